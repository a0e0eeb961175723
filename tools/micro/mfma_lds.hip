// Micro-benchmark (development): matrix pipe fed from LDS with random operand bits - what a (PT x CT) wave tile of the split-precision
// convolution kernels can reach when only the fragment reads compete with the MFMAs (no global loads, no barriers, no epilogue).
//   hipcc --offload-arch=gfx950 -O3 mfma_lds.hip -o mfma_lds && ./mfma_lds
// A k-step reads (PT + CT) x 2 16-byte fragments per lane and issues 3 x PT x CT MFMAs (hi.hi, hi.lo, lo.hi).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int PT, int CT, int WAVES, int WPE>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k(float *out, int iters, int zero) {
    extern __shared__ u4 sm[];                      // 4096 x 16 bytes
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
        unsigned int x = i * 2654435761u + 12345u;
        u4 v;
        for (int e = 0; e < 4; ++e) {
            x = x * 1664525u + 1013904223u;
            // two fp16 values in (-1, 1) with random mantissas
            const unsigned int a = (x >> 3) & 0x3FFu, b = (x >> 14) & 0x3FFu;
            v[e] = zero ? 0u : ((0x3800u | a | ((x & 1u) << 15)) | ((0x3800u | b | ((x & 2u) << 14)) << 16));
        }
        sm[i] = v;
    }
    __syncthreads();
    f16v acc[CT][PT];
    for (int c = 0; c < CT; ++c) for (int p = 0; p < PT; ++p) for (int e = 0; e < 16; ++e) acc[c][p][e] = 0.f;
    const int lane = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        u4 ph[PT], pl[PT], ch[CT], cl[CT];
        const int base = (lane * 9 + it * 37) & 4095;
#pragma unroll
        for (int p = 0; p < PT; ++p) { ph[p] = sm[(base + p * 290) & 4095]; pl[p] = sm[(base + p * 290 + 1) & 4095]; }
#pragma unroll
        for (int c = 0; c < CT; ++c) { ch[c] = sm[(base + 2048 + c * 290) & 4095]; cl[c] = sm[(base + 2048 + c * 290 + 1) & 4095]; }
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int p = 0; p < PT; ++p)
                    acc[c][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, term == 0 ? cl[c] : ch[c]),
                                                                       __builtin_bit_cast(h8, term == 1 ? pl[p] : ph[p]), acc[c][p], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CT; ++c) for (int p = 0; p < PT; ++p) for (int e = 0; e < 16; ++e) s += acc[c][p][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int PT, int CT, int WAVES, int WPE>
void run(int iters, int zero) {
    float *out; (void)hipMalloc(&out, 256 * 8 * 512 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * (4 * WPE / WAVES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k<PT, CT, WAVES, WPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<PT, CT, WAVES, WPE>), dim3(grid), dim3(64 * WAVES), 65536, 0, out, 10, zero);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PT, CT, WAVES, WPE>), dim3(grid), dim3(64 * WAVES), 65536, 0, out, iters, zero);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * WAVES * iters * 3.0 * PT * CT * 32768.0;
    printf("wave tile %dx%d  %d waves/SIMD  reads/MFMA %.2f  data %s: %8.1f us  %6.0f TF/s raw (%.1f %% of 2516.6) = %.0f TF/s algorithmic\n", PT, CT, WPE,
           2.0 * (PT + CT) / (3.0 * PT * CT), zero ? "zero" : "rand", ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 25.166, flop / ms / 1e9 / 3);
    (void)hipFree(out);
}

int main() {
    for (int zero = 0; zero < 2; ++zero) {
        run<2, 2, 8, 2>(20000, zero);      // today's conv3x3 wave tile
        run<2, 1, 8, 2>(40000, zero);
        run<1, 2, 8, 2>(40000, zero);      // 8-wave sparse tile
        run<2, 4, 8, 2>(10000, zero);
        run<3, 4, 4, 1>(8000, zero);
        run<4, 4, 4, 1>(6000, zero);
        run<2, 2, 4, 1>(20000, zero);
    }
    return 0;
}
