// Micro-benchmark (development): what would a "hi.hi on fp16 + both correction terms in one block-scaled fp8 MFMA" product cost on the
// MI355X matrix pipe, against today's three fp16 MFMAs per product?  Per 32 input channels and 32 x 32 outputs:
//   pattern A (shipped f16x2): 6 x v_mfma_f32_32x32x16_f16                       (192 pipe cycles)
//   pattern B (proposed)     : 2 x v_mfma_f32_32x32x16_f16 + 1 x v_mfma_scale_f32_32x32x64_f8f6f4   (128 pipe cycles)
// Random operand bits (the pipe is power-limited on real data), 4 accumulators per wave, two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_f8mix.hip -o mfma_f8mix && ./mfma_f8mix
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ unsigned int rnd(unsigned int &x) { x = x * 1664525u + 1013904223u; return x; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    unsigned int x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    h8 ah[2], bh[2], al[2], bl[2];
    v8i a8, b8;
    for (int j = 0; j < 2; ++j)
        for (int i = 0; i < 8; ++i) {
            ah[j][i] = (_Float16)(((int)(rnd(x) >> 16) - 32768) * (1.f / 32768.f));
            bh[j][i] = (_Float16)(((int)(rnd(x) >> 16) - 32768) * (1.f / 32768.f));
            al[j][i] = (_Float16)(((int)(rnd(x) >> 16) - 32768) * (1.f / 67108864.f));
            bl[j][i] = (_Float16)(((int)(rnd(x) >> 16) - 32768) * (1.f / 67108864.f));
        }
    for (int i = 0; i < 8; ++i) {          // fp8 e4m3 bytes with random mantissas, exponents around 1: 0x30..0x3F and sign
        unsigned int w = 0;
        for (int b = 0; b < 4; ++b) w |= (0x30u | (rnd(x) >> 28) | ((rnd(x) >> 31) << 7)) << (8 * b);
        a8[i] = (int)w;
        w = 0;
        for (int b = 0; b < 4; ++b) w |= (0x30u | (rnd(x) >> 28) | ((rnd(x) >> 31) << 7)) << (8 * b);
        b8[i] = (int)w;
    }
    f16v acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {          // 4 x (32 channels of products into each of the 4 accumulators)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (MODE == 0) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ks], bh[ks], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bl[ks], acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ks], bh[ks], acc[j], 0, 0, 0);
                    }
                } else {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0], bh[0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1], bh[1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[j], 0, 0, 0, 116, 0, 127);
                }
            }
        }
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(int iters) {
    float *out; (void)hipMalloc(&out, 256 * 2 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * 2;
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // algorithmic work: per (r, j): 32 channels x 32 x 32 outputs x 2 flop
    const double flop = (double)grid * 4 * iters * 16 * 2.0 * 32 * 32 * 32;
    printf("%s: %9.1f us  %6.0f TF/s algorithmic (products of fp32-class pairs)\n",
           MODE == 0 ? "A  6 x f16 MFMA per 32 channels          " : "B  2 x f16 + 1 x scaled fp8 (K = 64) MFMA", ms * 1e3, flop / ms / 1e9);
    (void)hipFree(out);
}

int main() {
    run<0>(4000); run<1>(4000); run<0>(20000); run<1>(20000);
    return 0;
}
