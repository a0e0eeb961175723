// Micro-benchmark (development): what the matrix pipe of an MI355X delivers when nothing else is in the way.
//   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
// Variants: waves per SIMD (1, 2, 4), independent accumulators per wave (1, 2, 4, 8), with / without an LDS read per MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS, bool RAND = false>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    __shared__ h8 sm[1024];
    if (LDS) for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = h8{1, 2, 3, 4, 5, 6, 7, 8};
    __syncthreads();
    h8 a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 0, 1, 0, 1, 0, 1, 0};
    if (RAND) {        // operands with random bits: what the pipe draws (and the clock does) on real data instead of 1.0 / 0.0
        unsigned int x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
        for (int i = 0; i < 8; ++i) {
            x = x * 1664525u + 1013904223u; a[i] = (_Float16)(((int)(x >> 16) - 32768) * (1.f / 32768.f));
            x = x * 1664525u + 1013904223u; b[i] = (_Float16)(((int)(x >> 16) - 32768) * (1.f / 32768.f));
        }
    }
    f16v acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                if (LDS) a = sm[(threadIdx.x + 64 * (r * NACC + j + it)) & 1023];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS, bool RAND = false>
void run(int wg_per_cu, int iters) {
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    hipLaunchKernelGGL((k<NACC, LDS, RAND>), dim3(grid), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS, RAND>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop = (double)grid * 4 * iters * 8 * NACC * 32768.0;
    printf("acc %d lds %d rand %d waves/SIMD %d: %.1f us  %.0f TF/s (%.1f %% of 2516.6)\n", NACC, (int)LDS, (int)RAND, wg_per_cu, ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 25.166);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) { run<4, false>(w, 4000 / w); }
    run<1, false>(2, 4000); run<2, false>(2, 4000); run<8, false>(2, 1000);
    for (int w : {1, 2, 4}) { run<4, true>(w, 4000 / w); }
    // long run: does the clock hold?
    run<4, false>(2, 40000);
    // random operand bits (power): short and long
    run<4, false, true>(2, 4000); run<4, false, true>(2, 40000); run<4, false, true>(1, 40000);
    return 0;
}
