// PDV RoI-grid pooling, one set-abstraction branch in ONE kernel: group the ball's points, two point-wise layers, max over the ball.
//
// Reference: pointnet2_modules.py:31-158 (StackSAModuleMSGAttention.forward: QueryAndGroup -> shared MLP of Conv2d 1x1 + BN + ReLU ->
// max over nsample), pointnet2_utils.py:192-211 (grouping with use_xyz / use_density), kde_utils.py:17-64 (Gaussian KDE, bandwidth 0.25).
// The layer-by-layer path writes the grouped tensor (grid points x 16 samples x (C + 4) floats: 1 GB for the 128-channel source of a
// 487-RoI frame), reads it back through two GEMM launches and a max launch.  Here a WAVE owns a grid point: its 16 sample rows
// [dx, dy, dz, density, features] go global -> LDS once (16-byte copies, the rows are L2-resident voxel-centroid features), and both
// layers run on v_mfma_f32_16x16x4_f32 (exact fp32 products and accumulation, the arithmetic of the fp32 engine) with all weights
// resident in LDS:
//   layer 1  D1[channel x sample] = W1 . X^T: A = W1[channel][k], B = X[sample][k], both as 16-byte LDS reads covering 4 k-steps
//            (channel order of the k-steps permuted identically on both sides: k-step s of lane group g is channel g * CP/4 + s);
//   layer 2  D2[sample x channel] = H . W2: the accumulator of layer 1 - lane (g, r) holds channels 16mb + 4g + i of sample r -
//            IS the A operand of the k-step (mb, i): the hidden activations never leave their registers; BN + ReLU in place;
//   max over the 16 samples = 3 in-lane maxima + 2 cross-lane steps per 16 output channels.
#include "common.h"

namespace dz {
namespace {

typedef float f4_t __attribute__((ext_vector_type(4)));
constexpr int SA_NS = 16, SA_WAVES = 8, SA_THREADS = SA_WAVES * 64;

template <int CP, int H1, int H2>
struct SaCfg {
    static constexpr int RS = CP + 4;                       // LDS row stride of X and W1 (floats): conflict-free 16-byte reads
    static constexpr int R2 = H1 + 4;                       // row stride of W2[channel2][hidden]
    static constexpr int OFF_W1 = 0, OFF_W2 = OFF_W1 + H1 * RS, OFF_SB = OFF_W2 + H2 * R2, OFF_X = OFF_SB + 2 * H1 + 2 * H2;
    static constexpr int OFF_G = OFF_X + SA_WAVES * SA_NS * RS, FLOATS = OFF_G + SA_WAVES * SA_NS * 4;
    static constexpr int LDS = FLOATS * 4;
    static_assert(CP % 16 == 0 && H1 % 16 == 0 && H2 % 16 == 0 && LDS <= 160 * 1024, "shape");
};

struct SaArgs {
    const float *new_xyz, *xyz, *feats;
    const uint32_t *bitmap, *prefix;
    const int *idx, *cnt;
    const float *w1, *s1, *b1, *w2, *s2, *b2;
    float *out;
    int mq, per_batch, c, cells_per_batch, ldw1, ldw2;
};

template <int CP, int H1, int H2>
__global__ __launch_bounds__(SA_THREADS) void k_sa_pool(SaArgs a) {
    using C = SaCfg<CP, H1, H2>;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 15, g = lane >> 4;
    constexpr int Q = CP / 4, M1 = H1 / 16, M2 = H2 / 16;
    // weights, transposed to [output channel][input channel]; folded BatchNorm scale / shift
    for (int i = tid; i < H1 * CP; i += SA_THREADS) sm[C::OFF_W1 + (i % H1) * C::RS + i / H1] = a.w1[(size_t)(i / H1) * a.ldw1 + i % H1];
    for (int i = tid; i < H2 * H1; i += SA_THREADS) sm[C::OFF_W2 + (i % H2) * C::R2 + i / H2] = a.w2[(size_t)(i / H2) * a.ldw2 + i % H2];
    for (int i = tid; i < H1; i += SA_THREADS) { sm[C::OFF_SB + i] = a.s1[i]; sm[C::OFF_SB + H1 + i] = a.b1[i]; }
    for (int i = tid; i < H2; i += SA_THREADS) { sm[C::OFF_SB + 2 * H1 + i] = a.s2[i]; sm[C::OFF_SB + 2 * H1 + H2 + i] = a.b2[i]; }
    float *const X = sm + C::OFF_X + wid * SA_NS * C::RS;
    float *const gs = sm + C::OFF_G + wid * SA_NS * 4;
    // the padding columns of my sample rows never change
    const int npad = CP - 4 - a.c;
    for (int i = lane; i < SA_NS * npad; i += 64) X[(i / npad) * C::RS + 4 + a.c + i % npad] = 0.f;
    __syncthreads();

    const int c4 = a.c >> 2;
    const float bw = 0.25f;
    for (int q = blockIdx.x * SA_WAVES + wid; q < a.mq; q += gridDim.x * SA_WAVES) {
        const int b = q / a.per_batch;
        const int batch_start = bitmap_rank(a.bitmap, a.prefix, (uint32_t)b * (uint32_t)a.cells_per_batch);
        const int cnt = a.cnt[q];
        const bool empty = cnt == 0;
        // ---- group: offsets, KDE density, features -> X (16 rows)
        int v = 0;
        if (lane < SA_NS) {
            v = batch_start + a.idx[(size_t)q * SA_NS + lane];
#pragma unroll
            for (int d = 0; d < 3; ++d) gs[lane * 4 + d] = empty ? 0.f : __fsub_rn(a.xyz[(size_t)v * 3 + d], a.new_xyz[(size_t)q * 3 + d]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (lane < SA_NS) {
            float dens = 0.f;
            if (!empty) {
                float acc = 0.f;
                for (int s = 0; s < cnt; ++s) {
                    float lp = 0.f;
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        const float u = __fdiv_rn(__fsub_rn(gs[lane * 4 + d], gs[s * 4 + d]), bw);
                        lp += -(u * u) / 2.f - 0.91893853320467274178f;
                    }
                    acc += expf(lp);
                }
                dens = acc / (bw * bw * bw * (float)cnt);
            }
            *reinterpret_cast<f4_t *>(X + lane * C::RS) = f4_t{gs[lane * 4], gs[lane * 4 + 1], gs[lane * 4 + 2], dens};
        }
        for (int t = lane; t < SA_NS * c4; t += 64) {
            const int row = t / c4, f = t % c4;
            const int vr = __shfl(v, row, 64);
            f4_t val = f4_t{0.f, 0.f, 0.f, 0.f};
            if (!empty) val = *reinterpret_cast<const f4_t *>(a.feats + (size_t)vr * a.c + 4 * f);
            *reinterpret_cast<f4_t *>(X + row * C::RS + 4 + 4 * f) = val;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // ---- layer 1: D1[channel x sample]
        f4_t h[M1];
#pragma unroll
        for (int mb = 0; mb < M1; ++mb) h[mb] = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < Q / 4; ++j) {
            const f4_t xb = *reinterpret_cast<const f4_t *>(X + r * C::RS + g * Q + 4 * j);
#pragma unroll
            for (int mb = 0; mb < M1; ++mb) {
                const f4_t wa = *reinterpret_cast<const f4_t *>(sm + C::OFF_W1 + (mb * 16 + r) * C::RS + g * Q + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[e], xb[e], h[mb], 0, 0, 0);
            }
        }
        // lane (g, r): h[mb][i] = pre-activation of channel 16mb + 4g + i, sample r
#pragma unroll
        for (int mb = 0; mb < M1; ++mb) {
            const f4_t sc = *reinterpret_cast<const f4_t *>(sm + C::OFF_SB + mb * 16 + g * 4);
            const f4_t sh = *reinterpret_cast<const f4_t *>(sm + C::OFF_SB + H1 + mb * 16 + g * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) h[mb][i] = fmaxf(fmaf(h[mb][i], sc[i], sh[i]), 0.f);
        }
        // ---- layer 2: D2[sample x channel2], k-step (mb, i) <-> hidden channel 16mb + 4g + i
        float res[M2];
#pragma unroll
        for (int m2 = 0; m2 < M2; ++m2) {
            f4_t o = f4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mb = 0; mb < M1; ++mb) {
                const f4_t wb = *reinterpret_cast<const f4_t *>(sm + C::OFF_W2 + (m2 * 16 + r) * C::R2 + mb * 16 + g * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_16x16x4f32(h[mb][i], wb[i], o, 0, 0, 0);
            }
            // lane (g, c = r): o[i] = sample 4g + i, channel2 16 m2 + c
            const float sc = sm[C::OFF_SB + 2 * H1 + m2 * 16 + r], sh = sm[C::OFF_SB + 2 * H1 + H2 + m2 * 16 + r];
            float mx = fmaxf(fmaxf(fmaxf(fmaf(o[0], sc, sh), fmaf(o[1], sc, sh)), fmaxf(fmaf(o[2], sc, sh), fmaf(o[3], sc, sh))), 0.f);
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            res[m2] = mx;
        }
        if (g == 0) {
#pragma unroll
            for (int m2 = 0; m2 < M2; ++m2) a.out[(size_t)q * H2 + m2 * 16 + r] = res[m2];
        }
        // the next grid point overwrites X and gs: every read above has returned (their MFMAs were issued in order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

template <int CP, int H1, int H2>
int launch_sa(const SaArgs &a, hipStream_t stream) {
    using C = SaCfg<CP, H1, H2>;
    static PerDeviceFlags done;
    if (int rc = reserve_lds(reinterpret_cast<const void *>(&k_sa_pool<CP, H1, H2>), C::LDS, done, "dz_pdv_sa_pool")) return rc;
    int grid = device_cus();
    if ((long)grid * SA_WAVES > a.mq) grid = (a.mq + SA_WAVES - 1) / SA_WAVES;
    hipLaunchKernelGGL((k_sa_pool<CP, H1, H2>), dim3(grid), dim3(SA_THREADS), C::LDS, stream, a);
    return DZ_OK;
}

}  // namespace
}  // namespace dz

using namespace dz;

extern "C" {

int dz_pdv_sa_pool_supported(int c, int cin_pad, int h1, int h2, int nsample, int relu1, int relu2) {
    return nsample == SA_NS && relu1 && relu2 && c % 4 == 0 && c + 4 <= cin_pad &&
           ((cin_pad == 80 && h1 == 32 && h2 == 32) || (cin_pad == 144 && h1 == 64 && h2 == 64));
}

int dz_pdv_sa_pool(const float *new_xyz, int mq, int per_batch, const float *xyz, const float *feats, int c, const uint32_t *bitmap,
                   const uint32_t *prefix, int cells_per_batch, const int *idx, const int *cnt, int nsample, const float *w1, int ldw1,
                   const float *s1, const float *b1, int h1, const float *w2, int ldw2, const float *s2, const float *b2, int h2, int cin_pad,
                   float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dz_pdv_sa_pool_supported(c, cin_pad, h1, h2, nsample, 1, 1)) {
        set_error("dz_pdv_sa_pool: no instance for c %d, cin_pad %d, widths %d / %d, nsample %d", c, cin_pad, h1, h2, nsample);
        return DZ_ERR_UNSUPPORTED;
    }
    DZ_CHECK_ARG(mq >= 0 && per_batch >= 1 && ldw1 >= h1 && ldw2 >= h2, "dz_pdv_sa_pool: bad sizes");
    if (mq == 0) return DZ_OK;
    DZ_CHECK_ARG(new_xyz && xyz && feats && bitmap && prefix && idx && cnt && w1 && s1 && b1 && w2 && s2 && b2 && out, "dz_pdv_sa_pool: null pointer");
    const SaArgs a{new_xyz, xyz, feats, bitmap, prefix, idx, cnt, w1, s1, b1, w2, s2, b2, out, mq, per_batch, c, cells_per_batch, ldw1, ldw2};
    int rc = cin_pad == 80 ? launch_sa<80, 32, 32>(a, stream) : launch_sa<144, 64, 64>(a, stream);
    if (rc) return rc;
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
