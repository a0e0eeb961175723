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
#include "hgemm.h"

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

// ------------------------------------------------------------------------------------------------------------------------------
// The same branch on pair16 operands (the head's split math modes): three v_mfma_f32_32x32x16 per product instead of fp32 MFMAs
// at an eighth of the rate.  A WAVE owns TWO grid points = 32 sample rows, and nothing of a row touches LDS:
//   rows     lane (row r = lane & 31, half h) gathers the 8 channels 16 s + 8 h .. + 7 of ITS sample row for every k-step s straight
//            from the centroid features (2 x 16 bytes per k-step; channel 0..3 = offset + density, computed in the lane), splits them
//            into (hi, lo): that IS its B operand of layer 1.  The loads of the NEXT tile are issued before this tile's matrix work.
//   density  the 16 offsets of a ball live in the 16 lanes of its half-row group: the Gaussian kernel sum runs over ds_bpermute reads,
//            each half-wave taking every other sample (two partial sums, one exchange).
//   layer 1  D[channel x row] = W1 . X^T, BatchNorm + ReLU + split on the accumulator, one exchange with lane ^ 32 completes the
//            8-channel groups: the lane's operand of layer 2 (pointnet.hip's chaining);
//   layer 2  transposed, D^T[row x channel]: rows 0..15 (grid point A) are accumulator registers 0..7 of both half-waves, rows 16..31
//            (B) registers 8..15: the max over a ball = 7 in-lane maxima + one exchange; half-wave h stores grid point A / B.
// Weights (pair16, 55 KB for 144 -> 64 -> 64) stay in LDS for the whole launch; persistent waves, no workgroup barrier in the loop.
template <int CP, int H>
struct SaHCfg {
    static constexpr int KS1 = CP / 16, KS2 = H / 16, NF = H / 32, C = CP - 16;       // k-steps of the layers, 32-channel fragments, feature channels
    static constexpr int RS1 = CP * 4 + 16, RS2 = H * 4 + 16;                         // LDS row strides (bytes): 16-byte reads of 16 consecutive rows hit 16 different slots
    static constexpr int OFF_W1 = 0, OFF_W2 = OFF_W1 + H * RS1, OFF_SB = OFF_W2 + H * RS2, LDS = OFF_SB + 4 * H * 4;
    static_assert(CP % 16 == 0 && H % 32 == 0 && (RS1 / 16) % 2 == 1 && (RS2 / 16) % 2 == 1 && LDS <= 160 * 1024, "shape");
};

struct SaHArgs {
    const float *new_xyz, *xyz, *feats;
    const uint32_t *bitmap, *prefix;
    const int *idx, *cnt;
    const float *w1, *s1, *b1, *w2, *s2, *b2;        // w1 (H, ldw1) / w2 (H, ldw2) pair16 rows
    float *out;
    int mq, per_batch, cells_per_batch, ldw1, ldw2, ldo;
    unsigned int feat_bytes;
};

template <int CP, int H, class M>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sa_pool_h(SaHArgs a) {
    using C = SaHCfg<CP, H>;
    constexpr int KS1 = C::KS1, KS2 = C::KS2, NF = C::NF, CF = C::C;
    extern __shared__ __attribute__((aligned(16))) unsigned char smh[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, l31 = lane & 31, h = lane >> 5;
    for (int i = tid; i < H * (CP / 4); i += 512) {                          // 16-byte pieces of W1's rows
        const int n = i / (CP / 4), pc = i % (CP / 4);
        *reinterpret_cast<v4u *>(smh + C::OFF_W1 + n * C::RS1 + pc * 16) = *reinterpret_cast<const v4u *>(a.w1 + (size_t)n * a.ldw1 + pc * 4);
    }
    for (int i = tid; i < H * (H / 4); i += 512) {
        const int n = i / (H / 4), pc = i % (H / 4);
        *reinterpret_cast<v4u *>(smh + C::OFF_W2 + n * C::RS2 + pc * 16) = *reinterpret_cast<const v4u *>(a.w2 + (size_t)n * a.ldw2 + pc * 4);
    }
    float *const sb = reinterpret_cast<float *>(smh + C::OFF_SB);             // s1 | b1 | s2 | b2
    for (int i = tid; i < H; i += 512) { sb[i] = a.s1[i]; sb[H + i] = a.b1[i]; sb[2 * H + i] = a.s2[i]; sb[3 * H + i] = a.b2[i]; }
    __syncthreads();
    const __amdgpu_buffer_rsrc_t frsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.feats), 0, a.feat_bytes, 0x00020000);
    const int ntiles = (a.mq + 1) >> 1;
    const int stride = gridDim.x * 8;

    // what a tile's gather needs before its row loads can go out
    struct Head { int v; int cnt; float nx, ny, nz; };
    auto head = [&](int t) {
        Head r{0, 0, 0.f, 0.f, 0.f};
        const int q = 2 * t + (l31 >> 4);
        if (t < ntiles && q < a.mq) {
            const int b = q / a.per_batch;
            r.cnt = a.cnt[q];
            r.v = bitmap_rank(a.bitmap, a.prefix, (uint32_t)b * (uint32_t)a.cells_per_batch) + a.idx[(size_t)q * SA_NS + (l31 & 15)];
            r.nx = a.new_xyz[(size_t)q * 3]; r.ny = a.new_xyz[(size_t)q * 3 + 1]; r.nz = a.new_xyz[(size_t)q * 3 + 2];
        }
        return r;
    };
    // the row's pieces: piece p = 2 s + k = feature channels 16 s + 8 h - 4 + 4 k .. + 3 (an empty ball, the offset / density slot and
    // the channels past the features read zeros through an out-of-range offset)
    struct Rows { f32x4v f[2 * KS1]; float px, py, pz; };
    auto gather = [&](const Head &hd, Rows &r) {
        const bool ok = hd.cnt > 0;
        r.px = r.py = r.pz = 0.f;                                            // (first: the density needs them before the features)
        if (ok) { r.px = a.xyz[(size_t)hd.v * 3]; r.py = a.xyz[(size_t)hd.v * 3 + 1]; r.pz = a.xyz[(size_t)hd.v * 3 + 2]; }
        const unsigned int rb = ok ? (unsigned int)hd.v * (unsigned int)(CF * 4) : OOB_OFFSET;
#pragma unroll
        for (int p = 0; p < 2 * KS1; ++p) {
            const int fo = 8 * p + 8 * h - 4 - 4 * (p & 1);                   // 16 (p / 2) + 8 h - 4 + 4 (p & 1)
            const unsigned int off = (ok && fo >= 0 && fo < CF) ? rb + (unsigned int)(fo * 4) : OOB_OFFSET;
            r.f[p] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(frsrc, off, 0, 0));
        }
    };
    auto wfrag = [&](int base, int rs, int r0, int s, v4u &hi, v4u &lo) {
        const unsigned char *p = smh + base + (r0 + l31) * rs + (2 * s + h) * 32;
        hi = *reinterpret_cast<const v4u *>(p);
        lo = *reinterpret_cast<const v4u *>(p + 16);
    };

    // The first iteration is a VIRTUAL tile of two empty balls: its result (the layer stack on zero rows) is what every empty ball gets,
    // and a tile whose two balls are both empty afterwards costs only its stores (RoIs over free space: most of their grid points).
    const int t0 = blockIdx.x * 8 + wid;
    Head hd{0, 0, 0.f, 0.f, 0.f}, hd_next = head(t0);
    Rows rows;
    gather(hd, rows);
    float res_empty[NF];
#pragma unroll
    for (int cb = 0; cb < NF; ++cb) res_empty[cb] = 0.f;
    bool virt = true;
    for (int t = t0 - stride; t < ntiles; t += stride) {
        const int q_mine = 2 * t + h;
        if (!virt && __builtin_amdgcn_ballot_w64(hd.cnt > 0) == 0ull) {      // (wave-uniform)
#pragma unroll
            for (int cb = 0; cb < NF; ++cb)
                if (q_mine < a.mq) a.out[(size_t)q_mine * a.ldo + cb * 32 + l31] = res_empty[cb];
            hd = hd_next;
            gather(hd, rows);
            hd_next = head(t + 2 * stride);
            continue;
        }
        // ---- offsets and the kernel density estimate of my sample (kde_utils.py:17-64: Gaussian, bandwidth 0.25, over the ball's cnt samples)
        const bool ok = hd.cnt > 0;
        const float gx = ok ? __fsub_rn(rows.px, hd.nx) : 0.f, gy = ok ? __fsub_rn(rows.py, hd.ny) : 0.f, gz = ok ? __fsub_rn(rows.pz, hd.nz) : 0.f;
        float dens = 0.f;
        {
            const float bw = 0.25f;
            float acc = 0.f;
            const int base = lane & 48;                                      // first lane of my ball in my half-wave
#pragma unroll 2
            for (int s2 = 0; s2 < SA_NS / 2; ++s2) {
                const int sidx = 2 * s2 + h;
                const float ox = __shfl(gx, base + sidx, 64), oy = __shfl(gy, base + sidx, 64), oz = __shfl(gz, base + sidx, 64);
                const float ux = __fdiv_rn(__fsub_rn(gx, ox), bw), uy = __fdiv_rn(__fsub_rn(gy, oy), bw), uz = __fdiv_rn(__fsub_rn(gz, oz), bw);
                float lp = 0.f;
                lp += -(ux * ux) / 2.f - 0.91893853320467274178f;
                lp += -(uy * uy) / 2.f - 0.91893853320467274178f;
                lp += -(uz * uz) / 2.f - 0.91893853320467274178f;
                acc += sidx < hd.cnt ? expf(lp) : 0.f;
            }
            acc += __shfl_xor(acc, 32, 64);
            if (ok) dens = acc / (bw * bw * bw * (float)hd.cnt);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- layer 1: D[channel x row]; the operand of k-step s = my row's pieces 2 s, 2 s + 1, split as they are consumed
        f32x16 acc[NF];
#pragma unroll
        for (int ct = 0; ct < NF; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
        v4u wh[NF], wl[NF];
#pragma unroll
        for (int ct = 0; ct < NF; ++ct) wfrag(C::OFF_W1, C::RS1, ct * 32, 0, wh[ct], wl[ct]);
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            f32x4v p0 = rows.f[2 * s], p1 = rows.f[2 * s + 1];
            if (s == 0 && h == 0) p0 = f32x4v{gx, gy, gz, dens};
            const float v0[4] = {p0[0], p0[1], p0[2], p0[3]}, v1[4] = {p1[0], p1[1], p1[2], p1[3]};
            uint2 h0, l0, h1, l1;
            split4<M>(v0, h0, l0);
            split4<M>(v1, h1, l1);
            const v4u xh = v4u{h0.x, h0.y, h1.x, h1.y}, xl = v4u{l0.x, l0.y, l1.x, l1.y};
            v4u nh[NF], nl[NF];                                              // the next k-step's weight fragments: read under this one's MFMAs
            if (s + 1 < KS1) {
#pragma unroll
                for (int ct = 0; ct < NF; ++ct) wfrag(C::OFF_W1, C::RS1, ct * 32, s + 1, nh[ct], nl[ct]);
            }
#pragma unroll
            for (int ct = 0; ct < NF; ++ct) {
                acc[ct] = M::mma(wl[ct], xh, acc[ct]);
                acc[ct] = M::mma(wh[ct], xl, acc[ct]);
                acc[ct] = M::mma(wh[ct], xh, acc[ct]);
            }
            if (s + 1 < KS1) {
#pragma unroll
                for (int ct = 0; ct < NF; ++ct) { wh[ct] = nh[ct]; wl[ct] = nl[ct]; }
            }
            __builtin_amdgcn_sched_barrier(0);                               // (keeps the splits and weight reads of later k-steps from piling up in registers)
        }
        // ---- the next tile's rows go out now (they land under the rest of this tile and the other wave of the SIMD); the head of the one after
        hd = hd_next;
        gather(hd, rows);
        hd_next = virt ? head(t0 + stride) : head(t + 2 * stride);
        __builtin_amdgcn_sched_barrier(0);
        // BatchNorm + ReLU + split; lane (row, h) holds channels 32 ct + 8 q + 4 h + {0..3}: the groups with (q & 1) == h stay, the others
        // are swapped with lane ^ 32 (pointnet.hip)
        v4u hh[KS2], hl[KS2];
#pragma unroll
        for (int ct = 0; ct < NF; ++ct) {
            uint2 ghi[4], glo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = ct * 32 + q * 8 + h * 4;
                const float4 s4 = *reinterpret_cast<const float4 *>(sb + c0), b4 = *reinterpret_cast<const float4 *>(sb + H + c0);
                const float v[4] = {fmaxf(fmaf(acc[ct][4 * q], s4.x, b4.x), 0.f), fmaxf(fmaf(acc[ct][4 * q + 1], s4.y, b4.y), 0.f),
                                    fmaxf(fmaf(acc[ct][4 * q + 2], s4.z, b4.z), 0.f), fmaxf(fmaf(acc[ct][4 * q + 3], s4.w, b4.w), 0.f)};
                split4<M>(v, ghi[q], glo[q]);
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const uint2 keep_hi = h ? ghi[2 * sl + 1] : ghi[2 * sl], keep_lo = h ? glo[2 * sl + 1] : glo[2 * sl];
                const uint2 send_hi = h ? ghi[2 * sl] : ghi[2 * sl + 1], send_lo = h ? glo[2 * sl] : glo[2 * sl + 1];
                uint2 recv_hi, recv_lo;
                recv_hi.x = (unsigned int)__shfl_xor((int)send_hi.x, 32, 64);
                recv_hi.y = (unsigned int)__shfl_xor((int)send_hi.y, 32, 64);
                recv_lo.x = (unsigned int)__shfl_xor((int)send_lo.x, 32, 64);
                recv_lo.y = (unsigned int)__shfl_xor((int)send_lo.y, 32, 64);
                const int s = 2 * ct + sl;
                hh[s] = h ? v4u{recv_hi.x, recv_hi.y, keep_hi.x, keep_hi.y} : v4u{keep_hi.x, keep_hi.y, recv_hi.x, recv_hi.y};
                hl[s] = h ? v4u{recv_lo.x, recv_lo.y, keep_lo.x, keep_lo.y} : v4u{keep_lo.x, keep_lo.y, recv_lo.x, recv_lo.y};
            }
        }
        // ---- layer 2, transposed: lane = channel cb * 32 + l31, rows 8 (e >> 2) + 4 h + (e & 3); max over each ball's 16 rows
#pragma unroll
        for (int cb = 0; cb < NF; ++cb) {
            f32x16 d;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = 0.f;
#pragma unroll
            for (int s = 0; s < KS2; ++s) {
                v4u whi, wlo;
                wfrag(C::OFF_W2, C::RS2, cb * 32, s, whi, wlo);
                d = M::mma(hl[s], whi, d);
                d = M::mma(hh[s], wlo, d);
                d = M::mma(hh[s], whi, d);
            }
            const float sc = sb[2 * H + cb * 32 + l31], sh = sb[3 * H + cb * 32 + l31];
            float ma = 0.f, mb = 0.f;                                       // (ReLU: the maximum is at least 0)
#pragma unroll
            for (int e = 0; e < 8; ++e) { ma = fmaxf(ma, fmaf(d[e], sc, sh)); mb = fmaxf(mb, fmaf(d[8 + e], sc, sh)); }
            ma = fmaxf(ma, __shfl_xor(ma, 32, 64));
            mb = fmaxf(mb, __shfl_xor(mb, 32, 64));
            if (virt) res_empty[cb] = ma;
            else if (q_mine < a.mq) a.out[(size_t)q_mine * a.ldo + cb * 32 + l31] = h ? mb : ma;
        }
        virt = false;
    }
}

template <int CP, int H, class M>
int launch_sa_h(const SaHArgs &a, hipStream_t stream) {
    using C = SaHCfg<CP, H>;
    static PerDeviceFlags done;
    if (int rc = reserve_lds(reinterpret_cast<const void *>(&k_sa_pool_h<CP, H, M>), C::LDS, done, "dz_pdv_sa_pool_split")) return rc;
    const int ntiles = (a.mq + 1) / 2;
    int grid = device_cus();
    if ((long)grid * 8 > ntiles) grid = (ntiles + 7) / 8;
    hipLaunchKernelGGL((k_sa_pool_h<CP, H, M>), dim3(grid), dim3(512), C::LDS, stream, a);
    return DZ_OK;
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

int dz_pdv_sa_pool_split_supported(int c, int cin_pad, int h1, int h2, int nsample, int relu1, int relu2) {
    return nsample == SA_NS && relu1 && relu2 && ((cin_pad == 80 && c == 64 && h1 == 32 && h2 == 32) || (cin_pad == 144 && c == 128 && h1 == 64 && h2 == 64));
}

// dz_pdv_sa_pool on pair16 operands (math = DZ_MATH_F16X2 / DZ_MATH_BF16X2): w1 (h1, ldw1) and w2 (h2, ldw2) are pair16 rows per OUTPUT
// channel (dz_pair16 packing, ld in channels), everything else as dz_pdv_sa_pool.
int dz_pdv_sa_pool_split(const float *new_xyz, int mq, int per_batch, const float *xyz, const float *feats, long feat_rows, int c,
                         const uint32_t *bitmap, const uint32_t *prefix, int cells_per_batch, const int *idx, const int *cnt, int nsample,
                         const float *w1, int ldw1, const float *s1, const float *b1, int h1, const float *w2, int ldw2, const float *s2,
                         const float *b2, int h2, int cin_pad, int math, float *out, int ldo, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (!dz_pdv_sa_pool_split_supported(c, cin_pad, h1, h2, nsample, 1, 1)) {
        set_error("dz_pdv_sa_pool_split: no instance for c %d, cin_pad %d, widths %d / %d, nsample %d", c, cin_pad, h1, h2, nsample);
        return DZ_ERR_UNSUPPORTED;
    }
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_pdv_sa_pool_split: math %d is not a split mode", math);
    DZ_CHECK_ARG(mq >= 0 && per_batch >= 1 && ldw1 >= cin_pad && ldw2 >= h1 && ldw1 % 4 == 0 && ldw2 % 4 == 0 && feat_rows >= 0 && ldo >= h2, "dz_pdv_sa_pool_split: bad sizes");
    if (mq == 0) return DZ_OK;
    DZ_CHECK_ARG(new_xyz && xyz && feats && bitmap && prefix && idx && cnt && w1 && s1 && b1 && w2 && s2 && b2 && out, "dz_pdv_sa_pool_split: null pointer");
    const size_t fb = (size_t)feat_rows * c * 4;
    if (fb >= 0x80000000ull) { set_error("dz_pdv_sa_pool_split: features of %zu bytes exceed the 2 GiB buffer-addressing limit", fb); return DZ_ERR_UNSUPPORTED; }
    const SaHArgs a{new_xyz, xyz, feats, bitmap, prefix, idx, cnt, w1, s1, b1, w2, s2, b2, out, mq, per_batch, cells_per_batch, ldw1, ldw2, ldo, (unsigned int)fb};
    int rc;
    if (cin_pad == 80) rc = math == DZ_MATH_F16X2 ? launch_sa_h<80, 32, MathF16>(a, stream) : launch_sa_h<80, 32, MathBF16>(a, stream);
    else rc = math == DZ_MATH_F16X2 ? launch_sa_h<144, 64, MathF16>(a, stream) : launch_sa_h<144, 64, MathBF16>(a, stream);
    if (rc) return rc;
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
