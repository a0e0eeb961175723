// PDV's encoder attention (one head of 192 channels over the 216 grid points of a RoI) on pair16 operands, with the key and value
// projections FOLDED AWAY: keys = values = the layer's input rows.
//
// Reference: detection/detzero_det/models/centerpoint_modules/pdv_utils/attention_utils.py:17-52 (nn.TransformerEncoderLayer: one
// head, post-norm), torch's MultiheadAttention arithmetic: q = x Wq + bq, k = x Wk + bk, v = x Wv + bv, softmax(q k^T / sqrt(E) + mask) v Wo + bo.
// With one head the projections fold into the two sides of the attention (detzero_amd/pdv_modules.py: PDVHead.attention):
//     scores  (x_i Wq + bq) . (x_j Wk + bk) = (x_i (Wq Wk^T) + Wk bq) . x_j + [terms without j: softmax-invariant]     -> q' = x Mq + uq
//     output  sum_j p_ij (x_j Wv + bv) Wo + bo = (sum_j p_ij x_j) (Wv Wo) + (bv Wo + bo)                                  -> o' = P x
// so this kernel computes o' = softmax(q' x^T) x from TWO row tensors (q' already in log2 units, x), and the layer needs two 192 x 192
// GEMMs (q', output) instead of four, reads x where it would read k and v, and never writes them.  (x: 841 k rows x 768 bytes per
// 8-frame pass of the two-stage detector; the unfolded fp32 path moved 3 more tensors of that size through HBM and ran
// k_attn1h_block at 1.8 ms.)
//
// One workgroup = one RoI, 7 waves x 32 queries (L <= 224).  The rows stream through LDS in 16 steps of the same shape - (channel half,
// 64 keys) = 64 half-rows of 384 bytes, double buffered, `buffer_load ... lds`, XOR-swizzled in 128-byte windows (conflict-free
// 16-byte reads down 16 consecutive rows) - twice: steps 0-7 for the scores, steps 8-15 for the weighted sum.  Splitting the 192
// channels in two halves keeps a wave inside 256 registers (2 waves per SIMD): 48 of query operands + 112 of scores, then 112 + 48 of
// output accumulators.
//   phase 1  S^T[key x query] += X_half . Q'_half^T: the whole score row of a query stays in registers: 7 accumulators, the lane (query,
//            half h) holding keys 8 (e / 4) + 4 h + e % 4 of each block of 32.
//   softmax  in-lane maximum / sum + one exchange with lane ^ 32; exp2 (the caller folds log2 e / sqrt(E) into q').
//   phase 2  O^T[channel x query] += X^T . P^T: each step's rows are transposed IN LDS (16-byte read, eight 2-byte writes) into [channel]
//            [key position] hi / lo planes, key positions permuted so that the 8 keys a lane's probability registers cover in a k-step
//            are contiguous: one 16-byte read per operand half.  The probabilities, split into (hi, lo), are the B operands as they sit
//            in the accumulators of phase 1 - nothing is moved.
//   output   1 / sum, split, one exchange with lane ^ 32 -> pair16 rows (32-byte stores), the operand of the output GEMM.
#include <stdlib.h>

#include "hgemm.h"

namespace dz {
namespace {

constexpr int SA_E = 192, SA_ROWB = SA_E * 4;                               // 768-byte rows
constexpr int SA_HB = SA_ROWB / 2, SA_HKS = 6, SA_HCF = 3;                   // a channel half: 384 bytes, 6 k-steps, 3 channel fragments
constexpr int SA_WAVES = 7, SA_THREADS = SA_WAVES * 64, SA_LMAX = SA_WAVES * 32, SA_NKB = SA_LMAX / 32;
constexpr int SA_STEP_KEYS = 64, SA_BLK = SA_STEP_KEYS * SA_HB;             // a step: 64 half-rows = 24 KB
constexpr int SA_VROW = 272, SA_VT = (SA_E / 2) * SA_VROW;                  // V^T of a step: [96 channels][128 B hi | 128 B lo | 16 pad]
constexpr int SA_OFF_ROWS = 0, SA_OFF_VT = 2 * SA_BLK, SA_OFF_MASK = SA_OFF_VT + SA_VT, SA_LDS = SA_OFF_MASK + 256;
static_assert(SA_LDS <= 160 * 1024 && SA_BLK % 1024 == 0, "LDS");

struct SelfAttnArgs {
    const float *q, *x;            // (r * l, 192) pair16: folded queries in log2 units, keys = values
    const uint8_t *kpm;            // (r, l) or null: 1 = key masked
    float *out;                    // (r * l, 192) pair16
    int l;
};

// slot of 16-byte piece `piece` (0..23) of half-row `row`: rows are 384 bytes, so rows r, r + 2, ... share their 128-byte bank window
__device__ __forceinline__ int swz(int piece, int row) { return (piece & ~7) | ((piece & 7) ^ ((row >> 1) & 7)); }

template <class M>
__global__ __launch_bounds__(SA_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_self_attn_h(SelfAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int l = a.l;
    const size_t row0 = (size_t)blockIdx.x * l;
    const unsigned char *xg = reinterpret_cast<const unsigned char *>(a.x) + row0 * SA_ROWB;
    const srsrc_t xrsrc = make_srsrc(xg, (unsigned int)(l * SA_ROWB));
    uint8_t *const mask_s = sm + SA_OFF_MASK;
    for (int i = tid; i < 256; i += SA_THREADS) mask_s[i] = (i >= l || (a.kpm && a.kpm[row0 + i])) ? 1 : 0;

    // step st (0..15; steps st and st + 8 fetch the same bytes): channel half (st >> 2) & 1 of keys 64 (st & 3) .. + 63 -> buffer st & 1,
    // 24 wave-instructions of 1 KB (rows past l: the last row again - those keys are masked)
    auto issue_step = [&](int st) {
        const int half = (st >> 2) & 1, blk = st & 3, buf = st & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int wi = j * SA_WAVES + wid;                              // LDS pieces [wi * 64, wi * 64 + 64)
            if (wi < SA_BLK / 1024) {
                const int sl = wi * 64 + lane, r = sl / 24, cs = sl % 24;
                const int key = min(blk * SA_STEP_KEYS + r, l - 1);
                const unsigned int voff = (unsigned int)(key * SA_ROWB + half * SA_HB + swz(cs, r) * 16);
                const unsigned int base = __builtin_amdgcn_readfirstlane((unsigned int)(SA_OFF_ROWS + buf * SA_BLK + wi * 1024));
                asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(base), "v"(voff), "s"(xrsrc) : "memory", "m0");
            }
        }
    };
    issue_step(0);

    const int qi = wid * 32 + l31;
    // ---- phase 1: the score row of my query
    f32x16 sc[SA_NKB];
#pragma unroll
    for (int kb = 0; kb < SA_NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[kb][e] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        // my query's operands of this half: k-step s = groups 2 s + h of the half-row (queries past l: the last row, never stored)
        v4u qh[SA_HKS], ql[SA_HKS];
        {
            const unsigned char *qp = reinterpret_cast<const unsigned char *>(a.q) + (row0 + min(qi, l - 1)) * SA_ROWB + half * SA_HB + h * 32;
#pragma unroll
            for (int s = 0; s < SA_HKS; ++s) {
                qh[s] = *reinterpret_cast<const v4u *>(qp + s * 64);
                ql[s] = *reinterpret_cast<const v4u *>(qp + s * 64 + 16);
            }
        }
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int st = half * 4 + blk;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                                // step st has landed; everyone is done with the other buffer
            issue_step(st + 1);
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const int kb = 2 * blk + sub;
                if (kb < SA_NKB) {
                    const int r = sub * 32 + l31;
                    const unsigned char *kr = sm + SA_OFF_ROWS + (st & 1) * SA_BLK + r * SA_HB;
#pragma unroll
                    for (int s = 0; s < SA_HKS; ++s) {
                        const int g = 2 * s + h;
                        const v4u khi = *reinterpret_cast<const v4u *>(kr + swz(2 * g, r) * 16);
                        const v4u klo = *reinterpret_cast<const v4u *>(kr + swz(2 * g + 1, r) * 16);
                        sc[kb] = M::mma(klo, qh[s], sc[kb]);
                        sc[kb] = M::mma(khi, ql[s], sc[kb]);
                        sc[kb] = M::mma(khi, qh[s], sc[kb]);
                    }
                }
            }
        }
    }
    // ---- softmax over the keys of my query (an all-masked row gives zeros: the caller never uses one)
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < SA_NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (mask_s[kb * 32 + 8 * (e >> 2) + 4 * h + (e & 3)]) sc[kb][e] = -INFINITY;
            mx = fmaxf(mx, sc[kb][e]);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (mx == -INFINITY) mx = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < SA_NKB; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sc[kb][e] = __builtin_amdgcn_exp2f(sc[kb][e] - mx);
            sum += sc[kb][e];
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = sum > 0.f ? 1.f / sum : 0.f;

    // ---- phase 2: O^T[channel x query], one channel half at a time
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        f32x16 o[SA_HCF];
#pragma unroll
        for (int ct = 0; ct < SA_HCF; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[ct][e] = 0.f;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int st = 8 + half * 4 + blk;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                                // rows of step st are in; everyone is done with V^T of the step before
            if (st + 1 < 16) issue_step(st + 1);
            // transpose: item = (row r = sl % 64, stored piece c = sl / 64); piece c = group c / 2 of the half-row, hi (even) or lo (odd)
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int sl = it * SA_THREADS + tid;
                if (sl < SA_STEP_KEYS * 24) {
                    const int r = sl & 63, c = sl >> 6;
                    const v4u v = *reinterpret_cast<const v4u *>(sm + SA_OFF_ROWS + (st & 1) * SA_BLK + r * SA_HB + swz(c, r) * 16);
                    const int rk = r & 15;
                    const int pos = (r & 48) + ((rk >> 2) & 1) * 8 + (rk >> 3) * 4 + (rk & 3);
                    unsigned char *dst = sm + SA_OFF_VT + (c >> 1) * 8 * SA_VROW + (c & 1) * 128 + pos * 2;
                    const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        *reinterpret_cast<unsigned short *>(dst + j * SA_VROW) = (unsigned short)(w[j >> 1] >> (16 * (j & 1)));
                }
            }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kb = 2 * blk + (t >> 1), e0 = 8 * (t & 1);
                if (kb < SA_NKB) {
                    // probabilities of this k-step: registers e0 .. e0 + 7 = keys 16 t + {0..3, 8..11} + 4 h of the step = positions 8 h .. 8 h + 7
                    uint2 h0, l0, h1, l1;
                    const float v0[4] = {sc[kb][e0], sc[kb][e0 + 1], sc[kb][e0 + 2], sc[kb][e0 + 3]};
                    const float v1[4] = {sc[kb][e0 + 4], sc[kb][e0 + 5], sc[kb][e0 + 6], sc[kb][e0 + 7]};
                    split4<M>(v0, h0, l0);
                    split4<M>(v1, h1, l1);
                    const v4u ph = v4u{h0.x, h0.y, h1.x, h1.y}, pl = v4u{l0.x, l0.y, l1.x, l1.y};
#pragma unroll
                    for (int ct = 0; ct < SA_HCF; ++ct) {
                        const unsigned char *vp = sm + SA_OFF_VT + (ct * 32 + l31) * SA_VROW + t * 32 + h * 16;
                        const v4u vhi = *reinterpret_cast<const v4u *>(vp);
                        const v4u vlo = *reinterpret_cast<const v4u *>(vp + 128);
                        o[ct] = M::mma(vlo, ph, o[ct]);
                        o[ct] = M::mma(vhi, pl, o[ct]);
                        o[ct] = M::mma(vhi, ph, o[ct]);
                    }
                }
            }
        }
        // ---- output rows of this half: lane (query, h) holds channels 32 ct + 8 q + 4 h + {0..3}; groups with (q & 1) == h stay, the others
        // are swapped with lane ^ 32
        unsigned char *op = reinterpret_cast<unsigned char *>(a.out) + (row0 + min(qi, l - 1)) * SA_ROWB + half * SA_HB + h * 32;
#pragma unroll
        for (int ct = 0; ct < SA_HCF; ++ct) {
            uint2 ghi[4], glo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float v[4] = {o[ct][4 * q] * inv, o[ct][4 * q + 1] * inv, o[ct][4 * q + 2] * inv, o[ct][4 * q + 3] * inv};
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
                const v4u oh = h ? v4u{recv_hi.x, recv_hi.y, keep_hi.x, keep_hi.y} : v4u{keep_hi.x, keep_hi.y, recv_hi.x, recv_hi.y};
                const v4u ol = h ? v4u{recv_lo.x, recv_lo.y, keep_lo.x, keep_lo.y} : v4u{keep_lo.x, keep_lo.y, recv_lo.x, recv_lo.y};
                if (qi < l) {                                               // group 2 (2 ct + sl) + h of the half-row
                    *reinterpret_cast<v4u *>(op + (2 * ct + sl) * 64) = oh;
                    *reinterpret_cast<v4u *>(op + (2 * ct + sl) * 64 + 16) = ol;
                }
            }
        }
    }
}

}  // namespace
}  // namespace dz

using namespace dz;

extern "C" {

int dz_self_attention_split_supported(int l, int e) { return e == SA_E && l >= 1 && l <= SA_LMAX; }

// o' = softmax(q' x^T + mask) x per group of l consecutive rows (one head, e = 192, l <= 224): q' (r * l, e) pair16 in log2 units
// (scores are used as exponents of 2), x (r * l, e) pair16 = keys = values, key_padding_mask (r, l) bytes or null, out (r * l, e) pair16.
int dz_self_attention_split(const float *q, const float *x, const unsigned char *key_padding_mask, int r, int l, int e, float *out,
                            int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(r >= 0 && dz_self_attention_split_supported(l, e), "dz_self_attention_split: one head of 192 channels over 1..224 rows (got l %d, e %d)", l, e);
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_self_attention_split: math %d is not a split mode", math);
    if (r == 0) return DZ_OK;
    DZ_CHECK_ARG(q && x && out, "dz_self_attention_split: null pointer");
    const SelfAttnArgs a{q, x, key_padding_mask, out, l};
    int rc;
    if (math == DZ_MATH_F16X2) {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_self_attn_h<MathF16>), SA_LDS, done, "dz_self_attention_split"))) return rc;
        hipLaunchKernelGGL(k_self_attn_h<MathF16>, dim3(r), dim3(SA_THREADS), SA_LDS, stream, a);
    } else {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_self_attn_h<MathBF16>), SA_LDS, done, "dz_self_attention_split"))) return rc;
        hipLaunchKernelGGL(k_self_attn_h<MathBF16>, dim3(r), dim3(SA_THREADS), SA_LDS, stream, a);
    }
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
