// Multi-head attention core on the 16-bit matrix cores with split-precision operands (the refiner's split math modes), gfx950.
//
// Reference: refining/detzero_refine/models/modules/transformer/multi_head_attention.py:207-288 (scores, key padding mask, softmax,
// weighted sum); the fp32 kernels of mha.hip are the exact-fp32 engine.  PRM's cross-attention (200 queries x 9600 keys x 8 heads x 96
// tracks per chunk) spends 2.5 ms in k_mha_block at 0.48 of the fp32-MFMA peak; with q, k, p and v carried as (hi, lo) 16-bit pairs
// (three v_mfma_f32_32x32x16 per product, fp32 accumulation, the arithmetic of dz_linear_forward_split) the matrix work shrinks 5x and
// the kernel is bound by the softmax's exponentials instead.
//
// A workgroup = one (batch, head), up to 8 waves of 32 queries; keys in blocks of 64 staged once per workgroup (double buffered):
//   K block  -> LDS as pair16 rows [key][4 groups of (16 B hi | 16 B lo)]: the A operand of S^T = K . Q^T is one 16-byte read per half;
//   V block  -> LDS transposed and split, [channel][key position] hi plane / lo plane, the key positions permuted so that the 8 keys
//               a lane contributes to a k-step of the second product are contiguous: A operand of O^T += V^T . P^T = one 16-byte read;
//   S^T tile (32 keys x 32 queries): lane (query, half h) holds keys 8 (i / 4) + 4 h + i % 4 of ITS query -> the softmax is in-lane
//               plus one exchange with lane ^ 32, and the probabilities of registers 8 t .. 8 t + 7, split into (hi, lo), ARE the lane's
//               B operand of k-step t of the second product: nothing is transposed or moved.
// Online softmax in units of log2 e (v_exp_f32), running maximum / sum per query, rescale only when a maximum moved.
#include "hgemm.h"

namespace dz {
namespace {

constexpr int AH_KEYS = 64, AH_ROW = 144;                  // keys per block; LDS row: 128 bytes + 16 (conflict-free 16-byte reads)
constexpr int AH_KT = AH_KEYS * AH_ROW, AH_VP = 32 * AH_ROW;                  // K tile; one plane (hi or lo) of the V^T tile
constexpr int AH_BUF = AH_KT + 2 * AH_VP + AH_KEYS, AH_LDS = 2 * AH_BUF;

template <class M>
__device__ __forceinline__ void split8(const float (&v)[8], v4u &hi, v4u &lo) {
    unsigned int a[4], c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) split2<M>(v[2 * j], v[2 * j + 1], a[j], c[j]);
    hi = v4u{a[0], a[1], a[2], a[3]};
    lo = v4u{c[0], c[1], c[2], c[3]};
}

template <class M, bool MASK>
__global__ __launch_bounds__(512) void k_mha_block_h(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                     const uint8_t *__restrict__ kpm, int lq, int lk, int heads, float scale,
                                                     float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[AH_LDS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int hd = blockIdx.y, b = blockIdx.z;
    const int e_dim = heads * 32;
    const int qi = (blockIdx.x * (blockDim.x >> 6) + wid) * 32 + l31;
    // B operand of S^T: Q[query l31][d = 16 s + 8 h .. + 7], scaled to log2 units, as (hi, lo)
    v4u qh[2], ql[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (qi < lq) {
            const float4 a = *reinterpret_cast<const float4 *>(q + ((size_t)b * lq + qi) * e_dim + hd * 32 + s * 16 + h * 8);
            const float4 c = *reinterpret_cast<const float4 *>(q + ((size_t)b * lq + qi) * e_dim + hd * 32 + s * 16 + h * 8 + 4);
            const float sc2 = scale * 1.44269504088896340736f;
            x[0] = a.x * sc2; x[1] = a.y * sc2; x[2] = a.z * sc2; x[3] = a.w * sc2;
            x[4] = c.x * sc2; x[5] = c.y * sc2; x[6] = c.z * sc2; x[7] = c.w * sc2;
        }
        split8<M>(x, qh[s], ql[s]);
    }
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const float *kb = k + (size_t)b * lk * e_dim + hd * 32, *vb = v + (size_t)b * lk * e_dim + hd * 32;
    const uint8_t *mb = MASK ? kpm + (size_t)b * lk : nullptr;
    // staging: thread (key = tid / 4, 8-channel group tid % 4) of the first 256 threads
    const bool stager = tid < 256;
    const int skey = tid >> 2, sg = tid & 3;
    // position of a key inside its 32-key tile in the V^T planes: key = 16 t + 8 g + 4 hh + r -> 16 t + 8 hh + 4 g + r
    const int spos = (skey & 32) | (skey & 16) | ((skey & 4) << 1) | ((skey & 8) >> 1) | (skey & 3);
    float4 kst[2], vst[2];
    unsigned char mst = 1;
    auto fetch = [&](int key0) {
        if (!stager) return;
        const int key = key0 + skey;
        if (key < lk) {
            const float *kp = kb + (size_t)key * e_dim + sg * 8, *vp = vb + (size_t)key * e_dim + sg * 8;
            kst[0] = *reinterpret_cast<const float4 *>(kp); kst[1] = *reinterpret_cast<const float4 *>(kp + 4);
            vst[0] = *reinterpret_cast<const float4 *>(vp); vst[1] = *reinterpret_cast<const float4 *>(vp + 4);
            mst = (MASK && sg == 0) ? mb[key] : 0;
        } else {
            kst[0] = kst[1] = vst[0] = vst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            mst = 1;
        }
    };
    auto stash = [&](int buf) {
        if (!stager) return;
        unsigned char *base = sm + buf * AH_BUF;
        const float kv[8] = {kst[0].x, kst[0].y, kst[0].z, kst[0].w, kst[1].x, kst[1].y, kst[1].z, kst[1].w};
        v4u hi, lo;
        split8<M>(kv, hi, lo);
        *reinterpret_cast<v4u *>(base + skey * AH_ROW + sg * 32) = hi;
        *reinterpret_cast<v4u *>(base + skey * AH_ROW + sg * 32 + 16) = lo;
        const float vv[8] = {vst[0].x, vst[0].y, vst[0].z, vst[0].w, vst[1].x, vst[1].y, vst[1].z, vst[1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            unsigned int vh, vl;
            M::split(vv[e], vh, vl);
            unsigned char *row = base + AH_KT + (sg * 8 + e) * AH_ROW + spos * 2;
            *reinterpret_cast<unsigned short *>(row) = (unsigned short)vh;
            *reinterpret_cast<unsigned short *>(row + AH_VP) = (unsigned short)vl;
        }
        if (sg == 0) base[AH_KT + 2 * AH_VP + skey] = mst;
    };
    const int nblocks = (lk + AH_KEYS - 1) / AH_KEYS;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int blk = 0; blk < nblocks; ++blk) {
        const int cur = blk & 1;
        if (blk + 1 < nblocks) fetch((blk + 1) * AH_KEYS);                  // lands under this block's MFMAs
        const unsigned char *base = sm + cur * AH_BUF;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            // ---- S^T tile: 32 keys x 32 queries
            f32x16 s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const unsigned char *kp = base + (mt * 32 + l31) * AH_ROW + (2 * st + h) * 32;
                const v4u khi = *reinterpret_cast<const v4u *>(kp), klo = *reinterpret_cast<const v4u *>(kp + 16);
                s = M::mma(klo, qh[st], s);
                s = M::mma(khi, ql[st], s);
                s = M::mma(khi, qh[st], s);
            }
            // lane (query l31, half h): s[i] = score of key 8 (i / 4) + 4 h + i % 4 of the tile
            unsigned int mw[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) mw[j] = *reinterpret_cast<const unsigned int *>(base + AH_KT + 2 * AH_VP + mt * 32 + 8 * j + 4 * h);
            float p[16], tmax = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool dead = ((mw[i >> 2] >> (8 * (i & 3))) & 0xFFu) != 0u;
                p[i] = dead ? -INFINITY : s[i];
                tmax = fmaxf(tmax, p[i]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);
            float alpha = 1.f, psum = 0.f;
            if (m_new != -INFINITY) {
                alpha = __builtin_amdgcn_exp2f(m_run - m_new);               // m_run = -inf -> 0
#pragma unroll
                for (int i = 0; i < 16; ++i) { p[i] = __builtin_amdgcn_exp2f(p[i] - m_new); psum += p[i]; }
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) p[i] = 0.f;
            }
            l_run = l_run * alpha + psum;                                    // in-lane partial over this half's keys; halves summed at the end
            m_run = m_new;
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) o[i] *= alpha;
            }
            // ---- O^T += V^T . P^T: k-step t <-> the lane's registers 8 t .. 8 t + 7
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const float pv[8] = {p[8 * t], p[8 * t + 1], p[8 * t + 2], p[8 * t + 3], p[8 * t + 4], p[8 * t + 5], p[8 * t + 6], p[8 * t + 7]};
                v4u ph, pl;
                split8<M>(pv, ph, pl);
                const unsigned char *vp = base + AH_KT + l31 * AH_ROW + (mt * 32 + 16 * t + 8 * h) * 2;
                const v4u vhi = *reinterpret_cast<const v4u *>(vp), vlo = *reinterpret_cast<const v4u *>(vp + AH_VP);
                o = M::mma(vlo, ph, o);
                o = M::mma(vhi, pl, o);
                o = M::mma(vhi, ph, o);
            }
        }
        if (blk + 1 < nblocks) stash(cur ^ 1);         // the other buffer was last read before the previous barrier
        __syncthreads();
    }
    float l = l_run + __shfl_xor(l_run, 32, 64);
    if (qi < lq) {
        const float inv = 1.f / l;                     // fully masked row -> NaN, as torch.softmax gives
        float *dst = out + ((size_t)b * lq + qi) * e_dim + hd * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j)                    // o[i] = O^T[d = 8 (i / 4) + 4 h + i % 4][query]
            *reinterpret_cast<float4 *>(dst + 8 * j + 4 * h) = make_float4(o[4 * j] * inv, o[4 * j + 1] * inv, o[4 * j + 2] * inv, o[4 * j + 3] * inv);
    }
}

template <class M>
void launch_block_h(const float *q, const float *k, const float *v, const uint8_t *mask, int batch, int lq, int lk, int heads, float scale,
                    float *out, hipStream_t stream) {
    const int qw = (lq + 31) / 32, blocks = (qw + 7) / 8;
    int nw = (qw + blocks - 1) / blocks;
    if (nw < 4) nw = 4;                                // the first 256 threads stage the key blocks
    const dim3 grid(blocks, heads, batch);
    if (mask)
        hipLaunchKernelGGL((k_mha_block_h<M, true>), grid, dim3(64 * nw), 0, stream, q, k, v, mask, lq, lk, heads, scale, out);
    else
        hipLaunchKernelGGL((k_mha_block_h<M, false>), grid, dim3(64 * nw), 0, stream, q, k, v, mask, lq, lk, heads, scale, out);
}

}  // namespace
}  // namespace dz

using namespace dz;

extern "C" {

int dz_mha_core_split(const float *q, const float *k, const float *v, const uint8_t *key_padding_mask, int batch, int lq, int lk, int heads,
                      float scale, float *out, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(batch >= 0 && lq >= 0 && lk >= 1 && heads >= 1 && heads <= 65535 && batch <= 65535, "dz_mha_core_split: bad sizes");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_mha_core_split: math %d is not a split mode", math);
    if (batch == 0 || lq == 0) return DZ_OK;
    DZ_CHECK_ARG(q && k && v && out, "dz_mha_core_split: null pointer");
    if (math == DZ_MATH_F16X2) launch_block_h<MathF16>(q, k, v, key_padding_mask, batch, lq, lk, heads, scale, out, stream);
    else launch_block_h<MathBF16>(q, k, v, key_padding_mask, batch, lq, lk, heads, scale, out, stream);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
