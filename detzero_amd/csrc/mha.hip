// Fused multi-head attention core for the refining module (GRM / PRM), fp32 MFMA, gfx950.
//
// Reference: refining/detzero_refine/models/modules/transformer/multi_head_attention.py:207-288
//   q = q * scaling; attn = bmm(q, k^T); masked_fill(key_padding_mask, -inf); softmax; bmm(attn, v)
// The reference materialises the (B*heads, Lq, Lk) score tensor (PRM: 200 x 9600 per head) and a
// head-averaged copy of it; here one wavefront owns 16 queries of one (batch, head) and streams the
// keys once with an online softmax - scores never leave registers.
//
// Register-only dataflow (no LDS, no barriers).  With v_mfma_f32_16x16x4_f32:
//   S^T (16 keys x 16 queries) = K_tile (16 x 32) . Q^T (32 x 16)
//       A = K rows straight from global memory (one 16-byte load per lane per 16-d slice),
//       B = Q^T, pre-scaled, resident in 8 VGPRs for the whole key loop.
//     C layout: lane (g = l>>4, r = l&15) holds keys 4g..4g+3 of query r -> the softmax reduction
//     over keys is 3 in-lane ops + 2 cross-lane shuffles, and each lane's running max / sum /
//     rescale factor belong to ITS query column.
//   O^T (32 x 16 queries) += V^T (32 x 16 keys) . P^T (16 keys x 16 queries)
//       B = P^T: MFMA number e takes k-slot g' <-> key 4g'+e, which is exactly register e of
//       lane (g', r): the probabilities feed the second GEMM without leaving their lane.
//       A = V[key 4g+e][d] loaded per lane (16 lanes read 64 contiguous bytes).
#include <stdlib.h>

#include "igemm.h"

namespace dz {

constexpr int HD = 32;  // head dim of every DetZero refiner config (256 / 8 heads)

// HDT: head dim (32 for the refiner's heads; 64 .. 256 for the single wide head of the PDV encoder layer, dz_attention_single_head)
template <bool MASK, int HDT = HD>
__global__ __launch_bounds__(256) void k_mha_core(const float *__restrict__ q, const float *__restrict__ k,
                                                  const float *__restrict__ v, const uint8_t *__restrict__ kpm,
                                                  int batch, int lq, int lk, int heads, float scale,
                                                  float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int qtiles = (lq + 15) / 16;
    const long item = (long)blockIdx.x * 4 + wid;
    const long nitems = (long)batch * heads * qtiles;
    if (item >= nitems) return;
    const int qt = (int)(item % qtiles);
    const int h = (int)((item / qtiles) % heads);
    const int b = (int)(item / ((long)qtiles * heads));
    const int e_dim = heads * HDT;
    constexpr int NS = HDT / 16;                    // 16-channel slices of a head

    // B operand of S^T: Q[query r][d = 16*s + 4g + e] * scale
    const int qi = qt * 16 + r;
    float qreg[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qi < lq) t = *reinterpret_cast<const float4 *>(q + ((size_t)b * lq + qi) * e_dim + h * HDT + s * 16 + g * 4);
        qreg[s][0] = t.x * scale; qreg[s][1] = t.y * scale; qreg[s][2] = t.z * scale; qreg[s][3] = t.w * scale;
    }

    f32x4 o[NS];
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;   // per query column r (replicated over g)
    const float *kb = k + (size_t)b * lk * e_dim + h * HDT;
    const float *vb = v + (size_t)b * lk * e_dim + h * HDT;
    const uint8_t *mb = kpm ? kpm + (size_t)b * lk : nullptr;

    for (int key0 = 0; key0 < lk; key0 += 16) {
        // ---- S^T tile
        const int krow = key0 + r;                      // A operand row = key
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (krow < lk) kv = *reinterpret_cast<const float4 *>(kb + (size_t)krow * e_dim + sl * 16 + g * 4);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qreg[sl][0], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qreg[sl][1], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qreg[sl][2], s, 0, 0, 0);
            s = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qreg[sl][3], s, 0, 0, 0);
        }
        // lane (g,r): s[e] = score(query r, key key0 + 4g + e); masked / out-of-range keys -> -inf
        // (branch-free selects on scalars: no conditional writes into the accumulator vector)
        float sv[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = key0 + g * 4 + e;
            bool dead = key >= lk;
            if (MASK) {
                const unsigned char mk = mb[min(key, lk - 1)];
                dead = dead | (mk != 0);
            }
            sv[e] = dead ? -INFINITY : s[e];
            tmax = fmaxf(tmax, sv[e]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        float alpha = 1.f;
        float p[4] = {0.f, 0.f, 0.f, 0.f};
        if (m_new != -INFINITY) {
            alpha = expf(m_run - m_new);               // m_run = -inf -> 0
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = expf(sv[e] - m_new);
        }
        l_run = l_run * alpha + (p[0] + p[1] + p[2] + p[3]);   // in-lane partial; reduced over g at the end
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < NS; ++dt)
#pragma unroll
            for (int e = 0; e < 4; ++e) o[dt][e] *= alpha;
        // ---- O^T += V^T . P^T
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = key0 + g * 4 + e;
#pragma unroll
            for (int dt = 0; dt < NS; ++dt) {
                const float vv = (key < lk) ? vb[(size_t)key * e_dim + dt * 16 + r] : 0.f;
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, p[e], o[dt], 0, 0, 0);
            }
        }
    }
    // row sums: every lane of column r holds the partial over its own keys
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // o[dt][e] = O^T[d = dt*16 + 4g + e][query r]
    if (qi < lq) {
        const float inv = 1.f / l_run;               // fully masked row -> NaN, as torch.softmax gives
        float *dst = out + ((size_t)b * lq + qi) * e_dim + h * HDT;
#pragma unroll
        for (int dt = 0; dt < NS; ++dt) {
            float4 w = make_float4(o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv);
            *reinterpret_cast<float4 *>(dst + dt * 16 + g * 4) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Block variant for long query lists (PRM: 200 queries x 9600 keys per head).  k_mha_core lets every wave stream the whole K / V
// of its (batch, head) from L2 - 13 waves re-read the same 2.4 MB, and V arrives through scalar 4-byte loads.  Here a workgroup
// (up to 8 waves) owns up to 256 queries of one (batch, head): key blocks of 64 are staged ONCE per workgroup in LDS (double buffered, one
// barrier per block; V transposed on the way in so that the A operand of the P.V product is one 16-byte LDS read), and every wave
// carries TWO 16-query column tiles, so each K / V fragment read from LDS feeds two MFMAs.  Same register-resident online softmax.
// ------------------------------------------------------------------------------------------------
constexpr int MB_KEYS = 64;                 // keys per staged block
constexpr int MB_KSTR = HD + 4;             // row stride of the K tile (floats): 16-byte aligned rows, staggered banks
constexpr int MB_VSTR = MB_KEYS + 4;        // row stride of the transposed V tile

template <bool MASK>
__global__ __launch_bounds__(512) void k_mha_block(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                   const uint8_t *__restrict__ kpm, int lq, int lk, int heads, float scale,
                                                   float *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) float ks[2][MB_KEYS * MB_KSTR];
    __shared__ __attribute__((aligned(16))) float vt[2][HD * MB_VSTR];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int e_dim = heads * HD;
    const int qbase = (blockIdx.x * (blockDim.x >> 6) + wid) * 32;      // up to 8 waves x 32 queries per workgroup
    const bool stager = tid < 256;                                      // 64 keys x 4 pieces of 8 channels per block
    float qreg[2][2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = qbase + t * 16 + r;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qi < lq) x = *reinterpret_cast<const float4 *>(q + ((size_t)b * lq + qi) * e_dim + h * HD + sl * 16 + g * 4);
            // scores are kept in units of log2(e): the softmax below runs on v_exp_f32 (2^x) directly
            const float sc2 = scale * 1.44269504088896340736f;
            qreg[t][sl][0] = x.x * sc2; qreg[t][sl][1] = x.y * sc2; qreg[t][sl][2] = x.z * sc2; qreg[t][sl][3] = x.w * sc2;
        }
    }
    f32x4 o[2][2];
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *kb = k + (size_t)b * lk * e_dim + h * HD;
    const float *vb = v + (size_t)b * lk * e_dim + h * HD;
    const uint8_t *mb = kpm ? kpm + (size_t)b * lk : nullptr;
    // staging: thread -> (key = tid / 4 [+0], 8 consecutive channels (tid % 4) * 8) of the block: two float4 of K and two of V
    const int skey = tid >> 2, sch = (tid & 3) * 8;
    float4 kst[2], vst[2];
    auto fetch = [&](int key0) {
        const int key = key0 + skey;
        if (!stager) return;
        if (key < lk) {
            const float *kp = kb + (size_t)key * e_dim + sch, *vp = vb + (size_t)key * e_dim + sch;
            kst[0] = *reinterpret_cast<const float4 *>(kp); kst[1] = *reinterpret_cast<const float4 *>(kp + 4);
            vst[0] = *reinterpret_cast<const float4 *>(vp); vst[1] = *reinterpret_cast<const float4 *>(vp + 4);
        } else {
            kst[0] = kst[1] = vst[0] = vst[1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stash = [&](int buf) {
        if (!stager) return;
        float *kd = ks[buf] + skey * MB_KSTR + sch;
        *reinterpret_cast<float4 *>(kd) = kst[0]; *reinterpret_cast<float4 *>(kd + 4) = kst[1];
        float *vd = vt[buf] + sch * MB_VSTR + skey;                    // transposed: [channel][key]
        vd[0] = vst[0].x; vd[MB_VSTR] = vst[0].y; vd[2 * MB_VSTR] = vst[0].z; vd[3 * MB_VSTR] = vst[0].w;
        vd[4 * MB_VSTR] = vst[1].x; vd[5 * MB_VSTR] = vst[1].y; vd[6 * MB_VSTR] = vst[1].z; vd[7 * MB_VSTR] = vst[1].w;
    };
    const int nblocks = (lk + MB_KEYS - 1) / MB_KEYS;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int blk = 0; blk < nblocks; ++blk) {
        const int cur = blk & 1, key0 = blk * MB_KEYS;
        if (blk + 1 < nblocks) fetch(key0 + MB_KEYS);                  // lands under this block's MFMAs
#pragma unroll
        for (int kt = 0; kt < MB_KEYS / 16; ++kt) {
            // ---- S^T tiles: A = K rows (key r of this 16-key tile), shared by the wave's two query tiles
            f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const float4 kv = *reinterpret_cast<const float4 *>(ks[cur] + (kt * 16 + r) * MB_KSTR + sl * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qreg[t][sl][0], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qreg[t][sl][1], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qreg[t][sl][2], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qreg[t][sl][3], s[t], 0, 0, 0);
                }
            }
            bool dead[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = key0 + kt * 16 + g * 4 + e;
                dead[e] = key >= lk;
                if (MASK) dead[e] = dead[e] | (mb[min(key, lk - 1)] != 0);
            }
            float p[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float sv[4], tmax = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) { sv[e] = dead[e] ? -INFINITY : s[t][e]; tmax = fmaxf(tmax, sv[e]); }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run[t], tmax);
                float alpha = 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) p[t][e] = 0.f;
                if (m_new != -INFINITY) {
                    alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[t][e] = __builtin_amdgcn_exp2f(sv[e] - m_new);
                }
                l_run[t] = l_run[t] * alpha + (p[t][0] + p[t][1] + p[t][2] + p[t][3]);
                m_run[t] = m_new;
                if (!__all(alpha == 1.f)) {              // the running maxima settle after a few tiles: no rescale then
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[t][dt][e] *= alpha;
                }
            }
            // ---- O^T += V^T . P^T: A = V^T[d = dt*16 + r][keys 4g .. 4g+3 of the tile] = one 16-byte read of the transposed tile
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const float4 vv = *reinterpret_cast<const float4 *>(vt[cur] + (dt * 16 + r) * MB_VSTR + kt * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.x, p[t][0], o[t][dt], 0, 0, 0);
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.y, p[t][1], o[t][dt], 0, 0, 0);
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.z, p[t][2], o[t][dt], 0, 0, 0);
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.w, p[t][3], o[t][dt], 0, 0, 0);
                }
            }
        }
        if (blk + 1 < nblocks) stash(cur ^ 1);         // the other buffer was last read before the previous barrier
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float l = l_run[t];
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const int qi = qbase + t * 16 + r;
        if (qi < lq) {
            const float inv = 1.f / l;                 // fully masked row -> NaN, as torch.softmax gives
            float *dst = out + ((size_t)b * lq + qi) * e_dim + h * HD;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
                *reinterpret_cast<float4 *>(dst + dt * 16 + g * 4) = make_float4(o[t][dt][0] * inv, o[t][dt][1] * inv, o[t][dt][2] * inv, o[t][dt][3] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// One WIDE head per sequence (the PDV encoder layer: R RoIs x L = 216 grid points x E = 192 channels).  k_mha_core<.., E> lets each of the
// 14 waves of a RoI stream the RoI's K and V (2 x 166 KB) from L2 on its own; here a workgroup owns the RoI (up to 8 waves x 32 queries)
// and stages K / V once, in blocks of 64 keys: K as rows [key][E + 4], V transposed [channel][64 + 4] (the A operand of the second
// product is then one 16-byte LDS read).  Single buffer (102 KB at E = 192), two barriers per block; a wave carries two 16-query tiles
// so that every K / V fragment read feeds two MFMAs.  Same fp32 arithmetic and online softmax as k_mha_block.
// ------------------------------------------------------------------------------------------------
template <bool MASK, int E>
__global__ __launch_bounds__(512) void k_attn1h_block(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                      const uint8_t *__restrict__ kpm, int l, float scale, float *__restrict__ out) {
    constexpr int NS = E / 16, KSTR = E + 4, VSTR = MB_KEYS + 4;
    extern __shared__ __attribute__((aligned(16))) float sm1h[];
    float *const ks = sm1h;                              // [64][KSTR]
    float *const vt = sm1h + MB_KEYS * KSTR;             // [E][VSTR]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nthr = blockDim.x;
    const int r = lane & 15, g = lane >> 4;
    const size_t seq = (size_t)blockIdx.x * l;
    const int qbase = wid * 32;
    float qreg[2][NS][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = qbase + t * 16 + r;
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qi < l) x = *reinterpret_cast<const float4 *>(q + (seq + qi) * E + sl * 16 + g * 4);
            const float sc2 = scale * 1.44269504088896340736f;       // scores in units of log2(e): the softmax runs on v_exp_f32
            qreg[t][sl][0] = x.x * sc2; qreg[t][sl][1] = x.y * sc2; qreg[t][sl][2] = x.z * sc2; qreg[t][sl][3] = x.w * sc2;
        }
    }
    f32x4 o[2][NS];
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < NS; ++dt) o[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float *kb = k + seq * E, *vb = v + seq * E;
    const uint8_t *mb = MASK ? kpm + seq : nullptr;
    const int nblocks = (l + MB_KEYS - 1) / MB_KEYS;
    for (int blk = 0; blk < nblocks; ++blk) {
        const int key0 = blk * MB_KEYS;
        __syncthreads();                                 // every wave is done with the previous block
        for (int i = tid; i < MB_KEYS * (E / 4); i += nthr) {
            const int key = i / (E / 4), c4 = i % (E / 4);
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (key0 + key < l) {
                kk = *reinterpret_cast<const float4 *>(kb + (size_t)(key0 + key) * E + c4 * 4);
                vv = *reinterpret_cast<const float4 *>(vb + (size_t)(key0 + key) * E + c4 * 4);
            }
            *reinterpret_cast<float4 *>(ks + key * KSTR + c4 * 4) = kk;
            float *vd = vt + (c4 * 4) * VSTR + key;
            vd[0] = vv.x; vd[VSTR] = vv.y; vd[2 * VSTR] = vv.z; vd[3 * VSTR] = vv.w;
        }
        __syncthreads();
#pragma unroll 1
        for (int kt = 0; kt < MB_KEYS / 16; ++kt) {
            if (key0 + kt * 16 >= l) break;              // (workgroup-uniform)
            f32x4 s[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) {
                const float4 kv = *reinterpret_cast<const float4 *>(ks + (kt * 16 + r) * KSTR + sl * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qreg[t][sl][0], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qreg[t][sl][1], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qreg[t][sl][2], s[t], 0, 0, 0);
                    s[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qreg[t][sl][3], s[t], 0, 0, 0);
                }
            }
            bool dead[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int key = key0 + kt * 16 + g * 4 + e;
                dead[e] = key >= l;
                if (MASK) dead[e] = dead[e] | (mb[min(key, l - 1)] != 0);
            }
            float p[2][4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float sv[4], tmax = -INFINITY;
#pragma unroll
                for (int e = 0; e < 4; ++e) { sv[e] = dead[e] ? -INFINITY : s[t][e]; tmax = fmaxf(tmax, sv[e]); }
                tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_run[t], tmax);
                float alpha = 1.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) p[t][e] = 0.f;
                if (m_new != -INFINITY) {
                    alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
#pragma unroll
                    for (int e = 0; e < 4; ++e) p[t][e] = __builtin_amdgcn_exp2f(sv[e] - m_new);
                }
                l_run[t] = l_run[t] * alpha + (p[t][0] + p[t][1] + p[t][2] + p[t][3]);
                m_run[t] = m_new;
                if (!__all(alpha == 1.f)) {
#pragma unroll
                    for (int dt = 0; dt < NS; ++dt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[t][dt][e] *= alpha;
                }
            }
#pragma unroll
            for (int dt = 0; dt < NS; ++dt) {
                const float4 vv = *reinterpret_cast<const float4 *>(vt + (dt * 16 + r) * VSTR + kt * 16 + g * 4);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.x, p[t][0], o[t][dt], 0, 0, 0);
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.y, p[t][1], o[t][dt], 0, 0, 0);
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.z, p[t][2], o[t][dt], 0, 0, 0);
                    o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv.w, p[t][3], o[t][dt], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float lsum = l_run[t];
        lsum += __shfl_xor(lsum, 16, 64);
        lsum += __shfl_xor(lsum, 32, 64);
        const int qi = qbase + t * 16 + r;
        if (qi < l) {
            const float inv = 1.f / lsum;              // fully masked row -> NaN, as torch.softmax gives
            float *dst = out + (seq + qi) * E;
#pragma unroll
            for (int dt = 0; dt < NS; ++dt)
                *reinterpret_cast<float4 *>(dst + dt * 16 + g * 4) = make_float4(o[t][dt][0] * inv, o[t][dt][1] * inv, o[t][dt][2] * inv, o[t][dt][3] * inv);
        }
    }
}

template <int E>
static bool launch_1h_block(const float *q, const float *k, const float *v, const unsigned char *mask, int r, int l, float scale, float *out, hipStream_t stream) {
    constexpr int LDS = (MB_KEYS * (E + 4) + E * (MB_KEYS + 4)) * 4;
    static_assert(LDS <= 160 * 1024, "LDS");
    const int nw = (l + 31) / 32;
    if (nw > 8) return false;
    static PerDeviceFlags done_m, done_u;
    if (mask) {
        if (reserve_lds(reinterpret_cast<const void *>(&k_attn1h_block<true, E>), LDS, done_m, "dz_attention_single_head")) return false;
        hipLaunchKernelGGL((k_attn1h_block<true, E>), dim3(r), dim3(64 * nw), LDS, stream, q, k, v, mask, l, scale, out);
    } else {
        if (reserve_lds(reinterpret_cast<const void *>(&k_attn1h_block<false, E>), LDS, done_u, "dz_attention_single_head")) return false;
        hipLaunchKernelGGL((k_attn1h_block<false, E>), dim3(r), dim3(64 * nw), LDS, stream, q, k, v, (const uint8_t *)nullptr, l, scale, out);
    }
    return true;
}

// One wide head (the PDV encoder layer: R sequences of L = 216 tokens, E = 192): k_mha_core with head dim E - one wave per 16 queries,
// keys streamed from L2, scores and probabilities in registers.  Returns false when E has no instance (the caller's VALU kernel runs).
template <int E>
static void launch_1h(const float *q, const float *k, const float *v, const unsigned char *mask, int r, int l, float scale, float *out, hipStream_t stream) {
    const long items = (long)r * ((l + 15) / 16);
    if (mask)
        hipLaunchKernelGGL((k_mha_core<true, E>), dim3(ceil_div(items, 4)), dim3(256), 0, stream, q, k, v, mask, r, l, l, 1, scale, out);
    else
        hipLaunchKernelGGL((k_mha_core<false, E>), dim3(ceil_div(items, 4)), dim3(256), 0, stream, q, k, v, mask, r, l, l, 1, scale, out);
}
bool attention_1h_mfma(const float *q, const float *k, const float *v, const unsigned char *mask, int r, int l, int e, float scale, float *out,
                       hipStream_t stream) {
    static const bool no_block = getenv("DZ_TUNE_ATT1H_NOBLOCK") != nullptr;
    if (!no_block && l > 64 && l <= 256) {             // K / V staged once per sequence
        if (e == 192 && launch_1h_block<192>(q, k, v, mask, r, l, scale, out, stream)) return true;
        if (e == 128 && launch_1h_block<128>(q, k, v, mask, r, l, scale, out, stream)) return true;
    }
    switch (e) {
        case 64: launch_1h<64>(q, k, v, mask, r, l, scale, out, stream); return true;
        case 128: launch_1h<128>(q, k, v, mask, r, l, scale, out, stream); return true;
        case 192: launch_1h<192>(q, k, v, mask, r, l, scale, out, stream); return true;
        case 256: launch_1h<256>(q, k, v, mask, r, l, scale, out, stream); return true;
        default: return false;
    }
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_mha_core(const float *q, const float *k, const float *v, const uint8_t *key_padding_mask, int batch, int lq,
                int lk, int heads, float scale, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(batch >= 0 && lq >= 0 && lk >= 1 && heads >= 1, "dz_mha_core: bad sizes");
    if (batch == 0 || lq == 0) return DZ_OK;
    DZ_CHECK_ARG(q && k && v && out, "dz_mha_core: null pointer");
    if (lq > 32 && batch <= 65535 && heads <= 65535) {           // long query lists: K / V staged once per 128 queries
        // a workgroup = up to 8 waves of 32 queries (PRM: 200 queries -> 7 waves, K / V staged once per (batch, head))
        const int qw = (lq + 31) / 32, blocks = (qw + 7) / 8;
        int nw = (qw + blocks - 1) / blocks;
        if (nw < 4) nw = 4;                                        // the first 256 threads stage the key blocks
        const dim3 grid(blocks, heads, batch);
        if (key_padding_mask)
            hipLaunchKernelGGL(k_mha_block<true>, grid, dim3(64 * nw), 0, stream, q, k, v, key_padding_mask, lq, lk, heads, scale, out);
        else
            hipLaunchKernelGGL(k_mha_block<false>, grid, dim3(64 * nw), 0, stream, q, k, v, key_padding_mask, lq, lk, heads, scale, out);
        DZ_LAUNCH_CHECK();
        return DZ_OK;
    }
    const long items = (long)batch * heads * ((lq + 15) / 16);
    if (key_padding_mask)
        hipLaunchKernelGGL(k_mha_core<true>, dim3(ceil_div(items, 4)), dim3(256), 0, stream, q, k, v, key_padding_mask,
                           batch, lq, lk, heads, scale, out);
    else
        hipLaunchKernelGGL(k_mha_core<false>, dim3(ceil_div(items, 4)), dim3(256), 0, stream, q, k, v, key_padding_mask,
                           batch, lq, lk, heads, scale, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
