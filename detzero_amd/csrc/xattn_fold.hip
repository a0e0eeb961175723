// Cross-attention of a FEW queries over a LONG memory with the key / value projections folded into the queries, fp32 MFMA, gfx950.
//
// Reference: refining/detzero_refine/models/modules/transformer/multi_head_attention.py:199-288 as called by the decoder layer
// (decoder.py:79-84) of the geometry refiner (geometry_transformer.py:118-140): 3 queries per object attend to 4096 memory points,
// 8 heads of 32 channels.  The reference projects the whole memory twice (K = Wk m + bk, V = Wv m + bv: 2 x 4096 x 256 x 256
// multiply-adds per object) to serve 3 queries.  For head h and query q
//     score(q, j) = scale * q_h . (Wk_h m_j + bk_h) = (scale * Wk_h^T q_h) . m_j + const(q, h)      (the constant drops out of the softmax)
//     out_h(q)    = sum_j p_j (Wv_h m_j + bv_h)     = Wv_h (sum_j p_j m_j) + bv_h                   (sum_j p_j = 1)
// so each (head, query) pair becomes ONE 256-channel folded query attending to the RAW memory rows, and the two projections shrink
// to (heads * queries) x 256 x 32 products before and after (k_fold_queries / k_fold_out).  With heads * queries <= 32 that is
// 4 * 32 * Lk * E multiply-adds instead of 2 * Lk * E * E + ..., and the memory is read from HBM exactly once: no K, no V, no
// split-precision copy of the memory.  Everything is fp32 (v_mfma_f32_16x16x4_f32), whatever math mode the linear layers run in.
//
// k_xattn_fold: grid (splits, B), 4 waves; a workgroup owns a key range of one object, a wave owns 16-key blocks of it:
//   the block's rows go HBM -> LDS with one direct-to-LDS dwordx4 per key row (wave-private double buffer, no barriers), XOR-swizzled
//   by the key so both MFMA operand patterns below are bank-conflict free;
//   S^T (16 keys x 32 rows) = m_tile (16 x 256) . Qf^T: A = m[key r][channel g*64 + s], B = Qf[row][g*64 + s] (LDS, 16-byte reads = 4 k-steps);
//   C layout: lane (g, r) holds keys 4g .. 4g+3 of folded query row r (+16) -> online softmax with 2 cross-lane ops per block;
//   ctx^T (256 channels x 32 rows) += m_tile^T . P^T: MFMA number e takes k-slot g <-> key 4g+e = register e of the lane (no
//   transposition of the probabilities), A = m[key 4g+e][channel] read from the same LDS tile; the rescale factor of a row is in-lane.
// Partial (max, sum, ctx) per split go to a workspace; k_fold_out merges them, normalises and applies Wv_h, bv.
#include <algorithm>

#include "common.h"
#include "hgemm.h"

namespace dz {
namespace {

constexpr int XF_E = 256, XF_ROWS = 32, XF_KB = 16, XF_WAVES = 4, XF_THREADS = 256;
constexpr int XF_ROWB = XF_E * 4;                       // bytes of a memory row
constexpr int XF_TILE = XF_KB * XF_ROWB;                // one 16-key block
constexpr int XF_OFF_Q = 0, XF_OFF_T = XF_ROWS * XF_ROWB, XF_LDS = XF_OFF_T + XF_WAVES * 2 * XF_TILE;
static_assert(XF_LDS <= 160 * 1024, "LDS");

__device__ __forceinline__ void xf_load16_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory", "m0");
}

// Qf[b][h * lq + qi][c] = scale * sum_d q[b][qi][h * hd + d] * Wk[h * hd + d][c]; rows >= heads * lq are zero.  Grid (heads + 1, B):
// one workgroup per (object, head) - the head's hd rows of Wk are read once per workgroup, coalesced - and one for the padding rows
__global__ __launch_bounds__(XF_THREADS) void k_fold_queries(const float *__restrict__ q, const float *__restrict__ wk_oi, int lq, int heads,
                                                              float scale, float *__restrict__ qf) {
    __shared__ float qs[XF_ROWS * XF_E];                 // the head's slice of the queries [qi][d]: lq * hd <= (32 / heads) * (256 / heads)
    const int hh = blockIdx.x, b = blockIdx.y, c = threadIdx.x, hd = XF_E / heads;
    float *dst = qf + (size_t)b * XF_ROWS * XF_E;
    if (hh == heads) {
        for (int r = heads * lq; r < XF_ROWS; ++r) dst[(size_t)r * XF_E + c] = 0.f;
        return;
    }
    for (int i = c; i < lq * hd; i += XF_THREADS) qs[i] = q[((size_t)b * lq + i / hd) * XF_E + hh * hd + i % hd] * scale;
    __syncthreads();
    for (int q0 = 0; q0 < lq; q0 += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int d = 0; d < hd; ++d) {
            const float w = wk_oi[(size_t)(hh * hd + d) * XF_E + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(qs[min(q0 + j, lq - 1) * hd + d], w, acc[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q0 + j < lq) dst[(size_t)(hh * lq + q0 + j) * XF_E + c] = acc[j];
    }
}

template <bool MASK>
__global__ __launch_bounds__(XF_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_xattn_fold(
    const float *__restrict__ qf, const float *__restrict__ mem, const uint8_t *__restrict__ kpm, int lk, int splits,
    float *__restrict__ part, float *__restrict__ pm, float *__restrict__ pl) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, r = lane & 15, g = lane >> 4;
    const int sp = blockIdx.x, b = blockIdx.y;

    // folded queries -> LDS, 16-byte granule gg of row rr at granule gg ^ (rr & 15)
    {
        const float4 *src = reinterpret_cast<const float4 *>(qf + (size_t)b * XF_ROWS * XF_E);
        for (int i = tid; i < XF_ROWS * 64; i += XF_THREADS) {
            const int rr = i >> 6, gg = i & 63;
            *reinterpret_cast<float4 *>(smem_raw + XF_OFF_Q + rr * XF_ROWB + ((gg ^ (rr & 15)) << 4)) = src[i];
        }
    }
    __syncthreads();

    const int nblk = (lk + XF_KB - 1) / XF_KB, per = (nblk + splits - 1) / splits;
    const int blk0 = sp * per, blk1 = min(blk0 + per, nblk);
    const srsrc_t rsrc = make_srsrc(mem + (size_t)b * lk * XF_E, (unsigned int)lk * XF_ROWB);
    const uint8_t *mb = MASK ? kpm + (size_t)b * lk : nullptr;
    const unsigned int tbase = XF_OFF_T + wid * 2 * XF_TILE;

    auto issue = [&](int blk, int buf) {
#pragma unroll
        for (int i = 0; i < XF_KB; ++i) {
            const int key = min(blk * XF_KB + i, lk - 1);
            xf_load16_lds(tbase + buf * XF_TILE + i * XF_ROWB, (unsigned int)((lane ^ i) << 4), rsrc, (unsigned int)key * XF_ROWB);
        }
    };

    f32x4 ctx[2][16];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int c = 0; c < 16; ++c) ctx[nb][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};

    int blk = blk0 + wid, it = 0;
    if (blk < blk1) issue(blk, 0);
    for (; blk < blk1; blk += XF_WAVES, ++it) {
        const int buf = it & 1;
        // the other buffer was last read two iterations ago: its LDS reads have returned (their MFMAs were issued), say so to the
        // hardware before the asynchronous writes go out
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (blk + XF_WAVES < blk1) {
            issue(blk + XF_WAVES, buf ^ 1);
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const unsigned char *tile = smem_raw + tbase + buf * XF_TILE;

        // ---- S^T: 16 keys x 32 rows over 256 channels (64 k-steps of 4 channels), two chains per row block
        f32x4 st[2][2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) st[nb][0] = st[nb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int off = r * XF_ROWB + (((g * 16 + j) ^ r) << 4);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(tile + off);
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(smem_raw + XF_OFF_Q + off);
            const f32x4 b1 = *reinterpret_cast<const f32x4 *>(smem_raw + XF_OFF_Q + 16 * XF_ROWB + off);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                st[0][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b0[e], st[0][j & 1], 0, 0, 0);
                st[1][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b1[e], st[1][j & 1], 0, 0, 0);
            }
        }
        // ---- online softmax: lane (g, r) holds keys 4g .. 4g+3 of rows r and 16 + r
        bool dead[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int key = blk * XF_KB + g * 4 + e;
            dead[e] = key >= lk;
            if (MASK) dead[e] = dead[e] | (mb[min(key, lk - 1)] != 0);
        }
        float p[2][4], alpha[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            float sv[4], tmax = -INFINITY;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sv[e] = dead[e] ? -INFINITY : st[nb][0][e] + st[nb][1][e];
                tmax = fmaxf(tmax, sv[e]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run[nb], tmax);
            alpha[nb] = 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) p[nb][e] = 0.f;
            if (m_new != -INFINITY) {
                alpha[nb] = expf(m_run[nb] - m_new);
#pragma unroll
                for (int e = 0; e < 4; ++e) p[nb][e] = expf(sv[e] - m_new);
            }
            l_run[nb] = l_run[nb] * alpha[nb] + (p[nb][0] + p[nb][1] + p[nb][2] + p[nb][3]);
            m_run[nb] = m_new;
        }
        if (__any(alpha[0] != 1.f || alpha[1] != 1.f)) {
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int c = 0; c < 16; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) ctx[nb][c][e] *= alpha[nb];
        }
        // ---- ctx^T += m_tile^T . P^T
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = g * 4 + e;
            const unsigned char *row = tile + kk * XF_ROWB + ((r & 3) << 2);
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float a = *reinterpret_cast<const float *>(row + (((4 * c + (r >> 2)) ^ kk) << 4));
                ctx[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[0][e], ctx[0][c], 0, 0, 0);
                ctx[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[1][e], ctx[1][c], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // ---- merge the 4 waves: each writes its ctx over its OWN tile buffers ([row][channel], granule ^ (row & 15)), max / sum over the
    // folded queries once every wave is past them
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        l_run[nb] += __shfl_xor(l_run[nb], 16, 64);
        l_run[nb] += __shfl_xor(l_run[nb], 32, 64);
#pragma unroll
        for (int c = 0; c < 16; ++c)      // ctx[nb][c][e] = ctx^T[channel 16c + 4g + e][row nb * 16 + r]
            *reinterpret_cast<f32x4 *>(smem_raw + tbase + (nb * 16 + r) * XF_ROWB + (((4 * c + g) ^ r) << 4)) = ctx[nb][c];
    }
    __syncthreads();
    float *ms = reinterpret_cast<float *>(smem_raw + XF_OFF_Q), *ls = ms + XF_WAVES * XF_ROWS;
    if (g == 0) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            ms[wid * XF_ROWS + nb * 16 + r] = m_run[nb];
            ls[wid * XF_ROWS + nb * 16 + r] = l_run[nb];
        }
    }
    __syncthreads();
    const size_t slot = (size_t)b * splits + sp;
    for (int row = 0; row < XF_ROWS; ++row) {
        float mx = -INFINITY;
#pragma unroll
        for (int w = 0; w < XF_WAVES; ++w) mx = fmaxf(mx, ms[w * XF_ROWS + row]);
        float acc = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < XF_WAVES; ++w) {
            const float mw = ms[w * XF_ROWS + row];
            const float f = mw == -INFINITY ? 0.f : expf(mw - mx);
            l += f * ls[w * XF_ROWS + row];
            acc += f * *reinterpret_cast<const float *>(smem_raw + XF_OFF_T + w * 2 * XF_TILE + row * XF_ROWB + (((tid >> 2) ^ (row & 15)) << 4) + ((tid & 3) << 2));
        }
        part[(slot * XF_ROWS + row) * XF_E + tid] = acc;
        if (tid == 0) {
            pm[slot * XF_ROWS + row] = mx;
            pl[slot * XF_ROWS + row] = l;
        }
    }
}

// out[b][qi][h * hd + d] = bv + sum_c ctx[h * lq + qi][c] * Wv[h * hd + d][c]; ctx = merged and normalised partials.  Grid (heads, B):
// a workgroup merges the lq rows of its head, then thread (d = t % hd, part = t / hd) sums its share of the 256 channels
__global__ __launch_bounds__(XF_THREADS) void k_fold_out(const float *__restrict__ part, const float *__restrict__ pm, const float *__restrict__ pl,
                                                         const float *__restrict__ wv_io, const float *__restrict__ bv, int lq, int heads, int splits,
                                                         float *__restrict__ out) {
    __shared__ float cn[XF_ROWS * XF_E];                 // the head's lq <= 32 / heads merged rows
    __shared__ float red[XF_THREADS];
    const int hh = blockIdx.x, b = blockIdx.y, t = threadIdx.x, hd = XF_E / heads;
    for (int qi = 0; qi < lq; ++qi) {
        const int row = hh * lq + qi;
        float mx = -INFINITY;
        for (int s = 0; s < splits; ++s) mx = fmaxf(mx, pm[((size_t)b * splits + s) * XF_ROWS + row]);
        float acc = 0.f, l = 0.f;
        for (int s = 0; s < splits; ++s) {
            const size_t slot = ((size_t)b * splits + s) * XF_ROWS + row;
            const float mw = pm[slot];
            const float f = mw == -INFINITY ? 0.f : expf(mw - mx);
            l += f * pl[slot];
            acc += f * part[slot * XF_E + t];
        }
        cn[qi * XF_E + t] = acc / l;                   // every key masked: 0 / 0 = NaN, as torch.softmax gives
    }
    __syncthreads();
    const int d = t % hd, pr = t / hd, nparts = XF_THREADS / hd, span = XF_E / nparts;
    for (int qi = 0; qi < lq; ++qi) {
        float acc = 0.f;
        for (int c = pr * span; c < (pr + 1) * span; ++c) acc = fmaf(cn[qi * XF_E + c], wv_io[(size_t)c * XF_E + hh * hd + d], acc);
        red[t] = acc;
        __syncthreads();
        if (pr == 0) {
            float v = bv[hh * hd + d];
            for (int k = 0; k < nparts; ++k) v += red[k * hd + d];
            out[((size_t)b * lq + qi) * XF_E + hh * hd + d] = v;
        }
        __syncthreads();
    }
}

int xf_splits(int b, int lk) {
    const int nblk = (lk + XF_KB - 1) / XF_KB;
    int s = (4 * device_cus() + b - 1) / b;                 // about 4 workgroups per CU over the launch
    s = std::min(s, std::max(1, nblk / 16));                // at least 4 key blocks per wave
    return std::max(1, std::min(s, 64));
}

}  // namespace
}  // namespace dz

using namespace dz;

extern "C" {

int dz_xattn_folded_supported(int lq, int e, int heads) { return e == XF_E && heads > 0 && e % heads == 0 && heads * lq <= XF_ROWS && lq > 0; }

size_t dz_xattn_folded_workspace_bytes(int b, int lk) {
    const size_t s = (size_t)xf_splits(std::max(b, 1), std::max(lk, 1));
    return ((size_t)b * XF_ROWS * XF_E + (size_t)b * s * XF_ROWS * (XF_E + 2)) * sizeof(float);
}

int dz_xattn_folded(const float *q, const float *mem, const uint8_t *key_padding_mask, const float *wk_oi, const float *wv_io, const float *bv,
                    int b, int lq, int lk, int e, int heads, float scale, float *workspace, size_t workspace_bytes, float *out, void *stream) {
    DZ_CHECK_ARG(dz_xattn_folded_supported(lq, e, heads) && lk > 0 && b >= 0,
                 "dz_xattn_folded: needs E = 256 and heads * queries <= 32 (got E %d, heads %d, queries %d, keys %d)", e, heads, lq, lk);
    if (b == 0) return DZ_OK;
    DZ_CHECK_ARG(q && mem && wk_oi && wv_io && bv && workspace && out, "dz_xattn_folded: null pointer");
    DZ_CHECK_ARG((size_t)lk * XF_ROWB < 0x80000000ull, "dz_xattn_folded: %d keys per object exceed the 2 GiB buffer-addressing limit", lk);
    DZ_CHECK_ARG(workspace_bytes >= dz_xattn_folded_workspace_bytes(b, lk), "dz_xattn_folded: workspace of %zu bytes, need %zu", workspace_bytes,
                 dz_xattn_folded_workspace_bytes(b, lk));
    hipStream_t st = (hipStream_t)stream;
    const int splits = xf_splits(b, lk);
    float *qf = workspace, *part = qf + (size_t)b * XF_ROWS * XF_E, *pm = part + (size_t)b * splits * XF_ROWS * XF_E, *pl = pm + (size_t)b * splits * XF_ROWS;
    hipLaunchKernelGGL(k_fold_queries, dim3(heads + 1, b), dim3(XF_THREADS), 0, st, q, wk_oi, lq, heads, scale, qf);
    int rc;
    if (key_padding_mask) {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_xattn_fold<true>), XF_LDS, done, "dz_xattn_folded"))) return rc;
        hipLaunchKernelGGL(k_xattn_fold<true>, dim3(splits, b), dim3(XF_THREADS), XF_LDS, st, qf, mem, key_padding_mask, lk, splits, part, pm, pl);
    } else {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_xattn_fold<false>), XF_LDS, done, "dz_xattn_folded"))) return rc;
        hipLaunchKernelGGL(k_xattn_fold<false>, dim3(splits, b), dim3(XF_THREADS), XF_LDS, st, qf, mem, (const uint8_t *)nullptr, lk, splits, part, pm, pl);
    }
    hipLaunchKernelGGL(k_fold_out, dim3(heads, b), dim3(XF_THREADS), 0, st, part, pm, pl, wv_io, bv, lq, heads, splits, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
