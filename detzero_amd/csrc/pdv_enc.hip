// PDV's encoder layer around its attention core, as TWO row-chain kernels on pair16 operands (the head's split math modes):
//
//   k_enc_front   pos = W_p2 . ReLU(BN(W_p1 . positional input)) + b          (FeedForwardPositionalEncoding, attention_utils.py:112-133)
//                 src = features + (row gets an encoding ? pos : 0)            (attention_utils.py:31-44)          -> pair16 rows
//                 q'  = src . Mq + uq                                          (folded query: csrc/pdv_attn.hip)   -> pair16 rows
//   [dz_self_attention_split: o' = softmax(q' src^T) src]
//   k_enc_back    x   = LayerNorm1(src + o' . Mvo + bvo)                       (nn.TransformerEncoderLayer, post-norm, attention_utils.py:17-52)
//                 y   = LayerNorm2(x + W2 . ReLU(W1 . x + b1) + b2)
//                 out = pooled + (RoI without points ? pooled : y)             (COMBINE, pdv_head.py:540-560)      -> fp32 rows
//
// Layer by layer (dz_linear_forward_split, dz_add_layernorm, conversions) every one of these lines is a pass over 841 k rows x 768
// bytes of HBM per 8-frame batch - 12 such tensors written and read back between the pooled features and the encoder's output.  Here a
// WAVE owns 32 rows and carries them through the whole chain in registers (the operand chaining of pointnet.hip: the accumulator of a
// 32 x 32 fragment, after bias / BatchNorm / ReLU / LayerNorm and the (hi, lo) split, becomes the next layer's MFMA operand with one
// exchange between lane and lane ^ 32); HBM sees the inputs once and the outputs once.
//   * weights stream through LDS in slices of 24 KB (a ring of three, two slices in flight, `buffer_load_dwordx4 ... lds`, unpadded
//     128-byte rows with the 16-byte pieces XOR-swizzled), one workgroup barrier per slice = per 18-40 MFMAs of each of the 8 waves;
//     a slice is 32 output channels x all inputs, or - for the layer whose input comes from memory - 32 INPUT channels x all 192
//     outputs, so that its operands arrive k-step by k-step under the MFMAs of the slice before (double-buffered registers);
//   * residuals ride on the matrix pipe: adding rows that already exist as MFMA operands (src, x, the split pooled features) is two
//     MFMAs per k-step against an identity fragment built in registers - no accumulator-layout loads, no reverse exchange;
//   * LayerNorm in registers: a row's 192 channels are 96 accumulator registers of its lane and 96 of lane ^ 32;
//   * every load in the loop is inline asm with counted `s_waitcnt vmcnt` (vector memory operations retire in order), the number of
//     loads and stores per step is fixed (rows past the end use an out-of-range buffer offset: zeros come back, stores are dropped).
#include <stdlib.h>

#include "hgemm.h"

namespace dz {
namespace {

constexpr int EN_E = 192, EN_F = 128, EN_P = 96;                  // model width, feed-forward width, hidden width of the positional encoder
constexpr int EN_THREADS = 512, EN_WAVES = 8, EN_TILE = EN_WAVES * 32;
constexpr int EN_SLICE = 24576, EN_RING = 3;
constexpr int EN_ROWB = EN_E * 4;                                  // 768 bytes: a row of 192 channels, pair16 or fp32
constexpr int EN_OFF_RING = 0, EN_OFF_VEC = EN_RING * EN_SLICE, EN_OFF_W0 = EN_OFF_VEC + 8192, EN_W0_ROW = 80;
constexpr int EN_LDS = EN_OFF_W0 + EN_P * EN_W0_ROW;
static_assert(EN_LDS <= 160 * 1024, "LDS");

template <class M>
__device__ __forceinline__ unsigned int one16() { return M::ID == 1 ? 0x3C00u : 0x3F80u; }

__device__ __forceinline__ void en_ld16(v4u &dst, srsrc_t rsrc, unsigned int voff) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void en_ld16o(v4u &dst, srsrc_t rsrc, unsigned int voff) {       // + 16 bytes
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void en_ld1(unsigned int &dst, srsrc_t rsrc, unsigned int voff) {
    asm volatile("buffer_load_ubyte %0, %1, %2, 0 offen" : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void en_st16(v4u v, srsrc_t rsrc, unsigned int voff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
__device__ __forceinline__ void en_st16o(v4u v, srsrc_t rsrc, unsigned int voff) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:16" ::"v"(v), "v"(voff), "s"(rsrc) : "memory");
}
#define EN_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ void en_pin(v4u &v) { asm volatile("" : "+v"(v)); }

// weight slices -> ring buffer `buf` (3 wave-instructions of 1 KB per wave; unit u at LDS offset u * 16)
//   slab: 192 rows x 128 bytes at byte column colb of rows of `rowb` bytes     -> [row n][8 pieces], piece p stored at p ^ ((n >> 1) & 7)
__device__ __forceinline__ void en_issue_slab(srsrc_t rsrc, int colb, int rowb, int buf, int tid, int wid) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = j * EN_THREADS + tid, n = u >> 3, pc = (u & 7) ^ ((n >> 1) & 7);
        const unsigned int off = (unsigned int)(n * rowb + colb + pc * 16);
        const unsigned int base = __builtin_amdgcn_readfirstlane((unsigned int)(EN_OFF_RING + buf * EN_SLICE + (j * EN_THREADS + wid * 64) * 16));
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(base), "v"(off), "s"(rsrc) : "memory", "m0");
    }
}
//   rows: 32 rows r0 .. r0 + 31 x k channels (k % 32 == 0, k <= 192)                -> [chunk of 32 channels][row n][8 pieces], same swizzle
__device__ __forceinline__ void en_issue_rows(srsrc_t rsrc, int r0, int k, int buf, int tid, int wid) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = j * EN_THREADS + tid, chunk = u >> 8, n = (u >> 3) & 31, pc = (u & 7) ^ ((n >> 1) & 7);
        const unsigned int off = chunk * 32 < k ? (unsigned int)((r0 + n) * k * 4 + chunk * 128 + pc * 16) : OOB_OFFSET;
        const unsigned int base = __builtin_amdgcn_readfirstlane((unsigned int)(EN_OFF_RING + buf * EN_SLICE + (j * EN_THREADS + wid * 64) * 16));
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(base), "v"(off), "s"(rsrc) : "memory", "m0");
    }
}

// fragment of k-step s (16 input channels) for the 32 output channels r0 + lane & 31 of a [chunk][rows_per_chunk rows][128 B] slice
__device__ __forceinline__ void en_wfrag(const unsigned char *sm, int base, int rows_per_chunk, int r0, int s, int l31, int h, v4u &hi, v4u &lo) {
    const unsigned char *p = sm + base + (((s >> 1) * rows_per_chunk + r0 + l31) << 7);
    const int sw = (l31 >> 1) & 7, p0 = (s & 1) * 4 + h * 2;
    hi = *reinterpret_cast<const v4u *>(p + ((p0 ^ sw) << 4));
    lo = *reinterpret_cast<const v4u *>(p + (((p0 + 1) ^ sw) << 4));
}

// accumulator fragment (lane (row, h) holds channels 8 q + 4 h + i of the fragment in v[4 q + i]) -> the MFMA operands of the two
// k-steps the fragment's 32 channels make (pointnet.hip): groups with (q & 1) == h stay, the others are swapped with lane ^ 32
template <class M>
__device__ __forceinline__ void en_to_ops(const float (&v)[16], int h, v4u &h0, v4u &l0, v4u &h1, v4u &l1) {
    uint2 ghi[4], glo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float t[4] = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        split4<M>(t, ghi[q], glo[q]);
    }
    v4u oh[2], ol[2];
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const uint2 keep_hi = h ? ghi[2 * sl + 1] : ghi[2 * sl], keep_lo = h ? glo[2 * sl + 1] : glo[2 * sl];
        const uint2 send_hi = h ? ghi[2 * sl] : ghi[2 * sl + 1], send_lo = h ? glo[2 * sl] : glo[2 * sl + 1];
        uint2 recv_hi, recv_lo;
        recv_hi.x = (unsigned int)__shfl_xor((int)send_hi.x, 32, 64);
        recv_hi.y = (unsigned int)__shfl_xor((int)send_hi.y, 32, 64);
        recv_lo.x = (unsigned int)__shfl_xor((int)send_lo.x, 32, 64);
        recv_lo.y = (unsigned int)__shfl_xor((int)send_lo.y, 32, 64);
        oh[sl] = h ? v4u{recv_hi.x, recv_hi.y, keep_hi.x, keep_hi.y} : v4u{keep_hi.x, keep_hi.y, recv_hi.x, recv_hi.y};
        ol[sl] = h ? v4u{recv_lo.x, recv_lo.y, keep_lo.x, keep_lo.y} : v4u{keep_lo.x, keep_lo.y, recv_lo.x, recv_lo.y};
    }
    h0 = oh[0]; l0 = ol[0]; h1 = oh[1]; l1 = ol[1];
}

// identity fragment of k-step sl of a 32-channel block as an A operand: lane (channel m = lane & 31, h) holds k = 8 h .. 8 h + 7 =
// channels 16 sl + 8 h + j: one where that is m
template <class M>
__device__ __forceinline__ v4u en_identity(int sl, int l31, int h) {
    const int j = l31 - 16 * sl - 8 * h;
    unsigned int w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 4; ++k) w[k] = (j >> 1) == k ? (one16<M>() << (16 * (j & 1))) : 0u;
    if (j < 0 || j > 7) { w[0] = w[1] = w[2] = w[3] = 0u; }
    return v4u{w[0], w[1], w[2], w[3]};
}

// acc += rows that exist as operands (identity weights: exact up to the fp32 additions)
template <class M>
__device__ __forceinline__ f32x16 en_add_rows(f32x16 acc, v4u ident, v4u xh, v4u xl) {
    acc = M::mma(ident, xl, acc);
    return M::mma(ident, xh, acc);
}

// LayerNorm of a row spread over NF fragments of this lane and of lane ^ 32 (channels 32 ct + 8 q + 4 h + i <-> f[ct][4 q + i])
template <int NF>
__device__ __forceinline__ void en_layernorm(f32x16 (&f)[NF], const float *gamma, const float *beta, float eps, int h) {
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < NF; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += f[ct][e];
    s += __shfl_xor(s, 32, 64);
    const float mean = s / (float)(NF * 32);
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < NF; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = f[ct][e] - mean; q += d * d; }
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q / (float)(NF * 32) + eps);
#pragma unroll
    for (int ct = 0; ct < NF; ++ct)
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            const int c0 = ct * 32 + qq * 8 + h * 4;
            const float4 g4 = *reinterpret_cast<const float4 *>(gamma + c0), b4 = *reinterpret_cast<const float4 *>(beta + c0);
            f[ct][4 * qq] = (f[ct][4 * qq] - mean) * rstd * g4.x + b4.x;
            f[ct][4 * qq + 1] = (f[ct][4 * qq + 1] - mean) * rstd * g4.y + b4.y;
            f[ct][4 * qq + 2] = (f[ct][4 * qq + 2] - mean) * rstd * g4.z + b4.z;
            f[ct][4 * qq + 3] = (f[ct][4 * qq + 3] - mean) * rstd * g4.w + b4.w;
        }
}

// a fragment's accumulators set to a per-channel vector (bias) in the accumulator layout
__device__ __forceinline__ f32x16 en_bias(const float *b, int ct, int h) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 b4 = *reinterpret_cast<const float4 *>(b + ct * 32 + q * 8 + h * 4);
        r[4 * q] = b4.x; r[4 * q + 1] = b4.y; r[4 * q + 2] = b4.z; r[4 * q + 3] = b4.w;
    }
    return r;
}

template <class M>
__device__ __forceinline__ void en_split8(v4u p0, v4u p1, float scale, v4u &hi, v4u &lo) {      // 8 fp32 values (two loads) -> operand
    const float v0[4] = {__uint_as_float(p0.x) * scale, __uint_as_float(p0.y) * scale, __uint_as_float(p0.z) * scale, __uint_as_float(p0.w) * scale};
    const float v1[4] = {__uint_as_float(p1.x) * scale, __uint_as_float(p1.y) * scale, __uint_as_float(p1.z) * scale, __uint_as_float(p1.w) * scale};
    uint2 h0, l0, h1, l1;
    split4<M>(v0, h0, l0);
    split4<M>(v1, h1, l1);
    hi = v4u{h0.x, h0.y, h1.x, h1.y};
    lo = v4u{l0.x, l0.y, l1.x, l1.y};
}

// ================================================================================================ back half
struct EncBackArgs {
    const float *op, *src, *pooled;            // (rows, 192): attention output and layer input as pair16, pooled features fp32
    const unsigned char *row_skip;             // (rows): 1 = the row's RoI has no points: out = 2 * pooled
    const float *wo, *w1, *w2;                 // pair16 rows per output channel: (192, 192), (128, 192), (192, 128)
    const float *bo, *g1, *be1, *b1, *b2, *g2, *be2;
    float *out;                                // (rows, 192) fp32, or pair16 (out_pair16: the operand of the FC stack that follows)
    long rows;
    float eps1, eps2;
    int out_pair16;
};
constexpr int VB_BO = 0, VB_G1 = 192, VB_BE1 = 384, VB_B1 = 576, VB_B2 = 704, VB_G2 = 896, VB_BE2 = 1088;

// OP16: the result as pair16 rows (the FC stack's operand) instead of fp32 - a template parameter, not a run-time branch: with both
// store sequences in the code the six LayerNorm2 accumulators were spilled around the branch (r04: 400 bytes of scratch per lane)
template <class M, bool OP16>
__global__ __launch_bounds__(EN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_enc_back(EncBackArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float *const vec = reinterpret_cast<float *>(sm + EN_OFF_VEC);
    const int tid_ = threadIdx.x, lane = tid_ & 63, wid = __builtin_amdgcn_readfirstlane(tid_ >> 6), l31_ = lane & 31, h_ = lane >> 5;
    for (int i = tid_; i < EN_E; i += EN_THREADS) {
        vec[VB_BO + i] = a.bo[i]; vec[VB_G1 + i] = a.g1[i]; vec[VB_BE1 + i] = a.be1[i];
        vec[VB_B2 + i] = a.b2[i]; vec[VB_G2 + i] = a.g2[i]; vec[VB_BE2 + i] = a.be2[i];
    }
    for (int i = tid_; i < EN_F; i += EN_THREADS) vec[VB_B1 + i] = a.b1[i];
    const unsigned int tot = (unsigned int)(a.rows * EN_ROWB);
    const srsrc_t r_op = make_srsrc(a.op, tot), r_src = make_srsrc(a.src, tot), r_po = make_srsrc(a.pooled, tot), r_out = make_srsrc(a.out, tot);
    const srsrc_t r_sk = make_srsrc(a.row_skip, (unsigned int)a.rows);
    const srsrc_t r_wo = make_srsrc(a.wo, EN_E * EN_E * 4), r_w1 = make_srsrc(a.w1, EN_F * EN_E * 4), r_w2 = make_srsrc(a.w2, EN_E * EN_F * 4);
    // slice i of a tile (16): 0-5 = input-channel slabs of Wo, 6-9 = output-channel blocks of W1, 10-15 = of W2
    auto issue_slice_t = [&](int i, int buf, int tid) {
        if (i < 6) en_issue_slab(r_wo, i * 128, EN_ROWB, buf, tid, wid);
        else if (i < 10) en_issue_rows(r_w1, (i - 6) * 32, EN_E, buf, tid, wid);
        else en_issue_rows(r_w2, (i - 10) * 32, EN_F, buf, tid, wid);
    };
    __syncthreads();
    issue_slice_t(0, 0, tid_);
    issue_slice_t(1, 1, tid_);
    int ring = 0;                                        // buffer of the slice about to be consumed
    const long ntiles = (a.rows + EN_TILE - 1) / EN_TILE;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // (opaque copies: the dozens of per-thread LDS / DMA offsets derived from these are loop invariants the compiler would otherwise
        // compute once and keep - or spill - across the whole loop)
        int tid = tid_, l31 = l31_, h = h_;
        asm volatile("" : "+v"(tid), "+v"(l31), "+v"(h));
        auto issue_slice = [&](int i, int buf) { issue_slice_t(i, buf, tid); };
        const long row = tile * EN_TILE + wid * 32 + l31;
        const bool rok = row < a.rows;
        const unsigned int rowb = rok ? (unsigned int)(row * EN_ROWB) : OOB_OFFSET;
        // operands of slab t (k-steps 2 t, 2 t + 1): the attention output one slab ahead (two register sets), the layer input - needed
        // only for the residual at the slab's end - within the slab
        v4u ja[4], jb[4], sj[4];
        auto load_o = [&](v4u (&j)[4], int t) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const unsigned int off = rowb + (unsigned int)((4 * t + 2 * sl + h) * 32);
                en_ld16(j[2 * sl], r_op, off);
                en_ld16o(j[2 * sl + 1], r_op, off);
            }
        };
        auto load_s = [&](int t) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const unsigned int off = rowb + (unsigned int)((4 * t + 2 * sl + h) * 32);
                en_ld16(sj[2 * sl], r_src, off);
                en_ld16o(sj[2 * sl + 1], r_src, off);
            }
        };
        unsigned int skip = 0u;
        load_o(ja, 0);
        en_ld1(skip, r_sk, rok ? (unsigned int)row : OOB_OFFSET);
        EN_WAIT(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) en_pin(ja[i]);
        asm volatile("" : "+v"(skip));

        // ---- x = LayerNorm1(src + o' . Mvo + bvo): all 192 outputs accumulate while the input channels stream by
        f32x16 acc[6];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) acc[ct] = en_bias(vec + VB_BO, ct, h);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            if (t > 0) {
                EN_WAIT(3);                              // (slice t; o' of this slab landed with the layer input of the slab before)
                if (t & 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) en_pin(jb[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) en_pin(ja[i]);
                }
            }
            __syncthreads();
            if (t & 1) load_o(ja, t + 1 < 6 ? t + 1 : 0); else load_o(jb, t + 1 < 6 ? t + 1 : 0);      // (the last one is never used: fixed load count)
            load_s(t);
            issue_slice(t + 2, ring + 2 >= EN_RING ? ring + 2 - EN_RING : ring + 2);
            const int wb = EN_OFF_RING + ring * EN_SLICE;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const v4u oh = (t & 1) ? jb[2 * sl] : ja[2 * sl], ol = (t & 1) ? jb[2 * sl + 1] : ja[2 * sl + 1];
#pragma unroll
                for (int ct = 0; ct < 6; ++ct) {
                    v4u whi, wlo;
                    en_wfrag(sm, wb, EN_E, ct * 32, sl, l31, h, whi, wlo);
                    acc[ct] = M::mma(wlo, oh, acc[ct]);
                    acc[ct] = M::mma(whi, ol, acc[ct]);
                    acc[ct] = M::mma(whi, oh, acc[ct]);
                    if (ct & 1) __builtin_amdgcn_sched_barrier(0);          // (keeps the weight reads of later fragments from piling up in registers)
                }
            }
            EN_WAIT(3);                                  // the layer input of this slab (the slice issued after it stays in flight)
#pragma unroll
            for (int i = 0; i < 4; ++i) en_pin(sj[i]);
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) acc[t] = en_add_rows<M>(acc[t], en_identity<M>(sl, l31, h), sj[2 * sl], sj[2 * sl + 1]);
            __builtin_amdgcn_sched_barrier(0);
            ring = ring + 1 == EN_RING ? 0 : ring + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        en_layernorm<6>(acc, vec + VB_G1, vec + VB_BE1, a.eps1, h);
        __builtin_amdgcn_sched_barrier(0);
        v4u xh[12], xl[12];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = acc[ct][e];
            en_to_ops<M>(v, h, xh[2 * ct], xl[2 * ct], xh[2 * ct + 1], xl[2 * ct + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- hidden = ReLU(W1 . x + b1)
        asm volatile("" : "+v"(tid), "+v"(l31), "+v"(h));      // (per phase: the address temporaries of one phase must not stay live - spilled - through the next)
        v4u hh[8], hl[8];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            EN_WAIT(3);
            __syncthreads();
            issue_slice(6 + ct + 2, ring + 2 >= EN_RING ? ring + 2 - EN_RING : ring + 2);
            const int wb = EN_OFF_RING + ring * EN_SLICE;
            f32x16 d = en_bias(vec + VB_B1, ct, h);
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                v4u whi, wlo;
                en_wfrag(sm, wb, 32, 0, s, l31, h, whi, wlo);
                d = M::mma(wlo, xh[s], d);
                d = M::mma(whi, xl[s], d);
                d = M::mma(whi, xh[s], d);
                if (s & 1) __builtin_amdgcn_sched_barrier(0);
            }
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = fmaxf(d[e], 0.f);
            en_to_ops<M>(v, h, hh[2 * ct], hl[2 * ct], hh[2 * ct + 1], hl[2 * ct + 1]);
            __builtin_amdgcn_sched_barrier(0);
            ring = ring + 1 == EN_RING ? 0 : ring + 1;
        }
        // ---- y = LayerNorm2(x + W2 . hidden + b2)
        asm volatile("" : "+v"(tid), "+v"(l31), "+v"(h));
        v4u pa[4], pb[4];                                // pooled features of a fragment: two k-steps x two 16-byte loads (fp32)
        auto load_p = [&](v4u (&p)[4], int c) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const unsigned int off = rowb + (unsigned int)((2 * c + sl) * 64 + h * 32);
                en_ld16(p[2 * sl], r_po, off);
                en_ld16o(p[2 * sl + 1], r_po, off);
            }
        };
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
            EN_WAIT(3);
            __syncthreads();
            if (ct == 5) load_p(pa, 0);
            {   // the slices of the NEXT tile (the same 16 every tile)
                const int nx = 10 + ct + 2;
                issue_slice(nx >= 16 ? nx - 16 : nx, ring + 2 >= EN_RING ? ring + 2 - EN_RING : ring + 2);
            }
            const int wb = EN_OFF_RING + ring * EN_SLICE;
            f32x16 d = en_bias(vec + VB_B2, ct, h);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                v4u whi, wlo;
                en_wfrag(sm, wb, 32, 0, s, l31, h, whi, wlo);
                d = M::mma(wlo, hh[s], d);
                d = M::mma(whi, hl[s], d);
                d = M::mma(whi, hh[s], d);
                if (s & 1) __builtin_amdgcn_sched_barrier(0);
            }
            d = en_add_rows<M>(d, en_identity<M>(0, l31, h), xh[2 * ct], xl[2 * ct]);
            d = en_add_rows<M>(d, en_identity<M>(1, l31, h), xh[2 * ct + 1], xl[2 * ct + 1]);
            acc[ct] = d;
            __builtin_amdgcn_sched_barrier(0);
            ring = ring + 1 == EN_RING ? 0 : ring + 1;
        }
        __builtin_amdgcn_sched_barrier(0);
        en_layernorm<6>(acc, vec + VB_G2, vec + VB_BE2, a.eps2, h);
        __builtin_amdgcn_sched_barrier(0);
        // ---- out = pooled + (skip ? pooled : y), fragment by fragment: the pooled rows arrive as operands (split here) one fragment ahead
        asm volatile("" : "+v"(tid), "+v"(l31), "+v"(h));
        const bool sk = skip != 0u;
        const float pscale = sk ? 2.f : 1.f;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            if (c + 1 < 6) { if (c & 1) load_p(pa, c + 1); else load_p(pb, c + 1); }
            if (c == 0) EN_WAIT(7); else if (c < 5) EN_WAIT(8); else EN_WAIT(4);
            if (c & 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) en_pin(pb[i]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) en_pin(pa[i]);
            }
            f32x16 d;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = sk ? 0.f : acc[c][e];
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                v4u ph, pl;
                en_split8<M>((c & 1) ? pb[2 * sl] : pa[2 * sl], (c & 1) ? pb[2 * sl + 1] : pa[2 * sl + 1], pscale, ph, pl);
                d = en_add_rows<M>(d, en_identity<M>(sl, l31, h), ph, pl);
            }
            if constexpr (OP16) {                        // (four stores either way: the counted waits hold)
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = d[e];
                v4u oh0, ol0, oh1, ol1;
                en_to_ops<M>(v, h, oh0, ol0, oh1, ol1);
                const unsigned int off = rowb + (unsigned int)((4 * c + h) * 32);
                en_st16(oh0, r_out, off);
                en_st16o(ol0, r_out, off);
                en_st16(oh1, r_out, off + 64u);
                en_st16o(ol1, r_out, off + 64u);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    en_st16(v4u{__float_as_uint(d[4 * q]), __float_as_uint(d[4 * q + 1]), __float_as_uint(d[4 * q + 2]), __float_as_uint(d[4 * q + 3])}, r_out,
                            rowb + (unsigned int)((c * 32 + q * 8 + h * 4) * 4));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    EN_WAIT(0);
}

// ================================================================================================ front half
struct EncFrontArgs {
    const float *pos_in;                       // (rows, pin) fp32, pin in {4, 8}
    const float *feats;                        // (rows, 192) fp32: pooled features
    const unsigned char *row_add;              // (rows): 1 = the row gets its positional encoding
    const float *w0, *w1, *wq;                 // pair16 rows per output channel: (96, 16), (192, 96), (192, 192)
    const float *s0, *b0, *b1, *uq;            // folded BatchNorm of the first layer, biases
    float *src, *q;                            // (rows, 192) pair16
    long rows;
    int pin;
};
constexpr int VF_S0 = 0, VF_B0 = 96, VF_B1 = 192, VF_UQ = 384;

template <class M>
__global__ __launch_bounds__(EN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_enc_front(EncFrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    float *const vec = reinterpret_cast<float *>(sm + EN_OFF_VEC);
    const int tid_ = threadIdx.x, lane = tid_ & 63, wid = __builtin_amdgcn_readfirstlane(tid_ >> 6), l31_ = lane & 31, h_ = lane >> 5;
    for (int i = tid_; i < EN_P; i += EN_THREADS) { vec[VF_S0 + i] = a.s0[i]; vec[VF_B0 + i] = a.b0[i]; }
    for (int i = tid_; i < EN_E; i += EN_THREADS) { vec[VF_B1 + i] = a.b1[i]; vec[VF_UQ + i] = a.uq[i]; }
    for (int i = tid_; i < EN_P * 4; i += EN_THREADS)       // the first layer's weights stay: 96 rows x 64 bytes (one k-step), padded rows
        *reinterpret_cast<v4u *>(sm + EN_OFF_W0 + (i >> 2) * EN_W0_ROW + (i & 3) * 16) = reinterpret_cast<const v4u *>(a.w0)[i];
    const unsigned int tot = (unsigned int)(a.rows * EN_ROWB);
    const srsrc_t r_fe = make_srsrc(a.feats, tot), r_src = make_srsrc(a.src, tot), r_q = make_srsrc(a.q, tot);
    const srsrc_t r_pi = make_srsrc(a.pos_in, (unsigned int)(a.rows * a.pin * 4)), r_ad = make_srsrc(a.row_add, (unsigned int)a.rows);
    const srsrc_t r_w1 = make_srsrc(a.w1, EN_E * EN_P * 4), r_wq = make_srsrc(a.wq, EN_E * EN_E * 4);
    const v4u id0 = en_identity<M>(0, l31_, h_), id1 = en_identity<M>(1, l31_, h_);
    // slice i of a tile (12): 0-5 = output-channel blocks of W_p2 (96 inputs), 6-11 = of Mq (192 inputs)
    auto issue_slice_t = [&](int i, int buf, int tid) {
        if (i < 6) en_issue_rows(r_w1, i * 32, EN_P, buf, tid, wid);
        else en_issue_rows(r_wq, (i - 6) * 32, EN_E, buf, tid, wid);
    };
    __syncthreads();
    issue_slice_t(0, 0, tid_);
    issue_slice_t(1, 1, tid_);
    int ring = 0;
    const long ntiles = (a.rows + EN_TILE - 1) / EN_TILE;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int tid = tid_, l31 = l31_, h = h_;              // (opaque copies: see k_enc_back)
        asm volatile("" : "+v"(tid), "+v"(l31), "+v"(h));
        auto issue_slice = [&](int i, int buf) { issue_slice_t(i, buf, tid); };
        const long row = tile * EN_TILE + wid * 32 + l31;
        const bool rok = row < a.rows;
        const unsigned int rowb = rok ? (unsigned int)(row * EN_ROWB) : OOB_OFFSET;
        v4u fa[4], fb[4];                                 // features of a fragment: two k-steps x two 16-byte loads (fp32)
        auto load_f = [&](v4u (&p)[4], int c) {
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                const unsigned int off = rowb + (unsigned int)((2 * c + sl) * 64 + h * 32);
                en_ld16(p[2 * sl], r_fe, off);
                en_ld16o(p[2 * sl + 1], r_fe, off);
            }
        };
        // ---- positional input: channels 0 .. pin - 1 of the one k-step (half-wave 0; the rest of the 16 are zeros)
        v4u pi0 = v4u{0u, 0u, 0u, 0u}, pi1 = v4u{0u, 0u, 0u, 0u};
        unsigned int add = 0u;
        {
            const unsigned int pb = (rok && h == 0) ? (unsigned int)(row * a.pin * 4) : OOB_OFFSET;
            en_ld16(pi0, r_pi, pb);
            en_ld16o(pi1, r_pi, a.pin > 4 ? pb : OOB_OFFSET);
            en_ld1(add, r_ad, rok ? (unsigned int)row : OOB_OFFSET);
            load_f(fa, 0);
        }
        EN_WAIT(0);
        en_pin(pi0); en_pin(pi1);
        asm volatile("" : "+v"(add));
#pragma unroll
        for (int i = 0; i < 4; ++i) en_pin(fa[i]);
        v4u xh0, xl0;
        en_split8<M>(pi0, pi1, 1.f, xh0, xl0);
        // ---- hidden = ReLU(BN(W_p1 . positional input)): resident weights
        v4u ph[6], pl[6];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) {
            const unsigned char *wp = sm + EN_OFF_W0 + (ct * 32 + l31) * EN_W0_ROW + h * 32;
            const v4u whi = *reinterpret_cast<const v4u *>(wp), wlo = *reinterpret_cast<const v4u *>(wp + 16);
            f32x16 d;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = 0.f;
            d = M::mma(wlo, xh0, d);
            d = M::mma(whi, xl0, d);
            d = M::mma(whi, xh0, d);
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = ct * 32 + q * 8 + h * 4;
                const float4 s4 = *reinterpret_cast<const float4 *>(vec + VF_S0 + c0), b4 = *reinterpret_cast<const float4 *>(vec + VF_B0 + c0);
                v[4 * q] = fmaxf(fmaf(d[4 * q], s4.x, b4.x), 0.f);
                v[4 * q + 1] = fmaxf(fmaf(d[4 * q + 1], s4.y, b4.y), 0.f);
                v[4 * q + 2] = fmaxf(fmaf(d[4 * q + 2], s4.z, b4.z), 0.f);
                v[4 * q + 3] = fmaxf(fmaf(d[4 * q + 3], s4.w, b4.w), 0.f);
            }
            en_to_ops<M>(v, h, ph[2 * ct], pl[2 * ct], ph[2 * ct + 1], pl[2 * ct + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- src = features + (add ? W_p2 . hidden + b : 0) -> operands of the query GEMM, and pair16 rows
        const bool ad = add != 0u;
        v4u sh[12], sl_[12];
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
            if (ct > 0) {
                EN_WAIT(7);
                if (ct & 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) en_pin(fb[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) en_pin(fa[i]);
                }
            }
            __syncthreads();
            if (ct + 1 < 6) { if (ct & 1) load_f(fa, ct + 1); else load_f(fb, ct + 1); }
            issue_slice(ct + 2, ring + 2 >= EN_RING ? ring + 2 - EN_RING : ring + 2);
            const int wb = EN_OFF_RING + ring * EN_SLICE;
            f32x16 d = en_bias(vec + VF_B1, ct, h);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                v4u whi, wlo;
                en_wfrag(sm, wb, 32, 0, s, l31, h, whi, wlo);
                d = M::mma(wlo, ph[s], d);
                d = M::mma(whi, pl[s], d);
                d = M::mma(whi, ph[s], d);
                if (s & 1) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = ad ? d[e] : 0.f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                v4u fh, fl;
                en_split8<M>((ct & 1) ? fb[2 * k] : fa[2 * k], (ct & 1) ? fb[2 * k + 1] : fa[2 * k + 1], 1.f, fh, fl);
                d = en_add_rows<M>(d, k ? id1 : id0, fh, fl);
            }
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = d[e];
            en_to_ops<M>(v, h, sh[2 * ct], sl_[2 * ct], sh[2 * ct + 1], sl_[2 * ct + 1]);
#pragma unroll
            for (int k = 0; k < 2; ++k) {               // group 2 (2 ct + k) + h of my row
                const unsigned int off = rowb + (unsigned int)((4 * ct + 2 * k + h) * 32);
                en_st16(sh[2 * ct + k], r_src, off);
                en_st16o(sl_[2 * ct + k], r_src, off);
            }
            __builtin_amdgcn_sched_barrier(0);
            ring = ring + 1 == EN_RING ? 0 : ring + 1;
        }
        // ---- q' = src . Mq + uq
#pragma unroll
        for (int ct = 0; ct < 6; ++ct) {
            EN_WAIT(7);
            __syncthreads();
            {
                const int nx = 6 + ct + 2;
                issue_slice(nx >= 12 ? nx - 12 : nx, ring + 2 >= EN_RING ? ring + 2 - EN_RING : ring + 2);
            }
            const int wb = EN_OFF_RING + ring * EN_SLICE;
            f32x16 d = en_bias(vec + VF_UQ, ct, h);
#pragma unroll
            for (int s = 0; s < 12; ++s) {
                v4u whi, wlo;
                en_wfrag(sm, wb, 32, 0, s, l31, h, whi, wlo);
                d = M::mma(wlo, sh[s], d);
                d = M::mma(whi, sl_[s], d);
                d = M::mma(whi, sh[s], d);
                if (s & 1) __builtin_amdgcn_sched_barrier(0);
            }
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = d[e];
            v4u qh0, ql0, qh1, ql1;
            en_to_ops<M>(v, h, qh0, ql0, qh1, ql1);
            const unsigned int off = rowb + (unsigned int)((4 * ct + h) * 32);
            en_st16(qh0, r_q, off);
            en_st16o(ql0, r_q, off);
            en_st16(qh1, r_q, off + 64u);
            en_st16o(ql1, r_q, off + 64u);
            __builtin_amdgcn_sched_barrier(0);
            ring = ring + 1 == EN_RING ? 0 : ring + 1;
        }
    }
    EN_WAIT(0);
}

template <class K, class A>
int en_launch(K kernel, const A &a, long rows, const char *who, PerDeviceFlags &done, hipStream_t stream) {
    if (int rc = reserve_lds(reinterpret_cast<const void *>(kernel), EN_LDS, done, who)) return rc;
    const long ntiles = (rows + EN_TILE - 1) / EN_TILE;
    int grid = device_cus();
    if (grid > ntiles) grid = (int)ntiles;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(EN_THREADS), EN_LDS, stream, a);
    return DZ_OK;
}

}  // namespace
}  // namespace dz

using namespace dz;

extern "C" {

// src = feats + (row_add ? W_p2 . ReLU(s0 * (W_p1 . pos_in) + b0) + b1 : 0) and q = src . Mq + uq, both (rows, 192) pair16.
// pos_in (rows, pin) fp32 with pin 4 or 8 (zero-padded to the 16 inputs of w0); w0 (96, 16), w1 (192, 96), wq (192, 192): pair16 rows per
// OUTPUT channel; feats (rows, 192) fp32; row_add (rows) bytes.
int dz_pdv_encoder_front(const float *pos_in, int pin, const float *feats, const unsigned char *row_add, long rows, const float *w0, const float *s0,
                         const float *b0, const float *w1, const float *b1, const float *wq, const float *uq, float *src, float *q, int math,
                         void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0 && (pin == 4 || pin == 8), "dz_pdv_encoder_front: pin 4 or 8 (got %d)", pin);
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_pdv_encoder_front: math %d is not a split mode", math);
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(pos_in && feats && row_add && w0 && s0 && b0 && w1 && b1 && wq && uq && src && q, "dz_pdv_encoder_front: null pointer");
    if ((size_t)rows * EN_ROWB >= 0x80000000ull) { set_error("dz_pdv_encoder_front: %ld rows exceed the 2 GiB buffer-addressing limit", rows); return DZ_ERR_UNSUPPORTED; }
    const EncFrontArgs a{pos_in, feats, row_add, w0, w1, wq, s0, b0, b1, uq, src, q, rows, pin};
    int rc;
    if (math == DZ_MATH_F16X2) { static PerDeviceFlags done; rc = en_launch(&k_enc_front<MathF16>, a, rows, "dz_pdv_encoder_front", done, stream); }
    else { static PerDeviceFlags done; rc = en_launch(&k_enc_front<MathBF16>, a, rows, "dz_pdv_encoder_front", done, stream); }
    if (rc) return rc;
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// out = pooled + (row_skip ? pooled : LN2(x + W2 . ReLU(W1 . x + b1) + b2)) with x = LN1(src + op . Wo + bo); op, src (rows, 192) pair16,
// pooled (rows, 192) fp32, out (rows, 192) fp32 or - out_pair16 - pair16; wo (192, 192), w1 (128, 192), w2 (192, 128) pair16 rows per OUTPUT channel.
int dz_pdv_encoder_back(const float *op, const float *src, const float *pooled, const unsigned char *row_skip, long rows, const float *wo,
                        const float *bo, const float *g1, const float *be1, float eps1, const float *w1, const float *b1, const float *w2,
                        const float *b2, const float *g2, const float *be2, float eps2, float *out, int out_pair16, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0, "dz_pdv_encoder_back: negative rows");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_pdv_encoder_back: math %d is not a split mode", math);
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(op && src && pooled && row_skip && wo && bo && g1 && be1 && w1 && b1 && w2 && b2 && g2 && be2 && out, "dz_pdv_encoder_back: null pointer");
    if ((size_t)rows * EN_ROWB >= 0x80000000ull) { set_error("dz_pdv_encoder_back: %ld rows exceed the 2 GiB buffer-addressing limit", rows); return DZ_ERR_UNSUPPORTED; }
    const EncBackArgs a{op, src, pooled, row_skip, wo, w1, w2, bo, g1, be1, b1, b2, g2, be2, out, rows, eps1, eps2, out_pair16 ? 1 : 0};
    int rc;
    if (math == DZ_MATH_F16X2) {
        if (a.out_pair16) { static PerDeviceFlags done; rc = en_launch(&k_enc_back<MathF16, true>, a, rows, "dz_pdv_encoder_back", done, stream); }
        else { static PerDeviceFlags done; rc = en_launch(&k_enc_back<MathF16, false>, a, rows, "dz_pdv_encoder_back", done, stream); }
    } else {
        if (a.out_pair16) { static PerDeviceFlags done; rc = en_launch(&k_enc_back<MathBF16, true>, a, rows, "dz_pdv_encoder_back", done, stream); }
        else { static PerDeviceFlags done; rc = en_launch(&k_enc_back<MathBF16, false>, a, rows, "dz_pdv_encoder_back", done, stream); }
    }
    if (rc) return rc;
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
