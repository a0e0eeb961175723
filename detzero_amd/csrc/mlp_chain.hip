// The memory branch of the refiner behind its PointNet encoder, in ONE kernel on pair16 operands:
//     h   = ReLU(BN_a(W_a . tap + per-object addend))            128 -> 512     (memory_mlp[0]; the addend is W_pool . max-pooled features)
//     mem = ReLU(BN_b(W_b . h))                                  512 -> 256     (memory_mlp[3])                      -> fp32 rows
//     K   = W_k . mem + b_k,  V = W_v . mem + b_v                256 -> 256     (decoder cross-attention in_proj)    -> fp32 rows (optional)
// Reference: refining/detzero_refine/models/modules/geometry_transformer.py:56-67,126-133 and position_transformer.py:60-72,108-117
// (memory_mlp on cat([intermediate, pooled])), transformer/multi_head_attention.py:199-236 (k / v projections of the memory).
//
// Layer by layer (dz_linear_forward_split) a chunk of 96 PRM tracks moves 921 600 rows through these four GEMMs: the 512-wide hidden
// tensor alone is 3.8 GB of HBM round trip, the memory is converted to pair16 and read twice more for K and V.  Here a WAVE owns 32
// rows and carries them through the whole chain in registers (the operand-chaining of pointnet.hip: BatchNorm + ReLU + (hi, lo) split
// on the accumulator, one exchange with lane ^ 32, and the 16 bytes hi | 16 bytes lo a lane then holds ARE its MFMA operand of the
// next layer):
//   * the 512-wide hidden layer never exists as a whole: it is produced in four slices of 128 channels, and each slice is contracted
//     into the 256 accumulators of the next layer at once (k ascending, so the sums are those of the layered path, bit for bit);
//   * one wave per SIMD with the whole register file (128 fp32 accumulators of the 256-wide layer + 64 of the slice + operands);
//   * weights stream through LDS in 32 KB slices (64 output x 128 input channels) through a ring of four buffers, THREE slices in flight
//     (every workgroup of the launch walks the same 1.3 MB of weights in step: with one slice of look-ahead a third of the time went
//     into waiting for L2), loaded with `buffer_load_dwordx4 ... lds` (no staging registers) into an XOR-swizzled layout that is
//     conflict-free for the fragment reads; 40 slices per 128 rows (24 without K / V), one workgroup barrier per slice = per 48 MFMAs;
//   * fp32 results leave straight from the accumulator layout: lane (row, half) stores the 16 bytes it holds of each 8-channel group, the
//     two half-wave lanes write adjacent pieces and four stores complete a 128-byte line (round 5; DZ_TUNE_CHAIN_DIRECT=0 = the earlier
//     transposition through a wave-private LDS window, 4 % slower).  The window still carries the tile's per-object addends.
// HBM traffic: the tap rows in (512 B each), memory / K / V rows out (1 KB each) - nothing else.
#include <stdlib.h>

#include <type_traits>

#include "hgemm.h"

#ifndef DZ_CHAIN_DIAG
#define DZ_CHAIN_DIAG 0        // timing experiments only (results are garbage): 1 = no workgroup barriers, 2 = no weight loads, 8 = no vmcnt waits,
#endif                          // 16 = weight fragments from registers instead of LDS, 32 = no MFMAs, 64 = no result stores

namespace dz {
namespace {

constexpr int MC_THREADS = 256, MC_WAVES = 4;
constexpr int MC_SLICE = 64 * 128 * 4, MC_NBUF = 4;        // one weight slice: 64 rows x 128 channels of pair16; ring of four, three in flight
constexpr int MC_WIN_ROW = 144, MC_WIN = 32 * MC_WIN_ROW;  // fp32 window of a 32 x 32 fragment
constexpr int MC_OFF_W = 0, MC_OFF_WIN = MC_NBUF * MC_SLICE, MC_OFF_V = MC_OFF_WIN + MC_WAVES * MC_WIN;
constexpr int MC_C1 = 512, MC_C2 = 256, MC_CIN = 128;
constexpr int MC_NVEC = 2 * MC_C1 + 2 * MC_C2 + 2 * MC_C2;            // sA bA | sB bB | bK bV
constexpr int MC_LDS = MC_OFF_V + MC_NVEC * 4;
static_assert(MC_LDS <= 160 * 1024, "LDS");

struct ChainArgs {
    const float *x;                          // (rows, 128) pair16: the tapped encoder layer
    const float *wa, *wb, *wk, *wv;          // (512, 128), (256, 512), (256, 256) x 2 pair16, rows = output channel
    const float *sa, *ba, *sb, *bb, *bk, *bv;
    const float *gshift;                     // (rows / group_rows, ldg) fp32 pre-BatchNorm addend of layer a
    float *mem, *k, *v;                      // (rows, 256) fp32
    long rows;
    int group_rows, ldg;
    unsigned int x_bytes;
    int direct;                              // result rows straight from the accumulator layout (0 = through the wave's LDS window)
};

__device__ __forceinline__ void mc_load16_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(b), "v"(voff), "s"(rsrc) : "memory", "m0");
}

template <class M>
__device__ __forceinline__ f32x16 mc_mma(v4u a, v4u b, f32x16 c) {
    if (DZ_CHAIN_DIAG & 32) { c[0] += __uint_as_float(a.x ^ b.x); return c; }
    return M::mma(a, b, c);
}

template <class M, bool KV>
__global__ __launch_bounds__(MC_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_mlp_chain(ChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // (an opaque base offset: the 8 KB of vectors sit above the 64 KB the LDS instructions' offset field reaches from address 0, and
    // the compiler would otherwise materialise every one of their ~60 constant addresses in a register of its own)
    unsigned int vec_off = MC_OFF_V;
    asm volatile("" : "+v"(vec_off));
    float *const vec = reinterpret_cast<float *>(smem_raw + vec_off);
    float *const sA = vec, *const bA = vec + MC_C1, *const sB = vec + 2 * MC_C1, *const bB = sB + MC_C2, *const bK = bB + MC_C2, *const bV = bK + MC_C2;
    for (int i = tid; i < MC_C1; i += MC_THREADS) { sA[i] = a.sa[i]; bA[i] = a.ba[i]; }
    for (int i = tid; i < MC_C2; i += MC_THREADS) {
        sB[i] = a.sb[i]; bB[i] = a.bb[i];
        bK[i] = KV ? a.bk[i] : 0.f; bV[i] = KV ? a.bv[i] : 0.f;
    }
    const srsrc_t xrsrc = make_srsrc(a.x, a.x_bytes);

    // ---- the slice stream of a tile: j = 0 .. NSL-1, 64 output rows x 128 input channels each
    //   j = 6 sa + hf (hf = 0, 1): W_a rows [128 sa + 64 hf, +64), all 128 input channels
    //   j = 6 sa + 2 + qd (qd = 0 .. 3): W_b rows [64 qd, +64), input channels [128 sa, +128)
    //   j = 24 + 8 pj + 2 qd + kk: W_k (pj = 0) / W_v (pj = 1) rows [64 qd, +64), input channels [128 kk, +128)
    constexpr int NSL = KV ? 40 : 24;
    static_assert(NSL % MC_NBUF == 0, "slice j of every tile lives in buffer j % 4");
    // unit u = i * 256 + tid of a slice = 16 bytes at LDS offset u * 16: chunk (32 channels) u >> 9, row (u >> 3) & 63, stored piece u & 7
    // holds source piece (u & 7) ^ ((row >> 1) & 7)
    auto issue_slice = [&](int j, int buf) {
        // (the slice index may be a run-time value: the descriptor is built from scalar selects, never indexed from memory)
        const float *wbase;
        unsigned int n0, k0, rowb, wbytes;
        if (j < 24) {
            const int sa = j / 6, r = j % 6;
            wbase = r < 2 ? a.wa : a.wb;
            wbytes = r < 2 ? MC_C1 * MC_CIN * 4 : MC_C2 * MC_C1 * 4;
            n0 = r < 2 ? sa * 128 + r * 64 : (r - 2) * 64;
            k0 = r < 2 ? 0 : sa * 128;
            rowb = r < 2 ? MC_CIN * 4 : MC_C1 * 4;
        } else {
            const int jj = j - 24, r = jj & 7;
            wbase = jj < 8 ? a.wk : a.wv;
            wbytes = MC_C2 * MC_C2 * 4;
            n0 = (r >> 1) * 64; k0 = (r & 1) * 128; rowb = MC_C2 * 4;
        }
        const srsrc_t rs = make_srsrc(wbase, wbytes);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int chunk = i >> 1, n = (i & 1) * 32 + (tid >> 3), pc = (tid & 7) ^ ((n >> 1) & 7);
            mc_load16_lds((unsigned int)(MC_OFF_W + buf * MC_SLICE + (i * MC_THREADS + wid * 64) * 16),
                          (n0 + n) * rowb + (k0 + chunk * 32) * 4 + pc * 16, rs);
        }
    };
    // weight fragment: k-step s (16 channels: chunk s >> 1, pieces 4 (s & 1) + 2 h, + 1) of rows [r0, r0 + 32) of the slice in `buf`.
    // Rows r0 + l31 with r0 a multiple of 32 share the swizzle of l31, so a fragment address is one of 8 lane-dependent bases (buffer
    // pair x (s & 1) x hi / lo) plus a compile-time offset below 64 KB; the bases are kept opaque, or the compiler materialises (and
    // spills) a register per (buffer, chunk, fragment) instead of using the offset field of ds_read_b128
    unsigned int wb[2][2][2];                            // (32-bit LDS offsets: a laundered POINTER loses its address space)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                const int sw = (l31 >> 1) & 7, pc = par * 4 + h * 2 + hl;
                wb[b][par][hl] = (unsigned int)(MC_OFF_W + b * 2 * MC_SLICE + (l31 << 7) + ((pc ^ sw) << 4));
                asm volatile("" : "+v"(wb[b][par][hl]));
            }
    auto wfrag = [&](int buf, int r0, int s, v4u &hi, v4u &lo) {
        if (DZ_CHAIN_DIAG & 16) { hi = v4u{0x3c003c00u + (unsigned)s, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u + (unsigned)r0}; lo = hi; return; }
        const int off = (buf & 1) * MC_SLICE + (((s >> 1) * 64 + r0) << 7);
        hi = *reinterpret_cast<const v4u *>(smem_raw + wb[buf >> 1][s & 1][0] + off);
        lo = *reinterpret_cast<const v4u *>(smem_raw + wb[buf >> 1][s & 1][1] + off);
    };
    // Slice j has landed in every wave's view once the wave's own loads for it are back and the workgroup has met.  Its loads went out
    // three slices ago; the two slices issued since stay in flight (8 loads per thread each; loads return in order) - unless result
    // stores were issued in between, which retire on their own schedule: then everything is drained (`drain`).  The buffer of slice
    // j - 1 is free after the barrier: slice j + 3 goes into it.
    auto begin_slice = [&](int j, bool drain) {
        if (!(DZ_CHAIN_DIAG & 8)) {
            if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        if (!(DZ_CHAIN_DIAG & 1)) __syncthreads();
        if (!(DZ_CHAIN_DIAG & 2)) issue_slice((j + 3) % NSL, (j + 3) & 3);
        return j & 3;
    };

    // accumulator fragment (32 channels x 32 rows) -> BatchNorm (+ addend) + optional ReLU; lane (l31, h): acc[4 q + e] = channel 8 q + 4 h + e
    auto finish = [&](const f32x16 &acc, const float *sc, const float *sh, const float *add, bool relu, float (&v)[16]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 s4 = sc ? *reinterpret_cast<const float4 *>(sc + q * 8 + h * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 b4 = *reinterpret_cast<const float4 *>(sh + q * 8 + h * 4);
            float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (add) g4 = *reinterpret_cast<const float4 *>(add + q * 8 + h * 4);
            v[4 * q] = fmaf(acc[4 * q] + g4.x, s4.x, b4.x);
            v[4 * q + 1] = fmaf(acc[4 * q + 1] + g4.y, s4.y, b4.y);
            v[4 * q + 2] = fmaf(acc[4 * q + 2] + g4.z, s4.z, b4.z);
            v[4 * q + 3] = fmaf(acc[4 * q + 3] + g4.w, s4.w, b4.w);
            if (relu) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = fmaxf(v[4 * q + e], 0.f);
            }
        }
    };
    // 32 finished channels of my row -> the two k-steps (hi, lo operands) they form: split + exchange of the half groups with lane ^ 32
    auto to_operands = [&](const float (&v)[16], v4u &h0, v4u &l0, v4u &h1, v4u &l1) {
        // (plain scalars, no arrays: a lane-dependent choice between two array elements would be lowered to an indexed stack slot)
        uint2 gh0, gl0, gh1, gl1, gh2, gl2, gh3, gl3;
        {
            const float t0[4] = {v[0], v[1], v[2], v[3]}, t1[4] = {v[4], v[5], v[6], v[7]}, t2[4] = {v[8], v[9], v[10], v[11]}, t3[4] = {v[12], v[13], v[14], v[15]};
            split4<M>(t0, gh0, gl0);
            split4<M>(t1, gh1, gl1);
            split4<M>(t2, gh2, gl2);
            split4<M>(t3, gh3, gl3);
        }
        auto pick = [&](unsigned int a0, unsigned int a1) { return h ? a1 : a0; };
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            const uint2 e_hi = sl ? gh2 : gh0, o_hi = sl ? gh3 : gh1, e_lo = sl ? gl2 : gl0, o_lo = sl ? gl3 : gl1;   // groups 2 sl, 2 sl + 1
            const uint2 keep_hi = {pick(e_hi.x, o_hi.x), pick(e_hi.y, o_hi.y)}, keep_lo = {pick(e_lo.x, o_lo.x), pick(e_lo.y, o_lo.y)};
            const uint2 send_hi = {pick(o_hi.x, e_hi.x), pick(o_hi.y, e_hi.y)}, send_lo = {pick(o_lo.x, e_lo.x), pick(o_lo.y, e_lo.y)};
            uint2 rh, rl;
            rh.x = (unsigned int)__shfl_xor((int)send_hi.x, 32, 64);
            rh.y = (unsigned int)__shfl_xor((int)send_hi.y, 32, 64);
            rl.x = (unsigned int)__shfl_xor((int)send_lo.x, 32, 64);
            rl.y = (unsigned int)__shfl_xor((int)send_lo.y, 32, 64);
            const v4u oh = {pick(keep_hi.x, rh.x), pick(keep_hi.y, rh.y), pick(rh.x, keep_hi.x), pick(rh.y, keep_hi.y)};
            const v4u ol = {pick(keep_lo.x, rl.x), pick(keep_lo.y, rl.y), pick(rl.x, keep_lo.x), pick(rl.y, keep_lo.y)};
            if (sl == 0) { h0 = oh; l0 = ol; } else { h1 = oh; l1 = ol; }
        }
    };
    // 32 finished channels x 32 rows -> global fp32 rows through the wave's window: lane (row = lane >> 1, half) stores 64 contiguous bytes.
    // Buffer stores with a 32-bit offset (rows past the end carry the out-of-range offset: dropped by the hardware, no branches)
    unsigned char *const win = smem_raw + MC_OFF_WIN + wid * MC_WIN;
    auto store_f32 = [&](const float (&v)[16], __amdgpu_buffer_rsrc_t rs, unsigned int off, unsigned int doff) {
        if (a.direct) {
            // lane (row l31, half h) holds channels 8 q + 4 h .. + 3 of its row: four 16-byte stores; lanes h = 0 / 1 write adjacent pieces
            // and the four q complete the row's 128 bytes (one L2 line) back to back - as many store instructions as the window route,
            // none of its 8 LDS accesses and 2 wave barriers per fragment
            if (DZ_CHAIN_DIAG & 64) return;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(v4u{__float_as_uint(v[4 * q]), __float_as_uint(v[4 * q + 1]), __float_as_uint(v[4 * q + 2]), __float_as_uint(v[4 * q + 3])},
                                                       rs, (int)doff + q * 32, 0, 0);
            return;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4 *>(win + l31 * MC_WIN_ROW + (q * 8 + h * 4) * 4) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const v4u *src = reinterpret_cast<const v4u *>(win + (lane >> 1) * MC_WIN_ROW + (lane & 1) * 64);
        const v4u t0 = src[0], t1 = src[1], t2 = src[2], t3 = src[3];
        if (DZ_CHAIN_DIAG & 64) return;
        __builtin_amdgcn_raw_buffer_store_b128(t0, rs, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(t1, rs, (int)off + 16, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(t2, rs, (int)off + 32, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128(t3, rs, (int)off + 48, 0, 0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const unsigned int out_bytes = (unsigned int)((size_t)a.rows * MC_C2 * 4);
    const __amdgpu_buffer_rsrc_t rsMem = make_rsrc(a.mem, out_bytes);

    const long ntiles = (a.rows + 31) / 32;
    const long nwaves = (long)gridDim.x * MC_WAVES;
    const long per = (ntiles + nwaves - 1) / nwaves;
    const long t_begin = ((long)blockIdx.x * MC_WAVES + wid) * per;
    __syncthreads();
    issue_slice(0, 0);
    issue_slice(1, 1);
    issue_slice(2, 2);

    for (long it = 0; it < per; ++it) {
        const long tile = t_begin + it, row0 = tile * 32;
        const bool live = tile < ntiles;                                     // wave-uniform
        const long row = row0 + l31;
        const bool rok = live && row < a.rows;
        // ---- tap rows: k-step s takes the 8-channel groups 2 s + h (32 contiguous bytes)
        v4u xh[8], xl[8];
        {
            const unsigned int off = rok ? (unsigned int)(row * (MC_CIN * 4)) + (unsigned int)(h * 32) : OOB_OFFSET;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(xh[s]) : "v"(off + (unsigned int)(s * 64)), "s"(xrsrc));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(xl[s]) : "v"(off + (unsigned int)(s * 64)), "s"(xrsrc));
            }
        }
        // fp32 output rows: lane (row0 + lane / 2, 64-byte half lane & 1); out of range = dropped (outputs are < 2 GiB, host-checked)
        const unsigned int rowoff = (live && row0 + (lane >> 1) < a.rows) ? (unsigned int)((row0 + (lane >> 1)) * (MC_C2 * 4)) + (unsigned int)((lane & 1) * 64) : OOB_OFFSET;
        const unsigned int rowoff_d = rok ? (unsigned int)(row * (MC_C2 * 4)) + (unsigned int)(h * 16) : OOB_OFFSET;
        // the tile's row of per-object addends (512 floats) rides into the wave's window (idle until the result stores): read from
        // global inside the slice stream, every load of it would make the compiler wait for the weight slices in flight behind it
        const bool has_gs = a.gshift != nullptr;
        float4 g0 = make_float4(0.f, 0.f, 0.f, 0.f), g1 = g0;
        if (has_gs && live) {
            const float4 *gp = reinterpret_cast<const float4 *>(a.gshift + (size_t)(row0 / a.group_rows) * a.ldg);
            g0 = gp[lane]; g1 = gp[64 + lane];
        }
        const float *gs = has_gs ? reinterpret_cast<const float *>(win) : nullptr;
        f32x16 accB[8];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct)
#pragma unroll
            for (int e = 0; e < 16; ++e) accB[ct][e] = 0.f;
#pragma unroll 1                     // (fully unrolled, the compiler hoists the four slices' addend loads to the top of the tile: 300 spilled registers)
        for (int sa2 = 0; sa2 < 2; ++sa2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {                                    // (two slices of the hidden layer per trip: buffer indices stay compile-time)
                const int sa = 2 * sa2 + u;
                // ---- 128 channels of the hidden layer, 64 per slice
                f32x16 accA[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) accA[ct][e] = 0.f;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    // (the first slice of a tile follows the previous tile's result stores and my row loads: drain)
                    const int buf = (6 * u + hf) & 3;
                    begin_slice(12 * sa2 + 6 * u + hf, sa == 0 && hf == 0);
                    if (u == 0 && hf == 0 && sa2 == 0) {
#pragma unroll
                        for (int s = 0; s < 8; ++s) asm volatile("" : "+v"(xh[s]), "+v"(xl[s]));
                        if (has_gs) {
                            reinterpret_cast<float4 *>(win)[lane] = g0;
                            reinterpret_cast<float4 *>(win)[64 + lane] = g1;
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        // (term-major over the two fragments: with one wave per SIMD a dependent MFMA issued back to back waits for
                        // its predecessor; every accumulator still receives lo.hi, hi.lo, hi.hi in that order)
                        v4u whi[2], wlo[2];
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) wfrag(buf, ct * 32, s, whi[ct], wlo[ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accA[hf * 2 + ct] = mc_mma<M>(wlo[ct], xh[s], accA[hf * 2 + ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accA[hf * 2 + ct] = mc_mma<M>(whi[ct], xl[s], accA[hf * 2 + ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accA[hf * 2 + ct] = mc_mma<M>(whi[ct], xh[s], accA[hf * 2 + ct]);
                    }
                }
                v4u sh_[8], sl_[8];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    float v[16];
                    const int c0 = sa * 128 + ct * 32;
                    finish(accA[ct], sA + c0, bA + c0, gs ? gs + c0 : nullptr, true, v);
                    to_operands(v, sh_[2 * ct], sl_[2 * ct], sh_[2 * ct + 1], sl_[2 * ct + 1]);
                }
                // ---- their contribution to the 256 channels of the second layer, 64 output channels per slice
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int buf = (6 * u + 2 + qd) & 3;
                    begin_slice(12 * sa2 + 6 * u + 2 + qd, false);
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        v4u whi[2], wlo[2];
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) wfrag(buf, ct * 32, s, whi[ct], wlo[ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accB[qd * 2 + ct] = mc_mma<M>(wlo[ct], sh_[s], accB[qd * 2 + ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accB[qd * 2 + ct] = mc_mma<M>(whi[ct], sl_[s], accB[qd * 2 + ct]);
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) accB[qd * 2 + ct] = mc_mma<M>(whi[ct], sh_[s], accB[qd * 2 + ct]);
                    }
                }
            }
        }
        // ---- memory rows: fp32 out, operands of the projections
        v4u mh[16], ml[16];
#pragma unroll
        for (int ct = 0; ct < 8; ++ct) {
            float v[16];
            finish(accB[ct], sB + ct * 32, bB + ct * 32, nullptr, true, v);
            store_f32(v, rsMem, rowoff + ct * 128, rowoff_d + ct * 128);
            if (KV) to_operands(v, mh[2 * ct], ml[2 * ct], mh[2 * ct + 1], ml[2 * ct + 1]);
        }
        if constexpr (KV) {
#pragma unroll 1
            for (int pj = 0; pj < 2; ++pj) {                                 // K, then V
                const __amdgpu_buffer_rsrc_t rsOut = make_rsrc(pj ? a.v : a.k, out_bytes);
                f32x16 acc[8];
#pragma unroll
                for (int ct = 0; ct < 8; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk) {
                        // (the first slice of K / V follows the stores of the memory / K rows: drain)
                        const int buf = (2 * qd + kk) & 3;
                        begin_slice(24 + 8 * pj + 2 * qd + kk, qd == 0 && kk == 0);
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            v4u whi[2], wlo[2];
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) wfrag(buf, ct * 32, s, whi[ct], wlo[ct]);
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) acc[qd * 2 + ct] = mc_mma<M>(wlo[ct], mh[kk * 8 + s], acc[qd * 2 + ct]);
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) acc[qd * 2 + ct] = mc_mma<M>(whi[ct], ml[kk * 8 + s], acc[qd * 2 + ct]);
#pragma unroll
                            for (int ct = 0; ct < 2; ++ct) acc[qd * 2 + ct] = mc_mma<M>(whi[ct], mh[kk * 8 + s], acc[qd * 2 + ct]);
                            if (s & 1) __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
#pragma unroll
                for (int ct = 0; ct < 8; ++ct) {
                    float v[16];
                    finish(acc[ct], nullptr, (pj ? bV : bK) + ct * 32, nullptr, false, v);
                    store_f32(v, rsOut, rowoff + ct * 128, rowoff_d + ct * 128);
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the slices issued for a tile that does not come
}

template <class M>
int launch_chain(const ChainArgs &a, bool kv, hipStream_t stream) {
    const long ntiles = (a.rows + 31) / 32;
    int grid = device_cus();
    if ((long)grid * MC_WAVES > ntiles) grid = (int)((ntiles + MC_WAVES - 1) / MC_WAVES);
    int rc;
    if (kv) {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_mlp_chain<M, true>), MC_LDS, done, "dz_mlp_chain_forward"))) return rc;
        hipLaunchKernelGGL((k_mlp_chain<M, true>), dim3(grid), dim3(MC_THREADS), MC_LDS, stream, a);
    } else {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_mlp_chain<M, false>), MC_LDS, done, "dz_mlp_chain_forward"))) return rc;
        hipLaunchKernelGGL((k_mlp_chain<M, false>), dim3(grid), dim3(MC_THREADS), MC_LDS, stream, a);
    }
    return DZ_OK;
}

}  // namespace
}  // namespace dz

using namespace dz;

extern "C" {

int dz_mlp_chain_forward(const float *x, long rows, const float *wa, const float *sa, const float *ba, const float *group_shift, int ldg, int group_rows,
                         const float *wb, const float *sb, const float *bb, const float *wk, const float *bk, const float *wv, const float *bv,
                         float *mem, float *k, float *v, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0 && group_rows >= 32 && group_rows % 32 == 0 && (!group_shift || ldg >= MC_C1),
                 "dz_mlp_chain_forward: group_rows a multiple of 32 (got %d), ldg >= 512", group_rows);
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_mlp_chain_forward: math %d is not a split mode", math);
    if (rows == 0) return DZ_OK;
    const bool kv = wk != nullptr;
    DZ_CHECK_ARG(x && wa && sa && ba && wb && sb && bb && mem && (!kv || (wv && bk && bv && k && v)), "dz_mlp_chain_forward: null pointer");
    DZ_CHECK_ARG(!group_shift || rows % group_rows == 0, "dz_mlp_chain_forward: rows not a multiple of group_rows");
    const size_t x_bytes = (size_t)rows * MC_CIN * 4;
    if ((size_t)rows * MC_C2 * 4 >= 0x80000000ull) { set_error("dz_mlp_chain_forward: %ld rows exceed the 2 GiB buffer-addressing limit of an output", rows); return DZ_ERR_UNSUPPORTED; }
    static const int direct = [] { const char *e = getenv("DZ_TUNE_CHAIN_DIRECT"); return e ? atoi(e) : 1; }();
    const ChainArgs a{x, wa, wb, wk, wv, sa, ba, sb, bb, bk, bv, group_shift, mem, k, v, rows, group_rows, ldg, (unsigned int)x_bytes, direct};
    const int rc = math == DZ_MATH_F16X2 ? launch_chain<MathF16>(a, kv, stream) : launch_chain<MathBF16>(a, kv, stream);
    if (rc) return rc;
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
