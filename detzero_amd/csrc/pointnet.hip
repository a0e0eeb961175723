// Fused PointNet encoder of the refining module on pair16 operands: three point-wise layers (Conv1d / Conv2d 1x1 + BatchNorm + ReLU)
// and the max over an object's points in ONE kernel, the activations never leaving the registers.
//
// Reference: refining/detzero_refine/models/modules/geometry_transformer.py:34-67,118-140 (memory / query encoders of the GRM:
// 11 -> 128 -> 128 -> 512 and 4 -> 128 -> 128 -> 256, torch.max over the points, a forward hook on the second layer's output) and
// position_transformer.py:43-124 (PRM: 32 -> 128 -> 128 -> 256 twice).
//
// Why: run layer by layer (dz_linear_forward_split) these stacks are HBM-bound - a 128-channel pair16 row is 512 bytes, and one chunk
// of 96 PRM tracks pushes 4.9 M rows through three layers: 10.6 GB of activation traffic for 0.5 TFLOP (r03: 3.5-4.2 TB/s on every
// layer, i.e. at the memory roofline, 5 ms per chunk).  Here a wave owns 32 rows at a time:
//   * layers 1 and 2 in the orientation of hgemm.h, D[channel x row] = W . X^T: the accumulator of a 32 x 32 fragment gives a lane 4
//     of the 8 channels of a pair16 group for its row; after BatchNorm + ReLU + the (hi, lo) split one cross-lane exchange with lane
//     ^ 32 completes the groups, and the 16 bytes hi | 16 bytes lo a lane then holds ARE its operand of the next layer's MFMA
//     (8 consecutive k of one row) - no LDS, no HBM between the layers;
//   * layer 3 in the transposed orientation, D^T[row x channel] = H2 . W3^T: the 32 rows of the tile are the 16 registers of a lane x
//     its two half-waves, so the max over the rows is 15 in-lane ops + one exchange, and every lane carries the running maximum of ITS
//     output channel across the consecutive tiles of the wave's row range - the (groups, C3) result is written with one atomic max
//     per (wave, group, channel) when the group changes, not per tile;
//   * weights: W1 and W2 (92 KB as pair16) stay in LDS for the whole launch (persistent workgroups, 8 waves), W3 streams through a
//     double buffer in slices of 32 output channels (16 KB, one barrier per slice, loaded once per 256 rows);
//   * the optional tap (layer 2's output, which the GRM / PRM memory branches feed to their second MLP) is stored straight from the
//     operand registers: 32 contiguous bytes per lane and 8-channel group.
// HBM traffic: the input rows (128 bytes each) + the tap when asked for.  Same arithmetic as dz_linear_forward_split (three 16-bit
// MFMAs per product, fp32 accumulation, k ascending), the same results up to the order of the max (exact).
#include <stdlib.h>

#include "hgemm.h"

#ifndef DZ_PN_DIAG
#define DZ_PN_DIAG 0           // timing experiments only (results are garbage): 1 = no workgroup barriers, 2 = no W3 loads, 4 = no layer-3 epilogue,
#endif                          // 8 = no vmcnt waits in the slice loop, 16 = weight fragments from registers, 32 = no hidden-layer epilogues

#ifndef PN_PREFETCH_X
#define PN_PREFETCH_X 1        // 0: the input rows of a tile are loaded at its top (the form up to r05h; timing comparisons)
#endif

namespace dz {

constexpr int PN_THREADS = 512, PN_WAVES = 8, PN_HID = 128, PN_CIN = 32;
constexpr int PN_ROWB = 144;                               // W1's LDS row (a 32-channel chunk): 128 bytes + 16 pad (conflict-free ds_read_b128)
constexpr int PN_W1 = PN_HID * PN_ROWB;                    // [128 rows][144]
// W2 and the W3 ring: unpadded 128-byte rows, the 16-byte piece p of row n stored at p ^ ((n >> 1) & 7) (conflict-free for the 16-lane
// groups ds_read_b128 is served in) - padded they would not leave room for a third slice in flight
constexpr int PN_W2 = 4 * PN_HID * 128;                    // [4 chunks][128 rows][128]
constexpr int PN_W3S = 4 * 32 * 128, PN_RING = 3;          // a slice of 32 output channels: [4 chunks][32 rows][128]; ring of three
constexpr int PN_OFF_W1 = 0, PN_OFF_W2 = PN_W1, PN_OFF_W3 = PN_OFF_W2 + PN_W2, PN_OFF_SS = PN_OFF_W3 + PN_RING * PN_W3S;
constexpr int PN_OFF_RUN = PN_OFF_SS + (4 * PN_HID + 2 * 512) * 4;      // (scale / shift of layers 1 and 2, then of layer 3)
constexpr int PN_LDS = PN_OFF_RUN + PN_WAVES * 16 * 32 * 4;             // + running maxima: [wave][slice][channel of the slice]
static_assert(PN_LDS <= 160 * 1024, "LDS");

struct PointNetArgs {
    const float *x;                // (rows, 32) pair16
    const float *w1, *w2, *w3;     // (128, 32), (128, 128), (c3, 128) pair16
    const float *s1, *b1, *s2, *b2, *s3, *b3;      // BatchNorm scale / shift per layer (fp32)
    float *tap;                    // (rows, 128) pair16 or null: layer 2's output
    float *out;                    // (rows / group_rows, c3) fp32, pre-filled with -inf
    long rows;
    int c3, group_rows;
    unsigned int x_bytes, w3_bytes;
    int x_cols_f32;                // 0: x is pair16; 16 / 32: x is (rows, x_cols_f32) fp32 and is split on the way in
};

template <class M>
__global__ __launch_bounds__(PN_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_pointnet3(PointNetArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *const ss = reinterpret_cast<float *>(smem_raw + PN_OFF_SS);       // s1[128] b1[128] s2[128] b2[128]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    // ---- resident weights: W1 (128 x 32) and W2 (128 x 128) as [chunk][row][144 bytes]
    {
        const v4u *g1 = reinterpret_cast<const v4u *>(a.w1);
        for (int i = tid; i < PN_HID * 8; i += PN_THREADS) {                 // 8 pieces of 16 bytes per row
            const int n = i >> 3, pc = i & 7;
            *reinterpret_cast<v4u *>(smem_raw + PN_OFF_W1 + n * PN_ROWB + pc * 16) = g1[i];
        }
        const v4u *g2 = reinterpret_cast<const v4u *>(a.w2);
        for (int i = tid; i < PN_HID * 32; i += PN_THREADS) {                // 32 pieces per row: chunk = piece / 8
            const int n = i >> 5, pc = i & 31;
            *reinterpret_cast<v4u *>(smem_raw + PN_OFF_W2 + (((pc >> 3) * PN_HID + n) << 7) + (((pc & 7) ^ ((n >> 1) & 7)) << 4)) = g2[i];
        }
        for (int i = tid; i < 512; i += PN_THREADS) {
            ss[4 * PN_HID + i] = (a.s3 && i < a.c3) ? a.s3[i] : 1.f;
            ss[4 * PN_HID + 512 + i] = (a.b3 && i < a.c3) ? a.b3[i] : 0.f;
        }
        if (tid < PN_HID) {
            ss[tid] = a.s1 ? a.s1[tid] : 1.f;
            ss[PN_HID + tid] = a.b1 ? a.b1[tid] : 0.f;
            ss[2 * PN_HID + tid] = a.s2 ? a.s2[tid] : 1.f;
            ss[3 * PN_HID + tid] = a.b2 ? a.b2[tid] : 0.f;
        }
    }
    const srsrc_t xrsrc = make_srsrc(a.x, a.x_bytes);
    const srsrc_t w3rsrc = make_srsrc(a.w3, a.w3_bytes);
    const int nsl = a.c3 / 32;                                               // W3 slices
    // this wave's contiguous range of 32-row tiles; every wave of the launch runs the same number of iterations (barriers)
    const long ntiles = (a.rows + 31) / 32;
    const long nwaves = (long)gridDim.x * PN_WAVES;
    const long per = (ntiles + nwaves - 1) / nwaves;
    const long t_begin = ((long)blockIdx.x * PN_WAVES + wid) * per;

    // W3 slices (32 output channels x 128 inputs = 16 KB) stream through a ring of three with `buffer_load_dwordx4 ... lds`: two 16-byte
    // units per thread, unit u = j * 512 + tid at LDS offset u * 16 = chunk u >> 8, row (u >> 3) & 31, stored piece u & 7
    auto w3_issue = [&](int slice, int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = j * PN_THREADS + tid, chunk = u >> 8, n = (u >> 3) & 31, pc = (u & 7) ^ ((n >> 1) & 7);
            const unsigned int off = (unsigned int)((slice * 32 + n) * (PN_HID * 4) + chunk * 128 + pc * 16);
            const unsigned int base = __builtin_amdgcn_readfirstlane((unsigned int)(PN_OFF_W3 + buf * PN_W3S + (j * PN_THREADS + wid * 64) * 16));
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(base), "v"(off), "s"(w3rsrc) : "memory", "m0");
        }
    };
    // weight fragment of k-step s (16 channels) for the 32 rows [r0, r0 + 32) of a [chunk][rows][144] tile (W1)
    auto wfrag = [&](int base, int rows_per_chunk, int r0, int s, v4u &hi, v4u &lo) {
        const unsigned char *p = smem_raw + base + ((s >> 1) * rows_per_chunk + r0 + l31) * PN_ROWB + ((s & 1) * 4 + h * 2) * 16;
        hi = *reinterpret_cast<const v4u *>(p);
        lo = *reinterpret_cast<const v4u *>(p + 16);
    };
    // ... and of an unpadded, swizzled [chunk][rows][128] tile (W2, the W3 ring); r0 a multiple of 32: the swizzle is that of l31
    const int sw = (l31 >> 1) & 7;
    auto wfrag_u = [&](int base, int rows_per_chunk, int r0, int s, v4u &hi, v4u &lo) {
        if (DZ_PN_DIAG & 16) { hi = v4u{0x3c003c00u + (unsigned)s, 0x3c003c00u, 0x3c003c00u + (unsigned)base, 0x3c003c00u + (unsigned)r0}; lo = hi; return; }
        const unsigned char *p = smem_raw + base + (((s >> 1) * rows_per_chunk + r0 + l31) << 7);
        const int p0 = (s & 1) * 4 + h * 2;
        hi = *reinterpret_cast<const v4u *>(p + ((p0 ^ sw) << 4));
        lo = *reinterpret_cast<const v4u *>(p + (((p0 + 1) ^ sw) << 4));
    };

    // running maximum of output channel (slice cb, column l31) over the rows of the current group: LDS, private to the wave
    // (16 registers less: the kernel sits at the 256-register limit of two waves per SIMD)
    float *const run = reinterpret_cast<float *>(smem_raw + PN_OFF_RUN) + wid * (16 * 32) + l31;      // run[cb * 32]
    if (h == 0) {
        for (int cb = 0; cb < 16; ++cb) run[cb * 32] = -INFINITY;
    }
    long cur_group = -1;
    auto flush = [&]() {
        if (h == 0) {
            for (int cb = 0; cb < nsl; ++cb) {
                const float v = run[cb * 32];
                run[cb * 32] = -INFINITY;
                if (cur_group < 0 || v == -INFINITY) continue;
                float *o = a.out + (size_t)cur_group * a.c3 + cb * 32 + l31;
                if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int *>(o), __float_as_int(v));      // (sign bit, not value: -0.0)
                else atomicMin(reinterpret_cast<unsigned int *>(o), __float_as_uint(v));
            }
        }
    };

    // operand registers of a layer's input: k-step s -> (hi, lo) of channels 16 s + 8 h .. + 7 of my row
    v4u xh[2], xl[2];
    v4u hh[8], hl[8];
    // The input rows of a tile are requested a whole tile ahead, as soon as layer 1 has consumed the registers they land in (with the
    // loads at the top of the tile every wave of the workgroup - they run in step, a barrier per W3 slice - sat out the HBM round trip
    // once per tile).  pair16 rows: 128 bytes = 4 groups of (16 hi | 16 lo), k-step s takes groups 2 s + h.  fp32 rows: k-step s takes
    // channels 16 s + 8 h .. + 7 = 32 contiguous bytes, which wait in xh / xl as they are and are split at the top of their tile (the
    // same bits dz_pair16_from_f32 would have written); a 16-column input has no second k-step (out-of-range offset: zeros)
    auto issue_x = [&](long tile_n) {
        const long row_n = tile_n * 32 + l31;
        const bool ok = tile_n < ntiles && row_n < a.rows;
        if (a.x_cols_f32 == 0) {
            const unsigned int off = ok ? (unsigned int)(row_n * (PN_CIN * 4)) + (unsigned int)(h * 32) : OOB_OFFSET;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(xh[s]) : "v"(off + (unsigned int)(s * 64)), "s"(xrsrc));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(xl[s]) : "v"(off + (unsigned int)(s * 64)), "s"(xrsrc));
            }
        } else {
            const unsigned int rb = (unsigned int)a.x_cols_f32 * 4u;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned int off = (ok && s * 16 + h * 8 < a.x_cols_f32) ? (unsigned int)row_n * rb + (unsigned int)((s * 16 + h * 8) * 4) : OOB_OFFSET;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(xh[s]) : "v"(off), "s"(xrsrc));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:16" : "=v"(xl[s]) : "v"(off), "s"(xrsrc));
            }
        }
    };

    __syncthreads();
    issue_x(t_begin);
    w3_issue(0, 0);
    w3_issue(nsl > 1 ? 1 : 0, 1);
    int ring = 0;                                        // buffer of the slice about to be consumed (wave-uniform)

    for (long it = 0; it < per; ++it) {
        const long tile = t_begin + it;
        const long row = tile * 32 + l31;
        const bool live = tile < ntiles && it < per;                        // wave-uniform
        const bool rok = live && row < a.rows;
        const long group = live ? (tile * 32) / a.group_rows : -1;
        if (group != cur_group) { flush(); cur_group = group; }
        // ---- my input rows: requested before the two W3 slices now in flight (2 loads per thread each; loads return in order), so
        // "at most four outstanding" covers them without draining the slices
        if (PN_PREFETCH_X) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else { issue_x(tile); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#pragma unroll
        for (int s = 0; s < 2; ++s) asm volatile("" : "+v"(xh[s]), "+v"(xl[s]));
        if (a.x_cols_f32 != 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float f0[4] = {__uint_as_float(xh[s].x), __uint_as_float(xh[s].y), __uint_as_float(xh[s].z), __uint_as_float(xh[s].w)};
                const float f1[4] = {__uint_as_float(xl[s].x), __uint_as_float(xl[s].y), __uint_as_float(xl[s].z), __uint_as_float(xl[s].w)};
                uint2 h0, l0, h1, l1;
                split4<M>(f0, h0, l0);
                split4<M>(f1, h1, l1);
                xh[s] = v4u{h0.x, h0.y, h1.x, h1.y};
                xl[s] = v4u{l0.x, l0.y, l1.x, l1.y};
            }
        }
        // ---- layers 1 and 2: D[channel x row], then BatchNorm + ReLU + split + completion of the 8-channel groups
        f32x16 acc[4];
        auto hidden_layer = [&](int wbase, auto ks_t, const v4u *bh, const v4u *bl, const float *sc, const float *sh) {
            constexpr int KSTEPS = decltype(ks_t)::value;           // 2: layer 1 (W1, padded rows); 8: layer 2 (W2, swizzled rows)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
#pragma unroll
            for (int s = 0; s < KSTEPS; ++s) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
                    v4u whi, wlo;
                    if constexpr (KSTEPS == 2) wfrag(wbase, PN_HID, ct * 32, s, whi, wlo);
                    else wfrag_u(wbase, PN_HID, ct * 32, s, whi, wlo);
                    if constexpr (M::TERMS != 1) {
                        acc[ct] = M::mma(wlo, bh[s], acc[ct]);
                        acc[ct] = M::mma(whi, bl[s], acc[ct]);
                    }
                    acc[ct] = M::mma(whi, bh[s], acc[ct]);
                }
            }
            if (DZ_PN_DIAG & 32) {
#pragma unroll
                for (int s = 0; s < 8; ++s) { hh[s] = v4u{__float_as_uint(acc[s >> 1][0]), 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; hl[s] = hh[s]; }
                return;
            }
            // lane (row l31, half h) holds channels 32 ct + 8 q + 4 h + {0..3} in acc[ct][4 q ..]: it keeps the groups g = 4 ct + q with
            // (q & 1) == h, sends the other two to lane ^ 32 and receives their missing halves from it.  (The four fragments finish
            // together and the matrix pipe idles during this epilogue - a quarter of the kernel by the DZ_PN_DIAG measurements; finishing
            // them one or two at a time so that the next ones' MFMAs could cover it was measured: 1927 / 1843 us against 1840)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                uint2 ghi[4], glo[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = ct * 32 + q * 8 + h * 4;
                    const float4 s4 = *reinterpret_cast<const float4 *>(sc + c0), b4 = *reinterpret_cast<const float4 *>(sh + c0);
                    const float v[4] = {fmaxf(fmaf(acc[ct][4 * q], s4.x, b4.x), 0.f), fmaxf(fmaf(acc[ct][4 * q + 1], s4.y, b4.y), 0.f),
                                        fmaxf(fmaf(acc[ct][4 * q + 2], s4.z, b4.z), 0.f), fmaxf(fmaf(acc[ct][4 * q + 3], s4.w, b4.w), 0.f)};
                    split4<M>(v, ghi[q], glo[q]);
                }
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {                             // k-step 2 ct + sl: groups q = 2 sl (lane half 0) and 2 sl + 1 (half 1)
                    const uint2 keep_hi = h ? ghi[2 * sl + 1] : ghi[2 * sl], keep_lo = h ? glo[2 * sl + 1] : glo[2 * sl];
                    const uint2 send_hi = h ? ghi[2 * sl] : ghi[2 * sl + 1], send_lo = h ? glo[2 * sl] : glo[2 * sl + 1];
                    uint2 recv_hi, recv_lo;
                    recv_hi.x = (unsigned int)__shfl_xor((int)send_hi.x, 32, 64);
                    recv_hi.y = (unsigned int)__shfl_xor((int)send_hi.y, 32, 64);
                    recv_lo.x = (unsigned int)__shfl_xor((int)send_lo.x, 32, 64);
                    recv_lo.y = (unsigned int)__shfl_xor((int)send_lo.y, 32, 64);
                    // channels 8 g .. 8 g + 3 come from half 0, 8 g + 4 .. + 7 from half 1
                    const int s = 2 * ct + sl;
                    hh[s] = h ? v4u{recv_hi.x, recv_hi.y, keep_hi.x, keep_hi.y} : v4u{keep_hi.x, keep_hi.y, recv_hi.x, recv_hi.y};
                    hl[s] = h ? v4u{recv_lo.x, recv_lo.y, keep_lo.x, keep_lo.y} : v4u{keep_lo.x, keep_lo.y, recv_lo.x, recv_lo.y};
                }
            }
        };
        hidden_layer(PN_OFF_W1, std::integral_constant<int, 2>{}, xh, xl, ss, ss + PN_HID);
        if (PN_PREFETCH_X) issue_x(tile + 1);              // (past the wave's range: another wave's rows or out of range - never used)
        {
            v4u ih[8], il[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) { ih[s] = hh[s]; il[s] = hl[s]; }
            hidden_layer(PN_OFF_W2, std::integral_constant<int, 8>{}, ih, il, ss + 2 * PN_HID, ss + 3 * PN_HID);
        }
        if (a.tap && rok) {                                                  // layer 2's output: group 2 s + h of my row = 32 contiguous bytes
            unsigned char *tp = reinterpret_cast<unsigned char *>(a.tap) + (size_t)row * (PN_HID * 4) + h * 32;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                *reinterpret_cast<v4u *>(tp + s * 64) = hh[s];
                *reinterpret_cast<v4u *>(tp + s * 64 + 16) = hl[s];
            }
        }
        // ---- layer 3, transposed: D^T[row x channel] per slice of 32 channels, max over the rows into the running maximum
        for (int cb = 0; cb < nsl; ++cb) {
            // slice cb has landed once my loads for it are back and the workgroup has met; its loads went out two slices ago, the
            // slice issued since (2 loads per thread; loads return in order) stays in flight.  The buffer of the previous slice is
            // free after the barrier: the slice after next goes into it (the next tile walks the same slices again)
            if (!(DZ_PN_DIAG & 8)) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (!(DZ_PN_DIAG & 1)) __syncthreads();
            if (!(DZ_PN_DIAG & 2)) {
                int nxt = cb + 2;
                if (nxt >= nsl) nxt -= nsl;
                if (nxt >= nsl) nxt -= nsl;                                  // (nsl = 1)
                int nb = ring + 2;
                if (nb >= PN_RING) nb -= PN_RING;
                w3_issue(nxt, nb);
            }
            f32x16 d;
#pragma unroll
            for (int e = 0; e < 16; ++e) d[e] = 0.f;
            const int wbase3 = PN_OFF_W3 + ring * PN_W3S;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                v4u whi, wlo;
                wfrag_u(wbase3, 32, 0, s, whi, wlo);
                if constexpr (M::TERMS != 1) {
                    d = M::mma(hl[s], whi, d);
                    d = M::mma(hh[s], wlo, d);
                }
                d = M::mma(hh[s], whi, d);
            }
            if (DZ_PN_DIAG & 4) { if (h == 0) run[cb * 32] = d[0]; ring = ring + 1 == PN_RING ? 0 : ring + 1; continue; }
            // lane: channel cb * 32 + l31, rows 8 (e >> 2) + 4 h + (e & 3)
            const int ch = cb * 32 + l31;
            const float sc = ss[4 * PN_HID + ch], sh = ss[4 * PN_HID + 512 + ch];
            float m = -INFINITY;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long r = tile * 32 + 8 * (e >> 2) + 4 * h + (e & 3);
                const float v = fmaxf(fmaf(d[e], sc, sh), 0.f);
                m = (live && r < a.rows) ? fmaxf(m, v) : m;
            }
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (h == 0) run[cb * 32] = fmaxf(run[cb * 32], m);
            ring = ring + 1 == PN_RING ? 0 : ring + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the slices issued for a tile that does not come
    flush();
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_pointnet3_forward(const float *x, long rows, const float *w1, const float *s1, const float *b1, const float *w2, const float *s2,
                         const float *b2, const float *w3, const float *s3, const float *b3, int c3, int group_rows, float *tap, float *out,
                         int x_cols_f32, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0 && (c3 == 256 || c3 == 512 || c3 == 128) && group_rows >= 32 && group_rows % 32 == 0,
                 "dz_pointnet3_forward: c3 in {128, 256, 512}, group_rows a multiple of 32 (got %d, %d)", c3, group_rows);
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_pointnet3_forward: math %d is not a split mode", math);
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(x && w1 && w2 && w3 && out && rows % group_rows == 0, "dz_pointnet3_forward: null pointer / rows not a multiple of group_rows");
    DZ_CHECK_ARG(x_cols_f32 == 0 || x_cols_f32 == 16 || x_cols_f32 == 32, "dz_pointnet3_forward: fp32 input rows of 16 or 32 columns (got %d)", x_cols_f32);
    const size_t x_bytes = (size_t)rows * (x_cols_f32 ? x_cols_f32 : PN_CIN) * 4;
    if (x_bytes >= 0x80000000ull) { set_error("dz_pointnet3_forward: input of %zu bytes exceeds the 2 GiB buffer-addressing limit", x_bytes); return DZ_ERR_UNSUPPORTED; }
    int rc = fill_u32(out, 0xFF800000u, (size_t)(rows / group_rows) * c3, stream);       // -inf
    if (rc) return rc;
    PointNetArgs a{x, w1, w2, w3, s1, b1, s2, b2, s3, b3, tap, out, rows, c3, group_rows, (unsigned int)x_bytes, (unsigned int)((size_t)c3 * PN_HID * 4), x_cols_f32};
    const long ntiles = (rows + 31) / 32;
    int grid = device_cus();
    if ((long)grid * PN_WAVES > ntiles) grid = (int)((ntiles + PN_WAVES - 1) / PN_WAVES);
    if (math == DZ_MATH_F16X2) {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_pointnet3<MathF16>), PN_LDS, done, "dz_pointnet3_forward"))) return rc;
        hipLaunchKernelGGL(k_pointnet3<MathF16>, dim3(grid), dim3(PN_THREADS), PN_LDS, stream, a);
    } else {
        static PerDeviceFlags done;
        if ((rc = reserve_lds(reinterpret_cast<const void *>(&k_pointnet3<MathBF16>), PN_LDS, done, "dz_pointnet3_forward"))) return rc;
        hipLaunchKernelGGL(k_pointnet3<MathBF16>, dim3(grid), dim3(PN_THREADS), PN_LDS, stream, a);
    }
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
