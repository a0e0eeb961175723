// Submanifold 3 x 3 x 3 sparse convolution with the inputs of a whole z slab staged ONCE per tile ("x-run" engine), gfx950,
// pair16 operands.  Reference call sites: detection/detzero_det/models/centerpoint_modules/backbone3d.py:93-121 (SparseBasicBlock:
// two SubMConv3d + BatchNorm + ReLU with the residual add), :243-280 (conv2 / conv3 / conv4 of VoxelResBackBone8x).
//
// Why: the gather engine (sparse_conv_h.hip, sparse_conv_w.h) fetches one input row per (output row, tap) pair - 13.7-15.3 rows of
// L2 -> CU traffic, load instructions and LDS stage stores per output row.  Rows of a level are stored in ascending linear key
// ((b*D + z)*H + y)*W + x, so for a tile of T CONSECUTIVE output rows the neighbours at one z offset tz (nine taps) are the active
// cells of ONE key interval [key(first) + tz*H*W - W - 1, key(last) + tz*H*W + W + 1], i.e. ONE contiguous range of input rows
// ("window"; about 1.2-1.6 T rows - 3.5-4.2 staged rows per output row instead of 13.7-15.3 gathered ones, as whole 1 KB runs).
// The kernel keeps that window resident in LDS and runs all nine taps of the slab from it; the tap shift is a per-lane row offset
// into the window (rank of the neighbour - first row of the window), a missing neighbour reads a zero row.
//
//   tile      T = WP * PT * 32 consecutive output rows x all COUT channels, one 512-thread workgroup per CU (persistent, XCD-aware
//             deal of runs of consecutive tiles, as k_spconv_h); wave (wp, wc) owns PT 32-row fragments x CT 32-channel fragments.
//   stage     (tz, 16-channel chunk kc, window pass): RCAP window rows x 64 bytes (one MFMA k-step of pair16: hi | lo of two 8-channel
//             groups) -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, 1 KB per wave instruction), double buffered.
//             A window longer than RCAP rows is walked in passes of RCAP rows (rows outside the pass read the zero row; ranks grow
//             with the tap index, so a row's taps are still accumulated in ascending order whatever the pass boundaries are).
//   step      G taps of a stage: their weight slices (COUT rows x 64 bytes each) -> LDS the same way, double buffered.  ONE barrier
//             per step: s_waitcnt vmcnt(0) (this wave's pieces of the next step's data, issued a whole step ago, have landed),
//             s_barrier (everyone's have, and everyone is done with the buffers the loads issued next will overwrite).
//   rows      64-byte rows are unpadded; 16-byte piece p of row r sits at slot p ^ ((r >> 2) & 3) (conflict-free for the 16-lane
//             groups ds_read_b128 is served in when the rows are consecutive); the direct loads realise the swizzle by permuting
//             which source piece a lane fetches.
//   table     the PACKED neighbour table (dz_build_neighbors_packed: one word per (tz, ty) and output row = rank below the centre
//             cell + three presence bits): 3 * PT words per lane and z slab instead of 27 * PT indices.
//   skipping  a (tap, 32-row fragment) whose lanes all miss (no neighbour, or outside the pass) issues no MFMAs (wave-uniform
//             branch on a ballot) - tap skipping at 32-row granularity without any mask table.
// Accumulation order per output element: tz, kc, tap, k - fixed, independent of the tile the row falls into.
// Epilogue = store_tile_pair16 (hgemm.h): BatchNorm scale / shift, residual, ReLU, split, 32-byte stores, staged through the
// window buffer the tile has finished with.
#include <stdlib.h>

#include "hgemm.h"

namespace dz {

struct SpConvXArgs {
    const float *in;            // pair16 rows (in_rows, cin)
    const int *nbr;             // packed table (9, cap)
    const int *win;             // (tiles, 3, 2): first input row, row count of the window of (tile, tz)
    const int *d_m_out;
    const float *w;             // (27, cout, cin) pair16
    const float *scale, *shift, *residual;
    float *out;
    int cin, cout, cap, relu;
    unsigned int in_bytes, w_bytes, nbr_bytes;
};

template <int COUT_, int WP_, int WC_, int PT_, int G_, int RCAP_>
struct XCfg {
    static constexpr int COUT = COUT_, WP = WP_, WC = WC_, PT = PT_, G = G_, RCAP = RCAP_;
    static constexpr int NW = WP * WC, THREADS = 64 * NW;
    static constexpr int CT = COUT / (32 * WC);
    static constexpr int BP = WP * PT * 32;                  // output rows per tile
    static constexpr int NGQ = 9 / G;                        // steps per stage
    static constexpr int WIN_BYTES = RCAP * 64, WSLOT_BYTES = G * COUT * 64;
    static constexpr int OFF_WIN = 0, OFF_W = 2 * WIN_BYTES, OFF_ZERO = OFF_W + 2 * WSLOT_BYTES, OFF_SS = OFF_ZERO + 64;
    static constexpr int LDS_BYTES = OFF_SS + 2 * COUT * 4;
    static constexpr int WJ = G * COUT / 16;                 // 1 KB direct loads of a weight slot
    static constexpr int WIN_J = RCAP / 16;                  // ... of a full window buffer
    static_assert(G == 9 || G == 3, "taps per step: a z slab or one of its window rows");
    static_assert(COUT % (32 * WC) == 0 && RCAP % 16 == 0, "shape");
    static_assert(WIN_BYTES >= NW * STG_WAVE_BYTES, "the epilogue staging windows live in a window buffer");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void x_load16_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory", "m0");
}

// position of a workgroup in its stream of steps (wave-uniform: lives in SGPRs)
struct XPos {
    int seq, tile;              // index in this workgroup's tile sequence, tile number
    int lo0, n0, lo1, n1, lo2, n2;      // windows of the tile's three z slabs (scalars, never indexed: the struct must stay in SGPRs)
    int wlo, wn;                // window of the current slab
    int tz, kc, pass, gq;
    int wb, ws;                 // window buffer / weight slot the step reads
    bool live, stage_first, tz_first, tile_first;
};

template <class C, class M>
__global__ __launch_bounds__(C::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_spconv_x(SpConvXArgs a) {
    constexpr int PT = C::PT, CT = C::CT, G = C::G, COUT = C::COUT, RCAP = C::RCAP, NW = C::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const sc_s = reinterpret_cast<float *>(smem + C::OFF_SS), *const sh_s = sc_s + COUT;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / C::WC, wc = wid % C::WC, l31 = lane & 31, kh = lane >> 5;
    const int m = min(*a.d_m_out, a.cap);
    const int ntiles = (m + C::BP - 1) / C::BP;
    const int nk = a.cin / 16;
    const srsrc_t prsrc = make_srsrc(a.in, a.in_bytes), crsrc = make_srsrc(a.w, a.w_bytes), nrsrc = make_srsrc(a.nbr, a.nbr_bytes);
    const unsigned int row_bytes = (unsigned int)a.cin * 4u, tap_bytes = (unsigned int)(COUT * a.cin * 4);
    const unsigned int nbr_row_bytes = (unsigned int)a.cap * 4u;
    for (int c = tid; c < COUT; c += C::THREADS) {
        sc_s[c] = a.scale ? a.scale[c] : 1.f;
        sh_s[c] = a.shift ? a.shift[c] : 0.f;
    }
    if (tid < 16) reinterpret_cast<unsigned int *>(smem + C::OFF_ZERO)[tid] = 0u;

    // ---- direct loads: lane L of a 1 KB load writes LDS bytes [16 L, 16 L + 16) of its run = row L >> 2, slot L & 3 of 16 rows; the
    // slot holds source piece slot ^ ((row >> 2) & 3), and (row >> 2) & 3 == (L >> 4) & 3 because runs start at multiples of 16 rows
    const int lrow = lane >> 2;
    const unsigned int lpiece = (unsigned int)((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const unsigned int w_voff = (unsigned int)lrow * row_bytes + lpiece;          // + (run's first cout) * row_bytes

    // ---- fragment read addresses
    const unsigned int khsw = (unsigned int)(((2 * kh) ^ ((l31 >> 2) & 3)) << 4);   // swizzled slot of my hi piece in a row whose (row >> 2) & 3 is l31's
    const unsigned int wrow = (unsigned int)((wc * CT * 32 + l31) * 64);            // weight rows: cout index = wc*CT*32 + ct*32 + l31

    // ---- this workgroup's tile sequence (XCD-aware: workgroup b runs on XCD b % 8; runs of XRUN consecutive tiles per XCD share
    // their overlapping windows in that XCD's L2)
    constexpr int XRUN = 8;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = gridDim.x >> 3;
    XPos it;
    auto enter_tile = [&]() {       // it.seq -> first live tile at or after it; loads its windows
        for (;; ++it.seq) {
            const int q = it.seq * nx + jx;
            if ((q / XRUN) * 8 * XRUN >= ntiles) { it.live = false; return; }
            it.tile = ((q / XRUN) * 8 + xcd) * XRUN + q % XRUN;
            if (it.tile < ntiles) break;
        }
        const int *wq = a.win + (size_t)it.tile * 6;
        it.lo0 = __builtin_amdgcn_readfirstlane(wq[0]); it.n0 = __builtin_amdgcn_readfirstlane(wq[1]);
        it.lo1 = __builtin_amdgcn_readfirstlane(wq[2]); it.n1 = __builtin_amdgcn_readfirstlane(wq[3]);
        it.lo2 = __builtin_amdgcn_readfirstlane(wq[4]); it.n2 = __builtin_amdgcn_readfirstlane(wq[5]);
        it.tz = it.n0 > 0 ? 0 : 1;              // (the centre slab of a live tile is never empty: dz_spconv_x_windows)
        it.wlo = it.n0 > 0 ? it.lo0 : it.lo1;
        it.wn = it.n0 > 0 ? it.n0 : it.n1;
        it.kc = it.pass = it.gq = 0;
        it.stage_first = it.tz_first = it.tile_first = true;
    };
    auto advance = [&]() {
        it.ws ^= 1;
        it.stage_first = it.tz_first = it.tile_first = false;
        if (++it.gq < C::NGQ) return;
        it.gq = 0;
        it.stage_first = true;
        it.wb ^= 1;
        if ((it.pass + 1) * RCAP < it.wn) { ++it.pass; return; }
        it.pass = 0;
        if (++it.kc < nk) return;
        it.kc = 0;
        it.tz_first = true;
        ++it.tz;
        if (it.tz == 1) { it.wlo = it.lo1; it.wn = it.n1; return; }        // (n1 > 0 always)
        if (it.tz == 2 && it.n2 > 0) { it.wlo = it.lo2; it.wn = it.n2; return; }
        ++it.seq;
        enter_tile();
    };
    it.seq = 0; it.wb = 0; it.ws = 0; it.live = true;
    enter_tile();
    if (!it.live) return;
    __syncthreads();

    // packed table words of my rows, one array per window row ty (separate arrays, statically indexed: a [3][PT] array picked with a
    // run-time ty is turned into a scratch array by the compiler): current z slab, next
    unsigned int pw0[PT], pw1[PT], pw2[PT], pn0[PT], pn1[PT], pn2[PT];
    auto issue = [&](const XPos &s) {
        // weights of the step's G taps -> slot s.ws
        {
            const unsigned int soff0 = (unsigned int)(s.tz * 9 + s.gq * G) * tap_bytes + (unsigned int)(s.kc * 64);
            const unsigned int lds0 = (unsigned int)(C::OFF_W + s.ws * C::WSLOT_BYTES);
#pragma unroll
            for (int i = 0; i < (C::WJ + NW - 1) / NW; ++i) {
                const int j = i * NW + wid;                         // run j: rows 16 j .. 16 j + 15 of the slot = tap j / (COUT/16), cout (16 j) % COUT ..
                if (C::WJ % NW == 0 || j < C::WJ)
                    x_load16_lds(lds0 + (unsigned int)(j * 1024), w_voff, crsrc,
                                 soff0 + (unsigned int)(j / (COUT / 16)) * tap_bytes + (unsigned int)((j % (COUT / 16)) * 16) * row_bytes);
            }
        }
        // the stage's window pass -> buffer s.wb
        if (s.stage_first) {
            const int wlo = s.wlo + s.pass * RCAP, wcnt = min(RCAP, s.wn - s.pass * RCAP);
            const unsigned int lds0 = (unsigned int)(C::OFF_WIN + s.wb * C::WIN_BYTES);
            for (int j = wid; j * 16 < wcnt; j += NW) {
                // (rows past the window's end re-read its last row: always inside the buffer, never referenced)
                const unsigned int voff = (unsigned int)min(lrow, wcnt - 1 - j * 16) * row_bytes + lpiece;
                x_load16_lds(lds0 + (unsigned int)(j * 1024), voff, prsrc, (unsigned int)(wlo + j * 16) * row_bytes + (unsigned int)(s.kc * 64));
            }
        }
        // packed table words of the slab
        if (s.tz_first) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int row = s.tile * C::BP + (wp * PT + pt) * 32 + l31;
                const unsigned int voff = row < m ? (unsigned int)row * 4u : OOB_OFFSET;
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(pn0[pt]) : "v"(voff), "s"(nrsrc), "s"((unsigned int)(s.tz * 3 + 0) * nbr_row_bytes) : "memory");
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(pn1[pt]) : "v"(voff), "s"(nrsrc), "s"((unsigned int)(s.tz * 3 + 1) * nbr_row_bytes) : "memory");
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(pn2[pt]) : "v"(voff), "s"(nrsrc), "s"((unsigned int)(s.tz * 3 + 2) * nbr_row_bytes) : "memory");
            }
        }
    };

    f32x16 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    struct Frag { v4u c_hi[CT], c_lo[CT], p_hi[PT], p_lo[PT]; bool any[PT]; };
    // the G taps of step `s`.  G == 9: all taps of the slab (ty static); G == 3: the taps of window row ty = s.gq, whose packed
    // words are picked with two selects (ONE copy of the code: a switch over s.gq triples it and its accumulator live ranges)
    auto compute = [&](const XPos &s) {
        const int wlo = s.wlo + s.pass * RCAP;
        const unsigned int wcnt = (unsigned int)min(RCAP, s.wn - s.pass * RCAP);
        const unsigned int win_o = (unsigned int)(C::OFF_WIN + s.wb * C::WIN_BYTES);
        const unsigned int w_hi_o = (unsigned int)(C::OFF_W + s.ws * C::WSLOT_BYTES) + wrow + khsw, w_lo_o = w_hi_o ^ 16u;
        unsigned int ew[PT];
        if constexpr (G != 9) {
            static_assert(G == 3, "G = 1 would need the tap's x position at run time");
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) ew[pt] = s.gq == 0 ? pw0[pt] : (s.gq == 1 ? pw1[pt] : pw2[pt]);
        }
        auto load_frag = [&](Frag &f, auto g_t) {
            constexpr int g = decltype(g_t)::value, ty = g / 3, tx = g % 3;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f.c_hi[ct] = *reinterpret_cast<const v4u *>(smem + w_hi_o + (g * COUT + ct * 32) * 64);
                f.c_lo[ct] = *reinterpret_cast<const v4u *>(smem + w_lo_o + (g * COUT + ct * 32) * 64);
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const unsigned int e = G != 9 ? ew[pt] : (ty == 0 ? pw0[pt] : ty == 1 ? pw1[pt] : pw2[pt]);
                const int off = (int)(e & 0x1FFFFFFFu) - wlo + (tx == 0 ? -1 : tx == 1 ? 0 : (int)((e >> 30) & 1u));
                const bool valid = ((e >> (29 + tx)) & 1u) != 0u && (unsigned int)off < wcnt;
                f.any[pt] = __ballot(valid) != 0ull;
                const unsigned int ra = (unsigned int)off * 64u + (unsigned int)(((2 * kh) ^ ((off >> 2) & 3)) << 4);
                const unsigned int po = valid ? win_o + ra : (unsigned int)C::OFF_ZERO;
                f.p_hi[pt] = *reinterpret_cast<const v4u *>(smem + po);
                f.p_lo[pt] = *reinterpret_cast<const v4u *>(smem + (po ^ 16u));
            }
        };
        auto mma = [&](const Frag &f) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                if (!f.any[pt]) continue;                           // wave-uniform
                // term-major: consecutive MFMAs go to different accumulators; each still receives lo.hi, hi.lo, hi.hi in that order
#pragma unroll
                for (int term = 3 - M::TERMS; term < 3; ++term)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct][pt] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[pt] : f.p_hi[pt], acc[ct][pt]);
            }
        };
        Frag f0, f1;
        load_frag(f0, std::integral_constant<int, 0>{});
        if constexpr (G == 3) {
            load_frag(f1, std::integral_constant<int, 1>{});
            mma(f0);
            load_frag(f0, std::integral_constant<int, 2>{});
            mma(f1);
            mma(f0);
        } else {
            load_frag(f1, std::integral_constant<int, 1>{}); mma(f0);
            load_frag(f0, std::integral_constant<int, 2>{}); mma(f1);
            load_frag(f1, std::integral_constant<int, 3>{}); mma(f0);
            load_frag(f0, std::integral_constant<int, 4>{}); mma(f1);
            load_frag(f1, std::integral_constant<int, 5>{}); mma(f0);
            load_frag(f0, std::integral_constant<int, 6>{}); mma(f1);
            load_frag(f1, std::integral_constant<int, 7>{}); mma(f0);
            load_frag(f0, std::integral_constant<int, 8>{}); mma(f1);
            mma(f0);
        }
    };

    issue(it);
    for (;;) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const XPos cur = it;
        if (cur.tz_first) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                asm volatile("" : "+v"(pn0[pt]), "+v"(pn1[pt]), "+v"(pn2[pt]));
                pw0[pt] = pn0[pt]; pw1[pt] = pn1[pt]; pw2[pt] = pn2[pt];
            }
        }
        advance();
        if (it.live) issue(it);
        compute(cur);
        if (!it.live || it.tile_first) {
            // last step of the tile: its window buffer (every wave is done with it after the barrier) stages the epilogue; the loads
            // in flight go to the other window buffer and the other weight slot
            __syncthreads();
            const int row0 = cur.tile * C::BP;
            store_tile_pair16<C, M>(acc, smem + C::OFF_WIN + cur.wb * C::WIN_BYTES, sc_s, sh_s, 0, a.cout, a.relu != 0,
                                    reinterpret_cast<const unsigned char *>(a.residual), reinterpret_cast<unsigned char *>(a.out), wp, wc, lane, wid,
                                    [&](int lr) {
                                        const int row = row0 + lr;
                                        return row < m ? (size_t)row * a.cout * 4 : ~size_t(0);
                                    });
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            if (!it.live) break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// windows of the tiles: one workgroup per tile; exact first / last referenced input row of every z slab
__global__ __launch_bounds__(256) void k_xwin(const int *__restrict__ nbr, int cap, const int *__restrict__ d_m_out, int tile_rows,
                                              int *__restrict__ win) {
    __shared__ int lo_s[3], hi_s[3];
    const int m = min(*d_m_out, cap);
    const int tile = blockIdx.x, row0 = tile * tile_rows;
    if (threadIdx.x < 3) { lo_s[threadIdx.x] = 0x7FFFFFFF; hi_s[threadIdx.x] = -1; }
    __syncthreads();
    const int row1 = min(m, row0 + tile_rows);
    for (int tz = 0; tz < 3; ++tz) {
        int lo = 0x7FFFFFFF, hi = -1;
        for (int i = threadIdx.x; i < 3 * tile_rows; i += blockDim.x) {
            const int ty = i / tile_rows, row = row0 + i % tile_rows;
            if (row >= row1) continue;
            const unsigned int e = (unsigned int)nbr[(size_t)(tz * 3 + ty) * cap + row];
            if (!(e >> 29)) continue;
            const int r = (int)(e & 0x1FFFFFFFu), l = (int)((e >> 29) & 1u), c = (int)((e >> 30) & 1u), rt = (int)(e >> 31);
            lo = min(lo, r - l);
            hi = max(hi, rt ? r + c : (c ? r : r - 1));
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            lo = min(lo, __shfl_xor(lo, d, 64));
            hi = max(hi, __shfl_xor(hi, d, 64));
        }
        if ((threadIdx.x & 63) == 0 && hi >= 0) { atomicMin(&lo_s[tz], lo); atomicMax(&hi_s[tz], hi); }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int tz = threadIdx.x;
        int lo = lo_s[tz], n = hi_s[tz] >= 0 ? hi_s[tz] - lo + 1 : 0;
        if (n == 0) lo = 0;
        if (tz == 1 && n == 0 && row0 < m) n = 1;        // a live tile always runs its centre slab (the kernel's step stream needs one stage per tile)
        win[(size_t)tile * 6 + 2 * tz] = lo;
        win[(size_t)tile * 6 + 2 * tz + 1] = n;
    }
}

using X32 = XCfg<32, 8, 1, 2, 9, 832>;
using X64 = XCfg<64, 8, 1, 2, 3, 832>;
using X128 = XCfg<128, 4, 2, 2, 3, 576>;

template <class C, class M>
static int launch_x(const SpConvXArgs &a, hipStream_t stream) {
    static PerDeviceFlags done;
    if (int rc = reserve_lds(reinterpret_cast<const void *>(&k_spconv_x<C, M>), C::LDS_BYTES, done, "dz_spconv_forward_split_x")) return rc;
    int grid = ceil_div(a.cap, C::BP);
    const int cus = device_cus();
    if (grid > cus) grid = cus;
    grid = (grid + 7) & ~7;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL((k_spconv_x<C, M>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

template <class M>
static int x_dispatch(const SpConvXArgs &a, hipStream_t stream) {
    if (a.cout == 32) return launch_x<X32, M>(a, stream);
    if (a.cout == 64) return launch_x<X64, M>(a, stream);
    return launch_x<X128, M>(a, stream);
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_spconv_x_tile_rows(int cin, int cout) {
    if (cin != cout) return 0;
    if (cout == 32) return X32::BP;
    if (cout == 64) return X64::BP;
    if (cout == 128) return X128::BP;
    return 0;
}

int dz_spconv_x_windows(const int *nbr_packed, int cap_out, const int *d_m_out, int tile_rows, int *windows, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(nbr_packed && d_m_out && windows, "dz_spconv_x_windows: null pointer");
    DZ_CHECK_ARG(tile_rows > 0 && tile_rows % 32 == 0 && cap_out >= 0, "dz_spconv_x_windows: bad tile_rows %d / cap %d", tile_rows, cap_out);
    if (cap_out == 0) return DZ_OK;
    hipLaunchKernelGGL(k_xwin, dim3(ceil_div(cap_out, tile_rows)), dim3(256), 0, stream, nbr_packed, cap_out, d_m_out, tile_rows, windows);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_spconv_forward_split_x(const float *in, int in_rows, int cin, const int *nbr_packed, const int *windows, int tile_rows, int cap_out,
                              const int *d_m_out, const float *w, const float *scale, const float *shift, const float *residual, int relu,
                              float *out, int cout, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(in && nbr_packed && windows && d_m_out && w && out, "dz_spconv_forward_split_x: null pointer");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2 || math == DZ_MATH_F16, "dz_spconv_forward_split_x: math %d is not a split mode", math);
    const int tr = dz_spconv_x_tile_rows(cin, cout);
    if (tr == 0) {
        set_error("dz_spconv_forward_split_x: %d -> %d channels (submanifold 32 -> 32, 64 -> 64, 128 -> 128 only)", cin, cout);
        return DZ_ERR_UNSUPPORTED;
    }
    DZ_CHECK_ARG(tile_rows == tr, "dz_spconv_forward_split_x: windows built for %d-row tiles, the %d-channel kernel uses %d", tile_rows, cout, tr);
    if (cap_out == 0) return DZ_OK;
    const size_t in_bytes = (size_t)in_rows * cin * sizeof(float), w_bytes = (size_t)27 * cout * cin * sizeof(float),
                 nbr_bytes = (size_t)9 * cap_out * sizeof(int);
    if (in_rows < 0 || in_bytes >= 0x80000000ull || nbr_bytes >= 0x80000000ull || cap_out >= (1 << 29)) {
        set_error("dz_spconv_forward_split_x: input of %zu / table of %zu bytes exceeds the 2 GiB buffer-addressing limit", in_bytes, nbr_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    SpConvXArgs a{in, nbr_packed, windows, d_m_out, w, scale, shift, residual, out, cin, cout, cap_out, relu,
                  (unsigned int)in_bytes, (unsigned int)w_bytes, (unsigned int)nbr_bytes};
    if (math == DZ_MATH_F16) return x_dispatch<MathF16H>(a, stream);
    return math == DZ_MATH_F16X2 ? x_dispatch<MathF16>(a, stream) : x_dispatch<MathBF16>(a, stream);
}

const char *dz_spconv_x_variant(int cin, int cout) {
    if (cin == 32 && cout == 32) return "k_spconv_x<32>";
    if (cin == 64 && cout == 64) return "k_spconv_x<64>";
    if (cin == 128 && cout == 128) return "k_spconv_x<128>";
    return "none";
}

}  // extern "C"
