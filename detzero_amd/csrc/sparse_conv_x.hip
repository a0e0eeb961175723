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
//   tile      two UNITS of UR = WP * 32 consecutive output rows x all COUT channels, one 512-thread workgroup per CU; wave (wp, wc) owns
//             one 32-row fragment of each unit x CT 32-channel fragments.  Workgroups are persistent and take tiles from per-XCD queues
//             (runs of 64 consecutive tiles per XCD: neighbouring tiles share window rows in that XCD's L2; one atomic ticket per tile,
//             fetched during the epilogue of the tile before); the last units of a launch are dealt one by one (half the fragments
//             idle) so that no workgroup is left with a whole tile while the others have finished.
//   stage     (tz, 16-channel chunk kc, window pass): RCAP window rows x 64 bytes (one MFMA k-step of pair16: hi | lo of two 8-channel
//             groups) -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, 1 KB per wave instruction), double buffered.
//             A window longer than RCAP rows (2-3 % of the slabs: a tile in a sparse region next to a dense slab; up to 30 000 rows)
//             is not staged at all: the stage runs in GATHER mode - each lane fetches its neighbour row's 2 x 16 bytes of the chunk
//             straight from global memory into the MFMA operand registers (as k_spconv_w does), same taps, same order.  (Walking
//             such a window in passes of RCAP rows costs up to 39 stages for one slab: measured, the slowest workgroup then took
//             1.8-2x the average one.)
//   step      TAPS taps of a stage (one window row ty, or the whole slab at 32 channels): their weight slices (COUT rows x 64 bytes
//             each) -> LDS the same way, D steps ahead in a ring of D + 1 slots.  ONE barrier per step: s_waitcnt vmcnt(N) with N =
//             the loads issued since for LATER steps (vector memory operations retire in order; every step issues the same number of
//             loads, absent rows through an out-of-range offset), then s_barrier (everyone's pieces have landed, and everyone is done
//             with the buffers the loads issued next will overwrite).  The MFMAs of a step's last tap are issued after the NEXT
//             step's barrier, so no barrier is followed by a cold matrix pipe.
//   rows      64-byte rows are unpadded; 16-byte piece p of row r sits at slot p ^ ((r >> 2) & 3) (conflict-free for the 16-lane
//             groups ds_read_b128 is served in when the rows are consecutive); the direct loads realise the swizzle by permuting
//             which source piece a lane fetches.
//   table     the PACKED neighbour table (dz_build_neighbors_packed: one word per (tz, ty) and output row = rank below the centre
//             cell + three presence bits): 3 * PT words per lane and z slab instead of 27 * PT indices.
//   skipping  a (fragment, tap) none of the fragment's 32 rows has a neighbour at issues no MFMAs (wave-uniform branch on a ballot) -
//             tap skipping at 32-row granularity without any mask table; the rows of a unit are processed in the order of their TAP SETS
//             (k_xwin sorts them; the epilogue writes through the row map), which makes whole fragments agree on what they skip.
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
    unsigned long long *dbg;    // DIAG bit 9 builds: per-wave cycle sums (8 words per wave) or null
    const int *perm;            // output row of each (unit, position) when the table is in tap-set order, or null
    int *queue;                 // 10 words behind the windows: next ticket of each XCD's tile queue, workgroups done, single-unit queue
    int xrun;                   // consecutive tiles per XCD run (8; DZ_TUNE_XRUN)
    int steal, singles;         // tail: take whole tiles from other XCDs' queues before the single units; single units per workgroup (1)
    int sload;                  // window words through scalar loads (1; DZ_TUNE_X_SLOAD)
};

template <int COUT_, int WP_, int WC_, int PT_, int TAPS_, int D_, int RCAP_>
struct XCfg {
    static constexpr int COUT = COUT_, WP = WP_, WC = WC_, PT = PT_, TAPS = TAPS_, D = D_, RCAP = RCAP_;
    static constexpr int NW = WP * WC, THREADS = 64 * NW;
    static constexpr int CT = COUT / (32 * WC);
    static constexpr int BP = WP * PT * 32;                  // output rows per tile
    static constexpr int UR = BP / 2;                        // ... per unit: a tile is two consecutive units (fragment pt of every wave = unit pt), the
                                                             // last tiles of a launch are single units (half the fragments idle): finer load balance
    static constexpr int SPS = 9 / TAPS;                     // steps per stage: a step = TAPS taps = 1 or 3 window rows ty
    static constexpr int NSLOT = D + 1;                      // weight slots: a step's slices are issued D steps ahead of their use
    static constexpr int WIN_BYTES = (RCAP + 1) * 64;        // + the zero row missing neighbours read
    static constexpr int WSLOT_BYTES = TAPS * COUT * 64;
    static constexpr int OFF_WIN = 0, OFF_W = 2 * WIN_BYTES, OFF_TRASH = OFF_W + NSLOT * WSLOT_BYTES, OFF_SS = OFF_TRASH + 1024;
    static constexpr int OFF_TK = OFF_SS + 2 * COUT * 4;     // tickets of the workgroup's next two tiles
    static constexpr int LDS_BYTES = OFF_TK + 16;
    static constexpr int R = COUT / 16;                      // 1 KB runs (16 output channels x 64 bytes) per tap slice
    static constexpr int WJ = TAPS * R, WPW = (WJ + NW - 1) / NW;               // 1 KB direct loads of a weight slot: in all, per wave
    static constexpr int WIN_J = RCAP / 16, WPWIN = WIN_J / NW;                 // ... of a full window buffer
    static constexpr int WINPW = WPWIN + 3 * PT;             // loads a wave issues at a stage's first step besides the weights
    static_assert(TAPS == 3 || TAPS == 9, "a step is one window row or a whole z slab");
    static_assert(PT == 2, "a unit is one fragment per wave");
    static_assert(D >= 1 && D <= SPS && D <= 3, "weight slices are issued 1..3 steps ahead, never before the stage's window");
    static_assert(COUT % (32 * WC) == 0 && NW % R == 0 && RCAP % (16 * NW) == 0, "shape");
    static_assert(RCAP * 64 >= NW * STG_WAVE_BYTES, "the epilogue staging windows live in a window buffer (below its zero row)");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void x_load16_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory", "m0");
}

// store_tile_pair16 (hgemm.h) for this kernel: the output rows of the lane's items come in `orow` (fetched through the row map a
// whole stage earlier: -1 = no row), and ALL residual groups of the tile are requested before the first fragment is staged - one
// round trip to memory per tile instead of one per 32 x 32 fragment (measured: the residual cost of the per-fragment form was as much
// as the rest of the epilogue).
template <class T, class M>
__device__ __forceinline__ void x_store_tile(const f32x16 (&acc)[T::CT][T::PT], unsigned char *smem_bytes, const float *sc_s, const float *sh_s,
                                             int cout, bool relu, const unsigned char *residual, unsigned char *out, int wc, int lane, int wid,
                                             const int (&orow)[T::PT][2]) {
    unsigned char *const stg = smem_bytes + wid * STG_WAVE_BYTES;
    const int h = lane >> 5, g = lane & 3;
    const size_t row_bytes = (size_t)cout * 4;
    uint4 rh[T::PT][T::CT][2], rl[T::PT][T::CT][2];
    if (residual) {
#pragma unroll
        for (int pt = 0; pt < T::PT; ++pt)
#pragma unroll
            for (int ct = 0; ct < T::CT; ++ct)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    rh[pt][ct][i] = rl[pt][ct][i] = make_uint4(0u, 0u, 0u, 0u);
                    if (orow[pt][i] >= 0) {
                        const unsigned char *rp = residual + (size_t)orow[pt][i] * row_bytes + (size_t)(wc * T::CT * 32 + ct * 32 + g * 8) * 4;
                        rh[pt][ct][i] = *reinterpret_cast<const uint4 *>(rp);
                        rl[pt][ct][i] = *reinterpret_cast<const uint4 *>(rp + 16);
                    }
                }
    }
#pragma unroll
    for (int pt = 0; pt < T::PT; ++pt) {
#pragma unroll
        for (int ct = 0; ct < T::CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = make_float4(acc[ct][pt][4 * j], acc[ct][pt][4 * j + 1], acc[ct][pt][4 * j + 2], acc[ct][pt][4 * j + 3]);
                *reinterpret_cast<float4 *>(stg + (lane & 31) * STG_ROW_BYTES + j * 32 + h * 16) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int lc = wc * T::CT * 32 + ct * 32 + g * 8;                 // my 8-channel group
            const float4 sc0 = *reinterpret_cast<const float4 *>(sc_s + lc), sc1 = *reinterpret_cast<const float4 *>(sc_s + lc + 4);
            const float4 sh0 = *reinterpret_cast<const float4 *>(sh_s + lc), sh1 = *reinterpret_cast<const float4 *>(sh_s + lc + 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = (lane >> 2) + 16 * i;
                const float4 va = *reinterpret_cast<const float4 *>(stg + r * STG_ROW_BYTES + g * 32);
                const float4 vb = *reinterpret_cast<const float4 *>(stg + r * STG_ROW_BYTES + g * 32 + 16);
                float v[8] = {fmaf(va.x, sc0.x, sh0.x), fmaf(va.y, sc0.y, sh0.y), fmaf(va.z, sc0.z, sh0.z), fmaf(va.w, sc0.w, sh0.w),
                              fmaf(vb.x, sc1.x, sh1.x), fmaf(vb.y, sc1.y, sh1.y), fmaf(vb.z, sc1.z, sh1.z), fmaf(vb.w, sc1.w, sh1.w)};
                if (residual) {
                    const uint4 a4 = rh[pt][ct][i], b4 = rl[pt][ct][i];
                    const unsigned int hw[4] = {a4.x, a4.y, a4.z, a4.w}, lw[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] += M::join(hw[k] & 0xFFFFu, lw[k] & 0xFFFFu);
                        v[2 * k + 1] += M::join(hw[k] >> 16, lw[k] >> 16);
                    }
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const float v0[4] = {v[0], v[1], v[2], v[3]}, v1[4] = {v[4], v[5], v[6], v[7]};
                uint2 h0, l0, h1, l1;
                split4<M>(v0, h0, l0);
                split4<M>(v1, h1, l1);
                if (orow[pt][i] >= 0) {
                    unsigned char *gp = out + (size_t)orow[pt][i] * row_bytes + (size_t)lc * 4;
                    *reinterpret_cast<uint4 *>(gp) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    *reinterpret_cast<uint4 *>(gp + 16) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// one stage of a workgroup's stream (wave-uniform: SGPRs; never indexed dynamically)
struct XStage {
    int u0, seq, tz, kc;        // first unit of the tile, the tile's number in the workgroup's sequence
    int wlo, wcnt;              // the window: first input row, rows (0: gather mode, or past the end of the stream - its loads fetch nothing)
    int wb;                     // window buffer
    bool half;                  // the tile is ONE unit (fragment 1 of every wave idle)
    bool live, gather, tz_first, tile_first;    // gather: the slab's window does not fit the buffer, operands come from global memory
};

// DIAG (development, DZ_TUNE_X_DIAG; timing experiments, results are garbage): bit 0 no MFMAs, 1 no fragment LDS reads, 2 no weight
// loads, 3 no window loads, 4 no barriers, 5 no epilogue, 7 no tap skipping
template <class C, class M, int DIAG = 0>
__global__ __launch_bounds__(C::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_spconv_x(SpConvXArgs a) {
    constexpr int PT = C::PT, CT = C::CT, COUT = C::COUT, RCAP = C::RCAP, NW = C::NW, D = C::D, TAPS = C::TAPS, SPS = C::SPS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *const sc_s = reinterpret_cast<float *>(smem + C::OFF_SS), *const sh_s = sc_s + COUT;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / C::WC, wc = wid % C::WC, l31 = lane & 31, kh = lane >> 5;
    const int m = min(*a.d_m_out, a.cap);
    const int nunits = (m + C::UR - 1) / C::UR;
    // tiles of two units while at least a unit per workgroup remains beyond them (a multiple of 64 = whole runs for all 8 queues), then
    // single units
    const int XRUN = a.xrun;
    // single-unit tickets per workgroup at the end of the launch (0 unless forced: measured best at 8, 16 and 32 frames per pass)
    const int singles = a.singles >= 0 ? a.singles : 0;
    const int nfull = (max(nunits - singles * (int)gridDim.x, 0) / 2) / (8 * XRUN) * (8 * XRUN);
    const int nk = a.cin / 16;
    const srsrc_t prsrc = make_srsrc(a.in, a.in_bytes), crsrc = make_srsrc(a.w, a.w_bytes), nrsrc = make_srsrc(a.nbr, a.nbr_bytes);
    const unsigned int row_bytes = (unsigned int)a.cin * 4u, tap_bytes = (unsigned int)(COUT * a.cin * 4);
    const unsigned int nbr_row_bytes = (unsigned int)a.cap * 4u;
    for (int c = tid; c < COUT; c += C::THREADS) {
        sc_s[c] = a.scale ? a.scale[c] : 1.f;
        sh_s[c] = a.shift ? a.shift[c] : 0.f;
    }
    if (tid < 32) reinterpret_cast<unsigned int *>(smem + C::OFF_WIN + (tid >> 4) * C::WIN_BYTES + RCAP * 64)[tid & 15] = 0u;    // the zero rows

    // ---- direct loads: lane L of a 1 KB load writes LDS bytes [16 L, 16 L + 16) of its run = row L >> 2, slot L & 3 of 16 rows; the
    // slot holds source piece slot ^ ((row >> 2) & 3), and (row >> 2) & 3 == (L >> 4) & 3 because runs start at multiples of 16 rows.
    // Run j = i * NW + wid of a buffer belongs to wave wid: everything that depends on the wave only is computed here, once.
    const int lrow = lane >> 2;
    const unsigned int lpiece = (unsigned int)((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    const unsigned int w_voff = (unsigned int)lrow * row_bytes + lpiece;
    const unsigned int wid_lds = (unsigned int)wid * 1024u;                                           // run wid of a buffer
    const unsigned int wid_rows = (unsigned int)(wid * 16) * row_bytes;                               // window run wid: 16 * wid rows in
    const unsigned int nw_rows = (unsigned int)(NW * 16) * row_bytes;
    const unsigned int w_soff_w = (unsigned int)(wid / C::R) * tap_bytes + (unsigned int)((wid % C::R) * 16) * row_bytes;   // weight run wid: tap wid / R, cout 16 (wid % R)
    const bool w_last_ok = C::WJ % NW == 0 || wid < C::WJ % NW;                                       // my run of the last, partial round of a weight slot exists

    // ---- fragment read addresses
    const unsigned int khsw = (unsigned int)(((2 * kh) ^ ((l31 >> 2) & 3)) << 4);   // swizzled slot of my hi piece in a row whose (row >> 2) & 3 is l31's
    const unsigned int wrow = (unsigned int)((wc * CT * 32 + l31) * 64);            // weight rows: cout index = wc*CT*32 + ct*32 + l31

    // ---- this workgroup's stream of stages.  Tiles come from one queue per XCD (workgroup b runs on XCD b % 8): XCD x owns the runs
    // x, x + 8, .. of XRUN consecutive tiles - neighbouring tiles share most of their windows in that XCD's L2 - and its workgroups
    // take them in order, one ticket (atomic add on the queue's counter) per tile: tiles differ in cost (gather-mode slabs, empty
    // slabs, skipped taps), a static deal left the slowest workgroup 20-25 % behind the average.  Thread 0 takes the tickets two tiles
    // ahead - the atomic is issued at the start of a tile's epilogue and read at its end, among the epilogue's own loads and stores -
    // and hands them to the other waves through LDS (tk_s[k & 1] = ticket of the workgroup's k-th tile); the last workgroup to finish
    // resets the counters for the next launch.  Inside a tile: tz, 16-channel chunk.
    // (The returning atomic is a plain, compiler-tracked one: an asynchronous one from inline asm does not survive the register copies
    // the compiler inserts at loop edges - they read the destination before the value has landed.)
    const int xcd = blockIdx.x & 7;
    int *const tk_s = reinterpret_cast<int *>(smem + C::OFF_TK);
    // a ticket: q < nfull / 8 = the q-th full tile of my XCD's queue; past those, 2^30 + s = the s-th single unit of the common queue
    // When my XCD's queue is exhausted the other XCDs' are tried before the single units (round 5): without that, the workgroups of an
    // XCD that finishes early eat the single units while a slower XCD still hands out whole tiles, and the launch ends on a whole
    // tile - the slowest workgroup ran 8.6 % / 7 % behind the average at 128 / 64 channels and 32 frames per pass
    // (profiles/r05_ab_notes.txt).  A stolen tile loses its L2 neighbourhood, but only the last few per launch are stolen.
    // (thread 0 only; `steal_from` remembers where the last probe round ended: 8 = every queue is empty)
    int steal_from = 1;
    const int steal_on = a.steal;
    auto take_ticket = [&]() {
        const int per = nfull / 8;
        if (steal_from == 1) {
            const int t = atomicAdd(a.queue + xcd, 1);
            if (t < per) return t | (xcd << 20);
            if (!steal_on) steal_from = 8;
        }
        while (steal_from < 8) {
            const int x2 = (xcd + steal_from) & 7;
            const int t = atomicAdd(a.queue + x2, 1);
            if (t < per) return t | (x2 << 20);
            ++steal_from;
        }
        return (int)(0x40000000 | atomicAdd(a.queue + 9, 1));
    };
    if (tid == 0) {
        tk_s[0] = take_ticket();
        tk_s[1] = take_ticket();
    }
    __syncthreads();
    struct Gen { int seq, u0, lo0, n0, lo1, n1, lo2, n2, tz, kc, wlo, wn, wb; bool live, started, half; } g;
    g.seq = -1; g.live = true; g.started = false; g.half = false; g.wb = 1; g.u0 = 0; g.tz = g.kc = 0; g.wlo = g.wn = 0;
    g.lo0 = g.n0 = g.lo1 = g.n1 = g.lo2 = g.n2 = 0;
    auto gen = [&]() {          // the next stage of the stream (live = false: past its end)
        XStage s;
        s.tile_first = s.tz_first = false;
        bool next_tile = !g.started;
        if (g.started && ++g.kc == nk) {
            g.kc = 0;
            s.tz_first = true;
            ++g.tz;
            if (g.tz == 1) { g.wlo = g.lo1; g.wn = g.n1; }                        // (n1 > 0 always)
            else if (g.tz == 2 && g.n2 > 0) { g.wlo = g.lo2; g.wn = g.n2; }
            else next_tile = true;
        }
        if (next_tile && g.live) {
            g.started = true;
            ++g.seq;                        // my g.seq-th tile: its ticket is in tk_s[g.seq & 1]
            const int q = __builtin_amdgcn_readfirstlane(tk_s[g.seq & 1]);
            if (q & 0x40000000) {
                g.u0 = 2 * nfull + (q & 0x3FFFFFFF);
                g.half = true;
                g.live = g.u0 < nunits;
            } else {
                const int qx = q >> 20, qq = q & 0xFFFFF;           // the queue (XCD) the ticket came from, its number there
                g.u0 = 2 * (((qq / XRUN) * 8 + qx) * XRUN + qq % XRUN);
                g.half = false;
            }
            if (g.live) {
                // windows of the tile's unit(s): per z slab the union of the two units' ranges
                // The twelve words come through the SCALAR cache in three s_load_dwordx4 with one lgkmcnt wait.  As plain loads the
                // compiler issued six global_load_dwordx2 ONE AFTER THE OTHER, each behind an `s_waitcnt vmcnt(0)` - six memory round
                // trips per tile, and each wait also drained the window / weight prefetches in flight: ~9.7 k of a tile's ~230 k cycles
                // at 128 channels, 3-6 % of the x-run kernels (in-kernel counter `gen`, profiles/r05b_xrun_cycles.txt).  (The table is
                // written by an earlier launch: the scalar cache cannot hold stale lines of it.  A single unit's second half is the
                // next unit's words - inside the buffer, not used.)
                const int *wq = a.win + (size_t)g.u0 * 6;
                typedef int v4i_t __attribute__((ext_vector_type(4)));
                v4i_t w0, w1, w2;
                if (a.sload) {
                    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(w0), "=&s"(w1), "=&s"(w2) : "s"(wq) : "memory");
                } else {            // (development knob DZ_TUNE_X_SLOAD=0: the round-4 loads)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        w0[q] = __builtin_amdgcn_readfirstlane(wq[q]);
                        w1[q] = __builtin_amdgcn_readfirstlane(wq[4 + q]);
                        w2[q] = __builtin_amdgcn_readfirstlane(wq[8 + q]);
                    }
                }
                const int ww[12] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w, w2.x, w2.y, w2.z, w2.w};
                int lo[3], n[3];
#pragma unroll
                for (int z = 0; z < 3; ++z) {
                    lo[z] = ww[2 * z];
                    n[z] = ww[2 * z + 1];
                    if (!g.half) {
                        const int lb = ww[6 + 2 * z], nb = ww[6 + 2 * z + 1];
                        if (n[z] == 0) { lo[z] = lb; n[z] = nb; }
                        else if (nb > 0) { const int hi = max(lo[z] + n[z], lb + nb); lo[z] = min(lo[z], lb); n[z] = hi - lo[z]; }
                    }
                }
                g.lo0 = lo[0]; g.n0 = n[0]; g.lo1 = lo[1]; g.n1 = n[1]; g.lo2 = lo[2]; g.n2 = n[2];
                g.tz = g.n0 > 0 ? 0 : 1;          // (the centre slab of a live unit is never empty: dz_spconv_x_windows)
                g.wlo = g.n0 > 0 ? g.lo0 : g.lo1;
                g.wn = g.n0 > 0 ? g.n0 : g.n1;
                g.kc = 0;
                s.tile_first = s.tz_first = true;
            }
        }
        g.wb ^= 1;
        s.live = g.live;
        s.u0 = g.u0; s.half = g.half; s.seq = g.seq; s.tz = g.tz; s.kc = g.kc; s.wb = g.wb;
        s.gather = g.live && g.wn > RCAP;
        s.wlo = g.wlo;
        s.wcnt = g.live && g.wn <= RCAP ? g.wn : 0;
        return s;
    };

    // packed table words of my rows, one array per window row ty (separate arrays, statically indexed: a [3][PT] array picked with a
    // run-time index is turned into a scratch array by the compiler): current z slab, next
    unsigned int pw0[PT], pw1[PT], pw2[PT], pn0[PT], pn1[PT], pn2[PT];
    // ---- issue, one load at a time (every load is a filler behind an MFMA, never a block of its own):
    // run I of my share of stage s's window pass (WPWIN per wave and stage, always: a run past the window's end - every run of a
    // stage past the end of the stream - fetches nothing, so the per-wave load counts stay static)
    auto issue_win = [&](const XStage &s, auto i_t) {
        if constexpr (DIAG & 8) return;
        constexpr int I = decltype(i_t)::value;
        const int left = s.wcnt - 1 - wid * 16 - I * NW * 16;     // rows of the window after the first row of my run
        // (rows past the window's end re-read its last row: always inside the buffer, never referenced)
        const unsigned int voff = left >= 0 ? (unsigned int)min(lrow, left) * row_bytes + lpiece : OOB_OFFSET;
        x_load16_lds((unsigned int)(C::OFF_WIN + s.wb * C::WIN_BYTES) + wid_lds + (unsigned int)(I * NW * 1024), voff, prsrc,
                     (unsigned int)s.wlo * row_bytes + (unsigned int)(s.kc * 64) + wid_rows + (unsigned int)I * nw_rows);
    };
    // table word (fragment PTI, window row TY) of stage s's slab
    auto issue_pw = [&](const XStage &s, auto pt_t, auto ty_t) {
        constexpr int PTI = decltype(pt_t)::value, TY = decltype(ty_t)::value;
        const int row = (s.u0 + PTI) * C::UR + wp * 32 + l31;          // fragment PTI = unit PTI of the tile
        // (with a table in tap-set order `row` is a POSITION; live rows keep the positions below m: k_xwin)
        const unsigned int voff = s.live && s.tz_first && row < m && !(PTI == 1 && s.half) ? (unsigned int)row * 4u : OOB_OFFSET;
        const unsigned int so = __builtin_amdgcn_readfirstlane((unsigned int)(s.tz * 3 + TY) * nbr_row_bytes);     // (scalar operand of the load)
        const srsrc_t rs = nrsrc;            // (named here: an asm operand alone does not capture a variable in a generic lambda)
        unsigned int &dst = TY == 0 ? pn0[PTI] : (TY == 1 ? pn1[PTI] : pn2[PTI]);
        asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rs), "s"(so) : "memory");
    };
    // run I of my share of the weight slices of step q of stage s -> slot `slot` (WPW per wave, always)
    auto issue_w = [&](const XStage &s, int q, int slot, auto i_t) {
        if constexpr (DIAG & 4) return;
        constexpr int I = decltype(i_t)::value;
        constexpr bool part = C::WJ % NW != 0 && I == C::WPW - 1;             // the last, partial round
        const bool ok = !part || w_last_ok;
        x_load16_lds(ok ? (unsigned int)(C::OFF_W + slot * C::WSLOT_BYTES) + wid_lds + (unsigned int)(I * NW * 1024) : (unsigned int)C::OFF_TRASH,
                     ok && s.live ? w_voff : OOB_OFFSET, crsrc,
                     (unsigned int)(s.tz * 9 + q * TAPS + I * (NW / C::R)) * tap_bytes + (unsigned int)(s.kc * 64) + w_soff_w);
    };

    f32x16 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // row addresses of the current window pass: byte offset (inside a window buffer) of the hi piece of the row the tap reads, or of
    // the zero row; per window row ty x tap tx x fragment.  anym: bit (ty*3 + tx)*2 + pt = some lane of my fragment pt has the tap
    // (16 bits each - a window buffer is < 64 KB -, fragments 2 p and 2 p + 1 share a register: registers are what limits this kernel)
    static_assert(C::WIN_BYTES < 65536 && PT % 2 == 0, "packed row addresses");
    unsigned int radr[3][3][PT / 2];
    unsigned int anym = 0u;
    struct Frag { v4u c_hi[CT], c_lo[CT], p_hi[PT], p_lo[PT]; unsigned int any; };       // any: bit pt = some lane of fragment pt has the tap
    Frag fa, fb, fp;
    fp.any = 0u;
#pragma unroll
    for (int i = 0; i < CT; ++i) fp.c_hi[i] = fp.c_lo[i] = v4u{0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < PT; ++i) fp.p_hi[i] = fp.p_lo[i] = v4u{0u, 0u, 0u, 0u};

    // A tap = FS slots.  Slot i issues MFMA i of the tap - term i / (CT*PT) (lo.hi, hi.lo, hi.hi), then fragment, then channel
    // fragment: consecutive MFMAs go to different accumulators, each accumulator still receives its three terms in order - and then
    // ONE filler: an LDS read of the next tap's operands, a direct load, a table word.  The vector-memory and LDS instructions are
    // thus issued in the shadow of an MFMA instead of as blocks that stall both waves of a SIMD at once (measured: with the loads of a
    // step issued back to back behind a tap, 20-30 % of the kernel was load issue).
    constexpr int FS = 3 * CT * PT;
    // loads a step issues: (first step of a stage) the next stage's window runs and table words, then the weight runs D steps ahead;
    // op j is issued by tap block j / OPB, behind its MFMAs
    auto nops_of = [](int q) constexpr { return (q == 0 ? C::WPWIN + 3 * PT : 0) + C::WPW; };
    auto opb_of = [nops_of](int q) constexpr { return (nops_of(q) + TAPS - 1) / TAPS; };
    // ... and how many of them block t of step q issues (gather mode: the loads younger than the block's last B-operand load)
    auto younger_of = [nops_of, opb_of](int q, int t) constexpr {
        const int left = nops_of(q) - t * opb_of(q);
        return left < 0 ? 0 : (left < opb_of(q) ? left : opb_of(q));        // (all of a block's loads are issued behind its operand reads)
    };
    auto mfma_slot = [&](const Frag &f, auto i_t) {
        constexpr int i = decltype(i_t)::value, term = i / (CT * PT), pt = (i % (CT * PT)) / CT, ct = i % CT;
        if constexpr (term < 3 - M::TERMS) return;
        if constexpr (DIAG & 1) { asm volatile("" ::"v"(f.c_hi[ct]), "v"(f.c_lo[ct]), "v"(f.p_hi[pt]), "v"(f.p_lo[pt])); return; }
        acc[ct][pt] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[pt] : f.p_hi[pt], acc[ct][pt]);
    };
    auto tap_block = [&](const Frag &f, auto &&fill) {        // fill(slot): the filler behind slot's MFMA
        // (a tap none of my lanes has issues no MFMAs - wave-uniform; one small branch per slot rather than two copies of the block:
        // the copies' register allocation does not fit the 256 registers of a wave)
        auto one = [&](auto i_t) {
            if ((f.any >> ((decltype(i_t)::value % (CT * PT)) / CT)) & 1u) mfma_slot(f, i_t);
            fill(i_t);
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
        one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
        if constexpr (FS > 6) {
            one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{}); one(std::integral_constant<int, 8>{});
            one(std::integral_constant<int, 9>{}); one(std::integral_constant<int, 10>{}); one(std::integral_constant<int, 11>{});
        }
    };
    static_assert(FS == 6 || FS == 12, "slots per tap");

    unsigned long long tm_vm = 0ull, tm_bar = 0ull, tm_step = 0ull, tm_epi = 0ull, tm_gen = 0ull, tm_n = 0ull;
    const unsigned long long tm_start = (DIAG & 512) ? __builtin_readcyclecounter() : 0ull;
    XStage cur = gen();
    XStage nxt = gen();
    __syncthreads();
    int ws = 0;                 // weight slot of the current step
    bool prev_gather = false;   // the stage before `cur` ran in gather mode
    auto slot_of = [&](int ahead) { const int v = ws + ahead; return v >= C::NSLOT ? v - C::NSLOT : v; };
    // prologue: stage 0's window + words, the weights of its first D steps (same order as in the loop: window, words, weights)
    if (cur.live) {
        auto all_win = [&](auto self, auto i_t) -> void {
            constexpr int i = decltype(i_t)::value;
            if constexpr (i < C::WPWIN) { issue_win(cur, i_t); self(self, std::integral_constant<int, i + 1>{}); }
        };
        all_win(all_win, std::integral_constant<int, 0>{});
        auto all_pw = [&](auto self, auto j_t) -> void {
            constexpr int j = decltype(j_t)::value;
            if constexpr (j < 3 * PT) { issue_pw(cur, std::integral_constant<int, j / 3>{}, std::integral_constant<int, j % 3>{}); self(self, std::integral_constant<int, j + 1>{}); }
        };
        all_pw(all_pw, std::integral_constant<int, 0>{});
        auto all_w = [&](auto self, int q, auto i_t) -> void {
            constexpr int i = decltype(i_t)::value;
            if constexpr (i < C::WPW) { issue_w(cur, q, q, i_t); self(self, q, std::integral_constant<int, i + 1>{}); }
        };
        all_w(all_w, 0, std::integral_constant<int, 0>{});
        if constexpr (D >= 2) all_w(all_w, 1, std::integral_constant<int, 0>{});
        if constexpr (D >= 3) all_w(all_w, 2, std::integral_constant<int, 0>{});
    }

    // one step = TAPS taps of the stage (window row Q, or the whole slab): wait for its data, (re)build its row addresses, run the
    // taps.  The MFMAs of a step's last tap are issued AFTER the next step's barrier (fp carries its operands across), so no barrier
    // is followed by a cold start.
    auto step = [&](auto q_t) {
        constexpr int Q = decltype(q_t)::value;
        const bool gm_prev = Q == 0 ? prev_gather : cur.gather;          // mode of the step before this one
        const bool GM = cur.gather;          // gather mode: the B operands come from global memory, not from the window (one small branch
                                             // per operand read: two copies of the step do not fit the register file)
        constexpr int NV = (D - 1) * C::WPW + ((Q >= 1 && Q <= D - 1) ? C::WINPW : 0);     // younger loads that may stay in flight
        unsigned long long t_top = 0ull;
        if constexpr (DIAG & 512) t_top = __builtin_readcyclecounter();
        if constexpr (!(DIAG & 256)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NV) : "memory");
        if constexpr (DIAG & 512) { const unsigned long long t = __builtin_readcyclecounter(); tm_vm += t - t_top; t_top = t; }
        if constexpr (!(DIAG & 16)) __syncthreads();
        if constexpr (DIAG & 512) { const unsigned long long t = __builtin_readcyclecounter(); tm_bar += t - t_top; t_top = t; }
        if (Q == 0 && cur.tz_first) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                asm volatile("" : "+v"(pn0[pt]), "+v"(pn1[pt]), "+v"(pn2[pt]));
                pw0[pt] = pn0[pt]; pw1[pt] = pn1[pt]; pw2[pt] = pn2[pt];
            }
        }
        // row addresses of window row ty (all three taps, all fragments)
        auto addr_row = [&](auto ty_t) {
            constexpr int ty = decltype(ty_t)::value;
            unsigned int am = 0u;
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const unsigned int e = ty == 0 ? pw0[pt] : (ty == 1 ? pw1[pt] : pw2[pt]);
                const int base = (int)(e & 0x1FFFFFFFu) - cur.wlo;
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) {
                    const int off = base + (tx == 0 ? -1 : tx == 1 ? 0 : (int)((e >> 30) & 1u));
                    const bool valid = ((e >> (29 + tx)) & 1u) != 0u && (GM || (unsigned int)off < (unsigned int)cur.wcnt);
                    if (__ballot(valid) != 0ull) am |= 1u << (tx * 2 + pt);
                    const unsigned int ra = valid ? (unsigned int)off * 64u + (unsigned int)(((2 * kh) ^ ((off >> 2) & 3)) << 4) : (unsigned int)(RCAP * 64);
                    if (pt & 1) radr[ty][tx][pt / 2] |= ra << 16;       // (not read in gather mode)
                    else radr[ty][tx][pt / 2] = ra;
                }
            }
            anym = (anym & ~(63u << (ty * 6))) | (am << (ty * 6));
        };
        const unsigned int win_o = (unsigned int)(C::OFF_WIN + cur.wb * C::WIN_BYTES);
        const unsigned int w_hi_o = (unsigned int)(C::OFF_W + ws * C::WSLOT_BYTES) + wrow + khsw, w_lo_o = w_hi_o ^ 16u;
        // operand reads of tap t of the step into f: weights from the slot, neighbour rows from the window (LDS) or - gather mode -
        // straight from global memory
        auto frag_reads = [&](Frag &f, auto t_t) {
            constexpr int t = decltype(t_t)::value, ty = TAPS == 3 ? Q : t / 3, tx = t % 3;
            f.any = (DIAG & 128) ? 3u : (anym >> ((ty * 3 + tx) * 2)) & 3u;       // (DIAG 128: no tap skipping)
            if constexpr (DIAG & 2) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(f.c_hi[ct]), "+v"(f.c_lo[ct]));
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) asm volatile("" : "+v"(f.p_hi[pt]), "+v"(f.p_lo[pt]));
                return;
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f.c_hi[ct] = *reinterpret_cast<const v4u *>(smem + w_hi_o + (t * COUT + ct * 32) * 64);
                f.c_lo[ct] = *reinterpret_cast<const v4u *>(smem + w_lo_o + (t * COUT + ct * 32) * 64);
            }
            if (GM) {
                // my lane's neighbour row itself: hi / lo half of 8-channel group 2 kc + kh (missing: out of range, zeros).  From inline
                // asm: a load the compiler tracks makes its waitcnt pass drain the whole queue - vmcnt(0) - at the join of the two
                // operand paths, and with it the direct loads in flight; the wait is in tap()
                const srsrc_t rs = prsrc;
                const unsigned int kso = __builtin_amdgcn_readfirstlane((unsigned int)(cur.kc * 64));
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const unsigned int e = ty == 0 ? pw0[pt] : (ty == 1 ? pw1[pt] : pw2[pt]);
                    const unsigned int idx = (e & 0x1FFFFFFFu) + (tx == 0 ? 0xFFFFFFFFu : tx == 1 ? 0u : ((e >> 30) & 1u));
                    const unsigned int vo = ((e >> (29 + tx)) & 1u) ? idx * row_bytes + (unsigned int)(kh * 32) : OOB_OFFSET;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(f.p_hi[pt]) : "v"(vo), "s"(rs), "s"(kso) : "memory");
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(f.p_lo[pt]) : "v"(vo), "s"(rs), "s"(kso) : "memory");
                }
            } else {
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const unsigned int po = win_o + ((pt & 1) ? radr[ty][tx][pt / 2] >> 16 : radr[ty][tx][pt / 2] & 0xFFFFu);
                    f.p_hi[pt] = *reinterpret_cast<const v4u *>(smem + po);
                    f.p_lo[pt] = *reinterpret_cast<const v4u *>(smem + (po ^ 16u));
                }
            }
        };
        // loads of the step, in issue order: (first step of a stage) the next stage's window runs and table words, then the weight
        // runs D steps ahead; op j goes behind MFMA slot (j % OPB) of tap block j / OPB, from the block's last slots backwards
        constexpr int NWIN = Q == 0 ? C::WPWIN : 0, NPW = Q == 0 ? 3 * PT : 0, NOPS = nops_of(Q), OPB = opb_of(Q);
        static_assert(OPB <= FS, "more loads per tap than MFMA slots");
        auto load_op = [&](auto j_t) {
            constexpr int j = decltype(j_t)::value;
            if constexpr (j < NWIN) issue_win(nxt, j_t);
            else if constexpr (j < NWIN + NPW) issue_pw(nxt, std::integral_constant<int, (j - NWIN) / 3>{}, std::integral_constant<int, (j - NWIN) % 3>{});
            else if constexpr (j < NOPS) {
                using I = std::integral_constant<int, j - NWIN - NPW>;
                if constexpr (Q + D < SPS) issue_w(cur, Q + D, slot_of(D), I{});        // step Q + D of this stage
                else issue_w(nxt, Q + D - SPS, slot_of(D), I{});                      // or Q + D - SPS of the next
            }
        };
        // tap block T of the step: the operand reads of tap T (fnew), then the MFMAs of the tap before it (fprev; the reads' latency
        // runs under them), then the block's share of the loads.  One branch for the MFMAs (a tap none of my lanes has issues none).
        // Measured alternatives: one branch per MFMA slot with one filler each, or the MFMAs in two halves around the reads (+15-25 %
        // per step: in this loop every branch costs ~20 cycles), two copies of the block (256 registers + spills at 64 / 128 channels).
        auto tap = [&](Frag &fprev, Frag &fnew, auto t_t) {
            constexpr int T = decltype(t_t)::value;
            if (cur.tz_first && (TAPS == 3 ? T == 0 : T % 3 == 0)) addr_row(std::integral_constant<int, TAPS == 3 ? Q : T / 3>{});
            // gather mode: fprev's B operands are global loads of the previous block (of this step, or the last block of the step
            // before it); younger than the last of them are only the loads that block issued behind its operand reads
            constexpr int QP = T > 0 ? Q : (Q + SPS - 1) % SPS, TP = T > 0 ? T - 1 : TAPS - 1;
            if (T > 0 ? GM : gm_prev) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger_of(QP, TP)) : "memory");
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) asm volatile("" : "+v"(fprev.p_hi[pt]), "+v"(fprev.p_lo[pt]));
            }
            frag_reads(fnew, t_t);
            // (one branch per fragment: a fragment none of whose lanes has the tap - every fragment 1 of a single-unit tile - issues
            // no MFMAs; each MFMA appears once in the code: two copies for the two tile kinds do not fit the register file)
            auto frag_mfmas = [&](auto self, auto i_t, auto pt_t) -> void {
                constexpr int i = decltype(i_t)::value;
                if constexpr (i < FS) {
                    if constexpr ((i % (CT * PT)) / CT == decltype(pt_t)::value) mfma_slot(fprev, i_t);
                    self(self, std::integral_constant<int, i + 1>{}, pt_t);
                }
            };
            if (fprev.any & 1u) frag_mfmas(frag_mfmas, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
            if (fprev.any & 2u) frag_mfmas(frag_mfmas, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
            {
                auto ops = [&](auto self, auto o_t) -> void {
                    constexpr int o = decltype(o_t)::value;
                    if constexpr (o < OPB && T * OPB + o < NOPS) { load_op(std::integral_constant<int, T * OPB + o>{}); self(self, std::integral_constant<int, o + 1>{}); }
                };
                ops(ops, std::integral_constant<int, 0>{});
            }
        };
        tap(fp, fa, std::integral_constant<int, 0>{});
        tap(fa, fb, std::integral_constant<int, 1>{});
        tap(fb, fp, std::integral_constant<int, 2>{});
        if constexpr (TAPS == 9) {
            tap(fp, fa, std::integral_constant<int, 3>{});
            tap(fa, fb, std::integral_constant<int, 4>{});
            tap(fb, fp, std::integral_constant<int, 5>{});
            tap(fp, fa, std::integral_constant<int, 6>{});
            tap(fa, fb, std::integral_constant<int, 7>{});
            tap(fb, fp, std::integral_constant<int, 8>{});
        }
        ws = slot_of(1);
        if (Q == SPS - 1 && GM) {
            // the stage's last operands cross the loop edge in fp: make sure they have landed before the compiler may copy registers
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger_of(SPS - 1, TAPS - 1)) : "memory");
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) asm volatile("" : "+v"(fp.p_hi[pt]), "+v"(fp.p_lo[pt]));
        }
        if constexpr (DIAG & 512) { tm_step += __builtin_readcyclecounter() - t_top; ++tm_n; }
    };

    int orow[PT][2];            // output rows of my epilogue items (fragment pt, row (lane >> 2) + 16 i of it), -1 = none
    while (cur.live) {
        const bool tile_last = !nxt.live || nxt.tile_first;
        if (tile_last) {
            // (round 5: the rows of the epilogue items - through the row map when the table is in tap-set order - are fetched at the START OF
            // THE EPILOGUE, not here: as plain loads in front of the stage the compiler's wait-count pass, which does not see the asm loads
            // of the stream, put an `s_waitcnt vmcnt(0)` before them and emptied the prefetch queue at the start of every tile's last
            // stage; as asm loads their results crossed the stage in registers the compiler is free to copy before they land)
        }
        step(std::integral_constant<int, 0>{});
        if constexpr (SPS == 3) {
            step(std::integral_constant<int, 1>{});
            step(std::integral_constant<int, 2>{});
        }
        if (tile_last) {
            // last stage of the tile: finish its last tap, then its window buffer (every wave is done with it after the barrier)
            // stages the epilogue; the loads in flight go to the other window buffer and to other weight slots
            unsigned long long t_e = 0ull;
            if constexpr (DIAG & 512) t_e = __builtin_readcyclecounter();
            if (cur.gather) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(younger_of(SPS - 1, TAPS - 1)) : "memory");
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) asm volatile("" : "+v"(fp.p_hi[pt]), "+v"(fp.p_lo[pt]));
            }
            tap_block(fp, [](auto) {});
            fp.any = 0u;
            __syncthreads();
            {
                const int nu = cur.half ? 1 : 2;
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int pos = (cur.u0 + pt) * C::UR + wp * 32 + (lane >> 2) + 16 * i;
                        orow[pt][i] = (pt < nu && pos < m) ? (a.perm ? a.perm[pos] : pos) : -1;
                    }
            }
            int next_ticket = 0;            // the ticket of the tile after next (thread 0; read at the end of the epilogue)
            if (tid == 0) next_ticket = take_ticket();
            if constexpr (DIAG & 32) {
#pragma unroll
                for (int i = 0; i < CT; ++i)
#pragma unroll
                    for (int j = 0; j < PT; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));
            } else
            x_store_tile<C, M>(acc, smem + C::OFF_WIN + cur.wb * C::WIN_BYTES, sc_s, sh_s, a.cout, a.relu != 0,
                               reinterpret_cast<const unsigned char *>(a.residual), reinterpret_cast<unsigned char *>(a.out), wc, lane, wid, orow);
#pragma unroll
            for (int i = 0; i < CT; ++i)
#pragma unroll
                for (int j = 0; j < PT; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
            if (tid == 0) tk_s[cur.seq & 1] = next_ticket;
            if constexpr (DIAG & 512) tm_epi += __builtin_readcyclecounter() - t_e;
            if (!nxt.live) break;
        }
        prev_gather = cur.gather;
        cur = nxt;
        if constexpr (DIAG & 512) {
            const unsigned long long t_g = __builtin_readcyclecounter();
            nxt = gen();
            tm_gen += __builtin_readcyclecounter() - t_g;
        } else nxt = gen();
    }
    if constexpr (DIAG & 512) {
        if (a.dbg && lane == 0) {
            unsigned long long *d = a.dbg + ((size_t)blockIdx.x * NW + wid) * 8;
            d[0] = tm_vm; d[1] = tm_bar; d[2] = tm_step; d[3] = tm_epi; d[4] = tm_gen; d[5] = tm_n; d[6] = __builtin_readcyclecounter() - tm_start; d[7] = 1ull;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the last workgroup to get here resets the queues (every other one has taken its last ticket: its atomics have returned)
    if (tid == 0) {
        if (atomicAdd(a.queue + 8, 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 10; ++i) a.queue[i] = 0;
        }
    }
}

// windows of the units: one workgroup per unit; exact first / last referenced input row of every z slab.
// With nbr_sorted / perm (optional): the rows of the unit in the order of their TAP SETS (the 27-bit presence mask as an integer;
// ascending in even units, descending in odd ones - the two units of a tile give every wave one light and one heavy fragment):
// perm[unit * rows + i] = output row at sorted position i, nbr_sorted = the packed table re-ordered the same way (zero words for
// positions past the level's last row).  The convolution's window is LDS-resident and its row map is per lane, so the order of a
// tile's rows is free - and with rows of similar tap sets in one 32-row fragment, 11-13 % fewer (fragment, tap) pairs have any lane
// to multiply for (tools/xrun_stats: 22.6 -> 19.9-20.4 taps per row at levels 3 / 4).
__global__ __launch_bounds__(256) void k_xwin(const int *__restrict__ nbr, int cap, const int *__restrict__ d_m_out, int tile_rows,
                                              int *__restrict__ win, int *__restrict__ nbr_sorted, int *__restrict__ perm) {
    __shared__ int lo_s[3], hi_s[3];
    __shared__ __attribute__((aligned(16))) unsigned int key_s[256];
    if (blockIdx.x == 0 && threadIdx.x < 16) win[(size_t)gridDim.x * 6 + threadIdx.x] = 0;       // the tile queues of dz_spconv_forward_split_x
    const int m = min(*d_m_out, cap);
    const int tile = blockIdx.x, row0 = tile * tile_rows;
    if (threadIdx.x < 3) { lo_s[threadIdx.x] = 0x7FFFFFFF; hi_s[threadIdx.x] = -1; }
    __syncthreads();
    const int row1 = min(m, row0 + tile_rows);
    // one row per thread (tile_rows <= 256): its nine table words, read once
    const int i = threadIdx.x, row = row0 + i;
    const bool mine = i < tile_rows && row < row1;
    unsigned int words[9];
    unsigned int mask = 0u;
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        words[g] = mine ? (unsigned int)nbr[(size_t)g * cap + row] : 0u;
        mask |= (words[g] >> 29) << (3 * g);
    }
#pragma unroll
    for (int tz = 0; tz < 3; ++tz) {
        int lo = 0x7FFFFFFF, hi = -1;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
            const unsigned int e = words[tz * 3 + ty];
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
    // ---- tap-set order: rank of my (mask, row) key among the unit's keys by counting (every thread walks the 256 keys in LDS:
    // broadcast reads, no exchange stages, no barriers in the loop).  Rows past the level's end sort LAST in either direction: the
    // live rows of the last unit keep the positions row0 .. m - 1, so "position < m" stays the test for a live position and nothing is
    // written past the table's `cap` columns
    const unsigned int k27 = (tile & 1) ? (0x7FFFFFFu - mask) : mask;            // odd units descending
    const unsigned int key = mine ? k27 : 0x8000000u;
    key_s[i] = key;
    __syncthreads();
    if (threadIdx.x < 3) {
        const int tz = threadIdx.x;
        int lo = lo_s[tz], n = hi_s[tz] >= 0 ? hi_s[tz] - lo + 1 : 0;
        if (n == 0) lo = 0;
        if (tz == 1 && n == 0 && row0 < m) n = 1;        // a live unit always runs its centre slab (the kernel's step stream needs one stage per tile)
        win[(size_t)tile * 6 + 2 * tz] = lo;
        win[(size_t)tile * 6 + 2 * tz + 1] = n;
    }
    if (!perm || i >= tile_rows) return;
    int rank = 0;
    for (int j = 0; j < tile_rows; j += 4) {
        const uint4 v = *reinterpret_cast<const uint4 *>(&key_s[j]);
        rank += (v.x < key || (v.x == key && j < i)) + (v.y < key || (v.y == key && j + 1 < i)) + (v.z < key || (v.z == key && j + 2 < i)) +
                (v.w < key || (v.w == key && j + 3 < i));
    }
    const int pos = row0 + rank;                                           // my row's position in the unit's tap-set order
    if (pos < cap) {
        perm[pos] = row;
#pragma unroll
        for (int g = 0; g < 9; ++g) nbr_sorted[(size_t)g * cap + pos] = (int)words[g];
    }
}

using X32 = XCfg<32, 8, 1, 2, 9, 1, 896>;
using X64 = XCfg<64, 8, 1, 2, 3, 2, 896>;
using X128 = XCfg<128, 4, 2, 2, 3, 2, 640>;
// development variants of the 32-channel configuration (DZ_TUNE_X32 = 1 / 2): 256-thread workgroups whose LDS lets TWO of them share a
// CU (one's epilogue / barrier waits under the other's steps): one window row per step with the full window, or whole slabs from a
// short window
using X32B = XCfg<32, 4, 1, 2, 3, 2, 448>;
using X32C = XCfg<32, 4, 1, 2, 9, 1, 320>;
static int x_run_len() {          // tiles per XCD run (development knob; a power of two in [1, 64])
    static const int v = getenv("DZ_TUNE_XRUN") ? atoi(getenv("DZ_TUNE_XRUN")) : 8;
    return v >= 1 && v <= 64 && (v & (v - 1)) == 0 ? v : 8;
}
static int x_steal() { static const int v = getenv("DZ_TUNE_X_STEAL") ? atoi(getenv("DZ_TUNE_X_STEAL")) : 1; return v; }
// single-unit tickets per workgroup at the end of a launch: -1 = by the launch's size (k_spconv_x), else forced (development knob).
// A/B at 32 frames per pass: 0 singles 1063.8 frames/s against 1059.6 with one per workgroup (round 4)
static int x_singles() { static const int v = getenv("DZ_TUNE_X_SINGLES") ? atoi(getenv("DZ_TUNE_X_SINGLES")) : -1; return v >= -1 && v <= 8 ? v : -1; }
static int x_sload() { static const int v = getenv("DZ_TUNE_X_SLOAD") ? atoi(getenv("DZ_TUNE_X_SLOAD")) : 1; return v; }
static int x32_variant() {
    static const int v = getenv("DZ_TUNE_X32") ? atoi(getenv("DZ_TUNE_X32")) : 0;
    return v;
}

template <class C, class M, int DIAG = 0>
static int launch_x(const SpConvXArgs &a, hipStream_t stream) {
    static PerDeviceFlags done;
    if (int rc = reserve_lds(reinterpret_cast<const void *>(&k_spconv_x<C, M, DIAG>), C::LDS_BYTES, done, "dz_spconv_forward_split_x")) return rc;
    int grid = ceil_div(a.cap, C::UR);
    const int cus = device_cus() * (C::THREADS <= 256 && 2 * C::LDS_BYTES <= 160 * 1024 ? 2 : 1);      // (workgroups that fit a CU in pairs)
    if (grid > cus) grid = cus;
    grid = (grid + 7) & ~7;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL((k_spconv_x<C, M, DIAG>), dim3(grid), dim3(C::THREADS), C::LDS_BYTES, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

[[maybe_unused]] static unsigned long long *x_dbg_buf = nullptr;      // (diag builds) cycle sums of the last timed launch

template <class C>
static int x_diag(const SpConvXArgs &a, hipStream_t stream, int d) {
    switch (d) {
        case 1: return launch_x<C, MathF16, 1>(a, stream);
        case 3: return launch_x<C, MathF16, 3>(a, stream);
        case 4: return launch_x<C, MathF16, 4>(a, stream);
        case 8: return launch_x<C, MathF16, 8>(a, stream);
        case 12: return launch_x<C, MathF16, 12>(a, stream);
        case 16: return launch_x<C, MathF16, 16>(a, stream);
        case 32: return launch_x<C, MathF16, 32>(a, stream);
        case 15: return launch_x<C, MathF16, 15>(a, stream);
        case 128: return launch_x<C, MathF16, 128>(a, stream);
        case 144: return launch_x<C, MathF16, 144>(a, stream);
        case 256: return launch_x<C, MathF16, 256>(a, stream);
        case 2: return launch_x<C, MathF16, 2>(a, stream);
        case 258: return launch_x<C, MathF16, 258>(a, stream);
        case 259: return launch_x<C, MathF16, 259>(a, stream);
        case 271: return launch_x<C, MathF16, 271>(a, stream);
        case 287: return launch_x<C, MathF16, 287>(a, stream);
        case 319: return launch_x<C, MathF16, 319>(a, stream);
        case 447: return launch_x<C, MathF16, 447>(a, stream);
        case 512: return launch_x<C, MathF16, 512>(a, stream);
        default: return launch_x<C, MathF16>(a, stream);
    }
}

template <class M>
static int x_dispatch(const SpConvXArgs &a, hipStream_t stream) {
#ifdef DZ_SPCONV_DIAG
    static const int diag = getenv("DZ_TUNE_X_DIAG") ? atoi(getenv("DZ_TUNE_X_DIAG")) : 0;
    if (diag && M::ID == 1 && M::TERMS == 3) {
        SpConvXArgs b = a;
        if (diag & 512) {
            static unsigned long long *dbg = nullptr;
            if (!dbg && hipMalloc(&dbg, 256 * 8 * 8 * sizeof(unsigned long long)) != hipSuccess) dbg = nullptr;
            if (dbg) (void)hipMemsetAsync(dbg, 0, 256 * 8 * 8 * sizeof(unsigned long long), stream);
            b.dbg = dbg;
            x_dbg_buf = dbg;
        }
        if (a.cout == 32) return x_diag<X32>(b, stream, diag);
        if (a.cout == 64) return x_diag<X64>(b, stream, diag);
        return x_diag<X128>(b, stream, diag);
    }
#endif
    if (a.cout == 32) {
        if (x32_variant() == 1) return launch_x<X32B, M>(a, stream);
        if (x32_variant() == 2) return launch_x<X32C, M>(a, stream);
        return launch_x<X32, M>(a, stream);
    }
    if (a.cout == 64) return launch_x<X64, M>(a, stream);
    return launch_x<X128, M>(a, stream);
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_spconv_x_tile_rows(int cin, int cout) {
    if (cin != cout) return 0;
    if (cout == 32) return x32_variant() == 1 ? X32B::UR : (x32_variant() == 2 ? X32C::UR : X32::UR);
    if (cout == 64) return X64::UR;
    if (cout == 128) return X128::UR;
    return 0;
}

int dz_spconv_x_window_rows(int cin, int cout) {
    if (cin != cout) return 0;
    if (cout == 32) return x32_variant() == 1 ? X32B::RCAP : (x32_variant() == 2 ? X32C::RCAP : X32::RCAP);
    if (cout == 64) return X64::RCAP;
    if (cout == 128) return X128::RCAP;
    return 0;
}

size_t dz_spconv_x_windows_words(int cap_out, int tile_rows) {
    return tile_rows > 0 && cap_out >= 0 ? (size_t)ceil_div(cap_out, tile_rows) * 6 + 16 : 0;
}

int dz_spconv_x_windows(const int *nbr_packed, int cap_out, const int *d_m_out, int tile_rows, int *windows, int *nbr_sorted, int *perm,
                        void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(nbr_packed && d_m_out && windows, "dz_spconv_x_windows: null pointer");
    DZ_CHECK_ARG(tile_rows > 0 && tile_rows % 32 == 0 && tile_rows <= 256 && cap_out >= 0, "dz_spconv_x_windows: bad tile_rows %d / cap %d", tile_rows, cap_out);
    DZ_CHECK_ARG((nbr_sorted == nullptr) == (perm == nullptr), "dz_spconv_x_windows: nbr_sorted and perm come together");
    if (cap_out == 0) return DZ_OK;
    hipLaunchKernelGGL(k_xwin, dim3(ceil_div(cap_out, tile_rows)), dim3(256), 0, stream, nbr_packed, cap_out, d_m_out, tile_rows, windows, nbr_sorted, perm);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_spconv_forward_split_x(const float *in, int in_rows, int cin, const int *nbr_packed, const int *perm, int *windows, int tile_rows,
                              int cap_out, const int *d_m_out, const float *w, const float *scale, const float *shift, const float *residual, int relu,
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
                  (unsigned int)in_bytes, (unsigned int)w_bytes, (unsigned int)nbr_bytes, nullptr, perm,
                  windows + (size_t)ceil_div(cap_out, tile_rows) * 6, x_run_len(), x_steal(), x_singles(), x_sload()};
    if (math == DZ_MATH_F16) return x_dispatch<MathF16H>(a, stream);
    return math == DZ_MATH_F16X2 ? x_dispatch<MathF16>(a, stream) : x_dispatch<MathBF16>(a, stream);
}

#ifdef DZ_SPCONV_DIAG
/* (diag builds only, not in the header) cycle sums of the last DZ_TUNE_X_DIAG=512 launch, averaged over the waves that ran */
int dz_spconv_x_debug_dump(void) {
    if (!x_dbg_buf) return -1;
    static unsigned long long h[256 * 8 * 8];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, x_dbg_buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    double s[7] = {0, 0, 0, 0, 0, 0, 0}, mx = 0;
    int n = 0;
    for (int w = 0; w < 256 * 8; ++w) {
        if (!h[w * 8 + 7]) continue;
        ++n;
        for (int k = 0; k < 7; ++k) s[k] += (double)h[w * 8 + k];
        if ((double)h[w * 8 + 6] > mx) mx = (double)h[w * 8 + 6];
    }
    if (!n) return -3;
    printf("x-dbg: waves %d  steps/wave %.0f | per step (cycles): vmcnt wait %.0f  barrier %.0f  body %.0f | per wave total: epilogue %.0f  gen %.0f  kernel %.0f (max %.0f)\n", n, s[5] / n,
           s[0] / s[5], s[1] / s[5], s[2] / s[5], s[3] / n, s[4] / n, s[6] / n, mx);
    fflush(stdout);
    return 0;
}
#endif

const char *dz_spconv_x_variant(int cin, int cout) {
    if (cin == 32 && cout == 32) return "k_spconv_x<32>";
    if (cin == 64 && cout == 64) return "k_spconv_x<64>";
    if (cin == 128 && cout == 128) return "k_spconv_x<128>";
    return "none";
}

}  // extern "C"
