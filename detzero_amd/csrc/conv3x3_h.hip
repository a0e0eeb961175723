// 3x3 stride-1 dense BEV convolution on pair16 operands with an image-tile-resident input (the bulk of the BEV
// backbone and head: backbone2d.py:41-62, center_head.py:14-48,81-102).
//
// conv2d_h.hip stages, per (tap, channel chunk), 128 pixels x 128 B of input next to the weight slice: 9 taps
// re-fetch (almost) the same pixels, and the measured limit of that kernel is L2->LDS traffic (10-23 TB/s), not the
// matrix pipe.  Here a 512-thread workgroup owns an 8-row x 32-column tile of output pixels (256 pixels) x BC output
// channels and keeps the tile's (8+2) x (32+2) input pixels of the current 32-channel chunk RESIDENT in LDS: all nine
// taps read their operand fragments from it at a row / column offset (a 32-pixel MFMA fragment = 32 consecutive x of
// one image row, so a tap shift is a pointer offset), and only the 3x3 weight slices stream through the NS-deep
// register pipeline + double-buffered LDS of hgemm.h.  Input bytes per MAC drop 6.6x, total staged bytes ~3x; a wave
// issues 24 MFMAs per tap against 2 global loads, 2 LDS writes and 16 LDS reads.
//   waves: 4 (pairs of image rows) x 2 (halves of BC); wave tile = 2 x 32 pixels x BC/2 channels.
//   per channel chunk: 9 taps, fully unrolled (stage indices, LDS offsets and vmcnt counts are static);
//   the next chunk's input tile is prefetched into registers during the taps and swapped in at the chunk boundary.
#include <stdlib.h>

#include <type_traits>

#include "hgemm.h"

namespace dz {

constexpr int C3_TW = 32, C3_KC = 32;
constexpr int C3_ROW_U4 = C3_KC / 4 + 1;                          // 144-byte LDS rows (conflict-free ds_read_b128)
constexpr int C3_PXW = C3_TW + 2;
constexpr int C3_NS = 3;                                          // weight stages in flight (9 taps % 3 == 0)
using v2u = __attribute__((ext_vector_type(2))) unsigned int;

// NT threads: 512 = 4 (pairs of image rows) x 2 (halves of BC) waves, one workgroup per CU; 256 = 4 x 1 waves, TWO workgroups
// per CU whose barrier phases and epilogues interleave on the matrix pipe
// PT = image rows per wave: 2 at two waves per SIMD; 3 (with NT = 256 and all BC channels in every wave: 3 x 4 fragments, 12
// accumulators) at ONE wave per SIMD with the whole register file - fewer LDS fragment reads and staged bytes per MFMA
// BC = 32 (WC = 1, 8 row pairs: 16 x 32-pixel tiles): grouped convolutions with few output channels per group - the head's output
// layer (6 groups of 64 -> 1..3 channels): memory-bound, every group reads its 64-channel slice of the input exactly once
template <int BC, int NT, int PT_>
struct C3Cfg {
    static constexpr int PT = PT_;
    static constexpr int WC = (PT == 2 && BC >= 64) ? NT / 256 : 1;             // channel splits over waves
    static constexpr int WP = NT / 64 / WC;                       // row groups
    static constexpr int TH = WP * PT;                            // tile height
    static constexpr int CT = BC / (32 * WC);                     // 32-channel fragments per wave
    static constexpr int PXH = TH + 2, PX_ROWS = C3_PXW * PXH, PX_PIECES = PX_ROWS * (C3_KC / 4);
    static constexpr int PXPT = (PX_PIECES + NT - 1) / NT;        // input pieces per thread (last partly idle)
    static constexpr int W_PIECES = BC * (C3_KC / 4);
    static constexpr int WPT = (W_PIECES + NT - 1) / NT;          // weight pieces per thread per tap (2 or 1; BC = 32: 1, half the threads idle)
    static constexpr int W_U4 = BC * C3_ROW_U4;                   // one weight buffer
    static constexpr int LDS_MAIN_BYTES = (PX_ROWS * C3_ROW_U4 + 2 * W_U4) * 16;
    // epilogue staging window of a wave: 32 pixels x SG 8-channel groups (32 bytes each) + 16 bytes of padding
    static constexpr int SG = (NT == 512 && PT == 2) ? 4 : 2;
    static constexpr int STG_ROW = SG * 32 + 16, STG_BYTES = 32 * STG_ROW;
    static constexpr int LDS_STG_END = LDS_MAIN_BYTES + (NT / 64) * STG_BYTES;
    static constexpr int LDS_BYTES = LDS_STG_END + 2 * BC * 4;                 // + BatchNorm scale / shift of the channel tile
    static_assert(BC == 32 || BC == 64 || BC == 128, "BC is 32, 64 or 128");
    static_assert(CT == 1 || CT == 2 || CT == 4, "1, 2 or 4 channel fragments per wave");
    static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS of a CU");
    static_assert(W_PIECES % NT == 0 || W_PIECES < NT, "weight slice must split evenly over the threads");
};

// DIAG (-DDZ_C3_DIAG builds only, timing experiments, results are garbage): bit 0 = no per-tap barriers, 1 = no weight LDS
// stores, 2 = no fragment LDS reads, 3 = no global loads, 4 = no MFMAs, 5 = no epilogue
// SPARSE (dz_conv2d_desc.in_rowidx, round 5): the input image is never materialised.  `in` holds the rows of a sparse level
// (in_row_channels pair16 channels each) and in_rowidx the row of every (pixel, z slab) of the zero-bordered image, -1 = empty:
// input channel chunk kc of a pixel is chunk kc % (nk / 2) of the row of slab kc / (nk / 2).  A thread keeps the two row indices of
// each of its PXPT pieces' pixels in registers (one 8-byte load per piece and TILE, issued behind the last input prefetch that
// needs the current tile's indices, one channel chunk before the next tile's first prefetch reads them); everything downstream of
// the input prefetch - LDS tile, fragments, MFMAs, epilogue - is the dense kernel.  This is HeightCompression
// (height_compression.py:20-24) + the ZeroPad2d of the first BEV block (backbone2d.py:41-46) fused into that block's convolution.
template <int BC, class M, bool OUT_F32, int NT = 512, int DIAG = 0, int PT = 2, bool SPARSE = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(PT == 2 ? 2 : 1, PT == 2 ? 2 : 1))) void k_conv3x3_h(dz_conv2d_desc p, int tiles_x, int tiles_y, unsigned int in_bytes,
                                                          unsigned int w_bytes, int skew_ticks, int q_sa, int q_sb, float q_act, int pair0, int ny) {
    using C = C3Cfg<BC, NT, PT>;
    constexpr int CT = C::CT, WPT = C::WPT, C3_THREADS = NT, C3_PXPT = C::PXPT, WC = C::WC, C3_TH = C::TH, C3_PX_ROWS = C::PX_ROWS,
                  C3_PX_PIECES = C::PX_PIECES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v4u *const px_s = reinterpret_cast<v4u *>(smem_raw);                     // [340][ROW_U4]
    v4u *const w_s = px_s + C3_PX_ROWS * C3_ROW_U4;                          // [2][BC][ROW_U4]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wp = wid / WC, wc = wid % WC;                                  // image-row pair, channel half
    // Persistent workgroups, XCD-aware: the dispatcher places workgroup b on XCD b % 8 (own L2 each); XCD k walks the k-th
    // contiguous eighth of the pixel tiles, its workgroups side by side (neighbouring tiles share halo rows in that L2).
    // A workgroup keeps ONE channel tile (its weight stream simply wraps around from tile to tile) and the load pipeline
    // never drains between tiles: the next tile's input is prefetched during the last channel chunk of the current one.
    // this launch covers the (group, channel tile) pairs pair0 .. pair0 + ny - 1 of the layer's ntn * groups (all of them unless their
    // number does not divide the 32 workgroups of an XCD: launch_c3_nt)
    const int ntn = p.cout_pad / BC, nty = ny;                  // channel tiles per group; pairs of this launch
    // pixel tiles: all of them, or - with a tile list (dz_bev_tile_list) - the tiles to run; the others' result (the layer's zero-input
    // response) is written by dz_bev_fill_empty_tiles
    const int *const tlist = p.in_tiles;
    const int npx = tlist ? tlist[0] : p.batch * tiles_x * tiles_y;
    const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3, nj = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const int grp = (pair0 + jloc % nty) / ntn;
    const int n0 = ((pair0 + jloc % nty) % ntn) * BC;
    const int per_xcd = (npx + 7) >> 3;
    const int band_lo = xcd * per_xcd, band_hi = min(npx, band_lo + per_xcd);
    const int tstep = nj / nty;                                   // workgroups of this XCD that share my channel tile
    int tile = band_lo + jloc / nty;
    if (tstep == 0 || tile >= band_hi) return;
    if (NT == 256 && skew_ticks > 0 && jloc >= nj / 2) {          // second workgroup of a CU: start out of phase with the first
        const unsigned long long t0 = wall_clock64();
        while (wall_clock64() - t0 < (unsigned long long)skew_ticks) __builtin_amdgcn_s_sleep(32);
    }

    const srsrc_t prsrc = make_srsrc(p.in, in_bytes);
    const srsrc_t crsrc = make_srsrc(p.w, w_bytes);
    int x0, y0, b;
    // position in the launch's tile sequence -> tile id.  With a tile list the id comes through the SCALAR cache (one s_load_dword + an
    // lgkmcnt wait): a plain load here is a vector load guarded by `s_waitcnt vmcnt(0)` - it would empty the weight / input prefetch
    // queue twice per tile (tile_geo of the next tile, tile_origin of the epilogue)
    auto tile_id = [&](int t) {
        if (!tlist) return t;
        const int *ptr = tlist + 2 + t;
        int id;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(id) : "s"(ptr) : "memory");
        return id;
    };
    auto tile_origin = [&](int t, int &ox, int &oy, int &ob) {          // t = a tile ID
        ox = (t % tiles_x) * C3_TW;
        oy = ((t / tiles_x) % tiles_y) * C3_TH;
        ob = t / (tiles_x * tiles_y);
    };
    // input pieces of a tile: pixel (y0 + in_off + ry, x0 + in_off + rx) of the padded image, rx < 34, ry < 10.  Piece i of a
    // thread is the same (ry, rx, 16-byte piece) in every tile, so nothing per piece is kept in registers: the byte offset
    // inside the tile is rebuilt from the thread index at every issue (a handful of VALU ops next to 32-cycle MFMAs) and the
    // tile's origin rides in the scalar offset of the load; pieces past the image edge get an out-of-range voffset (zeros).
    struct TileGeo { unsigned int base; int rows, cols; };       // byte offset of the tile's first input pixel; rows / columns left in the image
    auto tile_geo = [&](int t) {
        int ox, oy, ob;
        tile_origin(t, ox, oy, ob);
        TileGeo g;
        if constexpr (SPARSE) g.base = (unsigned int)((ob * p.in_hp + oy + p.in_off) * p.in_wp + ox + p.in_off);       // pixel index
        else g.base = (unsigned int)((((long)(ob * p.in_hp + oy + p.in_off) * p.in_wp + ox + p.in_off) * p.in_cstride + p.in_coff + grp * p.cin) * 4);
        g.rows = p.in_hp - (oy + p.in_off);
        g.cols = p.in_wp - (ox + p.in_off);
        return g;
    };
    int id_cur = tile_id(tile), id_next = id_cur;
    TileGeo geo = tile_geo(id_cur), geo_next = geo;
    const int prow = tid / (C3_KC / 4), pq = tid % (C3_KC / 4);
    bool has_next = tile + tstep < band_hi;
    unsigned int cvoff[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int idx = tid + i * C3_THREADS;
        cvoff[i] = idx < C::W_PIECES ? (unsigned int)((((long)grp * 9 * p.cout_pad + n0 + idx / (C3_KC / 4)) * p.cin + (idx % (C3_KC / 4)) * 4) * 4)
                                     : OOB_OFFSET;
    }
    const unsigned int tap_bytes = (unsigned int)((long)p.cout_pad * p.cin * 4);
    const int nk = p.cin / C3_KC;
    const int nchunks = nk * 9;

    v4u wst[C3_NS][WPT];          // weight stages: chunk j lives in stage j % 3 == tap % 3
    v4u pst[C3_PXPT];             // input tile of the next channel chunk
    auto issue_w = [&](v4u (&st)[WPT], int chunk) {
        // chunk = kc * 9 + tap of the current tile; past its end the stream wraps to the next tile's chunks (same
        // weights), or - after the last tile - to out-of-range offsets (zeros come back, nothing is fetched; the
        // per-wave load counts stay uniform)
        if constexpr (DIAG & 8) return;
        if (chunk >= nchunks) chunk = has_next ? chunk - nchunks : -1;
        const int kc = chunk / 9, tap = chunk - kc * 9;
        const unsigned int add = chunk >= 0 ? (unsigned int)tap * tap_bytes + (unsigned int)(kc * C3_KC * 4) : OOB_OFFSET;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st[i]) : "v"(cvoff[i] == OOB_OFFSET ? OOB_OFFSET : cvoff[i] + add), "s"(crsrc));
    };
    // SPARSE: rows of my pieces' pixels (z slab 0 / 1) in the tile the next issue_px call reads.  The loads are inline asm: their
    // results exist once a counted wait has covered them, so NOTHING may read ridx between issue_idx() and the own_idx() behind
    // that wait - the validity of a piece (tile edge, stream end) therefore travels separately, as a bit mask, and is applied there
    v2u ridx[SPARSE ? C3_PXPT : 1];
    [[maybe_unused]] unsigned int ridx_ok = 0u;
    [[maybe_unused]] const srsrc_t irsrc = make_srsrc(p.in_rowidx, SPARSE ? (unsigned int)((size_t)p.batch * p.in_hp * p.in_wp * 8) : 0u);
    [[maybe_unused]] auto issue_idx = [&](const TileGeo &g, bool live) {
        if constexpr (SPARSE) {
            int pr = prow;
            asm volatile("" : "+v"(pr));
            unsigned int okm = 0u;
#pragma unroll
            for (int i = 0; i < C3_PXPT; ++i) {
                const int r = pr + i * (C3_THREADS / (C3_KC / 4));
                const int ry = r / C3_PXW, rx = r - ry * C3_PXW;
                const bool ok = live && r < C3_PX_ROWS && ry < g.rows && rx < g.cols;
                const unsigned int off = ok ? (g.base + (unsigned int)(ry * p.in_wp + rx)) * 8u : OOB_OFFSET;
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(ridx[i]) : "v"(off), "s"(irsrc));
                okm |= ok ? 1u << i : 0u;
            }
            ridx_ok = okm;
        }
    };
    [[maybe_unused]] auto own_idx = [&]() {
        if constexpr (SPARSE) {
#pragma unroll
            for (int i = 0; i < C3_PXPT; ++i) {
                asm volatile("" : "+v"(ridx[i]));
                if (!((ridx_ok >> i) & 1u)) ridx[i] = v2u{0xFFFFFFFFu, 0xFFFFFFFFu};       // (idempotent: own_idx runs once per channel chunk)
            }
        }
    };
    auto issue_px = [&](int kc) {
        // input tile of channel chunk kc; kc == nk: chunk 0 of the next tile
        if constexpr (DIAG & 8) return;
        if constexpr (SPARSE) {
            const bool cur = kc < nk;
            const bool any = cur || has_next;
            const int kq = cur ? kc : 0, half = nk >> 1;
            const bool z1 = kq >= half;
            const unsigned int sbase = (unsigned int)((kq - (z1 ? half : 0)) * C3_KC * 4 + p.in_coff * 4);
            const unsigned int rowb = (unsigned int)p.in_row_channels * 4u;
#pragma unroll
            for (int i = 0; i < C3_PXPT; ++i) {
                const int ri = (int)(z1 ? ridx[i].y : ridx[i].x);
                const unsigned int off = (any && ri >= 0) ? (unsigned int)ri * rowb + (unsigned int)(pq * 16) : OOB_OFFSET;
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(pst[i]) : "v"(off), "s"(prsrc), "s"(sbase));
            }
            return;
        }
        const bool cur = kc < nk;
        const TileGeo g = cur ? geo : geo_next;
        const unsigned int sbase = g.base + (cur ? (unsigned int)(kc * C3_KC * 4) : 0u);
        const bool any = cur || has_next;
        int pr = prow;
        asm volatile("" : "+v"(pr));             // keeps the per-piece offsets from being hoisted into registers for the whole loop
#pragma unroll
        for (int i = 0; i < C3_PXPT; ++i) {
            const int r = pr + i * (C3_THREADS / (C3_KC / 4));
            const int ry = r / C3_PXW, rx = r - ry * C3_PXW;
            const bool ok = any && r < C3_PX_ROWS && ry < g.rows && rx < g.cols;
            const unsigned int off = ok ? (unsigned int)(((ry * p.in_wp + rx) * p.in_cstride + pq * 4) * 4) : OOB_OFFSET;
            // one wave per SIMD (PT == 3): the prefetched tile waits in the accumulation registers the 12 accumulators leave free
            if constexpr (PT == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(pst[i]) : "v"(off), "s"(prsrc), "s"(sbase));
            else asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=a"(pst[i]) : "v"(off), "s"(prsrc), "s"(sbase));
        }
    };
    auto own_w = [&](v4u (&st)[WPT]) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) asm volatile("" : "+v"(st[i]));
    };
    auto store_w = [&](const v4u (&st)[WPT], int buf) {
        if constexpr (DIAG & 2) return;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + i * C3_THREADS;
            if (C::W_PIECES % NT == 0 || idx < C::W_PIECES) w_s[buf * C::W_U4 + (idx / (C3_KC / 4)) * C3_ROW_U4 + idx % (C3_KC / 4)] = st[i];
        }
    };
    auto store_px = [&]() {
#pragma unroll
        for (int i = 0; i < C3_PXPT; ++i) {
            if constexpr (PT == 2) asm volatile("" : "+v"(pst[i]));
            else asm volatile("" : "+a"(pst[i]));
            const int idx = tid + i * C3_THREADS;
            if (idx < C3_PX_PIECES) px_s[(idx / (C3_KC / 4)) * C3_ROW_U4 + idx % (C3_KC / 4)] = pst[i];
        }
    };

    f32x16 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment base addresses (16-byte units): pixel fragment pt = image row 2*wp + pt, column lane & 31
    const int kg2 = (lane >> 5) * 2;
    const int pbase = ((PT * wp) * C3_PXW + (lane & 31)) * C3_ROW_U4 + kg2;
    const int wbase = (wc * CT * 32 + (lane & 31)) * C3_ROW_U4 + kg2;
    struct Frag { v4u p_hi[PT], p_lo[PT], c_hi[CT], c_lo[CT], p_x[PT], c_x[CT]; };        // (p_x, c_x: second fp8 piece, q16 prototype)
    auto load_frag = [&](Frag &f, int tap, int buf, int q) {
        if constexpr (DIAG & 4) {
#pragma unroll
            for (int i = 0; i < PT; ++i) asm volatile("" : "+v"(f.p_hi[i]), "+v"(f.p_lo[i]));
#pragma unroll
            for (int i = 0; i < CT; ++i) asm volatile("" : "+v"(f.c_hi[i]), "+v"(f.c_lo[i]));
            return;
        }
        const int ky = tap / 3, kx = tap - ky * 3;
        if constexpr (M::Q16) {
            // q16 chunk row (128 bytes = 8 pieces): fp16 hi of channels 8i..8i+7 in piece i (0..3), hi8 in pieces 4-5, lo8 in 6-7.
            // k-step q of the fp16 MFMA: piece 2q + kg.  The fp8 MFMA (issued with k-step 1) covers the 32 channels twice: its
            // k-block 0 (lanes 0-31) multiplies lo8(x) by hi8(w), block 1 (lanes 32-63) hi8(x) by lo8(w).
            const int kb = lane >> 5;
            const v4u *pp = px_s + (pbase - kg2) + (ky * C3_PXW + kx) * C3_ROW_U4;
            const v4u *cp = w_s + buf * C::W_U4 + (wbase - kg2);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                f.p_hi[pt] = pp[pt * C3_PXW * C3_ROW_U4 + 2 * q + kb];
                if (q == 1) {
                    f.p_lo[pt] = pp[pt * C3_PXW * C3_ROW_U4 + (kb ? 4 : 6)];
                    f.p_x[pt] = pp[pt * C3_PXW * C3_ROW_U4 + (kb ? 5 : 7)];
                }
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                f.c_hi[ct] = cp[ct * 32 * C3_ROW_U4 + 2 * q + kb];
                if (q == 1) {
                    f.c_lo[ct] = cp[ct * 32 * C3_ROW_U4 + (kb ? 6 : 4)];
                    f.c_x[ct] = cp[ct * 32 * C3_ROW_U4 + (kb ? 7 : 5)];
                }
            }
            return;
        }
        const v4u *pp = px_s + pbase + (ky * C3_PXW + kx) * C3_ROW_U4 + q * 4;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            f.p_hi[pt] = pp[pt * C3_PXW * C3_ROW_U4];
            f.p_lo[pt] = pp[pt * C3_PXW * C3_ROW_U4 + 1];
        }
        const v4u *cp = w_s + buf * C::W_U4 + wbase + q * 4;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            f.c_hi[ct] = cp[ct * 32 * C3_ROW_U4];
            f.c_lo[ct] = cp[ct * 32 * C3_ROW_U4 + 1];
        }
    };
    auto mma = [&](const Frag &f, int q = 0) {
        if constexpr (M::Q16) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = M::mma(f.c_hi[ct], f.p_hi[pt], acc[ct][pt]);
            if (q == 1) {
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < PT; ++pt) acc[ct][pt] = mma_f8(f.c_lo[ct], f.c_x[ct], f.p_lo[pt], f.p_x[pt], acc[ct][pt], q_sa, q_sb);
            }
            return;
        }
        if constexpr (DIAG & 16) {
#pragma unroll
            for (int i = 0; i < PT; ++i) asm volatile("" ::"v"(f.p_hi[i]), "v"(f.p_lo[i]));
#pragma unroll
            for (int i = 0; i < CT; ++i) asm volatile("" ::"v"(f.c_hi[i]), "v"(f.c_lo[i]));
            return;
        }
        // term-major: consecutive MFMAs go to different accumulators (measured 1.7 % faster than three in a row into the same
        // one); each accumulator still receives lo.hi, hi.lo, hi.hi in that order, so the result does not depend on it
#pragma unroll
        for (int term = 3 - M::TERMS; term < 3; ++term)              // single-product mode: only hi.hi (term 2)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
                    acc[ct][pt] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[pt] : f.p_hi[pt], acc[ct][pt]);
    };

    // BatchNorm scale / shift of my channel tile: read in every tile's epilogue (from global memory each (fragment, group) of
    // the epilogue paid a full load latency: the stores in between may alias, so the compiler cannot hoist the loads)
    float *const sc_s = reinterpret_cast<float *>(smem_raw + C::LDS_STG_END), *const sh_s = sc_s + BC;
    if (tid < BC) {
        const bool in = n0 + tid < p.g_cout[grp];
        sc_s[tid] = (in && p.scale) ? p.scale[grp * p.cout_pad + n0 + tid] : 1.f;
        sh_s[tid] = (in && p.shift) ? p.shift[grp * p.cout_pad + n0 + tid] : 0.f;
    }
    // ---- prologue: input tile of chunk 0 and weight slice of (chunk 0, tap 0) into LDS; taps 1..3 in flight
    if constexpr (SPARSE) {
        issue_idx(geo, true);
        asm volatile("s_waitcnt vmcnt(0)");
        own_idx();
    }
    issue_px(0);
    issue_w(wst[0], 0);
    asm volatile("s_waitcnt vmcnt(0)");
    store_px();
    own_w(wst[0]);
    store_w(wst[0], 0);
    __syncthreads();
    issue_w(wst[1], 1);
    issue_w(wst[2], 2);
    issue_w(wst[0], 3);
    Frag f0, f1;
    load_frag(f0, 0, 0, 0);

    int kcg = 0;                                                 // channel chunks done so far, over all tiles (LDS buffer parity)
    for (;;) {
    if (has_next) { id_next = tile_id(tile + tstep); geo_next = tile_geo(id_next); }
    for (int kc = 0; kc < nk; ++kc, ++kcg) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            constexpr int NW2 = 2 * WPT;                         // two younger weight stages stay in flight
            const int cur = t & 1;                               // 9 taps: the buffer parity flips every chunk, and
            const int buf = (kcg & 1) ? (cur ^ 1) : cur;         // every channel chunk (9 is odd)
            const int c = kc * 9 + t;
            // ---- phase 1: k-step-1 fragments, weights of chunk c+1 to the other LDS buffer, MFMAs of k-step 0
            load_frag(f1, t, buf, 1);
            // loads younger than chunk c+1's: chunks c+2, c+3, plus this chunk's input prefetch while it is the youngest
            if constexpr (!(DIAG & 8)) {
                if (t >= 1 && t <= 3) {
                    // (SPARSE: in chunk nk - 2 the next tile's PXPT index loads are in flight too, younger than the input prefetch)
                    if (SPARSE && kc == nk - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW2 + 2 * C3_PXPT));
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW2 + C3_PXPT));
                } else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW2));
            }
            if constexpr (SPARSE) {
                if (t == 0) own_idx();           // (the indices issued a chunk ago are covered by the wait above)
            }
            own_w(wst[(t + 1) % 3]);
            store_w(wst[(t + 1) % 3], buf ^ 1);
            mma(f0);
            // one LDS read / write behind each of the first MFMAs: the two waves of a SIMD run in lock step (one barrier
            // per tap), so memory instructions issued in a block of their own would leave the matrix pipe idle
            interleave_hint<0x100, M::TERMS == 1 ? (PT + CT) : 2 * (PT + CT), 1>();
            interleave_hint<0x200, WPT, 1>();
            if constexpr (!(DIAG & 1)) __syncthreads();
            if (t == 8) {
                // channel-chunk boundary: every wave has finished reading the old input tile (its last reads were the
                // k-step-1 fragments above, complete before the barrier); swap in the prefetched tile
                store_px();
                __syncthreads();
            }
            // ---- phase 2: fragments of chunk c+1, weights of chunk c+4 into the stage just freed, MFMAs of k-step 1
            load_frag(f0, (t + 1) % 9, buf ^ 1, 0);
            issue_w(wst[(t + 1) % 3], c + 4);
            if (t == 0) {
                issue_px(kc + 1);
                if constexpr (SPARSE) {
                    // the current tile's indices have served their last prefetch: fetch the next tile's into the same registers
                    if (kc == nk - 2) issue_idx(geo_next, has_next);
                }
            }
            mma(f1, 1);
            interleave_hint<0x100, M::TERMS == 1 ? (PT + CT) : 2 * (PT + CT), 1>();
        }
    }
    tile_origin(id_cur, x0, y0, b);

    // ---- epilogue: 32x32 accumulator: pixel column = lane & 31, channel = 8*(reg>>2) + 4*(lane>>5) + (reg&3)
    if constexpr (DIAG & 32) {                                   // no epilogue: the accumulators only have to stay alive
        float sacc = 0.f;
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int j = 0; j < PT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) asm volatile("" ::"v"(acc[i][j][e]));
        if (sacc == 12345.f) p.out[0] = sacc;
    } else {
    const int h = lane >> 5;
    const int gcout = p.g_cout[grp];
    const int ooff = p.out_coff + p.g_ooff[grp];
    if constexpr (!OUT_F32) {
        // pair16 result: the accumulator layout gives a lane 4 of the 8 channels of a group for ONE pixel (8 bytes of hi, 8 of
        // lo, 512 bytes away from its neighbour lane's) - stored directly that is 64 separate 8-byte requests per instruction
        // (measured: 19 % of the kernel).  Each 32-pixel x 32-channel fragment goes through a wave-private LDS window instead and
        // leaves as 128 contiguous bytes per pixel, 8 full lines per store instruction.
        constexpr int SG = C::SG, STG_ROW = C::STG_ROW;                        // 8-channel groups per staging round
        constexpr int LPR = 2 * SG, RPI = 64 / LPR;                            // lanes per pixel row, rows per store instruction
        unsigned char *const stg = smem_raw + C::LDS_MAIN_BYTES + wid * C::STG_BYTES;
        const int srow = lane / LPR, spiece = lane % LPR;
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int y = y0 + PT * wp + pt;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
                for (int j0 = 0; j0 < 4; j0 += SG) {
                    const int cbase = n0 + wc * CT * 32 + ct * 32 + j0 * 8;
#pragma unroll
                    for (int jj = 0; jj < SG; ++jj) {
                        const int j = j0 + jj;
                        const int lc = wc * CT * 32 + ct * 32 + 8 * j + 4 * h;
                        const float4 sc = *reinterpret_cast<const float4 *>(sc_s + lc), sh = *reinterpret_cast<const float4 *>(sh_s + lc);
                        float v[4] = {fmaf(acc[ct][pt][4 * j], sc.x, sh.x), fmaf(acc[ct][pt][4 * j + 1], sc.y, sh.y),
                                      fmaf(acc[ct][pt][4 * j + 2], sc.z, sh.z), fmaf(acc[ct][pt][4 * j + 3], sc.w, sh.w)};
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        if constexpr (M::Q16) {
                            // q16 out (SG == 4: the staged row is the whole 128-byte chunk of the fragment's 32 channels): fp16 hi of
                            // channel c at byte 2c, hi8 at 64 + c, lo8 at 96 + c; c = 8j + 4h + e
                            const int c0 = 8 * j + 4 * h;
                            unsigned int hh[2], h8 = 0u, l8 = 0u;
                            float r[4];
#pragma unroll
                            for (int e2 = 0; e2 < 2; ++e2) {
                                const float a0 = v[2 * e2], a1 = v[2 * e2 + 1];
                                const h2_t hp = __builtin_convertvector(f32x2v{__builtin_amdgcn_fmed3f(a0, -65504.f, 65504.f),
                                                                               __builtin_amdgcn_fmed3f(a1, -65504.f, 65504.f)}, h2_t);
                                hh[e2] = __builtin_bit_cast(unsigned int, hp);
                                const f32x2v hb = __builtin_convertvector(hp, f32x2v);
                                r[2 * e2] = a0 - hb.x; r[2 * e2 + 1] = a1 - hb.y;
                                v[2 * e2] = hb.x; v[2 * e2 + 1] = hb.y;
                            }
                            const float s8 = q_act, sl = q_act * 2048.f;
                            auto clamp8 = [](float x) { return __builtin_amdgcn_fmed3f(x, -448.f, 448.f); };
                            h8 = (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(clamp8(v[0] * s8), clamp8(v[1] * s8), (int)h8, false);
                            h8 = (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(clamp8(v[2] * s8), clamp8(v[3] * s8), (int)h8, true);
                            l8 = (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(clamp8(r[0] * sl), clamp8(r[1] * sl), (int)l8, false);
                            l8 = (unsigned int)__builtin_amdgcn_cvt_pk_fp8_f32(clamp8(r[2] * sl), clamp8(r[3] * sl), (int)l8, true);
                            unsigned char *w = stg + (lane & 31) * STG_ROW;
                            *reinterpret_cast<uint2 *>(w + 2 * c0) = make_uint2(hh[0], hh[1]);
                            *reinterpret_cast<unsigned int *>(w + 64 + c0) = h8;
                            *reinterpret_cast<unsigned int *>(w + 96 + c0) = l8;
                        } else {
                        uint2 hi, lo;
                        split4<M>(v, hi, lo);
                        unsigned char *w = stg + (lane & 31) * STG_ROW + jj * 32 + h * 8;
                        *reinterpret_cast<uint2 *>(w) = hi;
                        *reinterpret_cast<uint2 *>(w + 16) = lo;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const int gcol = cbase + (spiece >> 1) * 8;                // the 8-channel group of my 16-byte piece
#pragma unroll
                    for (int i = 0; i < 32 / RPI; ++i) {
                        const int r = srow + RPI * i, x = x0 + r;
                        const v4u d = *reinterpret_cast<const v4u *>(stg + r * STG_ROW + spiece * 16);
                        if constexpr (DIAG & 64) {                        // no global stores
                            asm volatile("" ::"v"(d));
                            continue;
                        }
                        if (y < p.ho && x < p.wo && gcol < gcout) {
                            const size_t op = ((size_t)b * p.out_hp + (size_t)y * p.out_sy + p.out_dy) * p.out_wp + (size_t)x * p.out_sx + p.out_dx;
                            unsigned char *g = reinterpret_cast<unsigned char *>(p.out) + (op * p.out_cstride + ooff + gcol) * 4 + (spiece & 1) * 16;
                            *reinterpret_cast<v4u *>(g) = d;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
            }
        }
    } else {
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int y = y0 + PT * wp + pt, x = x0 + (lane & 31);
        if (y >= p.ho || x >= p.wo) continue;
        const size_t op = ((size_t)b * p.out_hp + (size_t)y * p.out_sy + p.out_dy) * p.out_wp + (size_t)x * p.out_sx + p.out_dx;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wc * CT * 32 + ct * 32 + 8 * j + 4 * h;
                if (col >= gcout) continue;
                float *o = p.out + op * p.out_cstride + ooff + col;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int lc = col - n0 + e;
                    float v = fmaf(acc[ct][pt][4 * j + e], sc_s[lc], sh_s[lc]);
                    if (p.relu) v = fmaxf(v, 0.f);
                    if (col + e < gcout) o[e] = v;
                }
            }
        }
    }
    }
    }
    // ---- next tile: its first input chunk is already in LDS, its first weight slices are in flight
    if (!has_next) break;
    tile += tstep;
    id_cur = id_next;
    geo = geo_next;
    has_next = tile + tstep < band_hi;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)");      // nothing of this file's asm loads may stay in flight at exit
}

// q16 prototype (diag builds): tensor exponents from the environment (read once): activations scaled by 2^-DZ_TUNE_Q16_EA, weights by
// 2^-DZ_TUNE_Q16_EW; the kernels of every other math type ignore the three arguments
static int q16_env(const char *n, int d) { const char *v = getenv(n); return v ? atoi(v) : d; }
static int q16_sa() { static const int v = 127 + q16_env("DZ_TUNE_Q16_EW", -4); return v; }               // weights are the A operand
static int q16_sb() { static const int v = 127 + q16_env("DZ_TUNE_Q16_EA", 2) - 11; return v; }           // activations the B operand; both correction terms carry 2^-11
static float q16_act() { static const float v = ldexpf(1.f, -q16_env("DZ_TUNE_Q16_EA", 2)); return v; }

template <int BC, class M, bool OUT_F32, int NT, int DIAG = 0, int PT = 2, bool SPARSE = false>
static int launch_c3_nt(const dz_conv2d_desc &p, size_t w_bytes, hipStream_t stream) {
    using C = C3Cfg<BC, NT, PT>;
    const int tiles_x = ceil_div(p.wo, C3_TW), tiles_y = ceil_div(p.ho, C::TH);
    // (SPARSE: `in` is the level's rows - in_rows of them, in_row_channels channels each)
    const size_t in_bytes = SPARSE ? (size_t)p.in_rows * p.in_row_channels * sizeof(float) : (size_t)p.batch * p.in_hp * p.in_wp * p.in_cstride * sizeof(float);
    if (SPARSE && (size_t)p.batch * p.in_hp * p.in_wp * 8 >= 0x80000000ull) {
        set_error("dz_conv2d_forward_split: row-index image exceeds the 2 GiB buffer-addressing limit");
        return DZ_ERR_UNSUPPORTED;
    }
    if (in_bytes >= 0x80000000ull || w_bytes >= 0x80000000ull) {
        set_error("dz_conv2d_forward_split: image of %zu bytes / weights of %zu bytes exceed the 2 GiB buffer-addressing limit", in_bytes, w_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    static PerDeviceFlags lds_done;
    if (int rc_ = reserve_lds(reinterpret_cast<const void *>(&k_conv3x3_h<BC, M, OUT_F32, NT, DIAG, PT, SPARSE>), C::LDS_BYTES, lds_done, "dz_conv2d_forward_split")) return rc_;
    // persistent: 512 / NT workgroups per CU, a multiple of 8 x channel tiles so that every XCD gets the same number of
    // workgroups of every channel tile
    const int nty = p.cout_pad / BC * p.groups;
    const int slots = 32 * (PT == 2 ? 512 / NT : 1);            // workgroups of an XCD
    static const int skew = getenv("DZ_TUNE_C3_SKEW") ? atoi(getenv("DZ_TUNE_C3_SKEW")) : 0;     // 10 ns ticks
    static const int split = getenv("DZ_TUNE_C3_SPLIT") ? atoi(getenv("DZ_TUNE_C3_SPLIT")) : 1;  // development knob: 0 = one launch (r01-r04)
    // A workgroup keeps ONE (group, channel tile) pair, so an XCD runs a multiple of their number: with 3 or 6 pairs (the head's
    // 64 -> 384 layer and its grouped output layer) that is 30 of 32 workgroups - 16 CUs idle for the whole launch.  Such a layer runs
    // as two launches over 2 + 1 / 4 + 2 of its pairs, each on all 256 CUs (round 5: 27 rounds of tiles instead of 28.8)
    int parts[2][2] = {{0, nty}, {0, 0}};
    // (only where the launch is long enough to pay for a second one: at 8 frames per pass - 14 rounds - one launch measured 1.4 % faster)
    const long pair_tiles = (long)p.batch * tiles_x * tiles_y * nty;
    if (split && slots % nty != 0 && nty < slots && pair_tiles >= 5000) {
        int a = 1;
        while (a * 2 <= nty) a *= 2;                            // largest power of two below nty
        if (slots % a == 0 && slots % (nty - a) == 0) { parts[0][1] = a; parts[1][0] = a; parts[1][1] = nty - a; }
    }
    for (int k = 0; k < 2 && parts[k][1] > 0; ++k) {
        const int ny = parts[k][1];
        int per_xcd = slots / ny * ny;
        if (per_xcd < ny) per_xcd = ny;
        const long grid = 8L * per_xcd;
        hipLaunchKernelGGL((k_conv3x3_h<BC, M, OUT_F32, NT, DIAG, PT, SPARSE>), dim3((unsigned int)grid), dim3(NT), C::LDS_BYTES, stream, p, tiles_x,
                           tiles_y, (unsigned int)in_bytes, (unsigned int)w_bytes, skew, q16_sa(), q16_sb(), q16_act(), parts[k][0], ny);
        DZ_LAUNCH_CHECK();
    }
    return DZ_OK;
}

// Shipped configuration: 512 threads (4 row pairs x 2 channel halves), one workgroup per CU.  Built and measured on the MI355X, all
// slower, kept behind -DDZ_C3_DIAG for the record: two 256-thread workgroups per CU (DZ_TUNE_C3_NT=256: their phases
// do not de-synchronise, and a start skew changes nothing - the CU is throughput-bound), one wave per SIMD with 3 x 4 fragments
// (DZ_TUNE_C3_PT=3: fewer LDS reads per MFMA, but every barrier and wait is exposed: 543 vs 471 us), and the diag switches.
template <int BC, class M, bool OUT_F32>
static int launch_c3(const dz_conv2d_desc &p, size_t w_bytes, hipStream_t stream) {
#ifdef DZ_C3_DIAG
    static const int nt = getenv("DZ_TUNE_C3_NT") ? atoi(getenv("DZ_TUNE_C3_NT")) : 512;
    if constexpr (BC == 128 && !OUT_F32 && std::is_same<M, MathF16>::value) {
        static const int diag = getenv("DZ_TUNE_C3_DIAG") ? atoi(getenv("DZ_TUNE_C3_DIAG")) : 0;
        switch (diag) {
            case 1: return launch_c3_nt<BC, M, OUT_F32, 512, 1>(p, w_bytes, stream);
            case 2: return launch_c3_nt<BC, M, OUT_F32, 512, 2>(p, w_bytes, stream);
            case 4: return launch_c3_nt<BC, M, OUT_F32, 512, 4>(p, w_bytes, stream);
            case 8: return launch_c3_nt<BC, M, OUT_F32, 512, 8>(p, w_bytes, stream);
            case 15: return launch_c3_nt<BC, M, OUT_F32, 512, 15>(p, w_bytes, stream);
            case 16: return launch_c3_nt<BC, M, OUT_F32, 512, 16>(p, w_bytes, stream);
            case 32: return launch_c3_nt<BC, M, OUT_F32, 512, 32>(p, w_bytes, stream);
            case 47: return launch_c3_nt<BC, M, OUT_F32, 512, 47>(p, w_bytes, stream);
            case 64: return launch_c3_nt<BC, M, OUT_F32, 512, 64>(p, w_bytes, stream);
            default: break;
        }
        static const int pt = getenv("DZ_TUNE_C3_PT") ? atoi(getenv("DZ_TUNE_C3_PT")) : 2;
        if (pt == 3) return launch_c3_nt<BC, M, OUT_F32, 256, 0, 3>(p, w_bytes, stream);
        static const int q16 = getenv("DZ_TUNE_C3_Q16") ? atoi(getenv("DZ_TUNE_C3_Q16")) : 0;
        if (q16) return launch_c3_nt<BC, MathF16Q, OUT_F32, 512>(p, w_bytes, stream);
    }
    if constexpr (BC == 64) {
        if (nt == 256) return launch_c3_nt<BC, M, OUT_F32, 256>(p, w_bytes, stream);
    }
#endif
    return launch_c3_nt<BC, M, OUT_F32, 512>(p, w_bytes, stream);
}

// 0 = not eligible, 64 / 128 = channel tile of the resident-tile kernel
int conv3x3_h_variant(const dz_conv2d_desc &p) {
    static const int off = getenv("DZ_TUNE_NO_CONV3X3") ? atoi(getenv("DZ_TUNE_NO_CONV3X3")) : 0;
    if (off) return 0;
    if (p.kh != 3 || p.kw != 3 || p.stride != 1 || p.group_shift) return 0;
    if (p.cin % C3_KC != 0) return 0;
    // sparse input (in_rowidx): two z slabs of in_row_channels channels each, 128 output channels per tile, pair16 out
    if (p.in_rowidx && (p.groups != 1 || p.cout_pad % 128 != 0 || p.cin != 2 * p.in_row_channels || p.in_row_channels % C3_KC != 0 || p.in_coff != 0)) return 0;
    if (p.cout_pad == 32) {               // few output channels (per group): the 16 x 32-pixel tiles of the BC = 32 configuration
        const long t32 = (long)p.batch * ceil_div(p.wo, C3_TW) * ceil_div(p.ho, 16) * p.groups;
        return t32 >= 384 ? 32 : 0;
    }
    if (p.groups != 1 || p.cout_pad % 64 != 0) return 0;
#ifdef DZ_C3_DIAG
    static const int nt = getenv("DZ_TUNE_C3_NT") ? atoi(getenv("DZ_TUNE_C3_NT")) : 512;
#else
    constexpr int nt = 512;
#endif
    const int bc = (p.cout_pad % 128 == 0 && nt != 256) ? 128 : 64;
    // one 512-thread workgroup per CU: below ~1.5 waves of tiles the 4-wave kernels of conv2d_h.hip fill the chip better
    const long tiles = (long)p.batch * ceil_div(p.wo, C3_TW) * ceil_div(p.ho, 8) * (p.cout_pad / bc);
    if (tiles < 384 && !p.in_rowidx) return 0;        // (the sparse-input form exists in this kernel only)
    return bc;
}

#ifdef DZ_C3_DIAG
bool conv3x3_d_eligible(const dz_conv2d_desc &p);      // conv3x3_d.hip: direct-to-LDS 2 x 4-fragment variant (measured equal, not shipped)
int conv3x3_d_launch(const dz_conv2d_desc &p, int math, size_t w_bytes, hipStream_t stream);
#endif

int conv3x3_h_launch(const dz_conv2d_desc &p, int math, int out_f32, size_t w_bytes, hipStream_t stream) {
    const int bc = conv3x3_h_variant(p);
    if (p.in_rowidx) {
        if (bc != 128 || out_f32) {
            set_error("dz_conv2d_forward_split: in_rowidx (sparse input) needs a 3 x 3 stride-1 layer, 128-channel output tiles, two z slabs, pair16 output");
            return DZ_ERR_UNSUPPORTED;
        }
        if (math == DZ_MATH_F16X2) return launch_c3_nt<128, MathF16, false, 512, 0, 2, true>(p, w_bytes, stream);
        if (math == DZ_MATH_F16) return launch_c3_nt<128, MathF16H, false, 512, 0, 2, true>(p, w_bytes, stream);
        return launch_c3_nt<128, MathBF16, false, 512, 0, 2, true>(p, w_bytes, stream);
    }
#ifdef DZ_C3_DIAG
    if (bc == 128 && !out_f32 && conv3x3_d_eligible(p)) return conv3x3_d_launch(p, math, w_bytes, stream);
#endif
    auto go = [&](auto bc_t, auto m_t) {
        constexpr int BCV = decltype(bc_t)::value;
        using MM = typename decltype(m_t)::type;
        // fp32 output exists for the 32-channel configuration only (the head's grouped output layer - the one layer of the network that
        // leaves the pair16 domain); the 64 / 128-channel instances were never launched and spilled (round-4 review): the caller
        // (dz_conv2d_forward_split) sends such a layer to the generic kernel
        if constexpr (BCV == 32) return out_f32 ? launch_c3<BCV, MM, true>(p, w_bytes, stream) : launch_c3<BCV, MM, false>(p, w_bytes, stream);
        else return launch_c3<BCV, MM, false>(p, w_bytes, stream);
    };
    auto by_math = [&](auto bc_t) {
        if (math == DZ_MATH_F16X2) return go(bc_t, TypeTag<MathF16>{});
        if (math == DZ_MATH_F16) return go(bc_t, TypeTag<MathF16H>{});
        return go(bc_t, TypeTag<MathBF16>{});
    };
    if (bc == 32) return by_math(std::integral_constant<int, 32>{});
    if (bc == 128) return by_math(std::integral_constant<int, 128>{});
    return by_math(std::integral_constant<int, 64>{});
}

}  // namespace dz
