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

#include "hgemm.h"

namespace dz {

constexpr int C3_TW = 32, C3_TH = 8, C3_KC = 32;
constexpr int C3_ROW_U4 = C3_KC / 4 + 1;                          // 144-byte LDS rows (conflict-free ds_read_b128)
constexpr int C3_PXW = C3_TW + 2, C3_PXH = C3_TH + 2;
constexpr int C3_PX_ROWS = C3_PXW * C3_PXH;                       // 340 input pixels per tile
constexpr int C3_PX_PIECES = C3_PX_ROWS * (C3_KC / 4);            // 2720 16-byte pieces
constexpr int C3_THREADS = 512;
constexpr int C3_PXPT = (C3_PX_PIECES + C3_THREADS - 1) / C3_THREADS;   // 6 pieces per thread (last partly idle)
constexpr int C3_NS = 3;                                          // weight stages in flight (9 taps % 3 == 0)

template <int BC>
struct C3Cfg {
    static constexpr int CT = BC / 64;                            // 32-channel fragments per wave
    static constexpr int W_PIECES = BC * (C3_KC / 4);
    static constexpr int WPT = W_PIECES / C3_THREADS;             // weight pieces per thread per tap (2 or 1)
    static constexpr int W_U4 = BC * C3_ROW_U4;                   // one weight buffer
    static constexpr int LDS_BYTES = (C3_PX_ROWS * C3_ROW_U4 + 2 * W_U4) * 16;
    static_assert(BC == 64 || BC == 128, "BC is 64 or 128");
};

template <int BC, class M, bool OUT_F32>
__global__ __launch_bounds__(C3_THREADS) void k_conv3x3_h(dz_conv2d_desc p, int tiles_x, int tiles_y, unsigned int in_bytes,
                                                          unsigned int w_bytes) {
    using C = C3Cfg<BC>;
    constexpr int CT = C::CT, WPT = C::WPT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v4u *const px_s = reinterpret_cast<v4u *>(smem_raw);                     // [340][ROW_U4]
    v4u *const w_s = px_s + C3_PX_ROWS * C3_ROW_U4;                          // [2][BC][ROW_U4]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wp = wid >> 1, wc = wid & 1;                                   // image-row pair, channel half
    // Persistent workgroups, XCD-aware: the dispatcher places workgroup b on XCD b % 8 (own L2 each); XCD k walks the k-th
    // contiguous eighth of the pixel tiles, its workgroups side by side (neighbouring tiles share halo rows in that L2).
    // A workgroup keeps ONE channel tile (its weight stream simply wraps around from tile to tile) and the load pipeline
    // never drains between tiles: the next tile's input is prefetched during the last channel chunk of the current one.
    const int nty = p.cout_pad / BC;
    const int npx = p.batch * tiles_x * tiles_y;                 // pixel tiles
    const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3, nj = gridDim.x >> 3;      // gridDim.x is a multiple of 8
    const int n0 = (jloc % nty) * BC;
    const int per_xcd = (npx + 7) >> 3;
    const int band_lo = xcd * per_xcd, band_hi = min(npx, band_lo + per_xcd);
    const int tstep = nj / nty;                                   // workgroups of this XCD that share my channel tile
    int tile = band_lo + jloc / nty;
    if (tstep == 0 || tile >= band_hi) return;

    const srsrc_t prsrc = make_srsrc(p.in, in_bytes);
    const srsrc_t crsrc = make_srsrc(p.w, w_bytes);
    int x0, y0, b;
    auto tile_origin = [&](int t, int &ox, int &oy, int &ob) {
        ox = (t % tiles_x) * C3_TW;
        oy = ((t / tiles_x) % tiles_y) * C3_TH;
        ob = t / (tiles_x * tiles_y);
    };
    // input pieces of a tile: pixel (y0 + in_off + ry, x0 + in_off + rx) of the padded image, rx < 34, ry < 10
    auto tile_offsets = [&](int t, unsigned int (&off)[C3_PXPT]) {
        int ox, oy, ob;
        tile_origin(t, ox, oy, ob);
#pragma unroll
        for (int i = 0; i < C3_PXPT; ++i) {
            const int idx = tid + i * C3_THREADS;
            off[i] = OOB_OFFSET;
            if (idx < C3_PX_PIECES) {
                const int r = idx / (C3_KC / 4), q = idx % (C3_KC / 4);
                const int iy = oy + p.in_off + r / C3_PXW, ix = ox + p.in_off + r % C3_PXW;
                if (iy < p.in_hp && ix < p.in_wp)
                    off[i] = (unsigned int)((((long)(ob * p.in_hp + iy) * p.in_wp + ix) * p.in_cstride + p.in_coff + q * 4) * 4);
            }
        }
    };
    unsigned int pvoff[C3_PXPT], pvoff_next[C3_PXPT];
    tile_offsets(tile, pvoff);
    bool has_next = tile + tstep < band_hi;
    unsigned int cvoff[WPT];
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
        const int idx = tid + i * C3_THREADS;
        cvoff[i] = (unsigned int)(((n0 + idx / (C3_KC / 4)) * p.cin + (idx % (C3_KC / 4)) * 4) * 4);
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
        if (chunk >= nchunks) chunk = has_next ? chunk - nchunks : -1;
        const int kc = chunk / 9, tap = chunk - kc * 9;
        const unsigned int add = chunk >= 0 ? (unsigned int)tap * tap_bytes + (unsigned int)(kc * C3_KC * 4) : OOB_OFFSET;
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st[i]) : "v"(cvoff[i] + add), "s"(crsrc));
    };
    auto issue_px = [&](int kc) {
        // input tile of channel chunk kc; kc == nk: chunk 0 of the next tile
#pragma unroll
        for (int i = 0; i < C3_PXPT; ++i) {
            const unsigned int off = kc < nk ? pvoff[i] + (unsigned int)(kc * C3_KC * 4) : (has_next ? pvoff_next[i] : OOB_OFFSET);
            asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pst[i]) : "v"(off), "s"(prsrc));
        }
    };
    auto own_w = [&](v4u (&st)[WPT]) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) asm volatile("" : "+v"(st[i]));
    };
    auto store_w = [&](const v4u (&st)[WPT], int buf) {
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + i * C3_THREADS;
            w_s[buf * C::W_U4 + (idx / (C3_KC / 4)) * C3_ROW_U4 + idx % (C3_KC / 4)] = st[i];
        }
    };
    auto store_px = [&]() {
#pragma unroll
        for (int i = 0; i < C3_PXPT; ++i) {
            asm volatile("" : "+v"(pst[i]));
            const int idx = tid + i * C3_THREADS;
            if (idx < C3_PX_PIECES) px_s[(idx / (C3_KC / 4)) * C3_ROW_U4 + idx % (C3_KC / 4)] = pst[i];
        }
    };

    f32x16 acc[CT][2];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment base addresses (16-byte units): pixel fragment pt = image row 2*wp + pt, column lane & 31
    const int kg2 = (lane >> 5) * 2;
    const int pbase = ((2 * wp) * C3_PXW + (lane & 31)) * C3_ROW_U4 + kg2;
    const int wbase = (wc * CT * 32 + (lane & 31)) * C3_ROW_U4 + kg2;
    struct Frag { v4u p_hi[2], p_lo[2], c_hi[CT], c_lo[CT]; };
    auto load_frag = [&](Frag &f, int tap, int buf, int q) {
        const int ky = tap / 3, kx = tap - ky * 3;
        const v4u *pp = px_s + pbase + (ky * C3_PXW + kx) * C3_ROW_U4 + q * 4;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
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
    auto mma = [&](const Frag &f) {
        // term-major: consecutive MFMAs go to different accumulators (measured 1.7 % faster than three in a row into the same
        // one); each accumulator still receives lo.hi, hi.lo, hi.hi in that order, so the result does not depend on it
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
                    acc[ct][pt] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[pt] : f.p_hi[pt], acc[ct][pt]);
    };

    // ---- prologue: input tile of chunk 0 and weight slice of (chunk 0, tap 0) into LDS; taps 1..3 in flight
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
    if (has_next) tile_offsets(tile + tstep, pvoff_next);
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
            if (t >= 1 && t <= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW2 + C3_PXPT));
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NW2));
            own_w(wst[(t + 1) % 3]);
            store_w(wst[(t + 1) % 3], buf ^ 1);
            mma(f0);
            // one LDS read / write behind each of the first MFMAs: the two waves of a SIMD run in lock step (one barrier
            // per tap), so memory instructions issued in a block of their own would leave the matrix pipe idle
            interleave_hint<0x100, 2 * (2 + CT), 1>();
            interleave_hint<0x200, WPT, 1>();
            __syncthreads();
            if (t == 8) {
                // channel-chunk boundary: every wave has finished reading the old input tile (its last reads were the
                // k-step-1 fragments above, complete before the barrier); swap in the prefetched tile
                store_px();
                __syncthreads();
            }
            // ---- phase 2: fragments of chunk c+1, weights of chunk c+4 into the stage just freed, MFMAs of k-step 1
            load_frag(f0, (t + 1) % 9, buf ^ 1, 0);
            issue_w(wst[(t + 1) % 3], c + 4);
            if (t == 0) issue_px(kc + 1);
            mma(f1);
            interleave_hint<0x100, 2 * (2 + CT), 1>();
        }
    }
    tile_origin(tile, x0, y0, b);

    // ---- epilogue: 32x32 accumulator: pixel column = lane & 31, channel = 8*(reg>>2) + 4*(lane>>5) + (reg&3)
    const int h = lane >> 5;
    const int gcout = p.g_cout[0];
    const int ooff = p.out_coff + p.g_ooff[0];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) {
        const int y = y0 + 2 * wp + pt, x = x0 + (lane & 31);
        if (y >= p.ho || x >= p.wo) continue;
        const size_t op = ((size_t)b * p.out_hp + (size_t)y * p.out_sy + p.out_dy) * p.out_wp + (size_t)x * p.out_sx + p.out_dx;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wc * CT * 32 + ct * 32 + 8 * j + 4 * h;
                if (col >= gcout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = p.scale ? p.scale[col + e] : 1.f;
                    const float sh = p.shift ? p.shift[col + e] : 0.f;
                    v[e] = fmaf(acc[ct][pt][4 * j + e], sc, sh);
                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                }
                if (OUT_F32) {
                    float *o = p.out + op * p.out_cstride + ooff + col;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < gcout) o[e] = v[e];
                } else {
                    uint2 hi, lo;
                    split4<M>(v, hi, lo);
                    unsigned char *g = reinterpret_cast<unsigned char *>(p.out) + (op * p.out_cstride + ooff + (col & ~7)) * 4 + (col & 7) * 2;
                    *reinterpret_cast<uint2 *>(g) = hi;
                    *reinterpret_cast<uint2 *>(g + 16) = lo;
                }
            }
        }
    }
    // ---- next tile: its first input chunk is already in LDS, its first weight slices are in flight
    if (!has_next) break;
    tile += tstep;
#pragma unroll
    for (int i = 0; i < C3_PXPT; ++i) pvoff[i] = pvoff_next[i];
    has_next = tile + tstep < band_hi;
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)");      // nothing of this file's asm loads may stay in flight at exit
}

template <int BC, class M, bool OUT_F32>
static int launch_c3(const dz_conv2d_desc &p, size_t w_bytes, hipStream_t stream) {
    using C = C3Cfg<BC>;
    const int tiles_x = ceil_div(p.wo, C3_TW), tiles_y = ceil_div(p.ho, C3_TH);
    const size_t in_bytes = (size_t)p.batch * p.in_hp * p.in_wp * p.in_cstride * sizeof(float);
    if (in_bytes >= 0x80000000ull || w_bytes >= 0x80000000ull) {
        set_error("dz_conv2d_forward_split: image of %zu bytes / weights of %zu bytes exceed the 2 GiB buffer-addressing limit", in_bytes, w_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv3x3_h<BC, M, OUT_F32>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                C::LDS_BYTES) != hipSuccess) {
            set_error("dz_conv2d_forward_split: cannot reserve %d bytes of LDS", C::LDS_BYTES);
            return DZ_ERR_HIP;
        }
        attr_set = true;
    }
    // persistent: one 512-thread workgroup per CU, a multiple of 8 x channel tiles so that every XCD gets the same number of
    // workgroups of every channel tile
    const int nty = p.cout_pad / BC;
    int per_xcd = 32 / nty * nty;
    if (per_xcd < nty) per_xcd = nty;
    const long grid = 8L * per_xcd;
    hipLaunchKernelGGL((k_conv3x3_h<BC, M, OUT_F32>), dim3((unsigned int)grid), dim3(C3_THREADS), C::LDS_BYTES, stream, p, tiles_x,
                       tiles_y, (unsigned int)in_bytes, (unsigned int)w_bytes);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// 0 = not eligible, 64 / 128 = channel tile of the resident-tile kernel
int conv3x3_h_variant(const dz_conv2d_desc &p) {
    static const int off = getenv("DZ_TUNE_NO_CONV3X3") ? atoi(getenv("DZ_TUNE_NO_CONV3X3")) : 0;
    if (off) return 0;
    if (p.kh != 3 || p.kw != 3 || p.stride != 1 || p.groups != 1 || p.group_shift) return 0;
    if (p.cin % C3_KC != 0 || p.cout_pad % 64 != 0) return 0;
    const int bc = p.cout_pad % 128 == 0 ? 128 : 64;
    // one 512-thread workgroup per CU: below ~1.5 waves of tiles the 4-wave kernels of conv2d_h.hip fill the chip better
    const long tiles = (long)p.batch * ceil_div(p.wo, C3_TW) * ceil_div(p.ho, C3_TH) * (p.cout_pad / bc);
    if (tiles < 384) return 0;
    return bc;
}

int conv3x3_h_launch(const dz_conv2d_desc &p, int math, int out_f32, size_t w_bytes, hipStream_t stream) {
    const int bc = conv3x3_h_variant(p);
    if (bc == 128) {
        if (math == DZ_MATH_F16X2) return out_f32 ? launch_c3<128, MathF16, true>(p, w_bytes, stream) : launch_c3<128, MathF16, false>(p, w_bytes, stream);
        return out_f32 ? launch_c3<128, MathBF16, true>(p, w_bytes, stream) : launch_c3<128, MathBF16, false>(p, w_bytes, stream);
    }
    if (math == DZ_MATH_F16X2) return out_f32 ? launch_c3<64, MathF16, true>(p, w_bytes, stream) : launch_c3<64, MathF16, false>(p, w_bytes, stream);
    return out_f32 ? launch_c3<64, MathBF16, true>(p, w_bytes, stream) : launch_c3<64, MathBF16, false>(p, w_bytes, stream);
}

}  // namespace dz
