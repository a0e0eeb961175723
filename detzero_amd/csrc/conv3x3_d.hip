// 3x3 stride-1 dense BEV convolution on pair16 operands, second generation: 2 x 4-fragment wave tiles fed by direct-to-LDS loads.
//
// conv3x3_h.hip gives a wave 2 x 2 fragments (two image rows x 64 channels): 8 LDS fragment reads per 12 MFMAs.  On real operand
// bits the matrix pipe of the MI355X is power-limited, and the LDS-fed ceiling measured for that wave tile is 443-451 TF/s
// algorithmic (tools/micro/mfma_lds.hip) - which that kernel reaches.  A 2 x 4 wave tile (two image rows x all 128
// channels of the tile: 12 reads per 24 MFMAs) has a ceiling of 527 TF/s, but its 8 accumulators (128 registers) + double-buffered
// fragments (96) leave no room for the register staging of the weight / input pipeline.  gfx950 can load 16 bytes per lane
// straight into LDS (buffer_load_dwordx4 ... lds: LDS address = M0 + lane * 16), so here NOTHING is staged in registers:
//   * a 512-thread workgroup owns 16 rows x 32 columns of output pixels x 128 output channels; wave w = image rows 2w, 2w+1;
//   * channel chunks of 16 (one MFMA k-step): a pair16 row of a chunk is 64 bytes = 4 pieces [hi g0 | lo g0 | hi g1 | lo g1];
//   * LDS rows are unpadded (a wave-level direct load writes 1 KB contiguous), bank conflicts are avoided by an XOR swizzle:
//     piece p of row r lives in slot p ^ ((r >> 2) & 3); every lane fetches the global piece that belongs in ITS slot;
//   * the (18 x 34)-pixel input tile of a chunk is double-buffered in LDS (loaded during the 9 taps of the previous chunk), the
//     128 x 16-channel weight slice of a (tap, chunk) goes through a ring of three 8 KB buffers, two (tap, chunk) steps ahead;
//   * one workgroup barrier per (tap, chunk) step = per 24 MFMAs of a wave; vmcnt counts are static (18 steps unrolled).
// Same descriptor, same results (the accumulation order over taps and channels is unchanged) as conv3x3_h.hip.
//
// MEASURED (r02, MI355X): parity-green, LDS array cycles 131 M -> 85 M per launch, bank conflicts < 10 % - and exactly the same
// time as conv3x3_h.hip on a balanced problem (32 x 188 x 188 x 128 -> 128: 924 vs 924 us, the same 1.394 M CU cycles), 4 %
// slower on the headline shape, whose 1152 tiles of 16 x 32 pixels are 4.5 per CU.  Both kernels keep the matrix pipe busy 71 %
// of the cycles at the clock the power budget allows: the limit is the energy of the MFMAs themselves, not their feeding.
// Not shipped: this file is compiled only with -DDZ_C3_DIAG (DZ_TUNE_C3_D=1 selects it).
#ifdef DZ_C3_DIAG
#include <stdlib.h>

#include "hgemm.h"

namespace dz {

constexpr int D_TW = 32, D_TH = 16, D_NT = 512, D_BC = 128, D_NW = 3;
constexpr int D_PXW = D_TW + 2, D_PXH = D_TH + 2, D_PX = D_PXW * D_PXH;        // 612 input pixels per tile
constexpr int D_PX_LOADS = (D_PX * 4 + D_NT - 1) / D_NT;                      // 5 direct loads per thread per channel chunk
constexpr int D_PXBUF = D_PX_LOADS * D_NT * 16;                               // 40960 bytes (the tail beyond 612 pixels is never read)
constexpr int D_WBUF = D_BC * 64;                                             // 8192 bytes
constexpr int D_STG_ROW = 80, D_STG = 32 * D_STG_ROW;                         // epilogue staging window of a wave (2 groups per round)
constexpr int D_OFF_W = 2 * D_PXBUF, D_OFF_STG = D_OFF_W + D_NW * D_WBUF, D_OFF_SS = D_OFF_STG + (D_NT / 64) * D_STG;
constexpr int D_LDS_BYTES = D_OFF_SS + 2 * D_BC * 4;

// 16 bytes per lane from a buffer straight into LDS at lds_base + lane * 16 (lds_base wave-uniform)
__device__ __forceinline__ void load_to_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_base), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

template <class M>
__global__ __launch_bounds__(D_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_conv3x3_d(dz_conv2d_desc p, int tiles_x, int tiles_y,
                                                                                               unsigned int in_bytes, unsigned int w_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // persistent, XCD-aware schedule of conv3x3_h.hip: XCD k walks the k-th eighth of the pixel tiles, a workgroup keeps one channel tile
    const int nty = p.cout_pad / D_BC;
    const int npx = p.batch * tiles_x * tiles_y;
    const int xcd = blockIdx.x & 7, jloc = blockIdx.x >> 3, nj = gridDim.x >> 3;
    const int n0 = (jloc % nty) * D_BC;
    const int per_xcd = (npx + 7) >> 3;
    const int band_lo = xcd * per_xcd, band_hi = min(npx, band_lo + per_xcd);
    const int tstep = nj / nty;
    int tile = band_lo + jloc / nty;
    if (tstep == 0 || tile >= band_hi) return;

    const srsrc_t prsrc = make_srsrc(p.in, in_bytes);
    const srsrc_t crsrc = make_srsrc(p.w, w_bytes);
    int x0, y0, b;
    auto tile_origin = [&](int t, int &ox, int &oy, int &ob) {
        ox = (t % tiles_x) * D_TW;
        oy = ((t / tiles_x) % tiles_y) * D_TH;
        ob = t / (tiles_x * tiles_y);
    };
    struct TileGeo { unsigned int base; int rows, cols; };
    auto tile_geo = [&](int t) {
        int ox, oy, ob;
        tile_origin(t, ox, oy, ob);
        TileGeo g;
        g.base = (unsigned int)((((long)(ob * p.in_hp + oy + p.in_off) * p.in_wp + ox + p.in_off) * p.in_cstride + p.in_coff) * 4);
        g.rows = p.in_hp - (oy + p.in_off);
        g.cols = p.in_wp - (ox + p.in_off);
        return g;
    };
    TileGeo geo = tile_geo(tile), geo_next = geo;
    bool has_next = tile + tstep < band_hi;

    const unsigned int tap_bytes = (unsigned int)((long)p.cout_pad * p.cin * 4);
    const int nk = p.cin / 16;                          // channel chunks (even: cin % 32 == 0)
    const int nchunks = nk * 9;

    // weight slice of a step: 128 rows x 4 pieces = one piece per thread.  My LDS slot is (row r = tid >> 2, slot s = tid & 3): it holds
    // piece s ^ ((r >> 2) & 3) of that row
    const int wr = tid >> 2;
    const unsigned int wvoff = (unsigned int)(((n0 + wr) * p.cin) * 4 + (((tid & 3) ^ ((wr >> 2) & 3)) * 16));
    auto issue_w = [&](int chunk, int slot) {
        // chunk = kc * 9 + tap of the current tile; past its end the stream wraps to the next tile's steps (same weights), or - after
        // the last tile - to out-of-range offsets (zeros land in LDS, nothing is fetched; the per-wave load counts stay uniform)
        if (chunk >= nchunks) chunk = has_next ? chunk - nchunks : -1;
        const int kc = chunk / 9, tap = chunk - kc * 9;
        const unsigned int soff = chunk >= 0 ? (unsigned int)tap * tap_bytes + (unsigned int)(kc * 64) : 0u;
        load_to_lds((unsigned int)(D_OFF_W + slot * D_WBUF + wid * 1024), chunk >= 0 ? wvoff : OOB_OFFSET, crsrc, soff);
    };
    // input tile of a chunk: 612 pixels x 4 pieces in 5 loads per thread; load i of wave w fills pieces (i * 8 + w) * 64 + lane
    auto issue_px = [&](int i, int kc, int buf) {
        const bool cur = kc < nk;                        // kc == nk: chunk 0 of the next tile
        const TileGeo g = cur ? geo : geo_next;
        const unsigned int sbase = g.base + (cur ? (unsigned int)(kc * 64) : 0u);
        int pc = (i * 8 + wid) * 64 + lane;
        asm volatile("" : "+v"(pc));                     // keeps the per-piece offsets from being hoisted into registers for the whole loop
        const int r = pc >> 2;
        const int ry = r / D_PXW, rx = r - ry * D_PXW;
        const bool ok = (cur || has_next) && r < D_PX && ry < g.rows && rx < g.cols;
        const unsigned int voff = ok ? (unsigned int)(((ry * p.in_wp + rx) * p.in_cstride) * 4 + (((pc & 3) ^ ((r >> 2) & 3)) * 16)) : OOB_OFFSET;
        load_to_lds((unsigned int)(buf * D_PXBUF + (i * 8 + wid) * 1024), voff, prsrc, sbase);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addresses: k-group kg = lane >> 5 -> pieces 2 kg (hi) and 2 kg + 1 (lo)
    const int kg = lane >> 5, l31 = lane & 31;
    const unsigned int c_lds = (unsigned int)(D_OFF_W + l31 * 64 + (((2 * kg) ^ ((l31 >> 2) & 3)) * 16));     // + ct * 2048 + slot * 8192; lo = ^ 16
    struct Frag { v4u p_hi[2], p_lo[2], c_hi[4], c_lo[4]; };
    auto load_frag = [&](Frag &f, int tap, int pxbuf, int wslot) {
        const int ky = tap / 3, kx = tap - ky * 3;
        int lx = l31;
        asm volatile("" : "+v"(lx));                     // fragment addresses are rebuilt per step (a few VALU ops), not kept in 18+ registers
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int r = (2 * wid + pt + ky) * D_PXW + lx + kx;
            const unsigned int a = (unsigned int)(pxbuf * D_PXBUF + r * 64 + (((2 * kg) ^ ((r >> 2) & 3)) * 16));
            f.p_hi[pt] = *reinterpret_cast<const v4u *>(smem_raw + a);
            f.p_lo[pt] = *reinterpret_cast<const v4u *>(smem_raw + (a ^ 16u));
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const unsigned int a = c_lds + (unsigned int)(wslot * D_WBUF + ct * 2048);
            f.c_hi[ct] = *reinterpret_cast<const v4u *>(smem_raw + a);
            f.c_lo[ct] = *reinterpret_cast<const v4u *>(smem_raw + (a ^ 16u));
        }
    };
    auto mma = [&](const Frag &f) {
#pragma unroll
        for (int term = 3 - M::TERMS; term < 3; ++term)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
                    acc[ct][pt] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[pt] : f.p_hi[pt], acc[ct][pt]);
    };

    float *const sc_s = reinterpret_cast<float *>(smem_raw + D_OFF_SS), *const sh_s = sc_s + D_BC;
    if (tid < D_BC) {
        const bool in = n0 + tid < p.g_cout[0];
        sc_s[tid] = (in && p.scale) ? p.scale[n0 + tid] : 1.f;
        sh_s[tid] = (in && p.shift) ? p.shift[n0 + tid] : 0.f;
    }
    // ---- prologue: input tile of chunk 0, weight slices of steps 0 and 1
#pragma unroll
    for (int i = 0; i < D_PX_LOADS; ++i) issue_px(i, 0, 0);
    issue_w(0, 0);
    issue_w(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    Frag fa, fb;
    load_frag(fa, 0, 0, 0);

    for (;;) {
        if (has_next) geo_next = tile_geo(tile + tstep);
        for (int kc = 0; kc < nk; kc += 2) {
#pragma unroll
            for (int u = 0; u < 18; ++u) {
                const int t = u % 9, half = u / 9;               // tap, which chunk of the pair (= input buffer of this step)
                const int c = (kc + half) * 9 + t;
                // weights two steps ahead; the next chunk's input tile during taps 0..4
                issue_w(c + 2, (u + 2) % D_NW);
                if (t < D_PX_LOADS) issue_px(t, kc + half + 1, half ^ 1);
                // step c + 1's weights (issued in step c - 1) and, at tap 8, the next chunk's input tile are older than these
                {
                    const int younger = 1 + (t < D_PX_LOADS ? 1 : 0) + ((t >= 1 && t <= D_PX_LOADS) ? 1 : 0);
                    if (younger == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                }
                __syncthreads();
                const int tn = (t + 1) % 9, bufn = t == 8 ? (half ^ 1) : half;
                if (u & 1) {
                    load_frag(fa, tn, bufn, (u + 1) % D_NW);
                    mma(fb);
                } else {
                    load_frag(fb, tn, bufn, (u + 1) % D_NW);
                    mma(fa);
                }
                interleave_hint<0x100, M::TERMS == 1 ? 6 : 12, M::TERMS == 1 ? 1 : 2>();
            }
        }
        tile_origin(tile, x0, y0, b);

        // ---- epilogue (see conv3x3_h.hip): fragments through a wave-private LDS window, 64 contiguous bytes per pixel and round
        {
            const int h = lane >> 5;
            const int gcout = p.g_cout[0];
            const int ooff = p.out_coff + p.g_ooff[0];
            unsigned char *const stg = smem_raw + D_OFF_STG + wid * D_STG;
            const int srow = lane >> 2, spiece = lane & 3;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int y = y0 + 2 * wid + pt;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                    for (int j0 = 0; j0 < 4; j0 += 2) {
                        const int cbase = n0 + ct * 32 + j0 * 8;
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = j0 + jj;
                            const int lc = ct * 32 + 8 * j + 4 * h;
                            const float4 sc = *reinterpret_cast<const float4 *>(sc_s + lc), sh = *reinterpret_cast<const float4 *>(sh_s + lc);
                            float v[4] = {fmaf(acc[ct][pt][4 * j], sc.x, sh.x), fmaf(acc[ct][pt][4 * j + 1], sc.y, sh.y),
                                          fmaf(acc[ct][pt][4 * j + 2], sc.z, sh.z), fmaf(acc[ct][pt][4 * j + 3], sc.w, sh.w)};
                            if (p.relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                            }
                            uint2 hi, lo;
                            split4<M>(v, hi, lo);
                            unsigned char *w = stg + l31 * D_STG_ROW + jj * 32 + h * 8;
                            *reinterpret_cast<uint2 *>(w) = hi;
                            *reinterpret_cast<uint2 *>(w + 16) = lo;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const int gcol = cbase + (spiece >> 1) * 8;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const int r = srow + 16 * i, x = x0 + r;
                            const v4u d = *reinterpret_cast<const v4u *>(stg + r * D_STG_ROW + spiece * 16);
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
        }
        // ---- next tile: its first input chunk is in LDS, its first fragments are in registers, its weight stream is in flight
        if (!has_next) break;
        tile += tstep;
        geo = geo_next;
        has_next = tile + tstep < band_hi;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of this file's asm loads may stay in flight at exit
}

template <class M>
static int launch_c3d(const dz_conv2d_desc &p, size_t w_bytes, hipStream_t stream) {
    const int tiles_x = ceil_div(p.wo, D_TW), tiles_y = ceil_div(p.ho, D_TH);
    const size_t in_bytes = (size_t)p.batch * p.in_hp * p.in_wp * p.in_cstride * sizeof(float);
    if (in_bytes >= 0x80000000ull || w_bytes >= 0x80000000ull) {
        set_error("dz_conv2d_forward_split: image of %zu bytes / weights of %zu bytes exceed the 2 GiB buffer-addressing limit", in_bytes, w_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    static PerDeviceFlags lds_done;
    if (int rc_ = reserve_lds(reinterpret_cast<const void *>(&k_conv3x3_d<M>), D_LDS_BYTES, lds_done, "dz_conv2d_forward_split")) return rc_;
    const int nty = p.cout_pad / D_BC;
    int per_xcd = 32 / nty * nty;
    if (per_xcd < nty) per_xcd = nty;
    hipLaunchKernelGGL((k_conv3x3_d<M>), dim3((unsigned int)(8 * per_xcd)), dim3(D_NT), D_LDS_BYTES, stream, p, tiles_x, tiles_y,
                       (unsigned int)in_bytes, (unsigned int)w_bytes);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// eligibility on top of conv3x3_h_variant(p) == 128 (pair16 output only)
bool conv3x3_d_eligible(const dz_conv2d_desc &p) {
    static const int on = getenv("DZ_TUNE_C3_D") ? atoi(getenv("DZ_TUNE_C3_D")) : 0;
    return on && p.groups == 1 && p.cin % 32 == 0 && p.cout_pad % D_BC == 0;
}

int conv3x3_d_launch(const dz_conv2d_desc &p, int math, size_t w_bytes, hipStream_t stream) {
    if (math == DZ_MATH_F16) return launch_c3d<MathF16H>(p, w_bytes, stream);
    return math == DZ_MATH_F16X2 ? launch_c3d<MathF16>(p, w_bytes, stream) : launch_c3d<MathBF16>(p, w_bytes, stream);
}

}  // namespace dz
#endif  // DZ_C3_DIAG
