// Sparse 3-D convolution for the SMALL-CHANNEL levels (Cin in {16, 32}, Cout in {16, 32}) on pair16 operands:
// wave-private 32-row tiles, weights of ALL kernel taps resident in LDS, activations gathered straight from HBM/L2
// into the matrix-core operand registers (no LDS staging, no barriers in the loop).
//
// Why a second kernel next to k_spconv_h (sparse_conv_h.hip): at 16-32 channels a neighbour row is 64-128 bytes and a
// (tap, 128-row tile) step of k_spconv_h moves ~5 vector-memory instructions per wave through LDS for 6 MFMAs - its time
// follows the vector-memory path, not bytes or FLOPs.  Measured on MI355X (r02): the cost of a gather instruction is
// ~1.75 cycles of the CU's L1 per DISTINCT 128-byte line its 64 lanes touch, so the layout below is built around
// "one instruction = 16 whole rows":
//   * v_mfma_f32_16x16x32: lane (n = lane & 15, kg = lane >> 4) supplies 8 consecutive k of column n.  The 64 bytes
//     [hi(8 ch) | lo(8 ch) | hi(next 8 ch) | lo(next 8 ch)] of a pair16 row, in MEMORY ORDER, are taken as the 32 k values of
//     one MFMA step, so the four lanes of a pixel read 64 contiguous bytes and one load covers 16 rows = 16 lines.  The
//     split product  w.x ~= w_hi.x_hi + w_hi.x_lo + w_lo.x_hi  becomes two MFMAs on the same B operand:
//        A1 = [w_hi | w_hi | w_hi' | w_hi']   (-> w_hi.x_hi + w_hi.x_lo)      A2 = [w_lo | w_lo | w_lo' | w_lo']   (-> w_lo.x_hi + w_lo.x_lo)
//     (the lo.lo term k_spconv_h drops comes for free here: the sum is the full fp32-accumulated product of the pairs)
//   * the whole weight tensor (27 x Cout x Cin pair16, <= 108 KB) is loaded into LDS ONCE per workgroup, 16 output
//     channels x 16 bytes per unit, so a wave reads an A fragment as contiguous 256-byte runs;
//   * a wave owns 32 output rows (two 16-pixel halves sharing the A fragments); taps are skipped at 32-row granularity
//     (MFMAs at 16-row granularity);
//   * waves never synchronise: each runs a 3-stage software pipeline over its own stream of chunks (a chunk = up to G
//     taps of one tile): neighbour indices of chunk c+2, gathers of chunk c+1 and MFMAs of chunk c are in flight
//     together, across tile boundaries.  Loads are issued from inline asm and this file keeps the vmcnt bookkeeping
//     (every chunk issues exactly 2G index loads and NL gathers - absent slots use the out-of-range offset, which fetches
//     nothing - so the counts are static; the residual rows of a tile ride behind the gathers of its last chunk).
// Epilogue = k_spconv_h's: BatchNorm scale/shift (from LDS), residual, ReLU, split into (hi, lo), 8-byte stores.
//
// Reference call sites: detection/detzero_det/models/centerpoint_modules/backbone3d.py:243-259 (conv_input, conv1, conv2).
#pragma once
#include "hgemm.h"

namespace dz {

struct SpConvHArgs {
    const float *in;        // pair16 rows
    const int *nbr;
    const uint32_t *tile_masks;  // per-32-row tap masks (dz_build_neighbors) or null
    const int *d_m_out;
    const float *w;         // (kvol, cout_pad, cin) pair16
    const float *scale;
    const float *shift;
    const float *residual;  // pair16 rows or null
    float *out;             // pair16 rows
    int cin, cout, cout_pad, kvol, cap, relu;
    unsigned int in_bytes, w_bytes, nbr_bytes, mask_bytes;
    int diag;                // -DDZ_SPCONV_DIAG builds only (timing experiments; results are garbage when non-zero)
};

// chunk descriptor (wave-uniform, lives in SGPRs)
struct WChunk {
    int row0;            // first output row of the tile
    unsigned int taps;   // 5 bits per slot (slots >= nslots repeat the last tap)
    int nslots;          // slots that carry a tap of the tile
    int last;            // last chunk of its tile: epilogue after its MFMAs
    int live;            // 0: past the end of this wave's work (nothing loaded, nothing stored)
    unsigned int rowoff[2];   // per lane: byte offset of its row of each 16-pixel half in one tap of the table (or out of range)
};

template <int CIN, int COUT, int G, class M, int WAVES, int OCC, bool PK = false>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(OCC))) void k_spconv_w(SpConvHArgs a) {
    constexpr int KS = CIN / 16;            // 32-deep MFMA steps per tap (16 channels x (hi, lo))
    constexpr int KG = CIN / 8;             // 8-channel groups of a row
    constexpr int CB = COUT / 16;           // 16-channel output blocks
    constexpr int NI = 2 * G;               // index loads per chunk (one per 16-pixel half and slot)
    constexpr int NL = G * 2 * KS;          // gather loads per chunk
    constexpr int NR = 2 * 2 * CB;          // residual loads of a tile (8 bytes hi + 8 bytes lo per half and block)
    constexpr int ROWB = CIN * 4;           // bytes of a pair16 input row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v4u *const w_s = reinterpret_cast<v4u *>(smem_raw);                              // [kvol][CB][KG][2][16]
    float *const sc_s = reinterpret_cast<float *>(w_s + a.kvol * CB * KG * 2 * 16);      // [32] scale, [32] shift

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, kg = lane >> 4;
    const int m = min(*a.d_m_out, a.cap);

    // ---- weights -> LDS: unit [(tap, cb, kk, half, c)] = the `half` (hi / lo) values of channels 8kk..8kk+7 of cout 16cb + c
    {
        const v4u *wg = reinterpret_cast<const v4u *>(a.w);
        const int units = a.kvol * CB * KG * 2 * 16;
        for (int u = tid; u < units; u += 64 * WAVES) {
            const int c = u & 15, half = (u >> 4) & 1, kk = (u >> 5) % KG, cb = ((u >> 5) / KG) % CB, tap = (u >> 5) / (KG * CB);
            w_s[u] = wg[((size_t)(tap * a.cout_pad + cb * 16 + c) * CIN + kk * 8) / 4 + half];
        }
        if (tid < 32) {
            sc_s[tid] = (a.scale && tid < a.cout) ? a.scale[tid] : 1.f;
            sc_s[32 + tid] = (a.shift && tid < a.cout) ? a.shift[tid] : 0.f;
        }
    }
    __syncthreads();

    const srsrc_t prsrc = make_srsrc(a.in, a.in_bytes);
    const srsrc_t nrsrc = make_srsrc(a.nbr, a.nbr_bytes);
    const unsigned int res_bytes = a.residual ? (unsigned int)min((size_t)0x7FFFFFFFu, (size_t)a.cap * a.cout * 4) : 0u;
    const srsrc_t rrsrc = make_srsrc(a.residual ? a.residual : a.in, res_bytes);
    const unsigned int nbr_tap_bytes = (unsigned int)a.cap * 4u;

    // ---- this wave's tile sequence: workgroup b runs on XCD b % 8; at step s the workgroups of an XCD work on neighbouring
    // groups of WAVES consecutive tiles (runs of XRUN groups per XCD), i.e. all CUs of an XCD sweep through the table together
    // and share gathered rows in L1 / L2 while they are hot.  (Measured and dropped: contiguous equal-work ranges per wave or
    // per workgroup - the static deal already keeps 91 % of the wave slots busy, and distant ranges lose the sharing: -12 %.)
    constexpr int XRUN = 4;
    const int xcd = blockIdx.x & 7, jx = blockIdx.x >> 3, nx = gridDim.x >> 3;
    const int ntiles_w = (m + 31) >> 5;
    int seq = -1;                           // groups of this workgroup handed out so far
    auto tile_of = [&](int s) {
        const int q = s * nx + jx;
        return ((q / XRUN) * 8 * XRUN + xcd * XRUN + q % XRUN) * WAVES + wid;
    };
    // tap masks of this wave's next 64 tiles, one per lane (ONE vector load per 64 tiles; a scalar or uniform vector load per
    // tile would sit in the middle of the hand-counted load stream and drain it)
    int mask_base = 0;
    const srsrc_t mrsrc = make_srsrc(a.tile_masks, a.mask_bytes);
    auto load_masks = [&](int base) -> unsigned int {
        const int tile = tile_of(base + lane);
        unsigned int v;         // (asm: the compiler's own waitcnt pass would otherwise drain the stream at every tile switch)
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(tile < ntiles_w ? (unsigned int)tile * 4u : OOB_OFFSET), "s"(mrsrc));
        return v;
    };
    unsigned int mask_vec = load_masks(0);
    unsigned int rem = 0u;
    int cur_row0 = 0, cur_live = 0;
    unsigned int cur_rowoff[2] = {OOB_OFFSET, OOB_OFFSET};   // byte offset of this lane's two rows in one tap of the table
    bool started = false;
    // next chunk of this wave's stream.  Slots past the end of a tile's tap list repeat its last tap (their loads hit the
    // lines just fetched; their MFMAs are skipped through `nslots`), so no per-slot validity logic is needed downstream.
    auto next_chunk = [&]() {
        WChunk c;
        if (rem == 0u || !started) {                       // move to the next tile
            started = true;
            ++seq;
            if (seq - mask_base == 64) {                   // (drains the load pipeline once per 64 tiles)
                mask_base += 64;
                mask_vec = load_masks(mask_base);
            }
            const int tile = tile_of(seq);
            cur_live = tile < ntiles_w;
            cur_row0 = tile * 32;
            rem = (unsigned int)__builtin_amdgcn_readlane((int)mask_vec, seq - mask_base);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int row = cur_row0 + h * 16 + n;
                cur_rowoff[h] = (cur_live && row < m) ? (unsigned int)row * 4u : OOB_OFFSET;
            }
        }
        c.row0 = cur_row0;
        c.live = cur_live;
        c.taps = 0u;
        c.nslots = 0;
        int t = 0;
#pragma unroll
        for (int s = 0; s < G; ++s) {
            if (rem) { t = __ffs((int)rem) - 1; rem &= rem - 1; ++c.nslots; }
            c.taps |= (unsigned int)t << (5 * s);
        }
        c.last = (rem == 0u);
        c.rowoff[0] = cur_rowoff[0];
        c.rowoff[1] = cur_rowoff[1];
        return c;
    };

    // ---- pipeline registers
    int idx[2][G][2];                       // neighbour indices of a chunk: [slot][16-pixel half]
    v4u gat[2][G][2][KS];                   // gathered operands of a chunk: [slot][half][k-step]
    uint2 resv[2][NR];                      // residual halves of the tile whose last chunk sits in the slot
    unsigned int present[2] = {0u, 0u};     // bit 2*slot + half: that half of the tile has a neighbour at the slot's tap
    f32x4v acc[2][CB];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[h][cb][e] = 0.f;

    auto issue_idx = [&](const WChunk &c, int (&dst)[G][2]) {
#pragma unroll
        for (int s = 0; s < G; ++s) {
            // scalar: rides in the instruction's soffset.  PK: the packed table has one row per (tz, ty) = tap / 3 (the entry is loaded
            // once per slot all the same: a second load of it hits the line, and the hand-counted load stream keeps its shape)
            unsigned int toff = (PK ? ((c.taps >> (5 * s)) & 31u) / 3u : ((c.taps >> (5 * s)) & 31u)) * nbr_tap_bytes;
#ifdef DZ_SPCONV_DIAG
            if (a.diag == 3) toff = 0u;                        // every index load reads tap 0 (cache hits)
#endif
#pragma unroll
            for (int h = 0; h < 2; ++h)
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(dst[s][h]) : "v"(c.rowoff[h]), "s"(nrsrc), "s"(toff));
        }
    };
    // indices have landed (caller waited): gather the operands, remember which halves have any neighbour at all.
    // A missing neighbour is -1: (unsigned)(-1) * ROWB + kg * 16 (+ 64) stays just below 2^32, i.e. out of range -> zeros.
    auto issue_gather = [&](const WChunk &c, int (&ix)[G][2], v4u (&dst)[G][2][KS], unsigned int &pres) {
        pres = 0u;
#pragma unroll
        for (int s = 0; s < G; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm volatile("" : "+v"(ix[s][h]));
                if (PK) {
                    // packed entry (DZ_NBR_PACKED): r = active cells below the centre of the x window (bits 0..28), presence of the
                    // left / centre / right tap (bits 29..31): left = r - 1, centre = r, right = r + centre.  Branch-free: the tap's
                    // position in its window is wave-uniform, so its shift / offset / centre mask are scalars
                    const unsigned int tx = ((c.taps >> (5 * s)) & 31u) % 3u;
                    const unsigned int sh = 29u + tx, cm = tx >> 1;                     // cm = 1 for the right tap
                    const int dl = (int)((tx + 2u) / 3u) - 1;                           // -1 for the left tap, else 0
                    const unsigned int e = (unsigned int)ix[s][h];
                    const int keep = __builtin_amdgcn_sbfe((int)e, sh, 1u);              // 0 / -1: the tap's presence bit, sign-extended
                    const int v = (int)(e & 0x1FFFFFFFu) + dl + (int)((e >> 30) & cm);
                    ix[s][h] = (v & keep) | ~keep;
                }
                if (__ballot(ix[s][h] >= 0) != 0ull) pres |= 1u << (2 * s + h);
                unsigned int base = (unsigned int)ix[s][h] * (unsigned int)ROWB + (unsigned int)(kg * 16);
#ifdef DZ_SPCONV_DIAG
                if (a.diag == 2) base |= 0x80000000u;         // gathers out of range: same instructions, nothing fetched
                if (a.diag == 5) base = (unsigned int)(lane * 16);     // every gather hits the same 1 KB
#endif
                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(dst[s][h][0]) : "v"(base), "s"(prsrc));
                if constexpr (KS == 2)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:64" : "=v"(dst[s][h][KS - 1]) : "v"(base), "s"(prsrc));
            }
        }
        pres &= (1u << (2 * c.nslots)) - 1u;                   // repeated slots: loaded, not multiplied
        if (!c.live) pres = 0u;
#ifdef DZ_SPCONV_DIAG
        if (a.diag == 1) pres = 0u;                           // no MFMAs
#endif
    };
    // residual rows of the chunk's tile, in the accumulators' layout (issued only behind the gathers of a tile's LAST chunk)
    auto issue_res = [&](const WChunk &c, uint2 (&dst)[NR]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = c.row0 + h * 16 + n;
            const bool rok = c.live && row < m;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int col = cb * 16 + 4 * kg;
                const unsigned int off = rok ? (unsigned int)(((size_t)row * COUT + (col & ~7)) * 4 + (col & 7) * 2) : OOB_OFFSET;
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(dst[(h * CB + cb) * 2]) : "v"(off), "s"(rrsrc));
                asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:16" : "=v"(dst[(h * CB + cb) * 2 + 1]) : "v"(off), "s"(rrsrc));
            }
        }
    };
    const int wlane = (kg >> 1) * 32 + n;                    // this lane's 16-byte unit inside a (tap, block, k-step) weight slab
    auto mfma_chunk = [&](const WChunk &c, v4u (&src)[G][2][KS], unsigned int pres) {
#pragma unroll
        for (int s = 0; s < G; ++s) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(src[s][h][ks]));
            if (!((pres >> (2 * s)) & 3u)) continue;             // wave-uniform
            const int t = (int)((c.taps >> (5 * s)) & 31u);
            const v4u *wt = w_s + t * (CB * KG * 32) + wlane;    // unit (tap, cb, kk, half, c) = ((tap*CB + cb)*KG + kk)*32 + half*16 + c
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                v4u a1[CB], a2[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    a1[cb] = wt[(cb * KG + ks * 2) * 32];
                    if constexpr (M::TERMS == 1) {               // single product: w_hi meets x_hi only (the k-slots of x_lo get zeros)
                        if (kg & 1) a1[cb] = v4u{0u, 0u, 0u, 0u};
                    } else {
                        a2[cb] = wt[(cb * KG + ks * 2) * 32 + 16];
                    }
                }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!((pres >> (2 * s + h)) & 1u)) continue;  // wave-uniform
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) {
                        if constexpr (M::TERMS != 1) acc[h][cb] = M::mma16(a2[cb], src[s][h][ks], acc[h][cb]);
                        acc[h][cb] = M::mma16(a1[cb], src[s][h][ks], acc[h][cb]);
                    }
                }
            }
        }
    };
    // accumulator of a 16x16 fragment: pixel = lane & 15, channel = 4 * (lane >> 4) + reg
    auto epilogue = [&](const WChunk &c, uint2 (&rv)[NR]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int row = c.row0 + h * 16 + n;
            const bool rok = c.live && row < m;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int col = cb * 16 + 4 * kg;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc[h][cb][e], sc_s[col + e], sc_s[32 + col + e]);
                if (a.residual) {
                    asm volatile("" : "+v"(rv[(h * CB + cb) * 2]), "+v"(rv[(h * CB + cb) * 2 + 1]));
                    const uint2 rh = rv[(h * CB + cb) * 2], rl = rv[(h * CB + cb) * 2 + 1];
                    v[0] += M::join(rh.x & 0xFFFFu, rl.x & 0xFFFFu);
                    v[1] += M::join(rh.x >> 16, rl.x >> 16);
                    v[2] += M::join(rh.y & 0xFFFFu, rl.y & 0xFFFFu);
                    v[3] += M::join(rh.y >> 16, rl.y >> 16);
                }
                if (a.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                uint2 hi, lo;
                split4<M>(v, hi, lo);
#ifdef DZ_SPCONV_DIAG
                if (a.diag == 4 && row != 12345) continue;     // no stores
#endif
                if (rok) {
                    unsigned char *gp = reinterpret_cast<unsigned char *>(a.out) + ((size_t)row * COUT + (col & ~7)) * 4 + (col & 7) * 2;
                    *reinterpret_cast<uint2 *>(gp) = hi;
                    *reinterpret_cast<uint2 *>(gp + 16) = lo;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[h][cb][e] = 0.f;
            }
        }
    };
    const bool has_res = a.residual != nullptr;
    // wait until at most NI + NL (+ NR when `extra`) younger loads are outstanding
    auto wait_for = [&](bool extra) {
        if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI + NL + NR));
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI + NL));
    };
    static_assert(NI + NL + NR <= 63, "vmcnt is a 6-bit counter");

    // ---- prologue: D0 gathers in flight, D1 indices in flight
    WChunk d0 = next_chunk(), d1 = next_chunk(), d2;
    issue_idx(d0, idx[0]);
    issue_idx(d1, idx[1]);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NI));
    issue_gather(d0, idx[0], gat[0], present[0]);
    bool r0 = has_res && d0.last;           // residual loads issued behind gather(d0)
    if (r0) issue_res(d0, resv[0]);
    // steady state, unrolled by two so that every register set has a static index:
    //   1. indices of chunk c+2 -> idx[c & 1]              (its previous user, gather(c), has been issued)
    //   2. wait idx(c+1); gather(c+1) -> gat[(c+1) & 1]     (+ residual rows if c+1 ends its tile)
    //   3. wait gather(c) (+ its residual rows); MFMAs; epilogue if c ends its tile
    // younger loads at both waits: NI (indices of c+2) + NL (a chunk's gathers) [+ NR residual loads of the chunk in between]
    auto step = [&](auto par_t) {
        constexpr int A = decltype(par_t)::value, B = A ^ 1;
        d2 = next_chunk();
        issue_idx(d2, idx[A]);
        wait_for(r0);                                        // idx(d1) landed; behind it: gather(d0) [+ res(d0)], idx(d2)
        issue_gather(d1, idx[B], gat[B], present[B]);
        const bool r1 = has_res && d1.last;
        if (r1) issue_res(d1, resv[B]);
        wait_for(r1);                                        // gather(d0) [+ res(d0)] landed; behind: idx(d2), gather(d1) [+ res(d1)]
        mfma_chunk(d0, gat[A], present[A]);
        if (d0.last) epilogue(d0, resv[A]);
        d0 = d1; d1 = d2; r0 = r1;
    };
    while (d0.live) {
        step(std::integral_constant<int, 0>{});
        if (!d0.live) break;
        step(std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt vmcnt(0)");
}

template <int CIN, int COUT, int G, class M, int WAVES, int OCC, bool PK = false>
static int launch_spconv_w(const SpConvHArgs &a, hipStream_t stream) {
    const int lds = a.kvol * (COUT / 16) * (CIN / 8) * 2 * 16 * 16 + 64 * 4;
    static PerDeviceFlags lds_done;
    if (int rc_ = reserve_lds(reinterpret_cast<const void *>(&k_spconv_w<CIN, COUT, G, M, WAVES, OCC, PK>), 27 * (COUT / 16) * (CIN / 8) * 2 * 16 * 16 + 64 * 4, lds_done, "dz_spconv_forward_split")) return rc_;
    // persistent workgroups: as many as fit on the chip, a multiple of 8 for the XCD schedule
    const int cus = device_cus();
    const int per_cu = (4 * OCC) / WAVES > 0 ? (4 * OCC) / WAVES : 1;
    int grid = cus * per_cu;
    const int need = ceil_div(ceil_div(a.cap, 32), WAVES);
    if (grid > need) grid = need;
    grid = (grid + 7) & ~7;
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL((k_spconv_w<CIN, COUT, G, M, WAVES, OCC, PK>), dim3(grid), dim3(64 * WAVES), lds, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // namespace dz
