// Sparse 3-D convolution on pair16 operands, third generation: TILE-RESIDENT inputs.
//
// k_spconv_h / k_spconv_w (sparse_conv_h.hip, sparse_conv_w.h) gather every (output row, kernel tap) pair from L2: a neighbour row
// is fetched 8-15 times per convolution and their time follows that gather volume (r02: ~5 TB/s of gathered rows whatever the
// channel count; one fp16 MFMA instead of three changes them by 13 %).  Here a workgroup owns TR consecutive output rows and
// stages the rows those outputs read - the tile's HALO, 1.2-1.8x TR rows when the level is kept in the brick key order of
// common.h (LevelGeom layout 1) - in LDS ONCE per 16-channel chunk; all kernel taps then take their operands from LDS:
//   * dz_build_tiles (once per rulebook, shared by the convolutions of an indice_key like the table itself) turns the
//     output-stationary neighbour table into, per tile: the list of distinct input rows (halo) and a LOCAL table
//     ltab[tap][row] = position in that list (uint16, 0xFFFF = no neighbour) - half the bytes of the global table;
//   * the convolution is the resident-tile dense 3x3 kernel's design (conv3x3_d.hip) with the image tile replaced by the halo:
//     nothing is staged in registers (buffer_load_dwordx4 ... lds), 64-byte rows per 16-channel chunk with the XOR swizzle
//     slot = piece ^ ((row >> 2) & 3), the halo double-buffered (the next chunk arrives during the first taps of the current
//     one), weight slices of a (tap, chunk) step through a ring of three, the step's local-table slice through a ring of four,
//     one workgroup barrier per step, static vmcnt counts (every step issues the same number of loads; the ones that are not
//     needed are pointed at a zero region of LDS with an out-of-range offset: zeros arrive, nothing is fetched);
//   * a fragment's B operand is read from the LDS row ltab says (missing neighbour = the zero region): a tap costs no global
//     gather, no index load from HBM, no LDS write;
//   * 8 waves x (PT x 32 rows) x all output channels: the weights of a step are read from L2 once per TR = 512 rows (128 in
//     k_spconv_h), and a wave skips the MFMAs of a 32-row fragment that has no neighbour at the step's tap;
//   * a tile whose halo exceeds the LDS capacity (HL - 1 rows) is processed in passes over slices of its halo list, the
//     accumulators staying in registers - any input works, dense tiles cost extra passes.
// Same arithmetic as k_spconv_h: the three fp16 / bf16 MFMAs per product, taps ascending inside a channel chunk, chunks ascending.
//
// Reference call sites: detection/detzero_det/models/centerpoint_modules/backbone3d.py:64-121, :243-280.
#include <stdlib.h>

#include "hgemm.h"

namespace dz {

constexpr int T_THREADS = 512, T_WAVES = 8;
constexpr int T_HL = 896;                         // LDS rows of a halo buffer (the last one is never loaded: HL - 1 usable; 512-row tiles of the
                                                  // brick order read <= 840 rows on the 160k-point frames; two buffers + rings = 150 KB)
constexpr int T_XBUF = T_HL * 64;                 // bytes
constexpr int T_PXL = T_HL * 4 / T_THREADS;       // direct loads per thread per halo chunk (7)
constexpr int T_ZERO = 1024;                      // zero region: target of the dummy loads, source of missing neighbours
constexpr int T_NW = 3, T_NL = 4;                 // weight ring, local-table ring
constexpr int T_KVOL_MAX = 27;

// ---------------------------------------------------------------------------------------------------------------------------
// tile prepass
// ---------------------------------------------------------------------------------------------------------------------------
// layout of the local table: entry of (tile, tap k, row r of the tile) at ((tile * kvol + k) * TR + perm(r)) with
// perm(r) = (r >> 6) * 64 + (r & 31) * 2 + ((r >> 5) & 1): the two rows (l, l + 32) a lane of the convolution serves are one dword
__device__ __forceinline__ int ltab_perm(int r) { return (r >> 6) * 64 + (r & 31) * 2 + ((r >> 5) & 1); }

// exclusive scan of one value per thread over a block of NT threads (NT / 64 <= 16 waves); lds: >= 16 words
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds, uint32_t &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t s = lds[w];
        if (w < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return woff + incl - v;
}

// One workgroup per tile.  The distinct neighbour rows of the tile are collected in an LDS hash set (open addressing, identity
// hash: runs of consecutive rows keep their order), ranked by a scan over the slots, written as the halo list; the second sweep
// over the table looks every neighbour up again and writes its position.
template <int TR, int NT>
__global__ __launch_bounds__(NT) void k_build_tiles(const int *__restrict__ nbr, const int *__restrict__ d_m_out, int cap, int kvol,
                                                    int hstride, int *__restrict__ halo, int *__restrict__ nhalo,
                                                    unsigned short *__restrict__ ltab) {
    constexpr int HS = TR * 64;                   // slots (>= 2 x the 27 * TR candidates), a power of two
    constexpr int GROUPS = HS / 32;
    static_assert(GROUPS % NT == 0 || NT % GROUPS == 0, "groups per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *const hs = reinterpret_cast<uint32_t *>(smem_raw);                 // [HS] row + 1, 0 = empty
    uint32_t *const occ = hs + HS;                                                // [GROUPS] occupancy of 32 slots
    uint32_t *const base = occ + GROUPS;                                          // [GROUPS] occupied slots before the group
    uint32_t *const scan_s = base + GROUPS;                                       // [16]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = min(*d_m_out, cap);
    const int tile = blockIdx.x, row0 = tile * TR;
    if (row0 >= m) return;
    for (int s = tid; s < HS; s += NT) hs[s] = 0u;
    __syncthreads();
    const int ncand = kvol * TR;
    for (int idx = tid; idx < ncand; idx += NT) {
        const int k = idx / TR, r = idx - k * TR;
        const int row = row0 + r;
        const int v = row < m ? nbr[(size_t)k * cap + row] : -1;
        if (v < 0) continue;
        const uint32_t key = (uint32_t)v + 1u;
        uint32_t h = (uint32_t)v & (HS - 1);
        for (;;) {
            const uint32_t old = atomicCAS(&hs[h], 0u, key);
            if (old == 0u || old == key) break;
            h = (h + 1) & (HS - 1);
        }
    }
    __syncthreads();
    for (int g2 = wid; g2 < HS / 64; g2 += NT / 64) {
        const unsigned long long bal = __ballot(hs[g2 * 64 + lane] != 0u);
        if (lane == 0) { occ[2 * g2] = (uint32_t)bal; occ[2 * g2 + 1] = (uint32_t)(bal >> 32); }
    }
    __syncthreads();
    // exclusive scan of the group counts (GROUPS = 2 * TR entries; NT threads take GROUPS / NT consecutive groups each)
    constexpr int GPT = GROUPS / NT > 0 ? GROUPS / NT : 1;
    uint32_t cnt[GPT], sum = 0;
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid * GPT + j;
        cnt[j] = g < GROUPS ? (uint32_t)__popc(occ[g]) : 0u;
        sum += cnt[j];
    }
    uint32_t total;
    uint32_t run = block_excl_scan<NT>(sum, scan_s, total);
#pragma unroll
    for (int j = 0; j < GPT; ++j) {
        const int g = tid * GPT + j;
        if (g < GROUPS) base[g] = run;
        run += cnt[j];
    }
    if (tid == 0) nhalo[tile] = (int)total;
    __syncthreads();
    int *const hl = halo + (size_t)tile * hstride;
    for (int s = tid; s < HS; s += NT) {
        const uint32_t key = hs[s];
        if (key) hl[base[s >> 5] + __popc(occ[s >> 5] & ((1u << (s & 31)) - 1u))] = (int)(key - 1u);
    }
    unsigned short *const lt = ltab + (size_t)tile * kvol * TR;
    for (int idx = tid; idx < ncand; idx += NT) {
        const int k = idx / TR, r = idx - k * TR;
        const int row = row0 + r;
        const int v = row < m ? nbr[(size_t)k * cap + row] : -1;
        unsigned short pos = 0xFFFFu;
        if (v >= 0) {
            const uint32_t key = (uint32_t)v + 1u;
            uint32_t h = (uint32_t)v & (HS - 1);
            while (hs[h] != key) h = (h + 1) & (HS - 1);
            pos = (unsigned short)(base[h >> 5] + __popc(occ[h >> 5] & ((1u << (h & 31)) - 1u)));
        }
        lt[k * TR + ltab_perm(r)] = pos;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// convolution
// ---------------------------------------------------------------------------------------------------------------------------
struct SpConvTArgs {
    const float *in;              // pair16 rows (in_rows, cin)
    const int *halo;              // (ntiles, hstride)
    const int *nhalo;             // (ntiles)
    const unsigned short *ltab;   // (ntiles, kvol, TR) permuted (ltab_perm)
    const uint32_t *tile_masks;   // per 32 output rows: taps with a neighbour (dz_build_neighbors)
    const int *d_m_out;
    const float *w;               // (kvol, cout_pad, cin) pair16
    const float *scale, *shift;
    const float *residual;        // pair16 rows (cap, cout) or null
    float *out;                   // pair16 rows (cap, cout)
    int cin, cout, kvol, cap, relu, hstride;
    unsigned int in_bytes, w_bytes, ltab_bytes;
};

// 16 bytes (4 bytes) per lane from a buffer straight into LDS at lds_base + lane * 16 (lane * 4); lds_base wave-uniform
// (readfirstlane: both are wave-uniform by construction, the compiler cannot always prove it)
__device__ __forceinline__ void t_load16_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory");
}
__device__ __forceinline__ void t_load4_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory");
}

template <int N>
__device__ __forceinline__ void t_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// CT: 32-channel output fragments (cout_pad = 32 * CT); PT: 32-row fragments per wave (TR = 8 * PT * 32);
// PXS: halo loads per thread and step while the next chunk is being fetched (XSTEPS = ceil(PXL / PXS) steps; a chunk has at least
// XSTEPS + 1 steps - the tap list is padded with empty taps when the tile has fewer)
template <class M, int CT, int PT, int PXS>
__global__ __launch_bounds__(T_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_spconv_t(SpConvTArgs a) {
    constexpr int TR = T_WAVES * PT * 32;
    constexpr int BC = CT * 32;
    constexpr int WBUF = BC * 64;                               // bytes of a weight slice (16 channels)
    constexpr int WL = (BC * 4 + T_THREADS - 1) / T_THREADS;    // weight loads per thread and step (1)
    constexpr int XSTEPS = (T_PXL + PXS - 1) / PXS;
    constexpr int MINSTEPS = XSTEPS + 1;
    constexpr int OFF_X = 0, OFF_ZERO = 2 * T_XBUF, OFF_W = OFF_ZERO + T_ZERO, OFF_L = OFF_W + T_NW * WBUF;
    constexpr int LBUF = T_WAVES * 256;                         // a step's local-table slice: one dword per lane and wave
    constexpr int OFF_HAL = OFF_L + T_NL * LBUF, OFF_SS = OFF_HAL + T_HL * 4, OFF_TAP = OFF_SS + 2 * BC * 4;
    static_assert(WL == 1, "one weight piece per thread and step at most");
    using ET = HTile<TR, BC, 16, T_WAVES, 1>;                   // epilogue traits (store_tile_pair16)
    static_assert(ET::PT == PT && ET::CT == CT, "epilogue traits");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int *const hal_s = reinterpret_cast<int *>(smem_raw + OFF_HAL);
    float *const sc_s = reinterpret_cast<float *>(smem_raw + OFF_SS), *const sh_s = sc_s + BC;
    int *const tap_s = reinterpret_cast<int *>(smem_raw + OFF_TAP);          // [0] = number of steps per chunk, [1 + i] = tap of step i

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int m = min(*a.d_m_out, a.cap);
    // XCD-aware deal of the tiles: runs of XRUN consecutive tiles (neighbouring bricks: shared halo rows) per XCD
    constexpr int XRUN = 4;
    const int tile = (((int)(blockIdx.x >> 3) / XRUN) * 8 + (int)(blockIdx.x & 7)) * XRUN + (int)(blockIdx.x >> 3) % XRUN;
    const int row0 = tile * TR;
    if (row0 >= m) return;

    const srsrc_t prsrc = make_srsrc(a.in, a.in_bytes);
    const srsrc_t crsrc = make_srsrc(a.w, a.w_bytes);
    const srsrc_t lrsrc = make_srsrc(a.ltab, a.ltab_bytes);
    const int nk = a.cin / 16;
    const unsigned int row_bytes = (unsigned int)a.cin * 4u;
    const unsigned int tap_bytes = (unsigned int)(BC * a.cin * 4);
    const int nh = a.nhalo[tile];
    const int *const hl_g = a.halo + (size_t)tile * a.hstride;
    const int npass = nh > 0 ? (nh + (T_HL - 2)) / (T_HL - 1) : 1;

    // tap masks of my two fragments and of the tile; the tile's step list
    unsigned int fm[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int r32 = (row0 >> 5) + wid * PT + pt;
        fm[pt] = (r32 << 5) < m ? a.tile_masks[r32] : 0u;
        fm[pt] = __builtin_amdgcn_readfirstlane(fm[pt]);
    }
    if (tid < BC) {
        const bool in = tid < a.cout;
        sc_s[tid] = (in && a.scale) ? a.scale[tid] : 1.f;
        sh_s[tid] = (in && a.shift) ? a.shift[tid] : 0.f;
    }
    if (tid == 0) {
        unsigned int tm = 0u;
        for (int i = 0; i < TR / 32; ++i) {
            const int r32 = (row0 >> 5) + i;
            if ((r32 << 5) < m) tm |= a.tile_masks[r32];
        }
        int n = 0;
        for (int k = 0; k < a.kvol; ++k)
            if ((tm >> k) & 1u) tap_s[1 + n++] = k;
        for (int k = 0; k < a.kvol && n < MINSTEPS; ++k)         // pad with empty taps (their local-table rows are all 0xFFFF)
            if (!((tm >> k) & 1u)) tap_s[1 + n++] = k;
        tap_s[0] = n;
    }
    __syncthreads();
    const int ntap = __builtin_amdgcn_readfirstlane(tap_s[0]);
    if (ntap < MINSTEPS) return;                                  // (kvol < MINSTEPS: refused by the launcher)

    f32x16 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // weight piece of this thread: row wr = tid >> 2 of the slice, LDS slot tid & 3 holds piece (tid & 3) ^ ((wr >> 2) & 3)
    const int wr = tid >> 2;
    const unsigned int wvoff = wr < BC ? (unsigned int)(wr * a.cin * 4 + (((tid & 3) ^ ((wr >> 2) & 3)) * 16)) : OOB_OFFSET;
    const unsigned int w_lds = (unsigned int)(wid * 1024);
    const bool w_real = wid * 16 < BC;                                         // wave-uniform: waves beyond the slice write zeros into the zero region
    // local-table dword of this lane for tap k: byte ((tile * kvol + k) * TR + wid * 64 + l31 * 2) * 2
    const unsigned int lvoff = (unsigned int)((wid * 64 + l31 * 2) * 2);
    const unsigned int ltile = (unsigned int)((size_t)tile * a.kvol * TR * 2);

    struct It { int ti, kc; };                                   // position in the (chunk, tap) step sequence of a pass
    auto next = [&](It &it) { if (++it.ti == ntap) { it.ti = 0; ++it.kc; } };
    auto issue_w = [&](const It &it, int slot) {
        const bool ok = it.kc < nk;
        const int tap = __builtin_amdgcn_readfirstlane(tap_s[1 + it.ti]);
        t_load16_lds((unsigned int)(w_real ? OFF_W + slot * WBUF + w_lds : OFF_ZERO), ok ? wvoff : OOB_OFFSET, crsrc,
                     ok ? (unsigned int)tap * tap_bytes + (unsigned int)(it.kc * 64) : 0u);
    };
    auto issue_l = [&](const It &it, int slot) {
        const bool ok = it.kc < nk;
        const int tap = __builtin_amdgcn_readfirstlane(tap_s[1 + it.ti]);
        t_load4_lds((unsigned int)(OFF_L + slot * LBUF + wid * 256), ok ? lvoff : OOB_OFFSET, lrsrc, ok ? ltile + (unsigned int)(tap * TR * 2) : 0u);
    };
    // halo chunk kc into X buffer `buf`: load i of wave w fills pieces (i * 8 + w) * 64 + lane
    auto issue_x = [&](int i, int kc, int buf, bool real) {
        int pc = (i * T_WAVES + wid) * 64 + lane;
        asm volatile("" : "+v"(pc));
        const int r = pc >> 2;
        const int g = hal_s[r];
        const unsigned int voff = (real && g >= 0) ? (unsigned int)g * row_bytes + (unsigned int)(((pc & 3) ^ ((r >> 2) & 3)) * 16) : OOB_OFFSET;
        t_load16_lds((unsigned int)(real ? OFF_X + buf * T_XBUF + (i * T_WAVES + wid) * 1024 : OFF_ZERO), voff, prsrc, (unsigned int)(kc * 64));
    };

    struct Frag { v4u p_hi[PT], p_lo[PT], c_hi[CT], c_lo[CT]; };
    const unsigned int c_lds = (unsigned int)(OFF_W + l31 * 64 + (((2 * kg) ^ ((l31 >> 2) & 3)) * 16));
    // lt: the lane's local-table dword of the step (rows l31 and l31 + 32 of the wave: low / high half)
    auto load_frag = [&](Frag &f, unsigned int lt, int pass_base, int xbuf, int wslot) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const unsigned int e = (pt & 1) ? (lt >> 16) : (lt & 0xFFFFu);
            const unsigned int lr = e - (unsigned int)pass_base;
            const unsigned int ax = (unsigned int)(OFF_X + xbuf * T_XBUF) + lr * 64u + (((2u * kg) ^ ((lr >> 2) & 3u)) * 16u);
            const unsigned int adr = lr < (unsigned int)(T_HL - 1) ? ax : (unsigned int)(OFF_ZERO + kg * 32);
            f.p_hi[pt] = *reinterpret_cast<const v4u *>(smem_raw + adr);
            f.p_lo[pt] = *reinterpret_cast<const v4u *>(smem_raw + (adr ^ 16u));
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const unsigned int adr = c_lds + (unsigned int)(wslot * WBUF + ct * 2048);
            f.c_hi[ct] = *reinterpret_cast<const v4u *>(smem_raw + adr);
            f.c_lo[ct] = *reinterpret_cast<const v4u *>(smem_raw + (adr ^ 16u));
        }
    };
    static_assert(PT <= 2, "a local-table dword carries two rows");
    auto mma = [&](const Frag &f, int tap) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            if (!((fm[pt] >> tap) & 1u)) continue;               // wave-uniform: this 32-row fragment has no neighbour at the tap
#pragma unroll
            for (int term = 3 - M::TERMS; term < 3; ++term)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct)
                    acc[ct][pt] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[pt] : f.p_hi[pt], acc[ct][pt]);
        }
    };

    for (int pass = 0; pass < npass; ++pass) {
        const int pass_base = pass * (T_HL - 1);
        // halo slice of the pass -> LDS (row HL - 1 is never loaded: the slice has at most HL - 1 entries)
        for (int r = tid; r < T_HL; r += T_THREADS) hal_s[r] = (r < T_HL - 1 && pass_base + r < nh) ? hl_g[pass_base + r] : -1;
        __syncthreads();
        // ---- prologue: halo chunk 0, weights of steps 0 and 1, local tables of steps 0, 1, 2, the zero region
        t_load16_lds((unsigned int)OFF_ZERO, OOB_OFFSET, prsrc, 0u);
#pragma unroll
        for (int i = 0; i < T_PXL; ++i) issue_x(i, 0, 0, true);
        It iw{0, 0}, il{0, 0}, ic{0, 0};
        issue_w(iw, 0); next(iw);
        issue_w(iw, 1); next(iw);
        issue_l(il, 0); next(il);
        issue_l(il, 1); next(il);
        issue_l(il, 2); next(il);
        t_wait_vm<0>();
        __syncthreads();
        const unsigned int lt_0 = *reinterpret_cast<const unsigned int *>(smem_raw + OFF_L + 0 * LBUF + wid * 256 + (lane & 31) * 4);
        unsigned int lt_nxt = *reinterpret_cast<const unsigned int *>(smem_raw + OFF_L + 1 * LBUF + wid * 256 + (lane & 31) * 4);
        Frag fa, fb;
        load_frag(fa, lt_0, pass_base, 0, 0);
        const int nsteps = nk * ntap;
        // step u: (ic.kc, ic.ti).  iw is at step u + 2, il at step u + 3.
        auto step = [&](int u, Frag &fcur, Frag &fnxt) {
            const int tap = __builtin_amdgcn_readfirstlane(tap_s[1 + ic.ti]);
            const int kc = ic.kc;
            // halo of the next chunk during the first XSTEPS steps of this one (dummies otherwise: static load counts)
            {
                const bool real = ic.ti < XSTEPS && kc + 1 < nk;
#pragma unroll
                for (int j = 0; j < PXS; ++j) {
                    const int i = (ic.ti < XSTEPS ? ic.ti : 0) * PXS + j;
                    issue_x(i < T_PXL ? i : 0, kc + 1, (kc + 1) & 1, real && i < T_PXL);
                }
            }
            issue_w(iw, (u + 2) % T_NW); next(iw);
            issue_l(il, (u + 3) % T_NL); next(il);
            // everything issued before this step has landed: weights + local table of step u + 1 (and u + 2's table), and -
            // at a chunk's last step - the next chunk's halo (its last loads were issued at step XSTEPS - 1 < ntap - 1)
            t_wait_vm<PXS + WL + 1>();
            __syncthreads();
            It in = ic;
            next(in);
            const unsigned int lt_nn = *reinterpret_cast<const unsigned int *>(smem_raw + OFF_L + ((u + 2) % T_NL) * LBUF + wid * 256 + (lane & 31) * 4);
            if (u + 1 < nsteps) load_frag(fnxt, lt_nxt, pass_base, in.kc & 1, (u + 1) % T_NW);
            mma(fcur, tap);
            lt_nxt = lt_nn;
            ic = in;
        };
        for (int u = 0; u < nsteps; u += 2) {
            step(u, fa, fb);
            if (u + 1 < nsteps) step(u + 1, fb, fa);
        }
        t_wait_vm<0>();                                          // the look-ahead loads of the last steps (dummies) before LDS is reused
        __syncthreads();
    }

    // ---- epilogue through the idle halo buffers (hgemm.h)
    store_tile_pair16<ET, M>(acc, smem_raw, sc_s, sh_s, 0, a.cout, a.relu != 0, reinterpret_cast<const unsigned char *>(a.residual),
                             reinterpret_cast<unsigned char *>(a.out), wid, 0, lane, wid, [&](int lr) {
                                 const int row = row0 + lr;
                                 return row < m ? (size_t)row * a.cout * 4 : ~size_t(0);
                             });
}

template <int CT>
constexpr int t_lds_bytes() {
    return 2 * T_XBUF + T_ZERO + T_NW * CT * 32 * 64 + T_NL * T_WAVES * 256 + T_HL * 4 + 2 * CT * 32 * 4 + 32 * 4;
}

template <class M, int CT, int PT, int PXS>
static int launch_spconv_t(const SpConvTArgs &a, hipStream_t stream) {
    constexpr int LDS = t_lds_bytes<CT>();
    constexpr int TR = T_WAVES * PT * 32;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DZ_ERR_HIP;
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_spconv_t<M, CT, PT, PXS>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            set_error("dz_spconv_tiles_forward: cannot reserve %d bytes of LDS", LDS);
            return DZ_ERR_HIP;
        }
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    int grid = ceil_div(a.cap, TR);
    grid = (grid + 31) & ~31;            // whole runs of XRUN tiles on every XCD
    hipLaunchKernelGGL((k_spconv_t<M, CT, PT, PXS>), dim3(grid), dim3(T_THREADS), LDS, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

template <class M>
static int spconv_t_dispatch(const SpConvTArgs &a, int cout_pad, hipStream_t stream) {
    // kvol 27: the next halo chunk arrives over 4 steps (a chunk has >= 5); small kernels (3 x 1 x 1): over 2 steps (>= 3)
    if (a.kvol >= 5) {
        if (cout_pad == 32) return launch_spconv_t<M, 1, 2, 2>(a, stream);
        if (cout_pad == 64) return launch_spconv_t<M, 2, 2, 2>(a, stream);
        if (cout_pad == 128) return launch_spconv_t<M, 4, 2, 2>(a, stream);
    } else if (a.kvol >= 3) {
        if (cout_pad == 32) return launch_spconv_t<M, 1, 2, 4>(a, stream);
        if (cout_pad == 64) return launch_spconv_t<M, 2, 2, 4>(a, stream);
        if (cout_pad == 128) return launch_spconv_t<M, 4, 2, 4>(a, stream);
    }
    set_error("dz_spconv_tiles_forward: unsupported shape cout=%d kvol=%d", a.cout, a.kvol);
    return DZ_ERR_UNSUPPORTED;
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_spconv_tile_rows(void) { return T_WAVES * 2 * 32; }

size_t dz_build_tiles_halo_stride(int kvol) { return (size_t)(kvol < 1 ? 1 : kvol) * (T_WAVES * 2 * 32); }

int dz_build_tiles(const int *nbr, int kvol, int cap_out, const int *d_m_out, int *halo, int *nhalo, unsigned short *ltab, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int TR = T_WAVES * 2 * 32, NT = 1024;
    DZ_CHECK_ARG(nbr && d_m_out && halo && nhalo && ltab, "dz_build_tiles: null pointer");
    DZ_CHECK_ARG(kvol >= 1 && kvol <= T_KVOL_MAX && cap_out >= 0, "dz_build_tiles: kvol %d not in [1,27]", kvol);
    if (cap_out == 0) return DZ_OK;
    constexpr int LDS = TR * 64 * 4 + 2 * (TR * 64 / 32) * 4 + 64;
    static bool attr_set[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return DZ_ERR_HIP;
    if (dev < 0 || dev >= 16 || !attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&k_build_tiles<TR, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            set_error("dz_build_tiles: cannot reserve %d bytes of LDS", LDS);
            return DZ_ERR_HIP;
        }
        if (dev >= 0 && dev < 16) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_build_tiles<TR, NT>), dim3(ceil_div(cap_out, TR)), dim3(NT), LDS, stream, nbr, d_m_out, cap_out, kvol,
                       (int)dz_build_tiles_halo_stride(kvol), halo, nhalo, ltab);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_spconv_tiles_forward(const float *in, int in_rows, int cin, const int *halo, const int *nhalo, const unsigned short *ltab,
                            const uint32_t *tile_masks, int kvol, int cap_out, const int *d_m_out, const float *w, const float *scale,
                            const float *shift, const float *residual, int relu, float *out, int cout, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int TR = T_WAVES * 2 * 32;
    DZ_CHECK_ARG(in && halo && nhalo && ltab && tile_masks && d_m_out && w && out, "dz_spconv_tiles_forward: null pointer");
    DZ_CHECK_ARG(kvol >= 3 && kvol <= T_KVOL_MAX, "dz_spconv_tiles_forward: kvol %d not in [3,27]", kvol);
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2 || math == DZ_MATH_F16, "dz_spconv_tiles_forward: math %d is not a split mode", math);
    DZ_CHECK_ARG(cout % 8 == 0 && cin % 16 == 0, "dz_spconv_tiles_forward: cin must be a multiple of 16, cout of 8");
    if (cap_out == 0) return DZ_OK;
    const int cout_pad = cout < 32 ? 32 : cout;
    const size_t in_bytes = (size_t)in_rows * cin * sizeof(float);
    const size_t w_bytes = (size_t)kvol * cout_pad * cin * sizeof(float);
    const size_t ltab_bytes = (size_t)ceil_div(cap_out, TR) * kvol * TR * 2;
    if (in_rows < 0 || in_bytes >= 0x80000000ull || ltab_bytes >= 0x80000000ull) {
        set_error("dz_spconv_tiles_forward: input of %zu bytes / local table of %zu bytes exceed the 2 GiB buffer-addressing limit", in_bytes, ltab_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    SpConvTArgs a{in, halo, nhalo, ltab, tile_masks, d_m_out, w, scale, shift, residual, out, cin, cout, kvol, cap_out, relu,
                  (int)dz_build_tiles_halo_stride(kvol), (unsigned int)in_bytes, (unsigned int)w_bytes, (unsigned int)ltab_bytes};
    if (math == DZ_MATH_F16) return spconv_t_dispatch<MathF16H>(a, cout_pad, stream);
    return math == DZ_MATH_F16X2 ? spconv_t_dispatch<MathF16>(a, cout_pad, stream) : spconv_t_dispatch<MathBF16>(a, cout_pad, stream);
}

const char *dz_spconv_tiles_variant(int cin, int cout) {
    const int cout_pad = cout < 32 ? 32 : cout;
    if (cout_pad == 32) return "k_spconv_t<512x32>";
    if (cout_pad == 64) return "k_spconv_t<512x64>";
    if (cout_pad == 128) return "k_spconv_t<512x128>";
    return "none";
}

}  // extern "C"
