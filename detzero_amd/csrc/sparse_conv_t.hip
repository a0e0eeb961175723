// Sparse 3-D convolution on pair16 operands, third generation: TILE-RESIDENT inputs.
//
// k_spconv_h / k_spconv_w (sparse_conv_h.hip, sparse_conv_w.h) gather every (output row, kernel tap) pair from L2: a neighbour row
// is fetched 8-15 times per convolution.  Here a workgroup owns TR = 512 consecutive output rows and stages the rows those outputs
// read - the tile's HALO, 1.2-1.4x TR rows when the level is kept in the brick key order of common.h (LevelGeom layout 1) - in LDS
// ONCE per 16-channel chunk; all kernel taps then take their operands from LDS:
//   * dz_build_tiles (once per rulebook, shared by the convolutions of an indice_key like the table itself) turns the
//     output-stationary neighbour table into, per tile: the list of distinct input rows (halo); the tile's SLOT list (its non-empty
//     kernel taps, ascending); a LOCAL table ltab[slot][row] = position in the halo list (uint16, 0xFFFF = no neighbour) - half
//     the bytes of the global table; and an order of the tile's rows: rows are SORTED BY THEIR TAP SET inside the tile, so that
//     the 32-row MFMA fragments hold rows with similar neighbourhoods and a fragment can skip the slots none of its rows has
//     (26 -> ~20 of 27 taps issued per fragment on the 160k-point frames; spconv sorts whole tensors by mask for the same reason
//     - with a gather kernel that costs more in locality than it saves, with the inputs resident in LDS it is free);
//   * the convolution is the resident-tile dense 3x3 kernel's design (conv3x3_d.hip) with the image tile replaced by the halo:
//     nothing is staged in registers (buffer_load_dwordx4 ... lds), 64-byte rows per 16-channel chunk with the XOR swizzle
//     slot = piece ^ ((row >> 2) & 3), the halo double-buffered (the next chunk arrives during the first steps of the current
//     one), weights and local-table slices through rings, static vmcnt counts (every step issues the same number of loads; the
//     ones that are not needed go to a zero region of LDS with an out-of-range offset: zeros arrive, nothing is fetched);
//   * a STEP = G slots x 16 channels between two workgroup barriers (G = 4 / 2 / 1 for 32 / 64 / 128 output channels): the
//     bookkeeping of a step (~130 scalar / address instructions per wave: a wave issues one instruction at a time, so they
//     cost as much as 4 MFMAs) is paid once per G taps, and the inline-asm loads - scheduling barriers for the compiler - are
//     dealt out between the MFMA groups by hand;
//   * a fragment's B operand is read from the LDS row ltab says (missing neighbour = the zero region): a tap costs no global
//     gather, no index load from HBM, no LDS write;
//   * 8 waves x 64 rows x all output channels: the weights of a step are read from L2 once per 512 rows (128 in k_spconv_h);
//   * a tile whose halo exceeds the LDS capacity (HL - 1 rows) is processed in passes over slices of its halo list, the
//     accumulators staying in registers - any input works, dense tiles cost extra passes.
// Same arithmetic as k_spconv_h: the three fp16 / bf16 MFMAs per product, taps ascending inside a channel chunk, chunks ascending.
//
// Reference call sites: detection/detzero_det/models/centerpoint_modules/backbone3d.py:64-121, :243-280.
#include <stdlib.h>

#include "hgemm.h"

#ifdef DZ_BUILD_EXPERIMENTAL
namespace dz {

constexpr int T_THREADS = 512, T_WAVES = 8, T_TR = 512;
constexpr int T_HL = 896;                         // LDS rows of a halo buffer (the last one is never loaded: HL - 1 usable; 512-row tiles of the
                                                  // brick order read <= 840 rows on the 160k-point frames; two buffers + rings = 155-160 KB)
constexpr int T_XBUF = T_HL * 64;                 // bytes
constexpr int T_PXL = T_HL * 4 / T_THREADS;       // direct loads per thread per halo chunk (7)
constexpr int T_ZERO = 1024;                      // zero region: target of the dummy loads, source of missing neighbours
constexpr int T_KVOL_MAX = 27;
constexpr int T_SLOTS = 32;                       // slot capacity of a tile (>= kvol, a multiple of 4)
constexpr int T_INFO = 64;                        // int32 words of a tile's info record
constexpr int TI_NSLOTS = 0, TI_NHALO = 1, TI_TAP = 4, TI_FSLOT = 36;     // info record: [4 .. 35] tap of slot s (-1: padding), [36 .. 51] slot bits of sorted fragment f
constexpr int T_LTAB = T_SLOTS * T_TR;            // uint16 entries of a tile's local table (32 KB)

// uint16 index of the local-table entry of (slot s, sorted position q): [slot group s / 4][wave q / 64][lane q % 32][s % 4][q / 32 % 2] -
// the 16 bytes a lane needs for one group of four slots (its two rows l and l + 32) are contiguous, a wave's slice is 512 bytes
__device__ __forceinline__ int ltab_index(int s, int q) { return ((((s >> 2) * T_WAVES + (q >> 6)) * 32 + (q & 31)) * 4 + (s & 3)) * 2 + ((q >> 5) & 1); }

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

// ---------------------------------------------------------------------------------------------------------------------------
// tile prepass
// ---------------------------------------------------------------------------------------------------------------------------
// One workgroup per tile.  The distinct neighbour rows of the tile are collected in an LDS hash set (open addressing, identity
// hash: runs of consecutive rows keep their order), ranked by a scan over the slots and written as the halo list; the rows are
// sorted by (touches the z - 1 plane, touches the z + 1 plane, tap set) with a bitonic sort; the second sweep over the table
// looks every neighbour up again and writes its position at (slot of the tap, sorted position of the row).
template <int NT>
__global__ __launch_bounds__(NT) void k_build_tiles(const int *__restrict__ nbr, const int *__restrict__ d_m_out, int cap, int kvol,
                                                    int hstride, int *__restrict__ halo, int *__restrict__ tinfo,
                                                    unsigned short *__restrict__ ltab, unsigned short *__restrict__ rowmap) {
    constexpr int TR = T_TR;
    constexpr int HS = TR * 64;                   // slots (>= 2 x the 27 * TR candidates), a power of two
    constexpr int GROUPS = HS / 32;
    static_assert(GROUPS == NT, "one occupancy group per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *const hs = reinterpret_cast<uint32_t *>(smem_raw);                 // [HS] row + 1, 0 = empty
    uint32_t *const occ = hs + HS;                                                // [GROUPS] occupancy of 32 slots
    uint32_t *const base = occ + GROUPS;                                          // [GROUPS] occupied slots before the group
    uint32_t *const rmask = base + GROUPS;                                        // [TR] tap set of a row
    unsigned long long *const keys = reinterpret_cast<unsigned long long *>(rmask + TR);     // [TR] sort keys
    unsigned short *const inv = reinterpret_cast<unsigned short *>(keys + TR);    // [TR] sorted position of a row
    uint32_t *const scan_s = reinterpret_cast<uint32_t *>(inv + TR);              // [16], [16] = tile tap mask
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int m = min(*d_m_out, cap);
    const int tile = blockIdx.x, row0 = tile * TR;
    if (row0 >= m) return;
    for (int s = tid; s < HS; s += NT) hs[s] = 0u;
    if (tid < TR) rmask[tid] = 0u;
    if (tid == 0) scan_s[16] = 0u;
    {   // the whole local table of the tile = "no neighbour"
        uint4 *lt4 = reinterpret_cast<uint4 *>(ltab + (size_t)tile * T_LTAB);
        for (int i = tid; i < T_LTAB * 2 / 16; i += NT) lt4[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
    __syncthreads();
    const int ncand = kvol * TR;
    for (int idx = tid; idx < ncand; idx += NT) {
        const int k = idx / TR, r = idx - k * TR;
        const int row = row0 + r;
        const int v = row < m ? nbr[(size_t)k * cap + row] : -1;
        if (v < 0) continue;
        atomicOr(&rmask[r], 1u << k);
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
    if (tid < TR) {
        const uint32_t mk = rmask[tid];
        if (mk) atomicOr(&scan_s[16], mk);
        // sort key: rows of the tile by (reaches below, reaches above, tap set); rows past the end last
        const uint32_t zk = (kvol == 27) ? ((((mk & 0x1FFu) != 0u) ? 2u : 0u) | (((mk >> 18) != 0u) ? 1u : 0u)) : 0u;
        keys[tid] = (row0 + tid < m) ? (((unsigned long long)zk << 36) | ((unsigned long long)mk << 9) | (unsigned long long)tid)
                                     : (~0ull << 9) | (unsigned long long)tid;
    }
    __syncthreads();
    const uint32_t cnt = (uint32_t)__popc(occ[tid]);
    uint32_t total;
    base[tid] = block_excl_scan<NT>(cnt, scan_s, total);
    // bitonic sort of the TR keys (threads 0 .. TR - 1, one element each)
    for (int k = 2; k <= TR; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            if (tid < TR) {
                const int o = tid ^ j;
                if (o > tid) {
                    const unsigned long long a = keys[tid], b = keys[o];
                    if ((a > b) == ((tid & k) == 0)) { keys[tid] = b; keys[o] = a; }
                }
            }
        }
    __syncthreads();
    const uint32_t tmask = scan_s[16];
    int *const ti = tinfo + (size_t)tile * T_INFO;
    if (tid < TR) {
        const int r = (int)(keys[tid] & 511ull);
        inv[r] = (unsigned short)tid;
        rowmap[(size_t)tile * TR + tid] = (unsigned short)r;
        // slot bits of this sorted position's row, OR-ed over the 32 rows of its fragment
        uint32_t mk = rmask[r], sb = 0u;
        while (mk) {
            const int t = __ffs((int)mk) - 1;
            mk &= mk - 1;
            sb |= 1u << __popc(tmask & ((1u << t) - 1u));
        }
#pragma unroll
        for (int d = 16; d >= 1; d >>= 1) sb |= (uint32_t)__shfl_xor((int)sb, d, 64);
        if ((tid & 31) == 0) ti[TI_FSLOT + (tid >> 5)] = (int)sb;
    }
    if (tid < T_SLOTS) {                          // slot s = the s-th set bit of the tile's tap mask
        uint32_t rem = tmask;
        int tap = -1;
        for (int i = 0; i <= tid; ++i) {
            if (!rem) { tap = -1; break; }
            tap = __ffs((int)rem) - 1;
            rem &= rem - 1;
        }
        ti[TI_TAP + tid] = tap;
    }
    if (tid == 0) { ti[TI_NSLOTS] = __popc(tmask); ti[TI_NHALO] = (int)total; ti[2] = 0; ti[3] = 0; }
    __syncthreads();
    int *const hl = halo + (size_t)tile * hstride;
    for (int s = tid; s < HS; s += NT) {
        const uint32_t key = hs[s];
        if (key) hl[base[s >> 5] + __popc(occ[s >> 5] & ((1u << (s & 31)) - 1u))] = (int)(key - 1u);
    }
    unsigned short *const lt = ltab + (size_t)tile * T_LTAB;
    for (int idx = tid; idx < ncand; idx += NT) {
        const int k = idx / TR, r = idx - k * TR;
        const int row = row0 + r;
        const int v = row < m ? nbr[(size_t)k * cap + row] : -1;
        if (v < 0) continue;
        const uint32_t key = (uint32_t)v + 1u;
        uint32_t h = (uint32_t)v & (HS - 1);
        while (hs[h] != key) h = (h + 1) & (HS - 1);
        const int pos = (int)(base[h >> 5] + __popc(occ[h >> 5] & ((1u << (h & 31)) - 1u)));
        lt[ltab_index(__popc(tmask & ((1u << k) - 1u)), inv[r])] = (unsigned short)pos;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// convolution
// ---------------------------------------------------------------------------------------------------------------------------
struct SpConvTArgs {
    const float *in;              // pair16 rows (in_rows, cin)
    const int *halo;              // (ntiles, hstride)
    const int *tinfo;             // (ntiles, T_INFO)
    const unsigned short *ltab;   // (ntiles, T_LTAB)
    const unsigned short *rowmap; // (ntiles, TR): row (relative to the tile) at a sorted position
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
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory", "m0");
}
__device__ __forceinline__ void t_load4_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory", "m0");
}

template <int N>
__device__ __forceinline__ void t_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CT, int G, int D>
struct TCfg {
    static constexpr int BC = CT * 32;
    static constexpr int WBUF = BC * 64;                          // bytes of one slot's weight slice (16 channels)
    static constexpr int WSLOT = G * WBUF;                        // bytes of a step's weights
    static constexpr int WL = (G * BC * 4 + T_THREADS - 1) / T_THREADS;   // weight loads per thread and step
    static constexpr int LSLOT = G == 4 ? 4096 : G * 1024;        // bytes of a step's local table
    static constexpr int LL = G == 4 ? 1 : G;                     // local-table loads per thread and step (waves 0-3 real)
    // rings: a slot is rewritten right after the barrier that follows its last read.  G = 1: a step's weights are read during the
    // step before it (fragments are loaded one step ahead), its table two steps before; G > 1: the weights of slots 1 .. G - 1 are
    // read during the step itself, the table one step before
    static constexpr int NW = G == 1 ? D + 2 : D + 3, NL = G == 1 ? D + 3 : D + 2;
    static constexpr bool HREG = CT == 1;                         // halo-load offsets of a pass in registers instead of LDS (32 output channels: LDS is short, registers are not)
    static constexpr int OFF_X = 0, OFF_ZERO = 2 * T_XBUF, OFF_W = OFF_ZERO + T_ZERO, OFF_L = OFF_W + NW * WSLOT;
    static constexpr int OFF_HAL = OFF_L + NL * LSLOT, OFF_SS = OFF_HAL + (HREG ? T_TR * 2 : T_HL * 4), OFF_END = OFF_SS + 2 * BC * 4;
    static constexpr int LDS = OFF_END;
};

// CT: 32-channel output fragments (cout_pad = 32 * CT).  G: slots per step.  D: prefetch distance in steps - the loads a step
// issues (weights and local table of step u + D + 2 / u + D + 3, a share of the next halo chunk) have to land only by the wait of
// step u + D + 1.  PXS: halo loads per thread and step while the next chunk is fetched (XSTEPS = ceil(PXL / PXS) steps; a chunk
// has at least XSTEPS + D + 1 steps - the slot list is padded with empty slots when the tile has fewer taps).
template <class M, int CT, int G, int PXS, int D>
__global__ __launch_bounds__(T_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_spconv_t(SpConvTArgs a) {
    using C = TCfg<CT, G, D>;
    constexpr int PT = 2, TR = T_TR, BC = C::BC, WBUF = C::WBUF, WL = C::WL, LL = C::LL;
    constexpr int XSTEPS = (T_PXL + PXS - 1) / PXS;
    constexpr int MINSTEPS = XSTEPS + D + 1;
    constexpr int LPS = PXS + WL + LL;                          // loads per thread and step
    static_assert(D * LPS <= 63, "vmcnt is a 6-bit counter");
    static_assert(G == 1 || G == 2 || G == 4, "slots per step");
    static_assert(G * BC * 4 >= T_THREADS, "every thread loads a weight piece");
    using ET = HTile<TR, BC, 16, T_WAVES, 1>;                   // epilogue traits (store_tile_pair16)
    static_assert(ET::PT == PT && ET::CT == CT, "epilogue traits");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int *const hal_s = reinterpret_cast<int *>(smem_raw + C::OFF_HAL);
    float *const sc_s = reinterpret_cast<float *>(smem_raw + C::OFF_SS), *const sh_s = sc_s + BC;

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
    const int *const ti = a.tinfo + (size_t)tile * T_INFO;
    const int nh = __builtin_amdgcn_readfirstlane(ti[TI_NHALO]);
    const int nsl_real = __builtin_amdgcn_readfirstlane(ti[TI_NSLOTS]);
    const int *const hl_g = a.halo + (size_t)tile * a.hstride;
    const int npass = nh > 0 ? (nh + (T_HL - 2)) / (T_HL - 1) : 1;
    // steps of a chunk: the tile's slots in groups of G, padded (empty slots: no weights, no neighbours, no MFMAs) to MINSTEPS
    int nst = (nsl_real + G - 1) / G;
    if (nst < MINSTEPS) nst = MINSTEPS;
    // tap of every slot in a register (lane s = slot s; -1 = padding): read with v_readlane, no memory access in the loop
    const int tapvec = ti[TI_TAP + (lane & 31)];
    auto tap_at = [&](int s) { return __builtin_amdgcn_readlane(tapvec, s); };
    // slots my two fragments (sorted rows wid * 64 .. + 31, + 32 .. + 63) take part in
    unsigned int fs[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) fs[pt] = __builtin_amdgcn_readfirstlane((unsigned int)ti[TI_FSLOT + wid * PT + pt]);
    if (tid < BC) {
        const bool in = tid < a.cout;
        sc_s[tid] = (in && a.scale) ? a.scale[tid] : 1.f;
        sh_s[tid] = (in && a.shift) ? a.shift[tid] : 0.f;
    }

    f32x16 acc[CT][PT];
#pragma unroll
    for (int i = 0; i < CT; ++i)
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // weight piece j of this thread in a step's G slices: piece index pw = j * 512 + tid -> slot g = pw / (BC * 4) (wave-uniform: a
    // wave covers 64 consecutive pieces, a slice has BC * 4 >= 128), row wr, LDS piece-slot pw & 3 holds global piece (pw & 3) ^ ((wr >> 2) & 3)
    auto wpiece_g = [&](int j) { return (j * T_THREADS + wid * 64) / (BC * 4); };
    auto wpiece_voff = [&](int j) {
        const int pw = j * T_THREADS + tid;
        const int wr = (pw % (BC * 4)) >> 2;
        return (unsigned int)(wr * a.cin * 4 + (((pw & 3) ^ ((wr >> 2) & 3)) * 16));
    };
    // local-table bytes of this lane in a step's slice: waves 0-3 load it for the whole tile ((w', l31') pair p = wid * 64 + lane)
    const unsigned int ltile = (unsigned int)((size_t)tile * T_LTAB * 2);
    const bool l_real = wid < 4;

    struct It { int st, kc; };                                   // position in the (chunk, step) sequence of a pass
    auto next = [&](It &it) { if (++it.st == nst) { it.st = 0; ++it.kc; } };
    // weights / local table of the step at `it` (past the last step, empty slots: out-of-range offsets, zeros arrive)
    auto issue_w = [&](const It &it, int ring) {
#pragma unroll
        for (int j = 0; j < WL; ++j) {
            const int s = it.st * G + (G == 1 ? 0 : wpiece_g(j));
            const int tap = (it.kc < nk && s < T_SLOTS) ? tap_at(s) : -1;
            t_load16_lds((unsigned int)(C::OFF_W + ring * C::WSLOT + (j * T_THREADS + wid * 64) * 16), tap >= 0 ? wpiece_voff(j) : OOB_OFFSET, crsrc,
                         tap >= 0 ? (unsigned int)tap * tap_bytes + (unsigned int)(it.kc * 64) : 0u);
        }
    };
    auto issue_l = [&](const It &it, int ring) {
        if constexpr (G == 4) {
            // one dwordx4 per lane: the 4 slots x 2 rows of pair p; slot group = step
            const bool ok = it.kc < nk && l_real && it.st * 4 < T_SLOTS;
            t_load16_lds((unsigned int)(l_real ? C::OFF_L + ring * C::LSLOT + wid * 1024 : C::OFF_ZERO), ok ? (unsigned int)((wid * 64 + lane) * 16) : OOB_OFFSET,
                         lrsrc, ok ? ltile + (unsigned int)(it.st * 4096) : 0u);
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int s = it.st * G + g;
                const bool ok = it.kc < nk && l_real && s < T_SLOTS;
                t_load4_lds((unsigned int)(l_real ? C::OFF_L + ring * C::LSLOT + g * 1024 + wid * 256 : C::OFF_ZERO),
                            ok ? (unsigned int)((wid * 64 + lane) * 16 + (s & 3) * 4) : OOB_OFFSET, lrsrc, ok ? ltile + (unsigned int)((s >> 2) * 4096) : 0u);
            }
        }
    };
    // halo chunk kc into X buffer `buf`: load i of wave w fills pieces (i * 8 + w) * 64 + lane
    unsigned int xv[C::HREG ? T_PXL : 1];                         // (HREG) byte offset of this thread's piece of load i in the input
    auto x_voff = [&](int i, int g) {
        const int pc = (i * T_WAVES + wid) * 64 + lane, r = pc >> 2;
        return g >= 0 ? (unsigned int)g * row_bytes + (unsigned int)(((pc & 3) ^ ((r >> 2) & 3)) * 16) : OOB_OFFSET;
    };
    auto issue_x = [&](int i, int kc, int buf) {
        unsigned int voff;
        if constexpr (C::HREG) {
            voff = xv[0];
#pragma unroll
            for (int q = 1; q < T_PXL; ++q) voff = i == q ? xv[q] : voff;          // (i is a constant after unrolling, or wave-uniform)
        } else {
            int pc = (i * T_WAVES + wid) * 64 + lane;
            asm volatile("" : "+v"(pc));
            voff = x_voff(i, hal_s[pc >> 2]);
        }
        t_load16_lds((unsigned int)(C::OFF_X + buf * T_XBUF + (i * T_WAVES + wid) * 1024), voff, prsrc, (unsigned int)(kc * 64));
    };
    auto issue_dummy = [&]() { t_load16_lds((unsigned int)C::OFF_ZERO, OOB_OFFSET, prsrc, 0u); };

    // local-table entries of this lane for the G slots of a step (slot g: low half = row l31, high half = row l31 + 32)
    struct LT { unsigned int e[G]; };
    auto table_of = [&](int step) {
        LT t;
        const int ring = step % C::NL;
        if constexpr (G == 4) {
            const uint4 v = *reinterpret_cast<const uint4 *>(smem_raw + C::OFF_L + ring * C::LSLOT + (wid * 32 + l31) * 16);
            t.e[0] = v.x; t.e[1] = v.y; t.e[2] = v.z; t.e[3] = v.w;
        } else {
#pragma unroll
            for (int g = 0; g < G; ++g) t.e[g] = *reinterpret_cast<const unsigned int *>(smem_raw + C::OFF_L + ring * C::LSLOT + g * 1024 + (wid * 32 + l31) * 4);
        }
        return t;
    };

    struct Frag { v4u p_hi[PT], p_lo[PT], c_hi[CT], c_lo[CT]; };
    const unsigned int c_lds = (unsigned int)(C::OFF_W + l31 * 64 + (((2 * kg) ^ ((l31 >> 2) & 3)) * 16));
    auto load_frag = [&](Frag &f, unsigned int lt, int pass_base, int xbuf, int wring, int g) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const unsigned int e = (pt & 1) ? (lt >> 16) : (lt & 0xFFFFu);
            const unsigned int lr = e - (unsigned int)pass_base;
            const unsigned int ax = (unsigned int)(C::OFF_X + xbuf * T_XBUF) + lr * 64u + (((2u * kg) ^ ((lr >> 2) & 3u)) * 16u);
            const unsigned int adr = lr < (unsigned int)(T_HL - 1) ? ax : (unsigned int)(C::OFF_ZERO + kg * 32);
            f.p_hi[pt] = *reinterpret_cast<const v4u *>(smem_raw + adr);
            f.p_lo[pt] = *reinterpret_cast<const v4u *>(smem_raw + (adr ^ 16u));
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const unsigned int adr = c_lds + (unsigned int)(wring * C::WSLOT + g * WBUF + ct * 2048);
            f.c_hi[ct] = *reinterpret_cast<const v4u *>(smem_raw + adr);
            f.c_lo[ct] = *reinterpret_cast<const v4u *>(smem_raw + (adr ^ 16u));
        }
    };
    // MFMAs of one slot for fragment pt, term-major (every accumulator receives lo.hi, hi.lo, hi.hi in that order)
    auto mma_pt = [&](const Frag &f, auto pt_t) {
        constexpr int P = decltype(pt_t)::value;
#pragma unroll
        for (int term = 3 - M::TERMS; term < 3; ++term)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                acc[ct][P] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[P] : f.p_hi[P], acc[ct][P]);
    };
    auto mma_slot = [&](const Frag &f, int s) {                  // wave-uniform skips: fragments none of whose rows has the slot's tap
        if ((fs[0] >> s) & 1u) mma_pt(f, std::integral_constant<int, 0>{});
        if ((fs[1] >> s) & 1u) mma_pt(f, std::integral_constant<int, 1>{});
    };

    for (int pass = 0; pass < npass; ++pass) {
        const int pass_base = pass * (T_HL - 1);
        // halo slice of the pass -> LDS / registers (row HL - 1 is never loaded: the slice has at most HL - 1 entries)
        if constexpr (C::HREG) {
#pragma unroll
            for (int i = 0; i < T_PXL; ++i) {
                const int r = ((i * T_WAVES + wid) * 64 + lane) >> 2;
                xv[i] = x_voff(i, (r < T_HL - 1 && pass_base + r < nh) ? hl_g[pass_base + r] : -1);
            }
        } else {
            for (int r = tid; r < T_HL; r += T_THREADS) hal_s[r] = (r < T_HL - 1 && pass_base + r < nh) ? hl_g[pass_base + r] : -1;
        }
        __syncthreads();
        // ---- prologue: halo chunk 0, weights of steps 0 .. D + 1, local tables of steps 0 .. D + 2, the zero region
        issue_dummy();
#pragma unroll
        for (int i = 0; i < T_PXL; ++i) issue_x(i, 0, 0);
        // (look-ahead of the table: D + 3 steps with G = 1 - it is read one step before its fragments are -, D + 2 otherwise)
        constexpr int DL = G == 1 ? D + 3 : D + 2;
        It iw{0, 0}, il{0, 0}, ic{0, 0};
#pragma unroll
        for (int j = 0; j < D + 2; ++j) { issue_w(iw, j % C::NW); next(iw); }
#pragma unroll
        for (int j = 0; j < DL; ++j) { issue_l(il, j % C::NL); next(il); }
        t_wait_vm<0>();
        __syncthreads();
        LT lt_cur = table_of(0), lt_nxt = table_of(1);
        Frag fa, fb;
        load_frag(fa, lt_cur.e[0], pass_base, 0, 0, 0);
        const int nsteps = nk * nst;
        // Step u = (ic.kc, ic.st); iw is at step u + D + 2, il at step u + DL.  Everything a step needs was put in flight D + 2
        // steps earlier, and nothing between the barrier and the MFMAs waits on memory.  Inside a step the fragments of slot g + 1
        // are read while the MFMAs of slot g run (f0 / f1 alternate; G is 1 - steps come in pairs - or even).
        auto step = [&](int u, Frag &f0, Frag &f1) {
            const int kc = ic.kc, st = ic.st;
            // all loads issued D + 1 or more steps ago have landed: weights of steps <= u + 1, local tables of steps <= u + 2, and -
            // at a chunk's last step - the next chunk's halo (its last loads were issued at step XSTEPS - 1 <= nst - 2 - D)
            t_wait_vm<D * LPS>();
            __syncthreads();
            It in = ic;
            next(in);
            LT lt_nn;
            if constexpr (G == 1) lt_nn = table_of(u + 2); else lt_nxt = table_of(u + 1);
            const bool xreal = st < XSTEPS && kc + 1 < nk;
            // slot 0: fragments of slot 1 (or of the next step's slot 0), MFMAs of slot 0
            if constexpr (G == 1) load_frag(f1, lt_nxt.e[0], pass_base, in.kc & 1, (u + 1) % C::NW, 0);
            else load_frag(f1, lt_cur.e[1], pass_base, kc & 1, u % C::NW, 1);
            if constexpr (G == 1) {                              // one slot per step: its two fragments' MFMAs on either side of the loads
                if ((fs[0] >> st) & 1u) mma_pt(f0, std::integral_constant<int, 0>{});
            } else {
                mma_slot(f0, st * G);
            }
            // this step's share of the look-ahead loads (static count: the ones with nothing to fetch go to the zero region), dealt
            // out between the MFMA groups of the slots: the inline-asm loads are scheduling barriers for the compiler
#pragma unroll
            for (int j = 0; j < PXS; ++j) {
                const int i = (st < XSTEPS ? st : 0) * PXS + j;
                if (xreal && i < T_PXL) issue_x(i, kc + 1, (kc + 1) & 1); else issue_dummy();
                if (G >= 2 && j == PXS / 2 - 1) {               // slot 1 in the middle of the halo loads
                    if constexpr (G == 2) load_frag(f0, lt_nxt.e[0], pass_base, in.kc & 1, (u + 1) % C::NW, 0);
                    else if constexpr (G == 4) load_frag(f0, lt_cur.e[2], pass_base, kc & 1, u % C::NW, 2);
                    if constexpr (G >= 2) mma_slot(f1, st * G + 1);
                }
            }
            if constexpr (G == 4) {
                load_frag(f1, lt_cur.e[3], pass_base, kc & 1, u % C::NW, 3);
                mma_slot(f0, st * G + 2);
            }
            issue_w(iw, (u + D + 2) % C::NW);
            next(iw);
            if constexpr (G == 4) {
                load_frag(f0, lt_nxt.e[0], pass_base, in.kc & 1, (u + 1) % C::NW, 0);
                mma_slot(f1, st * G + 3);
            }
            issue_l(il, (u + DL) % C::NL);
            next(il);
            if constexpr (G == 1) {
                if ((fs[1] >> st) & 1u) mma_pt(f0, std::integral_constant<int, 1>{});
            }
            lt_cur = lt_nxt;
            if constexpr (G == 1) lt_nxt = lt_nn;
            ic = in;
        };
        if constexpr (G == 1) {
            for (int u = 0; u < nsteps; u += 2) {
                step(u, fa, fb);
                if (u + 1 < nsteps) step(u + 1, fb, fa);
            }
        } else {
            for (int u = 0; u < nsteps; ++u) step(u, fa, fb);
        }
        t_wait_vm<0>();                                          // the look-ahead loads of the last steps (dummies) before LDS is reused
        __syncthreads();
    }

    // ---- epilogue through the idle halo buffers (hgemm.h); sorted position -> row of the tile
    unsigned short *const rm_s = reinterpret_cast<unsigned short *>(smem_raw + C::OFF_HAL);          // (the halo list is done with)
    for (int q = tid; q < TR; q += T_THREADS) rm_s[q] = a.rowmap[(size_t)tile * TR + q];
    __syncthreads();
    store_tile_pair16<ET, M>(acc, smem_raw, sc_s, sh_s, 0, a.cout, a.relu != 0, reinterpret_cast<const unsigned char *>(a.residual),
                             reinterpret_cast<unsigned char *>(a.out), wid, 0, lane, wid, [&](int lq) {
                                 const int row = row0 + (int)rm_s[lq];
                                 return row < m ? (size_t)row * a.cout * 4 : ~size_t(0);
                             });
}

template <class M, int CT, int G, int PXS, int D>
static int launch_spconv_t(const SpConvTArgs &a, hipStream_t stream) {
    constexpr int LDS = TCfg<CT, G, D>::LDS;
    static_assert(LDS <= 160 * 1024, "LDS budget of a CU");
    static PerDeviceFlags lds_done;
    if (int rc_ = reserve_lds(reinterpret_cast<const void *>(&k_spconv_t<M, CT, G, PXS, D>), LDS, lds_done, "dz_spconv_tiles_forward")) return rc_;
    int grid = ceil_div(a.cap, T_TR);
    grid = (grid + 31) & ~31;            // whole runs of XRUN tiles on every XCD
    hipLaunchKernelGGL((k_spconv_t<M, CT, G, PXS, D>), dim3(grid), dim3(T_THREADS), LDS, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

template <class M>
static int spconv_t_dispatch(const SpConvTArgs &a, int cout_pad, hipStream_t stream) {
    // 3 x 3 x 3 kernels: 4 / 2 / 1 slots per step for 32 / 64 / 128 output channels (a step's weights are 8 KB in all three), the
    // next halo chunk arrives over 2 / 2 / 4 steps, prefetch distance 1 / 1 / 2 steps (what LDS leaves room for);
    // small kernels (3 x 1 x 1): one slot per step, the whole chunk in the first of its 3 steps, distance 1
    if (a.kvol >= 11) {
        if (cout_pad == 32) return launch_spconv_t<M, 1, 4, 4, 1>(a, stream);
        if (cout_pad == 64) return launch_spconv_t<M, 2, 2, 4, 1>(a, stream);
        if (cout_pad == 128) return launch_spconv_t<M, 4, 1, 2, 2>(a, stream);
    } else if (a.kvol >= 3 && cout_pad == 128) {
        return launch_spconv_t<M, 4, 1, 7, 1>(a, stream);
    }
    set_error("dz_spconv_tiles_forward: unsupported shape cout=%d kvol=%d", a.cout, a.kvol);
    return DZ_ERR_UNSUPPORTED;
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_spconv_tile_rows(void) { return T_TR; }
int dz_spconv_tile_info_words(void) { return T_INFO; }
int dz_spconv_tile_table_entries(void) { return T_LTAB; }

size_t dz_build_tiles_halo_stride(int kvol) { return (size_t)(kvol < 1 ? 1 : kvol) * T_TR; }

int dz_build_tiles(const int *nbr, int kvol, int cap_out, const int *d_m_out, int *halo, int *tinfo, unsigned short *ltab,
                   unsigned short *rowmap, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    constexpr int NT = 1024;
    DZ_CHECK_ARG(nbr && d_m_out && halo && tinfo && ltab && rowmap, "dz_build_tiles: null pointer");
    DZ_CHECK_ARG(kvol >= 1 && kvol <= T_KVOL_MAX && cap_out >= 0, "dz_build_tiles: kvol %d not in [1,27]", kvol);
    if (cap_out == 0) return DZ_OK;
    constexpr int LDS = T_TR * 64 * 4 + 2 * (T_TR * 64 / 32) * 4 + T_TR * 4 + T_TR * 8 + T_TR * 2 + 17 * 4 + 60;
    static PerDeviceFlags lds_done;
    if (int rc_ = reserve_lds(reinterpret_cast<const void *>(&k_build_tiles<NT>), LDS, lds_done, "dz_build_tiles")) return rc_;
    hipLaunchKernelGGL((k_build_tiles<NT>), dim3(ceil_div(cap_out, T_TR)), dim3(NT), LDS, stream, nbr, d_m_out, cap_out, kvol,
                       (int)dz_build_tiles_halo_stride(kvol), halo, tinfo, ltab, rowmap);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_spconv_tiles_forward(const float *in, int in_rows, int cin, const int *halo, const int *tinfo, const unsigned short *ltab,
                            const unsigned short *rowmap, int kvol, int cap_out, const int *d_m_out, const float *w, const float *scale,
                            const float *shift, const float *residual, int relu, float *out, int cout, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(in && halo && tinfo && ltab && rowmap && d_m_out && w && out, "dz_spconv_tiles_forward: null pointer");
    DZ_CHECK_ARG(kvol >= 3 && kvol <= T_KVOL_MAX, "dz_spconv_tiles_forward: kvol %d not in [3,27]", kvol);
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2 || math == DZ_MATH_F16, "dz_spconv_tiles_forward: math %d is not a split mode", math);
    DZ_CHECK_ARG(cout % 8 == 0 && cin % 16 == 0, "dz_spconv_tiles_forward: cin must be a multiple of 16, cout of 8");
    if (cap_out == 0) return DZ_OK;
    const int cout_pad = cout < 32 ? 32 : cout;
    const size_t in_bytes = (size_t)in_rows * cin * sizeof(float);
    const size_t w_bytes = (size_t)kvol * cout_pad * cin * sizeof(float);
    const size_t ltab_bytes = (size_t)ceil_div(cap_out, T_TR) * T_LTAB * 2;
    if (in_rows < 0 || in_bytes >= 0x80000000ull || ltab_bytes >= 0x80000000ull) {
        set_error("dz_spconv_tiles_forward: input of %zu bytes / local table of %zu bytes exceed the 2 GiB buffer-addressing limit", in_bytes, ltab_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    SpConvTArgs a{in, halo, tinfo, ltab, rowmap, d_m_out, w, scale, shift, residual, out, cin, cout, kvol, cap_out, relu,
                  (int)dz_build_tiles_halo_stride(kvol), (unsigned int)in_bytes, (unsigned int)w_bytes, (unsigned int)ltab_bytes};
    if (math == DZ_MATH_F16) return spconv_t_dispatch<MathF16H>(a, cout_pad, stream);
    return math == DZ_MATH_F16X2 ? spconv_t_dispatch<MathF16>(a, cout_pad, stream) : spconv_t_dispatch<MathBF16>(a, cout_pad, stream);
}

const char *dz_spconv_tiles_variant(int cin, int cout) {
    const int cout_pad = cout < 32 ? 32 : cout;
    if (cout_pad == 32) return cin <= 16 ? "k_spconv_t<512x32,c16>" : "k_spconv_t<512x32>";
    if (cout_pad == 64) return "k_spconv_t<512x64>";
    if (cout_pad == 128) return "k_spconv_t<512x128>";
    return "none";
}

}  // extern "C"
#else   // default build: the tile engine (measured slower than the gather / x-run engines) is not compiled; its entry points stay
        // exported so that include/detzero_hip.h == the library's symbols in every build, and say how to get the engine
#include "common.h"
using namespace dz;
extern "C" {
static int no_tiles(const char *fn) {
    set_error("%s: the tile-resident sparse engine is an experimental build option - rebuild with DZ_BUILD_EXPERIMENTAL=1 (python -m detzero_amd.build --force)", fn);
    return DZ_ERR_UNSUPPORTED;
}
int dz_spconv_tile_rows(void) { return 0; }
int dz_spconv_tile_info_words(void) { return 0; }
int dz_spconv_tile_table_entries(void) { return 0; }
size_t dz_build_tiles_halo_stride(int) { return 0; }
int dz_build_tiles(const int *, int, int, const int *, int *, int *, unsigned short *, unsigned short *, void *) { return no_tiles("dz_build_tiles"); }
int dz_spconv_tiles_forward(const float *, int, int, const int *, const int *, const unsigned short *, const unsigned short *, int, int, const int *,
                            const float *, const float *, const float *, const float *, int, float *, int, int, void *) {
    return no_tiles("dz_spconv_tiles_forward");
}
const char *dz_spconv_tiles_variant(int, int) { return "none (built without DZ_BUILD_EXPERIMENTAL)"; }
}  // extern "C"
#endif
