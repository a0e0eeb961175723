// Sparse-tensor index machinery on bitmaps (gfx950).
//
// Replaces spconv's hash-table "indice" generation for the call sites in
// detection/detzero_det/models/centerpoint_modules/backbone3d.py:243-280,302-307.
// MI355X-first design: instead of a hash table + sort, every level is one bit per grid cell
// (level 1 of the Waymo config: 41*1504*1504 bits = 11.6 MB, resident in the 256 MB Infinity
// Cache) plus an exclusive popcount prefix per 32-bit word.  A set-bit scan gives every active
// site its rank in ascending linear-key order, so feature rows are stored spatially sorted
// (x-neighbours are adjacent rows -> coalesced gathers) and a neighbour lookup is two loads +
// one popcount, no probing, no atomics on the read side.
#include <algorithm>
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace dz {

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char *last_error() { return g_err; }

__global__ void k_fill_u32(uint32_t *__restrict__ p, uint32_t v, size_t n) {
    const size_t n4 = n / 4;
    uint4 *p4 = reinterpret_cast<uint4 *>(p);
    const uint4 v4 = make_uint4(v, v, v, v);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p4[i] = v4;
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

int fill_u32(void *ptr, uint32_t value, size_t nwords, hipStream_t stream) {
    if (nwords == 0) return DZ_OK;
    if (((uintptr_t)ptr & 15u) != 0) { set_error("fill_u32: pointer not 16-byte aligned"); return DZ_ERR_INVALID; }
    hipLaunchKernelGGL(k_fill_u32, dim3(stream_grid((long)(nwords / 4 + 1), 256)), dim3(256), 0, stream,
                       reinterpret_cast<uint32_t *>(ptr), value, nwords);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// ------------------------------------------------------------------------------------------
// bitmap scan
// ------------------------------------------------------------------------------------------
// WPT = bitmap words per thread (a workgroup covers 256 * WPT words).  Large, mostly empty bitmaps use WPT = 4
// (fewer partial sums); small / dense ones WPT = 1: the coordinate emission walks the set bits of a thread's words
// serially, so a dense level needs many threads rather than long per-thread chains.
// line_flags (optional, one byte per 32 words = 128-byte line): lines whose flag is 0 hold no bit and are not read - a thread's
// WPT <= 4 aligned words lie in one line.
template <int WPT>
__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t *__restrict__ bitmap, size_t nwords,
                                                     uint32_t *__restrict__ partial, const unsigned char *__restrict__ line_flags) {
    __shared__ uint32_t lds[4];
    const size_t base = (size_t)blockIdx.x * (256 * WPT) + (size_t)threadIdx.x * WPT;
    uint32_t s = 0;
    if (!line_flags || (base < nwords && line_flags[base >> 5])) {
#pragma unroll
        for (int j = 0; j < WPT; ++j)
            if (base + j < nwords) s += __popc(bitmap[base + j]);
    }
    uint32_t total;
    block_excl_scan_256(s, lds, total);
    if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

// single block of 1024 threads: exclusive scan of `partial` in place, total -> d_total.  8192 entries per trip: loaded with
// coalesced accesses (entry j * 1024 + t), transposed through LDS so that a thread scans 8 CONSECUTIVE entries (a thread-
// contiguous global access pattern keeps one CU's address pipe busy with 64 cache lines per load: 46 us for the 45k partial sums
// of a batch of 16 level-1 grids), one block scan of the thread sums, and back; the next trip's loads are issued before the
// scan of the current one.
__global__ __launch_bounds__(1024) void k_scan_partials(uint32_t *__restrict__ partial, int nblocks, int *__restrict__ d_total) {
    constexpr int PPT = 8, TRIP = 1024 * PPT;
    __shared__ uint32_t buf[TRIP + TRIP / 32];          // (one pad word per 32: the 8-word thread stride stays conflict-light)
    __shared__ uint32_t wsum[16];
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    auto slot = [](int i) { return i + (i >> 5); };
    uint32_t carry = 0, nxt[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) nxt[j] = (j * 1024 + t < nblocks) ? partial[j * 1024 + t] : 0u;
    for (int base = 0; base < nblocks; base += TRIP) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) buf[slot(j * 1024 + t)] = nxt[j];
        if (base + TRIP < nblocks) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) { const int i = base + TRIP + j * 1024 + t; nxt[j] = i < nblocks ? partial[i] : 0u; }
        }
        __syncthreads();
        uint32_t v[PPT], s = 0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) { v[j] = buf[slot(t * PPT + j)]; s += v[j]; }
        uint32_t incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t u = __shfl_up(incl, off, 64);
            if (lane >= off) incl += u;
        }
        if (lane == 63) wsum[wid] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const uint32_t x = wsum[w]; if (w < wid) woff += x; tot += x; }
        uint32_t run = carry + woff + incl - s;
#pragma unroll
        for (int j = 0; j < PPT; ++j) { buf[slot(t * PPT + j)] = run; run += v[j]; }
        carry += tot;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PPT; ++j) { const int i = base + j * 1024 + t; if (i < nblocks) partial[i] = buf[slot(j * 1024 + t)]; }
        __syncthreads();
    }
    if (t == 0) *d_total = (int)carry;
}

// exact n / d, n % d for 32-bit n through one fp64 multiply + one fix-up (d > 0)
struct FastDiv { uint32_t d; double inv; };
__device__ __forceinline__ uint32_t fast_divmod(uint32_t n, const FastDiv f, uint32_t &rem) {
    uint32_t q = (uint32_t)__double2uint_rz((double)n * f.inv);
    uint32_t r = n - q * f.d;
    if ((int32_t)r < 0) { --q; r += f.d; }
    else if (r >= f.d) { ++q; r -= f.d; }
    rem = r;
    return q;
}
struct ScanDecode { FastDiv d0, d1, d2; };

// NZ: prefix[] is written only for words that have a bit set - the rank queries (bitmap_find / bitmap_rank) test the bit before
// they read the prefix, and a level-1 bitmap is > 97 % zero words (181 MB of prefix writes per 16 frames otherwise)
template <int MODE, int WPT, bool NZ = false>
__global__ __launch_bounds__(256) void k_scan_down(const uint32_t *__restrict__ bitmap, size_t nwords,
                                                   const uint32_t *__restrict__ partial,
                                                   uint32_t *__restrict__ prefix, ScanDecode dec,
                                                   int *__restrict__ coords_out, int cap_out,
                                                   const unsigned char *__restrict__ line_flags, const int *__restrict__ d_total) {
    constexpr int CHUNK = 256 * WPT;
    if (NZ) {
        // a chunk without a bit (half of the chunks of a level-1 grid) has nothing to write: its count is the difference of two
        // scanned partial sums
        const uint32_t p0 = partial[blockIdx.x], p1 = blockIdx.x + 1 < gridDim.x ? partial[blockIdx.x + 1] : (uint32_t)*d_total;
        if (p0 == p1) return;
    }
    __shared__ uint32_t lds[4];
    __shared__ __attribute__((aligned(16))) uint32_t w_s[CHUNK];
    __shared__ __attribute__((aligned(16))) uint32_t p_s[CHUNK];
    const size_t base = (size_t)blockIdx.x * CHUNK + (size_t)threadIdx.x * WPT;
    uint32_t wv[WPT];
    const bool live = !line_flags || (base < nwords && line_flags[base >> 5]);         // (an unflagged line holds no bit: not read)
#pragma unroll
    for (int j = 0; j < WPT; ++j) wv[j] = (live && base + j < nwords) ? bitmap[base + j] : 0u;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < WPT; ++j) s += __popc(wv[j]);
    uint32_t total;
    uint32_t run = block_excl_scan_256(s, lds, total) + partial[blockIdx.x];
    uint32_t pv[WPT];
#pragma unroll
    for (int j = 0; j < WPT; ++j) { pv[j] = run; run += __popc(wv[j]); }
#pragma unroll
    for (int j = 0; j < WPT; ++j)
        if (base + j < nwords && (!NZ || wv[j])) prefix[base + j] = pv[j];
    if (MODE < 0 || total == 0) return;
    // coordinate emission: words are re-distributed round-robin over the threads (dense clusters of
    // set bits are consecutive words; consecutive words per thread serialised them on one lane)
#pragma unroll
    for (int j = 0; j < WPT; ++j) { w_s[threadIdx.x * WPT + j] = wv[j]; p_s[threadIdx.x * WPT + j] = pv[j]; }
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < WPT; ++j) {
        const int idx = threadIdx.x + 256 * j;
        uint32_t word = w_s[idx];
        if (!word) continue;
        uint32_t r = p_s[idx];
        const uint32_t key0 = (uint32_t)(((size_t)blockIdx.x * CHUNK + idx) << 5);
        if (MODE == 2) {
            // brick keys (LevelGeom layout 1; dec = D, NBY, NBX): the 32 cells of a word share (b, y/8, x/8, z) and y%8 / 4
            uint32_t z, by, bx;
            const uint32_t brick = fast_divmod(key0 >> 6, dec.d0, z);
            const uint32_t t = fast_divmod(brick, dec.d2, bx);
            const uint32_t bb = fast_divmod(t, dec.d1, by);
            const int y0 = (int)(by * 8u + ((key0 >> 3) & 7u)), x0 = (int)(bx * 8u);
            while (word) {
                const int bit = __ffs((int)word) - 1;
                word &= word - 1;
                if ((int)r < cap_out) reinterpret_cast<int4 *>(coords_out)[r] = make_int4((int)bb, (int)z, y0 + (bit >> 3), x0 + (bit & 7));
                ++r;
            }
            continue;
        }
        // decode the word's first cell once (3 divisions), then walk the bits with carries only
        uint32_t c2, c1, c0;
        const uint32_t t1 = fast_divmod(key0, dec.d2, c2);
        const uint32_t t0 = fast_divmod(t1, dec.d1, c1);
        uint32_t bb = fast_divmod(t0, dec.d0, c0);
        int last = 0;
        while (word) {
            const int bit = __ffs((int)word) - 1;
            word &= word - 1;
            c2 += (uint32_t)(bit - last);
            last = bit;
            while (c2 >= dec.d2.d) {
                c2 -= dec.d2.d;
                if (++c1 == dec.d1.d) { c1 = 0; if (++c0 == dec.d0.d) { c0 = 0; ++bb; } }
            }
            if ((int)r < cap_out) {
                int4 o;
                if (MODE == 0) o = make_int4((int)bb, (int)c0, (int)c1, (int)c2);   // [b,z,y,x]
                else o = make_int4((int)bb, (int)c2, (int)c1, (int)c0);             // key (x,y,z) -> [b,z,y,x]
                reinterpret_cast<int4 *>(coords_out)[r] = o;
            }
            ++r;
        }
    }
}

size_t bitmap_scan_workspace_bytes(size_t nwords) {
    const size_t nblocks = (nwords + 255) / 256;          // the finest chunking (WPT = 1)
    return align_up(nblocks * sizeof(uint32_t), 256);
}

template <int WPT>
static void launch_scan(const uint32_t *bitmap, size_t nwords, uint32_t *prefix, int *d_total, int mode, const ScanDecode &dec,
                        int *coords_out, int cap_out, uint32_t *partial, hipStream_t stream, bool nonzero_only,
                        const unsigned char *lf) {
    const int nblocks = (int)((nwords + 256 * WPT - 1) / (256 * WPT));
    hipLaunchKernelGGL(k_scan_reduce<WPT>, dim3(nblocks), dim3(256), 0, stream, bitmap, nwords, partial, lf);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, stream, partial, nblocks, d_total);
    if (mode == 0 && nonzero_only)
        hipLaunchKernelGGL((k_scan_down<0, WPT, true>), dim3(nblocks), dim3(256), 0, stream, bitmap, nwords, partial, prefix, dec,
                           coords_out, cap_out, lf, d_total);
    else if (mode == 0)
        hipLaunchKernelGGL((k_scan_down<0, WPT>), dim3(nblocks), dim3(256), 0, stream, bitmap, nwords, partial, prefix, dec,
                           coords_out, cap_out, lf, d_total);
    else if (mode == 1)
        hipLaunchKernelGGL((k_scan_down<1, WPT>), dim3(nblocks), dim3(256), 0, stream, bitmap, nwords, partial, prefix, dec,
                           coords_out, cap_out, lf, d_total);
    else if (mode == 2 && nonzero_only)
        hipLaunchKernelGGL((k_scan_down<2, WPT, true>), dim3(nblocks), dim3(256), 0, stream, bitmap, nwords, partial, prefix, dec,
                           coords_out, cap_out, lf, d_total);
    else if (mode == 2)
        hipLaunchKernelGGL((k_scan_down<2, WPT>), dim3(nblocks), dim3(256), 0, stream, bitmap, nwords, partial, prefix, dec,
                           coords_out, cap_out, lf, d_total);
    else
        hipLaunchKernelGGL((k_scan_down<-1, WPT>), dim3(nblocks), dim3(256), 0, stream, bitmap, nwords, partial, prefix, dec,
                           coords_out, cap_out, lf, d_total);
}

int bitmap_scan(const uint32_t *bitmap, size_t nwords, uint32_t *prefix, int *d_total, int mode,
                ScanDims dims, int *coords_out, int cap_out, void *ws, size_t ws_bytes,
                hipStream_t stream, bool nonzero_only, const unsigned char *line_flags) {
    if (ws_bytes < bitmap_scan_workspace_bytes(nwords)) {
        set_error("bitmap_scan: workspace %zu < %zu", ws_bytes, bitmap_scan_workspace_bytes(nwords));
        return DZ_ERR_WORKSPACE;
    }
    uint32_t *partial = reinterpret_cast<uint32_t *>(ws);
    ScanDecode dec;
    dec.d0 = FastDiv{(uint32_t)(dims.d0 > 0 ? dims.d0 : 1), 1.0 / (double)(dims.d0 > 0 ? dims.d0 : 1)};
    dec.d1 = FastDiv{(uint32_t)(dims.d1 > 0 ? dims.d1 : 1), 1.0 / (double)(dims.d1 > 0 ? dims.d1 : 1)};
    dec.d2 = FastDiv{(uint32_t)(dims.d2 > 0 ? dims.d2 : 1), 1.0 / (double)(dims.d2 > 0 ? dims.d2 : 1)};
    if (nwords <= ((size_t)1 << 21))
        launch_scan<1>(bitmap, nwords, prefix, d_total, mode, dec, coords_out, cap_out, partial, stream, nonzero_only, line_flags);
    else            // (16 words per thread - 11k partial sums for a batch of level-1 grids - was measured: the reduce pass 34 -> 113 us,
                    // the emit pass 76 -> 142 us: a thread's 64 contiguous bytes are no longer one coalesced vector load per wavefront)
        launch_scan<4>(bitmap, nwords, prefix, d_total, mode, dec, coords_out, cap_out, partial, stream, nonzero_only, line_flags);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// ------------------------------------------------------------------------------------------
// level construction
// ------------------------------------------------------------------------------------------
__global__ void k_set_bits_from_coords(const int *__restrict__ coords, const int *__restrict__ d_n, int n_cap,
                                       LevelGeom lg, uint32_t *__restrict__ bitmap) {
    const int n = d_n ? min(*d_n, n_cap) : n_cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[i];
        if (!lg.inside(c.x, c.y, c.z, c.w)) continue;
        const uint32_t key = lg.key(c.x, c.y, c.z, c.w);
        atomicOr(&bitmap[key >> 5], 1u << (key & 31u));
    }
}

__global__ void k_rank_of_coords(const int *__restrict__ coords, const int *__restrict__ d_n, int n_cap, LevelGeom lg,
                                 const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ prefix, int *__restrict__ rank) {
    const int n = d_n ? min(*d_n, n_cap) : n_cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_cap; i += gridDim.x * blockDim.x) {
        int r = -1;
        if (i < n) {
            const int4 c = reinterpret_cast<const int4 *>(coords)[i];
            if (lg.inside(c.x, c.y, c.z, c.w)) r = bitmap_rank(bitmap, prefix, lg.key(c.x, c.y, c.z, c.w));
        }
        rank[i] = r;
    }
}

struct ConvGeom {
    int k[3], s[3], p[3];
    int od, oh, ow;  // output dims
};

// every active input marks the outputs whose window contains it
// per dimension: the (at most KMAX) output coordinates fed by input coordinate c: w = (c + p - t) / s for the taps
// t congruent to (c + p) mod s
__device__ __forceinline__ int outs_of(int c, int k, int s, int p, int od, int (&w)[3]) {
    int n = 0;
    const int cp = c + p;
    for (int t = cp % s; t < k && t <= cp; t += s) {
        const int o = (cp - t) / s;
        if (o < od && n < 3) w[n++] = o;
    }
    return n;
}

__global__ void k_mark_outputs(const int *__restrict__ coords_in, const int *__restrict__ d_m_in, int cap_in,
                               ConvGeom g, LevelGeom lo, uint32_t *__restrict__ bitmap_out) {
    const int m = min(*d_m_in, cap_in);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int4 c = reinterpret_cast<const int4 *>(coords_in)[i];
        int wz[3], wy[3], wx[3];
        const int nz = outs_of(c.y, g.k[0], g.s[0], g.p[0], g.od, wz);
        const int ny = outs_of(c.z, g.k[1], g.s[1], g.p[1], g.oh, wy);
        const int nx = outs_of(c.w, g.k[2], g.s[2], g.p[2], g.ow, wx);
        for (int a = 0; a < nz; ++a)
            for (int b = 0; b < ny; ++b) {
                // the x candidates of one row are adjacent cells: usually one bitmap word (in both layouts)
                uint32_t word_idx = 0xFFFFFFFFu, bits = 0u;
                for (int e = 0; e < nx; ++e) {
                    const uint32_t key = lo.key(c.x, wz[a], wy[b], wx[e]);
                    if ((key >> 5) != word_idx) {
                        // ~8 inputs feed every output: test before the atomic (a stale read only costs a redundant atomicOr)
                        if (bits && (__builtin_nontemporal_load(&bitmap_out[word_idx]) & bits) != bits) atomicOr(&bitmap_out[word_idx], bits);
                        word_idx = key >> 5;
                        bits = 0u;
                    }
                    bits |= 1u << (key & 31u);
                }
                if (bits && (__builtin_nontemporal_load(&bitmap_out[word_idx]) & bits) != bits) atomicOr(&bitmap_out[word_idx], bits);
            }
    }
}

// k_mark_outputs for windows that feed at most two outputs per dimension (every strided stage of the backbone: k 3 / s 2, and
// conv_out's (3,1,1) / (2,1,1)).  Atomics on this part resolve at the memory side (the bitmap is shared by the 8 XCDs' L2s) and
// bound the kernel; the inputs arrive in key order, so the 64 inputs of a wavefront mostly feed the same few output words: a
// lane keeps one (word, bits) per (z, y) output row, the lanes of a run of equal words OR their bits together in registers
// (segmented scan over the wavefront) and only the run's last lane tests the word and issues the atomic.
__global__ __launch_bounds__(256) void k_mark_outputs_w(const int *__restrict__ coords_in, const int *__restrict__ d_m_in, int cap_in,
                                                        ConvGeom g, LevelGeom lo, uint32_t *__restrict__ bitmap_out) {
    const int m = min(*d_m_in, cap_in);
    const int m_pad = (m + 63) & ~63;
    const int lane = threadIdx.x & 63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m_pad; i += gridDim.x * blockDim.x) {
        uint32_t widx[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, wbits[4] = {0u, 0u, 0u, 0u};
        if (i < m) {
            const int4 c = reinterpret_cast<const int4 *>(coords_in)[i];
            int wz[3], wy[3], wx[3];
            const int nz = outs_of(c.y, g.k[0], g.s[0], g.p[0], g.od, wz);
            const int ny = outs_of(c.z, g.k[1], g.s[1], g.p[1], g.oh, wy);
            const int nx = outs_of(c.w, g.k[2], g.s[2], g.p[2], g.ow, wx);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if (a >= nz || b >= ny || nx < 1) continue;
                    const uint32_t k0 = lo.key(c.x, wz[a], wy[b], wx[0]);
                    uint32_t w = k0 >> 5, bits = 1u << (k0 & 31u);
                    if (nx > 1) {
                        const uint32_t k1 = lo.key(c.x, wz[a], wy[b], wx[1]);
                        if ((k1 >> 5) == w) bits |= 1u << (k1 & 31u);
                        else atomicOr(&bitmap_out[k1 >> 5], 1u << (k1 & 31u));      // the second x candidate opens another word (1 input in 32)
                    }
                    widx[a * 2 + b] = w;
                    wbits[a * 2 + b] = bits;
                }
        }
        bool last[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            const uint32_t w = widx[sl];
            uint32_t bits = wbits[sl];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t wp = (uint32_t)__shfl_up((int)w, d, 64), bp = (uint32_t)__shfl_up((int)bits, d, 64);
                if (lane >= d && wp == w) bits |= bp;
            }
            const uint32_t wn = (uint32_t)__shfl_down((int)w, 1, 64);
            last[sl] = bits && (lane == 63 || wn != w);
            wbits[sl] = bits;
        }
        // ~8 inputs feed every output: test before the atomic (a stale read only costs a redundant atomicOr); the four tests of a lane
        // are loaded together
        uint32_t seen[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) seen[sl] = last[sl] ? __builtin_nontemporal_load(&bitmap_out[widx[sl]]) : 0xFFFFFFFFu;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)
            if (last[sl] && (seen[sl] & wbits[sl]) != wbits[sl]) atomicOr(&bitmap_out[widx[sl]], wbits[sl]);
    }
}

// nbr[t*cap + o] for every output row o and tap t.
//
// k_build_neighbors_rows (linear key layout, x window of 3 around an always-valid centre cell, or of 1): ONE thread per output
// row.  The x taps of a (tz, ty) row are consecutive cells of one bitmap row, and ranks follow the key order, so the bitmap and
// prefix words around the centre cell answer all of them: with r = number of active cells below the centre cell, the left
// neighbour (when its bit is set) is row r - 1, the centre r, the right one r + centre_bit; when the centre sits at bit 0 / 31
// the outer cell is bit 31 / 0 of the adjacent word (rank from that word's own prefix - a prefix entry is only valid where the
// word holds a bit, common.h).  A thread fetches the THREE words around the centre with one 12-byte load each from the bitmap
// and the prefix array: 2 * K0 * K1 independent, unconditional loads, the next z slab's issued before the current slab's taps are
// computed, no divergent second lookups (with 64 lanes nearly every wavefront has a lane at bit 0 or 31).
// A lane's tap mask is OR-reduced over its 32-row group in registers and stored - no atomics, no zero-fill of the mask words
// (rows past the count store 0).
#ifndef DZ_NBR_DIAG
#define DZ_NBR_DIAG 0       // development builds (tools/gpu_nbr_diag.sh): 1 no bitmap / prefix loads, 2 no table stores
#endif
typedef unsigned int nbr_u3 __attribute__((ext_vector_type(3)));

// the three x taps of one (tz, ty) row from the bitmap / prefix words around the centre cell.  EDGE = false: the centre word is
// w.y (words w.x / w.z before / after it); EDGE = true (the centre word is the first or last of the array): it is component ci.
template <bool EDGE>
__device__ __forceinline__ void nbr_row3(nbr_u3 w, nbr_u3 p, int ci, uint32_t bc, bool ok, bool lv, bool rv, int &vl, int &vc, int &vr) {
    uint32_t wc_ = w.y, pc_ = p.y, lw = w.x, lp = p.x, rw = w.z, rp = p.z;
    if (EDGE) {
        wc_ = ci == 1 ? w.y : (ci == 0 ? w.x : w.z);
        pc_ = ci == 1 ? p.y : (ci == 0 ? p.x : p.z);
        lw = ci == 1 ? w.x : (ci == 2 ? w.y : 0u); lp = ci == 1 ? p.x : p.y;
        rw = ci == 1 ? w.z : (ci == 0 ? w.y : 0u); rp = ci == 1 ? p.z : p.y;
    }
    if (!ok) wc_ = lw = rw = 0u;
    const int rank = (int)(pc_ + __popc(wc_ & ((1u << bc) - 1u)));         // active cells below the centre cell
    const uint32_t cbit = (wc_ >> bc) & 1u;
    vc = cbit ? rank : -1;
    // left / right cell: bit bc -+ 1 of the centre word, or bit 31 / 0 of the adjacent word (ranked from that word's own prefix)
    const bool lbit = bc != 0u ? ((wc_ >> (bc - 1u)) & 1u) != 0u : (lw >> 31) != 0u;
    const bool rbit = bc != 31u ? ((wc_ >> (bc + 1u)) & 1u) != 0u : (rw & 1u) != 0u;
    const int lrank = bc != 0u ? rank - 1 : (int)(lp + __popc(lw)) - 1;
    const int rrank = bc != 31u ? rank + (int)cbit : (int)rp;
    vl = lbit && lv ? lrank : -1;
    vr = rbit && rv ? rrank : -1;
}

template <int K0, int K1, int KW, bool XCD, bool PACK = false>
__global__ __launch_bounds__(256) void k_build_neighbors_rows(const int *__restrict__ coords_out, const int *__restrict__ d_m_out,
                                                              int cap_out, const uint32_t *__restrict__ bitmap_in,
                                                              const uint32_t *__restrict__ prefix_in, LevelGeom li, ConvGeom g,
                                                              int *__restrict__ nbr, uint32_t *__restrict__ tile_masks, int mask_rows,
                                                              uint32_t last_base, uint32_t index_bytes, uint32_t nbr_bytes) {
    // table, bitmap and prefix array through buffer descriptors: a scalar base (+ the tap's scalar row offset) and ONE 32-bit
    // per-lane byte offset, instead of a 64-bit address per access (the table is written through K0 * K1 * KW row pointers)
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(nbr, 0, nbr_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bmr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(bitmap_in), 0, index_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t pfr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(prefix_in), 0, index_bytes, 0x00020000);
    const uint32_t row_bytes = (uint32_t)cap_out * 4u;
    const int m = min(*d_m_out, cap_out);
    // every 32-row group of the mask buffer is visited (mask_rows = 32 x its words): whole wavefronts run the same trip count
    const int total = tile_masks ? mask_rows : ((m + 63) & ~63);
    // workgroups are dealt round-robin to the 8 XCDs (each with its own L2): XCD x takes the x-th CONTIGUOUS eighth of the live
    // rows, so that the bitmap / prefix lines of a row's y and z neighbours (hundreds to thousands of rows away in key order)
    // are fetched into the L2 that fetched them for the row itself, not into all eight (gridDim.x is a multiple of 8)
    const int live = min(total, (m + 63) & ~63);
    const int span = XCD ? ((live + 511) / 512) * 64 : live;
    const int lo = XCD ? (int)(blockIdx.x & 7u) * span : 0, hi = XCD ? min(live, lo + span) : live;
    const int first = lo + (XCD ? (int)(blockIdx.x >> 3) : (int)blockIdx.x) * (int)blockDim.x + (int)threadIdx.x;
    const int step = (XCD ? (int)(gridDim.x >> 3) : (int)gridDim.x) * (int)blockDim.x;
    if (tile_masks)             // mask words past the live rows: zero
        for (int o = live + blockIdx.x * blockDim.x + threadIdx.x; o < total; o += gridDim.x * blockDim.x)
            if ((threadIdx.x & 31) == 0) tile_masks[o >> 5] = 0u;
    for (int o = first; o < hi; o += step) {
        uint32_t bits = 0u;
        int4 c = make_int4(0, 0, 0, 0);
        if (o < m) c = reinterpret_cast<const int4 *>(coords_out)[o];
        const uint32_t voff = (uint32_t)o * 4u;
        const int uxc = c.w * g.s[2] - g.p[2] + (KW == 3 ? 1 : 0);       // centre cell: inside the grid (checked by the launcher)
        const bool lv = uxc > 0, rv = uxc + 1 < li.w;
        const int uz0 = c.y * g.s[0] - g.p[0], uy0 = c.z * g.s[1] - g.p[1];
        const uint32_t key00 = (uint32_t)(((c.x * li.d + uz0) * li.h + uy0) * li.w + uxc);
        // one z slab (K1 rows) at a time: the loads of slab tz + 1 are issued before the taps of slab tz are computed and stored
        struct Slab { nbr_u3 word[K1], pref[K1]; uint32_t kc[K1], base[K1]; bool ok[K1], edge; } sl[K0];
        auto fetch = [&](int tz, Slab &q) {
            q.edge = false;
#pragma unroll
            for (int ty = 0; ty < K1; ++ty) {
                q.ok[ty] = o < m && (unsigned)(uz0 + tz) < (unsigned)li.d && (unsigned)(uy0 + ty) < (unsigned)li.h;
                q.kc[ty] = q.ok[ty] ? key00 + (uint32_t)((tz * li.h + ty) * li.w) : 32u;
                const uint32_t wc = q.kc[ty] >> 5;
                q.edge |= wc - 1u > last_base;                                   // first or last word of the arrays
                q.base[ty] = min(wc > 0u ? wc - 1u : 0u, last_base);             // words base .. base + 2 (inside the arrays)
#if DZ_NBR_DIAG & 1
                q.word[ty] = nbr_u3{q.kc[ty], q.kc[ty] * 3u, q.kc[ty] * 5u}; q.pref[ty] = nbr_u3{q.kc[ty] >> 3, q.kc[ty] >> 4, q.kc[ty] >> 5};
#else
                if (KW == 3) {
                    q.word[ty] = __builtin_amdgcn_raw_buffer_load_b96(bmr, q.base[ty] * 4u, 0, 0);
                    q.pref[ty] = __builtin_amdgcn_raw_buffer_load_b96(pfr, q.base[ty] * 4u, 0, 0);   // (unwritten where a word is empty: not used)
                } else {
                    q.word[ty].y = __builtin_amdgcn_raw_buffer_load_b32(bmr, wc * 4u, 0, 0);
                    q.pref[ty].y = __builtin_amdgcn_raw_buffer_load_b32(pfr, wc * 4u, 0, 0);
                }
#endif
            }
        };
        fetch(0, sl[0]);
#pragma unroll
        for (int tz = 0; tz < K0; ++tz) {
            if (tz + 1 < K0) fetch(tz + 1, sl[tz + 1]);
            const Slab &q = sl[tz];
            const bool any_edge = KW == 3 && __any(q.edge);
#pragma unroll
            for (int ty = 0; ty < K1; ++ty) {
                const int r = tz * K1 + ty;
                const uint32_t bc = q.kc[ty] & 31u;
                int vl, vc, vr;
                if (KW == 1) {
                    const uint32_t w = q.ok[ty] ? q.word[ty].y : 0u;
                    vc = (w >> bc) & 1u ? (int)(q.pref[ty].y + __popc(w & ((1u << bc) - 1u))) : -1;
#if !(DZ_NBR_DIAG & 2)
                    if (o < m) __builtin_amdgcn_raw_buffer_store_b32(vc, tab, voff, (uint32_t)r * row_bytes, 0);
#endif
                    if (vc >= 0) bits |= 1u << r;
                    continue;
                }
                if (any_edge) nbr_row3<true>(q.word[ty], q.pref[ty], (int)((q.kc[ty] >> 5) - q.base[ty]), bc, q.ok[ty], lv, rv, vl, vc, vr);
                else nbr_row3<false>(q.word[ty], q.pref[ty], 1, bc, q.ok[ty], lv, rv, vl, vc, vr);
                const int tap = r * 3;
#if DZ_NBR_DIAG & 2
                bits ^= (uint32_t)(vl + vc + vr) & 0x8000000u;
#else
                if (PACK) {
                    // packed row entry (DZ_NBR_PACKED, include/detzero_hip.h): the three x taps are consecutive ranks, so one word
                    // holds r = active cells below the centre (bits 0..28) and the presence of left / centre / right (bits 29..31):
                    // left = r - 1, centre = r, right = r + centre.  r from whichever neighbour exists (a prefix word is only
                    // valid where its bitmap word holds a bit)
                    const uint32_t rr = vl >= 0 ? (uint32_t)vl + 1u : (vc >= 0 ? (uint32_t)vc : (vr >= 0 ? (uint32_t)vr : 0u));
                    const uint32_t e = rr | (vl >= 0 ? 1u << 29 : 0u) | (vc >= 0 ? 1u << 30 : 0u) | (vr >= 0 ? 1u << 31 : 0u);
                    if (o < m) __builtin_amdgcn_raw_buffer_store_b32((int)e, tab, voff, (uint32_t)r * row_bytes, 0);
                } else if (o < m) {
                    __builtin_amdgcn_raw_buffer_store_b32(vl, tab, voff, (uint32_t)tap * row_bytes, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(vc, tab, voff, (uint32_t)(tap + 1) * row_bytes, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(vr, tab, voff, (uint32_t)(tap + 2) * row_bytes, 0);
                }
#endif
                bits |= ((vl >= 0 ? 1u : 0u) | (vc >= 0 ? 2u : 0u) | (vr >= 0 ? 4u : 0u)) << tap;
            }
        }
        if (tile_masks) {
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, d, 64);
            if ((threadIdx.x & 31) == 0) tile_masks[o >> 5] = bits;
        }
    }
}

// k_build_neighbors_xwin: the PACKED submanifold table (k_build_neighbors_rows<3, 3, 3, true, true>) AND what the x-run convolution
// needs on top of it (csrc/sparse_conv_x.hip: dz_spconv_x_windows) in ONE launch - the windows of every unit of `tile_rows` output
// rows and, optionally, the unit's rows in tap-set order (perm + the re-ordered table).  Round 4 ran k_xwin as a second kernel that
// read the whole table back (105 us per level and step at 32 frames); here a workgroup owns one 256-row block (one or two units),
// every thread keeps the nine words of its row in registers, and the window bounds / the counting sort run on them.
// Block order: XCD x takes the x-th contiguous eighth of the live blocks (the bitmap / prefix lines of a row's y and z neighbours
// stay in one L2, as in k_build_neighbors_rows).
__global__ __launch_bounds__(256) void k_build_neighbors_xwin(const int *__restrict__ coords_out, const int *__restrict__ d_m_out, int cap_out,
                                                              const uint32_t *__restrict__ bitmap_in, const uint32_t *__restrict__ prefix_in,
                                                              LevelGeom li, int *__restrict__ nbr, uint32_t *__restrict__ tile_masks, int mask_rows,
                                                              uint32_t last_base, uint32_t index_bytes, uint32_t nbr_bytes, int tile_rows,
                                                              int *__restrict__ win, int *__restrict__ nbr_sorted, int *__restrict__ perm) {
    __shared__ int lo_s[2][3], hi_s[2][3];
    __shared__ __attribute__((aligned(16))) unsigned int key_s[256];
    const __amdgpu_buffer_rsrc_t tab = __builtin_amdgcn_make_buffer_rsrc(nbr, 0, nbr_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bmr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(bitmap_in), 0, index_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t pfr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(prefix_in), 0, index_bytes, 0x00020000);
    const uint32_t row_bytes = (uint32_t)cap_out * 4u;
    const int m = min(*d_m_out, cap_out);
    const int tid = threadIdx.x;
    const int nunits = (cap_out + tile_rows - 1) / tile_rows;
    if (blockIdx.x == 0 && tid < 16) win[(size_t)nunits * 6 + tid] = 0;       // the tile queues of dz_spconv_forward_split_x
    const int nblk = (m + 255) >> 8;                                          // live 256-row blocks
    const int per = (nblk + 7) >> 3;
    const int b_lo = (int)(blockIdx.x & 7u) * per, b_hi = min(nblk, b_lo + per);
    const int upb = 256 / tile_rows;                                          // units per block: 1 or 2
    const int su = tid / tile_rows, ui = tid - su * tile_rows;                // my unit of the block, my index in it
    // mask words and windows past the live blocks: zero
    for (int o = nblk * 256 + blockIdx.x * 256 + tid; o < mask_rows; o += gridDim.x * 256)
        if ((tid & 31) == 0) tile_masks[o >> 5] = 0u;
    for (int u = nblk * upb + blockIdx.x * 256 + tid; u < nunits; u += gridDim.x * 256) {
#pragma unroll
        for (int k = 0; k < 6; ++k) win[(size_t)u * 6 + k] = 0;
    }
    for (int blk = b_lo + (int)(blockIdx.x >> 3); blk < b_hi; blk += (int)(gridDim.x >> 3)) {
        const int o = blk * 256 + tid;
        const bool mine = o < m;
        if (tid < 6) { (&lo_s[0][0])[tid] = 0x7FFFFFFF; (&hi_s[0][0])[tid] = -1; }
        int4 c = make_int4(0, 0, 0, 0);
        if (mine) c = reinterpret_cast<const int4 *>(coords_out)[o];
        const uint32_t voff = (uint32_t)o * 4u;
        const int uxc = c.w;                                                  // centre cell (submanifold: the row's own x)
        const bool lv = uxc > 0, rv = uxc + 1 < li.w;
        const int uz0 = c.y - 1, uy0 = c.z - 1;
        const uint32_t key00 = (uint32_t)(((c.x * li.d + uz0) * li.h + uy0) * li.w + uxc);
        struct Slab { nbr_u3 word[3], pref[3]; uint32_t kc[3], base[3]; bool ok[3], edge; } sl[3];
        auto fetch = [&](int tz, Slab &q) {
            q.edge = false;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                q.ok[ty] = mine && (unsigned)(uz0 + tz) < (unsigned)li.d && (unsigned)(uy0 + ty) < (unsigned)li.h;
                q.kc[ty] = q.ok[ty] ? key00 + (uint32_t)((tz * li.h + ty) * li.w) : 32u;
                const uint32_t wc = q.kc[ty] >> 5;
                q.edge |= wc - 1u > last_base;
                q.base[ty] = min(wc > 0u ? wc - 1u : 0u, last_base);
                q.word[ty] = __builtin_amdgcn_raw_buffer_load_b96(bmr, q.base[ty] * 4u, 0, 0);
                q.pref[ty] = __builtin_amdgcn_raw_buffer_load_b96(pfr, q.base[ty] * 4u, 0, 0);
            }
        };
        uint32_t words[9];
        uint32_t bits = 0u;
        fetch(0, sl[0]);
#pragma unroll
        for (int tz = 0; tz < 3; ++tz) {
            if (tz + 1 < 3) fetch(tz + 1, sl[tz + 1]);
            const Slab &q = sl[tz];
            const bool any_edge = __any(q.edge);
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                const int r = tz * 3 + ty;
                const uint32_t bc = q.kc[ty] & 31u;
                int vl, vc, vr;
                if (any_edge) nbr_row3<true>(q.word[ty], q.pref[ty], (int)((q.kc[ty] >> 5) - q.base[ty]), bc, q.ok[ty], lv, rv, vl, vc, vr);
                else nbr_row3<false>(q.word[ty], q.pref[ty], 1, bc, q.ok[ty], lv, rv, vl, vc, vr);
                const uint32_t rr = vl >= 0 ? (uint32_t)vl + 1u : (vc >= 0 ? (uint32_t)vc : (vr >= 0 ? (uint32_t)vr : 0u));
                const uint32_t e = rr | (vl >= 0 ? 1u << 29 : 0u) | (vc >= 0 ? 1u << 30 : 0u) | (vr >= 0 ? 1u << 31 : 0u);
                words[r] = mine ? e : 0u;
                if (mine) __builtin_amdgcn_raw_buffer_store_b32((int)e, tab, voff, (uint32_t)r * row_bytes, 0);
                bits |= ((vl >= 0 ? 1u : 0u) | (vc >= 0 ? 2u : 0u) | (vr >= 0 ? 4u : 0u)) << (r * 3);
            }
        }
        {
            uint32_t gb = bits;
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) gb |= (uint32_t)__shfl_xor((int)gb, d, 64);
            if ((tid & 31) == 0 && o < mask_rows) tile_masks[o >> 5] = gb;
        }
        __syncthreads();                                                      // lo_s / hi_s initialised
        // ---- windows of my unit: first / last referenced input row of every z slab (as k_xwin, sparse_conv_x.hip)
#pragma unroll
        for (int tz = 0; tz < 3; ++tz) {
            int lo = 0x7FFFFFFF, hi = -1;
#pragma unroll
            for (int ty = 0; ty < 3; ++ty) {
                const uint32_t e = words[tz * 3 + ty];
                if (!(e >> 29)) continue;
                const int r = (int)(e & 0x1FFFFFFFu), l = (int)((e >> 29) & 1u), cc = (int)((e >> 30) & 1u), rt = (int)(e >> 31);
                lo = min(lo, r - l);
                hi = max(hi, rt ? r + cc : (cc ? r : r - 1));
            }
            // (a wavefront lies inside one unit: units are 128 or 256 rows)
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) {
                lo = min(lo, __shfl_xor(lo, d, 64));
                hi = max(hi, __shfl_xor(hi, d, 64));
            }
            if ((tid & 63) == 0 && hi >= 0) { atomicMin(&lo_s[su][tz], lo); atomicMax(&hi_s[su][tz], hi); }
        }
        const int unit = blk * upb + su, row0 = unit * tile_rows;
        unsigned int key = 0x8000000u;
        if (perm) {
            const unsigned int k27 = (unit & 1) ? (0x7FFFFFFu - (bits & 0x7FFFFFFu)) : (bits & 0x7FFFFFFu);      // odd units descending
            key = mine ? k27 : 0x8000000u;
            key_s[tid] = key;
        }
        __syncthreads();
        if (ui < 3 && unit < nunits) {
            const int tz = ui;
            int lo = lo_s[su][tz], n = hi_s[su][tz] >= 0 ? hi_s[su][tz] - lo + 1 : 0;
            if (n == 0) lo = 0;
            if (tz == 1 && n == 0 && row0 < m) n = 1;        // a live unit always runs its centre slab
            win[(size_t)unit * 6 + 2 * tz] = lo;
            win[(size_t)unit * 6 + 2 * tz + 1] = n;
        }
        if (perm) {
            // rank of my (tap set, row) key among my unit's keys by counting (broadcast LDS reads, no exchange stages)
            int rank = 0;
            const unsigned int *ks = key_s + su * tile_rows;
            for (int j = 0; j < tile_rows; j += 4) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&ks[j]);
                rank += (v.x < key || (v.x == key && j < ui)) + (v.y < key || (v.y == key && j + 1 < ui)) + (v.z < key || (v.z == key && j + 2 < ui)) +
                        (v.w < key || (v.w == key && j + 3 < ui));
            }
            const int pos = row0 + rank;
            if (pos < cap_out) {
                perm[pos] = o;
#pragma unroll
                for (int g = 0; g < 9; ++g) nbr_sorted[(size_t)g * cap_out + pos] = (int)words[g];
            }
        }
        __syncthreads();                                                      // key_s / lo_s are rewritten by the next block
    }
}

// general form (brick key layout, other kernel widths / paddings): one thread per (o, tz, ty) row of kW taps; the tap masks are
// collected with atomics into zero-filled words.
__global__ __launch_bounds__(256) void k_build_neighbors(const int *__restrict__ coords_out,
                                                         const int *__restrict__ d_m_out, int cap_out,
                                                         const uint32_t *__restrict__ bitmap_in,
                                                         const uint32_t *__restrict__ prefix_in, LevelGeom li,
                                                         ConvGeom g, int *__restrict__ nbr,
                                                         uint32_t *__restrict__ tile_masks) {
    // tile_masks (optional): word o/32 collects, for 32 consecutive output rows (the pixel side of one 32x32 MFMA fragment),
    // the kernel taps that have at least one neighbour - what the conv kernels need to skip empty taps without scanning the
    // table again
    const int m = min(*d_m_out, cap_out);
    const int rows = g.k[0] * g.k[1];
    // rows padded to whole 64-row groups: a wavefront then works on ONE group (its lanes share the mask word)
    const int m_pad = (m + 63) & ~63;
    const long total = (long)m_pad * rows;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int o = (int)(idx % m_pad);
        const int zy = (int)(idx / m_pad);
        const int tz = zy / g.k[1], ty = zy % g.k[1];
        uint32_t bits = 0u;
        if (o < m) {
            const int4 c = reinterpret_cast<const int4 *>(coords_out)[o];
            const int uz = c.y * g.s[0] - g.p[0] + tz;
            const int uy = c.z * g.s[1] - g.p[1] + ty;
            const bool row_ok = (unsigned)uz < (unsigned)li.d && (unsigned)uy < (unsigned)li.h;
            const int base_x = c.w * g.s[2] - g.p[2];
            for (int tx = 0; tx < g.k[2]; ++tx) {
                const int ux = base_x + tx;
                int v = -1;
                if (row_ok && (unsigned)ux < (unsigned)li.w) v = bitmap_find(bitmap_in, prefix_in, li.key(c.x, uz, uy, ux));
                const int tap = (tz * g.k[1] + ty) * g.k[2] + tx;
                nbr[(size_t)tap * cap_out + o] = v;
                if (v >= 0) bits |= 1u << tap;
            }
        }
        if (tile_masks) {       // uniform per launch; the loop bounds are wave-uniform too (total is a multiple of 64)
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) bits |= (uint32_t)__shfl_xor((int)bits, d, 64);
            if ((threadIdx.x & 31) == 0 && bits) atomicOr(&tile_masks[o >> 5], bits);
        }
    }
}

__global__ void k_scatter_rows(const float *__restrict__ src, const int *__restrict__ rank,
                               const int *__restrict__ d_n, int n_cap, int c_src, float *__restrict__ dst,
                               int c_dst) {
    const int n = d_n ? min(*d_n, n_cap) : n_cap;
    const long total = (long)n * c_dst;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / c_dst), ch = (int)(idx % c_dst);
        const int r = rank[i];
        if (r < 0) continue;
        dst[(size_t)r * c_dst + ch] = (ch < c_src) ? src[(size_t)i * c_src + ch] : 0.f;
    }
}

__global__ void k_gather_rows(const float *__restrict__ src, const int *__restrict__ idx_,
                              const int *__restrict__ d_n, int n_cap, int c, float *__restrict__ out) {
    const int n = d_n ? min(*d_n, n_cap) : n_cap;
    const long total = (long)n * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / c), ch = (int)(idx % c);
        out[idx] = src[(size_t)idx_[i] * c + ch];
    }
}

static bool geom_from(const int *k3, const int *s3, const int *p3, int d, int h, int w, ConvGeom &g) {
    for (int i = 0; i < 3; ++i) {
        g.k[i] = k3[i]; g.s[i] = s3[i]; g.p[i] = p3[i];
        if (g.k[i] < 1 || g.s[i] < 1 || g.p[i] < 0) return false;
    }
    g.od = (d + 2 * g.p[0] - g.k[0]) / g.s[0] + 1;
    g.oh = (h + 2 * g.p[1] - g.k[1]) / g.s[1] + 1;
    g.ow = (w + 2 * g.p[2] - g.k[2]) / g.s[2] + 1;
    return g.od > 0 && g.oh > 0 && g.ow > 0;
}

// scan of a level's bitmap with coordinate emission in the level's key layout
int level_scan(const uint32_t *bitmap, const LevelGeom &lg, uint32_t *prefix, int *d_total, int *coords_out, int cap_out, void *ws,
               size_t ws_bytes, hipStream_t stream, bool nonzero_only, const unsigned char *line_flags) {
    const size_t nwords = align_up((lg.cells() + 31) / 32, 8);        // = dz_index_words
    if (lg.layout == 0)
        return bitmap_scan(bitmap, nwords, prefix, d_total, 0, ScanDims{lg.d, lg.h, lg.w}, coords_out, cap_out, ws, ws_bytes, stream,
                           nonzero_only, line_flags);
    return bitmap_scan(bitmap, nwords, prefix, d_total, 2, ScanDims{lg.d, lg.nby, lg.nbx}, coords_out, cap_out, ws, ws_bytes, stream,
                       nonzero_only, line_flags);
}

}  // namespace dz

using namespace dz;

extern "C" {

const char *dz_version(void) { return "detzero_hip 0.1 (gfx950)"; }
const char *dz_last_error(void) { return dz::last_error(); }

int dz_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

size_t dz_index_words(int b, int d, int h, int w, int layout) {
    const size_t cells = make_level(b, d, h, w, layout).cells();
    // padded to a multiple of 8 words so the scan can use 32-byte vector loads
    return align_up((cells + 31) / 32, 8);
}

size_t dz_index_workspace_bytes(int b, int d, int h, int w, int layout) {
    return bitmap_scan_workspace_bytes(dz_index_words(b, d, h, w, layout));
}

static int check_cells(int b, int d, int h, int w, int layout) {
    if (b < 1 || d < 1 || h < 1 || w < 1 || (layout != DZ_LAYOUT_LINEAR && layout != DZ_LAYOUT_BRICK) ||
        make_level(b, d, h, w, layout).cells() >= 0xFFFFFFFFull) {
        set_error("grid %d x %d x %d x %d (layout %d) does not fit 32-bit cell keys", b, d, h, w, layout);
        return DZ_ERR_UNSUPPORTED;
    }
    return DZ_OK;
}


int dz_index_from_coords(const int *coords, const int *d_n, int n_cap, int b, int d, int h, int w, int layout,
                         uint32_t *bitmap, uint32_t *prefix, int *coords_out, int *d_m, int cap_out,
                         int *rank_of_input, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG((coords || n_cap == 0) && bitmap && prefix && coords_out && d_m && n_cap >= 0 && cap_out >= 0,
                 "dz_index_from_coords: null/negative argument");
    int rc = check_cells(b, d, h, w, layout);
    if (rc) return rc;
    const LevelGeom lg = make_level(b, d, h, w, layout);
    const size_t nwords = dz_index_words(b, d, h, w, layout);
    rc = fill_u32(bitmap, 0u, nwords, stream);
    if (rc) return rc;
    if (n_cap > 0)
        hipLaunchKernelGGL(k_set_bits_from_coords, dim3(stream_grid(n_cap, 256)), dim3(256), 0, stream, coords,
                           d_n, n_cap, lg, bitmap);
    rc = level_scan(bitmap, lg, prefix, d_m, coords_out, cap_out, ws, ws_bytes, stream, false);
    if (rc) return rc;
    if (rank_of_input && n_cap > 0)
        hipLaunchKernelGGL(k_rank_of_coords, dim3(stream_grid(n_cap, 256)), dim3(256), 0, stream, coords, d_n,
                           n_cap, lg, bitmap, prefix, rank_of_input);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_index_downsample(const int *coords_in, const int *d_m_in, int cap_in, int b, int d, int h, int w, int layout,
                        const int *h_k3, const int *h_s3, const int *h_p3, uint32_t *bitmap_out,
                        uint32_t *prefix_out, int *coords_out, int *d_m_out, int cap_out, void *ws,
                        size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(coords_in && d_m_in && bitmap_out && prefix_out && coords_out && d_m_out,
                 "dz_index_downsample: null argument");
    ConvGeom g;
    DZ_CHECK_ARG(geom_from(h_k3, h_s3, h_p3, d, h, w, g), "dz_index_downsample: bad kernel/stride/padding");
    int rc = check_cells(b, g.od, g.oh, g.ow, layout);
    if (rc) return rc;
    const LevelGeom lo = make_level(b, g.od, g.oh, g.ow, layout);
    const size_t nwords = dz_index_words(b, g.od, g.oh, g.ow, layout);
    rc = fill_u32(bitmap_out, 0u, nwords, stream);
    if (rc) return rc;
    static const int plain = getenv("DZ_TUNE_MARK_PLAIN") ? atoi(getenv("DZ_TUNE_MARK_PLAIN")) : 0;     // development knob
    const bool two = (g.k[0] + g.s[0] - 1) / g.s[0] <= 2 && (g.k[1] + g.s[1] - 1) / g.s[1] <= 2 && (g.k[2] + g.s[2] - 1) / g.s[2] <= 2;
    if (cap_in > 0 && two && !plain)
        hipLaunchKernelGGL(k_mark_outputs_w, dim3(stream_grid(cap_in, 256)), dim3(256), 0, stream, coords_in, d_m_in,
                           cap_in, g, lo, bitmap_out);
    else if (cap_in > 0)
        hipLaunchKernelGGL(k_mark_outputs, dim3(stream_grid(cap_in, 256)), dim3(256), 0, stream, coords_in, d_m_in,
                           cap_in, g, lo, bitmap_out);
    rc = level_scan(bitmap_out, lo, prefix_out, d_m_out, coords_out, cap_out, ws, ws_bytes, stream, false);
    if (rc) return rc;
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_tile_masks_words(int cap_out) { return cap_out < 0 ? 0 : tile_masks_words(cap_out); }

int dz_build_neighbors(const int *coords_out, const int *d_m_out, int cap_out, const uint32_t *bitmap_in,
                       const uint32_t *prefix_in, int b, int d, int h, int w, int layout, const int *h_k3, const int *h_s3,
                       const int *h_p3, int *nbr, uint32_t *tile_masks, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(coords_out && d_m_out && bitmap_in && prefix_in && nbr, "dz_build_neighbors: null argument");
    ConvGeom g;
    DZ_CHECK_ARG(geom_from(h_k3, h_s3, h_p3, d, h, w, g), "dz_build_neighbors: bad kernel/stride/padding");
    DZ_CHECK_ARG(layout == DZ_LAYOUT_LINEAR || layout == DZ_LAYOUT_BRICK, "dz_build_neighbors: bad layout %d", layout);
    if (cap_out == 0) return DZ_OK;
    const LevelGeom li = make_level(b, d, h, w, layout);
    // the per-row kernel: linear keys, three z taps, and an x window whose centre cell is inside the input grid for every output
    // (3 wide with padding 1: centre = x * s <= w - 1 because ow = (w - 1) / s + 1; 1 wide without padding: centre = x * s)
    static const int generic = getenv("DZ_TUNE_NBR_GENERIC") ? atoi(getenv("DZ_TUNE_NBR_GENERIC")) : 0;    // development knob
    const bool kw3 = g.k[2] == 3 && g.p[2] == 1 && (long)(g.ow - 1) * g.s[2] <= w - 1;
    const bool kw1 = g.k[2] == 1 && g.p[2] == 0 && (long)(g.ow - 1) * g.s[2] <= w - 1;
    const size_t table_bytes = (size_t)g.k[0] * g.k[1] * g.k[2] * cap_out * sizeof(int);         // (descriptor-addressed: below 4 GiB)
    if (!generic && layout == DZ_LAYOUT_LINEAR && g.k[0] == 3 && ((g.k[1] == 3 && kw3) || (g.k[1] == 1 && kw1)) &&
        table_bytes < 0xFFFFFFFFull) {
        const int mask_rows = tile_masks_words(cap_out) * 32;
        const size_t index_words = dz_index_words(b, d, h, w, layout);                       // (>= 8 words: padded to whole 32-byte units)
        const uint32_t last_base = (uint32_t)(index_words - 3);
        static const int gmax = getenv("DZ_TUNE_NBR_GRID") ? atoi(getenv("DZ_TUNE_NBR_GRID")) : 2048;           // development knob
        const dim3 grid((std::min(stream_grid(tile_masks ? mask_rows : cap_out, 256), gmax) + 7) & ~7);
        static const int flat = getenv("DZ_TUNE_NBR_FLAT") ? atoi(getenv("DZ_TUNE_NBR_FLAT")) : 0;             // development knob
#define DZ_NBR_ROWS(K1_, KW_, X_)                                                                                                   \
    hipLaunchKernelGGL((k_build_neighbors_rows<3, K1_, KW_, X_>), grid, dim3(256), 0, stream, coords_out, d_m_out, cap_out, bitmap_in, \
                       prefix_in, li, g, nbr, tile_masks, mask_rows, last_base, (uint32_t)(index_words * 4), (uint32_t)table_bytes)
        if (g.k[1] == 3) { if (flat) DZ_NBR_ROWS(3, 3, false); else DZ_NBR_ROWS(3, 3, true); }
        else { if (flat) DZ_NBR_ROWS(1, 1, false); else DZ_NBR_ROWS(1, 1, true); }
#undef DZ_NBR_ROWS
        DZ_LAUNCH_CHECK();
        return DZ_OK;
    }
    if (tile_masks) {
        const int rc = fill_u32(tile_masks, 0u, (size_t)tile_masks_words(cap_out), stream);
        if (rc) return rc;
    }
    const long work = (long)cap_out * g.k[0] * g.k[1];
    hipLaunchKernelGGL(k_build_neighbors, dim3(stream_grid(work, 256)), dim3(256), 0, stream, coords_out, d_m_out,
                       cap_out, bitmap_in, prefix_in, li, g, nbr, tile_masks);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_build_neighbors_packed(const int *coords_out, const int *d_m_out, int cap_out, const uint32_t *bitmap_in,
                              const uint32_t *prefix_in, int b, int d, int h, int w, int layout, const int *h_k3, const int *h_s3,
                              const int *h_p3, int *nbr, uint32_t *tile_masks, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(coords_out && d_m_out && bitmap_in && prefix_in && nbr && tile_masks, "dz_build_neighbors_packed: null argument");
    ConvGeom g;
    DZ_CHECK_ARG(geom_from(h_k3, h_s3, h_p3, d, h, w, g), "dz_build_neighbors_packed: bad kernel/stride/padding");
    const bool kw3 = g.k[2] == 3 && g.p[2] == 1 && (long)(g.ow - 1) * g.s[2] <= w - 1;
    const size_t table_bytes = (size_t)9 * cap_out * sizeof(int);
    if (layout != DZ_LAYOUT_LINEAR || g.k[0] != 3 || g.k[1] != 3 || !kw3 || table_bytes >= 0xFFFFFFFFull || cap_out >= (1 << 29)) {
        set_error("dz_build_neighbors_packed: needs linear keys, a 3 x 3 x 3 window with x padding 1 and fewer than 2^29 rows");
        return DZ_ERR_UNSUPPORTED;
    }
    if (cap_out == 0) return DZ_OK;
    const LevelGeom li = make_level(b, d, h, w, layout);
    const int mask_rows = tile_masks_words(cap_out) * 32;
    const size_t index_words = dz_index_words(b, d, h, w, layout);
    const dim3 grid((stream_grid(mask_rows, 256) + 7) & ~7);
    hipLaunchKernelGGL((k_build_neighbors_rows<3, 3, 3, true, true>), grid, dim3(256), 0, stream, coords_out, d_m_out, cap_out, bitmap_in,
                       prefix_in, li, g, nbr, tile_masks, mask_rows, (uint32_t)(index_words - 3), (uint32_t)(index_words * 4),
                       (uint32_t)table_bytes);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_build_neighbors_packed_x(const int *coords, const int *d_m, int cap, const uint32_t *bitmap, const uint32_t *prefix, int b, int d, int h,
                                int w, int *nbr, uint32_t *tile_masks, int tile_rows, int *windows, int *nbr_sorted, int *perm, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(coords && d_m && bitmap && prefix && nbr && tile_masks && windows, "dz_build_neighbors_packed_x: null argument");
    DZ_CHECK_ARG(b >= 1 && d >= 1 && h >= 1 && w >= 1 && cap >= 0, "dz_build_neighbors_packed_x: bad grid");
    DZ_CHECK_ARG(tile_rows == 128 || tile_rows == 256, "dz_build_neighbors_packed_x: tile_rows %d (128 or 256: dz_spconv_x_tile_rows)", tile_rows);
    DZ_CHECK_ARG((nbr_sorted == nullptr) == (perm == nullptr), "dz_build_neighbors_packed_x: nbr_sorted and perm come together");
    const size_t table_bytes = (size_t)9 * cap * sizeof(int);
    if (table_bytes >= 0xFFFFFFFFull || cap >= (1 << 29)) {
        set_error("dz_build_neighbors_packed_x: needs fewer than 2^29 rows");
        return DZ_ERR_UNSUPPORTED;
    }
    if (cap == 0) return DZ_OK;
    const LevelGeom li = make_level(b, d, h, w, DZ_LAYOUT_LINEAR);
    const int mask_rows = tile_masks_words(cap) * 32;
    const size_t index_words = dz_index_words(b, d, h, w, DZ_LAYOUT_LINEAR);
    const dim3 grid((std::min(ceil_div(cap, 256), 2048) + 7) & ~7);
    hipLaunchKernelGGL(k_build_neighbors_xwin, grid, dim3(256), 0, stream, coords, d_m, cap, bitmap, prefix, li, nbr, tile_masks, mask_rows,
                       (uint32_t)(index_words - 3), (uint32_t)(index_words * 4), (uint32_t)table_bytes, tile_rows, windows, nbr_sorted, perm);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_scatter_rows(const float *src, const int *rank, const int *d_n, int n_cap, int c_src, float *dst, int c_dst,
                    void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(src && rank && dst && c_src > 0 && c_dst >= c_src, "dz_scatter_rows: bad argument");
    if (n_cap == 0) return DZ_OK;
    hipLaunchKernelGGL(k_scatter_rows, dim3(stream_grid((long)n_cap * c_dst, 256)), dim3(256), 0, stream, src, rank,
                       d_n, n_cap, c_src, dst, c_dst);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_gather_rows(const float *src, const int *idx, const int *d_n, int n_cap, int c, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(src && idx && out && c > 0, "dz_gather_rows: bad argument");
    if (n_cap == 0) return DZ_OK;
    hipLaunchKernelGGL(k_gather_rows, dim3(stream_grid((long)n_cap * c, 256)), dim3(256), 0, stream, src, idx, d_n,
                       n_cap, c, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
