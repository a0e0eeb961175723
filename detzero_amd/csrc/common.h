// Shared host/device helpers for libdetzero_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/detzero_hip.h"

namespace dz {

void set_error(const char *fmt, ...);

#define DZ_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            dz::set_error(__VA_ARGS__);         \
            return DZ_ERR_INVALID;              \
        }                                       \
    } while (0)

#define DZ_HIP(expr)                                                                     \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            dz::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return DZ_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

#define DZ_LAUNCH_CHECK()                                                                \
    do {                                                                                 \
        hipError_t _e = hipGetLastError();                                               \
        if (_e != hipSuccess) {                                                          \
            dz::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return DZ_ERR_HIP;                                                           \
        }                                                                                \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

// words of a neighbour table's tile_masks buffer: one tap mask per 32 output rows (whole 16-byte units)
static inline int tile_masks_words(int cap) { return (cap / 32 + 1 + 3) & ~3; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// persistent-grid size for memory-bound grid-stride kernels: enough blocks to fill 256 CUs x 8
static inline int stream_grid(long work_items, int block) {
    long g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 2048) g = 2048;
    return (int)g;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: one process may drive several GPUs
// (ops._guarded), so the "already raised" flag of a kernel is kept per device.  `done` = a function-local static array.
constexpr int DZ_MAX_DEVICES = 32;
struct PerDeviceFlags { bool v[DZ_MAX_DEVICES] = {}; };
static inline int reserve_lds(const void *kernel, int bytes, PerDeviceFlags &done, const char *who) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { set_error("%s: hipGetDevice failed", who); return DZ_ERR_HIP; }
    if (dev >= 0 && dev < DZ_MAX_DEVICES && done.v[dev]) return DZ_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        set_error("%s: cannot reserve %d bytes of LDS on device %d", who, bytes, dev);
        return DZ_ERR_HIP;
    }
    if (dev >= 0 && dev < DZ_MAX_DEVICES) done.v[dev] = true;
    return DZ_OK;
}
// compute units of the CURRENT device (cached per device)
static inline int device_cus() {
    static int cus[DZ_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev >= 0 && dev < DZ_MAX_DEVICES && cus[dev]) return cus[dev];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < DZ_MAX_DEVICES) cus[dev] = n;
    return n;
}

constexpr uint32_t KEY_INVALID = 0xFFFFFFFFu;

// ---- bitmap index lookups -------------------------------------------------------------------
// rank of an ACTIVE cell `key` = number of active cells with a smaller key.
// CONTRACT: levels built by dz_voxelize_to_level carry a PARTIAL prefix (written only at words that hold a bit: their level-1
// bitmap is > 97 % empty words) - bitmap_rank on a key whose WORD is empty reads an unwritten prefix entry.  Rank only ACTIVE keys
// on such a level (bitmap_find tests the bit first and is always safe); ops.SparseLevel.prefix_partial records the property and
// the wrappers that rank arbitrary keys (PDV ball query / grouping: batch starts) refuse a partial level.
__device__ __forceinline__ int bitmap_rank(const uint32_t *__restrict__ bitmap,
                                           const uint32_t *__restrict__ prefix, uint32_t key) {
    const uint32_t w = key >> 5, bit = key & 31u;
    const uint32_t word = bitmap[w];
    return (int)(prefix[w] + __popc(word & ((1u << bit) - 1u)));
}
// rank or -1 when the cell is not active
__device__ __forceinline__ int bitmap_find(const uint32_t *__restrict__ bitmap,
                                           const uint32_t *__restrict__ prefix, uint32_t key) {
    const uint32_t w = key >> 5, bit = key & 31u;
    const uint32_t word = bitmap[w];
    if (!((word >> bit) & 1u)) return -1;
    return (int)(prefix[w] + __popc(word & ((1u << bit) - 1u)));
}

// ---- cell keys of a sparse level -------------------------------------------------------------------------------
// layout 0 (DZ_LAYOUT_LINEAR): key = ((b*D + z)*H + y)*W + x - rows in ascending (b, z, y, x) order.
// layout 1 (DZ_LAYOUT_BRICK):  the (y, x) plane is cut into 8 x 8 columns that run through all of z; a column ("brick") is
//   D * 64 consecutive keys = 2 * D bitmap words:  key = ((((b*NBY + y/8)*NBX + x/8)*D + z) << 6) | (y%8 << 3) | x%8.
//   Feature rows (= ranks of the set bits) of a spatial neighbourhood are then close together in memory: the rows a
//   128-row tile of a 3x3x3 convolution reads are 1.3-1.8x its own rows (4.3x in the linear order), which is what lets
//   the tile convolution (sparse_conv_t.hip) stage a tile's inputs in LDS once.
struct LevelGeom {
    int b, d, h, w, layout, nby, nbx;
    __host__ __device__ __forceinline__ uint32_t key(int bi, int z, int y, int x) const {
        if (layout == 0) return (uint32_t)(((bi * d + z) * h + y) * w + x);
        const uint32_t brick = (uint32_t)((bi * nby + (y >> 3)) * nbx + (x >> 3));
        return ((brick * (uint32_t)d + (uint32_t)z) << 6) | (uint32_t)(((y & 7) << 3) | (x & 7));
    }
    __host__ __device__ __forceinline__ bool inside(int bi, int z, int y, int x) const {
        return (unsigned)bi < (unsigned)b && (unsigned)z < (unsigned)d && (unsigned)y < (unsigned)h && (unsigned)x < (unsigned)w;
    }
    __host__ __device__ __forceinline__ size_t cells() const {
        return layout == 0 ? (size_t)b * d * h * w : (size_t)b * nby * nbx * d * 64;
    }
    // first key of batch item bi (its keys are one contiguous range in both layouts)
    __host__ __device__ __forceinline__ uint32_t batch_key(int bi) const {
        return layout == 0 ? (uint32_t)bi * (uint32_t)(d * h * w) : (uint32_t)bi * (uint32_t)(nby * nbx * d * 64);
    }
};
static inline LevelGeom make_level(int b, int d, int h, int w, int layout) {
    LevelGeom g;
    g.b = b; g.d = d; g.h = h; g.w = w; g.layout = layout ? 1 : 0;
    g.nby = (h + 7) / 8; g.nbx = (w + 7) / 8;
    return g;
}

// Exclusive scan of one value per thread across a 256-thread block (4 waves of 64).
// lds: >= 4 uint32.  Returns the exclusive prefix of `v`; `total` = block sum.
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *lds, uint32_t &total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t t = __shfl_up(incl, off, 64);
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds[wid] = incl;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t s = lds[w];
        if (w < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return woff + incl - v;
}

// Bitmap scan (3 launches): prefix[w] = #set bits in words < w ; *d_total = popcount of all.
// Optionally emits the coordinates of every set bit at its rank:
//   mode 0: key = ((b*D+z)*H+y)*W+x  -> coords [b,z,y,x]   (dims = D,H,W)
//   mode 2: brick keys of a LevelGeom (layout 1; dims = D, NBY, NBX) -> coords [b,z,y,x]
//   mode 1: key = ((b*GX+x)*GY+y)*GZ+z -> coords [b,z,y,x] (dims = GX,GY,GZ)   (DynamicMeanVFE order)
//   mode -1: no coordinates
struct ScanDims { int d0, d1, d2; };
// device fill (32-bit pattern) as a plain kernel: no memset nodes, so whole frames capture into a hipGraph
// as kernels only
int fill_u32(void *ptr, uint32_t value, size_t nwords, hipStream_t stream);
size_t bitmap_scan_workspace_bytes(size_t nwords);
int bitmap_scan(const uint32_t *bitmap, size_t nwords, uint32_t *prefix, int *d_total, int mode,
                ScanDims dims, int *coords_out, int cap_out, void *ws, size_t ws_bytes,
                hipStream_t stream, bool nonzero_only = false,      // nonzero_only (modes 0, 2): prefix[] valid only at words with a bit set
                const unsigned char *line_flags = nullptr);         // optional: one byte per 32 words, 0 = the line holds no bit (not read)
// scan of a level's bitmap, coordinates [b,z,y,x] emitted at their rank in the level's key layout
int level_scan(const uint32_t *bitmap, const LevelGeom &lg, uint32_t *prefix, int *d_total, int *coords_out, int cap_out, void *ws,
               size_t ws_bytes, hipStream_t stream, bool nonzero_only, const unsigned char *line_flags = nullptr);

}  // namespace dz
