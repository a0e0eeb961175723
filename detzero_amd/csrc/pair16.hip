// pair16 <-> fp32 conversion and the memory-bound row movers of the split-precision path (gfx950).
// Layout and arithmetic: hgemm.h.  These kernels only stand at the ends of the pipeline (input voxel features,
// tensors handed back to PyTorch callers) and at the sparse -> BEV hand-over (height_compression.py:20-24).
#include "hgemm.h"

namespace dz {

template <class M>
__global__ void k_pair16_from_f32(const float *__restrict__ src, long rows, int c_src, int c_dst, uint4 *__restrict__ dst) {
    const int groups = c_dst / 8;
    const long total = rows * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long r = idx / groups;
        const int g = (int)(idx % groups);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (g * 8 + e < c_src) ? src[r * c_src + g * 8 + e] : 0.f;
        const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        uint2 h0, l0, h1, l1;
        split4<M>(a, h0, l0);
        split4<M>(b, h1, l1);
        dst[idx * 2] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[idx * 2 + 1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

template <class M>
__global__ void k_pair16_to_f32(const uint4 *__restrict__ src, long rows, int c, float *__restrict__ dst) {
    const int groups = c / 8;
    const long total = rows * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const uint4 h = src[idx * 2], l = src[idx * 2 + 1];
        const unsigned int hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
        float4 o0, o1;
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[2 * e] = M::join(hw[e] & 0xFFFFu, lw[e] & 0xFFFFu);
            o[2 * e + 1] = M::join(hw[e] >> 16, lw[e] >> 16);
        }
        o0 = make_float4(o[0], o[1], o[2], o[3]);
        o1 = make_float4(o[4], o[5], o[6], o[7]);
        reinterpret_cast<float4 *>(dst)[idx * 2] = o0;
        reinterpret_cast<float4 *>(dst)[idx * 2 + 1] = o1;
    }
}

template <class M>
__global__ void k_scatter_rows_split(const float *__restrict__ src, const int *__restrict__ rank, const int *__restrict__ d_n,
                                     int n_cap, int c_src, uint4 *__restrict__ dst, int c_dst) {
    const int n = d_n ? min(*d_n, n_cap) : n_cap;
    const int groups = c_dst / 8;
    const long total = (long)n * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / groups), g = (int)(idx % groups);
        const int r = rank[i];
        if (r < 0) continue;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (g * 8 + e < c_src) ? src[(size_t)i * c_src + g * 8 + e] : 0.f;
        const float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        uint2 h0, l0, h1, l1;
        split4<M>(a, h0, l0);
        split4<M>(b, h1, l1);
        const size_t o = ((size_t)r * groups + g) * 2;
        dst[o] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[o + 1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

// bev[b][y+pad][x+pad][ch*D + z] <- feats[o][ch]: one thread moves the hi and lo halves of one channel
__global__ void k_sparse_to_bev_split(const unsigned short *__restrict__ feats, const int *__restrict__ coords,
                                      const int *__restrict__ d_m, int cap, int c, int d, int h, int w, int pad,
                                      unsigned short *__restrict__ bev) {
    const int m = min(*d_m, cap);
    const long total = (long)m * c;
    const int hp = h + 2 * pad, wp = w + 2 * pad;
    const int cd = c * d;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int o = (int)(idx / c), ch = (int)(idx % c);
        const int4 cc = reinterpret_cast<const int4 *>(coords)[o];  // [b,z,y,x]
        const size_t pix = ((size_t)cc.x * hp + (cc.z + pad)) * wp + (cc.w + pad);
        const unsigned short *s = feats + ((size_t)o * c + (ch & ~7)) * 2 + (ch & 7);   // 16-bit units: group start * 2
        const int oc = ch * d + cc.y;
        unsigned short *t = bev + (pix * cd + (oc & ~7)) * 2 + (oc & 7);
        t[0] = s[0];
        t[8] = s[8];
    }
}

// The same hand-over written DENSE for two z slabs (the encoded tensor of VoxelResBackBone8x): one thread per (pixel, group of 8
// output channels) looks the pixel's two cells up in the level's bitmap and writes the group - 4 channels of the z = 0 row
// interleaved with 4 of the z = 1 row (channel ch * 2 + z), zeros where a cell is empty or the pixel is border.  Every byte of
// the image is written once with 16-byte stores: no zero-fill pass over the 591 MB canvas and no 2-byte scatter stores.
__global__ __launch_bounds__(256) void k_sparse_to_bev_split_dense2(const uint2 *__restrict__ feats, const uint32_t *__restrict__ bitmap,
                                                                     const uint32_t *__restrict__ prefix, LevelGeom lg, int c, int pad,
                                                                     uint4 *__restrict__ bev, int feat_rows) {
    const int hp = lg.h + 2 * pad, wp = lg.w + 2 * pad;
    const int groups = c * 2 / 8, in_groups = c / 8;
    const long total = (long)lg.b * hp * wp * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int og = (int)(idx % groups);
        const long pix = idx / groups;
        const int xp = (int)(pix % wp), yp = (int)((pix / wp) % hp), b = (int)(pix / ((long)wp * hp));
        const int x = xp - pad, y = yp - pad;
        uint2 h0 = make_uint2(0u, 0u), l0 = h0, h1 = h0, l1 = h0;
        if ((unsigned)x < (unsigned)lg.w && (unsigned)y < (unsigned)lg.h) {
            int r0 = bitmap_find(bitmap, prefix, lg.key(b, 0, y, x)), r1 = bitmap_find(bitmap, prefix, lg.key(b, 1, y, x));
            // a level that overflowed its row capacity still has every site in its bitmap: ranks past the feature rows read as empty
            // cells (the pipeline's overflow flag reports the frame; nothing is read past the buffer)
            if (r0 >= feat_rows) r0 = -1;
            if (r1 >= feat_rows) r1 = -1;
            // input group og / 2 (16 B hi | 16 B lo = four uint2), its channels (og & 1) * 4 .. + 3
            if (r0 >= 0) { const uint2 *s = feats + ((size_t)r0 * in_groups + (og >> 1)) * 4 + (og & 1); h0 = s[0]; l0 = s[2]; }
            if (r1 >= 0) { const uint2 *s = feats + ((size_t)r1 * in_groups + (og >> 1)) * 4 + (og & 1); h1 = s[0]; l1 = s[2]; }
        }
        auto zip = [](uint2 a, uint2 z) {       // 16-bit interleave: a0 z0 a1 z1 a2 z2 a3 z3
            return make_uint4((a.x & 0xFFFFu) | (z.x << 16), (a.x >> 16) | (z.x & 0xFFFF0000u), (a.y & 0xFFFFu) | (z.y << 16),
                              (a.y >> 16) | (z.y & 0xFFFF0000u));
        };
        bev[idx * 2] = zip(h0, h1);
        bev[idx * 2 + 1] = zip(l0, l1);
    }
}

// Row-index image of a level with two z slabs: idx[b][y + pad][x + pad][z] = feature row of cell (b, z, y, x), -1 where the cell is
// empty, the pixel is border, or the rank lies past the feature rows (a level that overflowed its calibrated capacity).  8 bytes
// per pixel instead of the pixel's 2 x C channels: what the sparse-input convolution (conv3x3_h.hip, dz_conv2d_desc.in_rowidx)
// reads in place of the dense BEV image.
__global__ __launch_bounds__(256) void k_bev_row_index(const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ prefix, LevelGeom lg, int pad,
                                                       int feat_rows, int2 *__restrict__ idx) {
    const int hp = lg.h + 2 * pad, wp = lg.w + 2 * pad;
    const long total = (long)lg.b * hp * wp;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
        const int xp = (int)(pix % wp), yp = (int)((pix / wp) % hp), b = (int)(pix / ((long)wp * hp));
        const int x = xp - pad, y = yp - pad;
        int r0 = -1, r1 = -1;
        if ((unsigned)x < (unsigned)lg.w && (unsigned)y < (unsigned)lg.h) {
            r0 = bitmap_find(bitmap, prefix, lg.key(b, 0, y, x));
            r1 = bitmap_find(bitmap, prefix, lg.key(b, 1, y, x));
            if (r0 >= feat_rows) r0 = -1;
            if (r1 >= feat_rows) r1 = -1;
        }
        idx[pix] = make_int2(r0, r1);
    }
}

// ---- pixel tiles of the sparse-input convolution (8 x 32 output pixels, conv3x3_h.hip) whose 10 x 34 input halo holds no row at all:
// the convolution of such a tile is ReLU(shift) in every pixel.  k_bev_tile_flags marks them (one wavefront per tile), k_bev_tile_compact
// writes list[0] = n occupied, list[1] = n empty, then the occupied tile ids in ascending order, then the empty ones; the convolution
// walks only the occupied ones, k_bev_fill_tiles writes the constant into the empty ones - the same bits the kernel would have produced
// (acc = 0 -> fmaf(0, scale, shift) = shift -> ReLU -> the (hi, lo) split).
// Chebyshev distance (capped at DCAP) of every pixel to the nearest pixel that holds a row; outside the image there are none.
// D >= l + 1 on all pixels of a tile <=> the tile's input halo at dense layer l (1 = the sparse-input layer) lies in the region where the
// network's activations are its ZERO-INPUT RESPONSE: the tile's result is a copy of that response (dz_bev_fill_empty_tiles).
constexpr int BEV_DCAP = 8;
__global__ __launch_bounds__(256) void k_bev_tile_mind(const int2 *__restrict__ idx, int batch, int hp, int wp, int ho, int wo, int tiles_y, int tiles_x,
                                                       unsigned char *__restrict__ mind) {
    // one workgroup per tile: occupancy of the tile's pixels + BEV_DCAP pixels around them into LDS, then each of the 256 threads
    // grows a square around its pixel
    constexpr int R = BEV_DCAP - 1, SH = 8 + 2 * R, SW = 32 + 2 * R;
    __shared__ unsigned char occ[SH][SW + 2];
    __shared__ int tile_min;
    const int t = blockIdx.x;
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int y0 = ty * 8 - R, x0 = tx * 32 - R;             // image coordinates (unpadded) of the LDS window's origin
    if (threadIdx.x == 0) tile_min = BEV_DCAP;
    for (int k = threadIdx.x; k < SH * SW; k += 256) {
        const int ry = k / SW, rx = k - ry * SW;
        const int y = y0 + ry, x = x0 + rx;
        unsigned char o = 0;
        if ((unsigned)y < (unsigned)ho && (unsigned)x < (unsigned)wo) {
            const int2 v = idx[((size_t)b * hp + y + 1) * wp + x + 1];
            o = (v.x >= 0 || v.y >= 0) ? 1 : 0;
        }
        occ[ry][rx] = o;
    }
    __syncthreads();
    const int py = threadIdx.x >> 5, px = threadIdx.x & 31;
    int d = BEV_DCAP;
    if (ty * 8 + py < ho && tx * 32 + px < wo) {
        const int cy = py + R, cx = px + R;
        if (occ[cy][cx]) d = 0;
        else {
            for (int r = 1; r <= R && d == BEV_DCAP; ++r) {
                bool hit = false;
                for (int k = -r; k <= r; ++k)
                    hit |= occ[cy - r][cx + k] | occ[cy + r][cx + k] | occ[cy + k][cx - r] | occ[cy + k][cx + r];
                if (hit) d = r;
            }
        }
    }
    atomicMin(&tile_min, d);
    __syncthreads();
    if (threadIdx.x == 0) mind[t] = (unsigned char)tile_min;
}

// list l (blockIdx.x = l - 1, l = 1 .. nlists): [n occupied, n skippable, ids of the tiles with min D < l + 1 ascending, the others], stride `stride` ints
__global__ __launch_bounds__(1024) void k_bev_tile_compact(const unsigned char *__restrict__ mind, int n, int *__restrict__ lists, int stride) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, need = (int)blockIdx.x + 2;
    int *const list = lists + (size_t)blockIdx.x * stride;
    const int per = (n + 1023) / 1024, lo = min(n, tid * per), hi = min(n, lo + per);
    int c = 0;
    for (int i = lo; i < hi; ++i) c += mind[i] < need;
    part[tid] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                      // inclusive scan of the per-thread counts
        const int v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const int total = part[1023];
    int occ = part[tid] - c, emp = lo - occ;                 // occupied / skippable tiles before my range
    for (int i = lo; i < hi; ++i) {
        if (mind[i] < need) list[2 + occ++] = i;
        else list[2 + total + emp++] = i;
    }
    if (tid == 0) { list[0] = total; list[1] = n - total; }
}

template <class M>
__global__ __launch_bounds__(256) void k_bev_fill_tiles(const int *__restrict__ list, int tiles_y, int tiles_x, int ho, int wo, const float *__restrict__ shift,
                                                        int relu, int cout, float *__restrict__ out, int out_hp, int out_wp, int out_cstride, int out_coff,
                                                        const float *__restrict__ zero_resp) {
    const int n_occ = list[0], n_emp = list[1];
    if ((int)blockIdx.x >= n_emp) return;
    const int t = list[2 + n_occ + blockIdx.x];
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int groups = cout / 8;
    for (int k = threadIdx.x; k < 8 * 32 * groups; k += 256) {
        const int g = k % groups, px = k / groups, r = px / 32, c = px % 32;
        const int y = ty * 8 + r, x = tx * 32 + c;
        if (y >= ho || x >= wo) continue;
        uint4 *dst = reinterpret_cast<uint4 *>(out + (((size_t)b * out_hp + y + 1) * out_wp + x + 1) * out_cstride + out_coff + g * 8);
        if (zero_resp) {
            // the layer's response to an all-zero input at this pixel: a (1, out_hp, out_wp, cout) pair16 image computed once per model
            const uint4 *src = reinterpret_cast<const uint4 *>(zero_resp + (((size_t)(y + 1)) * out_wp + x + 1) * cout + g * 8);
            dst[0] = src[0];
            dst[1] = src[1];
            continue;
        }
        float v0[4], v1[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = shift ? shift[g * 8 + e] : 0.f, bb = shift ? shift[g * 8 + 4 + e] : 0.f;
            v0[e] = relu ? fmaxf(a, 0.f) : a;
            v1[e] = relu ? fmaxf(bb, 0.f) : bb;
        }
        uint2 h0, l0, h1, l1;
        split4<M>(v0, h0, l0);
        split4<M>(v1, h1, l1);
        dst[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
        dst[1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
    }
}

}  // namespace dz

using namespace dz;

extern "C" {

size_t dz_bev_tile_list_words(int batch, int ho, int wo) { return (size_t)batch * ceil_div(ho, 8) * ceil_div(wo, 32) + 2; }

int dz_bev_tile_list(const int *idx, int batch, int hp, int wp, int ho, int wo, int nlists, int *lists, unsigned char *mind_ws, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(idx && lists && mind_ws && batch > 0 && ho > 0 && wo > 0 && hp >= ho + 2 && wp >= wo + 2 && nlists >= 1 && nlists < BEV_DCAP,
                 "dz_bev_tile_list: bad argument (1 <= nlists <= %d)", BEV_DCAP - 1);
    const int ty = ceil_div(ho, 8), tx = ceil_div(wo, 32), n = batch * ty * tx;
    hipLaunchKernelGGL(k_bev_tile_mind, dim3(n), dim3(256), 0, stream, reinterpret_cast<const int2 *>(idx), batch, hp, wp, ho, wo, ty, tx, mind_ws);
    hipLaunchKernelGGL(k_bev_tile_compact, dim3(nlists), dim3(1024), 0, stream, mind_ws, n, lists, (int)dz_bev_tile_list_words(batch, ho, wo));
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_bev_fill_empty_tiles(const int *list, int batch, int ho, int wo, const float *shift, int relu, int cout, float *out, int out_hp, int out_wp,
                            int out_cstride, int out_coff, const float *zero_resp, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(list && out && batch > 0 && ho > 0 && wo > 0 && cout > 0 && cout % 8 == 0 && out_cstride % 8 == 0 && out_coff % 8 == 0 &&
                 out_hp >= ho + 2 && out_wp >= wo + 2, "dz_bev_fill_empty_tiles: bad argument");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2 || math == DZ_MATH_F16, "dz_bev_fill_empty_tiles: math %d is not a split mode", math);
    const int ty = ceil_div(ho, 8), tx = ceil_div(wo, 32), n = batch * ty * tx;
    if (math == DZ_MATH_BF16X2)
        hipLaunchKernelGGL(k_bev_fill_tiles<MathBF16>, dim3(n), dim3(256), 0, stream, list, ty, tx, ho, wo, shift, relu, cout, out, out_hp, out_wp, out_cstride,
                           out_coff, zero_resp);
    else
        hipLaunchKernelGGL(k_bev_fill_tiles<MathF16>, dim3(n), dim3(256), 0, stream, list, ty, tx, ho, wo, shift, relu, cout, out, out_hp, out_wp, out_cstride,
                           out_coff, zero_resp);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_bev_row_index(const uint32_t *bitmap, const uint32_t *prefix, int batch, int d, int h, int w, int layout, int pad, int feat_rows, int *idx,
                     void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(bitmap && prefix && idx && batch > 0 && h > 0 && w > 0 && pad >= 0 && feat_rows >= 0, "dz_bev_row_index: bad argument");
    if (d != 2) { set_error("dz_bev_row_index: %d z slabs (the sparse-input convolution reads exactly 2)", d); return DZ_ERR_UNSUPPORTED; }
    DZ_CHECK_ARG(layout == DZ_LAYOUT_LINEAR || layout == DZ_LAYOUT_BRICK, "dz_bev_row_index: bad layout %d", layout);
    const LevelGeom lg = make_level(batch, d, h, w, layout);
    const long total = (long)batch * (h + 2 * pad) * (w + 2 * pad);
    hipLaunchKernelGGL(k_bev_row_index, dim3(stream_grid(total, 256)), dim3(256), 0, stream, bitmap, prefix, lg, pad, feat_rows, reinterpret_cast<int2 *>(idx));
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_pair16_from_f32(const float *src, long rows, int c_src, int c_dst, int math, float *dst, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0 && c_src >= 1 && c_dst >= c_src && c_dst % 8 == 0, "dz_pair16_from_f32: bad sizes");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_pair16_from_f32: math %d is not a split mode", math);
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(src && dst, "dz_pair16_from_f32: null pointer");
    const dim3 grid(stream_grid(rows * (c_dst / 8), 256));
    if (math == DZ_MATH_F16X2)
        hipLaunchKernelGGL(k_pair16_from_f32<MathF16>, grid, dim3(256), 0, stream, src, rows, c_src, c_dst, reinterpret_cast<uint4 *>(dst));
    else
        hipLaunchKernelGGL(k_pair16_from_f32<MathBF16>, grid, dim3(256), 0, stream, src, rows, c_src, c_dst, reinterpret_cast<uint4 *>(dst));
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_pair16_to_f32(const float *src, long rows, int c, int math, float *dst, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0 && c >= 8 && c % 8 == 0, "dz_pair16_to_f32: bad sizes");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_pair16_to_f32: math %d is not a split mode", math);
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(src && dst, "dz_pair16_to_f32: null pointer");
    const dim3 grid(stream_grid(rows * (c / 8), 256));
    if (math == DZ_MATH_F16X2)
        hipLaunchKernelGGL(k_pair16_to_f32<MathF16>, grid, dim3(256), 0, stream, reinterpret_cast<const uint4 *>(src), rows, c, dst);
    else
        hipLaunchKernelGGL(k_pair16_to_f32<MathBF16>, grid, dim3(256), 0, stream, reinterpret_cast<const uint4 *>(src), rows, c, dst);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_scatter_rows_split(const float *src, const int *rank, const int *d_n, int n_cap, int c_src, float *dst, int c_dst,
                          int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n_cap >= 0 && c_src >= 1 && c_dst >= c_src && c_dst % 8 == 0, "dz_scatter_rows_split: bad sizes");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2, "dz_scatter_rows_split: math %d is not a split mode", math);
    if (n_cap == 0) return DZ_OK;
    DZ_CHECK_ARG(src && rank && dst, "dz_scatter_rows_split: null pointer");
    const dim3 grid(stream_grid((long)n_cap * (c_dst / 8), 256));
    if (math == DZ_MATH_F16X2)
        hipLaunchKernelGGL(k_scatter_rows_split<MathF16>, grid, dim3(256), 0, stream, src, rank, d_n, n_cap, c_src,
                           reinterpret_cast<uint4 *>(dst), c_dst);
    else
        hipLaunchKernelGGL(k_scatter_rows_split<MathBF16>, grid, dim3(256), 0, stream, src, rank, d_n, n_cap, c_src,
                           reinterpret_cast<uint4 *>(dst), c_dst);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_sparse_to_bev_split(const float *feats, const int *coords, const int *d_m, int cap, int c, int d, int h, int w,
                           int pad, float *bev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(feats && coords && d_m && bev && c > 0 && c % 8 == 0 && d > 0 && (c * d) % 8 == 0 && pad >= 0,
                 "dz_sparse_to_bev_split: bad argument");
    if (cap == 0) return DZ_OK;
    hipLaunchKernelGGL(k_sparse_to_bev_split, dim3(stream_grid((long)cap * c, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned short *>(feats), coords, d_m, cap, c, d, h, w, pad,
                       reinterpret_cast<unsigned short *>(bev));
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_sparse_to_bev_split_dense(const float *feats, int feat_rows, const uint32_t *bitmap, const uint32_t *prefix, int batch, int c, int d,
                                 int h, int w, int layout, int pad, float *bev, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(feats && bitmap && prefix && bev && batch > 0 && c > 0 && c % 8 == 0 && h > 0 && w > 0 && pad >= 0 && feat_rows >= 0,
                 "dz_sparse_to_bev_split_dense: bad argument");
    if (d != 2) { set_error("dz_sparse_to_bev_split_dense: %d z slabs (the dense form interleaves exactly 2)", d); return DZ_ERR_UNSUPPORTED; }
    DZ_CHECK_ARG(layout == DZ_LAYOUT_LINEAR || layout == DZ_LAYOUT_BRICK, "dz_sparse_to_bev_split_dense: bad layout %d", layout);
    const LevelGeom lg = make_level(batch, d, h, w, layout);
    const long total = (long)batch * (h + 2 * pad) * (w + 2 * pad) * (c * 2 / 8);
    hipLaunchKernelGGL(k_sparse_to_bev_split_dense2, dim3(stream_grid(total, 256)), dim3(256), 0, stream,
                       reinterpret_cast<const uint2 *>(feats), bitmap, prefix, lg, c, pad, reinterpret_cast<uint4 *>(bev), feat_rows);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
