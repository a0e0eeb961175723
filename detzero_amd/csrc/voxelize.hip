// Points -> voxels on gfx950: hard voxelization (spconv Point2VoxelCPU3d semantics), MeanVFE and
// DynamicMeanVFE.  Reference call sites:
//   detection/detzero_det/datasets/processor/data_processor.py:61-91   (hard voxelizer)
//   detection/detzero_det/models/centerpoint_modules/vfe.py:66-83      (MeanVFE)
//   detection/detzero_det/models/centerpoint_modules/vfe.py:109-147    (DynamicMeanVFE)
//
// The sequential definitions ("first 5 points of a voxel in input order", "voxels numbered by
// first appearance", "no new voxel after max_voxels") are reproduced exactly without any sort:
//   1. one pass sets a bit per occupied cell; a bitmap scan ranks the cells (sparse_index.hip);
//   2. max_points rounds of atomicMin pick, per voxel, its r-th smallest point index;
//   3. a second bitmap over POINT indices marks each voxel's first point; its scan is the
//      first-appearance rank, which is the output row (and the max_voxels cut-off).
// All kernels stream the point buffer with coalesced loads; the bitmaps live in L2/Infinity Cache.
#include "hgemm.h"

namespace dz {

struct VoxGeom {
    float lo[3], vs[3], hi[3];
    int g[3];      // gx, gy, gz
    int xy_mask;   // also apply mask_points_by_range (lo <= x,y <= hi, inclusive; common_utils.py:247-250)
};

// c_j = floor((p_j - lo_j) / vs_j) with one IEEE rounding per operation (no FMA contraction, no
// reciprocal): identical to numpy/torch fp32 and to spconv's CPU loop.
__device__ __forceinline__ bool voxel_coord(const float *p, const VoxGeom &g, int &cx, int &cy, int &cz) {
    if (g.xy_mask && !(p[0] >= g.lo[0] && p[0] <= g.hi[0] && p[1] >= g.lo[1] && p[1] <= g.hi[1])) return false;
    const float fx = floorf(__fdiv_rn(__fsub_rn(p[0], g.lo[0]), g.vs[0]));
    const float fy = floorf(__fdiv_rn(__fsub_rn(p[1], g.lo[1]), g.vs[1]));
    const float fz = floorf(__fdiv_rn(__fsub_rn(p[2], g.lo[2]), g.vs[2]));
    // compare in float first so that huge/NaN values cannot overflow the int conversion
    if (!(fx >= 0.f && fx < (float)g.g[0] && fy >= 0.f && fy < (float)g.g[1] && fz >= 0.f && fz < (float)g.g[2]))
        return false;
    cx = (int)fx; cy = (int)fy; cz = (int)fz;
    return true;
}

// ---- hard voxelization ---------------------------------------------------------------------
// key (z-major) per point + occupancy bit
// (n_per = points per frame of a batch of equally long frames stored back to back; the frame index extends the key)
__global__ void k_hard_keys(const float *__restrict__ pts, int n, int c, VoxGeom g, int n_per, uint32_t cells,
                            uint32_t *__restrict__ keys, uint32_t *__restrict__ bitmap) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *p = pts + (size_t)i * c;
        const float xyz[3] = {p[0], p[1], p[2]};
        int cx, cy, cz;
        uint32_t key = KEY_INVALID;
        if (voxel_coord(xyz, g, cx, cy, cz)) {
            key = (uint32_t)(i / n_per) * cells + (uint32_t)((cz * g.g[1] + cy) * g.g[0] + cx);
            atomicOr(&bitmap[key >> 5], 1u << (key & 31u));
        }
        keys[i] = key;
    }
}

// "the first max_points points of a voxel, in point order" = the max_points smallest point indices of the voxel.  One pass: every
// point offers its index to the voxel's slot 0 with atomicMin and carries the LARGER of (what was there, what it offered) on to
// slot 1, and so on - an insertion network whose result does not depend on the interleaving: an atomicMin conserves the multiset
// {slot, carried value}, every index is offered to slot 0, so slot 0 ends as the minimum and everything else has been offered to
// slot 1, and by induction slot q holds the (q+1)-th smallest.  An empty slot (0x7f7f7f7f) ends a point's walk: typically one or
// two atomics per point instead of the max_points full passes (rank lookup + atomicMin each) of the round-by-round version.
// Points arrive in scan order, so the lanes of a wavefront often hold runs of points of ONE voxel (their indices ascending): a
// point with p run-mates before it has p smaller indices in its voxel and can never sit above slot p - it starts its walk there
// (the induction still holds: the value of overall rank r starts at a slot <= r, and slot q keeps the minimum of what reaches
// it), and a point with max_points run-mates before it is dropped at once.
__global__ void k_hard_insert(const uint32_t *__restrict__ keys, int n, const uint32_t *__restrict__ bitmap,
                              const uint32_t *__restrict__ prefix, int *__restrict__ mins, int cap, int max_points) {
    const int lane = threadIdx.x & 63;
    const int n_pad = (n + 63) & ~63;               // whole wavefronts run the same trip count (the run positions use a ballot)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        const uint32_t key = i < n ? keys[i] : KEY_INVALID;
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64);
        const unsigned long long heads = __ballot(lane == 0 || prev != key);
        const int first = 63 - __clzll(heads & (~0ull >> (63 - lane)));      // first lane of this lane's run
        const int q0 = lane - first;
        if (key == KEY_INVALID || q0 >= max_points) continue;
        const int v = bitmap_rank(bitmap, prefix, key);
        int val = i;
        for (int q = q0; q < max_points; ++q) {
            const int old = atomicMin(&mins[(size_t)q * cap + v], val);
            if (old == 0x7f7f7f7f) break;
            val = max(old, val);
        }
    }
}

__global__ void k_hard_mark_first(const int *__restrict__ min0, const int *__restrict__ d_m, int cap,
                                  uint32_t *__restrict__ pt_bitmap) {
    const int m = min(*d_m, cap);
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < m; v += gridDim.x * blockDim.x) {
        const uint32_t i = (uint32_t)min0[v];
        atomicOr(&pt_bitmap[i >> 5], 1u << (i & 31u));
    }
}

// one thread per (voxel, slot): copies the slot's point (or zeros) into the output row given by
// the voxel's first-appearance rank
__global__ void k_hard_emit(const float *__restrict__ pts, int c, const int *__restrict__ mins, int cap,
                            int max_points, int max_voxels, const int *__restrict__ d_m,
                            const int *__restrict__ canon_coords, const uint32_t *__restrict__ pt_bitmap,
                            const uint32_t *__restrict__ pt_prefix, float *__restrict__ voxels,
                            int *__restrict__ coords_zyx, int *__restrict__ num_points,
                            int *__restrict__ d_num_voxels) {
    const int m = min(*d_m, cap);
    if (blockIdx.x == 0 && threadIdx.x == 0) *d_num_voxels = min(m, max_voxels);
    const long total = (long)m * max_points;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / max_points), r = (int)(idx % max_points);
        const int first = mins[v];
        const int row = bitmap_rank(pt_bitmap, pt_prefix, (uint32_t)first);
        if (row >= max_voxels) continue;
        const int pi = mins[(size_t)r * cap + v];
        float *dst = voxels + ((size_t)row * max_points + r) * c;
        if (pi != 0x7f7f7f7f) {
            const float *src = pts + (size_t)pi * c;
            for (int j = 0; j < c; ++j) dst[j] = src[j];
        } else {
            for (int j = 0; j < c; ++j) dst[j] = 0.f;
        }
        if (r == 0) {
            const int4 cc = reinterpret_cast<const int4 *>(canon_coords)[v];  // [b,z,y,x]
            coords_zyx[row * 3 + 0] = cc.y;
            coords_zyx[row * 3 + 1] = cc.z;
            coords_zyx[row * 3 + 2] = cc.w;
            int cnt = 0;
            for (int q = 0; q < max_points; ++q) cnt += (mins[(size_t)q * cap + v] != 0x7f7f7f7f);
            num_points[row] = cnt;
        }
    }
}

// Fused emit + MeanVFE (vfe.py:58-83 on the voxels of data_processor.py:47-95) for the batched frame pipeline:
// one thread per (voxel, channel) writes the mean of the voxel's (<= max_points) first points straight into the
// voxel's first-appearance row of a batch buffer, and [batch_index, z, y, x] into the coordinate buffer; the
// (max_voxels, max_points, C) tensor of the reference never exists.  Same summation order as k_mean_vfe.
__global__ void k_hard_emit_mean(const float *__restrict__ pts, int c, const int *__restrict__ mins, int cap,
                                 int max_points, int max_voxels, const int *__restrict__ d_m,
                                 const int *__restrict__ canon_coords, const uint32_t *__restrict__ pt_bitmap,
                                 const uint32_t *__restrict__ pt_prefix, int batch_index, int batch, int n_per, int cap_frame,
                                 float *__restrict__ feats, int c_stride, int *__restrict__ coords_bzyx,
                                 int *__restrict__ d_num_voxels) {
    // batch frames stored back to back (n_per points each): the first-appearance rank is global (frame-major); frame f's
    // voxels are the ranks [R_f, R_f+1) with R_f = number of first points below point index f*n_per, and go to rows
    // f*cap_frame + (rank - R_f) of the batch buffers, cut at max_voxels per frame
    const int m = min(*d_m, cap);
    if (blockIdx.x == 0 && (int)threadIdx.x < batch) {
        const int f = threadIdx.x;
        const int lo = bitmap_rank(pt_bitmap, pt_prefix, (uint32_t)f * (uint32_t)n_per);
        const int hi = f + 1 < batch ? bitmap_rank(pt_bitmap, pt_prefix, (uint32_t)(f + 1) * (uint32_t)n_per) : m;
        d_num_voxels[f] = min(hi - lo, max_voxels);
    }
    const long total = (long)m * c_stride;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / c_stride), ch = (int)(idx % c_stride);
        const int4 cc = reinterpret_cast<const int4 *>(canon_coords)[v];  // [frame,z,y,x]
        const int rank = bitmap_rank(pt_bitmap, pt_prefix, (uint32_t)mins[v]) -
                         bitmap_rank(pt_bitmap, pt_prefix, (uint32_t)cc.x * (uint32_t)n_per);
        if (rank >= max_voxels) continue;
        const size_t row = (size_t)cc.x * cap_frame + rank;
        float sum = 0.f;
        int cnt = 0;
        for (int q = 0; q < max_points; ++q) {
            const int pi = mins[(size_t)q * cap + v];
            if (pi != 0x7f7f7f7f) {
                ++cnt;
                if (ch < c) sum = __fadd_rn(sum, pts[(size_t)pi * c + ch]);
            }
        }
        feats[row * c_stride + ch] = ch < c ? __fdiv_rn(sum, fmaxf((float)cnt, 1.0f)) : 0.f;
        if (ch == 0) reinterpret_cast<int4 *>(coords_bzyx)[row] = make_int4(batch_index + cc.x, cc.y, cc.z, cc.w);
    }
}

// ---- voxelize straight into the level-1 sparse index (batched frame pipeline) ---------------------------------
// When a frame cannot exceed max_voxels (points per frame <= max_voxels), no voxel is ever dropped, so the occupancy
// bitmap of the hard voxelizer IS the level-1 index of the sparse backbone: keys are laid out in the LEVEL's geometry
// ((b*D + z)*H + y)*W + x, the scan writes the level's prefix / canonical coordinates / count, and the voxel means go
// to the voxel's canonical row directly (zero-padded to the backbone's input width, optionally as pair16).  The
// first-appearance ordering, the voxel list, dz_index_from_coords and dz_scatter_rows all disappear.
// line_flags (optional): one byte per 32 bitmap words (a 128-byte line), set for every line that receives a bit - the scan that
// follows skips the lines (> 75 % of a level-1 grid) that were never touched.
// Atomics on the bitmap resolve at the memory side and bound this kernel; points arrive in scan order, so the lanes of a
// wavefront hold runs of points of the same bitmap word: the run's bits are ORed together in registers (segmented scan) and its
// last lane issues ONE atomic (half the atomics on lidar-ordered points; any order stays correct).
__global__ void k_level_keys(const float *__restrict__ pts, int n, int c, VoxGeom g, int n_per, LevelGeom lg,
                             uint32_t *__restrict__ keys, uint32_t *__restrict__ bitmap, unsigned char *__restrict__ line_flags) {
    const int lane = threadIdx.x & 63;
    const int n_pad = (n + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        uint32_t key = KEY_INVALID;
        if (i < n) {
            const float *p = pts + (size_t)i * c;
            const float xyz[3] = {p[0], p[1], p[2]};
            int cx, cy, cz;
            if (voxel_coord(xyz, g, cx, cy, cz)) key = lg.key(i / n_per, cz, cy, cx);
            keys[i] = key;
        }
        const uint32_t w = key == KEY_INVALID ? KEY_INVALID : key >> 5;
        uint32_t bits = key == KEY_INVALID ? 0u : 1u << (key & 31u);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t wp = (uint32_t)__shfl_up((int)w, d, 64), bp = (uint32_t)__shfl_up((int)bits, d, 64);
            if (lane >= d && wp == w) bits |= bp;
        }
        const uint32_t wn = (uint32_t)__shfl_down((int)w, 1, 64);
        if (bits && (lane == 63 || wn != w)) {
            atomicOr(&bitmap[w], bits);
            if (line_flags) line_flags[w >> 5] = 1;
        }
    }
}

// one thread per (voxel row, group of 8 output channels)
template <int MATH>
__global__ void k_level_emit_mean(const float *__restrict__ pts, int c, const int *__restrict__ mins, int cap, int max_points,
                                  const int *__restrict__ d_m, float *__restrict__ feats, int c_dst) {
    const int m = min(*d_m, cap);
    const int groups = c_dst / 8;
    const long total = (long)m * groups;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / groups), gq = (int)(idx % groups);
        float sum[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = 0.f;
        int cnt = 0;
        // (a group of channel padding - every group behind the first at 5 point features - is zeros: it reads nothing)
        for (int q = 0; q < (gq * 8 < c ? max_points : 0); ++q) {
            const int pi = mins[(size_t)q * cap + v];
            if (pi != 0x7f7f7f7f) {
                ++cnt;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (gq * 8 + e < c) sum[e] = __fadd_rn(sum[e], pts[(size_t)pi * c + gq * 8 + e]);
            }
        }
        const float nrm = fmaxf((float)cnt, 1.0f);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum[e] = gq * 8 + e < c ? __fdiv_rn(sum[e], nrm) : 0.f;
        float *dst = feats + ((size_t)v * c_dst + gq * 8);
        if (MATH == 0) {
            reinterpret_cast<float4 *>(dst)[0] = make_float4(sum[0], sum[1], sum[2], sum[3]);
            reinterpret_cast<float4 *>(dst)[1] = make_float4(sum[4], sum[5], sum[6], sum[7]);
        } else {
            using M = typename std::conditional<MATH == 1, MathF16, MathBF16>::type;
            const float a[4] = {sum[0], sum[1], sum[2], sum[3]}, b[4] = {sum[4], sum[5], sum[6], sum[7]};
            uint2 h0, l0, h1, l1;
            split4<M>(a, h0, l0);
            split4<M>(b, h1, l1);
            reinterpret_cast<uint4 *>(dst)[0] = make_uint4(h0.x, h0.y, h1.x, h1.y);
            reinterpret_cast<uint4 *>(dst)[1] = make_uint4(l0.x, l0.y, l1.x, l1.y);
        }
    }
}

// ---- MeanVFE ---------------------------------------------------------------------------------
__global__ void k_mean_vfe(const float *__restrict__ voxels, const int *__restrict__ num_points,
                           const int *__restrict__ d_m, int cap, int max_points, int c, float *__restrict__ out,
                           int c_out) {
    const int m = d_m ? min(*d_m, cap) : cap;
    const long total = (long)m * c_out;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / c_out), ch = (int)(idx % c_out);
        float r = 0.f;
        if (ch < c) {
            float s = 0.f;
            for (int q = 0; q < max_points; ++q) s = __fadd_rn(s, voxels[((size_t)v * max_points + q) * c + ch]);
            const float nrm = fmaxf((float)num_points[v], 1.0f);
            r = __fdiv_rn(s, nrm);
        }
        out[idx] = r;
    }
}

// ---- dynamic voxelization ----------------------------------------------------------------------
// (bitmap atomics merged per run of equal words over the wavefront, as in k_level_keys)
__global__ void k_dyn_keys(const float *__restrict__ pts, int n, int stride, VoxGeom g, int batch,
                           uint32_t *__restrict__ keys, uint32_t *__restrict__ bitmap) {
    const int lane = threadIdx.x & 63;
    const int n_pad = (n + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        uint32_t key = KEY_INVALID;
        if (i < n) {
            const float *p = pts + (size_t)i * stride;
            const float xyz[3] = {p[1], p[2], p[3]};
            int cx, cy, cz;
            const int b = (int)p[0];
            if ((unsigned)b < (unsigned)batch && voxel_coord(xyz, g, cx, cy, cz))
                key = (uint32_t)(((b * g.g[0] + cx) * g.g[1] + cy) * g.g[2] + cz);
            keys[i] = key;
        }
        const uint32_t w = key == KEY_INVALID ? KEY_INVALID : key >> 5;
        uint32_t bits = key == KEY_INVALID ? 0u : 1u << (key & 31u);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t wp = (uint32_t)__shfl_up((int)w, d, 64), bp = (uint32_t)__shfl_up((int)bits, d, 64);
            if (lane >= d && wp == w) bits |= bp;
        }
        const uint32_t wn = (uint32_t)__shfl_down((int)w, 1, 64);
        if (bits && (lane == 63 || wn != w)) atomicOr(&bitmap[w], bits);
    }
}

// Deterministic sums: every value is added as a 64-bit fixed-point number (2^-28 units: exact for |v| >= 2^-4 at fp32 precision,
// 3.7e-9 absolute below; |v| * points per voxel up to 2^35), and integer addition commutes - the result does not depend on the
// order the atomics land in (fp32 atomics, as torch_scatter uses them in the reference, give run-to-run differences).
constexpr float DYN_FIX = 268435456.f;        // 2^28

// One thread per point.  Points arrive in scan order: the lanes of a wavefront hold runs of points of one voxel, whose fixed-point
// values are summed in registers (segmented scan; integer addition, so the result is the same in any grouping) - the run's last
// lane ranks the voxel and issues ONE atomic per channel and one for the count.
__global__ void k_dyn_accumulate(const float *__restrict__ pts, int n, int c, const uint32_t *__restrict__ keys,
                                 const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ prefix, int cap,
                                 unsigned long long *__restrict__ acc, int *__restrict__ counts) {
    const int lane = threadIdx.x & 63;
    const int n_pad = (n + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        const uint32_t key = i < n ? keys[i] : KEY_INVALID;
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64), next = (uint32_t)__shfl_down((int)key, 1, 64);
        const unsigned long long heads = __ballot(lane == 0 || prev != key);
        const int first = 63 - __clzll(heads & (~0ull >> (63 - lane)));           // first lane of this lane's run (runs are contiguous)
        const int run = lane - first + 1;                                         // points of the run up to this lane
        const bool last = key != KEY_INVALID && (lane == 63 || next != key);
        int v = cap;
        if (last) v = bitmap_rank(bitmap, prefix, key);
        int steps = 0;                                  // scan steps the longest run of this wavefront needs (runs of lidar points are short)
        while (steps < 6 && __ballot(lane - (1 << steps) >= first) != 0ull) ++steps;
        for (int ch = 0; ch < c; ++ch) {
            long long q = key != KEY_INVALID ? __float2ll_rn(pts[(size_t)i * (c + 1) + 1 + ch] * DYN_FIX) : 0ll;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j >= steps) break;                  // (wave-uniform: no run of this wavefront reaches back 2^j lanes)
                const int lo = __shfl_up((int)(unsigned int)q, 1 << j, 64), hi = __shfl_up((int)(q >> 32), 1 << j, 64);
                if (lane - (1 << j) >= first) q += (long long)(((unsigned long long)(unsigned int)hi << 32) | (unsigned int)lo);
            }
            // two's complement: wrap-around addition is signed addition
            if (v < cap) atomicAdd(&acc[(size_t)v * c + ch], (unsigned long long)q);
        }
        if (v < cap) atomicAdd(&counts[v], run);
    }
}

__global__ void k_dyn_divide(const unsigned long long *__restrict__ acc, float *__restrict__ feats, const int *__restrict__ counts,
                             const int *__restrict__ d_m, int cap, int c) {
    const int m = min(*d_m, cap);
    const long total = (long)m * c;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / c);
        const double sum = (double)(long long)acc[idx] * (1.0 / (double)DYN_FIX);
        feats[idx] = __fdiv_rn((float)sum, (float)counts[v]);
    }
}

static bool make_geom(const float *r6, const float *vs3, const int *g3, int xy_mask, VoxGeom &g) {
    g.xy_mask = xy_mask;
    for (int i = 0; i < 3; ++i) {
        g.lo[i] = r6[i]; g.hi[i] = r6[3 + i]; g.vs[i] = vs3[i]; g.g[i] = g3[i];
        if (!(g.vs[i] > 0.f) || g.g[i] < 1) return false;
    }
    return true;
}

struct HardWs {
    uint32_t *keys, *bitmap, *prefix, *pt_bitmap, *pt_prefix;
    int *mins, *canon_coords, *d_m;
    void *scan_ws;
    size_t scan_ws_bytes, total;
};
static HardWs carve_hard(void *ws, int n, size_t nwords, int max_points) {
    HardWs w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    char *base = (char *)ws;
    const size_t pt_words = align_up(((size_t)n + 31) / 32, 8);
    size_t o_keys = take((size_t)n * 4), o_bm = take(nwords * 4), o_pf = take(nwords * 4);
    size_t o_pb = take(pt_words * 4), o_pp = take(pt_words * 4);
    size_t o_min = take((size_t)max_points * n * 4), o_cc = take((size_t)n * 16), o_dm = take(256);
    w.scan_ws_bytes = bitmap_scan_workspace_bytes(nwords > pt_words ? nwords : pt_words);
    size_t o_sw = take(w.scan_ws_bytes);
    w.total = off;
    if (base) {
        w.keys = (uint32_t *)(base + o_keys); w.bitmap = (uint32_t *)(base + o_bm); w.prefix = (uint32_t *)(base + o_pf);
        w.pt_bitmap = (uint32_t *)(base + o_pb); w.pt_prefix = (uint32_t *)(base + o_pp);
        w.mins = (int *)(base + o_min); w.canon_coords = (int *)(base + o_cc); w.d_m = (int *)(base + o_dm);
        w.scan_ws = base + o_sw;
    }
    return w;
}

}  // namespace dz

using namespace dz;

extern "C" {

size_t dz_voxelize_hard_workspace_bytes(int n, int gx, int gy, int gz, int max_points) {
    if (n < 1) n = 1;
    return carve_hard(nullptr, n, dz_index_words(1, gz, gy, gx, DZ_LAYOUT_LINEAR), max_points).total;
}

// shared driver of the two emit flavours: feats != nullptr selects the fused emit + mean
static int voxelize_hard_impl(const float *points, int n, int c, const float *h_range6, const float *h_vsize3,
                              const int *h_grid3, int xy_range_mask, int max_points, int max_voxels, float *voxels,
                              int *coords_zyx, int *num_points, int batch_index, int batch, int cap_frame, float *feats,
                              int c_stride, int *coords_bzyx, int *d_num_voxels, void *ws, size_t ws_bytes, hipStream_t stream) {
    // n = points of ALL `batch` frames (equally long, back to back)
    VoxGeom g;
    DZ_CHECK_ARG(make_geom(h_range6, h_vsize3, h_grid3, xy_range_mask, g), "dz_voxelize_hard: bad geometry");
    const size_t cells = (size_t)g.g[0] * g.g[1] * g.g[2];
    if (cells * (size_t)batch >= 0xFFFFFFFFull) { set_error("dz_voxelize_hard: grid x batch too large for 32-bit keys"); return DZ_ERR_UNSUPPORTED; }
    if (n == 0) return fill_u32(d_num_voxels, 0u, (size_t)batch, stream);
    DZ_CHECK_ARG(points, "dz_voxelize_hard: null points");
    const int n_per = n / batch;
    const size_t nwords = dz_index_words(batch, g.g[2], g.g[1], g.g[0], DZ_LAYOUT_LINEAR);
    HardWs w = carve_hard(ws, n, nwords, max_points);
    if (ws_bytes < w.total) { set_error("dz_voxelize_hard: workspace %zu < %zu", ws_bytes, w.total); return DZ_ERR_WORKSPACE; }
    const size_t pt_words = align_up(((size_t)n + 31) / 32, 8);
    const int cap = n;  // a frame of n points opens at most n voxels

    int rc = fill_u32(w.bitmap, 0u, nwords, stream);
    if (!rc) rc = fill_u32(w.pt_bitmap, 0u, pt_words, stream);
    if (!rc) rc = fill_u32(w.mins, 0x7f7f7f7fu, (size_t)max_points * cap, stream);
    if (rc) return rc;
    const int grid_n = stream_grid(n, 256);
    hipLaunchKernelGGL(k_hard_keys, dim3(grid_n), dim3(256), 0, stream, points, n, c, g, n_per, (uint32_t)cells, w.keys, w.bitmap);
    rc = bitmap_scan(w.bitmap, nwords, w.prefix, w.d_m, 0, ScanDims{g.g[2], g.g[1], g.g[0]}, w.canon_coords, cap,
                         w.scan_ws, w.scan_ws_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_hard_insert, dim3(grid_n), dim3(256), 0, stream, w.keys, n, w.bitmap, w.prefix, w.mins, cap, max_points);
    hipLaunchKernelGGL(k_hard_mark_first, dim3(grid_n), dim3(256), 0, stream, w.mins, w.d_m, cap, w.pt_bitmap);
    // scan over the point-index bitmap: rank of a voxel's first point = first-appearance voxel id.
    // d_num_voxels is used as a scratch total here and overwritten by k_hard_emit.
    rc = bitmap_scan(w.pt_bitmap, pt_words, w.pt_prefix, d_num_voxels, -1, ScanDims{1, 1, 1}, nullptr, 0, w.scan_ws,
                     w.scan_ws_bytes, stream);
    if (rc) return rc;
    if (feats)
        hipLaunchKernelGGL(k_hard_emit_mean, dim3(stream_grid((long)cap * c_stride, 256)), dim3(256), 0, stream, points, c, w.mins,
                           cap, max_points, max_voxels, w.d_m, w.canon_coords, w.pt_bitmap, w.pt_prefix, batch_index, batch, n_per,
                           cap_frame, feats, c_stride, coords_bzyx, d_num_voxels);
    else
        hipLaunchKernelGGL(k_hard_emit, dim3(stream_grid((long)cap * max_points, 256)), dim3(256), 0, stream, points, c,
                           w.mins, cap, max_points, max_voxels, w.d_m, w.canon_coords, w.pt_bitmap, w.pt_prefix, voxels,
                           coords_zyx, num_points, d_num_voxels);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_voxelize_hard(const float *points, int n, int c, const float *h_range6, const float *h_vsize3,
                     const int *h_grid3, int xy_range_mask, int max_points, int max_voxels, float *voxels, int *coords_zyx,
                     int *num_points, int *d_num_voxels, void *ws, size_t ws_bytes, void *stream_) {
    DZ_CHECK_ARG(n >= 0 && c >= 3 && max_points >= 1 && max_voxels >= 1, "dz_voxelize_hard: bad sizes");
    DZ_CHECK_ARG(voxels && coords_zyx && num_points && d_num_voxels && ws, "dz_voxelize_hard: null pointer");
    return voxelize_hard_impl(points, n, c, h_range6, h_vsize3, h_grid3, xy_range_mask, max_points, max_voxels, voxels, coords_zyx,
                              num_points, 0, 1, 0, nullptr, 0, nullptr, d_num_voxels, ws, ws_bytes, (hipStream_t)stream_);
}

int dz_voxelize_hard_mean(const float *points, int n, int c, const float *h_range6, const float *h_vsize3,
                          const int *h_grid3, int xy_range_mask, int max_points, int max_voxels, int batch_index, float *feats,
                          int c_stride, int *coords_bzyx, int *d_num_voxels, void *ws, size_t ws_bytes, void *stream_) {
    DZ_CHECK_ARG(n >= 0 && c >= 3 && max_points >= 1 && max_voxels >= 1 && c_stride >= c, "dz_voxelize_hard_mean: bad sizes");
    DZ_CHECK_ARG(feats && coords_bzyx && d_num_voxels && ws, "dz_voxelize_hard_mean: null pointer");
    return voxelize_hard_impl(points, n, c, h_range6, h_vsize3, h_grid3, xy_range_mask, max_points, max_voxels, nullptr, nullptr,
                              nullptr, batch_index, 1, 0, feats, c_stride, coords_bzyx, d_num_voxels, ws, ws_bytes, (hipStream_t)stream_);
}

size_t dz_voxelize_hard_batched_workspace_bytes(int n_per_frame, int batch, int gx, int gy, int gz, int max_points) {
    const long n = (long)(n_per_frame < 1 ? 1 : n_per_frame) * (batch < 1 ? 1 : batch);
    return carve_hard(nullptr, (int)n, dz_index_words(batch < 1 ? 1 : batch, gz, gy, gx, DZ_LAYOUT_LINEAR), max_points).total;
}

int dz_voxelize_hard_mean_batched(const float *points, int n_per_frame, int batch, int c, const float *h_range6,
                                  const float *h_vsize3, const int *h_grid3, int xy_range_mask, int max_points, int max_voxels,
                                  float *feats, int c_stride, int *coords_bzyx, int cap_per_frame, int *d_num_voxels, void *ws,
                                  size_t ws_bytes, void *stream_) {
    DZ_CHECK_ARG(n_per_frame >= 0 && batch >= 1 && batch <= 256 && c >= 3 && max_points >= 1 && max_voxels >= 1 && c_stride >= c,
                 "dz_voxelize_hard_mean_batched: bad sizes");
    DZ_CHECK_ARG((long)n_per_frame * batch < 0x7FFFFFFFl, "dz_voxelize_hard_mean_batched: too many points");
    DZ_CHECK_ARG(cap_per_frame >= (max_voxels < n_per_frame ? max_voxels : n_per_frame),
                 "dz_voxelize_hard_mean_batched: cap_per_frame below min(max_voxels, points per frame)");
    DZ_CHECK_ARG(feats && coords_bzyx && d_num_voxels && ws, "dz_voxelize_hard_mean_batched: null pointer");
    return voxelize_hard_impl(points, n_per_frame * batch, c, h_range6, h_vsize3, h_grid3, xy_range_mask, max_points, max_voxels,
                              nullptr, nullptr, nullptr, 0, batch, cap_per_frame, feats, c_stride, coords_bzyx, d_num_voxels, ws,
                              ws_bytes, (hipStream_t)stream_);
}

static size_t level_ws_layout(long n, int max_points, int cap, size_t nwords, size_t *o_min, size_t *o_sw, size_t *sw_bytes, size_t *o_lf,
                              size_t *lf_words) {
    size_t off = align_up((size_t)(n < 1 ? 1 : n) * 4, 256);                 // keys first
    *o_min = off;
    off += align_up((size_t)max_points * (cap < 1 ? 1 : cap) * 4, 256);
    *sw_bytes = bitmap_scan_workspace_bytes(nwords);
    *o_sw = off;
    off += align_up(*sw_bytes, 256);
    *lf_words = align_up((nwords + 31) / 32, 16) / 4;                         // one byte per 32-word line, as whole 16-byte units
    *o_lf = off;
    off += align_up(*lf_words * 4, 256);
    return off;
}

size_t dz_voxelize_to_level_workspace_bytes(int n_per_frame, int batch, int max_points, int cap, int d, int h, int w, int layout) {
    size_t a, b, c, e, f;
    return level_ws_layout((long)n_per_frame * batch, max_points, cap, dz_index_words(batch, d, h, w, layout), &a, &b, &c, &e, &f);
}

int dz_voxelize_to_level(const float *points, int n_per_frame, int batch, int c, const float *h_range6, const float *h_vsize3,
                         const int *h_grid3, int xy_range_mask, int max_points, int max_voxels, int level_d, int layout, uint32_t *bitmap,
                         uint32_t *prefix, int *coords_out, int *d_m, int cap, float *feats, int c_dst, int math, void *ws,
                         size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n_per_frame >= 0 && batch >= 1 && c >= 3 && max_points >= 1 && c_dst >= c && c_dst % 8 == 0,
                 "dz_voxelize_to_level: bad sizes");
    DZ_CHECK_ARG(math >= 0 && math <= 2, "dz_voxelize_to_level: bad math mode %d", math);
    DZ_CHECK_ARG(bitmap && prefix && coords_out && d_m && feats && ws, "dz_voxelize_to_level: null pointer");
    VoxGeom g;
    DZ_CHECK_ARG(make_geom(h_range6, h_vsize3, h_grid3, xy_range_mask, g), "dz_voxelize_to_level: bad geometry");
    DZ_CHECK_ARG(level_d >= g.g[2], "dz_voxelize_to_level: level depth %d below the voxel grid's %d", level_d, g.g[2]);
    if (n_per_frame > max_voxels) {
        set_error("dz_voxelize_to_level: %d points per frame could exceed max_voxels %d - use dz_voxelize_hard_mean_batched",
                  n_per_frame, max_voxels);
        return DZ_ERR_UNSUPPORTED;
    }
    const long n = (long)n_per_frame * batch;
    DZ_CHECK_ARG(n < 0x7FFFFFFFl && cap >= (n < 1 ? 1 : 0), "dz_voxelize_to_level: too many points");
    DZ_CHECK_ARG(layout == DZ_LAYOUT_LINEAR || layout == DZ_LAYOUT_BRICK, "dz_voxelize_to_level: bad layout %d", layout);
    const LevelGeom lg = make_level(batch, level_d, g.g[1], g.g[0], layout);
    if (lg.cells() >= 0xFFFFFFFFull) { set_error("dz_voxelize_to_level: grid x batch too large for 32-bit keys"); return DZ_ERR_UNSUPPORTED; }
    const size_t nwords = dz_index_words(batch, level_d, g.g[1], g.g[0], layout);
    size_t o_min, o_sw, sw_bytes, o_lf, lf_words;
    const size_t need = level_ws_layout(n, max_points, cap, nwords, &o_min, &o_sw, &sw_bytes, &o_lf, &lf_words);
    if (ws_bytes < need) { set_error("dz_voxelize_to_level: workspace %zu < %zu", ws_bytes, need); return DZ_ERR_WORKSPACE; }
    uint32_t *keys = (uint32_t *)ws;
    int *mins = (int *)((char *)ws + o_min);
    static const int use_flags = getenv("DZ_TUNE_LINE_FLAGS") ? atoi(getenv("DZ_TUNE_LINE_FLAGS")) : 1;      // development knob
    unsigned char *line_flags = use_flags ? (unsigned char *)ws + o_lf : nullptr;
    int rc = fill_u32(bitmap, 0u, nwords, stream);
    if (!rc) rc = fill_u32(mins, 0x7f7f7f7fu, (size_t)max_points * cap, stream);
    if (!rc && line_flags) rc = fill_u32(line_flags, 0u, lf_words, stream);
    if (rc) return rc;
    if (n == 0) return level_scan(bitmap, lg, prefix, d_m, coords_out, cap, (char *)ws + o_sw, sw_bytes, stream, true, line_flags);
    DZ_CHECK_ARG(points, "dz_voxelize_to_level: null points");
    const int grid_n = stream_grid(n, 256);
    hipLaunchKernelGGL(k_level_keys, dim3(grid_n), dim3(256), 0, stream, points, (int)n, c, g, n_per_frame, lg, keys, bitmap,
                       line_flags);
    rc = level_scan(bitmap, lg, prefix, d_m, coords_out, cap, (char *)ws + o_sw, sw_bytes, stream, true, line_flags);
    if (rc) return rc;
    hipLaunchKernelGGL(k_hard_insert, dim3(grid_n), dim3(256), 0, stream, keys, (int)n, bitmap, prefix, mins, cap, max_points);
    const dim3 ge(stream_grid((long)cap * (c_dst / 8), 256));
    if (math == 0)
        hipLaunchKernelGGL(k_level_emit_mean<0>, ge, dim3(256), 0, stream, points, c, mins, cap, max_points, d_m, feats, c_dst);
    else if (math == 1)
        hipLaunchKernelGGL(k_level_emit_mean<1>, ge, dim3(256), 0, stream, points, c, mins, cap, max_points, d_m, feats, c_dst);
    else
        hipLaunchKernelGGL(k_level_emit_mean<2>, ge, dim3(256), 0, stream, points, c, mins, cap, max_points, d_m, feats, c_dst);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_mean_vfe(const float *voxels, const int *num_points, const int *d_m, int cap, int max_points, int c, float *out,
                int c_out_stride, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(voxels && num_points && out && c >= 1 && c_out_stride >= c && max_points >= 1, "dz_mean_vfe: bad argument");
    if (cap == 0) return DZ_OK;
    hipLaunchKernelGGL(k_mean_vfe, dim3(stream_grid((long)cap * c_out_stride, 256)), dim3(256), 0, stream, voxels,
                       num_points, d_m, cap, max_points, c, out, c_out_stride);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

static size_t dyn_layout(int n, size_t nwords, int cap, int c, size_t *o_keys, size_t *o_bm, size_t *o_pf, size_t *o_cnt,
                         size_t *o_acc, size_t *o_sw, size_t *sw_bytes) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    *o_keys = take((size_t)(n < 1 ? 1 : n) * 4);
    *o_bm = take(nwords * 4);
    *o_pf = take(nwords * 4);
    *o_cnt = take((size_t)(cap < 1 ? 1 : cap) * 4);
    *o_acc = take((size_t)(cap < 1 ? 1 : cap) * (c < 1 ? 1 : c) * 8);        // fixed-point sums
    *sw_bytes = bitmap_scan_workspace_bytes(nwords);
    *o_sw = take(*sw_bytes);
    return off;
}

size_t dz_voxelize_dynamic_workspace_bytes(int n, int batch, int gx, int gy, int gz, int c, int cap) {
    size_t a, b, d, e, f, g, h;
    return dyn_layout(n, dz_index_words(batch, gx, gy, gz, DZ_LAYOUT_LINEAR), cap, c, &a, &b, &d, &e, &h, &f, &g);
}

int dz_voxelize_dynamic_mean(const float *points_b, int n, int c, const float *h_range6, const float *h_vsize3,
                             const int *h_grid3, int xy_range_mask, int batch, float *feats, int *coords_bzyx,
                             int *d_num_voxels, int cap,
                             void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 0 && c >= 3 && batch >= 1 && cap >= 0, "dz_voxelize_dynamic_mean: bad sizes");
    DZ_CHECK_ARG(feats && coords_bzyx && d_num_voxels && ws, "dz_voxelize_dynamic_mean: null pointer");
    VoxGeom g;
    DZ_CHECK_ARG(make_geom(h_range6, h_vsize3, h_grid3, xy_range_mask, g), "dz_voxelize_dynamic_mean: bad geometry");
    const size_t cells = (size_t)batch * g.g[0] * g.g[1] * g.g[2];
    // the reference's int32 merge key overflows for b >= 24 on the Waymo grid (vfe.py:128-131); we refuse instead
    if (cells >= 0x7FFFFFFFull) { set_error("dz_voxelize_dynamic_mean: batch*grid exceeds int32 merge keys"); return DZ_ERR_UNSUPPORTED; }
    const size_t nwords = dz_index_words(batch, g.g[0], g.g[1], g.g[2], DZ_LAYOUT_LINEAR);
    size_t o_keys, o_bm, o_pf, o_cnt, o_acc, o_sw, sw_bytes;
    const size_t need = dyn_layout(n, nwords, cap, c, &o_keys, &o_bm, &o_pf, &o_cnt, &o_acc, &o_sw, &sw_bytes);
    if (ws_bytes < need) { set_error("dz_voxelize_dynamic_mean: workspace %zu < %zu", ws_bytes, need); return DZ_ERR_WORKSPACE; }
    char *base = (char *)ws;
    uint32_t *keys = (uint32_t *)(base + o_keys), *bitmap = (uint32_t *)(base + o_bm), *prefix = (uint32_t *)(base + o_pf);
    int *counts = (int *)(base + o_cnt);
    unsigned long long *acc = (unsigned long long *)(base + o_acc);

    int rc = fill_u32(bitmap, 0u, nwords, stream);
    if (!rc && cap > 0) rc = fill_u32(feats, 0u, (size_t)cap * c, stream);
    if (!rc && cap > 0) rc = fill_u32(acc, 0u, (size_t)cap * c * 2, stream);
    if (!rc && cap > 0) rc = fill_u32(counts, 0u, (size_t)cap, stream);
    if (rc) return rc;
    if (n > 0) {
        DZ_CHECK_ARG(points_b, "dz_voxelize_dynamic_mean: null points");
        hipLaunchKernelGGL(k_dyn_keys, dim3(stream_grid(n, 256)), dim3(256), 0, stream, points_b, n, c + 1, g, batch, keys,
                           bitmap);
    }
    rc = bitmap_scan(bitmap, nwords, prefix, d_num_voxels, 1, ScanDims{g.g[0], g.g[1], g.g[2]}, coords_bzyx, cap,
                         base + o_sw, sw_bytes, stream);
    if (rc) return rc;
    if (n > 0 && cap > 0) {
        hipLaunchKernelGGL(k_dyn_accumulate, dim3(stream_grid((long)n, 256)), dim3(256), 0, stream, points_b, n, c,
                           keys, bitmap, prefix, cap, acc, counts);
        hipLaunchKernelGGL(k_dyn_divide, dim3(stream_grid((long)cap * c, 256)), dim3(256), 0, stream, acc, feats, counts,
                           d_num_voxels, cap, c);
    }
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
