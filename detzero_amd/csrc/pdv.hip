// PDV second stage: the device kernels under detzero_amd/pdv_modules.py.
//
// Reference: detection/detzero_det/models/centerpoint_modules/pdv_head.py:269-637 on top of
//   utils/voxel_aggregation_utils.py:7-157        voxel centroids per backbone level, lookup in the sparse tensor
//   ops/pointnet2/pointnet2_stack/src/ball_query_count_gpu.cu:16-62, group_points_gpu.cu:71-102, pointnet2_utils.py:153-218
//                                                  stacked ball query (first nsample points in index order), grouping, Gaussian KDE
//   utils/density_utils.py:52-109 + ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36,377-404
//                                                  points per box part (6 x 6 x 6 cells of every RoI)
//   utils/attention_utils.py:7-52                 one transformer encoder layer over the 216 grid points of a RoI
//
// MI355X-first choices:
//   * Centroids reuse the bitmap machinery of the sparse index (sparse_index.hip): the stride-4 / stride-8 voxel grids are bitmaps,
//     a 3-kernel scan ranks their set bits, so the centroid lists come out in (b, z, y, x) order - the order torch.unique(dim=0)
//     gives the reference - without a sort.  The same bitmap answers "which feature row of x_conv3 is this voxel" (two loads + popc).
//   * The reference's ball query scans ALL points of a frame for every grid point (O(M x N), ~6e9 distance tests per scale at Waymo
//     sizes).  The points here are voxel centroids, at most one per cell of the level's grid, and their list is sorted by cell key:
//     walking the cells of the ball's bounding box in (z, y, x) order visits the candidates in ascending index order, so the first
//     nsample hits are exactly the reference's - a few hundred bitmap probes per grid point instead of 50 000 distance tests.
//   * Grouping, the KDE of the grouped offsets and the feature gather are one kernel writing the rows the shared MLP (dz_linear_forward)
//     consumes; the max over samples is dz_group_max.  The per-part point counts are one pass over the points with atomics.
#include <stdlib.h>

#include "common.h"

namespace dz {

bool attention_1h_mfma(const float *q, const float *k, const float *v, const unsigned char *mask, int r, int l, int e, float scale, float *out,
                       hipStream_t stream);      // mha.hip

// ------------------------------------------------------------------------------------------------ voxel centroids
struct CentroidGeom {
    float lo[3], vs[3];      // range minimum, voxel size x stride (float32 as the reference computes them)
    int g[3];                // cells per axis (x, y, z): trunc((hi - lo) / vs)
};

// voxel_aggregation_utils.py:29-39: index = (p - lo) / vs in float32; outside iff index < 0 or index >= grid; then .long()
// (round 5: one atomicOr per point was 0.29 ms per 8 two-sweep frames - the stride-4 / stride-8 cells hold dozens of points each and
// their atomics queue up on one address.  The lanes of a run of equal bitmap words OR their bits in registers, the run's last lane
// reads the word first and skips the atomic when its bits are there already - a stale read only costs a redundant atomic.)
__global__ void k_cen_keys(const float *__restrict__ pts, int n, int stride, CentroidGeom g, int batch, uint32_t *__restrict__ keys,
                           uint32_t *__restrict__ bitmap) {
    const int lane = threadIdx.x & 63;
    const int n_pad = (n + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        uint32_t key = KEY_INVALID;
        if (i < n) {
            const float *p = pts + (size_t)i * stride;
            const int b = (int)p[0];
            float q[3];
            bool ok = (unsigned)b < (unsigned)batch;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                q[a] = __fdiv_rn(__fsub_rn(p[1 + a], g.lo[a]), g.vs[a]);
                ok = ok && !(q[a] < 0.f) && !(q[a] >= (float)g.g[a]);
            }
            if (ok) {
                const int cx = (int)q[0], cy = (int)q[1], cz = (int)q[2];
                key = (uint32_t)(((b * g.g[2] + cz) * g.g[1] + cy) * g.g[0] + cx);
            }
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
            if ((__builtin_nontemporal_load(&bitmap[w]) & bits) != bits) atomicOr(&bitmap[w], bits);
        }
    }
}

// level 2: parent cell of every level-1 centroid (voxel_aggregation_utils.py:147-152: coordinates // grid_scaling)
__global__ void k_cen_parent_keys(const int *__restrict__ coords, const int *__restrict__ d_m, int cap, int scaling, int d2, int h2, int w2,
                                  uint32_t *__restrict__ keys, uint32_t *__restrict__ bitmap) {
    const int m = min(*d_m, cap);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int4 c = reinterpret_cast<const int4 *>(coords)[i];
        const uint32_t key = (uint32_t)(((c.x * d2 + c.y / scaling) * h2 + c.z / scaling) * w2 + c.w / scaling);
        atomicOr(&bitmap[key >> 5], 1u << (key & 31u));
        keys[i] = key;
    }
}

// sums[v][1..] += w * row[1..], counts[v] += w   (w = 1 for points, the point count of a level-1 centroid for level 2).
// One thread per row.  Rows arrive in scan order (points) or key order (level-1 centroids), so the lanes of a wavefront hold runs of
// rows of ONE cell: the run is summed in registers (segmented scan, a fixed tree per run) and its last lane issues one atomic per
// column - a fraction of the memory-side atomics of a thread per (row, column); any order stays correct.
__global__ void k_cen_accumulate(const float *__restrict__ rows, int n, const int *__restrict__ d_n, int stride, int cols,
                                 const int *__restrict__ weights, const uint32_t *__restrict__ keys, const uint32_t *__restrict__ bitmap,
                                 const uint32_t *__restrict__ prefix, int cap, float *__restrict__ sums, int *__restrict__ counts) {
    const int nn = d_n ? min(*d_n, n) : n;
    const int lane = threadIdx.x & 63;
    const int n_pad = (nn + 63) & ~63;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += gridDim.x * blockDim.x) {
        const uint32_t key = i < nn ? keys[i] : KEY_INVALID;
        const uint32_t prev = (uint32_t)__shfl_up((int)key, 1, 64), next = (uint32_t)__shfl_down((int)key, 1, 64);
        const unsigned long long heads = __ballot(lane == 0 || prev != key);
        const int first = 63 - __clzll(heads & (~0ull >> (63 - lane)));      // first lane of this lane's run
        const bool live = key != KEY_INVALID;
        const bool last = live && (lane == 63 || next != key);
        int v = cap;
        if (last) v = bitmap_rank(bitmap, prefix, key);
        int w = live ? (weights ? weights[i] : 1) : 0;
        int steps = 0;                                  // scan steps the longest run of this wavefront needs
        while (steps < 6 && __ballot(lane - (1 << steps) >= first) != 0ull) ++steps;
        int wsum = w;
        for (int j = 0; j < steps; ++j) {
            const int t = __shfl_up(wsum, 1 << j, 64);
            if (lane - (1 << j) >= first) wsum += t;
        }
        if (v < cap) {
            atomicAdd(&counts[v], wsum);
            sums[(size_t)v * cols] = rows[(size_t)i * stride];               // batch index column: identical for all members
        }
        for (int ch = 1; ch < cols; ++ch) {
            float x = 0.f;
            if (live) {
                x = rows[(size_t)i * stride + ch];
                if (weights) x = __fmul_rn(x, (float)w);
            }
            for (int j = 0; j < steps; ++j) {
                const float t = __shfl_up(x, 1 << j, 64);
                if (lane - (1 << j) >= first) x = __fadd_rn(x, t);
            }
            if (v < cap) atomicAdd(&sums[(size_t)v * cols + ch], x);
        }
    }
}

__global__ void k_cen_divide(float *__restrict__ sums, const int *__restrict__ counts, const int *__restrict__ d_m, int cap, int cols) {
    const int m = min(*d_m, cap);
    const long total = (long)m * cols;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int v = (int)(idx / cols), ch = (int)(idx % cols);
        if (ch) sums[idx] = __fdiv_rn(sums[idx], (float)counts[v]);
    }
}

// ------------------------------------------------------------------------------------------------ lookup in a sparse level
__global__ void k_index_lookup(const int *__restrict__ coords, const int *__restrict__ d_n, int n, const uint32_t *__restrict__ bitmap,
                               const uint32_t *__restrict__ prefix, LevelGeom lg, int *__restrict__ out) {
    const int nn = d_n ? min(*d_n, n) : n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int v = -1;
        if (i < nn) {
            const int4 c = reinterpret_cast<const int4 *>(coords)[i];
            if (lg.inside(c.x, c.y, c.z, c.w)) v = bitmap_find(bitmap, prefix, lg.key(c.x, c.y, c.z, c.w));
        }
        out[i] = v;
    }
}

// ------------------------------------------------------------------------------------------------ ball query on the cell bitmap
struct BallGeom {
    float lo[3], vs[3];      // x, y, z
    int B, D, H, W;          // bitmap dimensions (D = z cells, H = y, W = x)
};

// one thread per query.  idx (M, nsample): the first nsample points (ascending index within the query's batch item) with
// d^2 < r^2, the rest filled with the first hit; a ball without points -> all zeros and cnt = 0 (pointnet2_utils.py:78-83,186-189)
__global__ __launch_bounds__(256) void k_ball_query(const float *__restrict__ new_xyz, int mq, int per_batch, const float *__restrict__ xyz,
                                                    const uint32_t *__restrict__ bitmap, const uint32_t *__restrict__ prefix, BallGeom g,
                                                    float radius, int nsample, int *__restrict__ idx, int *__restrict__ cnt_out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= mq) return;
    const int b = q / per_batch;
    const float qx = new_xyz[(size_t)q * 3], qy = new_xyz[(size_t)q * 3 + 1], qz = new_xyz[(size_t)q * 3 + 2];
    const float r2 = __fmul_rn(radius, radius);
    const uint32_t batch_key = (uint32_t)b * (uint32_t)(g.D * g.H * g.W);
    const int batch_start = bitmap_rank(bitmap, prefix, batch_key);
    int lo[3], hi[3];
    const float qq[3] = {qx, qy, qz};
    const int dims[3] = {g.W, g.H, g.D};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // one cell of slack on both sides: a centroid lies inside its cell, the float divisions are monotone
        lo[a] = max((int)floorf((qq[a] - radius - g.lo[a]) / g.vs[a]) - 1, 0);
        hi[a] = min((int)floorf((qq[a] + radius - g.lo[a]) / g.vs[a]) + 1, dims[a] - 1);
    }
    int *row = idx + (size_t)q * nsample;
    int cnt = 0, first = 0;
    for (int cz = lo[2]; cz <= hi[2] && cnt < nsample; ++cz)
        for (int cy = lo[1]; cy <= hi[1] && cnt < nsample; ++cy) {
            // the x cells of a (z, y) row are consecutive keys: whole bitmap words, masked to the row's span, and only their SET bits
            // are visited - ascending bit = ascending x, the order of the cell-by-cell probe (round 5: most cells of a ball's box are
            // empty, each cost a word load and a test)
            const uint32_t line = batch_key + (uint32_t)((cz * g.H + cy) * g.W);
            if (hi[0] < lo[0]) continue;
            const uint32_t k0 = line + (uint32_t)lo[0], k1 = line + (uint32_t)hi[0];
            for (uint32_t w = k0 >> 5; w <= (k1 >> 5) && cnt < nsample; ++w) {
                const uint32_t full = bitmap[w];
                uint32_t word = full;
                if (w == (k0 >> 5)) word &= ~0u << (k0 & 31u);
                if (w == (k1 >> 5) && (k1 & 31u) != 31u) word &= (1u << ((k1 & 31u) + 1u)) - 1u;
                if (!word) continue;
                const uint32_t base = prefix[w];
                while (word && cnt < nsample) {
                    const int bit = __ffs((int)word) - 1;
                    word &= word - 1u;
                    const int v = (int)(base + __popc(full & ((1u << bit) - 1u)));
                    const float dx = __fsub_rn(qx, xyz[(size_t)v * 3]), dy = __fsub_rn(qy, xyz[(size_t)v * 3 + 1]), dz = __fsub_rn(qz, xyz[(size_t)v * 3 + 2]);
                    const float d2 = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
                    if (d2 < r2) {
                        if (cnt == 0) first = v - batch_start;
                        row[cnt++] = v - batch_start;
                    }
                }
            }
        }
    for (int s = cnt; s < nsample; ++s) row[s] = first;          // (an empty ball: first = 0)
    cnt_out[q] = cnt;
}

// one wave per query: rows (M * nsample, row_stride) = [dx, dy, dz, density, features (C), 0 ...] - pointnet2_utils.py:192-211 with
// the Gaussian KDE of kde_utils.py:17-64 (bandwidth 0.25: mean over the ball's points of prod_d N((g_e - g_s)_d / h) / h^3)
__global__ __launch_bounds__(256) void k_group_features(const float *__restrict__ new_xyz, int mq, int per_batch, const float *__restrict__ xyz,
                                                        const float *__restrict__ feats, int c, const uint32_t *__restrict__ bitmap,
                                                        const uint32_t *__restrict__ prefix, int cells_per_batch, const int *__restrict__ idx,
                                                        const int *__restrict__ cnt_in, int nsample, float *__restrict__ rows, int row_stride) {
    __shared__ float gs[4][32][3];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + w;
    if (q >= mq) return;                                             // (whole waves leave together: no barrier below)
    const int b = q / per_batch;
    const int batch_start = bitmap_rank(bitmap, prefix, (uint32_t)b * (uint32_t)cells_per_batch);
    const int cnt = cnt_in[q];
    const bool empty = cnt == 0;
    const float bw = 0.25f;
    if (lane < nsample) {
        const int v = batch_start + idx[(size_t)q * nsample + lane];
#pragma unroll
        for (int a = 0; a < 3; ++a) gs[w][lane][a] = empty ? 0.f : __fsub_rn(xyz[(size_t)v * 3 + a], new_xyz[(size_t)q * 3 + a]);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < nsample) {
        float dens = 0.f;
        if (!empty) {
            float acc = 0.f;
            for (int s = 0; s < cnt; ++s) {
                float lp = 0.f;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float u = __fdiv_rn(__fsub_rn(gs[w][lane][a], gs[w][s][a]), bw);
                    lp += -(u * u) / 2.f - 0.91893853320467274178f;
                }
                acc += expf(lp);
            }
            dens = acc / (bw * bw * bw * (float)cnt);
        }
        float *r = rows + ((size_t)q * nsample + lane) * row_stride;
        r[0] = gs[w][lane][0]; r[1] = gs[w][lane][1]; r[2] = gs[w][lane][2]; r[3] = dens;
    }
    // features: the wave copies nsample rows of c floats (and zeroes the padding columns)
    const int cols = row_stride - 4;
    for (int e = 0; e < nsample; ++e) {
        const int v = batch_start + idx[(size_t)q * nsample + e];
        float *r = rows + ((size_t)q * nsample + e) * row_stride + 4;
        for (int ch = lane; ch < cols; ch += 64) r[ch] = (ch < c && !empty) ? feats[(size_t)v * c + ch] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------------ points per box part
// density_utils.py:52-109 (find_num_points_per_part_multi): per point the first max_boxes boxes containing it (box order), then the
// cell of the point in each of those boxes' G x G x G grids; counts (B, O, G, G, G)
__global__ __launch_bounds__(256) void k_part_counts(const float *__restrict__ pts, int n, int stride, const float *__restrict__ rois, int batch,
                                                     int o, int gsz, int max_boxes, int *__restrict__ counts) {
    extern __shared__ float sb[];                                    // [chunk of 64 boxes][10]
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b_blk = blockIdx.y;                                    // one batch item per grid row: its boxes are staged in LDS
    float x = 0.f, y = 0.f, z = 0.f;
    bool mine = false;
    if (i < n) {
        const float *p = pts + (size_t)i * stride;
        mine = (int)p[0] == b_blk;
        x = p[1]; y = p[2]; z = p[3];
    }
    // (a batch's points are contiguous in a collated batch: most (block of points, batch item) pairs have nothing to do)
    if (!__syncthreads_or(mine ? 1 : 0)) return;
    int found = 0;
    const float *bx = rois + (size_t)b_blk * o * 7;
    for (int base = 0; base < o; base += 64) {
        const int nb = min(64, o - base);
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            const float *qb = bx + (size_t)(base + threadIdx.x) * 7;
            float *d = sb + threadIdx.x * 10;
            for (int j = 0; j < 7; ++j) d[j] = qb[j];
            d[7] = cosf(-qb[6]);
            d[8] = sinf(-qb[6]);
            const float rr = 0.5f * (fabsf(qb[3]) + fabsf(qb[4])) + 1e-2f;      // a radius that certainly covers the box footprint (+ slack)
            d[9] = rr * rr * 1.001f;
        }
        __syncthreads();
        if (!mine || found >= max_boxes) continue;
        for (int k = 0; k < nb && found < max_boxes; ++k) {
            const float *qb = sb + k * 10;
            const float sx = x - qb[0], sy = y - qb[1];
            if (sx * sx + sy * sy > qb[9]) continue;                                 // far outside the footprint: the exact test below would fail
            if ((double)fabsf(z - qb[2]) > (double)qb[5] / 2.0) continue;           // check_pt_in_box3d
            const float lx = sx * qb[7] + sy * (-qb[8]);
            const float ly = sx * qb[8] + sy * qb[7];
            if (!(((double)fabsf(lx) < (double)qb[3] / 2.0 + (double)1e-5f) && ((double)fabsf(ly) < (double)qb[4] / 2.0 + (double)1e-5f))) continue;
            ++found;
            // density_utils.py:76-92: rotate by -heading, move the origin to the box corner, divide by the cell size
            const float loc[3] = {lx, ly, z - qb[2]};
            int cell[3];
            bool ok = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float corner = __fadd_rn(loc[a], __fdiv_rn(qb[3 + a], 2.f));
                const float gq = __fdiv_rn(corner, __fdiv_rn(qb[3 + a], (float)gsz));
                ok = ok && !(gq < 0.f) && !(gq >= (float)gsz) && (gq == gq);
                cell[a] = (int)gq;
            }
            if (ok) atomicAdd(&counts[((((size_t)b_blk * o + base + k) * gsz + cell[0]) * gsz + cell[1]) * gsz + cell[2]], 1);
        }
    }
}

// The same counts with the boxes binned on a coarse BEV grid first (round 5).  k_part_counts tests every point against every RoI of its
// frame (320k points x 487 RoIs x 8 frames = 1.2e9 circle tests, ~0.5 ms per pass); a point lies in a handful of boxes at most.  Here
//   k_part_cells   per frame: the staged box table (the 10 floats of k_part_counts' LDS rows) and, per cell of a PC_G x PC_G grid over
//                  the bounding square of the frame's box circles, the ids (ascending) of the boxes whose circle reaches the cell -
//                  one 64-byte line per cell: [count | ids x 15], count -1 = more than 15 (the point kernel then walks all boxes);
//   k_part_counts_binned   one thread per point: its cell's line, then exactly the tests of k_part_counts on the listed boxes.
// The cell of a coordinate is trunc(clamp((v - lo) * inv, 0, G - 1)) for points and box bounds alike: float subtraction, multiplication,
// clamp and truncation are monotone, and a point that passes the circle test of box k has |p - c_k| < R_k (R_k = the test's radius
// with 0.2 % + 1 mm of slack for its rounding), so its cell lies inside the box's cell range; points outside the grid fall into the
// edge cells, which the clamped box ranges include.  Every box that could pass is therefore listed, in box order: bit-identical counts.
constexpr int PC_G = 64, PC_LINE = 16, PC_HDR = 16;               // cells per axis; ints per cell line; floats of the frame header

__device__ __forceinline__ int pc_cell(float v, float lo, float inv) {
    const float t = fminf(fmaxf(__fmul_rn(__fsub_rn(v, lo), inv), 0.f), (float)(PC_G - 1));     // (NaN -> 0)
    return (int)t;
}

__device__ __forceinline__ size_t pc_frame_floats(int o) { return (size_t)PC_HDR + (size_t)PC_G * PC_G * PC_LINE + (((size_t)o * 10 + 15) & ~(size_t)15); }

__global__ __launch_bounds__(256) void k_part_cells(const float *__restrict__ rois, int o, float *__restrict__ ws) {
    extern __shared__ float sb[];                                    // [o][4]: centre x, y, bin radius, -
    __shared__ float red[4][4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float *bx = rois + (size_t)b * o * 7;
    float *frame = ws + (size_t)b * pc_frame_floats(o);
    int *cells = reinterpret_cast<int *>(frame + PC_HDR);
    float *table = frame + PC_HDR + (size_t)PC_G * PC_G * PC_LINE;
    float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
    for (int k = tid; k < o; k += 256) {
        const float *qb = bx + (size_t)k * 7;
        const float rr = 0.5f * (fabsf(qb[3]) + fabsf(qb[4])) + 1e-2f;      // (k_part_counts' radius)
        const float rb = rr * 1.002f + 1e-3f;
        sb[k * 4] = qb[0]; sb[k * 4 + 1] = qb[1]; sb[k * 4 + 2] = rb;
        if (qb[0] - rb < x0) x0 = qb[0] - rb;                               // (comparisons, not fmin: a NaN box joins no bound)
        if (qb[0] + rb > x1) x1 = qb[0] + rb;
        if (qb[1] - rb < y0) y0 = qb[1] - rb;
        if (qb[1] + rb > y1) y1 = qb[1] + rb;
        if (blockIdx.y == 0) {
            float *d = table + (size_t)k * 10;
            for (int j = 0; j < 7; ++j) d[j] = qb[j];
            d[7] = cosf(-qb[6]);
            d[8] = sinf(-qb[6]);
            d[9] = rr * rr * 1.001f;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, d, 64)); x1 = fmaxf(x1, __shfl_xor(x1, d, 64));
        y0 = fminf(y0, __shfl_xor(y0, d, 64)); y1 = fmaxf(y1, __shfl_xor(y1, d, 64));
    }
    if (lane == 0) { red[w][0] = x0; red[w][1] = x1; red[w][2] = y0; red[w][3] = y1; }
    __syncthreads();
    x0 = fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
    x1 = fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
    y0 = fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2]));
    y1 = fmaxf(fmaxf(red[0][3], red[1][3]), fmaxf(red[2][3], red[3][3]));
    // (no finite box: one cell row / column, every list holds every box that compares at all - the clamps make any inv safe)
    const float invx = (x1 > x0 && x1 - x0 < INFINITY) ? (float)PC_G / (x1 - x0) : 0.f, invy = (y1 > y0 && y1 - y0 < INFINITY) ? (float)PC_G / (y1 - y0) : 0.f;
    const float lox = (x0 > -INFINITY && x0 < INFINITY) ? x0 : 0.f, loy = (y0 > -INFINITY && y0 < INFINITY) ? y0 : 0.f;
    if (blockIdx.y == 0 && tid == 0) { frame[0] = lox; frame[1] = loy; frame[2] = invx; frame[3] = invy; }
    // cell range of every box, once per workgroup: x0 | x1 << 8 | y0 << 16 | y1 << 24 (PC_G <= 256), an empty range for a box with a NaN
    // centre or radius (it can contain no point: its comparisons all fail in k_part_counts)
    static_assert(PC_G <= 256, "packed cell ranges");
    unsigned int *rng = reinterpret_cast<unsigned int *>(sb);
    for (int k = tid; k < o; k += 256) {
        const float bxc = sb[k * 4], byc = sb[k * 4 + 1], rb = sb[k * 4 + 2];
        unsigned int r = 1u;                                                // x0 = 1 > x1 = 0
        if (bxc == bxc && byc == byc && rb == rb)
            r = (unsigned int)pc_cell(bxc - rb, lox, invx) | (unsigned int)pc_cell(bxc + rb, lox, invx) << 8 | (unsigned int)pc_cell(byc - rb, loy, invy) << 16 |
                (unsigned int)pc_cell(byc + rb, loy, invy) << 24;
        rng[k * 4 + 3] = r;
    }
    __syncthreads();
    const int cell = blockIdx.y * 256 + tid;
    const unsigned int cy = (unsigned int)(cell / PC_G), cx = (unsigned int)(cell % PC_G);
    int *line = cells + (size_t)cell * PC_LINE;
    int cnt = 0;
    for (int k = 0; k < o; ++k) {
        const unsigned int r = rng[k * 4 + 3];
        if (cx < (r & 255u) || cx > ((r >> 8) & 255u) || cy < ((r >> 16) & 255u) || cy > (r >> 24)) continue;
        if (cnt < PC_LINE - 1) line[1 + cnt] = k;
        ++cnt;
    }
    line[0] = cnt < PC_LINE ? cnt : -1;
}

__global__ __launch_bounds__(256) void k_part_counts_binned(const float *__restrict__ pts, int n, int stride, const float *__restrict__ ws, int batch,
                                                            int o, int gsz, int max_boxes, int *__restrict__ counts) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float *p = pts + (size_t)i * stride;
    const int b = (int)p[0];
    if (b < 0 || b >= batch) return;
    const float x = p[1], y = p[2], z = p[3];
    const float *frame = ws + (size_t)b * pc_frame_floats(o);
    const int *cells = reinterpret_cast<const int *>(frame + PC_HDR);
    const float *table = frame + PC_HDR + (size_t)PC_G * PC_G * PC_LINE;
    const int cell = pc_cell(y, frame[1], frame[3]) * PC_G + pc_cell(x, frame[0], frame[2]);
    int line[PC_LINE];
    {
        const int4 *src = reinterpret_cast<const int4 *>(cells + (size_t)cell * PC_LINE);
#pragma unroll
        for (int j = 0; j < PC_LINE / 4; ++j) { const int4 v = src[j]; line[4 * j] = v.x; line[4 * j + 1] = v.y; line[4 * j + 2] = v.z; line[4 * j + 3] = v.w; }
    }
    const bool all = line[0] < 0;
    const int ncand = all ? o : line[0];
    int found = 0;
    for (int c = 0; c < ncand && found < max_boxes; ++c) {
        int k = c;
        if (!all) {
            // (a run-time index into the register copy of the line would become a scratch array: select)
            k = line[1];
#pragma unroll
            for (int j = 2; j < PC_LINE; ++j) k = (c == j - 1) ? line[j] : k;
        }
        const float *qb = table + (size_t)k * 10;
        const float sx = x - qb[0], sy = y - qb[1];
        if (sx * sx + sy * sy > qb[9]) continue;
        if ((double)fabsf(z - qb[2]) > (double)qb[5] / 2.0) continue;           // check_pt_in_box3d
        const float lx = sx * qb[7] + sy * (-qb[8]);
        const float ly = sx * qb[8] + sy * qb[7];
        if (!(((double)fabsf(lx) < (double)qb[3] / 2.0 + (double)1e-5f) && ((double)fabsf(ly) < (double)qb[4] / 2.0 + (double)1e-5f))) continue;
        ++found;
        const float loc[3] = {lx, ly, z - qb[2]};
        int cellp[3];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float corner = __fadd_rn(loc[a], __fdiv_rn(qb[3 + a], 2.f));
            const float gq = __fdiv_rn(corner, __fdiv_rn(qb[3 + a], (float)gsz));
            ok = ok && !(gq < 0.f) && !(gq >= (float)gsz) && (gq == gq);
            cellp[a] = (int)gq;
        }
        if (ok) atomicAdd(&counts[((((size_t)b * o + k) * gsz + cellp[0]) * gsz + cellp[1]) * gsz + cellp[2]], 1);
    }
}

// ------------------------------------------------------------------------------------------------ single-head attention, wide head
// out[r] = softmax(q[r] k[r]^T * scale + mask) v[r] for R independent sequences of L <= 256 tokens with E <= 256 channels (the PDV
// encoder layer: L = 216 grid points, E = 192, one head - dz_mha_core serves the refiner's 32-channel heads).  A workgroup = 8 query
// rows of one sequence, two per wave; lanes split the keys for the scores and the channels for the weighted sum.
constexpr int ATT_QB = 8;

__global__ __launch_bounds__(256) void k_attention_1h(const float *__restrict__ q, const float *__restrict__ k, const float *__restrict__ v,
                                                      const unsigned char *__restrict__ mask, int l, int e, float scale, float *__restrict__ out) {
    extern __shared__ float sm[];                                    // qs[ATT_QB][e], ps[ATT_QB][l]
    float *qs = sm, *ps = sm + ATT_QB * e;
    const int qblocks = (l + ATT_QB - 1) / ATT_QB;
    const int r = blockIdx.x / qblocks, q0 = (blockIdx.x % qblocks) * ATT_QB;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float *qr = q + (size_t)r * l * e, *kr = k + (size_t)r * l * e, *vr = v + (size_t)r * l * e;
    for (int t = threadIdx.x; t < ATT_QB * e; t += 256) {
        const int qi = q0 + t / e;
        qs[t] = qi < l ? qr[(size_t)qi * e + t % e] * scale : 0.f;
    }
    __syncthreads();
    const int qa = 2 * w, qb = 2 * w + 1;                            // this wave's two query slots
    float s[4][2];
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int j = lane + 64 * kk;
        s[kk][0] = s[kk][1] = -INFINITY;
        if (j < l && !(mask && mask[(size_t)r * l + j])) {
            const float4 *kp = reinterpret_cast<const float4 *>(kr + (size_t)j * e);
            float a0 = 0.f, a1 = 0.f;
            for (int c4 = 0; c4 < e / 4; ++c4) {
                const float4 kv = kp[c4];
                const float *x0 = qs + qa * e + c4 * 4, *x1 = qs + qb * e + c4 * 4;
                a0 += kv.x * x0[0] + kv.y * x0[1] + kv.z * x0[2] + kv.w * x0[3];
                a1 += kv.x * x1[0] + kv.y * x1[1] + kv.z * x1[2] + kv.w * x1[3];
            }
            s[kk][0] = a0; s[kk][1] = a1;
        }
        mx[0] = fmaxf(mx[0], s[kk][0]); mx[1] = fmaxf(mx[1], s[kk][1]);
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { mx[0] = fmaxf(mx[0], __shfl_xor(mx[0], d, 64)); mx[1] = fmaxf(mx[1], __shfl_xor(mx[1], d, 64)); }
    float sum[2] = {0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int t = 0; t < 2; ++t) { s[kk][t] = (s[kk][t] == -INFINITY) ? 0.f : expf(s[kk][t] - mx[t]); sum[t] += s[kk][t]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sum[0] += __shfl_xor(sum[0], d, 64); sum[1] += __shfl_xor(sum[1], d, 64); }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int j = lane + 64 * kk;
        if (j < l) { ps[qa * l + j] = s[kk][0] / sum[0]; ps[qb * l + j] = s[kk][1] / sum[1]; }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_block();
    float o[4][2];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) o[dd][0] = o[dd][1] = 0.f;
    for (int j = 0; j < l; ++j) {
        const float p0 = ps[qa * l + j], p1 = ps[qb * l + j];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const int d = lane + 64 * dd;
            if (d < e) { const float vv = vr[(size_t)j * e + d]; o[dd][0] += p0 * vv; o[dd][1] += p1 * vv; }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qi = q0 + 2 * w + t;
        if (qi >= l) continue;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const int d = lane + 64 * dd;
            if (d < e) out[((size_t)r * l + qi) * e + d] = o[dd][t];
        }
    }
}

}  // namespace dz

using namespace dz;

extern "C" {

static size_t cen_layout(int n, int batch, int gx, int gy, int gz, int scaling, int cap1, size_t *o_bm1, size_t *o_pf1, size_t *o_bm2, size_t *o_pf2,
                         size_t *o_keys2, size_t *o_sw, size_t *sw_bytes, size_t *words1, size_t *words2) {
    *words1 = dz_index_words(batch, gz, gy, gx, DZ_LAYOUT_LINEAR);
    const int d2 = (gz + scaling - 1) / scaling, h2 = (gy + scaling - 1) / scaling, w2 = (gx + scaling - 1) / scaling;
    *words2 = dz_index_words(batch, d2, h2, w2, DZ_LAYOUT_LINEAR);
    size_t off = align_up((size_t)(n < 1 ? 1 : n) * 4, 256);                 // keys of the points
    *o_bm1 = off; off += align_up(*words1 * 4, 256);
    *o_pf1 = off; off += align_up(*words1 * 4, 256);
    *o_bm2 = off; off += align_up(*words2 * 4, 256);
    *o_pf2 = off; off += align_up(*words2 * 4, 256);
    *o_keys2 = off; off += align_up((size_t)(cap1 < 1 ? 1 : cap1) * 4, 256);
    *sw_bytes = bitmap_scan_workspace_bytes(*words1 > *words2 ? *words1 : *words2);
    *o_sw = off; off += align_up(*sw_bytes, 256);
    return off;
}

size_t dz_pdv_centroids_workspace_bytes(int n, int batch, int gx, int gy, int gz, int scaling, int cap1) {
    size_t a, b, c, d, e, f, g, h, i;
    return cen_layout(n, batch, gx, gy, gz, scaling < 1 ? 1 : scaling, cap1, &a, &b, &c, &d, &e, &f, &g, &h, &i);
}

int dz_pdv_voxel_centroids(const float *points_b, int n, int c, const float *h_range6, const float *h_vsize3, const int *h_grid3, int batch,
                           int scaling, float *cen1, int *coords1, int *counts1, int *d_m1, int cap1, float *cen2, int *coords2, int *counts2,
                           int *d_m2, int cap2, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 0 && c >= 3 && batch >= 1 && scaling >= 1 && cap1 >= 0 && cap2 >= 0, "dz_pdv_voxel_centroids: bad sizes");
    DZ_CHECK_ARG(h_range6 && h_vsize3 && h_grid3 && cen1 && coords1 && counts1 && d_m1 && ws, "dz_pdv_voxel_centroids: null pointer");
    DZ_CHECK_ARG(!cen2 == !coords2 && !cen2 == !counts2 && !cen2 == !d_m2, "dz_pdv_voxel_centroids: the level-2 outputs go together");
    CentroidGeom g;
    for (int i = 0; i < 3; ++i) {
        g.lo[i] = h_range6[i]; g.vs[i] = h_vsize3[i]; g.g[i] = h_grid3[i];
        DZ_CHECK_ARG(g.vs[i] > 0.f && g.g[i] >= 1, "dz_pdv_voxel_centroids: bad geometry");
    }
    const size_t cells = (size_t)batch * g.g[0] * g.g[1] * g.g[2];
    if (cells >= 0x7FFFFFFFull) { set_error("dz_pdv_voxel_centroids: batch x grid exceeds 32-bit cell keys"); return DZ_ERR_UNSUPPORTED; }
    size_t o_bm1, o_pf1, o_bm2, o_pf2, o_keys2, o_sw, sw_bytes, words1, words2;
    const size_t need = cen_layout(n, batch, g.g[0], g.g[1], g.g[2], scaling, cap1, &o_bm1, &o_pf1, &o_bm2, &o_pf2, &o_keys2, &o_sw, &sw_bytes, &words1, &words2);
    if (ws_bytes < need) { set_error("dz_pdv_voxel_centroids: workspace %zu < %zu", ws_bytes, need); return DZ_ERR_WORKSPACE; }
    unsigned char *base = reinterpret_cast<unsigned char *>(ws);
    uint32_t *keys = reinterpret_cast<uint32_t *>(base), *bm1 = reinterpret_cast<uint32_t *>(base + o_bm1), *pf1 = reinterpret_cast<uint32_t *>(base + o_pf1);
    uint32_t *bm2 = reinterpret_cast<uint32_t *>(base + o_bm2), *pf2 = reinterpret_cast<uint32_t *>(base + o_pf2), *keys2 = reinterpret_cast<uint32_t *>(base + o_keys2);
    const int cols = 1 + c;
    int rc = fill_u32(bm1, 0u, words1, stream);
    if (rc) return rc;
    if (cap1 > 0) {
        rc = fill_u32(cen1, 0u, (size_t)cap1 * cols, stream); if (rc) return rc;
        rc = fill_u32(counts1, 0u, (size_t)cap1, stream); if (rc) return rc;
    }
    if (n > 0) {
        DZ_CHECK_ARG(points_b, "dz_pdv_voxel_centroids: null points");
        hipLaunchKernelGGL(k_cen_keys, dim3(stream_grid(n, 256)), dim3(256), 0, stream, points_b, n, cols, g, batch, keys, bm1);
    }
    rc = bitmap_scan(bm1, words1, pf1, d_m1, 0, ScanDims{g.g[2], g.g[1], g.g[0]}, coords1, cap1, base + o_sw, sw_bytes, stream);
    if (rc) return rc;
    if (n > 0 && cap1 > 0) {
        hipLaunchKernelGGL(k_cen_accumulate, dim3(stream_grid((long)n, 256)), dim3(256), 0, stream, points_b, n, (const int *)nullptr, cols, cols,
                           (const int *)nullptr, keys, bm1, pf1, cap1, cen1, counts1);
        hipLaunchKernelGGL(k_cen_divide, dim3(stream_grid((long)cap1 * cols, 256)), dim3(256), 0, stream, cen1, counts1, d_m1, cap1, cols);
    }
    if (cen2) {
        const int d2 = (g.g[2] + scaling - 1) / scaling, h2 = (g.g[1] + scaling - 1) / scaling, w2 = (g.g[0] + scaling - 1) / scaling;
        rc = fill_u32(bm2, 0u, words2, stream); if (rc) return rc;
        if (cap2 > 0) {
            rc = fill_u32(cen2, 0u, (size_t)cap2 * cols, stream); if (rc) return rc;
            rc = fill_u32(counts2, 0u, (size_t)cap2, stream); if (rc) return rc;
        }
        if (cap1 > 0)
            hipLaunchKernelGGL(k_cen_parent_keys, dim3(stream_grid(cap1, 256)), dim3(256), 0, stream, coords1, d_m1, cap1, scaling, d2, h2, w2, keys2, bm2);
        rc = bitmap_scan(bm2, words2, pf2, d_m2, 0, ScanDims{d2, h2, w2}, coords2, cap2, base + o_sw, sw_bytes, stream);
        if (rc) return rc;
        if (cap1 > 0 && cap2 > 0) {
            hipLaunchKernelGGL(k_cen_accumulate, dim3(stream_grid((long)cap1, 256)), dim3(256), 0, stream, cen1, cap1, d_m1, cols, cols, counts1,
                               keys2, bm2, pf2, cap2, cen2, counts2);
            hipLaunchKernelGGL(k_cen_divide, dim3(stream_grid((long)cap2 * cols, 256)), dim3(256), 0, stream, cen2, counts2, d_m2, cap2, cols);
        }
    }
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_index_lookup(const int *coords, const int *d_n, int n, const uint32_t *bitmap, const uint32_t *prefix, int b, int d, int h, int w, int layout,
                    int *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 0 && b >= 1 && d >= 1 && h >= 1 && w >= 1 && (layout == DZ_LAYOUT_LINEAR || layout == DZ_LAYOUT_BRICK), "dz_index_lookup: bad sizes");
    if (n == 0) return DZ_OK;
    DZ_CHECK_ARG(coords && bitmap && prefix && out, "dz_index_lookup: null pointer");
    hipLaunchKernelGGL(k_index_lookup, dim3(stream_grid(n, 256)), dim3(256), 0, stream, coords, d_n, n, bitmap, prefix, make_level(b, d, h, w, layout), out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_pdv_ball_query(const float *new_xyz, int mq, int per_batch, const float *xyz, const uint32_t *bitmap, const uint32_t *prefix, int b, int d, int h,
                      int w, const float *h_lo3, const float *h_vs3, float radius, int nsample, int *idx, int *cnt, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(mq >= 0 && per_batch >= 1 && nsample >= 1 && nsample <= 32 && radius > 0.f, "dz_pdv_ball_query: bad sizes (nsample <= 32)");
    if (mq == 0) return DZ_OK;
    DZ_CHECK_ARG(new_xyz && xyz && bitmap && prefix && h_lo3 && h_vs3 && idx && cnt, "dz_pdv_ball_query: null pointer");
    DZ_CHECK_ARG((mq + per_batch - 1) / per_batch <= b, "dz_pdv_ball_query: %d queries in groups of %d exceed %d batch items", mq, per_batch, b);
    BallGeom g;
    for (int i = 0; i < 3; ++i) { g.lo[i] = h_lo3[i]; g.vs[i] = h_vs3[i]; }
    g.B = b; g.D = d; g.H = h; g.W = w;
    hipLaunchKernelGGL(k_ball_query, dim3(ceil_div(mq, 256)), dim3(256), 0, stream, new_xyz, mq, per_batch, xyz, bitmap, prefix, g, radius, nsample, idx, cnt);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_pdv_group_features(const float *new_xyz, int mq, int per_batch, const float *xyz, const float *feats, int c, const uint32_t *bitmap,
                          const uint32_t *prefix, int cells_per_batch, const int *idx, const int *cnt, int nsample, float *rows, int row_stride,
                          void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(mq >= 0 && per_batch >= 1 && nsample >= 1 && nsample <= 32 && c >= 1 && row_stride >= c + 4, "dz_pdv_group_features: bad sizes");
    if (mq == 0) return DZ_OK;
    DZ_CHECK_ARG(new_xyz && xyz && feats && bitmap && prefix && idx && cnt && rows, "dz_pdv_group_features: null pointer");
    hipLaunchKernelGGL(k_group_features, dim3(ceil_div(mq, 4)), dim3(256), 0, stream, new_xyz, mq, per_batch, xyz, feats, c, bitmap, prefix, cells_per_batch,
                       idx, cnt, nsample, rows, row_stride);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_pdv_part_counts(const float *points_b, int n, int stride, const float *rois, int batch, int o, int grid, int max_boxes, int *counts, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 0 && stride >= 4 && batch >= 1 && o >= 0 && grid >= 1 && max_boxes >= 1, "dz_pdv_part_counts: bad sizes");
    if (o == 0) return DZ_OK;
    DZ_CHECK_ARG(rois && counts && (points_b || n == 0), "dz_pdv_part_counts: null pointer");
    const int rc = fill_u32(counts, 0u, (size_t)batch * o * grid * grid * grid, stream);
    if (rc) return rc;
    if (n == 0) return DZ_OK;
    hipLaunchKernelGGL(k_part_counts, dim3(ceil_div(n, 256), batch), dim3(256), 64 * 10 * sizeof(float), stream, points_b, n, stride, rois, batch, o, grid,
                       max_boxes, counts);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

size_t dz_pdv_part_counts_ws_bytes(int batch, int o) {
    if (batch < 1 || o < 0) return 0;
    return (size_t)batch * ((size_t)PC_HDR + (size_t)PC_G * PC_G * PC_LINE + (((size_t)o * 10 + 15) & ~(size_t)15)) * sizeof(float);
}

int dz_pdv_part_counts_binned(const float *points_b, int n, int stride, const float *rois, int batch, int o, int grid, int max_boxes, int *counts,
                              void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 0 && stride >= 4 && batch >= 1 && o >= 0 && grid >= 1 && max_boxes >= 1, "dz_pdv_part_counts_binned: bad sizes");
    if (o == 0) return DZ_OK;
    DZ_CHECK_ARG(rois && counts && (points_b || n == 0), "dz_pdv_part_counts_binned: null pointer");
    DZ_CHECK_ARG(ws && ws_bytes >= dz_pdv_part_counts_ws_bytes(batch, o) && ((uintptr_t)ws & 63) == 0,
                 "dz_pdv_part_counts_binned: workspace of %zu bytes, 64-byte aligned (dz_pdv_part_counts_ws_bytes)", dz_pdv_part_counts_ws_bytes(batch, o));
    // k_part_cells stages the o box circles (16 bytes each) in dynamic LDS next to 64 bytes of static LDS: beyond the default 64 KiB
    // limit of a launch (o > 4092) the all-pairs kernel runs instead
    if ((size_t)o * 16 + 64 > 64 * 1024) return dz_pdv_part_counts(points_b, n, stride, rois, batch, o, grid, max_boxes, counts, stream_);     // (box circles staged in LDS)
    const int rc = fill_u32(counts, 0u, (size_t)batch * o * grid * grid * grid, stream);
    if (rc) return rc;
    if (n == 0) return DZ_OK;
    hipLaunchKernelGGL(k_part_cells, dim3(batch, PC_G * PC_G / 256), dim3(256), (size_t)o * 16, stream, rois, o, (float *)ws);
    hipLaunchKernelGGL(k_part_counts_binned, dim3(ceil_div(n, 256)), dim3(256), 0, stream, points_b, n, stride, (const float *)ws, batch, o, grid, max_boxes,
                       counts);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_attention_single_head(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask, int r, int l, int e, float scale,
                             float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(r >= 0 && l >= 1 && l <= 256 && e >= 4 && e <= 256 && e % 4 == 0, "dz_attention_single_head: L <= 256, E <= 256, E %% 4 == 0 (got %d, %d)", l, e);
    if (r == 0) return DZ_OK;
    DZ_CHECK_ARG(q && k && v && out, "dz_attention_single_head: null pointer");
    // the matrix-core kernel (mha.hip: k_mha_core with head dim E, r03: 1.93 ms -> ~0.3 ms for 487 x 216 x 192) whenever E has an instance
    static const bool valu_only = getenv("DZ_TUNE_ATT1H_VALU") != nullptr;
    if (!valu_only && attention_1h_mfma(q, k, v, key_padding_mask, r, l, e, scale, out, stream)) {
        DZ_LAUNCH_CHECK();
        return DZ_OK;
    }
    const int qblocks = (l + ATT_QB - 1) / ATT_QB;
    hipLaunchKernelGGL(k_attention_1h, dim3(r * qblocks), dim3(256), (size_t)ATT_QB * (e + l) * sizeof(float), stream, q, k, v, key_padding_mask, l, e, scale, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
