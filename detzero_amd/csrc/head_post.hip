// CenterHead post-processing on the device (gfx950): score map, exact top-K, box decode with
// range / score masks, rotated BEV IoU, rotated NMS with the suppression sweep on the GPU, and the
// refiner's points-in-boxes test.
//
// Reference:
//   detection/detzero_det/models/centerpoint_modules/center_head.py:315-368
//   detection/detzero_det/utils/centernet_utils.py:138-230 (_topk, decode_bbox_from_heatmap)
//   detection/detzero_det/utils/model_nms_utils.py:6-25
//   utils/detzero_utils/ops/iou3d_nms/src/iou3d_nms_kernel.cu:15-232,328-335,386-430 (geometry)
//   utils/detzero_utils/ops/iou3d_nms/src/iou3d_nms.cpp:114-160 (host sweep, D2H copy -> removed)
//   utils/detzero_utils/ops/roiaware_pool3d/src/roiaware_pool3d_kernel.cu:16-36,352-374
//
// Parity notes: all geometry is evaluated with the reference's formula sequence in fp32 and with
// FMA contraction disabled, so that discrete decisions (IoU > thr, inside/outside) agree with the
// CPU restatement wherever libm and ocml agree on sin/cos/atan2.
#include <string.h>

#include "common.h"

#pragma clang fp contract(off)

#include "box_geom.h"

namespace dz {

constexpr int PAIR_THREADS = 128;

template <bool IOU>
__global__ void k_pairwise(const float *__restrict__ a, int na, const float *__restrict__ b, int nb,
                           float *__restrict__ out) {
    __shared__ P2 cp_s[16 * PAIR_THREADS];
    __shared__ float ang_s[16 * PAIR_THREADS];
    const long total = (long)na * nb;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int i = (int)(idx / nb), j = (int)(idx % nb);
        float A[7], B[7];
#pragma unroll
        for (int q = 0; q < 7; ++q) { A[q] = a[i * 7 + q]; B[q] = b[j * 7 + q]; }
        out[idx] = IOU ? rect_iou(A, B, cp_s + threadIdx.x, ang_s + threadIdx.x, PAIR_THREADS)
                       : rect_overlap(A, B, cp_s + threadIdx.x, ang_s + threadIdx.x, PAIR_THREADS);
    }
}

// ------------------------------------------------------------------------------------------
// rotated NMS
// ------------------------------------------------------------------------------------------
// mask[i][cb] bit j: iou(box i, box cb*64+j) > thr, only for j > i (upper triangle).
// One wavefront per (16 rows, 64 columns): lane = column box, the wave walks its 16 row boxes and
// every lane evaluates ONE rotated IoU per step; __ballot packs the 64 verdicts into the mask word.
// Pairs whose circumscribed circles are more than 5 cm apart cannot touch (the reference's corner
// test has a 1 cm margin), so their IoU is exactly 0 and the polygon clipping is skipped.
// Round 5: the wave first runs the circle test for all of its NMS_ROWS_PER_WAVE x 64 pairs and COMPACTS the survivors (ballot +
// prefix into an LDS list), then evaluates the rotated IoU of the list with all 64 lanes busy.  Before, every (row, 64 columns)
// step ran the polygon clipping whenever ANY lane had survived - a few active lanes per step: 215 us per 32 frames, the largest
// kernel of the post stage.  Same pairs, same rect_iou, same mask.
constexpr int NMS_ROWS_PER_WAVE = 16;    // 500 candidates -> 32 x 8 (upper triangle: ~150) wavefronts per frame
__global__ __launch_bounds__(64) void k_nms_mask(const float *__restrict__ boxes, const int *__restrict__ d_n, int n_cap,
                                                 float thr, unsigned long long *__restrict__ mask, int col_blocks) {
    // batch item = blockIdx.z: boxes (B,n_cap,7), d_n (B), mask (B,n_cap,col_blocks)
    boxes += (size_t)blockIdx.z * n_cap * 7;
    mask += (size_t)blockIdx.z * n_cap * col_blocks;
    const int n = d_n ? min(d_n[blockIdx.z], n_cap) : n_cap;
    const int cb = blockIdx.x;
    const int rb = blockIdx.y / (64 / NMS_ROWS_PER_WAVE), part = blockIdx.y % (64 / NMS_ROWS_PER_WAVE);
    const int row0 = rb * 64 + part * NMS_ROWS_PER_WAVE;
    if (row0 >= n) return;
    __shared__ P2 cp_s[16 * 64];
    __shared__ float ang_s[16 * 64];
    __shared__ unsigned short pair_s[NMS_ROWS_PER_WAVE * 64];          // surviving pairs: local row << 6 | column lane
    __shared__ unsigned long long bits_s[NMS_ROWS_PER_WAVE];
    const int t = threadIdx.x;
    const int rows = min(NMS_ROWS_PER_WAVE, n - row0);
    const int col = cb * 64 + t;
    if (cb < rb || cb * 64 >= n) {        // lower triangle / beyond n: nothing can be suppressed there
        if (t < rows) mask[(size_t)(row0 + t) * col_blocks + cb] = 0ull;
        return;
    }
    float B[7];
    const bool col_ok = col < n;
#pragma unroll
    for (int q = 0; q < 7; ++q) B[q] = col_ok ? boxes[col * 7 + q] : 0.f;
    const float rb2 = 0.25f * (B[3] * B[3] + B[4] * B[4]);          // squared half diagonal
    const float rbr = sqrtf(rb2);
    if (t < NMS_ROWS_PER_WAVE) bits_s[t] = 0ull;
    // ---- circle test of every pair, survivors appended in (row, column) order
    int npairs = 0;
    for (int i = 0; i < rows; ++i) {
        const int row = row0 + i;
        const float ax = boxes[row * 7], ay = boxes[row * 7 + 1], adx = boxes[row * 7 + 3], ady = boxes[row * 7 + 4];      // wave-uniform -> scalar loads
        bool near = false;
        if (col_ok && col > row) {
            const float dx = ax - B[0], dy = ay - B[1];
            const float reach = sqrtf(0.25f * (adx * adx + ady * ady)) + rbr + 0.05f;
            near = dx * dx + dy * dy <= reach * reach;
        }
        const unsigned long long m = __ballot(near);
        if (near) pair_s[npairs + __popcll(m & ((1ull << t) - 1ull))] = (unsigned short)((i << 6) | t);
        npairs += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- rotated IoU of the survivors, 64 at a time
    for (int k0 = 0; k0 < npairs; k0 += 64) {
        const int k = k0 + t;
        if (k < npairs) {
            const int pr = pair_s[k], i = pr >> 6, c = pr & 63;
            float A[7], C[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) { A[q] = boxes[(row0 + i) * 7 + q]; C[q] = boxes[(cb * 64 + c) * 7 + q]; }
            if (rect_iou(A, C, cp_s + t, ang_s + t, 64) > thr) atomicOr(&bits_s[i], 1ull << c);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (t < rows) mask[(size_t)(row0 + t) * col_blocks + cb] = bits_s[t];
}

// sequential suppression, one workgroup: 64 mask rows at a time are staged in LDS, wave 0 sweeps.
__global__ __launch_bounds__(256) void k_nms_sweep(const unsigned long long *__restrict__ mask, const int *__restrict__ d_n,
                                                   int n_cap, int col_blocks, int post_max, int *__restrict__ keep,
                                                   int *__restrict__ d_num_keep) {
    extern __shared__ unsigned long long rows[];   // 64 x col_blocks
    // batch item = blockIdx.x: mask (B,n_cap,col_blocks), keep (B,n_cap), d_num_keep (B)
    mask += (size_t)blockIdx.x * n_cap * col_blocks;
    keep += (size_t)blockIdx.x * n_cap;
    d_num_keep += blockIdx.x;
    const int n = d_n ? min(d_n[blockIdx.x], n_cap) : n_cap;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    unsigned long long remv = 0ull;   // lane l of wave 0 owns column block l
    int nk = 0;
    const int nrb = (n + 63) / 64;
    for (int rb = 0; rb < nrb; ++rb) {
        const int rows_here = min(64, n - rb * 64);
        for (int idx = threadIdx.x; idx < rows_here * col_blocks; idx += 256)
            rows[idx] = mask[(size_t)rb * 64 * col_blocks + idx];
        __syncthreads();
        if (wid == 0) {
            for (int r = 0; r < rows_here; ++r) {
                const unsigned long long own = __shfl(remv, rb, 64);
                if (!((own >> r) & 1ull) && nk < post_max) {
                    if (lane == 0) keep[nk] = rb * 64 + r;
                    ++nk;
                    if (lane < col_blocks) remv |= rows[r * col_blocks + lane];
                }
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *d_num_keep = nk;
}

// out[b][j] = [box(7) | score | label+1] of candidate keep[b][j] for j < min(num_keep[b], post_max), zeros after
// (model_nms_utils.py:22-25 selection + the 1-based labels of center_head.py:356)
__global__ void k_pack_detections(const float *__restrict__ boxes, const float *__restrict__ scores, const int *__restrict__ labels,
                                  const int *__restrict__ keep, const int *__restrict__ d_num_keep, int k, int post_max,
                                  float *__restrict__ out) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= post_max * 9) return;
    const int j = idx / 9, c = idx % 9;
    float v = 0.f;
    if (j < min(d_num_keep[b], post_max) && j < k) {
        const int src = keep[(size_t)b * k + j];
        v = c < 7 ? boxes[((size_t)b * k + src) * 7 + c] : (c == 7 ? scores[(size_t)b * k + src] : (float)(labels[(size_t)b * k + src] + 1));
    }
    out[((size_t)b * post_max + j) * 9 + c] = v;
}

// ------------------------------------------------------------------------------------------
// score map + top-K + decode
// ------------------------------------------------------------------------------------------
constexpr int HEAD_COLS = 12;  // center 0:2 | center_z 2 | dim 3:6 | rot 6:8 | iou 8 | hm 9:12

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// Exact top-K by a chip-wide radix select over the score bits (scores >= 0, so uint order == float
// order): three digit levels of 11 + 11 + 10 bits.  Every level is one grid-wide histogram pass (LDS
// private histograms, flushed with atomics) and a one-workgroup pick of the digit that contains the
// K-th largest key.  Per batch item the state is {hist[2048], prefix, need, n_gt, n_eq}.
constexpr int RADIX_BINS = 2048;
constexpr int TOPK_STATE_WORDS = RADIX_BINS + 8;
constexpr int ST_PREFIX = RADIX_BINS + 0, ST_NEED = RADIX_BINS + 1, ST_NGT = RADIX_BINS + 2, ST_NEQ = RADIX_BINS + 3;
constexpr int TIE_CAP = 4096;

__device__ __forceinline__ void hist_flush(const uint32_t *lh, uint32_t *gh) {
    for (int i = threadIdx.x; i < RADIX_BINS; i += blockDim.x) {
        const uint32_t v = lh[i];
        if (v) atomicAdd(&gh[i], v);
    }
}

// level 0 fused with the score map: keys[b][cls*HW + pix] = bits of sigmoid(hm)*clamp(iou,0,1)^2
__global__ __launch_bounds__(256) void k_score_hist0(const float *__restrict__ head, int hw, int ncls, int use_iou,
                                                     uint32_t *__restrict__ keys, uint32_t *__restrict__ state) {
    __shared__ uint32_t lh[RADIX_BINS];
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < RADIX_BINS; i += 256) lh[i] = 0u;
    __syncthreads();
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < hw; pix += gridDim.x * 256) {
        const float *row = head + ((size_t)b * hw + pix) * HEAD_COLS;
        float w = 1.f;
        if (use_iou) {
            const float iou = fminf(fmaxf(row[8], 0.f), 1.f);
            w = iou * iou;
        }
        for (int c = 0; c < ncls; ++c) {
            float sc = sigmoidf_(row[9 + c]);
            if (use_iou) sc = sc * w;
            const uint32_t kv = __float_as_uint(sc);
            keys[((size_t)b * ncls + c) * hw + pix] = kv;
            atomicAdd(&lh[kv >> 21], 1u);
        }
    }
    __syncthreads();
    hist_flush(lh, state + (size_t)b * TOPK_STATE_WORDS);
}

// levels 1 (bits 20..10) and 2 (bits 9..0): histogram of the keys that match the prefix found so far
template <int LEVEL>
__global__ __launch_bounds__(256) void k_radix_hist(const uint32_t *__restrict__ keys, int n, uint32_t *__restrict__ state) {
    __shared__ uint32_t lh[RADIX_BINS];
    const int b = blockIdx.y;
    uint32_t *st = state + (size_t)b * TOPK_STATE_WORDS;
    const uint32_t prefix = st[ST_PREFIX];
    constexpr uint32_t himask = LEVEL == 1 ? 0xFFE00000u : 0xFFFFFC00u;
    constexpr int shift = LEVEL == 1 ? 10 : 0;
    constexpr uint32_t dmask = LEVEL == 1 ? 2047u : 1023u;
    for (int i = threadIdx.x; i < RADIX_BINS; i += 256) lh[i] = 0u;
    __syncthreads();
    const uint32_t *kb = keys + (size_t)b * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t kv = kb[i];
        if ((kv & himask) == prefix) atomicAdd(&lh[(kv >> shift) & dmask], 1u);
    }
    __syncthreads();
    hist_flush(lh, st);
}

// one workgroup per batch item: digit d with  #keys(digit > d) < need <= #keys(digit >= d); clears the histogram
__global__ __launch_bounds__(256) void k_radix_pick(uint32_t *__restrict__ state, int level, int k, int n) {
    __shared__ uint32_t lds[4];
    uint32_t *st = state + (size_t)blockIdx.x * TOPK_STATE_WORDS;
    const int t = threadIdx.x;
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    const uint32_t need = level == 0 ? (uint32_t)min(k, n) : st[ST_NEED];
    const uint32_t prefix = level == 0 ? 0u : st[ST_PREFIX];
    // thread t owns bins 2047-8t ... 2040-8t (descending)
    uint32_t h[8], sum = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) { h[j] = st[RADIX_BINS - 1 - (8 * t + j)]; sum += h[j]; }
    uint32_t total;
    const uint32_t above = block_excl_scan_256(sum, lds, total);
#pragma unroll
    for (int j = 0; j < 8; ++j) st[RADIX_BINS - 1 - (8 * t + j)] = 0u;
    if (above < need && need <= above + sum) {
        uint32_t acc = above;
        int d = RADIX_BINS - 1 - 8 * t;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (acc + h[j] >= need) { d = RADIX_BINS - 1 - (8 * t + j); break; }
            acc += h[j];
        }
        st[ST_PREFIX] = prefix | ((uint32_t)d << shift);
        st[ST_NEED] = need - acc;
        if (level == 2) { st[ST_NGT] = 0u; st[ST_NEQ] = 0u; }
    }
}

// candidates: every key > T (any order); indices of keys == T (any order, up to TIE_CAP)
__global__ __launch_bounds__(256) void k_topk_collect(const uint32_t *__restrict__ keys, int n, uint32_t *__restrict__ state,
                                                      unsigned long long *__restrict__ cand, uint32_t *__restrict__ ties,
                                                      int maxk) {
    const int b = blockIdx.y;
    uint32_t *st = state + (size_t)b * TOPK_STATE_WORDS;
    const uint32_t T = st[ST_PREFIX];
    const uint32_t *kb = keys + (size_t)b * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const uint32_t kv = kb[i];
        if (kv > T) {
            const uint32_t pos = atomicAdd(&st[ST_NGT], 1u);
            if ((int)pos < maxk) cand[(size_t)b * maxk + pos] = ((unsigned long long)kv << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
        } else if (kv == T) {
            const uint32_t pos = atomicAdd(&st[ST_NEQ], 1u);
            if (pos < (uint32_t)TIE_CAP) ties[(size_t)b * TIE_CAP + pos] = (uint32_t)i;
        }
    }
}

template <int NW>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds, uint32_t &total) {
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
    for (int w = 0; w < NW; ++w) {
        uint32_t s = lds[w];
        if (w < wid) woff += s;
        tot += s;
    }
    __syncthreads();
    total = tot;
    return woff + incl - v;
}

struct DecodeArgs {
    const float *head;
    const uint32_t *keys;
    const uint32_t *state;
    const unsigned long long *gcand;
    const uint32_t *ties;
    float *boxes, *scores;
    int *labels, *counts;
    int hw, w, ncls, k, stride;
    float score_thresh;
    float lim[6], lo[3], vs[3];
};

constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_MAXK = 1024;

// one workgroup per batch item: gathers the <= K candidates found by the radix select (ties at the
// threshold value resolved to the smallest flat indices), bitonic sort (score desc, index asc), decode,
// masks, ordered compaction.
__global__ __launch_bounds__(TOPK_THREADS) void k_topk_decode(DecodeArgs a) {
    __shared__ uint32_t scan_lds[TOPK_THREADS / 64];
    __shared__ unsigned long long cand[TOPK_MAXK];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = a.ncls * a.hw;
    const uint32_t *keys = a.keys + (size_t)b * n;
    const int K = min(a.k, n);
    const uint32_t *st = a.state + (size_t)b * TOPK_STATE_WORDS;
    const uint32_t T = st[ST_PREFIX];
    const uint32_t need_eq = st[ST_NEED];          // number of keys == T to take (smallest indices first)
    const uint32_t n_gt = st[ST_NGT];              // == K - need_eq
    const uint32_t n_eq = st[ST_NEQ];              // keys == T in the whole map

    cand[tid] = (tid < (int)n_gt && tid < TOPK_MAXK) ? a.gcand[(size_t)b * TOPK_MAXK + tid] : 0ull;
    __syncthreads();
    if (n_eq == need_eq && n_eq <= (uint32_t)TIE_CAP) {
        // common case: every key equal to the threshold is selected, no ordering question
        for (uint32_t j = tid; j < need_eq; j += TOPK_THREADS) {
            const uint32_t pos = n_gt + j;
            if (pos < TOPK_MAXK)
                cand[pos] = ((unsigned long long)T << 32) | (uint32_t)(0xFFFFFFFFu - a.ties[(size_t)b * TIE_CAP + j]);
        }
    } else {
        // more ties than needed (flat / saturated score maps): take the first need_eq in index order
        uint32_t taken = 0u;
        for (int base = 0; base < n && taken < need_eq; base += TOPK_THREADS) {
            const int i = base + tid;
            const uint32_t flag = (i < n && keys[i] == T) ? 1u : 0u;
            uint32_t tot;
            const uint32_t ex = block_excl_scan<TOPK_THREADS / 64>(flag, scan_lds, tot);
            if (flag && taken + ex < need_eq) {
                const uint32_t pos = n_gt + taken + ex;
                if (pos < TOPK_MAXK) cand[pos] = ((unsigned long long)T << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
            }
            taken += tot;
        }
    }
    __syncthreads();

    // ---- bitonic sort of TOPK_MAXK 64-bit keys, descending (unused slots are 0 -> sink to the end)
    for (int size = 2; size <= TOPK_MAXK; size <<= 1) {
        for (int strd = size >> 1; strd > 0; strd >>= 1) {
            const int i = tid;
            const int j = i ^ strd;
            if (j > i) {
                const unsigned long long x = cand[i], y = cand[j];
                const bool desc = ((i & size) == 0);
                if (desc ? (x < y) : (x > y)) { cand[i] = y; cand[j] = x; }
            }
            __syncthreads();
        }
    }

    // ---- decode + masks + ordered compaction
    float box[7];
    float score = 0.f;
    int label = 0;
    uint32_t pass = 0u;
    if (tid < K) {
        const unsigned long long e = cand[tid];
        const uint32_t kv = (uint32_t)(e >> 32);
        const uint32_t flat = 0xFFFFFFFFu - (uint32_t)(e & 0xFFFFFFFFull);
        score = __uint_as_float(kv);
        label = (int)(flat / (uint32_t)a.hw);
        const int pix = (int)(flat % (uint32_t)a.hw);
        const float ys = (float)(pix / a.w), xs = (float)(pix % a.w);
        const float *row = a.head + ((size_t)b * a.hw + pix) * HEAD_COLS;
        const float fx = xs + row[0], fy = ys + row[1];
        box[0] = fx * (float)a.stride * a.vs[0] + a.lo[0];
        box[1] = fy * (float)a.stride * a.vs[1] + a.lo[1];
        box[2] = row[2];
        box[3] = expf(row[3]); box[4] = expf(row[4]); box[5] = expf(row[5]);
        box[6] = atan2f(row[7], row[6]);    // rot[:,1] = sin, rot[:,0] = cos (center_head.py:330-331)
        bool ok = box[0] >= a.lim[0] && box[1] >= a.lim[1] && box[2] >= a.lim[2] && box[0] <= a.lim[3] &&
                  box[1] <= a.lim[4] && box[2] <= a.lim[5];
        ok = ok && (score > a.score_thresh);
        pass = ok ? 1u : 0u;
    }
    uint32_t tot;
    const uint32_t pos = block_excl_scan<TOPK_THREADS / 64>(pass, scan_lds, tot);
    if (pass) {
        float *bo = a.boxes + ((size_t)b * a.k + pos) * 7;
        for (int q = 0; q < 7; ++q) bo[q] = box[q];
        a.scores[(size_t)b * a.k + pos] = score;
        a.labels[(size_t)b * a.k + pos] = label;
    }
    if (tid == 0) a.counts[b] = (int)tot;
}

// ------------------------------------------------------------------------------------------
// points in boxes (refiner crop)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_points_in_boxes(const float *__restrict__ boxes, const float *__restrict__ pts,
                                                         int t, int m, int *__restrict__ mask) {
    // block = 256 points of one batch item; boxes staged in LDS in chunks of 64 with cos/sin hoisted
    __shared__ float sb[64 * 9];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float *bx = boxes + (size_t)b * t * 7;
    float x = 0.f, y = 0.f, z = 0.f;
    if (i < m) {
        const float *p = pts + ((size_t)b * m + i) * 3;
        x = p[0]; y = p[1]; z = p[2];
    }
    for (int base = 0; base < t; base += 64) {
        const int nb = min(64, t - base);
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            const float *q = bx + (size_t)(base + threadIdx.x) * 7;
            float *d = sb + threadIdx.x * 9;
            for (int j = 0; j < 7; ++j) d[j] = q[j];
            d[7] = cosf(-q[6]);
            d[8] = sinf(-q[6]);
        }
        __syncthreads();
        if (i < m) {
            for (int k = 0; k < nb; ++k) {
                const float *q = sb + k * 9;
                int in = 0;
                if (!((double)fabsf(z - q[2]) > (double)q[5] / 2.0)) {
                    const float sx = x - q[0], sy = y - q[1];
                    const float lx = sx * q[7] + sy * (-q[8]);
                    const float ly = sx * q[8] + sy * q[7];
                    const bool inx = (double)fabsf(lx) < (double)q[3] / 2.0 + (double)1e-5f;
                    const bool iny = (double)fabsf(ly) < (double)q[4] / 2.0 + (double)1e-5f;
                    in = (inx && iny) ? 1 : 0;
                }
                mask[((size_t)b * t + base + k) * m + i] = in;
            }
        }
    }
}

// ---- object crop with compaction (daemon/prepare_object_data.py:250-273,310) -------------------------------------
// The reference builds the dense (T, M) int mask, copies it to the host and boolean-indexes the point array once per
// object.  Here the mask is a BITMAP (T rows of M bits, one __ballot per wavefront and box), the bitmap scan of
// sparse_index.hip ranks its set bits in (box, point) order - exactly the order of `[pts[mask[i]] for i in boxes]` -
// and one gather writes every object's points back to back; per-object offsets are the ranks at the row starts.
__global__ __launch_bounds__(256) void k_points_in_boxes_bits(const float *__restrict__ boxes, const float *__restrict__ pts,
                                                              int t, int m, int row_words, uint32_t *__restrict__ bitmap) {
    __shared__ float sb[64 * 9];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float x = 0.f, y = 0.f, z = 0.f;
    if (i < m) { x = pts[(size_t)i * 3]; y = pts[(size_t)i * 3 + 1]; z = pts[(size_t)i * 3 + 2]; }
    for (int base = 0; base < t; base += 64) {
        const int nb = min(64, t - base);
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            const float *q = boxes + (size_t)(base + threadIdx.x) * 7;
            float *d = sb + threadIdx.x * 9;
            for (int j = 0; j < 7; ++j) d[j] = q[j];
            d[7] = cosf(-q[6]);
            d[8] = sinf(-q[6]);
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            const float *q = sb + k * 9;
            bool in = false;
            if (i < m && !((double)fabsf(z - q[2]) > (double)q[5] / 2.0)) {       // same test as k_points_in_boxes
                const float sx = x - q[0], sy = y - q[1];
                const float lx = sx * q[7] + sy * (-q[8]);
                const float ly = sx * q[8] + sy * q[7];
                in = ((double)fabsf(lx) < (double)q[3] / 2.0 + (double)1e-5f) && ((double)fabsf(ly) < (double)q[4] / 2.0 + (double)1e-5f);
            }
            const unsigned long long bits = __ballot(in);
            if (lane == 0) {
                uint32_t *w = bitmap + (size_t)(base + k) * row_words + (i >> 5);
                w[0] = (uint32_t)bits;
                w[1] = (uint32_t)(bits >> 32);
            }
        }
    }
}

// points per box (roiaware_pool3d's points_in_boxes_num as tracking/.../data_processor.py:64-69 uses it): the bitmap kernel's
// test, one popcount of the ballot per wavefront and box, one atomicAdd per wavefront that saw a point - no (T, M) mask at all
__global__ __launch_bounds__(256) void k_points_in_boxes_count(const float *__restrict__ boxes, const float *__restrict__ pts,
                                                               int t, int m, int *__restrict__ counts) {
    __shared__ float sb[64 * 9];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    float x = 0.f, y = 0.f, z = 0.f;
    if (i < m) { x = pts[(size_t)i * 3]; y = pts[(size_t)i * 3 + 1]; z = pts[(size_t)i * 3 + 2]; }
    for (int base = 0; base < t; base += 64) {
        const int nb = min(64, t - base);
        __syncthreads();
        if ((int)threadIdx.x < nb) {
            const float *q = boxes + (size_t)(base + threadIdx.x) * 7;
            float *d = sb + threadIdx.x * 9;
            for (int j = 0; j < 7; ++j) d[j] = q[j];
            d[7] = cosf(-q[6]);
            d[8] = sinf(-q[6]);
        }
        __syncthreads();
        for (int k = 0; k < nb; ++k) {
            const float *q = sb + k * 9;
            bool in = false;
            if (i < m && !((double)fabsf(z - q[2]) > (double)q[5] / 2.0)) {       // same test as k_points_in_boxes
                const float sx = x - q[0], sy = y - q[1];
                const float lx = sx * q[7] + sy * (-q[8]);
                const float ly = sx * q[8] + sy * q[7];
                in = ((double)fabsf(lx) < (double)q[3] / 2.0 + (double)1e-5f) && ((double)fabsf(ly) < (double)q[4] / 2.0 + (double)1e-5f);
            }
            const unsigned long long bits = __ballot(in);
            if (lane == 0 && bits) atomicAdd(&counts[base + k], __popcll(bits));
        }
    }
}

// one thread per (kept point, payload word)
__global__ void k_crop_gather(const int *__restrict__ pairs, const int *__restrict__ d_total, int cap, const uint32_t *__restrict__ payload,
                              int words, uint32_t *__restrict__ out, int *__restrict__ out_index, const uint32_t *__restrict__ prefix,
                              int t, int row_words, int *__restrict__ offsets) {
    const int total = min(*d_total, cap);
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid <= t) offsets[gid] = gid < t ? (int)prefix[(size_t)gid * row_words] : *d_total;
    for (long idx = gid; idx < (long)total * words; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / words), wq = (int)(idx % words);
        const int pt = pairs[(size_t)r * 4 + 3];                 // [box, 0, 0, point]
        out[idx] = payload[(size_t)pt * words + wq];
        if (wq == 0 && out_index) out_index[r] = pt;
    }
}

}  // namespace dz

using namespace dz;

// ------------------------------------------------------------------------------------------------ RoI features from the BEV map
// center_head.py:408-432,461-486 (get_box_center with num_point = 5, absl_to_relative, centernet_utils.bilinear_interpolate_torch:233-262):
// per first-stage box the BEV features at its centre and at the middles of its front / back / left / right edges, bilinear with
// the reference's clamped corner indices (the weights use the CLAMPED x1 / y1, as the reference does), concatenated channel-wise.
__global__ __launch_bounds__(256) void k_roi_bev_features(const float *__restrict__ boxes, int n, const float *__restrict__ bev, long row_stride,
                                                          long pix_stride, int h, int w, int c, float x_lo, float y_lo, float sx, float sy,
                                                          float fstride, float *__restrict__ out) {
    const int i = blockIdx.x;
    if (i >= n) return;
    const float *b = boxes + (size_t)i * 7;
    const float cx = b[0], cy = b[1], hx = b[3] * 0.5f, hy = b[4] * 0.5f;
    const float cs = cosf(b[6]), sn = sinf(b[6]);
    // corners 0..3 of box_utils.boxes_to_corners_3d (template (+,+), (+,-), (-,-), (-,+)) rotated about z and shifted
    const float tx[4] = {hx, hx, -hx, -hx}, ty[4] = {hy, -hy, -hy, hy};
    float kx[4], ky[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        kx[k] = __fadd_rn(__fadd_rn(__fmul_rn(tx[k], cs), __fmul_rn(ty[k], -sn)), cx);
        ky[k] = __fadd_rn(__fadd_rn(__fmul_rn(tx[k], sn), __fmul_rn(ty[k], cs)), cy);
    }
    const float px[5] = {cx, (kx[0] + kx[1]) / 2.f, (kx[2] + kx[3]) / 2.f, (kx[0] + kx[3]) / 2.f, (kx[1] + kx[2]) / 2.f};
    const float py[5] = {cy, (ky[0] + ky[1]) / 2.f, (ky[2] + ky[3]) / 2.f, (ky[0] + ky[3]) / 2.f, (ky[1] + ky[2]) / 2.f};
#pragma unroll
    for (int p = 0; p < 5; ++p) {
        const float x = __fdiv_rn(__fdiv_rn(__fsub_rn(px[p], x_lo), sx), fstride), y = __fdiv_rn(__fdiv_rn(__fsub_rn(py[p], y_lo), sy), fstride);
        const float fx = floorf(x), fy = floorf(y);
        const long x0 = min(max((long)fx, 0l), (long)w - 1), x1 = min(max((long)fx + 1, 0l), (long)w - 1);
        const long y0 = min(max((long)fy, 0l), (long)h - 1), y1 = min(max((long)fy + 1, 0l), (long)h - 1);
        const float wa = __fmul_rn((float)x1 - x, (float)y1 - y), wb = __fmul_rn((float)x1 - x, y - (float)y0);
        const float wc = __fmul_rn(x - (float)x0, (float)y1 - y), wd = __fmul_rn(x - (float)x0, y - (float)y0);
        const float *ia = bev + y0 * row_stride + x0 * pix_stride, *ib = bev + y1 * row_stride + x0 * pix_stride;
        const float *ic = bev + y0 * row_stride + x1 * pix_stride, *id = bev + y1 * row_stride + x1 * pix_stride;
        for (int ch = threadIdx.x; ch < c; ch += 256)
            out[((size_t)i * 5 + p) * c + ch] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(ia[ch], wa), __fmul_rn(ib[ch], wb)), __fmul_rn(ic[ch], wc)), __fmul_rn(id[ch], wd));
    }
}

extern "C" {

static size_t crop_layout(int m, int t, int cap, size_t *o_pf, size_t *o_pairs, size_t *o_sw, size_t *sw_bytes, int *row_words) {
    const int rw = (int)align_up(((size_t)(m < 1 ? 1 : m) + 31) / 32, 8);          // 256-bit rows: rows stay word aligned
    *row_words = rw;
    const size_t nwords = (size_t)(t < 1 ? 1 : t) * rw;
    size_t off = align_up(nwords * 4, 256);
    *o_pf = off; off += align_up(nwords * 4, 256);
    *o_pairs = off; off += align_up((size_t)(cap < 1 ? 1 : cap) * 16, 256);
    *sw_bytes = bitmap_scan_workspace_bytes(nwords);
    *o_sw = off; off += align_up(*sw_bytes, 256);
    return off;
}

size_t dz_crop_points_workspace_bytes(int m, int t, int cap) {
    size_t a, b, c, d; int rw;
    return crop_layout(m, t, cap, &a, &b, &c, &d, &rw);
}

int dz_crop_points_in_boxes(const float *xyz, int m, const float *boxes, int t, const void *payload, int payload_words, void *out,
                            int *out_index, int *offsets, int *d_total, int cap, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(m >= 0 && t >= 0 && cap >= 0 && payload_words >= 1, "dz_crop_points_in_boxes: bad sizes");
    DZ_CHECK_ARG(offsets && d_total && ws, "dz_crop_points_in_boxes: null pointer");
    DZ_CHECK_ARG((size_t)t * (((size_t)m + 255) / 256 * 256) < 0xFFFFFFFFull, "dz_crop_points_in_boxes: boxes x points exceed 2^32 mask bits");
    size_t o_pf, o_pairs, o_sw, sw_bytes; int rw;
    const size_t need = crop_layout(m, t, cap, &o_pf, &o_pairs, &o_sw, &sw_bytes, &rw);
    if (ws_bytes < need) { set_error("dz_crop_points_in_boxes: workspace %zu < %zu", ws_bytes, need); return DZ_ERR_WORKSPACE; }
    if (t == 0 || m == 0) {
        int rc = fill_u32(offsets, 0u, (size_t)t + 1, stream);
        return rc ? rc : fill_u32(d_total, 0u, 1, stream);
    }
    DZ_CHECK_ARG(xyz && boxes && payload && out, "dz_crop_points_in_boxes: null pointer");
    uint32_t *bitmap = (uint32_t *)ws, *prefix = (uint32_t *)((char *)ws + o_pf);
    int *pairs = (int *)((char *)ws + o_pairs);
    const size_t nwords = (size_t)t * rw;
    int rc = fill_u32(bitmap, 0u, nwords, stream);                      // words past the last point block of a row stay 0
    if (rc) return rc;
    hipLaunchKernelGGL(k_points_in_boxes_bits, dim3(ceil_div(m, 256)), dim3(256), 0, stream, boxes, xyz, t, m, rw, bitmap);
    rc = bitmap_scan(bitmap, nwords, prefix, d_total, 0, ScanDims{1, 1, rw * 32}, pairs, cap, (char *)ws + o_sw, sw_bytes, stream);
    if (rc) return rc;
    const long work = (long)cap * payload_words > t + 1 ? (long)cap * payload_words : t + 1;
    hipLaunchKernelGGL(k_crop_gather, dim3(stream_grid(work, 256)), dim3(256), 0, stream, pairs, d_total, cap, (const uint32_t *)payload,
                       payload_words, (uint32_t *)out, out_index, prefix, t, rw, offsets);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(na >= 0 && nb >= 0, "dz_boxes_overlap_bev: negative size");
    if (na == 0 || nb == 0) return DZ_OK;
    DZ_CHECK_ARG(a && b && out, "dz_boxes_overlap_bev: null pointer");
    hipLaunchKernelGGL(k_pairwise<false>, dim3(stream_grid((long)na * nb, PAIR_THREADS)), dim3(PAIR_THREADS), 0, stream, a, na, b, nb, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(na >= 0 && nb >= 0, "dz_boxes_iou_bev: negative size");
    if (na == 0 || nb == 0) return DZ_OK;
    DZ_CHECK_ARG(a && b && out, "dz_boxes_iou_bev: null pointer");
    hipLaunchKernelGGL(k_pairwise<true>, dim3(stream_grid((long)na * nb, PAIR_THREADS)), dim3(PAIR_THREADS), 0, stream, a, na, b, nb, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

size_t dz_nms_workspace_bytes(int n_cap) {
    const size_t cb = (size_t)(n_cap + 63) / 64;
    return align_up((size_t)(n_cap > 0 ? n_cap : 1) * (cb > 0 ? cb : 1) * sizeof(unsigned long long), 256);
}

int dz_nms_rotated_batched(const float *boxes, const int *d_n, int batch, int n_cap, float thresh, int post_max, int *keep,
                           int *d_num_keep, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(keep && d_num_keep && n_cap >= 0 && post_max >= 0 && batch >= 0, "dz_nms_rotated: bad argument");
    if (batch == 0) return DZ_OK;
    if (n_cap == 0) return fill_u32(d_num_keep, 0u, (size_t)batch, stream);
    DZ_CHECK_ARG(boxes && ws, "dz_nms_rotated: null pointer");
    DZ_CHECK_ARG(n_cap <= 4096, "dz_nms_rotated: n_cap %d > 4096 (NMS_PRE_MAXSIZE of the reference configs)", n_cap);
    DZ_CHECK_ARG(batch <= 65535, "dz_nms_rotated: batch %d > 65535", batch);
    if (ws_bytes < (size_t)batch * dz_nms_workspace_bytes(n_cap)) { set_error("dz_nms_rotated: workspace too small"); return DZ_ERR_WORKSPACE; }
    const int cb = (n_cap + 63) / 64;
    unsigned long long *mask = (unsigned long long *)ws;
    hipLaunchKernelGGL(k_nms_mask, dim3(cb, cb * (64 / NMS_ROWS_PER_WAVE), batch), dim3(64), 0, stream, boxes, d_n, n_cap, thresh,
                       mask, cb);
    hipLaunchKernelGGL(k_nms_sweep, dim3(batch), dim3(256), (size_t)64 * cb * sizeof(unsigned long long), stream, mask, d_n,
                       n_cap, cb, post_max, keep, d_num_keep);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_nms_rotated(const float *boxes, const int *d_n, int n_cap, float thresh, int post_max, int *keep, int *d_num_keep,
                   void *ws, size_t ws_bytes, void *stream_) {
    return dz_nms_rotated_batched(boxes, d_n, 1, n_cap, thresh, post_max, keep, d_num_keep, ws, ws_bytes, stream_);
}

int dz_pack_detections(const float *boxes, const float *scores, const int *labels, const int *keep, const int *d_num_keep,
                       int batch, int k, int post_max, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(batch >= 0 && k >= 0 && post_max >= 0, "dz_pack_detections: bad sizes");
    if (batch == 0 || post_max == 0) return DZ_OK;
    DZ_CHECK_ARG(out && d_num_keep && (k == 0 || (boxes && scores && labels && keep)), "dz_pack_detections: null pointer");
    hipLaunchKernelGGL(k_pack_detections, dim3(ceil_div(post_max * 9, 256), batch), dim3(256), 0, stream, boxes, scores, labels,
                       keep, d_num_keep, k, post_max, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

static size_t decode_layout(int batch, int hw, int ncls, size_t *o_keys, size_t *o_state, size_t *o_cand, size_t *o_ties) {
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += align_up(bytes, 256); return o; };
    *o_keys = take((size_t)batch * hw * ncls * sizeof(uint32_t));
    *o_state = take((size_t)batch * TOPK_STATE_WORDS * sizeof(uint32_t));
    *o_cand = take((size_t)batch * TOPK_MAXK * sizeof(unsigned long long));
    *o_ties = take((size_t)batch * TIE_CAP * sizeof(uint32_t));
    return off;
}

size_t dz_centerhead_decode_workspace_bytes(int batch, int hw, int ncls, int k) {
    (void)k;
    size_t a, b, c, d;
    return decode_layout(batch, hw, ncls, &a, &b, &c, &d);
}

int dz_centerhead_decode(const float *head, int batch, int h, int w, int ncls, int k, float score_thresh,
                         const float *h_limit6, const float *h_range6, const float *h_vsize3, int stride, int use_iou,
                         float *boxes, float *scores, int *labels, int *d_counts, void *ws, size_t ws_bytes,
                         void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(head && boxes && scores && labels && d_counts && ws, "dz_centerhead_decode: null pointer");
    DZ_CHECK_ARG(batch >= 1 && h >= 1 && w >= 1 && ncls >= 1 && ncls <= 3, "dz_centerhead_decode: bad sizes (ncls<=3)");
    DZ_CHECK_ARG(k >= 1 && k <= TOPK_MAXK, "dz_centerhead_decode: K %d not in [1,%d]", k, TOPK_MAXK);
    const int hw = h * w;
    if (ws_bytes < dz_centerhead_decode_workspace_bytes(batch, hw, ncls, k)) {
        set_error("dz_centerhead_decode: workspace too small");
        return DZ_ERR_WORKSPACE;
    }
    size_t o_keys, o_state, o_cand, o_ties;
    decode_layout(batch, hw, ncls, &o_keys, &o_state, &o_cand, &o_ties);
    uint32_t *keys = (uint32_t *)((char *)ws + o_keys);
    uint32_t *state = (uint32_t *)((char *)ws + o_state);
    unsigned long long *gcand = (unsigned long long *)((char *)ws + o_cand);
    uint32_t *ties = (uint32_t *)((char *)ws + o_ties);
    const int n = ncls * hw;
    int rc = fill_u32(state, 0u, (size_t)batch * TOPK_STATE_WORDS, stream);
    if (rc) return rc;
    const dim3 gpix(stream_grid(hw, 256) > 512 ? 512 : stream_grid(hw, 256), batch);
    const dim3 gkey(stream_grid(n, 256) > 512 ? 512 : stream_grid(n, 256), batch);
    hipLaunchKernelGGL(k_score_hist0, gpix, dim3(256), 0, stream, head, hw, ncls, use_iou, keys, state);
    hipLaunchKernelGGL(k_radix_pick, dim3(batch), dim3(256), 0, stream, state, 0, k, n);
    hipLaunchKernelGGL(k_radix_hist<1>, gkey, dim3(256), 0, stream, keys, n, state);
    hipLaunchKernelGGL(k_radix_pick, dim3(batch), dim3(256), 0, stream, state, 1, k, n);
    hipLaunchKernelGGL(k_radix_hist<2>, gkey, dim3(256), 0, stream, keys, n, state);
    hipLaunchKernelGGL(k_radix_pick, dim3(batch), dim3(256), 0, stream, state, 2, k, n);
    hipLaunchKernelGGL(k_topk_collect, gkey, dim3(256), 0, stream, keys, n, state, gcand, ties, TOPK_MAXK);
    DecodeArgs a;
    a.head = head; a.keys = keys; a.state = state; a.gcand = gcand; a.ties = ties; a.boxes = boxes; a.scores = scores; a.labels = labels; a.counts = d_counts;
    a.hw = hw; a.w = w; a.ncls = ncls; a.k = k; a.stride = stride; a.score_thresh = score_thresh;
    for (int i = 0; i < 6; ++i) a.lim[i] = h_limit6[i];
    for (int i = 0; i < 3; ++i) { a.lo[i] = h_range6[i]; a.vs[i] = h_vsize3[i]; }
    hipLaunchKernelGGL(k_topk_decode, dim3(batch), dim3(TOPK_THREADS), 0, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_points_in_boxes_count(const float *boxes, const float *pts, int t, int m, int *counts, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(t >= 0 && m >= 0, "dz_points_in_boxes_count: negative size");
    if (t == 0) return DZ_OK;
    DZ_CHECK_ARG(boxes && counts && (pts || m == 0), "dz_points_in_boxes_count: null pointer");
    const int rc = fill_u32(counts, 0u, (size_t)t, stream);
    if (rc) return rc;
    if (m == 0) return DZ_OK;
    hipLaunchKernelGGL(k_points_in_boxes_count, dim3(ceil_div(m, 256)), dim3(256), 0, stream, boxes, pts, t, m, counts);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_points_in_boxes_v2(const float *boxes, const float *pts, int batch, int t, int m, int *mask, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(batch >= 0 && t >= 0 && m >= 0, "dz_points_in_boxes_v2: negative size");
    if (batch == 0 || t == 0 || m == 0) return DZ_OK;
    DZ_CHECK_ARG(boxes && pts && mask, "dz_points_in_boxes_v2: null pointer");
    hipLaunchKernelGGL(k_points_in_boxes, dim3(ceil_div(m, 256), batch), dim3(256), 0, stream, boxes, pts, t, m, mask);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_roi_bev_features(const float *boxes, int n, const float *bev, long row_stride, long pix_stride, int h, int w, int c, float x_lo, float y_lo,
                        float voxel_x, float voxel_y, int stride, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 0 && h >= 1 && w >= 1 && c >= 1 && voxel_x > 0.f && voxel_y > 0.f && stride >= 1, "dz_roi_bev_features: bad sizes");
    if (n == 0) return DZ_OK;
    DZ_CHECK_ARG(boxes && bev && out, "dz_roi_bev_features: null pointer");
    hipLaunchKernelGGL(k_roi_bev_features, dim3(n), dim3(256), 0, stream, boxes, n, bev, row_stride, pix_stride, h, w, c, x_lo, y_lo, voxel_x, voxel_y, (float)stride, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
