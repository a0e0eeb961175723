// Test-time augmentation on the device: the point-side transforms of the augmented
// copies, the restore of their boxes to the original frame, and weighted box fusion of all copies' detections.
//
// Reference:
//   detection/detzero_det/datasets/augmentor/test_time_augmentor.py:32-83       (flip / rotation / scaling of the points)
//   detection/detzero_det/models/centerpoint.py:131-208                          (CenterPoint.test_time_augment)
//   detection/detzero_det/utils/ensemble_utils/wbf_3d.py:10-203, ensemble.py:7-33 (weighted_boxes_fusion_3d, wbf_online)
//   utils/detzero_utils/ops/iou3d_nms/iou3d_nms_utils.py:74-107                  (boxes_iou3d_gpu)
//
// The reference fuses on the host: per candidate box one IoU kernel launch, a device sync and a numpy update.  Here one
// workgroup per (frame, class) walks its candidates in score order; the IoUs against the clusters built so far are
// evaluated across the workgroup, the best match is reduced, and one lane applies the reference's update rule with the
// reference's rounding (float32 accumulator fed by float64 products, float64 confidence sums) - no host round trips.
#include "common.h"

#pragma clang fp contract(off)

#include "box_geom.h"

namespace dz {

constexpr int TTA_MAX_OPS = 32;
enum { TTA_ORIGINAL = 0, TTA_FLIP_X = 1, TTA_FLIP_Y = 2, TTA_FLIP_XY = 3, TTA_ROT = 4, TTA_SCALE = 5 };

struct TtaOps {
    int n;
    int kind[TTA_MAX_OPS];
    float p0[TTA_MAX_OPS], p1[TTA_MAX_OPS];     // rot: cos, sin of the float32 angle; scale: factor; restore-rot: p0/p1 + angle in p2
    float p2[TTA_MAX_OPS];
};

__global__ __launch_bounds__(256) void k_tta_points(const float *__restrict__ pts, int n, int c, TtaOps ops, float *__restrict__ out) {
    const long total = (long)ops.n * n;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (long)gridDim.x * blockDim.x) {
        const int o = (int)(r / n);
        const float *p = pts + (size_t)(r - (long)o * n) * c;
        float *q = out + (size_t)r * c;
        float x = p[0], y = p[1], z = p[2];
        switch (ops.kind[o]) {
        case TTA_FLIP_X: y = -y; break;
        case TTA_FLIP_Y: x = -x; break;
        case TTA_FLIP_XY: x = -x; y = -y; break;
        case TTA_ROT: {                                  // points @ [[c, s, 0], [-s, c, 0], [0, 0, 1]] (common_utils.py:237-242)
            const float cs = ops.p0[o], sn = ops.p1[o];
            const float nx = x * cs + y * (-sn), ny = x * sn + y * cs;
            x = nx; y = ny;
            break;
        }
        case TTA_SCALE: x *= ops.p0[o]; y *= ops.p0[o]; z *= ops.p0[o]; break;
        default: break;
        }
        q[0] = x; q[1] = y; q[2] = z;
        for (int k = 3; k < c; ++k) q[k] = p[k];
    }
}

// boxes (F, T, M, dim >= 7) in place; rows of copy i come back to the original frame (centerpoint.py:165-203, 7-dim boxes)
__global__ __launch_bounds__(256) void k_tta_restore(float *__restrict__ boxes, int frames, int m, int dim, TtaOps ops) {
    const long total = (long)frames * ops.n * m;
    const float pi = 3.14159265358979323846f;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (long)gridDim.x * blockDim.x) {
        const int o = (int)((r / m) % ops.n);
        float *b = boxes + (size_t)r * dim;
        switch (ops.kind[o]) {
        case TTA_FLIP_X: b[1] = -b[1]; b[6] = -b[6]; break;
        case TTA_FLIP_Y: b[0] = -b[0]; b[6] = -(b[6] + pi); break;
        case TTA_FLIP_XY: b[0] = -b[0]; b[1] = -b[1]; b[6] = b[6] + pi; break;
        case TTA_ROT: {
            const float cs = ops.p0[o], sn = ops.p1[o];
            const float x = b[0], y = b[1];
            b[0] = x * cs + y * (-sn);
            b[1] = x * sn + y * cs;
            b[6] = b[6] + ops.p2[o];
            break;
        }
        case TTA_SCALE:
            for (int k = 0; k < 6; ++k) b[k] = b[k] / ops.p0[o];
            break;
        default: break;
        }
    }
}

// ---- weighted box fusion ----------------------------------------------------------------------------------------
struct WbfParams {
    int frames, cand, per_model, n_models, conf_max, allows_overflow;
    double iou_thr[3], skip_thr[3], wsum;
};

// workspace of one (frame, class) slot; `cap` = candidates per frame
struct WbfSlot {
    int *sel_idx, *order, *c_cnt, *c_first, *c_obj, *c_objc, *counts;     // c_objc: candidate lending the cluster its object id; counts: [0] selected, [1] clusters, [2] first candidate of the class
    double *sel_score, *c_conf, *c_score;
    float *c_acc, *c_fused;
};

__host__ __device__ inline size_t wbf_slot_bytes(int cap) {
    return (size_t)cap * (6 * sizeof(int) + 3 * sizeof(double) + 14 * sizeof(float)) + 64;      // (a multiple of 8: slots stay 8-byte aligned)
}

__device__ __forceinline__ WbfSlot wbf_slot(void *ws, int slot, int cap) {
    unsigned char *p = reinterpret_cast<unsigned char *>(ws) + (size_t)slot * wbf_slot_bytes(cap);
    WbfSlot s;
    s.sel_score = reinterpret_cast<double *>(p); p += (size_t)cap * sizeof(double);
    s.c_conf = reinterpret_cast<double *>(p); p += (size_t)cap * sizeof(double);
    s.c_score = reinterpret_cast<double *>(p); p += (size_t)cap * sizeof(double);
    s.counts = reinterpret_cast<int *>(p); p += 64;
    s.sel_idx = reinterpret_cast<int *>(p); p += (size_t)cap * sizeof(int);
    s.order = reinterpret_cast<int *>(p); p += (size_t)cap * sizeof(int);
    s.c_cnt = reinterpret_cast<int *>(p); p += (size_t)cap * sizeof(int);
    s.c_first = reinterpret_cast<int *>(p); p += (size_t)cap * sizeof(int);
    s.c_obj = reinterpret_cast<int *>(p); p += (size_t)cap * sizeof(int);
    s.c_objc = reinterpret_cast<int *>(p); p += (size_t)cap * sizeof(int);
    s.c_acc = reinterpret_cast<float *>(p); p += (size_t)cap * 7 * sizeof(float);
    s.c_fused = reinterpret_cast<float *>(p);
    return s;
}

// prefilter_boxes (wbf_3d.py:10-51): candidates of the class with score x weight >= threshold, in descending score order
// (equal scores: later candidate first - numpy's argsort()[::-1] leaves that order unspecified)
__global__ __launch_bounds__(256) void k_wbf_rank(const float *__restrict__ scores, const int *__restrict__ labels,
                                                  const double *__restrict__ weights, WbfParams p, void *ws) {
    const int l = blockIdx.x, f = blockIdx.y;
    const WbfSlot s = wbf_slot(ws, f * 3 + l, p.cand);
    __shared__ int n_s, first_s;
    if (threadIdx.x == 0) { n_s = 0; first_s = 0x7fffffff; }
    __syncthreads();
    for (int c = threadIdx.x; c < p.cand; c += blockDim.x) {
        if (labels[(size_t)f * p.cand + c] != l + 1) continue;
        atomicMin(&first_s, c);
        const double sc = (double)scores[(size_t)f * p.cand + c] * (weights ? weights[c / p.per_model] : 1.0);
        if (sc >= p.skip_thr[l]) {
            const int q = atomicAdd(&n_s, 1);
            s.sel_idx[q] = c;
            s.sel_score[q] = sc;
        }
    }
    __syncthreads();
    const int n = n_s;
    for (int q = threadIdx.x; q < n; q += blockDim.x) {
        const double sc = s.sel_score[q];
        const int c = s.sel_idx[q];
        int r = 0;
        for (int k = 0; k < n; ++k) {
            const double s2 = s.sel_score[k];
            r += (s2 > sc || (s2 == sc && s.sel_idx[k] > c)) ? 1 : 0;
        }
        s.order[r] = c;
    }
    if (threadIdx.x == 0) { s.counts[0] = n; s.counts[1] = 0; s.counts[2] = first_s; }
}

// boxes_iou3d_gpu (iou3d_nms_utils.py:74-107), one pair, float32 operation by operation
__device__ __forceinline__ float iou3d_pair(const float *a, const float *b, P2 *cp, float *ang, int ld) {
    const float a_max = a[2] + a[5] / 2, a_min = a[2] - a[5] / 2, b_max = b[2] + b[5] / 2, b_min = b[2] - b[5] / 2;
    const float bev = rect_overlap(a, b, cp, ang, ld);
    const float h = fmaxf(fminf(a_max, b_max) - fmaxf(a_min, b_min), 0.f);
    const float o3 = bev * h;
    const float vol_a = a[3] * a[4] * a[5], vol_b = b[3] * b[4] * b[5];
    return o3 / fmaxf(vol_a + vol_b - o3, 1e-6f);
}

constexpr int WBF_THREADS = 256;

// the clustering loop of weighted_boxes_fusion_3d (wbf_3d.py:168-190) for one (frame, class)
__global__ __launch_bounds__(WBF_THREADS) void k_wbf_cluster(const float *__restrict__ boxes, const float *__restrict__ scores,
                                                             const double *__restrict__ weights, const int *__restrict__ obj_ids,
                                                             WbfParams p, void *ws) {
    const int l = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const WbfSlot s = wbf_slot(ws, f * 3 + l, p.cand);
    __shared__ P2 cp_s[16 * WBF_THREADS];
    __shared__ float ang_s[16 * WBF_THREADS];
    __shared__ float red_iou[WBF_THREADS / 64];
    __shared__ int red_idx[WBF_THREADS / 64];
    __shared__ float bj_s[8];
    __shared__ int ncl_s;
    if (tid == 0) ncl_s = 0;
    const int n = s.counts[0];
    const double thr_d = p.iou_thr[l];          // compared as in the reference: the float32 IoU, as a double, against the double threshold
    for (int j = 0; j < n; ++j) {
        const int cand = s.order[j];
        if (tid < 7) bj_s[tid] = boxes[((size_t)f * p.cand + cand) * 7 + tid];
        __syncthreads();                                                   // bj_s, ncl_s and the previous update are visible
        const int ncl = ncl_s;
        float bj[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) bj[k] = bj_s[k];
        float best = -1.f;
        int best_idx = 0x7fffffff;
        for (int c = tid; c < ncl; c += WBF_THREADS) {
            const float iou = iou3d_pair(bj, s.c_fused + (size_t)c * 7, cp_s + tid, ang_s + tid, WBF_THREADS);
            if (iou > best) { best = iou; best_idx = c; }                  // first maximum (torch.argmax)
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float o_iou = __shfl_xor(best, d, 64);
            const int o_idx = __shfl_xor(best_idx, d, 64);
            if (o_iou > best || (o_iou == best && o_idx < best_idx)) { best = o_iou; best_idx = o_idx; }
        }
        if ((tid & 63) == 0) { red_iou[tid >> 6] = best; red_idx[tid >> 6] = best_idx; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < WBF_THREADS / 64; ++w)
                if (red_iou[w] > best || (red_iou[w] == best && red_idx[w] < best_idx)) { best = red_iou[w]; best_idx = red_idx[w]; }
            const double sj = (double)scores[(size_t)f * p.cand + cand] * (weights ? weights[cand / p.per_model] : 1.0);
            if (ncl == 0 || !((double)best > thr_d)) {                      // new cluster: the candidate itself
                float *acc = s.c_acc + (size_t)ncl * 7, *fu = s.c_fused + (size_t)ncl * 7;
                for (int k = 0; k < 7; ++k) { acc[k] = (float)(sj * (double)bj[k]); fu[k] = bj[k]; }
                s.c_conf[ncl] = sj; s.c_cnt[ncl] = 1; s.c_first[ncl] = cand; s.c_score[ncl] = sj;
                s.c_obj[ncl] = obj_ids ? obj_ids[(size_t)f * p.cand + cand] : -1;
                s.c_objc[ncl] = cand;
                ncl_s = ncl + 1;
            } else {                                                        // get_weighted_box (wbf_3d.py:53-96)
                const int c = best_idx;
                float *acc = s.c_acc + (size_t)c * 7, *fu = s.c_fused + (size_t)c * 7;
                const double conf = s.c_conf[c] + sj;
                const int cnt = s.c_cnt[c] + 1;
                for (int k = 0; k < 6; ++k) {
                    acc[k] = (float)((double)acc[k] + sj * (double)bj[k]);
                    fu[k] = (float)((double)acc[k] / conf);
                }
                // heading and 'max' confidence: the member with the largest confidence = the first one (score order)
                const int first = s.c_first[c];
                fu[6] = boxes[((size_t)f * p.cand + first) * 7 + 6];
                s.c_conf[c] = conf; s.c_cnt[c] = cnt;
                // object id (weighted_tracking_boxes_fusion_3d, wbf_3d.py:86-94): the most confident member that has one; members
                // are ordered by np.argsort(conf)[::-1], so among EQUAL confidences the member appended last wins
                if (obj_ids) {
                    const int id_new = obj_ids[(size_t)f * p.cand + cand];
                    if (id_new >= 0) {
                        const int holder = s.c_objc[c];
                        const double holder_sc = (double)scores[(size_t)f * p.cand + holder] * (weights ? weights[holder / p.per_model] : 1.0);
                        if (s.c_obj[c] < 0 || sj == holder_sc) { s.c_obj[c] = id_new; s.c_objc[c] = cand; }
                    }
                }
                const double first_sc = (double)scores[(size_t)f * p.cand + first] * (weights ? weights[first / p.per_model] : 1.0);
                s.c_score[c] = (double)(float)(p.conf_max ? first_sc : conf / cnt);
            }
        }
    }
    __syncthreads();
    // confidence rescale (wbf_3d.py:186-190), with the promotion rules of NumPy >= 2 for the float32 fused rows
    const int ncl = ncl_s;
    for (int c = tid; c < ncl; c += WBF_THREADS) {
        const int cnt = s.c_cnt[c];
        double sc = s.c_score[c];
        if (cnt == 1) {
            const double mn = p.allows_overflow ? 1.0 : ((double)cnt < p.wsum ? 1.0 : p.wsum);
            sc = sc * mn / p.wsum;
        } else {
            const float s32 = (float)sc;
            double t;
            if (p.allows_overflow || (double)cnt < p.wsum) t = (double)(s32 * (float)cnt);      // float32 x Python int
            else t = (double)s32 * p.wsum;                                                       // float32 x float64
            sc = (double)(float)(t / p.wsum);
        }
        s.c_score[c] = sc;
    }
    if (tid == 0) s.counts[1] = ncl;
}

// all classes of a frame, sorted by fused score (wbf_3d.py:198-203)
__global__ __launch_bounds__(256) void k_wbf_emit(const float *__restrict__ boxes, WbfParams p, void *ws, double *__restrict__ out_boxes,
                                                  double *__restrict__ out_scores, int *__restrict__ out_labels, int *__restrict__ out_obj,
                                                  int *__restrict__ out_count) {
    const int f = blockIdx.x;
    WbfSlot s[3];
    int ncl[3], first[3], pos0[3];
    for (int l = 0; l < 3; ++l) { s[l] = wbf_slot(ws, f * 3 + l, p.cand); ncl[l] = s[l].counts[1]; first[l] = s[l].counts[2]; }
    // concatenation order of the classes = order of their first appearance among the candidates
    for (int l = 0; l < 3; ++l) {
        pos0[l] = 0;
        for (int m = 0; m < 3; ++m)
            if (m != l && (first[m] < first[l] || (first[m] == first[l] && m < l))) pos0[l] += ncl[m];
    }
    const int total = ncl[0] + ncl[1] + ncl[2];
    for (int i = threadIdx.x; i < total; i += blockDim.x) {
        int l = 0, c = i;
        while (c >= ncl[l]) { c -= ncl[l]; ++l; }
        const double sc = s[l].c_score[c];
        const int pos = pos0[l] + c;
        int r = 0;
        for (int m = 0; m < 3; ++m)
            for (int k = 0; k < ncl[m]; ++k) {
                const double s2 = s[m].c_score[k];
                r += (s2 > sc || (s2 == sc && pos0[m] + k > pos)) ? 1 : 0;
            }
        double *ob = out_boxes + ((size_t)f * p.cand + r) * 7;
        const float *fu = s[l].c_fused + (size_t)c * 7;
        for (int k = 0; k < 7; ++k) ob[k] = (double)fu[k];
        out_scores[(size_t)f * p.cand + r] = sc;
        out_labels[(size_t)f * p.cand + r] = l + 1;
        if (out_obj) out_obj[(size_t)f * p.cand + r] = s[l].c_obj[c];
    }
    if (threadIdx.x == 0) out_count[f] = total;
}

static int tta_ops_from(const int *h_kind, const float *h_param, int n_ops, bool restore, TtaOps &ops) {
    if (!h_kind || !h_param || n_ops < 1 || n_ops > TTA_MAX_OPS) {
        set_error("tta: 1..%d operations expected", TTA_MAX_OPS);
        return DZ_ERR_INVALID;
    }
    ops.n = n_ops;
    for (int i = 0; i < TTA_MAX_OPS; ++i) { ops.kind[i] = 0; ops.p0[i] = ops.p1[i] = ops.p2[i] = 0.f; }
    for (int i = 0; i < n_ops; ++i) {
        if (h_kind[i] < 0 || h_kind[i] > TTA_SCALE) {
            set_error("tta: unknown operation code %d", h_kind[i]);
            return DZ_ERR_INVALID;
        }
        ops.kind[i] = h_kind[i];
        if (h_kind[i] == TTA_ROT) {
            // the angle enters the reference as a float32 tensor element; the restore rotates by its negation
            const float ang = restore ? (float)(-(double)h_param[i]) : h_param[i];
            ops.p0[i] = cosf(ang); ops.p1[i] = sinf(ang); ops.p2[i] = ang;
        } else {
            ops.p0[i] = h_param[i];
        }
    }
    return DZ_OK;
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_tta_augment_points(const float *points, int n, int c, const int *h_kind, const float *h_param, int n_ops, float *out,
                          void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 0 && c >= 3, "dz_tta_augment_points: bad sizes");
    TtaOps ops;
    const int rc = tta_ops_from(h_kind, h_param, n_ops, false, ops);
    if (rc) return rc;
    if (n == 0) return DZ_OK;
    DZ_CHECK_ARG(points && out, "dz_tta_augment_points: null pointer");
    hipLaunchKernelGGL(k_tta_points, dim3(stream_grid((long)n_ops * n, 256)), dim3(256), 0, stream, points, n, c, ops, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_tta_restore_boxes(float *boxes, int frames, int n_ops, int m, int dim, const int *h_kind, const float *h_param, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(frames >= 0 && m >= 0 && dim >= 7, "dz_tta_restore_boxes: bad sizes");
    TtaOps ops;
    const int rc = tta_ops_from(h_kind, h_param, n_ops, true, ops);
    if (rc) return rc;
    if (frames == 0 || m == 0) return DZ_OK;
    DZ_CHECK_ARG(boxes, "dz_tta_restore_boxes: null pointer");
    hipLaunchKernelGGL(k_tta_restore, dim3(stream_grid((long)frames * n_ops * m, 256)), dim3(256), 0, stream, boxes, frames, m, dim, ops);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

size_t dz_wbf_workspace_bytes(int frames, int cand) {
    if (frames < 1 || cand < 1) return 256;
    return (size_t)frames * 3 * wbf_slot_bytes(cand) + 256;
}

int dz_wbf_fuse_3d(const float *boxes, const float *scores, const int *labels, const int *obj_ids, int frames, int cand, int per_model,
                   const double *weights, int n_models, const double *h_iou_thr3, const double *h_skip_thr3, double weight_sum,
                   int conf_max, int allows_overflow, double *out_boxes, double *out_scores, int *out_labels, int *out_obj_ids,
                   int *out_count, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(frames >= 0 && cand >= 0 && per_model >= 1 && n_models >= 1 && (long)per_model * n_models >= cand,
                 "dz_wbf_fuse_3d: bad sizes (cand %d, %d models x %d)", cand, n_models, per_model);
    DZ_CHECK_ARG(h_iou_thr3 && h_skip_thr3 && weight_sum > 0, "dz_wbf_fuse_3d: thresholds / weight sum");
    if (frames == 0) return DZ_OK;
    DZ_CHECK_ARG(out_count, "dz_wbf_fuse_3d: null out_count");
    if (cand == 0) {
        DZ_HIP(hipMemsetAsync(out_count, 0, (size_t)frames * sizeof(int), stream));
        return DZ_OK;
    }
    DZ_CHECK_ARG(boxes && scores && labels && out_boxes && out_scores && out_labels && ws, "dz_wbf_fuse_3d: null pointer");
    DZ_CHECK_ARG(!obj_ids == !out_obj_ids, "dz_wbf_fuse_3d: obj_ids and out_obj_ids go together");
    DZ_CHECK_ARG(ws_bytes >= dz_wbf_workspace_bytes(frames, cand) && ((uintptr_t)ws & 7u) == 0, "dz_wbf_fuse_3d: workspace too small or misaligned");
    WbfParams p;
    p.frames = frames; p.cand = cand; p.per_model = per_model; p.n_models = n_models; p.conf_max = conf_max; p.allows_overflow = allows_overflow;
    for (int i = 0; i < 3; ++i) { p.iou_thr[i] = h_iou_thr3[i]; p.skip_thr[i] = h_skip_thr3[i]; }
    p.wsum = weight_sum;
    hipLaunchKernelGGL(k_wbf_rank, dim3(3, frames), dim3(256), 0, stream, scores, labels, weights, p, ws);
    hipLaunchKernelGGL(k_wbf_cluster, dim3(3, frames), dim3(WBF_THREADS), 0, stream, boxes, scores, weights, obj_ids, p, ws);
    hipLaunchKernelGGL(k_wbf_emit, dim3(frames), dim3(256), 0, stream, boxes, p, ws, out_boxes, out_scores, out_labels, out_obj_ids, out_count);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
