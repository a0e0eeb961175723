// Cropped object points -> inputs of the refining models (second half): the local-frame
// transform, fixed-size selection and per-point feature encoding that the reference does per object in numpy inside
// its DataLoader workers, here one pass over a whole batch of objects on the device.
//
//   GRM  refining/detzero_refine/datasets/waymo/waymo_geometry_dataset.py:73-131  (box frame per proposal, p2s)
//   PRM  refining/detzero_refine/datasets/waymo/waymo_position_dataset.py:66-155  (frame of the middle box, p2co)
//   helpers refining/detzero_refine/utils/data_utils.py:6-10,33-42,62-113; utils/detzero_utils/box_utils.py:28-53
//
// WHICH points are kept (sample_points draws with Python's random.sample) is decided on the host and arrives as index
// lists, so a run seeded like the reference keeps exactly the reference's points (object_features.py).
// Arithmetic follows the reference: points / boxes float64, yaw matrices rounded to float32 (rotate_yaw builds a float32
// array), box corners in float32 end to end, one rounding to float32 when the feature row is stored.
// Memory-bound (a PRM batch of 128 objects writes 1 GB of feature rows): the PRM rows are stored 16 bytes per thread,
// a wavefront covering 1 KB of consecutive channels.
#include "common.h"

namespace dz {

constexpr double kPi = 3.14159265358979323846;

__device__ __forceinline__ double wrap_heading(double a) {       // data_utils.py:33-42, step by step as there
    while (a >= kPi) a -= 2 * kPi;
    while (a < -kPi) a += 2 * kPi;
    return a;
}

// (p - c) @ rotate_yaw(yaw).T with the matrix entries rounded to float32
__device__ __forceinline__ void to_box_frame(const double *__restrict__ p, const double *__restrict__ c, double yaw, double (&o)[3]) {
    const double cs = (double)(float)cos(yaw), sn = (double)(float)sin(yaw);
    const double vx = p[0] - c[0], vy = p[1] - c[1], vz = p[2] - c[2];
    o[0] = vx * cs + vy * sn;
    o[1] = -vx * sn + vy * cs;
    o[2] = vz;
}

// last box f in [lo, hi) with box_offsets[f] <= p (boxes without points share their offset with the next box)
__device__ __forceinline__ int box_of_point(const int *__restrict__ box_offsets, int lo, int hi, int p) {
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (box_offsets[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

enum { GRM_XYZ = 1, GRM_INTENSITY = 2, GRM_P2S = 4, GRM_SCORE = 8 };

__global__ __launch_bounds__(256) void k_grm_memory(const double *__restrict__ pts, const int *__restrict__ box_offsets,
                                                    const double *__restrict__ traj, const double *__restrict__ score,
                                                    const int *__restrict__ obj_box_offsets, const int *__restrict__ mem_idx,
                                                    int mem_n, int batch, int enc, int cm, float *__restrict__ out) {
    const long total = (long)batch * mem_n;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < total; r += (long)gridDim.x * blockDim.x) {
        float *o = out + r * cm;
        const int idx = mem_idx[r];
        if (idx < 0) {
            for (int c = 0; c < cm; ++c) o[c] = 0.f;
            continue;
        }
        const int b = (int)(r / mem_n);
        const int o0 = obj_box_offsets[b], o1 = obj_box_offsets[b + 1];
        const int p = box_offsets[o0] + idx;
        const int f = box_of_point(box_offsets, o0, o1, p);
        const double *bx = traj + (size_t)f * 7;
        double v[3];
        to_box_frame(pts + (size_t)p * 4, bx, bx[6], v);
        int c = 0;
        if (enc & GRM_XYZ) { o[c++] = (float)v[0]; o[c++] = (float)v[1]; o[c++] = (float)v[2]; }
        if (enc & GRM_INTENSITY) o[c++] = (float)pts[(size_t)p * 4 + 3];
        if (enc & GRM_P2S) {
            for (int k = 0; k < 3; ++k) o[c++] = (float)(bx[3 + k] / 2 - v[k]);
            for (int k = 0; k < 3; ++k) o[c++] = (float)(bx[3 + k] / 2 + v[k]);
        }
        if (enc & GRM_SCORE) o[c++] = (float)score[f];
    }
}

__global__ __launch_bounds__(256) void k_grm_query(const double *__restrict__ pts, const int *__restrict__ box_offsets,
                                                   const double *__restrict__ traj, const int *__restrict__ query_box,
                                                   const int *__restrict__ query_idx, int q_n, long rows, float4 *__restrict__ out) {
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
        const int f = query_box[r / q_n];
        const int idx = query_idx[r];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f >= 0 && idx >= 0) {
            const size_t p = (size_t)box_offsets[f] + idx;
            const double *bx = traj + (size_t)f * 7;
            double v[3];
            to_box_frame(pts + p * 4, bx, bx[6], v);
            o = make_float4((float)v[0], (float)v[1], (float)v[2], (float)pts[p * 4 + 3]);
        }
        out[r] = o;
    }
}

// ---- PRM ------------------------------------------------------------------------------------------------------
// per box: trajectory in the frame of the object's middle box (init_coords_transform) and the 27 anchor coordinates
// of the p2co feature (8 corners, float32 arithmetic as boxes_to_corners_3d, then the centre)
__global__ __launch_bounds__(256) void k_prm_boxes(const double *__restrict__ traj, const int *__restrict__ obj_box_offsets, int batch,
                                                   int box_max, double *__restrict__ init_box, float *__restrict__ traj_local,
                                                   float *__restrict__ padding_mask, double *__restrict__ anchors,
                                                   double *__restrict__ frames) {
    const int total = batch * box_max;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < total; r += gridDim.x * blockDim.x) {
        const int b = r / box_max, t = r % box_max;
        const int o0 = obj_box_offsets[b], nb = obj_box_offsets[b + 1] - o0;
        double init[7];
        for (int k = 0; k < 7; ++k) init[k] = nb > 0 ? traj[(size_t)(o0 + nb / 2) * 7 + k] : 0.0;
        init[6] = wrap_heading(init[6]);
        if (t == 0) {
            for (int k = 0; k < 7; ++k) init_box[(size_t)b * 7 + k] = init[k];
            frames[2 * b] = (double)(float)cos(init[6]);
            frames[2 * b + 1] = (double)(float)sin(init[6]);
        }
        float *tl = traj_local + (size_t)r * 7;
        double *an = anchors + (size_t)r * 27;
        if (t >= nb) {
            for (int k = 0; k < 7; ++k) tl[k] = 0.f;
            for (int k = 0; k < 27; ++k) an[k] = 0.0;
            padding_mask[r] = 1.f;
            continue;
        }
        padding_mask[r] = 0.f;
        const double *bx = traj + (size_t)(o0 + t) * 7;
        double c[3];
        to_box_frame(bx, init, init[6], c);
        const double yaw = wrap_heading(wrap_heading(bx[6]) - init[6]);
        tl[0] = (float)c[0]; tl[1] = (float)c[1]; tl[2] = (float)c[2];
        tl[3] = (float)bx[3]; tl[4] = (float)bx[4]; tl[5] = (float)bx[5];
        tl[6] = (float)yaw;
        // corners: float32 template * dims, rotation about z, + centre (box_utils.py:44-51); no fused multiply-adds
        const float cf[3] = {(float)c[0], (float)c[1], (float)c[2]};
        const float df[3] = {(float)bx[3], (float)bx[4], (float)bx[5]};
        const float yf = (float)yaw;
        const float ca = cosf(yf), sa = sinf(yf);
        const float sx[8] = {1, 1, -1, -1, 1, 1, -1, -1}, sy[8] = {1, -1, -1, 1, 1, -1, -1, 1}, sz[8] = {-1, -1, -1, -1, 1, 1, 1, 1};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float x = __fmul_rn(df[0], sx[k] * 0.5f), y = __fmul_rn(df[1], sy[k] * 0.5f), z = __fmul_rn(df[2], sz[k] * 0.5f);
            const float xr = __fadd_rn(__fmul_rn(x, ca), __fmul_rn(y, -sa));
            const float yr = __fadd_rn(__fmul_rn(x, sa), __fmul_rn(y, ca));
            an[3 * k + 0] = (double)__fadd_rn(xr, cf[0]);
            an[3 * k + 1] = (double)__fadd_rn(yr, cf[1]);
            an[3 * k + 2] = (double)__fadd_rn(z, cf[2]);
        }
        an[24] = c[0]; an[25] = c[1]; an[26] = c[2];
    }
}

enum { PRM_XYZ = 0, PRM_INTENSITY = 1, PRM_P2CO = 2, PRM_SCORE = 3, PRM_CLASS = 4 };

struct PrmEncoding {
    int n;
    int code[8];
    int start[9];       // first channel of every entry; start[n] = channels per row
};

// One workgroup per box slot at a time, one thread per 4 consecutive channels of a feature row (a wave stores 1 KB
// contiguous); rows [0, q_n) of a box are its query points, rows [q_n, q_n + m_n) its memory points.  Everything that
// depends only on the box is wave-uniform; the per-item index math is 32-bit.  frames = (cos, sin) of every object's
// frame, as doubles holding the float32-rounded values (k_prm_boxes).
__global__ __launch_bounds__(256) void k_prm_points(const double *__restrict__ pts, const int *__restrict__ box_offsets,
                                                    const double *__restrict__ score, const int *__restrict__ obj_box_offsets,
                                                    const int *__restrict__ obj_cls, const int *__restrict__ q_idx,
                                                    const int *__restrict__ m_idx, int q_n, int m_n, int batch, int box_max,
                                                    PrmEncoding enc, const double *__restrict__ init_box,
                                                    const double *__restrict__ anchors, const double *__restrict__ frames,
                                                    float *__restrict__ query, float *__restrict__ memory) {
    const unsigned int ch = (unsigned int)enc.start[enc.n];
    const unsigned int quads = (ch + 3u) >> 2;
    const unsigned int q_items = (unsigned int)q_n * quads, items = (unsigned int)(q_n + m_n) * quads;
    const int slots = batch * box_max;
    // per channel: feature code and index inside the feature (the same for every row) - looked up instead of searched
    __shared__ unsigned char s_code[40], s_k[40];
    __shared__ double s_an[27];
    for (unsigned int c = threadIdx.x; c < 4u * quads; c += blockDim.x) {
        int e = 0;
        while (e + 1 < enc.n && (int)c >= enc.start[e + 1]) ++e;
        s_code[c] = c < ch ? (unsigned char)enc.code[e] : (unsigned char)255;
        s_k[c] = (unsigned char)(c - enc.start[e]);
    }
    for (int bt = blockIdx.x; bt < slots; bt += gridDim.x) {
        __syncthreads();                                            // s_code ready / previous box done with s_an
        const int b = bt / box_max, t = bt - b * box_max;
        const int o0 = obj_box_offsets[b], nb = obj_box_offsets[b + 1] - o0;
        float *const qrow = query + (size_t)bt * q_n * ch, *const mrow = memory + (size_t)bt * m_n * ch;
        if (t >= nb) {                                              // padding slot: zero rows
            if ((ch & 3u) == 0) {
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                for (unsigned int i = threadIdx.x; i < q_items; i += blockDim.x) reinterpret_cast<float4 *>(qrow)[i] = z;
                for (unsigned int i = threadIdx.x; i < items - q_items; i += blockDim.x) reinterpret_cast<float4 *>(mrow)[i] = z;
            } else {
                for (unsigned int i = threadIdx.x; i < (unsigned int)q_n * ch; i += blockDim.x) qrow[i] = 0.f;
                for (unsigned int i = threadIdx.x; i < (unsigned int)m_n * ch; i += blockDim.x) mrow[i] = 0.f;
            }
            continue;
        }
        const int f = o0 + t;
        const double *const ib = init_box + (size_t)b * 7;
        const double cx = ib[0], cy = ib[1], cz = ib[2], cs = frames[2 * b], sn = frames[2 * b + 1];
        const double sc = score[f];
        const int cls = obj_cls ? obj_cls[b] : 0;
        if (threadIdx.x < 27) s_an[threadIdx.x] = anchors[(size_t)bt * 27 + threadIdx.x];
        __syncthreads();
        const double *const bp = pts + (size_t)box_offsets[f] * 4;
        for (unsigned int i = threadIdx.x; i < items; i += blockDim.x) {
            const bool is_q = i < q_items;
            const unsigned int ii = is_q ? i : i - q_items;
            const unsigned int j = ii / quads, quad = ii - j * quads;
            const int idx = is_q ? q_idx[(size_t)f * q_n + j] : m_idx[(size_t)f * m_n + j];
            float *o = (is_q ? qrow : mrow) + (size_t)j * ch + 4u * quad;
            double v[3] = {0.0, 0.0, 0.0}, inten = 0.0;              // rows past the box's points are zero points
            if (idx >= 0) {
                const double *p = bp + (size_t)idx * 4;
                const double vx = p[0] - cx, vy = p[1] - cy;
                v[0] = vx * cs + vy * sn;
                v[1] = -vx * sn + vy * cs;
                v[2] = p[2] - cz;
                inten = p[3];
            }
            float val[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned int code = s_code[4u * quad + k], kk = s_k[4u * quad + k], k3 = kk % 3u;
                const double coord = k3 == 0 ? v[0] : (k3 == 1 ? v[1] : v[2]);
                double x = 0.0;
                if (code == PRM_XYZ) x = coord;
                else if (code == PRM_INTENSITY) x = inten;
                else if (code == PRM_P2CO) x = coord - s_an[kk];
                else if (code == PRM_SCORE) x = sc;
                else if (code == PRM_CLASS) x = cls == (int)kk + 1 ? 1.0 : 0.0;
                val[k] = (float)x;
            }
            if ((ch & 3u) == 0) {
                *reinterpret_cast<float4 *>(o) = make_float4(val[0], val[1], val[2], val[3]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4u * quad + k < ch) o[k] = val[k];
            }
        }
    }
}

// ---- device-side draw of the fixed-size selections (the fast alternative to replaying Python's random.sample on the host) ----
// Same distribution as sample_points (data_utils.py:12-30: a uniformly random k-subset in ascending order when n >= k, all rows
// otherwise), different random stream: selection sampling (Knuth 3.4.2 S) driven by a counter-based generator, so that set s
// of a call is a pure function of (seed, s) - reproducible, order-independent, restated bit for bit by oracle/object_features.py.
__host__ __device__ inline uint32_t draw_hash(uint64_t seed, uint32_t set, uint32_t i) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * ((uint64_t)set + 1) + (uint64_t)i * 0xD1B54A32D192ED03ull;     // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

__global__ __launch_bounds__(64) void k_draw_subsets(const int *__restrict__ counts, int n_sets, int k, uint64_t seed, int set0,
                                                     int *__restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_sets) return;
    const int n = counts[s];
    int *o = out + (size_t)s * k;
    int sel = 0;
    if (n < k) {
        for (; sel < n; ++sel) o[sel] = sel;
    } else {
        for (int i = 0; i < n && sel < k; ++i) {
            // keep row i with probability (k - sel) / (n - i): floor(r * (n - i) / 2^32) < k - sel
            const uint32_t r = draw_hash(seed, (uint32_t)(set0 + s), (uint32_t)i);
            if ((uint32_t)(((uint64_t)r * (uint32_t)(n - i)) >> 32) < (uint32_t)(k - sel)) o[sel++] = i;
        }
    }
    for (; sel < k; ++sel) o[sel] = -1;
}

static int grm_channels(int enc) {
    return ((enc & GRM_XYZ) ? 3 : 0) + ((enc & GRM_INTENSITY) ? 1 : 0) + ((enc & GRM_P2S) ? 6 : 0) + ((enc & GRM_SCORE) ? 1 : 0);
}

static int prm_channels(const int *codes, int n) {
    static const int width[5] = {3, 1, 27, 1, 3};
    int ch = 0;
    for (int i = 0; i < n; ++i) {
        if (codes[i] < 0 || codes[i] > 4) return -1;
        ch += width[codes[i]];
    }
    return ch;
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_draw_subsets(const int *counts, int n_sets, int k, unsigned long long seed, int first_set_id, int *out_idx, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n_sets >= 0 && k >= 1, "dz_draw_subsets: bad sizes");
    if (n_sets == 0) return DZ_OK;
    DZ_CHECK_ARG(counts && out_idx, "dz_draw_subsets: null pointer");
    hipLaunchKernelGGL(k_draw_subsets, dim3(ceil_div(n_sets, 64)), dim3(64), 0, stream, counts, n_sets, k, (uint64_t)seed, first_set_id, out_idx);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_grm_feature_channels(int encoding) { return grm_channels(encoding); }

int dz_prm_feature_channels(const int *h_encoding, int n_enc) {
    if (!h_encoding || n_enc < 1 || n_enc > 8) return -1;
    return prm_channels(h_encoding, n_enc);
}

int dz_grm_encode_points(const double *pts, const int *box_offsets, const double *traj, const double *score,
                         const int *obj_box_offsets, const int *mem_idx, int mem_n, const int *query_box, const int *query_idx,
                         int q_max, int q_n, int batch, int encoding, float *memory, float *query, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(batch >= 0 && mem_n >= 0 && q_max >= 0 && q_n >= 0, "dz_grm_encode_points: negative size");
    if (batch == 0) return DZ_OK;
    DZ_CHECK_ARG(box_offsets && traj && score && obj_box_offsets, "dz_grm_encode_points: null pointer");
    const int cm = grm_channels(encoding);
    DZ_CHECK_ARG(cm > 0 && (encoding & ~15) == 0, "dz_grm_encode_points: bad encoding flags 0x%x", encoding);
    if (mem_n > 0) {
        DZ_CHECK_ARG(mem_idx && memory, "dz_grm_encode_points: null memory pointers");
        hipLaunchKernelGGL(k_grm_memory, dim3(stream_grid((long)batch * mem_n, 256)), dim3(256), 0, stream, pts, box_offsets, traj, score,
                           obj_box_offsets, mem_idx, mem_n, batch, encoding, cm, memory);
        DZ_LAUNCH_CHECK();
    }
    const long qrows = (long)batch * q_max * q_n;
    if (qrows > 0) {
        DZ_CHECK_ARG(query_box && query_idx && query, "dz_grm_encode_points: null query pointers");
        DZ_CHECK_ARG(((uintptr_t)query & 15u) == 0, "dz_grm_encode_points: query output not 16-byte aligned");
        hipLaunchKernelGGL(k_grm_query, dim3(stream_grid(qrows, 256)), dim3(256), 0, stream, pts, box_offsets, traj, query_box, query_idx,
                           q_n, qrows, reinterpret_cast<float4 *>(query));
        DZ_LAUNCH_CHECK();
    }
    return DZ_OK;
}

int dz_prm_encode_points(const double *pts, const int *box_offsets, const double *traj, const double *score,
                         const int *obj_box_offsets, const int *obj_cls, const int *q_idx, const int *m_idx, int q_n, int m_n,
                         int batch, int box_max, const int *h_encoding, int n_enc, float *query, float *memory,
                         float *traj_local, float *padding_mask, double *init_box, double *anchors, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(batch >= 0 && box_max >= 1 && q_n >= 0 && m_n >= 0, "dz_prm_encode_points: bad sizes");
    DZ_CHECK_ARG(h_encoding && n_enc >= 1 && n_enc <= 8, "dz_prm_encode_points: 1..8 encoding entries");
    const int ch = prm_channels(h_encoding, n_enc);
    DZ_CHECK_ARG(ch > 0, "dz_prm_encode_points: unknown encoding code");
    if (batch == 0) return DZ_OK;
    DZ_CHECK_ARG(box_offsets && traj && score && obj_box_offsets && traj_local && padding_mask && init_box && anchors,
                 "dz_prm_encode_points: null pointer");
    static const int width[5] = {3, 1, 27, 1, 3};
    PrmEncoding enc;
    enc.n = n_enc;
    bool needs_cls = false;
    enc.start[0] = 0;
    for (int i = 0; i < 8; ++i) {
        enc.code[i] = i < n_enc ? h_encoding[i] : 0;
        enc.start[i + 1] = enc.start[i] + (i < n_enc ? width[h_encoding[i]] : 0);
        needs_cls |= i < n_enc && h_encoding[i] == PRM_CLASS;
    }
    DZ_CHECK_ARG(!needs_cls || obj_cls, "dz_prm_encode_points: the 'class' feature needs obj_cls");
    hipLaunchKernelGGL(k_prm_boxes, dim3(stream_grid((long)batch * box_max, 256)), dim3(256), 0, stream, traj, obj_box_offsets, batch, box_max,
                       init_box, traj_local, padding_mask, anchors, anchors + (size_t)batch * box_max * 27);
    DZ_LAUNCH_CHECK();
    if (q_n + m_n > 0) {
        DZ_CHECK_ARG((q_n == 0 || (q_idx && query)) && (m_n == 0 || (m_idx && memory)), "dz_prm_encode_points: null point pointers");
        DZ_CHECK_ARG((ch & 3) != 0 || ((((uintptr_t)query | (uintptr_t)memory) & 15u) == 0), "dz_prm_encode_points: outputs not 16-byte aligned");
        DZ_CHECK_ARG((long)(q_n + m_n) * (ch + 3) < (1l << 31), "dz_prm_encode_points: rows per box too large");
        int grid = batch * box_max;
        if (grid > 8192) grid = 8192;
        hipLaunchKernelGGL(k_prm_points, dim3(grid), dim3(256), 0, stream, pts, box_offsets, score, obj_box_offsets, obj_cls, q_idx, m_idx,
                           q_n, m_n, batch, box_max, enc, init_box, anchors, anchors + (size_t)batch * box_max * 27, query, memory);
        DZ_LAUNCH_CHECK();
    }
    return DZ_OK;
}

}  // extern "C"
