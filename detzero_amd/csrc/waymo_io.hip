// Frame assembly of the detector's input on the device ("data formats either side"):
// the reference builds a frame from the stored sweeps in numpy inside its DataLoader workers
// (detection/detzero_det/datasets/dataset.py:164-195, DatasetTemplate.merge_sweeps): drop the no-label-zone returns,
// tanh the intensity, move every sweep into the current frame's coordinates with inv(pose_cur) @ pose_sweep, append the
// time offset.  Here the raw (N,6) rows [x,y,z,intensity,elongation,NLZ] of all sweeps are uploaded once and one pass
// writes the kept rows, in the reference's order, as the (N',6) float32 frame the voxelizer reads.
// Arithmetic as in numpy: float32 coordinates times a float64 matrix accumulate in float64 and round once to float32.
#include "common.h"

namespace dz {

constexpr int MERGE_MAX_SWEEPS = 16;

struct SweepParams {
    int n;
    int offset[MERGE_MAX_SWEEPS + 1];
    double mat[MERGE_MAX_SWEEPS][12];       // rows 0..2 of inv(pose_cur) @ pose_sweep
    double dt[MERGE_MAX_SWEEPS];            // seconds
};

__global__ __launch_bounds__(256) void k_nlz_bits(const float *__restrict__ raw, int n, uint32_t *__restrict__ bitmap) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool keep = i < n && raw[(size_t)i * 6 + 5] == -1.0f;
    const unsigned long long b = __ballot(keep);
    if ((threadIdx.x & 63) == 0) {
        bitmap[(i >> 5)] = (uint32_t)b;
        bitmap[(i >> 5) + 1] = (uint32_t)(b >> 32);
    }
}

__global__ __launch_bounds__(256) void k_merge_sweeps(const float *__restrict__ raw, int n, const uint32_t *__restrict__ bitmap,
                                                      const uint32_t *__restrict__ prefix, SweepParams sp, float *__restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t word = bitmap[i >> 5], bit = 1u << (i & 31);
        if (!(word & bit)) continue;
        const int row = (int)(prefix[i >> 5] + __popc(word & (bit - 1u)));
        int s = 0;
        while (s + 1 < sp.n && i >= sp.offset[s + 1]) ++s;
        const float *p = raw + (size_t)i * 6;
        const double x = p[0], y = p[1], z = p[2];
        const double *m = sp.mat[s];
        float *o = out + (size_t)row * 6;
        o[0] = (float)(x * m[0] + y * m[1] + z * m[2] + m[3]);
        o[1] = (float)(x * m[4] + y * m[5] + z * m[6] + m[7]);
        o[2] = (float)(x * m[8] + y * m[9] + z * m[10] + m[11]);
        o[3] = tanhf(p[3]);
        o[4] = p[4];
        o[5] = (float)sp.dt[s];
    }
}

static size_t merge_layout(int n, size_t *o_pf, size_t *o_sw, size_t *sw_bytes) {
    const size_t nwords = align_up(((size_t)(n < 1 ? 1 : n) + 63) / 64 * 2, 8);
    size_t off = align_up(nwords * 4, 256);
    *o_pf = off; off += align_up(nwords * 4, 256);
    *sw_bytes = bitmap_scan_workspace_bytes(nwords);
    *o_sw = off; off += align_up(*sw_bytes, 256);
    return off;
}

}  // namespace dz

using namespace dz;

extern "C" {

size_t dz_merge_sweeps_workspace_bytes(int n_total) {
    size_t a, b, c;
    return merge_layout(n_total, &a, &b, &c);
}

int dz_merge_sweeps(const float *raw, int n_total, const int *h_sweep_offsets, const double *h_transforms, const double *h_time_offsets,
                    int n_sweeps, float *out, int *d_count, void *ws, size_t ws_bytes, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n_total >= 0 && n_sweeps >= 1 && n_sweeps <= MERGE_MAX_SWEEPS, "dz_merge_sweeps: 1..%d sweeps", MERGE_MAX_SWEEPS);
    DZ_CHECK_ARG(h_sweep_offsets && h_transforms && h_time_offsets && d_count && ws, "dz_merge_sweeps: null pointer");
    DZ_CHECK_ARG(h_sweep_offsets[0] == 0 && h_sweep_offsets[n_sweeps] == n_total, "dz_merge_sweeps: sweep offsets do not cover the rows");
    size_t o_pf, o_sw, sw_bytes;
    const size_t need = merge_layout(n_total, &o_pf, &o_sw, &sw_bytes);
    if (ws_bytes < need) { set_error("dz_merge_sweeps: workspace %zu < %zu", ws_bytes, need); return DZ_ERR_WORKSPACE; }
    if (n_total == 0) return fill_u32(d_count, 0u, 1, stream);
    DZ_CHECK_ARG(raw && out, "dz_merge_sweeps: null pointer");
    SweepParams sp;
    sp.n = n_sweeps;
    for (int s = 0; s <= MERGE_MAX_SWEEPS; ++s) sp.offset[s] = s <= n_sweeps ? h_sweep_offsets[s] : n_total;
    for (int s = 0; s < MERGE_MAX_SWEEPS; ++s) {
        for (int k = 0; k < 12; ++k) sp.mat[s][k] = s < n_sweeps ? h_transforms[s * 12 + k] : 0.0;
        sp.dt[s] = s < n_sweeps ? h_time_offsets[s] : 0.0;
        DZ_CHECK_ARG(s >= n_sweeps || sp.offset[s] <= sp.offset[s + 1], "dz_merge_sweeps: sweep offsets must not decrease");
    }
    uint32_t *bitmap = (uint32_t *)ws, *prefix = (uint32_t *)((char *)ws + o_pf);
    const size_t nwords = align_up(((size_t)n_total + 63) / 64 * 2, 8);
    int rc = fill_u32(bitmap, 0u, nwords, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_nlz_bits, dim3(ceil_div(n_total, 256)), dim3(256), 0, stream, raw, n_total, bitmap);
    rc = bitmap_scan(bitmap, nwords, prefix, d_count, -1, ScanDims{1, 1, 1}, nullptr, 0, (char *)ws + o_sw, sw_bytes, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(k_merge_sweeps, dim3(stream_grid(n_total, 256)), dim3(256), 0, stream, raw, n_total, bitmap, prefix, sp, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
