// Small memory-bound kernels of the refining module (GRM / PRM) on gfx950: point-wise max pooling of the
// PointNet encoders and the residual + LayerNorm of the transformer decoder layer.
// Reference: refining/detzero_refine/models/modules/geometry_transformer.py:124,137,
// position_transformer.py:108,117 (torch.max over points) and transformer/decoder.py:75-88
// (query + dropout(query2) -> nn.LayerNorm).  The GEMMs of the refiner run on igemm.h through
// dz_linear_forward, attention on mha.hip.
#include "common.h"

namespace dz {

// x (groups*len, c) row-major -> out (groups, c).  Block = 4 row lanes x 64 channels; every row is read as
// 256 contiguous bytes per 64-channel slab.
__global__ __launch_bounds__(256) void k_group_max(const float *__restrict__ x, int len, int c, float *__restrict__ out) {
    __shared__ float red[4][64];
    const int g = blockIdx.x;
    const int ch = blockIdx.y * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    float m = -INFINITY;
    if (ch < c) {
        const float *base = x + (size_t)g * len * c + ch;
        for (int l = rl; l < len; l += 4) m = fmaxf(m, base[(size_t)l * c]);
    }
    red[rl][threadIdx.x & 63] = m;
    __syncthreads();
    if (rl == 0 && ch < c) out[(size_t)g * c + ch] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]),
                                                           fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}

// out[i] = every int of row i of up to four (rows, w[k]) tensors is zero (PDV's empty-grid-point mask over the ball indices of all
// branches, pdv_head.py:521-523, without concatenating them first); one thread per (row, tensor) 16-byte piece is overkill: rows are
// 16-64 ints - one thread per row, vector loads
struct ZeroRowsArgs { const int *p[4]; int w[4]; int n; };
__global__ void k_rows_all_zero(ZeroRowsArgs a, int rows, unsigned char *__restrict__ out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += gridDim.x * blockDim.x) {
        unsigned int acc = 0u;
        for (int k = 0; k < a.n; ++k) {
            const int4 *q = reinterpret_cast<const int4 *>(a.p[k] + (size_t)i * a.w[k]);
            for (int j = 0; j < a.w[k] / 4; ++j) { const int4 v = q[j]; acc |= (unsigned int)(v.x | v.y | v.z | v.w); }
        }
        out[i] = acc == 0u;
    }
}

// one wavefront per row; c <= 1024, c % 64 == 0
template <int PER_LANE>
__global__ __launch_bounds__(256) void k_add_layernorm(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ gamma, const float *__restrict__ beta,
                                                       int rows, float eps, int do_norm, float *__restrict__ out,
                                                       const float *__restrict__ post = nullptr, const uint8_t *__restrict__ group_skip = nullptr,
                                                       int group_rows = 1) {
    constexpr int C = PER_LANE * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float v[PER_LANE];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
        const size_t i = (size_t)row * C + j * 64 + lane;
        v[j] = x[i] + (y ? y[i] : 0.f);
        s += v[j];
    }
    if (!do_norm) {
#pragma unroll
        for (int j = 0; j < PER_LANE; ++j) out[(size_t)row * C + j * 64 + lane] = v[j];
        return;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) { const float d = v[j] - mean; q += d * d; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int j = 0; j < PER_LANE; ++j) {
        const int ch = j * 64 + lane;
        float r = (v[j] - mean) * rstd * gamma[ch] + beta[ch];
        if (post) {                        // out = post + (the row's group is skipped ? post : LN(x + y))
            const float pv = post[(size_t)row * C + ch];
            r = pv + ((group_skip && group_skip[row / group_rows]) ? pv : r);
        }
        out[(size_t)row * C + ch] = r;
    }
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_group_max(const float *x, int groups, int len, int c, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(groups >= 0 && len >= 1 && c >= 1, "dz_group_max: bad sizes");
    if (groups == 0) return DZ_OK;
    DZ_CHECK_ARG(x && out, "dz_group_max: null pointer");
    hipLaunchKernelGGL(k_group_max, dim3(groups, ceil_div(c, 64)), dim3(256), 0, stream, x, len, c, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

/* out[i] = 1 when every int of row i of all n (n <= 4) row-major int32 tensors t[k] (rows x w[k], w[k] % 4 == 0) is zero. */
int dz_rows_all_zero(const int *const *t, const int *w, int n, int rows, unsigned char *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(n >= 1 && n <= 4 && rows >= 0 && t && w, "dz_rows_all_zero: 1..4 tensors");
    if (rows == 0) return DZ_OK;
    ZeroRowsArgs a{};
    a.n = n;
    for (int k = 0; k < n; ++k) {
        DZ_CHECK_ARG(t[k] && w[k] >= 4 && w[k] % 4 == 0, "dz_rows_all_zero: row widths are multiples of 4 ints");
        a.p[k] = t[k]; a.w[k] = w[k];
    }
    DZ_CHECK_ARG(out, "dz_rows_all_zero: null pointer");
    hipLaunchKernelGGL(k_rows_all_zero, dim3(stream_grid(rows, 256)), dim3(256), 0, stream, a, rows, out);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

int dz_add_layernorm(const float *x, const float *y, const float *gamma, const float *beta, int rows, int c, float eps,
                     int do_norm, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0, "dz_add_layernorm: negative rows");
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(x && out && (!do_norm || (gamma && beta)), "dz_add_layernorm: null pointer");
    const dim3 grid(ceil_div(rows, 4));
    switch (c) {
        case 64: hipLaunchKernelGGL(k_add_layernorm<1>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, eps, do_norm, out); break;
        case 192: hipLaunchKernelGGL(k_add_layernorm<3>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, eps, do_norm, out); break;
        case 128: hipLaunchKernelGGL(k_add_layernorm<2>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, eps, do_norm, out); break;
        case 256: hipLaunchKernelGGL(k_add_layernorm<4>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, eps, do_norm, out); break;
        case 512: hipLaunchKernelGGL(k_add_layernorm<8>, grid, dim3(256), 0, stream, x, y, gamma, beta, rows, eps, do_norm, out); break;
        default: set_error("dz_add_layernorm: c=%d not in {64,128,192,256,512}", c); return DZ_ERR_UNSUPPORTED;
    }
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

/* out = post + (group_skip[row / group_rows] ? post : LayerNorm(x + y)): the encoder layer's second normalisation with PDV's COMBINE
 * (pooled + attended features) and its "RoIs without points stay untouched" rule in the same pass.  c = 192. */
int dz_add_layernorm_combine(const float *x, const float *y, const float *gamma, const float *beta, int rows, int c, float eps, const float *post,
                             const unsigned char *group_skip, int group_rows, float *out, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(rows >= 0 && c == 192 && group_rows >= 1, "dz_add_layernorm_combine: c = 192, group_rows >= 1 (got %d, %d)", c, group_rows);
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(x && out && gamma && beta && post, "dz_add_layernorm_combine: null pointer");
    hipLaunchKernelGGL(k_add_layernorm<3>, dim3(ceil_div(rows, 4)), dim3(256), 0, stream, x, y, gamma, beta, rows, eps, 1, out, post, group_skip, group_rows);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
