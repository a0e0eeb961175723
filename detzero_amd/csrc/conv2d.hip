// Dense BEV convolutions (Conv2d 3x3 / 1x1, ConvTranspose2d with kernel == stride) with fused
// BatchNorm/bias + ReLU, channel-last implicit GEMM on the fp32 matrix cores (gfx950).
//
// Reference: detection/detzero_det/models/centerpoint_modules/backbone2d.py:33-120 (BaseBEVBackbone),
// center_head.py:14-48 (SeparateHead), :81-102 (shared_conv).  The reference runs these through
// cuDNN/MIOpen in NCHW with separate BN and ReLU passes; here the activations stay channel-last in
// zero-bordered images (the border is the conv padding, so the kernel has no bounds tests), BN is
// folded into a per-channel scale/shift epilogue, and every layer writes straight into the buffer
// (and channel offset) its consumer reads - the concat of backbone2d.py:107-108 is free.
#include <string.h>

#include "igemm.h"

namespace dz {

template <class T>
__global__ __launch_bounds__(256) void k_conv2d(dz_conv2d_desc p, long m_total, unsigned int in_bytes) {
    __shared__ __attribute__((aligned(16))) float smem[T::LDS_FLOATS];
    __shared__ int in_pix[T::BM];    // input pixel index of the (0,0) tap, -1 past the end
    __shared__ int out_pix[T::BM];   // output pixel index

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / T::WN, wn = wid % T::WN;
    const int ntn = p.cout_pad / T::BN;           // N tiles per group
    const int grp = blockIdx.y / ntn;
    const int n0 = (blockIdx.y % ntn) * T::BN;
    const long row0 = (long)blockIdx.x * T::BM;

    for (int r = tid; r < T::BM; r += 256) {
        const long mrow = row0 + r;
        int ip = -1, op = -1;
        if (mrow < m_total) {
            const int x = (int)(mrow % p.wo);
            const long t = mrow / p.wo;
            const int y = (int)(t % p.ho);
            const int b = (int)(t / p.ho);
            ip = (b * p.in_hp + y * p.stride + p.in_off) * p.in_wp + x * p.stride + p.in_off;
            op = (b * p.out_hp + y * p.out_sy + p.out_dy) * p.out_wp + x * p.out_sx + p.out_dx;
        }
        in_pix[r] = ip;
        out_pix[r] = op;
    }
    __syncthreads();

    f32x4 acc[T::MT][T::NT];
#pragma unroll
    for (int i = 0; i < T::MT; ++i)
#pragma unroll
        for (int j = 0; j < T::NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int taps = p.kh * p.kw;
    const int kchunks = p.cin / T::KC;
    const int nchunks = taps * kchunks;
    const float *wg = p.w + (size_t)grp * taps * p.cin * p.cout_pad;
    const long cbase = p.in_coff + (long)grp * p.cin;

    // per-thread A byte offsets (tap (0,0), channel chunk 0), fixed for the whole tile; rows past the end of
    // the image get an out-of-range offset and read as zeros
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(p.in, in_bytes);
    unsigned int voff[T::A_PER_THREAD];
#pragma unroll
    for (int i = 0; i < T::A_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        voff[i] = OOB_OFFSET;
        if (T::A_F4 % T::THREADS == 0 || idx < T::A_F4) {
            const int rr = idx / (T::KC / 4), q = idx % (T::KC / 4);
            const int ip = in_pix[rr];
            if (ip >= 0) voff[i] = (unsigned int)(((long)ip * p.in_cstride + cbase + q * 4) * 4);
        }
    }
    Stage<T> st;
    int ky = 0, kx = 0, kc = 0;          // chunk iterator without divisions
    const float *wp = wg;                 // weight slice of the current chunk
    unsigned int aoff = 0;                // byte offset of the current (tap, chunk) relative to voff
    auto issue = [&]() {
        load_a_buf<T>(st, rsrc, voff, aoff);
        load_b<T>(st, wp, p.cout_pad, n0, tid);
    };
    auto advance = [&]() {
        wp += (size_t)T::KC * p.cout_pad;
        if (++kc == kchunks) {
            kc = 0;
            if (++kx == p.kw) { kx = 0; ++ky; }
        }
        aoff = (unsigned int)(((ky * p.in_wp + kx) * p.in_cstride + kc * T::KC) * 4);
    };
    gemm_pipeline<T>(nchunks, smem, st, issue, advance, acc, wm, wn, lane, tid);

    const int r = lane & 15, g = lane >> 4;
    const int gcout = p.g_cout[grp];
    const int ooff = p.out_coff + p.g_ooff[grp];
#pragma unroll
    for (int nt = 0; nt < T::NT; ++nt) {
        const int col = n0 + wn * T::NT * 16 + nt * 16 + r;
        if (col >= gcout) continue;
        const float sc = p.scale ? p.scale[grp * p.cout_pad + col] : 1.f;
        const float sh = p.shift ? p.shift[grp * p.cout_pad + col] : 0.f;
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int lr = wm * T::MT * 16 + mt * 16 + g * 4 + e;
                const int op = out_pix[lr];
                if (op >= 0) {
                    float a = acc[mt][nt][e];
                    if (p.group_shift) a += p.group_shift[((row0 + lr) / p.group_rows) * p.cout_pad + col];
                    float v = fmaf(a, sc, sh);
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.out[(size_t)op * p.out_cstride + ooff + col] = v;
                }
            }
        }
    }
}

template <class T>
static int launch_conv(const dz_conv2d_desc &p, hipStream_t stream) {
    const long m_total = (long)p.batch * p.ho * p.wo;
    dim3 grid(ceil_div(m_total, T::BM), (p.cout_pad / T::BN) * p.groups);
    const size_t in_bytes = (size_t)p.batch * p.in_hp * p.in_wp * p.in_cstride * sizeof(float);
    if (in_bytes >= 0x80000000ull) {
        set_error("dz_conv2d_forward: input image of %zu bytes exceeds the 2 GiB buffer-addressing limit", in_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(k_conv2d<T>, grid, dim3(256), 0, stream, p, m_total, (unsigned int)in_bytes);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// variant ids: tile BM x BN x KC
enum ConvVariant { CV_NONE = 0, CV_64_64_16, CV_128_16_16, CV_128_64_32, CV_96_64_32, CV_64_64_32, CV_48_64_32,
                   CV_128_32_32, CV_128_16_32 };
static const char *kConvVariantName[] = {"none", "k_conv2d<64x64x16>", "k_conv2d<128x16x16>", "k_conv2d<128x64x32>",
                                         "k_conv2d<96x64x32>", "k_conv2d<64x64x32>", "k_conv2d<48x64x32>",
                                         "k_conv2d<128x32x32>", "k_conv2d<128x16x32>"};

// The BEV layers are a few hundred tiles for 256 CUs, so the M tile is chosen to fill the chip: with T
// equal tiles the busiest CU runs ceil(T/256) of them while the average is T/256; e.g. 94x94x256 as
// 64x64 tiles = 556 tiles -> 72 % fill, as 48x64 tiles = 740 -> 96 %.  Larger tiles amortise the
// per-chunk barrier / staging better, which `tile_eff` prices in (measured, approximate).
static ConvVariant conv2d_select(const dz_conv2d_desc &p) {
    const long m_total = (long)p.batch * p.ho * p.wo;
    if (p.cin % 32 != 0 && p.cin % 16 == 0) {
        if (p.cout_pad % 64 == 0) return CV_64_64_16;
        if (p.cout_pad % 16 == 0) return CV_128_16_16;
    }
    if (p.cin % 32 == 0) {
        if (p.cout_pad % 64 == 0) {
            static const int bms[4] = {128, 96, 64, 48};
            static const ConvVariant cvs[4] = {CV_128_64_32, CV_96_64_32, CV_64_64_32, CV_48_64_32};
            static const double tile_eff[4] = {1.00, 0.97, 0.92, 0.86};
            const int cus = 256;
            double best = -1.0;
            ConvVariant pick = CV_64_64_32;
            for (int i = 0; i < 4; ++i) {
                const long tiles = (long)ceil_div(m_total, bms[i]) * (p.cout_pad / 64) * p.groups;
                const long rounds = (tiles + cus - 1) / cus;
                const double useful = (double)m_total / ((double)ceil_div(m_total, bms[i]) * bms[i]);   // ragged last tile
                const double score = tile_eff[i] * useful * (double)tiles / (double)(rounds * cus);
                if (score > best) { best = score; pick = cvs[i]; }
            }
            return pick;
        }
        if (p.cout_pad % 32 == 0) return CV_128_32_32;
        if (p.cout_pad % 16 == 0) return CV_128_16_32;
    }
    return CV_NONE;
}

static int conv2d_dispatch(const dz_conv2d_desc &p, hipStream_t stream) {
    const long m_total = (long)p.batch * p.ho * p.wo;
    if (m_total == 0) return DZ_OK;
    switch (conv2d_select(p)) {
        case CV_64_64_16: return launch_conv<TileCfg<64, 64, 16, 2, 2>>(p, stream);
        case CV_128_16_16: return launch_conv<TileCfg<128, 16, 16, 4, 1>>(p, stream);
        case CV_128_64_32: return launch_conv<TileCfg<128, 64, 32, 2, 2>>(p, stream);
        case CV_96_64_32: return launch_conv<TileCfg<96, 64, 32, 2, 2>>(p, stream);
        case CV_64_64_32: return launch_conv<TileCfg<64, 64, 32, 2, 2>>(p, stream);
        case CV_48_64_32: return launch_conv<TileCfg<48, 64, 32, 1, 4>>(p, stream);
        case CV_128_32_32: return launch_conv<TileCfg<128, 32, 32, 4, 1>>(p, stream);
        case CV_128_16_32: return launch_conv<TileCfg<128, 16, 32, 4, 1>>(p, stream);
        default: break;
    }
    set_error("dz_conv2d_forward: unsupported channels cin=%d cout_pad=%d", p.cin, p.cout_pad);
    return DZ_ERR_UNSUPPORTED;
}

}  // namespace dz

using namespace dz;

namespace dz {
// y[r][n] = act(scale[n] * sum over the splits of part[r][s * cout_pad + n] + shift[n]): the second half of dz_linear_forward_splitk
__global__ void k_splitk_reduce(const float *__restrict__ part, int rows, int cout, int cout_pad, int splits, const float *__restrict__ scale,
                                const float *__restrict__ shift, int relu, float *__restrict__ y, int y_stride) {
    const long total = (long)rows * cout;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int r = (int)(idx / cout), n = (int)(idx % cout);
        const float *p = part + (size_t)r * splits * cout_pad + n;
        float a = 0.f;
        for (int s = 0; s < splits; ++s) a += p[(size_t)s * cout_pad];
        float v = fmaf(a, scale ? scale[n] : 1.f, shift ? shift[n] : 0.f);
        if (relu) v = fmaxf(v, 0.f);
        y[(size_t)r * y_stride + n] = v;
    }
}
}  // namespace dz

extern "C" {

int dz_conv2d_forward(const dz_conv2d_desc *d, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(d && d->in && d->out && d->w, "dz_conv2d_forward: null pointer");
    DZ_CHECK_ARG(d->groups >= 1 && d->groups <= 8, "dz_conv2d_forward: groups %d not in [1,8]", d->groups);
    DZ_CHECK_ARG(!d->phase_groups, "dz_conv2d_forward: phase_groups is a dz_conv2d_forward_split feature (launch the phases one by one here)");
    DZ_CHECK_ARG(!d->in_rowidx, "dz_conv2d_forward: in_rowidx (sparse input) is a dz_conv2d_forward_split feature");
    DZ_CHECK_ARG(!d->group_shift || (d->group_rows >= 1 && d->groups == 1), "dz_conv2d_forward: group_shift needs group_rows >= 1, groups == 1");
    DZ_CHECK_ARG(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->cin >= 16, "dz_conv2d_forward: bad kernel/cin");
    DZ_CHECK_ARG(d->in_cstride % 4 == 0 && d->in_coff % 4 == 0 && d->cout_pad % 16 == 0,
                 "dz_conv2d_forward: channel strides must keep 16-byte alignment");
    for (int g = 0; g < d->groups; ++g)
        DZ_CHECK_ARG(d->g_cout[g] >= 1 && d->g_cout[g] <= d->cout_pad, "dz_conv2d_forward: bad g_cout[%d]", g);
    // the last tap of the last enumerated pixel must stay inside the input image
    DZ_CHECK_ARG((d->ho - 1) * d->stride + d->in_off + d->kh - 1 < d->in_hp &&
                 (d->wo - 1) * d->stride + d->in_off + d->kw - 1 < d->in_wp && d->in_off >= 0,
                 "dz_conv2d_forward: taps leave the input image");
    DZ_CHECK_ARG((d->ho - 1) * d->out_sy + d->out_dy < d->out_hp && (d->wo - 1) * d->out_sx + d->out_dx < d->out_wp,
                 "dz_conv2d_forward: output leaves the output image");
    return conv2d_dispatch(*d, stream);
}

const char *dz_conv2d_variant(const dz_conv2d_desc *d) { return d ? kConvVariantName[conv2d_select(*d)] : "none"; }

int dz_linear_forward(const float *x, int rows, int cin, int x_stride, const float *w, int cout, int cout_pad,
                      const float *scale, const float *shift, const float *group_shift, int group_rows, int relu,
                      float *y, int y_stride, void *stream_) {
    DZ_CHECK_ARG(rows >= 0 && x_stride >= cin && y_stride >= cout, "dz_linear_forward: bad sizes");
    if (group_rows < 1) group_rows = 1;
    // the A operand is fetched through 32-bit buffer offsets: feed the engine at most ~2 GiB of rows per launch
    long max_rows = (long)(0x7FF00000ull / ((size_t)x_stride * sizeof(float)));
    if (group_shift) max_rows = max_rows / group_rows * group_rows;     // keep launches aligned to row groups
    DZ_CHECK_ARG(max_rows >= 1, "dz_linear_forward: one row group exceeds the 2 GiB addressing window");
    for (long r0 = 0; r0 < rows || r0 == 0; r0 += max_rows) {
        const int n = (int)((rows - r0) < max_rows ? (rows - r0) : max_rows);
        dz_conv2d_desc d = {};
        d.in = x + (size_t)r0 * x_stride; d.out = y + (size_t)r0 * y_stride; d.w = w; d.scale = scale; d.shift = shift;
        d.batch = 1; d.ho = 1; d.wo = n;
        d.in_hp = 1; d.in_wp = n; d.in_cstride = x_stride; d.in_coff = 0; d.cin = cin;
        d.kh = 1; d.kw = 1; d.stride = 1; d.in_off = 0;
        d.out_hp = 1; d.out_wp = n; d.out_cstride = y_stride; d.out_coff = 0;
        d.out_sy = 1; d.out_sx = 1; d.out_dy = 0; d.out_dx = 0;
        d.groups = 1; d.cout_pad = cout_pad; d.g_cout[0] = cout; d.g_ooff[0] = 0; d.relu = relu;
        d.group_shift = group_shift ? group_shift + (size_t)(r0 / group_rows) * cout_pad : nullptr;
        d.group_rows = group_rows;
        if (n == 0) return DZ_OK;
        const int rc = dz_conv2d_forward(&d, stream_);
        if (rc) return rc;
    }
    return DZ_OK;
}

// A few rows against a very long input (the PDV head's first FC layer: 487 RoIs x 41 472 inputs -> 256): as one GEMM that is 32 workgroups
// walking 2 592 k-steps each.  Here the input channels are cut into `splits` (<= 8) groups that run as the groups of ONE grouped 1 x 1
// convolution (each group's weight rows are contiguous in w) into a (rows, splits * cout_pad) workspace; a second pass sums them
// and applies scale / shift / ReLU.  fp32 sums in a different order than dz_linear_forward.
size_t dz_linear_splitk_workspace_bytes(int rows, int cout_pad, int splits) { return (size_t)rows * cout_pad * splits * sizeof(float); }

int dz_linear_forward_splitk(const float *x, int rows, int cin, int x_stride, const float *w, int cout, int cout_pad, const float *scale,
                             const float *shift, int relu, float *y, int y_stride, int splits, float *workspace, size_t workspace_bytes,
                             void *stream_) {
    DZ_CHECK_ARG(rows >= 0 && x_stride >= cin && y_stride >= cout && splits >= 1 && splits <= 8 && cin % (splits * 32) == 0,
                 "dz_linear_forward_splitk: 1..8 splits of a multiple of 32 channels each (cin %d, splits %d)", cin, splits);
    if (rows == 0) return DZ_OK;
    DZ_CHECK_ARG(workspace && workspace_bytes >= dz_linear_splitk_workspace_bytes(rows, cout_pad, splits), "dz_linear_forward_splitk: workspace too small");
    DZ_CHECK_ARG((size_t)rows * x_stride * sizeof(float) < 0x7FF00000ull, "dz_linear_forward_splitk: input beyond the 2 GiB addressing window");
    dz_conv2d_desc d = {};
    d.in = x; d.out = workspace; d.w = w; d.scale = nullptr; d.shift = nullptr;
    d.batch = 1; d.ho = 1; d.wo = rows;
    d.in_hp = 1; d.in_wp = rows; d.in_cstride = x_stride; d.in_coff = 0; d.cin = cin / splits;
    d.kh = 1; d.kw = 1; d.stride = 1; d.in_off = 0;
    d.out_hp = 1; d.out_wp = rows; d.out_cstride = splits * cout_pad; d.out_coff = 0;
    d.out_sy = 1; d.out_sx = 1; d.out_dy = 0; d.out_dx = 0;
    d.groups = splits; d.cout_pad = cout_pad; d.relu = 0; d.group_rows = 1;
    for (int g = 0; g < splits; ++g) { d.g_cout[g] = cout_pad; d.g_ooff[g] = g * cout_pad; }
    const int rc = dz_conv2d_forward(&d, stream_);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    hipLaunchKernelGGL(k_splitk_reduce, dim3(stream_grid((long)rows * cout, 256)), dim3(256), 0, stream, workspace, rows, cout, cout_pad, splits, scale, shift,
                       relu, y, y_stride);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // extern "C"
