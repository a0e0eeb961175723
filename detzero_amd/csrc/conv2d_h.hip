// Dense BEV convolutions on the 16-bit matrix cores with split-precision (pair16) operands: same layers,
// descriptors and fused epilogue as conv2d.hip (Conv2d 3x3 / 1x1, ConvTranspose2d with kernel == stride,
// BatchNorm/bias + ReLU folded), fp32-class results at ~5x the fp32-MFMA rate (hgemm.h).
//
// Reference: detection/detzero_det/models/centerpoint_modules/backbone2d.py:33-120, center_head.py:14-48, :81-102.
// Input image and weights are pair16; the output image is pair16 (feeds the next layer) or plain fp32 (the
// last head convolution, which feeds the decoder).
#include <stdlib.h>

#include "hgemm.h"

namespace dz {

// conv3x3_h.hip: 3x3 stride-1 layers with enough tiles run on the image-tile-resident kernel
int conv3x3_h_variant(const dz_conv2d_desc &p);
int conv3x3_h_launch(const dz_conv2d_desc &p, int math, int out_f32, size_t w_bytes, hipStream_t stream);

template <class T, class M, bool OUT_F32, int NS>
__global__ __launch_bounds__(256) void k_conv2d_h(dz_conv2d_desc p, long m_total, unsigned int in_bytes, unsigned int w_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v4u *const smem = reinterpret_cast<v4u *>(smem_raw);
    __shared__ int in_pix[T::BP];    // input pixel index of the (0,0) tap, -1 past the end
    __shared__ int out_pix[T::BP];   // output pixel index
    __shared__ __attribute__((aligned(16))) float sc_s[T::BC], sh_s[T::BC];     // scale / shift of my channel tile (pair16 epilogue)

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wp = wid / T::WC, wc = wid % T::WC;
    const int ntn = p.cout_pad / T::BC;           // channel tiles per group
    // XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (each XCD has its own L2), so remap the
    // linear id bijectively such that every XCD walks one CONTIGUOUS band of the image, channel tiles of a pixel
    // tile adjacent: neighbouring tiles share their halo rows and taps in that XCD's L2 instead of re-fetching them
    const int nty = ntn * p.groups;
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int by = lid % nty;
    const int grp = by / ntn;
    const int n0 = (by % ntn) * T::BC;
    const long row0 = (long)(lid / nty) * T::BP;
    // phase groups (ConvTranspose2d with kernel == stride in one launch): group g = phase (g / out_sx, g % out_sx) of the output
    // pixel block; all phases read the same input channels and share scale / shift.  The XCD-aware order above puts the phases
    // of a pixel tile next to each other, so the input tile comes from HBM once and from that XCD's L2 for the other phases
    const bool phases = p.phase_groups != 0;
    const int ph_dy = phases ? grp / p.out_sx : 0, ph_dx = phases ? grp % p.out_sx : 0;
    const int ssg = phases ? 0 : grp;                 // scale / shift / input-channel group

    for (int r = tid; r < T::BP; r += 256) {
        const long mrow = row0 + r;
        int ip = -1, op = -1;
        if (mrow < m_total) {
            const int x = (int)(mrow % p.wo);
            const long t = mrow / p.wo;
            const int y = (int)(t % p.ho);
            const int b = (int)(t / p.ho);
            ip = (b * p.in_hp + y * p.stride + p.in_off) * p.in_wp + x * p.stride + p.in_off;
            op = (b * p.out_hp + y * p.out_sy + p.out_dy + ph_dy) * p.out_wp + x * p.out_sx + p.out_dx + ph_dx;
        }
        in_pix[r] = ip;
        out_pix[r] = op;
    }
    if constexpr (!OUT_F32) {
        for (int c = tid; c < T::BC; c += 256) {
            const bool in = n0 + c < p.g_cout[grp];
            sc_s[c] = (in && p.scale) ? p.scale[ssg * p.cout_pad + n0 + c] : 1.f;
            sh_s[c] = (in && p.shift) ? p.shift[ssg * p.cout_pad + n0 + c] : 0.f;
        }
    }
    __syncthreads();

    f32x16 acc[T::CT][T::PT];
#pragma unroll
    for (int i = 0; i < T::CT; ++i)
#pragma unroll
        for (int j = 0; j < T::PT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int taps = p.kh * p.kw;
    const int kchunks = p.cin / T::KC;
    const int nchunks = taps * kchunks;
    const long cbase = p.in_coff + (long)ssg * p.cin;

    const srsrc_t prsrc = make_srsrc(p.in, in_bytes);
    const srsrc_t crsrc = make_srsrc(p.w, w_bytes);
    unsigned int pvoff[T::P_PER_THREAD], cvoff[T::C_PER_THREAD];
#pragma unroll
    for (int i = 0; i < T::P_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        pvoff[i] = OOB_OFFSET;
        if (T::P_PIECES % T::THREADS == 0 || idx < T::P_PIECES) {
            const int rr = idx / (T::KC / 4), q = idx % (T::KC / 4);
            const int ip = in_pix[rr];
            if (ip >= 0) pvoff[i] = (unsigned int)(((long)ip * p.in_cstride + cbase + q * 4) * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < T::C_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        cvoff[i] = OOB_OFFSET;
        if (T::C_PIECES % T::THREADS == 0 || idx < T::C_PIECES) {
            const int n = idx / (T::KC / 4), q = idx % (T::KC / 4);
            cvoff[i] = (unsigned int)((((long)grp * taps * p.cout_pad + n0 + n) * p.cin + q * 4) * 4);
        }
    }
    int ky = 0, kx = 0, kc = 0;
    unsigned int padd = 0, cadd = 0;
    const unsigned int tap_bytes = (unsigned int)((long)p.cout_pad * p.cin * 4);
    unsigned int tap_base = 0;
    auto issue = [&](HStage<T> &st, auto) { load_hstage<T>(st, prsrc, pvoff, padd, crsrc, cvoff, cadd); };
    // chunk order: channel chunk outermost, then ky, kx innermost - the kx taps of one image row re-read the same
    // cache lines shifted by one pixel, back to back, while they are still in the CU's L1
    auto advance = [&]() {
        tap_base += tap_bytes;
        if (++kx == p.kw) {
            kx = 0;
            if (++ky == p.kh) { ky = 0; ++kc; tap_base = 0; }
        }
        padd = (unsigned int)(((ky * p.in_wp + kx) * p.in_cstride + kc * T::KC) * 4);
        cadd = tap_base + (unsigned int)(kc * T::KC * 4);
    };
    hgemm_pipeline<T, M, NS>(nchunks, smem, issue, advance, acc, wp, wc, lane, tid);

    // accumulator of a 32x32 fragment: pixel = lane & 31, channel = 8*(reg>>2) + 4*(lane>>5) + (reg&3)
    const int h = lane >> 5;
    const int gcout = p.g_cout[grp];
    const int ooff = p.out_coff + p.g_ooff[grp];
    if constexpr (!OUT_F32) {
        if (!p.group_shift) {          // (the per-row-group addend of the PointNet concat keeps the direct path below)
            store_tile_pair16<T, M>(acc, smem_raw, sc_s, sh_s, n0, gcout, p.relu != 0, nullptr, reinterpret_cast<unsigned char *>(p.out), wp, wc,
                                    lane, wid, [&](int lr) {
                                        const int op = out_pix[lr];
                                        return op >= 0 ? ((size_t)op * p.out_cstride + ooff) * 4 : ~size_t(0);
                                    });
            return;
        }
    }
    if constexpr (OUT_F32) {
        if (p.group_max) {
            // Fused max over row groups (PointNet: torch.max over the points of an object, geometry_transformer.py:124,137): the tile's
            // rows belong to ONE group (group_rows % BP == 0, checked by the launcher), so the wave reduces its PT fragments in
            // registers, the 32 rows of a fragment with shuffles, and issues one atomic max per (wave, channel) on the fp32 result
            // (pre-filled with -inf).  The (rows x cout) activation is never written: 2 x rows x cout x 4 bytes of HBM traffic less.
            const long grow = row0 / p.group_rows;
            float *const orow = p.out + (size_t)grow * p.out_cstride + ooff;
#pragma unroll
            for (int ct = 0; ct < T::CT; ++ct) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = n0 + wc * T::CT * 32 + ct * 32 + 8 * j + 4 * h;
                    float v[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
                    for (int pt = 0; pt < T::PT; ++pt) {
                        const int lr = wp * T::PT * 32 + pt * 32 + (lane & 31);
                        if (out_pix[lr] < 0 || col >= gcout) continue;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float sc = p.scale ? p.scale[ssg * p.cout_pad + col + e] : 1.f;
                            const float sh = p.shift ? p.shift[ssg * p.cout_pad + col + e] : 0.f;
                            float a = acc[ct][pt][4 * j + e];
                            if (p.group_shift) a += p.group_shift[(size_t)grow * p.cout_pad + col + e];
                            a = fmaf(a, sc, sh);
                            if (p.relu) a = fmaxf(a, 0.f);
                            v[e] = fmaxf(v[e], a);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
#pragma unroll
                        for (int d = 1; d < 32; d <<= 1) v[e] = fmaxf(v[e], __shfl_xor(v[e], d, 64));
                    }
                    if ((lane & 31) == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (col + e >= gcout || v[e] == -INFINITY) continue;
                            // float max through integer atomics: non-negative values order like ints, negative ones inversely like uints
                            // (by the SIGN BIT, not by value: -0.0 compares >= 0 but its pattern is INT_MIN as an int and would
                            // never beat the -inf prefill; as an unsigned it is the smallest "negative" pattern - the right maximum)
                            if (__float_as_int(v[e]) >= 0) atomicMax(reinterpret_cast<int *>(orow + col + e), __float_as_int(v[e]));
                            else atomicMin(reinterpret_cast<unsigned int *>(orow + col + e), __float_as_uint(v[e]));
                        }
                    }
                }
            }
            return;
        }
    }
#pragma unroll
    for (int pt = 0; pt < T::PT; ++pt) {
        const int lr = wp * T::PT * 32 + pt * 32 + (lane & 31);
        const int op = out_pix[lr];
        if (op < 0) continue;
#pragma unroll
        for (int ct = 0; ct < T::CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wc * T::CT * 32 + ct * 32 + 8 * j + 4 * h;       // first of 4 consecutive channels
                if (col >= gcout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float sc = p.scale ? p.scale[ssg * p.cout_pad + col + e] : 1.f;
                    const float sh = p.shift ? p.shift[ssg * p.cout_pad + col + e] : 0.f;
                    float a = acc[ct][pt][4 * j + e];
                    // per-row-group addend of the PointNet concat (linear layers: output pixel index = row)
                    if (p.group_shift) a += p.group_shift[(size_t)(op / p.group_rows) * p.cout_pad + col + e];
                    v[e] = fmaf(a, sc, sh);
                    if (p.relu) v[e] = fmaxf(v[e], 0.f);
                }
                if (OUT_F32) {
                    float *o = p.out + (size_t)op * p.out_cstride + ooff + col;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < gcout) o[e] = v[e];
                } else {
                    uint2 hi, lo;
                    split4<M>(v, hi, lo);
                    // group of 8 channels starting at (col & ~7): 16 bytes hi | 16 bytes lo; this lane owns slot 4*h..4*h+3
                    unsigned char *g = reinterpret_cast<unsigned char *>(p.out) +
                                       ((size_t)op * p.out_cstride + ooff + (col & ~7)) * 4 + (col & 7) * 2;
                    *reinterpret_cast<uint2 *>(g) = hi;
                    *reinterpret_cast<uint2 *>(g + 16) = lo;
                }
            }
        }
    }
}

template <class T, class M, bool OUT_F32, int NS>
static int launch_conv_h(const dz_conv2d_desc &p, size_t w_bytes, hipStream_t stream) {
    const long m_total = (long)p.batch * p.ho * p.wo;
    dim3 grid(ceil_div(m_total, T::BP) * (p.cout_pad / T::BC) * p.groups);
    const size_t in_bytes = (size_t)p.batch * p.in_hp * p.in_wp * p.in_cstride * sizeof(float);
    if (in_bytes >= 0x80000000ull || w_bytes >= 0x80000000ull) {
        set_error("dz_conv2d_forward_split: image of %zu bytes / weights of %zu bytes exceed the 2 GiB buffer-addressing limit", in_bytes, w_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    static PerDeviceFlags lds_done;
    if (int rc_ = reserve_lds(reinterpret_cast<const void *>(&k_conv2d_h<T, M, OUT_F32, NS>), T::LDS_BYTES, lds_done, "dz_conv2d_forward_split")) return rc_;
    hipLaunchKernelGGL((k_conv2d_h<T, M, OUT_F32, NS>), grid, dim3(256), T::LDS_BYTES, stream, p, m_total, (unsigned int)in_bytes,
                       (unsigned int)w_bytes);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// variant ids: BP (pixels) x BC (channels) x KC
enum ConvHVariant { CH_NONE = 0, CH_128_128, CH_128_64, CH_64_128, CH_64_64, CH_128_32 };
static const char *kConvHVariantName[] = {"none", "k_conv2d_h<128x128x32>", "k_conv2d_h<128x64x32>", "k_conv2d_h<64x128x32>",
                                          "k_conv2d_h<64x64x32>", "k_conv2d_h<128x32x32>"};

static ConvHVariant conv2d_h_select(const dz_conv2d_desc &p) {
    if (p.cin % 32 != 0) return CH_NONE;
    const long m_total = (long)p.batch * p.ho * p.wo;
    if (p.cout_pad % 64 != 0) return p.cout_pad % 32 == 0 ? CH_128_32 : CH_NONE;
    // development knob: force a tile shape (1 = 128 x 128, 2 = 128 x 64, 3 = 64 x 128, 4 = 64 x 64) for the layers it divides
    static const int forced = getenv("DZ_TUNE_CONV2D_H") ? atoi(getenv("DZ_TUNE_CONV2D_H")) : 0;
    if (forced >= 1 && forced <= 4 && p.cout_pad % (forced == 1 || forced == 3 ? 128 : 64) == 0 && m_total >= 4096) return (ConvHVariant)forced;
    // chip fill: two workgroups are resident per CU; prefer the largest tile that keeps >= ~90 % of the slots busy
    struct Cand { ConvHVariant v; int bp, bc; double eff; };
    static const Cand cands[4] = {{CH_128_128, 128, 128, 1.00}, {CH_128_64, 128, 64, 0.90}, {CH_64_128, 64, 128, 0.88}, {CH_64_64, 64, 64, 0.78}};
    const int slots = 512;
    double best = -1.0;
    ConvHVariant pick = CH_64_64;
    for (int i = 0; i < 4; ++i) {
        if (p.cout_pad % cands[i].bc != 0) continue;
        const long tiles = (long)ceil_div(m_total, cands[i].bp) * (p.cout_pad / cands[i].bc) * p.groups;
        const long rounds = (tiles + slots - 1) / slots;
        const double useful = (double)m_total / ((double)ceil_div(m_total, cands[i].bp) * cands[i].bp);
        const double score = cands[i].eff * useful * (double)tiles / (double)(rounds * slots);
        if (score > best) { best = score; pick = cands[i].v; }
    }
    return pick;
}

template <class M, bool OUT_F32>
static int conv2d_h_dispatch(const dz_conv2d_desc &p, size_t w_bytes, hipStream_t stream) {
    switch (conv2d_h_select(p)) {
        case CH_128_128: return launch_conv_h<HTile<128, 128, 32, 2, 2>, M, OUT_F32, 2>(p, w_bytes, stream);
        case CH_128_64: return launch_conv_h<HTile<128, 64, 32, 2, 2>, M, OUT_F32, 3>(p, w_bytes, stream);
        case CH_64_128: return launch_conv_h<HTile<64, 128, 32, 2, 2>, M, OUT_F32, 3>(p, w_bytes, stream);
        case CH_64_64: return launch_conv_h<HTile<64, 64, 32, 2, 2>, M, OUT_F32, 3>(p, w_bytes, stream);
        case CH_128_32: return launch_conv_h<HTile<128, 32, 32, 4, 1>, M, OUT_F32, 3>(p, w_bytes, stream);
        default: break;
    }
    set_error("dz_conv2d_forward_split: unsupported channels cin=%d cout_pad=%d (cin %% 32, cout_pad %% 32 required)", p.cin, p.cout_pad);
    return DZ_ERR_UNSUPPORTED;
}

}  // namespace dz

using namespace dz;

extern "C" {

int dz_conv2d_forward_split(const dz_conv2d_desc *d, int math, int out_f32, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(d && d->in && d->out && d->w, "dz_conv2d_forward_split: null pointer");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2 || math == DZ_MATH_F16, "dz_conv2d_forward_split: math %d is not a split mode", math);
    DZ_CHECK_ARG(d->groups >= 1 && d->groups <= 8, "dz_conv2d_forward_split: groups %d not in [1,8]", d->groups);
    DZ_CHECK_ARG(!d->group_shift || (d->group_rows >= 1 && d->groups == 1 && d->kh == 1 && d->kw == 1 && d->batch == 1 && d->ho == 1 &&
                                     d->out_hp == 1 && d->out_sx == 1 && d->out_dx == 0),
                 "dz_conv2d_forward_split: group_shift is for linear layers (1x1, one image row = the rows), group_rows >= 1");
    DZ_CHECK_ARG(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->cin >= 32, "dz_conv2d_forward_split: bad kernel/cin");
    DZ_CHECK_ARG(d->in_cstride % 8 == 0 && d->in_coff % 8 == 0 && d->cin % 8 == 0,
                 "dz_conv2d_forward_split: input channel stride / offset must be multiples of the 8-channel pair16 group");
    for (int g = 0; g < d->groups; ++g) {
        DZ_CHECK_ARG(d->g_cout[g] >= 1 && d->g_cout[g] <= d->cout_pad, "dz_conv2d_forward_split: bad g_cout[%d]", g);
        DZ_CHECK_ARG(out_f32 || (d->g_cout[g] % 8 == 0 && d->g_ooff[g] % 8 == 0),
                     "dz_conv2d_forward_split: pair16 output needs channel counts / offsets in multiples of 8");
    }
    DZ_CHECK_ARG(out_f32 || (d->out_cstride % 8 == 0 && d->out_coff % 8 == 0),
                 "dz_conv2d_forward_split: pair16 output needs channel stride / offset in multiples of 8");
    DZ_CHECK_ARG((d->ho - 1) * d->stride + d->in_off + d->kh - 1 < d->in_hp &&
                 (d->wo - 1) * d->stride + d->in_off + d->kw - 1 < d->in_wp && d->in_off >= 0,
                 "dz_conv2d_forward_split: taps leave the input image");
    DZ_CHECK_ARG(!d->phase_groups || (d->groups == d->out_sy * d->out_sx && d->kh == 1 && d->kw == 1 && d->stride == 1 && !d->group_shift && !d->group_max),
                 "dz_conv2d_forward_split: phase_groups needs groups == out_sy * out_sx (<= 8) and a 1x1 kernel");
    const int ph_y = d->phase_groups ? d->out_sy - 1 : 0, ph_x = d->phase_groups ? d->out_sx - 1 : 0;
    DZ_CHECK_ARG((d->ho - 1) * d->out_sy + d->out_dy + ph_y < d->out_hp && (d->wo - 1) * d->out_sx + d->out_dx + ph_x < d->out_wp,
                 "dz_conv2d_forward_split: output leaves the output image");
    if ((long)d->batch * d->ho * d->wo == 0) return DZ_OK;
    const size_t w_bytes = (size_t)d->groups * d->kh * d->kw * d->cout_pad * d->cin * sizeof(float);
    if (d->in_rowidx) {
        DZ_CHECK_ARG(d->in_row_channels >= 32 && d->in_rows >= 0, "dz_conv2d_forward_split: in_rowidx needs in_row_channels / in_rows");
        if (!conv3x3_h_variant(*d) || out_f32) {
            set_error("dz_conv2d_forward_split: in_rowidx (sparse input) is implemented for 3 x 3 stride-1 layers with 128-channel output tiles, "
                      "two z slabs (cin == 2 * in_row_channels) and pair16 output");
            return DZ_ERR_UNSUPPORTED;
        }
    }
    {
        const int bc3 = conv3x3_h_variant(*d);
        // in_tiles lists 8 x 32-pixel tiles (dz_bev_tile_list): only the 64 / 128-channel variants of k_conv3x3_h walk that grid.  The
        // 32-channel variant has 16 x 32 tiles - a list there would be decoded on the wrong grid (wrong pixels): refused.  A layer
        // that no tile variant takes (small images, strided layers) runs on the generic kernel, which computes EVERY pixel: the list is
        // ignored there (results stay correct; the caller's fill of the skipped tiles rewrites what was computed).
        if (d->in_tiles && bc3 && !((bc3 == 64 || bc3 == 128) && !out_f32)) {
            set_error("dz_conv2d_forward_split: in_tiles lists 8 x 32 pixel tiles; this layer runs on the 32-channel tile kernel (16 x 32 tiles) - pass no list");
            return DZ_ERR_UNSUPPORTED;
        }
        if (bc3 && (!out_f32 || bc3 == 32)) return conv3x3_h_launch(*d, math, out_f32, w_bytes, stream);      // (fp32 output: 32-channel tiles only)
    }
    if (math == DZ_MATH_F16X2)
        return out_f32 ? conv2d_h_dispatch<MathF16, true>(*d, w_bytes, stream) : conv2d_h_dispatch<MathF16, false>(*d, w_bytes, stream);
    if (math == DZ_MATH_F16)
        return out_f32 ? conv2d_h_dispatch<MathF16H, true>(*d, w_bytes, stream) : conv2d_h_dispatch<MathF16H, false>(*d, w_bytes, stream);
    return out_f32 ? conv2d_h_dispatch<MathBF16, true>(*d, w_bytes, stream) : conv2d_h_dispatch<MathBF16, false>(*d, w_bytes, stream);
}

int dz_linear_forward_split(const float *x, long rows, int cin, int x_stride, const float *w, int cout, int cout_pad, const float *scale,
                            const float *shift, const float *group_shift, int group_rows, int relu, float *y, int y_stride, int math,
                            int out_f32, int group_max, void *stream_) {
    DZ_CHECK_ARG(rows >= 0 && x_stride >= cin && y_stride >= cout && cin % 32 == 0 && cout_pad % 32 == 0 && cout <= cout_pad,
                 "dz_linear_forward_split: bad sizes (cin and cout_pad in multiples of 32)");
    if (group_max) {
        // y = (rows / group_rows, y_stride) fp32: max over every group's rows, fused into the layer's epilogue
        DZ_CHECK_ARG(out_f32 && group_rows >= 128 && group_rows % 128 == 0 && rows % group_rows == 0 && y_stride % 4 == 0 && ((uintptr_t)y & 15u) == 0,
                     "dz_linear_forward_split: group_max needs fp32 output, group_rows %% 128 == 0 (got %d), rows %% group_rows == 0, a 16-byte aligned result", group_rows);
        const int rc = fill_u32(y, 0xFF800000u, (size_t)(rows / group_rows) * y_stride, (hipStream_t)stream_);        // -inf
        if (rc) return rc;
    }
    if (group_rows < 1) group_rows = 1;
    // the rows are fetched through 32-bit buffer offsets: at most ~2 GiB of them per launch
    long max_rows = (long)(0x7FF00000ull / ((size_t)x_stride * sizeof(float)));
    if (group_shift || group_max) max_rows = max_rows / group_rows * group_rows;
    DZ_CHECK_ARG(max_rows >= 1, "dz_linear_forward_split: one row group exceeds the 2 GiB addressing window");
    for (long r0 = 0; r0 < rows; r0 += max_rows) {
        const int n = (int)((rows - r0) < max_rows ? (rows - r0) : max_rows);
        dz_conv2d_desc d = {};
        d.in = x + (size_t)r0 * x_stride; d.out = y + (size_t)(group_max ? r0 / group_rows : r0) * y_stride; d.w = w; d.scale = scale; d.shift = shift;
        d.batch = 1; d.ho = 1; d.wo = n;
        d.in_hp = 1; d.in_wp = n; d.in_cstride = x_stride; d.in_coff = 0; d.cin = cin;
        d.kh = 1; d.kw = 1; d.stride = 1; d.in_off = 0;
        d.out_hp = 1; d.out_wp = n; d.out_cstride = y_stride; d.out_coff = 0;
        d.out_sy = 1; d.out_sx = 1; d.out_dy = 0; d.out_dx = 0;
        d.groups = 1; d.cout_pad = cout_pad; d.g_cout[0] = cout; d.g_ooff[0] = 0; d.relu = relu;
        d.group_shift = group_shift ? group_shift + (size_t)(r0 / group_rows) * cout_pad : nullptr;
        d.group_rows = group_rows;
        d.group_max = group_max ? 1 : 0;
        const int rc = dz_conv2d_forward_split(&d, math, out_f32, stream_);
        if (rc) return rc;
    }
    return DZ_OK;
}

const char *dz_conv2d_variant_split(const dz_conv2d_desc *d) {
    if (!d) return "none";
    const int bc = conv3x3_h_variant(*d);
    if (bc) return bc == 128 ? "k_conv3x3_h<8x32x128>" : bc == 64 ? "k_conv3x3_h<8x32x64>" : "k_conv3x3_h<16x32x32>";
    return kConvHVariantName[conv2d_h_select(*d)];
}

}  // extern "C"
