// Sparse 3-D convolution forward on the 16-bit matrix cores with split-precision (pair16) operands: the
// output-stationary gather -> implicit GEMM -> direct store of sparse_conv.hip (same rulebook, tap skipping,
// fused BatchNorm + bias + residual + ReLU epilogue), fp32-class results at ~5x the fp32-MFMA rate (hgemm.h).
//
// Reference call sites: detection/detzero_det/models/centerpoint_modules/backbone3d.py:64-121, :243-280.
#include <stdlib.h>

#include "hgemm.h"
#include "sparse_conv_w.h"
#ifdef DZ_BUILD_EXPERIMENTAL
#include "sparse_conv_d.h"          // k_spconv_d: operand tiles by direct-to-LDS loads in the gather engine (measured slower)
#endif

namespace dz {

constexpr int KVOL_MAX_H = 27;

// GN = false: the tile's slice of the neighbour table is staged in LDS and scanned for its tap mask first.
// GN = true (tile_masks given): no table in LDS and no per-tile scan - the tap mask comes precomputed and each
// thread fetches the neighbour index of the rows it gathers straight from the table, NS chunks ahead of the gather
// that uses it (a register ring, filled by EXTRA loads behind every stage's loads; see hgemm_pipeline).  Less LDS per
// workgroup = more resident workgroups for the small-channel levels, and the tile prologue disappears.
template <class T, class M, int NS, bool GN, int OCC, int DIAG = 0>
__global__ __launch_bounds__(T::THREADS) __attribute__((amdgpu_waves_per_eu(OCC))) void k_spconv_h(SpConvHArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    v4u *const smem = reinterpret_cast<v4u *>(smem_raw);
    int *const nbr_s = reinterpret_cast<int *>(smem + T::LDS_U4);       // [kvol][BP]   (GN = false only)
    __shared__ unsigned int mask_s;
    __shared__ __attribute__((aligned(16))) float sc_s[T::BC], sh_s[T::BC];     // BatchNorm scale / shift of my channel tile
    constexpr int P = T::P_PER_THREAD;
    constexpr int NST = T::P_PER_THREAD + T::C_PER_THREAD;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wp = wid / T::WC, wc = wid % T::WC;
    const int m = min(*a.d_m_out, a.cap);
    const int ntiles = (m + T::BP - 1) / T::BP;
    const int kchunks = a.cin / T::KC;
    const int n0 = blockIdx.y * T::BC;          // channel tile (cout_pad may be split over blockIdx.y)
    const srsrc_t prsrc = make_srsrc(a.in, a.in_bytes);
    const srsrc_t crsrc = make_srsrc(a.w, a.w_bytes);
    const srsrc_t nrsrc = make_srsrc(a.nbr, a.nbr_bytes);
    unsigned int cvoff[T::C_PER_THREAD];
#pragma unroll
    for (int i = 0; i < T::C_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        cvoff[i] = OOB_OFFSET;
        if (T::C_PIECES % T::THREADS == 0 || idx < T::C_PIECES) {
            const int n = idx / (T::KC / 4), q = idx % (T::KC / 4);
            cvoff[i] = (unsigned int)(((n0 + n) * a.cin + q * 4) * 4);
        }
    }
    const unsigned int tap_bytes = (unsigned int)(a.cout_pad * a.cin * 4);
    const unsigned int nbr_tap_bytes = (unsigned int)a.cap * 4u;
    for (int c = tid; c < T::BC; c += T::THREADS) {
        const bool in = n0 + c < a.cout;
        sc_s[c] = (in && a.scale) ? a.scale[n0 + c] : 1.f;
        sh_s[c] = (in && a.shift) ? a.shift[n0 + c] : 0.f;
    }
    __syncthreads();

    // XCD-aware persistent schedule: workgroup b runs on XCD b % 8 (private L2 each).  Tiles are dealt to the XCDs
    // in runs of XRUN consecutive (spatially sorted) row tiles: a run shares its gathered neighbour rows in one L2,
    // while the round-robin of runs keeps the eight XCDs evenly loaded (whole contiguous eighths were measured slower:
    // tile cost follows the local point density)
    constexpr int XRUN = 16;
    const int xcd = blockIdx.x & 7;
    // (a ticket counter per XCD instead of this static deal was measured: the ticket's round trip per tile costs more
    // than the idle tails it removes)
    for (int t = blockIdx.x >> 3;; t += gridDim.x >> 3) {
        const int tile = ((t / XRUN) * 8 + xcd) * XRUN + t % XRUN;
        if ((t / XRUN) * 8 * XRUN >= ntiles) break;
        if (tile >= ntiles) continue;
        const int row0 = tile * T::BP;
        unsigned int taps;
        if constexpr (GN) {
            taps = 0u;
#pragma unroll
            for (int i = 0; i < T::BP / 32; ++i) taps |= a.tile_masks[tile * (T::BP / 32) + i];
            taps = __builtin_amdgcn_readfirstlane(taps);
        } else {
            if (tid == 0) mask_s = 0u;
            __syncthreads();
            unsigned int local = 0u;
            for (int idx = tid; idx < a.kvol * T::BP; idx += T::THREADS) {
                const int k = idx / T::BP, r = idx % T::BP;
                const int row = row0 + r;
                const int v = (row < m) ? a.nbr[(size_t)k * a.cap + row] : -1;
                nbr_s[idx] = v;
                if (v >= 0) local |= 1u << k;
            }
            if (local) atomicOr(&mask_s, local);
            __syncthreads();
            taps = mask_s;
        }

        f32x16 acc[T::CT][T::PT];
#pragma unroll
        for (int i = 0; i < T::CT; ++i)
#pragma unroll
            for (int j = 0; j < T::PT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        const int nchunks = __popc(taps) * kchunks;
        if (nchunks > 0) {
            unsigned int rem = taps;
            int tap = __ffs((int)rem) - 1, kc = 0;
            // chunk order: channel chunk outermost, taps innermost - neighbouring taps gather mostly the same input
            // rows, so their KC-channel slices are re-read back to back while they are still in the CU's L1
            auto advance = [&]() {
                rem &= rem - 1;
                if (rem == 0u) { rem = taps; ++kc; }
                tap = __ffs((int)rem) - 1;
            };
            if constexpr (GN) {
                // neighbour-index ring: slot s holds the indices of the chunk that will be gathered into stage s next
                int nb[NS][P];
                unsigned int nvoff[P];          // byte offset of this thread's rows in one tap of the table
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const int idx = tid + i * T::THREADS;
                    const int row = row0 + idx / (T::KC / 4);
                    nvoff[i] = ((T::P_PIECES % T::THREADS == 0 || idx < T::P_PIECES) && row < m) ? (unsigned int)row * 4u : OOB_OFFSET;
                }
                unsigned int rem_a = taps;                  // tap iterator of the ring: NS chunks ahead of (rem, tap)
                int tap_a = __ffs((int)rem_a) - 1;
                auto fetch_nbr = [&](int (&dst)[P]) {
                    const unsigned int toff = (unsigned int)tap_a * nbr_tap_bytes;
#pragma unroll
                    for (int i = 0; i < P; ++i)
                        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "=v"(dst[i]) : "v"(nvoff[i] == OOB_OFFSET ? OOB_OFFSET : nvoff[i] + toff), "s"(nrsrc));
                    rem_a &= rem_a - 1;
                    if (rem_a == 0u) rem_a = taps;          // keeps cycling past the last chunk: those fetches are never used
                    tap_a = __ffs((int)rem_a) - 1;
                };
#pragma unroll
                for (int s = 0; s < NS; ++s) fetch_nbr(nb[s]);
                asm volatile("s_waitcnt vmcnt(0)");
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int i = 0; i < P; ++i) asm volatile("" : "+v"(nb[s][i]));
                auto issue = [&](HStage<T> &st, auto s_t) {
                    constexpr int S = decltype(s_t)::value;
                    // the fetch into slot S was issued NS calls ago, right behind that call's stage loads
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * (NST + (DIAG == 7 ? 0 : P))));
                    unsigned int pvoff[P];
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        asm volatile("" : "+v"(nb[S][i]));
                        const int q = (tid + i * T::THREADS) % (T::KC / 4);
                        const int rb = nb[S][i];
                        pvoff[i] = (nvoff[i] != OOB_OFFSET && rb >= 0) ? (unsigned int)rb * (unsigned int)(a.cin * 4) + (unsigned int)(q * 16) : OOB_OFFSET;
                    }
                    if constexpr (DIAG == 8) {      // gathers out of range: instructions issued, nothing fetched
#pragma unroll
                        for (int i = 0; i < P; ++i) pvoff[i] = OOB_OFFSET;
                    }
                    unsigned int cv2[T::C_PER_THREAD];
#pragma unroll
                    for (int i = 0; i < T::C_PER_THREAD; ++i) cv2[i] = DIAG == 9 ? OOB_OFFSET : cvoff[i];
                    load_hstage<T>(st, prsrc, pvoff, (unsigned int)(kc * T::KC * 4), crsrc, cv2,
                                   (unsigned int)tap * tap_bytes + (unsigned int)(kc * T::KC * 4));
                    if constexpr (DIAG != 7) fetch_nbr(nb[S]);
                };
                hgemm_pipeline<T, M, NS, DIAG == 7 ? 0 : P, DIAG, (T::THREADS == 512)>(nchunks, smem, issue, advance, acc, wp, wc, lane, tid);
            } else {
                auto issue = [&](HStage<T> &st, auto) {
                    unsigned int pvoff[P];
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        const int idx = tid + i * T::THREADS;
                        pvoff[i] = OOB_OFFSET;
                        if (T::P_PIECES % T::THREADS == 0 || idx < T::P_PIECES) {
                            const int rr = idx / (T::KC / 4), q = idx % (T::KC / 4);
                            const int rb = nbr_s[tap * T::BP + rr];
                            pvoff[i] = rb >= 0 ? (unsigned int)rb * (unsigned int)(a.cin * 4) + (unsigned int)(q * 16) : OOB_OFFSET;
                        }
                    }
                    load_hstage<T>(st, prsrc, pvoff, (unsigned int)(kc * T::KC * 4), crsrc, cvoff,
                                   (unsigned int)tap * tap_bytes + (unsigned int)(kc * T::KC * 4));
                };
                hgemm_pipeline<T, M, NS>(nchunks, smem, issue, advance, acc, wp, wc, lane, tid);
            }
        }

        // epilogue through the (now idle) tile buffers: see store_tile_pair16
        store_tile_pair16<T, M>(acc, smem_raw, sc_s, sh_s, n0, a.cout, a.relu != 0, reinterpret_cast<const unsigned char *>(a.residual),
                                reinterpret_cast<unsigned char *>(a.out), wp, wc, lane, wid, [&](int lr) {
                                    const int row = row0 + lr;
                                    return row < m ? (size_t)row * a.cout * 4 : ~size_t(0);
                                });
        __syncthreads();      // the next tile's first stage (and, without the ring, its table slice) overwrites the buffers
    }
}

// development knob: DZ_TUNE_<name>=<int> in the environment overrides a tile choice (read once)
static int tune(const char *name, int dflt) {
    const char *v = getenv(name);
    return v ? atoi(v) : dflt;
}

template <class T, class M, int NS, bool GN, int OCC, int DIAG = 0>
static int launch_spconv_h_impl(const SpConvHArgs &a, hipStream_t stream) {
    constexpr int LDS = T::LDS_BYTES + (GN ? 0 : KVOL_MAX_H * T::BP * 4);
    static PerDeviceFlags lds_done;
    if (int rc_ = reserve_lds(reinterpret_cast<const void *>(&k_spconv_h<T, M, NS, GN, OCC, DIAG>), LDS, lds_done, "dz_spconv_forward_split")) return rc_;
    int grid = ceil_div(a.cap, T::BP);
    if (grid > 2048) grid = 2048;
    grid = (grid + 7) & ~7;            // a multiple of 8: see the XCD schedule in the kernel
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL((k_spconv_h<T, M, NS, GN, OCC, DIAG>), dim3(grid, a.cout_pad / T::BC), dim3(T::THREADS), LDS, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

// NS / NSG: register stages of the LDS-table and of the register-ring variant (the ring costs registers: one stage
// less where the extra registers would cost a resident wave)
// OCC / OCCG: the resident waves per SIMD the register allocation has to leave room for
template <class T, class M, int NS, int NSG = NS, int OCC = 1, int OCCG = OCC>
static int launch_spconv_h(const SpConvHArgs &a, hipStream_t stream, bool ring_ok = true) {
    static const int no_gn = tune("DZ_TUNE_SPCONV_NOGN", 0);
    if (a.tile_masks && a.nbr_bytes && !no_gn && ring_ok) return launch_spconv_h_impl<T, M, NSG, true, OCCG>(a, stream);
    return launch_spconv_h_impl<T, M, NS, false, OCC>(a, stream);
}

// Tile choices (per launch at 16 frames, r01e; alternatives that were built and measured slower or equal are listed in
// 64-row tiles with 4 waves - the r01d defaults, kept below as knob 1 -, 256-row tiles with 64 x 64 wave tiles,
// 4 register stages, 128 x 128 with 4 waves).
template <class M>
static int spconv_h_dispatch(const SpConvHArgs &a, hipStream_t stream) {
    // (t128: 3 since round 5 - the 128-channel gather kernel now serves only the strided 64 -> 128 layer and conv_out, the x-run engine
    // has the submanifold layers: 1079.8 / 1078.1 against 1074.4 / 1075.0 frames/s, profiles/r05_ab_notes.txt; 4 = the r03-r04 tile)
    static const int t64 = tune("DZ_TUNE_SPCONV64", 0), t128 = tune("DZ_TUNE_SPCONV128", 3), tw = tune("DZ_TUNE_SPCONV_W", 1);
    // small-channel levels: wave-private tiles with all weights resident in LDS (sparse_conv_w.h)
    if (tw && a.cout_pad == 32 && a.tile_masks && a.nbr_bytes) {
        if (a.cin == 16 && a.cout == 16) {
            // workgroup size (waves) x waves per SIMD: development knob.  Round 5: 4 waves per workgroup (three workgroups per CU)
            // instead of 6 (two per CU): 238 vs 282 us per launch at 32 frames - the waves of a smaller workgroup drift apart less
            // before the weights' LDS reads and share the CU with two independent neighbours
            static const int w16 = tune("DZ_TUNE_W16", 2);
            if (w16 == 0) return launch_spconv_w<16, 16, 4, M, 6, 3>(a, stream);
            if (w16 == 1) return launch_spconv_w<16, 16, 4, M, 8, 4>(a, stream);
            if (w16 == 6) return launch_spconv_w<16, 16, 4, M, 2, 3>(a, stream);
            if (w16 == 7) return launch_spconv_w<16, 16, 4, M, 3, 3>(a, stream);
            return launch_spconv_w<16, 16, 4, M, 4, 3>(a, stream);
        }
        if (a.cin == 16 && a.cout == 32) {
            static const int w1632 = tune("DZ_TUNE_W1632", 0);
            if (w1632 == 1) return launch_spconv_w<16, 32, 3, M, 4, 3>(a, stream);
            if (w1632 == 2) return launch_spconv_w<16, 32, 3, M, 3, 3>(a, stream);
            return launch_spconv_w<16, 32, 3, M, 6, 3>(a, stream);
        }
        if (a.cin == 32 && a.cout == 32) return launch_spconv_w<32, 32, 2, M, 12, 3>(a, stream);
    }
    if (a.cin == 16 && a.cout_pad == 32) return launch_spconv_h<HTile<128, 32, 16, 4, 1>, M, 4, 3, 4, 5>(a, stream);
    if (a.cin == 32 && a.cout_pad == 32) return launch_spconv_h<HTile<128, 32, 32, 4, 1>, M, 3, 3, 1, 3>(a, stream);
#ifdef DZ_BUILD_EXPERIMENTAL
    static const int td = tune("DZ_TUNE_SPCONV_D", 0);          // operand tiles by direct-to-LDS loads (sparse_conv_d.h)
    if (td && a.tile_masks && a.nbr_bytes && a.cin % 32 == 0) {
        if (a.cout_pad == 64 && (td & 1)) return launch_spconv_d<64, M>(a, stream);
        if (a.cout_pad == 128 && (td & 2)) return launch_spconv_d<128, M>(a, stream);
    }
#endif
    if ((a.cin == 32 || a.cin == 64) && a.cout_pad == 64) {
        if (t64 == 1) return launch_spconv_h<HTile<64, 64, 32, 2, 2>, M, 4, 3, 3, 3>(a, stream);       // 4 waves of 32 x 32 (r01d)
        if (t64 == 2) return launch_spconv_h<HTile<128, 64, 32, 4, 2>, M, 3, 3, 4, 4>(a, stream);      // 8 waves of 32 x 32 (r01e-r03b)
        // 8 waves of 64 x 32 over 256 rows: 12 instead of 16 fragment reads per 12 MFMAs, the weight slice fetched once per 256 rows
        // (r03: LDS fragment reads had become the largest single item of the diag breakdown of the 128-channel kernel, -31 %; inside
        // the detector, A/B on one box, two rounds: 861.2 / 861.9 against 855.3 / 853.3 frames/s, +0.2 % on another box)
        return launch_spconv_h<HTile<256, 64, 32, 4, 2>, M, 3, 3, 2, 2>(a, stream);
    }
    if ((a.cin == 64 || a.cin == 128) && a.cout_pad == 128) {
#ifdef DZ_SPCONV_DIAG
        if (t128 >= 11 && t128 <= 20 && a.tile_masks && a.nbr_bytes) {
            using DT = HTile<128, 128, 32, 4, 2>;
            if (t128 == 11) return launch_spconv_h_impl<DT, M, 3, true, 2, 1>(a, stream);
            if (t128 == 12) return launch_spconv_h_impl<DT, M, 3, true, 2, 2>(a, stream);
            if (t128 == 13) return launch_spconv_h_impl<DT, M, 3, true, 2, 3>(a, stream);
            if (t128 == 14) return launch_spconv_h_impl<DT, M, 3, true, 2, 4>(a, stream);
            if (t128 == 16) return launch_spconv_h_impl<DT, M, 3, true, 2, 6>(a, stream);
            if (t128 == 17) return launch_spconv_h_impl<DT, M, 3, true, 2, 7>(a, stream);
            if (t128 == 18) return launch_spconv_h_impl<DT, M, 3, true, 2, 8>(a, stream);
            if (t128 == 20) return launch_spconv_h_impl<DT, M, 3, true, 2, 10>(a, stream);
            return launch_spconv_h_impl<DT, M, 3, true, 2, 9>(a, stream);
        }
#endif
        if (t128 == 1) return launch_spconv_h<HTile<64, 128, 32, 2, 2>, M, 3>(a, stream, !(a.cin == 128 && a.kvol == 27));   // 4 waves (r01d)
        if (t128 == 2) return launch_spconv_h<HTile<128, 128, 32, 4, 2>, M, 3, 3, 2, 2>(a, stream);    // three register stages (r01e-r02f)
        // 8 waves of 64 x 64 over 256 rows, two register stages (240 registers): 16 instead of 24 fragment reads per 24 MFMAs.  r03,
        // A/B inside the detector: -0.2 % / -0.45 % of a pass on two boxes, +3 % together with the 256-row 64-channel tile on a third:
        // not the default
        if (t128 == 3) return launch_spconv_h<HTile<256, 128, 32, 4, 2>, M, 2, 2, 2, 2>(a, stream);
        // 8 waves of 32 x 64, four register stages with the ring (202 registers at two waves per SIMD): +0.3 % of a pass over three,
        // A/B inside the detector on one box (tools/gpu_ab_env.sh) - gather latency is not what limits this kernel
        return launch_spconv_h<HTile<128, 128, 32, 4, 2>, M, 3, 4, 2, 2>(a, stream);
    }
    set_error("dz_spconv_forward_split: unsupported channels cin=%d cout=%d", a.cin, a.cout);
    return DZ_ERR_UNSUPPORTED;
}

}  // namespace dz

using namespace dz;

template <class M>
static int spconv_w_packed_dispatch(const SpConvHArgs &a, hipStream_t stream) {
    if (a.cin == 16 && a.cout == 16) return launch_spconv_w<16, 16, 4, M, 6, 3, true>(a, stream);
    if (a.cin == 16 && a.cout == 32) return launch_spconv_w<16, 32, 3, M, 6, 3, true>(a, stream);
    if (a.cin == 32 && a.cout == 32) return launch_spconv_w<32, 32, 2, M, 12, 3, true>(a, stream);
    set_error("dz_spconv_forward_split_packed: %d -> %d channels (the packed table feeds the 16 -> 16, 16 -> 32 and 32 -> 32 kernels)", a.cin, a.cout);
    return DZ_ERR_UNSUPPORTED;
}

extern "C" {

int dz_spconv_forward_split(const float *in, int in_rows, int cin, const int *nbr, const uint32_t *tile_masks, int kvol, int cap_out,
                            const int *d_m_out, const float *w, const float *scale, const float *shift, const float *residual,
                            int relu, float *out, int cout, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(in && nbr && d_m_out && w && out, "dz_spconv_forward_split: null pointer");
    DZ_CHECK_ARG(kvol >= 1 && kvol <= KVOL_MAX_H, "dz_spconv_forward_split: kvol %d not in [1,27]", kvol);
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2 || math == DZ_MATH_F16, "dz_spconv_forward_split: math %d is not a split mode", math);
    DZ_CHECK_ARG(cout % 8 == 0 && cin % 8 == 0, "dz_spconv_forward_split: channels must be multiples of the 8-channel pair16 group");
    if (cap_out == 0) return DZ_OK;
    const int cout_pad = cout < 32 ? 32 : cout;
    const size_t in_bytes = (size_t)in_rows * cin * sizeof(float);
    const size_t w_bytes = (size_t)kvol * cout_pad * cin * sizeof(float);
    if (in_rows < 0 || in_bytes >= 0x80000000ull) {
        set_error("dz_spconv_forward_split: input of %zu bytes exceeds the 2 GiB buffer-addressing limit", in_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    // the table is addressed through a buffer descriptor when it fits its 2 GiB window (else: staged through LDS)
    const size_t nbr_bytes = (size_t)kvol * cap_out * sizeof(int);
    SpConvHArgs a{in, nbr, tile_masks, d_m_out, w, scale, shift, residual, out, cin, cout, cout_pad, kvol, cap_out, relu,
                  (unsigned int)in_bytes, (unsigned int)w_bytes, nbr_bytes < 0x80000000ull ? (unsigned int)nbr_bytes : 0u,
                  tile_masks ? (unsigned int)tile_masks_words(cap_out) * 4u : 0u, 0};
#ifdef DZ_SPCONV_DIAG
    a.diag = tune("DZ_TUNE_W_DIAG", 0);
#endif
    if (math == DZ_MATH_F16) return spconv_h_dispatch<MathF16H>(a, stream);
    return math == DZ_MATH_F16X2 ? spconv_h_dispatch<MathF16>(a, stream) : spconv_h_dispatch<MathBF16>(a, stream);
}

int dz_spconv_forward_split_packed(const float *in, int in_rows, int cin, const int *nbr_packed, const uint32_t *tile_masks, int cap_out,
                                   const int *d_m_out, const float *w, const float *scale, const float *shift, const float *residual,
                                   int relu, float *out, int cout, int math, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    DZ_CHECK_ARG(in && nbr_packed && tile_masks && d_m_out && w && out, "dz_spconv_forward_split_packed: null pointer");
    DZ_CHECK_ARG(math == DZ_MATH_F16X2 || math == DZ_MATH_BF16X2 || math == DZ_MATH_F16, "dz_spconv_forward_split_packed: math %d is not a split mode", math);
    if (cap_out == 0) return DZ_OK;
    const int kvol = 27, cout_pad = 32;
    const size_t in_bytes = (size_t)in_rows * cin * sizeof(float);
    const size_t w_bytes = (size_t)kvol * cout_pad * cin * sizeof(float);
    const size_t nbr_bytes = (size_t)9 * cap_out * sizeof(int);
    if (in_rows < 0 || in_bytes >= 0x80000000ull || nbr_bytes >= 0x80000000ull) {
        set_error("dz_spconv_forward_split_packed: input of %zu / table of %zu bytes exceeds the 2 GiB buffer-addressing limit", in_bytes, nbr_bytes);
        return DZ_ERR_UNSUPPORTED;
    }
    SpConvHArgs a{in, nbr_packed, tile_masks, d_m_out, w, scale, shift, residual, out, cin, cout, cout_pad, kvol, cap_out, relu,
                  (unsigned int)in_bytes, (unsigned int)w_bytes, (unsigned int)nbr_bytes, (unsigned int)tile_masks_words(cap_out) * 4u, 0};
    if (math == DZ_MATH_F16) return spconv_w_packed_dispatch<MathF16H>(a, stream);
    return math == DZ_MATH_F16X2 ? spconv_w_packed_dispatch<MathF16>(a, stream) : spconv_w_packed_dispatch<MathBF16>(a, stream);
}

const char *dz_spconv_variant_split(int cin, int cout) {
    const int cout_pad = cout < 32 ? 32 : cout;
    if (tune("DZ_TUNE_SPCONV_W", 1) && cout_pad == 32) {        // (needs the table's tile masks, which dz_build_neighbors always writes)
        if (cin == 16 && cout == 16) return "k_spconv_w<16x16>";
        if (cin == 16 && cout == 32) return "k_spconv_w<16x32>";
        if (cin == 32 && cout == 32) return "k_spconv_w<32x32>";
    }
    if (cin == 16 && cout_pad == 32) return "k_spconv_h<128x32x16>";
    if (cin == 32 && cout_pad == 32) return "k_spconv_h<128x32x32>";
    if ((cin == 32 || cin == 64) && cout_pad == 64)
        return tune("DZ_TUNE_SPCONV64", 0) == 2 ? "k_spconv_h<128x64x32>" : "k_spconv_h<256x64x32>";
    if ((cin == 64 || cin == 128) && cout_pad == 128)
        return tune("DZ_TUNE_SPCONV128", 3) == 3 ? "k_spconv_h<256x128x32>" : "k_spconv_h<128x128x32>";
    return "none";
}

}  // extern "C"
