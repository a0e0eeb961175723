// Split-precision implicit-GEMM tile engine (gfx950, wave64): fp32-class results on the 16-bit matrix cores.
//
// Every fp32 value x is carried as a PAIR of 16-bit floats  x ~= hi + lo,  hi = rn16(x), lo = rn16(x - hi)
// (fp16: 22 significant bits, bf16: 16), and a product of two such values is evaluated as
//        a.b  ~=  a_hi.b_hi + a_lo.b_hi + a_hi.b_lo            (the lo.lo term, <= 2^-22 relative, is dropped)
// with three v_mfma_f32_32x32x16_{f16,bf16} accumulating in fp32.  The 16-bit matrix pipe runs 16x the rate
// of v_mfma_f32_16x16x4_f32 (2.5 PF/s vs 157 TF/s dense, /opt/skills/guides/MI355X_MICROARCH.md), so three
// of them per product still leave ~5x the fp32-MFMA throughput, and the error of the fp16 pair (measured
// through the whole CenterPoint network: 5e-7 on the head maps) is of the order of fp32 summation-order noise.
//
// "pair16" storage (activations in HBM, weights, LDS tiles): a row of C channels (C % 8 == 0) is C 32-bit
// words, exactly the size of the fp32 row it replaces; each group of 8 channels is 16 bytes of hi values
// followed by 16 bytes of lo values.  A 16-byte load of either half IS the matrix-core operand of its lane
// (8 consecutive k), so tiles go HBM -> registers -> LDS -> MFMA without a single conversion or shuffle; the
// producing kernel's epilogue does the split once per output value.
//
// Orientation: D[cout x pixel] = W[cout x k] . X^T[k x pixel].  The 32x32 accumulator then holds, per lane,
// 4 consecutive output channels of one pixel per register quad -> 8-byte hi / lo stores straight into the
// pair16 row of that pixel.
//
// The chunk pipeline (double-buffered LDS, one barrier per KC-channel chunk, register double buffer of the
// fragments, global loads two chunks ahead) is the one of igemm.h.
#pragma once
#include "igemm.h"

namespace dz {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4v = __attribute__((ext_vector_type(4))) float;
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef __bf16 b8_t __attribute__((ext_vector_type(8)));

struct MathF16 {
    static constexpr int ID = 1;
    static constexpr int TERMS = 3;            // MFMAs per product: hi.hi + lo.hi + hi.lo
    static constexpr bool Q16 = false;         // (MathF16Q: the q16 tensor format of the fp16 + fp8 prototype)
    static __device__ __forceinline__ f32x16 mma(v4u a, v4u b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4v mma16(v4u a, v4u b, f32x4v c) {        // 16x16x32: lane = (row or column) & 15, k group = lane >> 4
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8_t, a), __builtin_bit_cast(h8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(float v, unsigned int &hi, unsigned int &lo) {
        // saturate instead of overflowing to inf: |v| beyond the fp16 range keeps hi = +-65504 and the rest in lo
        const float vc = fminf(fmaxf(v, -65504.f), 65504.f);
        const _Float16 h = (_Float16)vc;
        const float rem = fminf(fmaxf(v - (float)h, -65504.f), 65504.f);
        const _Float16 l = (_Float16)rem;
        hi = (unsigned int)__builtin_bit_cast(unsigned short, h);
        lo = (unsigned int)__builtin_bit_cast(unsigned short, l);
    }
    static __device__ __forceinline__ float join(unsigned int hi, unsigned int lo) {
        return (float)__builtin_bit_cast(_Float16, (unsigned short)hi) + (float)__builtin_bit_cast(_Float16, (unsigned short)lo);
    }
};

struct MathBF16 {
    static constexpr int ID = 2;
    static constexpr int TERMS = 3;
    static constexpr bool Q16 = false;
    static __device__ __forceinline__ f32x16 mma(v4u a, v4u b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ f32x4v mma16(v4u a, v4u b, f32x4v c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(b8_t, a), __builtin_bit_cast(b8_t, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(float v, unsigned int &hi, unsigned int &lo) {
        const __bf16 h = (__bf16)v;
        const __bf16 l = (__bf16)(v - (float)h);
        hi = (unsigned int)__builtin_bit_cast(unsigned short, h);
        lo = (unsigned int)__builtin_bit_cast(unsigned short, l);
    }
    static __device__ __forceinline__ float join(unsigned int hi, unsigned int lo) {
        return __uint_as_float(hi << 16) + __uint_as_float(lo << 16);
    }
};

// DZ_MATH_F16 ('f16'): the same pair16 tensors (results are still stored as hi + lo), but a product is the single fp16 MFMA hi.hi -
// plain fp16 inputs with fp32 accumulation, a third of the matrix work.  NOT fp32-class: inputs carry 11 significant bits per layer
// (opt-in; boxes move by up to ~1e-2 against the fp32 path - tests/test_gpu_f16.py states the measured tolerance).
struct MathF16H : MathF16 {
    static constexpr int TERMS = 1;
};
template <class T>
struct TypeTag { using type = T; };        // carries a math type through generic lambdas

// PROTOTYPE (diag builds of conv3x3_h.hip only): hi.hi on the fp16 pipe + both correction terms in ONE block-scaled fp8
// MFMA.  Tensor format "q16": a 32-channel block is 128 bytes = [hi: 32 x fp16][hi8: 32 x fp8 e4m3 of hi * 2^-E][lo8: 32 x fp8 of
// (x - hi) * 2^(11-E)] - the footprint of pair16, 15 significant bits, one power-of-two scale E per tensor.
struct MathF16Q : MathF16 {
    static constexpr bool Q16 = true;
};
typedef int v8i_t __attribute__((ext_vector_type(8)));
// D += 2^(sa + sb - 254) * sum over k of A[k] B[k], 64 fp8 values per row: lanes 0-31 hold k 0..31, lanes 32-63 k 32..63
__device__ __forceinline__ f32x16 mma_f8(v4u a0, v4u a1, v4u b0, v4u b1, f32x16 c, int sa, int sb) {
    const v8i_t a = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
    const v8i_t b = {(int)b0.x, (int)b0.y, (int)b0.z, (int)b0.w, (int)b1.x, (int)b1.y, (int)b1.z, (int)b1.w};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
}

// 4 consecutive channels -> the two 8-byte halves (hi, lo) of their slot in a pair16 group.  Two values at a time on the packed
// conversions of gfx950 (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32, v_pk_add_f32): ~4.5 VALU ops per value instead of ~10 for the
// scalar M::split (same roundings and the same saturation rule, so the same bits).
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef __bf16 b2_t __attribute__((ext_vector_type(2)));

template <class M>
__device__ __forceinline__ void split2(float a, float b, unsigned int &hi, unsigned int &lo) {
    if constexpr (M::ID == 1) {
        f32x2v v = {__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b, -65504.f, 65504.f)};
        const h2_t h = __builtin_convertvector(v, h2_t);
        f32x2v rem = f32x2v{a, b} - __builtin_convertvector(h, f32x2v);
        rem.x = __builtin_amdgcn_fmed3f(rem.x, -65504.f, 65504.f);
        rem.y = __builtin_amdgcn_fmed3f(rem.y, -65504.f, 65504.f);
        const h2_t l = __builtin_convertvector(rem, h2_t);
        hi = __builtin_bit_cast(unsigned int, h);
        lo = __builtin_bit_cast(unsigned int, l);
    } else {
        const f32x2v v = {a, b};
        const b2_t h = __builtin_convertvector(v, b2_t);
        const unsigned int hb = __builtin_bit_cast(unsigned int, h);
        const f32x2v back = {__uint_as_float(hb << 16), __uint_as_float(hb & 0xFFFF0000u)};
        const b2_t l = __builtin_convertvector(v - back, b2_t);
        hi = hb;
        lo = __builtin_bit_cast(unsigned int, l);
    }
}

template <class M>
__device__ __forceinline__ void split4(const float (&v)[4], uint2 &hi, uint2 &lo) {
    split2<M>(v[0], v[1], hi.x, lo.x);
    split2<M>(v[2], v[3], hi.y, lo.y);
}

// BP pixels (MFMA N side) x BC output channels (MFMA M side), KC channels of one tap per chunk, 4 or 8 waves as WP x WC
template <int BP_, int BC_, int KC_, int WP_, int WC_>
struct HTile {
    static constexpr int BP = BP_, BC = BC_, KC = KC_, WP = WP_, WC = WC_;
    static constexpr int THREADS = 64 * WP_ * WC_;
    static constexpr int PT = BP / (32 * WP);             // 32-pixel fragments per wave
    static constexpr int CT = BC / (32 * WC);             // 32-channel fragments per wave
    static constexpr int ROW_U4 = KC / 4 + 1;             // LDS row stride in 16-byte units (+1: conflict-free ds_read_b128)
    static constexpr int P_PIECES = BP * (KC / 4);        // 16-byte pieces of the pixel chunk
    static constexpr int C_PIECES = BC * (KC / 4);
    static constexpr int P_PER_THREAD = (P_PIECES + THREADS - 1) / THREADS;
    static constexpr int C_PER_THREAD = (C_PIECES + THREADS - 1) / THREADS;
    static constexpr int PS_U4 = BP * ROW_U4;             // one buffer
    static constexpr int CS_U4 = BC * ROW_U4;
    static constexpr int LDS_U4 = 2 * (PS_U4 + CS_U4);    // double buffered
    static constexpr int LDS_BYTES = LDS_U4 * 16;
    static_assert(WP * WC == 4 || WP * WC == 8, "4 or 8 waves per workgroup");
    static_assert(BP % (32 * WP) == 0 && BC % (32 * WC) == 0, "tile must split into 32x32 fragments");
    static_assert(KC == 16 || KC == 32, "KC is one or two 16-deep MFMA steps");
};

template <class T>
struct HFrag {
    v4u p_hi[T::PT], p_lo[T::PT], c_hi[T::CT], c_lo[T::CT];
};

// ps / cs point at the lane's row and k-group: row (lane & 31), 16-byte unit 2 * (lane >> 5)
template <class T>
__device__ __forceinline__ void load_hfrag(HFrag<T> &f, const v4u *__restrict__ ps, const v4u *__restrict__ cs, int q) {
#pragma unroll
    for (int pt = 0; pt < T::PT; ++pt) {
        f.p_hi[pt] = ps[pt * 32 * T::ROW_U4 + q * 4];
        f.p_lo[pt] = ps[pt * 32 * T::ROW_U4 + q * 4 + 1];
    }
#pragma unroll
    for (int ct = 0; ct < T::CT; ++ct) {
        f.c_hi[ct] = cs[ct * 32 * T::ROW_U4 + q * 4];
        f.c_lo[ct] = cs[ct * 32 * T::ROW_U4 + q * 4 + 1];
    }
}

// TM (term-major): the three terms outermost, so that consecutive MFMAs go to different accumulators when a wave has more than
// one (same bits: every accumulator still receives lo.hi, hi.lo, hi.hi in that order)
template <class T, class M, bool TM = false>
__device__ __forceinline__ void mma_hfrag(const HFrag<T> &f, f32x16 (&acc)[T::CT][T::PT]) {
    if constexpr (M::TERMS == 1) {              // single product on the hi halves (the lo fragment reads are dead code and disappear)
#pragma unroll
        for (int ct = 0; ct < T::CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < T::PT; ++pt) acc[ct][pt] = M::mma(f.c_hi[ct], f.p_hi[pt], acc[ct][pt]);
    } else if constexpr (TM) {
#pragma unroll
        for (int term = 0; term < 3; ++term)
#pragma unroll
            for (int ct = 0; ct < T::CT; ++ct)
#pragma unroll
                for (int pt = 0; pt < T::PT; ++pt)
                    acc[ct][pt] = M::mma(term == 0 ? f.c_lo[ct] : f.c_hi[ct], term == 1 ? f.p_lo[pt] : f.p_hi[pt], acc[ct][pt]);
    } else {
#pragma unroll
        for (int ct = 0; ct < T::CT; ++ct)
#pragma unroll
            for (int pt = 0; pt < T::PT; ++pt) {
                acc[ct][pt] = M::mma(f.c_lo[ct], f.p_hi[pt], acc[ct][pt]);
                acc[ct][pt] = M::mma(f.c_hi[ct], f.p_lo[pt], acc[ct][pt]);
                acc[ct][pt] = M::mma(f.c_hi[ct], f.p_hi[pt], acc[ct][pt]);
            }
    }
}

template <class T>
struct HStage {
    v4u p[T::P_PER_THREAD];
    v4u c[T::C_PER_THREAD];
};

// Global loads of the 16-byte pieces.  They are issued from inline asm so that THIS file owns the vmcnt
// bookkeeping: with builtin loads the compiler's waitcnt pass cannot see through the rotating register stages of
// the NS-deep pipeline and drains the whole queue (vmcnt(0)) once per trip.  Contract:
//   * every thread issues exactly NST = P_PER_THREAD + C_PER_THREAD loads per stage (threads without a piece use
//     the out-of-range offset: zeros come back, nothing is fetched), so the per-wave counter is uniform;
//   * loads return in order, so "stage s has landed" == vmcnt(<= loads issued after it) (wait_hstage);
//   * a stage's registers are touched only through wait_hstage's "+v" operands before they are stored to LDS.
// Offsets past the buffer read as zeros (missing neighbours, rows past the end of the image).
using srsrc_t = v4u;     // buffer resource words held in 4 consecutive SGPRs

__device__ __forceinline__ srsrc_t make_srsrc(const void *base, unsigned int bytes) {
    const unsigned long long a = (unsigned long long)base;
    srsrc_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned int)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned int)(a >> 32) & 0xFFFFu);      // stride 0
    r.z = __builtin_amdgcn_readfirstlane(bytes);                                   // num_records (bytes)
    r.w = 0x00020000u;
    return r;
}

template <class T>
__device__ __forceinline__ void load_hstage(HStage<T> &st, srsrc_t prsrc, const unsigned int (&pvoff)[T::P_PER_THREAD],
                                            unsigned int padd, srsrc_t crsrc, const unsigned int (&cvoff)[T::C_PER_THREAD],
                                            unsigned int cadd) {
#pragma unroll
    for (int i = 0; i < T::P_PER_THREAD; ++i)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st.p[i]) : "v"(pvoff[i] + padd), "s"(prsrc));
#pragma unroll
    for (int i = 0; i < T::C_PER_THREAD; ++i)
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(st.c[i]) : "v"(cvoff[i] + cadd), "s"(crsrc));
}

// wait until at most N younger loads are outstanding, then hand the stage's registers back to the compiler
template <class T, int N>
__device__ __forceinline__ void wait_hstage(HStage<T> &st) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));
#pragma unroll
    for (int i = 0; i < T::P_PER_THREAD; ++i) asm volatile("" : "+v"(st.p[i]));
#pragma unroll
    for (int i = 0; i < T::C_PER_THREAD; ++i) asm volatile("" : "+v"(st.c[i]));
}

template <class T>
__device__ __forceinline__ void store_hstage(const HStage<T> &st, v4u *__restrict__ Ps, v4u *__restrict__ Cs, int tid) {
#pragma unroll
    for (int i = 0; i < T::P_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        if (T::P_PIECES % T::THREADS == 0 || idx < T::P_PIECES) Ps[(idx / (T::KC / 4)) * T::ROW_U4 + idx % (T::KC / 4)] = st.p[i];
    }
#pragma unroll
    for (int i = 0; i < T::C_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        if (T::C_PIECES % T::THREADS == 0 || idx < T::C_PIECES) Cs[(idx / (T::KC / 4)) * T::ROW_U4 + idx % (T::KC / 4)] = st.c[i];
    }
}

// ---- epilogue of a pair16 result, staged through LDS ---------------------------------------------------------------------
// The MFMA accumulator layout gives a lane 4 of the 8 channels of a pair16 group for ONE row (row = lane & 31, channel =
// 8*(reg>>2) + 4*(lane>>5) + (reg&3)): stored from there, every store instruction is 64 separate 8-byte requests 256-512 bytes
// apart, the residual is read the same way, and each (fragment, group) step waits for its own scale / shift load because the
// stores in between may alias (measured on the resident-tile 3x3 kernel: 19 % of the kernel).  Instead each 32-row x 32-channel
// fragment is written as raw fp32 into a wave-private LDS window (32 rows x 144 bytes, in the tile buffers the main loop has
// finished with), and read back with a lane per (row, 8-channel group): 32 contiguous bytes of fp32 in, scale / shift from LDS,
// 32 contiguous bytes of residual in, 32 contiguous bytes of hi | lo out - four lanes cover a 128-byte line.
// Caller contract: every wave of the workgroup is past the main loop's last barrier (no LDS fragment reads outstanding), and a
// workgroup barrier separates this call from the next write into the tile buffers.
constexpr int STG_ROW_BYTES = 144, STG_WAVE_BYTES = 32 * STG_ROW_BYTES;

// row_off(lr): byte offset of row lr of the tile in the output (and residual) image, channel 0 of this launch's block;
// ~size_t(0) when the row does not exist.  sc_s / sh_s: scale and shift of the workgroup's BC channels (LDS).
template <class T, class M, class RowOff>
__device__ __forceinline__ void store_tile_pair16(const f32x16 (&acc)[T::CT][T::PT], unsigned char *smem_bytes, const float *sc_s,
                                                  const float *sh_s, int n0, int cout, bool relu, const unsigned char *residual,
                                                  unsigned char *out, int wp, int wc, int lane, int wid, RowOff &&row_off) {
    static_assert((T::THREADS / 64) * STG_WAVE_BYTES <= T::LDS_BYTES, "staging windows must fit in the tile buffers");
    unsigned char *const stg = smem_bytes + wid * STG_WAVE_BYTES;
    const int h = lane >> 5, g = lane & 3;
#pragma unroll
    for (int pt = 0; pt < T::PT; ++pt) {
#pragma unroll
        for (int ct = 0; ct < T::CT; ++ct) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = make_float4(acc[ct][pt][4 * j], acc[ct][pt][4 * j + 1], acc[ct][pt][4 * j + 2], acc[ct][pt][4 * j + 3]);
                *reinterpret_cast<float4 *>(stg + (lane & 31) * STG_ROW_BYTES + j * 32 + h * 16) = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int lc = wc * T::CT * 32 + ct * 32 + g * 8;                 // my 8-channel group inside the BC tile
            const int col = n0 + lc;
            const float4 sc0 = *reinterpret_cast<const float4 *>(sc_s + lc), sc1 = *reinterpret_cast<const float4 *>(sc_s + lc + 4);
            const float4 sh0 = *reinterpret_cast<const float4 *>(sh_s + lc), sh1 = *reinterpret_cast<const float4 *>(sh_s + lc + 4);
            size_t off[2];
            float4 va[2], vb[2];
            uint4 rh[2], rl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {                                     // all loads of both items before the first store
                const int r = (lane >> 2) + 16 * i;
                off[i] = row_off(wp * T::PT * 32 + pt * 32 + r);
                if (col >= cout) off[i] = ~size_t(0);
                va[i] = *reinterpret_cast<const float4 *>(stg + r * STG_ROW_BYTES + g * 32);
                vb[i] = *reinterpret_cast<const float4 *>(stg + r * STG_ROW_BYTES + g * 32 + 16);
                rh[i] = rl[i] = make_uint4(0u, 0u, 0u, 0u);
                if (residual && off[i] != ~size_t(0)) {
                    const unsigned char *rp = residual + off[i] + (size_t)col * 4;
                    rh[i] = *reinterpret_cast<const uint4 *>(rp);
                    rl[i] = *reinterpret_cast<const uint4 *>(rp + 16);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[8] = {fmaf(va[i].x, sc0.x, sh0.x), fmaf(va[i].y, sc0.y, sh0.y), fmaf(va[i].z, sc0.z, sh0.z), fmaf(va[i].w, sc0.w, sh0.w),
                              fmaf(vb[i].x, sc1.x, sh1.x), fmaf(vb[i].y, sc1.y, sh1.y), fmaf(vb[i].z, sc1.z, sh1.z), fmaf(vb[i].w, sc1.w, sh1.w)};
                if (residual) {
                    const unsigned int hw[4] = {rh[i].x, rh[i].y, rh[i].z, rh[i].w}, lw[4] = {rl[i].x, rl[i].y, rl[i].z, rl[i].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        v[2 * k] += M::join(hw[k] & 0xFFFFu, lw[k] & 0xFFFFu);
                        v[2 * k + 1] += M::join(hw[k] >> 16, lw[k] >> 16);
                    }
                }
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                const float v0[4] = {v[0], v[1], v[2], v[3]}, v1[4] = {v[4], v[5], v[6], v[7]};
                uint2 h0, l0, h1, l1;
                split4<M>(v0, h0, l0);
                split4<M>(v1, h1, l1);
                if (off[i] != ~size_t(0)) {
                    unsigned char *gp = out + off[i] + (size_t)col * 4;
                    *reinterpret_cast<uint4 *>(gp) = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    *reinterpret_cast<uint4 *>(gp + 16) = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
}

// Schedule of gemm_pipeline (igemm.h) with a DEEPER global-load pipeline: at 16-bit MFMA rates a chunk lasts only
// a few hundred cycles per wave, less than an L2 / HBM round trip, so the loads run NS chunks ahead in NS
// register stages (the loop is unrolled NS times so that every stage index is static):
//   per chunk c, phase 1 = MFMAs of k-step 0 || LDS reads of k-step 1 || LDS writes of chunk c+1 (loaded NS chunks ago)
//   barrier
//   phase 2 = MFMAs of k-step 1 || LDS reads of chunk c+1 || global loads of chunk c+1+NS into the stage just freed
// `issue(stage)` loads the chunk the caller's iterator points at, `advance()` moves the iterator (chunk order).
// `issue(stage, slot)` loads the chunk the caller's iterator points at (slot = std::integral_constant<int, chunk % NS>, the
// static index of `stage`); it may issue EXTRA further loads AFTER the stage's NST ones in every call (the sparse conv's
// neighbour-list prefetch) - the vmcnt bookkeeping below accounts for them.
// DIAG (development, -DDZ_SPCONV_DIAG builds only; results are garbage for 1-4): 1 = no MFMAs, 2 = no LDS fragment
// reads, 3 = no LDS stage stores, 4 = no global loads - what is left tells which resource a kernel's time goes to.
template <class T>
__device__ __forceinline__ void keep_hfrag(HFrag<T> &f) {
#pragma unroll
    for (int i = 0; i < T::PT; ++i) asm volatile("" ::"v"(f.p_hi[i]), "v"(f.p_lo[i]));
#pragma unroll
    for (int i = 0; i < T::CT; ++i) asm volatile("" ::"v"(f.c_hi[i]), "v"(f.c_lo[i]));
}

template <class T, class M, int NS, int EXTRA = 0, int DIAG = 0, bool FREE2 = false, class Issue, class Advance>
__device__ __forceinline__ void hgemm_pipeline(int nchunks, v4u *__restrict__ smem, Issue &&issue, Advance &&advance,
                                               f32x16 (&acc)[T::CT][T::PT], int wp, int wc, int lane, int tid) {
    v4u *const Ps0 = smem, *const Cs0 = smem + 2 * T::PS_U4;          // [2][PS], [2][CS]
    constexpr int Q = T::KC / 16;
    const int poff = (wp * T::PT * 32 + (lane & 31)) * T::ROW_U4 + (lane >> 5) * 2;
    const int coff = (wc * T::CT * 32 + (lane & 31)) * T::ROW_U4 + (lane >> 5) * 2;
    constexpr int NST = T::P_PER_THREAD + T::C_PER_THREAD;          // staging loads / stores per chunk

    static_assert(NS >= 1 && NS <= 4, "1..4 register stages");
    HStage<T> st[NS];
    auto do_issue = [&](auto &stage, auto slot) {
        if constexpr (DIAG != 4) issue(stage, slot);
    };
    auto do_store = [&](auto &stage, v4u *ps, v4u *cs) {
        if constexpr (DIAG != 3) store_hstage<T>(stage, ps, cs, tid);
    };
    auto do_frag = [&](HFrag<T> &f, const v4u *ps, const v4u *cs, int q) {
        if constexpr (DIAG != 2) load_hfrag<T>(f, ps, cs, q);
    };
    auto do_mma = [&](HFrag<T> &f) {
        if constexpr (DIAG == 1) keep_hfrag<T>(f); else mma_hfrag<T, M, DIAG == 10>(f, acc);
    };
    do_issue(st[0], std::integral_constant<int, 0>{});
    if constexpr (DIAG != 4) wait_hstage<T, 0>(st[0]);
    do_store(st[0], Ps0, Cs0);
    __syncthreads();
    // the steady-state loop below is entered only through the branch with UNCONDITIONAL prefetches, so that the
    // compiler's s_waitcnt vmcnt(N) bookkeeping knows exactly NS stages are in flight at the loop header (with
    // conditional prefetches it falls back to vmcnt(0) there, which serialises the pipeline again)
    const bool steady = nchunks > 2 * NS;
    auto issue_at = [&](auto j_t) {
        constexpr int S = decltype(j_t)::value % NS;
        advance();
        do_issue(st[S], std::integral_constant<int, S>{});
    };
    if (steady) {
        issue_at(std::integral_constant<int, 1>{});
        if (NS >= 2) issue_at(std::integral_constant<int, 2>{});
        if (NS >= 3) issue_at(std::integral_constant<int, 3>{});
        if (NS >= 4) issue_at(std::integral_constant<int, 4>{});
    } else {
        if (1 < nchunks) issue_at(std::integral_constant<int, 1>{});
        if (NS >= 2 && 2 < nchunks) issue_at(std::integral_constant<int, 2>{});
        if (NS >= 3 && 3 < nchunks) issue_at(std::integral_constant<int, 3>{});
        if (NS >= 4 && 4 < nchunks) issue_at(std::integral_constant<int, 4>{});
    }
    HFrag<T> f0, f1;
    do_frag(f0, Ps0 + poff, Cs0 + coff, 0);
    // chunk c with its stage index S = (c + 1) % NS static; has1: chunk c+1 exists, has2: chunk c+1+NS exists
    auto body = [&](int c, auto s_t, auto static_t, bool has1, bool has2) {
        constexpr int S = decltype(s_t)::value;
        constexpr bool ALL = decltype(static_t)::value;              // both known true at compile time: no branches
        const int cur = c & 1;
        const v4u *Pc = Ps0 + cur * T::PS_U4 + poff, *Cc = Cs0 + cur * T::CS_U4 + coff;
        const v4u *Pn = Ps0 + (cur ^ 1) * T::PS_U4 + poff, *Cn = Cs0 + (cur ^ 1) * T::CS_U4 + coff;
        // ---- phase 1: memory instructions first, MFMAs behind them (sched_barrier pins the order: without it the
        // scheduler hoists the MFMAs above the staging and the LDS-write latency lands in front of the barrier)
        if (Q == 2) do_frag(f1, Pc, Cc, 1);
        if (ALL) {
            if constexpr (DIAG != 4) wait_hstage<T, (NS - 1) * (NST + EXTRA) + EXTRA>(st[S]);     // steady state: the NS-1 younger stages stay in flight
            do_store(st[S], Ps0 + (cur ^ 1) * T::PS_U4, Cs0 + (cur ^ 1) * T::CS_U4);
        } else if (has1) {
            if constexpr (DIAG != 4) wait_hstage<T, 0>(st[S]);                  // tail: fewer stages may be in flight - drain
            do_store(st[S], Ps0 + (cur ^ 1) * T::PS_U4, Cs0 + (cur ^ 1) * T::CS_U4);
        }
        __builtin_amdgcn_sched_barrier(0);
        do_mma(f0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        // ---- phase 2
        if (ALL || has1) do_frag(f0, Pn, Cn, 0);
        if (ALL || has2) { advance(); do_issue(st[S], std::integral_constant<int, S>{}); }
        if (Q == 2) {
            // phase 2 order: pinned (loads, then MFMAs) or left to the compiler's scheduler (DIAG 6 / FREE2: measured
            // 3-5 % faster for the 8-wave sparse tiles, where it interleaves the address math with the MFMAs)
            if constexpr (DIAG != 6 && !FREE2) __builtin_amdgcn_sched_barrier(0);
            do_mma(f1);
            if constexpr (DIAG != 6 && !FREE2) __builtin_amdgcn_sched_barrier(0);
        }
    };
    using TT = std::integral_constant<bool, true>;
    using FF = std::integral_constant<bool, false>;
    int c = 0;
    // steady state: NS chunks per trip, every predicate true
    for (; c + 2 * NS < nchunks; c += NS) {
        body(c, std::integral_constant<int, 1 % NS>{}, TT{}, true, true);
        if (NS >= 2) body(c + 1, std::integral_constant<int, 2 % NS>{}, TT{}, true, true);
        if (NS >= 3) body(c + 2, std::integral_constant<int, 3 % NS>{}, TT{}, true, true);
        if (NS >= 4) body(c + 3, std::integral_constant<int, 4 % NS>{}, TT{}, true, true);
    }
    // tail (c is a multiple of NS): same stage rotation, run-time predicates
    for (; c < nchunks; c += NS) {
        body(c, std::integral_constant<int, 1 % NS>{}, FF{}, c + 1 < nchunks, c + 1 + NS < nchunks);
        if (NS >= 2 && c + 1 < nchunks) body(c + 1, std::integral_constant<int, 2 % NS>{}, FF{}, c + 2 < nchunks, c + 2 + NS < nchunks);
        if (NS >= 3 && c + 2 < nchunks) body(c + 2, std::integral_constant<int, 3 % NS>{}, FF{}, c + 3 < nchunks, c + 3 + NS < nchunks);
        if (NS >= 4 && c + 3 < nchunks) body(c + 3, std::integral_constant<int, 4 % NS>{}, FF{}, c + 4 < nchunks, c + 4 + NS < nchunks);
    }
}

}  // namespace dz
