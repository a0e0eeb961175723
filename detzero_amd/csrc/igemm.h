// fp32 implicit-GEMM tile engine shared by the sparse 3-D convolution and the dense BEV
// convolution (gfx950, wave64).
//
//   C[BM x BN] += A[BM x K] . B[K x BN],   K walked in chunks of KC channels of one kernel tap.
//
// Matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32: bitwise a k-ordered fmaf chain, 157 TF/s chip
// peak, 32 cycles/issue per SIMD).  A workgroup is 4 waves laid out WM x WN over the tile; a wave
// owns MT x NT accumulator fragments of 16x16.  Operand fetch per 16-deep k slice: one
// ds_read_b128 per A fragment (4 consecutive k of the lane's row) and one ds_read_b32 per
// (k, B fragment); at 32 cycles per MFMA the LDS is <5 % busy, so plain padded layouts suffice:
//   As[BM][KC+4]  (row-major, 16-B aligned rows; read 16 B per lane)
//   Bs[KC][BN+4]  (k-major; lanes of a 16-lane group read 16 consecutive columns)
// Within an MFMA lane l supplies k-slot g = l>>4; we map slot g, step e of slice q to channel
// kappa = 16q + 4g + e on BOTH operands (any bijection of k is a valid GEMM), which is what lets
// A come from a single 16-byte LDS read.
//
// Global->LDS staging is register-staged, software pipelined and DOUBLE-BUFFERED in LDS: while the
// MFMAs of chunk i read buffer i&1, the loads of chunk i+1 (issued one chunk earlier) land in
// registers and are written to buffer (i+1)&1, so there is ONE barrier per chunk and HBM/L2 latency
// hides behind 32-128 MFMAs per wave even at 1-2 workgroups per CU.
#pragma once
#include <type_traits>

#include "common.h"

namespace dz {

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int BM_, int BN_, int KC_, int WM_, int WN_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, KC = KC_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 256;
    static constexpr int MT = BM / (16 * WM);
    static constexpr int NT = BN / (16 * WN);
    static constexpr int LDA = KC + 4;
    static constexpr int LDB = BN + 4;
    static constexpr int A_F4 = BM * (KC / 4);            // float4 elements of the A chunk
    static constexpr int B_F4 = KC * (BN / 4);            // float4 elements of the B chunk
    static constexpr int A_PER_THREAD = (A_F4 + THREADS - 1) / THREADS;
    static constexpr int B_PER_THREAD = (B_F4 + THREADS - 1) / THREADS;
    static constexpr int AS_FLOATS = BM * LDA;            // one buffer
    static constexpr int BS_FLOATS = KC * LDB;
    static constexpr int LDS_FLOATS = 2 * (AS_FLOATS + BS_FLOATS);   // double buffered
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM % (16 * WM) == 0 && BN % (16 * WN) == 0, "tile must split into 16x16 fragments");
    static_assert(KC % 16 == 0, "KC must be a multiple of the 16-deep k slice");
};

// MFMAs of one staged chunk.  Operands of k-slice q+1 are fetched from LDS before the MFMAs of slice q
// are issued (explicit register double buffer), so the matrix pipe never waits on an LDS round trip.
template <class T>
struct Frag {
    f32x4 a[T::MT];
    float b[4][T::NT];
};

template <class T>
__device__ __forceinline__ void load_frag(Frag<T> &f, const float *__restrict__ ap, const float *__restrict__ bp, int q) {
#pragma unroll
    for (int mt = 0; mt < T::MT; ++mt) f.a[mt] = *reinterpret_cast<const f32x4 *>(ap + mt * 16 * T::LDA + q * 16);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int nt = 0; nt < T::NT; ++nt) f.b[e][nt] = bp[(q * 16 + e) * T::LDB + nt * 16];
}

template <class T>
__device__ __forceinline__ void mma_frag(const Frag<T> &c, f32x4 (&acc)[T::MT][T::NT]) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < T::NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[mt][e], c.b[e][nt], acc[mt][nt], 0, 0, 0);
}

// scheduling hint: interleave `n` groups of (mfma_per MFMAs, one instruction of class `mask`)
// masks (LLVM sched_group_barrier): 0x008 MFMA, 0x020 VMEM read, 0x100 DS read, 0x200 DS write
template <int MASK, int N, int MFMA_PER>
__device__ __forceinline__ void interleave_hint() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, MFMA_PER, 0);
        __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
    }
}

// Register staging buffers for one chunk.
template <class T>
struct Stage {
    float4 a[T::A_PER_THREAD];
    float4 b[T::B_PER_THREAD];
};

// B chunk: rows [k0, k0+KC) of a (K x ldw) weight slice, columns [n0, n0+BN)
template <class T>
__device__ __forceinline__ void load_b(Stage<T> &st, const float *__restrict__ w, int ldw, int n0, int tid) {
#pragma unroll
    for (int i = 0; i < T::B_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        if (T::B_F4 % T::THREADS == 0 || idx < T::B_F4) {
            const int kk = idx / (T::BN / 4), q = idx % (T::BN / 4);
            st.b[i] = *reinterpret_cast<const float4 *>(w + (size_t)kk * ldw + n0 + q * 4);
        }
    }
}

// A operand rows are fetched with raw buffer loads: an offset past `num_records` returns zeros, so rows
// without a neighbour (sparse conv) or past the end of the image need no branch - the chunk loop stays one
// basic block, which is what lets the scheduler interleave staging with the MFMAs.
using v4u = __attribute__((ext_vector_type(4))) unsigned int;
constexpr unsigned int OOB_OFFSET = 0x80000000u;       // + any chunk offset < 2^31 stays out of range (buffers < 2 GiB)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}

// byte offsets voff[i] (row start + 16*q, or OOB_OFFSET) computed by the caller; `add` = chunk byte offset
template <class T>
__device__ __forceinline__ void load_a_buf(Stage<T> &st, __amdgpu_buffer_rsrc_t rsrc, const unsigned int (&voff)[T::A_PER_THREAD],
                                           unsigned int add) {
#pragma unroll
    for (int i = 0; i < T::A_PER_THREAD; ++i) {
        const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i] + add, 0, 0);
        st.a[i] = make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
    }
}

template <class T>
__device__ __forceinline__ void store_stage(const Stage<T> &st, float *__restrict__ As, float *__restrict__ Bs, int tid) {
#pragma unroll
    for (int i = 0; i < T::A_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        if (T::A_F4 % T::THREADS == 0 || idx < T::A_F4) {
            const int r = idx / (T::KC / 4), q = idx % (T::KC / 4);
            *reinterpret_cast<float4 *>(&As[r * T::LDA + q * 4]) = st.a[i];
        }
    }
#pragma unroll
    for (int i = 0; i < T::B_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        if (T::B_F4 % T::THREADS == 0 || idx < T::B_F4) {
            const int kk = idx / (T::BN / 4), q = idx % (T::BN / 4);
            *reinterpret_cast<float4 *>(&Bs[kk * T::LDB + q * 4]) = st.b[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The chunk pipeline.  `issue()` loads the chunk the caller's iterator points at into `st` (global ->
// registers), `advance()` moves the iterator.  Per chunk c (two 16-deep k slices q0, q1 when KC = 32):
//
//   P1:  MFMAs(q0 of c)  ||  LDS reads of q1 of c  ||  wait for chunk c+1's global loads, write them to
//                                                      the other LDS buffer
//   --- barrier ---
//   P2:  MFMAs(q1 of c)  ||  LDS reads of q0 of c+1 (other buffer)  ||  issue global loads of chunk c+2
//
// One barrier per chunk; every LDS / global latency of a wave sits in the shadow of its own MFMAs, so the
// matrix pipe stays fed even when the co-resident waves of a SIMD run in lock step.
// Hazards: the buffer written in P1 of c was last read by fragment loads issued before the barrier of
// chunk c-1 (LDS services requests in arrival order); it is read again only after the barrier of c.
// ------------------------------------------------------------------------------------------------
template <class T, class Issue, class Advance>
__device__ __forceinline__ void gemm_pipeline(int nchunks, float *__restrict__ smem, Stage<T> &st, Issue &&issue,
                                              Advance &&advance, f32x4 (&acc)[T::MT][T::NT], int wm, int wn, int lane,
                                              int tid) {
    float *const As0 = smem, *const Bs0 = smem + 2 * T::AS_FLOATS;     // [2][AS], [2][BS]
    constexpr int Q = T::KC / 16;
    static_assert(Q == 1 || Q == 2, "KC must be 16 or 32");
    const int r = lane & 15, g = lane >> 4;
    const int aoff = (wm * T::MT * 16 + r) * T::LDA + g * 4;
    const int boff = (g * 4) * T::LDB + wn * T::NT * 16 + r;
    constexpr int MFMA_N = 4 * T::MT * T::NT;                     // MFMAs per k slice
    constexpr int NLD = T::MT + 2 * T::NT;                        // ~LDS read instructions per slice (b reads pair up)
    constexpr int NST = T::A_PER_THREAD + T::B_PER_THREAD;        // staging loads / stores per chunk
    constexpr int REST1 = MFMA_N - (Q == 2 ? NLD : 0);
    constexpr int PER1 = (REST1 / NST) > 0 ? (REST1 / NST) : 1;
    constexpr int REST2 = MFMA_N - NLD;
    constexpr int PER2 = (REST2 / NST) > 0 ? (REST2 / NST) : 1;

    issue();
    store_stage<T>(st, As0, Bs0, tid);
    __syncthreads();
    if (nchunks > 1) { advance(); issue(); }
    Frag<T> f0, f1;
    load_frag<T>(f0, As0 + aoff, Bs0 + boff, 0);
    // the loop body, specialised at compile time on "a chunk c+1 / c+2 exists" so that it has no branches
    auto body = [&](int c, auto has1_t, auto has2_t) {
        constexpr bool HAS1 = decltype(has1_t)::value, HAS2 = decltype(has2_t)::value;
        const int cur = c & 1;
        const float *Ac = As0 + cur * T::AS_FLOATS + aoff, *Bc = Bs0 + cur * T::BS_FLOATS + boff;
        const float *An = As0 + (cur ^ 1) * T::AS_FLOATS + aoff, *Bn = Bs0 + (cur ^ 1) * T::BS_FLOATS + boff;
        // ---- P1
        if (Q == 2) load_frag<T>(f1, Ac, Bc, 1);
        if (HAS1) store_stage<T>(st, As0 + (cur ^ 1) * T::AS_FLOATS, Bs0 + (cur ^ 1) * T::BS_FLOATS, tid);
        mma_frag<T>(f0, acc);
        if (Q == 2) interleave_hint<0x100, NLD, 1>();
        if (HAS1) interleave_hint<0x200, NST, PER1>();
        __syncthreads();
        // ---- P2
        if (HAS1) load_frag<T>(f0, An, Bn, 0);
        if (HAS2) { advance(); issue(); }
        if (Q == 2) {
            mma_frag<T>(f1, acc);
            if (HAS1) interleave_hint<0x100, NLD, 1>();
            if (HAS2) interleave_hint<0x020, NST, PER2>();
        }
    };
    using TT = std::integral_constant<bool, true>;
    using FF = std::integral_constant<bool, false>;
    int c = 0;
    for (; c + 2 < nchunks; ++c) body(c, TT{}, TT{});
    if (c + 1 < nchunks) { body(c, TT{}, FF{}); ++c; }
    body(c, FF{}, FF{});
}

}  // namespace dz
