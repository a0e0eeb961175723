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
__device__ __forceinline__ void mma_chunk(const float *__restrict__ As, const float *__restrict__ Bs,
                                          f32x4 (&acc)[T::MT][T::NT], int wm, int wn, int lane) {
    const int r = lane & 15, g = lane >> 4;
    const float *ap = As + (wm * T::MT * 16 + r) * T::LDA + g * 4;
    const float *bp = Bs + (g * 4) * T::LDB + wn * T::NT * 16 + r;
    constexpr int Q = T::KC / 16;
    Frag<T> f[2];
    load_frag<T>(f[0], ap, bp, 0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (q + 1 < Q) load_frag<T>(f[(q + 1) & 1], ap, bp, q + 1);
        // keep the prefetch above the MFMAs (hipcc otherwise sinks every ds_read next to its first use
        // and waits lgkmcnt(0) in front of each group of MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        const Frag<T> &c = f[q & 1];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int mt = 0; mt < T::MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < T::NT; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.a[mt][e], c.b[e][nt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
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

// A chunk: row r of the tile comes from element offset rowoff[r] (or is zero when rowoff[r] < 0);
// `rowoff` lives in LDS and already includes tap / channel-chunk offsets via `add`.
template <class T>
__device__ __forceinline__ void load_a(Stage<T> &st, const float *__restrict__ in, const int *rowbase, long scale,
                                       long add, int tid) {
#pragma unroll
    for (int i = 0; i < T::A_PER_THREAD; ++i) {
        const int idx = tid + i * T::THREADS;
        if (T::A_F4 % T::THREADS == 0 || idx < T::A_F4) {
            const int r = idx / (T::KC / 4), q = idx % (T::KC / 4);
            const int rb = rowbase[r];
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rb >= 0) v = *reinterpret_cast<const float4 *>(in + (long)rb * scale + add + q * 4);
            st.a[i] = v;
        }
    }
}

// A chunk from per-thread row pointers computed once per tile (nullptr = zero row): no LDS lookup and
// no 64-bit multiply in the chunk loop.
template <class T>
__device__ __forceinline__ void load_a_ptr(Stage<T> &st, const float *const (&rowp)[T::A_PER_THREAD], long add) {
#pragma unroll
    for (int i = 0; i < T::A_PER_THREAD; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowp[i] != nullptr) v = *reinterpret_cast<const float4 *>(rowp[i] + add);
        st.a[i] = v;
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

}  // namespace dz
