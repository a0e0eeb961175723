// Sparse 3-D convolution for the 64 / 128-channel levels with the operand tiles written by the memory pipeline itself
// (`buffer_load_dwordx4 ... lds`), gfx950.  Same tile (128 output rows x BC output channels, 8 waves of 32 x BC/2), same chunk order
// (32-channel chunk outermost, taps of the tile innermost), same arithmetic and results as k_spconv_h (sparse_conv_h.hip) - what
// changes is how a (tap, chunk) pair of operand tiles reaches LDS:
//   k_spconv_h: global -> registers (NS stages in flight) -> ds_write_b128 -> LDS.  The r03 diag breakdown of the 128-channel kernel
//     puts 17 % of its time on those stores (13 cycles of the VGPR -> LDS path per wave instruction, 32 of them per chunk) and
//     31 % on the fragment reads;
//   here: the gathered rows and the weight slice land in a ring of four LDS stages straight from L2 (three chunks in flight, `vmcnt`
//     counted by hand: every chunk is exactly 2 + BC/64 loads per thread, absent neighbours use the out-of-range offset and arrive
//     as zeros), unpadded 128-byte rows with the 16-byte piece p of row n stored at p ^ ((n >> 1) & 7) (conflict-free for the
//     16-lane groups ds_read_b128 is served in); the staging registers are gone, the tile's slice of the neighbour table sits in LDS
//     (one 4-byte direct load per (tap, row)) instead of a register ring.
// Reference call sites: detection/detzero_det/models/centerpoint_modules/backbone3d.py:261-280 (conv3, conv4, conv_out).
#pragma once
#include "sparse_conv_w.h"

namespace dz {

template <int BC_>
struct DTile {              // the members store_tile_pair16 reads
    static constexpr int BP = 128, BC = BC_, KC = 32, WP = 4, WC = 2, THREADS = 512, PT = 1, CT = BC / 64;
    static constexpr int A_BYTES = BP * 128, B_BYTES = BC * 128, STAGE = A_BYTES + B_BYTES, NB = 4;
    static constexpr int OFF_TAB = NB * STAGE, TAB_BYTES = 7 * THREADS * 4;            // (27 x 128 entries, loaded as 7 x 512)
    // scale / shift live in the dynamic allocation too: a static __shared__ array would sit in front of it and shift every LDS address
    // the direct loads are given (they address LDS from offset 0 of the workgroup's allocation)
    static constexpr int OFF_SS = OFF_TAB + TAB_BYTES, LDS_BYTES = OFF_SS + 2 * BC * 4;
    static constexpr int AL = BP * 8 / THREADS, BL = BC * 8 / THREADS, L = AL + BL;        // loads per thread and chunk
    static_assert(LDS_BYTES + 4096 <= 160 * 1024, "LDS");
};

__device__ __forceinline__ void d_load16_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory", "m0");
}
__device__ __forceinline__ void d_load4_lds(unsigned int lds_base, unsigned int voff, srsrc_t rsrc, unsigned int soff) {
    const unsigned int b = __builtin_amdgcn_readfirstlane(lds_base), so = __builtin_amdgcn_readfirstlane(soff);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dword %1, %2, %3 offen lds" ::"s"(b), "v"(voff), "s"(rsrc), "s"(so) : "memory", "m0");
}

template <int BC, class M>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_spconv_d(SpConvHArgs a) {
    using T = DTile<BC>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *const sc_s = reinterpret_cast<float *>(smem_raw + T::OFF_SS), *const sh_s = sc_s + BC;
    int *const tab = reinterpret_cast<int *>(smem_raw + T::OFF_TAB);               // [27][128]
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wp = wid / T::WC, wc = wid % T::WC, l31 = lane & 31, h = lane >> 5;
    const int m = min(*a.d_m_out, a.cap);
    const int ntiles = (m + T::BP - 1) / T::BP;
    const int kchunks = a.cin / T::KC;
    const int n0 = blockIdx.y * BC;
    const srsrc_t prsrc = make_srsrc(a.in, a.in_bytes), crsrc = make_srsrc(a.w, a.w_bytes), nrsrc = make_srsrc(a.nbr, a.nbr_bytes);
    const unsigned int row_bytes = (unsigned int)a.cin * 4u, tap_bytes = (unsigned int)(a.cout_pad * a.cin * 4);
    for (int c = tid; c < BC; c += T::THREADS) {
        const bool in = n0 + c < a.cout;
        sc_s[c] = (in && a.scale) ? a.scale[n0 + c] : 1.f;
        sh_s[c] = (in && a.shift) ? a.shift[n0 + c] : 0.f;
    }
    // staging geometry of a thread: unit u = i * 512 + tid = 16 bytes at offset u * 16 of a tile = row u >> 3, stored piece u & 7
    unsigned int a_sw[T::AL], b_voff[T::BL];
    int a_row[T::AL];
#pragma unroll
    for (int i = 0; i < T::AL; ++i) {
        const int u = i * T::THREADS + tid, r = u >> 3;
        a_row[i] = r;
        a_sw[i] = (unsigned int)(((u & 7) ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < T::BL; ++i) {
        const int u = i * T::THREADS + tid, n = u >> 3;
        b_voff[i] = (unsigned int)(n0 + n) * row_bytes + (unsigned int)(((u & 7) ^ ((n >> 1) & 7)) << 4);
    }
    // fragment addresses of a lane inside a stage: rows (block of 32) + l31, pieces 4 q + 2 h (+ 1)
    const int sw = (l31 >> 1) & 7;
    unsigned int fo[2][2];                              // [q][hi / lo]: swizzled piece offsets
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl) fo[q][hl] = (unsigned int)((((q * 4 + h * 2 + hl) ^ sw) << 4) + (l31 << 7));
    __syncthreads();

    constexpr int XRUN = 16;                            // the XCD-aware tile deal of k_spconv_h
    const int xcd = blockIdx.x & 7;
    for (int t = blockIdx.x >> 3;; t += gridDim.x >> 3) {
        const int tile = ((t / XRUN) * 8 + xcd) * XRUN + t % XRUN;
        if ((t / XRUN) * 8 * XRUN >= ntiles) break;
        if (tile >= ntiles) continue;
        const int row0 = tile * T::BP;
        unsigned int taps = 0u;
#pragma unroll
        for (int i = 0; i < T::BP / 32; ++i) taps |= a.tile_masks[tile * (T::BP / 32) + i];
        taps = __builtin_amdgcn_readfirstlane(taps);
        f32x16 acc[T::CT][T::PT];
#pragma unroll
        for (int i = 0; i < T::CT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][0][e] = 0.f;
        const int ntap = __popc(taps), nchunks = ntap * kchunks;
        if (nchunks > 0) {
            // ---- the tile's slice of the neighbour table -> LDS: tab[tap][row] for every tap (27 x 128 entries, 7 loads per thread);
            // rows past the end read as 0 (their results are never stored)
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int u = i * T::THREADS + tid, tp = u >> 7, r = u & 127;
                const bool ok = tp < a.kvol && row0 + r < m;
                d_load4_lds((unsigned int)(T::OFF_TAB + (i * T::THREADS + wid * 64) * 4), ok ? (unsigned int)(row0 + r) * 4u : OOB_OFFSET, nrsrc,
                            (unsigned int)(tp < a.kvol ? tp : 0) * (unsigned int)a.cap * 4u);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // chunk iterators: (rem_i, tap_i, kc_i) the chunk issued next, (rem, ...) implicit in the consume order (same sequence)
            unsigned int rem_i = taps;
            int tap_i = __ffs((int)rem_i) - 1, kc_i = 0;
            auto issue = [&](int buf) {
                const unsigned int abase = (unsigned int)(buf * T::STAGE), bbase = abase + T::A_BYTES;
#pragma unroll
                for (int i = 0; i < T::AL; ++i) {
                    const int nb = tab[tap_i * T::BP + a_row[i]];
                    d_load16_lds(abase + (unsigned int)((i * T::THREADS + wid * 64) * 16), nb >= 0 ? (unsigned int)nb * row_bytes + a_sw[i] : OOB_OFFSET, prsrc,
                                 (unsigned int)(kc_i * 128));
                }
#pragma unroll
                for (int i = 0; i < T::BL; ++i)
                    d_load16_lds(bbase + (unsigned int)((i * T::THREADS + wid * 64) * 16), b_voff[i], crsrc, (unsigned int)tap_i * tap_bytes + (unsigned int)(kc_i * 128));
                rem_i &= rem_i - 1;
                if (rem_i == 0u) { rem_i = taps; ++kc_i; }
                tap_i = __ffs((int)rem_i) - 1;
            };
            constexpr int D = T::NB - 1;                // chunks in flight
#pragma unroll
            for (int d = 0; d < D; ++d)
                if (d < nchunks) issue(d);
            for (int c = 0; c < nchunks; ++c) {
                // chunk c has landed when at most the loads of the chunks issued after it are outstanding
                const int younger = min(D - 1, nchunks - 1 - c);
                if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * T::L) : "memory");
                else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(T::L) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();                        // ... in every wave's view; and every wave is done with chunk c - 1: its stage is free
                if (c + D < nchunks) issue((c + D) & (T::NB - 1));
                const unsigned char *ab = smem_raw + (c & (T::NB - 1)) * T::STAGE + ((wp * 32) << 7);
                const unsigned char *bb = smem_raw + (c & (T::NB - 1)) * T::STAGE + T::A_BYTES + ((wc * T::CT * 32) << 7);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const v4u p_hi = *reinterpret_cast<const v4u *>(ab + fo[q][0]), p_lo = *reinterpret_cast<const v4u *>(ab + fo[q][1]);
#pragma unroll
                    for (int ct = 0; ct < T::CT; ++ct) {
                        const v4u c_hi = *reinterpret_cast<const v4u *>(bb + ((ct * 32) << 7) + fo[q][0]);
                        const v4u c_lo = *reinterpret_cast<const v4u *>(bb + ((ct * 32) << 7) + fo[q][1]);
                        if constexpr (M::TERMS != 1) {
                            acc[ct][0] = M::mma(c_lo, p_hi, acc[ct][0]);
                            acc[ct][0] = M::mma(c_hi, p_lo, acc[ct][0]);
                        }
                        acc[ct][0] = M::mma(c_hi, p_hi, acc[ct][0]);
                    }
                }
            }
            __syncthreads();                            // every wave is past its last fragment read: the stages become epilogue windows
        }
        store_tile_pair16<T, M>(acc, smem_raw, sc_s, sh_s, n0, a.cout, a.relu != 0, reinterpret_cast<const unsigned char *>(a.residual),
                                reinterpret_cast<unsigned char *>(a.out), wp, wc, lane, wid, [&](int lr) {
                                    const int row = row0 + lr;
                                    return row < m ? (size_t)row * a.cout * 4 : ~size_t(0);
                                });
        __syncthreads();
    }
}

template <int BC, class M>
static int launch_spconv_d(const SpConvHArgs &a, hipStream_t stream) {
    using T = DTile<BC>;
    static PerDeviceFlags done;
    if (int rc = reserve_lds(reinterpret_cast<const void *>(&k_spconv_d<BC, M>), T::LDS_BYTES, done, "dz_spconv_forward_split")) return rc;
    int grid = ceil_div(a.cap, T::BP);
    if (grid > 2048) grid = 2048;
    grid = (grid + 7) & ~7;            // a multiple of 8: the XCD schedule in the kernel
    if (grid < 8) grid = 8;
    hipLaunchKernelGGL((k_spconv_d<BC, M>), dim3(grid, a.cout_pad / BC), dim3(T::THREADS), T::LDS_BYTES, stream, a);
    DZ_LAUNCH_CHECK();
    return DZ_OK;
}

}  // namespace dz
