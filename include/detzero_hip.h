/*
 * libdetzero_hip — C ABI of the MI355X (gfx950) implementation of DetZero's per-frame detection
 * hot path.  Plain pointers and sizes only (no torch types); every pointer is a caller-owned
 * DEVICE pointer unless the name starts with `h_`; every call is asynchronous on `stream`
 * (a hipStream_t passed as void*), allocates nothing, never exits the process and returns
 * DZ_OK (0) or a negative error code (message via dz_last_error()).
 *
 * Counts that are only known on the device (number of voxels, of active sites, of kept boxes)
 * live in device int32 words (`d_*`); kernels read them, so a whole frame can be enqueued — or
 * captured in a hipGraph — without a host round trip.  `cap` arguments are the row capacities
 * of the caller's buffers (upper bounds).
 *
 * Each entry point names the reference interface it stands in for (paths relative to the
 * reference repository root).  spconv / cumm / torch_scatter are un-vendored pip dependencies
 * of the reference; their Python call sites are cited instead.
 */
#ifndef DETZERO_HIP_H
#define DETZERO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DZ_OK 0
#define DZ_ERR_INVALID (-1)     /* bad argument */
#define DZ_ERR_WORKSPACE (-2)   /* workspace too small */
#define DZ_ERR_HIP (-3)         /* HIP runtime error */
#define DZ_ERR_UNSUPPORTED (-4) /* shape not supported by this build */

const char *dz_version(void);
const char *dz_last_error(void);
/* number of compute units of the current device (used by callers to size persistent grids) */
int dz_device_cu_count(void);

/* ---------------------------------------------------------------------------------------------
 * Points -> voxels
 * ------------------------------------------------------------------------------------------- */

/* Hard voxelization == spconv.utils.Point2VoxelCPU3d(...).point_to_voxel(points)
 * as called from detection/detzero_det/datasets/processor/data_processor.py:69-83.
 *   points (n,c) f32; h_range6 = [x0,y0,z0,x1,y1,z1]; h_vsize3; h_grid3 = (gx,gy,gz)
 *   voxels (max_voxels,max_points,c) f32 zero padded; coords_zyx (max_voxels,3) i32;
 *   num_points (max_voxels) i32; d_num_voxels: device i32, number of voxels produced.
 * Voxel order = first appearance in the input, points per voxel = first max_points in input order.
 * xy_range_mask != 0 additionally applies DataProcessor.mask_points_and_boxes_outside_range
 * (data_processor.py:24-37: keep lo <= x,y <= hi, inclusive) so device-resident frames need no
 * separate compaction pass. */
size_t dz_voxelize_hard_workspace_bytes(int n, int gx, int gy, int gz, int max_points);
int dz_voxelize_hard(const float *points, int n, int c, const float *h_range6, const float *h_vsize3,
                     const int *h_grid3, int xy_range_mask, int max_points, int max_voxels, float *voxels,
                     int *coords_zyx, int *num_points, int *d_num_voxels, void *ws, size_t ws_bytes,
                     void *stream);
/* dz_voxelize_hard + MeanVFE (vfe.py:58-83) fused for batched pipelines: row r < *d_num_voxels of `feats`
 * (row stride c_stride >= c, extra channels zero) receives the mean of voxel r's first <= max_points points, and
 * coords_bzyx row r = [batch_index, z, y, x]; rows beyond the count are left untouched (callers pre-fill the
 * coordinate rows with -1 so that dz_index_from_coords ignores them).  Same workspace as dz_voxelize_hard. */
int dz_voxelize_hard_mean(const float *points, int n, int c, const float *h_range6, const float *h_vsize3,
                          const int *h_grid3, int xy_range_mask, int max_points, int max_voxels, int batch_index,
                          float *feats, int c_stride, int *coords_bzyx, int *d_num_voxels, void *ws, size_t ws_bytes,
                          void *stream);
/* the same for `batch` equally long frames stored back to back (points (batch*n_per_frame, c)) in ONE launch chain:
 * frame f's voxels (first-appearance order within the frame, cut at max_voxels) go to rows f*cap_per_frame + r of
 * feats / coords_bzyx with batch index f; d_num_voxels (batch) receives the per-frame counts. */
size_t dz_voxelize_hard_batched_workspace_bytes(int n_per_frame, int batch, int gx, int gy, int gz, int max_points);
int dz_voxelize_hard_mean_batched(const float *points, int n_per_frame, int batch, int c, const float *h_range6,
                                  const float *h_vsize3, const int *h_grid3, int xy_range_mask, int max_points,
                                  int max_voxels, float *feats, int c_stride, int *coords_bzyx, int cap_per_frame,
                                  int *d_num_voxels, void *ws, size_t ws_bytes, void *stream);
/* Voxelize a batch straight into the level-1 sparse index of the backbone (data_processor.py:61-91 + vfe.py:66-83 +
 * the SparseConvTensor construction of backbone3d.py:302-307 in one chain): fills the level's bitmap / prefix /
 * canonical coordinates / count (the arrays dz_index_from_coords would produce for the frames' voxels, level shape
 * (level_d, gy, gx)) and writes every voxel's mean to its canonical row of `feats` (cap, c_dst), channels >= c zero,
 * fp32 (math 0) or pair16.  Valid only when a frame cannot exceed max_voxels (n_per_frame <= max_voxels; otherwise
 * DZ_ERR_UNSUPPORTED: the first-appearance cut of the reference needs dz_voxelize_hard_mean_batched). */
size_t dz_voxelize_to_level_workspace_bytes(int n_per_frame, int batch, int max_points, int cap, int d, int h, int w, int layout);
int dz_voxelize_to_level(const float *points, int n_per_frame, int batch, int c, const float *h_range6,
                         const float *h_vsize3, const int *h_grid3, int xy_range_mask, int max_points, int max_voxels,
                         int level_d, int layout, uint32_t *bitmap, uint32_t *prefix, int *coords_out, int *d_m, int cap,
                         float *feats, int c_dst, int math, void *ws, size_t ws_bytes, void *stream);

/* MeanVFE.forward — detection/detzero_det/models/centerpoint_modules/vfe.py:66-83.
 * out (m, c_out_stride) f32: columns [0,c) = sum over slots / max(num_points,1); columns
 * [c, c_out_stride) are written as 0 (channel padding used by the sparse backbone). */
int dz_mean_vfe(const float *voxels, const int *num_points, const int *d_m, int cap, int max_points,
                int c, float *out, int c_out_stride, void *stream);

/* DynamicMeanVFE.forward — vfe.py:109-147 (torch.unique + torch_scatter.scatter_mean).
 *   points_b (n,1+c) f32 rows [b,x,y,z,...]; feats (cap,c) f32; coords_bzyx (cap,4) i32 in
 *   ascending merge-key order (key = b*gx*gy*gz + cx*gy*gz + cy*gz + cz). */
size_t dz_voxelize_dynamic_workspace_bytes(int n, int batch, int gx, int gy, int gz, int c, int cap);
int dz_voxelize_dynamic_mean(const float *points_b, int n, int c, const float *h_range6,
                             const float *h_vsize3, const int *h_grid3, int xy_range_mask, int batch, float *feats,
                             int *coords_bzyx, int *d_num_voxels, int cap, void *ws, size_t ws_bytes,
                             void *stream);

/* ---------------------------------------------------------------------------------------------
 * Sparse tensor index ("indice" machinery of spconv.pytorch.SparseConvTensor / SubMConv3d /
 * SparseConv3d — call sites detection/detzero_det/models/centerpoint_modules/backbone3d.py:
 * 243-280, 302-307).  A level is a bit per cell of the (B,D,H,W) grid plus an exclusive
 * popcount prefix per 32-bit word (dz_voxelize_to_level writes it only at words that hold at least one bit - the only ones
 * the rank query of an ACTIVE cell reads; its level-1 bitmap is > 97 % empty words); active sites are numbered in ascending cell
 * key, which is also their row in the feature matrix.  `layout` selects the key of cell (b, z, y, x):
 *   DZ_LAYOUT_LINEAR  ((b*D+z)*H+y)*W+x: rows in ascending (b, z, y, x) order - the canonical order of SURVEY App. C;
 *   DZ_LAYOUT_BRICK   the (y, x) plane cut into 8 x 8 columns through all of z, column-major:
 *                     ((((b*ceil(H/8) + y/8)*ceil(W/8) + x/8)*D + z) << 6) | (y%8 << 3) | x%8.  Rows of a spatial neighbourhood are
 *                     then neighbours in memory (the order the backbone keeps its levels in: dz_spconv_tiles_forward stages the
 *                     1.3-1.8x halo of a 128..512-row tile in LDS instead of gathering every (row, tap) pair from L2).  spconv's own
 *                     row order is a hash-table detail: parity is stated on rows sorted by the linear key.
 * Every function below that takes (b, d, h, w) takes the layout next to it; bitmap / prefix sizes depend on it.
 * ------------------------------------------------------------------------------------------- */
#define DZ_LAYOUT_LINEAR 0
#define DZ_LAYOUT_BRICK 1
size_t dz_index_words(int b, int d, int h, int w, int layout);            /* uint32 words in bitmap / prefix */
size_t dz_index_workspace_bytes(int b, int d, int h, int w, int layout);

/* Build a level from (n,4) i32 [b,z,y,x] coordinates in any order (duplicates allowed).
 * d_n may be NULL (then n_cap rows are all valid).  Writes bitmap, prefix, canonical coords and
 * *d_m; rank_of_input (n_cap) receives the canonical row of each input row (may be NULL). */
int dz_index_from_coords(const int *coords, const int *d_n, int n_cap, int b, int d, int h, int w, int layout,
                         uint32_t *bitmap, uint32_t *prefix, int *coords_out, int *d_m, int cap_out,
                         int *rank_of_input, void *ws, size_t ws_bytes, void *stream);

/* Output level of a regular sparse convolution (SparseConv3d, backbone3d.py:256-277):
 * out dims = floor((in + 2p - k)/s) + 1; a site is active iff >=1 active input in its window. */
int dz_index_downsample(const int *coords_in, const int *d_m_in, int cap_in, int b, int d, int h,
                        int w, int layout, const int *h_k3, const int *h_s3, const int *h_p3, uint32_t *bitmap_out,
                        uint32_t *prefix_out, int *coords_out, int *d_m_out, int cap_out, void *ws,
                        size_t ws_bytes, void *stream);

/* Rulebook in output-stationary form: nbr[t*cap_out + o] = input row feeding output row o
 * through kernel tap t = (tz*kH+ty)*kW+tx (input coordinate = o*s - p + t), or -1.
 * SubMConv3d = (k=3,s=1,p=1) with the output level equal to the input level. */
int dz_build_neighbors(const int *coords_out, const int *d_m_out, int cap_out, const uint32_t *bitmap_in,
                       const uint32_t *prefix_in, int b, int d, int h, int w, int layout, const int *h_k3,
                       const int *h_s3, const int *h_p3, int *nbr, uint32_t *tile_masks, void *stream);
/* tile_masks (dz_tile_masks_words(cap_out) words, 16-byte aligned, or NULL): bit t of word o/32 = some row of the 32-row
 * group o/32 has a neighbour at tap t (what the conv kernels need to skip empty taps without scanning the table). */
int dz_tile_masks_words(int cap_out);
/* The same rulebook PACKED (DZ_NBR_PACKED) for 3 x 3 x 3 windows with x padding 1 on linear keys (every 27-tap table of
 * VoxelResBackBone8x, backbone3d.py:243-280): rows are in key order, so the three x taps of a (tz, ty) row are consecutive input
 * rows - one word per (tz, ty) and output row holds them: nbr[(tz*3+ty)*cap_out + o] = r | left << 29 | centre << 30 | right << 31
 * with r = active input cells below the centre cell of the window; left neighbour = r - 1, centre = r, right = r + centre (each
 * only when its bit is set).  9 words per output row instead of 27: a third of the table bytes written here and read by every
 * convolution of the level.  tile_masks as dz_build_neighbors (required).  DZ_ERR_UNSUPPORTED for other windows / layouts. */
int dz_build_neighbors_packed(const int *coords_out, const int *d_m_out, int cap_out, const uint32_t *bitmap_in,
                              const uint32_t *prefix_in, int b, int d, int h, int w, int layout, const int *h_k3,
                              const int *h_s3, const int *h_p3, int *nbr, uint32_t *tile_masks, void *stream);

/* dst[rank[i]][0:c_src] = src[i][0:c_src]; dst[rank[i]][c_src:c_dst] = 0 (rows with rank<0 skipped) */
int dz_scatter_rows(const float *src, const int *rank, const int *d_n, int n_cap, int c_src, float *dst,
                    int c_dst, void *stream);

/* Sparse convolution forward with fused epilogue
 *   out[o] = relu?( (sum_t in[nbr[t][o]] . W[t]) * scale + shift (+ residual[o]) )
 * == SubMConv3d/SparseConv3d + BatchNorm1d(eval, folded into scale/shift together with the conv
 * bias) + residual add + ReLU of backbone3d.py:77-81,105-121.  fp32 in, fp32 MFMA accumulate.
 *   in (in_rows, cin) f32 (in_rows = row capacity of the buffer, < 2 GiB), cin in {16,32,64,128};
 *   w (kvol,cin,cout) f32; cout in {16,32,64,128}. */
int dz_spconv_forward(const float *in, int in_rows, int cin, const int *nbr, int kvol, int cap_out, const int *d_m_out,
                      const float *w, const float *scale, const float *shift, const float *residual,
                      int relu, float *out, int cout, void *stream);

/* spconv SparseConvTensor.dense() + HeightCompression reshape (height_compression.py:20-24),
 * written channel-last into a zero-bordered BEV image: bev[b][y+pad][x+pad][c*D + z] = feats[o][c].
 * The caller zero-fills `bev` (hipMemsetAsync) before the call. */
int dz_sparse_to_bev(const float *feats, const int *coords, const int *d_m, int cap, int c, int d, int h,
                     int w, int pad, float *bev, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Dense BEV network (torch.nn.Conv2d / ConvTranspose2d / BatchNorm2d / ReLU of
 * backbone2d.py:33-120 and center_head.py:14-48,81-102), channel-last implicit GEMM on fp32 MFMA.
 * ------------------------------------------------------------------------------------------- */
typedef struct dz_conv2d_desc {
    const float *in;      /* (B, in_hp, in_wp, in_cstride) channel-last, zero border included   */
    float *out;           /* (B, out_hp, out_wp, out_cstride)                                  */
    const float *w;       /* (groups, kh*kw, cin, cout_pad)                                    */
    const float *scale;   /* (groups*cout_pad) or NULL (=1)                                    */
    const float *shift;   /* (groups*cout_pad) or NULL (=0)                                    */
    int batch, ho, wo;    /* logical output extent enumerated by the kernel                    */
    int in_hp, in_wp, in_cstride, in_coff, cin;
    int kh, kw, stride, in_off; /* input pixel = (y*stride + ky + in_off, x*stride + kx + in_off) */
    int out_hp, out_wp, out_cstride, out_coff;
    int out_sy, out_sx, out_dy, out_dx; /* output pixel = (y*out_sy + out_dy, x*out_sx + out_dx) (pad included) */
    int groups, cout_pad;
    int g_cout[8];        /* valid output channels of each group                               */
    int g_ooff[8];        /* output channel offset of each group (added to out_coff)           */
    int relu;
    const float *group_shift; /* optional (n_row_groups, cout_pad): added to the accumulator BEFORE scale/shift, */
    int group_rows;           /* row group = output row / group_rows (per-object bias of the PointNet concat)   */
    int group_max;            /* dz_linear_forward_split only: out = (row groups, out_cstride) fp32 pre-filled with -inf, the max over each
                                 group's rows is taken in the epilogue (no (rows, cout) result is written)          */
    int phase_groups;         /* dz_conv2d_forward_split only, != 0: the `groups` are the out_sy x out_sx PHASES of a ConvTranspose2d with
                                 kernel == stride (backbone2d.py:89-99) in ONE launch: every group reads the SAME cin input channels
                                 (no per-group input offset), has its own weights w[g], shares scale / shift (cout_pad entries), and
                                 writes output pixel (y*out_sy + out_dy + g / out_sx, x*out_sx + out_dx + g % out_sx); the phases of a
                                 pixel tile are scheduled next to each other on one XCD, so the input image is fetched from HBM once */
    const int *in_rowidx;     /* dz_conv2d_forward_split only, != NULL: SPARSE INPUT of a 3 x 3 stride-1 layer - the image is never built.
                                 `in` = the rows of a sparse level with two z slabs (in_rows x in_row_channels pair16), in_rowidx =
                                 (batch, in_hp, in_wp, 2) int32: the row of every (pixel, z slab) of the zero-bordered image, -1 = empty
                                 (dz_bev_row_index).  Logical input channel j = z * in_row_channels + c (z-MAJOR: the layer's weights
                                 are permuted accordingly - HeightCompression's own order is c * 2 + z); cin = 2 * in_row_channels;
                                 in_cstride is ignored.  HeightCompression (height_compression.py:20-24) + the first block's ZeroPad2d
                                 (backbone2d.py:41-46) fused into that block's convolution */
    int in_row_channels, in_rows;
    const int *in_tiles;      /* 3 x 3 stride-1 resident-tile layers, optional: a list of dz_bev_tile_list (8 x 32 pixel tiles) - the launch covers its tiles to run only.
                                 Walked by the 64- / 128-channel tile kernels (pair16 output); IGNORED by the generic kernel (every pixel computed); refused
                                 (DZ_ERR_UNSUPPORTED) on the 32-channel tile kernel, whose tiles are 16 x 32 */
} dz_conv2d_desc;
int dz_conv2d_forward(const dz_conv2d_desc *h_desc, void *stream);
/* name of the kernel instance dz_conv2d_forward / dz_spconv_forward dispatch to (for profiling reports) */
const char *dz_conv2d_variant(const dz_conv2d_desc *h_desc);
const char *dz_spconv_variant(int cin, int cout);

/* ---------------------------------------------------------------------------------------------
 * Split-precision engine: the same sparse / dense convolutions on the 16-bit matrix cores.
 * Every fp32 value x travels as a pair of 16-bit floats (hi = rn16(x), lo = rn16(x - hi)); a product is
 * hi.hi + lo.hi + hi.lo in three v_mfma_f32_32x32x16_{f16,bf16} with fp32 accumulation (csrc/hgemm.h).
 * fp16 pairs carry 22 significant bits (results indistinguishable from fp32 summation-order noise, values
 * saturate at +-131008), bf16 pairs 16 bits with the full fp32 exponent range.
 * "pair16" layout of a row of C channels (C % 8 == 0): C 32-bit words, i.e. the byte size of the fp32 row;
 * each group of 8 channels = 16 bytes of hi values then 16 bytes of lo values.  Tensors keep the shapes,
 * strides and capacities of their fp32 counterparts (pointers are typed float* for that reason).
 * There is no fp32 conv in the reference to replace here beyond the ones named above: these entry points
 * are the throughput path of the same layers ("fp16/bf16-in / fp32-acc MFMA").
 * ------------------------------------------------------------------------------------------- */
#define DZ_MATH_F32 0
#define DZ_MATH_F16X2 1
#define DZ_MATH_BF16X2 2
#define DZ_MATH_F16 3      /* convolution entry points only: fp16-pair tensors (DZ_MATH_F16X2 storage), ONE fp16 MFMA per product (hi halves): plain-fp16 inputs, fp32 accumulation - a third of the matrix work, NOT fp32-class (opt-in fast mode) */
/* dst (rows, c_dst) pair16 <- src (rows, c_src) f32, channels c_src..c_dst-1 zero; c_dst % 8 == 0 */
int dz_pair16_from_f32(const float *src, long rows, int c_src, int c_dst, int math, float *dst, void *stream);
/* dst (rows, c) f32 <- src (rows, c) pair16 (hi + lo) */
int dz_pair16_to_f32(const float *src, long rows, int c, int math, float *dst, void *stream);
/* dz_scatter_rows writing pair16 rows */
int dz_scatter_rows_split(const float *src, const int *rank, const int *d_n, int n_cap, int c_src, float *dst,
                          int c_dst, int math, void *stream);
/* dz_spconv_forward on pair16 operands: in / residual / out pair16 rows, w (kvol, cout_pad, cin) pair16 with
 * cout_pad = max(cout, 32) (rows cout..cout_pad-1 zero), scale / shift (cout) f32. */
int dz_spconv_forward_split(const float *in, int in_rows, int cin, const int *nbr, const uint32_t *tile_masks, int kvol,
                            int cap_out, const int *d_m_out, const float *w, const float *scale, const float *shift,
                            const float *residual, int relu, float *out, int cout, int math, void *stream);
/* tile_masks: the buffer dz_build_neighbors filled for this table (NULL: the kernel scans the table itself, slower). */
/* Tile-resident sparse convolution (csrc/sparse_conv_t.hip; the same SubMConv3d / SparseConv3d call sites as dz_spconv_forward:
 * backbone3d.py:64-121, :243-280).  dz_build_tiles turns a neighbour table of dz_build_neighbors (kvol x cap_out, once per
 * indice_key) into, per tile of dz_spconv_tile_rows() = 512 consecutive output rows:
 *   halo   (ntiles x dz_build_tiles_halo_stride(kvol) int32)  the DISTINCT input rows the tile reads;
 *   tinfo  (ntiles x dz_spconv_tile_info_words() = 64 int32)  [0] slots = non-empty kernel taps of the tile, [1] halo rows,
 *          [4..35] tap of slot s (ascending; -1 beyond), [36..51] bit s of word f: sorted fragment f (32 rows) has a row with slot s;
 *   rowmap (ntiles x 512 uint16)  the tile's rows sorted by their tap set: row (relative to the tile) at sorted position q;
 *   ltab   (ntiles x dz_spconv_tile_table_entries() = 16384 uint16)  position in the halo list of the neighbour of (slot s, sorted
 *          position q), 0xFFFF = none, at [s / 4][q / 64][q % 32][s % 4][q / 32 % 2].
 * dz_spconv_tiles_forward stages a tile's halo rows in LDS once per 16-channel chunk and runs every kernel tap from there -
 * same operands, same result convention (pair16 in / out, BatchNorm scale / shift, residual, ReLU) as dz_spconv_forward_split.
 * kvol 27 with cout in {16, 32, 64, 128}, kvol in [3, 10] with cout 128; cin % 16 == 0.  Any row order of the level is correct;
 * the brick layout (DZ_LAYOUT_BRICK) is what keeps a tile's halo small. */
int dz_spconv_tile_rows(void);
int dz_spconv_tile_info_words(void);
int dz_spconv_tile_table_entries(void);
size_t dz_build_tiles_halo_stride(int kvol);
int dz_build_tiles(const int *nbr, int kvol, int cap_out, const int *d_m_out, int *halo, int *tinfo, unsigned short *ltab,
                   unsigned short *rowmap, void *stream);
int dz_spconv_tiles_forward(const float *in, int in_rows, int cin, const int *halo, const int *tinfo, const unsigned short *ltab,
                            const unsigned short *rowmap, int kvol, int cap_out, const int *d_m_out, const float *w,
                            const float *scale, const float *shift, const float *residual, int relu, float *out, int cout,
                            int math, void *stream);
const char *dz_spconv_tiles_variant(int cin, int cout);
/* dz_spconv_forward_split over a PACKED 27-tap table (dz_build_neighbors_packed): the small-channel levels (16 -> 16, 16 -> 32,
 * 32 -> 32), whose HBM bytes are one third table in the unpacked form.  Same arithmetic, bit-identical results. */
int dz_spconv_forward_split_packed(const float *in, int in_rows, int cin, const int *nbr_packed, const uint32_t *tile_masks,
                                   int cap_out, const int *d_m_out, const float *w, const float *scale, const float *shift,
                                   const float *residual, int relu, float *out, int cout, int math, void *stream);
/* Submanifold 3 x 3 x 3 convolutions with the inputs of a z slab staged once per tile (csrc/sparse_conv_x.hip, the "x-run" engine;
 * the SubMConv3d pairs of SparseBasicBlock, backbone3d.py:93-121, at 32 / 64 / 128 channels: conv2, conv3, conv4 of
 * VoxelResBackBone8x, backbone3d.py:261-280).  Rows of a level are in ascending linear key, so the nine taps of one z offset of a tile
 * of T consecutive output rows read ONE contiguous range of input rows (a "window").
 *   dz_spconv_x_tile_rows(cin, cout): rows per UNIT of the kernel for this layer (a tile is two consecutive units; the last tiles of
 *     a launch are single units, for load balance), 0 = layer not covered (cin != cout, other widths).
 *   dz_spconv_x_windows: from the PACKED table of the level (dz_build_neighbors_packed, input level == output level) the windows of
 *     every unit: windows[(unit*3 + tz)*2 + {0, 1}] = first input row, number of rows (0 = slab has no neighbour; the centre slab of a
 *     live unit always has one), followed by 16 words of tile-queue state the convolution kernels use (zeroed here, left zero by
 *     every launch).  dz_spconv_x_windows_words(cap_out, tile_rows) int32 in all.  Once per indice_key, shared by the level's
 *     convolutions (which therefore must not run concurrently on the same windows buffer).
 *   dz_spconv_forward_split_x: operands, result convention and arithmetic of dz_spconv_forward_split (pair16 in / residual / out,
 *     w (27, cout, cin) pair16, BatchNorm scale / shift, ReLU); per output element the products are accumulated in the order
 *     (tz, 16-channel chunk, tap), so results agree with dz_spconv_forward_split to fp32 summation-order noise, not bit for bit. */
int dz_spconv_x_tile_rows(int cin, int cout);
/* rows of a z-slab window the kernel can stage in LDS for this layer (0 = layer not covered); a (tile, slab) whose window is longer
 * runs in gather mode - same taps, same accumulation order, operands fetched per lane instead of from the staged window */
int dz_spconv_x_window_rows(int cin, int cout);
size_t dz_spconv_x_windows_words(int cap_out, int tile_rows);
int dz_spconv_x_windows(const int *nbr_packed, int cap_out, const int *d_m_out, int tile_rows, int *windows, int *nbr_sorted, int *perm,
                        void *stream);
/* dz_build_neighbors_packed (3 x 3 x 3 submanifold table of a level onto itself, linear keys) and dz_spconv_x_windows in ONE launch:
 * the packed table, its per-32-row tap masks, the windows of every unit of tile_rows (128 | 256 = dz_spconv_x_tile_rows) rows and -
 * optionally, both or neither - the table / row map in tap-set order.  Same outputs, bit for bit, as the two calls; a workgroup
 * owns a 256-row block and keeps the nine words of a row in registers instead of reading the table back (round 5: the second
 * launch cost 105 us per level and step at 32 frames).  (b, d, h, w) = the level's grid; coords / d_m / bitmap / prefix its index. */
int dz_build_neighbors_packed_x(const int *coords, const int *d_m, int cap, const uint32_t *bitmap, const uint32_t *prefix, int b, int d,
                                int h, int w, int *nbr_packed, uint32_t *tile_masks, int tile_rows, int *windows, int *nbr_sorted,
                                int *perm, void *stream);
/* nbr_sorted (9 x cap_out) / perm (cap_out), both or neither: the rows of every unit re-ordered by their tap set (ascending in even
 * units, descending in odd ones): perm[position] = output row, nbr_sorted = the packed table in that order.  Fragments of 32 rows
 * with similar tap sets let the kernel skip 11-13 % more (fragment, tap) pairs; pass both to dz_spconv_forward_split_x. */
int dz_spconv_forward_split_x(const float *in, int in_rows, int cin, const int *nbr_packed, const int *perm, int *windows, int tile_rows,
                              int cap_out, const int *d_m_out, const float *w, const float *scale, const float *shift,
                              const float *residual, int relu, float *out, int cout, int math, void *stream);
/* (nbr_packed = the level's packed table and perm = NULL, or nbr_sorted and its perm) */
const char *dz_spconv_x_variant(int cin, int cout);
/* HeightCompression WITHOUT the dense image (round 5): idx (batch, h + 2 pad, w + 2 pad, 2) int32 = the feature row of every
 * (pixel, z slab) of a two-slab level, -1 = empty cell / border / rank >= feat_rows (overflowed capacity).  The first block of
 * BaseBEVBackbone reads the level's rows through it (dz_conv2d_desc.in_rowidx): height_compression.py:20-24 and the ZeroPad2d of
 * backbone2d.py:41-46 never touch HBM.  DZ_ERR_UNSUPPORTED for d != 2. */
int dz_bev_row_index(const uint32_t *bitmap, const uint32_t *prefix, int batch, int d, int h, int w, int layout, int pad, int feat_rows,
                     int *idx, void *stream);
/* Pixel tiles (8 x 32 output pixels) of the first BEV block whose result is the network's ZERO-INPUT RESPONSE (round 5).  With D = the
 * Chebyshev distance of a pixel to the nearest pixel holding a row (nothing outside the image), the 3 x 3 stride-1 layer number l of the
 * block (l = 1: the sparse-input layer) sees, in a tile whose pixels all have D >= l + 1, exactly what it sees on an all-zero input - and
 * produces what it produces there.  dz_bev_tile_list writes, for l = 1 .. nlists (< 8), list l at lists + (l - 1) * dz_bev_tile_list_words:
 * [n to run, n skippable, ids of the tiles to run ascending ..., the skippable ones ...] (tile id = (b * tiles_y + ty) * tiles_x + tx);
 * mind_ws = one byte per tile.  Pass list l as dz_conv2d_desc.in_tiles of layer l: the launch walks the tiles to run only;
 * dz_bev_fill_empty_tiles gives the others their result: zero_resp = the layer's output on an all-zero input (1, out_hp, out_wp, cout)
 * pair16, computed once per model by the same kernels - or NULL for layer 1, whose response is the constant ReLU(shift).  Bit-identical
 * to running every tile (a pixel's result depends on its 3 x 3 input neighbourhood only). */
size_t dz_bev_tile_list_words(int batch, int ho, int wo);
int dz_bev_tile_list(const int *idx, int batch, int hp, int wp, int ho, int wo, int nlists, int *lists, unsigned char *mind_ws, void *stream);
int dz_bev_fill_empty_tiles(const int *list, int batch, int ho, int wo, const float *shift, int relu, int cout, float *out, int out_hp, int out_wp,
                            int out_cstride, int out_coff, const float *zero_resp, int math, void *stream);
/* dz_sparse_to_bev on pair16 rows / images (the 16-bit halves are moved, no arithmetic) */
int dz_sparse_to_bev_split(const float *feats, const int *coords, const int *d_m, int cap, int c, int d, int h,
                           int w, int pad, float *bev, void *stream);
/* the same image written whole (zeros included) from the level's own index instead of scattered into a zero-filled canvas:
 * for d == 2 z slabs (VoxelResBackBone8x's encoded tensor; DZ_ERR_UNSUPPORTED otherwise).  bitmap / prefix = the level's index
 * (dz_index_downsample), batch / d / h / w / layout its grid.  Same bytes as fill + dz_sparse_to_bev_split
 * (height_compression.py:20-24: channel = ch * D + z).  feat_rows = rows of the `feats` buffer (the level's row capacity): a cell
 * whose rank is not below it - a level that overflowed a calibrated capacity keeps every site in its bitmap - is written as empty,
 * exactly what the scatter form does with the rows it drops; nothing is read past the buffer. */
int dz_sparse_to_bev_split_dense(const float *feats, int feat_rows, const uint32_t *bitmap, const uint32_t *prefix, int batch, int c,
                                 int d, int h, int w, int layout, int pad, float *bev, void *stream);
/* dz_conv2d_forward on pair16 images: desc->in pair16, desc->w (groups, kh*kw, cout_pad, cin) pair16
 * (cout_pad % 32 == 0, cin % 32 == 0), desc->out pair16 or, with out_f32 != 0, plain fp32 (last head conv). */
int dz_conv2d_forward_split(const dz_conv2d_desc *h_desc, int math, int out_f32, void *stream);
const char *dz_conv2d_variant_split(const dz_conv2d_desc *h_desc);
/* dz_linear_forward on pair16 rows: x (rows, x_stride words) pair16, w (cout_pad, cin) pair16 (cin, cout_pad % 32 == 0), y pair16
 * rows or, with out_f32 != 0, fp32 rows; group_shift (row groups, cout_pad) fp32 as in dz_linear_forward.
 * group_max != 0 (with out_f32): the torch.max over the points of an object that follows the PointNet encoders
 * (geometry_transformer.py:124,137, position_transformer.py:108,117) fused into the layer: y = (rows / group_rows, y_stride) fp32,
 * the maximum over each group's rows; the (rows, cout) activation is not written.  group_rows % 128 == 0. */
int dz_linear_forward_split(const float *x, long rows, int cin, int x_stride, const float *w, int cout, int cout_pad, const float *scale,
                            const float *shift, const float *group_shift, int group_rows, int relu, float *y, int y_stride, int math,
                            int out_f32, int group_max, void *stream);
const char *dz_spconv_variant_split(int cin, int cout);
/* Cross-attention of a few queries over a long memory with the key / value projections folded into the queries (csrc/xattn_fold.hip;
 * multi_head_attention.py:199-288 as called by decoder.py:79-84 for the geometry refiner: 3 queries x 4096 memory points x 8 heads).
 * Equal to  out = softmax(scale * (q_h) . (Wk_h m + bk_h)) (Wv_h m + bv_h)  per head, computed WITHOUT projecting the memory:
 * q (b, lq, e) fp32 = the projected queries (Wq x + bq, NOT yet scaled); mem (b, lk, e) fp32 = the raw memory rows; wk_oi (e, e) = Wk as
 * stored by torch (rows = output channel); wv_io (e, e) = Wv^T (rows = input channel); bv (e); key_padding_mask (b, lk) bytes or NULL
 * (non-zero = ignore); out (b, lq, e) fp32 = the heads' outputs before out_proj.  bk cancels in the softmax.  fp32 throughout.
 * Supported (dz_xattn_folded_supported) when e == 256, e % heads == 0 and heads * lq <= 32; workspace from dz_xattn_folded_workspace_bytes. */
int dz_xattn_folded_supported(int lq, int e, int heads);
size_t dz_xattn_folded_workspace_bytes(int b, int lk);
int dz_xattn_folded(const float *q, const float *mem, const uint8_t *key_padding_mask, const float *wk_oi, const float *wv_io, const float *bv,
                    int b, int lq, int lk, int e, int heads, float scale, float *workspace, size_t workspace_bytes, float *out, void *stream);
/* The refiner's memory branch behind its PointNet encoder in one kernel (csrc/mlp_chain.hip; geometry_transformer.py:56-67,126-133,
 * position_transformer.py:60-72,108-117, multi_head_attention.py:199-236):  h = ReLU(sa * (Wa x + group_shift[row / group_rows]) + ba)
 * (128 -> 512), mem = ReLU(sb * (Wb h) + bb) (512 -> 256), and - when wk is not NULL - K = Wk mem + bk, V = Wv mem + bv (256 -> 256).
 * x (rows, 128) pair16 (the tapped encoder layer); wa (512, 128), wb (256, 512), wk / wv (256, 256) pair16, rows = output channel;
 * group_shift (rows / group_rows, ldg >= 512) fp32 or NULL; mem / k / v (rows, 256) fp32.  Split math only; group_rows % 32 == 0;
 * rows * 1024 < 2 GiB.  Bit-identical to the same layers run one dz_linear_forward_split each. */
int dz_mlp_chain_forward(const float *x, long rows, const float *wa, const float *sa, const float *ba, const float *group_shift, int ldg, int group_rows,
                         const float *wb, const float *sb, const float *bb, const float *wk, const float *bk, const float *wv, const float *bv,
                         float *mem, float *k, float *v, int math, void *stream);
/* Fused PointNet encoder (csrc/pointnet.hip; geometry_transformer.py:34-67,118-140, position_transformer.py:43-124): three
 * point-wise layers 32 -> 128 -> 128 -> c3 (c3 in {128, 256, 512}; BatchNorm scale / shift + ReLU after each) and the max over
 * every group of group_rows consecutive rows (group_rows % 32 == 0, rows % group_rows == 0) in one kernel; the activations never
 * leave the registers.  x (rows, 32) pair16 - or, with x_cols_f32 = 16 / 32, (rows, x_cols_f32) fp32 rows split on the way in (columns
 * beyond x_cols_f32 count as zero); w1 (128, 32), w2 (128, 128), w3 (c3, 128) pair16; out (rows / group_rows, c3) fp32;
 * tap (rows, 128) pair16 or NULL = the second layer's output (the reference reads it through a forward hook).  Split math only. */
int dz_pointnet3_forward(const float *x, long rows, const float *w1, const float *s1, const float *b1, const float *w2, const float *s2,
                         const float *b2, const float *w3, const float *s3, const float *b3, int c3, int group_rows, float *tap, float *out,
                         int x_cols_f32, int math, void *stream);

/* ---------------------------------------------------------------------------------------------
 * CenterHead decode + NMS (center_head.py:315-368, centernet_utils.py:138-230,
 * model_nms_utils.py:6-25, utils/detzero_utils/ops/iou3d_nms/)
 * ------------------------------------------------------------------------------------------- */
/* head (B, HW, 12) channel-last, columns [center 0:2 | center_z 2 | dim 3:6 | rot 6:8 | iou 8 | hm 9:12].
 * Produces, per batch item, the top-K (score = sigmoid(hm)*clamp(iou,0,1)^2) candidates in
 * descending score order that pass the score threshold and the centre range test:
 *   boxes (B,K,7) [x,y,z,dx,dy,dz,heading], scores (B,K), labels (B,K) i32 0-based, d_counts (B). */
size_t dz_centerhead_decode_workspace_bytes(int batch, int hw, int ncls, int k);
int dz_centerhead_decode(const float *head, int batch, int h, int w, int ncls, int k, float score_thresh,
                         const float *h_limit6, const float *h_range6, const float *h_vsize3, int stride,
                         int use_iou, float *boxes, float *scores, int *labels, int *d_counts, void *ws,
                         size_t ws_bytes, void *stream);

/* RoI features of the first-stage boxes (center_head.py:408-432,461-486: get_box_center with num_point 5 + absl_to_relative +
 * centernet_utils.bilinear_interpolate_torch:233-262): boxes (n, 7) of ONE frame; bev = that frame's (h, w, c) map, channel stride 1,
 * pixel / row strides in floats; map cell of a point = (p - lo) / voxel / stride; out (n, 5 * c) = [centre | front | back | left | right]. */
int dz_roi_bev_features(const float *boxes, int n, const float *bev, long row_stride, long pix_stride, int h, int w, int c, float x_lo, float y_lo,
                        float voxel_x, float voxel_y, int stride, float *out, void *stream);

/* iou3d_nms_cuda.nms_gpu (iou3d_nms.cpp:114-160 + nms_kernel iou3d_nms_kernel.cu:386-430), with the
 * suppression sweep done on the device.  boxes (n_cap,7) already in descending score order;
 * *d_n of them valid (d_n may be NULL).  keep (n_cap) i32 receives kept indices in order,
 * *d_num_keep their number (limited to post_max). */
size_t dz_nms_workspace_bytes(int n_cap);
int dz_nms_rotated(const float *boxes, const int *d_n, int n_cap, float thresh, int post_max, int *keep,
                   int *d_num_keep, void *ws, size_t ws_bytes, void *stream);
/* the same for `batch` frames in one launch pair: boxes (batch,n_cap,7), d_n (batch), keep (batch,n_cap),
 * d_num_keep (batch); ws >= batch * dz_nms_workspace_bytes(n_cap) */
int dz_nms_rotated_batched(const float *boxes, const int *d_n, int batch, int n_cap, float thresh, int post_max,
                           int *keep, int *d_num_keep, void *ws, size_t ws_bytes, void *stream);
/* final selection of model_nms_utils.py:22-25 / center_head.py:350-360 as one gather:
 * out (batch, post_max, 9) rows [x,y,z,dx,dy,dz,heading,score,label(1-based)] of the kept candidates, zeros after */
int dz_pack_detections(const float *boxes, const float *scores, const int *labels, const int *keep,
                       const int *d_num_keep, int batch, int k, int post_max, float *out, void *stream);

/* iou3d_nms_cuda.boxes_overlap_bev_gpu / boxes_iou_bev_gpu (iou3d_nms.cpp:60-111) : (na,nb) f32 */
int dz_boxes_overlap_bev(const float *a, int na, const float *b, int nb, float *out, void *stream);
int dz_boxes_iou_bev(const float *a, int na, const float *b, int nb, float *out, void *stream);

/* gather rows: out[i] = src[idx[i]] for i < *d_n (used to apply the NMS keep list on device) */
int dz_gather_rows(const float *src, const int *idx, const int *d_n, int n_cap, int c, float *out,
                   void *stream);

/* ---------------------------------------------------------------------------------------------
 * Refining module, secondary kernel set
 * ------------------------------------------------------------------------------------------- */
/* Points per box: counts[t] = number of points inside box t, the inside test of dz_points_in_boxes_v2, no (T, M) mask
 * (roiaware_pool3d_utils.points_in_boxes_num_gpu -> points_in_boxes_num kernel, used by tracking/detzero_track/datasets/
 * data_processor.py:64-69).  boxes (t,7) f32, pts (m,3) f32, counts (t,) i32 (overwritten). */
int dz_points_in_boxes_count(const float *boxes, const float *pts, int t, int m, int *counts, void *stream);

/* roiaware_pool3d_cuda.points_in_boxes_gpu_v2 (roiaware_pool3d.cpp:135-155,
 * roiaware_pool3d_kernel.cu:352-374): mask (B,T,M) i32, 1 where point m is inside box t. */
int dz_points_in_boxes_v2(const float *boxes, const float *pts, int batch, int t, int m, int *mask,
                          void *stream);
/* Object crop with compaction - the consumer side of points_in_boxes_gpu_v2 in daemon/prepare_object_data.py:250-273,310
 * (`obj_pts = pts[obj_pts_mask[idx, :]]` per object) without the dense (T, M) mask or its D2H copy.
 *   xyz (m,3) f32: point coordinates used for the inside test (same test as dz_points_in_boxes_v2);
 *   boxes (t,7) f32 (already enlarged by the caller); payload (m, payload_words) 32-bit words: the rows to gather
 *   (e.g. 4 float64 = 8 words); out (cap, payload_words): the kept rows of box 0, then box 1, ... each in ascending
 *   point order; out_index (cap) i32 or NULL: source point of every kept row; offsets (t+1) i32: row range of each
 *   box; *d_total: number of kept rows (rows beyond cap are dropped, compare with cap). */
size_t dz_crop_points_workspace_bytes(int m, int t, int cap);
int dz_crop_points_in_boxes(const float *xyz, int m, const float *boxes, int t, const void *payload, int payload_words,
                            void *out, int *out_index, int *offsets, int *d_total, int cap, void *ws, size_t ws_bytes,
                            void *stream);

/* multi_head_attention_forward core (refining/detzero_refine/models/transformer/
 * multi_head_attention.py:207-288): q (B,Lq,E), k,v (B,Lk,E) already projected, E = heads*32;
 * key_padding_mask (B,Lk) u8 (1 = ignore) or NULL; out (B,Lq,E).  softmax(q*scale . k^T) . v */
int dz_mha_core(const float *q, const float *k, const float *v, const uint8_t *key_padding_mask, int batch,
                int lq, int lk, int heads, float scale, float *out, void *stream);
/* The same core on the 16-bit matrix cores with q, k, v and the probabilities carried as (hi, lo) 16-bit pairs (csrc/mha_h.hip; three
 * MFMAs per product, fp32 accumulation and softmax): the arithmetic class of dz_linear_forward_split, for long key lists (PRM's
 * cross-attention: 200 queries x 9600 keys).  Same arguments; math = DZ_MATH_F16X2 or DZ_MATH_BF16X2; head dim 32. */
int dz_mha_core_split(const float *q, const float *k, const float *v, const uint8_t *key_padding_mask, int batch, int lq, int lk, int heads,
                      float scale, float *out, int math, void *stream);

/* y (rows, cout) = relu?( x (rows, cin) . W (cin, cout_pad) * scale + shift ) — the 1x1 Conv1d/Conv2d
 * + BatchNorm + ReLU stacks of the GRM/PRM PointNet encoders
 * (refining/detzero_refine/models/modules/geometry_transformer.py:34-67), same MFMA engine. */
int dz_linear_forward(const float *x, int rows, int cin, int x_stride, const float *w, int cout, int cout_pad,
                      const float *scale, const float *shift, const float *group_shift, int group_rows, int relu,
                      float *y, int y_stride, void *stream);

/* dz_linear_forward for a few rows against a very long input (the PDV head's first FC layer, pdv_head.py:154-172: 41 472 -> 256 for
 * some hundred RoIs): the input channels are cut into `splits` (1..8, cin % (splits * 32) == 0) groups that run side by side (one
 * grouped launch into the workspace) and are summed in a second pass with scale / shift / ReLU.  Same arguments otherwise. */
size_t dz_linear_splitk_workspace_bytes(int rows, int cout_pad, int splits);
int dz_linear_forward_splitk(const float *x, int rows, int cin, int x_stride, const float *w, int cout, int cout_pad, const float *scale,
                             const float *shift, int relu, float *y, int y_stride, int splits, float *workspace, size_t workspace_bytes,
                             void *stream);

/* out (groups, c) = max over the `len` rows of each group of x (groups*len, c): the point-wise max pooling of
 * the GRM/PRM PointNet encoders (geometry_transformer.py:124,137; position_transformer.py:108,117). */
int dz_group_max(const float *x, int groups, int len, int c, float *out, void *stream);

/* out = LayerNorm(x + y) * gamma + beta over rows of 256 channels (decoder.py:75-88: residual + post-LN);
 * y may be NULL; do_norm == 0 gives the plain sum x + y (with_pos_embed, decoder.py:50-51). */
/* out[i] = 1 when every int of row i of all n (<= 4) row-major int32 tensors t[k] (rows x w[k], w[k] % 4 == 0) is zero: PDV's
 * empty-grid-point mask `(ball_idxs == 0).all(-1)` (pdv_head.py:521-523) over the per-branch ball indices, without concatenating them. */
int dz_rows_all_zero(const int *const *t, const int *w, int n, int rows, unsigned char *out, void *stream);
int dz_add_layernorm(const float *x, const float *y, const float *gamma, const float *beta, int rows, int c,
                     float eps, int do_norm, float *out, void *stream);
/* out = post + (group_skip[row / group_rows] ? post : LayerNorm(x + y)), c = 192: the PDV encoder layer's second normalisation together
 * with COMBINE (pooled + attended features) and the untouched rows of RoIs without points (pdv_head.py:540-560, attention_utils.py:31-44). */
int dz_add_layernorm_combine(const float *x, const float *y, const float *gamma, const float *beta, int rows, int c, float eps, const float *post,
                             const unsigned char *group_skip, int group_rows, float *out, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Object features: cropped object points -> inputs of the refining models.
 * Replaces, for inference, WaymoGeometryDataset.extract_track_feature
 * (refining/detzero_refine/datasets/waymo/waymo_geometry_dataset.py:73-131) and WaymoPositionDataset.extract_track_feature
 * (waymo_position_dataset.py:66-155) with their helpers (refining/detzero_refine/utils/data_utils.py:6-113,
 * utils/detzero_utils/box_utils.py:28-53), batched over objects.
 *
 * Common inputs (device): pts (P,4) float64 [x,y,z global, intensity] of all boxes, object-major then frame order;
 * box_offsets (F+1) first point of every box; traj (F,7) float64 global boxes; score (F) float64;
 * obj_box_offsets (B+1) first box of every object.  WHICH points survive the fixed-size selection is the caller's
 * (data_utils.py:12-30 draws with Python's random.sample): index lists, -1 = zero row.
 * ------------------------------------------------------------------------------------------------------------------ */
/* Device-side alternative to the host draw: set s of the call (global id first_set_id + s) keeps, of its counts[s] rows, a
 * uniformly random k-subset in ascending order (all rows when counts[s] < k, padded with -1) - the distribution of
 * sample_points, from a counter-based generator keyed by (seed, set id) instead of Python's random stream.  out_idx (n_sets,k). */
int dz_draw_subsets(const int *counts, int n_sets, int k, unsigned long long seed, int first_set_id, int *out_idx, void *stream);
#define DZ_GRM_XYZ 1
#define DZ_GRM_INTENSITY 2
#define DZ_GRM_P2S 4
#define DZ_GRM_SCORE 8
int dz_grm_feature_channels(int encoding);           /* channels of a memory row for these flags (order: xyz, intensity, p2s, score) */
/* mem_idx (B,mem_n): index into the object's concatenated points; query_box (B,q_max): box id in [0,F) or -1 (padding);
 * query_idx (B,q_max,q_n): index into that box's points.  memory (B,mem_n,channels) and query (B,q_max,q_n,4) float32:
 * points in the frame of their own box (waymo_geometry_dataset.py:75, data_utils.py:62-71). */
int dz_grm_encode_points(const double *pts, const int *box_offsets, const double *traj, const double *score,
                         const int *obj_box_offsets, const int *mem_idx, int mem_n, const int *query_box, const int *query_idx,
                         int q_max, int q_n, int batch, int encoding, float *memory, float *query, void *stream);

#define DZ_PRM_XYZ 0
#define DZ_PRM_INTENSITY 1
#define DZ_PRM_P2CO 2
#define DZ_PRM_SCORE 3
#define DZ_PRM_CLASS 4
int dz_prm_feature_channels(const int *h_encoding, int n_enc);   /* channels of a row for this (host) list of codes, -1 if invalid */
/* q_idx (F,q_n) / m_idx (F,m_n): per box, index into its points.  Outputs: query (B,box_max,q_n,channels), memory
 * (B,box_max,m_n,channels), traj_local (B,box_max,7) and padding_mask (B,box_max) float32, init_box (B,7) float64 - points
 * and boxes in the frame of the object's middle box (waymo_position_dataset.py:72-78, data_utils.py:74-113), boxes past
 * an object's own are zero with mask 1.  anchors: scratch of B*box_max*27 + 2*B doubles.  obj_cls (B) 1-based, may be
 * NULL without the 'class' feature. */
int dz_prm_encode_points(const double *pts, const int *box_offsets, const double *traj, const double *score,
                         const int *obj_box_offsets, const int *obj_cls, const int *q_idx, const int *m_idx, int q_n, int m_n,
                         int batch, int box_max, const int *h_encoding, int n_enc, float *query, float *memory,
                         float *traj_local, float *padding_mask, double *init_box, double *anchors, void *stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Test-time augmentation: point transforms of the augmented copies
 * (detection/detzero_det/datasets/augmentor/test_time_augmentor.py:32-83), restore of their boxes
 * (detection/detzero_det/models/centerpoint.py:165-203) and weighted box fusion
 * (detection/detzero_det/utils/ensemble_utils/wbf_3d.py:10-203 as called by ensemble.py:7-33), all on the device.
 * Operations: host arrays of codes + one float parameter each (angle in radians / scale factor, 0 otherwise).
 * ------------------------------------------------------------------------------------------------------------------ */
#define DZ_TTA_ORIGINAL 0
#define DZ_TTA_FLIP_X 1
#define DZ_TTA_FLIP_Y 2
#define DZ_TTA_FLIP_XY 3
#define DZ_TTA_ROT 4
#define DZ_TTA_SCALE 5
/* points (n,c) -> out (n_ops, n, c): copy i = points under operation i (columns >= 3 copied). */
int dz_tta_augment_points(const float *points, int n, int c, const int *h_kind, const float *h_param, int n_ops, float *out,
                          void *stream);
/* boxes (frames, n_ops, m, dim >= 7) in place: rows of copy i back to the original frame (columns >= 7 untouched). */
int dz_tta_restore_boxes(float *boxes, int frames, int n_ops, int m, int dim, const int *h_kind, const float *h_param, void *stream);
size_t dz_wbf_workspace_bytes(int frames, int cand);
/* weighted_boxes_fusion_3d with iou_type '3d' per frame: boxes (frames, cand, 7) float32, scores (frames, cand), labels
 * (frames, cand) in 1..3 (0 = padding); candidate c belongs to model c / per_model (weights: device doubles per model or NULL
 * = 1; weight_sum = their sum).  conf_max: 0 'avg', 1 'max'.  Outputs sorted by fused score: out_boxes (frames, cand, 7) and
 * out_scores (frames, cand) float64 as in the reference, out_labels, out_count (frames). */
int dz_wbf_fuse_3d(const float *boxes, const float *scores, const int *labels, const int *obj_ids, int frames, int cand, int per_model,
                   const double *weights, int n_models, const double *h_iou_thr3, const double *h_skip_thr3, double weight_sum,
                   int conf_max, int allows_overflow, double *out_boxes, double *out_scores, int *out_labels, int *out_obj_ids,
                   int *out_count, void *ws, size_t ws_bytes, void *stream);
/* obj_ids / out_obj_ids (frames, cand), both or neither: weighted_tracking_boxes_fusion_3d (wbf_3d.py:205-265) - a fused box
 * carries the object id of its most confident member that has one (>= 0), else -1. */

/* ------------------------------------------------------------------------------------------------------------------
 * Frame assembly from stored sweeps (DatasetTemplate.merge_sweeps, detection/detzero_det/datasets/dataset.py:164-195):
 * raw (n_total,6) float32 rows [x,y,z,intensity,elongation,NLZ] of n_sweeps sweeps back to back (host offsets,
 * n_sweeps+1 entries); h_transforms: per sweep rows 0..2 of inv(pose_current) @ pose_sweep (12 doubles, row-major);
 * h_time_offsets: seconds.  out (n_total,6) float32 receives the rows with NLZ == -1 in their stored order as
 * [x',y',z',tanh(intensity),elongation,time offset]; *d_count their number. */
size_t dz_merge_sweeps_workspace_bytes(int n_total);
int dz_merge_sweeps(const float *raw, int n_total, const int *h_sweep_offsets, const double *h_transforms, const double *h_time_offsets,
                    int n_sweeps, float *out, int *d_count, void *ws, size_t ws_bytes, void *stream);

/* ------------------------------------------------------------------------------------------------
 * PDV second stage (detection/detzero_det/models/centerpoint_modules/pdv_head.py:269-637)
 * ------------------------------------------------------------------------------------------------ */
/* voxel_aggregation_utils.get_centroids_per_voxel_layer (:96-157): centroids of the points per voxel of the first feature
 * location's grid (voxel size already multiplied by its stride; grid = trunc((hi - lo) / vsize) as float32, :29-36) and, weighted
 * by the point counts, per voxel of a second location `scaling` times coarser.  points_b (n, 1+c) [b, x, y, z, f...];
 * cen (cap, 1+c) [b, mean x, y, z, f...], coords (cap, 4) int32 [b, z, y, x], counts (cap,), *d_m = number of voxels; rows come
 * in ascending (b, z, y, x) order (the order of torch.unique(dim=0) in the reference).  cen2 .. d_m2 may all be NULL. */
size_t dz_pdv_centroids_workspace_bytes(int n, int batch, int gx, int gy, int gz, int scaling, int cap1);
int dz_pdv_voxel_centroids(const float *points_b, int n, int c, const float *h_range6, const float *h_vsize3, const int *h_grid3,
                           int batch, int scaling, float *cen1, int *coords1, int *counts1, int *d_m1, int cap1, float *cen2,
                           int *coords2, int *counts2, int *d_m2, int cap2, void *ws, size_t ws_bytes, void *stream);
/* voxel_aggregation_utils.get_nonempty_voxel_feature_indices (:59-78) without the dense hash table: out[i] = row of cell
 * coords[i] = [b, z, y, x] in the sparse level (bitmap, prefix of dz_index_*), or -1. */
int dz_index_lookup(const int *coords, const int *d_n, int n, const uint32_t *bitmap, const uint32_t *prefix, int b, int d, int h,
                    int w, int layout, int *out, void *stream);
/* pointnet2_stack ball_query_count (src/ball_query_count_gpu.cu:16-62 + pointnet2_utils.py:78-83,186-189) for points that are
 * voxel centroids: at most one per cell of the (b, d, h, w) bitmap, listed in cell-key order (xyz (np, 3)).  Queries new_xyz
 * (mq, 3), `per_batch` consecutive queries per batch item.  idx (mq, nsample) int32: the first nsample points with d^2 < r^2 in
 * index order (relative to the batch item's first point), padded with the first hit, all 0 for an empty ball; cnt (mq,) hits. */
int dz_pdv_ball_query(const float *new_xyz, int mq, int per_batch, const float *xyz, const uint32_t *bitmap, const uint32_t *prefix,
                      int b, int d, int h, int w, const float *h_lo3, const float *h_vs3, float radius, int nsample, int *idx,
                      int *cnt, void *stream);
/* QueryAndGroup with use_xyz and use_density (pointnet2_utils.py:192-211, kde_utils.py:17-64, bandwidth 0.25): rows
 * (mq * nsample, row_stride) = [offset xyz, KDE density, the point's c features, zeros]. */
int dz_pdv_group_features(const float *new_xyz, int mq, int per_batch, const float *xyz, const float *feats, int c,
                          const uint32_t *bitmap, const uint32_t *prefix, int cells_per_batch, const int *idx, const int *cnt,
                          int nsample, float *rows, int row_stride, void *stream);
/* One branch of StackSAModuleMSGAttention in one kernel (csrc/pdv_sa.hip; pointnet2_modules.py:31-158): grouping as dz_pdv_group_features,
 * two point-wise layers (Conv2d 1x1 + folded BatchNorm + ReLU; w1 (cin_pad, ldw1) / w2 (h1, ldw2) fp32, rows = input channel, scale / shift
 * per output channel) and the max over the ball's nsample rows -> out (mq, h2) fp32.  Exact fp32 (v_mfma_f32_16x16x4_f32).  Instances
 * (dz_pdv_sa_pool_supported): nsample 16, (cin_pad, h1, h2) = (80, 32, 32) or (144, 64, 64), c % 4 == 0, c + 4 <= cin_pad, ReLU on both layers. */
int dz_pdv_sa_pool_supported(int c, int cin_pad, int h1, int h2, int nsample, int relu1, int relu2);
int dz_pdv_sa_pool(const float *new_xyz, int mq, int per_batch, const float *xyz, const float *feats, int c, const uint32_t *bitmap,
                   const uint32_t *prefix, int cells_per_batch, const int *idx, const int *cnt, int nsample, const float *w1, int ldw1,
                   const float *s1, const float *b1, int h1, const float *w2, int ldw2, const float *s2, const float *b2, int h2, int cin_pad,
                   float *out, void *stream);
/* The same branch on pair16 operands (math = DZ_MATH_F16X2 / DZ_MATH_BF16X2; the PDV head's split modes): w1 (h1, ldw1) / w2 (h2, ldw2) are
 * pair16 rows per OUTPUT channel (ld in channels, dz_pair16 packing), features stay fp32 rows (feat_rows x c) and are split in the
 * kernel.  Instances: (c, cin_pad, h1, h2) = (64, 80, 32, 32) and (128, 144, 64, 64), nsample 16.  Same outputs as dz_pdv_sa_pool up
 * to the pair16 product error (~2^-21 relative per product, fp32 accumulation). */
int dz_pdv_sa_pool_split_supported(int c, int cin_pad, int h1, int h2, int nsample, int relu1, int relu2);
int dz_pdv_sa_pool_split(const float *new_xyz, int mq, int per_batch, const float *xyz, const float *feats, long feat_rows, int c,
                         const uint32_t *bitmap, const uint32_t *prefix, int cells_per_batch, const int *idx, const int *cnt, int nsample,
                         const float *w1, int ldw1, const float *s1, const float *b1, int h1, const float *w2, int ldw2, const float *s2,
                         const float *b2, int h2, int cin_pad, int math, float *out, int ldo, void *stream);
/* (ldo: row stride of out in floats, >= h2 - the branches of a head write side by side into one (mq, sum of widths) tensor.  Grid points
 * whose ball is empty skip the arithmetic: their result is the layer stack applied to a zero row, computed once per wave.) */
/* density_utils.find_num_points_per_part_multi (:52-109) on points_in_multi_boxes (roiaware_pool3d_kernel.cu:377-404): counts
 * (batch, o, grid, grid, grid) int32 of the points (n, stride) [b, x, y, z, ...] per cell of every RoI (batch, o, 7), a point
 * counting for the first max_boxes RoIs (in RoI order) that contain it. */
int dz_pdv_part_counts(const float *points_b, int n, int stride, const float *rois, int batch, int o, int grid, int max_boxes,
                       int *counts, void *stream);
/* The same counts, bit for bit, with the RoIs binned on a 64 x 64 BEV grid first (a point then tests the handful of boxes whose
 * footprint circle reaches its cell instead of all o of its frame: ~0.5 ms -> tens of microseconds per 8 frames of 320k points).
 * ws: dz_pdv_part_counts_ws_bytes(batch, o) bytes of device scratch, 64-byte aligned (per frame: grid origin / scale, one 64-byte
 * line of box ids per cell, the staged box table).  More than 4096 RoIs per frame: runs dz_pdv_part_counts. */
size_t dz_pdv_part_counts_ws_bytes(int batch, int o);
int dz_pdv_part_counts_binned(const float *points_b, int n, int stride, const float *rois, int batch, int o, int grid, int max_boxes,
                              int *counts, void *ws, size_t ws_bytes, void *stream);
/* softmax(q k^T * scale + key padding mask) v for r independent sequences of l <= 256 tokens, one head of e <= 256 channels
 * (nn.MultiheadAttention core of attention_utils.TransformerEncoder; q, k, v, out (r, l, e) f32; mask (r, l) bytes or NULL). */
int dz_attention_single_head(const float *q, const float *k, const float *v, const unsigned char *key_padding_mask, int r, int l,
                             int e, float scale, float *out, void *stream);
/* The same layer with the key / value projections folded into its two GEMMs (one head): o' = softmax(q' x^T + mask) x over every group
 * of l consecutive rows, where x (r * l, e) pair16 are the layer's input rows (keys = values) and q' = x (Wq Wk^T) + Wk bq, scaled by
 * log2(e) / sqrt(E), (r * l, e) pair16; the caller applies (Wv Wo, bv Wo + bo) to o' (pdv_modules.py: PDVHead.attention).  e = 192,
 * l <= 224, math = DZ_MATH_F16X2 / DZ_MATH_BF16X2; out (r * l, e) pair16.  A fully masked group gives zero rows. */
/* The rest of the encoder layer around that attention, as two row-chain kernels (csrc/pdv_enc.hip; activations stay in registers):
 *   front: src = feats + (row_add ? W_p2 . ReLU(s0 * (W_p1 . pos_in) + b0) + b1 : 0), q = src . wq + uq          -> (rows, 192) pair16 each
 *   back:  x = LN1(src + op . wo + bo), y = LN2(x + w2 . ReLU(w1 . x + b1) + b2), out = pooled + (row_skip ? pooled : y) -> (rows, 192) fp32
 * Weights are pair16 rows per OUTPUT channel: w0 (96, 16), w1 (192, 96), wq / wo (192, 192), back w1 (128, 192), w2 (192, 128); pos_in
 * (rows, pin) fp32 with pin 4 or 8; feats / pooled (rows, 192) fp32; row_add / row_skip (rows) bytes; math = DZ_MATH_F16X2 / BF16X2. */
int dz_pdv_encoder_front(const float *pos_in, int pin, const float *feats, const unsigned char *row_add, long rows, const float *w0, const float *s0,
                         const float *b0, const float *w1, const float *b1, const float *wq, const float *uq, float *src, float *q, int math,
                         void *stream);
int dz_pdv_encoder_back(const float *op, const float *src, const float *pooled, const unsigned char *row_skip, long rows, const float *wo,
                        const float *bo, const float *g1, const float *be1, float eps1, const float *w1, const float *b1, const float *w2,
                        const float *b2, const float *g2, const float *be2, float eps2, float *out, int out_pair16, int math, void *stream);
/* (out_pair16: the result as pair16 rows - the operand of the head's FC stack - instead of fp32) */
int dz_self_attention_split_supported(int l, int e);
int dz_self_attention_split(const float *q, const float *x, const unsigned char *key_padding_mask, int r, int l, int e, float *out,
                            int math, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DETZERO_HIP_H */
