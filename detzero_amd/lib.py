"""ctypes binding of libdetzero_hip.so (the C ABI declared in include/detzero_hip.h).

PyTorch is used only for device memory and streams: every call takes raw device pointers
(``tensor.data_ptr()``) and the current HIP stream.  There is NO fallback: if the shared library
is missing or a call fails, a DetZeroHipError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdetzero_hip.so')


ERR_INVALID, ERR_WORKSPACE, ERR_HIP, ERR_UNSUPPORTED = -1, -2, -3, -4        # include/detzero_hip.h


class DetZeroHipError(RuntimeError):
    pass


c_int, c_float, c_size_t, c_void_p, c_double = ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_double
_I3 = c_int * 3
_F3 = c_float * 3
_F6 = c_float * 6
_I8 = c_int * 8


class Conv2dDesc(ctypes.Structure):
    """Mirror of ``dz_conv2d_desc`` (include/detzero_hip.h)."""
    _fields_ = [
        ('inp', c_void_p), ('out', c_void_p), ('w', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
        ('batch', c_int), ('ho', c_int), ('wo', c_int),
        ('in_hp', c_int), ('in_wp', c_int), ('in_cstride', c_int), ('in_coff', c_int), ('cin', c_int),
        ('kh', c_int), ('kw', c_int), ('stride', c_int), ('in_off', c_int),
        ('out_hp', c_int), ('out_wp', c_int), ('out_cstride', c_int), ('out_coff', c_int),
        ('out_sy', c_int), ('out_sx', c_int), ('out_dy', c_int), ('out_dx', c_int),
        ('groups', c_int), ('cout_pad', c_int),
        ('g_cout', _I8), ('g_ooff', _I8),
        ('relu', c_int),
        ('group_shift', c_void_p), ('group_rows', c_int), ('group_max', c_int), ('phase_groups', c_int),
        ('in_rowidx', c_void_p), ('in_row_channels', c_int), ('in_rows', c_int), ('in_tiles', c_void_p),
    ]


# name -> (restype, argtypes); must list every symbol of include/detzero_hip.h (tests check this)
_SIGS = {
    'dz_version': (ctypes.c_char_p, []),
    'dz_last_error': (ctypes.c_char_p, []),
    'dz_device_cu_count': (c_int, []),
    'dz_voxelize_hard_workspace_bytes': (c_size_t, [c_int] * 5),
    'dz_voxelize_hard': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dz_voxelize_hard_mean': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                      c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dz_voxelize_hard_batched_workspace_bytes': (c_size_t, [c_int] * 6),
    'dz_voxelize_hard_mean_batched': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                              c_int, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dz_voxelize_to_level_workspace_bytes': (c_size_t, [c_int] * 8),
    'dz_voxelize_to_level': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    'dz_mean_vfe': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'dz_voxelize_dynamic_workspace_bytes': (c_size_t, [c_int] * 7),
    'dz_voxelize_dynamic_mean': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                         c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'dz_index_words': (c_size_t, [c_int] * 5),
    'dz_index_workspace_bytes': (c_size_t, [c_int] * 5),
    'dz_index_from_coords': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dz_index_downsample': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t,
                                    c_void_p]),
    'dz_draw_subsets': (c_int, [c_void_p, c_int, c_int, ctypes.c_ulonglong, c_int, c_void_p, c_void_p]),
    'dz_grm_feature_channels': (c_int, [c_int]),
    'dz_grm_encode_points': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                     c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dz_prm_feature_channels': (c_int, [c_void_p, c_int]),
    'dz_prm_encode_points': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                     c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p]),
    'dz_tta_augment_points': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'dz_tta_restore_boxes': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dz_wbf_workspace_bytes': (c_size_t, [c_int, c_int]),
    'dz_wbf_fuse_3d': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_double,
                               c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dz_merge_sweeps_workspace_bytes': (c_size_t, [c_int]),
    'dz_merge_sweeps': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'dz_linear_forward_split': (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                        c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'dz_tile_masks_words': (c_int, [c_int]),
    'dz_build_neighbors': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dz_build_neighbors_packed': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dz_scatter_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'dz_spconv_forward': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_int, c_void_p, c_int, c_void_p]),
    'dz_sparse_to_bev': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                 c_void_p]),
    'dz_conv2d_forward': (c_int, [ctypes.POINTER(Conv2dDesc), c_void_p]),
    'dz_conv2d_variant': (ctypes.c_char_p, [ctypes.POINTER(Conv2dDesc)]),
    'dz_spconv_variant': (ctypes.c_char_p, [c_int, c_int]),
    'dz_pair16_from_f32': (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dz_pair16_to_f32': (c_int, [c_void_p, ctypes.c_long, c_int, c_int, c_void_p, c_void_p]),
    'dz_scatter_rows_split': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'dz_spconv_forward_split': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'dz_spconv_forward_split_packed': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'dz_spconv_tile_rows': (c_int, []),
    'dz_spconv_tile_info_words': (c_int, []),
    'dz_spconv_tile_table_entries': (c_int, []),
    'dz_build_tiles_halo_stride': (c_size_t, [c_int]),
    'dz_build_tiles': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dz_spconv_tiles_forward': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'dz_spconv_tiles_variant': (ctypes.c_char_p, [c_int, c_int]),
    'dz_spconv_x_tile_rows': (c_int, [c_int, c_int]),
    'dz_spconv_x_window_rows': (c_int, [c_int, c_int]),
    'dz_spconv_x_windows_words': (c_size_t, [c_int, c_int]),
    'dz_spconv_x_windows': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'dz_build_neighbors_packed_x': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int,
                                            c_void_p, c_void_p, c_void_p, c_void_p]),
    'dz_spconv_forward_split_x': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
    'dz_spconv_x_variant': (ctypes.c_char_p, [c_int, c_int]),
    'dz_bev_tile_list_words': (c_size_t, [c_int, c_int, c_int]),
    'dz_bev_tile_list': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'dz_bev_fill_empty_tiles': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'dz_bev_row_index': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dz_sparse_to_bev_split': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                       c_void_p]),
    'dz_sparse_to_bev_split_dense': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                             c_void_p]),
    'dz_conv2d_forward_split': (c_int, [ctypes.POINTER(Conv2dDesc), c_int, c_int, c_void_p]),
    'dz_conv2d_variant_split': (ctypes.c_char_p, [ctypes.POINTER(Conv2dDesc)]),
    'dz_spconv_variant_split': (ctypes.c_char_p, [c_int, c_int]),
    'dz_xattn_folded_supported': (c_int, [c_int, c_int, c_int]),
    'dz_xattn_folded_workspace_bytes': (ctypes.c_size_t, [c_int, c_int]),
    'dz_xattn_folded': (c_int, [c_void_p] * 6 + [c_int] * 5 + [ctypes.c_float, c_void_p, ctypes.c_size_t, c_void_p, c_void_p]),
    'dz_roi_bev_features': (c_int, [c_void_p, c_int, c_void_p, ctypes.c_long, ctypes.c_long, c_int, c_int, c_int] + [ctypes.c_float] * 4 + [c_int, c_void_p, c_void_p]),
    'dz_linear_splitk_workspace_bytes': (ctypes.c_size_t, [c_int, c_int, c_int]),
    'dz_linear_forward_splitk': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p,
                                         ctypes.c_size_t, c_void_p]),
    'dz_mha_core_split': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p, c_int, c_void_p]),
    'dz_pdv_sa_pool_supported': (c_int, [c_int] * 7),
    'dz_pdv_sa_pool': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                               c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dz_pdv_encoder_front': (c_int, [c_void_p, c_int, c_void_p, c_void_p, ctypes.c_long] + [c_void_p] * 9 + [c_int, c_void_p]),
    'dz_pdv_encoder_back': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_void_p]),
    'dz_self_attention_split_supported': (c_int, [c_int, c_int]),
    'dz_self_attention_split': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'dz_pdv_sa_pool_split_supported': (c_int, [c_int] * 7),
    'dz_pdv_sa_pool_split': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.c_long, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'dz_mlp_chain_forward': (c_int, [c_void_p, ctypes.c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int] + [c_void_p] * 10 + [c_int, c_void_p]),
    'dz_pointnet3_forward': (c_int, [c_void_p, ctypes.c_long] + [c_void_p] * 9 + [c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'dz_centerhead_decode_workspace_bytes': (c_size_t, [c_int] * 4),
    'dz_centerhead_decode': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    'dz_nms_workspace_bytes': (c_size_t, [c_int]),
    'dz_nms_rotated': (c_int, [c_void_p, c_void_p, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    'dz_nms_rotated_batched': (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                       c_void_p]),
    'dz_pack_detections': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dz_boxes_overlap_bev': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'dz_boxes_iou_bev': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'dz_gather_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dz_pdv_centroids_workspace_bytes': (c_size_t, [c_int] * 7),
    'dz_pdv_voxel_centroids': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'dz_index_lookup': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dz_pdv_ball_query': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_float,
                                  c_int, c_void_p, c_void_p, c_void_p]),
    'dz_pdv_group_features': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                      c_void_p, c_int, c_void_p]),
    'dz_pdv_part_counts': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dz_pdv_part_counts_ws_bytes': (ctypes.c_size_t, [c_int, c_int]),
    'dz_pdv_part_counts_binned': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    'dz_attention_single_head': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_void_p]),
    'dz_points_in_boxes_count': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dz_points_in_boxes_v2': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dz_crop_points_workspace_bytes': (c_size_t, [c_int] * 3),
    'dz_crop_points_in_boxes': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int, c_void_p, c_size_t, c_void_p]),
    'dz_mha_core': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                            c_void_p]),
    'dz_linear_forward': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                  c_int, c_int, c_void_p, c_int, c_void_p]),
    'dz_group_max': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'dz_rows_all_zero': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'dz_add_layernorm_combine': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    'dz_add_layernorm': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int, c_void_p, c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises DetZeroHipError when it is missing: the product path
    has no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DetZeroHipError(
            'libdetzero_hip.so not found at %s - build it with `python -m detzero_amd.build` '
            '(hipcc --offload-arch=gfx950); there is no CPU fallback' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what):
    if rc != 0:
        msg = load().dz_last_error()
        raise DetZeroHipError('%s failed (rc=%d): %s' % (what, rc, msg.decode() if msg else ''))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DetZeroHipError('expected a device tensor (the HIP path has no CPU fallback)')
        if t is not None and not t.is_contiguous():
            raise DetZeroHipError('expected a contiguous tensor')


def i3(v):
    return _I3(*[int(x) for x in v])


def f3(v):
    return _F3(*[float(x) for x in v])


def f6(v):
    return _F6(*[float(x) for x in v])
