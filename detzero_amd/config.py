"""Minimal EasyDict-compatible config tree (easydict is not installed here).

Mirrors what the hot path needs of /root/reference/utils/detzero_utils/config_utils.py:59-94:
attribute access, ``.get``, yaml loading with ``_BASE_CONFIG_`` includes.
"""
import os

import yaml


class AttrDict(dict):
    def __init__(self, d=None, **kw):
        super().__init__()
        d = dict(d or {}, **kw)
        for k, v in d.items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, AttrDict):
            return AttrDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(AttrDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, AttrDict._wrap(v))

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def merge_new_config(config, new_config, base_dir='.'):
    """config_utils.py:59-76 (include path resolved against base_dir instead of the CWD)."""
    if '_BASE_CONFIG_' in new_config:
        path = new_config['_BASE_CONFIG_']
        if not os.path.isabs(path):
            path = os.path.join(base_dir, path)
        with open(path, 'r') as f:
            config.update(AttrDict(yaml.safe_load(f)))
    for key, val in new_config.items():
        if not isinstance(val, dict):
            config[key] = val
            continue
        if key not in config:
            config[key] = AttrDict()
        merge_new_config(config[key], val, base_dir)
    return config


def cfg_from_yaml_file(cfg_file, config=None, base_dir=None):
    config = AttrDict() if config is None else config
    base_dir = base_dir if base_dir is not None else os.path.dirname(os.path.dirname(os.path.abspath(cfg_file)))
    with open(cfg_file, 'r') as f:
        new_config = yaml.safe_load(f)
    merge_new_config(config, new_config, base_dir)
    return config


def centerpoint_1sweep_cfg(voxel_size=(0.1, 0.1, 0.15), max_voxels_test=200000):
    """In-code copy of the VALUES of tools/cfgs/det_model_cfgs/centerpoint_1sweep.yaml and
    det_dataset_cfgs/waymo_1sweep.yaml that the inference path reads."""
    return AttrDict({
        'CLASS_NAMES': ['Vehicle', 'Pedestrian', 'Cyclist'],
        'DATA_CONFIG': {
            'POINT_CLOUD_RANGE': [-75.2, -75.2, -2, 75.2, 75.2, 4.0],
            'DATA_PROCESSOR': [
                {'NAME': 'mask_points_and_boxes_outside_range', 'REMOVE_OUTSIDE_BOXES': True},
                {'NAME': 'shuffle_points', 'SHUFFLE_ENABLED': {'train': True, 'test': False}},
                {'NAME': 'transform_points_to_voxels', 'VOXEL_SIZE': list(voxel_size), 'MAX_POINTS_PER_VOXEL': 5,
                 'MAX_NUMBER_OF_VOXELS': {'train': 150000, 'test': max_voxels_test}},
            ],
        },
        'MODEL': {
            'NAME': 'CenterPoint',
            'SECOND_STAGE': False,
            'VFE': {'NAME': 'MeanVFE'},
            'BACKBONE_3D': {'NAME': 'VoxelResBackBone8x'},
            'MAP_TO_BEV': {'NAME': 'HeightCompression', 'NUM_BEV_FEATURES': 256},
            'BACKBONE_2D': {'NAME': 'BaseBEVBackbone', 'LAYER_NUMS': [5, 5], 'LAYER_STRIDES': [1, 2],
                            'NUM_FILTERS': [128, 256], 'UPSAMPLE_STRIDES': [1, 2],
                            'NUM_UPSAMPLE_FILTERS': [256, 256]},
            'DENSE_HEAD': {
                'NAME': 'CenterHead', 'CLASS_AGNOSTIC': False,
                'CLASS_NAMES_EACH_HEAD': [['Vehicle', 'Pedestrian', 'Cyclist']],
                'SHARED_CONV_CHANNEL': 64, 'USE_BIAS_BEFORE_NORM': True, 'NUM_HM_CONV': 2, 'IOU_WEIGHT': 1,
                'SEPARATE_HEAD_CFG': {
                    'HEAD_ORDER': ['center', 'center_z', 'dim', 'rot', 'iou'],
                    'HEAD_DICT': {'center': {'out_channels': 2, 'num_conv': 2},
                                  'center_z': {'out_channels': 1, 'num_conv': 2},
                                  'dim': {'out_channels': 3, 'num_conv': 2},
                                  'rot': {'out_channels': 2, 'num_conv': 2},
                                  'iou': {'out_channels': 1, 'num_conv': 2}},
                },
                'TARGET_ASSIGNER_CONFIG': {'FEATURE_MAP_STRIDE': 8, 'NUM_MAX_OBJS': 500, 'GAUSSIAN_OVERLAP': 0.1,
                                           'MIN_RADIUS': 2},
                'POST_PROCESSING': {
                    'SCORE_THRESH': 0.03, 'POST_CENTER_LIMIT_RANGE': [-80, -80, -10.0, 80, 80, 10.0],
                    'MAX_OBJ_PER_SAMPLE': 500,
                    'NMS_CONFIG': {'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.7, 'NMS_PRE_MAXSIZE': 4096,
                                   'NMS_POST_MAXSIZE': 500},
                },
            },
            'POST_PROCESSING': {'RECALL_THRESH_LIST': [0.3, 0.5, 0.7], 'SCORE_THRESH': 0.03,
                                'OUTPUT_RAW_SCORE': False, 'EVAL_METRIC': 'waymo'},
        },
    })


def centerpoint_3sweeps_cfg(voxel_size=(0.1, 0.1, 0.15)):
    """VALUES of det_model_cfgs/centerpoint_3sweeps.yaml + det_dataset_cfgs/waymo_3sweeps.yaml read on the inference path:
    the 1-sweep network with DynamicMeanVFE on 6 point features (x, y, z, intensity, elongation, offset), SWEEP_COUNT [-1, 1],
    voxelization left to the model (transform_points_to_voxels_placeholder, centerpoint_3sweeps.yaml:15-16), 400k voxels at test."""
    cfg = centerpoint_1sweep_cfg(voxel_size, max_voxels_test=400000)
    cfg.DATA_CONFIG.SWEEP_COUNT = [-1, 1]
    cfg.DATA_CONFIG.POINT_FEATURE_ENCODING = {
        'encoding_type': 'absolute_coordinates_encoding',
        'used_feature_list': ['x', 'y', 'z', 'intensity', 'elongation', 'offset'],
        'src_feature_list': ['x', 'y', 'z', 'intensity', 'elongation', 'offset']}
    cfg.DATA_CONFIG.DATA_PROCESSOR[2].NAME = 'transform_points_to_voxels_placeholder'
    cfg.MODEL.VFE.NAME = 'DynamicMeanVFE'
    return cfg


def centerpoint_pdv_cfg(voxel_size=(0.1, 0.1, 0.15)):
    """VALUES of det_model_cfgs/centerpoint_pdv_3sweeps.yaml read on the inference path: the 3-sweep detector with SECOND_STAGE and
    the PDVHead RoI head (voxel-centroid aggregation on x_conv3 / x_conv4, 6x6x6 RoI grid pooling with density features, one
    attention layer, FC heads)."""
    cfg = centerpoint_3sweeps_cfg(voxel_size)
    cfg.MODEL.SECOND_STAGE = True
    cfg.MODEL.ROI_HEAD = {
        'NAME': 'PDVHead', 'CLASS_AGNOSTIC': True, 'SHARED_FC': [256, 256], 'CLS_FC': [256, 256], 'REG_FC': [256, 256], 'DP_RATIO': 0.3,
        'NMS_CONFIG': {'TRAIN': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 9000, 'NMS_POST_MAXSIZE': 512, 'NMS_THRESH': 0.8},
                       'TEST': {'NMS_TYPE': 'nms_gpu', 'MULTI_CLASSES_NMS': False, 'NMS_PRE_MAXSIZE': 1024, 'NMS_POST_MAXSIZE': 512, 'NMS_THRESH': 0.7}},
        'VOXEL_AGGREGATION': {'NUM_FEATURES': [64, 128], 'FEATURE_LOCATIONS': ['x_conv3', 'x_conv4']},
        'ROI_GRID_POOL': {
            'FEATURE_LOCATIONS': ['x_conv3', 'x_conv4'], 'GRID_SIZE': 6,
            'POOL_LAYERS': {'x_conv3': {'MLPS': [[32, 32], [32, 32]], 'POOL_RADIUS': [0.8, 1.2], 'NSAMPLE': [16, 16], 'POOL_METHOD': 'max_pool', 'USE_DENSITY': True},
                            'x_conv4': {'MLPS': [[64, 64], [64, 64]], 'POOL_RADIUS': [1.2, 2.4], 'NSAMPLE': [16, 16], 'POOL_METHOD': 'max_pool', 'USE_DENSITY': True}},
            'ATTENTION': {'ENABLED': True, 'NUM_FEATURES': 192, 'NUM_HEADS': 1, 'NUM_HIDDEN_FEATURES': 128, 'NUM_LAYERS': 1,
                          'POSITIONAL_ENCODER': 'density_grid_points', 'MAX_NUM_BOXES': 20, 'DROPOUT': 0.1, 'COMBINE': True, 'MASK_EMPTY_POINTS': True}},
        'TARGET_CONFIG': {'BOX_CODER': 'ResidualCoder', 'ROI_PER_IMAGE': 128, 'FG_RATIO': 0.5, 'SAMPLE_ROI_BY_EACH_CLASS': True, 'CLS_SCORE_TYPE': 'roi_iou',
                          'CLS_FG_THRESH': 0.75, 'CLS_BG_THRESH': 0.25, 'CLS_BG_THRESH_LO': 0.1, 'HARD_BG_RATIO': 0.8, 'REG_FG_THRESH': 0.55},
        'LOSS_CONFIG': {'CLS_LOSS': 'BinaryCrossEntropy', 'REG_LOSS': 'smooth-l1', 'CORNER_LOSS_REGULARIZATION': True,
                        'LOSS_WEIGHTS': {'rcnn_cls_weight': 1.0, 'rcnn_reg_weight': 1.0, 'rcnn_corner_weight': 1.0, 'code_weights': [1.0] * 7}},
    }
    cfg.MODEL.POST_PROCESSING.NMS_CONFIG = {'MULTI_CLASSES_NMS': False, 'NMS_TYPE': 'nms_gpu', 'NMS_THRESH': 0.7, 'NMS_PRE_MAXSIZE': 4096, 'NMS_POST_MAXSIZE': 500}
    return cfg
