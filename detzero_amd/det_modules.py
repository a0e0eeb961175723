"""CenterPoint modules under the reference's registry names, backed by libdetzero_hip.

Mirror of /root/reference/detection/detzero_det/models/centerpoint_modules/ (``__init__.py:8-17``):
same class names, constructor kwargs, ``forward(batch_dict) -> batch_dict`` contract, batch_dict
keys and ``state_dict()`` layout, so checkpoints of the reference load unchanged.
Underneath there is no spconv / cuDNN / torch_scatter: every module drives the HIP kernels through
``detzero_amd.ops``.  Inference only (the reference's training path - autograd through spconv,
target assignment, losses - is out of scope).

Internal data layout (differs from the reference on purpose):
  * sparse features: rows sorted by linear voxel key, 16-channel padded input;
  * dense activations: channel-last, zero-bordered images kept under private ``_nhwc_*`` keys of the
    batch_dict; the NCHW tensors the reference's keys promise are zero-copy permuted views.
"""
import contextlib
import os
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .lib import DetZeroHipError

PACKED_TABLES = bool(os.environ.get('DZ_TUNE_PACKED_TABLES'))     # opt-in: packed 27-tap tables for the small-channel levels (measured equal)

K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)


def _inference_only(module):
    if module.training:
        raise DetZeroHipError('%s: only the inference path is implemented on the HIP backend '
                              '(call .eval()); training is out of scope' % type(module).__name__)


def fold_bn(bn, conv_bias=None):
    """Eval-mode BatchNorm (+ preceding conv bias) as y = x*scale + shift, computed in fp64."""
    w = bn.weight.detach().double()
    b = bn.bias.detach().double()
    mean = bn.running_mean.detach().double()
    var = bn.running_var.detach().double()
    scale = w / torch.sqrt(var + bn.eps)
    shift = b - mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias.detach().double() * scale
    return scale.float().contiguous(), shift.float().contiguous()


_FROM_ENTRY = object()
# bumped whenever a module's lazily built device-side caches (kernel-layout weights, packed pairs, zero-response images) may be rebuilt by
# the next pass: concurrent sub-passes (centerpoint.FramePipeline._call_split) run ONE AFTER THE OTHER on the first pass of a generation,
# so that a cache one of them builds on its stream is complete before the other reads it on another stream
CACHE_GEN = [0]


class _Cached(nn.Module):
    """Modules that cache kernel-layout parameters; the cache is dropped when weights change."""

    def __init__(self):
        super().__init__()
        self._plan = None
        self.act_exp = None
        self.math = 0       # ops.MATH_MODES: 0 = fp32 MFMA, 1 = fp16-pair split, 2 = bf16-pair split, 3 = fp16 pairs with one product (csrc/hgemm.h)
        self.register_load_state_dict_post_hook(lambda module, incompatible: module.invalidate())

    def invalidate(self):
        self._plan = None
        CACHE_GEN[0] += 1

    def set_math(self, mode):
        if ops.math_id(mode) != self.math:
            CACHE_GEN[0] += 1
        self.math = ops.math_id(mode)
        return self

    def set_prescale(self, exps):
        """Per-stage power-of-two pre-scale of the activations kept as fp16 pairs: exps = {stage name: e} (centerpoint.select_prescale)
        or None.  A tensor of stage g is stored as value * 2^e_g; the factor is folded, exactly, into the folded-BN scale / shift of
        the layer that writes it (`_p`), so fp16 pairs keep their 22 bits whatever the checkpoint's activation range is."""
        self.act_exp = {k: int(v) for k, v in exps.items()} if exps else None
        CACHE_GEN[0] += 1
        return self

    def _e(self, name):
        """Exponent of stage `name` in the active math mode (0 unless the tensors are fp16 pairs and a pre-scale is set)."""
        if getattr(self, 'act_exp', None) is None or ops.storage_math(self.math) != 1:
            return 0
        return int(self.act_exp.get(name, 0))

    def _p(self, entry, e_in=0, e_out=0, key='w', scale=_FROM_ENTRY, shift=_FROM_ENTRY):
        """(weights, scale, shift) of a plan entry as the kernels of the active math mode take them.  fp32: the plan's own tensors.
        Split modes: weights packed as pairs (cached); for fp16 pairs with the per-output-channel pre-scale of ops.weight_prescale, and
        scale' = scale * 2^(e_out - e_in) / 2^e_c, shift' = shift * 2^e_out for an input stored at 2^e_in and an output at 2^e_out -
        all exact (powers of two).  scale=None in the plan (a layer without BatchNorm) becomes the vector of those factors."""
        scale = entry.get('scale') if scale is _FROM_ENTRY else scale
        shift = entry.get('shift') if shift is _FROM_ENTRY else shift
        if not self.math:
            return entry[key], scale, shift
        sm = ops.storage_math(self.math)
        if sm != 1:
            return self._w(entry, key), scale, shift
        cache = entry.setdefault('_pre', {})
        ck = (key, int(e_in), int(e_out), None if scale is None else scale.data_ptr(), None if shift is None else shift.data_ptr())
        if ck not in cache:
            wk = (key, 'w')
            if wk not in cache:
                ws, inv = ops.weight_prescale(entry[key])
                cache[wk] = (ops.pack_weight_split(ws, sm), inv)
            wp, inv = cache[wk]
            k = inv * (2.0 ** (int(e_out) - int(e_in)))
            ref = scale if scale is not None else shift
            if ref is not None and k.numel() != ref.numel():
                # per-group channel padding of the vectors (the head's output layer: 6 groups x 16 columns padded to 32)
                g = k.shape[0] if k.dim() > 1 else 1
                kk = k.reshape(g, -1)
                pad = ref.numel() // g - kk.shape[1]
                if pad < 0:
                    raise DetZeroHipError('_p: %d weight columns per group but %d scale / shift entries' % (kk.shape[1], ref.numel() // g))
                k = torch.cat([kk, kk.new_ones(g, pad)], dim=1)
            k = k.reshape(-1).contiguous()
            sc = (scale * k if scale is not None else k).contiguous()
            sh = None if shift is None else ((shift * (2.0 ** int(e_out))).contiguous() if e_out else shift)
            cache[ck] = (wp, sc, sh)
        return cache[ck]

    def _w(self, entry, key='w'):
        """Weights of a plan entry in the layout of the active math mode (split layouts packed once, cached)."""
        if not self.math:
            return entry[key]
        ck = '%s_split%d' % (key, ops.storage_math(self.math))
        if ck not in entry:
            entry[ck] = ops.pack_weight_split(entry[key], ops.storage_math(self.math))
        return entry[ck]

    def _apply(self, fn, *a, **kw):
        self._plan = None
        CACHE_GEN[0] += 1
        return super()._apply(fn, *a, **kw)


# ================================================================================================
# sparse containers / parameter holders (names and weight layouts of spconv.pytorch 2.x)
# ================================================================================================
class SparseConvTensor:
    """What the reference reads from spconv's tensor: features, indices, spatial_shape, batch_size,
    dense() (pdv_head.py:567-637, height_compression.py:21)."""

    def __init__(self, features, indices, spatial_shape, batch_size, level=None, padded=None, math=0):
        self._features = features  # fp32 rows, or None: decoded from the padded matrix on first use (most consumers never ask)
        self.indices = indices
        self.spatial_shape = list(spatial_shape)
        self.batch_size = batch_size
        self._level = level
        self._padded = padded      # (capacity-sized feature matrix, SparseLevel) for the fast path
        self._math = math          # encoding of the padded matrix (0 = fp32, else pair16); .features is always fp32
        # (the padded rows may carry the level's power-of-two pre-scale, SparseLevel.act_exp: removed on decoding)

    @property
    def features(self):
        if self._features is None:
            rows = self._padded[0][:self.indices.shape[0]]
            self._features = ops.level_rows_f32(rows, self._padded[1], self._math)
        return self._features

    @features.setter
    def features(self, value):
        self._features = value

    def feature_rows(self, rows):
        """features[rows] (rows: int64 indices) WITHOUT decoding the whole level: a pair16 tensor is gathered as raw 32-bit words and only
        the gathered rows are converted (the PDV head reads ~60 k of a level's ~1.7 M rows per 8 frames; round 4 decoded the level)."""
        if self._features is not None or self._padded is None or not self._math:
            return self.features[rows]
        return ops.level_rows_f32(self._padded[0][rows].contiguous(), self._padded[1], self._math)

    def replace_feature(self, new_features):
        return SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self._level)

    def dense(self):
        """(B, C, D, H, W) like spconv's .dense()."""
        c = self.features.shape[1]
        d, h, w = self.spatial_shape
        bev = ops.sparse_to_bev(self.features, self._level, c, pad=0)          # (B,H,W,C*D)
        return bev.view(self.batch_size, h, w, c, d).permute(0, 3, 4, 1, 2)


class _SparseConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None,
                 subm=False):
        super().__init__()
        k = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        s = (stride,) * 3 if isinstance(stride, int) else tuple(stride)
        p = (padding,) * 3 if isinstance(padding, int) else tuple(padding)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = k, s, p
        self.indice_key, self.subm = indice_key, subm
        # spconv 2.x implicit-GEMM layout: (Cout, kD, kH, kW, Cin)
        self.weight = nn.Parameter(torch.empty(out_channels, *k, in_channels))
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter('bias', None)

    def taps(self, cin_pad=None):
        """(kvol, Cin[padded], Cout) fp32 contiguous."""
        co, kd, kh, kw, ci = self.weight.shape
        w = self.weight.detach().permute(1, 2, 3, 4, 0).reshape(kd * kh * kw, ci, co)
        if cin_pad is not None and cin_pad > ci:
            w = torch.cat([w, w.new_zeros(w.shape[0], cin_pad - ci, co)], dim=1)
        return w.float().contiguous()


class SubMConv3d(_SparseConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        # SubMConv3d ignores stride/padding: always "same"
        super().__init__(in_channels, out_channels, kernel_size, 1, 1, bias, indice_key, subm=True)


class SparseConv3d(_SparseConvBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True, indice_key=None):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, bias, indice_key, subm=False)


class SparseSequential(nn.Sequential):
    pass


class SparseBasicBlock(nn.Module):
    """backbone3d.py:85-121."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, norm_fn=None, downsample=None, indice_key=None):
        super().__init__()
        assert norm_fn is not None
        self.conv1 = SubMConv3d(inplanes, planes, 3, stride=stride, padding=1, bias=True, indice_key=indice_key)
        self.bn1 = norm_fn(planes)
        self.relu = nn.ReLU()
        self.conv2 = SubMConv3d(planes, planes, 3, stride=stride, padding=1, bias=True, indice_key=indice_key)
        self.bn2 = norm_fn(planes)
        self.downsample = downsample
        self.stride = stride


def post_act_block(in_channels, out_channels, kernel_size, indice_key=None, stride=1, padding=0, conv_type='subm',
                   norm_fn=None):
    """backbone3d.py:64-83."""
    if conv_type == 'subm':
        conv = SubMConv3d(in_channels, out_channels, kernel_size, bias=False, indice_key=indice_key)
    elif conv_type == 'spconv':
        conv = SparseConv3d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=False,
                            indice_key=indice_key)
    else:
        raise NotImplementedError(conv_type)
    return SparseSequential(conv, norm_fn(out_channels), nn.ReLU())


# ================================================================================================
# VFE
# ================================================================================================
class MeanVFE(nn.Module):
    """vfe.py:58-83."""

    def __init__(self, model_cfg, num_point_features, **kwargs):
        super().__init__()
        self.num_point_features = num_point_features

    def get_output_feature_dim(self):
        return self.num_point_features

    def forward(self, batch_dict, **kwargs):
        voxels, num_points = batch_dict['voxels'], batch_dict['voxel_num_points']
        batch_dict['voxel_features'] = ops.mean_vfe(voxels.float().contiguous(), num_points.int().contiguous())
        return batch_dict


class DynamicMeanVFE(nn.Module):
    """vfe.py:86-147: voxelization on the device from ``batch_dict['points']`` (N,1+C)."""

    def __init__(self, model_cfg, num_point_features, voxel_size, grid_size, point_cloud_range, **kwargs):
        super().__init__()
        self.num_point_features = num_point_features
        self.voxel_size = [float(v) for v in voxel_size]
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.grid_size = [int(g) for g in grid_size]

    def get_output_feature_dim(self):
        return self.num_point_features

    @torch.no_grad()
    def forward(self, batch_dict, **kwargs):
        points = batch_dict['points'].float().contiguous()
        feats, coords = ops.voxelize_dynamic(points, self.point_cloud_range, self.voxel_size,
                                             batch_dict['batch_size'])
        batch_dict['voxel_features'] = feats
        batch_dict['voxel_coords'] = coords
        return batch_dict


# ================================================================================================
# sparse 3-D backbone
# ================================================================================================
class VoxelResBackBone8x(_Cached):
    """backbone3d.py:231-338.  21 sparse convolutions executed as fused
    gather -> MFMA -> (BN, bias, residual, ReLU) kernels on a bitmap-indexed sparse tensor."""

    CIN_PAD = 16

    def __init__(self, model_cfg, input_channels, grid_size, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        channels = list(model_cfg.get('CHANNELS', [16, 32, 64, 128])) if hasattr(model_cfg, 'get') else [16, 32, 64, 128]
        norm_fn = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        grid_size = [int(g) for g in grid_size]
        self.sparse_shape = [grid_size[2] + 1, grid_size[1], grid_size[0]]
        self.input_channels = input_channels

        self.conv_input = SparseSequential(
            SubMConv3d(input_channels, channels[0], 3, padding=1, bias=False, indice_key='subm1'),
            norm_fn(channels[0]), nn.ReLU())
        block = post_act_block
        self.conv1 = SparseSequential(
            SparseBasicBlock(channels[0], channels[0], norm_fn=norm_fn, indice_key='res1'),
            SparseBasicBlock(channels[0], channels[0], norm_fn=norm_fn, indice_key='res1'))
        self.conv2 = SparseSequential(
            block(channels[0], channels[1], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv2', conv_type='spconv'),
            SparseBasicBlock(channels[1], channels[1], norm_fn=norm_fn, indice_key='res2'),
            SparseBasicBlock(channels[1], channels[1], norm_fn=norm_fn, indice_key='res2'))
        self.conv3 = SparseSequential(
            block(channels[1], channels[2], 3, norm_fn=norm_fn, stride=2, padding=1, indice_key='spconv3', conv_type='spconv'),
            SparseBasicBlock(channels[2], channels[2], norm_fn=norm_fn, indice_key='res3'),
            SparseBasicBlock(channels[2], channels[2], norm_fn=norm_fn, indice_key='res3'))
        self.conv4 = SparseSequential(
            block(channels[2], channels[3], 3, norm_fn=norm_fn, stride=2, padding=(0, 1, 1), indice_key='spconv4', conv_type='spconv'),
            SparseBasicBlock(channels[3], channels[3], norm_fn=norm_fn, indice_key='res4'),
            SparseBasicBlock(channels[3], channels[3], norm_fn=norm_fn, indice_key='res4'))
        last_pad = model_cfg.get('last_pad', 0) if hasattr(model_cfg, 'get') else 0
        self.conv_out = SparseSequential(
            SparseConv3d(channels[3], channels[3], (3, 1, 1), stride=(2, 1, 1), padding=last_pad, bias=False,
                         indice_key='spconv_down2'),
            norm_fn(channels[3]), nn.ReLU())
        self.num_point_features = channels[3]
        # Row order of the sparse levels (ops.LAYOUT_*) and convolution engine of the split math modes.
        #   engine 'gather' + LAYOUT_LINEAR (default of rounds 1-3): one gather per (row, tap) pair (sparse_conv_h.hip, sparse_conv_w.h);
        #   engine 'tiles'  + LAYOUT_BRICK: tile-resident inputs (sparse_conv_t.hip) - built and parity-tested in round 3, measured
        #   SLOWER inside the detector on MI355X (743 vs 865 frames/s), so it is opt-in: set_sparse_engine(model, 'tiles').
        #   engine 'xrun' + LAYOUT_LINEAR (default since round 4: +7 % frames/s, A/B on one box): the submanifold convolutions of the
        #   32 / 64 / 128-channel levels stage each z slab's window of input rows once per tile (sparse_conv_x.hip, packed tables +
        #   ops.build_windows); every other convolution as 'gather'.
        self.layout = int(os.environ.get('DZ_TUNE_LAYOUT', ops.LAYOUT_LINEAR))          # (environment: A/B runs on one box)
        self.engine = os.environ.get('DZ_TUNE_SPCONV_ENGINE', 'xrun')
        # output widths whose convolutions run on the tile engine when it is selected (the others keep the gather kernels)
        self.tile_couts = tuple(int(c) for c in os.environ.get('DZ_TUNE_SPCONV_TILE_COUTS', '16,32,64,128').split(',') if c)
        self.backbone_channels = {'x_conv1': channels[0], 'x_conv2': channels[1], 'x_conv3': channels[2],
                                  'x_conv4': channels[3]}
        self.channels = channels

    def set_engine(self, engine):
        """'gather' (rows in the canonical linear-key order), 'xrun' (gather + the z-slab window kernel for the submanifold
        convolutions of the 32 / 64 / 128-channel levels; same row order) or 'tiles' (tile-resident convolution; rows in brick order)."""
        if engine not in ('gather', 'tiles', 'xrun'):
            raise DetZeroHipError('unknown sparse engine %r (gather | xrun | tiles)' % (engine,))
        if engine == 'tiles' and ops.L.load().dz_spconv_tile_rows() == 0:
            raise DetZeroHipError("sparse engine 'tiles' is an experimental build option (measured slower, DESIGN.md 2d): rebuild the library "
                                  'with DZ_BUILD_EXPERIMENTAL=1 (python -m detzero_amd.build --force)')
        self.engine = engine
        self.layout = ops.LAYOUT_BRICK if engine == 'tiles' else ops.LAYOUT_LINEAR

    # ---- kernel-layout parameters -------------------------------------------------------------
    def plan(self):
        if self._plan is not None:
            return self._plan

        def conv_bn(conv, bn, cin_pad=None):
            scale, shift = fold_bn(bn, conv.bias)
            return {'w': conv.taps(cin_pad), 'scale': scale, 'shift': shift, 'k': conv.kernel_size,
                    's': conv.stride, 'p': conv.padding}

        def blk(b):
            return (conv_bn(b.conv1, b.bn1), conv_bn(b.conv2, b.bn2))

        p = {'conv_input': conv_bn(self.conv_input[0], self.conv_input[1], self.CIN_PAD)}
        p['conv1'] = [blk(self.conv1[0]), blk(self.conv1[1])]
        for name in ('conv2', 'conv3', 'conv4'):
            seq = getattr(self, name)
            p[name] = {'down': conv_bn(seq[0][0], seq[0][1]), 'blocks': [blk(seq[1]), blk(seq[2])]}
        p['conv_out'] = conv_bn(self.conv_out[0], self.conv_out[1])
        self._plan = p
        return p

    # ---- execution ------------------------------------------------------------------------------
    def _res_block(self, x, nbr, level, params, e=0):
        """(all tensors of a level - block inputs, hidden activations, the residual - share the level's exponent e)"""
        c1, c2 = params
        y = ops.spconv_forward(x, nbr, level, *self._p(c1, e, e), None, True, math=self.math)
        return ops.spconv_forward(y, nbr, level, *self._p(c2, e, e), x, True, math=self.math)

    def build_pyramid(self, voxel_features, voxel_coords, batch_size, d_n=None, overlap=True, side_key=0, caps=None, level1=None,
                      staggered=False, exact=False):
        """Everything of the backbone that depends only on voxel COORDINATES: the level-1 index + feature scatter and
        the output sets / bitmaps / neighbour tables of every stage.  Returns {'x': level-1 rows, 'steps': [...]}.

        overlap=True builds the tables of the deeper stages on a side stream (inside a captured graph: a parallel
        branch) so that they run under the convolutions of the earlier stages; events order each stage's tables
        before their first use.  overlap=False keeps everything on the current stream (used when the whole
        preparation stage is itself overlapped with the previous batch, see StreamingDetector).

        staggered=True (with overlap): only level 1 is built here; `run_pyramid` asks for the index of stage i + 1 when it
        starts the convolutions of stage i (pyr['build_stage']), gated on the main stream's progress - each stage's ~0.1-0.3 ms
        of index kernels then runs under the convolutions of the stage before it instead of all of them crowding under the
        memory-bound level-1 convolutions.

        caps: optional row capacities of the 4 strided stages (conv2, conv3, conv4, conv_out).  The default is the
        worst case (min(cells, 8 x inputs)), which is safe but grows with the batch; calibrated capacities
        (FramePipeline.calibrate) keep large batches inside the 2 GiB buffer-addressing window.  Rows beyond a
        capacity are DROPPED by the kernels; pyr['overflow'] (device bool) reports it.

        exact=True (the plugin path, `run`): every strided stage's row count is read back (one host sync per stage) and the level
        shrunk to it before its tables and feature rows are sized - the worst case grows 8x per stage and passes the 2 GiB
        buffer-addressing window at ~4 frames of 160k points; exact sizes take dozens."""
        p = self.plan()
        dev = voxel_features.device
        if level1 is None:
            n = voxel_features.shape[0]
            lvl1 = ops.SparseLevel(batch_size, self.sparse_shape, max(n, 1), dev, layout=self.layout)
            rank = lvl1.build_from_coords(voxel_coords, d_n)
        else:
            # ops.voxelize_to_level already produced the level-1 index and its feature rows (voxel_features = those rows)
            lvl1, rank = level1, None
        main = torch.cuda.current_stream(dev)
        side = self._side_stream(dev, side_key) if overlap else main
        if overlap:
            side.wait_stream(main)
        steps = []          # (down neighbour table or None, same-level table or None, level, ready event or None)
        tiled = self.engine == 'tiles' and self.math != 0

        # opt-in (PACKED_TABLES): tables read only by the small-channel split-math kernels (<= 32 output channels: conv_input, conv1,
        # conv2) built packed - a third of the words; the index chain gains what the decode costs the convolutions
        pack = PACKED_TABLES and self.math != 0 and not tiled and self.layout == 0 and os.environ.get('DZ_TUNE_SPCONV_W', '1') != '0'

        xrun = self.engine == 'xrun' and self.math != 0 and self.layout == 0
        xrun_couts = tuple(int(c) for c in os.environ.get('DZ_TUNE_XRUN_COUTS', '32,64,128').split(',') if c)

        def table(src, dst, k, s, p, cout):
            # (asked BEFORE the packed table is built: a width / capacity the x-run kernel does not cover would otherwise build the
            # table twice - packed for nothing, then plain)
            if (xrun and src is dst and cout in xrun_couts and tuple(k) == (3, 3, 3) and dst.cap < (1 << 29)
                    and ops.L.load().dz_spconv_x_tile_rows(int(cout), int(cout)) != 0):
                # submanifold table of a level the x-run kernel covers: packed words + the tiles' windows
                nbr = ops.neighbors_xrun(dst, cout)          # (one launch: packed table + windows + tap-set order)
                if getattr(nbr, 'xwin', None) is not None:
                    return nbr
            nbr = src.neighbors_to(dst, k, s, p, packed=pack and cout <= 32)
            return ops.build_tiles(nbr, dst) if tiled and cout in self.tile_couts and k[0] * k[1] * k[2] >= 3 else nbr

        def hand_over(step):                        # tensors born on the side stream are consumed on the main stream
            nbr_d, nbr_s, lvl, _ = step
            # (the level's bitmap / prefix / workspace too: the dense BEV hand-over and the PDV lookups read them on the main stream)
            for t in (nbr_d, nbr_s, lvl.coords, lvl.d_m, lvl.bitmap, lvl.prefix, lvl.ws):
                if t is not None:
                    t.record_stream(main)
                    if getattr(t, 'tile_masks', None) is not None:
                        t.tile_masks.record_stream(main)
                    for tt in getattr(t, 'tiles', None) or ():
                        tt.record_stream(main)
                    if getattr(t, 'xwin', None) is not None:
                        for tt in t.xwin:
                            if torch.is_tensor(tt):
                                tt.record_stream(main)
        ch = self.channels
        with torch.cuda.stream(side):
            nbr1 = table(lvl1, lvl1, K3, S1, P1, ch[0])
            steps.append((None, nbr1, lvl1, side.record_event() if overlap else None))
        x = voxel_features if level1 is not None else ops.scatter_rows(voxel_features, rank, self.CIN_PAD, lvl1.cap, d_n, math=self.math)
        pyr = {'x': x, 'steps': steps}
        names = ('conv2', 'conv3', 'conv4', 'conv_out')

        def build_stage(li, gate=None):
            """Index of stage names[li] (output set, strided table, its submanifold table) on the side stream; gate = an event of the
            main stream the work waits for."""
            name = names[li]
            level = steps[li][2]
            with torch.cuda.stream(side):
                if gate is not None:
                    side.wait_event(gate)
                dp = p[name]['down'] if name != 'conv_out' else p[name]
                nxt = level.downsample(dp['k'], dp['s'], dp['p'], cap=None if caps is None else int(caps[li]))
                if exact and caps is None:
                    m = nxt.num_active()                    # (host sync) rows of the level: its buffers need no more
                    nxt.cap = max(int(m), 1)
                    nxt.coords = nxt.coords[:nxt.cap]
                co = ch[min(li + 1, 3)]
                nbr_d = table(level, nxt, dp['k'], dp['s'], dp['p'], co)
                nbr_s = table(nxt, nxt, K3, S1, P1, co) if name != 'conv_out' else None
                if name == 'conv_out' and caps is not None:
                    # overflow flag of the calibrated capacities (same stream as the index build: it is covered by
                    # the last stage's event, so the main stream does not wait for the whole pyramid up front)
                    lv = [st[2] for st in steps[1:]] + [nxt]
                    key = (tuple(l.cap for l in lv), str(dev))
                    cache = self.__dict__.setdefault('_cap_limits', {})
                    if key not in cache:    # created on the first (eager, warm-up) call: no H2D copy inside a graph capture
                        cache[key] = torch.tensor(key[0], dtype=torch.int32, device=dev)
                    pyr['overflow'] = (torch.cat([l.d_m for l in lv]) > cache[key]).any()
                steps.append((nbr_d, nbr_s, nxt, side.record_event() if overlap else None))
            if overlap:
                hand_over(steps[-1])
                if 'overflow' in pyr and name == 'conv_out':
                    pyr['overflow'].record_stream(main)
        if overlap:
            hand_over(steps[0])
        if staggered and overlap:
            pyr['build_stage'] = build_stage
        else:
            for li in range(4):
                build_stage(li)
        return pyr

    def run_pyramid(self, pyr):
        """The 21 sparse convolutions over a prepared pyramid.  Returns dict of (features, SparseLevel)."""
        p = self.plan()
        mm = self.math
        steps = pyr['steps']
        x = pyr['x']
        main = torch.cuda.current_stream(x.device)
        build_stage = pyr.pop('build_stage', None)          # staggered pyramid: stage i + 1's index is requested when stage i's convs start

        def ready(ev):
            if ev is not None:
                main.wait_event(ev)
        if build_stage is not None:
            build_stage(0)                                  # conv2's index under the level-1 convolutions
        _, nbr, lvl1, ev = steps[0]
        ready(ev)
        ci = p['conv_input']
        e = self._e('x_conv1')                                # (the voxel features themselves are stored unscaled)
        x = ops.spconv_forward(x, nbr, lvl1, *self._p(ci, 0, e), None, True, math=mm)
        for bp in p['conv1']:
            x = self._res_block(x, nbr, lvl1, bp, e)
        lvl1.act_exp = e
        out = {'x_conv1': (x, lvl1)}
        level = lvl1
        for i, name in enumerate(('conv2', 'conv3', 'conv4')):
            if build_stage is not None:
                build_stage(i + 1, gate=main.record_event())       # the next stage's index under this stage's convolutions
            dp = p[name]['down']
            nbr_d, nbr, nxt, ev = steps[i + 1]
            ready(ev)
            e_prev, e = e, self._e('x_conv%d' % (i + 2))
            x = ops.spconv_forward(x, nbr_d, nxt, *self._p(dp, e_prev, e), None, True, in_level=level, math=mm)
            for bp in p[name]['blocks']:
                x = self._res_block(x, nbr, nxt, bp, e)
            nxt.act_exp = e
            out['x_conv%d' % (i + 2)] = (x, nxt)
            level = nxt
        dp = p['conv_out']
        nbr_d, _, nxt, ev = steps[4]
        ready(ev)
        e_prev, e = e, self._e('encoded')
        x = ops.spconv_forward(x, nbr_d, nxt, *self._p(dp, e_prev, e), None, True, in_level=level, math=mm)
        nxt.act_exp = e
        out['encoded'] = (x, nxt)
        return out

    def run(self, voxel_features, voxel_coords, batch_size, d_n=None, side_key=0, exact=False):
        """Capacity-sized execution without host syncs (exact=True: exact-sized levels, one sync per strided stage).  Returns dict of
        (features, SparseLevel)."""
        return self.run_pyramid(self.build_pyramid(voxel_features, voxel_coords, batch_size, d_n, True, side_key, exact=exact))

    def _side_stream(self, dev, key=0):
        pool = self.__dict__.setdefault('_side_streams', {})
        k = (str(dev), key)
        if k not in pool:
            # (DZ_TUNE_SIDE_PRIORITY: development knob, -1 = high-priority queue for the index pyramid)
            pool[k] = torch.cuda.Stream(device=dev, priority=int(os.environ.get('DZ_TUNE_SIDE_PRIORITY', '0')))
        return pool[k]

    def forward(self, batch_dict):
        _inference_only(self)
        vf = batch_dict['voxel_features'].float().contiguous()
        vc = batch_dict['voxel_coords'].int().contiguous()
        batch_size = batch_dict['batch_size']
        with torch.no_grad():
            res = self.run(vf, vc, batch_size, exact=True)        # (the plugin path reads the row counts back anyway: as_tensor below)

        def as_tensor(item):
            feats, level = item
            m = level.num_active()
            return SparseConvTensor(None, level.coords[:m], level.shape, batch_size, level=level, padded=(feats, level), math=self.math)
        batch_dict.update({'encoded_spconv_tensor': as_tensor(res['encoded']), 'encoded_spconv_tensor_stride': 8})
        batch_dict.update({'multi_scale_3d_features': {k: as_tensor(res[k]) for k in ('x_conv1', 'x_conv2', 'x_conv3', 'x_conv4')}})
        batch_dict.update({'multi_scale_3d_strides': {'x_conv1': 1, 'x_conv2': 2, 'x_conv3': 4, 'x_conv4': 8}})
        return batch_dict


# ================================================================================================
# map to BEV
# ================================================================================================
class HeightCompression(nn.Module):
    """height_compression.py:4-26."""

    def __init__(self, model_cfg, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_bev_features = self.model_cfg.NUM_BEV_FEATURES

    def forward(self, batch_dict):
        t = batch_dict['encoded_spconv_tensor']
        feats, level = t._padded if t._padded is not None else (t.features, t._level)
        math = t._math if t._padded is not None else 0
        c = feats.shape[1]
        bev = ops.sparse_to_bev(feats, level, c, pad=1, math=math)      # (B, H+2, W+2, C*D)
        e = int(getattr(level, 'act_exp', 0) or 0) if t._padded is not None else 0
        batch_dict['_nhwc_spatial_features'] = bev
        batch_dict['_nhwc_math'] = math                                 # encoding of the private channel-last images
        batch_dict['_nhwc_exp'] = e                                     # ... and their power-of-two pre-scale (values * 2^e)
        plain = ops.pair16_to_f32(bev, math) if math else bev
        if e:
            plain = plain * (2.0 ** -e)
        batch_dict['spatial_features'] = plain[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)   # NCHW view
        batch_dict['spatial_features_stride'] = batch_dict['encoded_spconv_tensor_stride']
        return batch_dict


# ================================================================================================
# dense BEV backbone
# ================================================================================================
def _conv_weight_taps(weight, cout_pad=None):
    """torch Conv2d weight (Cout,Cin,kh,kw) -> (kh*kw, Cin, Cout_pad)."""
    co, ci, kh, kw = weight.shape
    w = weight.detach().permute(2, 3, 1, 0).reshape(kh * kw, ci, co).float()
    if cout_pad is not None and cout_pad > co:
        w = torch.cat([w, w.new_zeros(kh * kw, ci, cout_pad - co)], dim=2)
    return w.contiguous()


def _pad_vec(v, n, fill=0.0):
    if v.numel() == n:
        return v.contiguous()
    out = v.new_full((n,), fill)
    out[:v.numel()] = v
    return out


# Zero-bordered activation images are written only in their interior, so inside a FramePipeline step (which owns a
# workspace dict and runs its dense stage on one stream) they are allocated and zeroed ONCE and reused by every later
# step instead of being re-zeroed (~0.5 ms of memsets per 16-frame step).  Outside a workspace (module forward(),
# whose outputs the caller may keep) every call gets fresh buffers.
_WORKSPACE = [None]


@contextlib.contextmanager
def workspace(bufs):
    prev = _WORKSPACE[0]
    _WORKSPACE[0] = bufs
    try:
        yield
    finally:
        _WORKSPACE[0] = prev


def bordered_zeros(name, shape, dev):
    ws = _WORKSPACE[0]
    if ws is None:
        return torch.zeros(shape, dtype=torch.float32, device=dev)
    key = (name, tuple(shape), str(dev))
    buf = ws.get(key)
    if buf is None:
        buf = ws[key] = torch.zeros(shape, dtype=torch.float32, device=dev)
    return buf


def nchw_to_padded_nhwc(x, pad=1):
    b, c, h, w = x.shape
    out = x.new_zeros((b, h + 2 * pad, w + 2 * pad, c))
    out[:, pad:pad + h, pad:pad + w, :] = x.permute(0, 2, 3, 1)
    return out


def _recode(img, enc, math, e_src=0, e_dst=0):
    """Channel-last image from encoding `enc` (0 = fp32, else pair16 of that mode) at pre-scale 2^e_src to the encoding of `math` at
    2^e_dst."""
    if ops.storage_math(enc) == ops.storage_math(math) and e_src == e_dst:
        return img
    plain = ops.pair16_to_f32(img, enc) if enc else img
    if e_src != e_dst:
        plain = plain * (2.0 ** (e_dst - e_src))
    return ops.pair16_from_f32(plain, math=math) if math else plain


SKIP_EMPTY_TILES = os.environ.get('DZ_TUNE_SKIP_EMPTY_TILES', '1') != '0'  # development switch: 0 = the sparse-input convolution runs every pixel tile
SPARSE_BEV_INPUT = os.environ.get('DZ_TUNE_SPARSE_BEV', '1') != '0'     # development switch: 0 = dense BEV image (r01-r04)
FIRST_BLOCK_WIDE_GROUPS = os.environ.get('DZ_TUNE_FIRST_BLOCK_WIDE', '1') != '0'     # development switch: 0 = the first BEV block in the concatenation's frame groups (r05)
FUSED_DEBLOCK_PHASES = os.environ.get('DZ_TUNE_DEBLOCK_PHASES', '1') != '0'     # development switch: 0 = one launch per phase (r01-r04)


def conv_layer(inp, in_shape, w, scale, shift, relu, out, out_shape, *, cin, in_cstride, in_coff=0, ksize=3,
               stride=1, in_off=0, out_cstride, out_coff=0, out_s=1, out_d=(0, 0), groups=1, cout_pad=None,
               g_cout=None, g_ooff=None, ho=None, wo=None, batch=1, math=0, out_f32=False, phase_groups=False, in_rowidx=None,
               in_row_channels=0, in_rows=0, in_tiles=None):
    """One dz_conv2d_forward[_split] call.  in_shape/out_shape = (Hp, Wp) of the (padded) images.
    math != 0: w is the pack_weight_split layout (..., cout_pad, cin)."""
    if cout_pad is None:
        cout_pad = w.shape[-2] if math else w.shape[-1]
    ops.conv2d(dict(
        inp=inp.data_ptr(), out=out.data_ptr(), w=w.data_ptr(),
        scale=scale.data_ptr() if scale is not None else None,
        shift=shift.data_ptr() if shift is not None else None,
        batch=batch, ho=ho, wo=wo, in_hp=in_shape[0], in_wp=in_shape[1], in_cstride=in_cstride, in_coff=in_coff,
        cin=cin, kh=ksize, kw=ksize, stride=stride, in_off=in_off,
        out_hp=out_shape[0], out_wp=out_shape[1], out_cstride=out_cstride, out_coff=out_coff,
        out_sy=out_s, out_sx=out_s, out_dy=out_d[0], out_dx=out_d[1],
        groups=groups, cout_pad=cout_pad,
        g_cout=g_cout if g_cout is not None else [cout_pad], g_ooff=g_ooff if g_ooff is not None else [0],
        relu=1 if relu else 0, phase_groups=1 if phase_groups else 0,
        in_rowidx=in_rowidx.data_ptr() if in_rowidx is not None else None, in_row_channels=int(in_row_channels), in_rows=int(in_rows),
        in_tiles=in_tiles.data_ptr() if in_tiles is not None else None),
        math=math, out_f32=out_f32, tiles=in_tiles)


class BaseBEVBackbone(_Cached):
    """backbone2d.py:6-120 for the layouts used by every DetZero config: per level
    ZeroPad+Conv3x3(stride s)+BN+ReLU, n x [Conv3x3+BN+ReLU], and a ConvTranspose2d(k=s) deblock."""

    def __init__(self, model_cfg, input_channels):
        super().__init__()
        self.model_cfg = model_cfg
        layer_nums = list(model_cfg.LAYER_NUMS)
        layer_strides = list(model_cfg.LAYER_STRIDES)
        num_filters = list(model_cfg.NUM_FILTERS)
        upsample_strides = list(model_cfg.UPSAMPLE_STRIDES)
        num_upsample_filters = list(model_cfg.NUM_UPSAMPLE_FILTERS)
        assert len(layer_nums) == len(layer_strides) == len(num_filters) == len(upsample_strides)
        c_in_list = [input_channels, *num_filters[:-1]]
        self.blocks = nn.ModuleList()
        self.deblocks = nn.ModuleList()
        for idx in range(len(layer_nums)):
            cur = [nn.ZeroPad2d(1),
                   nn.Conv2d(c_in_list[idx], num_filters[idx], kernel_size=3, stride=layer_strides[idx], padding=0, bias=False),
                   nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()]
            for _ in range(layer_nums[idx]):
                cur.extend([nn.Conv2d(num_filters[idx], num_filters[idx], kernel_size=3, padding=1, bias=False),
                            nn.BatchNorm2d(num_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()])
            self.blocks.append(nn.Sequential(*cur))
            s = upsample_strides[idx]
            if not (isinstance(s, int) and s >= 1):
                raise DetZeroHipError('BaseBEVBackbone: only integer UPSAMPLE_STRIDES >= 1 are supported')
            self.deblocks.append(nn.Sequential(
                nn.ConvTranspose2d(num_filters[idx], num_upsample_filters[idx], s, stride=s, bias=False),
                nn.BatchNorm2d(num_upsample_filters[idx], eps=1e-3, momentum=0.01), nn.ReLU()))
        self.num_bev_features = sum(num_upsample_filters)
        self.layer_strides, self.upsample_strides = layer_strides, upsample_strides
        self.num_filters, self.num_upsample_filters = num_filters, num_upsample_filters
        self.input_channels = input_channels

    def plan(self):
        if self._plan is not None:
            return self._plan
        levels = []
        for idx, blk in enumerate(self.blocks):
            convs = []
            mods = list(blk)
            i = 1
            while i < len(mods):
                conv, bn = mods[i], mods[i + 1]
                scale, shift = fold_bn(bn, conv.bias)
                convs.append({'w': _conv_weight_taps(conv.weight), 'scale': scale, 'shift': shift,
                              'stride': conv.stride[0], 'cin': conv.in_channels, 'cout': conv.out_channels})
                if idx == 0 and len(convs) == 1 and conv.in_channels % 2 == 0:
                    # HeightCompression's channel order is c * D + z (height_compression.py:22); the sparse-input convolution reads
                    # the two slabs' rows one after the other (z * C + c): the same weights, input channels permuted
                    wt_ = convs[0]['w']
                    c_half = conv.in_channels // 2
                    convs[0]['w_zmajor'] = wt_.view(wt_.shape[0], c_half, 2, wt_.shape[2]).permute(0, 2, 1, 3).reshape(wt_.shape).contiguous()
                i += 3
            de, dbn = self.deblocks[idx][0], self.deblocks[idx][1]
            scale, shift = fold_bn(dbn, de.bias)
            s = de.stride[0]
            wt = de.weight.detach().float()                            # (Cin, Cout, s, s)
            phases = [[{'w': wt[:, :, dy, dx].contiguous().unsqueeze(0).contiguous()} for dx in range(s)] for dy in range(s)]
            # (s*s, Cin, Cout): phase g = (dy, dx) = (g // s, g % s) - the group order of dz_conv2d_desc.phase_groups
            w_phases = wt.permute(2, 3, 0, 1).reshape(s * s, wt.shape[0], wt.shape[1]).contiguous()
            levels.append({'convs': convs, 'de': {'phases': phases, 'w_phases': w_phases, 'scale': scale, 'shift': shift, 's': s,
                                                  'cin': de.in_channels, 'cout': de.out_channels}})
        self._plan = levels
        return levels

    def sparse_input_ok(self, row_channels, slabs):
        """The first block's convolution can read a two-slab sparse level directly (dz_conv2d_desc.in_rowidx) instead of the dense
        BEV image: split math, 3 x 3 stride 1, 128-channel output tiles, input = 2 x row_channels."""
        cv = self.plan()[0]['convs'][0]
        return bool(self.math and SPARSE_BEV_INPUT and slabs == 2 and cv['stride'] == 1 and cv['cin'] == 2 * row_channels and
                    row_channels % 32 == 0 and cv['cout'] % 128 == 0)

    def _zero_response(self, lvl, h, w, rows_c, dev, nb0):
        """Outputs of the first block's layers on an ALL-ZERO input, one (1, H + 2, W + 2, C) pair16 image per layer - what every pixel far
        enough from any data computes (dz_bev_tile_list).  Computed once per (shape, math, frames per launch) with the detector's own
        kernels on nb0 all-zero frames: which kernel runs a layer depends on the launch's tile count, and the images must carry ITS bits."""
        e_in, e_mid = self._e('encoded'), self._e('spatial_features_2d')
        key = ('zero_resp', h, w, int(self.math), str(dev), int(nb0), e_in, e_mid)
        if key not in lvl:
            ridx = torch.full((nb0, h + 2, w + 2, 2), -1, dtype=torch.int32, device=dev)
            rows = torch.zeros((8, rows_c), dtype=torch.float32, device=dev)
            outs = []
            x, xh, xw, xc = None, h, w, 2 * rows_c
            with workspace(None):
                for ci, cv in enumerate(lvl['convs']):
                    if cv['stride'] != 1:
                        raise DetZeroHipError('BaseBEVBackbone: zero-response tiles need a stride-1 first block')
                    y = torch.zeros((nb0, xh + 2, xw + 2, cv['cout']), dtype=torch.float32, device=dev)
                    if ci == 0:
                        conv_layer(rows, (xh + 2, xw + 2), *self._p(cv, e_in, e_mid, key='w_zmajor'), True, y, (xh + 2, xw + 2), cin=cv['cin'],
                                   in_cstride=xc, ksize=3, stride=1, in_off=0, out_cstride=cv['cout'], out_d=(1, 1), ho=xh, wo=xw, batch=nb0,
                                   math=self.math, in_rowidx=ridx, in_row_channels=rows_c, in_rows=rows.shape[0])
                    else:
                        conv_layer(x, (xh + 2, xw + 2), *self._p(cv, e_mid, e_mid), True, y, (xh + 2, xw + 2), cin=cv['cin'], in_cstride=xc,
                                   ksize=3, stride=1, in_off=0, out_cstride=cv['cout'], out_d=(1, 1), ho=xh, wo=xw, batch=nb0, math=self.math)
                    outs.append(y[:1].clone())
                    x, xc = y, cv['cout']
            lvl[key] = outs
            # one entry = six (1, H+2, W+2, C) images (~110 MB at 188 x 188 x 128): ragged last groups and other batch sizes add keys -
            # keep the four most recent
            order = lvl.setdefault('_zero_resp_keys', [])
            order.append(key)
            while len(order) > 4:
                lvl.pop(order.pop(0), None)
        return lvl[key]

    def _level_convs(self, li, lvl, x, xh, xw, xc, batch, dev, sparse_in=None, out_last=None):
        """The 3 x 3 convolutions of block li over `batch` frames -> (activation, H, W, C).  out_last: where the block's LAST
        convolution writes (a zero-bordered buffer of the right shape, e.g. a frame slice of a larger one) instead of a workspace image."""
        convs = lvl['convs']
        bufs = None
        tiles = zero = None
        e_in, e_mid = self._e('encoded'), self._e('spatial_features_2d')     # exponents of the BEV input and of every tensor of this module
        if sparse_in is not None and li == 0 and SKIP_EMPTY_TILES:
            # pixel tiles far enough from any data compute the network's zero-input response: they are left out of the launches of the
            # block's layers (16-24 % of the tiles of a 160k-point frame at the first layer, 9-15 % at the sixth: the corners of the BEV
            # square beyond the sensor's range) and receive a copy of that response
            nl = min(len(convs), 6) if all(cv['stride'] == 1 for cv in convs) else 1
            key = ('zero_resp', xh, xw, int(self.math), str(dev), int(batch), e_in, e_mid)
            if nl > 1 and key not in lvl and torch.cuda.is_current_stream_capturing():
                nl = 1        # (the response images are computed by an eager pass: a capture without one before it skips at the first layer only)
                if not getattr(self, '_warned_capture_nl1', False):
                    self._warned_capture_nl1 = True
                    import warnings
                    warnings.warn('BaseBEVBackbone: graph capture of a %d-frame pass without an eager pass of the same size before it - empty tiles '
                                  'are skipped at the first layer only (run one eager pass first to skip them in all six)' % batch)
            tiles = ops.bev_tile_list(sparse_in[1], xh, xw, nl)
            zero = self._zero_response(lvl, xh, xw, sparse_in[0].shape[1], dev, batch) if nl > 1 else None
        for ci, cv in enumerate(convs):
            s = cv['stride']
            oh, ow = (xh + 2 - 3) // s + 1, (xw + 2 - 3) // s + 1
            if bufs is None:
                bufs = [bordered_zeros('bev2d.l%d.%d' % (li, k), (batch, oh + 2, ow + 2, cv['cout']), dev) for k in range(2)]
            y = out_last if (out_last is not None and ci == len(convs) - 1) else bufs[ci % 2]
            if sparse_in is not None and li == 0 and ci == 0:
                # (weights with the input channels in z-major order: the rows of slab 0, then of slab 1)
                rows, ridx = sparse_in
                wz, scz, shz = self._p(cv, e_in, e_mid, key='w_zmajor')
                conv_layer(rows, (xh + 2, xw + 2), wz, scz, shz, True, y, (oh + 2, ow + 2),
                           cin=cv['cin'], in_cstride=xc, ksize=3, stride=s, in_off=0, out_cstride=cv['cout'],
                           out_d=(1, 1), ho=oh, wo=ow, batch=batch, math=self.math, in_rowidx=ridx, in_row_channels=rows.shape[1],
                           in_rows=rows.shape[0], in_tiles=tiles[0] if tiles is not None else None)
                if tiles is not None:
                    ops.bev_fill_empty_tiles(tiles[0], batch, oh, ow, shz, True, cv['cout'], y, self.math)      # (ReLU(shift))
            else:
                tl = tiles[ci] if (tiles is not None and ci < tiles.shape[0]) else None
                wc, scc, shc = self._p(cv, e_in if (li == 0 and ci == 0) else e_mid, e_mid)
                conv_layer(x, (xh + 2, xw + 2), wc, scc, shc, True, y, (oh + 2, ow + 2),
                           cin=cv['cin'], in_cstride=xc, ksize=3, stride=s, in_off=0, out_cstride=cv['cout'],
                           out_d=(1, 1), ho=oh, wo=ow, batch=batch, math=self.math, in_tiles=tl)
                if tl is not None:
                    ops.bev_fill_empty_tiles(tl, batch, oh, ow, shc, True, cv['cout'], y, self.math, zero_resp=zero[ci])
            x, xh, xw, xc = y, oh, ow, cv['cout']
        return x, xh, xw, xc

    def _level_deblock(self, lvl, x, xh, xw, xc, concat, coff, h, w, batch):
        """ConvTranspose2d (kernel == stride) + BN + ReLU of a block's output into channels [coff, coff + cout) of the concatenation."""
        de = lvl['de']
        s = de['s']
        ctot = self.num_bev_features
        e_mid = self._e('spatial_features_2d')
        if xh * s != h or xw * s != w:
            raise DetZeroHipError('BaseBEVBackbone: deblock output %dx%d does not match %dx%d' % (xh * s, xw * s, h, w))
        if self.math and 1 < s * s <= 8 and FUSED_DEBLOCK_PHASES:
            # the s x s phases of the ConvTranspose2d as the groups of ONE launch (dz_conv2d_desc.phase_groups): the phases of a
            # pixel tile run next to each other on one XCD, the level's image is read from HBM once instead of s x s times
            conv_layer(x, (xh + 2, xw + 2), *self._p(de, e_mid, e_mid, key='w_phases'), True, concat,
                       (h + 2, w + 2), cin=de['cin'], in_cstride=xc, ksize=1, stride=1, in_off=1,
                       out_cstride=ctot, out_coff=coff, out_s=s, out_d=(1, 1), ho=xh, wo=xw, batch=batch, math=self.math,
                       groups=s * s, g_cout=[de['cout']] * (s * s), g_ooff=[0] * (s * s), phase_groups=True)
            return coff + de['cout']
        for dy in range(s):
            for dx in range(s):
                conv_layer(x, (xh + 2, xw + 2), *self._p(de['phases'][dy][dx], e_mid, e_mid, scale=de['scale'], shift=de['shift']), True, concat,
                           (h + 2, w + 2), cin=de['cin'], in_cstride=xc, ksize=1, stride=1, in_off=1,
                           out_cstride=ctot, out_coff=coff, out_s=s, out_d=(dy + 1, dx + 1), ho=xh, wo=xw,
                           batch=batch, math=self.math)
        return coff + de['cout']

    def run(self, bev, batch, sparse_in=None):
        """bev (B, H+2, W+2, Cin) zero-bordered channel-last -> concat (B, H+2, W+2, sum(upsample)) zero-bordered.
        In a split math mode both images are pair16 (same shapes).
        sparse_in = (rows (R, C) pair16, row index image (B, H+2, W+2, 2) int32) instead of `bev` (sparse_input_ok): the image is
        never built - HeightCompression fused into the first convolution."""
        plan = self.plan()
        if sparse_in is not None:
            rows, ridx = sparse_in
            dev = rows.device
            h, w = ridx.shape[1] - 2, ridx.shape[2] - 2
        else:
            dev = bev.device
            h, w = bev.shape[1] - 2, bev.shape[2] - 2
        concat = bordered_zeros('bev2d.concat', (batch, h + 2, w + 2, self.num_bev_features), dev)
        x, xh, xw, xc = bev, h, w, (bev.shape[3] if sparse_in is None else 2 * rows.shape[1])
        coff = 0
        for li, lvl in enumerate(plan):
            x, xh, xw, xc = self._level_convs(li, lvl, x, xh, xw, xc, batch, dev, sparse_in=sparse_in if li == 0 else None)
            coff = self._level_deblock(lvl, x, xh, xw, xc, concat, coff, h, w, batch)
        return concat

    def grouped_fits(self, nb, h, w):
        """Whether `run_grouped` can hold the images it keeps for ALL nb frames (the first block's output and every deeper block's
        activations) inside the 2 GiB window the kernels address through 32-bit offsets: at the Waymo size (190 x 190 x 128 words,
        18.5 MB per frame) that is 116 frames; beyond it the caller runs every layer per frame group."""
        plan = self.plan()
        xh, xw, worst = h, w, 0
        for lvl in plan:
            s = lvl['convs'][0]['stride']
            xh, xw = (xh + 2 - 3) // s + 1, (xw + 2 - 3) // s + 1
            worst = max(worst, (xh + 2) * (xw + 2) * lvl['convs'][-1]['cout'] * 4)
        return nb * worst < 2 ** 31

    def run_grouped(self, nb, group, consume, bev=None, sparse_in=None):
        """`run` for more frames than one concatenation image can hold (the kernels address an image through 32-bit offsets): the
        FIRST block and the deblocks + `consume(concat, g0, ng)` run in frame groups, the DEEPER blocks over all nb frames at once.
        Why not everything per group: at 16 frames the 94 x 94 layers of the Waymo config are 4.5 tiles per persistent workgroup -
        every launch runs 5 rounds where 4.5 would do (90 % of the chip); over 32 frames they are exactly 9."""
        plan = self.plan()
        if sparse_in is not None:
            rows, ridx = sparse_in
            dev = rows.device
            h, w = ridx.shape[1] - 2, ridx.shape[2] - 2
            cin0 = 2 * rows.shape[1]
        else:
            dev = bev.device
            h, w = bev.shape[1] - 2, bev.shape[2] - 2
            cin0 = bev.shape[3]
        l0 = plan[0]
        s0 = l0['convs'][0]['stride']
        h0, w0 = (h + 2 - 3) // s0 + 1, (w + 2 - 3) // s0 + 1
        c0 = l0['convs'][-1]['cout']
        if len(l0['convs']) < 2:
            raise DetZeroHipError('BaseBEVBackbone.run_grouped: the first block needs at least two convolutions')
        x0_all = bordered_zeros('bev2d.x0_all', (nb, h0 + 2, w0 + 2, c0), dev)
        # the first block's own images are a quarter of the concatenation's channels: it runs in groups as large as ITS widest image
        # allows (all 32 frames of the default pass: 6 launches instead of 12, half the launch ramps and tails)
        widest = max([cin0] + [cv['cout'] for cv in l0['convs']])
        group1 = max(group, min(nb, (2 ** 31 - 1) // ((h + 2) * (w + 2) * widest * 4))) if FIRST_BLOCK_WIDE_GROUPS else group
        for g0 in range(0, nb, group1):
            ng = min(group1, nb - g0)
            sp = (rows, ridx[g0:g0 + ng]) if sparse_in is not None else None
            self._level_convs(0, l0, None if sp is not None else bev[g0:g0 + ng], h, w, cin0, ng, dev, sparse_in=sp, out_last=x0_all[g0:g0 + ng])
        outs = [(x0_all, h0, w0, c0)]
        x, xh, xw, xc = outs[0]
        for li in range(1, len(plan)):
            x, xh, xw, xc = self._level_convs(li, plan[li], x, xh, xw, xc, nb, dev)
            outs.append((x, xh, xw, xc))
        for g0 in range(0, nb, group):
            ng = min(group, nb - g0)
            concat = bordered_zeros('bev2d.concat', (ng, h + 2, w + 2, self.num_bev_features), dev)
            coff = 0
            for lvl, (x, xh, xw, xc) in zip(plan, outs):
                coff = self._level_deblock(lvl, x[g0:g0 + ng], xh, xw, xc, concat, coff, h, w, ng)
            consume(concat, g0, ng)

    def forward(self, data_dict):
        _inference_only(self)
        with torch.no_grad():
            bev = data_dict.get('_nhwc_spatial_features', None)
            enc = data_dict.get('_nhwc_math', 0)
            e_src = int(data_dict.get('_nhwc_exp', 0) or 0)
            if bev is None:
                bev, enc = nchw_to_padded_nhwc(data_dict['spatial_features'].float()), 0
                e_src = 0
            bev = _recode(bev, enc, self.math, e_src, self._e('encoded'))
            concat = self.run(bev, bev.shape[0])
        e_out = self._e('spatial_features_2d')
        data_dict['_nhwc_spatial_features_2d'] = concat
        data_dict['_nhwc_math'] = self.math
        data_dict['_nhwc_exp'] = e_out
        plain = ops.pair16_to_f32(concat, self.math) if self.math else concat
        if e_out:
            plain = plain * (2.0 ** -e_out)
        data_dict['spatial_features_2d'] = plain[:, 1:-1, 1:-1, :].permute(0, 3, 1, 2)
        return data_dict


# ================================================================================================
# CenterHead
# ================================================================================================
class SeparateHead(nn.Module):
    """center_head.py:14-48 (parameter holder; executed batched by CenterHead)."""

    def __init__(self, input_channels, sep_head_dict, init_bias=-2.19, use_bias=False):
        super().__init__()
        self.sep_head_dict = sep_head_dict
        for cur_name in self.sep_head_dict:
            output_channels = self.sep_head_dict[cur_name]['out_channels']
            num_conv = self.sep_head_dict[cur_name]['num_conv']
            fc_list = []
            for _ in range(num_conv - 1):
                fc_list.append(nn.Sequential(
                    nn.Conv2d(input_channels, input_channels, kernel_size=3, stride=1, padding=1, bias=use_bias),
                    nn.BatchNorm2d(input_channels), nn.ReLU()))
            fc_list.append(nn.Conv2d(input_channels, output_channels, kernel_size=3, stride=1, padding=1, bias=True))
            fc = nn.Sequential(*fc_list)
            if 'hm' in cur_name:
                fc[-1].bias.data.fill_(init_bias)
            else:
                for m in fc.modules():
                    if isinstance(m, nn.Conv2d):
                        nn.init.kaiming_normal_(m.weight.data)
                        if m.bias is not None:
                            nn.init.constant_(m.bias, 0)
            self.__setattr__(cur_name, fc)


class CenterHead(_Cached):
    """center_head.py:51-488, inference path: shared conv, six branches (batched into one 64->384
    conv and one grouped 384->12 conv), decode + rotated NMS on the device.
    ``gt_boxes`` is optional: the reference runs its CPU target assignment even in eval (:448-453) and needs the key;
    here ``assign_targets`` runs when the batch carries it and fills ``forward_ret_dict['target_dicts']`` the same way."""

    COLS = {'center': (0, 2), 'center_z': (2, 1), 'dim': (3, 3), 'rot': (6, 2), 'iou': (8, 1), 'hm': (9, 3)}

    def __init__(self, model_cfg, input_channels, num_class, class_names, grid_size, point_cloud_range, voxel_size,
                 tta=False, predict_boxes_when_training=True):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.grid_size = grid_size
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_size = [float(v) for v in voxel_size]
        self.TTA = tta
        self.iou_weight = self.model_cfg.get('IOU_WEIGHT', 0)
        self.feature_map_stride = self.model_cfg.TARGET_ASSIGNER_CONFIG.get('FEATURE_MAP_STRIDE', None)
        self.class_names = class_names
        self.class_names_each_head = []
        self.class_id_mapping_each_head = []
        for cur_class_names in self.model_cfg.CLASS_NAMES_EACH_HEAD:
            self.class_names_each_head.append([x for x in cur_class_names if x in class_names])
            self.class_id_mapping_each_head.append(
                torch.from_numpy(np.array([self.class_names.index(x) for x in cur_class_names if x in class_names])))
        total_classes = sum(len(x) for x in self.class_names_each_head)
        assert total_classes == len(self.class_names), f'class_names_each_head={self.class_names_each_head}'
        if max(len(x) for x in self.class_names_each_head) > 3 or min(len(x) for x in self.class_names_each_head) < 1:
            raise DetZeroHipError('CenterHead: 1 to 3 classes per head (the decode kernel keeps the heat map in columns 9:12)')
        use_bias = self.model_cfg.get('USE_BIAS_BEFORE_NORM', False)
        self.shared_conv = nn.Sequential(
            nn.Conv2d(input_channels, self.model_cfg.SHARED_CONV_CHANNEL, 3, stride=1, padding=1, bias=use_bias),
            nn.BatchNorm2d(self.model_cfg.SHARED_CONV_CHANNEL), nn.ReLU())
        self.heads_list = nn.ModuleList()
        self.separate_head_cfg = self.model_cfg.SEPARATE_HEAD_CFG
        for cur_class_names in self.class_names_each_head:
            cur_head_dict = {k: dict(v) for k, v in self.separate_head_cfg.HEAD_DICT.items()}
            cur_head_dict['hm'] = dict(out_channels=len(cur_class_names), num_conv=self.model_cfg.NUM_HM_CONV)
            self.heads_list.append(SeparateHead(self.model_cfg.SHARED_CONV_CHANNEL, cur_head_dict, init_bias=-2.19,
                                                use_bias=use_bias))
        self.head_names = list(self.heads_list[0].sep_head_dict.keys())
        for name in self.head_names:
            cfgd = self.heads_list[0].sep_head_dict[name]
            if name not in self.COLS or cfgd['num_conv'] != 2 or (cfgd['out_channels'] != self.COLS[name][1] and name != 'hm'):
                raise DetZeroHipError('CenterHead: unsupported branch %s %s' % (name, dict(cfgd)))
        if sorted(self.head_names) != sorted(self.COLS):
            raise DetZeroHipError('CenterHead: branches must be %s (got %s)' % (sorted(self.COLS), self.head_names))
        self.predict_boxes_when_training = predict_boxes_when_training
        self.forward_ret_dict = {}
        self.input_channels = input_channels

    def plan(self):
        if self._plan is not None:
            return self._plan
        sc, sbn = self.shared_conv[0], self.shared_conv[1]
        s_scale, s_shift = fold_bn(sbn, sc.bias)
        c = self.model_cfg.SHARED_CONV_CHANNEL
        order = ['center', 'center_z', 'dim', 'rot', 'iou', 'hm']      # fixed column layout of the decode kernel
        heads = []
        for head, names in zip(self.heads_list, self.class_names_each_head):
            w1, sc1, sh1, w2, b2 = [], [], [], [], []
            for name in order:
                fc = getattr(head, name)
                conv1, bn1, conv2 = fc[0][0], fc[0][1], fc[1]
                a, b = fold_bn(bn1, conv1.bias)
                w1.append(_conv_weight_taps(conv1.weight)); sc1.append(a); sh1.append(b)
                w2.append(_conv_weight_taps(conv2.weight, 16))
                b2.append(_pad_vec(conv2.bias.detach().float(), 16))
            g_cout = [len(names) if n == 'hm' else self.COLS[n][1] for n in order]
            heads.append({
                'hidden': {'w': torch.cat(w1, dim=2).contiguous(), 'scale': torch.cat(sc1).contiguous(),
                           'shift': torch.cat(sh1).contiguous()},
                'final': {'w': torch.stack(w2, dim=0).contiguous(),          # (6, 9, 64, 16)
                          'shift': torch.cat(b2).contiguous(),
                          # the split engine works on 32-channel fragments: biases padded per group to 32
                          'shift32': torch.cat([_pad_vec(b, 32) for b in b2]).contiguous(),
                          'g_cout': g_cout, 'g_ooff': [self.COLS[n][0] for n in order]}})
        self._plan = {
            'shared': {'w': _conv_weight_taps(sc.weight), 'scale': s_scale, 'shift': s_shift, 'cin': sc.in_channels},
            'heads': heads, 'hidden': heads[0]['hidden'], 'final': heads[0]['final'],
            'c': c, 'order': order,
        }
        return self._plan

    def run_shared(self, concat, batch):
        """concat (B,H+2,W+2,Cin) zero-bordered -> the shared 3x3 conv's map (B,H+2,W+2,C), zero-bordered (center_head.py:443)."""
        p = self.plan()
        hp, wp = concat.shape[1], concat.shape[2]
        c = p['c']
        shared = bordered_zeros('head.shared', (batch, hp, wp, c), concat.device)
        e = self._e('spatial_features_2d')                    # (the head's hidden maps share the exponent of its input)
        conv_layer(concat, (hp, wp), *self._p(p['shared'], e, e), True, shared, (hp, wp),
                   cin=p['shared']['cin'], in_cstride=concat.shape[3], out_cstride=c, out_d=(1, 1), ho=hp - 2, wo=wp - 2, batch=batch,
                   math=self.math)
        return shared

    def run_head(self, shared, batch, index=0):
        """One SeparateHead (center_head.py:13-48, :445-447) on the shared map -> head map (B, H*W, 12) channel-last: its six
        branches as one 64 -> 384 conv and one grouped 384 -> 12 conv; a head with fewer than 3 classes leaves columns 9 + n .. 11 unwritten."""
        p = self.plan()
        hp_ = p['heads'][index]
        dev = shared.device
        hp, wp = shared.shape[1], shared.shape[2]
        h, w = hp - 2, wp - 2
        c = p['c']
        mm = self.math
        hidden = bordered_zeros('head.hidden', (batch, hp, wp, 6 * c), dev)
        e = self._e('spatial_features_2d')
        conv_layer(shared, (hp, wp), *self._p(hp_['hidden'], e, e), True, hidden, (hp, wp),
                   cin=c, in_cstride=c, out_cstride=6 * c, out_d=(1, 1), ho=h, wo=w, batch=batch, math=mm)
        head = torch.empty((batch, h * w, 12), dtype=torch.float32, device=dev)
        # (output layer: no BatchNorm - the plan holds no scale; fp32 output at exponent 0)
        conv_layer(hidden, (hp, wp), *self._p(hp_['final'], e, 0, scale=None, shift=hp_['final']['shift32' if mm else 'shift']), False, head, (h, w),
                   cin=c, in_cstride=6 * c, out_cstride=12, out_d=(0, 0), groups=6, cout_pad=32 if mm else 16,
                   g_cout=hp_['final']['g_cout'], g_ooff=hp_['final']['g_ooff'], ho=h, wo=w, batch=batch, math=mm, out_f32=True)
        return head, h, w

    def run_convs(self, concat, batch):
        """concat (B,H+2,W+2,Cin) zero-bordered -> head map (B, H*W, 12) channel-last of the FIRST head (the single-head layout of
        every DetZero config; FramePipeline's route)."""
        return self.run_head(self.run_shared(concat, batch), batch, 0)

    def assign_targets(self, gt_boxes, feature_map_size=None, **kwargs):
        """center_head.py:202-260 (host-side, like the reference's)."""
        from .target_assign import assign_targets
        return assign_targets(self, gt_boxes, feature_map_size)

    def decode_batched_nosync(self, head, h, w, index=0):
        """head (B,H*W,12) -> boxes (B,K,7), scores (B,K), labels (B,K) i32 (0-based), keep (B,K) i32, d_nk (B,) i32:
        top-K decode and rotated NMS of all frames in one launch sequence, counts stay on the device."""
        post = self.model_cfg.POST_PROCESSING
        nms = post.NMS_CONFIG
        if nms.NMS_TYPE != 'nms_gpu':
            raise DetZeroHipError('CenterHead: NMS_TYPE %s not supported (nms_gpu only)' % nms.NMS_TYPE)
        k = post.MAX_OBJ_PER_SAMPLE
        # candidates come out in descending score order and K <= NMS_PRE_MAXSIZE, so the reference's
        # topk(pre_max) + sort (model_nms_utils.py:15-20) is the identity here
        if k > nms.NMS_PRE_MAXSIZE:
            raise DetZeroHipError('MAX_OBJ_PER_SAMPLE > NMS_PRE_MAXSIZE is not supported')
        boxes, scores, labels, counts = ops.centerhead_decode(
            head, h, w, len(self.class_names_each_head[index]), k, post.SCORE_THRESH, post.POST_CENTER_LIMIT_RANGE,
            self.point_cloud_range, self.voxel_size, self.feature_map_stride, use_iou=self.iou_weight > 0)
        keep, d_nk = ops.nms_rotated_batched_nosync(boxes, counts, nms.NMS_THRESH, nms.NMS_POST_MAXSIZE)
        return boxes, scores, labels, keep, d_nk

    def decode_nosync(self, head, h, w, index=0):
        boxes, scores, labels, keep, d_nk = self.decode_batched_nosync(head, h, w, index)
        return [(boxes[b], scores[b], labels[b], keep[b], d_nk[b:b + 1]) for b in range(head.shape[0])]

    def generate_predicted_boxes(self, heads, h, w):
        """center_head.py:315-385: every head decodes and suppresses on its own; a frame's result is the heads' boxes one after the
        other, labels mapped through the head's class list.  `heads`: one head map or a list (one per head)."""
        heads = heads if isinstance(heads, (list, tuple)) else [heads]
        ret = [{'pred_boxes': [], 'pred_scores': [], 'pred_labels': []} for _ in range(heads[0].shape[0])]
        for index, head in enumerate(heads):
            mapping = self.class_id_mapping_each_head[index].to(head.device)
            for b, (boxes, scores, labels, keep, d_nk) in enumerate(self.decode_nosync(head, h, w, index)):
                nk = int(d_nk.item())
                sel = keep[:nk].long()
                ret[b]['pred_boxes'].append(boxes[sel])
                ret[b]['pred_scores'].append(scores[sel])
                ret[b]['pred_labels'].append(mapping[labels[sel].long()] + 1)
        return [{k: (v[0] if len(v) == 1 else torch.cat(v, dim=0)) for k, v in d.items()} for d in ret]

    def forward(self, data_dict):
        _inference_only(self)
        with torch.no_grad():
            concat = data_dict.get('_nhwc_spatial_features_2d', None)
            enc = data_dict.get('_nhwc_math', 0)
            e_src = int(data_dict.get('_nhwc_exp', 0) or 0)
            if concat is None:
                concat, enc = nchw_to_padded_nhwc(data_dict['spatial_features_2d'].float()), 0
                e_src = 0
            concat = _recode(concat, enc, self.math, e_src, self._e('spatial_features_2d'))
            shared = self.run_shared(concat, concat.shape[0])
            heads, preds = [], []
            for index, names in enumerate(self.class_names_each_head):
                head, h, w = self.run_head(shared, concat.shape[0], index)
                heads.append(head)
                preds.append({n: head.view(head.shape[0], h, w, 12)[..., o:o + (len(names) if n == 'hm' else c)].permute(0, 3, 1, 2)
                              for n, (o, c) in self.COLS.items()})
            if data_dict.get('gt_boxes', None) is not None:              # center_head.py:448-453
                self.forward_ret_dict['target_dicts'] = self.assign_targets(data_dict['gt_boxes'], feature_map_size=(h, w))
            self.forward_ret_dict['pred_dicts'] = preds
            pred_dicts = self.generate_predicted_boxes(heads, h, w)
            data_dict['final_box_dicts'] = pred_dicts
            if self.predict_boxes_when_training:         # second stage (center_head.py:461-486): first-stage boxes become the RoIs
                rois, roi_scores, roi_labels = self.reorder_rois(data_dict['batch_size'], pred_dicts)
                data_dict.update({'rois': rois, 'roi_scores': roi_scores, 'roi_labels': roi_labels, 'has_class_labels': True})
                if 'spatial_features_2d' in data_dict:              # center_head.py:461-486: BEV features at five points of every box
                    data_dict['roi_features'] = self.roi_features(data_dict['spatial_features_2d'], pred_dicts, rois.shape[1])
        return data_dict

    def roi_features(self, spatial_features_2d, pred_dicts, num_max_rois):
        """center_head.py:408-432,461-486 + reorder_rois_for_refining_features:388-406: (B, max boxes, 5 * C) bilinear BEV features at
        the centre and the four edge middles of every first-stage box (num_point = 5, :69), zero rows behind a frame's boxes."""
        b, c = spatial_features_2d.shape[0], spatial_features_2d.shape[1]
        out = spatial_features_2d.new_zeros((b, num_max_rois, 5 * c), dtype=torch.float32)
        hwc = spatial_features_2d.float().permute(0, 2, 3, 1)
        if hwc.stride(3) != 1:
            hwc = hwc.contiguous()
        for i, d in enumerate(pred_dicts):
            n = d['pred_boxes'].shape[0]
            if n:
                out[i, :n] = ops.roi_bev_features(d['pred_boxes'][:, :7], hwc[i], self.point_cloud_range[0], self.point_cloud_range[1],
                                                  self.voxel_size[0], self.voxel_size[1], self.feature_map_stride)
        return out

    @staticmethod
    def reorder_rois(batch_size, pred_dicts):
        """center_head.py:388-406 (the ``roi_features`` of :398-406 come from `roi_features` above): (B, max boxes, 7) RoIs padded with
        zero rows, scores, 1-based labels."""
        num_max_rois = max(1, max(len(d['pred_boxes']) for d in pred_dicts))
        ref = pred_dicts[0]['pred_boxes']
        rois = ref.new_zeros((batch_size, num_max_rois, ref.shape[-1]))
        roi_scores = ref.new_zeros((batch_size, num_max_rois))
        roi_labels = ref.new_zeros((batch_size, num_max_rois)).long()
        for b in range(batch_size):
            n = len(pred_dicts[b]['pred_boxes'])
            rois[b, :n] = pred_dicts[b]['pred_boxes']
            roi_scores[b, :n] = pred_dicts[b]['pred_scores']
            roi_labels[b, :n] = pred_dicts[b]['pred_labels']
        return rois, roi_scores, roi_labels


__all__ = {
    'MeanVFE': MeanVFE,
    'DynamicMeanVFE': DynamicMeanVFE,
    'VoxelResBackBone8x': VoxelResBackBone8x,
    'HeightCompression': HeightCompression,
    'BaseBEVBackbone': BaseBEVBackbone,
    'CenterHead': CenterHead,
}
