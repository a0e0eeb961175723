"""PDV second stage under the reference's module names, on the HIP kernels of csrc/pdv.hip.

Mirror of /root/reference/detection/detzero_det/models/centerpoint_modules/pdv_head.py (``RoIHeadTemplate`` :17-267,
``VoxelAggregationHead`` :269-575, ``PDVHead`` :594-637) with its helpers - ``voxel_aggregation_utils.py``, ``density_utils.py``,
``attention_utils.py``, ``box_coder_utils.ResidualCoder``, ``pointnet2_stack.pointnet2_modules.StackSAModuleMSGAttention`` -
for inference: same constructor arguments, ``batch_dict`` keys and ``state_dict()`` names / shapes (tests/test_pdv.py compares
with the manifest recorded from the reference class), so ``centerpoint_pdv_*`` checkpoints load unchanged.

Data flow of ``PDVHead.forward`` (eval):
  points -> voxel centroids per feature location          dz_pdv_voxel_centroids  (bitmap + scan: no sort, (b,z,y,x) order)
         -> rows of x_conv3 / x_conv4 under each centroid  dz_index_lookup         (the level's own bitmap: no dense hash table)
  RoIs   -> 6x6x6 grid points                              (a few thousand floats: torch elementwise)
         -> per location and radius: stacked ball query    dz_pdv_ball_query       (cell walk in index order instead of O(M x N))
            grouping + KDE density + feature gather        dz_pdv_group_features
            shared MLP, max over the samples               dz_linear_forward x2, dz_group_max
  points x RoIs -> points per box part                     dz_pdv_part_counts
  216 grid points per RoI -> one encoder layer              dz_linear_forward, dz_attention_single_head, dz_add_layernorm
  -> shared FC / regression / confidence heads              dz_linear_forward
  -> box decoding (ResidualCoder, a few flops per RoI)      torch elementwise
Training paths (proposal target layer, losses) are out of scope, like everywhere in this backend.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn

from . import iou3d_nms_utils, ops
from . import lib as L
from .det_modules import _Cached
from .lib import DetZeroHipError
from .refine_modules import _mha_plan, _pad16, _run_stack, _stack_plan


# ================================================================================================
# parameter holders (names and shapes of the reference modules)
# ================================================================================================
class StackSAModuleMSGAttention(nn.Module):
    """pointnet2_modules.py:31-158: per radius a grouper (no parameters) and a shared MLP of Conv2d 1x1 + BN + ReLU."""

    def __init__(self, *, radii, nsamples, mlps, use_xyz=True, pool_method='max_pool', use_density=False):
        super().__init__()
        if pool_method != 'max_pool' or not use_xyz:
            raise DetZeroHipError('StackSAModuleMSG: the HIP backend implements use_xyz + max_pool (the PDV configs)')
        self.radii, self.nsamples, self.use_density = list(radii), list(nsamples), bool(use_density)
        self.groupers = nn.ModuleList([nn.Module() for _ in radii])
        self.mlps = nn.ModuleList()
        for spec in mlps:
            spec = list(spec)
            spec[0] += 3 + (1 if use_density else 0)
            layers = []
            for k in range(len(spec) - 1):
                layers += [nn.Conv2d(spec[k], spec[k + 1], kernel_size=1, bias=False), nn.BatchNorm2d(spec[k + 1]), nn.ReLU()]
            self.mlps.append(nn.Sequential(*layers))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)


class FeedForwardPositionalEncoding(nn.Module):
    """attention_utils.py:112-133."""

    def __init__(self, d_input, d_output):
        super().__init__()
        self.ffn = nn.Sequential(nn.Conv1d(d_input, d_output // 2, 1), nn.BatchNorm1d(d_output // 2), nn.ReLU(d_output // 2),
                                 nn.Conv1d(d_output // 2, d_output, 1))


class TransformerEncoder(nn.Module):
    """attention_utils.py:7-52 (parameter holder: torch's own encoder modules give the reference's state-dict names)."""

    def __init__(self, attention_cfg, pos_encoder=None):
        super().__init__()
        self.attention_cfg = attention_cfg
        self.pos_encoder = pos_encoder
        layer = nn.TransformerEncoderLayer(attention_cfg.NUM_FEATURES, attention_cfg.NUM_HEADS, attention_cfg.NUM_HIDDEN_FEATURES, attention_cfg.DROPOUT)
        self.transformer_encoder = nn.TransformerEncoder(layer, attention_cfg.NUM_LAYERS, enable_nested_tensor=False)


def get_positional_encoder(pool_cfg):
    """attention_utils.py:136-154 (the feed-forward encoders; the frequency encoding has no parameters and is not used by any config)."""
    att = pool_cfg.ATTENTION
    d_in = {'grid_points': 3, 'density': 1, 'density_grid_points': 4}.get(att.POSITIONAL_ENCODER)
    if d_in is None:
        raise DetZeroHipError('PDV: POSITIONAL_ENCODER %r is not provided by the HIP backend' % att.POSITIONAL_ENCODER)
    return FeedForwardPositionalEncoding(d_input=d_in, d_output=att.NUM_FEATURES)


class ResidualCoder(object):
    """box_coder_utils.py:5-80, decode only."""

    def __init__(self, code_size=7, **kwargs):
        self.code_size = code_size

    @staticmethod
    def decode_torch(box_encodings, anchors):
        xa, ya, za, dxa, dya, dza, ra = torch.split(anchors[..., :7], 1, dim=-1)
        xt, yt, zt, dxt, dyt, dzt, rt = torch.split(box_encodings[..., :7], 1, dim=-1)
        diagonal = torch.sqrt(dxa ** 2 + dya ** 2)
        return torch.cat([xt * diagonal + xa, yt * diagonal + ya, zt * dza + za, torch.exp(dxt) * dxa, torch.exp(dyt) * dya,
                          torch.exp(dzt) * dza, rt + ra], dim=-1)


def rotate_points_along_z(points, angle):
    """common_utils.py:209-231 for (B, N, 3+) points and (B,) angles, written out instead of the batched matmul."""
    cosa, sina = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    x, y = points[..., 0], points[..., 1]
    out = points.clone()
    out[..., 0] = x * cosa - y * sina
    out[..., 1] = x * sina + y * cosa
    return out


# ================================================================================================
# device wrappers
# ================================================================================================
def voxel_centroids(points_b, point_cloud_range, voxel_size, stride, batch_size, scaling=None):
    """voxel_aggregation_utils.get_centroids_per_voxel_layer for one or two feature locations.
    -> [(centroids (M, 1+C), coords (M, 4) int32 bzyx, counts (M,), grid dims (D, H, W)), ...] exact-sized (one host sync)."""
    lib = L.load()
    L.require_cuda(points_b)
    n, cols = points_b.shape
    c = cols - 1
    vs = (torch.tensor(voxel_size).float() * stride).numpy()                               # float32, as the reference builds it
    rng = np.asarray(point_cloud_range, dtype=np.float32)
    grid = ((rng[3:6] - rng[0:3]) / vs).astype(np.int64)                                   # .long(): truncation
    dev = points_b.device
    cells = int(batch_size * grid[0] * grid[1] * grid[2])
    cap1 = max(min(n, cells), 1)
    two = scaling is not None
    cen1 = torch.empty((cap1, cols), dtype=torch.float32, device=dev)
    co1 = torch.empty((cap1, 4), dtype=torch.int32, device=dev)
    cn1 = torch.empty((cap1,), dtype=torch.int32, device=dev)
    dm = torch.zeros((2,), dtype=torch.int32, device=dev)
    cen2 = torch.empty((cap1, cols), dtype=torch.float32, device=dev) if two else None
    co2 = torch.empty((cap1, 4), dtype=torch.int32, device=dev) if two else None
    cn2 = torch.empty((cap1,), dtype=torch.int32, device=dev) if two else None
    sc = int(scaling) if two else 1
    ws = torch.empty((lib.dz_pdv_centroids_workspace_bytes(n, batch_size, int(grid[0]), int(grid[1]), int(grid[2]), sc, cap1),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.dz_pdv_voxel_centroids(L.ptr(points_b), n, c, L.f6(rng), L.f3(vs), L.i3(grid), batch_size, sc, L.ptr(cen1), L.ptr(co1), L.ptr(cn1),
                                        L.ptr(dm[0:1]), cap1, L.ptr(cen2), L.ptr(co2), L.ptr(cn2), L.ptr(dm[1:2]) if two else None, cap1 if two else 0,
                                        L.ptr(ws), ws.numel(), L.stream())
    L.check(rc, 'dz_pdv_voxel_centroids')
    m1, m2 = (int(v) for v in dm.tolist())
    out = [(cen1[:m1], co1[:m1], cn1[:m1], (int(grid[2]), int(grid[1]), int(grid[0])), vs)]
    if two:
        d2 = tuple((int(g) + sc - 1) // sc for g in (grid[2], grid[1], grid[0]))
        out.append((cen2[:m2], co2[:m2], cn2[:m2], d2, vs * np.float32(sc)))
    return out


def index_lookup(coords_bzyx, level):
    """Row of every cell in the sparse level, -1 where the level has no such cell (voxel_aggregation_utils.py:59-78)."""
    n = coords_bzyx.shape[0]
    out = torch.empty((max(n, 1),), dtype=torch.int32, device=coords_bzyx.device)
    with torch.cuda.device(coords_bzyx.device):
        rc = L.load().dz_index_lookup(L.ptr(coords_bzyx), None, n, L.ptr(level.bitmap), L.ptr(level.prefix), level.batch, *level.shape, level.layout,
                                      L.ptr(out), L.stream())
    L.check(rc, 'dz_index_lookup')
    return out[:n]


def _full_prefix(level, who):
    if getattr(level, 'prefix_partial', False):
        raise L.DetZeroHipError('%s ranks cells that may be inactive: it needs a level with a full prefix (ops.SparseLevel.build_from_coords), '
                                'not one written by dz_voxelize_to_level (prefix only at occupied words)' % who)
    if level.layout != ops.LAYOUT_LINEAR:
        raise L.DetZeroHipError('%s walks cells in the linear key order: the level must use LAYOUT_LINEAR' % who)


def ball_query(new_xyz, per_batch, xyz, level, lo, vs, radius, nsample):
    """-> idx (M, nsample) int32 (within-batch indices, padded with the first hit, zeros for an empty ball), cnt (M,) int32."""
    _full_prefix(level, 'dz_pdv_ball_query')
    mq = new_xyz.shape[0]
    dev = new_xyz.device
    idx = torch.empty((max(mq, 1), nsample), dtype=torch.int32, device=dev)
    cnt = torch.empty((max(mq, 1),), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.load().dz_pdv_ball_query(L.ptr(new_xyz), mq, per_batch, L.ptr(xyz), L.ptr(level.bitmap), L.ptr(level.prefix), level.batch, *level.shape,
                                        L.f3(lo), L.f3(vs), float(radius), nsample, L.ptr(idx), L.ptr(cnt), L.stream())
    L.check(rc, 'dz_pdv_ball_query')
    return idx[:mq], cnt[:mq]


def group_features(new_xyz, per_batch, xyz, feats, level, idx, cnt, row_stride):
    _full_prefix(level, 'dz_pdv_group_features')
    mq, nsample = idx.shape
    rows = torch.empty((max(mq * nsample, 1), row_stride), dtype=torch.float32, device=new_xyz.device)
    cells = level.shape[0] * level.shape[1] * level.shape[2]
    with torch.cuda.device(new_xyz.device):
        rc = L.load().dz_pdv_group_features(L.ptr(new_xyz), mq, per_batch, L.ptr(xyz), L.ptr(feats), feats.shape[1], L.ptr(level.bitmap), L.ptr(level.prefix),
                                            cells, L.ptr(idx), L.ptr(cnt), nsample, L.ptr(rows), row_stride, L.stream())
    L.check(rc, 'dz_pdv_group_features')
    return rows[:mq * nsample]


FUSED_SA = [True]            # development switch: dz_pdv_sa_pool (group + MLP + max in one kernel) vs group_features + layer launches


def sa_pool_supported(c, stack, nsample):
    return (FUSED_SA[0] and len(stack) == 2 and
            bool(L.load().dz_pdv_sa_pool_supported(int(c), stack[0]['w'].shape[0], stack[0]['cout'], stack[1]['cout'], int(nsample),
                                                   int(stack[0]['relu']), int(stack[1]['relu']))))


def sa_pool(new_xyz, per_batch, xyz, feats, level, idx, cnt, stack):
    """dz_pdv_sa_pool: grouping + the two layers of `stack` + max over the ball -> (M, cout) fp32."""
    _full_prefix(level, 'dz_pdv_sa_pool')
    mq, nsample = idx.shape
    l1, l2 = stack
    out = torch.empty((mq, l2['cout']), dtype=torch.float32, device=new_xyz.device)
    cells = level.shape[0] * level.shape[1] * level.shape[2]
    with torch.cuda.device(new_xyz.device):
        rc = L.load().dz_pdv_sa_pool(L.ptr(new_xyz), mq, per_batch, L.ptr(xyz), L.ptr(feats), feats.shape[1], L.ptr(level.bitmap), L.ptr(level.prefix), cells,
                                     L.ptr(idx), L.ptr(cnt), nsample, L.ptr(l1['w']), l1['w'].shape[1], L.ptr(l1['scale']), L.ptr(l1['shift']), l1['cout'],
                                     L.ptr(l2['w']), l2['w'].shape[1], L.ptr(l2['scale']), L.ptr(l2['shift']), l2['cout'], l1['w'].shape[0], L.ptr(out), L.stream())
    L.check(rc, 'dz_pdv_sa_pool')
    return out


def sa_pool_split_supported(c, stack, nsample, math):
    return (FUSED_SA[0] and SPLIT_SA[0] and math in (1, 2) and len(stack) == 2 and
            bool(L.load().dz_pdv_sa_pool_split_supported(int(c), stack[0]['w'].shape[0], stack[0]['cout'], stack[1]['cout'], int(nsample),
                                                         int(stack[0]['relu']), int(stack[1]['relu']))))


SPLIT_SA = [True]            # development switch: the pair16 instance of the fused branch in the split math modes


def sa_pool_split(new_xyz, per_batch, xyz, feats, level, idx, cnt, stack, math, out=None):
    """dz_pdv_sa_pool_split: sa_pool on pair16 operands (the head's split math modes).  out: a (M, cout) column block of a wider
    row-major tensor (the branches' results side by side: no concatenation pass)."""
    from .refine_modules import _split_w
    _full_prefix(level, 'dz_pdv_sa_pool_split')
    mq, nsample = idx.shape
    l1, l2 = stack
    w1, w2 = _split_w(l1, math), _split_w(l2, math)
    if out is None:
        out = torch.empty((mq, l2['cout']), dtype=torch.float32, device=new_xyz.device)
    if tuple(out.shape) != (mq, l2['cout']) or out.stride(1) != 1 or out.dtype != torch.float32:
        raise DetZeroHipError('sa_pool_split: out must be a (M, cout) fp32 column block')
    cells = level.shape[0] * level.shape[1] * level.shape[2]
    with torch.cuda.device(new_xyz.device):
        rc = L.load().dz_pdv_sa_pool_split(L.ptr(new_xyz), mq, per_batch, L.ptr(xyz), L.ptr(feats), feats.shape[0], feats.shape[1], L.ptr(level.bitmap),
                                           L.ptr(level.prefix), cells, L.ptr(idx), L.ptr(cnt), nsample, L.ptr(w1), w1.shape[1], L.ptr(l1['scale32']),
                                           L.ptr(l1['shift32']), l1['cout'], L.ptr(w2), w2.shape[1], L.ptr(l2['scale32']), L.ptr(l2['shift32']),
                                           l2['cout'], l1['w'].shape[0], int(math), L.ptr(out), out.stride(0), L.stream())
    L.check(rc, 'dz_pdv_sa_pool_split')
    return out


PART_COUNTS_BINNED = os.environ.get('DZ_TUNE_PART_BINNED', '1') != '0'      # 0: every point against every RoI of its frame (dz_pdv_part_counts)


def part_counts(points_b, rois, grid_size, max_num_boxes):
    """density_utils.find_num_points_per_part_multi -> (B, O, G, G, G) int32."""
    b, o = rois.shape[0], rois.shape[1]
    counts = torch.empty((b, o, grid_size, grid_size, grid_size), dtype=torch.int32, device=rois.device)
    r7 = rois[..., :7].float().contiguous()
    lib = L.load()
    with torch.cuda.device(rois.device):
        if PART_COUNTS_BINNED and o > 0:
            # RoIs binned on a BEV grid first (dz_pdv_part_counts_binned: the same counts, bit for bit)
            nb = int(lib.dz_pdv_part_counts_ws_bytes(b, o))
            ws = torch.empty((nb,), dtype=torch.uint8, device=rois.device)
            rc = lib.dz_pdv_part_counts_binned(L.ptr(points_b), points_b.shape[0], points_b.shape[1], L.ptr(r7), b, o, grid_size, max_num_boxes, L.ptr(counts),
                                               L.ptr(ws), nb, L.stream())
            L.check(rc, 'dz_pdv_part_counts_binned')
        else:
            rc = lib.dz_pdv_part_counts(L.ptr(points_b), points_b.shape[0], points_b.shape[1], L.ptr(r7), b, o, grid_size, max_num_boxes, L.ptr(counts), L.stream())
            L.check(rc, 'dz_pdv_part_counts')
    return counts


def attention_single_head(q, k, v, key_padding_mask, scale):
    r, l, e = q.shape
    out = torch.empty_like(q)
    m8 = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
    with torch.cuda.device(q.device):
        rc = L.load().dz_attention_single_head(L.ptr(q), L.ptr(k), L.ptr(v), L.ptr(m8), r, l, e, float(scale), L.ptr(out), L.stream())
    L.check(rc, 'dz_attention_single_head')
    return out


def self_attention_split(qp, xp, key_padding_mask, r, l, math):
    """dz_self_attention_split: softmax(q' x^T + mask) x per group of l rows; q', x and the result (r * l, 192) pair16."""
    out = torch.empty_like(xp)
    m8 = None if key_padding_mask is None else key_padding_mask.to(torch.uint8).contiguous()
    with torch.cuda.device(xp.device):
        rc = L.load().dz_self_attention_split(L.ptr(qp), L.ptr(xp), L.ptr(m8), r, l, xp.shape[1], L.ptr(out), int(math), L.stream())
    L.check(rc, 'dz_self_attention_split')
    return out


class _LazyDict(dict):
    """dict whose callable values are evaluated on first read (diagnostic tensors nobody may ask for)."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if callable(v) and not torch.is_tensor(v):
            v = v()
            dict.__setitem__(self, k, v)
        return v


FOLDED_ATTENTION = [True]    # development switch: the encoder layer on split operands with folded key / value projections
FUSED_ENCODER = [True]       # ... and its two halves around the attention as row-chain kernels (csrc/pdv_enc.hip)


def encoder_front(pos_in, feats, row_add, f, math):
    """dz_pdv_encoder_front -> (src, q') pair16 rows."""
    rows = feats.shape[0]
    src, q = torch.empty_like(feats), torch.empty_like(feats)
    with torch.cuda.device(feats.device):
        rc = L.load().dz_pdv_encoder_front(L.ptr(pos_in), pos_in.shape[1], L.ptr(feats), L.ptr(row_add), rows, L.ptr(f['w0']), L.ptr(f['s0']), L.ptr(f['b0']),
                                           L.ptr(f['w1']), L.ptr(f['b1']), L.ptr(f['wq']), L.ptr(f['uq']), L.ptr(src), L.ptr(q), int(math), L.stream())
    L.check(rc, 'dz_pdv_encoder_front')
    return src, q


def encoder_back(op, srcp, pooled, row_skip, f, math, out_pair16=False):
    """dz_pdv_encoder_back -> pooled + (row_skip ? pooled : encoder output), fp32 rows (or pair16 rows: the FC stack's operand)."""
    out = torch.empty_like(pooled)
    with torch.cuda.device(pooled.device):
        rc = L.load().dz_pdv_encoder_back(L.ptr(op), L.ptr(srcp), L.ptr(pooled), L.ptr(row_skip), pooled.shape[0], L.ptr(f['wo']), L.ptr(f['bo']),
                                          L.ptr(f['g1']), L.ptr(f['be1']), float(f['eps1']), L.ptr(f['fw1']), L.ptr(f['fb1']), L.ptr(f['fw2']), L.ptr(f['fb2']),
                                          L.ptr(f['g2']), L.ptr(f['be2']), float(f['eps2']), L.ptr(out), 1 if out_pair16 else 0, int(math), L.stream())
    L.check(rc, 'dz_pdv_encoder_back')
    return out


def _seq_plan(seq, cin_pad=None):
    return _stack_plan(nn.Sequential(*[m for m in seq if not isinstance(m, nn.Dropout)]), cin_pad=cin_pad)


# ================================================================================================
# the head
# ================================================================================================
class PDVHead(_Cached):
    def __init__(self, input_channels, model_cfg, point_cloud_range, voxel_size, num_class=1, **kwargs):
        super().__init__()
        self.model_cfg = model_cfg
        self.num_class = num_class
        self.pool_cfg = model_cfg.ROI_GRID_POOL
        self.point_cloud_range = [float(v) for v in point_cloud_range]
        self.voxel_size = [float(v) for v in voxel_size]
        if model_cfg.TARGET_CONFIG.BOX_CODER != 'ResidualCoder':
            raise DetZeroHipError('PDVHead: BOX_CODER %s is not provided' % model_cfg.TARGET_CONFIG.BOX_CODER)
        self.box_coder = ResidualCoder(**model_cfg.TARGET_CONFIG.get('BOX_CODER_CONFIG', {}))
        if model_cfg.get('DENSITY_CONFIDENCE', {}).get('ENABLED') or model_cfg.VOXEL_AGGREGATION.get('USE_EMPTY_VOXELS'):
            raise DetZeroHipError('PDVHead: DENSITY_CONFIDENCE / USE_EMPTY_VOXELS are not used by any DetZero config and not provided')
        att = self.pool_cfg.get('ATTENTION', {})
        if not att.get('ENABLED'):
            raise DetZeroHipError('PDVHead: the HIP backend implements the attention variant of the DetZero configs')
        layer_cfg = self.pool_cfg.POOL_LAYERS
        c_out = 0
        self.roi_grid_pool_layers = nn.ModuleList()
        for i, src in enumerate(self.pool_cfg.FEATURE_LOCATIONS):
            mlps = [[model_cfg.VOXEL_AGGREGATION.NUM_FEATURES[i]] + list(m) for m in layer_cfg[src].MLPS]
            self.roi_grid_pool_layers.append(StackSAModuleMSGAttention(
                radii=layer_cfg[src].POOL_RADIUS, nsamples=layer_cfg[src].NSAMPLE, mlps=mlps, use_xyz=True,
                pool_method=layer_cfg[src].POOL_METHOD, use_density=layer_cfg[src].get('USE_DENSITY')))
            c_out += sum(m[-1] for m in mlps)
        assert att.NUM_FEATURES == c_out, 'ATTENTION.NUM_FEATURES must equal voxel aggregation output dimension of %d.' % c_out
        if att.NUM_HEADS != 1 or att.NUM_LAYERS != 1:
            raise DetZeroHipError('PDVHead: one encoder layer with one head (the DetZero configs)')
        self.attention_head = TransformerEncoder(att, get_positional_encoder(self.pool_cfg))
        for p in self.attention_head.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        g = self.pool_cfg.GRID_SIZE
        pre = g * g * g * c_out
        shared = []
        for k in range(len(model_cfg.SHARED_FC)):
            shared += [nn.Conv1d(pre, model_cfg.SHARED_FC[k], kernel_size=1, bias=False), nn.BatchNorm1d(model_cfg.SHARED_FC[k]), nn.ReLU()]
            pre = model_cfg.SHARED_FC[k]
            if k != len(model_cfg.SHARED_FC) - 1 and model_cfg.DP_RATIO > 0:
                shared.append(nn.Dropout(model_cfg.DP_RATIO))
        self.shared_fc_layer = nn.Sequential(*shared)
        self.reg_layers = self.make_fc_layers(pre, self.box_coder.code_size * self.num_class, model_cfg.REG_FC)
        self.cls_layers = self.make_fc_layers(pre, self.num_class, model_cfg.CLS_FC)
        self.c_out = c_out
        self.forward_ret_dict = None

    def make_fc_layers(self, input_channels, output_channels, fc_list):
        """pdv_head.py:35-49."""
        layers, pre = [], input_channels
        for k in range(len(fc_list)):
            layers += [nn.Conv1d(pre, fc_list[k], kernel_size=1, bias=False), nn.BatchNorm1d(fc_list[k]), nn.ReLU()]
            pre = fc_list[k]
            if self.model_cfg.DP_RATIO >= 0 and k == 0:
                layers.append(nn.Dropout(self.model_cfg.DP_RATIO))
        layers.append(nn.Conv1d(pre, output_channels, kernel_size=1, bias=True))
        return nn.Sequential(*layers)

    # ---- kernel-layout parameters
    def plan(self):
        if self._plan is not None:
            return self._plan
        g3 = self.pool_cfg.GRID_SIZE ** 3
        p = {'pool': []}
        for layer in self.roi_grid_pool_layers:
            cin = layer.mlps[0][0].weight.shape[1]
            p['pool'].append([{'stack': _seq_plan(m, cin_pad=_pad16(cin)), 'stride': _pad16(cin)} for m in layer.mlps])
        enc = self.attention_head.transformer_encoder.layers[0]
        p['pos'] = _seq_plan(self.attention_head.pos_encoder.ffn, cin_pad=16)
        p['mha'] = _mha_plan(enc.self_attn)
        p['enc'] = {'w1': enc.linear1.weight.detach().float().t().contiguous(), 'b1': enc.linear1.bias.detach().float().contiguous(),
                    'w2': enc.linear2.weight.detach().float().t().contiguous(), 'b2': enc.linear2.bias.detach().float().contiguous(),
                    'one1': torch.ones(enc.linear1.out_features, device=enc.linear1.weight.device),
                    'one2': torch.ones(enc.linear2.out_features, device=enc.linear1.weight.device),
                    'ln': [(n.weight.detach().float().contiguous(), n.bias.detach().float().contiguous(), n.eps) for n in (enc.norm1, enc.norm2)]}
        if getattr(enc, 'norm_first', False):
            raise DetZeroHipError('PDVHead: post-norm encoder layers only')
        # one head: the key / value projections fold into the query and output GEMMs (csrc/pdv_attn.hip), in float64 on the host:
        #   scores (x_i wq + bq) . (x_j wk + bk) = (x_i (wq wk^T) + wk bq) . x_j + terms constant in j;  output (P x) (wv wo) + (bv wo + bo)
        m = p['mha']
        d = lambda t: t.detach().double()       # noqa: E731
        c2 = m['scale'] * 1.4426950408889634   # scores in log2 units (the kernel exponentiates with v_exp_f32)
        lyr = lambda w, b: {'w': w.float().contiguous(), 'cout': w.shape[1], 'scale32': torch.ones(w.shape[1], device=w.device),      # noqa: E731
                            'shift32': b.float().contiguous()}
        p['fold'] = {'q': lyr(d(m['wq']) @ d(m['wk']).t() * c2, (d(m['wk']) @ d(m['bq'])) * c2),
                     'o': lyr(d(m['wv']) @ d(m['wo']), d(m['bv']) @ d(m['wo']) + d(m['bo'])),
                     'f1': lyr(p['enc']['w1'], p['enc']['b1']), 'f2': lyr(p['enc']['w2'], p['enc']['b2'])}
        # shared FC: the reference flattens (RoI, C, 216) channel-major; the pooled rows here are (RoI, 216, C) -> permute the columns once
        shared = [m for m in self.shared_fc_layer if not isinstance(m, nn.Dropout)]
        w0 = shared[0].weight.detach()
        first = nn.Conv1d(w0.shape[1], w0.shape[0], 1, bias=False)
        first.weight.data = w0.view(w0.shape[0], self.c_out, g3).permute(0, 2, 1).reshape(w0.shape[0], g3 * self.c_out, 1).contiguous()
        p['shared'] = _stack_plan(nn.Sequential(first.to(w0.device), *shared[1:]))
        p['reg'] = _seq_plan(self.reg_layers)
        p['cls'] = _seq_plan(self.cls_layers)
        self._plan = p
        return p

    # ---- proposals (pdv_head.py:51-107)
    @torch.no_grad()
    def proposal_layer(self, batch_dict, nms_config):
        if batch_dict.get('rois', None) is not None:
            return batch_dict
        batch_size = batch_dict['batch_size']
        box_preds_all, cls_preds_all = batch_dict['batch_box_preds'], batch_dict['batch_cls_preds']
        rois = box_preds_all.new_zeros((batch_size, nms_config.NMS_POST_MAXSIZE, box_preds_all.shape[-1]))
        roi_scores = box_preds_all.new_zeros((batch_size, nms_config.NMS_POST_MAXSIZE))
        roi_labels = box_preds_all.new_zeros((batch_size, nms_config.NMS_POST_MAXSIZE), dtype=torch.long)
        if nms_config.MULTI_CLASSES_NMS:
            raise NotImplementedError
        for index in range(batch_size):
            mask = (batch_dict['batch_index'] == index) if batch_dict.get('batch_index', None) is not None else index
            box_preds, cls_preds = box_preds_all[mask], cls_preds_all[mask]
            cur_scores, cur_labels = torch.max(cls_preds, dim=1)
            selected, _ = iou3d_nms_utils.nms_gpu(box_preds[:, 0:7], cur_scores, nms_config.NMS_THRESH, pre_maxsize=nms_config.NMS_PRE_MAXSIZE)
            selected = selected[:nms_config.NMS_POST_MAXSIZE]
            rois[index, :len(selected)] = box_preds[selected]
            roi_scores[index, :len(selected)] = cur_scores[selected]
            roi_labels[index, :len(selected)] = cur_labels[selected]
        batch_dict['rois'], batch_dict['roi_scores'], batch_dict['roi_labels'] = rois, roi_scores, roi_labels + 1
        batch_dict['has_class_labels'] = cls_preds_all.shape[-1] > 1
        batch_dict.pop('batch_index', None)
        return batch_dict

    # ---- voxel centroids and the feature rows under them (pdv_head.py:598-637)
    def get_point_voxel_features(self, batch_dict):
        locs = list(self.model_cfg.VOXEL_AGGREGATION.FEATURE_LOCATIONS)
        if len(locs) > 2:
            raise DetZeroHipError('PDVHead: one or two VOXEL_AGGREGATION.FEATURE_LOCATIONS')
        strides = batch_dict['multi_scale_3d_strides']
        points = batch_dict['points'].float().contiguous()
        scaling = int(strides[locs[1]] / strides[locs[0]]) if len(locs) == 2 else None
        levels = voxel_centroids(points, self.point_cloud_range, self.voxel_size, strides[locs[0]], batch_dict['batch_size'], scaling)
        point_features, point_coords, self._point_index = {}, {}, {}
        for loc, (cen, coords, _, dims, vs) in zip(locs, levels):
            x_conv = batch_dict['multi_scale_3d_features'][loc]
            level = getattr(x_conv, '_level', None)
            if level is None:                                   # a tensor of another backend: index its coordinates first
                level = ops.SparseLevel(x_conv.batch_size, x_conv.spatial_shape, max(x_conv.indices.shape[0], 1), points.device)
                level.build_from_coords(x_conv.indices.int().contiguous(), want_rank=False)
            rows = index_lookup(coords, level)
            n_rows = x_conv.indices.shape[0]
            # (ranks at or beyond the feature rows exist only when a calibrated level overflowed its capacity - FramePipeline.two_stage
            # refuses such a pass; the mask keeps the gather inside the tensor for any other caller)
            sel = torch.nonzero((rows >= 0) & (rows < n_rows)).flatten()
            point_coords[loc] = cen[sel][:, :4].contiguous()
            if hasattr(x_conv, 'feature_rows'):          # (our tensors: gather first, decode the gathered rows only)
                point_features[loc] = x_conv.feature_rows(rows[sel].long()).contiguous()
            else:
                point_features[loc] = x_conv.features[rows[sel].long()].contiguous()
            self._point_index[loc] = (coords[sel].contiguous(), dims, vs)
        return point_features, point_coords

    # ---- RoI grid pooling (pdv_head.py:375-473)
    @staticmethod
    def get_dense_grid_points(rois, batch_size_rcnn, grid_size):
        dense_idx = rois.new_ones((grid_size, grid_size, grid_size)).nonzero().repeat(batch_size_rcnn, 1, 1).float()
        size = rois.view(batch_size_rcnn, -1)[:, 3:6]
        return (dense_idx + 0.5) / grid_size * size.unsqueeze(dim=1) - (size.unsqueeze(dim=1) / 2)

    def get_global_grid_points_of_roi(self, batch_dict, grid_size):
        rois = batch_dict['rois'].view(-1, batch_dict['rois'].shape[-1])
        local = self.get_dense_grid_points(rois, rois.shape[0], grid_size)
        glob = rotate_points_along_z(local.clone(), rois[:, 6]) + rois[:, 0:3].unsqueeze(dim=1)
        return glob, local

    def stack_math(self):
        """Arithmetic of this head's MLP / FC stacks = the head's own math mode (set_math), NOT the refiner's global one: 'f32'
        pins them to the exact-fp32 engine; the opt-in 'f16' detector mode runs them as f16 pairs (there is no single-product
        linear kernel)."""
        m = int(getattr(self, 'math', 0) or 0)
        return 1 if m == 3 else m

    def roi_grid_pool(self, batch_dict, cat_balls=True):
        """cat_balls=False: the fourth result is the list of per-branch ball-index tensors (M, nsample) instead of their concatenation."""
        p = self.plan()
        batch_size = batch_dict['batch_size']
        g = self.pool_cfg.GRID_SIZE
        glob, local = self.get_global_grid_points_of_roi(batch_dict, g)
        new_xyz = glob.reshape(-1, 3).float().contiguous()
        per_batch = new_xyz.shape[0] // batch_size
        lo = np.asarray(self.point_cloud_range[:3], dtype=np.float32)
        pooled, balls = [], []
        wide, placed = None, []        # the branches' results side by side (written in place by the split kernels)
        for k, loc in enumerate(self.pool_cfg.FEATURE_LOCATIONS):
            coords, dims, vs = self._point_index[loc]
            xyz = batch_dict['point_coords'][loc][:, 1:4].contiguous()
            feats = batch_dict['point_features'][loc].float().contiguous()
            level = ops.SparseLevel(batch_size, dims, max(coords.shape[0], 1), new_xyz.device)
            level.build_from_coords(coords, want_rank=False)      # rank of a centroid's cell = its row (rows are in cell-key order)
            layer = self.roi_grid_pool_layers[k]
            for s, (radius, nsample) in enumerate(zip(layer.radii, layer.nsamples)):
                if xyz.shape[0] == 0:          # no centroid on an active cell (empty / out-of-range frame): every ball is empty
                    couts = p['pool'][k][s]['stack'][-1]['cout']
                    pooled.append(new_xyz.new_zeros((new_xyz.shape[0], couts)))
                    balls.append(torch.zeros((new_xyz.shape[0], nsample), dtype=torch.int32, device=new_xyz.device))
                    continue
                idx, cnt = ball_query(new_xyz, per_batch, xyz, level, lo, vs, radius, nsample)
                stack = p['pool'][k][s]['stack']
                if sa_pool_split_supported(feats.shape[1], stack, nsample, self.stack_math()):       # group + MLP + max in one kernel
                    if wide is None:
                        wide = torch.empty((new_xyz.shape[0], self.c_out), dtype=torch.float32, device=new_xyz.device)
                    c0 = sum(t.shape[1] for t in pooled)
                    pooled.append(sa_pool_split(new_xyz, per_batch, xyz, feats, level, idx, cnt, stack, self.stack_math(),
                                                out=wide[:, c0:c0 + stack[-1]['cout']]))
                    placed.append(True)
                elif sa_pool_supported(feats.shape[1], stack, nsample):   # (exact fp32)
                    pooled.append(sa_pool(new_xyz, per_batch, xyz, feats, level, idx, cnt, stack))
                else:
                    rows = group_features(new_xyz, per_batch, xyz, feats, level, idx, cnt, p['pool'][k][s]['stride'])
                    out, _ = _run_stack(rows, stack, math=self.stack_math())
                    pooled.append(ops.group_max(out, new_xyz.shape[0], nsample))
                balls.append(idx)
        in_place = wide is not None and len(placed) == len(pooled)          # every branch wrote its column block of `wide`
        all_pooled = (wide if in_place else torch.cat(pooled, dim=-1)).view(-1, g ** 3, self.c_out)
        if not cat_balls:
            return all_pooled, glob.view(batch_size, -1, 3), local, balls
        all_balls = torch.cat(balls, dim=-1).view(-1, g ** 3, sum(b.shape[1] for b in balls))
        return all_pooled, glob.view(batch_size, -1, 3), local, all_balls

    def get_positional_input(self, points, rois, local_roi_grid_points):
        att = self.pool_cfg.ATTENTION
        ppp = part_counts(points.float().contiguous(), rois, self.pool_cfg.GRID_SIZE, att.MAX_NUM_BOXES)
        ppp = ppp.view(ppp.shape[0] * ppp.shape[1], -1, 1).float()
        ppp = torch.log10(ppp + 0.5) - (math.log10(0.5) if self.model_cfg.get('DENSITY_LOG_SHIFT') else 0)
        if att.POSITIONAL_ENCODER == 'grid_points':
            return local_roi_grid_points
        if att.POSITIONAL_ENCODER == 'density':
            return ppp
        return torch.cat((local_roi_grid_points, ppp), dim=-1)

    # ---- the encoder layer (attention_utils.py:17-52 around nn.TransformerEncoderLayer, post-norm, ReLU)
    def attention(self, point_features, positional_input, key_padding_mask, combine=False, pair16_ok=False):
        """combine: return pooled + attended features (COMBINE) instead of the attended ones.  pair16_ok: the fused split path may return
        ('pair16', rows (R * L, E) pair16, math) - the FC stack's operand - instead of an fp32 tensor."""
        p = self.plan()
        r, l, e = point_features.shape
        feats = point_features.reshape(r * l, e).contiguous()
        pos_in = positional_input.reshape(r * l, -1).float()
        empty = key_padding_mask.all(-1)                                   # RoIs without any point: left untouched (:31-44)
        add_pos = (~key_padding_mask) & (~empty)[:, None]
        m = p['mha']
        sm = self.stack_math()
        if (FOLDED_ATTENTION[0] and FUSED_ENCODER[0] and combine and sm in (1, 2) and m['heads'] == 1 and e == 192 and
                L.load().dz_self_attention_split_supported(l, e) and self._fused_encoder_ok(p, positional_input)):
            return self._attention_fused(p, sm, point_features, positional_input, key_padding_mask, empty, add_pos, r, l, e, pair16_ok)
        pos_rows = torch.nn.functional.pad(pos_in, (0, 16 - pos_in.shape[1]))      # (one launch: zero-padded to the stack's input width)
        pos, _ = _run_stack(pos_rows, p['pos'], math=self.stack_math())
        # feats + pos where add_pos, feats elsewhere, in one pass: pos * 1.0 and feats + 0.0 are exact (the encodings are finite)
        src = torch.addcmul(feats, pos, add_pos.reshape(r * l, 1).to(feats.dtype))
        if FOLDED_ATTENTION[0] and sm in (1, 2) and m['heads'] == 1 and L.load().dz_self_attention_split_supported(l, e) and e % 32 == 0:
            return self._attention_split(p, sm, point_features, src, key_padding_mask, empty, r, l, e, combine)
        q = ops.linear(src, m['wq'], m['one'], m['bq'], False, e)
        k = ops.linear(src, m['wk'], m['one'], m['bk'], False, e)
        v = ops.linear(src, m['wv'], m['one'], m['bv'], False, e)
        mask = key_padding_mask & (~empty)[:, None]                        # (an all-masked row would divide 0 by 0; its result is discarded)
        o = attention_single_head(q.view(r, l, e), k.view(r, l, e), v.view(r, l, e), mask, float(e) ** -0.5)
        o = ops.linear(o.view(r * l, e), m['wo'], m['one'], m['bo'], False, e)
        enc = p['enc']
        x = ops.add_layernorm(src, o, *enc['ln'][0])
        h = ops.linear(x, enc['w1'], enc['one1'], enc['b1'], True, enc['w1'].shape[1])
        y = ops.linear(h, enc['w2'], enc['one2'], enc['b2'], False, enc['w2'].shape[1])
        y = ops.add_layernorm(x, y, *enc['ln'][1])
        out = torch.where(empty[:, None, None], point_features, y.view(r, l, e))
        return point_features + out if combine else out

    @staticmethod
    def _fused_encoder_ok(p, positional_input):
        pos, enc = p['pos'], p['enc']
        if 'pos1_unit_scale' not in p:
            # the fused front chain applies the second positional layer as w1 x + b1 (no per-channel scale): a BatchNorm behind that
            # layer (any non-identity scale) takes the layered path instead.  Checked once per plan (one host sync; plans are rebuilt
            # when the weights are reloaded)
            sc = pos[1].get('scale') if len(pos) == 2 else None
            p['pos1_unit_scale'] = sc is None or bool(torch.all(sc[:192] == 1).item())
        return (p['pos1_unit_scale'] and positional_input.shape[-1] in (4, 8) and len(pos) == 2 and pos[0]['cout'] == 96 and pos[0]['relu'] and pos[1]['cout'] == 192 and
                not pos[1]['relu'] and tuple(enc['w1'].shape) == (192, 128) and tuple(enc['w2'].shape) == (128, 192))

    def _attention_fused(self, p, sm, point_features, positional_input, key_padding_mask, empty, add_pos, r, l, e, pair16_out=False):
        """COMBINE'd encoder layer in three launches: front chain (positional encoder, src, folded query), attention over the input rows,
        back chain (output projection, both LayerNorms, feed-forward block, pooled + result)."""
        from .refine_modules import _split_w
        key = 'fused%d' % sm
        if key not in p:
            f, enc, pos = p['fold'], p['enc'], p['pos']
            w0 = pos[0]['w'][:16]                                  # (16, 96): the stack's inputs are zero-padded to 16
            p[key] = {'w0': ops.pack_weight_split(w0, sm), 's0': pos[0]['scale'][:96].contiguous(), 'b0': pos[0]['shift'][:96].contiguous(),
                      'w1': ops.pack_weight_split(pos[1]['w'][:96], sm), 'b1': pos[1]['shift'][:192].contiguous(),
                      'wq': _split_w(f['q'], sm), 'uq': f['q']['shift32'], 'wo': _split_w(f['o'], sm), 'bo': f['o']['shift32'],
                      'g1': enc['ln'][0][0], 'be1': enc['ln'][0][1], 'eps1': enc['ln'][0][2], 'g2': enc['ln'][1][0], 'be2': enc['ln'][1][1],
                      'eps2': enc['ln'][1][2], 'fw1': _split_w(f['f1'], sm), 'fb1': f['f1']['shift32'], 'fw2': _split_w(f['f2'], sm), 'fb2': f['f2']['shift32']}
        f = p[key]
        pooled = point_features.reshape(r * l, e).contiguous()
        pos_in = positional_input.reshape(r * l, -1).float().contiguous()
        add = add_pos.reshape(r * l).to(torch.uint8)
        mask = key_padding_mask & (~empty)[:, None]
        skip = empty.to(torch.uint8)[:, None].expand(r, l).reshape(r * l).contiguous()
        # the row tensors are addressed through 32-bit buffer offsets: RoIs in chunks of < 2 GiB of rows (11 k RoIs of 216 x 192)
        step = max(1, ((1 << 31) - (1 << 20)) // (l * e * 4))
        outs = []
        for r0 in range(0, r, step):
            r1 = min(r, r0 + step)
            a, b = r0 * l, r1 * l
            srcp, qp = encoder_front(pos_in[a:b], pooled[a:b], add[a:b], f, sm)
            op = self_attention_split(qp, srcp, mask[r0:r1], r1 - r0, l, sm)
            outs.append(encoder_back(op, srcp, pooled[a:b], skip[a:b], f, sm, out_pair16=pair16_out))
        out = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
        return ('pair16', out, sm) if pair16_out else out.view(r, l, e)

    def _attention_split(self, p, sm, point_features, src, key_padding_mask, empty, r, l, e, combine=False):
        """The encoder layer on pair16 operands: q' GEMM, dz_self_attention_split over the input rows themselves, output GEMM, and the
        feed-forward block; LayerNorms and residuals in fp32 (dz_add_layernorm)."""
        from .refine_modules import _split_w
        f, enc = p['fold'], p['enc']
        lin = lambda xp, k, relu, f32: ops.linear_split(xp, _split_w(f[k], sm), f[k]['scale32'], f[k]['shift32'], relu, f[k]['cout'], sm, out_f32=f32)   # noqa: E731
        srcp = ops.pair16_from_f32(src, math=sm)
        qp = lin(srcp, 'q', False, False)
        mask = key_padding_mask & (~empty)[:, None]                        # (an all-masked group's result is discarded below)
        op = self_attention_split(qp, srcp, mask, r, l, sm)
        x = ops.add_layernorm(src, lin(op, 'o', False, True), *enc['ln'][0])
        hp = lin(ops.pair16_from_f32(x, math=sm), 'f1', True, False)
        if combine:     # pooled + (RoI without points ? pooled : LayerNorm(x + ffn)) in the normalisation's own pass
            g2, b2, eps2 = enc['ln'][1]
            return ops.add_layernorm_combine(x, lin(hp, 'f2', False, True), g2, b2, eps2, point_features.reshape(r * l, e), empty.to(torch.uint8), l).view(r, l, e)
        y = ops.add_layernorm(x, lin(hp, 'f2', False, True), *enc['ln'][1])
        return torch.where(empty[:, None, None], point_features, y.view(r, l, e))

    def generate_predicted_boxes(self, batch_size, rois, cls_preds, box_preds):
        """pdv_head.py:238-266."""
        code_size = self.box_coder.code_size
        batch_cls_preds = None if cls_preds is None else cls_preds.view(batch_size, -1, cls_preds.shape[-1])
        roi_ry = rois[:, :, 6].reshape(-1)
        roi_xyz = rois[:, :, 0:3].reshape(-1, 3)
        local_rois = rois.clone().detach()
        local_rois[:, :, 0:3] = 0
        boxes = self.box_coder.decode_torch(box_preds.view(batch_size, -1, code_size), local_rois).view(-1, code_size)
        boxes = rotate_points_along_z(boxes.unsqueeze(dim=1), roi_ry).squeeze(dim=1)
        boxes[:, 0:3] += roi_xyz
        return batch_cls_preds, boxes.view(batch_size, -1, code_size)

    @torch.no_grad()
    def forward(self, batch_dict):
        if self.training:
            raise DetZeroHipError('PDVHead: only the inference path is implemented on the HIP backend (call .eval())')
        p = self.plan()
        batch_dict['point_features'], batch_dict['point_coords'] = self.get_point_voxel_features(batch_dict)
        self.proposal_layer(batch_dict, nms_config=self.model_cfg.NMS_CONFIG['TEST'])
        pooled, _, local, balls = self.roi_grid_pool(batch_dict, cat_balls=False)
        g3 = self.pool_cfg.GRID_SIZE ** 3
        if not self.pool_cfg.ATTENTION.get('MASK_EMPTY_POINTS'):
            mask = torch.zeros(pooled.shape[:2], dtype=torch.bool, device=pooled.device)
        elif len(balls) <= 4 and all(b.shape[1] % 4 == 0 for b in balls):
            mask = ops.rows_all_zero(balls).view(-1, g3)            # (ball_idxs == 0).all(-1) over the branches, no concatenation
        else:
            mask = (torch.cat(balls, dim=-1) == 0).all(-1).view(-1, g3)
        ball_idxs = lambda: torch.cat(balls, dim=-1).view(-1, g3, sum(b.shape[1] for b in balls))     # noqa: E731  (built when somebody reads it)
        pos_in = self.get_positional_input(batch_dict['points'], batch_dict['rois'], local)
        from .refine_modules import _run_stack_split, _splittable
        nroi = pooled.shape[0]
        att = self.attention(pooled, pos_in, mask, combine=bool(self.pool_cfg.ATTENTION.get('COMBINE')),
                             pair16_ok=_splittable(p['shared'], nroi, self.stack_math()))
        if isinstance(att, tuple):          # the fused encoder's result as pair16 rows = the first FC layer's operand: no conversion pass
            _, attp, sm = att
            shared, _ = _run_stack_split(attp.view(nroi, -1), p['shared'], math=sm)
            att = lambda: ops.pair16_to_f32(attp, sm).view(pooled.shape)      # noqa: E731  (fp32 view for whoever reads forward_ret_dict)
        else:
            rows = att.reshape(att.shape[0], -1).contiguous()               # (RoI, 216 * C): the shared FC's columns were permuted to match
            shared, _ = _run_stack(rows, p['shared'], math=self.stack_math())
        rcnn_reg, _ = _run_stack(shared, p['reg'], math=self.stack_math())
        rcnn_cls, _ = _run_stack(shared, p['cls'], math=self.stack_math())
        cls_preds, box_preds = self.generate_predicted_boxes(batch_dict['batch_size'], batch_dict['rois'], rcnn_cls.contiguous(), rcnn_reg.contiguous())
        batch_dict['batch_cls_preds'], batch_dict['batch_box_preds'] = cls_preds, box_preds
        batch_dict['cls_preds_normalized'] = False
        self.forward_ret_dict = _LazyDict({'pooled_features': pooled, 'ball_idxs': ball_idxs, 'positional_input': pos_in, 'key_padding_mask': mask,
                                 'attention_output': att, 'rcnn_cls': rcnn_cls, 'rcnn_reg': rcnn_reg})
        return batch_dict
