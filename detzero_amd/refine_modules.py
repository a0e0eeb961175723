"""Refining module (GRM / PRM) regression networks under the reference's names, on the HIP backend.

Mirror of /root/reference/refining/detzero_refine/models/modules/:
  geometry_transformer.py:11-156 (GeometryTransformer), position_transformer.py:14-141 (PositionTransformer),
  head/{geometry_head.py,position_head.py} (decoder stacks), transformer/{decoder.py, multi_head_attention.py,
  ffn.py, position_encoding.py, conv_module.py}, target_assign.py:73-104 (decode),
  utils/detzero_utils/model_utils.py:81-135 (make_linear/fc/conv_layers).
Same constructor arguments, ``forward(data_dict)`` keys and ``state_dict()`` names/shapes (checked against the
reference's own manifest in tests/golden/refine_golden.npz).  Inference only.

Execution (all tensors channel-last, one row per point / query):
  * every Conv1d/Conv2d(1x1)/Linear + BatchNorm(eval) + ReLU is one ``dz_linear_forward`` (fp32 MFMA, folded BN);
  * ``torch.max`` over points = ``dz_group_max``; the ``repeat`` + ``cat`` of the pooled feature with the
    per-point feature is never materialised: its product with the first MLP layer is a per-object vector that
    enters the GEMM epilogue as a row-group addend;
  * attention = in-projections (``dz_linear_forward``) + ``dz_mha_core`` (scores never leave registers) +
    out-projection; residual + LayerNorm = ``dz_add_layernorm``.
"""
import contextlib

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .det_modules import _Cached, _inference_only, fold_bn
from .lib import DetZeroHipError


# ------------------------------------------------------------------------------------------------
# parameter holders with the reference's module tree (names matter for state_dict compatibility)
# ------------------------------------------------------------------------------------------------
def make_linear_layers(linear_cfg, input_channels, output_channels, output_use_norm=False):
    """model_utils.py:81-97."""
    layers, c_in = [], input_channels
    for k in range(len(linear_cfg)):
        layers.extend([nn.Linear(c_in, linear_cfg[k], bias=False), nn.BatchNorm1d(linear_cfg[k], eps=1e-3, momentum=0.01), nn.ReLU()])
        c_in = linear_cfg[k]
    if output_use_norm:
        layers.extend([nn.Linear(c_in, output_channels, bias=False), nn.BatchNorm1d(output_channels, eps=1e-3, momentum=0.01), nn.ReLU()])
    else:
        layers.append(nn.Linear(c_in, output_channels, bias=True))
    return nn.Sequential(*layers)


def make_fc_layers(linear_cfg, input_channels, output_channels, output_use_norm=False):
    """model_utils.py:99-115."""
    layers, c_in = [], input_channels
    for k in range(len(linear_cfg)):
        layers.extend([nn.Conv1d(c_in, linear_cfg[k], kernel_size=1, bias=False), nn.BatchNorm1d(linear_cfg[k], eps=1e-3, momentum=0.01), nn.ReLU()])
        c_in = linear_cfg[k]
    if output_use_norm:
        layers.extend([nn.Conv1d(c_in, output_channels, kernel_size=1, bias=False), nn.BatchNorm1d(output_channels, eps=1e-3, momentum=0.01), nn.ReLU()])
    else:
        layers.append(nn.Conv1d(c_in, output_channels, kernel_size=1, bias=True))
    return nn.Sequential(*layers)


def make_conv_layers(conv_cfg, input_channels, output_channels, output_use_norm=False):
    """model_utils.py:117-135."""
    layers, c_in = [], input_channels
    for k in range(len(conv_cfg)):
        layers.extend([nn.Conv2d(c_in, conv_cfg[k], kernel_size=1, bias=False), nn.BatchNorm2d(conv_cfg[k], eps=1e-3, momentum=0.01), nn.ReLU()])
        c_in = conv_cfg[k]
    if output_use_norm:
        layers.extend([nn.Conv2d(c_in, output_channels, kernel_size=1, bias=False), nn.BatchNorm2d(output_channels, eps=1e-3, momentum=0.01), nn.ReLU()])
    else:
        layers.append(nn.Conv2d(c_in, output_channels, kernel_size=1, bias=True))
    return nn.Sequential(*layers)


class MultiheadAttention(nn.Module):
    """Parameter layout of transformer/multi_head_attention.py:7-62 (in_proj_weight/bias, out_proj)."""

    def __init__(self, embed_dim, num_heads, dropout=0.0):
        super().__init__()
        self.embed_dim, self.num_heads = embed_dim, num_heads
        self.head_dim = embed_dim // num_heads
        self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * embed_dim))
        self.out_proj = nn.Linear(embed_dim, embed_dim, bias=True)
        nn.init.xavier_uniform_(self.in_proj_weight)


class PositionEmbeddingLearned(nn.Module):
    """transformer/position_encoding.py:4-21."""

    def __init__(self, input_channel, num_pos_feats=288):
        super().__init__()
        self.position_embedding_head = nn.Sequential(
            nn.Conv1d(input_channel, num_pos_feats, kernel_size=1), nn.BatchNorm1d(num_pos_feats), nn.ReLU(inplace=True),
            nn.Conv1d(num_pos_feats, num_pos_feats, kernel_size=1))


class TransformerDecoderLayer(nn.Module):
    """transformer/decoder.py:7-46 (parameters only; executed by ``decoder_layer_forward``)."""

    def __init__(self, d_model, nhead, dim_feedforward=2048, dropout=0.1, activation='relu', self_posembed=None,
                 cross_posembed=None, cross_only=False):
        super().__init__()
        if activation != 'relu':
            raise DetZeroHipError('TransformerDecoderLayer: only relu is implemented (all DetZero configs)')
        if cross_posembed is not None:
            raise DetZeroHipError('TransformerDecoderLayer: cross_posembed is not used by any DetZero config')
        self.cross_only = cross_only
        if not cross_only:
            self.self_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.multihead_attn = MultiheadAttention(d_model, nhead, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1, self.norm2, self.norm3 = nn.LayerNorm(d_model), nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.self_posembed = self_posembed
        self.nhead = nhead


class ConvModule(nn.Module):
    """conv + bn(+relu) with the attribute names of transformer/conv_module.py:182-320 ('conv', 'bn')."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=1, bias=False)      # bias='auto' with a norm -> no bias
        self.bn = nn.BatchNorm1d(out_channels)


class FFN(nn.Module):
    """transformer/ffn.py:6-67: per head [ConvModule(C->64)] + Conv1d(64->classes, bias)."""

    def __init__(self, in_channels, heads, head_conv=64):
        super().__init__()
        self.heads = heads
        for head, (classes, num_conv) in heads.items():
            if num_conv != 2:
                raise DetZeroHipError('FFN: num_conv must be 2 (all DetZero configs)')
            self.__setattr__(head, nn.Sequential(ConvModule(in_channels, head_conv), nn.Conv1d(head_conv, classes, kernel_size=1, bias=True)))


class _Head(nn.Module):
    """head/geometry_head.py:13-95 / head/position_head.py:13-82: decoder layers + prediction heads."""

    def __init__(self, pos_dims, common_heads, num_decoder_layers=1, num_heads=8, hidden_channel=256, ffn_channel=256,
                 dropout=0.1, activation='relu', cross_only=False, auxiliary=True, **kwargs):
        super().__init__()
        if num_decoder_layers != 1:
            raise DetZeroHipError('refiner heads: num_decoder_layers must be 1 (all DetZero configs)')
        self.decoder = nn.ModuleList([TransformerDecoderLayer(
            hidden_channel, num_heads, ffn_channel, dropout, activation,
            self_posembed=PositionEmbeddingLearned(pos_dims, hidden_channel), cross_only=cross_only)])
        self.prediction_heads = nn.ModuleList([FFN(hidden_channel, dict(common_heads))])


class GeometryHead(_Head):
    def __init__(self, num_classes=3, **kw):
        super().__init__(3, {'geometry_cls': (num_classes, 2), 'geometry_reg': (num_classes * 3, 2)}, **kw)


class PositionHead(_Head):
    def __init__(self, num_classes=3, **kw):
        super().__init__(4, {'center_reg': (3, 2), 'heading_cls': (12, 2), 'heading_reg': (12, 2)}, **kw)


# ------------------------------------------------------------------------------------------------
# kernel-layout parameter preparation
# ------------------------------------------------------------------------------------------------
def _pad16(n):
    return (n + 15) // 16 * 16


def _wt(weight, cin_pad=None, cout_pad=None):
    """(Cout, Cin[,1[,1]]) conv/linear weight -> (Cin_pad, Cout_pad) fp32 for dz_linear_forward."""
    w = weight.detach().float().reshape(weight.shape[0], weight.shape[1]).t().contiguous()
    ci, co = w.shape
    cin_pad = ci if cin_pad is None else cin_pad
    cout_pad = _pad16(co) if cout_pad is None else cout_pad
    out = w.new_zeros((cin_pad, cout_pad))
    out[:ci, :co] = w
    return out


def _vec(v, n, fill=0.0):
    out = v.new_full((n,), fill, dtype=torch.float32)
    out[:v.numel()] = v.detach().float()
    return out


def _stack_plan(seq, cin_pad=None):
    """nn.Sequential of [conv|linear, bn, relu]* (+ optional final conv with bias) -> list of layer dicts."""
    mods = list(seq)
    layers, i = [], 0
    while i < len(mods):
        lin = mods[i]
        cout = lin.weight.shape[0]
        if i + 1 < len(mods) and isinstance(mods[i + 1], (nn.BatchNorm1d, nn.BatchNorm2d)):
            scale, shift = fold_bn(mods[i + 1], lin.bias)
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            i += 3 if relu else 2
        else:
            scale = torch.ones(cout, device=lin.weight.device)
            shift = lin.bias.detach().float() if lin.bias is not None else torch.zeros(cout, device=lin.weight.device)
            relu = False
            i += 1
        cp = _pad16(cout)
        layers.append({'w': _wt(lin.weight, cin_pad if not layers else None, cp), 'scale': _vec(scale, cp, 1.0),
                       'shift': _vec(shift, cp), 'relu': relu, 'cout': cout})
    return layers


# Arithmetic of the refiner's big MLP stacks: 0 = fp32 matrix cores (default), 1 / 2 = split-precision pairs on the 16-bit
# matrix cores (csrc/hgemm.h) - set by the model's forward from its `math` (set_math).  Only stacks with many
# rows switch (the conversion pass has to pay), and only when every hidden width is a multiple of 32.
_REFINE_MATH = [0]
SPLIT_MIN_ROWS = 2048


@contextlib.contextmanager
def _math_mode(m):
    prev = _REFINE_MATH[0]
    _REFINE_MATH[0] = int(m)
    try:
        yield
    finally:
        _REFINE_MATH[0] = prev


def _with_math(fn):
    """Model forward under the module's arithmetic (`set_math('f16x2')`; default fp32)."""
    def wrapper(self, data_dict):
        with _math_mode(self.math):
            return fn(self, data_dict)
    return wrapper


def _r32(n):
    return (n + 31) // 32 * 32


def _split_w(l, m, key='w'):
    """pair16 weights (cout_pad32, cin_pad32) of a layer dict, packed once per math mode; also scale / shift of that width."""
    ck = '%s_split%d' % (key, m)
    if ck not in l:
        w = l[key]
        ci, co = w.shape
        wp = w.new_zeros((_r32(ci), _r32(co)))
        wp[:ci, :co] = w
        l[ck] = ops.pack_weight_split(wp, m)
        if 'scale32' not in l:
            l['scale32'], l['shift32'] = _vec(l['scale'], _r32(co), 1.0), _vec(l['shift'], _r32(co))
    return l[ck]


def _splittable(layers, rows, math=None):
    m = _REFINE_MATH[0] if math is None else math
    return bool(m) and rows >= SPLIT_MIN_ROWS and all(l['cout'] % 32 == 0 for l in layers[:-1])


def _gmax_ok(gmax):
    return gmax is not None and gmax[1] >= 128 and gmax[1] % 128 == 0


FOLDED_XATTN = [True]         # development switch: dz_xattn_folded for few-query cross-attention vs projecting the memory to K and V
FOLD_MIN_KEYS = 256
SPLIT_ATTENTION = [True]      # development switch: dz_mha_core_split for long key lists in the split math modes vs the fp32 core
SPLIT_ATTN_MIN_KEYS = 1024
FUSED_CHAIN = [True]          # development switch: dz_mlp_chain_forward (memory MLP + K / V in one kernel) vs one launch per layer
FUSED_POINTNET = [True]       # development switch: the fused encoder kernel (csrc/pointnet.hip) vs layer-by-layer launches


def _pointnet3_ok(xp, layers, gmax, m, x_f32=False):
    """x_f32: xp is still the fp32 rows (16 or 32 columns, contiguous) - the kernel splits them itself, no conversion pass."""
    cols_ok = (xp.shape[1] in (16, 32) and xp.is_contiguous()) if x_f32 else xp.shape[1] == 32
    return (FUSED_POINTNET[0] and m in (1, 2) and gmax is not None and gmax[1] % 32 == 0 and len(layers) == 3 and cols_ok
            and layers[0]['cout'] == 128 and layers[1]['cout'] == 128 and layers[2]['cout'] in (128, 256, 512)
            and all(l['relu'] for l in layers))


def _run_stack_split(xp, layers, keep_pair=False, math=None, gmax=None, tap=False, x_f32=False):
    """xp: pair16 rows.  Hidden layers stay pair16; the last one returns fp32 unless keep_pair.
    gmax = (groups, rows per group): the max over each group's rows is fused into the last layer (-> (groups, cout) fp32); a
    32 -> 128 -> 128 -> C PointNet encoder runs as ONE kernel (dz_pointnet3_forward; outs[1] = the second layer's rows)."""
    m = _REFINE_MATH[0] if math is None else math
    if x_f32 and not _pointnet3_ok(xp, layers, gmax, m, True):          # fp32 rows the fused encoder cannot take as they are
        xp, x_f32 = ops.pair16_from_f32(xp, c_dst=_r32(xp.shape[1]), math=m), False
    if _pointnet3_ok(xp, layers, gmax, m, x_f32):
        trip = [(_split_w(l, m), l['scale32'], l['shift32']) for l in layers]
        pooled, tap = ops.pointnet3(xp, trip, gmax[1], m, want_tap=tap, x_f32=x_f32)
        return pooled, [None, tap, pooled]
    outs = []
    for li, l in enumerate(layers):
        last = li == len(layers) - 1
        w = _split_w(l, m)
        if last and gmax is not None:
            if _gmax_ok(gmax):
                xp = ops.linear_split(xp, w, l['scale32'], l['shift32'], l['relu'], l['cout'], m, group_rows=gmax[1], group_max=True)
            else:
                xp = ops.group_max(ops.linear_split(xp, w, l['scale32'], l['shift32'], l['relu'], l['cout'], m, out_f32=True), gmax[0], gmax[1])
        else:
            xp = ops.linear_split(xp, w, l['scale32'], l['shift32'], l['relu'], l['cout'], m, out_f32=last and not keep_pair)
        outs.append(xp)
    return xp, outs


def _run_stack(x, layers, upto=None, math=None, gmax=None):
    """math: None = the refiner's global mode (set_refine_math); an explicit mode (0 = fp32 engine, 1 / 2 = split pairs) pins the
    stack's arithmetic whatever the refiner is set to (the PDV head passes its own)."""
    layers = layers if upto is None else layers[:upto]
    if _splittable(layers, x.shape[0], math):
        m = _REFINE_MATH[0] if math is None else math
        return _run_stack_split(x.contiguous(), layers, math=m, gmax=gmax, x_f32=True)      # (converted there unless the fused encoder takes fp32 rows)
    outs = []
    for li, l in enumerate(layers):
        if ops.linear_splitk_ok(x.shape[0], x.shape[1]) and x.shape[1] == l['w'].shape[0]:       # a few rows, a very long input
            x = ops.linear_splitk(x, l['w'], l['scale'], l['shift'], l['relu'], l['cout'])
        else:
            x = ops.linear(x, l['w'], l['scale'], l['shift'], l['relu'], l['cout'])
        outs.append(x)
    if gmax is not None:
        x = ops.group_max(x, gmax[0], gmax[1])
        outs[-1] = x
    return x, outs


def _pad_cols(x, n):
    if x.shape[1] == n:
        return x.contiguous()
    out = x.new_zeros((x.shape[0], n))
    out[:, :x.shape[1]] = x
    return out


def _mha_plan(m):
    e = m.embed_dim
    w, b = m.in_proj_weight.detach().float(), m.in_proj_bias.detach().float()
    one = torch.ones(e, device=w.device)
    return {'wq': w[:e].t().contiguous(), 'wk': w[e:2 * e].t().contiguous(), 'wv': w[2 * e:].t().contiguous(), 'wk_oi': w[e:2 * e].contiguous(),
            'bq': b[:e].contiguous(), 'bk': b[e:2 * e].contiguous(), 'bv': b[2 * e:].contiguous(),
            'wo': m.out_proj.weight.detach().float().t().contiguous(), 'bo': m.out_proj.bias.detach().float().contiguous(),
            'one': one, 'heads': m.num_heads, 'scale': float(m.head_dim) ** -0.5, 'e': e}


def _mha_forward(p, q_rows, k_rows, v_rows, b, lq, lk, key_padding_mask, kv=None):
    """rows are (B*L, E) channel-last.  multi_head_attention.py:199-288.  kv: the key / value projections of the memory when the
    caller already has them (dz_mlp_chain_forward produces them with the memory rows)."""
    e = p['e']
    q = ops.linear(q_rows, p['wq'], p['one'], p['bq'], False, e)
    am = _REFINE_MATH[0] if (SPLIT_ATTENTION[0] and lk >= SPLIT_ATTN_MIN_KEYS and _REFINE_MATH[0] in (1, 2)) else 0   # long key lists only
    if kv is not None:
        o = ops.mha_core(q.view(b, lq, e), kv[0].view(b, lk, e), kv[1].view(b, lk, e), key_padding_mask, p['heads'], p['scale'], math=am)
        return ops.linear(o.view(b * lq, e), p['wo'], p['one'], p['bo'], False, e)
    if FOLDED_XATTN[0] and k_rows is v_rows and lk >= FOLD_MIN_KEYS and ops.xattn_folded_supported(lq, e, p['heads']):
        # a few queries over a long memory (GRM: 3 x 4096): the key / value projections fold into the queries, the memory is read once
        o = ops.xattn_folded(q.view(b, lq, e), k_rows.view(b, lk, e), key_padding_mask, p['wk_oi'], p['wv'], p['bv'], p['heads'], p['scale'])
        return ops.linear(o.view(b * lq, e), p['wo'], p['one'], p['bo'], False, e)
    m = _REFINE_MATH[0]
    if m and k_rows is v_rows and k_rows.shape[0] >= SPLIT_MIN_ROWS and e % 32 == 0:
        # key / value projections of a long memory: one conversion of the memory rows, two split GEMMs
        if 'kv_split%d' % m not in p:
            p['kv_split%d' % m] = (ops.pack_weight_split(p['wk'], m), ops.pack_weight_split(p['wv'], m))
        wk, wv = p['kv_split%d' % m]
        mp = ops.pair16_from_f32(k_rows, c_dst=e, math=m)
        k = ops.linear_split(mp, wk, p['one'], p['bk'], False, e, m, out_f32=True)
        v = ops.linear_split(mp, wv, p['one'], p['bv'], False, e, m, out_f32=True)
    else:
        k = ops.linear(k_rows, p['wk'], p['one'], p['bk'], False, e)
        v = ops.linear(v_rows, p['wv'], p['one'], p['bv'], False, e)
    o = ops.mha_core(q.view(b, lq, e), k.view(b, lk, e), v.view(b, lk, e), key_padding_mask, p['heads'], p['scale'], math=am)
    return ops.linear(o.view(b * lq, e), p['wo'], p['one'], p['bo'], False, e)


def decoder_layer_plan(layer):
    pe = layer.self_posembed.position_embedding_head
    one = torch.ones(layer.linear1.out_features, device=layer.linear1.weight.device)
    return {
        'pos': _stack_plan(pe, cin_pad=16),
        'sa': None if layer.cross_only else _mha_plan(layer.self_attn), 'ca': _mha_plan(layer.multihead_attn),
        'w1': layer.linear1.weight.detach().float().t().contiguous(), 'b1': layer.linear1.bias.detach().float().contiguous(),
        'w2': layer.linear2.weight.detach().float().t().contiguous(), 'b2': layer.linear2.bias.detach().float().contiguous(),
        'one1': one, 'one2': torch.ones(layer.linear2.out_features, device=one.device),
        'ln': [(n.weight.detach().float().contiguous(), n.bias.detach().float().contiguous(), n.eps)
               for n in (layer.norm1, layer.norm2, layer.norm3)],
    }


def decoder_layer_forward(p, query, memory, query_pos, b, lq, lk, sa_mask=None, ca_mask=None, memory_kv=None):
    """decoder.py:48-92.  query (B*Lq, C), memory (B*Lk, C), query_pos (B*Lq, pos_dims) -> (B*Lq, C)."""
    pos, _ = _run_stack(_pad_cols(query_pos, 16), p['pos'])
    if p['sa'] is not None:
        qk = ops.add_layernorm(query, pos, None, None, norm=False)              # q = k = v = query + pos
        q2 = _mha_forward(p['sa'], qk, qk, qk, b, lq, lq, sa_mask)
        g, be, eps = p['ln'][0]
        query = ops.add_layernorm(query, q2, g, be, eps)
    qp = ops.add_layernorm(query, pos, None, None, norm=False)
    q2 = _mha_forward(p['ca'], qp, memory, memory, b, lq, lk, ca_mask, kv=memory_kv)
    g, be, eps = p['ln'][1]
    query = ops.add_layernorm(query, q2, g, be, eps)
    h = ops.linear(query, p['w1'], p['one1'], p['b1'], True, p['w1'].shape[1])
    q2 = ops.linear(h, p['w2'], p['one2'], p['b2'], False, p['w2'].shape[1])
    g, be, eps = p['ln'][2]
    return ops.add_layernorm(query, q2, g, be, eps)


def ffn_plan(ffn):
    plan = {}
    for name in ffn.heads:
        seq = getattr(ffn, name)
        cm, last = seq[0], seq[1]
        scale, shift = fold_bn(cm.bn, cm.conv.bias)
        c1 = cm.conv.weight.shape[0]
        co = last.weight.shape[0]
        plan[name] = [{'w': _wt(cm.conv.weight, None, _pad16(c1)), 'scale': _vec(scale, _pad16(c1), 1.0), 'shift': _vec(shift, _pad16(c1)),
                       'relu': True, 'cout': c1},
                      {'w': _wt(last.weight, None, _pad16(co)), 'scale': _vec(torch.ones(co, device=scale.device), _pad16(co), 1.0),
                       'shift': _vec(last.bias, _pad16(co)), 'relu': False, 'cout': co}]
    return plan


def ffn_forward(plan, x):
    return {name: _run_stack(x, layers)[0] for name, layers in plan.items()}


class _PointNetPlan:
    """encoder (3 layers, tap the 2nd) + max pool + concat-MLP, as used for the `memory` branch of GRM and PRM."""

    def __init__(self, encoder, mlp, cin, pooled_first):
        self.enc = _stack_plan(encoder, cin_pad=_pad16(cin))
        self.cin_pad = _pad16(cin)
        c_mid = self.enc[1]['cout']                 # hooked intermediate (encoder[5] = ReLU after the 2nd conv)
        c_pool = self.enc[2]['cout']
        first = list(mlp)[0]
        w = first.weight.detach().float().reshape(first.weight.shape[0], first.weight.shape[1]).t().contiguous()   # (Cin, Cout)
        if pooled_first:                            # PRM: cat([pooled, intermediate])
            w_pool, w_mid = w[:c_pool], w[c_pool:]
        else:                                       # GRM: cat([intermediate, pooled])
            w_mid, w_pool = w[:c_mid], w[c_mid:]
        self.w_pool = w_pool.contiguous()
        self.mlp = _stack_plan(mlp)
        self.mlp[0]['w'] = w_mid.contiguous()
        self.ones = torch.ones(w.shape[1], device=w.device)
        self.zeros = torch.zeros(w.shape[1], device=w.device)

    def _chain_ok(self, rows, length, m):
        """dz_mlp_chain_forward applies: 128 -> 512 -> 256 with ReLU on both layers, whole 32-row tiles per object."""
        return (FUSED_CHAIN[0] and m in (1, 2) and len(self.mlp) == 2 and self.mlp[0]['w'].shape[0] == 128 and self.mlp[0]['cout'] == 512 and
                self.mlp[1]['cout'] == 256 and self.mlp[0]['relu'] and self.mlp[1]['relu'] and length % 32 == 0 and rows * 1024 < 2 ** 31)

    def forward(self, pts_rows, groups, length, kv_plan=None):
        """-> memory rows (groups * length, E); with kv_plan (the cross-attention's _mha_plan) -> (memory, (K, V) or None): the key /
        value projections come out of the same kernel as the memory rows when dz_mlp_chain_forward applies."""
        mem = self._forward(pts_rows, groups, length, kv_plan)
        if kv_plan is None:
            return mem[0] if isinstance(mem, tuple) else mem
        return mem if isinstance(mem, tuple) else (mem, None)

    def _forward(self, pts_rows, groups, length, kv_plan):
        if _splittable(self.enc + self.mlp, pts_rows.shape[0]) and self.mlp[0]['cout'] % 32 == 0:
            m = _REFINE_MATH[0]
            # max over the points fused; the tapped layer pair16; 16- / 32-column fp32 rows go in as they are
            pooled, outs = _run_stack_split(pts_rows.contiguous(), self.enc, gmax=(groups, length), tap=True, x_f32=True)
            gshift = ops.linear(pooled, self.w_pool, self.ones, self.zeros, False, self.w_pool.shape[1])
            l0 = self.mlp[0]
            w0 = _split_w(l0, m)
            if outs[1] is not None and self._chain_ok(pts_rows.shape[0], length, m):
                l1 = self.mlp[1]
                w1 = _split_w(l1, m)
                kv = None
                if kv_plan is not None and kv_plan['e'] == 256:
                    if 'kv_split%d' % m not in kv_plan:
                        kv_plan['kv_split%d' % m] = (ops.pack_weight_split(kv_plan['wk'], m), ops.pack_weight_split(kv_plan['wv'], m))
                    wk, wv = kv_plan['kv_split%d' % m]
                    kv = (wk, kv_plan['bk'], wv, kv_plan['bv'])
                res = ops.mlp_chain(outs[1], (w0, l0['scale32'], l0['shift32']), (w1, l1['scale32'], l1['shift32']), gshift, length, m, kv=kv)
                return (res[0], (res[1], res[2])) if kv is not None else res
            only = len(self.mlp) == 1
            y = ops.linear_split(outs[1], w0, l0['scale32'], l0['shift32'], l0['relu'], l0['cout'], m, out_f32=only,
                                 group_shift=gshift, group_rows=length)
            if not only:
                y, _ = _run_stack_split(y, self.mlp[1:])
            return y
        x = _pad_cols(pts_rows, self.cin_pad)
        feat, outs = _run_stack(x, self.enc)
        pooled = ops.group_max(feat, groups, length)                               # (G, Cpool)
        gshift = ops.linear(pooled, self.w_pool, self.ones, self.zeros, False, self.w_pool.shape[1])   # (G, C1) pre-BN addend
        l0 = self.mlp[0]
        y = ops.linear(outs[1], l0['w'], l0['scale'], l0['shift'], l0['relu'], l0['cout'], group_shift=gshift, group_rows=length)
        y, _ = _run_stack(y, self.mlp[1:])
        return y


# ------------------------------------------------------------------------------------------------
# GRM
# ------------------------------------------------------------------------------------------------
class GeometryTransformer(_Cached):
    def __init__(self, model_cfg, query_point_dims=None, memory_point_dims=None):
        super().__init__()
        self.model_cfg = model_cfg
        self.query_point_dims, self.memory_point_dims = query_point_dims, memory_point_dims
        self.embed_dims = model_cfg.get('EMBED_DIMS', 256)
        self.anchor_sizes = model_cfg.get('ANCHOR_SIZES', [[4.8, 1.8, 1.5], [10.0, 2.6, 3.2], [2.0, 1.0, 1.6]])
        e = self.embed_dims
        # (sic) the reference builds `memory_encoder` on query_point_dims and `query_encoder` on memory_point_dims
        self.memory_encoder = make_fc_layers(model_cfg.MEMORY_ENCODER, query_point_dims, e * 2, output_use_norm=True)
        self.memory_mlp = make_fc_layers(model_cfg.REGRESSION_MLP, e * 2 + model_cfg.MEMORY_ENCODER[1], e, output_use_norm=True)
        self.query_encoder = make_fc_layers(model_cfg.QUERY_ENCODER, memory_point_dims, e, output_use_norm=True)
        self.query_mlp = make_linear_layers(model_cfg.REGRESSION_MLP, e, e, output_use_norm=True)
        dec = dict(model_cfg.DECODER)
        if dec.pop('NAME') != 'GeometryHead':
            raise DetZeroHipError('GeometryTransformer: DECODER.NAME must be GeometryHead')
        self.decoder = GeometryHead(**dec)
        self.preds_dict = {}

    def plan(self):
        if self._plan is None:
            self._plan = {
                'memory': _PointNetPlan(self.memory_encoder, self.memory_mlp, self.query_point_dims, pooled_first=False),
                'q_enc': _stack_plan(self.query_encoder, cin_pad=_pad16(self.memory_point_dims)),
                'q_mlp': _stack_plan(self.query_mlp),
                'layer': decoder_layer_plan(self.decoder.decoder[0]),
                'ffn': ffn_plan(self.decoder.prediction_heads[0]),
                'anchors': torch.tensor(self.anchor_sizes, dtype=torch.float32, device=self.query_mlp[0].weight.device),
            }
        return self._plan

    @torch.no_grad()
    @_with_math
    def forward(self, data_dict):
        _inference_only(self)
        p = self.plan()
        m_pts = data_dict['geo_memory_points'].float()
        b, lm, cm = m_pts.shape
        memory = p['memory'].forward(m_pts.reshape(b * lm, cm), b, lm)                        # (B*Lm, E)
        q_pts = data_dict['geo_query_points'].float()
        _, pp, npts, cq = q_pts.shape
        qf, _ = _run_stack(_pad_cols(q_pts.reshape(b * pp * npts, cq), _pad16(cq)), p['q_enc'], gmax=(b * pp, npts))
        qf, _ = _run_stack(qf, p['q_mlp'])                                                       # (B*pp, E)
        qpos = data_dict['geo_query_boxes'][..., 3:6].float().reshape(b * pp, 3).contiguous()
        out = decoder_layer_forward(p['layer'], qf, memory, qpos, b, pp, lm)
        preds = ffn_forward(p['ffn'], out)
        e = self.embed_dims
        data_dict['query'] = qf.view(b, pp, e).permute(0, 2, 1)
        data_dict['memory'] = memory.view(b, lm, e).permute(0, 2, 1)
        self.preds_dict = {k: v.view(1, b, pp, -1) for k, v in preds.items()}                 # (layers, B, queries, ch)
        self.preds_dict['geo_query_num'] = data_dict['geo_query_num']
        data_dict['batch_box_preds'] = self.generate_predicted_boxes(self.preds_dict, p['anchors'])
        return data_dict

    def generate_predicted_boxes(self, preds, anchors):
        """geometry_transformer.py:91-116 + target_assign.py:74-89: per query pick the arg-max anchor class, size =
        reg * anchor + anchor, then average over the valid queries of each object (a few flops per object)."""
        cls, reg = preds['geometry_cls'][0], preds['geometry_reg'][0]            # (B, Q, 3), (B, Q, 9)
        b, q, n = cls.shape
        size = reg.reshape(b, q, n, 3) * anchors + anchors
        pick = cls.argmax(dim=-1)
        size = torch.gather(size, 2, pick[..., None, None].expand(b, q, 1, 3)).squeeze(2)        # (B, Q, 3)
        qn = torch.as_tensor(preds['geo_query_num'], device=cls.device).long()
        valid = (torch.arange(q, device=cls.device)[None, :] < qn[:, None]).float()
        mean = (size * valid[..., None]).sum(1) / qn[:, None].float()
        boxes = torch.zeros((b, 7), dtype=torch.float32, device=cls.device)
        boxes[:, 3:6] = mean
        return boxes


# ------------------------------------------------------------------------------------------------
# PRM
# ------------------------------------------------------------------------------------------------
class PositionTransformer(_Cached):
    MEM_PTS_PER_BOX = 48        # hard-coded in the reference (position_head.py:91)

    def __init__(self, model_cfg, query_point_dims=None, memory_point_dims=None):
        super().__init__()
        self.model_cfg = model_cfg
        self.query_point_dims, self.memory_point_dims = query_point_dims, memory_point_dims
        self.embed_dims = model_cfg.get('EMBED_DIMS', 256)
        e = self.embed_dims
        self.query_encoder = make_conv_layers(model_cfg.QUERY_ENCODER, query_point_dims, e, output_use_norm=True)
        self.query_mlp = make_fc_layers(model_cfg.REGRESSION_MLP, e, e, output_use_norm=True)
        self.memory_encoder = make_fc_layers(model_cfg.MEMORY_ENCODER, memory_point_dims, e, output_use_norm=True)
        self.memory_mlp = make_fc_layers(model_cfg.REGRESSION_MLP, e + model_cfg.MEMORY_ENCODER[1], e, output_use_norm=True)
        dec = dict(model_cfg.DECODER)
        if dec.pop('NAME') != 'PositionHead':
            raise DetZeroHipError('PositionTransformer: DECODER.NAME must be PositionHead')
        self.decoder = PositionHead(**dec)
        self.preds_dict = {}
        self.dir_bin_num = 12

    def plan(self):
        if self._plan is None:
            dev = self.query_mlp[0].weight.device
            self._plan = {
                'memory': _PointNetPlan(self.memory_encoder, self.memory_mlp, self.memory_point_dims, pooled_first=True),
                'q_enc': _stack_plan(self.query_encoder, cin_pad=_pad16(self.query_point_dims)),
                'q_mlp': _stack_plan(self.query_mlp),
                'layer': decoder_layer_plan(self.decoder.decoder[0]),
                'ffn': ffn_plan(self.decoder.prediction_heads[0]),
                'anchor_angles': torch.arange(12, dtype=torch.float32, device=dev) * (2 * np.pi / 12) - np.pi,
            }
        return self._plan

    @torch.no_grad()
    @_with_math
    def forward(self, data_dict):
        _inference_only(self)
        p = self.plan()
        local_pts = data_dict['pos_query_points'].float()
        global_pts = data_dict['pos_memory_points'].float()
        traj = data_dict['pos_trajectory'].float()
        b, nb, npts, cq = local_pts.shape
        gp = global_pts.shape[2]
        if gp != self.MEM_PTS_PER_BOX:
            raise DetZeroHipError('PositionTransformer: %d memory points per box (the reference hard-codes 48)' % gp)
        e = self.embed_dims
        qf, _ = _run_stack(_pad_cols(local_pts.reshape(b * nb * npts, cq), _pad16(cq)), p['q_enc'], gmax=(b * nb, npts))
        qf, _ = _run_stack(qf, p['q_mlp'])                                                       # (B*nb, E)
        qpos = torch.cat([traj[..., :3], traj[..., 6:]], dim=-1).reshape(b * nb, -1).contiguous()
        lk = nb * gp
        # (B*Lk, E); the cross-attention's K / V come with it when the fused chain runs, unless the attention folds them away
        fold = FOLDED_XATTN[0] and lk >= FOLD_MIN_KEYS and ops.xattn_folded_supported(nb, e, p['layer']['ca']['heads'])
        if fold:
            memory, mkv = p['memory'].forward(global_pts.reshape(b * lk, global_pts.shape[3]), b, lk), None
        else:
            memory, mkv = p['memory'].forward(global_pts.reshape(b * lk, global_pts.shape[3]), b, lk, kv_plan=p['layer']['ca'])
        kpm = data_dict['padding_mask'].to(torch.bool)
        ca = kpm.reshape(b, nb, 1).repeat(1, 1, gp).reshape(b, -1)
        out = decoder_layer_forward(p['layer'], qf, memory, qpos, b, nb, lk, sa_mask=kpm, ca_mask=ca, memory_kv=mkv)
        preds = {k: v.view(b, nb, -1) for k, v in ffn_forward(p['ffn'], out).items()}
        preds['size_reg'] = traj[:, :, 3:6]
        data_dict['query'] = qf.view(b, nb, e).permute(0, 2, 1)
        data_dict['memory'] = memory.view(b, lk, e).permute(0, 2, 1)
        data_dict['query_pos'] = qpos.view(b, nb, -1)
        self.preds_dict = preds
        # target_assign.py:91-102
        center = preds['center_reg'] + traj[:, :, :3]
        dir_reg = preds['heading_reg'] * (np.pi / self.dir_bin_num) + p['anchor_angles']
        pick = preds['heading_cls'].argmax(dim=-1, keepdim=True)
        heading = torch.gather(dir_reg, 2, pick)
        data_dict['batch_box_preds'] = torch.cat([center, preds['size_reg'], heading], dim=-1)
        return data_dict


# ------------------------------------------------------------------------------------------------
# CRM
# ------------------------------------------------------------------------------------------------
class ConfidencePointnet(_Cached):
    """refining/detzero_refine/models/modules/confidence_pointnet.py:9-116 (inference): two nested PointNets - points of a
    box (encoder, max, concat, encoder, max), then boxes of a track (MLP, max, concat, MLP) - and two sigmoid heads;
    `pred_score = sqrt(score_reg * iou_reg)` per box.  Same module tree / state_dict as the reference.  Like the reference
    it does not mask the padded boxes: their zero rows take part in the max over the track."""

    def __init__(self, model_cfg, query_point_dims=None, memory_point_dims=None):
        super().__init__()
        self.model_cfg = model_cfg
        self.query_point_dims = query_point_dims
        self.embed_dims = model_cfg.get('EMBED_DIMS', 256)
        e = self.embed_dims
        self.pts_encoder_1 = make_conv_layers(model_cfg.ENCODER_MLP, query_point_dims, e, output_use_norm=True)
        self.pts_encoder_2 = make_conv_layers([], e + model_cfg.ENCODER_MLP[1], e, output_use_norm=True)
        self.pts_mlp = make_fc_layers(model_cfg.REGRESSION_MLP, e, e, output_use_norm=True)
        self.regression_mlp = make_fc_layers(model_cfg.REGRESSION_MLP, e * 2, e, output_use_norm=True)
        self.heads = nn.ModuleDict()
        self.preds_dict = {}
        self.score_thresh = model_cfg.get('SCORE_THRESH', [0.25, 0.5])
        self.tasks = {'score_reg': 1, 'iou_reg': 1}
        for task in self.tasks:
            self.heads[task] = make_fc_layers([int(e / 2)], e, self.tasks[task], output_use_norm=False)

    def plan(self):
        if self._plan is None:
            if len(list(self.pts_mlp)) < 6:
                raise DetZeroHipError('ConfidencePointnet: REGRESSION_MLP must have at least one hidden layer (the reference taps pts_mlp[5])')
            reg = _stack_plan(self.regression_mlp)
            first = list(self.regression_mlp)[0]
            w = first.weight.detach().float().reshape(first.weight.shape[0], first.weight.shape[1]).t().contiguous()     # (2E, C1)
            e = self.embed_dims
            reg[0]['w'] = w[e:].contiguous()                       # cat([pooled, tapped]) (confidence_pointnet.py:104-105): tapped rows
            self._plan = {
                'points': _PointNetPlan(self.pts_encoder_1, self.pts_encoder_2, self.query_point_dims, pooled_first=True),
                'boxes': _stack_plan(self.pts_mlp),
                'reg': reg, 'reg_w_pool': w[:e].contiguous(),
                'ones': torch.ones(w.shape[1], device=w.device), 'zeros': torch.zeros(w.shape[1], device=w.device),
                'heads': {k: _stack_plan(v) for k, v in self.heads.items()},
            }
        return self._plan

    @torch.no_grad()
    @_with_math
    def forward(self, data_dict):
        _inference_only(self)
        p = self.plan()
        pts = data_dict['conf_points'].float()
        b, nb, npts, c = pts.shape
        feat = p['points'].forward(pts.reshape(b * nb * npts, c), b * nb, npts)              # (B*nb*npts, E)
        box = ops.group_max(feat, b * nb, npts)                                              # (B*nb, E)
        box, outs = _run_stack(box, p['boxes'])
        tapped = outs[1]                                                                     # pts_mlp[5]
        pooled = ops.group_max(box, b, nb)                                                   # (B, E)
        gshift = ops.linear(pooled, p['reg_w_pool'], p['ones'], p['zeros'], False, p['reg_w_pool'].shape[1])
        l0 = p['reg'][0]
        y = ops.linear(tapped, l0['w'], l0['scale'], l0['shift'], l0['relu'], l0['cout'], group_shift=gshift, group_rows=nb)
        y, _ = _run_stack(y, p['reg'][1:])
        preds = {k: torch.sigmoid(_run_stack(y, layers)[0].view(b, nb, -1)) for k, layers in p['heads'].items()}
        self.preds_dict = preds
        data_dict['pred_score'] = torch.sqrt(preds['score_reg'].squeeze(2) * preds['iou_reg'].squeeze(2))
        return data_dict


__all__ = {'GeometryTransformer': GeometryTransformer, 'PositionTransformer': PositionTransformer,
           'ConfidencePointnet': ConfidencePointnet}
