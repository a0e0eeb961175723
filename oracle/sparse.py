"""ORACLE (test infrastructure only): spconv-style sparse 3-D convolution, restated on CPU.

spconv is an un-vendored pip dependency of the reference ("spconv v2.x", docs/INSTALL.md:10) and
cannot be installed here -> PARITY UNPINNED for this file.  Semantics follow SURVEY.md Appendix C
and are cross-checked against dense ``torch.nn.functional.conv3d`` in tests/test_oracle_sparse.py.
Network topology: /root/reference/detection/detzero_det/models/centerpoint_modules/backbone3d.py:231-338
(VoxelResBackBone8x), :85-121 (SparseBasicBlock), :64-83 (post_act_block);
height_compression.py:20-24 for the dense BEV map.
"""
import numpy as np
import torch


def lin_key(coords, shape):
    """((b*D+z)*H+y)*W+x as int64; coords (M,4) [b,z,y,x]."""
    d, h, w = (int(s) for s in shape)
    c = coords.astype(np.int64)
    return ((c[:, 0] * d + c[:, 1]) * h + c[:, 2]) * w + c[:, 3]


def canonical_order(coords, shape):
    return np.argsort(lin_key(coords, shape), kind='stable')


def out_shape_of(shape, k, s, p):
    return [(int(shape[i]) + 2 * p[i] - k[i]) // s[i] + 1 for i in range(3)]


def conv_out_coords(coords, shape, k, s, p):
    """Active output set of a regular (strided) sparse conv: every w with >=1 active input in its
    window; sorted by linear key (canonical order)."""
    oshape = out_shape_of(shape, k, s, p)
    c = coords.astype(np.int64)
    outs = []
    for tz in range(k[0]):
        for ty in range(k[1]):
            for tx in range(k[2]):
                nz = c[:, 1] + p[0] - tz
                ny = c[:, 2] + p[1] - ty
                nx = c[:, 3] + p[2] - tx
                ok = (nz % s[0] == 0) & (ny % s[1] == 0) & (nx % s[2] == 0)
                wz, wy, wx = nz // s[0], ny // s[1], nx // s[2]
                ok &= (wz >= 0) & (wz < oshape[0]) & (wy >= 0) & (wy < oshape[1]) & (wx >= 0) & (wx < oshape[2])
                outs.append(np.stack([c[ok, 0], wz[ok], wy[ok], wx[ok]], axis=1))
    allc = np.concatenate(outs, axis=0)
    key = lin_key(allc, oshape)
    _, first = np.unique(key, return_index=True)
    return allc[first].astype(np.int32), oshape


def build_rulebook(in_coords, in_shape, out_coords, k, s, p):
    """All (in_idx, out_idx, tap) with in = out*s - p + tap active.  tap = (tz*kH+ty)*kW+tx.
    SubM conv = (k=3, s=1, p=1, out_coords = in_coords)."""
    in_key = lin_key(in_coords, in_shape)
    order = np.argsort(in_key, kind='stable')
    skey = in_key[order]
    d, h, w = (int(x) for x in in_shape)
    oc = out_coords.astype(np.int64)
    ins, outs, taps = [], [], []
    for tz in range(k[0]):
        for ty in range(k[1]):
            for tx in range(k[2]):
                uz = oc[:, 1] * s[0] - p[0] + tz
                uy = oc[:, 2] * s[1] - p[1] + ty
                ux = oc[:, 3] * s[2] - p[2] + tx
                ok = (uz >= 0) & (uz < d) & (uy >= 0) & (uy < h) & (ux >= 0) & (ux < w)
                q = ((oc[:, 0] * d + uz) * h + uy) * w + ux
                pos = np.searchsorted(skey, q)
                pos_c = np.minimum(pos, skey.size - 1)
                hit = ok & (skey[pos_c] == q) if skey.size else np.zeros_like(ok)
                oi = np.nonzero(hit)[0]
                ins.append(order[pos_c[oi]])
                outs.append(oi)
                taps.append(np.full(oi.size, (tz * k[1] + ty) * k[2] + tx, dtype=np.int64))
    return np.concatenate(ins), np.concatenate(outs), np.concatenate(taps)


def neighbor_table(in_coords, in_shape, out_coords, k, s, p):
    """Dense (kvol, M_out) int32 table, -1 where the tap has no active input (same information as
    the rulebook, output-stationary form)."""
    i, o, t = build_rulebook(in_coords, in_shape, out_coords, k, s, p)
    tab = np.full((k[0] * k[1] * k[2], out_coords.shape[0]), -1, dtype=np.int32)
    tab[t, o] = i
    return tab


def weight_to_taps(w):
    """spconv-2.x layout (Cout,kD,kH,kW,Cin) -> (kvol, Cin, Cout)."""
    co, kd, kh, kw, ci = w.shape
    return w.permute(1, 2, 3, 4, 0).reshape(kd * kh * kw, ci, co).contiguous()


def sparse_conv(feats, rulebook, w_taps, m_out, bias=None):
    """out[o] = bias + sum_t feats[i] @ W[t] over rulebook triples; fp32 torch CPU; taps ascending."""
    i, o, t = rulebook
    out = torch.zeros((m_out, w_taps.shape[2]), dtype=feats.dtype)
    for tap in range(w_taps.shape[0]):
        sel = np.nonzero(t == tap)[0]
        if sel.size == 0:
            continue
        ii = torch.from_numpy(i[sel]); oo = torch.from_numpy(o[sel])
        out.index_add_(0, oo, feats[ii] @ w_taps[tap])
    if bias is not None:
        out = out + bias[None, :]
    return out


def bn_eval(x, sd, prefix, eps):
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    m, v = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    return (x - m) / torch.sqrt(v + eps) * w + b


class Level:
    def __init__(self, coords, shape):
        self.coords = coords
        self.shape = list(shape)


def backbone_forward(sd, voxel_features, voxel_coords, sparse_shape, prefix='backbone3d.', last_pad=0, dtype=torch.float32):
    """VoxelResBackBone8x.forward (backbone3d.py:289-338) with BN in eval mode (eps 1e-3, :239).
    voxel_coords (M,4) int32 [b,z,y,x] in ANY order; internally re-ordered canonically.
    Returns dict with per-stage (features, coords, shape) and rulebooks for parity checks.
    dtype: torch.float32 = the reference's arithmetic; torch.float64 (with a float64 state dict) = the error-budget yardstick of
    tests/test_gpu_full_parity.py - the same fp32 weights and voxel features evaluated without rounding noise."""
    eps = 1e-3
    order = canonical_order(voxel_coords, sparse_shape)
    coords = voxel_coords[order]
    x = torch.from_numpy(np.ascontiguousarray(voxel_features[order])).to(dtype)
    out = {'rulebooks': {}}
    K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)

    def subm(x, coords, shape, key, wname, bias_name=None):
        if key not in out['rulebooks']:
            out['rulebooks'][key] = (build_rulebook(coords, shape, coords, K3, S1, P1), coords.shape[0])
        rb, m = out['rulebooks'][key]
        b = sd[prefix + bias_name] if bias_name else None
        return sparse_conv(x, rb, weight_to_taps(sd[prefix + wname]), m, b)

    def basic_block(x, coords, shape, key, name):
        idt = x
        y = subm(x, coords, shape, key, name + '.conv1.weight', name + '.conv1.bias')
        y = torch.relu(bn_eval(y, sd, prefix + name + '.bn1', eps))
        y = subm(y, coords, shape, key, name + '.conv2.weight', name + '.conv2.bias')
        y = bn_eval(y, sd, prefix + name + '.bn2', eps)
        return torch.relu(y + idt)

    def down(x, coords, shape, key, name, k, s, p):
        oc, oshape = conv_out_coords(coords, shape, k, s, p)
        rb = build_rulebook(coords, shape, oc, k, s, p)
        out['rulebooks'][key] = (rb, oc.shape[0])
        y = sparse_conv(x, rb, weight_to_taps(sd[prefix + name + '.0.weight']), oc.shape[0])
        y = torch.relu(bn_eval(y, sd, prefix + name + '.1', eps))
        return y, oc, oshape

    shape = list(sparse_shape)
    x = subm(x, coords, shape, 'subm1', 'conv_input.0.weight')
    x = torch.relu(bn_eval(x, sd, prefix + 'conv_input.1', eps))
    x = basic_block(x, coords, shape, 'res1', 'conv1.0')
    x = basic_block(x, coords, shape, 'res1', 'conv1.1')
    out['x_conv1'] = (x, coords, shape)
    pads = {2: (1, 1, 1), 3: (1, 1, 1), 4: (0, 1, 1)}
    for stage in (2, 3, 4):
        x, coords, shape = down(x, coords, shape, 'spconv%d' % stage, 'conv%d.0' % stage, (3, 3, 3), (2, 2, 2), pads[stage])
        x = basic_block(x, coords, shape, 'res%d' % stage, 'conv%d.1' % stage)
        x = basic_block(x, coords, shape, 'res%d' % stage, 'conv%d.2' % stage)
        out['x_conv%d' % stage] = (x, coords, shape)
    lp = (last_pad, last_pad, last_pad) if isinstance(last_pad, int) else tuple(last_pad)
    x, coords, shape = down(x, coords, shape, 'spconv_down2', 'conv_out', (3, 1, 1), (2, 1, 1), lp)
    out['encoded'] = (x, coords, shape)
    return out


def to_bev(feats, coords, shape, batch_size):
    """HeightCompression (height_compression.py:20-24): dense (B,C,D,H,W) -> (B, C*D, H, W)."""
    d, h, w = shape
    c = feats.shape[1]
    dense = torch.zeros((batch_size, c, d, h, w), dtype=feats.dtype)
    cc = torch.from_numpy(coords.astype(np.int64))
    dense[cc[:, 0], :, cc[:, 1], cc[:, 2], cc[:, 3]] = feats
    return dense.reshape(batch_size, c * d, h, w)
