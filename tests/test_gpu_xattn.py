"""dz_xattn_folded (csrc/xattn_fold.hip): cross-attention of a few queries over a long memory without projecting the memory, against
torch.nn.functional-free float64 arithmetic of multi_head_attention.py:199-288 (the key bias included there, to show it cancels) and
against the library's own project-then-attend path (dz_mha_core)."""
import numpy as np
import pytest
import torch


def _mha64(x_q, mem, wq, bq, wk, bk, wv, bv, heads, mask):
    b, lq, e = x_q.shape
    hd = e // heads
    q = (x_q @ wq.T + bq) * hd ** -0.5
    k = mem @ wk.T + bk
    v = mem @ wv.T + bv
    q = q.reshape(b, lq, heads, hd).transpose(0, 2, 1, 3)
    k = k.reshape(b, -1, heads, hd).transpose(0, 2, 1, 3)
    v = v.reshape(b, -1, heads, hd).transpose(0, 2, 1, 3)
    s = q @ k.transpose(0, 1, 3, 2)
    if mask is not None:
        s = np.where(mask[:, None, None, :] != 0, -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    p = p / p.sum(-1, keepdims=True)
    return (p @ v).transpose(0, 2, 1, 3).reshape(b, lq, e)


@pytest.mark.gpu
@pytest.mark.parametrize('b,lq,lk,heads,masked', [(5, 3, 4096, 8, False), (2, 4, 1000, 8, True), (1, 1, 300, 8, False), (130, 3, 4096, 8, False),
                                                 (3, 2, 523, 16, True), (2, 8, 2048, 4, False)])
def test_folded_cross_attention(device, b, lq, lk, heads, masked):
    from detzero_amd import ops
    e = 256
    rng = np.random.default_rng(b * 1000 + lk)
    x_q = rng.standard_normal((b, lq, e))
    mem = rng.standard_normal((b, lk, e)) * 1.5
    wq, wk, wv = (rng.standard_normal((e, e)) / np.sqrt(e) for _ in range(3))
    bq, bk, bv = (0.2 * rng.standard_normal(e) for _ in range(3))
    mask = None
    if masked:
        mask = (rng.random((b, lk)) < 0.3).astype(np.uint8)
        mask[0, 64:160] = 1                    # whole 16-key blocks without a live key
        mask[-1, :48] = 1                      # ... also at the start of a wave's range (running maximum still -inf)
    want = _mha64(x_q, mem, wq, bq, wk, bk, wv, bv, heads, mask)

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    assert ops.xattn_folded_supported(lq, e, heads)
    q = t(x_q @ wq.T + bq)
    tm = None if mask is None else torch.from_numpy(mask).to(device)
    got = ops.xattn_folded(q, t(mem), tm, t(wk), t(wv.T), t(bv), heads, (e // heads) ** -0.5)
    torch.cuda.synchronize()
    err = float(np.abs(got.cpu().numpy() - want).max())
    # the library's project-then-attend path on the same inputs
    one = torch.ones(e, device=device)
    k = ops.linear(t(mem).view(b * lk, e), t(wk.T), one, t(bk), False, e).view(b, lk, e)
    v = ops.linear(t(mem).view(b * lk, e), t(wv.T), one, t(bv), False, e).view(b, lk, e)
    if heads * 32 == e:
        old = ops.mha_core(q, k, v, tm, heads, (e // heads) ** -0.5)
        err_old = float(np.abs(old.cpu().numpy() - want).max())
    else:
        err_old = float('nan')
    print('folded cross-attention b %d lq %d lk %d heads %d: |folded - f64| %.2e, |projected - f64| %.2e' % (b, lq, lk, heads, err, err_old))
    assert err <= 2e-5


@pytest.mark.gpu
def test_folded_cross_attention_all_keys_masked_is_nan(device):
    """torch.softmax over a fully masked row is NaN (multi_head_attention.py:262-268 masks with -inf); so is ours - for that object only."""
    from detzero_amd import ops
    e, b, lq, lk = 256, 2, 3, 512
    g = torch.Generator().manual_seed(3)
    q, mem = torch.randn((b, lq, e), generator=g).to(device), torch.randn((b, lk, e), generator=g).to(device)
    wk, wv, bv = torch.randn((e, e), generator=g).to(device) / 16, torch.randn((e, e), generator=g).to(device) / 16, torch.zeros(e, device=device)
    mask = torch.zeros((b, lk), dtype=torch.uint8, device=device)
    mask[1] = 1
    out = ops.xattn_folded(q, mem, mask, wk, wv, bv, 8, 32 ** -0.5)
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1]).all()


@pytest.mark.gpu
def test_folded_cross_attention_refuses_what_it_cannot_fold(device):
    from detzero_amd import ops
    from detzero_amd.lib import DetZeroHipError
    assert not ops.xattn_folded_supported(5, 256, 8) and not ops.xattn_folded_supported(3, 128, 4)
    x = torch.zeros((1, 5, 256), device=device)
    with pytest.raises(DetZeroHipError):
        ops.xattn_folded(x, torch.zeros((1, 64, 256), device=device), None, torch.zeros((256, 256), device=device), torch.zeros((256, 256), device=device),
                         torch.zeros(256, device=device), 8, 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize('name,mid,tol', [('f16x2', 1, 1.2e-5), ('bf16x2', 2, 3e-4)])       # 10x the observed 1.2e-6 / 2.9e-5
@pytest.mark.parametrize('b,lq,lk,heads,masked', [(3, 200, 9600, 8, True), (2, 37, 1000, 8, False), (1, 300, 130, 4, True), (2, 5, 64, 8, True)])
def test_attention_core_on_split_operands(device, name, mid, tol, b, lq, lk, heads, masked):
    """dz_mha_core_split (q, k, v and the probabilities as 16-bit pairs on the 16-bit matrix cores; multi_head_attention.py:207-288) against
    float64 on the host and against the exact-fp32 core (dz_mha_core) on the same inputs, key padding masks with whole dead blocks,
    ragged query / key counts."""
    from detzero_amd import ops
    e = heads * 32
    rng = np.random.default_rng(lq * 7 + lk)
    q = rng.standard_normal((b, lq, e))
    k = rng.standard_normal((b, lk, e))
    v = rng.standard_normal((b, lk, e)) * 2.0
    mask = None
    if masked:
        mask = (rng.random((b, lk)) < 0.35).astype(np.uint8)
        if lk >= 256:
            mask[0, 64 * (lk // 128):64 * (lk // 128) + 64] = 1      # a whole staged block without a live key
        mask[-1, :min(70, lk - 1)] = 1                                # ... and at the start (running maximum still -inf)
        mask[:, lk - 1] = 0                                           # (a fully masked row is NaN in both: tested for dz_xattn_folded)
    scale = 32 ** -0.5
    qq = q.reshape(b, lq, heads, 32).transpose(0, 2, 1, 3) * scale
    kk = k.reshape(b, lk, heads, 32).transpose(0, 2, 1, 3)
    vv = v.reshape(b, lk, heads, 32).transpose(0, 2, 1, 3)
    s = qq @ kk.transpose(0, 1, 3, 2)
    if mask is not None:
        s = np.where(mask[:, None, None, :] != 0, -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    want = ((p / p.sum(-1, keepdims=True)) @ vv).transpose(0, 2, 1, 3).reshape(b, lq, e)

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    tm = None if mask is None else torch.from_numpy(mask).to(device)
    got = ops.mha_core(t(q), t(k), t(v), tm, heads, scale, math=mid)
    ref = ops.mha_core(t(q), t(k), t(v), tm, heads, scale)
    torch.cuda.synchronize()
    err, err32 = float(np.abs(got.cpu().numpy() - want).max()), float(np.abs(ref.cpu().numpy() - want).max())
    print('attention core %s b %d lq %d lk %d heads %d: |split - f64| %.2e, |fp32 core - f64| %.2e' % (name, b, lq, lk, heads, err, err32))
    assert err <= tol


@pytest.mark.gpu
@pytest.mark.parametrize('r,l,e', [(9, 216, 192), (3, 100, 192), (5, 256, 128), (4, 40, 192), (2, 216, 64)])
def test_single_head_attention_of_the_pdv_encoder(device, r, l, e):
    """dz_attention_single_head (nn.MultiheadAttention with one head over the grid points of a RoI, attention_utils.py:17-52): the
    workgroup-per-sequence kernel with K / V staged in LDS (64 < L <= 256, E = 192 / 128), and the per-wave kernel it falls back to,
    against float64 on the host; key padding masks with fully dead 16-key tiles."""
    from detzero_amd import pdv_modules as pm
    rng = np.random.default_rng(r * 100 + l)
    q, k, v = (rng.standard_normal((r, l, e)) for _ in range(3))
    mask = (rng.random((r, l)) < 0.3).astype(np.uint8)
    mask[0, 16:48] = 1
    mask[-1, :min(20, l - 1)] = 1
    mask[:, l - 1] = 0
    scale = float(e) ** -0.5
    s = (q * scale) @ k.transpose(0, 2, 1)
    s = np.where(mask[:, None, :] != 0, -np.inf, s)
    s = s - s.max(-1, keepdims=True)
    p = np.exp(s)
    want = (p / p.sum(-1, keepdims=True)) @ v

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    got = pm.attention_single_head(t(q), t(k), t(v), torch.from_numpy(mask.astype(bool)).to(device), scale)
    err = float(np.abs(got.cpu().numpy() - want).max())
    print('single-head attention r %d l %d e %d: |got - f64| %.2e' % (r, l, e, err))
    assert err <= 2e-5
