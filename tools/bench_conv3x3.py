"""One BEV 3x3 layer (16 x 188 x 188, 128 -> 128 channels, pair16) timed alone: us per launch and algorithmic TF/s.
Development tool: with a -DDZ_C3_DIAG build, DZ_TUNE_C3_DIAG=<bits> removes one effect at a time (results are garbage)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from detzero_amd import ops                      # noqa: E402
from detzero_amd.det_modules import conv_layer   # noqa: E402


def q16_pack(x, e):
    """(..., C) fp32 -> same-shape float32-typed q16 bits: per 32 channels [hi fp16 x32 | fp8(hi * 2^-e) x32 | fp8((x - hi) * 2^(11-e)) x32]."""
    shp = x.shape
    xb = x.reshape(-1, shp[-1] // 32, 32).float()
    hi = xb.half()
    hf = hi.float()
    hi8 = (hf * 2.0 ** (-e)).clamp(-448, 448).to(torch.float8_e4m3fn)
    lo8 = ((xb - hf) * 2.0 ** (11 - e)).clamp(-448, 448).to(torch.float8_e4m3fn)
    by = torch.cat([hi.view(torch.uint8).reshape(*xb.shape[:2], 64), hi8.view(torch.uint8), lo8.view(torch.uint8)], dim=-1)
    return by.reshape(-1, shp[-1] * 4).view(torch.float32).reshape(shp)


def q16_unpack(q, e):
    shp = q.shape
    by = q.contiguous().view(torch.uint8).reshape(-1, shp[-1] // 32, 128)
    hi = by[..., :64].contiguous().view(torch.float16).float()
    lo = by[..., 96:].contiguous().view(torch.float8_e4m3fn).float() * 2.0 ** (e - 11)
    return (hi + lo).reshape(shp)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--hw', type=int, default=188)
    ap.add_argument('--cin', type=int, default=128)
    ap.add_argument('--cout', type=int, default=128)
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--math', type=int, default=1, help='1 = f16x2, 2 = bf16x2, 3 = f16 (single product)')
    ap.add_argument('--q16', action='store_true', help='fp16 + fp8 prototype (diag build, DZ_TUNE_C3_Q16=1): q16 tensors, checked against fp32 torch')
    ap.add_argument('--data', default='randn', choices=['randn', 'relu', 'zero', 'const'])
    a = ap.parse_args()
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    h = w = a.hw
    x = torch.zeros(a.batch, h + 2, w + 2, a.cin)
    x[:, 1:-1, 1:-1] = torch.randn(a.batch, h, w, a.cin, generator=g)
    wr = torch.randn(9, a.cin, a.cout, generator=g) * 0.05
    if a.data == 'relu':
        x = x.clamp_min(0)
    elif a.data == 'zero':
        x, wr = x * 0, wr * 0
    elif a.data == 'const':
        x, wr = (x != 0).float(), wr * 0 + 0.5
    ea, ew = int(os.environ.get('DZ_TUNE_Q16_EA', '2')), int(os.environ.get('DZ_TUNE_Q16_EW', '-4'))
    if a.q16:
        xp = q16_pack(x, ea).to(dev)
        wt = q16_pack(wr.transpose(-1, -2).contiguous(), ew).to(dev)
    else:
        xp = ops.pair16_from_f32(x.to(dev), math=1)
        wt = ops.pack_weight_split(wr.to(dev), 1)
    scale = torch.ones(wt.shape[-2], device=dev)
    shift = torch.zeros(wt.shape[-2], device=dev)
    y = torch.zeros(a.batch, h + 2, w + 2, a.cout, device=dev)

    def run():
        conv_layer(xp, (h + 2, w + 2), wt, scale, shift, True, y, (h + 2, w + 2), cin=a.cin, in_cstride=a.cin, ksize=3,
                   stride=1, in_off=0, out_cstride=a.cout, out_d=(1, 1), ho=h, wo=w, batch=a.batch, math=a.math)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    if a.q16 or os.environ.get('DZ_CHECK'):
        ref = torch.nn.functional.conv2d(x[:, 1:-1, 1:-1].permute(0, 3, 1, 2).double(), wr.reshape(3, 3, a.cin, a.cout).permute(3, 2, 0, 1).double(),
                                         padding=1).permute(0, 2, 3, 1).clamp_min(0)
        got = (q16_unpack(y.cpu(), ea) if a.q16 else ops.pair16_to_f32(y, 1).cpu())[:, 1:-1, 1:-1].double()
        print('max |err| / max |ref| = %.3e   mean %.3e' % (float((got - ref).abs().max() / ref.abs().max()), float((got - ref).abs().mean() / ref.abs().mean())))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    fl = 2.0 * a.batch * h * w * 9 * a.cin * a.cout
    print('conv3x3 %dx%dx%d %d->%d data=%s diag=%s  %.1f us  %.1f TF/s algorithmic' % (a.batch, h, w, a.cin, a.cout, a.data,
          os.environ.get('DZ_TUNE_C3_DIAG', '0'), us, fl / us * 1e-6))


if __name__ == '__main__':
    main()
