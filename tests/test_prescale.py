"""Host logic of the exact power-of-two pre-scales of the fp16-pair arithmetic (round-5 review, item 2a): `ops.weight_prescale`
(per-output-channel weight factors) and `centerpoint.prescale_exponents` (per-stage activation exponents).  CPU only."""
import math

import torch

from detzero_amd import ops
from detzero_amd.centerpoint import F16_PAIR_TARGET_PEAK, PRESCALE_STAGES, prescale_exponents


def test_weight_prescale_is_exact_and_places_every_channel():
    g = torch.Generator().manual_seed(0)
    for shape in ((27, 16, 16), (9, 128, 256), (6, 9, 64, 16), (64, 128)):
        w = torch.randn(shape, generator=g) * torch.logspace(-6, 2, shape[-1])          # channels from 1e-6 to 1e2
        w[..., 3] = 0.0                                                                  # an all-zero output channel keeps factor 1
        ws, inv = ops.weight_prescale(w)
        red = (-3, -2) if w.dim() >= 3 else (-2,)
        assert inv.shape == w.abs().amax(dim=red).shape
        back = ws * inv.reshape(inv.shape[:-1] + (1,) * len(red) + inv.shape[-1:])
        assert torch.equal(back, w)                                                      # powers of two: bit-exact round trip
        top = ws.abs().amax(dim=red)
        live = w.abs().amax(dim=red) > 0
        assert bool(((top[live] >= 2.0 ** 13) & (top[live] < 2.0 ** 15)).all())
        assert bool((inv[~live] == 1.0).all())
        m, e = torch.frexp(inv)
        assert bool((m == 0.5).all())                                                    # every factor is a power of two


def test_prescale_exponents():
    peaks = {'x_conv1': 37.5, 'x_conv2': 1.0e6, 'x_conv3': 2.0 ** -20, 'x_conv4': 2048.0, 'encoded': 0.0, 'spatial_features_2d': float('inf')}
    e = prescale_exponents(peaks)
    assert set(e) == set(PRESCALE_STAGES)
    for k in ('x_conv1', 'x_conv2', 'x_conv3', 'x_conv4'):
        assert F16_PAIR_TARGET_PEAK / 2 < peaks[k] * 2.0 ** e[k] <= F16_PAIR_TARGET_PEAK
    assert e['x_conv4'] == 0 and e['encoded'] == 0 and e['spatial_features_2d'] == 0 and e['x_conv2'] == -9 and e['x_conv3'] == 31
    assert math.log2(F16_PAIR_TARGET_PEAK) == 11
