#!/usr/bin/env python3
"""filtered_lrelu at StyleGAN3-layer shapes: the fused gfx950 kernel vs the three-launch route (upfirdn2d ->
filtered_lrelu_act_ -> upfirdn2d, what runs when the fused kernel declines) vs the PyTorch-op form (what a checkout
without its plugins runs).  Algorithmic bytes = x + y (+ sign tensor when written); time = HIP events.
One JSON line per case."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import numpy as np
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import filtered_lrelu, upfirdn2d
import gnerf_hip

dev = torch.device('cuda', 0)


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def lowpass(taps, factor):
    k = np.arange(taps) - (taps - 1) / 2
    f = np.sinc(k / factor) * np.kaiser(taps, 8.0)
    return torch.tensor(f / f.sum(), dtype=torch.float32, device=dev)


CASES = [  # up, down, shape (StyleGAN3-T/R layers use 6 taps per branch; pad so that out = in * up / down)
    (2, 2, (4, 128, 256, 256)),
    (4, 2, (4, 128, 128, 128)),
    (2, 1, (4, 64, 256, 256)),
    (2, 4, (4, 128, 256, 256)),
]

with torch.no_grad():
    for up, down, shape in CASES:
        fu, fd = lowpass(6 * up, up), (lowpass(6 * down, down) if down > 1 else None)
        fut, fdt = 6 * up - 1, (6 * down - 1 if down > 1 else 0)
        tot = fut + fdt - (up - 1)            # padding that keeps out = in * up / down
        pad = [tot // 2 + tot % 2 + (up - 1), tot // 2] * 2
        for dt, nm in ((torch.float16, 'f16'), (torch.float32, 'f32')):
            es = 2 if dt == torch.float16 else 4
            x = torch.randn(*shape, device=dev, dtype=dt)
            b = torch.randn(shape[1], device=dev, dtype=dt)
            kw = dict(up=up, down=down, padding=pad, gain=1.414, slope=0.2, clamp=256)
            y = filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw)
            one = torch.ones([1, 1], device=dev)
            r = gnerf_hip.filtered_lrelu(x, fu, one if fd is None else fd, b, torch.empty([0]), up, down, *pad, 0, 0, 1.414, 0.2, 256.0, False, False)
            assert r[2] == 0 and torch.equal(r[0], y)
            nbytes = (x.numel() + y.numel()) * es
            t_fused = timeit(lambda: filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, **kw))
            t_signs = timeit(lambda: gnerf_hip.filtered_lrelu(x, fu, one if fd is None else fd, b, torch.empty([0]), up, down, *pad, 0, 0, 1.414, 0.2, 256.0, False, True))

            def three():
                t = x + b[None, :, None, None]
                t = upfirdn2d.upfirdn2d(t, fu, up=up, padding=pad, gain=up ** 2)
                gnerf_hip.filtered_lrelu_act_(t, torch.empty([0]), 0, 0, 1.414, 0.2, 256.0, False)
                return upfirdn2d.upfirdn2d(t, fd, down=down)
            y3 = three()
            err = float((y3.float() - y.float()).abs().max())
            t_three = timeit(three, reps=3)
            t_ref = timeit(lambda: filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, impl='ref', **kw), reps=2)
            print(json.dumps({'case': f'up{up} down{down} {list(shape)}->{list(y.shape[2:])} {nm}', 'fused_ms': round(t_fused, 4),
                              'fused_with_signs_ms': round(t_signs, 4), 'three_launch_ms': round(t_three, 4), 'pytorch_ops_ms': round(t_ref, 4),
                              'algorithmic_MB': round(nbytes / 1e6, 1), 'fused_GBs': round(nbytes / t_fused / 1e6, 1),
                              'fused_frac_of_8TBs': round(nbytes / t_fused / 1e6 / 8000, 3), 'max_abs_diff_vs_three_launch': err}), flush=True)
