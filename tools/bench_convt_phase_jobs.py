#!/usr/bin/env python3
"""The transposed convolution's three job shapes (GNERF_CONVT_PHASE_JOBS: 0 = a workgroup walks all four phases of a position tile, 1 = one phase per
workgroup, 2 = phase pairs {4 taps, 1 tap} / {2, 2}) on the shapes the generator runs, interleaved in ONE process (the launcher reads the variable
per call); results are checked equal between the modes.   usage: python tools/bench_convt_phase_jobs.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import gnerf_hip
dev = torch.device('cuda', 0)


def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return min(ts)


# (n, cin, cout, h, w): the superresolution's two x2 layers at 1 / 4 / 8 frames per call, the backbone's fp32-grade x2 layers (3 x cin split channels)
SHAPES = [(4, 256, 128, 256, 256), (8, 256, 128, 256, 256), (1, 256, 128, 256, 256), (4, 128, 128, 256, 256),
          (4, 32, 256, 128, 128), (8, 32, 256, 128, 128), (1, 32, 256, 128, 128),
          (4, 3 * 512, 512, 32, 32), (4, 3 * 512, 256, 64, 64), (4, 3 * 256, 128, 128, 128), (1, 3 * 512, 512, 32, 32)]
for (n, cin, cout, h, w) in SHAPES:
    g = torch.Generator(device='cpu').manual_seed(1)
    x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (2 * cin ** 0.5)).to(dev)
    wp = gnerf_hip.pack_conv_transpose3x3_weights(wt)
    jobs = n * ((h + 1 + 7) // 8) * ((w + 1 + 31) // 32) * (cout // 128)
    row = {'shape': [n, cin, cout, h, w], 'whole_tile_jobs': jobs, 'GFLOP': round(2e-9 * n * h * w * cin * cout * 9, 1)}
    outs = {}
    for rnd in range(2):
        for mode in ('0', '2', '1'):
            os.environ['GNERF_CONVT_PHASE_JOBS'] = mode
            if rnd == 0:
                outs[mode] = gnerf_hip.conv_transpose3x3_s2(x, wp)
            t = timeit(lambda: gnerf_hip.conv_transpose3x3_s2(x, wp))
            row[f'ms_mode{mode}'] = round(min(t, row.get(f'ms_mode{mode}', 1e9)), 4)
    os.environ.pop("GNERF_CONVT_PHASE_JOBS", None)
    t = timeit(lambda: gnerf_hip.conv_transpose3x3_s2(x, wp))
    row['ms_launcher_choice'] = round(t, 4)
    row['modes_equal'] = bool(torch.equal(outs['0'], outs['1']) and torch.equal(outs['0'], outs['2']))
    print(json.dumps(row), flush=True)
