#!/usr/bin/env python3
"""The clock the chip HOLDS under the fused 3x3 convolution (MI355X_MICROARCH.md, DVFS give-back item 6): the -DGNERF_CONV_STAMPS diagnostic build
stamps s_memtime / s_memrealtime at every workgroup's start and end; after >= 2 s of back-to-back launches on random data the quotient of the
differences x 100 MHz is the in-kernel clock, and matrix FLOP / (SIMDs x 1024 FLOP per cycle) / kernel cycles is the matrix pipe's share of them.
usage: GNERF_HIP_LIB=g-nerf_amd/gnerf_hip/variants/libgnerf_D:GNERF_CONV_STAMPS.so python tools/conv_clock.py [--zeros]"""
import os, sys, json, time, ctypes, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import numpy as np
import torch
import gnerf_hip

ap = argparse.ArgumentParser()
ap.add_argument('--zeros', action='store_true', help='all-zero operands (the clock the chip holds on trivial data, for contrast)')
ap.add_argument('--seconds', type=float, default=2.5)
args = ap.parse_args()
dev = torch.device('cuda', 0)
lib = gnerf_hip.load()
fn = lib.gnerf_debug_conv_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]; fn.restype = ctypes.c_int
for (n, cin, cout, h, w) in [(4, 128, 128, 512, 512), (4, 256, 256, 256, 256)]:
    g = torch.Generator(device='cpu').manual_seed(1)
    x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
    if args.zeros: x.zero_(); wt.zero_()
    wpk = gnerf_hip.pack_conv3x3_weights(wt)
    dco = (torch.rand(n, cout, generator=g) + 0.5).to(dev); nxt = (torch.rand(n, cout, generator=g) + 0.5).to(dev)
    bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
    run = lambda: gnerf_hip.conv3x3_epilogue(x, wpk, bias, scale=dco, next_scale=nxt, gain=2 ** 0.5, clamp=256.0)
    run(); torch.cuda.synchronize()
    t0 = time.time(); launches = 0
    while time.time() - t0 < args.seconds:
        for _ in range(200): run()
        torch.cuda.synchronize(); launches += 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 200
    wgs = n * (h // 8) * (w // 32) * (cout // 128)
    slots = min(wgs, 16384)
    st = np.zeros((slots, 8), dtype=np.uint64)
    assert fn(st.ctypes.data, slots) == 0
    st = st.astype(np.float64)
    cyc, ticks = st[:, 1] - st[:, 0], st[:, 3] - st[:, 2]
    ok = ticks > 0
    clock = float(np.median(cyc[ok] / ticks[ok])) * 100e6
    flop = 2.0 * n * h * w * cin * cout * 9
    simds = 256 * 4
    mfma_cycles = flop / 16384 * 16 / simds                  # v_mfma_f32_16x16x32_f16: 16 384 FLOP, 16 cycles of its SIMD's matrix pipe
    kernel_cycles = ms * 1e-3 * clock
    seg = {'setup': st[ok, 4] - st[ok, 0], 'first_tile_and_weights_arrive': st[ok, 5] - st[ok, 4], 'main_loop': st[ok, 6] - st[ok, 5],
           'barrier_and_epilogue_to_lds': st[ok, 7] - st[ok, 6], 'store_loop_to_end': st[ok, 1] - st[ok, 7]}
    span = (st[ok, 3].max() - st[ok, 2].min()) / 100.0       # us from the first workgroup's start to the last one's end (last launch)
    print(json.dumps({'shape': [n, cin, cout, h, w], 'operands': 'zeros' if args.zeros else 'random', 'launches_before': launches, 'ms': round(ms, 4),
                      'PFLOPs': round(flop / ms * 1e-12, 4), 'in_kernel_clock_GHz': round(clock * 1e-9, 3),
                      'peak_at_that_clock_PFLOPs': round(2.5 * clock / 2.4e9, 3), 'frac_of_peak_at_that_clock': round(flop / ms * 1e-12 / (2.5 * clock / 2.4e9), 3),
                      'frac_of_2.5_PFLOPs': round(flop / ms * 1e-12 / 2.5, 3), 'matrix_pipe_cycles_per_simd': round(mfma_cycles), 'kernel_cycles': round(kernel_cycles),
                      'workgroup_cycles_median': float(np.median(cyc[ok])), 'workgroup_us_median': float(np.median(ticks[ok])) / 100.0, 'first_start_to_last_end_us': round(span, 1),
                      'workgroup_segments_cycles_median': {k: float(np.median(v)) for k, v in seg.items()},
                      'note': 'diagnostic build: the stamps cost two scalar memory-clock reads per workgroup; ms is this build\'s'}))
