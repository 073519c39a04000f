#!/usr/bin/env python3
"""GB/s of the HBM-bound custom ops at the shapes a G-NeRF forward issues.
Algorithmic bytes = tensors in + tensors out; time = HIP events on the launch stream.  Prints one JSON line per case."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
from torch_utils.ops import bias_act, upfirdn2d
import gnerf_hip

dev = torch.device('cuda', 0)
PEAK = 8000.0


def timeit(fn, reps=20):
    for _ in range(3):
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


def report(name, ms, nbytes):
    gbs = nbytes / ms / 1e6
    print(json.dumps({'op': name, 'ms': round(ms, 4), 'algorithmic_MB': round(nbytes / 1e6, 2), 'GBs': round(gbs, 1), 'frac_of_8TBs': round(gbs / PEAK, 3)}))


with torch.no_grad():
    f = upfirdn2d.setup_filter([1, 3, 3, 1], device=dev)
    for dt, nm in ((torch.float16, 'f16'), (torch.float32, 'f32')):
        es = 2 if dt == torch.float16 else 4
        x = torch.randn(4, 128, 512, 512, device=dev, dtype=dt)
        b = torch.randn(128, device=dev, dtype=dt)
        ms = timeit(lambda: bias_act.bias_act(x, b, act='lrelu', clamp=256))
        report(f'bias_act lrelu+clamp [4,128,512,512] {nm}', ms, (2 * x.numel() + 128) * es)
        xcl = x.contiguous(memory_format=torch.channels_last)
        ms = timeit(lambda: bias_act.bias_act(xcl, b, act='lrelu', clamp=256))
        report(f'bias_act lrelu+clamp [4,128,512,512] {nm} channels_last', ms, (2 * x.numel() + 128) * es)
        xb = torch.randn(4, 128, 513, 513, device=dev, dtype=dt)
        ms = timeit(lambda: upfirdn2d.upfirdn2d(xb, f, padding=[1, 1, 1, 1], gain=4))
        report(f'upfirdn2d blur 4x4 [4,128,513,513]->512 {nm}', ms, (xb.numel() + 4 * 128 * 512 * 512) * es)
        xu = torch.randn(4, 96, 128, 128, device=dev, dtype=dt)
        ms = timeit(lambda: upfirdn2d.upfirdn2d(xu, f, up=2, padding=[2, 1, 2, 1], gain=4))
        report(f'upfirdn2d up2 4x4 [4,96,128,128]->256 {nm}', ms, (xu.numel() + 4 * 96 * 256 * 256) * es)
        xi = torch.randn(4, 3, 256, 256, device=dev, dtype=dt)
        ms = timeit(lambda: upfirdn2d.upfirdn2d(xi, f, up=2, padding=[2, 1, 2, 1], gain=4))
        report(f'upfirdn2d up2 4x4 [4,3,256,256]->512 {nm}', ms, (xi.numel() + 4 * 3 * 512 * 512) * es)
    planes = torch.randn(4, 3, 32, 256, 256, device=dev)
    ms = timeit(lambda: gnerf_hip.planes_to_nhwc(planes))
    report('planes NCHW->NHWC [12,32,256,256] f32', ms, 2 * planes.numel() * 4)
    # round 2: the producer's fused last step (upsample2d + add, channels_last out, max|planes|), the modconv kernels
    img = torch.randn(4, 96, 128, 128, device=dev)
    yy = torch.randn(4, 96, 256, 256, device=dev)
    ms = timeit(lambda: gnerf_hip.upsample2x_add_nhwc(img, yy, f, with_absmax=True))
    report('upsample2d + add -> channels_last (+absmax) [4,96,128,128]->256 f32', ms, (img.numel() + 2 * yy.numel()) * 4)
    ms = timeit(lambda: upfirdn2d.upsample2d(img, f).add_(yy))
    report('(the same as two launches: upsample2d, add_, NCHW; the NCHW->NHWC repack would follow) f32', ms, (img.numel() + 2 * yy.numel()) * 4)
    ms = timeit(lambda: gnerf_hip.planes_absmax(yy))
    report('planes_absmax [4,96,256,256] f32', ms, yy.numel() * 4)
    for dt, nm in ((torch.float16, 'f16'), (torch.float32, 'f32')):
        es = 2 if dt == torch.float16 else 4
        x = torch.randn(4, 128, 512, 512, device=dev, dtype=dt)
        b = torch.randn(128, device=dev, dtype=dt)
        sc = torch.rand(4, 128, device=dev) + 0.5
        nz = torch.randn(512, 512, device=dev)
        ms = timeit(lambda: gnerf_hip.modconv_epilogue(x, b, scale=sc, noise=nz, round_noise=True, act='lrelu', gain=1.41, clamp=256))
        report(f'modconv_epilogue (demod + noise + bias + lrelu + clamp) [4,128,512,512] {nm}', ms, 2 * x.numel() * es + nz.numel() * 4)
        ms = timeit(lambda: gnerf_hip.scale_channels(x, sc))
        report(f'scale_channels [4,128,512,512] {nm}', ms, 2 * x.numel() * es)
    # channels_last forms (the fp16 blocks' layout on the fast path)
    for dt, nm in ((torch.float16, 'f16'),):
        es = 2
        x = torch.randn(4, 128, 512, 512, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
        b = torch.randn(128, device=dev, dtype=dt)
        sc = torch.rand(4, 128, device=dev) + 0.5
        ms = timeit(lambda: gnerf_hip.modconv_epilogue(x, b, scale=sc, act='lrelu', gain=1.41, clamp=256))
        report(f'channels_last modconv_epilogue (demod + bias + lrelu + clamp) [4,128,512,512] {nm}', ms, 2 * x.numel() * es)
        ms = timeit(lambda: gnerf_hip.modconv_epilogue(x, b, scale=sc, act='lrelu', gain=1.41, clamp=256, next_scale=sc))
        report(f'channels_last modconv_epilogue + next layer input scale [4,128,512,512] {nm}', ms, 2 * x.numel() * es)
        ms = timeit(lambda: gnerf_hip.scale_channels(x, sc))
        report(f'channels_last scale_channels [4,128,512,512] {nm}', ms, 2 * x.numel() * es)
        xb = torch.randn(4, 128, 513, 513, device=dev, dtype=dt).contiguous(memory_format=torch.channels_last)
        ms = timeit(lambda: upfirdn2d.upfirdn2d(xb, f, padding=[1, 1, 1, 1], gain=4))
        report(f'channels_last upfirdn2d blur 4x4 [4,128,513,513]->512 {nm}', ms, (xb.numel() + 4 * 128 * 512 * 512) * es)
        ms = timeit(lambda: gnerf_hip.blur_epilogue_channels_last(xb, f, [1, 1, 1, 1], blur_gain=4, bias=b, scale=sc, act='lrelu', gain=1.41, clamp=256, next_scale=sc))
        report(f'channels_last blur 4x4 + epilogue (+ next scale) in one pass [4,128,513,513]->512 {nm}', ms, (xb.numel() + 4 * 128 * 512 * 512) * es)
        ms = timeit(lambda: gnerf_hip.modconv_epilogue(upfirdn2d.upfirdn2d(xb, f, padding=[1, 1, 1, 1], gain=4), b, scale=sc, act='lrelu', gain=1.41, clamp=256, next_scale=sc))
        report(f'(the same as two passes) {nm}', ms, (xb.numel() + 3 * 4 * 128 * 512 * 512) * es)
        wrgb, srgb, brgb = torch.randn(3, 128, 1, 1, device=dev), torch.randn(4, 128, device=dev) / 11, torch.randn(3, device=dev)
        ms = timeit(lambda: gnerf_hip.torgb_channels_last(x, wrgb, srgb, brgb, clamp=256))
        report(f'channels_last ToRGB (modulated 1x1 conv to 3 ch + bias + clamp) [4,128,512,512] {nm}', ms, x.numel() * es + 4 * 3 * 512 * 512 * es)
    wgt, sty = torch.randn(512, 512, 3, 3, device=dev), torch.randn(4, 512, device=dev)
    ms = timeit(lambda: gnerf_hip.modulate_weights(wgt, sty, True, out_dtype=torch.float32))
    report('modulate_weights [512,512,3,3] x 4 styles -> f32', ms, wgt.numel() * 4 + 4 * wgt.numel() * 4)
    import gnerf_generator as GG
    ms = timeit(lambda: GG._modulated_weights(wgt, sty, True, False))
    report('(the PyTorch-op chain it replaces: 7 launches) f32', ms, wgt.numel() * 4 + 4 * wgt.numel() * 4)
    # torch's own elementwise path for scale: same bytes as bias_act
    x = torch.randn(4, 128, 512, 512, device=dev, dtype=torch.float16)
    ms = timeit(lambda: torch.nn.functional.leaky_relu(x, 0.2))
    report('(torch leaky_relu, same bytes, for scale) f16', ms, 2 * x.numel() * 2)

# grid_sample_gradfix's sampler (never executed by G-NeRF; for completeness): a smooth warp of a [8,64,256,256] image, as the ADA
# pipe's geometric augmentations would issue it.  Algorithmic bytes: image + grid + output (forward); + the fp32 image gradient (adjoint).
with torch.no_grad():
    n, c, hw = 8, 64, 256
    ys, xs = torch.meshgrid(torch.linspace(-1, 1, hw, device=dev), torch.linspace(-1, 1, hw, device=dev), indexing='ij')
    base = torch.stack([xs, ys], -1)[None].repeat(n, 1, 1, 1)
    grid = (base * 1.05 + 0.02 * torch.sin(7 * base.flip(-1))).contiguous()
    for dt, nm in ((torch.float16, 'f16'), (torch.float32, 'f32')):
        es = 2 if dt == torch.float16 else 4
        img = torch.randn(n, c, hw, hw, device=dev, dtype=dt)
        go = torch.randn(n, c, hw, hw, device=dev, dtype=dt)
        ms = timeit(lambda: gnerf_hip.grid_sample_2d(img, grid))
        report(f'grid_sample fwd [8,64,256,256] {nm}', ms, 2 * img.numel() * es + grid.numel() * 4)
        ms_t = timeit(lambda: torch.nn.functional.grid_sample(img, grid.to(dt), mode='bilinear', padding_mode='zeros', align_corners=False))
        report(f'(torch grid_sample fwd, same bytes, for scale) {nm}', ms_t, 2 * img.numel() * es + grid.numel() * 4)
        ms = timeit(lambda: gnerf_hip.grid_sample_2d_backward(go, img, grid))
        report(f'grid_sample adjoint (image + grid gradients) [8,64,256,256] {nm}', ms, 2 * img.numel() * es + img.numel() * 4 * 2 + 2 * grid.numel() * 4)
        ms_t = timeit(lambda: torch.ops.aten.grid_sampler_2d_backward(go, img, grid.to(dt), 0, 0, False, [True, True]))
        report(f'(aten grid_sampler_2d_backward, for scale) {nm}', ms_t, 2 * img.numel() * es + img.numel() * 4 * 2 + 2 * grid.numel() * 4)

# The matrix-bound kernels (csrc/conv3x3.hip), live since round 6: PFLOP/s against the dense f16 peak of 2.5 PFLOP/s (for the fp32-grade forms:
# the f16 work they execute -- three products per fp32 product -- and the fp32-equivalent TFLOP/s next to it).  Comparisons with MIOpen need its
# solver search and live in tools/bench_conv3x3.py, bench_conv_transpose.py, bench_conv_f32grade.py.
with torch.no_grad():
    def conv_line(name, ms, gflop_f16, extra=None):
        row = {'op': name, 'ms': round(ms, 4), 'GFLOP_f16_executed': round(gflop_f16, 2), 'PFLOPs': round(gflop_f16 / ms * 1e-3, 3), 'frac_of_2.5PF': round(gflop_f16 / ms * 1e-3 / 2.5, 3)}
        row.update(extra or {})
        print(json.dumps(row))
    g = torch.Generator(device='cpu').manual_seed(1)
    for (n, cin, cout, h, w) in [(4, 128, 128, 512, 512), (4, 256, 256, 256, 256), (8, 128, 128, 512, 512), (8, 256, 256, 256, 256), (1, 128, 128, 512, 512)]:
        x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev)
        wpk = gnerf_hip.pack_conv3x3_weights(wt)
        dco = (torch.rand(n, cout, generator=g) + 0.5).to(dev); nxt = (torch.rand(n, cout, generator=g) + 0.5).to(dev); bias = (torch.randn(cout, generator=g) * 0.1).to(dev)
        ms = timeit(lambda: gnerf_hip.conv3x3_epilogue(x, wpk, bias, scale=dco, next_scale=nxt, gain=2 ** 0.5, clamp=256.0))
        conv_line(f'fused 3x3 convolution + epilogue [{n},{cin}->{cout},{h},{w}] f16 channels_last', ms, 2e-9 * n * h * w * cin * cout * 9)
        del x
    for (n, cin, cout, h, w) in [(4, 256, 128, 256, 256), (4, 128, 128, 256, 256), (8, 256, 128, 256, 256), (1, 256, 128, 256, 256), (4, 32, 256, 128, 128)]:
        x = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).half().contiguous(memory_format=torch.channels_last)
        wp = gnerf_hip.pack_conv_transpose3x3_weights((torch.randn(cout, cin, 3, 3, generator=g) / (2 * cin ** 0.5)).to(dev))
        ms = timeit(lambda: gnerf_hip.conv_transpose3x3_s2(x, wp))
        conv_line(f'stride-2 transposed 3x3 convolution as phase convolutions [{n},{cin}->{cout},{h},{w} -> {2 * h + 1},{2 * w + 1}] f16 channels_last', ms, 2e-9 * n * h * w * cin * cout * 9)
        del x
    for (n, cin, cout, h, w) in [(4, 512, 512, 64, 64), (4, 256, 256, 128, 128), (4, 128, 128, 256, 256)]:
        x32 = (torch.randn(n, cin, h, w, generator=g) * 0.5).to(dev).contiguous(memory_format=torch.channels_last)
        w3 = gnerf_hip.pack_conv3x3_weights_f32x3((torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5)).to(dev))
        x3 = gnerf_hip.split_f16x3(x32)
        ms = timeit(lambda: gnerf_hip.conv3x3_f32x3_epilogue(x3, w3))
        gf = 2e-9 * n * h * w * cin * cout * 9
        conv_line(f'fp32-grade 3x3 convolution (f16 hi/lo x 3) [{n},{cin}->{cout},{h},{w}] f32 channels_last', ms, 3 * gf, {'TFLOPs_fp32_equivalent': round(gf / ms, 1), 'frac_of_157.3TF_fp32_matrix_peak': round(gf / ms / 157.3, 2)})
        ms_s = timeit(lambda: gnerf_hip.split_f16x3(x32))
        report(f'split_f16x3 (fp32 -> [hi | lo | hi] f16) [{n},{cin},{h},{w}]', ms_s, x32.numel() * 10)
        del x32, x3
