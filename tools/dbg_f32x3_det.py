#!/usr/bin/env python3
"""Which piece of the fp32-grade backbone path is not bit-reproducible from call to call?  (round 6 debugging aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import torch.nn.functional as F
import gnerf_hip
dev = torch.device('cuda', 0)
g = torch.Generator().manual_seed(0)
for (n, c, o, h, w) in [(2, 512, 512, 64, 64), (2, 128, 128, 256, 256), (2, 512, 256, 64, 64)]:
    x = (torch.randn(n, c, h, w, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(o, c, 3, 3, generator=g).to(dev)
    st = (torch.rand(n, c, generator=g) + 0.5).to(dev)
    a, b = gnerf_hip.split_f16x3(x, st), gnerf_hip.split_f16x3(x, st)
    print('split equal', torch.equal(a, b))
    w3, wp = gnerf_hip.pack_conv3x3_weights_f32x3(wt), gnerf_hip.pack_conv_transpose3x3_weights_f32x3(wt)
    bias = torch.randn(o, device=dev); sc = torch.rand(n, o, device=dev); nz = torch.randn(h, w, device=dev)
    ys = [gnerf_hip.conv3x3_f32x3_epilogue(a, w3, bias, scale=sc, noise=nz, gain=1.4) for _ in range(3)]
    print('conv f32x3 equal', [torch.equal(ys[0], y) for y in ys[1:]], 'finite', bool(torch.isfinite(ys[0]).all()))
    ts = [gnerf_hip.conv_transpose3x3_s2_f32x3(a, wp) for _ in range(3)]
    print('transposed f32x3 equal', [torch.equal(ts[0], t) for t in ts[1:]], 'finite', bool(torch.isfinite(ts[0]).all()))
    be = [gnerf_hip.blur_epilogue_channels_last(ts[0], torch.ones(4, 4, device=dev) / 16, [1, 1, 1, 1], blur_gain=4, bias=bias, scale=sc, act='lrelu', gain=1.4) for _ in range(2)]
    print('blur epilogue fp32 nhwc equal', torch.equal(be[0], be[1]))
    w1 = torch.randn(96, o, 1, 1, device=dev)
    cs = [F.conv2d(gnerf_hip.scale_channels(ys[0], sc), w1) for _ in range(3)]
    print('1x1 conv (MIOpen, fp32 channels_last) equal', [torch.equal(cs[0], t) for t in cs[1:]])
    xv = ys[0].permute(0, 2, 3, 1).reshape(n, h * w, o)
    wm = torch.randn(n, o, 96, device=dev)
    bs = [torch.baddbmm(torch.randn(1, 1, 96, device=dev) * 0 + 1, xv, wm) for _ in range(3)]
    print('1x1 as baddbmm equal', [torch.equal(bs[0], t) for t in bs[1:]])
import gnerf_generator as GG, gnerf_harness as H
torch.manual_seed(2)
G = GG.Generator().eval().requires_grad_(False).to(dev)
z = torch.randn(2, 512, device=dev)
c = torch.cat([H.camera_label(H.orbit_pose(7 + 11 * i, 120)) for i in range(2)]).to(dev)
for flag in (True, False):
    GG._F32X3 = flag
    outs = []
    with torch.no_grad():
        ws = G.mapping(z, c)
        for _ in range(3):
            outs.append(G.backbone.synthesis(ws, noise_mode='const'))
    print('backbone planes, F32X3 =', flag, 'equal', [torch.equal(outs[0], t) for t in outs[1:]], 'max diff', [float((outs[0] - t).abs().max()) for t in outs[1:]])
