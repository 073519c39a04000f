import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
import gnerf_generator as GG, gnerf_harness as H, gnerf_hip
dev = torch.device('cuda', 0)
torch.manual_seed(0)
G = GG.Generator().eval().requires_grad_(False).to(dev)
calls = {'repack': 0, 'absmax': 0}
r0, a0 = gnerf_hip.planes_to_nhwc, gnerf_hip.planes_absmax
def rp(*a, **k): calls['repack'] += 1; return r0(*a, **k)
def am(*a, **k): calls['absmax'] += 1; return a0(*a, **k)
gnerf_hip.planes_to_nhwc, gnerf_hip.planes_absmax = rp, am
import training.volumetric_rendering.renderer as RR
with torch.no_grad():
    ws1 = G.mapping(torch.randn(1, 512, device=dev), torch.zeros(1, 25, device=dev))
    cams = torch.cat([H.camera_label(H.orbit_pose(i, 240)) for i in range(10)]).to(dev)
    for flow in ('fast', 'reference'):
        GG._MODCONV_FAST = flow == 'fast'
        G.backbone.synthesis.b256.emit_channels_last = flow == 'fast'
        for S in (96, 48):
            G.rendering_kwargs['depth_resolution'] = G.rendering_kwargs['depth_resolution_importance'] = S
            calls.update(repack=0, absmax=0)
            G.synthesis(ws1, cams[:1], neural_rendering_resolution=64, cache_backbone=True, noise_mode='const')
            for i in range(10):
                G.synthesis(ws1, cams[i:i + 1], neural_rendering_resolution=64, use_cached_backbone=True)
            print(flow, S, dict(calls), 'planes strides', tuple(G._last_planes.stride()), 'tag', hasattr(G._last_planes, '_gnerf_absmax'))
    # why does the first call repack?
    GG._MODCONV_FAST = True
    G.backbone.synthesis.b256.emit_channels_last = True
    planes = G.backbone.synthesis(ws1, noise_mode='const')
    v = planes.view(1, 3, 32, 256, 256)
    print('stride', v.stride(), 'inter', RR._interleaved_view(v.detach()) is not None, 'base is planes', v._base is planes, 'tag', getattr(planes, '_gnerf_absmax', None), 'ver', planes._version,
          'absmax', RR._producer_absmax(v))
    import traceback
    def rp2(*a, **k):
        traceback.print_stack(limit=6); return r0(*a, **k)
    gnerf_hip.planes_to_nhwc = rp2
    G.synthesis(ws1, cams[:1], neural_rendering_resolution=64, cache_backbone=True, noise_mode='const')
