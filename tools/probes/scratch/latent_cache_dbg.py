import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
import gnerf_generator as GG, gnerf_harness as H, gnerf_hip
dev = torch.device('cuda', 0)
torch.manual_seed(6)
G = GG.Generator().eval().requires_grad_(False).to(dev)
c = H.camera_label(H.orbit_pose(3, 120)).to(dev)
calls = {'n': 0}
real = gnerf_hip.modulate_weights
def counting(*a, **k):
    calls['n'] += 1
    return real(*a, **k)
gnerf_hip.modulate_weights = counting
with torch.no_grad():
    ws = G.mapping(torch.randn(1, 512, device=dev), c)
    run = lambda w: G.synthesis(w, c, noise_mode='const', neural_rendering_resolution=64)
    torch.manual_seed(1); a = run(ws); first = calls['n']
    torch.manual_seed(1); b = run(ws); second = calls['n'] - first
    print('first', first, 'second', second, 'equal', torch.equal(a['image'], b['image']), float((a['image'] - b['image']).abs().max()),
          'raw', float((a['image_raw'] - b['image_raw']).abs().max()), 'depth', float((a['image_depth'] - b['image_depth']).abs().max()))
    torch.manual_seed(1); b2 = run(ws)
    print('third equal second', torch.equal(b2['image'], b['image']))
