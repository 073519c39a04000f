import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch, gnerf_hip
dev = torch.device('cuda', 0)
gen = torch.Generator().manual_seed(11)
img = (torch.randn(1, 3, 512, 512, generator=gen) * 0.7).to(dev)
want = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
got = gnerf_hip.to_uint8_nhwc(img)
bad = (got != want)
print('n bad', int(bad.sum()), 'of', bad.numel())
idx = bad.nonzero()[:8]
for i in idx:
    n, y, x, c = [int(v) for v in i]
    v = img[n, c, y, x]
    print(float(v), float(v * 127.5), float(v * 127.5 + 128), int(got[n, y, x, c]), int(want[n, y, x, c]))
