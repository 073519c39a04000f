#!/usr/bin/env python3
"""Does HIP-graph capture of the orbit frame work while an RCCL process group (and its watchdog thread) is alive?
One rank, backend nccl, world_size 1 -- the closest a one-GPU box gets to the 8-GPU bench's situation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
import gnerf_harness as H, gen_videos_mi355x as gv
dev = torch.device('cuda', 0)
t = torch.ones(4, device=dev); dist.all_reduce(t); dist.barrier()
with torch.no_grad():
    G = gv.build_random_generator(0, dev)
    z = torch.randn(1, 512).to(dev)
    gv.render_orbit(G, z, 24, 64, dev, rank=0, world=24, double_depth=True)
    for rep in range(3):
        dist.all_reduce(t)                              # keep the watchdog busy right before the capture
        frames, _, _ = gv.render_orbit(G, z, 24, 64, dev, 0, 1, double_depth=False, use_graph=True)
        full = H.gather_frames(frames, 24)
        dist.barrier()
        print('capture + replay under RCCL ok', rep, tuple(frames.shape), float(frames.float().std()))
dist.destroy_process_group()
