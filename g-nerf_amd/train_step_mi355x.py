#!/usr/bin/env python3
"""Data-parallel training step around the renderer (BASELINE config 5).

What it keeps from the reference's loop (g_nerf/training/training_loop.py:314-437): one process per GPU, per-rank batch of
4 at neural rendering resolution 64 with 48+48 samples (train.py:252,330-335), identical initial weights on every rank
(:234-238), loss.backward() through the renderer, the manual exchange of ONE flat gradient vector -- SUM all-reduce,
/ world, nan_to_num(0, 1e5, -1e5), scatter back (:388-396) -- then Adam(betas=(0.0, 0.99), eps=1e-8) (train.py:242).

What it does not have: the StyleGAN2 backbone, super-resolution, encoder, discriminator, datasets and the SSIM / VGG
losses live in the reference tree (MIOpen / PyTorch modules, SURVEY section 2: out of scope) and are absent on the GPU
box.  Their place is taken by (a) the tri-planes as a leaf parameter [B,3,32,256,256] -- exactly the tensor the backbone
hands the renderer (triplane.py:74), so the renderer's forward and backward run at the training shape -- (b) an L1 loss
on the rendered feature image and depth against synthetic targets, and (c) `--grad-mb` of ballast parameters so that the
flat gradient vector has the size of the reference generator's (30.7 M fp32 = 123 MB) and the
collective moves the same bytes over xGMI.

    python g-nerf_amd/train_step_mi355x.py --steps 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 g-nerf_amd/train_step_mi355x.py --steps 20

Prints one JSON line on rank 0: ms per step (max over ranks), rays/s over all ranks, and the per-phase split measured
with events on rank 0 (forward, backward, gradient exchange, optimiser).
"""

import argparse
import json
import os
import sys
import time

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import gnerf_harness as H  # noqa: E402

RENDERING = dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1,
                 clamp_mode='softplus', disparity_space_sampling=False)


class RendererTrainer(torch.nn.Module):
    """Everything of the generator that is downstream of the backbone and upstream of the super-resolution: cameras ->
    rays -> ImportanceRenderer(planes, decoder) -> feature image + depth (triplane.py:54-82)."""

    def __init__(self, batch, plane_res=256, ballast_floats=0, rendering=None):
        super().__init__()
        from training.volumetric_rendering.renderer import ImportanceRenderer
        from training.volumetric_rendering.ray_sampler import RaySampler
        self.renderer, self.ray_sampler = ImportanceRenderer(), RaySampler()
        self.decoder = H.TriPlaneDecoder()
        self.planes = torch.nn.Parameter(torch.randn(batch, 3, 32, plane_res, plane_res))
        self.ballast = torch.nn.Parameter(torch.zeros(ballast_floats)) if ballast_floats else None
        self.rendering = dict(RENDERING if rendering is None else rendering)

    def forward(self, c, res):
        cam2world, intrinsics = c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3)
        o, d = self.ray_sampler(cam2world, intrinsics, res)
        feat, depth, _ = self.renderer(self.planes, self.decoder, o, d, self.rendering)
        n = c.shape[0]
        return feat.permute(0, 2, 1).reshape(n, 32, res, res), depth.permute(0, 2, 1).reshape(n, 1, res, res)


def synthetic_batch(batch, res, device, seed):
    """Per-rank data: cameras on the gen_videos orbit (a different stretch per rank) and constant-free random targets."""
    g = torch.Generator().manual_seed(seed)
    c = torch.cat([H.camera_label(H.orbit_pose(int(i), 240)) for i in torch.randint(0, 240, [batch], generator=g)])
    return c.to(device), torch.rand(batch, 32, res, res, generator=g).mul(2).sub(1).to(device), torch.rand(batch, 1, res, res, generator=g).mul(1.05).add(2.25).to(device)


def train_step(model, opt, c, target, target_depth, res, bucket_bytes=None, timers=None):
    """One optimiser step; returns the loss.  `timers`: optional list to receive (name, start_event, end_event)."""
    def mark(name):
        if timers is None or not c.is_cuda:
            return lambda: None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()

        def done():
            b.record()
            timers.append((name, a, b))
        return done

    opt.zero_grad(set_to_none=True)
    t = mark('forward')
    img, depth = model(c, res)
    loss = (img - target).abs().mean() + (depth - target_depth).abs().mean()
    if model.ballast is not None:
        loss = loss + 1e-9 * model.ballast.sum()            # gives the ballast a (constant) gradient to exchange
    t()
    t = mark('backward')
    loss.backward()
    t()
    t = mark('exchange')
    H.allreduce_flat_grads(H.params_with_grad(model), bucket_bytes=bucket_bytes)
    t()
    t = mark('optimizer')
    opt.step()
    t()
    return loss.detach()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4, help='items per GPU (config 5: 32 over 8 GPUs)')
    ap.add_argument('--res', type=int, default=64, help='neural rendering resolution (train.py:252)')
    ap.add_argument('--plane-res', type=int, default=256)
    ap.add_argument('--grad-mb', type=float, default=123.0, help='size of the flat gradient vector (reference G: 123 MB)')
    ap.add_argument('--bucket-mb', type=float, default=0.0, help='exchange the flat vector in pieces of this size (0 = one collective)')
    ap.add_argument('--device', default=None)
    args = ap.parse_args()

    rank, world, local_rank = H.init_from_env()
    use_gpu = torch.cuda.is_available() if args.device is None else args.device.startswith('cuda')
    # (modulo: a rehearsal of several ranks on a one-GPU box with GNERF_DIST_BACKEND=gloo shares the card)
    dev = torch.device('cuda', local_rank % max(1, torch.cuda.device_count())) if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(dev)
    torch.manual_seed(0)                                                    # same initial weights on every rank ...
    own = 4 * (args.batch * 3 * 32 * args.plane_res ** 2 + 4257)
    ballast = max(0, int(args.grad_mb * 1e6 - own) // 4) if args.grad_mb > 0 else 0
    model = RendererTrainer(args.batch, args.plane_res, ballast).to(dev)
    H.broadcast_module(model)                                               # ... and made sure of (training_loop.py:234-238)
    opt = torch.optim.Adam(model.parameters(), lr=0.0025, betas=(0.0, 0.99), eps=1e-8)
    c, target, target_depth = synthetic_batch(args.batch, args.res, dev, seed=100 + rank)
    torch.manual_seed(1000 + rank)                                          # per-rank sampling noise (training_loop.py:142-143)
    bucket = int(args.bucket_mb * 1e6) if args.bucket_mb > 0 else None

    def sync():
        if use_gpu:
            torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if use_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(model, opt, c, target, target_depth, args.res, bucket)
    sync()
    timers = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = train_step(model, opt, c, target, target_depth, args.res, bucket, timers)
    sync()
    elapsed = H.max_over_ranks(time.perf_counter() - t0, dev)
    H.check_ddp_consistency(model)                                          # misc.py:202-213
    if rank == 0:
        phases = {}
        for name, a, b in timers:
            phases[name] = phases.get(name, 0.0) + a.elapsed_time(b) / args.steps
        n_grad = sum(p.numel() for p in model.parameters())
        print(json.dumps({
            'workload': f'config 5 (renderer part): {args.batch} items/GPU x {args.res}^2 rays x (48+48) samples, planes {args.plane_res}^2 as leaf '
                        f'parameter, L1 loss, flat-gradient exchange of {4 * n_grad / 1e6:.1f} MB, Adam',
            'n_gpus': world, 'steps': args.steps, 'ms_per_step': 1e3 * elapsed / args.steps,
            'rays_per_s': world * args.batch * args.res ** 2 * args.steps / elapsed,
            'phase_ms_rank0': {k: round(v, 3) for k, v in phases.items()}, 'loss': float(loss),
            'exchange': 'one all-reduce' if bucket is None else f'{args.bucket_mb} MB buckets, async', 'device': str(dev)}))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
