#!/usr/bin/env python3
"""Data-parallel training step (BASELINE config 5; SURVEY.md sections 3.4, 8d, 8e): G + D forward/backward on an FFHQ-shape
synthetic batch, one process per GPU, the reference's two flat-gradient all-reduces over RCCL.

The step is the reference's (g_nerf/training/training_loop.py:314-437) with the pieces that cannot exist here replaced as
SURVEY section 8d prescribes:
  * batch of 4 per GPU at neural rendering resolution 64, 48+48 samples (train.py:252,330-335); the identity encoder is out
    of scope (torchvision's ResNeXt50), so the latent z is part of the synthetic batch;
  * ws = G.mapping(z, c); G.synthesis(ws, c, neural_rendering_resolution=64) in TRAINING mode: un-fused modulated
    convolutions ('inference_only', train.py:304), random noise in the backbone, fp16 super-resolution on the GPU;
  * loss = (L1(image) + L1(image_raw)) weighted by `factor` + 1.2 softplus(-D(image_depth, c)) (:341-375; the SSIM and
    VGG terms need pytorch_msssim and a downloaded network: dropped, as SURVEY says);
  * loss.backward(); ONE flat vector of all G gradients: SUM all-reduce, / world, nan_to_num(0, 1e5, -1e5), scatter back
    (:388-396); Adam(betas 0.9/0.999, :311);
  * D step: softplus(D(fake.detach())), softplus(-D(real_depth)) + R1 on the real depth image through a double backward
    (:402-423), the second flat all-reduce (:427-436), Adam(betas 0/0.99, train.py:242);
  * D = Discriminator(c_dim=25, img_resolution=64, img_channels=1) with mbstd_group_size=4 (train.py's default of 3 does
    not divide the per-GPU batch of 4: SURVEY section 1).
The renderer's forward and backward inside G.synthesis are this repo's fused kernels; bias_act / upfirdn2d and their first-
and second-order gradients (R1) are the native ops; the convolutions are MIOpen's.

`--mode renderer` keeps round 1's stand-in (tri-planes as a leaf parameter, L1 on the feature image, ballast up to the
reference generator's 123 MB of gradients): the renderer's share of the step in isolation.

    python g-nerf_amd/train_step_mi355x.py --steps 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 g-nerf_amd/train_step_mi355x.py --steps 10

Prints one JSON line on rank 0: ms per step (max over ranks), images/s over all ranks, the per-phase split on rank 0.
"""

import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import gnerf_harness as H  # noqa: E402

RENDERING = dict(depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1,
                 clamp_mode='softplus', disparity_space_sampling=False)
R1_GAMMA = 1.0


# ---------------------------------------------------------------------------------------------------------------------
# the G + D step


def generator_loss(G, D, batch, res=64, **synthesis_kwargs):
    """training_loop.py:328-375 without the SSIM / VGG terms.  Returns (loss, parts, generated images)."""
    ws = G.mapping(batch['z'], batch['c'])
    gen = G.synthesis(ws, batch['c'], neural_rendering_resolution=res, **synthesis_kwargs)
    real = batch['loss_image']
    real_raw = F.interpolate(real, size=(res, res), mode='bilinear', align_corners=False, antialias=True)      # ssim_resize, :180,:337
    l1 = (real - gen['image'].float()).abs().mean((1, 2, 3))
    l1_raw = (real_raw - gen['image_raw'].float()).abs().mean((1, 2, 3))
    factor = batch['factor']
    recon = ((l1 + l1_raw) * factor).sum() / (factor.sum() + 1e-6)
    loss_gan = F.softplus(-D(gen['image_depth'], batch['c'])).mean()
    return recon + 1.2 * loss_gan, dict(l1=l1.detach().mean(), l1_raw=l1_raw.detach().mean(), gan=loss_gan.detach()), gen


def discriminator_backward(D, fake_depth, batch, r1_gamma=R1_GAMMA, **d_kwargs):
    """training_loop.py:402-423: both backward passes of the D step (gradients accumulate in D's parameters)."""
    loss_gen = F.softplus(D(fake_depth.detach(), batch['c'], **d_kwargs))
    loss_gen.mean().backward()
    real = batch['depth_image'].detach().requires_grad_(True)
    logits = D(real, batch['condition_c'], **d_kwargs)
    loss_real = F.softplus(-logits)
    r1_grad, = torch.autograd.grad(outputs=[logits.sum()], inputs=[real], create_graph=True, only_inputs=True)
    loss_r1 = r1_grad.square().sum([1, 2, 3]) * (r1_gamma / 2)
    (loss_real + loss_r1).mean().backward()
    return dict(d_gen=loss_gen.detach().mean(), d_real=loss_real.detach().mean(), d_r1=loss_r1.detach().mean())


def gd_train_step(G, D, opt_G, opt_D, batch, res=64, bucket_bytes=None, timers=None, synthesis_kwargs=None, d_kwargs=None):
    """One optimiser step of G and one of D; returns the dict of loss terms."""
    mark = _marker(timers, batch['c'])
    synthesis_kwargs, d_kwargs = synthesis_kwargs or {}, d_kwargs or {}
    opt_G.zero_grad(set_to_none=True)
    G.requires_grad_(True)
    t = mark('G forward')
    loss, parts, gen = generator_loss(G, lambda img, c: D(img, c, **d_kwargs), batch, res, **synthesis_kwargs)
    t()
    t = mark('G backward')
    loss.backward()
    t()
    t = mark('G exchange')
    H.allreduce_flat_grads(H.params_with_grad(G), bucket_bytes=bucket_bytes)
    t()
    t = mark('G optimizer')
    opt_G.step()
    G.requires_grad_(False)
    t()
    opt_D.zero_grad(set_to_none=True)
    D.requires_grad_(True)
    t = mark('D forward+backward (R1)')
    parts.update(discriminator_backward(D, gen['image_depth'], batch, **d_kwargs))
    D.requires_grad_(False)
    t()
    t = mark('D exchange')
    H.allreduce_flat_grads(list(D.parameters()), bucket_bytes=bucket_bytes)
    t()
    t = mark('D optimizer')
    opt_D.step()
    t()
    parts['loss'] = loss.detach()
    return parts


def synthetic_gd_batch(n, device, seed):
    """Per-rank batch with the shapes of dataset.py:1036-1045: uint8-range loss image [3,512,512] mapped to
    [-1,1], camera labels [25] on the gen_videos orbit, a real depth image [1,64,64] inside the ray limits, factor 1."""
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, 240, [2 * n], generator=g)
    c = torch.cat([H.camera_label(H.orbit_pose(int(i), 240)) for i in idx[:n]])
    cond_c = torch.cat([H.camera_label(H.orbit_pose(int(i), 240)) for i in idx[n:]])
    img = torch.randint(0, 256, [n, 3, 512, 512], generator=g, dtype=torch.uint8)
    out = dict(z=torch.randn(n, 512, generator=g), c=c, condition_c=cond_c, loss_image=img.float() / 127.5 - 1,
               depth_image=torch.rand(n, 1, 64, 64, generator=g) * 1.05 + 2.25, factor=torch.ones(n))
    return {k: v.to(device) for k, v in out.items()}


# ---------------------------------------------------------------------------------------------------------------------
# round 1's renderer-only stand-in


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


def _marker(timers, probe):
    def mark(name):
        if timers is None or not probe.is_cuda:
            return lambda: None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()

        def done():
            b.record()
            timers.append((name, a, b))
        return done
    return mark


def train_step(model, opt, c, target, target_depth, res, bucket_bytes=None, timers=None):
    """One optimiser step of the renderer-only stand-in; returns the loss."""
    mark = _marker(timers, c)
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
    ap.add_argument('--mode', choices=['full', 'renderer'], default='full', help='full = G + D step (config 5); renderer = the renderer-only stand-in')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=4, help='items per GPU (config 5: 32 over 8 GPUs)')
    ap.add_argument('--res', type=int, default=64, help='neural rendering resolution (train.py:252)')
    ap.add_argument('--plane-res', type=int, default=256)
    ap.add_argument('--grad-mb', type=float, default=123.0, help='renderer mode: size of the flat gradient vector (reference G: 123 MB)')
    ap.add_argument('--bucket-mb', type=float, default=0.0, help='exchange the flat vector in pieces of this size (0 = one collective)')
    ap.add_argument('--force-fp32', action='store_true', help='full mode: no fp16 in the super-resolution and discriminator blocks')
    ap.add_argument('--device', default=None)
    ap.add_argument('--marked', action='store_true', help='bracket the timed steps with the marker kernel of tools/orbit_marked.py (MARKED_SCRIPT=... tools/prof_orbit.sh: '
                                                          'per-kernel statistics of the steps alone, warm-up excluded)')
    ap.add_argument('--solver-search', action='store_true',
                    help='torch.backends.cudnn.benchmark on, as training_loop.py:133,144 has it.  Off by default here: MIOpen then times every solver of every '
                         'forward / backward-data / backward-weights shape of G and D on first use, and the first step takes more than seven minutes')
    args = ap.parse_args()
    H.configure_backend(bool(args.solver_search))

    rank, world, local_rank = H.init_from_env()
    use_gpu = torch.cuda.is_available() if args.device is None else args.device.startswith('cuda')
    # (modulo: a rehearsal of several ranks on a one-GPU box with GNERF_DIST_BACKEND=gloo shares the card)
    dev = torch.device('cuda', local_rank % max(1, torch.cuda.device_count())) if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(dev)
    torch.manual_seed(0)                                                    # same initial weights on every rank ...
    bucket = int(args.bucket_mb * 1e6) if args.bucket_mb > 0 else None

    def sync():
        if use_gpu:
            torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        if use_gpu:
            torch.cuda.synchronize()

    if args.mode == 'full':
        from torch_utils import custom_ops
        custom_ops.verbosity = 'none'
        import gnerf_generator
        G = gnerf_generator.Generator().train().requires_grad_(False).to(dev)
        D = gnerf_generator.Discriminator(c_dim=25, img_resolution=args.res, img_channels=1, mbstd_group_size=min(4, args.batch)).train().requires_grad_(False).to(dev)
        H.broadcast_module(G)                                               # ... and made sure of (training_loop.py:234-238)
        H.broadcast_module(D)
        # (on the GPU: torch's fused Adam -- one kernel per optimiser instead of the multi-tensor form's passes; same update formula)
        fused = dict(fused=True) if use_gpu and os.environ.get('GNERF_FUSED_ADAM', '1') == '1' else {}
        opt_G = torch.optim.Adam(G.parameters(), lr=0.0025, betas=(0.9, 0.999), eps=1e-8, **fused)          # training_loop.py:311
        opt_D = torch.optim.Adam(D.parameters(), lr=0.002, betas=(0.0, 0.99), eps=1e-8, **fused)            # train.py:242
        batch = synthetic_gd_batch(args.batch, dev, seed=100 + rank)
        torch.manual_seed(1000 + rank)                                      # per-rank noise (training_loop.py:142-143)
        kw = dict(synthesis_kwargs=dict(force_fp32=True), d_kwargs=dict(force_fp32=True)) if args.force_fp32 else {}

        def one(timers=None):
            return gd_train_step(G, D, opt_G, opt_D, batch, args.res, bucket, timers, **kw)
        modules = (G, D)
        n_grad = (sum(p.numel() for p in G.parameters()), sum(p.numel() for p in D.parameters()))
        what = (f'config 5: G + D step, {args.batch} items/GPU, 64^2 rays x (48+48) samples, SR to 512^2 '
                f'({"fp32" if args.force_fp32 or not use_gpu else "fp16 SR / D blocks"}), loss L1 + L1 + 1.2 softplus(-D(depth)), R1 on D, two flat all-reduces of '
                f'{4 * n_grad[0] / 1e6:.1f} MB (G) and {4 * n_grad[1] / 1e6:.1f} MB (D), Adam x2; random-init FFHQ-config generator, synthetic batch')
    else:
        own = 4 * (args.batch * 3 * 32 * args.plane_res ** 2 + 4257)
        ballast = max(0, int(args.grad_mb * 1e6 - own) // 4) if args.grad_mb > 0 else 0
        model = RendererTrainer(args.batch, args.plane_res, ballast).to(dev)
        H.broadcast_module(model)
        fused = dict(fused=True) if use_gpu and os.environ.get('GNERF_FUSED_ADAM', '1') == '1' else {}
        opt = torch.optim.Adam(model.parameters(), lr=0.0025, betas=(0.0, 0.99), eps=1e-8, **fused)
        c, target, target_depth = synthetic_batch(args.batch, args.res, dev, seed=100 + rank)
        torch.manual_seed(1000 + rank)

        def one(timers=None):
            return {'loss': train_step(model, opt, c, target, target_depth, args.res, bucket, timers)}
        modules = (model,)
        n_grad = (sum(p.numel() for p in model.parameters()),)
        what = (f'config 5 (renderer part): {args.batch} items/GPU x {args.res}^2 rays x (48+48) samples, planes {args.plane_res}^2 as leaf '
                f'parameter, L1 loss, flat-gradient exchange of {4 * n_grad[0] / 1e6:.1f} MB, Adam')

    for _ in range(args.warmup):
        one()
    sync()
    if args.marked and use_gpu:
        import gnerf_hip
        gnerf_hip.torch_rand(424242, dev, 1, 0)
        torch.cuda.synchronize()
    timers = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        parts = one(timers)
    sync()
    elapsed = H.max_over_ranks(time.perf_counter() - t0, dev)
    if args.marked and use_gpu:
        gnerf_hip.torch_rand(424242, dev, 1, 0)
        torch.cuda.synchronize()
    for m in modules:
        H.check_ddp_consistency(m)                                          # misc.py:202-213
    if rank == 0:
        phases = {}
        for name, a, b in timers:
            phases[name] = phases.get(name, 0.0) + a.elapsed_time(b) / args.steps
        print(json.dumps({
            'workload': what, 'n_gpus': world, 'steps': args.steps, 'frames': args.steps * args.batch, 'ms_per_step': 1e3 * elapsed / args.steps,
            'images_per_s': world * args.batch * args.steps / elapsed,
            'rays_per_s': world * args.batch * args.res ** 2 * args.steps / elapsed,
            'phase_ms_rank0': {k: round(v, 3) for k, v in phases.items()},
            'losses': {k: float(v) for k, v in parts.items()},
            'exchange': 'one all-reduce per optimiser' if bucket is None else f'{args.bucket_mb} MB buckets, async', 'device': str(dev)}))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
