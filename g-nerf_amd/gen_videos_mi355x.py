#!/usr/bin/env python3
"""Counterpart of G-NeRF's gen_videos.py for MI355X (BASELINE configs 1, 3, 4).

What it keeps from the reference (g_nerf/gen_videos.py:83-224): network pickle loading through the reference's own
`legacy`/`dnnlib`, the orbit cameras and intrinsics, `ws = G.mapping(z, c=0)` once, the x2 rule on the checkpoint's
depth resolutions, `G.synthesis(ws, c, noise_mode='const', neural_rendering_resolution=res)` per frame, the uint8
conversion.  What it changes: no cv2/imageio/mrcfile (frames are returned / saved as a uint8 array), `torch.no_grad()`,
the StyleGAN2 backbone runs ONCE per orbit (`cache_backbone` / `use_cached_backbone`, triplane.py:66-71; `ws` is constant),
the device is whatever `--device` says, and frames are sharded over ranks -- one process per GPU under
`torch.distributed.run`, a contiguous block of frames each, one gather of uint8 frames at the end.

Run with this repo's g-nerf_amd/ in front of the reference's g_nerf/ on PYTHONPATH:

    PYTHONPATH=.../g-nerf_amd:.../G-NeRF/g_nerf python g-nerf_amd/gen_videos_mi355x.py --network G.pkl --encoder E.pkl --id-image face.png
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 g-nerf_amd/gen_videos_mi355x.py --random-init --frames 240 ...

`--random-init` builds the FFHQ-config TriPlaneGenerator with seeded random weights and a random z instead of loading
pickles (there are no checkpoints or network access in the build environment); where the reference tree is absent it is
this repo's inference-only equivalent (gnerf_generator.Generator).  `--graph` replays the per-frame launch sequence from a
HIP graph captured once (FrameProgram); `--shapes out.npy` also extracts the 512^3 density volume of gen_videos.py --shapes.
"""

import argparse
import sys
import time

import numpy as np
import torch

import gnerf_harness as H


def ffhq_generator_kwargs(depth_resolution=48, depth_resolution_importance=48):
    """G_kwargs of the FFHQ configuration, as train.py:238-377 assembles them from its option defaults."""
    rendering = {
        'image_resolution': 512, 'disparity_space_sampling': False, 'clamp_mode': 'softplus',
        'superresolution_module': 'training.superresolution.SuperresolutionHybrid8XDC',
        'c_gen_conditioning_zero': False, 'gpc_reg_prob': 0.5, 'c_scale': 1, 'superresolution_noise_mode': 'none',
        'density_reg': 0.25, 'density_reg_p_dist': 0.004, 'reg_type': 'l1', 'decoder_lr_mul': 1, 'sr_antialias': True,
        'depth_resolution': depth_resolution, 'depth_resolution_importance': depth_resolution_importance,
        'ray_start': 2.25, 'ray_end': 3.3, 'box_warp': 1, 'avg_camera_radius': 2.7, 'avg_camera_pivot': [0, 0, 0.2],
    }
    return dict(z_dim=512, w_dim=512, c_dim=25, img_resolution=512, img_channels=3,
                mapping_kwargs=dict(num_layers=2), channel_base=32768, channel_max=512, fused_modconv_default='inference_only',
                rendering_kwargs=rendering, num_fp16_res=0, conv_clamp=None, sr_num_fp16_res=4,
                sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only', w_dim=512))


def build_random_generator(seed, device, **kw):
    """Seeded random-init FFHQ-config generator: the reference's TriPlaneGenerator when its tree is importable (its renderer
    and ops then resolve to this repo through the overlay), else this repo's inference-only equivalent with the same layer
    graph and parameter names (gnerf_generator.Generator; the GPU box has no reference tree)."""
    torch.manual_seed(seed)
    try:
        from training.triplane import TriPlaneGenerator
        G = TriPlaneGenerator(**ffhq_generator_kwargs(**kw))
    except ImportError:
        import gnerf_generator
        G = gnerf_generator.Generator(rendering_kwargs=ffhq_generator_kwargs(**kw)['rendering_kwargs'])
    return G.eval().requires_grad_(False).to(device)


def load_generator(network_pkl, device):
    import dnnlib
    import legacy
    with dnnlib.util.open_url(network_pkl) as f:
        return legacy.load_network_pkl(f)['G_ema'].to(device).eval().requires_grad_(False)


def identity_latent(args, device, z_dim):
    if args.encoder and args.id_image:
        import dnnlib
        import legacy
        from PIL import Image
        with dnnlib.util.open_url(args.encoder) as f:
            E = legacy.load_network_pkl(f)['E'].to(device).eval()
        img = np.asarray(Image.open(args.id_image).convert('RGB')).transpose(2, 0, 1)[None]
        return E(torch.from_numpy(img.copy()).to(device) / 127.5 - 1)                       # gen_videos.py:119,131
    return torch.randn(1, z_dim, generator=torch.Generator().manual_seed(args.seed + 1)).to(device)


def orbit_latents(G, z, device):
    """ws of the whole orbit: the mapping network runs once, on a zero camera label (gen_videos.py:147-150)."""
    c0 = H.camera_label(H.lookat_pose(3.14 / 2, 3.14 / 2, G.rendering_kwargs['avg_camera_radius'], device))
    return G.mapping(z=z, c=torch.zeros_like(c0).repeat(z.shape[0], 1))


class FrameProgram:
    """One orbit frame -- rays, the two uniform draws, the fused render, superresolution, uint8 conversion, ~190 launches --
    captured ONCE into a HIP graph and replayed per camera.  No entry point of the native library allocates or synchronises,
    and torch's graph-safe generator advances the draws from replay to replay.  The backbone runs once, before the capture
    (ws is constant over the orbit).  Reusable across orbits of the same generator, latent and resolution.

    The graph bakes in the ADDRESSES of everything the captured frame read: the cached backbone planes, the renderer's NHWC
    copy of them and its folded decoder weights.  The program therefore owns strong references to all three and puts them
    back in place before every replay -- an eager frame with cache_backbone=True in between would otherwise replace (and
    free) them and the replay would read recycled memory."""

    @torch.no_grad()
    def __init__(self, G, ws, res, device, batch=1):
        self.G, self.ws, self.res = G, ws, res
        self.c = H.camera_label(H.orbit_pose(0, 120, G.rendering_kwargs['avg_camera_radius'], device=device)).repeat(batch, 1)
        self._frame(cache=True, cached=False)
        self.planes = G._last_planes
        self._pin = None
        if hasattr(G.renderer, 'pin_planes'):
            p = self.planes
            self._pin = G.renderer.pin_planes(p.view(len(p), 3, 32, p.shape[-2], p.shape[-1]))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._frame()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.frame, self.raw = self._frame()
        self._decoder = G.renderer.__dict__.get('_gnerf_decoder_cache')
        # ... and the per-latent constants of the layers (styles, modulated weights: gnerf_generator._per_latent), which the captured
        # kernels read in place; another latent through the same generator replaces the cache entries, not these tensors
        self._latent_constants = getattr(sys.modules.get(type(G).__module__), 'latent_cache_tensors', lambda g: [])(G)

    def _frame(self, cache=False, cached=True):
        out = self.G.synthesis(ws=self.ws, c=self.c, noise_mode='const', neural_rendering_resolution=self.res,
                               cache_backbone=cache, use_cached_backbone=cached)
        return H.to_uint8(out['image']), H.to_uint8(out['image_raw'])

    def __call__(self, c):
        self.c.copy_(c)
        self.G._last_planes = self.planes                         # what an eager use_cached_backbone frame after this one reads, too
        if self._pin is not None:
            self.G.renderer.repin(self._pin)
        self.graph.replay()
        return self.frame.clone(), self.raw.clone()


@torch.no_grad()
def render_orbit(G, z, n_frames, res, device, rank=0, world=1, double_depth=True, frame_seed=None, use_graph=False, program=None,
                 frames_per_call=1):
    """This rank's frames of the orbit: (uint8 [n_local,512,512,3], uint8 raw [n_local,res,res,3], (lo, hi)).
    use_graph (GPU only): frames come from a FrameProgram (`program`, or one captured here).
    frames_per_call > 1 (one latent only): that many cameras go through the renderer and the superresolution as one batch; the
    renderer treats them as views of the one set of planes, each with the draws and the depth clamp of a call of its own."""
    if double_depth:                                                                        # gen_videos.py:127-128
        G.rendering_kwargs['depth_resolution'] = int(G.rendering_kwargs['depth_resolution'] * 2)
        G.rendering_kwargs['depth_resolution_importance'] = int(G.rendering_kwargs['depth_resolution_importance'] * 2)
    radius = G.rendering_kwargs['avg_camera_radius']
    lo, hi = H.shard_range(n_frames, rank, world)
    if hi == lo:
        return None, None, (lo, hi)
    k = max(int(frames_per_call), 1)
    if k > 1 and (z.shape[0] != 1 or frame_seed is not None):
        raise ValueError('frames_per_call > 1 needs a single latent and no per-frame reseeding')
    # every camera of this rank's block made on the host in one set of batched ops (per frame: ~0.15 ms of small tensor ops each) and
    # moved in ONE copy (per frame: a pose and an intrinsics upload, two blocking host-to-device copies in front of every frame's launches)
    labels = H.orbit_labels(range(lo, hi), n_frames, radius).to(device)
    if k > 1:
        cams = [labels[j:j + k] for j in range(0, hi - lo, k)]
    else:
        cams = [labels[j:j + 1].repeat(z.shape[0], 1) if z.shape[0] > 1 else labels[j:j + 1] for j in range(hi - lo)]
    frames, raws = [], []
    if (use_graph or program is not None) and device.type == 'cuda':
        assert frame_seed is None, 'per-frame reseeding and graph replay do not mix'
        if program is None:
            program = FrameProgram(G, orbit_latents(G, z, device), res, device, batch=z.shape[0] if k == 1 else k)
        for c in cams:
            n = c.shape[0]
            if n < program.c.shape[0]:                                                      # the last, shorter block of an orbit
                c = torch.cat([c, c[-1:].expand(program.c.shape[0] - n, -1)])
            f, r = program(c)
            frames.append(f[:n])
            raws.append(r[:n])
    else:
        ws = orbit_latents(G, z, device)
        for j, c in enumerate(cams):
            if frame_seed is not None:
                torch.manual_seed(frame_seed + lo + j)                                      # reproducible renderer draws per frame
            n = c.shape[0]
            if k > 1 and n < k:
                # the last, shorter block of an orbit keeps the batch size of the others (repeat its last camera, drop the extra frames):
                # a new batch size is a new set of convolution shapes -- a MIOpen solver search in the middle of the orbit
                c = torch.cat([c, c[-1:].expand(k - n, -1)])
            out = G.synthesis(ws=ws, c=c, noise_mode='const', neural_rendering_resolution=res, cache_backbone=(j == 0), use_cached_backbone=(j > 0))
            frames.append(H.to_uint8(out['image'][:n]))
            raws.append(H.to_uint8(out['image_raw'][:n]))
    return torch.cat(frames), torch.cat(raws), (lo, hi)


def voxel_samples(lo, hi, n, cube_length, device):
    """Points lo..hi-1 of gen_videos.py's create_samples(N=n, voxel_origin=[0,0,0], cube_length) (gen_videos.py:33-56), made on
    the device chunk by chunk instead of as one [n^3, 3] host tensor (1.6 GB at n = 512).  The arithmetic is the reference's,
    quirks included: the y and x indices come from FLOAT divisions of the linear index without a floor (a sheared lattice), in
    fp32 (indices above 2^24 are rounded)."""
    idx = torch.arange(lo, hi, dtype=torch.int64, device=device)
    origin, size = -cube_length / 2, cube_length / (n - 1)
    f = idx.float()
    s2 = (idx % n).float()
    s1 = (f / n) % n
    s0 = ((f / n) / n) % n
    return torch.stack([s0 * size + origin, s1 * size + origin, s2 * size + origin], dim=-1).unsqueeze(0)


@torch.no_grad()
def extract_density_grid(G, ws, resolution=512, max_batch=10000000, crop=True):
    """The density volume gen_videos.py --shapes writes to .mrc (gen_videos.py:189-224): sigma at resolution^3 lattice points of
    the box, flipped along the first axis, borders zeroed.  The reference calls G.sample_mixed per chunk of 10^7 points, which
    re-runs the StyleGAN2 backbone for every chunk (14 passes at 512^3); here the planes are made once and every chunk goes
    through ImportanceRenderer.run_model (the fused point-query kernel on a GPU).  Returns a float32 tensor [r, r, r] on ws' device."""
    device = ws.device
    planes = G.backbone.synthesis(ws, noise_mode='const')
    planes = planes.view(len(planes), 3, 32, planes.shape[-2], planes.shape[-1])
    total = resolution ** 3
    sigmas = torch.empty(total, dtype=torch.float32, device=device)
    for head in range(0, total, max_batch):
        hi = min(total, head + max_batch)
        pts = voxel_samples(head, hi, resolution, G.rendering_kwargs['box_warp'], device)
        if hasattr(G.renderer, 'query_sigma'):                                              # this repo's renderer: densities only
            sigmas[head:hi] = G.renderer.query_sigma(planes, G.decoder, pts, G.rendering_kwargs).reshape(-1)
        else:
            dirs = torch.zeros_like(pts)
            dirs[..., -1] = -1
            sigmas[head:hi] = G.renderer.run_model(planes, G.decoder, pts, dirs, G.rendering_kwargs)['sigma'].reshape(-1)
    vol = sigmas.reshape(resolution, resolution, resolution).flip(0)
    if crop:                                                                                # gen_videos.py:213-220
        pad, pad_top = int(30 * resolution / 256), int(38 * resolution / 256)
        vol = vol.clone()
        vol[:pad] = 0
        vol[-pad:] = 0
        vol[:, :pad] = 0
        vol[:, -pad_top:] = 0
        vol[:, :, :pad] = 0
        vol[:, :, -pad:] = 0
    return vol


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--network', help='generator pickle (G_ema)')
    ap.add_argument('--encoder', help='identity encoder pickle (E)')
    ap.add_argument('--id-image', help='identity reference image')
    ap.add_argument('--random-init', action='store_true', help='seeded random generator + random z instead of pickles')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--frames', type=int, default=120)                  # gen_videos.py:151
    ap.add_argument('--res', type=int, default=64)                      # neural rendering resolution, gen_videos.py:81
    ap.add_argument('--no-double-depth', action='store_true', help='keep the checkpoint depth resolutions (the reference CLI doubles them)')
    ap.add_argument('--device', default='cuda' if torch.cuda.is_available() else 'cpu')
    ap.add_argument('--graph', action='store_true', help='replay the per-frame launch sequence from a captured HIP graph')
    ap.add_argument('--frames-per-call', type=int, default=1, help='cameras per synthesis call (views of the one latent in one renderer launch)')
    ap.add_argument('--out', default=None, help='write frames to this .npy (rank 0)')
    ap.add_argument('--shapes', default=None, help='also extract the 512^3 density volume (gen_videos.py --shapes) and save it to this .npy (rank 0)')
    ap.add_argument('--voxel-res', type=int, default=512)
    ap.add_argument('--no-solver-search', action='store_true', help='leave torch.backends.cudnn.benchmark off (MIOpen takes its first heuristic pick per shape)')
    args = ap.parse_args()
    H.configure_backend(False if args.no_solver_search else None)

    rank, world, local_rank = H.init_from_env()
    # (modulo: a rehearsal of several ranks on a one-GPU box with GNERF_DIST_BACKEND=gloo shares the card)
    device = torch.device('cuda', local_rank % max(1, torch.cuda.device_count())) if args.device == 'cuda' else torch.device(args.device)
    if device.type == 'cuda':
        torch.cuda.set_device(device)
    if args.random_init:
        G = build_random_generator(args.seed, device)
    else:
        assert args.network, 'give --network or --random-init'
        G = load_generator(args.network, device)
    z = identity_latent(args, device, G.z_dim)

    # One untimed frame first: library load, MIOpen's per-shape kernel search (tens of seconds on a fresh machine) and the
    # allocator's first touches are one-off costs of the process, not of the orbit.
    render_orbit(G, z, args.frames, args.res, device, rank=0, world=args.frames, double_depth=False)
    if device.type == 'cuda':
        torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    frames, raws, (lo, hi) = render_orbit(G, z, args.frames, args.res, device, rank, world, double_depth=not args.no_double_depth, use_graph=args.graph,
                                          frames_per_call=args.frames_per_call)
    full = H.gather_frames(frames, args.frames)
    if device.type == 'cuda':
        torch.cuda.synchronize()
    elapsed = H.max_over_ranks(time.perf_counter() - t0, device)
    if rank == 0:
        print(f'{args.frames} frames on {world} rank(s){" (HIP graph replay)" if args.graph else ""}: {elapsed:.3f} s = {args.frames / elapsed:.2f} frames/s '
              f'(neural rendering {args.res}x{args.res}, {G.rendering_kwargs["depth_resolution"]}+{G.rendering_kwargs["depth_resolution_importance"]} samples)')
        if args.out:
            np.save(args.out, full.cpu().numpy())
        if args.shapes:
            if device.type == 'cuda':
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            vol = extract_density_grid(G, orbit_latents(G, z, device), args.voxel_res)
            if device.type == 'cuda':
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(f'density volume {args.voxel_res}^3: {dt:.3f} s = {args.voxel_res ** 3 / dt / 1e9:.2f} G points/s')
            np.save(args.shapes, vol.cpu().numpy())


if __name__ == '__main__':
    main()
