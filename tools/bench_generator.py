#!/usr/bin/env python3
"""Frames/s of the whole generator forward on one MI355X -- BASELINE configs 3 and 4 (one GPU's share):
  config 3: G.synthesis for a batch of 4: StyleGAN2 backbone -> tri-planes -> fused renderer (64x64 rays, 48+48) -> superresolution
            to 512x512 (fp16, native bias_act / upfirdn2d); phase split from HIP events.
  config 4: gen_videos.py's orbit, one frame per step, backbone cached (ws is constant over the orbit), 30 frames = one GPU's
            share of 240 frames on 8 GPUs, with the reference CLI's doubled sampling (96+96) and with 48+48; uint8 conversion
            of every frame included, eager and replayed from a captured HIP graph.
Random-init FFHQ-config generator (gnerf_generator.Generator: the reference's layer graph and parameter names, validated
against the reference class in tests/test_generator_cpu.py); there are no checkpoints in the build environment.
With --ref-ops the custom ops run in their PyTorch-op forms (what a G-NeRF checkout without its CUDA plugins does on this
GPU) and the renderer in its PyTorch-op form, for scale.  One JSON line per measurement."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT]
import torch
from torch_utils import custom_ops
custom_ops.verbosity = 'none'
import gnerf_harness as H, gnerf_generator

ap = argparse.ArgumentParser()
ap.add_argument('--ref-ops', action='store_true')
ap.add_argument('--reference-flow', action='store_true', help="the reference's layer code (GNERF_MODCONV_FAST=0) on the overlay's ops, NCHW planes")
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--frames', type=int, default=30)
ap.add_argument('--only', type=int, default=0, help='3 or 4: run only that configuration (for profiling)')
ap.add_argument('--miopen-find', type=int, default=1, help='torch.backends.cudnn.benchmark (MIOpen searches its solvers per shape on first use; training_loop.py:144 turns it on)')
ap.add_argument('--frames-per-call', type=int, nargs='*', default=[], help='config 4 also with this many cameras per synthesis call (views of the one latent)')
args = ap.parse_args()
dev = torch.device('cuda', 0)
H.configure_backend(bool(args.miopen_find))
torch.manual_seed(0)
G = gnerf_generator.Generator().eval().requires_grad_(False).to(dev)
with torch.no_grad():
    for n, p in G.named_parameters():           # non-trivial noise / biases, as in a trained network
        if n.endswith('noise_strength') or n.endswith('.bias'):
            p.add_(torch.randn_like(p) * 0.1)

if args.reference_flow:
    gnerf_generator._MODCONV_FAST = False
    G.backbone.synthesis.b256.emit_channels_last = False
if args.ref_ops:
    from torch_utils.ops import bias_act, upfirdn2d
    _b, _u = bias_act.bias_act, upfirdn2d.upfirdn2d
    bias_act.bias_act = lambda *a, **k: _b(*a, **{**k, 'impl': 'ref'})
    upfirdn2d.upfirdn2d = lambda *a, **k: _u(*a, **{**k, 'impl': 'ref'})
    G.renderer.forward = G.renderer._forward_torch


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / reps)
    return best, out


with torch.no_grad():
    # ---------------- config 3
    N = args.batch if args.only != 4 else 1
    z = torch.randn(N, 512, device=dev)
    c = torch.cat([H.camera_label(H.orbit_pose(7 * i, 240)) for i in range(N)]).to(dev)
    ws = G.mapping(z, c)
    t, out = timed(lambda: G.synthesis(ws, c, neural_rendering_resolution=64, noise_mode='const'), 10)
    assert out['image'].shape == (N, 3, 512, 512) and torch.isfinite(out['image']).all()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    o, d = G.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), 64)
    split = [0.0, 0.0, 0.0]
    for _ in range(5):
        ev[0].record()
        planes = G.backbone.synthesis(ws, noise_mode='const')
        ev[1].record()
        feat, depth, _ = G.renderer(planes.view(N, 3, 32, 256, 256), G.decoder, o, d, G.rendering_kwargs)
        fi = feat.permute(0, 2, 1).reshape(N, 32, 64, 64).contiguous()
        ev[2].record()
        G.superresolution(fi[:, :3], fi, ws, noise_mode='none')
        ev[3].record()
        torch.cuda.synchronize()
        for k in range(3):
            split[k] += ev[k].elapsed_time(ev[k + 1]) / 5
    if args.only != 4:
      print(json.dumps({'config': 3, 'workload': f'full generator forward, batch {N}, render 64x64 x (48+48), SR to 512x512 fp16',
                        'ops': 'pytorch-op forms' if args.ref_ops else 'native gfx950', 'ms_per_batch': round(t * 1e3, 3), 'frames_per_s': round(N / t, 1),
                        'phase_ms': {'backbone': round(split[0], 3), 'renderer': round(split[1], 3), 'superresolution': round(split[2], 3)}}), flush=True)

    # ---------------- config 4 (one GPU's share of the orbit)
    z1 = z[:1]
    ws1 = G.mapping(z1, torch.zeros(1, 25, device=dev))
    for S in ((96, 48) if args.only != 3 else ()):
        G.rendering_kwargs['depth_resolution'] = G.rendering_kwargs['depth_resolution_importance'] = S
        cams = torch.cat([H.camera_label(H.orbit_pose(i, 240)) for i in range(args.frames)]).to(dev)
        G.synthesis(ws1, cams[:1], neural_rendering_resolution=64, cache_backbone=True, noise_mode='const')

        def orbit():
            frames = []
            for i in range(args.frames):
                o_ = G.synthesis(ws1, cams[i:i + 1], neural_rendering_resolution=64, use_cached_backbone=True)
                frames.append(H.to_uint8(o_['image']))
            return torch.cat(frames)
        t, frames = timed(orbit, 1)
        line = {'config': 4, 'workload': f'orbit share of one GPU: {args.frames} frames, 64x64 rays x ({S}+{S}), cached backbone, SR to 512x512 fp16, uint8 frames',
                'ops': 'pytorch-op forms' if args.ref_ops else ('native gfx950, reference layer code' if args.reference_flow else 'native gfx950'), 'frames_per_s': round(args.frames / t, 1), 'ms_per_frame': round(t / args.frames * 1e3, 3)}
        if not args.ref_ops:
            # the per-frame sequence replayed from a HIP graph (no entry point allocates or synchronises)
            cam = cams[:1].clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    H.to_uint8(G.synthesis(ws1, cam, neural_rendering_resolution=64, use_cached_backbone=True)['image'])
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                frame = H.to_uint8(G.synthesis(ws1, cam, neural_rendering_resolution=64, use_cached_backbone=True)['image'])

            def orbit_graph():
                fr = []
                for i in range(args.frames):
                    cam.copy_(cams[i:i + 1])
                    graph.replay()
                    fr.append(frame.clone())
                return torch.cat(fr)
            tg, _ = timed(orbit_graph, 1)
            line.update(graph_frames_per_s=round(args.frames / tg, 1), graph_ms_per_frame=round(tg / args.frames * 1e3, 3))
        print(json.dumps(line), flush=True)
        for k in ([] if args.ref_ops else args.frames_per_call):
            def orbit_k():
                return torch.cat([H.to_uint8(G.synthesis(ws1, cams[i:i + k], neural_rendering_resolution=64, use_cached_backbone=True)['image'])
                                  for i in range(0, args.frames, k)])
            t, fr = timed(orbit_k, 1)
            assert fr.shape[0] == args.frames
            line = {'config': 4, 'workload': f'orbit share of one GPU: {args.frames} frames, {k} cameras per call, 64x64 rays x ({S}+{S}), cached backbone, SR to 512x512 fp16, uint8 frames',
                    'frames_per_call': k, 'frames_per_s': round(args.frames / t, 1), 'ms_per_frame': round(t / args.frames * 1e3, 3)}
            cam = cams[:k].clone()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):
                    H.to_uint8(G.synthesis(ws1, cam, neural_rendering_resolution=64, use_cached_backbone=True)['image'])
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                frame = H.to_uint8(G.synthesis(ws1, cam, neural_rendering_resolution=64, use_cached_backbone=True)['image'])

            def orbit_graph_k():
                fr = []
                for i in range(0, args.frames - args.frames % k, k):
                    cam.copy_(cams[i:i + k])
                    graph.replay()
                    fr.append(frame.clone())
                return torch.cat(fr)
            tg, fr = timed(orbit_graph_k, 1)
            line.update(graph_frames_per_s=round(fr.shape[0] / tg, 1), graph_ms_per_frame=round(tg / fr.shape[0] * 1e3, 3))
            print(json.dumps(line), flush=True)
