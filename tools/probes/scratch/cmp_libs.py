import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path[:0] = [os.path.join(ROOT, 'g-nerf_amd'), ROOT, os.path.join(ROOT, 'tests')]
    import torch, numpy as np
    import gnerf_hip
    from test_gpu_parity import _random_scene
    dev = torch.device('cuda', 0)
    planes, dec, o, d, nc, nf = _random_scene(7, N=2, res=16, S=48, F=48, hw=(64, 64))
    nhwc = gnerf_hip.planes_to_nhwc(planes.to(dev))
    outs = []
    for rep in range(3):
        rgb, depth, wsum = gnerf_hip.render_forward(nhwc, 2, [t.to(dev) for t in dec], o.to(dev), d.to(dev), nc.to(dev), nf.to(dev),
                                                    depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1.0, image_width=16, debug=False)
        outs.append(rgb.cpu().numpy())
    np.save(sys.argv[2], np.stack(outs))
else:
    import numpy as np
    res = {}
    for name, lib in (('new', ''), ('old', os.path.join(ROOT, 'g-nerf_amd/gnerf_hip/variants/libgnerf_D:GNERF_OLD_BLEND.so'))):
        env = dict(os.environ)
        if lib: env['GNERF_HIP_LIB'] = lib
        subprocess.run([sys.executable, __file__, 'child', f'/tmp/cmp_{name}.npy'], env=env, check=True)
        res[name] = np.load(f'/tmp/cmp_{name}.npy')
    a, b = res['new'], res['old']
    print('run-to-run identical new:', bool((a[0] == a[1]).all() and (a[0] == a[2]).all()), ' old:', bool((b[0] == b[1]).all()))
    d = np.abs(a[0] - b[0])
    print('shape', d.shape, 'max diff', d.max(), 'n > 1e-5:', int((d > 1e-5).sum()))
    rays = np.argwhere(d.max(-1) > 1e-5)
    print('rays with differences (item, ray):', rays[:40].tolist(), 'count', len(rays))
    for it, r in rays[:5]:
        print(it, r, 'diff per channel', np.round(d[it, r], 5).tolist())
