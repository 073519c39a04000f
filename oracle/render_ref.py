"""CPU oracle for the tri-plane importance renderer.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (g-nerf_amd/) never does: it fails loudly when the
HIP library is missing.

This is a from-scratch, ray-major restatement (explicit noise arguments, no
hidden RNG, own bilinear sampler) of what the reference computes on its pure
PyTorch path.  Each function names the reference lines it restates
(paths relative to /root/reference/g_nerf/).  It is pinned to the reference by
tests/golden/*.npz, which tests/golden/make_golden.py produced by importing the
reference in the build container (see tests/test_oracle_golden.py).

Everything is plain torch CPU in a selectable dtype (float32 = the reference's
arithmetic; float64 = a higher-precision "truth" used to judge which of two fp32
implementations is closer).
"""

import math
import torch

# ----------------------------------------------------------------------------
# Ray generation.  training/volumetric_rendering/ray_sampler.py:24-63


def make_rays(cam2world, intrinsics, resolution):
    """cam2world [N,4,4], intrinsics [N,3,3] -> origins [N,res*res,3], dirs [N,res*res,3].

    Ray m = i*res + j looks through the centre of pixel (row i, column j):
    x_cam = (j+.5)/res, y_cam = (i+.5)/res (ray_sampler.py:43-44 flips the
    meshgrid so that x follows the column index).
    """
    N = cam2world.shape[0]
    dt = cam2world.dtype
    fx = intrinsics[:, 0, 0:1]
    fy = intrinsics[:, 1, 1:2]
    cx = intrinsics[:, 0, 2:3]
    cy = intrinsics[:, 1, 2:3]
    sk = intrinsics[:, 0, 1:2]
    # torch.arange(res) * (1/res) + 0.5/res, exactly as ray_sampler.py:43
    ticks = torch.arange(resolution, dtype=torch.float32) * (1.0 / resolution) + (0.5 / resolution)
    ticks = ticks.to(dt)
    x_cam = ticks.repeat(resolution).unsqueeze(0).expand(N, -1)            # column index varies fastest
    y_cam = ticks.repeat_interleave(resolution).unsqueeze(0).expand(N, -1)
    # ray_sampler.py:51-52 (z_cam == 1)
    x_lift = (x_cam - cx + cy * sk / fy - sk * y_cam / fy) / fx
    y_lift = (y_cam - cy) / fy
    ones = torch.ones_like(x_lift)
    cam_pts = torch.stack((x_lift, y_lift, ones, ones), dim=-1)             # [N,M,4]
    world = torch.einsum('nij,nmj->nmi', cam2world, cam_pts)[:, :, :3]      # ray_sampler.py:56
    cam_loc = cam2world[:, :3, 3]
    dirs = world - cam_loc[:, None, :]
    dirs = torch.nn.functional.normalize(dirs, dim=2)                       # ray_sampler.py:59 (eps 1e-12)
    origins = cam_loc[:, None, :].expand(-1, dirs.shape[1], -1).contiguous()
    return origins, dirs


# ----------------------------------------------------------------------------
# Depth proposals.  renderer.py:169-192


def torch_linspace(start, end, steps, dtype):
    """Element k of torch.linspace(start, end, steps): ATen fills the first half
    as start + k*step and the second half as end - (steps-1-k)*step
    (aten/src/ATen/native/RangeFactories.cpp), all in the result dtype."""
    start_t = torch.tensor(start, dtype=dtype)
    end_t = torch.tensor(end, dtype=dtype)
    step = (end_t - start_t) / (steps - 1)
    k = torch.arange(steps)
    lo = start_t + step * k.to(dtype)
    hi = end_t - step * (steps - 1 - k).to(dtype)
    return torch.where(k < steps // 2, lo, hi)


def stratified_depths(noise, ray_start, ray_end, disparity=False):
    """noise [R,S] in [0,1) -> depths [R,S].  renderer.py:169-192.

    ray_start / ray_end are python floats (renderer.py:187-190) or [R] tensors
    (the 'auto' box-limit branch, renderer.py:183-186, which uses
    math_utils.linspace = start + k/(S-1) * (end-start), math_utils.py:101-118).
    """
    R, S = noise.shape
    dt = noise.dtype
    if disparity:                                                           # renderer.py:174-181
        base = torch_linspace(0.0, 1.0, S, dt)[None, :]
        d = base + noise * (1.0 / (S - 1))
        return 1.0 / (1.0 / ray_start * (1.0 - d) + 1.0 / ray_end * d)
    if isinstance(ray_start, torch.Tensor):
        steps = (torch.arange(S, dtype=torch.float32) / (S - 1)).to(dt)     # math_utils.py:107
        rs = ray_start.reshape(R, 1).to(dt)
        re = ray_end.reshape(R, 1).to(dt)
        base = rs + steps[None, :] * (re - rs)
        delta = (re - rs) / (S - 1)
        return base + noise * delta
    base = torch_linspace(ray_start, ray_end, S, dt)[None, :]
    delta = (ray_end - ray_start) / (S - 1)                                 # python double, renderer.py:189
    return base + noise * delta


# ----------------------------------------------------------------------------
# Tri-plane feature lookup.  renderer.py:23-65


def plane_uv(points, box_warp):
    """points [..., 3] -> uv [3, ..., 2] in grid_sample's [-1,1] convention.

    renderer.py:61 scales by 2/box_warp; project_onto_planes (renderer.py:39-53)
    multiplies by inv(plane_axes) and keeps two components.  For the three axes
    of generate_planes (renderer.py:23-37) this is the selection
        plane 0: (u,v) = (x,y)   plane 1: (x,z)   plane 2: (z,x)
    (inverse of a permutation matrix = its transpose; the matmul with exact 0/1
    entries is an exact selection for finite inputs)."""
    p = points * (2.0 / box_warp)
    x, y, z = p[..., 0], p[..., 1], p[..., 2]
    return torch.stack((torch.stack((x, y), -1), torch.stack((x, z), -1), torch.stack((z, x), -1)), 0)


def bilinear_zeros(plane_chw, uv):
    """plane_chw [C,H,W]; uv [P,2] -> [P,C].  Semantics of
    F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False)
    as called at renderer.py:64: u indexes W, v indexes H, pixel = ((g+1)*size-1)/2,
    out-of-range taps contribute zero."""
    C, H, W = plane_chw.shape
    dt = plane_chw.dtype
    ix = ((uv[:, 0] + 1) * W - 1) / 2
    iy = ((uv[:, 1] + 1) * H - 1) / 2
    x0 = torch.floor(ix)
    y0 = torch.floor(iy)
    x1 = x0 + 1
    y1 = y0 + 1
    w_nw = (x1 - ix) * (y1 - iy)
    w_ne = (ix - x0) * (y1 - iy)
    w_sw = (x1 - ix) * (iy - y0)
    w_se = (ix - x0) * (iy - y0)
    flat = plane_chw.reshape(C, H * W)
    out = torch.zeros(uv.shape[0], C, dtype=dt)
    for xs, ys, ws in ((x0, y0, w_nw), (x1, y0, w_ne), (x0, y1, w_sw), (x1, y1, w_se)):
        ok = (xs >= 0) & (xs <= W - 1) & (ys >= 0) & (ys <= H - 1)
        xi = xs.clamp(0, W - 1).long()
        yi = ys.clamp(0, H - 1).long()
        tap = flat[:, yi * W + xi].t()                                       # [P,C]
        out = out + tap * (ws * ok.to(dt))[:, None]
    return out


def triplane_features(planes_item, points, box_warp):
    """planes_item [3,C,H,W], points [P,3] -> mean over the 3 planes [P,C]
    (sample_from_planes renderer.py:55-65 + the mean at triplane.py:126)."""
    uv = plane_uv(points, box_warp)
    acc = None
    for p in range(3):
        f = bilinear_zeros(planes_item[p], uv[p])
        acc = f if acc is None else acc + f
    return acc / 3.0


# ----------------------------------------------------------------------------
# Decoder MLP.  triplane.py:113-136 + networks_stylegan2.py:101-134


def fold_decoder(weight1, bias1, weight2, bias2, lr_mul=1.0):
    """Raw OSGDecoder parameters -> effective (W1,b1,W2,b2) with
    FullyConnectedLayer's runtime gains folded in (networks_stylegan2.py:118-127:
    weight_gain = lr_mul/sqrt(in_features), bias_gain = lr_mul)."""
    g1 = lr_mul / math.sqrt(weight1.shape[1])
    g2 = lr_mul / math.sqrt(weight2.shape[1])
    return weight1 * g1, bias1 * lr_mul, weight2 * g2, bias2 * lr_mul


def decoder_mlp(x, W1, b1, W2, b2):
    """x [P,32] -> sigma [P], rgb [P,32].  triplane.py:124-136: addmm, Softplus
    (beta 1, threshold 20), addmm; rgb = sigmoid(o[1:])*1.002 - 0.001; sigma = o[0]."""
    h = torch.nn.functional.softplus(x @ W1.t() + b1)
    o = h @ W2.t() + b2
    rgb = torch.sigmoid(o[:, 1:]) * (1 + 2 * 0.001) - 0.001
    return o[:, 0], rgb


# ----------------------------------------------------------------------------
# Ray marching.  ray_marcher.py:25-57


def march_weights(sigma, depths):
    """sigma [R,S], depths [R,S] (ascending) -> weights [R,S-1], depth midpoints [R,S-1]."""
    delta = depths[:, 1:] - depths[:, :-1]
    sig_mid = (sigma[:, :-1] + sigma[:, 1:]) / 2
    t_mid = (depths[:, :-1] + depths[:, 1:]) / 2
    sig_mid = torch.nn.functional.softplus(sig_mid - 1)                      # ray_marcher.py:33
    alpha = 1 - torch.exp(-(sig_mid * delta))
    shifted = torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha + 1e-10], -1)
    trans = torch.cumprod(shifted, -1)[:, :-1]                               # exclusive product
    return alpha * trans, t_mid


def composite(colors, sigma, depths, white_back=False, depth_clamp=None):
    """colors [R,S,C], sigma [R,S], depths [R,S] -> rgb [R,C], depth [R], weights [R,S-1].

    depth_clamp=(lo,hi) reproduces ray_marcher.py:49-50 (NaN -> +inf, then clamp
    to the min/max over ALL depths of the call); None returns the unclamped value
    (still NaN -> inf)."""
    w, t_mid = march_weights(sigma, depths)
    c_mid = (colors[:, :-1] + colors[:, 1:]) / 2
    rgb = (w[:, :, None] * c_mid).sum(1)
    w_total = w.sum(1)
    depth = (w * t_mid).sum(1) / w_total
    depth = torch.nan_to_num(depth, float('inf'))
    if depth_clamp is not None:
        depth = torch.clamp(depth, depth_clamp[0], depth_clamp[1])
    if white_back:
        rgb = rgb + 1 - w_total[:, None]
    rgb = rgb * 2 - 1
    return rgb, depth, w


# ----------------------------------------------------------------------------
# Importance resampling.  renderer.py:194-253


def importance_depths(depths_c, weights_c, noise_f, eps=1e-5):
    """depths_c [R,S], weights_c [R,S-1], noise_f [R,F] in [0,1) -> fine depths [R,F] (unsorted).

    renderer.py:203-206: w' = avg_pool1d(max_pool1d(w, 2, 1, padding=1), 2, 1) + 0.01
    renderer.py:208-209: bins = depth midpoints (S-1 of them), pdf from w'[1:-1] (S-3)
    renderer.py:232-252: normalised cdf with leading 0, searchsorted(right=True),
    clamp below/above, linear inversion with the denom<eps -> 1 rule."""
    R, S = depths_c.shape
    w = weights_c
    neg = torch.full_like(w[:, :1], float('-inf'))
    padded = torch.cat([neg, w, neg], 1)
    mx = torch.maximum(padded[:, :-1], padded[:, 1:])                        # [R,S]
    sm = (mx[:, :-1] + mx[:, 1:]) / 2 + 0.01                                 # [R,S-1]
    bins = 0.5 * (depths_c[:, :-1] + depths_c[:, 1:])                        # [R,S-1]
    pw = sm[:, 1:-1] + eps                                                   # [R,S-3]
    pdf = pw / pw.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)                 # [R,S-2]
    n_w = pw.shape[1]
    u = noise_f.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_w)
    cdf_b = torch.gather(cdf, 1, below)
    cdf_a = torch.gather(cdf, 1, above)
    bin_b = torch.gather(bins, 1, below)
    bin_a = torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < eps, torch.ones_like(denom), denom)
    return bin_b + (u - cdf_b) / denom * (bin_a - bin_b)


# ----------------------------------------------------------------------------
# Whole renderer.  renderer.py:88-140


def render(planes, decoder, origins, dirs, options, noise_c, noise_f, stages=None, sigma_noise=None):
    """planes [N,3,C,H,W]; decoder = effective (W1,b1,W2,b2) from fold_decoder;
    origins, dirs [N,M,3]; noise_c [N,M,S] (the reference's first draw,
    rand_like([N,M,S,1])); noise_f [N*M,F] (its second draw, rand(N*M,F)).
    sigma_noise = (coarse [N*M,S], fine [N*M,F]): the two randn_like draws of
    run_model (renderer.py:146-147) ALREADY multiplied by options['density_noise'],
    added to the respective pass's densities; None = density_noise off.

    Returns rgb [N,M,C], depth [N,M,1], weight_sum [N,M,1] like
    ImportanceRenderer.forward.  If `stages` is a dict it receives the
    intermediate tensors (for stage-by-stage debugging of the HIP kernel)."""
    N, M, _ = origins.shape
    dt = planes.dtype
    S = int(options['depth_resolution'])
    F = int(options.get('depth_resolution_importance', 0))
    box_warp = options['box_warp']
    assert options.get('clamp_mode', 'softplus') == 'softplus'              # ray_marcher.py:32-35
    white_back = bool(options.get('white_back', False))
    disparity = bool(options.get('disparity_space_sampling', False))
    W1, b1, W2, b2 = decoder
    C_out = W2.shape[0] - 1
    R = N * M
    o = origins.reshape(R, 3)
    d = dirs.reshape(R, 3)
    item = torch.arange(N).repeat_interleave(M)
    ray_start, ray_end = options['ray_start'], options['ray_end']

    def shade(depths):
        S_ = depths.shape[1]
        pts = o[:, None, :] + depths[:, :, None] * d[:, None, :]             # renderer.py:105
        sig = torch.empty(R, S_, dtype=dt)
        col = torch.empty(R, S_, C_out, dtype=dt)
        for n in range(N):
            sel = item == n
            feat = triplane_features(planes[n], pts[sel].reshape(-1, 3), box_warp)
            s_, c_ = decoder_mlp(feat, W1, b1, W2, b2)
            sig[sel] = s_.reshape(-1, S_)
            col[sel] = c_.reshape(-1, S_, C_out)
        return sig, col

    depths_c = stratified_depths(noise_c.reshape(R, S), ray_start, ray_end, disparity)
    sig_c, col_c = shade(depths_c)
    if sigma_noise is not None:
        sig_c = sig_c + sigma_noise[0].reshape(R, S).to(dt)                  # renderer.py:146-147, coarse run_model
    if stages is not None:
        stages.update(depths_coarse=depths_c, sigma_coarse=sig_c, colors_coarse=col_c)
    if F > 0:
        w_c, _ = march_weights(sig_c, depths_c)                              # renderer.py:118
        depths_f = importance_depths(depths_c, w_c, noise_f.reshape(R, F)).detach()    # renderer.py:198 no_grad, :211 detach
        sig_f, col_f = shade(depths_f)
        if sigma_noise is not None:
            sig_f = sig_f + sigma_noise[1].reshape(R, F).to(dt)              # ... and the fine one
        all_d = torch.cat([depths_c, depths_f], 1)                           # renderer.py:157-167
        all_d, order = torch.sort(all_d, dim=1, stable=True)
        all_s = torch.gather(torch.cat([sig_c, sig_f], 1), 1, order)
        all_c = torch.gather(torch.cat([col_c, col_f], 1), 1, order[:, :, None].expand(-1, -1, C_out))
        if stages is not None:
            stages.update(weights_coarse=w_c, depths_fine=depths_f, sigma_fine=sig_f,
                          depths_all=all_d, sigma_all=all_s)
    else:
        all_d, all_s, all_c = depths_c, sig_c, col_c
    clamp = (all_d.min(), all_d.max())                                       # ray_marcher.py:50 (global!)
    rgb, depth, w = composite(all_c, all_s, all_d, white_back, clamp)
    if stages is not None:
        stages.update(weights_all=w)
    return rgb.reshape(N, M, C_out), depth.reshape(N, M, 1), w.sum(1).reshape(N, M, 1)


def query_points(planes, decoder, points, box_warp):
    """ImportanceRenderer.run_model (renderer.py:142-148) for arbitrary points
    [N,P,3] -> sigma [N,P,1], rgb [N,P,C] (density_noise off)."""
    N, P, _ = points.shape
    W1, b1, W2, b2 = decoder
    sig = []
    col = []
    for n in range(N):
        feat = triplane_features(planes[n], points[n], box_warp)
        s_, c_ = decoder_mlp(feat, W1, b1, W2, b2)
        sig.append(s_)
        col.append(c_)
    return torch.stack(sig)[..., None], torch.stack(col)


# ----------------------------------------------------------------------------
# Camera helpers used by the harness.  camera_utils.py:89-106,155-174


def lookat_pose(yaw, pitch, radius, dtype=torch.float32):
    """LookAtPoseSampler.sample(yaw, pitch, radius=radius) with zero stddev ->
    cam2world [1,4,4] (camera_utils.py:89-106 + create_cam2world_matrix :155-174)."""
    theta = torch.tensor([[yaw]], dtype=dtype)
    phi = torch.tensor([[pitch]], dtype=dtype)
    org = torch.zeros(1, 3, dtype=dtype)
    org[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - theta)
    org[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - theta)
    org[:, 1:2] = radius * torch.cos(phi)

    def unit(v):
        return v / torch.norm(v, dim=-1, keepdim=True)

    fwd = unit(unit(-org))
    up = torch.tensor([[0.0, 1.0, 0.0]], dtype=dtype)
    right = -unit(torch.cross(up, fwd, dim=-1))
    up2 = unit(torch.cross(fwd, right, dim=-1))
    rot = torch.eye(4, dtype=dtype)[None].clone()
    rot[:, :3, :3] = torch.stack((right, up2, fwd), dim=-1)
    trans = torch.eye(4, dtype=dtype)[None].clone()
    trans[:, :3, 3] = org
    return trans @ rot
