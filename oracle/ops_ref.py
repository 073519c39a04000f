"""CPU oracle for the StyleGAN custom ops.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path never does.

numpy restatements (float64 internally unless told otherwise) of what the
reference's `impl='ref'` paths and native plugins compute:
    bias_act        torch_utils/ops/bias_act.py:92-122 (forward), and the
                    grad=1 / grad=2 forms the plugin evaluates
                    (torch_utils/ops/bias_act.cu:27-151)
    upfirdn2d       torch_utils/ops/upfirdn2d.py:168-213
    filtered_lrelu  torch_utils/ops/filtered_lrelu.py:122-155
Pinned by tests/golden/ops_*.npz (generated from the reference, see
tests/golden/make_golden.py).
"""

import numpy as np

SELU_SCALE = 1.0507009873554804934193349852946
SELU_ALPHA = 1.6732632423543772848170429916717

# name -> (plugin index, default alpha, default gain, which saved tensor the
# gradient is written in terms of, has a second derivative)      bias_act.py:23-33
ACTIVATIONS = {
    'linear':   (1, 0.0, 1.0,          '',  False),
    'relu':     (2, 0.0, np.sqrt(2.0), 'y', False),
    'lrelu':    (3, 0.2, np.sqrt(2.0), 'y', False),
    'tanh':     (4, 0.0, 1.0,          'y', True),
    'sigmoid':  (5, 0.0, 1.0,          'y', True),
    'elu':      (6, 0.0, 1.0,          'y', True),
    'selu':     (7, 0.0, 1.0,          'y', True),
    'softplus': (8, 0.0, 1.0,          'y', True),
    'swish':    (9, 0.0, np.sqrt(2.0), 'x', True),
}


def _act(name, x, alpha):
    if name == 'linear':
        return x
    if name == 'relu':
        return np.maximum(x, 0)
    if name == 'lrelu':
        return np.where(x > 0, x, x * alpha)
    if name == 'tanh':
        return np.tanh(x)
    if name == 'sigmoid':
        return 1 / (1 + np.exp(-x))
    if name == 'elu':
        return np.where(x >= 0, x, np.expm1(np.minimum(x, 0)))
    if name == 'selu':
        return SELU_SCALE * np.where(x >= 0, x, SELU_ALPHA * np.expm1(np.minimum(x, 0)))
    if name == 'softplus':
        return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))
    if name == 'swish':
        return x / (1 + np.exp(-x))
    raise KeyError(name)


def _act_d1(name, x, alpha):
    """d act / dx evaluated at pre-activation x."""
    if name == 'linear':
        return np.ones_like(x)
    if name == 'relu':
        return (x > 0).astype(x.dtype)
    if name == 'lrelu':
        return np.where(x > 0, 1.0, alpha)
    if name == 'tanh':
        return 1 - np.tanh(x) ** 2
    s = 1 / (1 + np.exp(-x))
    if name == 'sigmoid':
        return s * (1 - s)
    if name == 'elu':
        return np.where(x >= 0, 1.0, np.exp(np.minimum(x, 0)))
    if name == 'selu':
        return SELU_SCALE * np.where(x >= 0, 1.0, SELU_ALPHA * np.exp(np.minimum(x, 0)))
    if name == 'softplus':
        return s
    if name == 'swish':
        return s + x * s * (1 - s)
    raise KeyError(name)


def _act_d2(name, x, alpha):
    """d2 act / dx2 evaluated at pre-activation x."""
    if name in ('linear', 'relu', 'lrelu'):
        return np.zeros_like(x)
    if name == 'tanh':
        t = np.tanh(x)
        return -2 * t * (1 - t * t)
    s = 1 / (1 + np.exp(-x))
    if name == 'sigmoid':
        return s * (1 - s) * (1 - 2 * s)
    if name == 'elu':
        return np.where(x >= 0, 0.0, np.exp(np.minimum(x, 0)))
    if name == 'selu':
        return SELU_SCALE * np.where(x >= 0, 0.0, SELU_ALPHA * np.exp(np.minimum(x, 0)))
    if name == 'softplus':
        return s * (1 - s)
    if name == 'swish':
        ds = s * (1 - s)
        return 2 * ds + x * ds * (1 - 2 * s)
    raise KeyError(name)


def _resolve(act, alpha, gain, clamp):
    _, def_alpha, def_gain, _, _ = ACTIVATIONS[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    return alpha, gain, clamp


def _bias_shape(x, dim):
    return [-1 if i == dim else 1 for i in range(x.ndim)]


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Forward: clamp(act(x + b) * gain).  bias_act.py:92-122."""
    alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    out_dtype = x.dtype
    z = x.astype(np.float64)
    if b is not None:
        z = z + b.astype(np.float64).reshape(_bias_shape(x, dim))
    y = _act(act, z, alpha) * gain
    if clamp >= 0:
        y = np.clip(y, -clamp, clamp)
    return y.astype(out_dtype)


def bias_act_grad(dy, x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """First-order: dL/dx given dL/dy (what plugin grad=1 returns, bias_act.cu:56-146:
    dy * act'(x+b) * gain, zeroed where the forward output was clamped)."""
    alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    z = x.astype(np.float64)
    if b is not None:
        z = z + b.astype(np.float64).reshape(_bias_shape(x, dim))
    g = dy.astype(np.float64) * _act_d1(act, z, alpha) * gain
    if clamp >= 0:
        y = _act(act, z, alpha) * gain
        g = np.where((y > -clamp) & (y < clamp), g, 0.0)
    return g.astype(x.dtype)


def bias_act_grad2(d_dx, dy, x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Second-order: d/dx of <d_dx, bias_act_grad(dy, x)> (plugin grad=2,
    bias_act.cu:76-131: d_dx * dy * act''(x+b) * gain, same clamp mask)."""
    alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    z = x.astype(np.float64)
    if b is not None:
        z = z + b.astype(np.float64).reshape(_bias_shape(x, dim))
    g = d_dx.astype(np.float64) * dy.astype(np.float64) * _act_d2(act, z, alpha) * gain
    if clamp >= 0:
        y = _act(act, z, alpha) * gain
        g = np.where((y > -clamp) & (y < clamp), g, 0.0)
    return g.astype(x.dtype)


# ----------------------------------------------------------------------------


def _pair(v):
    if isinstance(v, int):
        return v, v
    a, b = v
    return int(a), int(b)


def _pad4(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    padding = [int(p) for p in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    return tuple(padding)


def upfirdn2d_out_size(in_size, up, down, pad0, pad1, taps):
    """upfirdn2d.cpp:39-40."""
    return (in_size * up + pad0 + pad1 - taps + down) // down


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
    """x [N,C,H,W]; f None | [taps] (separable) | [fh,fw].  upfirdn2d.py:168-213:
    zero-insert upsample, pad (negative = crop), correlate with the flipped
    filter (true convolution unless flip_filter), keep every down-th sample.
    gain multiplies the signal once (the reference splits it as gain**(ndim/2)
    per separable pass)."""
    N, C, H, W = x.shape
    upx, upy = _pair(up)
    downx, downy = _pair(down)
    px0, px1, py0, py1 = _pad4(padding)
    if f is None:
        f = np.ones([1, 1], dtype=np.float32)
    f = np.asarray(f, dtype=np.float64)
    if f.ndim == 1:
        f2 = np.outer(f, f)
    else:
        f2 = f
    fh, fw = f2.shape
    xd = x.astype(np.float64)
    upH, upW = H * upy, W * upx
    z = np.zeros((N, C, upH, upW), dtype=np.float64)
    z[:, :, ::upy, ::upx] = xd
    z = np.pad(z, [(0, 0), (0, 0), (max(py0, 0), max(py1, 0)), (max(px0, 0), max(px1, 0))])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    k = f2 * gain
    if not flip_filter:
        k = k[::-1, ::-1]
    oh, ow = z.shape[2] - fh + 1, z.shape[3] - fw + 1
    assert oh >= 1 and ow >= 1
    y = np.zeros((N, C, oh, ow), dtype=np.float64)
    for i in range(fh):
        for j in range(fw):
            y += k[i, j] * z[:, :, i:i + oh, j:j + ow]
    y = y[:, :, ::downy, ::downx]
    return y.astype(x.dtype)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0,
                   gain=np.sqrt(2.0), slope=0.2, clamp=None, flip_filter=False):
    """filtered_lrelu.py:122-155: bias -> up-FIR (gain up**2) -> lrelu*gain, clamp -> down-FIR."""
    y = bias_act(x, b)
    y = upfirdn2d(y, fu, up=up, padding=padding, gain=up ** 2, flip_filter=flip_filter)
    y = bias_act(y, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    y = upfirdn2d(y, fd, down=down, flip_filter=flip_filter)
    return y


def setup_filter(f, normalize=True, flip_filter=False, gain=1, separable=None):
    """upfirdn2d.py:72-116."""
    if f is None:
        f = 1
    f = np.asarray(f, dtype=np.float32)
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.size >= 8)
    if f.ndim == 1 and not separable:
        f = np.outer(f, f)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f[tuple(slice(None, None, -1) for _ in range(f.ndim))]
    f = f * (gain ** (f.ndim / 2))
    return np.ascontiguousarray(f, dtype=np.float32)


def grid_sample_2d(image, grid):
    """torch.nn.functional.grid_sample(image, grid, mode='bilinear', padding_mode='zeros', align_corners=False), the call
    grid_sample_gradfix.grid_sample makes (grid_sample_gradfix.py:45).  image [N,C,H,W], grid [N,Ho,Wo,2] (x, y in [-1,1])
    -> [N,C,Ho,Wo]; float64 arithmetic.  pixel = ((g + 1) * size - 1) / 2, taps outside the image contribute zero."""
    image = np.asarray(image, dtype=np.float64)
    grid = np.asarray(grid, dtype=np.float64)
    N, C, H, W = image.shape
    ix = ((grid[..., 0] + 1) * W - 1) / 2
    iy = ((grid[..., 1] + 1) * H - 1) / 2
    x0, y0 = np.floor(ix), np.floor(iy)
    fx, fy = ix - x0, iy - y0
    out = np.zeros((N, C) + grid.shape[1:3], dtype=np.float64)
    n_idx = np.arange(N)[:, None, None]
    for dy, wy in ((0, 1 - fy), (1, fy)):
        for dx, wx in ((0, 1 - fx), (1, fx)):
            xx, yy = (x0 + dx).astype(np.int64), (y0 + dy).astype(np.int64)
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            vals = image[n_idx, :, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]          # [N,Ho,Wo,C]
            out += np.moveaxis(vals * (wx * wy * ok)[..., None], -1, 1)
    return out
