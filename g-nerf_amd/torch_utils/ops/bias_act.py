"""Fused bias + activation: y = clamp(act(x + b) * gain).

Interface of the reference's torch_utils/ops/bias_act.py (activation_funcs :23, bias_act :54,
_bias_act_ref :92, _bias_act_cuda :128).  GPU tensors run the hand-written gfx950 kernel
(csrc/bias_act.hip) for the forward pass and for the first- and second-order gradients; CPU tensors, or
impl='ref', use PyTorch ops, which is what the reference does too.

Autograd design: two Function classes take the static configuration as a (non-tensor) argument instead
of being generated per configuration.  `_Forward` saves only what the activation's gradient formula
needs ('x', 'y' or nothing -- the `ref` field of the table below), `_Gradient` evaluates the plugin's
grad=1 form and is itself differentiable through the grad=2 form.
"""

import collections
import os

import numpy as np
import torch

from .. import custom_ops


class _Spec(dict):
    """Attribute-style dict (stands in for dnnlib.EasyDict so that this module has no other dependencies)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


_F = torch.nn.functional
_ROWS = (
    # name       function                                      alpha  gain         idx ref  2nd
    ('linear',   lambda x, **_: x,                             0,     1,           1,  '',  False),
    ('relu',     lambda x, **_: _F.relu(x),                    0,     np.sqrt(2),  2,  'y', False),
    ('lrelu',    lambda x, alpha, **_: _F.leaky_relu(x, alpha), 0.2,  np.sqrt(2),  3,  'y', False),
    ('tanh',     lambda x, **_: torch.tanh(x),                 0,     1,           4,  'y', True),
    ('sigmoid',  lambda x, **_: torch.sigmoid(x),              0,     1,           5,  'y', True),
    ('elu',      lambda x, **_: _F.elu(x),                     0,     1,           6,  'y', True),
    ('selu',     lambda x, **_: _F.selu(x),                    0,     1,           7,  'y', True),
    ('softplus', lambda x, **_: _F.softplus(x),                0,     1,           8,  'y', True),
    ('swish',    lambda x, **_: torch.sigmoid(x) * x,          0,     np.sqrt(2),  9,  'x', True),
)
# cuda_idx selects the kernel; ref names the saved tensor the gradient is written in terms of.
activation_funcs = {name: _Spec(func=fn, def_alpha=a, def_gain=g, cuda_idx=i, ref=r, has_2nd_grad=s2)
                    for name, fn, a, g, i, r, s2 in _ROWS}

_plugin = None
_null_tensor = torch.empty([0])

_Config = collections.namedtuple('_Config', 'dim act alpha gain clamp')


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='bias_act_plugin', sources=['bias_act.hip'], headers=['common.h'],
                                        source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'))
    return True


def _configure(dim, act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    return _Config(dim, act, float(spec.def_alpha if alpha is None else alpha),
                   float(spec.def_gain if gain is None else gain), float(-1 if clamp is None else clamp))


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """x: any shape; b: optional 1-D bias matching x.shape[dim]; act: a key of `activation_funcs`;
    alpha / gain: None = the activation's defaults; clamp: None or a non-negative bound applied last;
    impl: 'cuda' (the GPU kernel when x is on a GPU) or 'ref' (PyTorch ops).
    Returns a tensor shaped and typed like x.  First and second order gradients are supported."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _bias_act_cuda(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp).apply(x, b)
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """PyTorch-op implementation; autograd supplies every gradient order."""
    assert isinstance(x, torch.Tensor)
    cfg = _configure(dim, act, alpha, gain, clamp)
    with torch.autograd.profiler.record_function('_bias_act_ref'):
        y = x
        if b is not None:
            assert isinstance(b, torch.Tensor) and b.ndim == 1
            assert 0 <= dim < x.ndim
            assert b.shape[0] == x.shape[dim]
            shape = [1] * x.ndim
            shape[dim] = -1
            y = y + b.reshape(shape)
        y = activation_funcs[act].func(y, alpha=cfg.alpha)
        if cfg.gain != 1:
            y = y * cfg.gain
        if cfg.clamp >= 0:
            y = y.clamp(-cfg.clamp, cfg.clamp)
    return y


# ---------------------------------------------------------------------------------------------
# GPU path


def _layout_of(t):
    return torch.channels_last if t.ndim > 2 and t.stride(1) == 1 else torch.contiguous_format


def _kernel(cfg, x, b, xref, yref, dy, order):
    return _plugin.bias_act(x, b, xref, yref, dy, order, cfg.dim, activation_funcs[cfg.act].cuda_idx, cfg.alpha, cfg.gain, cfg.clamp)


def _reduce_to_bias(t, dim):
    return t.sum([i for i in range(t.ndim) if i != dim])


class _Forward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, b, cfg):
        spec = activation_funcs[cfg.act]
        ctx.cfg = cfg
        ctx.layout = _layout_of(x)
        ctx.identity = cfg.act == 'linear' and cfg.gain == 1 and cfg.clamp < 0
        x = x.contiguous(memory_format=ctx.layout)
        b = _null_tensor if b is None else b.contiguous()
        y = x if (ctx.identity and b is _null_tensor) else _kernel(cfg, x, b, _null_tensor, _null_tensor, _null_tensor, 0)
        keep_x = 'x' in spec.ref or spec.has_2nd_grad
        ctx.save_for_backward(x if keep_x else _null_tensor, b if keep_x else _null_tensor, y if 'y' in spec.ref else _null_tensor)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, b, y = ctx.saved_tensors
        dx = db = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            dy = dy.contiguous(memory_format=ctx.layout)
            dx = dy if ctx.identity else _Gradient.apply(dy, x, b, y, ctx.cfg)
        if ctx.needs_input_grad[1]:
            db = _reduce_to_bias(dx, ctx.cfg.dim)
        return dx, db, None


class _Gradient(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, x, b, y, cfg):
        ctx.cfg = cfg
        ctx.layout = _layout_of(dy)
        dx = _kernel(cfg, dy, b, x, y, _null_tensor, 1)
        ctx.save_for_backward(dy if activation_funcs[cfg.act].has_2nd_grad else _null_tensor, x, b, y)
        return dx

    @staticmethod
    def backward(ctx, d_dx):
        dy, x, b, y = ctx.saved_tensors
        cfg = ctx.cfg
        second = activation_funcs[cfg.act].has_2nd_grad
        d_dx = d_dx.contiguous(memory_format=ctx.layout)
        d_dy = d_x = d_b = None
        if ctx.needs_input_grad[0]:
            d_dy = _Gradient.apply(d_dx, x, b, y, cfg)
        if second and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            d_x = _kernel(cfg, d_dx, b, x, y, dy, 2)
        if second and ctx.needs_input_grad[2]:
            d_b = _reduce_to_bias(d_x, cfg.dim)
        return d_dy, d_x, d_b, None, None


class _Bound:
    """What `_bias_act_cuda(...)` returns: the forward Function bound to one static configuration
    (callers use `.apply(x, b)`, as with the reference's per-configuration classes)."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.identity = cfg.act == 'linear' and cfg.gain == 1 and cfg.clamp < 0

    def apply(self, x, b=None):
        # inference fast path: no graph wanted, so skip the autograd.Function machinery (~3 us of host time per call)
        if not (torch.is_grad_enabled() and (x.requires_grad or (b is not None and b.requires_grad))):
            x = x.contiguous(memory_format=_layout_of(x))
            if b is None:
                return x if self.identity else _kernel(self.cfg, x, _null_tensor, _null_tensor, _null_tensor, _null_tensor, 0)
            return _kernel(self.cfg, x, b.contiguous(), _null_tensor, _null_tensor, _null_tensor, 0)
        return _Forward.apply(x, b, self.cfg)


_bias_act_cuda_cache = dict()


def _bias_act_cuda(dim=1, act='linear', alpha=None, gain=None, clamp=None):
    cfg = _configure(dim, act, alpha, gain, clamp)
    if cfg not in _bias_act_cuda_cache:
        _bias_act_cuda_cache[cfg] = _Bound(cfg)
    return _bias_act_cuda_cache[cfg]
