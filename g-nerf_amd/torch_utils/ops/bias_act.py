"""Fused bias + activation with the reference's interface (torch_utils/ops/bias_act.py:
activation_funcs :23, bias_act :54, _bias_act_ref :92, _bias_act_cuda :128).

GPU tensors run the hand-written gfx950 kernel (csrc/bias_act.hip) for the forward pass and for the
first- and second-order gradients; CPU tensors, or impl='ref', use PyTorch ops like the reference."""

import os

import numpy as np
import torch

from .. import custom_ops


class _Spec(dict):
    """Attribute-style dict (stands in for dnnlib.EasyDict so this module has no other dependencies)."""
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _spec(func, def_alpha, def_gain, cuda_idx, ref, has_2nd_grad):
    return _Spec(func=func, def_alpha=def_alpha, def_gain=def_gain, cuda_idx=cuda_idx, ref=ref, has_2nd_grad=has_2nd_grad)


_F = torch.nn.functional
# cuda_idx selects the kernel; ref names the saved tensor the gradient is expressed in ('x', 'y' or neither).
activation_funcs = {
    'linear':   _spec(lambda x, **_: x,                          0,   1,          1, '',  False),
    'relu':     _spec(lambda x, **_: _F.relu(x),                 0,   np.sqrt(2), 2, 'y', False),
    'lrelu':    _spec(lambda x, alpha, **_: _F.leaky_relu(x, alpha), 0.2, np.sqrt(2), 3, 'y', False),
    'tanh':     _spec(lambda x, **_: torch.tanh(x),              0,   1,          4, 'y', True),
    'sigmoid':  _spec(lambda x, **_: torch.sigmoid(x),           0,   1,          5, 'y', True),
    'elu':      _spec(lambda x, **_: _F.elu(x),                  0,   1,          6, 'y', True),
    'selu':     _spec(lambda x, **_: _F.selu(x),                 0,   1,          7, 'y', True),
    'softplus': _spec(lambda x, **_: _F.softplus(x),             0,   1,          8, 'y', True),
    'swish':    _spec(lambda x, **_: torch.sigmoid(x) * x,       0,   np.sqrt(2), 9, 'x', True),
}

_plugin = None
_null_tensor = torch.empty([0])


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(
            module_name='bias_act_plugin',
            sources=['bias_act.hip'],
            headers=['common.h'],
            source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'),
        )
    return True


def _resolve(act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    spec = activation_funcs[act]
    return (spec, float(spec.def_alpha if alpha is None else alpha), float(spec.def_gain if gain is None else gain),
            float(-1 if clamp is None else clamp))


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b) * gain).

    x: any shape; b: optional 1-D bias matching x.shape[dim]; act: key of `activation_funcs`;
    alpha / gain: None = the activation's defaults; clamp: None or a non-negative bound;
    impl: 'cuda' (the GPU kernel when x is on a GPU) or 'ref' (PyTorch ops).
    Supports first and second order gradients."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _bias_act_cuda(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp).apply(x, b)
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """PyTorch-op implementation (autograd supplies every gradient order)."""
    assert isinstance(x, torch.Tensor)
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    with torch.autograd.profiler.record_function('_bias_act_ref'):
        if b is not None:
            assert isinstance(b, torch.Tensor) and b.ndim == 1
            assert 0 <= dim < x.ndim
            assert b.shape[0] == x.shape[dim]
            x = x + b.reshape([-1 if i == dim else 1 for i in range(x.ndim)])
        x = spec.func(x, alpha=alpha)
        if gain != 1:
            x = x * gain
        if clamp >= 0:
            x = x.clamp(-clamp, clamp)
    return x


_bias_act_cuda_cache = dict()


def _bias_act_cuda(dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """autograd.Function (cached per static-argument tuple) around the plugin's bias_act entry point."""
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    key = (dim, act, alpha, gain, clamp)
    if key in _bias_act_cuda_cache:
        return _bias_act_cuda_cache[key]

    trivial = act == 'linear' and gain == 1 and clamp < 0       # op reduces to "+ b"
    keeps_x = 'x' in spec.ref or spec.has_2nd_grad
    keeps_y = 'y' in spec.ref
    kernel_args = (dim, spec.cuda_idx, alpha, gain, clamp)

    def memory_format_of(t):
        return torch.channels_last if t.ndim > 2 and t.stride(1) == 1 else torch.contiguous_format

    def sum_to_bias(t):
        return t.sum([i for i in range(t.ndim) if i != dim])

    class BiasActCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):
            ctx.memory_format = memory_format_of(x)
            x = x.contiguous(memory_format=ctx.memory_format)
            b = b.contiguous() if b is not None else _null_tensor
            y = x
            if not trivial or b is not _null_tensor:
                y = _plugin.bias_act(x, b, _null_tensor, _null_tensor, _null_tensor, 0, *kernel_args)
            ctx.save_for_backward(x if keeps_x else _null_tensor, b if keeps_x else _null_tensor, y if keeps_y else _null_tensor)
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy if trivial else BiasActCudaGrad.apply(dy, x, b, y)
            if ctx.needs_input_grad[1]:
                db = sum_to_bias(dx)
            return dx, db

    class BiasActCudaGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):
            ctx.memory_format = memory_format_of(dy)
            dx = _plugin.bias_act(dy, b, x, y, _null_tensor, 1, *kernel_args)
            ctx.save_for_backward(dy if spec.has_2nd_grad else _null_tensor, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActCudaGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _plugin.bias_act(d_dx, b, x, y, dy, 2, *kernel_args)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = sum_to_bias(d_x)
            return d_dy, d_x, d_b, None

    _bias_act_cuda_cache[key] = BiasActCuda
    return BiasActCuda
