"""ctypes binding of libgnerf_hip.so (C ABI in include/gnerf_hip.h).

This is the only place that touches the native library.  PyTorch is used for what it is
good at here -- device memory, the current HIP stream, dtypes -- and nothing else: every
function below hands raw device pointers to a hand-written gfx950 kernel.

There is NO fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNERF_HIP_LIB points tools/ablate.py at a timing-only variant build; everything else uses the in-tree library.
LIB_PATH = os.environ.get('GNERF_HIP_LIB') or os.path.join(_HERE, 'libgnerf_hip.so')

_lib = None

F32, F16, F64 = 0, 1, 2
_DTYPE_CODE = {torch.float32: F32, torch.float16: F16, torch.float64: F64}

MAX_SAMPLES = 256
DEBUG_SLOTS = 8
ABI_VERSION = 11
# decoder arithmetic of the fused renderer (GNERF_MLP_* in include/gnerf_hip.h)
MLP_MODES = {'auto': 0, 'f16x3': 1, 'f32': 2}

_c_p = ctypes.c_void_p
_c_i = ctypes.c_int
_c_i64 = ctypes.c_int64
_c_f = ctypes.c_float


class RenderParams(ctypes.Structure):
    """struct gnerf_render_params (include/gnerf_hip.h)."""
    _fields_ = [
        ('planes_nhwc', _c_p), ('n_items', ctypes.c_int32), ('plane_h', ctypes.c_int32), ('plane_w', ctypes.c_int32),
        ('ray_origins', _c_p), ('ray_dirs', _c_p), ('rays_per_item', ctypes.c_int32), ('image_width', ctypes.c_int32),
        ('w1', _c_p), ('b1', _c_p), ('w2', _c_p), ('b2', _c_p),
        ('depth_resolution', ctypes.c_int32), ('depth_resolution_importance', ctypes.c_int32),
        ('ray_start', _c_f), ('ray_end', _c_f),
        ('ray_start_per_ray', _c_p), ('ray_end_per_ray', _c_p),
        ('box_warp', _c_f), ('white_back', ctypes.c_int32), ('disparity_space_sampling', ctypes.c_int32),
        ('noise_coarse', _c_p), ('noise_fine', _c_p),
        ('out_rgb', _c_p), ('out_depth', _c_p), ('out_wsum', _c_p),
        ('workspace', _c_p), ('debug', _c_p),
        ('planes_absmax', _c_p), ('mlp_mode', ctypes.c_int32), ('planes_interleaved', ctypes.c_int32),
        ('planes_shared', ctypes.c_int32), ('depth_clamp_per_item', ctypes.c_int32),
        ('cam2world', _c_p), ('intrinsics', _c_p), ('rng_mode', ctypes.c_int32), ('rng_per_item', ctypes.c_int32),
        ('rng_seed', ctypes.c_uint64), ('rng_offset_coarse', ctypes.c_uint64), ('rng_offset_fine', ctypes.c_uint64),
        ('rng_offset_item_stride', ctypes.c_uint64), ('rng_threads_coarse', ctypes.c_uint32), ('rng_threads_fine', ctypes.c_uint32),
        ('sigma_noise_coarse', _c_p), ('sigma_noise_fine', _c_p),
    ]


class RenderGrads(ctypes.Structure):
    """struct gnerf_render_grads (include/gnerf_hip.h)."""
    _fields_ = [
        ('grad_rgb', _c_p), ('grad_depth', _c_p), ('grad_wsum', _c_p),
        ('grad_planes_nhwc', _c_p),
        ('grad_w1', _c_p), ('grad_b1', _c_p), ('grad_w2', _c_p), ('grad_b2', _c_p),
        ('scatter_stage', _c_p),
    ]


# name -> (restype, argtypes); must list every function include/gnerf_hip.h declares (tests check this).
SIGNATURES = {
    'gnerf_abi_version': (_c_i, []),
    'gnerf_last_error': (ctypes.c_char_p, []),
    'gnerf_build_info': (ctypes.c_char_p, []),
    'gnerf_clock_sample': (_c_i, [_c_p, ctypes.c_double, _c_p]),
    'gnerf_bias_act': (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i64, _c_i, _c_i64, _c_i, _c_i, _c_f, _c_f, _c_f, _c_p]),
    'gnerf_upfirdn2d': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, ctypes.POINTER(_c_i64), _c_i, _c_i, ctypes.POINTER(_c_i64),
                               _c_i, _c_i, ctypes.POINTER(_c_i64), _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_p]),
    'gnerf_filtered_lrelu_act': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, ctypes.POINTER(_c_i64), _c_i, _c_i, _c_i, _c_i,
                                        _c_f, _c_f, _c_f, _c_i, _c_p]),
    'gnerf_filtered_lrelu': (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, ctypes.POINTER(_c_i64),
                                    _c_i, _c_i, ctypes.POINTER(_c_i64), _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i,
                                    _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_f, _c_f, _c_i, _c_p]),
    'gnerf_grid_sample_2d': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, ctypes.POINTER(_c_i64), _c_i, _c_i, _c_p]),
    'gnerf_grid_sample_2d_backward': (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, ctypes.POINTER(_c_i64), _c_i, _c_i, _c_p]),
    'gnerf_planes_to_nhwc': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_planes_to_nhwc_stats': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p]),
    'gnerf_planes_absmax': (_c_i, [_c_p, _c_i64, _c_p, _c_p]),
    'gnerf_planes_from_nhwc': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_make_rays': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_p, _c_p, _c_p]),
    'gnerf_torch_rand_plan': (_c_i, [_c_i64, _c_i, _c_i, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)]),
    'gnerf_torch_rand': (_c_i, [_c_p, _c_i64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32, _c_p]),
    'gnerf_to_uint8_nhwc': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_render_workspace_bytes': (ctypes.c_size_t, []),
    'gnerf_render_forward': (_c_i, [ctypes.POINTER(RenderParams), _c_p]),
    'gnerf_render_backward': (_c_i, [ctypes.POINTER(RenderParams), ctypes.POINTER(RenderGrads), _c_p]),
    'gnerf_render_backward_stage_bytes': (ctypes.c_size_t, [ctypes.POINTER(RenderParams)]),
    'gnerf_render_backward_exchange_bytes': (ctypes.c_size_t, [ctypes.POINTER(RenderParams)]),
    'gnerf_query_points': (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_i, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_p]),
    'gnerf_query_points_backward': (_c_i, [_c_p, _c_i, _c_i, _c_i, _c_p, _c_i, _c_f, _c_p, _c_p, _c_p, _c_p, _c_p, _c_p,
                                           _c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_p]),
    'gnerf_modulate_weights': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_normalise_styles': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_p]),
    'gnerf_scale_channels': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_modconv_epilogue': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i, _c_i, _c_p, _c_i, _c_f, _c_f, _c_f, _c_p]),
    'gnerf_conv3x3_epilogue_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i, _c_p, _c_f, _c_f, _c_f, _c_p, _c_p]),
    'gnerf_conv_transpose3x3_s2_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_conv3x3_epilogue_torgb_nhwc': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i, _c_p, _c_f, _c_f, _c_f, _c_p, _c_p, _c_f, _c_p, _c_p]),
    'gnerf_split_f16x3_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_p, _c_p]),
    'gnerf_make_rays_and_draws': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_p, _c_p, _c_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32,
                                         _c_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, _c_p]),
    'gnerf_conv3x3_f32x3_epilogue_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_p, _c_f, _c_f, _c_f, _c_p, _c_p]),
    'gnerf_conv_transpose3x3_s2_f32x3_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_upsample2x_add_nhwc': (_c_i, [_c_p, _c_p, ctypes.POINTER(_c_f), _c_i, _c_f, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p]),
    'gnerf_scale_channels_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p]),
    'gnerf_modconv_epilogue_nhwc': (_c_i, [_c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_p, _c_p, _c_i, _c_i, _c_p, _c_i, _c_f, _c_f, _c_f, _c_p, _c_p]),
    'gnerf_torgb_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_f, _c_p]),
    'gnerf_torgb_nhwc_accumulate': (_c_i, [_c_p, _c_p, _c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_f, _c_p]),
    'gnerf_blur4_epilogue_nhwc': (_c_i, [_c_p, _c_p, _c_p, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_i, _c_f, _c_p, _c_p, _c_i, _c_f, _c_f, _c_f, _c_p, _c_p]),
}


def profiled(name):
    """Decorator: the call runs inside torch.autograd.profiler.record_function(name) WHILE a profiler is collecting (Kineto,
    or emit_nvtx -> roctx ranges that rocprofv3 --marker-trace shows), and as a plain call otherwise -- a record_function entered with
    no profiler attached still costs microseconds of host time per call, which an orbit frame of ~190 launches cannot afford.
    The reference opens the same ranges with misc.profiled_function (misc.py:102-107; conv2d_resample.py:47, bias_act.py:92, ...)."""
    import functools

    def deco(fn):
        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            if torch.autograd._profiler_enabled():
                with torch.autograd.profiler.record_function(name):
                    return fn(*args, **kwargs)
            return fn(*args, **kwargs)
        return wrapper
    return deco


def load():
    """Load the library once.  Raises RuntimeError (never falls back) if it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError(f'{LIB_PATH} is missing: build it with g-nerf_amd/csrc/build.sh '
                           f'(or python -c "import __graft_entry__ as g; g.build()"). There is no fallback path.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.gnerf_abi_version() != ABI_VERSION:
        raise RuntimeError(f'libgnerf_hip.so ABI version {lib.gnerf_abi_version()} != {ABI_VERSION}: rebuild it (g-nerf_amd/csrc/build.sh)')
    _lib = lib
    return lib


def is_available():
    return os.path.isfile(LIB_PATH)


# The thin PyTorch-ROCm C++ extension over the same C ABI (csrc/torch_binding.cpp -> gnerf_torch_ext.so): pybind entry points
# with the reference plugins' exact signatures (bias_act.cpp:36, upfirdn2d.cpp:20, filtered_lrelu.cpp:20,217) plus
# render_forward.  It is the default binding of the public ops (custom_ops.get_plugin) because a call costs ~3 us of host
# time instead of ~11 through ctypes; GNERF_HIP_BINDING=ctypes forces the ctypes route, which stays complete and is what
# everything falls back to when the extension has not been built.  Either way the kernels are libgnerf_hip.so's.
EXT_PATH = os.path.join(_HERE, 'gnerf_torch_ext.so')
_ext = None


def ext():
    """The extension module, or None (not built, or GNERF_HIP_BINDING=ctypes).  GNERF_HIP_BINDING=ext makes absence an error."""
    global _ext
    if _ext is None:
        want = os.environ.get('GNERF_HIP_BINDING', '')
        if want == 'ctypes' or os.environ.get('GNERF_HIP_LIB'):          # variant builds of the library are ctypes-only
            _ext = False
        elif not os.path.isfile(EXT_PATH):
            if want == 'ext':
                raise RuntimeError(f'{EXT_PATH} is missing: build it with g-nerf_amd/csrc/build.sh')
            _ext = False
        else:
            load()
            import importlib.util
            spec = importlib.util.spec_from_file_location('gnerf_torch_ext', EXT_PATH)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            # abi_version() is the header version compiled INTO the extension (its struct layouts); load() has already held the
            # library to ABI_VERSION, and library_abi_version() is what the extension's own link resolved to
            if mod.abi_version() != ABI_VERSION or mod.library_abi_version() != ABI_VERSION:
                raise RuntimeError(f'gnerf_torch_ext.so was built against ABI {mod.abi_version()} (library it links: '
                                   f'{mod.library_abi_version()}) != {ABI_VERSION}: rebuild (csrc/build.sh)')
            _ext = mod
    return _ext or None


def _check(code, what):
    if code != 0:
        msg = load().gnerf_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what} failed ({code}): {msg}')


# The three helpers below sit on every call; written for low host overhead (the public ops are called ~45 times per
# generator forward): raw stream handle without building a Stream object, plain ints for pointers (argtypes are
# c_void_p), and no device context switch when the tensor already lives on the current device.
_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream(t):
    if _raw_stream is not None:
        return _raw_stream(t.device.index if t.device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(t.device).cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


class _NoSwitch:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def _on_device(device):
    """Context that makes `device` current for the launch; free when it already is."""
    if device.index is None or device.index == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def _strides(t):
    return (ctypes.c_int64 * t.ndim)(*t.stride())


def _is_dense(t):
    """Non-overlapping and dense in SOME dimension order (what ATen's is_non_overlapping_and_dense checks)."""
    if t.is_contiguous():
        return True
    expected = 1
    for stride, size in sorted((st, sz) for sz, st in zip(t.shape, t.stride()) if sz != 1):
        if stride != expected:
            return False
        expected *= size
    return True


def _same_layout(a, b):
    """has_same_layout of the reference's bias_act.cpp:18-29: strides are compared only where the size is >= 2 (a size-1
    dimension's stride is arbitrary, e.g. after .contiguous() on [N,C,1,1])."""
    return all(sa == sb for sz, sa, sb in zip(a.shape, a.stride(), b.stride()) if sz >= 2)


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and t.device.type != 'cuda':
            raise RuntimeError('gnerf_hip: tensor is not on a GPU device')


# ----------------------------------------------------------------------------


@profiled('gnerf_hip::bias_act')
def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
    """Same contract as bias_act_plugin.bias_act (reference bias_act.cpp:36): absent tensors are
    empty tensors (numel 0) or None; returns a new tensor laid out like x."""
    def opt(t):
        return None if t is None or t.numel() == 0 else t
    b, xref, yref, dy = opt(b), opt(xref), opt(yref), opt(dy)
    _require_cuda(x, b, xref, yref, dy)
    if x.dtype not in _DTYPE_CODE:
        raise RuntimeError(f'bias_act: unsupported dtype {x.dtype}')
    if not _is_dense(x):
        raise RuntimeError('bias_act: x must be non-overlapping and dense')
    for name, t in (('xref', xref), ('yref', yref), ('dy', dy)):
        if t is not None and (t.shape != x.shape or t.dtype != x.dtype or not _same_layout(t, x)):
            raise RuntimeError(f'bias_act: {name} must have the same shape, dtype and layout as x')
    size_b, step_b = 0, 1
    if b is not None:
        if b.ndim != 1 or b.dtype != x.dtype or not b.is_contiguous():
            raise RuntimeError('bias_act: b must be a contiguous 1-D tensor of the same dtype as x')
        if not 0 <= dim < x.ndim or b.numel() != x.shape[dim]:
            raise RuntimeError('bias_act: b has the wrong number of elements or dim is out of bounds')
        size_b, step_b = b.numel(), x.stride(dim)
    y = torch.empty_like(x)
    with _on_device(x.device):
        code = load().gnerf_bias_act(_ptr(x), _ptr(b), _ptr(xref), _ptr(yref), _ptr(dy), _ptr(y), _DTYPE_CODE[x.dtype],
                                     x.numel(), size_b, step_b, int(grad), int(act), float(alpha), float(gain), float(clamp), _stream(x))
    _check(code, 'gnerf_bias_act')
    return y


@profiled('gnerf_hip::upfirdn2d')
def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
    """Same contract as upfirdn2d_plugin.upfirdn2d (reference upfirdn2d.cpp:20): x [N,C,H,W] in NCHW or
    channels_last, f float32 [fh,fw] on x's device; returns y in x's memory format."""
    _require_cuda(x, f)
    if x.ndim != 4 or f.ndim != 2 or f.dtype != torch.float32:
        raise RuntimeError('upfirdn2d: x must be rank 4 and f a rank-2 float32 tensor')
    if x.dtype not in _DTYPE_CODE:
        raise RuntimeError(f'upfirdn2d: unsupported dtype {x.dtype}')
    if x.numel() == 0 or f.numel() == 0:
        raise RuntimeError('upfirdn2d: x and f must not be empty')
    n, c, ih, iw = x.shape
    fh, fw = f.shape
    ow = (iw * upx + padx0 + padx1 - fw + downx) // downx
    oh = (ih * upy + pady0 + pady1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError('upfirdn2d: output must be at least 1x1')
    mf = torch.channels_last if (x.stride(1) == 1 and c > 1) else torch.contiguous_format
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device, memory_format=mf)
    with _on_device(x.device):
        code = load().gnerf_upfirdn2d(_ptr(x), _ptr(f), _ptr(y), _DTYPE_CODE[x.dtype], n, c, ih, iw, _strides(x),
                                      fh, fw, _strides(f), oh, ow, _strides(y), upx, upy, downx, downy, padx0, pady0,
                                      1 if flip else 0, float(gain), _stream(x))
    _check(code, 'gnerf_upfirdn2d')
    return y


@profiled('gnerf_hip::filtered_lrelu_act_')
def filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, write_signs):
    """Same contract as filtered_lrelu_plugin.filtered_lrelu_act_ (reference filtered_lrelu.cpp:217):
    in-place on x; returns the sign tensor written (or an empty tensor)."""
    _require_cuda(x)
    if x.ndim != 4 or x.dtype not in _DTYPE_CODE:
        raise RuntimeError('filtered_lrelu_act_: x must be a rank-4 float tensor')
    n, c, h, w = x.shape
    read_signs = si is not None and si.numel() > 0
    so = torch.empty([0], dtype=torch.uint8, device=x.device)
    s_h = s_w = 0
    mode = 0
    s = None
    if read_signs:
        _require_cuda(si)
        if si.dtype != torch.uint8 or si.ndim != 4 or not si.is_contiguous():
            raise RuntimeError('filtered_lrelu_act_: si must be a contiguous rank-4 uint8 tensor')
        s, s_h, s_w, mode = si, si.shape[2], si.shape[3] * 4, 2
    elif write_signs:
        s_w = (w + 15) & ~15
        s_h = h
        so = torch.empty([n, c, s_h, s_w // 4], dtype=torch.uint8, device=x.device)
        s, mode, sx, sy = so, 1, 0, 0
    with _on_device(x.device):
        code = load().gnerf_filtered_lrelu_act(_ptr(x), _ptr(s), _DTYPE_CODE[x.dtype], n, c, h, w, _strides(x), s_h, s_w, int(sx), int(sy),
                                               float(gain), float(slope), float(clamp), mode, _stream(x))
    _check(code, 'gnerf_filtered_lrelu_act')
    return so


E_UNSUPPORTED = -3


@profiled('gnerf_hip::filtered_lrelu')
def filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filters, write_signs):
    """Same contract as filtered_lrelu_plugin.filtered_lrelu (reference filtered_lrelu.cpp:20-213): returns
    (y, so, rc); rc = -1 with empty tensors means "no fused kernel for this configuration" and the caller runs the
    generic three-launch route (filtered_lrelu.py:225-231).  Anything else that goes wrong raises."""
    _require_cuda(x)
    if x.ndim != 4 or x.numel() == 0:
        raise RuntimeError('filtered_lrelu: x must be a non-empty rank-4 tensor')
    for f, name in ((fu, 'fu'), (fd, 'fd')):
        _require_cuda(f)
        if f.dtype != torch.float32 or f.ndim not in (1, 2) or f.numel() == 0:
            raise RuntimeError(f'filtered_lrelu: {name} must be a non-empty float32 tensor of rank 1 or 2')
    _require_cuda(b)
    if b.dtype != x.dtype or b.ndim != 1 or b.shape[0] != x.shape[1]:
        raise RuntimeError('filtered_lrelu: b must be a vector with one entry per channel of x, same dtype')
    if up < 1 or down < 1:
        raise RuntimeError('filtered_lrelu: up and down must be at least 1')
    none = (torch.empty([0], device=x.device), torch.empty([0], device=x.device), -1)
    if x.dtype not in (torch.float32, torch.float16):
        return none
    if (fu.ndim == 2 and tuple(fu.shape) != (1, 1)) or (fd.ndim == 2 and tuple(fd.shape) != (1, 1)):
        return none                                            # non-separable filters: generic route
    n, c, xh, xw = x.shape
    fut, fdt = fu.shape[-1] - 1, fd.shape[-1] - 1
    cw, chh = xw * up + (px0 + px1) - fut, xh * up + (py0 + py1) - fut
    if not (cw > fdt and chh > fdt):
        raise RuntimeError('filtered_lrelu: upsampled buffer must be at least the size of downsampling filter')
    yw, yh = (cw - fdt + (down - 1)) // down, (chh - fdt + (down - 1)) // down
    if yw < 1 or yh < 1:
        raise RuntimeError('filtered_lrelu: output must be at least 1x1')
    channels_last = x.stride(1) == 1 and c > 1
    y = torch.empty([n, c, yh, yw], dtype=x.dtype, device=x.device,
                    memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    read_signs = si is not None and si.numel() > 0
    so = torch.empty([0], dtype=torch.uint8, device=x.device)
    s, s_h, s_w, mode = None, 0, 0, 0
    if write_signs:
        s_h = yh * down - (down - 1) + fdt
        s_w = (yw * down - (down - 1) + fdt + 15) & ~15
        so = torch.empty([n, c, s_h, s_w >> 2], dtype=torch.uint8, device=x.device)
        s, mode = so, 1
    elif read_signs:
        _require_cuda(si)
        if si.dtype != torch.uint8 or si.ndim != 4 or not si.is_contiguous() or si.shape[0] != n or si.shape[1] != c:
            raise RuntimeError('filtered_lrelu: signs must be a contiguous uint8 [n, c, h, w/4] tensor matching x')
        s, s_h, s_w, mode = si, si.shape[2], si.shape[3] * 4, 2
    fu_c, fd_c, b_c = fu.contiguous(), fd.contiguous(), b.contiguous()
    with _on_device(x.device):
        code = load().gnerf_filtered_lrelu(_ptr(x), _ptr(fu_c), _ptr(fd_c), _ptr(b_c), _ptr(s), _ptr(y), _DTYPE_CODE[x.dtype],
                                           n, c, xh, xw, _strides(x), yh, yw, _strides(y),
                                           fu.shape[-1], fu.ndim, fd.shape[-1], fd.ndim, int(up), int(down), int(px0), int(py0),
                                           s_h, s_w, int(sx), int(sy), mode, float(gain), float(slope), float(clamp),
                                           1 if flip_filters else 0, _stream(x))
    if code == E_UNSUPPORTED:
        return none
    _check(code, 'gnerf_filtered_lrelu')
    return y, so, 0


def grid_sample_supported(image, grid):
    """True when the native sampler covers this call (GPU tensors, float16/float32 image, 4-D, positive strides)."""
    return (image.is_cuda and grid.is_cuda and image.ndim == 4 and grid.ndim == 4 and grid.shape[-1] == 2 and grid.shape[0] == image.shape[0]
            and image.dtype in (torch.float32, torch.float16) and image.numel() > 0 and grid.numel() > 0
            and image.stride(2) > 0 and image.stride(3) > 0)


@profiled('gnerf_hip::grid_sample_2d')
def grid_sample_2d(image, grid):
    """Bilinear, zero padding, align_corners=False (what grid_sample_gradfix.grid_sample evaluates, grid_sample_gradfix.py:45):
    image [N,C,H,W], grid [N,Ho,Wo,2] -> [N,C,Ho,Wo] in image's dtype."""
    _require_cuda(image, grid)
    n, c, h, w = image.shape
    ho, wo = grid.shape[1], grid.shape[2]
    g = grid.float().contiguous()
    out = torch.empty([n, c, ho, wo], dtype=image.dtype, device=image.device)
    with _on_device(image.device):
        code = load().gnerf_grid_sample_2d(_ptr(image), _ptr(g), _ptr(out), _DTYPE_CODE[image.dtype], n, c, h, w, _strides(image), ho, wo, _stream(image))
    _check(code, 'gnerf_grid_sample_2d')
    return out


@profiled('gnerf_hip::grid_sample_2d_backward')
def grid_sample_2d_backward(grad_out, image, grid, need_image=True, need_grid=True):
    """The adjoint (aten::grid_sampler_2d_backward upstream, grid_sample_gradfix.py:62-77): returns (grad_image, grad_grid), each
    None when not requested; grad_image in image's dtype, grad_grid in grid's."""
    _require_cuda(grad_out, image, grid)
    n, c, h, w = image.shape
    ho, wo = grid.shape[1], grid.shape[2]
    g = grid.float().contiguous()
    go = grad_out.to(image.dtype).contiguous()
    gi = torch.zeros([n, c, h, w], dtype=torch.float32, device=image.device) if need_image else None
    gg = torch.zeros([n, ho, wo, 2], dtype=torch.float32, device=image.device) if need_grid else None
    with _on_device(image.device):
        code = load().gnerf_grid_sample_2d_backward(_ptr(go), _ptr(image), _ptr(g), _ptr(gi), _ptr(gg), _DTYPE_CODE[image.dtype],
                                                    n, c, h, w, _strides(image), ho, wo, _stream(image))
    _check(code, 'gnerf_grid_sample_2d_backward')
    return (None if gi is None else gi.to(image.dtype)), (None if gg is None else gg.to(grid.dtype))


@profiled('gnerf_hip::planes_to_nhwc')
def planes_to_nhwc(planes, with_absmax=False):
    """[N,3,C,H,W] (or [NP,C,H,W]) float32 NCHW -> [NP,H,W,C] contiguous.  with_absmax: also return max |planes| as a
    one-element device tensor, measured by the same pass (render_forward's planes_absmax)."""
    _require_cuda(planes)
    if planes.dtype != torch.float32:
        raise RuntimeError('planes_to_nhwc: planes must be float32')
    p = planes.reshape(-1, *planes.shape[-3:]).contiguous()
    np_, c, h, w = p.shape
    out = torch.empty([np_, h, w, c], dtype=torch.float32, device=p.device)
    if with_absmax:
        amax = torch.empty([1], dtype=torch.float32, device=p.device)
        with _on_device(p.device):
            code = load().gnerf_planes_to_nhwc_stats(_ptr(p), _ptr(out), np_, c, h, w, _ptr(amax), _stream(p))
        _check(code, 'gnerf_planes_to_nhwc_stats')
        return out, amax
    with _on_device(p.device):
        code = load().gnerf_planes_to_nhwc(_ptr(p), _ptr(out), np_, c, h, w, _stream(p))
    _check(code, 'gnerf_planes_to_nhwc')
    return out


@profiled('gnerf_hip::planes_absmax')
def planes_absmax(planes):
    """max |x| of a contiguous float32 device tensor -> one-element device tensor (NaN if any element is NaN)."""
    _require_cuda(planes)
    if planes.dtype != torch.float32 or not planes.is_contiguous() or planes.numel() == 0:
        raise RuntimeError('planes_absmax: expected a non-empty contiguous float32 tensor')
    amax = torch.empty([1], dtype=torch.float32, device=planes.device)
    with _on_device(planes.device):
        code = load().gnerf_planes_absmax(_ptr(planes), planes.numel(), _ptr(amax), _stream(planes))
    _check(code, 'gnerf_planes_absmax')
    return amax


@profiled('gnerf_hip::planes_from_nhwc')
def planes_from_nhwc(planes_nhwc, n_items=None):
    """[NP,H,W,C] float32 -> [NP,C,H,W] contiguous ([N,3,C,H,W] when n_items is given)."""
    _require_cuda(planes_nhwc)
    if planes_nhwc.dtype != torch.float32 or planes_nhwc.ndim != 4 or not planes_nhwc.is_contiguous():
        raise RuntimeError('planes_from_nhwc: expected a contiguous float32 [NP,H,W,C] tensor')
    np_, h, w, c = planes_nhwc.shape
    out = torch.empty([np_, c, h, w], dtype=torch.float32, device=planes_nhwc.device)
    with _on_device(planes_nhwc.device):
        code = load().gnerf_planes_from_nhwc(_ptr(planes_nhwc), _ptr(out), np_, c, h, w, _stream(planes_nhwc))
    _check(code, 'gnerf_planes_from_nhwc')
    return out if n_items is None else out.view(n_items, np_ // n_items, c, h, w)


@profiled('gnerf_hip::to_uint8_nhwc')
def to_uint8_nhwc(img):
    """(img * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous() for a float32 [N,C,H,W] GPU tensor in one launch
    (gen_videos.py:173 + the frame writer's layout).  Returns uint8 [N,H,W,C]."""
    _require_cuda(img)
    if img.dtype != torch.float32 or img.ndim != 4 or not (1 <= img.shape[1] <= 64):
        raise RuntimeError('to_uint8_nhwc: expected a float32 [N,C,H,W] tensor with 1..64 channels')
    x = img.detach().contiguous()
    n, c, h, w = x.shape
    out = torch.empty([n, h, w, c], dtype=torch.uint8, device=x.device)
    with _on_device(x.device):
        code = load().gnerf_to_uint8_nhwc(_ptr(x), _ptr(out), n, c, h, w, _stream(x))
    _check(code, 'gnerf_to_uint8_nhwc')
    return out


@profiled('gnerf_hip::make_rays')
def make_rays(cam2world, intrinsics, resolution):
    _require_cuda(cam2world, intrinsics)
    c2w = cam2world.to(torch.float32).contiguous()
    k = intrinsics.to(torch.float32).contiguous()
    n = c2w.shape[0]
    if c2w.shape != (n, 4, 4) or k.shape != (n, 3, 3):
        raise RuntimeError('make_rays: expected cam2world [N,4,4] and intrinsics [N,3,3]')
    o = torch.empty([n, resolution * resolution, 3], dtype=torch.float32, device=c2w.device)
    d = torch.empty_like(o)
    with _on_device(c2w.device):
        code = load().gnerf_make_rays(_ptr(c2w), _ptr(k), n, int(resolution), _ptr(o), _ptr(d), _stream(c2w))
    _check(code, 'gnerf_make_rays')
    return o, d


@profiled('gnerf_hip::make_rays_and_draws')
def make_rays_and_draws(cam2world, intrinsics, resolution, S, F, generator=None):
    """make_rays(cam2world, intrinsics, resolution) AND the renderer's two uniform draws -- torch.rand([N,M,S,1]) then torch.rand(N*M, F)
    (renderer.py:190,241) -- in ONE launch (gnerf_make_rays_and_draws).  The draws are the device generator's: the values torch.rand would
    have returned, bit for bit, and the generator is left where those two calls would have left it (torch_philox_plan).  Returns
    (origins [N,M,3], dirs [N,M,3], noise_coarse [N,M,S,1], noise_fine [N*M,F] or None).  Not inside a graph capture (the generator's
    offset lives on the device then): the caller draws with torch.rand."""
    _require_cuda(cam2world, intrinsics)
    c2w = cam2world.to(torch.float32).contiguous()
    k = intrinsics.to(torch.float32).contiguous()
    n, dev = c2w.shape[0], c2w.device
    if c2w.shape != (n, 4, 4) or k.shape != (n, 3, 3):
        raise RuntimeError('make_rays_and_draws: expected cam2world [N,4,4] and intrinsics [N,3,3]')
    m = int(resolution) * int(resolution)
    plan = torch_philox_plan(dev, n, m, int(S), int(F), generator=generator, advance=False)
    o = torch.empty([n, m, 3], dtype=torch.float32, device=dev)
    d = torch.empty_like(o)
    nc = torch.empty([n, m, int(S), 1], dtype=torch.float32, device=dev)
    nf = torch.empty([n * m, int(F)], dtype=torch.float32, device=dev) if F > 0 else None
    with _on_device(dev):
        code = load().gnerf_make_rays_and_draws(_ptr(c2w), _ptr(k), n, int(resolution), _ptr(o), _ptr(d),
                                                _ptr(nc), nc.numel(), plan.offset_coarse, plan.threads_coarse,
                                                _ptr(nf), 0 if nf is None else nf.numel(), plan.offset_fine, plan.threads_fine, plan.seed, _stream(c2w))
    _check(code, 'gnerf_make_rays_and_draws')
    commit_philox_plan(plan)
    return o, d, nc, nf


_workspaces = {}
_EMPTY = torch.empty([0])          # "absent tensor" for the C++ binding, as the reference's _null_tensor (bias_act.py:38)


def _workspace(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        ws = torch.zeros([max(int(load().gnerf_render_workspace_bytes()), 16)], dtype=torch.uint8, device=device)     # zeroed once; calls leave it zeroed
        _workspaces[key] = ws
    return ws


def release_workspaces():
    """Kept for callers of earlier versions: render_backward's staging buffer is allocated per call now (torch's caching
    allocator owns it), so there is nothing cached to drop.  The small per-stream render workspaces stay."""


# ---------------------------------------------------------------------------- surroundings of the modulated convolution


@profiled('gnerf_hip::modulate_weights')
def modulate_weights(weight, styles, demodulate=True, out_dtype=torch.float32, want_weights=True, want_dcoefs=False, transposed=False,
                     channels_last=False):
    """Per-sample modulated (+ demodulated) convolution weights in one launch (networks_stylegan2.py:61-75), with the fp16
    pre-normalisation of :62-64 when out_dtype is float16 and demodulate.  weight [O,I,k,k], styles [N,I] float32.
    Returns (w, dcoefs): w [N,O,I,k,k] in out_dtype (or None), or -- transposed -- [N,I,O,k,k], the form conv_transpose2d takes;
    with channels_last the memory of every sample's 4-D weight is channels_last ([O,k,k,I] / [I,k,k,O]; the returned tensor is
    a strided view with the logical shape above).  dcoefs [N,O] float32 or None."""
    _require_cuda(weight, styles)
    w32, s32 = weight.detach().to(torch.float32).contiguous(), styles.detach().to(torch.float32).contiguous()
    o, i, kh, kw = w32.shape
    n = s32.shape[0]
    if s32.shape != (n, i) or out_dtype not in (torch.float32, torch.float16):
        raise RuntimeError('modulate_weights: styles must be [N, I] and out_dtype float32 or float16')
    out = view = None
    if want_weights:
        a, b = (i, o) if transposed else (o, i)
        if channels_last:
            out = torch.empty([n, a, kh, kw, b], dtype=out_dtype, device=w32.device)
            view = out.permute(0, 1, 4, 2, 3)
        else:
            out = view = torch.empty([n, a, b, kh, kw], dtype=out_dtype, device=w32.device)
    dco = torch.empty([n, o], dtype=torch.float32, device=w32.device) if (want_dcoefs and demodulate) else None
    prenorm = 1 if (out_dtype == torch.float16 and demodulate) else 0
    with _on_device(w32.device):
        code = load().gnerf_modulate_weights(_ptr(w32), _ptr(s32), _ptr(out), _DTYPE_CODE[out_dtype], _ptr(dco), n, o, i, kh * kw,
                                             1 if demodulate else 0, prenorm, (1 if transposed else 0) + (2 if channels_last else 0), _stream(w32))
    _check(code, 'gnerf_modulate_weights')
    return view, dco


@profiled('gnerf_hip::normalise_styles')
def normalise_styles(styles):
    """styles [N,I] / max|styles[n]| per row (networks_stylegan2.py:64)."""
    _require_cuda(styles)
    s32 = styles.detach().to(torch.float32).contiguous()
    out = torch.empty_like(s32)
    with _on_device(s32.device):
        code = load().gnerf_normalise_styles(_ptr(s32), _ptr(out), s32.shape[0], s32.shape[1], _stream(s32))
    _check(code, 'gnerf_normalise_styles')
    return out


def is_channels_last(x):
    """True for a 4-D tensor whose MEMORY is [N,H,W,C] with C > 1 (and not also NCHW-contiguous)."""
    return x.ndim == 4 and x.shape[1] > 1 and x.stride(1) == 1 and x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()


def _activation_layout(x, what):
    """'nchw' or 'nhwc' for a dense float16/float32 4-D activation tensor; anything else raises."""
    if x.ndim != 4 or x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'{what}: x must be a 4-D float16/float32 tensor')
    if x.is_contiguous():
        return 'nchw'
    if is_channels_last(x):
        return 'nhwc'
    raise RuntimeError(f'{what}: x must be contiguous (NCHW) or channels_last')


@profiled('gnerf_hip::scale_channels')
def scale_channels(x, scale):
    """x [N,C,H,W] (NCHW contiguous or channels_last, float16/32) * scale [N,C] float32, the product formed in x's dtype
    (networks_stylegan2.py:77).  The result has x's memory format."""
    _require_cuda(x, scale)
    layout = _activation_layout(x, 'scale_channels')
    n, c, h, w = x.shape
    s32 = scale.detach().to(torch.float32).contiguous()
    if s32.numel() != n * c:
        raise RuntimeError('scale_channels: scale must have N*C elements')
    y = torch.empty_like(x)
    with _on_device(x.device):
        if layout == 'nhwc':
            code = load().gnerf_scale_channels_nhwc(_ptr(x), _ptr(s32), _ptr(y), _DTYPE_CODE[x.dtype], n, h * w, c, _stream(x))
        else:
            code = load().gnerf_scale_channels(_ptr(x), _ptr(s32), _ptr(y), _DTYPE_CODE[x.dtype], n * c, h * w, _stream(x))
    _check(code, 'gnerf_scale_channels')
    return y


@profiled('gnerf_hip::modconv_epilogue')
def modconv_epilogue(x, bias=None, scale=None, noise=None, round_noise=False, act='lrelu', alpha=0.2, gain=1.0, clamp=None, next_scale=None):
    """Everything after the modulated convolution in one pass (networks_stylegan2.py:79-83 / :96-97 then :331-333):
    t = x * scale[n,c] + noise (rounded to x's dtype; skipped when both are None), y = clamp(act(t + bias[c]) * gain).
    x [N,C,H,W] NCHW contiguous or channels_last, float16/32; scale [N,C] float32; noise float32 [H,W] or [N,1,H,W]; bias [C] (any
    float dtype).  next_scale [N,C] (channels_last only): y is additionally multiplied by it in x's dtype -- the next layer's
    `x * styles` folded into this pass.  The result has x's memory format."""
    _require_cuda(x, bias, scale, noise, next_scale)
    layout = _activation_layout(x, 'modconv_epilogue')
    if act not in ('linear', 'lrelu'):
        raise RuntimeError('modconv_epilogue: act must be linear or lrelu')
    n, c, h, w = x.shape
    s32 = None if scale is None else scale.detach().to(torch.float32).contiguous()
    nz = None if noise is None else noise.detach().to(torch.float32).contiguous()
    per_item = 0
    if nz is not None:
        if nz.numel() == n * h * w and n > 1:
            per_item = 1
        elif nz.numel() != h * w:
            raise RuntimeError('modconv_epilogue: noise must have H*W or N*H*W elements')
    b = None if bias is None else bias.detach().to(x.dtype).contiguous()
    if (s32 is not None and s32.numel() != n * c) or (b is not None and b.numel() != c):
        raise RuntimeError('modconv_epilogue: scale must have N*C and bias C elements')
    nx = None if next_scale is None else next_scale.detach().to(torch.float32).contiguous()
    if nx is not None and (layout != 'nhwc' or nx.numel() != n * c):
        raise RuntimeError('modconv_epilogue: next_scale needs a channels_last x and N*C elements')
    y = torch.empty_like(x)
    with _on_device(x.device):
        if layout == 'nhwc':
            code = load().gnerf_modconv_epilogue_nhwc(_ptr(x), _ptr(y), _DTYPE_CODE[x.dtype], n, h * w, c, _ptr(s32), _ptr(nz), per_item,
                                                      1 if round_noise else 0, _ptr(b), 3 if act == 'lrelu' else 1, float(alpha), float(gain),
                                                      float(-1 if clamp is None else clamp), _ptr(nx), _stream(x))
        else:
            code = load().gnerf_modconv_epilogue(_ptr(x), _ptr(y), _DTYPE_CODE[x.dtype], n * c, h * w, c, _ptr(s32), _ptr(nz), per_item,
                                                 1 if round_noise else 0, _ptr(b), 3 if act == 'lrelu' else 1, float(alpha), float(gain),
                                                 float(-1 if clamp is None else clamp), _stream(x))
    _check(code, 'gnerf_modconv_epilogue')
    return y


def clock_under_load(run, microseconds=3000.0, device=None):
    """MHz the shader clock holds while `run()` (which enqueues work on the current stream for at least `microseconds`) executes: a
    one-wave sampler on a side stream (gnerf_clock_sample) reads the shader-cycle counter against the 100 MHz reference meanwhile."""
    dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
    out = torch.zeros(2, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    run()                                                   # the load is already running when the sampler starts (it does not wait for it)
    with _on_device(dev):
        _check(load().gnerf_clock_sample(out.data_ptr(), float(microseconds), ctypes.c_void_p(side.cuda_stream)), 'gnerf_clock_sample')
    run()
    torch.cuda.synchronize(dev)
    cyc, ticks = [int(v) for v in out.tolist()]
    return 100.0 * cyc / ticks if ticks else None


def _pad_input_channels(w9):
    """[9, O, I] -> [9, O, I rounded up to a multiple of 64] with zeros (the kernels read the weights in 64-channel chunks; the activations'
    missing channels are masked to zero by the kernel)."""
    i = w9.shape[2]
    pad = -i % 64
    return (torch.nn.functional.pad(w9, (0, pad)) if pad else w9).contiguous()


def pack_conv3x3_weights(weight, dtype=torch.float16):
    """[O, I, 3, 3] -> the tap-major [9, O, I] form gnerf_conv3x3_epilogue_nhwc reads (w_packed[ky * 3 + kx, o, c] = weight[o, c, ky, kx])."""
    o, i = weight.shape[:2]
    return _pad_input_channels(weight.detach().to(dtype).permute(2, 3, 0, 1).reshape(9, o, i))


def conv3x3_epilogue_supported(x, c_out):
    """Does the fused convolution + epilogue kernel take this activation tensor?  (float16, channels_last, 8 x 32 pixel tiles, input channels
    in multiples of 8 -- the last 64-channel chunk is zero-padded --, output channels in blocks of 128.)"""
    return (x.is_cuda and x.dtype == torch.float16 and x.ndim == 4 and is_channels_last(x) and x.shape[2] % 8 == 0 and x.shape[3] % 32 == 0
            and x.shape[1] % 8 == 0 and c_out % 128 == 0 and x.shape[1] * x.shape[2] * x.shape[3] * 2 < (1 << 31))


@profiled('gnerf_hip::conv3x3_epilogue')
def conv3x3_epilogue(x, w_packed, bias=None, scale=None, noise=None, round_noise=False, alpha=0.2, gain=1.0, clamp=None, next_scale=None):
    """conv2d(x, w, padding=1) followed by modconv_epilogue(act='lrelu') in ONE launch (csrc/conv3x3.hip): x [N,C,H,W] float16
    channels_last, w_packed = pack_conv3x3_weights(w) [9,O,C padded to a multiple of 64] float16; scale / next_scale [N,O] float32, noise float32 [H,W], bias [O].
    Returns a channels_last [N,O,H,W] float16 tensor.  Shapes outside conv3x3_epilogue_supported raise (GNERF_E_UNSUPPORTED)."""
    _require_cuda(x, w_packed, bias, scale, noise, next_scale)
    n, c, h, w = x.shape
    o = w_packed.shape[1]
    if not is_channels_last(x) or x.dtype != torch.float16 or tuple(w_packed.shape) != (9, o, -(-c // 64) * 64) or w_packed.dtype != torch.float16 or not w_packed.is_contiguous():
        raise RuntimeError('conv3x3_epilogue: x must be channels_last float16 [N,C,H,W] and w_packed contiguous float16 [9,O,C]')
    def f32(t, numel, what):
        if t is None:
            return None
        t = t.detach().to(torch.float32).contiguous()
        if t.numel() != numel:
            raise RuntimeError(f'conv3x3_epilogue: {what} must have {numel} elements')
        return t if t.data_ptr() % 16 == 0 else t.clone()
    s32, nx, nz = f32(scale, n * o, 'scale'), f32(next_scale, n * o, 'next_scale'), f32(noise, h * w, 'noise')
    b = None if bias is None else bias.detach().to(torch.float16).contiguous()
    if b is not None and b.numel() != o:
        raise RuntimeError('conv3x3_epilogue: bias must have O elements')
    y = torch.empty([n, o, h, w], dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    with _on_device(x.device):
        code = load().gnerf_conv3x3_epilogue_nhwc(_ptr(x), _ptr(w_packed), _ptr(y), n, h, w, c, o, _ptr(s32), _ptr(nz), 1 if round_noise else 0, _ptr(b),
                                                  float(alpha), float(gain), float(-1 if clamp is None else clamp), _ptr(nx), _stream(x))
    _check(code, 'gnerf_conv3x3_epilogue_nhwc')
    return y


def torgb_weights(weight, styles):
    """float16 [N, 3, C] = half(weight[o, c] * styles[n, c]): the 1 x 1 weights gnerf_torgb_nhwc forms per workgroup, as conv3x3_epilogue_torgb takes
    them (a constant of (latent, weight): cache it).  weight [3, C(, 1, 1)] float32, styles [N, C] float32 (ToRGB's affine output x its weight gain)."""
    w = weight.detach().to(torch.float32).reshape(1, 3, -1)
    return (w * styles.detach().to(torch.float32)[:, None, :]).to(torch.float16).contiguous()


def conv3x3_epilogue_torgb_supported(x, c_out):
    return conv3x3_epilogue_supported(x, c_out) and c_out == 128


@profiled('gnerf_hip::conv3x3_epilogue_torgb')
def conv3x3_epilogue_torgb(x, w_packed, img, rgb_w, rgb_bias=None, rgb_clamp=None, bias=None, scale=None, noise=None, round_noise=False, alpha=0.2, gain=1.0, clamp=None):
    """img += ToRGB(conv3x3_epilogue(x, ...)) in ONE launch that stores no layer output (csrc/conv3x3.hip, ABI 11): the last layer of a block whose x
    nothing else reads, with the block's ToRGB (rgb_w = torgb_weights(weight, styles) float16 [N,3,128], rgb_bias [3], rgb_clamp) in its epilogue and
    the result added to the running image img float32 [N,3,H,W] (dense NCHW) in place -- conv3x3_epilogue followed by torgb_channels_last(...,
    accumulate_into=img) with the same roundings.  x [N,C,H,W] float16 channels_last, w_packed [9,128,C padded] float16, scale [N,128] float32 (required)."""
    _require_cuda(x, w_packed, img, rgb_w, rgb_bias, bias, scale, noise)
    n, c, h, w = x.shape
    o = w_packed.shape[1]
    if not is_channels_last(x) or x.dtype != torch.float16 or tuple(w_packed.shape) != (9, 128, -(-c // 64) * 64) or w_packed.dtype != torch.float16 or not w_packed.is_contiguous():
        raise RuntimeError('conv3x3_epilogue_torgb: x must be channels_last float16 [N,C,H,W] and w_packed contiguous float16 [9,128,C]')
    if img.dtype != torch.float32 or tuple(img.shape) != (n, 3, h, w) or not img.is_contiguous():
        raise RuntimeError('conv3x3_epilogue_torgb: img must be a dense float32 [N,3,H,W] tensor')
    if rgb_w.dtype != torch.float16 or tuple(rgb_w.shape) != (n, 3, 128) or not rgb_w.is_contiguous() or scale is None:
        raise RuntimeError('conv3x3_epilogue_torgb: rgb_w must be float16 [N,3,128] (torgb_weights) and a demodulation scale is required')
    s32 = scale.detach().to(torch.float32).contiguous()
    nz = None if noise is None else noise.detach().to(torch.float32).contiguous()
    if s32.numel() != n * o or (nz is not None and nz.numel() != h * w):
        raise RuntimeError('conv3x3_epilogue_torgb: scale must have N x 128 elements, noise H x W')
    s32 = s32 if s32.data_ptr() % 16 == 0 else s32.clone()
    b = None if bias is None else bias.detach().to(torch.float16).contiguous()
    rb = None if rgb_bias is None else rgb_bias.detach().to(torch.float16).to(torch.float32).contiguous()       # (the stand-alone layer adds its float16 bias)
    if (b is not None and b.numel() != o) or (rb is not None and rb.numel() != 3):
        raise RuntimeError('conv3x3_epilogue_torgb: bias must have 128 elements, rgb_bias 3')
    with _on_device(x.device):
        code = load().gnerf_conv3x3_epilogue_torgb_nhwc(_ptr(x), _ptr(w_packed), n, h, w, c, _ptr(s32), _ptr(nz), 1 if round_noise else 0, _ptr(b),
                                                        float(alpha), float(gain), float(-1 if clamp is None else clamp),
                                                        _ptr(rgb_w), _ptr(rb), float(-1 if rgb_clamp is None else rgb_clamp), _ptr(img), _stream(x))
    _check(code, 'gnerf_conv3x3_epilogue_torgb_nhwc')
    return img


def _split_weights_f16x3(weight):
    """[O, I, kh, kw] float32 -> [O, 3 I, kh, kw] float32 holding [hi | hi | lo] along the input channels, hi = half(w), lo = half(w - hi): the
    weight operand of the fp32-grade convolution (gnerf_conv3x3_f32x3_epilogue_nhwc), against activations split as [hi | lo | hi]."""
    w = weight.detach().to(torch.float32)
    hi = w.to(torch.float16).to(torch.float32)
    lo = (w - hi).to(torch.float16).to(torch.float32)
    return torch.cat([hi, hi, lo], 1)


def pack_conv3x3_weights_f32x3(weight):
    """pack_conv3x3_weights of the [hi | hi | lo] split of a float32 weight [O, I, 3, 3]: float16 [9, O, 3 I padded to a multiple of 64]."""
    return pack_conv3x3_weights(_split_weights_f16x3(weight))


def pack_conv_transpose3x3_weights_f32x3(weight):
    """pack_conv_transpose3x3_weights of the [hi | hi | lo] split of a float32 weight [O, I, 3, 3] (correlation form)."""
    return pack_conv_transpose3x3_weights(_split_weights_f16x3(weight))


_split_overflow = {}


def split_overflow_flag(device):
    """The sticky device int split_f16x3 reports out-of-range activations in (one per device; read it with .item() when a check is wanted)."""
    f = _split_overflow.get(device)
    if f is None:
        f = _split_overflow[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return f


@profiled('gnerf_hip::split_f16x3')
def split_f16x3(x, scale=None):
    """x float32 channels_last [N, C, H, W] (C % 8 == 0), scale float32 [N, C] or None -> float16 channels_last [N, 3C, H, W] = [hi | lo | hi] of
    x * scale (csrc/conv3x3.hip): the activation operand of conv3x3_f32x3_epilogue / conv_transpose3x3_s2_f32x3."""
    _require_cuda(x, scale)
    n, c, h, w = x.shape
    if x.dtype != torch.float32 or not is_channels_last(x) or c % 8:
        raise RuntimeError('split_f16x3: x must be a channels_last float32 [N,C,H,W] tensor with C % 8 == 0')
    s32 = None if scale is None else scale.detach().to(torch.float32).contiguous()
    if s32 is not None and s32.numel() != n * c:
        raise RuntimeError('split_f16x3: scale must have N * C elements')
    y = torch.empty([n, 3 * c, h, w], dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    with _on_device(x.device):
        code = load().gnerf_split_f16x3_nhwc(_ptr(x), _ptr(s32), _ptr(y), n, h * w, c, _ptr(split_overflow_flag(x.device)), _stream(x))
    _check(code, 'gnerf_split_f16x3_nhwc')
    return y


def conv3x3_f32x3_supported(x, c_out):
    """Does the fp32-grade convolution take this float32 activation tensor (as its [hi | lo | hi] split)?  channels_last is not required of x:
    the caller converts; 8 x 32 pixel tiles, input channels in eights, output channels in blocks of 128."""
    return (x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and x.shape[2] % 8 == 0 and x.shape[3] % 32 == 0 and x.shape[1] % 8 == 0
            and c_out % 128 == 0 and 3 * x.shape[1] * x.shape[2] * x.shape[3] * 2 < (1 << 31))


def conv_transpose3x3_s2_f32x3_supported(x, c_out):
    return (x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and x.shape[1] % 8 == 0 and c_out % 128 == 0
            and 3 * x.shape[1] * x.shape[2] * x.shape[3] * 2 < (1 << 31))


@profiled('gnerf_hip::conv3x3_f32x3_epilogue')
def conv3x3_f32x3_epilogue(x3, w3_packed, bias=None, scale=None, noise=None, alpha=0.2, gain=1.0, clamp=None, next_scale=None):
    """The fp32-grade form of conv3x3_epilogue: x3 = split_f16x3(x) [N,3C,H,W] float16 channels_last, w3_packed = pack_conv3x3_weights_f32x3(w);
    bias float32 [O]; returns a channels_last float32 [N,O,H,W] tensor (nothing is rounded on the way out)."""
    _require_cuda(x3, w3_packed, bias, scale, noise, next_scale)
    n, c3, h, w = x3.shape
    o = w3_packed.shape[1]
    if not is_channels_last(x3) or x3.dtype != torch.float16 or tuple(w3_packed.shape) != (9, o, -(-c3 // 64) * 64) or w3_packed.dtype != torch.float16 or not w3_packed.is_contiguous():
        raise RuntimeError('conv3x3_f32x3_epilogue: x3 must be channels_last float16 [N,3C,H,W] and w3_packed contiguous float16 [9,O,3C]')
    def f32(t, numel, what):
        if t is None:
            return None
        t = t.detach().to(torch.float32).contiguous()
        if t.numel() != numel:
            raise RuntimeError(f'conv3x3_f32x3_epilogue: {what} must have {numel} elements')
        return t if t.data_ptr() % 16 == 0 else t.clone()
    s32, nx, nz, b = f32(scale, n * o, 'scale'), f32(next_scale, n * o, 'next_scale'), f32(noise, h * w, 'noise'), f32(bias, o, 'bias')
    y = torch.empty([n, o, h, w], dtype=torch.float32, device=x3.device, memory_format=torch.channels_last)
    with _on_device(x3.device):
        code = load().gnerf_conv3x3_f32x3_epilogue_nhwc(_ptr(x3), _ptr(w3_packed), _ptr(y), n, h, w, c3, o, _ptr(s32), _ptr(nz), _ptr(b),
                                                        float(alpha), float(gain), float(-1 if clamp is None else clamp), _ptr(nx), _stream(x3))
    _check(code, 'gnerf_conv3x3_f32x3_epilogue_nhwc')
    return y


@profiled('gnerf_hip::conv_transpose3x3_s2_f32x3')
def conv_transpose3x3_s2_f32x3(x3, w3_phases):
    """The fp32-grade form of conv_transpose3x3_s2: x3 = split_f16x3(x), w3_phases = pack_conv_transpose3x3_weights_f32x3(w); returns a
    channels_last float32 [N,O,2H+1,2W+1] tensor."""
    _require_cuda(x3, w3_phases)
    n, c3, h, w = x3.shape
    o = w3_phases.shape[1]
    if not is_channels_last(x3) or x3.dtype != torch.float16 or tuple(w3_phases.shape) != (9, o, -(-c3 // 64) * 64) or w3_phases.dtype != torch.float16 or not w3_phases.is_contiguous():
        raise RuntimeError('conv_transpose3x3_s2_f32x3: x3 must be channels_last float16 [N,3C,H,W] and w3_phases contiguous float16 [9,O,3C]')
    y = torch.empty([n, o, 2 * h + 1, 2 * w + 1], dtype=torch.float32, device=x3.device, memory_format=torch.channels_last)
    with _on_device(x3.device):
        code = load().gnerf_conv_transpose3x3_s2_f32x3_nhwc(_ptr(x3), _ptr(w3_phases), _ptr(y), n, h, w, c3, o, _stream(x3))
    _check(code, 'gnerf_conv_transpose3x3_s2_f32x3_nhwc')
    return y


def pack_conv_transpose3x3_weights(weight, dtype=torch.float16):
    """[O, I, 3, 3] (the correlation-form weight of a x2 layer: conv_transpose2d(x, weight.transpose(0, 1), stride=2)) -> the [9, O, I] form
    gnerf_conv_transpose3x3_s2_nhwc reads: the taps grouped by OUTPUT PHASE (py, px) = (oy & 1, ox & 1) -- phase (0,0): (ky, kx) = (0,0),
    (0,2), (2,0), (2,2); phase (0,1): (0,1), (2,1); phase (1,0): (1,0), (1,2); phase (1,1): (1,1)."""
    order = [(0, 0), (0, 2), (2, 0), (2, 2), (0, 1), (2, 1), (1, 0), (1, 2), (1, 1)]
    w = weight.detach().to(dtype)
    return _pad_input_channels(torch.stack([w[:, :, ky, kx] for ky, kx in order]))


def conv_transpose3x3_s2_supported(x, c_out):
    """Does the phase-decomposed transposed convolution take this activation tensor?  (float16, channels_last, input channels in
    multiples of 8, output channels in blocks of 128; any height and width.)"""
    return (x.is_cuda and x.dtype == torch.float16 and x.ndim == 4 and is_channels_last(x) and x.shape[1] % 8 == 0 and c_out % 128 == 0
            and x.shape[1] * x.shape[2] * x.shape[3] * 2 < (1 << 31))


@profiled('gnerf_hip::conv_transpose3x3_s2')
def conv_transpose3x3_s2(x, w_phases):
    """conv_transpose2d(x, w.transpose(0, 1), stride=2) for a 3x3 kernel (csrc/conv3x3.hip, MODE 1): x [N,C,H,W] float16 channels_last,
    w_phases = pack_conv_transpose3x3_weights(w) [9,O,C padded to a multiple of 64] float16.  Returns a channels_last [N,O,2H+1,2W+1] float16 tensor."""
    _require_cuda(x, w_phases)
    n, c, h, w = x.shape
    o = w_phases.shape[1]
    if not is_channels_last(x) or x.dtype != torch.float16 or tuple(w_phases.shape) != (9, o, -(-c // 64) * 64) or w_phases.dtype != torch.float16 or not w_phases.is_contiguous():
        raise RuntimeError('conv_transpose3x3_s2: x must be channels_last float16 [N,C,H,W] and w_phases contiguous float16 [9,O,C]')
    y = torch.empty([n, o, 2 * h + 1, 2 * w + 1], dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    with _on_device(x.device):
        code = load().gnerf_conv_transpose3x3_s2_nhwc(_ptr(x), _ptr(w_phases), _ptr(y), n, h, w, c, o, _stream(x))
    _check(code, 'gnerf_conv_transpose3x3_s2_nhwc')
    return y


@profiled('gnerf_hip::blur_epilogue_channels_last')
def blur_epilogue_channels_last(x, f, padding, blur_gain=1.0, bias=None, scale=None, act='lrelu', alpha=0.2, gain=1.0, clamp=None, next_scale=None,
                                flip_filter=False):
    """upfirdn2d(x, f, padding=padding, gain=blur_gain) with a 4x4 filter, then modconv_epilogue (no noise), in one pass over a
    channels_last x [N,C,H,W] (float16 / float32, C filling 16-byte vectors).  padding = [x0, x1, y0, y1].  Bit-identical to the
    two calls.  Returns a channels_last tensor."""
    _require_cuda(x, f, bias, scale, next_scale)
    if not is_channels_last(x) or x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError('blur_epilogue_channels_last: x must be a channels_last float16/float32 tensor')
    if act not in ('linear', 'lrelu'):
        raise RuntimeError('blur_epilogue_channels_last: act must be linear or lrelu')
    n, c, h, w = x.shape
    f32 = f.detach().to(torch.float32).contiguous()
    if f32.shape != (4, 4) or c % (16 // x.element_size()) != 0:
        raise RuntimeError('blur_epilogue_channels_last: a 4x4 filter and whole 16-byte channel vectors are required')
    px0, px1, py0, py1 = [int(v) for v in padding]
    oh, ow = h + py0 + py1 - 3, w + px0 + px1 - 3
    if oh < 1 or ow < 1:
        raise RuntimeError('blur_epilogue_channels_last: output must be at least 1x1')
    def vec(t, dtype):          # dense, and 16-byte aligned (the kernel fetches the per-channel operands as vectors): a view at an odd offset is copied
        if t is None:
            return None
        t = t.detach().to(dtype).contiguous()
        return t if t.data_ptr() % 16 == 0 else t.clone()
    s32, nx, b = vec(scale, torch.float32), vec(next_scale, torch.float32), vec(bias, x.dtype)
    if (s32 is not None and s32.numel() != n * c) or (nx is not None and nx.numel() != n * c) or (b is not None and b.numel() != c):
        raise RuntimeError('blur_epilogue_channels_last: scale / next_scale must have N*C and bias C elements')
    y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    with _on_device(x.device):
        code = load().gnerf_blur4_epilogue_nhwc(_ptr(x), _ptr(f32), _ptr(y), _DTYPE_CODE[x.dtype], n, c, h, w, oh, ow, px0, py0, 1 if flip_filter else 0,
                                                float(blur_gain), _ptr(s32), _ptr(b), 3 if act == 'lrelu' else 1, float(alpha), float(gain),
                                                float(-1 if clamp is None else clamp), _ptr(nx), _stream(x))
    _check(code, 'gnerf_blur4_epilogue_nhwc')
    return y


TORGB_CHANNELS = (32, 64, 128, 256, 512)


@profiled('gnerf_hip::torgb_channels_last')
def torgb_channels_last(x, weight, styles, bias=None, clamp=None, accumulate_into=None):
    """ToRGBLayer to three channels on a channels_last float16 x [N,C,H,W] (networks_stylegan2.py:349-367): weight [3,C,1,1] or [3,C]
    float32, styles [N,C] float32 (weight_gain applied), bias [3].  Returns float16 [N,3,H,W], NCHW.  See include/gnerf_hip.h.
    accumulate_into: a contiguous float32 [N,3,H,W] image; the layer's output (rounded to float16) is added to it IN PLACE and the
    image is returned -- the block's `img.add_(y.to(torch.float32))` in the same launch."""
    _require_cuda(x, weight, styles, bias)
    if x.dtype != torch.float16 or not is_channels_last(x) or x.shape[1] not in TORGB_CHANNELS:
        raise RuntimeError('torgb_channels_last: x must be a channels_last float16 tensor with 32, 64, 128, 256 or 512 channels')
    n, c, h, w = x.shape
    w32 = weight.detach().to(torch.float32).reshape(-1).contiguous()
    s32 = styles.detach().to(torch.float32).contiguous()
    if w32.numel() != 3 * c or s32.numel() != n * c:
        raise RuntimeError('torgb_channels_last: weight must be [3,C] and styles [N,C]')
    b = None if bias is None else bias.detach().to(torch.float16).contiguous()
    if accumulate_into is not None:
        img = accumulate_into
        _require_cuda(img)
        if img.dtype != torch.float32 or tuple(img.shape) != (n, 3, h, w) or not img.is_contiguous():
            raise RuntimeError('torgb_channels_last: accumulate_into must be a contiguous float32 [N,3,H,W] tensor')
        with _on_device(x.device):
            code = load().gnerf_torgb_nhwc_accumulate(_ptr(x), _ptr(w32), _ptr(s32), _ptr(b), _ptr(img), n, h * w, c,
                                                      float(-1 if clamp is None else clamp), _stream(x))
        _check(code, 'gnerf_torgb_nhwc_accumulate')
        return img
    y = torch.empty([n, 3, h, w], dtype=torch.float16, device=x.device)
    with _on_device(x.device):
        code = load().gnerf_torgb_nhwc(_ptr(x), _ptr(w32), _ptr(s32), _ptr(b), _ptr(y), n, h * w, c, float(-1 if clamp is None else clamp), _stream(x))
    _check(code, 'gnerf_torgb_nhwc')
    return y


def planes_layout(planes_nhwc, n_items, what):
    """0 for [3N,H,W,32] (one NHWC image per plane), 1 for [N,H,W,96] (planes interleaved per texel: channels_last memory of the
    backbone's [N,96,H,W] output).  Anything else raises."""
    if planes_nhwc.dtype != torch.float32 or not planes_nhwc.is_contiguous() or planes_nhwc.ndim != 4:
        raise RuntimeError(f'{what}: planes must be a contiguous float32 4-D tensor')
    if planes_nhwc.shape[3] == 32 and planes_nhwc.shape[0] == 3 * n_items:
        return 0
    if planes_nhwc.shape[3] == 96 and planes_nhwc.shape[0] == n_items:
        return 1
    raise RuntimeError(f'{what}: planes_nhwc must be [3N,H,W,32] or [N,H,W,96] (3 planes of 32 channels per item)')


@profiled('gnerf_hip::upsample2x_add_nhwc')
def upsample2x_add_nhwc(img, y, f, flip=False, gain=4.0, with_absmax=False):
    """upfirdn2d(img, f, up=2, padding=[2,1,2,1], gain) + y written channels_last in one launch (the tri-plane producer's last
    step, networks_stylegan2.py:456-463).  img [N,C,h,w], y [N,C,2h,2w] or None, both float32 NCHW-contiguous; f the 4x4 filter.
    Returns a [N,C,2h,2w] tensor with channels_last strides (its memory is [N,2h,2w,C]) and, with_absmax, max |out| [1].
    Returns None when the kernel does not cover the shape (C % 32, w % 16, h % 2) -- the caller composes the ops instead."""
    _require_cuda(img, y)
    if img.dtype != torch.float32 or img.ndim != 4 or not img.is_contiguous() or tuple(f.shape) != (4, 4):
        return None
    n, c, h, w = img.shape
    if c % 32 or w % 16 or h % 2:
        return None
    if y is not None and (y.dtype != torch.float32 or tuple(y.shape) != (n, c, 2 * h, 2 * w) or not y.is_contiguous()):
        return None
    taps = _filter_taps(f)
    out = torch.empty([n, c, 2 * h, 2 * w], dtype=torch.float32, device=img.device, memory_format=torch.channels_last)
    amax = torch.empty([1], dtype=torch.float32, device=img.device) if with_absmax else None
    with _on_device(img.device):
        code = load().gnerf_upsample2x_add_nhwc(_ptr(img), _ptr(y), taps, 1 if flip else 0, float(gain), _ptr(out), n, c, h, w, _ptr(amax), _stream(img))
    _check(code, 'gnerf_upsample2x_add_nhwc')
    return (out, amax) if with_absmax else out


_filter_tap_cache = {}


def _filter_taps(f):
    """The 16 taps of a 4x4 filter as a host float array (one device read per filter tensor and version, then cached)."""
    key = (f.data_ptr(), f._version if not f.is_inference() else None, f.device)
    taps = _filter_tap_cache.get(key)
    if taps is None:
        if len(_filter_tap_cache) > 64:
            _filter_tap_cache.clear()
        taps = (ctypes.c_float * 16)(*f.detach().float().cpu().reshape(-1).tolist())
        _filter_tap_cache[key] = taps
    return taps


def last_mlp_choice(device):
    """Decoder arithmetic the last mlp='auto' render call on `device`'s current stream picked: 'f16x3' or 'f32' (None if no such
    call ran).  Reads the render workspace (synchronises); for tests and diagnostics."""
    ws = _workspaces.get((device.index, torch.cuda.current_stream(device).cuda_stream))
    if ws is None:
        return None
    return {1: 'f16x3', 2: 'f32'}.get(int(ws.view(torch.int32)[4].item()))


def _render_params(planes_nhwc, n_items, decoder, ray_origins, ray_dirs, noise_coarse, noise_fine,
                   depth_resolution, depth_resolution_importance, ray_start, ray_end, box_warp,
                   white_back, disparity_space_sampling, image_width, what, planes_absmax=None, mlp='auto',
                   planes_shared=False, depth_clamp_per_item=False, cameras=None, rng=None):
    """Validate the arguments shared by render_forward / render_backward and fill a RenderParams.
    Returns (params, keepalive, rays_per_item); `keepalive` holds the converted tensors the pointers refer to.
    cameras = (cam2world [N,4,4], intrinsics [N,3,3], res) with ray_origins = ray_dirs = None: rays made in the kernel;
    rng = a TorchPhiloxPlan with noise_coarse = noise_fine = None: draws made in the kernel (gnerf_render_params, ABI 8)."""
    w1, b1, w2, b2 = decoder
    _require_cuda(planes_nhwc, ray_origins, ray_dirs, noise_coarse, noise_fine, w1, b1, w2, b2)
    dev = planes_nhwc.device
    if cameras is not None or rng is not None:
        return _render_params_generated(planes_nhwc, n_items, decoder, ray_origins, ray_dirs, noise_coarse, noise_fine,
                                        depth_resolution, depth_resolution_importance, ray_start, ray_end, box_warp, white_back,
                                        disparity_space_sampling, image_width, what, planes_absmax, mlp, planes_shared, depth_clamp_per_item,
                                        cameras, rng)

    def f32c(t):
        return t.to(torch.float32).contiguous()
    interleaved = planes_layout(planes_nhwc, 1 if planes_shared else n_items, what)
    if tuple(w1.shape) != (64, 32) or tuple(b1.shape) != (64,) or tuple(w2.shape) != (33, 64) or tuple(b2.shape) != (33,):
        raise RuntimeError(f'{what}: decoder must be the 32->64->33 OSGDecoder MLP')
    o, d = f32c(ray_origins), f32c(ray_dirs)
    if o.shape != d.shape or o.ndim != 3 or o.shape[0] != n_items or o.shape[2] != 3:
        raise RuntimeError(f'{what}: rays must be [N,M,3]')
    m = o.shape[1]
    S, F = int(depth_resolution), int(depth_resolution_importance)
    nc = f32c(noise_coarse)
    if nc.numel() != n_items * m * S:
        raise RuntimeError(f'{what}: noise_coarse must have N*M*S elements')
    nf = None
    if F > 0:
        if noise_fine is None:
            raise RuntimeError(f'{what}: noise_fine required when depth_resolution_importance > 0')
        nf = f32c(noise_fine)
        if nf.numel() != n_items * m * F:
            raise RuntimeError(f'{what}: noise_fine must have N*M*F elements')
    w1, b1, w2, b2 = f32c(w1), f32c(b1), f32c(w2), f32c(b2)
    rs_t = re_t = None
    if isinstance(ray_start, torch.Tensor) or isinstance(ray_end, torch.Tensor):
        rs_t = f32c(torch.as_tensor(ray_start, device=dev).expand(n_items, m, 1) if not isinstance(ray_start, torch.Tensor) else ray_start).reshape(-1)
        re_t = f32c(torch.as_tensor(ray_end, device=dev).expand(n_items, m, 1) if not isinstance(ray_end, torch.Tensor) else ray_end).reshape(-1)
        if rs_t.numel() != n_items * m or re_t.numel() != n_items * m:
            raise RuntimeError(f'{what}: per-ray ray_start / ray_end must have N*M elements')
        ray_start = ray_end = 0.0
    p = RenderParams()
    p.planes_nhwc = planes_nhwc.data_ptr(); p.n_items = n_items; p.plane_h = planes_nhwc.shape[1]; p.plane_w = planes_nhwc.shape[2]
    p.ray_origins = o.data_ptr(); p.ray_dirs = d.data_ptr(); p.rays_per_item = m; p.image_width = int(image_width)
    p.w1, p.b1, p.w2, p.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    p.depth_resolution = S; p.depth_resolution_importance = F
    p.ray_start = float(ray_start); p.ray_end = float(ray_end)
    p.ray_start_per_ray = None if rs_t is None else rs_t.data_ptr()
    p.ray_end_per_ray = None if re_t is None else re_t.data_ptr()
    p.box_warp = float(box_warp); p.white_back = int(bool(white_back)); p.disparity_space_sampling = int(bool(disparity_space_sampling))
    p.noise_coarse = nc.data_ptr(); p.noise_fine = None if nf is None else nf.data_ptr()
    if mlp not in MLP_MODES:
        raise RuntimeError(f"{what}: mlp must be one of {sorted(MLP_MODES)}")
    p.mlp_mode = MLP_MODES[mlp]
    if planes_absmax is not None:
        _require_cuda(planes_absmax)
        if planes_absmax.dtype != torch.float32 or planes_absmax.numel() != 1:
            raise RuntimeError(f'{what}: planes_absmax must be a one-element float32 device tensor')
    p.planes_absmax = _ptr(planes_absmax)
    p.planes_interleaved = interleaved
    p.planes_shared = int(bool(planes_shared)); p.depth_clamp_per_item = int(bool(depth_clamp_per_item))
    return p, (planes_nhwc, o, d, nc, nf, w1, b1, w2, b2, rs_t, re_t, planes_absmax), m


class TorchPhiloxPlan:
    """Where torch's device generator stands before the renderer's two uniform draws (renderer.py:190 rand_like([N,M,S,1]), :241
    rand(N*M, F)) and how ATen would have laid them out on this device -- what gnerf_render_params.rng_* carry (include/gnerf_hip.h,
    oracle/philox_ref.py).  per_item: N separate calls of one item each (the draws of the batched-views form)."""
    __slots__ = ('seed', 'offset_coarse', 'offset_fine', 'item_stride', 'threads_coarse', 'threads_fine', 'per_item', 'end_offset',
                 'numel_coarse', 'numel_fine', 'generator')


_device_geometry = {}


def torch_rand_geometry(numel, device):
    """(threads, philox offset increment) of `torch.rand(numel, device=device)` (gnerf_torch_rand_plan)."""
    geo = _device_geometry.get(device.index)
    if geo is None:
        pr = torch.cuda.get_device_properties(device)
        geo = _device_geometry[device.index] = (int(pr.multi_processor_count), int(pr.max_threads_per_multi_processor))
    thr, inc = ctypes.c_uint32(0), ctypes.c_uint64(0)
    _check(load().gnerf_torch_rand_plan(int(numel), geo[0], geo[1], ctypes.byref(thr), ctypes.byref(inc)), 'gnerf_torch_rand_plan')
    return thr.value, inc.value


def torch_philox_plan(device, n_items, rays_per_item, S, F, per_item=False, generator=None, advance=True):
    """Plan the renderer's two draws on `device`'s generator (default: torch's default generator of that device) and -- advance=True --
    move the generator past them, exactly as the torch.rand calls would have: a seeded run that renders with in-kernel draws leaves the
    generator where the reference's run leaves it.  Not usable while the stream is capturing a graph (graph-safe generators keep their
    offset on the device): the caller draws tensors then."""
    gen = generator if generator is not None else torch.cuda.default_generators[device.index]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('torch_philox_plan: the stream is capturing a graph; draw with torch.rand instead')
    plan = TorchPhiloxPlan()
    plan.seed = int(gen.initial_seed()) & 0xFFFFFFFFFFFFFFFF
    plan.per_item = bool(per_item)
    units = rays_per_item if per_item else n_items * rays_per_item
    plan.threads_coarse, inc_c = torch_rand_geometry(units * S, device)
    plan.threads_fine, inc_f = torch_rand_geometry(units * F, device) if F > 0 else (0, 0)
    plan.numel_coarse, plan.numel_fine = units * S, units * F
    plan.generator = gen
    off = int(gen.get_offset())
    plan.offset_coarse, plan.offset_fine = off, off + inc_c
    plan.item_stride = inc_c + inc_f if per_item else 0
    plan.end_offset = off + (inc_c + inc_f) * (n_items if per_item else 1)
    if advance:
        gen.set_offset(plan.end_offset)
    return plan


def commit_philox_plan(plan):
    """Move the plan's generator past its two draws (for plans made with advance=False: a caller that wants to know that the launch was
    accepted before the generator moves)."""
    plan.generator.set_offset(plan.end_offset)


@profiled('gnerf_hip::torch_rand')
def torch_rand(numel, device, seed, offset):
    """Element for element what torch.rand(numel, device=device) returns with the device generator at (seed, offset): the render
    kernels' in-kernel draw as a stand-alone kernel (gnerf_torch_rand).  The generator is not touched."""
    threads, _ = torch_rand_geometry(numel, device)
    out = torch.empty(int(numel), dtype=torch.float32, device=device)
    with _on_device(device):
        _check(load().gnerf_torch_rand(out.data_ptr(), int(numel), int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset), threads, _stream(out)), 'gnerf_torch_rand')
    return out


def _render_params_generated(planes_nhwc, n_items, decoder, ray_origins, ray_dirs, noise_coarse, noise_fine,
                             depth_resolution, depth_resolution_importance, ray_start, ray_end, box_warp, white_back,
                             disparity_space_sampling, image_width, what, planes_absmax, mlp, planes_shared, depth_clamp_per_item, cameras, rng):
    """_render_params for calls that make their rays and / or their draws in the kernel (ABI 8)."""
    w1, b1, w2, b2 = decoder
    dev = planes_nhwc.device

    def f32c(t):
        return t.to(torch.float32).contiguous()
    interleaved = planes_layout(planes_nhwc, 1 if planes_shared else n_items, what)
    if tuple(w1.shape) != (64, 32) or tuple(b1.shape) != (64,) or tuple(w2.shape) != (33, 64) or tuple(b2.shape) != (33,):
        raise RuntimeError(f'{what}: decoder must be the 32->64->33 OSGDecoder MLP')
    if isinstance(ray_start, torch.Tensor) or isinstance(ray_end, torch.Tensor):
        raise RuntimeError(f'{what}: in-kernel rays / draws take scalar ray limits')
    S, F = int(depth_resolution), int(depth_resolution_importance)
    p = RenderParams()
    keep = [planes_nhwc]
    if cameras is not None:
        if ray_origins is not None or ray_dirs is not None:
            raise RuntimeError(f'{what}: give rays or cameras, not both')
        c2w, intr, res = cameras
        _require_cuda(c2w, intr)
        c2w, intr = f32c(c2w), f32c(intr)
        if tuple(c2w.shape) != (n_items, 4, 4) or tuple(intr.shape) != (n_items, 3, 3):
            raise RuntimeError(f'{what}: cameras must be cam2world [N,4,4] and intrinsics [N,3,3]')
        m, image_width = int(res) * int(res), int(res)
        p.cam2world, p.intrinsics = c2w.data_ptr(), intr.data_ptr()
        keep += [c2w, intr]
    else:
        o, d = f32c(ray_origins), f32c(ray_dirs)
        if o.shape != d.shape or o.ndim != 3 or o.shape[0] != n_items or o.shape[2] != 3:
            raise RuntimeError(f'{what}: rays must be [N,M,3]')
        m = o.shape[1]
        p.ray_origins, p.ray_dirs = o.data_ptr(), d.data_ptr()
        keep += [o, d]
    if rng is not None:
        if noise_coarse is not None or noise_fine is not None:
            raise RuntimeError(f'{what}: give noise tensors or an rng plan, not both')
        p.rng_mode = 1
        p.rng_per_item = int(rng.per_item)
        p.rng_seed, p.rng_offset_coarse, p.rng_offset_fine = rng.seed, rng.offset_coarse, rng.offset_fine
        p.rng_offset_item_stride, p.rng_threads_coarse, p.rng_threads_fine = rng.item_stride, rng.threads_coarse, rng.threads_fine
    else:
        nc = f32c(noise_coarse)
        nf = f32c(noise_fine) if F > 0 else None
        if nc.numel() != n_items * m * S or (nf is not None and nf.numel() != n_items * m * F):
            raise RuntimeError(f'{what}: noise tensors must have N*M*S and N*M*F elements')
        p.noise_coarse, p.noise_fine = nc.data_ptr(), None if nf is None else nf.data_ptr()
        keep += [nc, nf]
    w1, b1, w2, b2 = f32c(w1), f32c(b1), f32c(w2), f32c(b2)
    keep += [w1, b1, w2, b2, planes_absmax]
    p.planes_nhwc = planes_nhwc.data_ptr(); p.n_items = n_items; p.plane_h = planes_nhwc.shape[1]; p.plane_w = planes_nhwc.shape[2]
    p.rays_per_item = m; p.image_width = int(image_width)
    p.w1, p.b1, p.w2, p.b2 = w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    p.depth_resolution = S; p.depth_resolution_importance = F
    p.ray_start = float(ray_start); p.ray_end = float(ray_end)
    p.box_warp = float(box_warp); p.white_back = int(bool(white_back)); p.disparity_space_sampling = int(bool(disparity_space_sampling))
    if mlp not in MLP_MODES:
        raise RuntimeError(f"{what}: mlp must be one of {sorted(MLP_MODES)}")
    p.mlp_mode = MLP_MODES[mlp]
    if planes_absmax is not None:
        _require_cuda(planes_absmax)
        if planes_absmax.dtype != torch.float32 or planes_absmax.numel() != 1:
            raise RuntimeError(f'{what}: planes_absmax must be a one-element float32 device tensor')
    p.planes_absmax = _ptr(planes_absmax)
    p.planes_interleaved = interleaved
    p.planes_shared = int(bool(planes_shared)); p.depth_clamp_per_item = int(bool(depth_clamp_per_item))
    return p, tuple(keep), m


def render_generated_supported(S, F, ray_start=0.0, ray_end=1.0, disparity_space_sampling=False, plan=None, numel_planes=None):
    """Do the render kernels make rays / draws themselves for these options?  (48+48 and 96+96 samples, plain stratified sampling; the
    pipelined kernel at its compile-time sample counts, which GNERF_RENDER_KERNEL / GNERF_PIPE_FULL=0 can take away; planes of one item
    below 4 GB.)  plan: a TorchPhiloxPlan whose generator geometry is checked too -- the kernel reproduces ATen's draw only when its
    thread count is a power of two or covers the draw (raygen.h: torch_rand_draw), which depends on the device's CU count."""
    if not (int(S) == int(F) and int(S) in (48, 96) and not disparity_space_sampling and not isinstance(ray_start, torch.Tensor) and not isinstance(ray_end, torch.Tensor)):
        return False
    if os.environ.get('GNERF_RENDER_KERNEL', 'pipe') != 'pipe' or os.environ.get('GNERF_PIPE_FULL', '1') == '0':
        return False
    if numel_planes is not None and int(numel_planes) * 4 >= (1 << 32):
        return False
    if plan is not None:
        for thr, numel in ((plan.threads_coarse, plan.numel_coarse), (plan.threads_fine, plan.numel_fine)):
            if numel and not (thr >= numel or (thr > 0 and thr & (thr - 1) == 0)):
                return False
        if plan.offset_coarse % 4 or plan.offset_fine % 4 or plan.item_stride % 4:
            return False
    return True


@profiled('gnerf_hip::render_forward')
def render_forward(planes_nhwc, n_items, decoder, ray_origins, ray_dirs, noise_coarse, noise_fine, *,
                   depth_resolution, depth_resolution_importance, ray_start, ray_end, box_warp,
                   white_back=False, disparity_space_sampling=False, image_width=0, debug=False, planes_absmax=None, mlp='auto',
                   planes_shared=False, depth_clamp_per_item=False, cameras=None, rng=None, sigma_noise=None):
    """planes_nhwc [3N,H,W,32]; decoder = (w1,b1,w2,b2) effective fp32 weights; rays [N,M,3];
    sigma_noise = (coarse [N*M,S], fine [N*M,F] or None): density noise ALREADY multiplied by density_noise, added to the two passes'
    densities before their ray marches (renderer.py:146-147); forward only, tensor rays and draws only.
    cameras = (cam2world [N,4,4], intrinsics [N,3,3], res) with ray_origins = ray_dirs = None: the kernel makes the rays gnerf_make_rays
    would (RaySampler.forward); rng = torch_philox_plan(...) with noise_coarse = noise_fine = None: the kernel makes the draws torch.rand
    would (both: render_generated_supported; bit-identical to the tensor forms, tests/test_gpu_parity.py).
    noise_coarse [N*M,S]; noise_fine [N*M,F] or None; ray_start/ray_end floats or [N*M] tensors.
    planes_shared: planes_nhwc holds ONE item's planes ([3,H,W,32] or [1,H,W,96]) that all N items of rays read (N views of one
    object in one launch).  depth_clamp_per_item: the final depth clamp (ray_marcher.py:49-50) takes its range from each item's
    own samples instead of the whole call's, so that item i's outputs equal those of a call with item i alone.
    mlp: decoder arithmetic, 'auto' (decided on the device from planes_absmax -- the one-element tensor planes_to_nhwc(...,
    with_absmax=True) returns; measured by the call itself when None -- and the decoder's weights), 'f16x3' or 'f32'.
    Returns (rgb [N,M,32], depth [N,M,1], wsum [N,M,1][, debug [N*M,8,S+F]])."""
    e = ext()
    if e is not None and not debug and planes_nhwc.dtype == torch.float32 and planes_nhwc.is_contiguous() and cameras is None and rng is None and sigma_noise is None:
        # the C++ binding: same validation and the same C ABI call, without ctypes marshalling
        def f32c(t):
            return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.to(torch.float32).contiguous()
        if mlp not in MLP_MODES:
            raise RuntimeError(f"render_forward: mlp must be one of {sorted(MLP_MODES)}")
        per_ray = isinstance(ray_start, torch.Tensor) or isinstance(ray_end, torch.Tensor)
        dev = planes_nhwc.device
        n, m_ = ray_origins.shape[0], ray_origins.shape[1]
        if per_ray:
            rs_t = f32c(ray_start if isinstance(ray_start, torch.Tensor) else torch.as_tensor(ray_start, device=dev).expand(n, m_, 1)).reshape(-1)
            re_t = f32c(ray_end if isinstance(ray_end, torch.Tensor) else torch.as_tensor(ray_end, device=dev).expand(n, m_, 1)).reshape(-1)
            rs, re = 0.0, 0.0
        else:
            rs_t = re_t = _EMPTY
            rs, re = float(ray_start), float(ray_end)
        w1, b1, w2, b2 = decoder
        with _on_device(dev):
            return e.render_forward(planes_nhwc, n_items, f32c(w1), f32c(b1), f32c(w2), f32c(b2), f32c(ray_origins), f32c(ray_dirs),
                                    f32c(noise_coarse), _EMPTY if noise_fine is None else f32c(noise_fine),
                                    int(depth_resolution), int(depth_resolution_importance), rs, re, rs_t, re_t, float(box_warp),
                                    bool(white_back), bool(disparity_space_sampling), int(image_width),
                                    _EMPTY if planes_absmax is None else planes_absmax, MLP_MODES[mlp], _workspace(dev),
                                    bool(planes_shared), bool(depth_clamp_per_item))
    p, keep, m = _render_params(planes_nhwc, n_items, decoder, ray_origins, ray_dirs, noise_coarse, noise_fine,
                                depth_resolution, depth_resolution_importance, ray_start, ray_end, box_warp,
                                white_back, disparity_space_sampling, image_width, 'render_forward', planes_absmax, mlp,
                                planes_shared, depth_clamp_per_item, cameras, rng)
    dev = planes_nhwc.device
    rgb = torch.empty([n_items, m, 32], dtype=torch.float32, device=dev)
    depth = torch.empty([n_items, m, 1], dtype=torch.float32, device=dev)
    wsum = torch.empty([n_items, m, 1], dtype=torch.float32, device=dev)
    dbg = torch.zeros([n_items * m, DEBUG_SLOTS, p.depth_resolution + p.depth_resolution_importance], dtype=torch.float32, device=dev) if debug else None
    ws = _workspace(dev)
    p.out_rgb, p.out_depth, p.out_wsum = rgb.data_ptr(), depth.data_ptr(), wsum.data_ptr()
    p.workspace = ws.data_ptr(); p.debug = None if dbg is None else dbg.data_ptr()
    if sigma_noise is not None:
        sc, sf = sigma_noise
        _require_cuda(sc, sf)
        sc = sc.detach().to(torch.float32).contiguous()
        sf = None if sf is None else sf.detach().to(torch.float32).contiguous()
        if sc.numel() != n_items * m * p.depth_resolution or (p.depth_resolution_importance > 0 and (sf is None or sf.numel() != n_items * m * p.depth_resolution_importance)):
            raise RuntimeError('render_forward: sigma_noise must be ([N*M,S], [N*M,F]) tensors')
        p.sigma_noise_coarse, p.sigma_noise_fine = sc.data_ptr(), None if sf is None else sf.data_ptr()
        keep = keep + (sc, sf)
    with _on_device(dev):
        code = load().gnerf_render_forward(ctypes.byref(p), _stream(planes_nhwc))
    _check(code, 'gnerf_render_forward')
    del keep
    if debug:
        return rgb, depth, wsum, dbg
    return rgb, depth, wsum


@profiled('gnerf_hip::render_backward')
def render_backward(planes_nhwc, n_items, decoder, ray_origins, ray_dirs, noise_coarse, noise_fine, grad_rgb, grad_depth, grad_wsum, *,
                    depth_resolution, depth_resolution_importance, ray_start, ray_end, box_warp,
                    white_back=False, disparity_space_sampling=False, image_width=0, need_planes=True, need_decoder=True,
                    staged_scatter=True, planes_absmax=None):
    """Gradient of render_forward for the same arguments (the forward pass is recomputed inside the kernel).
    grad_rgb [N,M,32], grad_depth [N,M,1], grad_wsum [N,M,1]; any of them may be None (zeros).
    staged_scatter: make the plane gradient in two passes through a staging buffer (per-texel aggregation in LDS before the
    atomics; see include/gnerf_hip.h) -- the default; False = the single-pass form.
    planes_absmax: max |planes| as for render_forward (the staged form's first pass picks its decoder arithmetic from it on the
    device; measured by the call when None).
    Returns (grad_planes_nhwc [3N,H,W,32] or None, (grad_w1, grad_b1, grad_w2, grad_b2) or None), all float32."""
    p, keep, m = _render_params(planes_nhwc, n_items, decoder, ray_origins, ray_dirs, noise_coarse, noise_fine,
                                depth_resolution, depth_resolution_importance, ray_start, ray_end, box_warp,
                                white_back, disparity_space_sampling, image_width, 'render_backward', planes_absmax)
    dev = planes_nhwc.device
    _require_cuda(grad_rgb, grad_depth, grad_wsum)
    grads_in = []
    for t, n in ((grad_rgb, 32), (grad_depth, 1), (grad_wsum, 1)):
        if t is not None:
            t = t.to(torch.float32).contiguous()
            if t.numel() != n_items * m * n:
                raise RuntimeError('render_backward: output gradients must match the forward outputs')
        grads_in.append(t)
    g = RenderGrads()
    g.grad_rgb, g.grad_depth, g.grad_wsum = [None if t is None else t.data_ptr() for t in grads_in]
    g_planes = torch.zeros_like(planes_nhwc) if need_planes else None
    g_dec = None
    if need_decoder:
        g_dec = (torch.zeros([64, 32], dtype=torch.float32, device=dev), torch.zeros([64], dtype=torch.float32, device=dev),
                 torch.zeros([33, 64], dtype=torch.float32, device=dev), torch.zeros([33], dtype=torch.float32, device=dev))
        g.grad_w1, g.grad_b1, g.grad_w2, g.grad_b2 = [t.data_ptr() for t in g_dec]
    g.grad_planes_nhwc = None if g_planes is None else g_planes.data_ptr()
    stage = None
    if g_planes is not None and staged_scatter:
        # Staging buffer of the two-pass scatter (gnerf_render_backward_stage_bytes: bounded, the passes run over batches of ray
        # tiles).  Allocated per call on the current stream: torch's caching allocator hands the block back to the rest of the
        # step afterwards (a buffer cached here would be invisible to it).  Out of memory -> the single-pass form, same result.
        nbytes = int(load().gnerf_render_backward_stage_bytes(ctypes.byref(p)))
        try:
            stage = torch.empty([nbytes], dtype=torch.uint8, device=dev)
            g.scatter_stage = stage.data_ptr()
        except torch.OutOfMemoryError:
            stage = None
    elif g_planes is None and g_dec is not None and staged_scatter:
        # decoder gradients only: the small per-sample exchange buffer of the pipelined path (1.5 KB per ray at 48+48)
        try:
            stage = torch.empty([int(load().gnerf_render_backward_exchange_bytes(ctypes.byref(p)))], dtype=torch.uint8, device=dev)
            g.scatter_stage = stage.data_ptr()
        except torch.OutOfMemoryError:
            stage = None
    with _on_device(dev):
        code = load().gnerf_render_backward(ctypes.byref(p), ctypes.byref(g), _stream(planes_nhwc))
    _check(code, 'gnerf_render_backward')
    del keep, grads_in, stage
    return g_planes, g_dec


@profiled('gnerf_hip::query_points')
def query_points(planes_nhwc, n_items, decoder, points, box_warp, want_rgb=True):
    """run_model for arbitrary points [N,P,3] -> sigma [N,P,1], rgb [N,P,32] (rgb None when want_rgb is False)."""
    w1, b1, w2, b2 = [t.to(torch.float32).contiguous() for t in decoder]
    _require_cuda(planes_nhwc, points, w1)
    pts = points.to(torch.float32).contiguous()
    if pts.ndim != 3 or pts.shape[0] != n_items or pts.shape[2] != 3:
        raise RuntimeError('query_points: points must be [N,P,3]')
    n_pts = pts.shape[1]
    interleaved = planes_layout(planes_nhwc, n_items, 'query_points')
    sigma = torch.empty([n_items, n_pts, 1], dtype=torch.float32, device=pts.device)
    rgb = torch.empty([n_items, n_pts, 32], dtype=torch.float32, device=pts.device) if want_rgb else None
    with _on_device(pts.device):
        code = load().gnerf_query_points(_ptr(planes_nhwc), n_items, planes_nhwc.shape[1], planes_nhwc.shape[2], _ptr(pts), n_pts,
                                         float(box_warp), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(sigma), _ptr(rgb), interleaved, _stream(pts))
    _check(code, 'gnerf_query_points')
    return sigma, rgb


@profiled('gnerf_hip::query_points_backward')
def query_points_backward(planes_nhwc, n_items, decoder, points, box_warp, grad_sigma, grad_rgb, need_planes=True, need_decoder=True):
    """Gradient of query_points for the same arguments: grad_sigma [N,P,1] / grad_rgb [N,P,32] (either may be None).
    Returns (grad_planes_nhwc or None, (grad_w1, grad_b1, grad_w2, grad_b2) or None), float32."""
    w1, b1, w2, b2 = [t.to(torch.float32).contiguous() for t in decoder]
    _require_cuda(planes_nhwc, points, w1, grad_sigma, grad_rgb)
    interleaved = planes_layout(planes_nhwc, n_items, 'query_points_backward')
    pts = points.to(torch.float32).contiguous()
    if pts.ndim != 3 or pts.shape[0] != n_items or pts.shape[2] != 3:
        raise RuntimeError('query_points_backward: points must be [N,P,3]')
    n_pts = pts.shape[1]
    gs = None if grad_sigma is None else grad_sigma.to(torch.float32).contiguous()
    gc = None if grad_rgb is None else grad_rgb.to(torch.float32).contiguous()
    if (gs is not None and gs.numel() != n_items * n_pts) or (gc is not None and gc.numel() != n_items * n_pts * 32):
        raise RuntimeError('query_points_backward: output gradients must match the forward outputs')
    dev = pts.device
    g_planes = torch.zeros_like(planes_nhwc) if need_planes else None
    g_dec = None
    if need_decoder:
        g_dec = (torch.zeros([64, 32], dtype=torch.float32, device=dev), torch.zeros([64], dtype=torch.float32, device=dev),
                 torch.zeros([33, 64], dtype=torch.float32, device=dev), torch.zeros([33], dtype=torch.float32, device=dev))
    gd = g_dec if g_dec is not None else (None, None, None, None)
    with _on_device(dev):
        code = load().gnerf_query_points_backward(_ptr(planes_nhwc), n_items, planes_nhwc.shape[1], planes_nhwc.shape[2], _ptr(pts), n_pts,
                                                  float(box_warp), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(gs), _ptr(gc),
                                                  _ptr(g_planes), _ptr(gd[0]), _ptr(gd[1]), _ptr(gd[2]), _ptr(gd[3]), interleaved, _stream(pts))
    _check(code, 'gnerf_query_points_backward')
    return g_planes, g_dec
