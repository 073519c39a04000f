"""Plugin loader with the reference's interface (torch_utils/custom_ops.py:61-157):
`get_plugin(module_name, sources, headers, source_dir, **build_kwargs)` returns an object exposing the
plugin's entry points; `verbosity` controls the status line.

The reference JIT-compiles CUDA sources with nvcc at first use.  Here the three plugins are views of
ONE ahead-of-time-built library, libgnerf_hip.so (hand-written gfx950 kernels, C ABI in
include/gnerf_hip.h), reached through the thin PyTorch C++ extension gnerf_torch_ext.so
(csrc/torch_binding.cpp: pybind entry points with the reference plugins' exact signatures), or through
the ctypes binding in gnerf_hip/__init__.py when the extension is not built / GNERF_HIP_BINDING=ctypes.
`sources`, `headers`, `source_dir` and the build keywords are accepted and ignored.  A missing library
is an error: there is no silent fallback for GPU tensors."""

import gnerf_hip

verbosity = 'brief'     # 'none', 'brief', 'full'

_cached_plugins = dict()


class _BiasActPlugin:
    """bias_act_plugin (reference bias_act.cpp:98-101)."""
    bias_act = staticmethod(gnerf_hip.bias_act)


class _Upfirdn2dPlugin:
    """upfirdn2d_plugin (reference upfirdn2d.cpp:106-109)."""
    upfirdn2d = staticmethod(gnerf_hip.upfirdn2d)


class _FilteredLReluPlugin:
    """filtered_lrelu_plugin (reference filtered_lrelu.cpp:298-302)."""
    filtered_lrelu_act_ = staticmethod(gnerf_hip.filtered_lrelu_act_)

    # Return code -1 = "no fused kernel for these parameters": the caller then runs the generic upfirdn2d ->
    # filtered_lrelu_act_ -> upfirdn2d sequence, exactly as the reference does for configurations its fused
    # kernel lacks (filtered_lrelu.cpp:55-60, filtered_lrelu.py:225-231).
    filtered_lrelu = staticmethod(gnerf_hip.filtered_lrelu)


_PLUGINS = {
    'bias_act_plugin': _BiasActPlugin,
    'upfirdn2d_plugin': _Upfirdn2dPlugin,
    'filtered_lrelu_plugin': _FilteredLReluPlugin,
}


def get_plugin(module_name, sources, headers=None, source_dir=None, **build_kwargs):
    assert verbosity in ['none', 'brief', 'full']
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    if module_name not in _PLUGINS:
        raise RuntimeError(f'custom_ops.get_plugin: unknown plugin "{module_name}" (this build ships {sorted(_PLUGINS)})')
    if verbosity != 'none':
        print(f'Setting up PyTorch plugin "{module_name}"... ', end='' if verbosity == 'brief' else '\n', flush=True)
    try:
        gnerf_hip.load()
    except Exception:
        if verbosity == 'brief':
            print('Failed!')
        raise
    if verbosity == 'full':
        print(f'Done setting up PyTorch plugin "{module_name}" ({gnerf_hip.LIB_PATH}).')
    elif verbosity == 'brief':
        print('Done.')
    plugin = _PLUGINS[module_name]
    ext = gnerf_hip.ext()
    if ext is not None:                 # same entry points, C++ binding (lower host cost per call)
        plugin = type(plugin.__name__ + 'Ext', (), {name: staticmethod(getattr(ext, name))
                                                    for name in vars(plugin) if not name.startswith('_')})
    _cached_plugins[module_name] = plugin
    return plugin
