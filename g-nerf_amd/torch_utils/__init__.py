# Overlay package: provides torch_utils.custom_ops and torch_utils.ops.{bias_act, upfirdn2d,
# filtered_lrelu, grid_sample_gradfix}.  With the reference's g_nerf/ later on sys.path, the rest of
# torch_utils (persistence, misc, training_stats, ops.conv2d_resample, ...) resolves there unchanged.
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
