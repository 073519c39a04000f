# Overlay package: provides only G-NeRF's hot-path modules (training.volumetric_rendering.*).
# When the reference's g_nerf/ directory is also on sys.path (after this one), every other
# `training.*` module (triplane, networks_stylegan2, superresolution, ...) resolves there unchanged.
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
