# MI355X-native replacements for training.volumetric_rendering.{renderer, ray_marcher, ray_sampler, math_utils}.
from pkgutil import extend_path
__path__ = extend_path(__path__, __name__)
