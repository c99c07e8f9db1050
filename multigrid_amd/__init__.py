"""MI355X-native batched MultiGrid simulator (step / reset / gen_obs as HIP kernels)."""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
__version__ = "0.1.0"
