"""xenoverse_amd — MI355X-native batched environment-step engine for Xenoverse worlds.

Hand-written HIP kernels (gfx950) behind a C-ABI (include/xeno.h, libxeno_hip.so) with the gymnasium
VectorEnv surface and the reference's set_task / task-dict API on top.  GPU only: importing the package is
cheap, but constructing any engine or env without libxeno_hip.so and a ROCm GPU raises.
"""
__version__ = "0.1.0"

from ._lib import XenoError, load as load_library  # noqa: F401
from .engine import Engine  # noqa: F401
from .registration import make_vec, register_with_gymnasium  # noqa: F401,E402

register_with_gymnasium()      # the reference's ids as vector entry points when gymnasium is importable
