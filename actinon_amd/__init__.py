"""actinon_amd -- MI355X-native trace/radiance path of the Actinon ray tracer behind its scene_s render seam.

The product is two C libraries (actinon_amd/lib): libactinon_hip.so (HIP kernels + C ABI, include/actinon_hip.h)
and libactinon_host.so (plain-C scene assembly + render driver, include/acn_scene.h).  This package is ctypes
plumbing over them for tests and benchmarks; importing it fails if the libraries are not built."""
import os as _os
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see actinon_hip.hip (concurrent lanes); no effect once HIP is initialised

from . import abi
from ._lib import AcnError, check, hip, host
from .scene import (Flat, Handle, Scene, cps_from_cl, detmath_eval, device_count, main_pass_positions, run_script, v3)

__all__ = ["abi", "AcnError", "check", "hip", "host", "Flat", "Handle", "Scene", "cps_from_cl", "detmath_eval", "run_script",
           "device_count", "main_pass_positions", "v3"]
