"""MI355X-native MeshAnything inference engine (hot path: point cloud -> VQ face tokens -> mesh)."""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: the decode step is ~120 dependent launches of 2-8 MB
# each, so the argument fetch of every launch is on the critical path (measured -0.8 % per step,
# profiles/r01_ab_dev_kernarg.txt).  Must be set before the HIP runtime initialises; an explicit user setting wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .config import MAConfig, DTYPE_F32, DTYPE_BF16  # noqa: F401,E402
