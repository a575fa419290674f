"""MI355X-native MeshAnything inference engine (hot path: point cloud -> VQ face tokens -> mesh)."""
from .config import MAConfig, DTYPE_F32, DTYPE_BF16  # noqa: F401
