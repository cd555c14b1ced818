"""MI355X-native replacement of the reference's CUDA extension package `viditq_extension`
(ViDiT-Q/kernels/viditq_extension): same module names (`fused`, `qgemm`, `nn`), same function names and
argument order, backed by hand-written HIP kernels for gfx950 behind a C ABI (include/wanq_hip.h)."""
from . import _C  # noqa: F401  (raises ImportError when libwanq_hip.so is missing)
from . import fused, qgemm  # noqa: F401
