"""Host side of the MI355X-native quantized Wan2.1 DiT: model definition, kernel-mode blocks, pipeline,
sequence parallelism.  Mirrors the layout of ViDiT-Q/examples/Wan2.1/wan/."""
from . import configs  # noqa: F401
