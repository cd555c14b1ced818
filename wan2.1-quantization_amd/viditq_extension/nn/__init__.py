from .base import QuantParams  # noqa: F401
from .layernorm import LayerNormGeneral  # noqa: F401
from .qlinear import W8A8OF16LinearDynamicInputScale  # noqa: F401
