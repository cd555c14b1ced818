from ..base.quant_layer import QuarotQuantizedLinear  # noqa: F401
