from ..base.quant_layer import SQQuantizedLinear  # noqa: F401
