"""SQQuantizedLinear: SmoothQuant, channel mask only (ViDiT-Q/quant_utils/qdiff/smooth_quant/sq_quant_layer.py:6-68)."""
from ..base.quant_layer import QuantizedLinear


class SQQuantizedLinear(QuantizedLinear):
    """SmoothQuant: channel mask only (smooth_quant/sq_quant_layer.py:6-68)."""
    uses_mask = True

    def __init__(self, in_features, out_features, bias, device, quant_config, fp_module):
        super().__init__(in_features, out_features, bias, device, quant_config, fp_module)
        self.alpha = quant_config.smooth_quant.alpha

    def update_quantized_weight_scaled(self):
        assert self.channel_mask is not None
        self.w_quantizer.init_done = False
        self._requantize(self.fp_module.weight.data.float() / self.channel_mask.reshape(1, -1))
        self.w_quantizer.init_done = True
