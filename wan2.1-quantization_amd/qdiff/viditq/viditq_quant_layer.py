"""ViDiTQuantizedLinear: channel mask x random-Hadamard rotation (viditq/viditq_quant_layer.py:8-73)."""
from ..base.quant_layer import QuantizedLinear


class ViDiTQuantizedLinear(QuantizedLinear):
    uses_mask = True
    uses_rotation = True

    def __init__(self, in_features, out_features, bias, device, quant_config, fp_module):
        super().__init__(in_features, out_features, bias, device, quant_config, fp_module)
        self.alpha = quant_config.viditq.alpha

    def update_quantized_weight_rotated_and_scaled(self):
        """W1 = Q(W / mask); W2 = Q((W1.double() @ R).float()): the weight is quantised TWICE, the second
        pass re-fits delta / zero_point (reference :40-50).  Kept as is."""
        assert self.channel_mask is not None and self.rotation_signs is not None
        self.w_quantizer.init_done = False
        w1 = self.w_quantizer(self.fp_module.weight.data.float() / self.channel_mask.reshape(1, -1))
        self._requantize(self._rotate_weight(w1))
        self.w_quantizer.init_done = True
