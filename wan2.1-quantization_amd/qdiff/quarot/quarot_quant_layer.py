"""QuarotQuantizedLinear: QuaRot, Hadamard rotation only (ViDiT-Q/quant_utils/qdiff/quarot/quarot_quant_layer.py:7-69)."""
from ..base.quant_layer import QuantizedLinear


class QuarotQuantizedLinear(QuantizedLinear):
    """QuaRot: rotation only (quarot/quarot_quant_layer.py:7-69)."""
    uses_rotation = True

    def update_quantized_weight_rotated(self):
        self.w_quantizer.init_done = False
        self._requantize(self._rotate_weight(self.fp_module.weight.data.float()))
        self.w_quantizer.init_done = True
