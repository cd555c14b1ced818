"""LayerNormGeneral: LayerNorm(no bias) -> adaLN modulate -> per-token int8 quantise (+row sum), one kernel.
Interface of ViDiT-Q/kernels/viditq_extension/nn/layernorm.py:6-54."""
import torch
import torch.nn as nn

from .. import fused
from .base import QuantParams


class LayerNormGeneral(nn.Module):
    """Module form of `fused.layernorm_nobias_t2i_quant_sum_fuse`.

    Differences from the CUDA module it stands in for, all of them relaxations: the input may be fp16, bf16 or fp32 (the
    statistics are fp32 either way), `hidden_size` is not limited to 4096 / 8192 (one wave per row up to 2048 columns, four
    beyond), the scale / sum buffers of `quant_params` may be fp16 (the reference's layout) or fp32, and a bad argument
    raises RuntimeError instead of aborting the process.  `act_sum` is kept for signature compatibility: the row sum is
    written whenever `quant_params.sum_input` exists, which is what every caller in the reference sets up."""

    def __init__(self, hidden_size, act_sum=False, eps=1e-6):
        super().__init__()
        self.hidden_size = hidden_size
        self.act_sum = act_sum
        self.variance_eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=torch.float16), requires_grad=False)

    @classmethod
    def from_layer_norm(cls, layer_norm):
        m = cls(layer_norm.normalized_shape[0], eps=layer_norm.eps)
        if layer_norm.weight is not None:
            m.weight = nn.Parameter(layer_norm.weight.detach().to(torch.float16), requires_grad=False)
        return m

    def forward(self, input, shift_msa, scale_msa, quant_params: QuantParams):
        """input [..., C] (fp16/bf16/fp32) -> int8 of the same shape; scale (and sum) written into quant_params."""
        c = input.shape[-1]
        x2 = input.contiguous().view(-1, c)
        out = torch.empty(x2.shape, dtype=torch.int8, device=x2.device)
        fused.layernorm_nobias_t2i_quant_sum_fuse(out, x2, self.weight, shift_msa.reshape(-1, c), scale_msa.reshape(-1, c),
                                                  quant_params.sum_input, quant_params.scale_input, self.variance_eps)
        return out.view(input.shape)
