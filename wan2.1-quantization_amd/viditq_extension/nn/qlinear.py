"""W8A8OF16LinearDynamicInputScale: int8 weight x int8 activation linear with per-token dynamic input scale.
Interface and buffers of ViDiT-Q/kernels/viditq_extension/nn/qlinear.py:8-124
(weight int8 [N,K], bias fp16 [N], scale_weight fp16 [N], zp_weight int16 [N] or None)."""
import torch
import torch.nn as nn

from .. import qgemm
from .base import QuantParams


class W8A8OF16LinearDynamicInputScale(nn.Module):
    """Module form of `qgemm.w8a8_of16_bias_weight_asym / _sym` (one int8 MFMA GEMM with the dequantisation in its epilogue).

    Relaxations against the CUDA module: no padding of M to 128 rows or of the scale / sum vectors is needed (any M, N % 8
    == 0, K % 16 == 0), the GEMM runs on the caller's current stream rather than the legacy default stream, and shape or
    dtype violations raise RuntimeError instead of tripping a C assert.  The symmetric variant really calls the symmetric
    epilogue (the reference's forward always takes the asymmetric kernel and needs a zero-point buffer for it)."""

    def __init__(self, in_features, out_features, has_bias=True, weight_sym=True):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.has_bias, self.weight_sym = has_bias, weight_sym
        self.register_buffer("weight", torch.empty(out_features, in_features, dtype=torch.int8))
        self.register_buffer("bias", torch.empty(out_features, dtype=torch.float16) if has_bias else None)
        self.register_buffer("scale_weight", torch.empty(out_features, dtype=torch.float16))
        self.register_buffer("zp_weight", None if weight_sym else torch.empty(out_features, dtype=torch.int16))

    @classmethod
    def from_linear(cls, linear, weight_sym=True, init_only=False):
        """Per-output-channel weight quantisation rule of the reference (qlinear.py:85-102):
        sym:  s = absmax/127, q = clamp(rne(w/s));  asym: s = (max-min)/255, zp = rne(min/s)+128, q = clamp(rne(w/s)-zp)."""
        q = cls(linear.in_features, linear.out_features, linear.bias is not None, weight_sym)
        if init_only:
            return q
        w = linear.weight.data
        if w.dtype != torch.float16:
            raise AssertionError("from_linear expects an fp16 nn.Linear")
        q = q.to(w.device)
        if linear.bias is not None:
            q.bias.copy_(linear.bias.data.to(torch.float16))
        if weight_sym:
            s = w.abs().amax(dim=-1) / 127.0
            codes = torch.round(w / s[:, None]).clamp_(-128, 127)
        else:
            lo, hi = w.amin(dim=-1), w.amax(dim=-1)
            s = (hi - lo) / 255.0
            zp = torch.round(lo / s) + 128
            codes = (torch.round(w / s[:, None]) - zp[:, None]).clamp_(-128, 127)
            q.zp_weight.copy_(zp.to(torch.int16))
        q.weight.copy_(codes.to(torch.int8))
        q.scale_weight.copy_(s.to(torch.float16))
        return q

    def forward(self, input, quant_params: QuantParams, out_dtype=torch.float16):
        """input int8 [..., K] with its per-token scale/sum in quant_params -> [..., N]."""
        k = input.shape[-1]
        a = input.reshape(-1, k)
        zp = self.zp_weight
        if zp is not None and quant_params.sum_input is None:
            raise RuntimeError("asymmetric weights need QuantParams(has_sum_input=True)")
        y = qgemm.w8a8_linear(a, self.weight, quant_params.scale_input[: a.shape[0]], self.scale_weight, self.bias,
                              quant_params.sum_input[: a.shape[0]] if zp is not None else None, zp, out_dtype=out_dtype)
        return y.view(*input.shape[:-1], self.out_features)
