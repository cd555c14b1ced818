"""Quantizers with the reference's interface (ViDiT-Q/quant_utils/qdiff/base/base_quantizer.py:13-162): modules
with `delta` / `zero_point` buffers ([G,1]) and an `init_done` flag.  All heavy lifting is HIP
(viditq_extension.fused): row statistics, static quantisation and the per-token dynamic quantiser; the [G]-sized
parameter arithmetic is plain fp32 torch (IEEE, bit-identical to the reference's CPU result).
Tensors must live on the GPU: there is no CPU path."""
import torch
import torch.nn as nn

from viditq_extension import fused

from ..config import ListConfig


class BaseQuantizer(nn.Module):
    def __init__(self, quant_config):
        super().__init__()
        self.n_bits = quant_config["n_bits"]
        self.sym = quant_config.get("sym", False)
        if isinstance(self.n_bits, list):
            raise AssertionError("when multiple n_bits are adopted, use the MixedPrecisionBaseQuantizer")
        self.register_buffer("delta", None)
        self.register_buffer("zero_point", None)
        if not isinstance(self.n_bits, ListConfig):
            self.n_levels = self.levels(self.n_bits, self.sym)
        self.init_done = False
        self.module_name = None

    @staticmethod
    def levels(n_bits, sym):
        return 2 ** (n_bits - 1) - 1 if sym else 2 ** n_bits  # base_quantizer.py:32


def _full(t, v):
    return torch.full_like(t, float(v))  # divide by a TENSOR: torch-GPU turns `/ python_scalar` into a reciprocal multiply


def static_params(x, n_bits, sym):
    """(delta [G], zero_point [G]) of StaticQuantizer.init_quant_params (base_quantizer.py:70-90)."""
    assert x.dim() == 2
    lo, hi, am = fused.row_minmax(x.contiguous())
    if sym:
        return am / _full(am, 2 ** (n_bits - 1) - 1), torch.zeros_like(am)
    n_levels = 2 ** n_bits
    hi, lo = hi.clamp_min(0.0), lo.clamp_max(0.0)
    delta = (hi - lo) / _full(hi, n_levels - 1)
    return delta, torch.round(lo / delta) + n_levels / 2


class StaticQuantizer(BaseQuantizer):
    """Per-row (output channel) static quantizer for weights."""

    def init_quant_params(self, x):
        delta, zp = static_params(x, self.n_bits, self.sym)
        if not torch.all(delta > 1e-6):
            raise AssertionError("unexpected small delta exists")  # the reference drops into ipdb here (:94-97)
        self.delta, self.zero_point = delta.unsqueeze(-1), zp.unsqueeze(-1)

    def codes_and_dequant(self, x, want_codes=True, want_dequant=True):
        if self.init_done is not True:
            self.init_quant_params(x)
        n = self.n_levels
        codes, deq = fused.weight_quant(x.contiguous(), self.delta.reshape(-1).float().contiguous(),
                                        self.zero_point.reshape(-1).float().contiguous(), -n - 1, n, want_codes, want_dequant)
        if codes is not None and self.n_bits < 8:
            # The reference's clamp is one level looser than the bit-width: a row whose extremes tie exactly at a
            # .5 boundary (lo = -hi, e.g. lattice-valued rotated weights) produces a 17th level.  Integer STORAGE has 2^b
            # levels, so the codes saturate; the fake-quant `weight.data` keeps the reference's value.
            codes.clamp_(-(2 ** (self.n_bits - 1)), 2 ** (self.n_bits - 1) - 1)
        return codes, deq

    def quantize(self, x):
        return self.codes_and_dequant(x, True, False)[0].float()

    def forward(self, x):
        return self.codes_and_dequant(x, False, True)[1]


class DynamicQuantizer(BaseQuantizer):
    """Per-token dynamic quantizer for activations (sym, 8 bit: the only activation format of the Wan configs)."""

    def _check(self):
        if not self.sym or self.n_bits != 8:
            raise NotImplementedError("the HIP activation quantiser implements symmetric 8-bit per-token quantisation")

    def quantize_int8(self, x, premul=None, rotation=None, want_sum=True):
        """int8 codes + fp32 (scale [T], sum [T]).  premul / rotation = (had_k, hadk): the ViDiT transform fused in."""
        self._check()
        x = x.contiguous()
        rows = x.shape[0]
        qs = torch.empty(2, rows, dtype=torch.float32, device=x.device)
        if premul is None and rotation is None:
            q = fused.quant_sum(x, qs[1] if want_sum else None, qs[0])
        else:
            q = fused.rotate_quant(x, premul, rotation, qs[1] if want_sum else None, qs[0])
        self.delta, self.zero_point = qs[0].unsqueeze(-1), torch.zeros(rows, 1, device=x.device)
        return q, qs[0], qs[1]

    def quantize(self, x):
        assert x.dim() == 2
        return self.quantize_int8(x, want_sum=False)[0].float()

    def forward(self, x):
        q, scale, _ = self.quantize_int8(x, want_sum=False)
        return q.float() * scale.unsqueeze(-1)
