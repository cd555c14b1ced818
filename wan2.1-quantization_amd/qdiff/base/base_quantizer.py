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
    """Per-group (per-token) dynamic quantizer (base_quantizer.py:101-162).  Symmetric (:116-128; every Wan configuration): one
    fused HIP pass gives int8 codes, scale and sum.  Asymmetric (:130-149; selected by no Wan configuration, pinned by
    tests/golden/a2_dynamic_asym.npz): the row minimum / maximum, delta and zero point are those of the static asymmetric
    quantiser evaluated per call (same equations, eps floor 1e-8), on the same HIP kernels (row_minmax + weight_quant)."""

    # eps floors of delta: symmetric (:122-127) / asymmetric (:139-146).  MixedPrecisionDynamicQuantizer overrides both.
    _sym_floor, _asym_floor = 1e-6, 1e-8

    # ---- asymmetric branch ------------------------------------------------------------------------
    def _asym_params(self, x):
        lo, hi, _ = fused.row_minmax(x)
        hi, lo = hi.clamp_min(0.0), lo.clamp_max(0.0)
        delta = (hi - lo) / _full(hi, self.n_levels - 1)
        delta = torch.where(delta < self._asym_floor, _full(delta, self._asym_floor), delta)  # (the reference drops into ipdb first, then floors: :139-146)
        zp = torch.round(lo / delta) + self.n_levels / 2
        self.delta, self.zero_point = delta.unsqueeze(-1), zp.unsqueeze(-1)
        return delta.contiguous(), zp.contiguous()

    def _asym(self, x, want_codes, want_dequant, lo=None, hi=None):
        x = x.contiguous().float()
        delta, zp = self._asym_params(x)
        n = self.n_levels
        return fused.weight_quant(x, delta, zp, -n - 1 if lo is None else lo, n if hi is None else hi, want_codes, want_dequant)

    def quantize_int8(self, x, premul=None, rotation=None, want_sum=True):
        """int8 codes + fp32 (scale [T], sum [T]).  premul / rotation = (had_k, hadk): the ViDiT transform fused in.
        Asymmetric: the per-token zero point is left in `self.zero_point`; the caller adds its rank-one term
        (QuantizedLinear.forward).  int8 storage saturates the code 128 the reference's loose clamp admits at an exact tie (D9)."""
        x = x.contiguous()
        rows = x.shape[0]
        if not self.sym:
            if premul is not None or rotation is not None:
                # (x * mask) @ R first, in fp32, then the asymmetric quantiser on the transformed row, as the reference's
                # ViDiTQuantizedLinear.forward orders it (viditq_quant_layer.py:60-73 with base_quantizer.py:130-149)
                y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                fused.rotate_quant(x, premul, rotation, None, None, out_fp=y, quantize=False)
                x = y
            codes, _ = self._asym(x, True, False, -128, 127)
            scale = self.delta.reshape(-1).float().contiguous()
            return codes, scale, (codes.float().sum(dim=1) * scale) if want_sum else None
        qs = torch.empty(2, rows, dtype=torch.float32, device=x.device)
        if self.n_bits != 8 or self._sym_floor != 1e-6:
            # any other range (n_bits < 8; the mixed-precision class has no floor): codes still fit int8 and the GEMM is unchanged.
            # Under a transform the transformed row is written out in fp32 first and quantised by the plain kernel (simulation
            # mode only: kernel-mode activations are always 8-bit)
            if not 2 <= self.n_bits <= 8:
                raise NotImplementedError(f"int8 activation codes hold 2..8 bits, not {self.n_bits}")
            if premul is not None or rotation is not None:
                y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
                fused.rotate_quant(x, premul, rotation, None, None, out_fp=y, quantize=False)
                x = y
            q = fused.quant_sum_levels(x, qs[1] if want_sum else None, qs[0], 2 ** (self.n_bits - 1) - 1, self._sym_floor)
        elif premul is None and rotation is None:
            q = fused.quant_sum(x, qs[1] if want_sum else None, qs[0])
        else:
            q = fused.rotate_quant(x, premul, rotation, qs[1] if want_sum else None, qs[0])
        self.delta, self.zero_point = qs[0].unsqueeze(-1), torch.zeros(rows, 1, device=x.device)
        return q, qs[0], qs[1]

    def forward_with_quant_params(self, x, delta, mixed_precision=None):
        """base_quantizer.py:164-206: fake-quant with a PRECOMPUTED delta of x's shape (the reference's block-wise attention-map and
        pre-softmax quantisers), optionally with a per-element bit-width tensor (0 bits = masked to zero).  Symmetric quantisers only,
        as the reference asserts; like the reference, `delta` is floored at 1e-6 IN PLACE.  One HIP launch (wanq_fake_quant_with_delta)."""
        assert self.sym
        if delta.shape != x.shape:
            delta = delta.expand_as(x)
        d32 = delta if (delta.dtype == torch.float32 and delta.is_contiguous()) else delta.float().contiguous()
        d32.clamp_min_(1e-6)
        bits = None if mixed_precision is None else mixed_precision.to(device=x.device, dtype=torch.int32).expand_as(x).contiguous()
        return fused.fake_quant_with_delta(x.contiguous(), d32, self.n_bits, bits)

    def quantize(self, x):
        assert x.dim() == 2
        if not self.sym:  # (8 bits: codes live in int8 storage, which saturates the 2^b-th level of the reference's loose clamp, D9)
            return self._asym(x, True, False, *((-128, 127) if self.n_bits == 8 else (None, None)))[0].float()
        return self.quantize_int8(x, want_sum=False)[0].float()

    def forward(self, x):
        if not self.sym:
            return self._asym(x, False, True)[1]
        q, scale, _ = self.quantize_int8(x, want_sum=False)
        return q.float() * scale.unsqueeze(-1)
