"""Quantizers holding parameters for a LIST of bit-widths (interface of
ViDiT-Q/quant_utils/qdiff/base/mixed_precision_quantizer.py:15-186): `n_bits` is a ListConfig, `i_bitwidth`
selects the active entry, `bitwidth_refactor(i)` switches it."""
import torch

from ..config import ListConfig
from .base_quantizer import BaseQuantizer, DynamicQuantizer, StaticQuantizer, static_params


class MixedPrecisionStaticQuantizer(StaticQuantizer):
    def __init__(self, quant_config):
        BaseQuantizer.__init__(self, quant_config)
        assert isinstance(quant_config["n_bits"], ListConfig) and quant_config.get("i_bitwidth", None) is not None
        self.bitwidth_list = quant_config["n_bits"]
        self.i_bitwidth = quant_config["i_bitwidth"]
        self.n_bits = self.bitwidth_list[self.i_bitwidth]
        self.n_levels = self.levels(self.n_bits, self.sym)
        self.register_buffer("delta_list", None)
        self.register_buffer("zero_point_list", None)

    def init_quant_params(self, x):
        ds, zs = zip(*[static_params(x, b, self.sym) for b in self.bitwidth_list])
        for d in ds:
            assert torch.all(d > 1e-7), "unexpected small delta exists"
        self.delta_list = torch.stack([d.unsqueeze(-1) for d in ds])
        self.zero_point_list = torch.stack([z.unsqueeze(-1) for z in zs])
        self.delta, self.zero_point = self.delta_list[self.i_bitwidth], self.zero_point_list[self.i_bitwidth]

    def bitwidth_refactor(self, i_bitwidth):
        self.i_bitwidth = i_bitwidth
        self.n_bits = self.bitwidth_list[i_bitwidth]
        self.n_levels = self.levels(self.n_bits, self.sym)  # (the reference leaves n_levels stale here)
        self.delta, self.zero_point = self.delta_list[i_bitwidth], self.zero_point_list[i_bitwidth]


class MixedPrecisionDynamicQuantizer(DynamicQuantizer):
    """mixed_precision_quantizer.py:126-186: the per-token dynamic quantiser at the ACTIVE entry of a bit-width list (n_levels is
    recomputed from it on every call, :139).  Two differences from DynamicQuantizer, pinned by tests/golden/a7_mixed_dynamic.npz:
    the symmetric branch has NO eps floor (:141-146) -- a tiny row keeps its own delta -- and the asymmetric floor is 1e-6
    (:158-165), not 1e-8.  Where the reference is undefined -- an all-zero row in the symmetric branch is 0 / 0: NaN codes, NaN
    output -- this class returns delta 0, codes 0, output 0 (repo-defined; SURVEY D5's rule for behaviour the reference breaks)."""
    _sym_floor, _asym_floor = 0.0, 1e-6

    def __init__(self, quant_config):
        BaseQuantizer.__init__(self, quant_config)
        assert isinstance(quant_config["n_bits"], ListConfig) and quant_config.get("i_bitwidth", None) is not None
        self.bitwidth_list = quant_config["n_bits"]
        self.i_bitwidth = quant_config["i_bitwidth"]
        self.n_bits = self.bitwidth_list[self.i_bitwidth]
        self.n_levels = self.levels(self.n_bits, self.sym)

    def bitwidth_refactor(self, i_bitwidth):
        self.i_bitwidth = i_bitwidth
        self.n_bits = self.bitwidth_list[i_bitwidth]
        self.n_levels = self.levels(self.n_bits, self.sym)
