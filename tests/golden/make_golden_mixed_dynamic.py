#!/usr/bin/env python3
"""Golden fixture for the reference's MixedPrecisionDynamicQuantizer (quant_utils/qdiff/base/mixed_precision_quantizer.py:126-186;
selected for activations whose `n_bits` is a list at base/quant_layer.py:48-52): the per-token dynamic quantiser for a LIST of
bit-widths, `bitwidth_refactor(i)` between calls.  Differences from DynamicQuantizer that the fixture pins:
  * symmetric: delta = absmax / (2^(b-1) - 1) with NO eps floor (an all-zero row divides 0 by 0: the reference's codes and output
    are NaN there; a tiny row keeps its own tiny delta);
  * asymmetric: eps floor 1e-6 (DynamicQuantizer: 1e-8), reached through `import ipdb; ipdb.set_trace()` -- ipdb is not installed
    here, so the generator registers an arithmetic-free stand-in whose set_trace() returns, which is what a user continuing from
    the breakpoint gets;
  * n_levels is recomputed from the ACTIVE n_bits on every call (the static class leaves it stale after bitwidth_refactor).
    python tests/golden/make_golden_mixed_dynamic.py   (build container only: imports /root/reference)"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "gen"))  # omegaconf stand-in
sys.path.insert(0, "/root/reference/ViDiT-Q/quant_utils")
_ipdb = types.ModuleType("ipdb")
_ipdb.set_trace = lambda *a, **k: None
sys.modules.setdefault("ipdb", _ipdb)
from omegaconf import OmegaConf  # noqa: E402
from qdiff.base.mixed_precision_quantizer import MixedPrecisionDynamicQuantizer  # noqa: E402
from qdiff.base.quant_layer import QuantizedLinear  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(1)


def main():
    g = torch.Generator().manual_seed(707)
    T, C = 10, 256
    x = torch.randn(T, C, generator=g) * torch.exp(0.7 * torch.randn(C, generator=g))
    x[1] = x[1].abs() + 0.05             # all positive
    x[2] = -x[2].abs() - 0.05            # all negative
    x[3] = (torch.arange(C, dtype=torch.float32) - 127.0) / 2.0   # lattice: exact .5 ties at 8 bits (delta == 1)
    x[4] *= 1e-9                         # tiny row: |delta| far below every eps
    x[5] = 0.0                           # all-zero row: 0 / 0 in the symmetric branch
    x[6, 17] = 40.0                      # one outlier channel
    out = {"x": x, "bits": np.array([8, 6, 4])}
    with np.errstate(all="ignore"):
        for sym in (True, False):
            tag = "sym" if sym else "asym"
            q = MixedPrecisionDynamicQuantizer(OmegaConf.create({"n_bits": [8, 6, 4], "i_bitwidth": 0, "sym": sym}))
            q.module_name = "golden"
            for i, bits in enumerate((8, 6, 4)):
                q.bitwidth_refactor(i)
                assert q.n_bits == bits
                out[f"{tag}_q{bits}"] = q.quantize(x.clone()).to(torch.float32)       # float: NaN must survive
                out[f"{tag}_delta{bits}"] = q.delta.reshape(-1).clone()
                out[f"{tag}_zp{bits}"] = q.zero_point.reshape(-1).clone()
                out[f"{tag}_dequant{bits}"] = q.forward(x.clone())
            # back to the first entry: parameters depend on the active bit-width only, not on the history
            q.bitwidth_refactor(0)
            out[f"{tag}_q8_again"] = q.quantize(x.clone()).to(torch.float32)
    # a QuantizedLinear with a list of activation bit-widths (symmetric, as every Wan configuration has activations) at 6 bits
    lin = torch.nn.Linear(C, 24)
    lin.weight.data = torch.randn(24, C, generator=g) * 0.05
    lin.weight.data[:, 11] *= 6.0
    lin.bias.data = torch.randn(24, generator=g) * 0.1
    cfg = OmegaConf.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": [8, 6, 4], "i_bitwidth": 1, "sym": True}})
    ql = QuantizedLinear(C, 24, True, "cpu", cfg, lin)
    ql.a_quantizer.module_name = "golden"
    assert type(ql.a_quantizer).__name__ == "MixedPrecisionDynamicQuantizer"
    rows = [0, 1, 2, 3, 4, 6, 7, 8, 9]   # without the all-zero row (NaN in the reference)
    out.update({"w": lin.weight.data, "b": lin.bias.data, "lin_rows": np.array(rows), "y6": ql(x[rows].reshape(1, len(rows), C))[0]})
    np.savez_compressed(os.path.join(HERE, "a7_mixed_dynamic.npz"), **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()})
    print({k: tuple(np.shape(v)) for k, v in out.items()})
    print("NaN rows (sym 8):", np.unique(np.argwhere(np.isnan(out["sym_q8"].numpy()))[:, 0]))


if __name__ == "__main__":
    main()
