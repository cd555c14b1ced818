#!/usr/bin/env python3
"""Golden fixture for the ASYMMETRIC branch of the reference's DynamicQuantizer (quant_utils/qdiff/base/base_quantizer.py:130-157:
x_max clipped to >= 0, x_min to <= 0, delta = (x_max - x_min) / (2^n - 1), zero_point = round(x_min / delta) + 2^n / 2,
x_int = round(x / delta) - zero_point clamped to [-2^n - 1, 2^n]) and for a QuantizedLinear whose activations use it
(base/quant_layer.py).  No Wan configuration selects this branch (every one quantises activations symmetrically); it is pinned
here so that the port of the class is complete.     python tests/golden/make_golden_dyn_asym.py   (build container only)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "gen"))  # omegaconf stand-in
sys.path.insert(0, "/root/reference/ViDiT-Q/quant_utils")
from omegaconf import OmegaConf  # noqa: E402
from qdiff.base.base_quantizer import DynamicQuantizer  # noqa: E402
from qdiff.base.quant_layer import QuantizedLinear  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(1)


def main():
    g = torch.Generator().manual_seed(2025)
    T, C = 9, 256
    x = torch.randn(T, C, generator=g) * torch.exp(0.7 * torch.randn(C, generator=g))
    x[1] = x[1].abs() + 0.05            # all positive: x_min clips to 0
    x[2] = -x[2].abs() - 0.05           # all negative: x_max clips to 0
    x[3] = torch.arange(C, dtype=torch.float32) - 100.5  # range 255 -> delta == 1: exact .5 ties
    x[4] *= 1e-3
    out = {"x": x}
    for bits in (8, 4):
        q = DynamicQuantizer(OmegaConf.create({"n_bits": bits, "sym": False}))
        q.module_name = "golden"
        out[f"q{bits}"] = q.quantize(x.clone()).to(torch.int32)
        out[f"delta{bits}"] = q.delta.reshape(-1).clone()
        out[f"zp{bits}"] = q.zero_point.reshape(-1).clone()
        out[f"dequant{bits}"] = q.forward(x.clone())
    # a QuantizedLinear with asymmetric 8-bit activations and asymmetric 8-bit weights
    lin = torch.nn.Linear(C, 24)
    lin.weight.data = torch.randn(24, C, generator=g) * 0.05
    lin.weight.data[:, 11] *= 6.0
    lin.bias.data = torch.randn(24, generator=g) * 0.1
    cfg = OmegaConf.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": False}})
    ql = QuantizedLinear(C, 24, True, "cpu", cfg, lin)
    ql.a_quantizer.module_name = "golden"
    out.update({"w": lin.weight.data, "b": lin.bias.data, "y": ql(x.reshape(1, T, C))[0]})
    np.savez_compressed(os.path.join(HERE, "a2_dynamic_asym.npz"), **{k: v.numpy() for k, v in out.items()})
    print({k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
