#!/usr/bin/env python3
"""Golden fixture for DynamicQuantizer.forward_with_quant_params of the reference (quant_utils/qdiff/base/base_quantizer.py:164-206):
the fake-quant step of its block-wise attention-map / pre-softmax quantisers with a PRECOMPUTED delta of x's own shape and an optional
per-element bit-width map (`mixed_precision`; 0 bits = masked to zero).  Symmetric quantiser only (it asserts so); delta below 1e-6 is
set to 1e-6 IN PLACE; plain form: delta / (2^b - 1), codes clamped to [0, 2^b - 1]; mixed form: delta / (2^bits - 1), codes clipped
from above only.           python tests/golden/make_golden_fwqp.py        (build container only: imports /root/reference)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "gen"))
sys.path.insert(0, "/root/reference/ViDiT-Q/quant_utils")
from omegaconf import OmegaConf  # noqa: E402
from qdiff.base.base_quantizer import DynamicQuantizer  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(1)


def main():
    g = torch.Generator().manual_seed(164)
    R, C, B = 24, 64, 8  # an attention-map-like tensor in [0, 1]; delta = the maximum of each 8 x 8 block, expanded to x's shape
    x = torch.softmax(torch.randn(R, C, generator=g) * 3.0, dim=-1)
    x[3] = 0.0
    x[8:16, 8:16] *= 1e-7     # a block whose maximum is below eps
    blk = x.reshape(R // B, B, C // B, B).amax(dim=(1, 3), keepdim=True).expand(R // B, B, C // B, B).reshape(R, C).contiguous()
    xs = torch.randn(R, C, generator=g)  # pre-softmax scores: signed values (the reference uses the method for those too, :168)
    blk_s = xs.abs().reshape(R // B, B, C // B, B).amax(dim=(1, 3), keepdim=True).expand(R // B, B, C // B, B).reshape(R, C).contiguous()
    bits = torch.tensor([0, 2, 4, 8])[torch.randint(0, 4, (R // B, C // B), generator=g)]
    bits = bits.reshape(R // B, 1, C // B, 1).expand(R // B, B, C // B, B).reshape(R, C).contiguous()
    out = {"x": x, "delta": blk, "xs": xs, "delta_s": blk_s, "bits": bits.to(torch.int32)}
    for b in (8, 4):
        q = DynamicQuantizer(OmegaConf.create({"n_bits": b, "sym": True}))
        q.module_name = "golden"
        d = blk.clone()
        out[f"y{b}"] = q.forward_with_quant_params(x.clone(), d)
        out[f"delta_after{b}"] = d  # floored in place
        out[f"ys{b}"] = q.forward_with_quant_params(xs.clone(), blk_s.clone())  # signed input: negative codes clamp to 0
    q = DynamicQuantizer(OmegaConf.create({"n_bits": 8, "sym": True}))
    q.module_name = "golden"
    out["y_mixed"] = q.forward_with_quant_params(x.clone(), blk.clone(), mixed_precision=bits.clone())
    out["ys_mixed"] = q.forward_with_quant_params(xs.clone(), blk_s.clone(), mixed_precision=bits.clone())  # no lower clamp in this form
    np.savez_compressed(os.path.join(HERE, "a16_forward_with_quant_params.npz"), **{k: v.numpy() for k, v in out.items()})
    print({k: tuple(v.shape) for k, v in out.items()})


if __name__ == "__main__":
    main()
