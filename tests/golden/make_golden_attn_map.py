#!/usr/bin/env python3
"""Golden fixture for the attention-map quantiser: the reference's OWN QuantizedAttentionMapOpenSORA
(/root/reference/ViDiT-Q/quant_utils/qdiff/base/quant_attn.py:118-173, group 'row') applied to a seeded post-softmax map, and the
`attn @ v` it feeds (examples/Wan2.1/models/quant_opensora.py:459-476), on CPU with the omegaconf stand-in of make_golden.py.

    python tests/golden/make_golden_attn_map.py        (in the build container; writes tests/golden/a16_attn_map.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference/ViDiT-Q/quant_utils")
sys.path.insert(0, os.path.join(HERE, "gen"))  # omegaconf stand-in
from omegaconf import OmegaConf  # noqa: E402
from qdiff.base.quant_attn import QuantizedAttentionMapOpenSORA  # noqa: E402


def main():
    g = torch.Generator().manual_seed(161)
    BS, H, N, D = 1, 3, 45, 128
    q = torch.randn(BS, H, N, D, generator=g) * 1.5
    k = torch.randn(BS, H, N, D, generator=g) * 1.5
    k[:, :, 7] *= 3.0  # a dominant key column
    v = torch.randn(BS, H, N, D, generator=g)
    attn = ((q * D ** -0.5) @ k.transpose(-2, -1)).to(torch.float32).softmax(dim=-1)  # quant_opensora.py:461-468
    out = {"q": q, "k": k, "v": v, "attn": attn}
    for tag, bits, sym in (("8a", 8, False), ("8s", 8, True), ("4s", 4, True)):
        cfg = OmegaConf.create({"attn": {"attn_map": {"group": "row", "n_bits": bits, "sym": sym}, "qk": {"reorder_file_path": None}}})
        m = QuantizedAttentionMapOpenSORA(cfg)
        m.attn_map_quantizer.module_name = "golden"
        aq = m(attn.clone())
        out[f"attn_q_{tag}"] = aq
        out[f"x_{tag}"] = aq @ v  # quant_opensora.py:476
    # ---- the FULL recipe of quant_opensora.py:431-476: the q / k / v quantisers (8-bit symmetric DynamicQuantizer with the
    # reshapes of :431-440: q and k per (token, head) over head_dim, v per (head, channel) over all tokens), the map from the
    # quantised q and k, the map quantiser, then attn @ v with the quantised v
    from qdiff.base.base_quantizer import DynamicQuantizer
    qcfg = OmegaConf.create({"n_bits": 8, "sym": True})
    qq, kq, vq = DynamicQuantizer(qcfg), DynamicQuantizer(qcfg), DynamicQuantizer(qcfg)
    for z in (qq, kq, vq):
        z.module_name = "golden"
    q8 = qq(q.reshape([-1, D])).reshape([BS, H, N, D])
    k8 = kq(k.reshape([-1, D])).reshape([BS, H, N, D])
    v8 = vq(v.permute([0, 1, 3, 2]).reshape([-1, N])).reshape([BS, H, D, N]).permute([0, 1, 3, 2])
    attn8 = ((q8 * D ** -0.5) @ k8.transpose(-2, -1)).to(torch.float32).softmax(dim=-1)
    out.update({"full_q8": q8, "full_k8": k8, "full_v8": v8, "full_attn": attn8})
    for tag, bits, sym in (("8a", 8, False), ("8s", 8, True), ("4s", 4, True)):
        cfg = OmegaConf.create({"attn": {"attn_map": {"group": "row", "n_bits": bits, "sym": sym}, "qk": {"reorder_file_path": None}}})
        m = QuantizedAttentionMapOpenSORA(cfg)
        m.attn_map_quantizer.module_name = "golden"
        out[f"full_x_{tag}"] = m(attn8.clone()) @ v8
    np.savez_compressed(os.path.join(HERE, "a16_attn_map.npz"), **{k_: t.numpy() for k_, t in out.items()})
    print({k_: tuple(t.shape) for k_, t in out.items()})


if __name__ == "__main__":
    main()
