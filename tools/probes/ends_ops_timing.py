#!/usr/bin/env python3
"""Per-operator device time of the fp32 ends at the headline size: csrc/embed_head.hip entry points against their torch forms."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "wan2.1-quantization_amd"))
from wan import ops  # noqa: E402
from wan.modules.model import sinusoidal_embedding_1d  # noqa: E402
from wan.quant_wanx import QuantWanModel  # noqa: E402


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(n):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / n * 1e3


def main():
    torch.manual_seed(0)
    with torch.device("cuda"):
        m = QuantWanModel(None, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=1536, ffn_dim=8960, freq_dim=256,
                          text_dim=4096, out_dim=16, num_heads=12, num_layers=1, eps=1e-6).eval()
    x = torch.randn(16, 21, 60, 104, device="cuda")
    ctx = torch.randn(512, 4096, device="cuda") * 0.1
    t = torch.tensor([500], device="cuda")
    h = torch.randn(32760, 1536, device="cuda")
    e = torch.randn(1, 1536, device="cuda")
    hd, pe, te, tp, tx = m.head, m.patch_embedding, m.time_embedding, m.time_projection, m.text_embedding
    mod = hd.modulation.view(2, 1536)
    with torch.no_grad():
        rows = [
            ("patch embedding", lambda: m._patch_embed(x), lambda: ops.patch_embed(x, pe.weight, pe.bias)),
            ("sinusoid + time MLPs", lambda: tp(te(sinusoidal_embedding_1d(256, t).float())),
             lambda: ops.linear_f32(ops.linear_f32(ops.linear_f32(ops.time_sinusoid(t, 256), te[0].weight, te[0].bias, out_act="silu"),
                                                   te[2].weight, te[2].bias), tp[1].weight, tp[1].bias, in_act="silu")),
            ("text embedding", lambda: tx(ctx), lambda: m._text_embed_hip(ctx)),
            ("head + unpatchify", lambda: m.unpatchify(hd(h.unsqueeze(0), e), [(21, 30, 52)])[0].float(),
             lambda: ops.head(h, mod, e[0], hd.head.weight, hd.head.bias, hd.eps, latent_shape=(16, 21, 60, 104), patch=(1, 2, 2))),
        ]
        for name, a, b in rows:
            print(f"{name:22s} torch {timed(a):8.1f} us   hip {timed(b):8.1f} us")


if __name__ == "__main__":
    main()
