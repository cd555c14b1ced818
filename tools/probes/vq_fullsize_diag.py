"""Diagnostic: fake_quant_cols_ at [32760, 1536] against the oracle on the CPU and on the GPU (where do they differ?)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "wan2.1-quantization_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import viditq_extension.fused as fused
from oracle import wan_ref as wr

L, C, H = 32760, 1536, 12
g = torch.Generator(device="cuda").manual_seed(33)
for dtype in (torch.float32, torch.bfloat16):
    v = (torch.randn(L, C, device="cuda", generator=g) * torch.exp(torch.randn(C, device="cuda", generator=g))).to(dtype)
    v[:, 3] = 0
    ref_gpu = wr.v_fake_quant(v.float().view(L, H, 128), 8).reshape(L, C)
    ref_cpu = wr.v_fake_quant(v.float().cpu().view(L, H, 128), 8).reshape(L, C)
    out, colmax = fused.fake_quant_cols_(v.clone(), 8)
    o = out.float().cpu()
    a = ref_cpu.to(dtype).float()
    b = ref_gpu.to(dtype).float().cpu()
    print(dtype, "kernel vs CPU oracle: differing", int((o != a).sum()), "max", float((o - a).abs().max()),
          "| GPU oracle vs CPU oracle: differing", int((b != a).sum()), "max", float((b - a).abs().max()))
    d = (o != a).nonzero()
    if len(d):
        r, c = d[0].tolist()
        delta = float(v.float()[:, c].abs().max()) / 127
        print("  first diff at", r, c, "x", float(v[r, c]), "kernel", float(o[r, c]), "oracle", float(a[r, c]), "delta", delta, "x/delta", float(v[r, c]) / delta)
