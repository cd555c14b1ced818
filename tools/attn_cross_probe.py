#!/usr/bin/env python3
"""Cross-attention shape (Lq = 32760, 12 heads) against the number of keys: intercept = per-workgroup fixed cost (Q load, first
tile latency, O store), slope = per-tile cost."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops  # noqa: E402

L, H = 32760, 12
q = torch.randn(L, H * 128, device="cuda").to(torch.bfloat16)
for Lk in (64, 128, 256, 512, 1024, 2048):
    k = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        ops.attention(q, k, v, H)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        ops.attention(q, k, v, H)
    e.record()
    torch.cuda.synchronize()
    print(f"Lk={Lk:5d}: {s.elapsed_time(e) / 20 * 1e3:7.1f} us")
