#!/usr/bin/env python3
"""One self-attention launch (L x L) and one cross-attention launch (L x 512) of the cfg-B block: driver for PMC passes
(see tools/attn_traffic_summary.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops  # noqa: E402

L, T, H = 32760, 512, 12
LAUNCHES = [("self_attn", L, L), ("cross_attn", L, T)]


def main():
    for name, Lq, Lk in LAUNCHES:
        q = torch.randn(Lq, H * 128, device="cuda").to(torch.bfloat16)
        k = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
        v = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
        torch.cuda.synchronize()
        ops.attention(q, k, v, H)
        torch.cuda.synchronize()
        print(name, Lq, Lk, flush=True)


if __name__ == "__main__":
    main()
