#!/usr/bin/env python3
"""Attention kernel timing at cfg-B (L=32760, 12 heads, d=128): HIP kernel vs torch SDPA."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops  # noqa: E402


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


for (Lq, Lk, H) in [(32760, 32760, 12), (32760, 512, 12), (9450, 75600, 5)]:
    q = torch.randn(Lq, H * 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    fl = 4.0 * Lq * Lk * 128 * H
    t = timeit(lambda: ops.attention(q, k, v, H))
    print(f"hip  attention Lq={Lq} Lk={Lk} H={H}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TFLOPS ({fl/t/2.5e15*100:.1f}% of bf16 MFMA peak)")
    t = timeit(lambda: ops.attention_sdpa(q, k, v, H))
    print(f"sdpa attention Lq={Lq} Lk={Lk} H={H}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TFLOPS")

# the per-rank shape under 4-way sequence parallelism (3 of 12 heads, all 32760 tokens): 384 workgroups on 256 CUs
Lq = Lk = 32760
for H in (3, 1):
    q = torch.randn(Lq, H * 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    fl = 4.0 * Lq * Lk * 128 * H
    for s in (1, 2):
        t = timeit(lambda: ops.attention(q, k, v, H, splits=s))
        print(f"hip  attention Lq={Lq} Lk={Lk} H={H} splits={s}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TFLOPS")
