#!/usr/bin/env python3
"""Attention kernel timing at cfg-B (L=32760, 12 heads, d=128): bf16 HIP kernel vs the int8 Q.K^T form (interleaved rounds in
ONE process, median and minimum) vs torch SDPA, plus the quantisation error of the int8 form against the bf16 kernel."""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops  # noqa: E402


def attention_sdpa(q, k, v, num_heads):
    """The same contraction through torch SDPA -- a timing reference only."""
    Lq, C = q.shape
    d = C // num_heads
    qh = q.view(1, Lq, num_heads, d).transpose(1, 2)
    kh = k.view(1, k.shape[0], num_heads, d).transpose(1, 2)
    vh = v.view(1, v.shape[0], num_heads, d).transpose(1, 2)
    o = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh, scale=1.0 / math.sqrt(d))
    return o.transpose(1, 2).reshape(Lq, C)


def time_once(fn, iters=3):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def ab(fns, rounds=7):
    for f in fns.values():
        f()
    ts = {n: [] for n in fns}
    for _ in range(rounds):
        for n, f in fns.items():
            ts[n].append(time_once(f))
    return {n: (sorted(v)[len(v) // 2], min(v)) for n, v in ts.items()}


shapes = [(32760, 32760, 12), (32760, 512, 12), (9450, 75600, 5)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    shapes = shapes[:1]
for (Lq, Lk, H) in shapes:
    g = torch.Generator(device="cuda").manual_seed(0)
    xq = torch.randn(Lq, H * 128, device="cuda", generator=g)
    xk = torch.randn(Lk, H * 128, device="cuda", generator=g)
    v = torch.randn(Lk, H * 128, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.ones(H * 128, device="cuda")
    q, k = ops.rmsnorm_rope_(xq.clone(), w, None, 128).to(torch.bfloat16), ops.rmsnorm_rope_(xk.clone(), w, None, 128).to(torch.bfloat16)
    q8, k8 = ops.rmsnorm_rope_q8(xq, w, None, 128, False), ops.rmsnorm_rope_q8(xk, w, None, 128, True)
    fl = 4.0 * Lq * Lk * 128 * H
    r = ab({"bf16": lambda: ops.attention(q, k, v, H), "qk8": lambda: ops.attention_qk8(q8, k8, v, H)})
    for n, (med, mn) in r.items():
        print(f"{n:5s} attention Lq={Lq} Lk={Lk} H={H}: median {med*1e3:8.3f} ms  min {mn*1e3:8.3f} ms  {fl/med/1e12:7.1f} TFLOP/s "
              f"({fl/med/2.5e15*100:.1f}% of the bf16 MFMA peak)")
    print(f"      qk8 / bf16 time ratio {r['qk8'][0] / r['bf16'][0]:.3f}")
    o16, o8 = ops.attention(q, k, v, H).float(), ops.attention_qk8(q8, k8, v, H).float()
    print(f"      int8 Q.K^T vs bf16 kernel: rel L2 {((o8 - o16).norm() / o16.norm()).item():.3e}  max abs {(o8 - o16).abs().max().item():.3e}")
    if Lq * Lk <= 32760 * 32760:
        t = time_once(lambda: attention_sdpa(q, k, v, H))
        print(f"sdpa  attention Lq={Lq} Lk={Lk} H={H}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TFLOP/s")

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
# the per-rank shape under 4-way sequence parallelism (3 of 12 heads, all 32760 tokens): 384 workgroups on 256 CUs
Lq = Lk = 32760
for H in (3, 1):
    q = torch.randn(Lq, H * 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(Lk, H * 128, device="cuda").to(torch.bfloat16)
    fl = 4.0 * Lq * Lk * 128 * H
    for s in (1, 2):
        t = time_once(lambda: ops.attention(q, k, v, H, splits=s), 5)
        print(f"hip  attention Lq={Lq} Lk={Lk} H={H} splits={s}: {t*1e3:8.3f} ms {fl/t/1e12:7.1f} TFLOPS")
