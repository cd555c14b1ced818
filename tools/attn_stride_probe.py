#!/usr/bin/env python3
"""Does the K/V row stride matter to the attention kernel's LDS-DMA staging?  Same 3 heads, once as columns of a 12-head
token-major buffer (row stride 1536 elements: a 64-key tile is 64 separate 256-B segments per head) and once head-packed
(row stride 384)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops  # noqa: E402

L, H = 32760, 2


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


big = [torch.randn(L, 12 * 128, device="cuda").to(torch.bfloat16) for _ in range(3)]
for name, cols in (("stride 1536", slice(0, H * 128)), ("stride 1536, heads 5-6", slice(5 * 128, (5 + H) * 128))):
    q, k, v = (t[:, cols] for t in big)
    print(f"{name:28s}: {timeit(lambda: ops.attention(q, k, v, H, splits=1)):.3f} ms")
q, k, v = (t[:, : H * 128].contiguous() for t in big)
print(f"{'stride 256 (packed)':28s}: {timeit(lambda: ops.attention(q, k, v, H, splits=1)):.3f} ms")
# one head, fully contiguous K/V rows (a 64-key tile = one 16-KiB block)
q, k, v = (t[:, :128].contiguous() for t in big)
print(f"{'1 head, stride 128':28s}: {timeit(lambda: ops.attention(q, k, v, 1, splits=1)):.3f} ms  (x2 = {2*timeit(lambda: ops.attention(q, k, v, 1, splits=1)):.3f})")
