#!/usr/bin/env python3
"""LayerNorm + ViDiT transform + quantise for q / k / v of a cfg-B block: three passes vs the one-pass multi kernel."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused  # noqa: E402
from qdiff.quarot import quarot_utils as qu  # noqa: E402

DEV = "cuda"
L, C = 32760, 1536
x = torch.randn(L, C, device=DEV)
sh, sc = torch.randn(1, C, device=DEV) * 0.1, torch.randn(1, C, device=DEV) * 0.1
pms = [torch.randn(C, device=DEV) for _ in range(3)]
rot = qu.kernel_rotation_params(C, DEV)
qs = [torch.empty(L, C, dtype=torch.int8, device=DEV) for _ in range(3)]
scales, sums = [torch.zeros(L, device=DEV) for _ in range(3)], [torch.zeros(L, device=DEV) for _ in range(3)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def three():
    for i in range(3):
        fused.layernorm_rotate_quant(qs[i], x, None, sh, sc, pms[i], rot, sums[i], scales[i], 1e-6)


print(f"three passes : {timeit(three):7.1f} us")
print(f"one pass (3) : {timeit(lambda: fused.layernorm_rotate_quant_multi(qs, x, None, sh, sc, pms, rot, sums, scales, 1e-6)):7.1f} us")
print(f"plain LN+quant (no transform), one consumer: {timeit(lambda: fused.layernorm_nobias_t2i_quant_sum_fuse(qs[0], x, None, sh, sc, sums[0], scales[0], 1e-6)):7.1f} us")
