#!/usr/bin/env python3
"""The attention-map quantiser (three streamed passes, csrc/attn_map.hip) at the cfg-B self-attention shape, next to the plain
flash-attention kernel: bf16 q / k and the reference's full recipe (int8 q / k + fake-quantised v)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from viditq_extension import fused  # noqa: E402
from wan import ops  # noqa: E402

L, H, d = 32760, 12, 128
g = torch.Generator(device="cuda").manual_seed(0)
q, k, v = (torch.randn(L, H * d, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3))


def timeit(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


t_plain = timeit(lambda: ops.attention(q, k, v, H))
t_map = timeit(lambda: ops.attention_map_quant(q, k, v, H, 8, False))
ident = torch.zeros(L, d // 2, 2, device="cuda")
ident[..., 0] = 1.0  # rotary = identity: the kernel only quantises
q8, k8 = ops.rmsnorm_rope_q8(q, None, ident, d, False), ops.rmsnorm_rope_q8(k, None, ident, d, True)
vq = v.clone()
fused.fake_quant_cols_(vq, 8)
t_qk8 = timeit(lambda: ops.attention_qk8(q8, k8, v, H))
t_full = timeit(lambda: ops.attention_map_quant(q8, k8, vq, H, 8, False))
print(f"L={L} heads={H}: plain bf16 attention {t_plain:.2f} ms | attention-map quantiser (bf16 q/k) {t_map:.2f} ms = {t_map / t_plain:.2f}x | "
      f"int8 Q.K^T attention {t_qk8:.2f} ms | full recipe (int8 q/k, quantised v, quantised map) {t_full:.2f} ms = {t_full / t_plain:.2f}x")
