#!/usr/bin/env python3
"""HIP-event timing of the row-wise (HBM-bound) kernels at the shapes of one cfg-B block pass: GB/s against algorithmic bytes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused  # noqa: E402
from qdiff.quarot import quarot_utils as qu  # noqa: E402
from wan import ops  # noqa: E402

DEV = "cuda"
L, C, F = 32760, 1536, 8960


def timeit(fn, iters=20, warm=3):
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


def report(name, t, nbytes):
    print(f"{name:58s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:5.2f} TB/s")


g = torch.Generator(device=DEV).manual_seed(0)
x32 = torch.randn(L, C, device=DEV, generator=g)
sh, sc = torch.randn(1, C, device=DEV, generator=g) * 0.1, torch.randn(1, C, device=DEV, generator=g) * 0.1
s_, u_ = torch.zeros(L, device=DEV), torch.zeros(L, device=DEV)
q = torch.empty(L, C, dtype=torch.int8, device=DEV)
report("LN+modulate+quant fp32 -> int8 [L,1536]", timeit(lambda: fused.layernorm_nobias_t2i_quant_sum_fuse(q, x32, None, sh, sc, u_, s_, 1e-6)), L * C * 5)
rot = qu.kernel_rotation_params(C, DEV)
pms = [torch.randn(C, device=DEV, generator=g) for _ in range(3)]
qs = [torch.empty(L, C, dtype=torch.int8, device=DEV) for _ in range(3)]
ss = [torch.zeros(L, device=DEV) for _ in range(3)]
us = [torch.zeros(L, device=DEV) for _ in range(3)]
report("LN+modulate+rotate+quant x3 fp32 -> 3 x int8 [L,1536]", timeit(lambda: fused.layernorm_rotate_quant_multi(qs, x32, None, sh, sc, pms, rot, us, ss, 1e-6)), L * C * 7)
report("LN+modulate+rotate+quant x1", timeit(lambda: fused.layernorm_rotate_quant(q, x32, None, sh, sc, pms[0], rot, u_, s_, 1e-6)), L * C * 5)
xb = torch.randn(L, C, device=DEV, generator=g).to(torch.bfloat16)
report("quant bf16 -> int8 [L,1536] (attention output)", timeit(lambda: fused.quant_sum(xb, u_, s_)), L * C * 3)
report("rotate+quant bf16 -> int8 [L,1536]", timeit(lambda: fused.rotate_quant(xb, pms[0], rot, u_, s_)), L * C * 3)
hb = torch.randn(L, F, device=DEV, generator=g).to(torch.bfloat16)
report("quant bf16 -> int8 [L,8960] (GELU output)", timeit(lambda: fused.quant_sum(hb, u_, s_)), L * F * 3)
report("GELU + quant bf16 -> int8 [L,8960] (ffn.0 pre-activation -> ffn.2 input: the step's form)", timeit(lambda: fused.gelu_quant_sum(hb, u_, s_)), L * F * 3)
rot_f = qu.kernel_rotation_params(F, DEV)
pm_f = torch.randn(F, device=DEV, generator=g)
qf = None
t_f = timeit(lambda: fused.rotate_quant(hb, pm_f, rot_f, u_, s_))
report("rotate(140x64, MFMA mix)+quant bf16 -> int8 [L,8960]", t_f, L * F * 3)
print(f"{'':58s} {2 * 140 * 140 * 64 * 3 * L / t_f / 1e12:8.1f} TFLOP/s bf16 MFMA (3-term split, unpadded)")
w = torch.ones(C, device=DEV)
rope = torch.randn(L, 64, 2, device=DEV, generator=g)
qb = torch.randn(L, C, device=DEV, generator=g).to(torch.bfloat16)
report("RMSNorm+RoPE bf16 in place [L,1536]", timeit(lambda: ops.rmsnorm_rope_(qb, w, rope, 128)), L * C * 4 + L * 64 * 8)
report("RMSNorm+RoPE+int8 (per-head) [L,1536]", timeit(lambda: ops.rmsnorm_rope_q8(qb, w, rope, 128, True)), L * C * 3 + L * 64 * 8)
r = torch.zeros(C, device=DEV)
report("calibration column absmax fp32 [L,1536]", timeit(lambda: fused.col_absmax_(r, x32)), L * C * 4)
