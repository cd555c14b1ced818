#!/usr/bin/env python3
"""Bitwise run-to-run determinism of every HIP entry point the block uses (no atomics anywhere, so any difference is a race):
each case is launched REPS times on the same inputs and compared with its first result."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused  # noqa: E402
import viditq_extension.qgemm as qgemm  # noqa: E402
from qdiff.quarot import quarot_utils as qu  # noqa: E402
from wan import ops  # noqa: E402

DEV, REPS = "cuda", int(os.environ.get("REPS", 60))
g = torch.Generator(device=DEV).manual_seed(0)


def check(name, fn):
    first = fn()
    first = [t.clone() for t in (first if isinstance(first, (list, tuple)) else [first])]
    bad = 0
    for _ in range(REPS):
        out = fn()
        out = out if isinstance(out, (list, tuple)) else [out]
        bad += any(not torch.equal(a, b) for a, b in zip(first, out))
    print(f"{name:55s} {'DETERMINISTIC' if bad == 0 else f'{bad}/{REPS} runs differ'}", flush=True)


for (Lq, Lk, H, kl) in [(270, 270, 4, None), (270, 24, 4, None), (1000, 1000, 2, 937), (4680, 4680, 12, None), (300, 5000, 1, None)]:
    q = torch.randn(Lq, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    k = torch.randn(Lk, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(Lk, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    check(f"attention Lq={Lq} Lk={Lk} H={H} klen={kl}", lambda: ops.attention(q, k, v, H, kl, splits=1))
    check(f"attention Lq={Lq} Lk={Lk} H={H} klen={kl} splits=2", lambda: ops.attention(q, k, v, H, kl, splits=2))

for (M, N, K) in [(270, 512, 512), (270, 1024, 512), (24, 512, 512), (4680, 1536, 1536), (4680, 8960, 1536), (4680, 1536, 8960)]:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=DEV, generator=g)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=DEV, generator=g)
    sa, asum = torch.rand(M, device=DEV, generator=g) * 0.01, torch.rand(M, device=DEV, generator=g)
    sw, zp, bias = torch.rand(N, device=DEV, generator=g) * 0.01, torch.randn(N, device=DEV, generator=g), torch.randn(N, device=DEV, generator=g)
    gate, res = torch.randn(N, device=DEV, generator=g), torch.randn(M, N, device=DEV, generator=g)
    check(f"gemm {M}x{N}x{K} bf16", lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16))
    check(f"gemm {M}x{N}x{K} bf16 gelu", lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, gelu=True))
    check(f"gemm {M}x{N}x{K} fp32 gate+res", lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.float32, gate=gate, residual=res,
                                                                    out=torch.empty_like(res)))

for (rows, C) in [(270, 512), (270, 1536), (4680, 1536), (270, 5120)]:
    x = torch.randn(rows, C, device=DEV, generator=g)
    sh, sc = torch.randn(1, C, device=DEV, generator=g) * 0.1, torch.randn(1, C, device=DEV, generator=g) * 0.1
    pms = [torch.randn(C, device=DEV, generator=g) for _ in range(3)]
    rot = qu.kernel_rotation_params(C, DEV)

    def multi():
        qs = [torch.empty(rows, C, dtype=torch.int8, device=DEV) for _ in range(3)]
        sc_, su_ = [torch.zeros(rows, device=DEV) for _ in range(3)], [torch.zeros(rows, device=DEV) for _ in range(3)]
        fused.layernorm_rotate_quant_multi(qs, x, None, sh, sc, pms, rot, su_, sc_, 1e-6)
        return qs + sc_ + su_

    def single():
        q_ = torch.empty(rows, C, dtype=torch.int8, device=DEV)
        s_, u_ = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
        fused.layernorm_rotate_quant(q_, x, None, sh, sc, pms[0], rot, u_, s_, 1e-6)
        return [q_, s_, u_]

    def plain():
        q_ = torch.empty(rows, C, dtype=torch.int8, device=DEV)
        s_, u_ = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
        fused.layernorm_nobias_t2i_quant_sum_fuse(q_, x, None, sh, sc, u_, s_, 1e-6)
        return [q_, s_, u_]

    check(f"LN+rotate+quant x3 [{rows},{C}]", multi)
    check(f"LN+rotate+quant    [{rows},{C}]", single)
    check(f"LN+quant           [{rows},{C}]", plain)
    xb = x.to(torch.bfloat16)
    check(f"quant bf16         [{rows},{C}]", lambda: [fused.quant_sum(xb, torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV))])

for (rows, H) in [(270, 4), (4680, 12)]:
    x = torch.randn(rows, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.rand(H * 128, device=DEV, generator=g) + 0.5
    rope = torch.randn(rows, 64, 2, device=DEV, generator=g)
    check(f"rmsnorm_rope       [{rows},{H*128}]", lambda: ops.rmsnorm_rope_(x.clone(), w, rope, 128, eps=1e-6))

# ---- round 2 kernels: int8-QK attention, per-head int8 form, scattered store, W4A8 GEMM, 8960 transform, column fake-quant,
# fused step
for (Lq, Lk, H) in [(270, 270, 4), (4680, 4680, 12), (300, 5000, 1)]:
    w = torch.rand(H * 128, device=DEV, generator=g) + 0.5
    rope = torch.randn(max(Lq, Lk), 64, 2, device=DEV, generator=g)
    xq = torch.randn(Lq, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    xk = torch.randn(Lk, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(Lk, H * 128, device=DEV, generator=g).to(torch.bfloat16)

    def qk8():
        q8 = ops.rmsnorm_rope_q8(xq, w, rope[:Lq].contiguous(), 128, False)
        k8 = ops.rmsnorm_rope_q8(xk, w, rope[:Lk].contiguous(), 128, True)
        return [q8.codes, q8.scales, k8.codes, k8.scales, ops.attention_qk8(q8, k8, v, H)]

    check(f"rmsnorm_rope_q8 + attention_qk8 Lq={Lq} Lk={Lk} H={H}", qk8)

from wan.distributed.parallel import SeqParallel  # noqa: E402
sp = SeqParallel(False)
sp.size = 4
x = torch.randn(1170, 1536, device=DEV, generator=g).to(torch.bfloat16)
w = torch.rand(1536, device=DEV, generator=g) + 0.5
rope = torch.randn(1170, 64, 2, device=DEV, generator=g)
numel, hmap, _ = sp.packed_layout(1170, 1536, 128, [(0, 128), (128, 384)], DEV)
check("rmsnorm_rope_scatter [1170,1536] P=4", lambda: ops.rmsnorm_rope_scatter(x, w, rope, 128, torch.zeros(numel, dtype=torch.bfloat16, device=DEV), hmap))

for (M, N, K) in [(270, 512, 512), (4680, 1536, 8960), (1000, 520, 1536)]:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=DEV, generator=g)
    wp = qgemm.pack_w4(torch.randint(-8, 8, (N, K), dtype=torch.int8, device=DEV, generator=g), bias=8)
    sa, asum = torch.rand(M, device=DEV, generator=g) * 0.01, torch.rand(M, device=DEV, generator=g)
    sw, zp, bias = torch.rand(N, device=DEV, generator=g) * 0.01, torch.randn(N, device=DEV, generator=g), torch.randn(N, device=DEV, generator=g)
    check(f"gemm W4A8 {M}x{N}x{K} bf16", lambda: qgemm.w8a8_linear(a, wp, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, w4=True))

hb = torch.randn(2340, 8960, device=DEV, generator=g).to(torch.bfloat16)
pm = torch.randn(8960, device=DEV, generator=g)
rot = qu.kernel_rotation_params(8960, DEV)


def r140():
    s_, u_ = torch.zeros(2340, device=DEV), torch.zeros(2340, device=DEV)
    return [fused.rotate_quant(hb, pm, rot, u_, s_), s_, u_]


check("rotate(140x64)+quant [2340,8960]", r140)
vv = torch.randn(4680, 1536, device=DEV, generator=g).to(torch.bfloat16)
check("col_absmax + fake_quant_cols [4680,1536]", lambda: list(fused.fake_quant_cols_(vv.clone(), 8)))
from wan.utils.fused_step import lincomb  # noqa: E402
ins = [torch.randn(16 * 21 * 60 * 104, device=DEV, generator=g) for _ in range(6)]
coef = torch.randn(3, 6).numpy()


def lc():
    outs = [torch.empty_like(ins[0]) for _ in range(3)]
    lincomb(coef, ins, outs)
    return outs


check("lincomb 3 x 6 latent-sized", lc)

# ---- round 5: the 13824-column transform, the narrow-range quantiser, the fake-quant with a precomputed delta, both attention forms
from viditq_extension import _C  # noqa: E402

for rows in (67, 1200):
    x = (torch.randn(rows, 13824, device=DEV, generator=g) * 2).to(torch.bfloat16)
    pm = torch.randn(13824, device=DEV, generator=g)
    rot = qu.kernel_rotation_params(13824, DEV)

    def r108():
        s, u = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
        return [fused.rotate_quant(x, pm, rot, u, s), s, u]
    check(f"rotate108 + quantise [{rows}, 13824]", r108)
for (rows, C, lv) in [(270, 1536, 31), (33, 8960, 7)]:
    x = torch.randn(rows, C, device=DEV, generator=g)

    def qlv():
        s, u = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
        return [fused.quant_sum_levels(x, u, s, lv, 0.0), s, u]
    check(f"quant_rows_levels [{rows}, {C}] levels={lv}", qlv)
x = torch.rand(512, 512, device=DEV, generator=g)
d = x.reshape(64, 8, 64, 8).amax(dim=(1, 3), keepdim=True).expand(64, 8, 64, 8).reshape(512, 512).contiguous()
check("fake_quant_with_delta [512, 512]", lambda: fused.fake_quant_with_delta(x, d, 8))
q = torch.randn(515, 4 * 128, device=DEV, generator=g).to(torch.bfloat16)
k = torch.randn(640, 4 * 128, device=DEV, generator=g).to(torch.bfloat16)
v = torch.randn(640, 4 * 128, device=DEV, generator=g).to(torch.bfloat16)
for form, name in ((0, "8 waves"), (1 << 40, "4 waves")):
    prev = _C.lib.wanq_attention_select_form(form)
    check(f"attention 515 x 640 x 4, {name}", lambda: ops.attention(q, k, v, 4, 601, splits=1))
    _C.lib.wanq_attention_select_form(prev)
