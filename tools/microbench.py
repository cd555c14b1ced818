#!/usr/bin/env python3
"""Kernel micro-benchmarks (HIP-event timed) for the hot-path kernels at cfg-B shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused  # noqa: E402
import viditq_extension.qgemm as qgemm  # noqa: E402

dev = "cuda"


def timeit(fn, iters=20, warm=5):
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


def main():
    L = 32760
    print(torch.cuda.get_device_name(0))
    for (M, N, K) in [(L, 1536, 1536), (L, 8960, 1536), (L, 1536, 8960), (512, 1536, 1536), (9450, 5120, 5120)]:
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
        w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
        sa = torch.rand(M, device=dev) * 0.01
        asum = torch.rand(M, device=dev)
        sw = torch.rand(N, device=dev) * 0.01
        zp = torch.randn(N, device=dev)
        bias = torch.randn(N, device=dev)
        t = timeit(lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.float16))
        ops = 2.0 * M * N * K
        print(f"gemm_w8a8 M={M} N={N} K={K}: {t*1e6:8.1f} us  {ops/t/1e12:7.1f} TOPS  ({ops/t/5.03e15*100:.1f}% of int8 MFMA peak)")
        xb = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        wb = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        t2 = timeit(lambda: torch.nn.functional.linear(xb, wb))
        print(f"   torch bf16 linear (hipBLASLt): {t2*1e6:8.1f} us  {ops/t2/1e12:7.1f} TFLOPS")
    for dt in (torch.float32, torch.float16):
        for C in (1536, 8960):
            x = torch.randn(L, C, device=dev, dtype=dt)
            sc = torch.zeros(L, device=dev)
            sm = torch.zeros(L, device=dev)
            t = timeit(lambda: fused.quant_sum(x, sm, sc))
            by = L * C * (x.element_size() + 1) + 8 * L
            print(f"quant_rows {dt} C={C}: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s")
            q = torch.empty(L, C, dtype=torch.int8, device=dev)
            sh = torch.randn(1, C, device=dev)
            scl = torch.randn(1, C, device=dev)
            t = timeit(lambda: fused.layernorm_nobias_t2i_quant_sum_fuse(q, x, None, sh, scl, sm, sc, 1e-6))
            print(f"ln_t2i_quant {dt} C={C}: {t*1e6:8.1f} us  {by/t/1e9:7.0f} GB/s")
            run = torch.zeros(C, device=dev)
            t = timeit(lambda: fused.col_absmax_(run, x))
            print(f"col_absmax {dt} C={C}: {t*1e6:8.1f} us  {L*C*x.element_size()/t/1e9:7.0f} GB/s")
    # attention baseline: torch SDPA at cfg-B (1 x 12 heads x 32760 x 128)
    q = torch.randn(1, 12, L, 128, device=dev, dtype=torch.bfloat16)
    k = torch.randn_like(q)
    v = torch.randn_like(q)
    t = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v), iters=3, warm=1)
    fl = 4.0 * L * L * 128 * 12
    print(f"torch SDPA bf16 L={L} H=12 d=128: {t*1e3:8.2f} ms  {fl/t/1e12:7.1f} TFLOPS")


if __name__ == "__main__":
    main()
