#!/usr/bin/env python3
"""Library yardstick for the int8 GEMM shapes of the 1.3B block: torch._int_mm (hipBLASLt/rocBLAS i8->i32, no epilogue)
and bf16 torch.matmul, against wanq_gemm_w8a8 with its full dequant epilogue."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from viditq_extension import qgemm  # noqa: E402


def timeit(fn, iters=10, warm=3):
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


M = 32768
for (N, K) in [(1536, 1536), (8960, 1536), (1536, 8960)]:
    a = torch.randint(-127, 127, (M, K), device="cuda", dtype=torch.int8)
    w = torch.randint(-127, 127, (N, K), device="cuda", dtype=torch.int8)
    ops = 2.0 * M * N * K
    try:
        t = timeit(lambda: torch._int_mm(a, w.t()))
        print(f"torch._int_mm   M={M} N={N} K={K}: {t*1e3:7.3f} ms {ops/t/1e12:7.1f} TOPS")
    except Exception as ex:  # noqa: BLE001
        print("torch._int_mm failed:", ex)
    ab, wb = a.to(torch.bfloat16), w.to(torch.bfloat16)
    t = timeit(lambda: ab @ wb.t())
    print(f"bf16 matmul     M={M} N={N} K={K}: {t*1e3:7.3f} ms {ops/t/1e12:7.1f} TFLOPS")
    sa = torch.rand(M, device="cuda").half() * 0.01
    suma = a.float().sum(1).half()
    sw = torch.rand(N, device="cuda").half() * 0.01
    zp = torch.zeros(N, device="cuda", dtype=torch.int16)
    bias = torch.zeros(N, device="cuda").half()
    t = timeit(lambda: qgemm.w8a8_of16_bias_weight_asym(a, w, bias, sa, sw, suma, zp))
    print(f"wanq_gemm_w8a8  M={M} N={N} K={K}: {t*1e3:7.3f} ms {ops/t/1e12:7.1f} TOPS")
