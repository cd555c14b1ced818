#!/usr/bin/env python3
"""Start-time stagger scan of the ping-pong int8 GEMM (WANQ_GEMM_STAGGER="cohorts:ns", read per launch): the four block shapes,
median of 7 rounds x 10 launches per setting, settings interleaved.  usage: gemm_stagger_scan.py"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.qgemm as qgemm
dev = "cuda"
os.environ["WANQ_GEMM_STAGGER_SCAN"] = "1"
L = 32760
SHAPES = [("CxC bf16", L, 1536, 1536, torch.bfloat16, False), ("ffn.0 bf16", L, 8960, 1536, torch.bfloat16, False),
          ("CxC f32+res", L, 1536, 1536, torch.float32, True), ("ffn.2 f32+res", L, 1536, 8960, torch.float32, True)]
SETTINGS = ["1:0", "2:1500", "2:3000", "2:6000", "2:12000", "4:1000", "4:2000", "4:4000", "4:8000", "8:1000", "8:2000", "8:4000"]
if len(sys.argv) > 1:
    SETTINGS = sys.argv[1:]
for name, M, N, K, od, res in SHAPES:
    g = torch.Generator(device=dev).manual_seed(1)
    a = (torch.randn(M, K, device=dev, generator=g) * 30).round().clamp(-128, 127).to(torch.int8)
    w = (torch.randn(N, K, device=dev, generator=g) * 30).round().clamp(-128, 127).to(torch.int8)
    sa = torch.rand(M, device=dev, generator=g) * 0.01; asum = a.float().sum(1) * sa
    sw = torch.rand(N, device=dev, generator=g) * 0.01; zp = torch.randn(N, device=dev, generator=g).round(); bias = torch.randn(N, device=dev, generator=g)
    gate = torch.randn(N, device=dev, generator=g) if res else None
    x = torch.randn(M, N, device=dev, generator=g) if res else None
    out = torch.empty(M, N, device=dev, dtype=od)
    ts = {s: [] for s in SETTINGS}
    for r in range(7):
        for s in SETTINGS:
            os.environ["WANQ_GEMM_STAGGER"] = s
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=od, gate=gate, residual=x, out=out)
            e1.record(); torch.cuda.synchronize()
            ts[s].append(e0.elapsed_time(e1) / 10 * 1e3)
    base = sorted(ts[SETTINGS[0]])[3]
    print(name + ": " + "  ".join(f"{s} {sorted(ts[s])[3]:.1f}us x{base / sorted(ts[s])[3]:.3f}" for s in SETTINGS), flush=True)
