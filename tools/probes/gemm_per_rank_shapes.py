"""Per-rank GEMM shapes under sequence parallelism (tokens sharded over sp = 1, 2, 4 ranks at cfg-B): how well does the persistent
256 x 256-tile kernel fill 256 CUs when a rank has 128 / 64 / 32 m-tiles?  Time per launch and the ideal (sp = 1 time / sp)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.qgemm as qgemm
dev = "cuda"
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (N, K, name, od, kw) in [(1536, 1536, "C x C bf16", torch.bfloat16, {}), (8960, 1536, "ffn.0 bf16", torch.bfloat16, {}), (1536, 8960, "ffn.2 f32+res", torch.float32, {"res": True}),
                             (1536, 1536, "C x C f32+res", torch.float32, {"res": True})]:
    base = None
    for sp in (1, 2, 4):
        M = 32760 // sp
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev); w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
        sa = torch.rand(M, device=dev) * 0.01; asum = torch.rand(M, device=dev); sw = torch.rand(N, device=dev) * 0.01
        zp = torch.randn(N, device=dev).round(); bias = torch.randn(N, device=dev); gate = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev) if kw.get("res") else None
        t = timeit(lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=od, gate=gate if res is not None else None, residual=res, out=res))
        base = base or t
        tiles = -(-M // 256) * (N // 256)
        print(f"{name:14s} sp={sp} M={M:6d}: {tiles:5d} tiles = {tiles / 256:5.2f} rounds  {t:7.1f} us  ideal {base / sp:7.1f} us  efficiency {base / sp / t:5.2f}", flush=True)
