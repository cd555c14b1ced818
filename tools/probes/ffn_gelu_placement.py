"""Where should the FFN's tanh-GELU run: in the ffn.0 GEMM epilogue (then a plain per-token quantiser) or in the quantiser
(gelu_quant_sum, the reference's own split: K/csrc/fused/fused.cu gelu_quant_sum behind an fp16 GEMM)?  cfg-B shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.qgemm as qgemm
import viditq_extension.fused as fused
dev = "cuda"
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M, N, K = 32760, 8960, 1536
a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
sa = torch.rand(M, device=dev) * 0.001; asum = a.float().sum(1) * sa
sw = torch.rand(N, device=dev) * 0.01; zp = torch.randn(N, device=dev).round(); bias = torch.randn(N, device=dev)
ssum = torch.empty(M, device=dev); sc = torch.empty(M, device=dev)
for rep in range(2):
    t_g = timeit(lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, gelu=True))
    t_p = timeit(lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, gelu=False))
    h = qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, gelu=False)
    t_q = timeit(lambda: fused.quant_sum(h, ssum, sc))
    t_gq = timeit(lambda: fused.gelu_quant_sum(h, ssum, sc))
    print(f"GEMM+GELU {t_g:7.1f} us + quant {t_q:7.1f} us = {t_g + t_q:7.1f} | GEMM {t_p:7.1f} us + gelu_quant {t_gq:7.1f} us = {t_p + t_gq:7.1f}", flush=True)
