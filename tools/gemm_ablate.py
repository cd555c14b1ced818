"""int8 GEMM at the three cfg-B block shapes: full dequant epilogue (bf16 out) against the raw int32 output."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    return s.elapsed_time(e) / iters * 1e-3
for (M, N, K) in [(32760, 1536, 1536), (32760, 1536, 8960), (32760, 8960, 1536)]:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
    sa = torch.rand(M, device=dev) * 0.01; asum = torch.rand(M, device=dev)
    sw = torch.rand(N, device=dev) * 0.01; zp = torch.randn(N, device=dev); bias = torch.randn(N, device=dev)
    t = timeit(lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16))
    t32 = timeit(lambda: qgemm.w8a8_o32(a, w))
    print(f"M={M} N={N} K={K}: bf16-epi {t*1e6:8.1f} us {2.0*M*N*K/t/1e12:7.1f} TOPS | o32 {t32*1e6:8.1f} us")

# W4A8: the same shapes with the weight packed 2 codes per byte (unpacked in registers next to the MFMAs) + the 14B per-rank shapes
print("--- W8A8 vs W4A8 (packed weight, in-kernel nibble unpack), bf16 epilogue")
for (M, N, K) in [(32760, 1536, 1536), (32760, 1536, 8960), (32760, 8960, 1536), (9450, 5120, 5120), (9450, 13824, 5120), (9450, 5120, 13824)]:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev)
    wp = qgemm.pack_w4(w, bias=8)
    sa = torch.rand(M, device=dev) * 0.01; asum = torch.rand(M, device=dev)
    sw = torch.rand(N, device=dev) * 0.01; zp = torch.randn(N, device=dev); bias = torch.randn(N, device=dev)
    t8 = timeit(lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16))
    t4 = timeit(lambda: qgemm.w8a8_linear(a, wp, sa, sw, bias, asum, zp - 8.0, out_dtype=torch.bfloat16, w4=True))
    print(f"M={M} N={N} K={K}: W8 {t8*1e6:8.1f} us {2.0*M*N*K/t8/1e12:7.1f} TOPS | W4 {t4*1e6:8.1f} us {2.0*M*N*K/t4/1e12:7.1f} TOPS  ({t8/t4:.3f}x)")
