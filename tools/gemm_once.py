import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.qgemm as qgemm
dev = "cuda"
for (M, N, K) in [(32760, 1536, 1536), (32760, 8960, 1536), (32760, 1536, 8960)]:
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
    sa = torch.rand(M, device=dev) * 0.01; asum = torch.rand(M, device=dev)
    sw = torch.rand(N, device=dev) * 0.01; zp = torch.randn(N, device=dev); bias = torch.randn(N, device=dev)
    for _ in range(3):
        qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
