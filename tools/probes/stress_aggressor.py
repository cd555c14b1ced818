"""Aggressor half of the two-process stress: loops one workload (attn | attnbig | gemm | gemmbig | tiny | tinylds | torch) for N seconds.
usage: stress_aggressor.py <kind> <seconds>.  Only `gemm` (the 128x128 register-staged int8 GEMM, which can share a CU with the
victim's workgroups) makes the transform kernels of stress_victim.py deviate; see csrc/rowwise.hip."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops
import viditq_extension.qgemm as qgemm
kind = sys.argv[1]
DEV = "cuda"
if os.environ.get("SHIFT_VA"):
    _pad = torch.empty(int(os.environ["SHIFT_VA"]) << 20, dtype=torch.uint8, device=DEV)  # shift every later allocation
t_end = time.time() + float(sys.argv[2])
if kind == "attn":
    q = torch.randn(270, 512, device=DEV).to(torch.bfloat16); k = torch.randn(270, 512, device=DEV).to(torch.bfloat16); v = torch.randn(270, 512, device=DEV).to(torch.bfloat16)
    while time.time() < t_end:
        for _ in range(50): ops.attention(q, k, v, 4)
        torch.cuda.synchronize()
elif kind == "attnbig":
    q = torch.randn(4680, 1536, device=DEV).to(torch.bfloat16); k = torch.randn(4680, 1536, device=DEV).to(torch.bfloat16); v = torch.randn(4680, 1536, device=DEV).to(torch.bfloat16)
    while time.time() < t_end:
        for _ in range(20): ops.attention(q, k, v, 12)
        torch.cuda.synchronize()
elif kind == "gemm":
    a = torch.randint(-128, 128, (270, 512), dtype=torch.int8, device=DEV); w = torch.randint(-128, 128, (1024, 512), dtype=torch.int8, device=DEV)
    sa = torch.rand(270, device=DEV); sw = torch.rand(1024, device=DEV)
    while time.time() < t_end:
        for _ in range(50): qgemm.w8a8_linear(a, w, sa, sw, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
elif kind == "tiny":
    a = torch.randn(4096, device=DEV)
    while time.time() < t_end:
        for _ in range(200): a.add_(1.0)
        torch.cuda.synchronize()
elif kind == "tinylds":
    a = torch.randn(64, 64, device=DEV)
    while time.time() < t_end:
        for _ in range(100): (a @ a)
        torch.cuda.synchronize()
elif kind == "gemmbig":
    a = torch.randint(-128, 128, (32760, 1536), dtype=torch.int8, device=DEV); w = torch.randint(-128, 128, (1536, 1536), dtype=torch.int8, device=DEV)
    sa = torch.rand(32760, device=DEV); sw = torch.rand(1536, device=DEV)
    while time.time() < t_end:
        for _ in range(20): qgemm.w8a8_linear(a, w, sa, sw, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
elif kind == "torch":
    a = torch.randn(2048, 2048, device=DEV)
    while time.time() < t_end:
        for _ in range(20): (a @ a).relu_()
        torch.cuda.synchronize()
print("aggressor", kind, "done", flush=True)
