"""The same pairing inside ONE process on two streams (v1 GEMM beside the transform kernel): 0 deviations in 6000 launches."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused
import viditq_extension.qgemm as qgemm
from qdiff.quarot import quarot_utils as qu
DEV = "cuda"
rows, C = 270, 512
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.randn(rows, C, device=DEV, generator=g)
pm = torch.randn(C, device=DEV, generator=g)
rot = qu.kernel_rotation_params(C, DEV)
a = torch.randint(-128, 128, (270, 512), dtype=torch.int8, device=DEV); w = torch.randint(-128, 128, (1024, 512), dtype=torch.int8, device=DEV)
sa = torch.rand(270, device=DEV); sw = torch.rand(1024, device=DEV)
s_, u_ = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
ref = fused.rotate_quant(x, pm, rot, u_, s_).clone()
torch.cuda.synchronize()
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
bad = 0
N = int(os.environ.get("ITERS", 3000))
outs = []
for i in range(N):
    with torch.cuda.stream(sA):
        for _ in range(4): qgemm.w8a8_linear(a, w, sa, sw, out_dtype=torch.bfloat16)
    with torch.cuda.stream(sB):
        q = fused.rotate_quant(x, pm, rot, u_, s_)
        bad_t = (q != ref).any()
        outs.append(bad_t)
    if i % 200 == 199:
        torch.cuda.synchronize()
        bad += int(torch.stack(outs).sum()); outs = []
torch.cuda.synchronize()
bad += int(torch.stack(outs).sum()) if outs else 0
print("same process, two streams: rotation glitches", bad, "of", N, flush=True)
