import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops
L, H = 32760, 12
q = torch.randn(L, H * 128, device="cuda").to(torch.bfloat16)
k = torch.randn(L, H * 128, device="cuda").to(torch.bfloat16)
v = torch.randn(L, H * 128, device="cuda").to(torch.bfloat16)
for _ in range(3):
    ops.attention(q, k, v, H)
torch.cuda.synchronize()
