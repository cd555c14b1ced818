import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused
from wan import ops
dev = "cuda"; L = 32760; C = 1536
x = torch.randn(L, C, device=dev)
sh = torch.randn(1, C, device=dev); sc = torch.randn(1, C, device=dev)
q = torch.empty(L, C, dtype=torch.int8, device=dev); s = torch.zeros(L, device=dev); sm = torch.zeros(L, device=dev)
xb = torch.randn(L, C, device=dev).to(torch.bfloat16)
hb = torch.randn(L, 8960, device=dev).to(torch.bfloat16)
s2 = torch.zeros(L, device=dev); sm2 = torch.zeros(L, device=dev)
run = torch.zeros(C, device=dev)
w = torch.ones(C, device=dev)
rope = torch.randn(L, 64, 2, device=dev)
for _ in range(3):
    fused.layernorm_nobias_t2i_quant_sum_fuse(q, x, None, sh, sc, sm, s, 1e-6)   # rowwise_kernel<1,3,true>
    fused.quant_sum(xb, sm, s)                                                   # rowwise_kernel<1,3,false>
    fused.quant_sum(hb, sm2, s2)                                                 # rowwise_kernel<4,5,false>
    fused.col_absmax_(run, x)                                                    # col_absmax_kernel<F32>
    ops.rmsnorm_rope_(xb, w, rope, 128)                                          # rmsnorm_rope_kernel<1,3>
torch.cuda.synchronize()
