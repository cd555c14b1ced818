"""Victim half of the two-process stress: one rowwise kernel (KIND = ln_rot | ln | rot | rot_nopremul | rot_k1 | quant) launched
ITERS times on fixed inputs and compared bitwise with its first result.  Run it while stress_aggressor.py runs in another process."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused
from qdiff.quarot import quarot_utils as qu
DEV = "cuda"
rows, C = 270, 512
g = torch.Generator(device=DEV).manual_seed(0)
x = torch.randn(rows, C, device=DEV, generator=g)
sh, sc = torch.randn(1, C, device=DEV, generator=g) * 0.1, torch.randn(1, C, device=DEV, generator=g) * 0.1
pm = torch.randn(C, device=DEV, generator=g)
rot = qu.kernel_rotation_params(C, DEV)
KIND = os.environ.get("KIND", "ln_rot")
xb = x.to(torch.bfloat16)
def run():
    q_ = torch.empty(rows, C, dtype=torch.int8, device=DEV)
    s_, u_ = torch.zeros(rows, device=DEV), torch.zeros(rows, device=DEV)
    if KIND == "ln_rot":
        fused.layernorm_rotate_quant(q_, x, None, sh, sc, pm, rot, u_, s_, 1e-6)
    elif KIND == "ln":
        fused.layernorm_nobias_t2i_quant_sum_fuse(q_, x, None, sh, sc, u_, s_, 1e-6)
    elif KIND == "rot":
        return fused.rotate_quant(x, pm, rot, u_, s_)
    elif KIND == "rot_nopremul":
        return fused.rotate_quant(x, None, rot, u_, s_)
    elif KIND == "rot_k1":
        return fused.rotate_quant(x[:, :128].contiguous(), None, (1, None), u_, s_)
    elif KIND == "quant":
        return fused.quant_sum(xb, u_, s_)
    return q_
ref = run().clone()
bad = 0
N = int(os.environ.get("ITERS", 20000))
for i in range(N):
    if not torch.equal(run(), ref):
        bad += 1
print("victim", KIND, "glitches:", bad, "of", N, flush=True)
