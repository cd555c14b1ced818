#!/usr/bin/env python3
"""Upper bound of the 'attention output -> per-token int8 quant for `o`' fusion (SURVEY 8(f)1 back end).

The per-token scale of the attention output spans all 12 heads, which twelve different attention workgroups produce.  The
cheapest fused form lets every attention workgroup publish its head's row maximum (atomic max on float bits) and leaves a
finishing pass that quantises with the ready-made maximum.  That pass still reads the bf16 output and writes the int8 codes;
what it saves against today's stand-alone quant_sum is only the in-kernel row-maximum -- which quant_sum computes in registers
on the single read of the row.  This script times exactly that difference with the existing kernels: quant_sum (dynamic) against
quant_sum_static (the row maximum is an INPUT), same tensor, interleaved rounds, and checks the codes are identical."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.fused as fused
dev = "cuda"
L, C = 32760, 1536
g = torch.Generator(device=dev).manual_seed(0)
o = torch.randn(L, C, device=dev, generator=g).to(torch.bfloat16)
s0, u0 = torch.empty(L, device=dev), torch.empty(L, device=dev)
q0 = fused.quant_sum(o, u0, s0)
amax = o.float().abs().amax(1)
s1, u1 = amax.clone(), torch.empty(L, device=dev)
q1 = fused.quant_sum_static(o, u1, s1)
print("codes identical:", bool(torch.equal(q0, q1)), " sums identical:", bool(torch.equal(u0, u1)))
ts = {"dynamic": [], "static": []}
for r in range(9):
    for name in ("dynamic", "static"):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            if name == "dynamic":
                fused.quant_sum(o, u0, s0)
            else:
                s1.copy_(amax)  # (quant_sum_static leaves the scale in place of the maximum: restore the input; 131 KB)
                fused.quant_sum_static(o, u1, s1)
        e1.record(); torch.cuda.synchronize()
        ts[name].append(e0.elapsed_time(e1) / 20 * 1e3)
for name in ts:
    t = sorted(ts[name])
    print(f"{name:8s} median {t[4]:6.1f} us  min {t[0]:6.1f} us   ({3 * L * C / t[4] / 1e6:.2f} TB/s)")
