#!/usr/bin/env python3
"""One launch of each of the ten W8A8 GEMMs of a cfg-B DiT block (exact shapes, output dtypes and epilogues), in block
order.  Driver for PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass):
    rocprofv3 --pmc FETCH_SIZE --kernel-include-regex gemm_w8a8 --output-format csv -d <dir> -- python3 tools/gemm_block_shapes.py
`tools/gemm_traffic_summary.py` turns the two counter CSVs into profiles/*_gemm_traffic.{csv,json}."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
import viditq_extension.qgemm as qgemm  # noqa: E402

L, T, C, F = 32760, 512, 1536, 8960
# name, M, N, K, out dtype, gelu, gate+residual
LAUNCHES = [
    ("self_attn.q", L, C, C, torch.bfloat16, False, False),
    ("self_attn.k", L, C, C, torch.bfloat16, False, False),
    ("self_attn.v", L, C, C, torch.bfloat16, False, False),
    ("self_attn.o", L, C, C, torch.float32, False, True),
    ("cross_attn.q", L, C, C, torch.bfloat16, False, False),
    ("cross_attn.k", T, C, C, torch.bfloat16, False, False),
    ("cross_attn.v", T, C, C, torch.bfloat16, False, False),
    ("cross_attn.o", L, C, C, torch.float32, False, True),
    ("ffn.0", L, F, C, torch.bfloat16, True, False),
    ("ffn.2", L, C, F, torch.float32, False, True),
]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for name, M, N, K, odt, gelu, res in LAUNCHES:
        a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
        w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        sa = torch.rand(M, device=dev, generator=g) * 0.01
        asum = torch.rand(M, device=dev, generator=g)
        sw = torch.rand(N, device=dev, generator=g) * 0.01
        zp = torch.randn(N, device=dev, generator=g)
        bias = torch.randn(N, device=dev, generator=g)
        gate = residual = out = None
        if res:
            gate = torch.randn(N, device=dev, generator=g)
            residual = torch.randn(M, N, device=dev, generator=g)
            out = residual  # in place, as the block does
        torch.cuda.synchronize()
        qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=odt, gelu=gelu, gate=gate, residual=residual, out=out)
        torch.cuda.synchronize()
        print(name, M, N, K, flush=True)


if __name__ == "__main__":
    main()
