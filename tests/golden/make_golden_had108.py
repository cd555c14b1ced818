#!/usr/bin/env python3
"""Golden fixture for the Hadamard transform of 13824 columns (Wan2.1-14B ffn.2 in_features) = H_108 (x) H_128.

The reference cannot rotate 13824 columns: get_hadK's precedence reaches `n % 144 == 0` first and asserts is_pow2(13824 // 144) = 96
(quarot_utils.py:110-112) before its K = 108 branch (:118-121), whose co-factor 128 IS a power of two.  This repository
defines the behaviour the reference's own table gives once that precedence accident is skipped: K = 108, the reference's
get_had108() table, the reference's own butterfly loop and fp32 sqrt (matmul_hadU, :158-179).  The fixture is made by running the
reference's matmul_hadU with get_hadK answering (get_had108(), 108) for this one size -- every arithmetic step is the reference's.
    python tests/golden/make_golden_had108.py      (build container only: imports /root/reference)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "gen"))
sys.path.insert(0, "/root/reference/ViDiT-Q/quant_utils")
from qdiff.quarot import quarot_utils  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(4)


def main():
    n = 13824
    try:
        quarot_utils.get_hadK(n)
        raise SystemExit("expected the reference's get_hadK(13824) to assert")
    except AssertionError:
        pass
    had108 = quarot_utils.get_had108()
    assert tuple(had108.shape) == (108, 108)
    orig = quarot_utils.get_hadK
    quarot_utils.get_hadK = lambda m, transpose=False: ((had108.T if transpose else had108), 108) if m == n else orig(m, transpose)
    g = torch.Generator().manual_seed(108)
    s = (torch.randint(0, 2, (n,), generator=g) * 2 - 1).to(torch.float64)
    x = torch.randn(2, n, generator=g, dtype=torch.float32)
    x[1] *= torch.exp(0.7 * torch.randn(n, generator=g, dtype=torch.float32))  # outlier channels
    x = x.to(torch.float64)  # fp32-representable inputs (stored as fp32), transformed in fp64
    hx = quarot_utils.matmul_hadU(x)
    # orthogonality on unit vectors: hadU(e_i) . hadU(e_j) = delta_ij for a few (i, j)
    idx = [0, 1, 127, 128, 6911, n - 1]
    E = torch.zeros(len(idx), n, dtype=torch.float64)
    E[torch.arange(len(idx)), idx] = 1.0
    HE = quarot_utils.matmul_hadU(E)
    gram = HE @ HE.T
    out = dict(signs=s.numpy(), K=np.int64(108), x=x.numpy().astype(np.float32), hadU_x=hx.numpy(), unit_rows=np.array(idx),
               unit_gram_err=np.float64((gram - torch.eye(len(idx), dtype=torch.float64)).abs().max()),
               hadU_units_first_col_block=HE[:, :256].numpy(), table_row0=had108[0].numpy().astype(np.int8), table_row1=had108[1].numpy().astype(np.int8),
               table_checksum=np.int64((had108.to(torch.int64) * torch.arange(1, 109).view(-1, 1) * torch.arange(1, 109).view(1, -1)).sum()))
    np.savez_compressed(os.path.join(HERE, "a5_hadamard_13824.npz"), **out)
    print({k: np.shape(v) for k, v in out.items()}, "gram err", out["unit_gram_err"])


if __name__ == "__main__":
    main()
