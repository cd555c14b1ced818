"""INTEGRATION.md section 2 shows the ctypes binding a maintainer of the reference would write for one entry point.  This test EXECUTES
that block as printed (from the repository root, as the text says) and holds the function it defines to the oracle."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import kernel_ref as kr

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_ctypes_binding_printed_in_integration_md_runs_and_matches_the_oracle(monkeypatch):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2."):text.index("## 3.")]
    block = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "ctypes.CDLL" in block and "wanq_gemm_w8a8" in block
    monkeypatch.chdir(ROOT)
    ns = {}
    exec(compile(block, "INTEGRATION.md#2", "exec"), ns)
    f = ns["w8a8_of16_bias_weight_asym"]
    g = torch.Generator().manual_seed(5)
    M, N, K = 333, 256, 384  # a ragged row count
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, generator=g)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, generator=g)
    sa = (torch.rand(M, generator=g) * 0.02 + 1e-3).half()
    sw = (torch.rand(N, generator=g) * 0.02 + 1e-3).half()
    bias = torch.randn(N, generator=g).half()
    zp = torch.randint(-20, 20, (N,), dtype=torch.int16, generator=g)
    asum = (a.float().sum(1) * sa.float()).half()
    out = f(*(t.cuda() for t in (a, w, bias, sa, sw, asum, zp)))
    torch.cuda.synchronize()
    ref = kr.w8a8_epilogue(kr.w8a8_o32(a.numpy(), w.numpy()), sa.numpy(), sw.numpy(), bias.numpy(), asum.numpy(), zp.numpy())  # fp32 truth
    assert out.dtype == torch.float16 and out.shape == (M, N)
    d = np.abs(out.float().cpu().numpy() - ref)
    assert (d <= np.maximum(np.abs(ref) * 2 ** -10, 1e-3)).all()  # within fp16 rounding of the fp32 value (the bar of test_kbench_gemm_golden)
