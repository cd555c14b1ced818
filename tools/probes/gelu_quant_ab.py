#!/usr/bin/env python3
"""A/B of builds of csrc/rowwise.hip on the headline GELU + per-token quantise launch ([32760, 8960] bf16 -> int8 + scale + sum):
every library on the command line is loaded with ctypes, timed interleaved, outputs must be bit-equal to the first's."""
import ctypes
import sys

import torch

V, I, I64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.wanq_quant_rows.argtypes = [V, I, V, V, V, I, I64, I, I, I, V]
    libs.append((path.split("/")[-1], lib))
rows, cols = 32760, 8960
g = torch.Generator(device="cuda").manual_seed(0)
x = (torch.randn(rows, cols, device="cuda", generator=g) * 1.5).to(torch.bfloat16)
st = torch.cuda.current_stream().cuda_stream
outs = []
for name, lib in libs:
    q = torch.empty(rows, cols, dtype=torch.int8, device="cuda")
    sc, sm = torch.empty(rows, device="cuda"), torch.empty(rows, device="cuda")
    outs.append((q, sc, sm))
times = {name: [] for name, _ in libs}
for r in range(12):
    for (name, lib), (q, sc, sm) in zip(libs, outs):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            rc = lib.wanq_quant_rows(x.data_ptr(), 1, q.data_ptr(), sc.data_ptr(), sm.data_ptr(), 2, rows, cols, 1, 0, st)
            assert rc == 0, rc
        e1.record()
        torch.cuda.synchronize()
        if r >= 2:
            times[name].append(e0.elapsed_time(e1) / 5 * 1e3)
for (name, _), (q, sc, sm) in zip(libs, outs):
    t = sorted(times[name])
    same = all(torch.equal(a, b) for a, b in zip((q, sc, sm), outs[0]))
    print(f"{name:24s} median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f} us   {880.6e6 / t[len(t) // 2] / 1e6:5.2f} TB/s   bit-equal to first: {same}")
