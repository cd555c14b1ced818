#!/usr/bin/env python3
"""A/B timing of attention-kernel VARIANTS in one process (interleaved rounds, median / min): every shared library given on
the command line (built from different revisions of csrc/attention.hip, e.g. into lib/variants/) is loaded with ctypes and its
wanq_attention_fwd is called on the same tensors.  Timing across gpurun boxes differs by +-3 %, so kernel edits are compared
this way.  usage: ab_attn_variants.py libA.so libB.so ..."""
import ctypes
import math
import sys

import torch

libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    lib.wanq_attention_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + \
        [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_void_p]
    lib.wanq_attention_fwd.restype = ctypes.c_int
    libs.append((path.split("/")[-1], lib))

for (Lq, Lk, H) in [(32760, 32760, 12), (32760, 512, 12)]:
    g = torch.Generator(device="cuda").manual_seed(0)
    q = torch.randn(Lq, H * 128, device="cuda", generator=g).to(torch.bfloat16)
    k = torch.randn(Lk, H * 128, device="cuda", generator=g).to(torch.bfloat16)
    v = torch.randn(Lk, H * 128, device="cuda", generator=g).to(torch.bfloat16)
    outs = {}

    def run(lib, o):
        rc = lib.wanq_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1, Lq, Lk, H, 128, H * 128, H * 128, H * 128,
                                    H * 128, 1.0 / math.sqrt(128), torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    ts = {n: [] for n, _ in libs}
    for n, lib in libs:
        outs[n] = torch.empty_like(q)
        run(lib, outs[n])
    for _ in range(9):
        for n, lib in libs:
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                run(lib, outs[n])
            e.record()
            torch.cuda.synchronize()
            ts[n].append(s.elapsed_time(e) / 3)
    fl = 4.0 * Lq * Lk * 128 * H
    base = None
    for n, _ in libs:
        t = sorted(ts[n])
        med = t[len(t) // 2]
        base = base or med
        d = (outs[n].float() - outs[libs[0][0]].float()).abs().max().item()
        print(f"Lq={Lq} Lk={Lk} H={H} {n:28s} median {med:7.3f} ms  min {t[0]:7.3f} ms  {fl / med / 1e9:7.1f} TFLOP/s  x{base / med:.3f}  max|o - o_first| {d:.2e}")
