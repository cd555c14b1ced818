#!/usr/bin/env python3
"""A/B timing of int8-GEMM kernel VARIANTS in one process (interleaved rounds, median / min): every shared library given on the
command line (built from different revisions of csrc/gemm_w8a8.hip, e.g. into lib/variants/) is loaded with ctypes and its
wanq_gemm_w8a8 / wanq_gemm_w4a8 are called on the same tensors; outputs must be BIT-EQUAL to the first library's.
Bit-equality here is NOT a race screen: the residual-prefetch epilogue passed it while a missing workgroup barrier let the next
tile's LDS-DMA overwrite other waves' residual buffers (waves run in lock-step in this harness); the full model caught it, and
tests/test_gpu_gemm.py::test_fp32_gate_residual_in_place_many_tiles (short K, in place, repeated) now pins it.
usage: ab_gemm_variants.py libA.so libB.so ...   (build: hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -shared csrc/gemm_w8a8.hip csrc/runtime.hip)"""
import ctypes
import sys

import torch

F16, BF16, F32, I32 = 0, 1, 2, 3
V = ctypes.c_void_p
ARGS = [V, V, V, ctypes.c_int, V, V, ctypes.c_int, V, V, ctypes.c_int, V, ctypes.c_int, V, V, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
        ctypes.c_int, V]
libs = []
for path in sys.argv[1:]:
    lib = ctypes.CDLL(path)
    for f in (lib.wanq_gemm_w8a8, lib.wanq_gemm_w4a8):
        f.argtypes = ARGS
        f.restype = ctypes.c_int
    libs.append((path.split("/")[-1], lib))
st = torch.cuda.current_stream().cuda_stream
dev = "cuda"


def pack_w4(w):  # the wanq_pack_w4 layout for codes in [-8, 7] stored as u = code + 8 (oracle: tests use the library's packer)
    u = (w.to(torch.int16) + 8).to(torch.uint8)
    N, K = u.shape
    u = u.view(N, K // 16, 4, 4)  # 16-code chunk -> 4 groups of 4 bytes: groups 0, 2 are low nibbles, 1, 3 high nibbles
    lo = torch.stack([u[:, :, 0], u[:, :, 2]], 2)
    hi = torch.stack([u[:, :, 1], u[:, :, 3]], 2)
    return (lo | (hi << 4)).reshape(N, K // 2).contiguous()


CASES = [  # (M, N, K, out dtype, epilogue flags, w4)
    (32760, 1536, 1536, BF16, 0, False), (32760, 8960, 1536, BF16, 1, False), (32760, 1536, 8960, F32, 2, False),
    (32760, 1536, 1536, F32, 2, False), (32760, 1536, 1536, I32, 0, False),
    (32760, 1536, 8960, BF16, 0, True), (32760, 8960, 1536, BF16, 1, True),
    (9450, 5120, 5120, BF16, 0, False), (9450, 13824, 5120, BF16, 1, False), (9450, 5120, 13824, F32, 2, False),
    (9450, 5120, 13824, BF16, 0, True), (1000, 1536, 1536, BF16, 0, False),
]
for (M, N, K, od, epi, w4) in CASES:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    if w4:
        wc = torch.randint(-8, 8, (N, K), dtype=torch.int8, device=dev, generator=g)
        w = pack_w4(wc)
    else:
        w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    sa = torch.rand(M, device=dev, generator=g) * 0.01
    asum = a.float().sum(1)
    sw = torch.rand(N, device=dev, generator=g) * 0.01
    zp = torch.randn(N, device=dev, generator=g).round()
    bias = torch.randn(N, device=dev, generator=g)
    gate = torch.randn(N, device=dev, generator=g)
    tdt = {BF16: torch.bfloat16, F32: torch.float32, I32: torch.int32}[od]
    res = torch.randn(M, N, device=dev, generator=g).to(tdt) if epi & 2 else None
    outs = {}

    def run(lib, o):
        f = lib.wanq_gemm_w4a8 if w4 else lib.wanq_gemm_w8a8
        if od == I32:
            rc = f(a.data_ptr(), w.data_ptr(), o.data_ptr(), od, None, None, F32, None, None, F32, None, F32, None, None, 0, M, N, K, st)
        else:
            rc = f(a.data_ptr(), w.data_ptr(), o.data_ptr(), od, sa.data_ptr(), asum.data_ptr(), F32, sw.data_ptr(), bias.data_ptr(), F32,
                   zp.data_ptr(), F32, gate.data_ptr() if epi & 2 else None, res.data_ptr() if epi & 2 else None, epi, M, N, K, st)
        assert rc == 0, rc

    ts = {n: [] for n, _ in libs}
    for n, lib in libs:
        outs[n] = torch.empty(M, N, device=dev, dtype=tdt)
        run(lib, outs[n])
    for _ in range(7):
        for n, lib in libs:
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10):
                run(lib, outs[n])
            e.record()
            torch.cuda.synchronize()
            ts[n].append(s.elapsed_time(e) / 10)
    ops = 2.0 * M * N * K
    base = None
    for n, _ in libs:
        t = sorted(ts[n])
        med = t[len(t) // 2]
        base = base or med
        a0, b0 = outs[n], outs[libs[0][0]]
        same = torch.equal(a0.view(torch.int32 if od != BF16 else torch.int16), b0.view(torch.int32 if od != BF16 else torch.int16))
        print(f"M={M} N={N} K={K} out={['f16','bf16','f32','i32'][od]} epi={epi} w4={int(w4)} {n:24s} median {med * 1e3:7.1f} us  min {t[0] * 1e3:7.1f} us "
              f"{ops / med / 1e9:7.1f} TOP/s  x{base / med:.3f}  bit-equal to first: {same}", flush=True)
