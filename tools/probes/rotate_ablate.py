"""Timing ablations of the q / k / v transform kernel (rotate_kernel<12,1,4,LN,MULTI>, three outputs from one pass over x at
[32760, 1536]): the library is built four times with parts of csrc/rotate.hip compiled out (-DWANQ_ROT_ABLATE_*; the outputs of
those builds are wrong by design) and the same call is timed on each.  usage: python tools/probes/rotate_ablate.py"""
import ctypes, glob, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "wan2.1-quantization_amd")
out = os.path.join(ROOT, "tools", "probes", "_bin")
os.makedirs(out, exist_ok=True)
srcs = sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))
variants = [("full", []), ("no lane-exchange stages", ["-DWANQ_ROT_ABLATE_LANE"]), ("no Paley mix", ["-DWANQ_ROT_ABLATE_MIX"]),
            ("cheap quantise (no exact-division check)", ["-DWANQ_ROT_ABLATE_QUANT"]),
            ("none of the three", ["-DWANQ_ROT_ABLATE_LANE", "-DWANQ_ROT_ABLATE_MIX", "-DWANQ_ROT_ABLATE_QUANT"])]
L, C = 32760, 1536
x = torch.randn(L, C, device="cuda")
sh, sc = torch.randn(1, C, device="cuda") * 0.1, torch.randn(1, C, device="cuda") * 0.1
pms = [torch.randn(C, device="cuda") for _ in range(3)]
qs = [torch.empty(L, C, dtype=torch.int8, device="cuda") for _ in range(3)]
ss = [torch.zeros(L, device="cuda") for _ in range(3)]
us = [torch.zeros(L, device="cuda") for _ in range(3)]
vp, i, i64, f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
arr = lambda ts: (ctypes.c_void_p * 3)(*[t.data_ptr() for t in ts])
for k, (name, flags) in enumerate(variants):
    lib = os.path.join(out, f"libwanq_rot{k}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-DWANQ_ALLOW_ABLATIONS", *flags, "-o", lib, *srcs])
    Lb = ctypes.CDLL(lib)
    fn = Lb.wanq_layernorm_rotate_quant_rows_multi
    fn.argtypes = [vp, i, vp, vp, vp, i, i64, i64, f, i, vp, i, vp, vp, vp, i, i64, i, vp]
    call = lambda: fn(x.data_ptr(), 2, None, sh.data_ptr(), sc.data_ptr(), 2, C, L, 1e-6, 3, arr(pms), 12, arr(qs), arr(ss), arr(us), 2, L, C,
                      torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        call()
    e.record(); torch.cuda.synchronize()
    print(f"{name:46s} {s.elapsed_time(e) / 20 * 1e3:8.1f} us")
