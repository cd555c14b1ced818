"""Build the library with the shelved 8-wave variant of the 8960 transform kernel (tools/probes/rotate140_8waves.hip.inc) and
-DWANQ_R140_STAMP into tools/probes/_bin/, run it once at [32760, 8960] and print its per-phase s_memtime stamps (shader-clock
cycles per row).  usage: python tools/probes/r140_stamp_run.py"""
import ctypes, glob, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "wan2.1-quantization_amd")
out = os.path.join(ROOT, "tools", "probes", "_bin")
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libwanq_r140_stamp.so")
import shutil
var = os.path.join(ROOT, "tools", "probes", "rotate140_8waves.hip.inc")
tmp = os.path.join(out, "csrc"); shutil.rmtree(tmp, ignore_errors=True); shutil.copytree(os.path.join(PKG, "csrc"), tmp)
shutil.copy(var, os.path.join(tmp, "rotate140.hip"))
srcs = sorted(glob.glob(os.path.join(tmp, "*.hip")))
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-shared", "-DWANQ_R140_STAMP", "-I", os.path.join(PKG, "csrc"), "-o", lib, *srcs])
L = ctypes.CDLL(lib)
vp, i, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
L.wanq_rotate_quant_rows.argtypes = [vp, i, vp, i, vp, i, vp, vp, vp, i, i64, i, vp]
rows, n = 32760, 8960
x = torch.randn(rows, n, device="cuda").to(torch.bfloat16)
pm = torch.randn(n, device="cuda")
q = torch.empty(rows, n, dtype=torch.int8, device="cuda")
sc, su = torch.zeros(rows, device="cuda"), torch.zeros(rows, device="cuda")
for _ in range(2):
    rc = L.wanq_rotate_quant_rows(x.data_ptr(), 1, pm.data_ptr(), 140, None, 2, q.data_ptr(), sc.data_ptr(), su.data_ptr(), 2, rows, n,
                                  torch.cuda.current_stream().cuda_stream)
    assert rc == 0
torch.cuda.synchronize()
