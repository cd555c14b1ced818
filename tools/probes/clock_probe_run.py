"""Shader clock held inside the attention and the persistent GEMM kernels (diagnostic build -DWANQ_CLOCK_PROBE: one workgroup
reads clock64 / wall_clock64 around its main loop; the library prints one [clock] line per launch).
usage: clock_probe_run.py lib_clock.so   (build: hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -shared -DWANQ_CLOCK_PROBE csrc/*.hip)"""
import ctypes, math, sys, torch
lib = ctypes.CDLL(sys.argv[1])
lib.wanq_attention_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_void_p]
lib.wanq_gemm_w8a8.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_int] + [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
Lq = Lk = 32760; H = 12
q = torch.randn(Lq, H * 128, device="cuda", generator=g).to(torch.bfloat16)
k = torch.randn(Lk, H * 128, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(Lk, H * 128, device="cuda", generator=g).to(torch.bfloat16)
o = torch.empty_like(q)
print("--- attention, 12 back-to-back launches (the library synchronises and prints after each)", flush=True)
for _ in range(12):
    lib.wanq_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1, Lq, Lk, H, 128, H*128, H*128, H*128, H*128, 1.0/math.sqrt(128), st)
torch.cuda.synchronize()
for (M, N, K) in [(32760, 1536, 1536), (32760, 8960, 1536), (32760, 1536, 8960)]:
    a = torch.randint(-127, 128, (M, K), device="cuda", dtype=torch.int8, generator=g)
    w = torch.randint(-127, 128, (N, K), device="cuda", dtype=torch.int8, generator=g)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    sa = torch.rand(M, device="cuda", generator=g) * 0.01
    asum = a.float().sum(1)
    sw = torch.rand(N, device="cuda", generator=g) * 0.01
    zp = torch.zeros(N, device="cuda")
    print(f"--- gemm {M}x{N}x{K}", flush=True)
    for _ in range(12):
        rc = lib.wanq_gemm_w8a8(a.data_ptr(), w.data_ptr(), out.data_ptr(), 1, sa.data_ptr(), asum.data_ptr(), 2, sw.data_ptr(), None, 2, zp.data_ptr(), 2, None, None, 0, M, N, K, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
