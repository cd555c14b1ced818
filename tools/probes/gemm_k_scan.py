import ctypes, sys, torch
sys.path.insert(0, "tools")
F16, BF16, F32, I32 = 0, 1, 2, 3
V = ctypes.c_void_p
ARGS = [V, V, V, ctypes.c_int, V, V, ctypes.c_int, V, V, ctypes.c_int, V, ctypes.c_int, V, V, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, V]
lib = ctypes.CDLL(sys.argv[1]); lib.wanq_gemm_w8a8.argtypes = ARGS
st = torch.cuda.current_stream().cuda_stream
dev = "cuda"
for N, od, epi in [(1536, BF16, 0), (1536, F32, 2), (8960, BF16, 1)]:
  for K in [128, 256, 512, 1536, 3072, 8960]:
    M = 32760
    g = torch.Generator(device=dev).manual_seed(1)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
    sa = torch.rand(M, device=dev) * 0.01; asum = a.float().sum(1); sw = torch.rand(N, device=dev) * 0.01
    zp = torch.randn(N, device=dev).round(); bias = torch.randn(N, device=dev); gate = torch.randn(N, device=dev)
    tdt = {BF16: torch.bfloat16, F32: torch.float32}[od]
    res = torch.randn(M, N, device=dev).to(tdt)
    o = torch.empty(M, N, device=dev, dtype=tdt)
    def run():
        rc = lib.wanq_gemm_w8a8(a.data_ptr(), w.data_ptr(), o.data_ptr(), od, sa.data_ptr(), asum.data_ptr(), F32, sw.data_ptr(), bias.data_ptr(), F32, zp.data_ptr(), F32, gate.data_ptr() if epi & 2 else None, res.data_ptr() if epi & 2 else None, epi, M, N, K, st)
        assert rc == 0
    for _ in range(5): run()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): run()
        e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) / 10)
    ts.sort()
    tiles = 128 * (N // 256); rounds = tiles / 256
    print(f"N={N} out={od} epi={epi} K={K:5d}: {ts[2]*1e3:7.1f} us  per round {ts[2]*1e3/rounds:6.2f} us  per K-tile-round {ts[2]*1e3/rounds/(K/128):6.3f} us", flush=True)
