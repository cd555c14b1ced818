import ctypes, math, sys, torch
lib = ctypes.CDLL(sys.argv[1])
lib.wanq_attention_fwd.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_void_p]
Lq = Lk = 32760; H = 12
g = torch.Generator(device="cuda").manual_seed(0)
q = torch.randn(Lq, H * 128, device="cuda", generator=g).to(torch.bfloat16)
k = torch.randn(Lk, H * 128, device="cuda", generator=g).to(torch.bfloat16)
v = torch.randn(Lk, H * 128, device="cuda", generator=g).to(torch.bfloat16)
o = torch.empty_like(q)
for _ in range(3):
    lib.wanq_attention_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), 1, Lq, Lk, H, 128, H*128, H*128, H*128, H*128, 1.0/math.sqrt(128), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
