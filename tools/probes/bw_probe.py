import torch
L,F=32760,8960
hb=torch.randn(L,F,device="cuda").to(torch.bfloat16)
out8=torch.empty(L,F,dtype=torch.int8,device="cuda")
outb=torch.empty_like(hb)
def timeit(fn,iters=10):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/iters*1e3
t=timeit(lambda: torch.ops.aten.copy_(out8, hb)); print(f"torch bf16->int8 cast copy: {t:.1f} us  {(L*F*3)/t/1e6:.2f} TB/s")
t=timeit(lambda: outb.copy_(hb)); print(f"torch bf16 copy (1:1): {t:.1f} us  {(L*F*4)/t/1e6:.2f} TB/s")
x32=torch.randn(L,1536*2,device="cuda"); y32=torch.empty_like(x32)
t=timeit(lambda: y32.copy_(x32)); print(f"torch fp32 copy 402 MB x2: {t:.1f} us  {(L*3072*8)/t/1e6:.2f} TB/s")
t=timeit(lambda: hb.abs().amax()); print(f"torch read-only abs-amax 587 MB (2 kernels): {t:.1f} us")
