"""The REAL RCCL backend (torch.distributed "nccl") on the one GPU of the box, world size 1: every collective call FORM the
multi-GPU path uses (wan/distributed/parallel.py, fsdp.py, bench.py) with its dtypes, shapes and async semantics.  Two ranks
cannot share a device under RCCL, so this is as far as a one-GPU box can take the product's own backend; the multi-rank layout
logic is covered under gloo (tests/test_distributed_cpu.py) and by the host-staged rehearsals."""
import os
import sys

import torch
import torch.distributed as dist

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{sys.argv[1]}", rank=0, world_size=1, device_id=dev)
except Exception as e:  # the box cannot bring RCCL up at all (environment): reported as a skip, not as a parity failure
    print(f"rccl_init_failed: {e!r}")
    sys.exit(3)
assert dist.get_backend() == "nccl"
g = torch.Generator(device=dev).manual_seed(0)
side = torch.cuda.Stream()
ok = 0
# head-scatter all-to-all images [P, Lp, w]: bf16 q / k / v, int8 codes + fp32 scale planes of the int8 Q.K^T form
for dt, shape in [(torch.bfloat16, (1, 4095, 768)), (torch.int8, (1, 4095, 768)), (torch.float32, (1, 6, 4095))]:
    send = (torch.randn(shape, device=dev, generator=g) * 50).to(dt)
    recv = torch.empty_like(send)
    work = dist.all_to_all_single(recv, send, async_op=True)   # runs on the process group's stream
    busy = torch.randn(2048, 2048, device=dev) @ torch.randn(2048, 2048, device=dev)  # compute keeps going underneath
    work.wait()                                                # current stream waits for the collective
    assert torch.equal(recv, send), dt
    ok += 1
# the send image a producer kernel wrote into a slice of a flat buffer (scatter_packed)
flat = torch.randn(3 * 4095 * 256, device=dev, generator=g).to(torch.bfloat16)
send = flat[4095 * 256:2 * 4095 * 256].view(1, 4095, 256)
recv = torch.empty_like(send)
dist.all_to_all_single(recv, send)
assert torch.equal(recv, send)
ok += 1
# final all-gather of the head output, the CFG all-gather, the fsdp gather of a block's byte string on a side stream
x = torch.randn(4095, 16, device=dev, generator=g)
out = torch.empty(4095, 16, device=dev)
dist.all_gather_into_tensor(out, x.contiguous())
assert torch.equal(out, x)
w = torch.randint(0, 256, (1 << 20,), device=dev, dtype=torch.uint8, generator=g)
full = torch.empty_like(w)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    dist.all_gather_into_tensor(full, w)
torch.cuda.current_stream().wait_stream(side)
assert torch.equal(full, w)
ok += 2
# ParallelPlan: sub-groups (new_group on a device-bound communicator) and the CFG all-gather in its list form
grp = dist.new_group([0])
lst = [torch.empty_like(x) for _ in range(1)]
dist.all_gather(lst, x.contiguous(), group=grp)
assert torch.equal(lst[0], x)
recv = torch.empty(1, 64, 128, device=dev, dtype=torch.bfloat16)
send = torch.randn(1, 64, 128, device=dev, generator=g).to(torch.bfloat16)
dist.all_to_all_single(recv, send, group=grp, async_op=True).wait()
assert torch.equal(recv, send)
ok += 2
# bench.py: barrier, max-over-ranks of the step time on the device
dist.barrier()
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert t.item() == 1.25
ok += 1
torch.cuda.synchronize()
dist.destroy_process_group()
print(f"rccl_single_rank ok={ok}")
