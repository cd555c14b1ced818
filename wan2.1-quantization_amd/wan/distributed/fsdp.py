"""`--dit_fsdp`: the DiT blocks' weights sharded over the ranks (ZeRO-3 for inference).

Reference: shard_model() wraps every WanAttentionBlock in torch FSDP with FULL_SHARD, so each rank stores 1/P of every block's
parameters and all-gathers a block's parameters right before its forward (ViDiT-Q/examples/Wan2.1/wan/distributed/fsdp.py:
10-32, applied at wan/text2video.py:106-107).  Here the sharded state is the kernel-mode blocks' INTEGER weights -- int8 codes,
or packed 4-bit nibbles, the only large tensors left after quantisation (scales, zero points, biases and norm weights are a
few KB per layer and stay replicated):

  * every block's weight tensors are flattened into one byte string with a fixed layout (all blocks have the same shapes),
    padded to a multiple of P, and each rank keeps its 1/P slice (`shard`);
  * two full-size byte buffers exist per rank; before block i runs, its slices are all-gathered into buffer i % 2 and the
    block's weight tensors are re-pointed at views of that buffer; the all-gather of block i+1 is issued BEFORE block i's
    compute, on a side stream and on a process group OF ITS OWN (`gather_group`: same ranks as `group`, separate RCCL
    communicator).  ProcessGroupNCCL runs all collectives of one group on one internal stream in issue order, so on the group
    the Ulysses exchange uses, block i's q / k / v all-to-alls would queue behind the 0.35-GB gather issued just before them;
    with its own communicator the gather can fly under the block's 10+ ms of GEMMs and attention (`wait()` only makes the
    compute stream wait for it).  Not yet timed on real multi-rank RCCL (no multi-GPU box from this container);
  * per rank memory for the blocks: total / P + 2 blocks (14B W8: 14 GB -> 1.75 GB + 0.7 GB at P = 8).

Backend agnostic (plain torch.distributed): tested under gloo with world_size 2 on CPU tensors, rehearsed on the one-GPU box."""
import torch
import torch.distributed as dist


def _weight_slots(block):
    """(owner module, attribute) of every sharded tensor of a kernel-mode block, in a fixed order."""
    out = []
    for attn in (block.self_attn, block.cross_attn):
        for l in "qkvo":
            lin = getattr(attn, l)
            if getattr(lin, "weight", None) is not None:
                out.append((lin, "weight"))
    for name in ("ffn0", "ffn2"):
        lin = getattr(block, name)
        if getattr(lin, "weight", None) is not None:
            out.append((lin, "weight"))
    return out


class ShardedBlocks:
    def __init__(self, blocks, group=None, slots_fn=_weight_slots, gather_group=None):
        self.blocks, self.group, self.slots_fn = list(blocks), group, slots_fn
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.P > 1 else 0
        # Which communicator carries the weight gathers.  DEFAULT: the sharding group itself (`group`) -- ProcessGroupNCCL then runs a
        # block's gather and its Ulysses all-to-alls on one stream, in issue order: correct, the gather of block i+1 simply queues
        # behind what was issued before it.  WANQ_FSDP_SHARED_GROUP=0 opts in to a communicator of their own (so that they do not
        # queue behind the all-to-alls of the block being computed): that two-communicator overlap has never run on real
        # multi-rank RCCL from here, so it is not the default until it has.  dist.new_group is
        # collective over WORLD, so the separate group is only created implicitly when `group` IS the world (every rank builds its
        # ShardedBlocks at the same point, as bench.py / quant_generate.py do); a proper subgroup must bring its own `gather_group`.
        import logging
        import os

        if self.P > 1 and gather_group is None and os.environ.get("WANQ_FSDP_SHARED_GROUP", "1") == "0":
            is_world = group is None or group is dist.group.WORLD or dist.get_world_size(group) == dist.get_world_size()
            if is_world:
                gather_group = dist.new_group(dist.get_process_group_ranks(group if group is not None else dist.group.WORLD))
        own = self.P > 1 and gather_group is not None
        self.gather_group = gather_group if own else group
        if self.P > 1 and self.rank == 0:
            logging.getLogger(__name__).info(
                "dit_fsdp: weight gathers on %s", "a communicator of their own (WANQ_FSDP_SHARED_GROUP=0 or gather_group given)" if own
                else "the sharding group (shared with the Ulysses exchange; WANQ_FSDP_SHARED_GROUP=0 selects a separate one)")
        first = slots_fn(self.blocks[0])
        self.layout = []  # (byte offset, nbytes, shape, dtype) per slot
        off = 0
        for owner, attr in first:
            t = getattr(owner, attr)
            nb = t.numel() * t.element_size()
            self.layout.append((off, nb, tuple(t.shape), t.dtype))
            off += (nb + 15) // 16 * 16
        self.full_bytes = (off + 16 * self.P - 1) // (16 * self.P) * (16 * self.P)
        self.shard_bytes = self.full_bytes // self.P
        dev = getattr(first[0][0], first[0][1]).device
        self.shards = []
        for blk in self.blocks:
            slots = slots_fn(blk)
            assert [(tuple(getattr(o, a).shape), getattr(o, a).dtype) for o, a in slots] == [(s, d) for _, _, s, d in self.layout], \
                "all blocks must have identical weight shapes to share one shard layout"
            flat = torch.zeros(self.full_bytes, dtype=torch.uint8, device=dev)
            for (o, a), (boff, nb, _, _) in zip(slots, self.layout):
                flat[boff:boff + nb] = getattr(o, a).contiguous().view(-1).view(torch.uint8)
            self.shards.append(flat[self.rank * self.shard_bytes:(self.rank + 1) * self.shard_bytes].clone())
            for o, a in slots:  # the full copy is dropped: the block now owns no weight storage
                setattr(o, a, torch.empty(0, dtype=getattr(o, a).dtype, device=dev))
        self.buffers = [torch.empty(self.full_bytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._pending = {}

    def bytes_per_rank(self):
        return len(self.blocks) * self.shard_bytes + 2 * self.full_bytes

    # ---- gather ------------------------------------------------------------------------------------
    def prefetch(self, i):
        """Start the all-gather of block i's weights into buffer i % 2 (no-op if already in flight)."""
        if i >= len(self.blocks) or i in self._pending:
            return
        buf = self.buffers[i % 2]
        if self.P == 1:
            buf[:self.shard_bytes].copy_(self.shards[i])
            self._pending[i] = None
            return
        if self.side is not None:
            # the buffer was last read by block i-2: the side stream must not overwrite it before that block's kernels are done
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self._pending[i] = dist.all_gather_into_tensor(buf, self.shards[i], group=self.gather_group, async_op=True)
        else:
            self._pending[i] = dist.all_gather_into_tensor(buf, self.shards[i], group=self.gather_group, async_op=True)

    def materialize(self, i):
        """Block i's weight tensors become views of the gathered buffer (waits for the gather; starts the next one)."""
        self.prefetch(i)
        work = self._pending.pop(i)
        if work is not None:
            work.wait()
            if self.side is not None:
                torch.cuda.current_stream().wait_stream(self.side)
        buf = self.buffers[i % 2]
        for (o, a), (boff, nb, shape, dtype) in zip(self.slots_fn(self.blocks[i]), self.layout):
            setattr(o, a, buf[boff:boff + nb].view(dtype).view(shape))
        # the next block's gather flies under block i's compute (issued before its kernels).  After the last block the next
        # pass's block 0 is fetched (cond -> uncond -> next step), which needs block 0 and block n-1 in different buffers
        n = len(self.blocks)
        if i + 1 < n:
            self.prefetch(i + 1)
        elif n % 2 == 0:
            self.prefetch(0)

    def release(self, i):
        dev = self.buffers[0].device
        for (o, a), (_, _, _, dtype) in zip(self.slots_fn(self.blocks[i]), self.layout):
            setattr(o, a, torch.empty(0, dtype=dtype, device=dev))

    def run(self, fn):
        """for i, blk: materialize -> fn(blk) -> release."""
        for i, blk in enumerate(self.blocks):
            self.materialize(i)
            fn(blk)
            self.release(i)
