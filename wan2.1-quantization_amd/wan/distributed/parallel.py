"""One-node parallelism for the denoising step: Ulysses sequence parallelism (all-to-all over xGMI through
torch.distributed, backend "nccl" == RCCL on ROCm) x classifier-free-guidance parallelism.

What the reference has: Ulysses/ring sequence parallel through xfuser + yunchang
(wan/distributed/xdit_context_parallel.py:66-192; yunchang.comm.all_to_all.SeqAllToAll4D, not vendored) and
nothing else; cond/uncond passes run back to back on the same ranks (wan/text2video.py:255-258).

What this module does, MI355X-first:
  * Ulysses: every token-local op (LN/quant, all GEMMs, cross-attention, FFN) runs on a [L/P, C] shard; per
    self-attention q, k, v go through ONE all-to-all each (head-scatter / sequence-gather, rank order ==
    sequence order so the per-rank RoPE slice stays valid), attention runs on H/P heads over the full
    sequence, and one all-to-all brings the output back.  The exchanges are pipelined over two head chunks
    (wan/quant_wanx_hip.py): all of them are issued async (they run on the process group's own HIP stream, in issue
    order), chunk 0 of q / k / v flies under the k / v GEMMs, chunk 1 and the way back of chunk 0 under the
    attention of the other chunk; xGMI is point-to-point, so an all-to-all drives all links at once.
  * CFG parallelism: the conditional and unconditional passes of a step are independent until the guidance
    combine, so with an even number of GPUs half of them run each pass (no per-block communication at all)
    and one 2 MB all-gather per step joins them.  1.3B has 12 heads, so Ulysses alone cannot use 8 GPUs
    (12 % 8 != 0, asserted by the reference at fp_generate.py:341-342); cfg 2 x ulysses 4 can.
All layout code is plain torch and backend agnostic: the CPU tests run it under gloo with world_size 2.
"""
import torch
import torch.distributed as dist


class _Done:
    def __init__(self, t):
        self.t = t

    def wait(self):
        return self.t


class _Pending:
    def __init__(self, work, out, post):
        self.work, self.out, self.post = work, out, post

    def wait(self):
        self.work.wait()  # NCCL/RCCL: makes the CURRENT stream wait for the collective's stream (no host block)
        return self.post(self.out)


class SeqParallel:
    """Ulysses helper over one process group (size P).  Tensors are 2-D token-major [tokens, channels]."""

    def __init__(self, group=None):
        self.group = group
        self.size = dist.get_world_size(group) if (dist.is_initialized() and group is not False) else 1
        self.rank = dist.get_rank(group) if self.size > 1 else 0
        # all-to-all accounting (bench.py reports it per step): calls and bytes this rank sends to OTHER ranks
        self.a2a_calls, self.a2a_bytes_sent = 0, 0

    def _count(self, send):
        self.a2a_calls += 1
        self.a2a_bytes_sent += send.numel() * send.element_size() * (self.size - 1) // self.size

    # ---- sequence sharding ------------------------------------------------------------------------
    def padded_len(self, L):
        return -(-L // self.size) * self.size

    def shard_rows(self, x):
        """x [Lpad, ...] (already padded to a multiple of P) -> this rank's contiguous row block."""
        lp = x.shape[0] // self.size
        return x[self.rank * lp:(self.rank + 1) * lp]

    def all_gather_rows(self, x):
        """[Lp, X] on every rank -> [P*Lp, X] in rank order (usp_dit_forward's final all_gather,
        xdit_context_parallel.py:142)."""
        if self.size == 1:
            return x
        out = torch.empty((self.size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous(), group=self.group)
        return out

    # ---- head-scatter / sequence-gather all-to-all and its inverse ---------------------------------
    # `cols=(c0, c1)` restricts the exchange to columns c0..c1 of every rank's head group (a whole number of heads): the block
    # pipelines the exchange of one head chunk under the attention of the previous one.
    def scatter_heads(self, x, async_op=False, cols=None):
        """[Lp, C] (all heads, local tokens) -> [P*Lp, w] (this rank's head group -- or its column slice -- all tokens)."""
        P = self.size
        if P == 1:
            y = x if cols is None else x[:, cols[0]:cols[1]]
            return _Done(y) if async_op else y
        lp, c = x.shape
        send = x.view(lp, P, c // P)
        if cols is not None:
            send = send[:, :, cols[0]:cols[1]]
        w = send.shape[2]
        send = send.transpose(0, 1).contiguous()  # [P, Lp, w]: block r goes to rank r
        recv = torch.empty_like(send)
        self._count(send)
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=async_op)
        post = lambda r: r.view(P * lp, w)  # noqa: E731  blocks arrive in rank order == sequence order
        return _Pending(work, recv, post) if async_op else post(recv)

    # ---- send images written by the producer kernel (no pack pass) ---------------------------------------------------
    def packed_layout(self, lp, c, head_dim, chunks, device):
        """Where every head of a [Lp, C] tensor goes so that each head chunk's all-to-all send image [P, Lp, w] is contiguous:
        returns (numel of the flat buffer, head_map int64 [H, 2] = (element offset of the head's row 0, row stride),
        [(offset, w)] per chunk).  chunks: [(c0, c1)] column ranges inside a rank's head group, covering it."""
        P = self.size
        g = c // P  # columns of one rank's head group
        key = (lp, c, head_dim, tuple(chunks), str(device))
        cache = self.__dict__.setdefault("_layouts", {})
        if key not in cache:
            assert sorted(chunks) == list(chunks) and chunks[0][0] == 0 and chunks[-1][1] == g and \
                all(a[1] == b[0] for a, b in zip(chunks, chunks[1:])), "chunks must tile the head group"
            rows = []
            for h in range(c // head_dim):
                r, cc = divmod(h * head_dim, g)
                c0, c1 = next(ch for ch in chunks if ch[0] <= cc < ch[1])
                w = c1 - c0
                rows.append((P * lp * c0 + r * lp * w + (cc - c0), w))
            cache[key] = (lp * c, torch.tensor(rows, dtype=torch.int64, device=device), [(P * lp * c0, c1 - c0) for c0, c1 in chunks])
        return cache[key]

    def scatter_packed(self, flat, lp, off, w, async_op=False):
        """The all-to-all of one chunk whose send image already sits at flat[off : off + P*Lp*w] as [P, Lp, w]."""
        P = self.size
        send = flat[off:off + P * lp * w].view(P, lp, w)
        recv = torch.empty_like(send)
        self._count(send)
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=async_op)
        post = lambda r: r.view(P * lp, w)  # noqa: E731
        return _Pending(work, recv, post) if async_op else post(recv)

    def gather_heads(self, x, async_op=False, out=None, cols=None):
        """[P*Lp, w] -> [Lp, P*w]: inverse of scatter_heads.  With `out` ([Lp, C]) and `cols` the chunk lands in columns
        c0..c1 of every head group of `out` (which is returned)."""
        P = self.size
        if P == 1:
            if out is None:
                return _Done(x) if async_op else x
            out[:, cols[0]:cols[1]] = x
            return _Done(out) if async_op else out
        l, w = x.shape
        lp = l // P
        send = x.contiguous().view(P, lp, w)  # block r = tokens of rank r
        recv = torch.empty_like(send)
        self._count(send)
        work = dist.all_to_all_single(recv, send, group=self.group, async_op=async_op)
        if out is None:
            post = lambda r: r.transpose(0, 1).reshape(lp, P * w)  # noqa: E731  block s = head group s
        else:
            c0, c1 = cols

            def post(r):
                out.view(lp, P, out.shape[1] // P)[:, :, c0:c1] = r.transpose(0, 1)
                return out
        return _Pending(work, recv, post) if async_op else post(recv)


class ParallelPlan:
    """world = cfg_degree x sp_degree.  Rank r: cfg index r // sp, sp index r % sp (sp ranks are adjacent)."""

    def __init__(self, world, rank, cfg_degree, sp_degree):
        assert cfg_degree * sp_degree == world and cfg_degree in (1, 2)
        self.world, self.rank, self.cfg_degree, self.sp_degree = world, rank, cfg_degree, sp_degree
        self.cfg_index, self.sp_index = rank // sp_degree, rank % sp_degree
        self.sp_group = self.cfg_group = None
        if world > 1:
            for c in range(cfg_degree):  # every rank must create every group, in the same order
                g = dist.new_group([c * sp_degree + s for s in range(sp_degree)])
                if c == self.cfg_index:
                    self.sp_group = g
            for s in range(sp_degree):
                g = dist.new_group([c * sp_degree + s for c in range(cfg_degree)])
                if s == self.sp_index:
                    self.cfg_group = g
        self.sp = SeqParallel(self.sp_group if sp_degree > 1 else False)

    @staticmethod
    def choose(world, num_heads, cfg_parallel=True):
        """Largest CFG degree allowed, then Ulysses over the rest; heads % sp == 0 is required
        (fp_generate.py:341-342)."""
        cfg = 2 if (cfg_parallel and world % 2 == 0) else 1
        sp = world // cfg
        if num_heads % sp != 0:
            cfg, sp = 1, world
        if num_heads % sp != 0:
            raise ValueError(f"num_heads={num_heads} is not divisible by the Ulysses degree {sp} (world {world})")
        return cfg, sp

    def describe(self):
        return f"cfg{self.cfg_degree}xsp{self.sp_degree}"

    def gather_cfg(self, mine):
        """Every rank gets (cond, uncond): one all-gather of the DiT output inside the cfg group."""
        if self.cfg_degree == 1:
            return None
        out = [torch.empty_like(mine) for _ in range(2)]
        dist.all_gather(out, mine.contiguous(), group=self.cfg_group)
        return out[0], out[1]
