"""The N>1 path on CPU: world_size 2 and 4 under gloo.  Covers the Ulysses all-to-all layout (forward o inverse
= identity; sequence order == rank order), Ulysses attention == full attention, the row sharding / all-gather,
and the cfg x sp process-group plan.  The layout code is the same code the GPU path runs over RCCL."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "wan2.1-quantization_amd")


def _init(rank, world, port):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)


def _ulysses_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from oracle import wan_ref as wr
        from wan.distributed.parallel import ParallelPlan, SeqParallel

        sp = SeqParallel(None)
        assert sp.size == world and sp.rank == rank
        H, d, L = max(4, world), 8, 10 * world
        g = torch.Generator().manual_seed(0)
        full = [torch.randn(L, H * d, generator=g) for _ in range(3)]
        lp = L // world
        loc = [sp.shard_rows(t).contiguous() for t in full]
        # identity: gather_heads(scatter_heads(x)) == x, sync and async forms
        s = sp.scatter_heads(loc[0])
        assert s.shape == (L, H * d // world)
        # rank r holds head group r of ALL tokens, in sequence order
        assert torch.equal(s, full[0].view(L, world, -1)[:, rank])
        assert torch.equal(sp.gather_heads(s), loc[0])
        assert torch.equal(sp.gather_heads(sp.scatter_heads(loc[1], async_op=True).wait(), async_op=True).wait(), loc[1])
        # column-chunked exchange (pipelined Ulysses): two chunks of every rank's head group, reassembled through `out`
        cw = loc[2].shape[1] // world
        h0 = cw // 2
        back = torch.full_like(loc[2], float("nan"))
        whole = sp.scatter_heads(loc[2])
        for c0, c1 in ((0, h0), (h0, cw)):
            part = sp.scatter_heads(loc[2], async_op=True, cols=(c0, c1)).wait()
            assert torch.equal(part, whole[:, c0:c1])
            sp.gather_heads(part, async_op=True, out=back, cols=(c0, c1)).wait()
        assert torch.equal(back, loc[2])
        # Ulysses attention == full attention (key padding masked by k_len)
        k_len = L - 3
        qs, ks, vs = (sp.scatter_heads(t) for t in loc)
        hp = H // world
        o = wr.attention(qs.view(L, hp, d), ks.view(L, hp, d), vs.view(L, hp, d), k_len).reshape(L, hp * d)
        o_loc = sp.gather_heads(o)
        ref = wr.attention(full[0].view(L, H, d), full[1].view(L, H, d), full[2].view(L, H, d), k_len).reshape(L, H * d)
        torch.testing.assert_close(o_loc, ref[rank * lp:(rank + 1) * lp], rtol=1e-5, atol=1e-6)
        # final all-gather restores the sequence
        assert torch.equal(sp.all_gather_rows(loc[2]), full[2])
        assert sp.padded_len(L + 1) == L + world
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _plan_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from wan.distributed.parallel import ParallelPlan

        cfg, sp = ParallelPlan.choose(world, num_heads=12)
        assert (cfg, sp) == ((2, world // 2) if world % 2 == 0 else (1, world))
        plan = ParallelPlan(world, rank, cfg, sp)
        assert plan.sp.size == sp and plan.cfg_index == rank // sp and plan.sp_index == rank % sp
        mine = torch.full((2, 3), float(plan.cfg_index * 10 + plan.sp_index))
        cond, uncond = plan.gather_cfg(mine)
        assert torch.all(cond == plan.sp_index) and torch.all(uncond == 10 + plan.sp_index)
        if sp > 1:  # the Ulysses group only spans ranks of the same cfg index
            x = torch.full((2 * sp, 4 * sp), float(rank))
            s = plan.sp.scatter_heads(x)
            expect = torch.cat([torch.full((2 * sp, 4), float(plan.cfg_index * sp + r)) for r in range(sp)])
            assert torch.equal(s, expect)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _run(worker, world, port):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_ulysses_layout_and_attention_world2():
    _run(_ulysses_worker, 2, 29611)


def test_ulysses_layout_and_attention_world4():
    _run(_ulysses_worker, 4, 29612)


def test_ulysses_layout_and_attention_world8():
    """BASELINE config 4's Ulysses degree (14B: 40 heads over 8 ranks), on the layout code alone."""
    _run(_ulysses_worker, 8, 29615)


def test_cfg_x_sp_plan_world2():
    _run(_plan_worker, 2, 29613)


def test_cfg_x_sp_plan_world4():
    _run(_plan_worker, 4, 29614)


def test_plan_choice_rules():
    from wan.distributed.parallel import ParallelPlan

    assert ParallelPlan.choose(1, 12) == (1, 1)
    assert ParallelPlan.choose(2, 12) == (2, 1)
    assert ParallelPlan.choose(4, 12) == (2, 2)
    assert ParallelPlan.choose(8, 12) == (2, 4)        # 12 heads: pure Ulysses-8 is impossible
    assert ParallelPlan.choose(8, 40, cfg_parallel=False) == (1, 8)  # 14B, the reference's ulysses_size=8
    assert ParallelPlan.choose(8, 40) == (2, 4)
    with pytest.raises(ValueError):
        ParallelPlan.choose(8, 12, cfg_parallel=False)


def test_head_chunks_and_split_kv_policy():
    """Host policy of the pipelined Ulysses exchange and of split-KV attention (no GPU): chunks tile the rank's heads, are
    whole rounds of the 256 CUs where a head is 128 workgroups, and split-KV turns on exactly for half-empty last rounds."""
    from wan import ops
    from wan.quant_wanx_hip import _head_chunks

    cpu = torch.device("cpu")  # _head_chunks assumes 256 CUs off-GPU
    for heads, L in [(6, 32760), (3, 32760), (12, 32760), (5, 75600), (20, 75600), (10, 75600), (2, 192), (1, 192), (6, 4680)]:
        ch = _head_chunks(heads, L, cpu)
        assert ch[0][0] == 0 and ch[-1][1] == heads and all(a[1] == b[0] and a[1] > a[0] for a, b in zip(ch, ch[1:])) and len(ch) <= 4
    assert _head_chunks(6, 32760, cpu) == [(0, 2), (2, 4), (4, 6)]   # never 3 + 3: two half-empty rounds each
    assert _head_chunks(3, 32760, cpu) == [(0, 2), (2, 3)]
    assert _head_chunks(6, 4680, cpu) == [(0, 6)]                    # 6 heads x 19 workgroups do not even fill one round
    assert ops.attention_splits(32760, 32760, 12, cpu, ncu=256) == 1
    assert ops.attention_splits(32760, 32760, 6, cpu, ncu=256) == 1
    assert ops.attention_splits(32760, 32760, 3, cpu, ncu=256) == 2
    assert ops.attention_splits(32760, 32760, 2, cpu, ncu=256) == 1  # exactly one full round
    assert ops.attention_splits(32760, 32760, 1, cpu, ncu=256) >= 2
    assert ops.attention_splits(32760, 512, 12, cpu, ncu=256) == 1


# ---------------------------------------------------------------------------------------------------------------------
# --dit_fsdp: block weights sharded over the ranks, all-gathered one block ahead (wan/distributed/fsdp.py)
class _ToyLin:
    def __init__(self, w):
        self.weight = w


class _ToyAttn:
    def __init__(self, g, dim, dtype):
        for l in "qkvo":
            setattr(self, l, _ToyLin(_toy_w(g, dim, dim, dtype)))


def _toy_w(g, n, k, dtype):
    import torch
    if dtype == torch.uint8:
        return torch.randint(0, 256, (n, k // 2), generator=g, dtype=torch.uint8)
    return torch.randint(-128, 128, (n, k), generator=g, dtype=torch.int8)


class _ToyBlock:
    """The attribute layout of WanAttentionBlockWithHipKernel that ShardedBlocks walks (self_attn / cross_attn q,k,v,o; ffn0/2)."""

    def __init__(self, seed, dim=48, ffn=80):
        import torch
        g = torch.Generator().manual_seed(seed)
        self.self_attn, self.cross_attn = _ToyAttn(g, dim, torch.int8), _ToyAttn(g, dim, torch.int8)
        self.ffn0, self.ffn2 = _ToyLin(_toy_w(g, ffn, dim, torch.uint8)), _ToyLin(_toy_w(g, dim, ffn, torch.uint8))  # packed W4

    def checksum(self):
        import torch
        tot = torch.zeros((), dtype=torch.int64)
        for a in (self.self_attn, self.cross_attn):
            for l in "qkvo":
                tot += getattr(a, l).weight.to(torch.int64).sum() * 3 + getattr(a, l).weight[0, 1].to(torch.int64)
        return int(tot + self.ffn0.weight.to(torch.int64).sum() * 5 + self.ffn2.weight.to(torch.int64)[1, 2])


def _fsdp_worker(rank, world, port, q):
    import traceback
    try:
        import torch
        _init(rank, world, port)
        from wan.distributed.fsdp import ShardedBlocks

        blocks = [_ToyBlock(s) for s in range(int(os.environ.get('WANQ_TOY_BLOCKS', '5')))]
        want = [b.checksum() for b in blocks]
        full = sum(getattr(a, l).weight.numel() for a in (blocks[0].self_attn, blocks[0].cross_attn) for l in "qkvo") + \
            blocks[0].ffn0.weight.numel() + blocks[0].ffn2.weight.numel()
        sh = ShardedBlocks(blocks, None)
        assert sh.P == world and sh.full_bytes >= full and sh.full_bytes % (16 * world) == 0
        assert all(s.numel() == sh.full_bytes // world for s in sh.shards)
        assert blocks[2].ffn0.weight.numel() == 0  # the block gave up its storage
        assert sh.bytes_per_rank() < len(blocks) * full / world + 2 * sh.full_bytes + 1
        got = []
        for _ in range(2):  # two passes (cond / uncond) reuse the double buffer
            got = []
            sh.run(lambda b: got.append(b.checksum()))
            assert got == want, (rank, got, want)
            assert blocks[-1].self_attn.q.weight.numel() == 0  # released after use
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        q.put((rank, traceback.format_exc()))


@pytest.mark.parametrize("n_blocks", [5, 4])  # even counts also prefetch the next pass's block 0 behind the last block
def test_dit_fsdp_shard_gather_roundtrip_world2(n_blocks, monkeypatch):
    monkeypatch.setenv("WANQ_TOY_BLOCKS", str(n_blocks))
    _run(_fsdp_worker, 2, 29671 + n_blocks)


# ---------------------------------------------------------------------------------------------------------------------
# Ulysses send images written in place by the producer (SeqParallel.packed_layout / scatter_packed)
def _packed_worker(rank, world, port, q):
    import traceback
    try:
        import torch
        _init(rank, world, port)
        from wan.distributed.parallel import SeqParallel

        sp = SeqParallel(None)
        lp, heads, d = 5, 8, 4  # 8 heads over `world` ranks, head chunks of 1 + 2 + 1 heads (world 2)
        c = heads * d
        g = c // world
        hp = heads // world
        chunks = [(0, d), (d, (hp - 1) * d), ((hp - 1) * d, g)] if hp >= 3 else [(0, g)]
        x = (torch.arange(lp * c, dtype=torch.float32).view(lp, c) + 1000 * rank)
        numel, hmap, where = sp.packed_layout(lp, c, d, chunks, "cpu")
        assert numel == lp * c and hmap.shape == (heads, 2) and len(where) == len(chunks)
        # what wanq_rmsnorm_rope_scatter's store does, in index form
        flat = torch.full((numel,), float("nan"))
        for h in range(heads):
            for r in range(lp):
                o = int(hmap[h, 0]) + r * int(hmap[h, 1])
                flat[o:o + d] = x[r, h * d:(h + 1) * d]
        assert not torch.isnan(flat).any()  # the images tile the buffer exactly
        for (c0, c1), (off, w) in zip(chunks, where):
            a = sp.scatter_packed(flat, lp, off, w)
            b = sp.scatter_heads(x, cols=(c0, c1))
            assert torch.equal(a, b), (rank, c0, c1)
            pa = sp.scatter_packed(flat, lp, off, w, async_op=True).wait()
            assert torch.equal(pa, b)
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        q.put((rank, traceback.format_exc()))


def test_packed_send_images_equal_transpose_pack_world2():
    _run(_packed_worker, 2, 29691)


def _q8_exchange_worker(rank, world, port, q):
    """The int8 Q.K^T exchange of wan/quant_wanx_hip.py::_self_attention_qk8_ulysses in index form: per-(token, head) codes
    and the two scale planes, quantised on the token shard, must arrive as exactly the head group's slice of what ONE rank
    holding all tokens would have produced (codes, both planes, the 64-row zero padding of the key planes)."""
    import traceback
    try:
        import torch
        _init(rank, world, port)
        from wan import ops
        from wan.distributed.parallel import SeqParallel

        sp = SeqParallel(None)
        d, H, lp = 128, 2 * world, 7
        L, C = lp * world, 2 * world * 128
        g = torch.Generator().manual_seed(3)
        codes_full = torch.randint(-127, 128, (L, C), generator=g, dtype=torch.int8)
        delta_full = torch.rand(H, L, generator=g) + 0.01
        planes_full = torch.stack([delta_full, -12582912.0 * delta_full])  # [2, H, L] as wanq_rmsnorm_rope_q8 writes them
        # this rank's shard, in the Q8Rows layout of the producer (stride padded to 64 for keys)
        loc = ops.Q8Rows.__new__(ops.Q8Rows)
        loc.rows, loc.cols, loc.heads, loc.stride = lp, C, H, 64
        loc.codes = codes_full[rank * lp:(rank + 1) * lp].contiguous()
        loc.scales = torch.zeros(2, H, 64)
        loc.scales[:, :, :lp] = planes_full[:, :, rank * lp:(rank + 1) * lp]
        planes = loc.scales[:, :, :lp].permute(2, 1, 0).reshape(lp, 2 * H)
        hp = H // world
        for ch in ((0, d), (d, hp * d)):  # two head chunks of the rank's head group
            cw = sp.scatter_heads(loc.codes, async_op=True, cols=ch)
            pw = sp.scatter_heads(planes, async_op=True, cols=(2 * ch[0] // d, 2 * ch[1] // d))
            for for_keys in (False, True):
                got = ops.Q8Rows.from_exchange(cw.wait(), pw.wait(), d, for_keys)
                h0, h1 = rank * hp + ch[0] // d, rank * hp + ch[1] // d
                assert torch.equal(got.codes, codes_full[:, h0 * d:h1 * d])
                assert got.stride == (64 if for_keys else L) and got.heads == h1 - h0
                assert torch.equal(got.scales[:, :, :L], planes_full[:, h0:h1]) and not bool(got.scales[:, :, L:].any())
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        q.put((rank, traceback.format_exc()))


def test_int8_qk_exchange_layout_world2():
    _run(_q8_exchange_worker, 2, 29693)


# ---------------------------------------------------------------------------------------------------------------------------
# Ulysses for the FP model and the calibration reduction (the reference patches usp_attn_forward / usp_dit_forward onto the FP
# model, W/wan/text2video.py:89-100, and get_calib_data_wanx.py runs under it): WanModel.forward(..., sp) under gloo.  The
# attention core is the HIP kernel in the product; here -- CPU, test only -- wan.ops.attention is replaced by the oracle's fp32
# softmax definition, everything else (sharding, per-rank rotary slice, head exchange, final all-gather, hooks) is product code.
def _fp_ulysses_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from oracle import wan_ref as wr
        from wan import calib, ops
        from wan.configs import seq_len_for
        from wan.distributed.parallel import SeqParallel
        from wan.modules.model import WanModel

        def attention_ref(q_, k_, v_, num_heads, k_len=None, **kw):
            d = q_.shape[1] // num_heads
            o = wr.attention(q_.float().view(-1, num_heads, d), k_.float().view(-1, num_heads, d), v_.float().view(-1, num_heads, d), k_len)
            return o.reshape(q_.shape[0], -1)  # fp32: no autocast on the CPU for the o projection behind it

        ops.attention = attention_ref
        calib.fused.col_absmax_ = lambda running, x: running.copy_(torch.maximum(running, x.float().abs().amax(0)))
        torch.manual_seed(0)
        heads = 4
        model = WanModel(dim=64, ffn_dim=128, num_heads=heads, num_layers=2, text_dim=32, freq_dim=32).eval()
        g = torch.Generator().manual_seed(1)
        torch.nn.init.xavier_uniform_(model.head.head.weight, generator=g)
        shape = (16, 3, 6, 10)  # 3 * 3 * 5 = 45 tokens: padded to a multiple of the world size, odd per-rank tails
        latent = torch.randn(shape, generator=g)
        ctx = torch.randn(7, 32, generator=g) * 0.1
        t = torch.tensor([321])
        sp = SeqParallel(None)
        sl = seq_len_for(shape, sp_size=world)
        assert sl % world == 0 and sl >= 45
        with torch.no_grad():
            ref = model([latent], t, [ctx], sl)[0]
            hooks_ref = calib.add_hooks(model)
            model([latent], t, [ctx], sl)
            want = {n: h.running.clone() for n, h in hooks_ref.items()}
            for h in hooks_ref.values():
                h.hook_handle.remove()
            hooks = calib.add_hooks(model)
            out = model([latent], t, [ctx], sl, sp)[0]
        torch.testing.assert_close(out, ref, rtol=2e-4, atol=2e-5)
        got = calib.gather_and_save_activation(hooks)  # MAX all-reduce over the ranks' token shards
        assert set(got) == set(want)
        for n in want:
            # padded rows are zeros in front of the first block only; behind it they carry bias terms on BOTH paths
            torch.testing.assert_close(got[n][0], want[n], rtol=2e-4, atol=2e-5, msg=n)
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_fp_model_and_calibration_under_ulysses_world2():
    _run(_fp_ulysses_worker, 2, 29641)


def test_fp_model_and_calibration_under_ulysses_world4():
    _run(_fp_ulysses_worker, 4, 29642)


# ---------------------------------------------------------------------------------------------------------------------------
# BASELINE configs 4 and 5 at world size 8 (gloo): the EXACT plan objects `bench.py --preset 14B-ulysses` / `--preset 14B-w4a8-fsdp`
# build -- Ulysses degree 8 on 40 heads, 9450 tokens and 5 heads per rank, the head chunks of the pipelined exchange and their
# packed send images; the --dit_fsdp shard layout of a block with the 14B weight shapes (FFN weights packed 4-bit) -- so that the
# first real 8-GPU run meets no layout it has not seen.
def _preset_ulysses_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from wan.configs import SIZE_CONFIGS, WAN_CONFIGS, latent_shape, seq_len_for
        from wan.distributed.parallel import ParallelPlan
        from wan.quant_wanx_hip import _head_chunks

        cfg = WAN_CONFIGS["t2v-14B"]
        H, C = cfg["num_heads"], cfg["dim"]
        d = C // H
        plan = ParallelPlan(world, rank, *ParallelPlan.choose(world, H, False))  # bench.py: --no-cfg-parallel under the preset
        assert plan.describe() == "cfg1xsp8" and plan.sp.size == 8 and plan.sp.rank == rank and plan.cfg_degree == 1
        shape = latent_shape(SIZE_CONFIGS["1280*720"], 81)
        assert tuple(shape) == (16, 21, 90, 160)
        seq_len = seq_len_for(shape, sp_size=plan.sp_degree)
        lp = seq_len // world
        assert seq_len == 75600 and lp == 9450
        chunks = _head_chunks(H // world, seq_len, torch.device("cpu"))
        assert chunks == [(0, 1), (1, 2), (2, 4), (4, 5)]  # a head is 296 workgroups > 256 CUs: units of one head, 1 + 1 + 2 + 1
        cols = [(a * d, b * d) for a, b in chunks]
        numel, hmap, where = plan.sp.packed_layout(lp, C, d, cols, torch.device("cpu"))
        assert numel == lp * C and hmap.shape == (H, 2)
        assert where == [(world * lp * c0, c1 - c0) for c0, c1 in cols]
        # head h of the local [lp, C] tensor: destination rank h // 5, column (h % 5) * d inside that rank's head group
        for h in (0, 4, 5, 17, 39):
            r, cc = divmod(h * d, C // world)
            c0, c1 = next(ch for ch in cols if ch[0] <= cc < ch[1])
            assert hmap[h].tolist() == [world * lp * c0 + r * lp * (c1 - c0) + (cc - c0), c1 - c0]
        # one exchange at the preset's per-rank size through the packed images (int8 stand-in for the bf16 payload: the layout
        # code never looks at the dtype), against the transpose-pack form chunk by chunk
        g = torch.Generator().manual_seed(rank)
        x = torch.randint(-128, 128, (lp, C), generator=g, dtype=torch.int8)
        flat = torch.empty(numel, dtype=torch.int8)
        for h in range(H):  # what rmsnorm_rope_scatter writes: head h's [lp, d] block at (offset, row stride)
            off, stride = hmap[h].tolist()
            flat.as_strided((lp, d), (stride, 1), off).copy_(x[:, h * d:(h + 1) * d])
        for (c0, c1), (off, w) in zip(cols, where):
            got = plan.sp.scatter_packed(flat, lp, off, w)
            want = plan.sp.scatter_heads(x, cols=(c0, c1))
            assert got.shape == (seq_len, c1 - c0) and torch.equal(got, want)
        assert plan.sp.a2a_calls == 8 and plan.sp.a2a_bytes_sent == 2 * lp * (C // world) * world * 7 // 8
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_preset_14b_ulysses_plan_world8():
    _run(_preset_ulysses_worker, 8, 29651)


class _Preset14BBlock:
    """Weight slots of one kernel-mode 14B block under quant_configs/w4a8_mixed.yaml: attention projections int8 [5120, 5120], FFN
    weights packed 4-bit (uint8 [N, K / 2])."""

    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        C, F = 5120, 13824

        def w(n, k, packed):
            t = torch.empty(n, k // 2 if packed else k, dtype=torch.uint8 if packed else torch.int8)
            t.view(torch.uint8)[::997, ::13] = torch.randint(0, 256, t[::997, ::13].shape, generator=g, dtype=torch.uint8)  # sparse marks
            return t

        self.self_attn, self.cross_attn = _ToyAttn.__new__(_ToyAttn), _ToyAttn.__new__(_ToyAttn)
        for a in (self.self_attn, self.cross_attn):
            for l in "qkvo":
                setattr(a, l, _ToyLin(w(C, C, False)))
        self.ffn0, self.ffn2 = _ToyLin(w(F, C, True)), _ToyLin(w(C, F, True))

    def marks(self):
        out = []
        for a in (self.self_attn, self.cross_attn):
            for l in "qkvo":
                out.append(getattr(a, l).weight.view(torch.uint8)[::997, ::13].clone())
        return out + [self.ffn0.weight[::997, ::13].clone(), self.ffn2.weight[::997, ::13].clone()]


def _preset_fsdp_worker(rank, world, port, q):
    try:
        _init(rank, world, port)
        from wan.distributed.fsdp import ShardedBlocks

        blk = _Preset14BBlock(0)  # every rank builds the same block (as every rank loads the same checkpoint)
        want = blk.marks()
        C, F = 5120, 13824
        sizes = [C * C] * 8 + [F * C // 2, C * F // 2]
        sh = ShardedBlocks([blk], None)
        assert sh.P == 8 and [nb for _, nb, _, _ in sh.layout] == sizes
        offs = [o for o, _, _, _ in sh.layout]
        assert offs == [sum(sizes[:i]) for i in range(10)]  # every slot is a multiple of 16 B: no padding between slots
        assert sh.full_bytes == sum(sizes) and sh.full_bytes % (16 * 8) == 0 and sh.shard_bytes == sh.full_bytes // 8 == 35061760
        assert sh.bytes_per_rank() == sh.shard_bytes + 2 * sh.full_bytes
        assert blk.ffn0.weight.numel() == 0
        got = []
        sh.run(lambda b: got.extend(b.marks()))  # one all-gather of 280 MB over the 8 ranks, views handed back to the block
        assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
        assert blk.self_attn.q.weight.numel() == 0  # released after use
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, traceback.format_exc()))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_preset_14b_w4a8_fsdp_shard_layout_world8():
    _run(_preset_fsdp_worker, 8, 29652)
