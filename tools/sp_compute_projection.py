#!/usr/bin/env python3
"""Per-rank COMPUTE time of the cfg-B denoising step under cfg x Ulysses parallelism, measured on ONE GPU -- a projection of the
1 / 2 / 4 / 8-GPU rows, never a measurement of them.

One kernel-mode block of the headline workload (C = 1536, 12 heads, F = 8960, quant_configs/w8a8_all_linears.yaml: every Linear W8A8,
ViDiT-Q on q / k / v) runs exactly the code path a rank runs under sequence parallelism (WanAttentionBlockWithHipKernel.forward
with an `sp` object): token shard of L / P rows for every row-wise kernel and GEMM, q / k / v written into the exchange's send
images, attention over ALL tokens for this rank's H / P heads in the pipelined head chunks -- with the all-to-alls replaced by
the local memory moves of `LocalSeqParallel` (same shapes, same pack / unpack copies, no communication).  What comes out is the
compute a rank has to do; the exchange is priced beside it from its byte count and the xGMI link rate
(/opt/skills/guides/MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, point to point):
    bytes leaving a GPU per block-pass = 4 tensors (q, k, v out, o back) x (L / P) x C x 2 B x (P - 1) / P, spread over P - 1 links.
The projected step is  passes_per_rank x 30 blocks x t_block + the non-block remainder of the measured N = 1 step; "hidden" assumes
the pipelined exchange hides entirely under the attention of the other head chunk, "exposed" that none of it does and that the
links deliver half their nominal rate.

usage: python tools/sp_compute_projection.py [--step-ms-n1 407]    (the N = 1 step time of the same box, from bench.py)"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
sys.path.insert(0, ROOT)

from wan.distributed.parallel import ParallelPlan, SeqParallel, _Done  # noqa: E402

L, C, F, H, BLOCKS, LC = 32760, 1536, 8960, 12, 30, 512
GRID = (21, 30, 52)
LINK_GBS = 153.0


class LocalSeqParallel(SeqParallel):
    """SeqParallel of size P whose exchanges are local memory moves of the right shapes (values are meaningless)."""

    def __init__(self, P):
        self.group, self.size, self.rank = None, P, 0

    def scatter_heads(self, x, async_op=False, cols=None):
        P = self.size
        lp, c = x.shape
        send = x.view(lp, P, c // P)
        if cols is not None:
            send = send[:, :, cols[0]:cols[1]]
        w = send.shape[2]
        y = send.transpose(0, 1).contiguous().view(P * lp, w)  # the pack copy of the real method; "received" = what was packed
        return _Done(y) if async_op else y

    def scatter_packed(self, flat, lp, off, w, async_op=False):
        y = flat[off:off + self.size * lp * w].view(self.size * lp, w)
        return _Done(y) if async_op else y

    def gather_heads(self, x, async_op=False, out=None, cols=None):
        P = self.size
        l, w = x.shape
        lp = l // P
        r = x.contiguous().view(P, lp, w)
        if out is None:
            y = r.transpose(0, 1).reshape(lp, P * w)
        else:
            out.view(lp, P, out.shape[1] // P)[:, :, cols[0]:cols[1]] = r.transpose(0, 1)
            y = out
        return _Done(y) if async_op else y


def headline_block(dev):
    from oracle import wan_ref as wr  # rope frequencies only (a tool, not the product path)
    from qdiff import config as qcfg
    from qdiff.base.quant_model import quant_layer_refactor_
    from qdiff.utils import apply_func_to_submodules
    from wan import calib
    from wan.modules.model import WanAttentionBlock
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel

    torch.manual_seed(7)
    blk = WanAttentionBlock("t2v_cross_attn", C, F, H, cross_attn_norm=True)
    for m in blk.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    cfg = qcfg.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                       "viditq": {"alpha": 0.5665, "layer_name_regex": r"self_attn\.(q|k|v)$"}, "remain_fp_regex": None})
    blk = blk.to(dev)
    apply_func_to_submodules(blk, torch.nn.Linear, quant_layer_refactor_, name=None, parent_module=None, quant_config=cfg,
                             full_name=None, remain_fp_regex=cfg.remain_fp_regex)
    gen = torch.Generator().manual_seed(11)
    act_mask = torch.rand(C, generator=gen) * 3 + 0.2
    for name in ("q", "k", "v"):
        calib.init_rotation_and_channel_mask_(getattr(blk.self_attn, name), "x", {"x": act_mask[None]}, gen)
    return WanAttentionBlockWithHipKernel.from_float(blk, None), wr.rope_freqs(C // H)


def time_block(hb, freqs, P, dev, reps=6):
    from wan import ops
    from wan.quant_wanx_hip import _FpSrc

    lp = L // P
    assert lp * P == L
    g = torch.Generator(device=dev).manual_seed(P)
    x0 = torch.randn(lp, C, device=dev, generator=g)
    e0 = torch.randn(1, 6, C, device=dev, generator=g) * 0.3
    ctx = _FpSrc(torch.randn(LC, C, device=dev, generator=g).to(torch.bfloat16), torch.bfloat16)
    rope = ops.rope_table(freqs, GRID, dev)[:lp].contiguous()  # rank 0's slice of the positions
    sp = None if P == 1 else LocalSeqParallel(P)
    ts = []
    for i in range(reps + 4):
        x = x0.clone()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        hb(x, e0, rope, L, ctx, sp)
        e.record()
        torch.cuda.synchronize()
        if i >= 4:
            ts.append(s.elapsed_time(e))
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step-ms-n1", type=float, default=0.0, help="measured N = 1 step time on this box (bench.py ms_per_step); "
                    "0 = blocks only")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    hb, freqs = headline_block(dev)
    time_block(hb, freqs, 1, dev, reps=2)  # clocks and allocator warm before the first timed configuration
    t1 = None
    print("PROJECTION from one GPU (per-rank compute measured, exchange priced from bytes) -- not a multi-GPU measurement")
    print(f"{'N':>2} {'plan':>9} {'tokens/rank':>11} {'block ms':>9} {'step ms (hidden)':>17} {'steps/s':>8} {'speed-up':>8} "
          f"{'a2a MB/block':>12} {'a2a ms/block @1 link-rate':>26} {'step ms (exposed)':>18} {'steps/s':>8}")
    for n in (1, 2, 4, 8):
        cfg_deg, sp_deg = ParallelPlan.choose(n, H)
        tb = time_block(hb, freqs, sp_deg, dev)
        if n == 1:
            t1 = tb
        rest = max(0.0, args.step_ms_n1 - 2 * BLOCKS * t1) if args.step_ms_n1 else 0.0  # embeddings, head, CFG + scheduler, gaps
        passes = 2 // cfg_deg
        step_hidden = passes * BLOCKS * tb + rest
        a2a_bytes = 4 * (L // sp_deg) * C * 2 * (sp_deg - 1) / sp_deg if sp_deg > 1 else 0.0
        a2a_ms = (a2a_bytes / max(1, sp_deg - 1)) / (LINK_GBS * 1e9) * 1e3 if sp_deg > 1 else 0.0  # each peer's share over its own link
        step_exposed = step_hidden + passes * BLOCKS * 2.0 * a2a_ms  # "exposed": nothing hidden AND the links at half their nominal rate
        base = 2 * BLOCKS * t1 + rest
        print(f"{n:>2} {f'cfg{cfg_deg}xsp{sp_deg}':>9} {L // sp_deg:>11} {tb:>9.3f} {step_hidden:>17.1f} {1e3 / step_hidden:>8.2f} "
              f"{base / step_hidden:>8.2f} {a2a_bytes / 1e6:>12.1f} {a2a_ms:>26.3f} {step_exposed:>18.1f} {1e3 / step_exposed:>8.2f}")


if __name__ == "__main__":
    main()
