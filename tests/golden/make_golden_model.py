#!/usr/bin/env python3
"""Golden vectors for the FP backbone the quantized layers are embedded in, from the reference's OWN wan/modules/model.py.

Run ONLY in the build container (where /root/reference exists):    python tests/golden/make_golden_model.py

The `wan` package is not importable here (flash_attn, xfuser, easydict ... are absent), but wan/modules/model.py itself needs
only torch + the diffusers configuration mixins (stand-in: tests/golden/gen/diffusers/) + `.attention.flash_attention`.  It
is loaded as a stand-alone module inside a synthetic package; three things are substituted IN THIS PROCESS ONLY:
  * `flash_attention` (asserts on the external flash_attn package, absent): replaced by the fp32 definition
    softmax(q k^T / sqrt(d)) v with the k_lens key mask -- the attention CORE is therefore not pinned by these fixtures
    (external dependency), everything around it is the reference's code: sinusoidal embedding, RoPE tables and
    rope_apply (float64 complex), WanRMSNorm / WanLayerNorm, WanAttentionBlock's modulation / residual / FFN dataflow, cross
    attention, Head, patch embedding, unpatchify;
  * `WanSelfAttention.forward`: the file's version is a debugging edit that cannot run (model.py:144-146 leaves q without
    norm_q and without the [B, L, heads, d] view, then rope_apply indexes dimension 3: SURVEY defect D1); the reference's own
    working form of the same method, usp_attn_forward (wan/distributed/xdit_context_parallel.py:162-170: q = norm_q(q(x)).view,
    k likewise, rope on both, attention, o), is used without its sequence-parallel exchange and with model.py's k_lens mask;
  * `torch.cuda.synchronize` (called unconditionally in WanModel.forward, model.py:619): a no-op on this CPU-only container.
A tiny T2V configuration (dim 256, 2 heads of 128, ffn 512, 2 blocks, text_len 32) with seeded random parameters is run in fp32
on one latent [16, 3, 8, 6] (36 tokens, sequence padded to 40) and stored in tests/golden/model_tiny.npz: the inputs and the
outputs of each stage; the parameters are regenerated from per-name seeds (seeded_parameters_), only their names are stored."""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ViDiT-Q/examples/Wan2.1/wan/modules"
sys.path.insert(0, os.path.join(HERE, "gen"))


def load_pkg():
    pkg = types.ModuleType("refwan"); pkg.__path__ = []
    sub = types.ModuleType("refwan.modules"); sub.__path__ = [REF]
    sys.modules["refwan"], sys.modules["refwan.modules"] = pkg, sub
    out = {}
    for name in ("attention", "model"):
        spec = importlib.util.spec_from_file_location("refwan.modules." + name, os.path.join(REF, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["refwan.modules." + name] = mod
        spec.loader.exec_module(mod)
        out[name] = mod
    return out["model"]


def softmax_attention(q, k, v, q_lens=None, k_lens=None, dropout_p=0., softmax_scale=None, q_scale=None, causal=False,
                      window_size=(-1, -1), deterministic=False, dtype=torch.bfloat16, version=None):
    """[B, Lq, H, d] x [B, Lk, H, d] -> [B, Lq, H, d], fp32, keys >= k_lens[b] masked (what varlen flash attention computes)."""
    assert not causal and window_size == (-1, -1) and q_scale is None
    B, Lq, H, d = q.shape
    s = torch.einsum("bqhd,bkhd->bhqk", q.float(), k.float()) * (softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(d))
    if k_lens is not None:
        mask = torch.arange(k.shape[1])[None, :] >= k_lens[:, None]
        s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    return torch.einsum("bhqk,bkhd->bqhd", torch.softmax(s, dim=-1), v.float())


def seeded_parameters_(model):
    """Every parameter from its own seed (crc32 of its name), so that a test can rebuild the identical model from the names alone
    (the fixture stays small).  Matrices ~ N(0, 1/fan_in), norm weights 1 + 0.3 N, biases / modulation 0.3 N, head 0.05 N."""
    import zlib
    for name, p in model.named_parameters():
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        r = torch.randn(p.shape, generator=g)
        if "modulation" in name or name.endswith(".bias"):
            p.copy_(0.3 * r)
        elif "norm" in name and name.endswith(".weight"):
            p.copy_(1.0 + 0.3 * r)
        elif name.endswith("head.head.weight"):
            p.copy_(0.05 * r)
        else:
            p.copy_(r / math.sqrt(p[0].numel()))


def seeded_layer_input(name, n, rows=8):
    """[rows, n] activations with two outlier channels, from the layer's name."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32((name + ".x").encode()))
    x = torch.randn(rows, n, generator=g)
    x[:, [5, n - 7]] *= 10.0
    return x


def seeded_vidit(name, n):
    """(act_mask fp32 [n] in [0.5, 1.5) with two outlier channels, rotation signs float64 [n]) of a ViDiT layer, from its name."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32((name + ".vidit").encode()))
    act_mask = torch.rand(n, generator=g) + 0.5
    act_mask[[3, n // 2]] *= 9.0
    signs = (torch.randint(0, 2, (n,), generator=g) * 2 - 1).double()
    return act_mask, signs


def main():
    torch.set_grad_enabled(False)
    torch.set_num_threads(1)
    torch.cuda.synchronize = lambda *a, **k: None
    import builtins
    real_print = builtins.print
    m = load_pkg()
    m.flash_attention = softmax_attention

    def self_attn_forward(self, x, seq_lens, grid_sizes, freqs):  # xdit_context_parallel.py:155-192 minus the all-to-all
        b, s, n, d = *x.shape[:2], self.num_heads, self.head_dim
        q = self.norm_q(self.q(x)).view(b, s, n, d)
        k = self.norm_k(self.k(x)).view(b, s, n, d)
        v = self.v(x).view(b, s, n, d)
        x = m.flash_attention(q=m.rope_apply(q, grid_sizes, freqs), k=m.rope_apply(k, grid_sizes, freqs), v=v, k_lens=seq_lens,
                              window_size=self.window_size)
        return self.o(x.flatten(2))

    m.WanSelfAttention.forward = self_attn_forward
    torch.manual_seed(0)
    model = m.WanModel(model_type="t2v", patch_size=(1, 2, 2), text_len=32, in_dim=16, dim=256, ffn_dim=512, freq_dim=64, text_dim=64,
                       out_dim=16, num_heads=2, num_layers=2, eps=1e-6).float().eval()
    seeded_parameters_(model)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(16, 3, 8, 6, generator=g)
    ctx = torch.randn(20, 64, generator=g)
    t = torch.tensor([417])
    seq_len = 40
    stages = {}
    hooks = [model.blocks[0].register_forward_hook(lambda mod, a, k, o: stages.__setitem__("block0_out", o.clone()), with_kwargs=True),
             model.blocks[0].register_forward_pre_hook(lambda mod, a, k: stages.__setitem__("block0_in", (a[0].clone(), {n: (v.clone() if torch.is_tensor(v) else v) for n, v in k.items()})), with_kwargs=True),
             model.blocks[1].register_forward_hook(lambda mod, a, k, o: stages.__setitem__("block1_out", o.clone()), with_kwargs=True)]
    builtins.print = lambda *a, **k: None  # the forward prints a timing line per block
    out = model([x], t, [ctx], seq_len)[0]
    builtins.print = real_print
    for h in hooks:
        h.remove()
    bin_x, bin_kw = stages["block0_in"]
    # stand-alone pieces
    emb = m.sinusoidal_embedding_1d(64, t)
    q = torch.randn(1, seq_len, 2, 128, generator=g)
    grid = torch.tensor([[3, 4, 3]])
    roped = m.rope_apply(q, grid, model.freqs)
    rms = m.WanRMSNorm(256, eps=1e-6)
    rms.weight.copy_(torch.rand(256, generator=g) + 0.5)
    xr = torch.randn(5, 256, generator=g) * 3
    arrs = {"in_x": x, "in_ctx": ctx, "in_t": t, "seq_len": np.int64(seq_len), "out": out, "sin_emb": emb, "freqs_real": model.freqs.real[:16],
            "freqs_imag": model.freqs.imag[:16], "rope_in": q, "rope_out": roped, "rms_w": rms.weight, "rms_in": xr, "rms_out": rms(xr),
            "ln_out": m.WanLayerNorm(256, eps=1e-6)(xr), "block0_in": bin_x, "block0_e": bin_kw["e"], "block0_context": bin_kw["context"],
            "block0_out": stages["block0_out"], "block1_out": stages["block1_out"]}
    arrs["param_names"] = np.array(sorted(n for n, _ in model.named_parameters()))

    # ---- the same model in the reference's SIMULATION mode: its own qdiff package swaps the block Linears for QuantizedLinear /
    # ViDiTQuantizedLinear (quant_layer_refactor_ with QuantWanModel's keyword arguments, W/wan/quant_wanx.py:85-97), the ViDiT
    # layers get a channel mask and a rotation the way ptq_wanx.py:334-344 gives them (here from per-name seeds), and the
    # fake-quantised model runs the same input.  W8A8 on all twenty block Linears, ViDiT on self-attention q / k / v.
    sys.path.insert(0, "/root/reference/ViDiT-Q/quant_utils")
    import torch.nn as nn
    from omegaconf import OmegaConf
    from qdiff.base.base_quantizer import BaseQuantizer
    from qdiff.base.quant_model import quant_layer_refactor_, set_init_done_
    from qdiff.quarot import quarot_utils
    from qdiff.utils import apply_func_to_submodules
    qcfg = OmegaConf.create({"remain_fp_regex": r"text_embedding|time_embedding|time_projection|head\.head",
                             "weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                             "viditq": {"alpha": 0.5665, "layer_name_regex": r"self_attn\.(q|k|v)$"}})
    apply_func_to_submodules(model, class_type=nn.Linear, function=quant_layer_refactor_, name=None, parent_module=None,
                             quant_config=qcfg, full_name=None, remain_fp_regex=qcfg.remain_fp_regex)
    n_vidit = 0
    for name, mod in model.named_modules():
        if type(mod).__name__ == "ViDiTQuantizedLinear":
            act_mask, signs = seeded_vidit(name, mod.in_features)
            mod.get_channel_mask(act_mask)
            mod.rotation_matrix = quarot_utils.matmul_hadU(torch.diag(signs))  # random_hadamard_matrix with the sign draw made explicit
            mod.update_quantized_weight_rotated_and_scaled()
            n_vidit += 1
        if hasattr(mod, "a_quantizer") and mod.a_quantizer is not None:
            mod.a_quantizer.module_name = name
    apply_func_to_submodules(model, class_type=BaseQuantizer, function=set_init_done_)
    stages.clear()
    # every quantised block's input and output (teacher-forced per-block parity: a block fed the reference's own block input;
    # `e` and `context` are those of the FP run -- time / text embeddings stay FP -- and block 0's input is `block0_in`)
    hq = [model.blocks[0].register_forward_hook(lambda mod, a, k, o: stages.__setitem__("qb0", o.clone()), with_kwargs=True),
          model.blocks[1].register_forward_pre_hook(lambda mod, a, k: stages.__setitem__("qb1_in", a[0].clone()), with_kwargs=True),
          model.blocks[1].register_forward_hook(lambda mod, a, k, o: stages.__setitem__("qb1", o.clone()), with_kwargs=True)]
    builtins.print = lambda *a, **k: None
    out_q = model([x], t, [ctx], seq_len)[0]
    builtins.print = real_print
    for h in hq:
        h.remove()
    arrs["quant_out"], arrs["quant_block0_out"] = out_q, stages["qb0"]
    assert torch.equal(stages["qb1_in"], stages["qb0"])
    arrs["quant_block1_out"] = stages["qb1"]
    # every quantized Linear of block 0 on its own: seeded inputs (regenerable from the layer's name) -> its simulation-mode output
    for n_, mod in model.blocks[0].named_modules():
        if hasattr(mod, "w_quantizer"):
            xin = seeded_layer_input("blocks.0." + n_, mod.in_features)
            arrs["layer_out.blocks.0." + n_] = mod(xin.unsqueeze(0))[0]
    arrs["quant_classes"] = np.array(sorted(f"{n}={type(m).__name__}" for n, m in model.named_modules() if hasattr(m, "w_quantizer")))
    real_print("simulation mode:", n_vidit, "ViDiT layers,", len(arrs["quant_classes"]), "quantized; out vs FP rel",
               float((out_q - out).norm() / out.norm()))
    np.savez_compressed(os.path.join(HERE, "model_tiny.npz"), **{k: (v.detach().numpy() if torch.is_tensor(v) else v) for k, v in arrs.items()})
    real_print("out", tuple(out.shape), float(out.abs().mean()), "block0", float(arrs["block0_out"].abs().mean()), "params", len(arrs["param_names"]))


if __name__ == "__main__":
    main()
