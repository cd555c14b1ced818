"""The FP backbone around the quantized layers, pinned to the reference's own wan/modules/model.py (fixture
tests/golden/model_tiny.npz from tests/golden/make_golden_model.py: the reference file run stand-alone in fp32 on a tiny T2V
configuration, attention core = the fp32 softmax definition, self-attention forward = the reference's usp_attn_forward q / k
handling because model.py's own is a non-runnable debug edit).

CPU part: the oracle's restatements (oracle/wan_ref.py: RoPE tables, rope_apply, RMSNorm, LayerNorm, whole block) against the
reference's outputs.  GPU part: this repository's WanModel (same parameter names, rebuilt from the same per-name seeds) against
the reference's stage outputs and final output."""
import math
import os
import sys
import zlib

import numpy as np
import pytest
import torch

from oracle import wan_ref as wr

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gm():
    return np.load(os.path.join(HERE, "golden", "model_tiny.npz"))


def seeded_parameters_(model):
    """tests/golden/make_golden_model.py::seeded_parameters_ (every parameter from crc32(name))."""
    with torch.no_grad():
        for name, p in model.named_parameters():
            g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
            r = torch.randn(p.shape, generator=g)
            if "modulation" in name or name.endswith(".bias"):
                v = 0.3 * r
            elif "norm" in name and name.endswith(".weight"):
                v = 1.0 + 0.3 * r
            elif name.endswith("head.head.weight"):
                v = 0.05 * r
            else:
                v = r / math.sqrt(p[0].numel())
            p.copy_(v.to(p.device, p.dtype))


def test_oracle_rope_norms_vs_reference_model_py(gm):
    freqs = wr.rope_freqs(128)
    ref = torch.complex(torch.from_numpy(gm["freqs_real"]), torch.from_numpy(gm["freqs_imag"]))
    torch.testing.assert_close(freqs[:16], ref, rtol=0, atol=1e-15)  # float64 tables: identical construction
    q = torch.from_numpy(gm["rope_in"])[0]  # [40, 2, 128]; 36 real tokens on a 3 x 4 x 3 grid, 4 padding rows left unrotated
    out = wr.rope_apply(q, (3, 4, 3), freqs)
    np.testing.assert_allclose(out.numpy(), gm["rope_out"][0], rtol=0, atol=1e-6)
    x = torch.from_numpy(gm["rms_in"])
    np.testing.assert_allclose(wr.rms_norm(x, torch.from_numpy(gm["rms_w"]), 1e-6).numpy(), gm["rms_out"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(wr.layer_norm(x, 1e-6).numpy(), gm["ln_out"], rtol=2e-6, atol=2e-6)


def test_oracle_fp_block_vs_reference_block(gm):
    """oracle BlockRef (quant=False) == WanAttentionBlock.forward of the reference: modulation, both attentions, FFN, residuals."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from wan.modules.model import WanAttentionBlock

    blk = WanAttentionBlock("t2v_cross_attn", 256, 512, 2, (-1, -1), True, True, 1e-6)

    class Holder(torch.nn.Module):  # the reference names its parameters blocks.0.<...>
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([blk])

    seeded_parameters_(Holder())
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    ref_blk = wr.block_from_state(sd, 2, eps=1e-6, quant=False)
    x = torch.from_numpy(gm["block0_in"])[0]            # [40, 256]
    e0 = torch.from_numpy(gm["block0_e"])                # [1, 6, 256]
    ctx = torch.from_numpy(gm["block0_context"])[0]     # [32, 256]
    out = ref_blk(x, e0, (3, 4, 3), 36, ctx, wr.rope_freqs(128))
    ref = gm["block0_out"][0]
    assert float(np.abs(out.numpy() - ref).max()) < 2e-5 * float(np.abs(ref).max())


@pytest.mark.gpu
def test_wan_model_forward_vs_reference_model_py(gm):
    """wan.modules.model.WanModel on the GPU (HIP RMSNorm+RoPE and bf16 flash attention inside) vs the reference's fp32 run."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from wan.modules.model import WanModel, sinusoidal_embedding_1d

    torch.manual_seed(0)
    with torch.device("cuda"):
        model = WanModel(model_type="t2v", patch_size=(1, 2, 2), text_len=32, in_dim=16, dim=256, ffn_dim=512, freq_dim=64, text_dim=64,
                         out_dim=16, num_heads=2, num_layers=2, eps=1e-6).eval()
    assert sorted(n for n, _ in model.named_parameters()) == list(gm["param_names"])  # the reference's parameter names
    seeded_parameters_(model)
    np.testing.assert_allclose(sinusoidal_embedding_1d(64, torch.from_numpy(gm["in_t"]).cuda()).float().cpu().numpy(), gm["sin_emb"],
                               rtol=1e-6, atol=1e-6)
    x, ctx, t = (torch.from_numpy(gm[k]).cuda() for k in ("in_x", "in_ctx", "in_t"))
    stages = {}
    h0 = model.blocks[0].register_forward_hook(lambda m, a, o: stages.__setitem__("b0", o))
    h1 = model.blocks[1].register_forward_hook(lambda m, a, o: stages.__setitem__("b1", o))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):  # as the pipeline runs it (text2video.py:213)
        out = model([x], t, [ctx], int(gm["seq_len"]))[0]
    h0.remove(); h1.remove()

    def rel(a, b):
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    b0 = stages["b0"].float().cpu().numpy().reshape(gm["block0_out"].shape)
    b1 = stages["b1"].float().cpu().numpy().reshape(gm["block1_out"].shape)
    # real tokens only: the reference lets the 4 padding rows of the sequence flow through the blocks too; they never reach the output
    e0, e1, eo = rel(b0[:, :36], gm["block0_out"][:, :36]), rel(b1[:, :36], gm["block1_out"][:, :36]), rel(out.float().cpu().numpy(), gm["out"])
    print(f"WanModel vs reference model.py: block 0 {e0:.2e}, block 1 {e1:.2e}, output {eo:.2e}")
    # bf16 autocast matmuls + bf16 q / k / v / P inside the HIP attention against the reference's fp32 run
    assert out.shape == (16, 3, 8, 6) and max(e0, e1, eo) < 2e-2


@pytest.mark.gpu
def test_fp_model_fused_glue_against_the_torch_expressions(gm, monkeypatch):
    """The FP blocks on the GPU run LayerNorm + modulate, RMSNorm + RoPE and gate + residual as the library's fused kernels
    (wan/modules/model.py::fused_fp); WANQ_FP_FUSED=0 evaluates the reference's torch expressions.  Same model, same inputs, both ways:
    what the first block's Linears see in front of any 16-bit rounding (the LayerNorm + modulate outputs: the calibration hooks' view)
    agrees to fp32 rounding; block outputs and the model output to the 16-bit roundings inside the attention."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from wan.modules.model import WanModel

    torch.manual_seed(0)
    with torch.device("cuda"):
        model = WanModel(model_type="t2v", patch_size=(1, 2, 2), text_len=32, in_dim=16, dim=256, ffn_dim=512, freq_dim=64, text_dim=64,
                         out_dim=16, num_heads=2, num_layers=2, eps=1e-6).eval()
    seeded_parameters_(model)
    x, ctx, t = (torch.from_numpy(gm[k]).cuda() for k in ("in_x", "in_ctx", "in_t"))

    def run(flag):
        monkeypatch.setenv("WANQ_FP_FUSED", flag)
        seen, hooks = {}, []
        for name in ("blocks.0.self_attn.q", "blocks.0.self_attn.o", "blocks.0.cross_attn.q", "blocks.0.ffn.0", "blocks.1.self_attn.q", "head.head"):
            mod = model.get_submodule(name)
            hooks.append(mod.register_forward_pre_hook(lambda m, a, name=name: seen.__setitem__(name, a[0].detach().float().clone())))
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):  # as the pipeline runs it (text2video.py:213)
            out = model([x], t, [ctx], int(gm["seq_len"]))[0]
        for h in hooks:
            h.remove()
        return out.float(), seen

    o1, s1 = run("1")
    o0, s0 = run("0")

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    assert s1["blocks.0.self_attn.q"].dtype == torch.float32 and s1["blocks.0.self_attn.q"].shape == s0["blocks.0.self_attn.q"].shape
    assert rel(s1["blocks.0.self_attn.q"], s0["blocks.0.self_attn.q"]) < 2e-6        # LayerNorm + modulate (shift e0, scale e1)
    tol = 1e-2                                                                         # behind the 16-bit Linears and attention operands
    assert rel(s1["blocks.0.self_attn.o"][:, :36], s0["blocks.0.self_attn.o"][:, :36]) < tol
    for name in ("blocks.0.cross_attn.q", "blocks.0.ffn.0", "blocks.1.self_attn.q", "head.head"):
        assert rel(s1[name][:, :36], s0[name][:, :36]) < tol, name
    assert rel(o1, o0) < tol
    ref = torch.from_numpy(gm["out"]).cuda()
    assert rel(o1, ref) < 2e-2 and rel(o0, ref) < 2e-2


def seeded_vidit(name, n):
    """tests/golden/make_golden_model.py::seeded_vidit."""
    g = torch.Generator().manual_seed(zlib.crc32((name + ".vidit").encode()))
    act_mask = torch.rand(n, generator=g) + 0.5
    act_mask[[3, n // 2]] *= 9.0
    signs = (torch.randint(0, 2, (n,), generator=g) * 2 - 1).double()
    return act_mask, signs


def seeded_layer_input(name, n, rows=8):
    """tests/golden/make_golden_model.py::seeded_layer_input."""
    g = torch.Generator().manual_seed(zlib.crc32((name + ".x").encode()))
    x = torch.randn(rows, n, generator=g)
    x[:, [5, n - 7]] *= 10.0
    return x


def _block0_state_and_vidit():
    from oracle import qdiff_ref as qr

    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from wan.modules.model import WanAttentionBlock

    blk = WanAttentionBlock("t2v_cross_attn", 256, 512, 2, (-1, -1), True, True, 1e-6)

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([blk])

    seeded_parameters_(Holder())
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    vidit = {}
    for l in "qkv":
        act_mask, signs = seeded_vidit(f"blocks.0.self_attn.{l}", 256)
        mask = qr.vidit_channel_mask(sd[f"self_attn.{l}.weight"].numpy(), act_mask.numpy(), 0.5665)
        vidit[f"self_attn.{l}"] = (torch.from_numpy(mask), torch.from_numpy(np.ascontiguousarray(qr.hadamard_from_signs(signs.numpy()))))
    return sd, vidit


def test_oracle_fake_quant_layers_and_block_vs_reference_simulation_mode(gm):
    """The reference's SIMULATION mode -- its qdiff package swapped into its model.py by quant_layer_refactor_, masks / rotations
    given the way ptq_wanx.py gives them -- against the oracle (W8A8 on the ten Linears, ViDiT scale + rotate on self-attention
    q / k / v).  Layer level (each quantized Linear of block 0 on a seeded input): plain layers to fp32 rounding, ViDiT layers
    up to a code flipping at a .5 boundary.  Block level: within the quantisation noise itself -- every re-quantisation down the
    block turns a last-bit difference of its input into a flipped code now and then, so two faithful evaluations drift apart
    by a fraction of the fake-quant-vs-FP gap (measured 3.8e-3 against 8.2e-3)."""
    sd, vidit = _block0_state_and_vidit()
    ref_blk = wr.block_from_state(sd, 2, eps=1e-6, quant=True, vidit=vidit)
    for name in wr.LINEARS:
        x = seeded_layer_input("blocks.0." + name, sd[name + ".weight"].shape[1])
        out, ref = ref_blk.lin[name](x).numpy(), gm["layer_out.blocks.0." + name]
        err = float(np.linalg.norm(out - ref) / np.linalg.norm(ref))
        assert err < (1e-3 if name in vidit else 1e-6), (name, err)
    x = torch.from_numpy(gm["block0_in"])[0]
    out = ref_blk(x, torch.from_numpy(gm["block0_e"]), (3, 4, 3), 36, torch.from_numpy(gm["block0_context"])[0], wr.rope_freqs(128))
    ref = gm["quant_block0_out"][0]
    err = float(np.linalg.norm(out.numpy()[:36] - ref[:36]) / np.linalg.norm(ref[:36]))
    fp_gap = float(np.linalg.norm(gm["block0_out"][0][:36] - ref[:36]) / np.linalg.norm(ref[:36]))
    print(f"oracle fake-quant block vs reference simulation mode: {err:.2e} (fake-quant vs FP itself: {fp_gap:.2e})")
    assert err < 0.6 * fp_gap


@pytest.mark.gpu
def test_qdiff_layers_vs_reference_simulation_mode_layers(gm):
    """This repository's quantized Linears (int8 GEMM inside) on the same seeded inputs as the reference's simulation-mode layers of
    block 0: plain QuantizedLinear 2e-5 of the output range, ViDiT layers within a flipped code."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from qdiff import config as qcfg
    from qdiff.base.quant_layer import QuantizedLinear
    from qdiff.viditq.viditq_quant_layer import ViDiTQuantizedLinear

    sd, _ = _block0_state_and_vidit()
    base = {"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True}}
    for name in wr.LINEARS:
        w, b = sd[name + ".weight"].cuda(), sd[name + ".bias"].cuda()
        lin = torch.nn.Linear(w.shape[1], w.shape[0]).cuda()
        lin.weight.data, lin.bias.data = w, b
        if name in ("self_attn.q", "self_attn.k", "self_attn.v"):
            ql = ViDiTQuantizedLinear(w.shape[1], w.shape[0], True, "cuda", qcfg.create(dict(base, viditq={"alpha": 0.5665, "layer_name_regex": ""})), lin)
            act_mask, signs = seeded_vidit("blocks.0." + name, w.shape[1])
            ql.get_channel_mask(act_mask.cuda())
            ql.rotation_signs = signs
            ql.update_quantized_weight_rotated_and_scaled()
            tol = 5e-3
        else:
            ql = QuantizedLinear(w.shape[1], w.shape[0], True, "cuda", qcfg.create(base), lin)
            tol = 2e-5
        ql.w_quantizer.init_done = True
        x = seeded_layer_input("blocks.0." + name, w.shape[1]).cuda()
        out, ref = ql(x.unsqueeze(0))[0].float().cpu().numpy(), gm["layer_out.blocks.0." + name]
        assert np.abs(out - ref).max() < tol * np.abs(ref).max() + tol, (name, float(np.abs(out - ref).max()), float(np.abs(ref).max()))


@pytest.mark.gpu
def test_kernel_mode_model_vs_reference_simulation_mode(gm):
    """The whole quantized model of this repository -- simulation mode (qdiff modules) and kernel mode (HIP blocks) -- against the
    reference's simulation-mode output on the same input, parameters, masks and rotations."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from qdiff import config as qcfg
    from qdiff.base.quant_layer import QuantizedLinear
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    torch.manual_seed(0)
    with torch.device("cuda"):
        fp = WanModel(model_type="t2v", patch_size=(1, 2, 2), text_len=32, in_dim=16, dim=256, ffn_dim=512, freq_dim=64, text_dim=64,
                      out_dim=16, num_heads=2, num_layers=2, eps=1e-6).eval()
    seeded_parameters_(fp)
    cfg = qcfg.create({"remain_fp_regex": r"text_embedding|time_embedding|time_projection|head\.head",
                       "weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                       "viditq": {"alpha": 0.5665, "layer_name_regex": r"self_attn\.(q|k|v)$"}})
    model = QuantWanModel.from_float(fp, cfg)
    model.quant_layer_refactor()
    classes = sorted(f"{n}={type(m).__name__}" for n, m in model.named_modules() if isinstance(m, QuantizedLinear))
    assert classes == list(gm["quant_classes"])  # the same twenty layers, the same six ViDiT ones
    for name, mod in model.named_modules():
        if type(mod).__name__ == "ViDiTQuantizedLinear":
            act_mask, signs = seeded_vidit(name, mod.in_features)
            mod.get_channel_mask(act_mask.cuda())
            mod.rotation_signs = signs
            mod.update_quantized_weight_rotated_and_scaled()
    model.set_init_done()
    model.eval()
    x, ctx, t = (torch.from_numpy(gm[k]).cuda() for k in ("in_x", "in_ctx", "in_t"))
    ref, ref_fp = gm["quant_out"], gm["out"]

    def rel(a, b):
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))

    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        sim = model([x], t, [ctx], int(gm["seq_len"]))[0].float().cpu().numpy()
    model.hardware_forward_refactor()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        hw = model([x], t, [ctx], int(gm["seq_len"]))[0].float().cpu().numpy()
    print(f"vs the reference's simulation mode: ours simulation {rel(sim, ref):.2e}, ours kernel mode {rel(hw, ref):.2e}; "
          f"(reference quantized vs reference FP {rel(ref, ref_fp):.2e})")
    # model level: two faithful evaluations of the same fake-quant recipe drift apart by a fraction of the quantisation noise
    # (every re-quantisation flips a code now and then; bf16 attention on top): bar = the reference's own quantized-vs-FP gap
    gap = rel(ref, ref_fp)
    assert rel(sim, ref) < 1.5 * gap and rel(hw, ref) < 1.5 * gap and rel(hw, ref_fp) < 2.5 * gap


def _tiny_quant_model():
    """This repository's QuantWanModel with the fixture's parameters, masks and rotations (the model of
    test_kernel_mode_model_vs_reference_simulation_mode), in simulation mode."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from qdiff import config as qcfg
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    torch.manual_seed(0)
    with torch.device("cuda"):
        fp = WanModel(model_type="t2v", patch_size=(1, 2, 2), text_len=32, in_dim=16, dim=256, ffn_dim=512, freq_dim=64, text_dim=64,
                      out_dim=16, num_heads=2, num_layers=2, eps=1e-6).eval()
    seeded_parameters_(fp)
    cfg = qcfg.create({"remain_fp_regex": r"text_embedding|time_embedding|time_projection|head\.head",
                       "weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                       "viditq": {"alpha": 0.5665, "layer_name_regex": r"self_attn\.(q|k|v)$"}})
    model = QuantWanModel.from_float(fp, cfg)
    model.quant_layer_refactor()
    for name, mod in model.named_modules():
        if type(mod).__name__ == "ViDiTQuantizedLinear":
            act_mask, signs = seeded_vidit(name, mod.in_features)
            mod.get_channel_mask(act_mask.cuda())
            mod.rotation_signs = signs
            mod.update_quantized_weight_rotated_and_scaled()
    model.set_init_done()
    return model.eval()


def _spectral_concentration(err):
    """Share of the error's energy in its largest singular direction.  Code flips are independent per element: for a [T, C] matrix
    of such noise the share is about (1/T)(1 + sqrt(T/C))^2 (0.05 at 36 x 256).  A wrong term of the dequantisation equation is
    structured -- a zero-point error is the outer product (per-token sum) x (per-channel delta zp . scale), a bias error is
    1 x delta b, a wrong per-token or per-channel scale is the signal times a rank-one factor -- and concentrates it."""
    s = np.linalg.svd(err.astype(np.float64), compute_uv=False)
    return float(s[0] ** 2 / (s ** 2).sum())


@pytest.mark.gpu
def test_kernel_mode_blocks_teacher_forced_vs_reference_simulation_mode(gm):
    """block-level parity that can tell a faithful implementation from a merely similar quantiser.  Every
    kernel-mode block is fed the REFERENCE'S OWN block input of its simulation-mode run (block 0: `block0_in`; block 1: the
    reference's `quant_block0_out`), with its `e` and `context`, and compared with the reference's output of that block
    (W/wan/quant_wanx_cuda.py:170-310 dataflow, Q/viditq/viditq_quant_layer.py:52-73 layers): no error travels from block to block.

    What bounds such a comparison.  A dynamic quantiser turns an input difference dx << delta into a code flip with probability
    |dx| / delta, i.e. into an error of power dx . delta instead of dx^2: fp32 reduction-order noise (1e-7) in front of the first
    quantisers of a block comes out at about 1e-4, and that, in front of the next quantiser down the block (attention output ->
    `o`, GELU output -> `ffn.2`), at sqrt(1e-4 . 10 / 127) = 3e-3 -- the level of the quantisation noise itself.  Two faithful
    evaluations of this recipe therefore agree either bit for bit or to a few 1e-3 at dim 256, nothing in between (the oracle,
    a plain CPU restatement, sits 3.8e-3 from the reference on block 0); a relative-error bar alone cannot be tighter.
    The test therefore carries two measures per configuration:
      * relative error of the block output over the real tokens; bars (a) 5e-3, (b) / (c) 7.5e-3 = the measured flip floors
        (3.6e-3 / 2.5e-3, 5.7e-3 / 4.4e-3, 5.6e-3 / 4.3e-3 for blocks 0 / 1) x 1.3;
      * the SHAPE of the error: flip noise is unstructured (spectral concentration 0.06-0.11 measured at 36 x 256), an error in
        an additive term of the dequantisation equation is rank-one structured (0.8-1.0).  Bar 0.2.
    and proves its own sensitivity: the same block with one layer's zero points off by 0.05 code or one layer's bias off by 5e-4
    of the output range -- additive faults BELOW the flip floor in relative error -- must fail the shape measure.  (A wrong
    multiplicative factor is not visible this way -- 0.4 % on one layer's weight scales moves block 0 from 3.6e-3 to 4.5e-3 with
    an unstructured error -- which is why scales and zero points are pinned bit for bit at the layer level:
    tests/test_gpu_qdiff.py, test_gpu_gemm.py.)
    Configurations: (a) fp32 activations, attention core = the fp32 softmax definition (what the fixture's generator gave the
    reference in place of the external flash_attn); (b) fp32 activations, the HIP attention kernel (bf16 operands, bf16 P);
    (c) the shipped configuration (bf16 activations between the kernels)."""
    from wan import ops
    from wan.quant_wanx_hip import _FpSrc

    model = _tiny_quant_model()
    e0 = torch.from_numpy(gm["block0_e"]).cuda().float()
    ctx = torch.from_numpy(gm["block0_context"])[0].cuda().float().contiguous()
    rope = model._rope((3, 4, 3), torch.device("cuda"))
    blocks_io = [(gm["block0_in"][0], gm["quant_block0_out"][0]), (gm["quant_block0_out"][0], gm["quant_block1_out"][0])]

    def fp32_attention(q, k, v, num_heads, k_len=None, out=None, splits=None):
        Lq, C = q.shape
        d = C // num_heads
        kl = k.shape[0] if k_len is None else min(int(k_len), k.shape[0])
        s = torch.einsum("qhd,khd->hqk", q.float().view(Lq, num_heads, d), k.float().view(-1, num_heads, d)[:kl]) / d ** 0.5
        return torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), v.float().view(-1, num_heads, d)[:kl]).reshape(Lq, C).to(q.dtype)

    def run_block(i, act_dtype):
        x = torch.from_numpy(blocks_io[i][0]).cuda().float().contiguous().clone()
        out = model.hip_blocks[i](x, e0, rope, 36, _FpSrc(ctx, act_dtype)).float().cpu().numpy()[:36]  # real tokens only
        ref = blocks_io[i][1][:36]
        err = out - ref
        return float(np.linalg.norm(err) / np.linalg.norm(ref)), _spectral_concentration(err)

    real_attention = ops.attention
    bars, shape_bar = {"a": 5.0e-3, "b": 7.5e-3, "c": 7.5e-3}, 0.2
    results, caught = {}, {}
    try:
        for tag, act_dtype, attn in (("a", torch.float32, fp32_attention), ("b", torch.float32, real_attention), ("c", torch.bfloat16, real_attention)):
            ops.attention = attn
            model.hardware_forward_refactor(act_dtype=act_dtype)
            for i in range(2):
                results[(tag, i)] = run_block(i, act_dtype)
            if tag != "a":
                continue
            # ---- sensitivity: two small additive faults in block 0, one at a time
            hb = model.hip_blocks[0]
            lin = hb.ffn2
            lin.zp_weight += 0.05
            caught["ffn.2 zero points + 0.05 code"] = run_block(0, act_dtype)
            lin.zp_weight -= 0.05
            lin = hb.cross_attn.o
            db = 5e-4 * float(np.abs(blocks_io[0][1]).max())
            lin.bias += db
            caught["cross_attn.o bias + 5e-4 of the output range"] = run_block(0, act_dtype)
            lin.bias -= db
    finally:
        ops.attention = real_attention
    print("teacher-forced kernel-mode blocks vs the reference's simulation mode (rel error, spectral concentration): " +
          ", ".join(f"({t}) block {i}: {e:.2e} / {c:.3f}" for (t, i), (e, c) in sorted(results.items())))
    print("injected faults, configuration (a), block 0: " + ", ".join(f"{k}: {e:.2e} / {c:.3f}" for k, (e, c) in caught.items()))
    for (tag, i), (e, c) in results.items():
        assert e < bars[tag] and c < shape_bar, (tag, i, e, c)
    for k, (e, c) in caught.items():
        assert e >= bars["a"] or c >= shape_bar, f"the test does not see the fault '{k}': {e:.2e} / {c:.3f}"


def test_from_pretrained_and_from_float_copy_every_tensor_without_initial_draws(tmp_path):
    """WanModel.from_pretrained (config.json + *.safetensors, the layout of a Wan2.1 checkpoint directory) and QuantWanModel.from_float
    build their module tree without drawing initial values (`_skip_init`): every parameter must then come from the source -- equal
    values, own storage -- and the RNG stream must not move (a skipped draw is the point)."""
    import json

    from safetensors.torch import save_file

    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    torch.manual_seed(3)
    src = WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, freq_dim=64, text_len=32).eval()
    with torch.no_grad():
        src.head.head.weight.normal_()
    json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in src.config.items()}, open(tmp_path / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in src.state_dict().items()}, str(tmp_path / "model.safetensors"))
    state = torch.get_rng_state()
    a = WanModel.from_pretrained(str(tmp_path))
    b = QuantWanModel.from_pretrained(str(tmp_path), None)
    c = QuantWanModel.from_float(src, None)
    assert torch.equal(torch.get_rng_state(), state)
    want = src.state_dict()
    for m in (a, b, c):
        got = m.state_dict()
        assert got.keys() == want.keys() and m.config == src.config and torch.equal(m.freqs, src.freqs)
        for k in want:
            assert torch.equal(got[k], want[k]) and got[k].data_ptr() != want[k].data_ptr(), k
