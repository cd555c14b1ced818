"""GPU parity of the attention front-end and of the whole kernel-mode DiT block against the CPU oracle of
the reference's simulation path (oracle/wan_ref.py).

Tolerance of the block test (stated): activations between kernels are bf16 (8 significant bits)
where the oracle is fp32, and every re-quantisation to int8 can move a code by one step when its fp input
moves by a bf16 ulp.  Measured against the oracle the block output's relative Frobenius error is ~3e-3; the
test allows 1e-2, and additionally demands that the kernel-mode block is much closer to the fake-quant
oracle than the quantisation error itself (distance oracle_quant <-> oracle_fp)."""
import numpy as np
import pytest
import torch

from oracle import wan_ref as wr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm()).item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rmsnorm_rope_vs_oracle(dtype):
    from wan import ops

    n, d = 3, 128
    grid = (3, 5, 7)
    L = 3 * 5 * 7 + 6  # 6 padding rows stay unrotated
    g = torch.Generator().manual_seed(2)
    x = torch.randn(L, n * d, generator=g).to(dtype)
    w = torch.rand(n * d, generator=g) + 0.5
    freqs = wr.rope_freqs(d)
    ref = wr.rope_apply(wr.rms_norm(x.float(), w, 1e-6).view(L, n, d), grid, freqs).reshape(L, n * d)
    table = ops.rope_table(freqs, grid, DEV)
    assert table.shape == (105, 64, 2)
    y = ops.rmsnorm_rope_(x.to(DEV).clone(), w.to(DEV), table, d, eps=1e-6)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-6
    np.testing.assert_allclose(y.float().cpu().numpy(), ref.numpy(), rtol=tol, atol=tol)
    # norm only / rope only
    y = ops.rmsnorm_rope_(x.to(DEV).clone(), w.to(DEV), None, d, eps=1e-6)
    np.testing.assert_allclose(y.float().cpu().numpy(), wr.rms_norm(x.float(), w, 1e-6).numpy(), rtol=tol, atol=tol)
    y = ops.rmsnorm_rope_(x.to(DEV).clone(), None, table, d)
    np.testing.assert_allclose(y.float().cpu().numpy(), wr.rope_apply(x.float().view(L, n, d), grid, freqs).reshape(L, -1).numpy(),
                               rtol=tol, atol=tol)


@pytest.mark.parametrize("dim,heads,P,chunks_h", [(1536, 12, 4, [(0, 1), (1, 3)]), (1536, 12, 2, [(0, 2), (2, 4), (4, 6)]),
                                                  (5120, 40, 8, [(0, 5)]), (512, 4, 2, [(0, 1), (1, 2)])])
def test_rmsnorm_rope_scatter_writes_the_ulysses_send_images(dim, heads, P, chunks_h):
    """wanq_rmsnorm_rope_scatter == wanq_rmsnorm_rope followed by the [Lp, P, w] -> [P, Lp, w] pack of every head chunk,
    bit for bit (the store address changes, nothing else)."""
    from wan import ops
    from wan.distributed.parallel import SeqParallel

    d, lp = 128, 77
    g = torch.Generator().manual_seed(dim + P)
    x = torch.randn(lp, dim, generator=g).to(torch.bfloat16).to(DEV)
    w = (torch.rand(dim, generator=g) + 0.5).to(DEV)
    table = ops.rope_table(wr.rope_freqs(d), (7, 11, 1), DEV)
    sp = SeqParallel(False)
    sp.size = P  # layout arithmetic only
    chunks = [(a * d, b * d) for a, b in chunks_h]
    numel, hmap, where = sp.packed_layout(lp, dim, d, chunks, DEV)
    flat = ops.rmsnorm_rope_scatter(x, w, table, d, torch.full((numel,), float("nan"), dtype=torch.bfloat16, device=DEV), hmap, eps=1e-6)
    ref = ops.rmsnorm_rope_(x.clone(), w, table, d, eps=1e-6)
    assert not torch.isnan(flat.float()).any()
    for (c0, c1), (off, wd) in zip(chunks, where):
        want = ref.view(lp, P, dim // P)[:, :, c0:c1].transpose(0, 1).contiguous()
        assert torch.equal(flat[off:off + P * lp * wd].view(P, lp, wd), want)


def make_block(dim, ffn, heads, seed):
    from wan.modules.model import WanAttentionBlock

    torch.manual_seed(seed)
    blk = WanAttentionBlock("t2v_cross_attn", dim, ffn, heads, cross_attn_norm=True)
    for m in blk.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    blk.norm3.weight.data.uniform_(0.5, 1.5)
    blk.norm3.bias.data.normal_(std=0.1)
    for nm in (blk.self_attn.norm_q, blk.self_attn.norm_k, blk.cross_attn.norm_q, blk.cross_attn.norm_k):
        nm.weight.data.uniform_(0.5, 1.5)
    return blk


@pytest.mark.parametrize("dim,ffn,heads,grid,pad,lc", [(256, 512, 2, (3, 6, 10), 4, 40), (1536, 8960, 12, (2, 6, 8), 0, 64),
                                                   (5120, 13824, 40, (1, 6, 8), 0, 32)])  # last: the 14B block shapes
def test_kernel_mode_block_vs_simulation_oracle(dim, ffn, heads, grid, pad, lc):
    from wan import ops
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel

    blk = make_block(dim, ffn, heads, 0)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    L = n_tok + pad
    g = torch.Generator().manual_seed(1)
    x = torch.randn(L, dim, generator=g)
    x[:, 5] *= 12.0  # an outlier channel, as real DiT activations have
    x[n_tok:] = 0
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)

    ref_q = wr.block_from_state(sd, heads, quant=True)(x, e0, grid, n_tok, ctx, freqs)
    ref_fp = wr.block_from_state(sd, heads, quant=False)(x, e0, grid, n_tok, ctx, freqs)

    hb = WanAttentionBlockWithHipKernel.from_float(blk.to(DEV))
    xd = x.to(DEV).clone()
    from wan.quant_wanx_hip import _FpSrc

    out = hb(xd, e0.to(DEV), ops.rope_table(freqs, grid, DEV), n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16))
    assert out.data_ptr() == xd.data_ptr()  # residual stream updated in place
    got = out.float().cpu()[:n_tok]

    err_q = rel_err(got, ref_q[:n_tok])
    quant_noise = rel_err(ref_q[:n_tok], ref_fp[:n_tok])
    print(f"dim={dim}: rel err vs fake-quant oracle {err_q:.2e}; fake-quant vs fp {quant_noise:.2e}")
    assert err_q < 1e-2
    assert err_q < 0.5 * quant_noise + 5e-3


def test_weight_codes_and_params_match_oracle():
    """HipLinearW8A8.from_float: delta / zero_point / dequantised weight identical to the oracle's."""
    from wan.quant_wanx_hip import HipLinearW8A8

    w = torch.randn(96, 256, generator=torch.Generator().manual_seed(4)) * 0.1
    w[0] = w[0].abs()
    w[1] = -w[1].abs()
    hl = HipLinearW8A8.from_float(w.to(DEV), None)
    delta, zp = wr.static_params(w)
    assert torch.equal(hl.scale_weight.cpu(), delta.view(-1)) and torch.equal(hl.zp_weight.cpu(), zp.view(-1))
    deq = (hl.weight.float().cpu() + zp) * delta
    assert torch.equal(deq, wr.static_fake_quant(w))


@pytest.mark.parametrize("Lq,Lk,H,klen", [(32, 64, 1, None), (300, 300, 2, None), (256, 64, 3, None), (100, 512, 2, None),
                                           (1000, 777, 2, None), (515, 640, 12, 601), (2050, 2050, 4, None),
                                           (1, 17, 1, None), (5, 63, 2, None), (17, 65, 1, 33)])  # one query, one ragged tile, a masked second tile
def test_flash_attention_vs_fp32_softmax(Lq, Lk, H, klen):
    """No fixture of the reference pins attention (flash_attn is an external library): the pin is the fp32
    definition softmax(QK^T/sqrt(d))V on the same bf16 inputs.  Tolerance: P is rounded to bf16 before PV
    (as flash-attn does) -> abs error <= 2^-8 * max|v| on outputs of magnitude <= max|v|."""
    from wan import ops

    d = 128
    g = torch.Generator().manual_seed(Lq * 31 + Lk)
    q = (torch.randn(Lq, H * d, generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(Lk, H * d, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(Lk, H * d, generator=g).to(torch.bfloat16)
    if Lk > 70:  # one dominant key for some queries: exercises the running-max rescale
        k[69] *= 4.0
    ref = wr.attention(q.float().view(Lq, H, d), k.float().view(Lk, H, d), v.float().view(Lk, H, d), klen).reshape(Lq, H * d)
    out = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), H, klen)
    err = (out.float().cpu() - ref).abs().max().item()
    assert err < 3e-2, err
    assert rel_err(out.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("Lq,Lk,H,klen,splits", [(300, 2100, 2, None, 2), (257, 2500, 1, 2437, 3), (64, 4096, 3, None, 4), (500, 130, 2, None, 5)])
def test_flash_attention_split_kv_matches_unsplit_and_fp32(Lq, Lk, H, klen, splits):
    """Split-KV (partials + merge kernel): same result as one workgroup per (query block, head) up to the fp32 merge order,
    and within the usual tolerance of the fp32 definition; ragged last tile in the last share, more splits than tiles."""
    from wan import ops

    d = 128
    g = torch.Generator().manual_seed(Lq + Lk + splits)
    q = (torch.randn(Lq, H * d, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    k = (torch.randn(Lk, H * d, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    v = torch.randn(Lk, H * d, generator=g).to(torch.bfloat16).to(DEV)
    k[Lk // 2 + 5] *= 4.0  # a dominant key in a later share: the merge must rescale the earlier partials
    one = ops.attention(q, k, v, H, klen, splits=1)
    many = ops.attention(q, k, v, H, klen, splits=splits)
    assert float((many.float() - one.float()).abs().max()) <= 2.0 ** -7 * float(one.float().abs().max())  # one bf16 rounding apart
    ref = wr.attention(q.float().cpu().view(Lq, H, d), k.float().cpu().view(Lk, H, d), v.float().cpu().view(Lk, H, d), klen).reshape(Lq, H * d)
    assert (many.float().cpu() - ref).abs().max().item() < 3e-2 and rel_err(many.float().cpu(), ref) < 1e-2


@pytest.mark.parametrize("Lq,Lk,H,klen", [(300, 300, 2, None), (515, 640, 12, 601), (1, 17, 1, None), (17, 65, 1, 33), (100, 512, 2, None),
                                           (130, 1100, 2, 1061), (257, 1500, 3, None), (2050, 2050, 4, None)])
def test_flash_attention_both_workgroup_forms_bit_identical(Lq, Lk, H, klen):
    """The bf16 kernel exists with 8 waves per workgroup (three ring stages, one workgroup per CU: the long self-attention) and with 4
    (two stages, two workgroups per CU: cross-attention and every key sequence up to 1024 by default).  `wanq_attention_select_form`
    forces either form, so that BOTH see the edge cases whatever the dispatch rule is: one query, one ragged tile, a masked second tile,
    ragged key counts on either side of the default bound.  Same arithmetic per wave -> bit-identical outputs; each within the usual
    tolerance of the fp32 definition."""
    from viditq_extension import _C
    from wan import ops

    d = 128
    g = torch.Generator().manual_seed(Lq * 31 + Lk)  # (the data of test_flash_attention_vs_fp32_softmax for the shapes they share)
    q = (torch.randn(Lq, H * d, generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(Lk, H * d, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(Lk, H * d, generator=g).to(torch.bfloat16)
    if Lk > 70:
        k[69] *= 4.0
    ref = wr.attention(q.float().view(Lq, H, d), k.float().view(Lk, H, d), v.float().view(Lk, H, d), klen).reshape(Lq, H * d)
    prev = _C.lib.wanq_attention_select_form(0)           # always 8 waves
    try:
        o8 = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), H, klen, splits=1)
        _C.lib.wanq_attention_select_form(1 << 40)        # always 4 waves
        o4 = ops.attention(q.to(DEV), k.to(DEV), v.to(DEV), H, klen, splits=1)
    finally:
        _C.lib.wanq_attention_select_form(prev)
    assert torch.equal(o8, o4)
    # accuracy of both (they are one result): the rel-Frobenius bar of test_flash_attention_vs_fp32_softmax, which carries the abs bar on its data
    assert rel_err(o8.float().cpu(), ref) < 1e-2


def test_attention_split_heuristic():
    from wan import ops

    dev = torch.device(DEV)
    assert ops.attention_splits(32760, 32760, 12, dev) == 1   # 1536 workgroups: 6 full rounds of 256 CUs
    assert ops.attention_splits(32760, 32760, 6, dev) == 1    # 3 full rounds
    assert ops.attention_splits(32760, 32760, 3, dev) == 2    # 1.5 rounds would cost 2
    assert ops.attention_splits(32760, 32760, 1, dev) >= 2    # one head alone fills half the GPU
    assert ops.attention_splits(32760, 512, 12, dev) == 1     # cross-attention: nothing to share


def test_flash_attention_strided_views_of_packed_qkv():
    from wan import ops

    L, H, d = 200, 2, 128
    qkv = torch.randn(L, 3 * H * d, generator=torch.Generator().manual_seed(8)).to(torch.bfloat16).to(DEV)
    q, k, v = qkv[:, : H * d], qkv[:, H * d: 2 * H * d], qkv[:, 2 * H * d:]
    out = ops.attention(q, k, v, H)
    ref = ops.attention(q.contiguous(), k.contiguous(), v.contiguous(), H)
    assert torch.equal(out, ref)


def test_kernel_mode_block_with_vidit_and_fp_layers_vs_oracle():
    """The reference's shipped configuration on one real-size block: ViDiT-Q (channel mask x Hadamard rotation)
    W8A8 on self_attn q/k/v, every other Linear left FP (remain_fp_regex) -- kernel-mode block vs the simulation
    oracle with the same masks and the same rotation signs."""
    from oracle import qdiff_ref as qr
    from qdiff import config as qcfg
    from qdiff.base.quant_model import quant_layer_refactor_
    from qdiff.utils import apply_func_to_submodules
    from wan import calib, ops
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    dim, ffn, heads, grid, lc = 1536, 8960, 12, (2, 6, 8), 64
    blk = make_block(dim, ffn, heads, 3)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(2)
    x = torch.randn(n_tok, dim, generator=g)
    x[:, 9] *= 15.0
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    act_mask = (torch.rand(dim, generator=g) * 3 + 0.2)

    cfg = qcfg.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                       "viditq": {"alpha": 0.5665, "layer_name_regex": ""},
                       "remain_fp_regex": r"self_attn\.(?!q$)(?!k$)(?!v$)[^.]+|ffn.*|cross_attn"})
    blk = blk.to(DEV)
    apply_func_to_submodules(blk, torch.nn.Linear, quant_layer_refactor_, name=None, parent_module=None, quant_config=cfg,
                             full_name=None, remain_fp_regex=cfg.remain_fp_regex)
    gen = torch.Generator().manual_seed(11)
    vidit = {}
    for name in ("q", "k", "v"):
        lin = getattr(blk.self_attn, name)
        assert type(lin).__name__ == "ViDiTQuantizedLinear"
        calib.init_rotation_and_channel_mask_(lin, "x", {"x": act_mask[None]}, gen)
        R = torch.from_numpy(qr.hadamard_from_signs(lin.rotation_signs.numpy()))
        vidit["self_attn." + name] = (lin.channel_mask.cpu(), R)
    assert type(blk.self_attn.o).__name__ == "Linear" and type(blk.ffn[0]).__name__ == "Linear"

    # oracle: fake-quant ViDiT on q/k/v, FP elsewhere
    lin = {}
    for nm in wr.LINEARS:
        w, b = sd[nm + ".weight"], sd[nm + ".bias"]
        lin[nm] = wr.FakeQuantLinear(w, b, 8, 8, False, *vidit[nm]) if nm in vidit else wr.FpLinear(w, b)
    norm_w = {k: sd[k + ".weight"].float() for k in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k")}
    ref = wr.BlockRef(lin, norm_w, sd["modulation"], heads, 1e-6, (sd["norm3.weight"].float(), sd["norm3.bias"].float()))(x, e0, grid, n_tok, ctx, freqs)

    hb = WanAttentionBlockWithHipKernel.from_float(blk, None)
    assert hb.self_attn.q.quantized and hb.self_attn.q.rot[0] == 12 and not hb.self_attn.o.quantized and not hb.ffn0.quantized
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16))
    err = rel_err(out.float().cpu(), ref)
    print(f"shipped-config block (ViDiT q/k/v + FP rest): rel err vs oracle {err:.2e}")
    assert err < 1e-2


@pytest.mark.parametrize("act", [{"n_bits": 8, "sym": False}, {"n_bits": 6, "sym": True}, {"n_bits": 4, "sym": False}])
def test_kernel_mode_block_with_asymmetric_or_narrow_activations_vs_simulation_oracle(act):
    """Kernel mode with an activation quantiser the fused producers do not implement -- asymmetric (Q/base/base_quantizer.py:130-149),
    below 8 bits -- on ALL ten Linears of a block, ViDiT mask + rotation on q / k / v (viditq_quant_layer.py:60-73 feeds whatever
    `a_quantizer` the config names): every such layer takes the unfused path (fp32 activation -> the layer's own quantiser kernels
    -> int8 GEMM -> zero-point rank-one term), against the simulation oracle with the same quantiser.  No Wan configuration selects
    these; the reference's kernel mode has no such path (its kernels are symmetric 8-bit), so the parity target is simulation mode."""
    from oracle import qdiff_ref as qr
    from qdiff import config as qcfg
    from qdiff.base.quant_model import quant_layer_refactor_
    from qdiff.utils import apply_func_to_submodules
    from wan import calib, ops
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    dim, ffn, heads, grid, lc = 256, 512, 2, (2, 5, 7), 24
    blk = make_block(dim, ffn, heads, 7)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(9)
    x = torch.randn(n_tok, dim, generator=g)
    x[:, 3] *= 10.0
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    act_mask = torch.rand(dim, generator=g) * 3 + 0.2
    cfg = qcfg.create({"weight": {"n_bits": 8, "sym": False}, "act": act, "viditq": {"alpha": 0.5665, "layer_name_regex": r"self_attn\.(q|k|v)$"}})
    blk = blk.to(DEV)
    apply_func_to_submodules(blk, torch.nn.Linear, quant_layer_refactor_, name=None, parent_module=None, quant_config=cfg, full_name=None,
                             remain_fp_regex=None)
    gen = torch.Generator().manual_seed(11)
    vidit = {}
    for name in ("q", "k", "v"):
        lin = getattr(blk.self_attn, name)
        assert type(lin).__name__ == "ViDiTQuantizedLinear", type(lin).__name__
        calib.init_rotation_and_channel_mask_(lin, "x", {"x": act_mask[None]}, gen)
        vidit["self_attn." + name] = (lin.channel_mask.cpu(), torch.from_numpy(qr.hadamard_from_signs(lin.rotation_signs.numpy())))
    ref = wr.block_from_state(sd, heads, quant=True, a_bits=act["n_bits"], a_sym=act["sym"], vidit=vidit)(x, e0, grid, n_tok, ctx, freqs)
    ref8 = wr.block_from_state(sd, heads, quant=True, vidit=vidit)(x, e0, grid, n_tok, ctx, freqs)
    fp = wr.block_from_state(sd, heads, quant=False)(x, e0, grid, n_tok, ctx, freqs)
    hb = WanAttentionBlockWithHipKernel.from_float(blk, None)
    assert all(l.act_quantizer is not None for l in (hb.self_attn.q, hb.self_attn.o, hb.cross_attn.k, hb.ffn0, hb.ffn2))
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    err, noise = rel_err(out, ref), rel_err(ref, fp)
    print(f"kernel-mode block, activations {act}: rel err vs its simulation oracle {err:.2e} (the recipe's own distance from FP {noise:.2e}; "
          f"8-bit symmetric recipe vs this one {rel_err(ref8, ref):.2e})")
    assert err < 0.5 * noise + 5e-3


def test_config5_viditq_block_14b_shapes_rotated_4bit_ffn_vs_oracle_and_quality():
    """quant_configs/w4a8_mixed_viditq.yaml on one 14B-shape block (dim 5120, ffn 13824, 40 heads): FFN weights 4 bit (packed), the
    rest 8, ViDiT-Q mask + rotation on self-attention q / k / v AND on ffn.0 (5120 = 40 x 128) / ffn.2 (13824 = 108 x 128: the width
    the reference's get_hadK asserts on; its own K = 108 branch here, csrc/rotate108.hip).  Masks from the activations the FP block
    really sees.  (a) parity: kernel-mode block vs the simulation oracle with the same masks / signs / bit-widths; (b) quality: the
    rotated 4-bit FFN against the plain 4-bit FFN of w4a8_mixed.yaml, both measured against the FP block."""
    import os

    from qdiff import config as qcfg
    from qdiff.base.quant_layer import QuantizedLinear
    from wan import calib, ops
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel
    from wan.quant_wanx_hip import _FpSrc

    dim, ffn, heads, grid, lc = 5120, 13824, 40, (1, 6, 8), 32
    qdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wan2.1-quantization_amd", "quant_configs")
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=1, text_dim=64, freq_dim=64).eval()
    blk = make_block(dim, ffn, heads, 0)
    g = torch.Generator().manual_seed(5)
    # outlier INPUT channels of the FFN weights, as trained DiTs have them: what a 4-bit grid per output row pays for
    for lin_ in (blk.ffn[0], blk.ffn[2]):
        cols = torch.randperm(lin_.in_features, generator=g)[: lin_.in_features // 200]
        lin_.weight.data[:, cols] *= 8.0
    fp.blocks[0].load_state_dict(blk.state_dict())
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    x = torch.randn(n_tok, dim, generator=g)
    x[:, 5] *= 12.0
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    norm_w = {k: sd[k + ".weight"].float() for k in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k")}
    norm3 = (sd["norm3.weight"].float(), sd["norm3.bias"].float())

    # ---- FP oracle block, recording every Linear's per-channel input absmax (what get_calib_data_wanx.py's hooks collect)
    seen = {}

    class Rec(wr.FpLinear):
        def __init__(self, name, w, b):
            super().__init__(w, b)
            self.name = name

        def __call__(self, xx):
            seen[self.name] = xx.abs().amax(dim=0)
            return super().__call__(xx)

    fp_ref = wr.BlockRef({nm: Rec(nm, sd[nm + ".weight"], sd[nm + ".bias"]) for nm in wr.LINEARS}, norm_w, sd["modulation"], heads, 1e-6, norm3)(
        x, e0, grid, n_tok, ctx, freqs)

    def build(cfg_name):
        model = QuantWanModel.from_float(fp, qcfg.load(os.path.join(qdir, cfg_name)))
        model.quant_layer_refactor()
        gen = torch.Generator().manual_seed(11)
        for name, mod in model.named_modules():
            if isinstance(mod, QuantizedLinear) and (mod.uses_mask or mod.uses_rotation):
                key = name.split("blocks.0.")[1]
                calib.init_rotation_and_channel_mask_(mod, name, {name: seen[key].clamp_min(1e-3)[None]}, gen)
        model.bitwidth_refactor()
        model.set_init_done()
        model.hardware_forward_refactor()
        return model

    rope = ops.rope_table(freqs, grid, DEV)
    run = lambda m: m.hip_blocks[0](x.to(DEV).clone(), e0.to(DEV), rope, n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()  # noqa: E731
    mv = build("w4a8_mixed_viditq.yaml")
    b0, hb = mv.blocks[0], mv.hip_blocks[0]
    assert b0.ffn[0].w_quantizer.n_bits == 4 and b0.ffn[2].w_quantizer.n_bits == 4 and b0.self_attn.q.w_quantizer.n_bits == 8
    assert hb.ffn0.w_bits == 4 and hb.ffn0.weight.dtype == torch.uint8 and hb.ffn0.rot[0] == 40 and hb.ffn2.rot[0] == 108
    assert hb.self_attn.q.rot[0] == 40 and hb.self_attn.o.rot is None
    out_v = run(mv)

    # ---- (a) parity against the simulation oracle with the same masks, signs and bit-widths
    lin = {}
    for nm in wr.LINEARS:
        owner = b0.self_attn if nm.startswith("self_attn") else b0.cross_attn if nm.startswith("cross_attn") else None
        ql = getattr(owner, nm.split(".")[1]) if owner is not None else b0.ffn[int(nm.split(".")[1])]
        bits = 4 if nm.startswith("ffn") else 8
        if ql.rotation_signs is not None:
            lin[nm] = wr.FakeQuantLinear(sd[nm + ".weight"], sd[nm + ".bias"], bits, 8, False, ql.channel_mask.float().cpu(),
                                         wr.hadamard_rotation(ql.rotation_signs.cpu(), strict=False))
        else:
            lin[nm] = wr.FakeQuantLinear(sd[nm + ".weight"], sd[nm + ".bias"], bits, 8, False)
    ref_v = wr.BlockRef(lin, norm_w, sd["modulation"], heads, 1e-6, norm3)(x, e0, grid, n_tok, ctx, freqs)
    err = rel_err(out_v, ref_v)
    # ---- (b) quality against the FP block: plain 4-bit FFN (w4a8_mixed.yaml) vs rotated 4-bit FFN
    out_p = run(build("w4a8_mixed.yaml"))

    def psnr(a, b):
        rng = (b.max() - b.min()).item()
        return 10 * np.log10(rng * rng / (a.double() - b.double()).pow(2).mean().item())

    ep, ev = rel_err(out_p, fp_ref), rel_err(out_v, fp_ref)
    print(f"config-5 block with ViDiT on q/k/v + ffn.0 / ffn.2: rel err vs its oracle {err:.2e}; against the FP block: plain W4 FFN rel L2 "
          f"{ep:.2e} / PSNR {psnr(out_p, fp_ref):.1f} dB, rotated W4 FFN rel L2 {ev:.2e} / PSNR {psnr(out_v, fp_ref):.1f} dB")
    assert err < 1.5e-2
    assert ev < 0.7 * ep  # the rotation buys back a good part of what the plain 4-bit grid loses to the outlier channels


def test_config5_w4a8_mixed_block_14b_shapes_and_checkpoint_roundtrip(tmp_path):
    """BASELINE config 5 on one 14B-shape block (dim 5120, ffn 13824, 40 heads): quant_configs/w4a8_mixed.yaml ->
    quant_layer_refactor + bitwidth_refactor (FFN weights 4 bit, the rest 8; Q/base/quant_model.py:76-105,
    mixed_precision_quantizer.py:56-186) -> kernel-mode block with the 4-bit weights PACKED in HBM and expanded inside the
    GEMM, vs the simulation oracle with 4-bit FFN fake-quant weights.  Then the integer checkpoint round trip: packed uint8
    [N, K/2] on disk, loaded back by hardware_forward_refactor into a second model, bit-equal output."""
    import os

    from qdiff import config as qcfg
    from qdiff.base.quant_layer import QuantizedLinear
    from wan import ops
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel
    from wan.quant_wanx_hip import _FpSrc

    dim, ffn, heads, grid, lc = 5120, 13824, 40, (1, 6, 8), 32
    cfgp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wan2.1-quantization_amd", "quant_configs", "w4a8_mixed.yaml")
    quant_config = qcfg.load(cfgp)
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=1, text_dim=64, freq_dim=64).eval()
    blk = make_block(dim, ffn, heads, 0)
    fp.blocks[0].load_state_dict(blk.state_dict())
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    model = QuantWanModel.from_float(fp, quant_config)
    model.quant_layer_refactor()
    model.bitwidth_refactor()
    model.set_init_done()
    b0 = model.blocks[0]
    assert b0.ffn[0].w_quantizer.n_bits == 4 and b0.ffn[2].w_quantizer.n_bits == 4 and b0.self_attn.q.w_quantizer.n_bits == 8
    assert b0.ffn[0]._codes.dtype == torch.uint8 and tuple(b0.ffn[0]._codes.shape) == (ffn, dim // 2)  # packed in the model
    model.hardware_forward_refactor()
    hb = model.hip_blocks[0]
    assert hb.ffn0.w_bits == 4 and hb.ffn0.weight.dtype == torch.uint8 and hb.ffn2.weight.shape == (dim, ffn // 2)
    assert hb.self_attn.q.w_bits == 8

    n_tok = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n_tok, dim, generator=g)
    x[:, 5] *= 12.0
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    lin = {}
    for nm in wr.LINEARS:
        lin[nm] = wr.FakeQuantLinear(sd[nm + ".weight"], sd[nm + ".bias"], 4 if nm.startswith("ffn") else 8, 8, False)
    norm_w = {k: sd[k + ".weight"].float() for k in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k")}
    ref = wr.BlockRef(lin, norm_w, sd["modulation"], heads, 1e-6, (sd["norm3.weight"].float(), sd["norm3.bias"].float()))(x, e0, grid, n_tok, ctx, freqs)
    ref8 = wr.block_from_state(sd, heads, quant=True)(x, e0, grid, n_tok, ctx, freqs)
    rope = ops.rope_table(freqs, grid, DEV)
    out = hb(x.to(DEV).clone(), e0.to(DEV), rope, n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    err = rel_err(out, ref)
    print(f"config-5 block (W4 FFN, W8 rest): rel err vs 4-bit oracle {err:.2e}; 4-bit vs 8-bit oracle {rel_err(ref, ref8):.2e}")
    assert err < 1.5e-2 and err < 0.5 * rel_err(ref, ref8) + 5e-3

    # ---- on-disk artefact: packed codes, loaded into a fresh model's kernel-mode blocks
    path = str(tmp_path / "int_weight.pt")
    sdw = model.quantize_and_save_weight(path)
    assert sdw["blocks.0.ffn.0.weight"].dtype == torch.uint8 and tuple(sdw["blocks.0.ffn.0.weight"].shape) == (ffn, dim // 2)
    assert sdw["blocks.0.self_attn.q.weight"].dtype == torch.int8
    model2 = QuantWanModel.from_float(fp, quant_config)
    model2.quant_layer_refactor()
    model2.bitwidth_refactor()
    model2.set_init_done()
    model2.hardware_forward_refactor()
    model2.hardware_forward_refactor(path)
    out2 = model2.hip_blocks[0](x.to(DEV).clone(), e0.to(DEV), rope, n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    assert torch.equal(out2, out)
    # the file, not the in-memory layers, is what the kernel-mode blocks hold afterwards
    tampered = dict(sdw)
    tampered["blocks.0.ffn.2.scale_weight"] = sdw["blocks.0.ffn.2.scale_weight"] * 2
    torch.save(tampered, path)
    model2.hardware_forward_refactor(path)
    out3 = model2.hip_blocks[0](x.to(DEV).clone(), e0.to(DEV), rope, n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    assert not torch.equal(out3, out)
    with pytest.raises(KeyError):
        bad = {k: v for k, v in sdw.items() if k != "blocks.0.cross_attn.o.scale_weight"}
        torch.save(bad, path)
        model2.hardware_forward_refactor(path)


def test_reference_format_checkpoint_layout_and_reload(tmp_path):
    """quantize_and_save_weight(reference_format=True) on a model quantised with the reference's shipped config.yaml (W8A8 on
    self_attn q / k / v, the rest FP): the state dict has the key set / dtypes the reference's kernel-mode loader reads (SURVEY
    Appendix C: `blocks.i.self_attn.{q,k,v}.{weight int8, scale_weight f16, zp_weight f16 (integer valued), bias f16}`, FP
    `o` / `cross_attn` / `ffn` as fp16 parameters, `norm1/2.weight` = ones, no fp_module / fp_weight / quantizer entries), and
    hardware_forward_refactor(path) loads it back: the kernel-mode output moves only by the fp16 rounding of the scales."""
    import os

    import yaml

    from qdiff import config as qcfg
    from wan.configs import seq_len_for
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wan2.1-quantization_amd", "quant_configs", "config.yaml")) as fh:
        raw = yaml.safe_load(fh)
    raw["viditq"] = None  # the reference's kernel path has no activation transform: export the plain W8A8 form of the same layers
    quant_config = qcfg.create({k: v for k, v in raw.items() if v is not None})
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64, freq_dim=64).eval()
    g = torch.Generator(device=DEV).manual_seed(2)
    torch.nn.init.xavier_uniform_(fp.head.head.weight, generator=g)
    model = QuantWanModel.from_float(fp, quant_config)
    model.quant_layer_refactor()
    model.set_init_done()
    path = str(tmp_path / "int_weight.pt")
    sd = model.quantize_and_save_weight(path, reference_format=True)
    for l in "qkv":
        base = f"blocks.1.self_attn.{l}"
        assert sd[base + ".weight"].dtype == torch.int8 and tuple(sd[base + ".weight"].shape) == (256, 256)
        assert sd[base + ".scale_weight"].dtype == torch.float16 and sd[base + ".zp_weight"].dtype == torch.float16 and sd[base + ".bias"].dtype == torch.float16
        assert torch.equal(sd[base + ".zp_weight"], sd[base + ".zp_weight"].round())  # integer valued: the loader copies it into int16
        assert base + ".act_premul" not in sd
    assert sd["blocks.0.self_attn.o.weight"].dtype == torch.float16 and sd["blocks.0.ffn.0.weight"].dtype == torch.float16
    assert torch.equal(sd["blocks.0.norm1.weight"], torch.ones(256, dtype=torch.float16)) and "blocks.1.norm2.weight" in sd
    assert not any(("fp_module" in k or "fp_weight" in k or "quantizer" in k) for k in sd)
    # reload into kernel mode
    shape = (16, 2, 8, 6)
    gl = torch.Generator(device=DEV).manual_seed(3)
    latent = torch.randn(shape, generator=gl, device=DEV)
    ctx = torch.randn(16, 64, generator=gl, device=DEV) * 0.1
    t = torch.tensor([500], device=DEV)
    model.hardware_forward_refactor()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        ref = model([latent], t, [ctx], seq_len_for(shape))[0].float()
    model.hardware_forward_refactor(path)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        out = model([latent], t, [ctx], seq_len_for(shape))[0].float()
    err = rel_err(out.cpu(), ref.cpu())
    print(f"kernel mode from the reference-format checkpoint vs from the in-memory layers: {err:.2e}")
    assert 0 < err < 5e-3  # fp16 scales / biases: not bit-equal, and not far


def _tiny_kernel_mode_model(raw_cfg, seed=0, dim=256, ffn=512, heads=2, layers=2):
    from qdiff import config as qcfg
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    quant_config = qcfg.create(raw_cfg)
    torch.manual_seed(seed)
    with torch.device(DEV):
        fp = WanModel(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=layers, text_dim=64, freq_dim=64).eval()
    g = torch.Generator(device=DEV).manual_seed(seed + 2)
    torch.nn.init.xavier_uniform_(fp.head.head.weight, generator=g)
    for m in fp.modules():
        if isinstance(m, torch.nn.Linear) and m.bias is not None:
            m.bias.data.normal_(std=0.02, generator=g)
    model = QuantWanModel.from_float(fp, quant_config)
    model.quant_layer_refactor()
    model.set_init_done()
    return model


def test_reference_format_roundtrip_with_symmetric_weights(tmp_path):
    """the reference's checkpoint format carries a `zp_weight` for every layer (all zeros when the weight quantiser
    is symmetric); the kernel-mode loader must take such a file back -- save(reference_format=True) -> load -> same output up to
    the fp16 rounding of the scales -- and must still refuse a NON-zero zero point for a layer that has none."""
    from wan.configs import seq_len_for

    model = _tiny_kernel_mode_model({"weight": {"n_bits": 8, "sym": True}, "act": {"n_bits": 8, "sym": True}, "remain_fp_regex": "head"})
    path = str(tmp_path / "int_weight.pt")
    sd = model.quantize_and_save_weight(path, reference_format=True)
    zp = sd["blocks.0.self_attn.q.zp_weight"]
    assert zp.dtype == torch.float16 and not bool(zp.any())
    shape = (16, 2, 8, 6)
    gl = torch.Generator(device=DEV).manual_seed(3)
    latent, ctx, t = torch.randn(shape, generator=gl, device=DEV), torch.randn(16, 64, generator=gl, device=DEV) * 0.1, torch.tensor([500], device=DEV)
    model.hardware_forward_refactor()
    assert model.hip_blocks[0].self_attn.q.zp_weight is None
    ref = model([latent], t, [ctx], seq_len_for(shape))[0].float()
    model.hardware_forward_refactor(path)
    out = model([latent], t, [ctx], seq_len_for(shape))[0].float()
    err = rel_err(out.cpu(), ref.cpu())
    print(f"symmetric-weight reference-format round trip: {err:.2e}")
    assert 0 < err < 5e-3
    bad = dict(sd)
    bad["blocks.0.self_attn.q.zp_weight"] = zp + 1
    torch.save(bad, path)
    with pytest.raises(KeyError):
        model.hardware_forward_refactor(path)


def test_context_kv_cache_is_bit_equal_and_follows_the_context_tensor():
    """cross_attn.k (+ RMSNorm) / cross_attn.v of the text context are step-invariant (W/wan/modules/model.py:178-200) and are
    kept per live context tensor across forward calls: same bits as recomputing them, fewer GEMM launches, recomputed for
    another tensor and after an in-place write to the same one."""
    from viditq_extension import qgemm
    from wan.configs import seq_len_for

    model = _tiny_kernel_mode_model({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True}, "remain_fp_regex": "head"})
    model.hardware_forward_refactor()
    shape = (16, 2, 8, 6)
    gl = torch.Generator(device=DEV).manual_seed(5)
    latent, t = torch.randn(shape, generator=gl, device=DEV), torch.tensor([500], device=DEV)
    ctx_a, ctx_b = (torch.randn(16, 64, generator=gl, device=DEV) * 0.1 for _ in range(2))
    sl = seq_len_for(shape)

    def run(ctx):
        tm = []
        qgemm.set_timer(tm)
        try:
            y = model([latent], t, [ctx], sl)[0]
        finally:
            qgemm.set_timer(None)
        return y, len(tm)

    model.context_cache = False
    ya, n_off = run(ctx_a)
    yb, _ = run(ctx_b)
    model.context_cache = True
    y1, n_first = run(ctx_a)
    y2, n_again = run(ctx_a)
    y3, n_other = run(ctx_b)
    assert torch.equal(y1, ya) and torch.equal(y2, ya) and torch.equal(y3, yb)
    nb = len(model.hip_blocks)
    assert n_first == n_off and n_other == n_off and n_again == n_off - 2 * nb  # k and v GEMMs of every block gone
    ctx_a.mul_(2.0)  # in-place write: the version counter moves and the entry is recomputed
    y4, n_written = run(ctx_a)
    model.context_cache = False
    y4_ref, _ = run(ctx_a)
    assert n_written == n_off and torch.equal(y4, y4_ref) and not torch.equal(y4, ya)


def test_oracle_evaluated_on_the_gpu_is_the_cpu_oracle():
    """oracle/wan_ref.py::BlockRef runs on either device (the whole-output parity tests of tests/test_gpu_fullsize.py evaluate it on
    the GPU): same block, same inputs, both devices.  The quantiser divisions go through fp64 on the GPU precisely so that both
    evaluations round the same quotients; what remains is fp32 summation order in LayerNorm / softmax in front of the quantisers,
    i.e. the recipe's code-flip floor (tests/test_model_golden.py), and exact agreement of the fake-quantised weights."""
    from oracle import wan_ref as wr

    dim, heads, ffn, Lx, lc, grid = 256, 2, 512, 4 * 6 * 5, 12, (4, 6, 5)
    g = torch.Generator().manual_seed(3)
    sd = {}
    for name in wr.LINEARS:
        n_out, n_in = (ffn, dim) if name == "ffn.0" else (dim, ffn) if name == "ffn.2" else (dim, dim)
        sd[name + ".weight"] = torch.randn(n_out, n_in, generator=g) / n_in ** 0.5
        sd[name + ".bias"] = torch.randn(n_out, generator=g) * 0.05
    for k in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k"):
        sd[k + ".weight"] = torch.rand(dim, generator=g) + 0.5
    sd["norm3.weight"], sd["norm3.bias"] = torch.rand(dim, generator=g) + 0.5, torch.randn(dim, generator=g) * 0.1
    sd["modulation"] = torch.randn(1, 6, dim, generator=g) / dim ** 0.5
    x, e0, ctx = torch.randn(Lx, dim, generator=g), torch.randn(1, 6, dim, generator=g) * 0.3, torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    cpu = wr.block_from_state(sd, heads, quant=True)
    dev = wr.block_from_state({k: v.to(DEV) for k, v in sd.items()}, heads, quant=True)
    for name in wr.LINEARS:
        assert torch.equal(cpu.lin[name].weight, dev.lin[name].weight.cpu()), name  # static quantiser: identical on both devices
    a = cpu(x, e0, grid, Lx, ctx, freqs)
    b = dev(x.to(DEV), e0.to(DEV), grid, Lx, ctx.to(DEV), freqs.to(DEV)).cpu()
    err = float((a.double() - b.double()).norm() / a.double().norm())
    assert err < 5e-3, err
    # the per-token quantiser alone, on identical inputs, IS identical (the division rule)
    t = torch.randn(64, dim, generator=g) * torch.exp(torch.randn(dim, generator=g))
    assert torch.equal(wr.dyn_fake_quant(t), wr.dyn_fake_quant(t.to(DEV)).cpu())
