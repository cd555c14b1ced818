"""Quantized Q.K^T attention: per-(token, head) symmetric int8 q / k, score matrix on the int8 matrix cores.
Oracle: the reference's recipe restated in oracle/wan_ref.py (qk_fake_quant / attention_qk_quant -- DynamicQuantizer over
head_dim for every (token, head), W/models/quant_opensora.py:431-436, Q/base/quant_attn.py:168-174) with fp32 softmax."""
import numpy as np
import pytest
import torch

from oracle import qdiff_ref as qr
from oracle import wan_ref as wr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rand(L, H, seed, spread=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(L, H * 128, generator=g)
    if spread:  # per-channel spread + a few outlier tokens, so that per-(token, head) scales really differ
        x = x * torch.exp(0.5 * torch.randn(H * 128, generator=g))
        x[:: max(1, L // 7)] *= 4.0
    return x


@pytest.mark.parametrize("L,H,dtype", [(70, 2, torch.float32), (300, 12, torch.bfloat16), (9, 40, torch.bfloat16)])
def test_rmsnorm_rope_q8_codes_and_scales_vs_oracle(L, H, dtype):
    """int8 codes / scales of the fused RMSNorm + RoPE + per-head quantiser vs the oracle chain rms_norm -> rope_apply (float64
    rotation) -> DynamicQuantizer on [tokens*heads, 128]: scales 1e-5 rel, codes within 1 LSB on < 0.5 % of the elements (the
    kernel normalises and rotates in fp32: codes move only at .5 boundaries); the constant plane is -12582912 * scale exactly."""
    from wan import ops

    C, d = H * 128, 128
    x = _rand(L, H, L + H).to(dtype)
    w = torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5
    grid = (1, 3, L // 3) if L % 3 == 0 else (1, 1, L)
    freqs = wr.rope_freqs(d)
    table = ops.rope_table(freqs, grid, DEV)
    q8, fp = ops.rmsnorm_rope_q8(x.to(DEV), w.to(DEV), table, d, True, want_fp=True)
    assert q8.stride % 64 == 0 and q8.stride >= L
    ref = wr.rope_apply(wr.rms_norm(x.float(), w, 1e-6).view(L, H, d), grid, freqs)          # [L, H, d] fp32
    oq, oscale = qr.dynamic_quantize_sym(ref.reshape(L * H, d).numpy())
    scale = q8.scales[0, :, :L].t().contiguous().cpu().numpy().reshape(-1)                      # [L*H]
    np.testing.assert_allclose(scale, oscale.reshape(-1), rtol=2e-5)
    np.testing.assert_array_equal(q8.scales[1, :, :L].cpu().numpy(), (-12582912.0 * q8.scales[0, :, :L]).cpu().numpy())
    dq = np.abs(q8.codes.cpu().numpy().astype(np.int32).reshape(L * H, d) - oq.astype(np.int32))
    assert dq.max() <= 1 and (dq != 0).mean() < 5e-3, (dq.max(), (dq != 0).mean())
    # the bf16 row written beside the codes is what the plain kernel writes
    plain = ops.rmsnorm_rope_(x.to(DEV).clone(), w.to(DEV), table, d, out=torch.empty(L, C, dtype=torch.bfloat16, device=DEV))
    assert torch.equal(fp, plain)


def _ref_from_codes(q8, k8, v, H, k_len=None):
    """fp32 definition on the kernel's own codes: softmax((q8*dq)(k8*dk)^T / sqrt(d)) v."""
    Lq, Lk = q8.codes.shape[0], k8.codes.shape[0]
    q = q8.codes.float().view(Lq, H, 128) * q8.scales[0, :, :Lq].t().unsqueeze(-1)
    k = k8.codes.float().view(Lk, H, 128) * k8.scales[0, :, :Lk].t().unsqueeze(-1)
    return wr.attention(q.cpu(), k.cpu(), v.float().view(Lk, H, 128).cpu(), k_len).reshape(Lq, H * 128)


@pytest.mark.parametrize("Lq,Lk,H,klen,splits", [(256, 256, 2, None, 1), (300, 333, 3, None, 1), (513, 700, 2, 650, 1),
                                                (64, 40, 4, None, 1), (1000, 4096, 2, None, 1), (520, 4100, 1, 4000, 3),
                                                (4680, 4680, 12, None, None)])
def test_attention_qk8_vs_fp32_definition(Lq, Lk, H, klen, splits):
    """Ragged query / key counts, key-length masking, a short (cross-attention-like) key set, split-KV, and the cfg-A size."""
    from wan import ops

    w = torch.ones(H * 128, device=DEV)
    q8 = ops.rmsnorm_rope_q8(_rand(Lq, H, 1).to(DEV), w, None, 128, False)
    k8 = ops.rmsnorm_rope_q8(_rand(Lk, H, 2).to(DEV), w, None, 128, True)
    v = _rand(Lk, H, 3, spread=False).to(torch.bfloat16).to(DEV)
    out = ops.attention_qk8(q8, k8, v, H, klen, splits=splits).float().cpu()
    ref = _ref_from_codes(q8, k8, v, H, klen)
    assert float((out - ref).abs().max()) < 3e-2 and float((out - ref).norm() / ref.norm()) < 1e-2
    if splits and splits > 1:
        one = ops.attention_qk8(q8, k8, v, H, klen, splits=1).float().cpu()
        assert float((out - one).abs().max()) < 2e-2


def test_attention_qk8_integer_scores_are_exact():
    """Peaked scores: with codes +-127 on one matching key per query the int8 dot is 128*127*127 = 2064512 (the largest
    magnitude the magic-number accumulator has to carry) and softmax puts all the mass there: o[q] == v[key(q)] exactly."""
    from wan import ops

    L, H = 192, 2
    q8 = ops.Q8Rows(L, H * 128, 128, DEV)
    k8 = ops.Q8Rows(L, H * 128, 128, DEV, 64)
    g = torch.Generator().manual_seed(4)
    signs = (torch.randint(0, 2, (L, H * 128), generator=g) * 2 - 1).to(torch.int8)
    k8.codes.copy_((signs * 127).to(DEV))
    perm = torch.randperm(L, generator=g)
    q8.codes.copy_((signs[perm] * 127).to(DEV))          # query i matches key perm[i]: dot = +2064512; others ~ N(0, 127^2 sqrt(128))
    q8.scales[0].fill_(0.01)
    k8.scales[0].fill_(0.02)
    k8.scales[1].copy_(-12582912.0 * k8.scales[0])
    v = torch.randn(L, H * 128, generator=g).to(torch.bfloat16).to(DEV)
    out = ops.attention_qk8(q8, k8, v, H)
    assert torch.equal(out.cpu(), v.cpu()[perm])


def test_attention_qk8_close_to_bf16_attention_and_oracle_recipe():
    """End to end against the reference recipe on fp32 inputs (quantisation inside the oracle): kernel vs oracle within the
    bf16 P.V tolerance, and the quantisation itself costs < 2 % against unquantised attention on these inputs."""
    from wan import ops

    L, H = 700, 4
    xq, xk = _rand(L, H, 7), _rand(L, H, 8)
    v = _rand(L, H, 9, spread=False).to(torch.bfloat16)
    w = torch.ones(H * 128)
    qn, kn = wr.rms_norm(xq, w, 1e-6).view(L, H, 128), wr.rms_norm(xk, w, 1e-6).view(L, H, 128)
    ref_q = wr.attention_qk_quant(qn, kn, v.float().view(L, H, 128)).reshape(L, H * 128)
    ref_fp = wr.attention(qn, kn, v.float().view(L, H, 128)).reshape(L, H * 128)
    q8 = ops.rmsnorm_rope_q8(xq.to(DEV), w.to(DEV), None, 128, False)
    k8 = ops.rmsnorm_rope_q8(xk.to(DEV), w.to(DEV), None, 128, True)
    out = ops.attention_qk8(q8, k8, v.to(DEV), H).float().cpu()
    e_oracle = float((out - ref_q).norm() / ref_q.norm())
    e_quant = float((ref_q - ref_fp).norm() / ref_fp.norm())
    print(f"int8 Q.K^T: kernel vs recipe oracle {e_oracle:.2e}; recipe vs unquantised attention {e_quant:.2e}")
    assert e_oracle < 1e-2 and e_quant < 2e-2


def test_kernel_mode_block_with_quantized_qk_vs_oracle():
    """attn.qk in the quant config: kernel-mode block (all linears W8A8 + int8 Q.K^T in self-attention) vs the simulation
    oracle with the same recipe (BlockRef(qk_bits=8))."""
    from test_gpu_block import make_block, rel_err
    from wan import ops
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    dim, ffn, heads, grid, lc = 1536, 8960, 12, (2, 6, 8), 64
    blk = make_block(dim, ffn, heads, 0)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n_tok, dim, generator=g)
    x[:, 5] *= 12.0
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    ref = wr.block_from_state(sd, heads, quant=True, qk_bits=8, cross_qk_bits=8)(x, e0, grid, n_tok, ctx, freqs)
    ref_noqk = wr.block_from_state(sd, heads, quant=True)(x, e0, grid, n_tok, ctx, freqs)
    hb = WanAttentionBlockWithHipKernel.from_float(blk.to(DEV), attn_qk8=True, cross_attn_qk8=True)
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    err = rel_err(out, ref)
    print(f"block with int8 Q.K^T: rel err vs recipe oracle {err:.2e}; recipe vs FP-attention oracle {rel_err(ref, ref_noqk):.2e}")
    assert err < 1e-2


@pytest.mark.parametrize("rows,cols,dtype,bits", [(777, 1536, torch.float32, 8), (300, 640, torch.bfloat16, 8), (64, 128, torch.float32, 4),
                                                  (1, 8, torch.float32, 8)])
def test_v_fake_quant_per_column_vs_oracle(rows, cols, dtype, bits):
    """attn.v: DynamicQuantizer over all tokens for every (head, channel) (W/models/quant_opensora.py:438-440)."""
    import viditq_extension.fused as fused

    g = torch.Generator().manual_seed(rows + cols)
    v = (torch.randn(rows, cols, generator=g) * torch.exp(torch.randn(cols, generator=g))).to(dtype)
    v[:, 3] = 0  # an all-zero channel: eps rule
    heads = max(1, cols // 128)
    ref = wr.v_fake_quant(v.float().view(rows, heads, cols // heads), bits).reshape(rows, cols)
    out, colmax = fused.fake_quant_cols_(v.to(DEV).clone(), bits)
    np.testing.assert_array_equal(colmax.cpu().numpy(), v.float().abs().amax(0).numpy())
    if dtype == torch.float32:
        np.testing.assert_array_equal(out.cpu().numpy(), ref.numpy())  # bit-exact: IEEE division, rne, fp32 product
    else:
        np.testing.assert_array_equal(out.float().cpu().numpy(), ref.to(dtype).float().numpy())
    assert out[:, 3].abs().max().item() == 0


def test_kernel_mode_block_with_quantized_v_vs_oracle(tmp_path):
    """attn.v / cross_attn.v (+ qk) through the quant config: block vs the simulation oracle with the same recipe; attn_map group
    'row' / 'column' is picked up, its 'block' mode refused with the reason."""
    from test_gpu_block import make_block, rel_err
    from wan import ops
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    dim, ffn, heads, grid, lc = 512, 1024, 4, (2, 6, 8), 64
    blk = make_block(dim, ffn, heads, 0)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n_tok, dim, generator=g)
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    ref = wr.block_from_state(sd, heads, quant=True, v_bits=8, cross_v_bits=8)(x, e0, grid, n_tok, ctx, freqs)
    ref_fp_v = wr.block_from_state(sd, heads, quant=True)(x, e0, grid, n_tok, ctx, freqs)
    hb = WanAttentionBlockWithHipKernel.from_float(blk.to(DEV), attn_v_bits=8, cross_attn_v_bits=8)
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    err = rel_err(out, ref)
    print(f"block with quantised v: rel err vs recipe oracle {err:.2e}; recipe vs FP-v oracle {rel_err(ref, ref_fp_v):.2e}")
    assert err < 1e-2

    # the config surface: attn.v is picked up by hardware_forward_refactor, attn.attn_map raises with the reason
    from qdiff import config as qcfg
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    base = {"model": {"model_id": "wan2.1", "model_type": "wanx"}, "remain_fp_regex": "text_embedding|time_embedding|time_projection|head\\.head",
            "weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True}}
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64, freq_dim=64).eval()
    m = QuantWanModel.from_float(fp, qcfg.create(dict(base, attn={"v": {"n_bits": 8, "sym": True}, "qk": {"n_bits": 8, "sym": True}})))
    m.quant_layer_refactor()
    m.set_init_done()
    m.hardware_forward_refactor()
    assert m.hip_blocks[0].attn_v_bits == 8 and m.hip_blocks[0].attn_qk8 and m.hip_blocks[0].cross_attn_v_bits is None
    m2 = QuantWanModel.from_float(fp, qcfg.create(dict(base, attn={"attn_map": {"n_bits": 8, "group": "block"}})))
    m2.quant_layer_refactor()
    m2.set_init_done()
    with pytest.raises(NotImplementedError, match="CogVideoX"):  # the 'block' mode is tied to another model's grid and reorder tables
        m2.hardware_forward_refactor()
    m3 = QuantWanModel.from_float(fp, qcfg.create(dict(base, attn={"attn_map": {"n_bits": 8, "sym": False, "group": "row"}},
                                                       cross_attn={"attn_map": {"n_bits": 4, "sym": True, "group": "column"}})))
    m3.quant_layer_refactor()
    m3.set_init_done()
    m3.hardware_forward_refactor()
    assert m3.hip_blocks[0].attn_map == (8, False) and m3.hip_blocks[0].cross_attn_map == (4, True)


@pytest.mark.parametrize("bits", [8, 4])
def test_v_fake_quant_kernel_vs_reference_golden(golden, bits):
    """wanq_col_absmax + wanq_fake_quant_cols on the token-major [tokens, heads*128] tensor == the reference's v quantiser
    (fixture a16_qkv_attn, made by its DynamicQuantizer with the reshape of quant_opensora.py:438-440), bit for bit."""
    import viditq_extension.fused as fused

    g = golden("a16_qkv_attn")
    v = torch.from_numpy(g["v"])[0].permute(1, 0, 2).reshape(g["v"].shape[2], -1).contiguous()      # [N, H*D]
    ref = torch.from_numpy(g[f"v{bits}"])[0].permute(1, 0, 2).reshape(v.shape)
    out, _ = fused.fake_quant_cols_(v.to(DEV).clone(), bits)
    assert torch.equal(out.cpu(), ref)


def test_qk_int8_codes_vs_reference_golden(golden):
    """The per-(token, head) int8 form the int8 Q.K^T attention consumes (wanq_rmsnorm_rope_q8, here with an identity rotation and
    no norm so that only its quantiser acts) dequantises to exactly what the reference's q / k quantiser returns (a16_qkv_attn)."""
    from wan import ops

    g = golden("a16_qkv_attn")
    n_tok, H, D = g["q"].shape[2], g["q"].shape[1], g["q"].shape[3]
    ident = torch.zeros(n_tok, D // 2, 2, device=DEV)
    ident[..., 0] = 1.0  # cos = 1, sin = 0
    for name in ("q", "k"):
        x = torch.from_numpy(g[name])[0].permute(1, 0, 2).reshape(n_tok, H * D).contiguous().to(DEV)
        ref = torch.from_numpy(g[f"{name}8"])[0].permute(1, 0, 2).reshape(n_tok, H * D)
        q8 = ops.rmsnorm_rope_q8(x, None, ident, D, for_keys=(name == "k"))
        delta = q8.scales[0, :, :n_tok].t().contiguous()  # [tokens, heads]
        deq = q8.codes.float().view(n_tok, H, D) * delta.unsqueeze(-1)
        assert torch.equal(deq.view(n_tok, H * D).cpu(), ref)


@pytest.mark.parametrize("Lq,Lk,H,klen,bits,sym", [(45, 45, 3, None, 8, False), (300, 300, 2, None, 8, True), (100, 512, 2, None, 8, False),
                                                    (515, 640, 4, 601, 8, False), (130, 77, 1, None, 4, True), (1000, 777, 2, None, 8, False)])
def test_attention_map_quant_vs_oracle(Lq, Lk, H, klen, bits, sym):
    """The streamed attention-map quantiser (three passes: row statistics, column maxima, quantised P.V) against the oracle's
    materialised form (oracle/wan_ref.py::attention_map_quant, pinned by the reference's own QuantizedAttentionMapOpenSORA in
    tests/golden/a16_attn_map.npz): ragged query / key counts, key masking, cross shapes, 8 and 4 bits, both quantiser forms.
    Tolerance: the flash-attention bar (P~ goes to the P.V MFMA as bf16) plus one quantisation step of a column."""
    from wan import ops

    d = 128
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    q = (torch.randn(Lq, H * d, generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(Lk, H * d, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(Lk, H * d, generator=g).to(torch.bfloat16)
    if Lk > 70:
        k[69] *= 3.0  # a dominant key column
    ref = wr.attention_map_quant(q.float().view(Lq, H, d), k.float().view(Lk, H, d), v.float().view(Lk, H, d), klen, bits, sym).reshape(Lq, H * d)
    fp = wr.attention(q.float().view(Lq, H, d), k.float().view(Lk, H, d), v.float().view(Lk, H, d), klen).reshape(Lq, H * d)
    out = ops.attention_map_quant(q.to(DEV), k.to(DEV), v.to(DEV), H, bits, sym, klen).float().cpu()
    err, noise = (out - ref).norm() / ref.norm(), (ref - fp).norm() / fp.norm()
    levels = (2 ** (bits - 1) - 1) if sym else (2 ** bits - 1)
    # a code may flip at a .5 boundary (the map is recomputed, exp2-based, on the GPU): one step of a column (<= 1 / levels) times |v|
    step = v.float().abs().max().item() / levels
    assert torch.isfinite(out).all() and (out - ref).abs().max().item() < 3e-2 + 1.5 * step, (out - ref).abs().max().item()
    assert err.item() < (1.2e-2 if bits == 8 else 0.5 * noise.item() + 1e-2), (err.item(), noise.item())  # 4 bits: well inside the recipe's own noise


@pytest.mark.parametrize("Lq,Lk,H,klen,bits,sym", [(45, 45, 3, None, 8, False), (300, 300, 2, None, 8, True), (515, 640, 4, 601, 8, False),
                                                    (100, 512, 2, None, 4, True), (1000, 777, 2, None, 8, False)])
def test_full_quantised_attention_recipe_vs_oracle(Lq, Lk, H, klen, bits, sym):
    """attn.qk + attn.v + attn.attn_map in ONE attention, as the reference applies them (W/models/quant_opensora.py:431-476):
    q / k as per-(token, head) int8 codes (S on the int8 matrix cores in all three passes), v fake-quantised per (head, channel),
    the map quantised per key column -- against the oracle's composition (pinned by the fixture's full_* keys,
    tests/test_oracle_golden.py).  The q / k codes are bit-exact by the quantiser tests; the bar is the map quantiser's."""
    from viditq_extension import fused
    from wan import ops

    d = 128
    g = torch.Generator().manual_seed(Lq * 5 + Lk)
    q = (torch.randn(Lq, H * d, generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(Lk, H * d, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(Lk, H * d, generator=g).to(torch.bfloat16)
    if Lk > 70:
        k[69] *= 3.0
    kl = Lk if klen is None else klen
    qf, kf, vf = q.float().view(Lq, H, d), k.float().view(Lk, H, d), v.float().view(Lk, H, d)
    vq = torch.cat([wr.v_fake_quant(vf[:kl], 8), vf[kl:]])
    ref = wr.attention_map_quant(wr.qk_fake_quant(qf, 8), wr.qk_fake_quant(kf, 8), vq, klen, bits, sym).reshape(Lq, H * d)
    ident = torch.zeros(max(Lq, Lk), d // 2, 2, device=DEV)
    ident[..., 0] = 1.0  # rotary = identity, RMSNorm weight None: the kernel only quantises
    q8 = ops.rmsnorm_rope_q8(q.to(DEV), None, ident, d, False)
    k8 = ops.rmsnorm_rope_q8(k.to(DEV), None, ident, d, True)
    vd = v.to(DEV).clone()
    fused.fake_quant_cols_(vd[:kl], 8)
    out = ops.attention_map_quant(q8, k8, vd, H, bits, sym, klen).float().cpu()
    levels = (2 ** (bits - 1) - 1) if sym else (2 ** bits - 1)
    step = v.float().abs().max().item() / levels
    err = ((out - ref).norm() / ref.norm()).item()
    fp = wr.attention(qf, kf, vf, klen).reshape(Lq, H * d)
    noise = ((ref - fp).norm() / fp.norm()).item()
    assert torch.isfinite(out).all() and (out - ref).abs().max().item() < 3e-2 + 1.5 * step, (out - ref).abs().max().item()
    assert err < (1.5e-2 if bits == 8 else 0.5 * noise + 1e-2), (err, noise)


def test_full_quantised_attention_recipe_matches_reference_golden():
    """The same entry point on the fixture's inputs, against the reference's own output of its whole recipe (full_x_*)."""
    import os

    from viditq_extension import fused
    from wan import ops

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "a16_attn_map.npz")))
    q, k, v = (torch.from_numpy(g[n])[0].permute(1, 0, 2).reshape(45, 3 * 128).to(torch.bfloat16) for n in ("q", "k", "v"))
    ident = torch.zeros(45, 64, 2, device=DEV)
    ident[..., 0] = 1.0
    q8 = ops.rmsnorm_rope_q8(q.to(DEV), None, ident, 128, False)
    k8 = ops.rmsnorm_rope_q8(k.to(DEV), None, ident, 128, True)
    vd = v.to(DEV).clone()
    fused.fake_quant_cols_(vd, 8)
    for tag, bits, sym in (("8a", 8, False), ("8s", 8, True), ("4s", 4, True)):
        out = ops.attention_map_quant(q8, k8, vd, 3, bits, sym).float().cpu().view(45, 3, 128)
        ref = torch.from_numpy(g[f"full_x_{tag}"])[0].permute(1, 0, 2)
        # the fixture's q / k / v are fp32; rounding them to bf16 in front of the quantisers moves codes by at most one step
        assert ((out - ref).norm() / ref.norm()).item() < (6e-2 if bits == 4 else 3e-2)


def test_kernel_mode_block_with_the_full_quantised_attention_recipe_vs_oracle():
    """A kernel-mode block with attn.qk + attn.v + attn.attn_map (and the same for cross-attention) against the simulation oracle
    with the same three quantisers (BlockRef(qk_bits, v_bits, attn_map)); the config surface accepts the combination."""
    from test_gpu_block import make_block, rel_err
    from wan import ops
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    dim, ffn, heads, grid, lc = 512, 1024, 4, (2, 6, 8), 64
    blk = make_block(dim, ffn, heads, 0)
    sd = {k_: v_.detach().clone() for k_, v_ in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n_tok, dim, generator=g)
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    kw = dict(qk_bits=8, cross_qk_bits=8, v_bits=8, cross_v_bits=8, attn_map=(8, False), cross_attn_map=(8, True))
    ref = wr.block_from_state(sd, heads, quant=True, **kw)(x, e0, grid, n_tok, ctx, freqs)
    ref_plain = wr.block_from_state(sd, heads, quant=True)(x, e0, grid, n_tok, ctx, freqs)
    hb = WanAttentionBlockWithHipKernel.from_float(blk.to(DEV), attn_qk8=True, cross_attn_qk8=True, attn_v_bits=8, cross_attn_v_bits=8,
                                                   attn_map=(8, False), cross_attn_map=(8, True))
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    err = rel_err(out, ref)
    print(f"block with the full quantised-attention recipe: rel err vs recipe oracle {err:.2e}; recipe vs plain-attention oracle {rel_err(ref, ref_plain):.2e}")
    assert err < 1.5e-2

    from qdiff import config as qcfg
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    base = {"model": {"model_id": "wan2.1", "model_type": "wanx"}, "remain_fp_regex": "text_embedding|time_embedding|time_projection|head\\.head",
            "weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True}}
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64, freq_dim=64).eval()
    full = {"qk": {"n_bits": 8, "sym": True}, "v": {"n_bits": 8, "sym": True}, "attn_map": {"n_bits": 8, "sym": False, "group": "row"}}
    m = QuantWanModel.from_float(fp, qcfg.create(dict(base, attn=full)))
    m.quant_layer_refactor()
    m.set_init_done()
    m.hardware_forward_refactor()
    b0 = m.hip_blocks[0]
    assert b0.attn_qk8 and b0.attn_v_bits == 8 and b0.attn_map == (8, False) and b0.cross_attn_map is None
    lat = torch.randn(16, 2, 8, 6, device=DEV)
    y = m([lat], torch.tensor([300], device=DEV), [torch.randn(16, 64, device=DEV) * 0.1], 24)[0]
    assert torch.isfinite(y).all()


def test_attention_map_quant_ignores_padded_query_rows():
    """with a padded sequence (rows >= q_len are padding, e.g. the last rank's tail under Ulysses) a key column's
    quantisation step must come from the REAL queries only, as in the reference, whose map has no padding rows.  The padded
    rows here are built to dominate every column maximum if they were counted; with q_len the real rows are bit-equal to the
    unpadded call and the padded output rows are zero."""
    from wan import ops

    d, H, Lq, pad, Lk = 128, 2, 83, 13, 150
    g = torch.Generator().manual_seed(11)
    q = (torch.randn(Lq + pad, H * d, generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(Lk, H * d, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(Lk, H * d, generator=g).to(torch.bfloat16)
    q[Lq:] = (k[:pad].float() * 4.0).to(torch.bfloat16)  # padding rows that put ~all their mass on one key each
    ref = ops.attention_map_quant(q[:Lq].contiguous().to(DEV), k.to(DEV), v.to(DEV), H, 8, False)
    out = ops.attention_map_quant(q.to(DEV), k.to(DEV), v.to(DEV), H, 8, False, q_len=Lq)
    assert torch.equal(out[:Lq], ref) and not bool(out[Lq:].any())
    counted = ops.attention_map_quant(q.to(DEV), k.to(DEV), v.to(DEV), H, 8, False)  # padding counted: other steps, other rows
    assert not torch.equal(counted[:Lq], ref)


def test_attention_map_quant_matches_reference_golden(golden=None):
    """The same entry point on the inputs of the reference-generated fixture (q, k, v as bf16): against the reference's own
    `attn_quantised @ v`."""
    import os

    from wan import ops

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "a16_attn_map.npz")))
    q, k, v = (torch.from_numpy(g[n])[0].permute(1, 0, 2).reshape(45, 3 * 128).to(torch.bfloat16) for n in ("q", "k", "v"))
    for tag, bits, sym in (("8a", 8, False), ("8s", 8, True), ("4s", 4, True)):
        out = ops.attention_map_quant(q.to(DEV), k.to(DEV), v.to(DEV), 3, bits, sym).float().cpu().view(45, 3, 128)
        ref = torch.from_numpy(g[f"x_{tag}"])[0].permute(1, 0, 2)
        # the fixture's q / k / v are fp32; rounding them to bf16 for the kernel moves the scores by 2^-8 relative
        assert ((out - ref).norm() / ref.norm()).item() < (6e-2 if bits == 4 else 3e-2)


def test_kernel_mode_block_with_quantized_attention_map_vs_oracle():
    """attn.attn_map / cross_attn.attn_map: kernel-mode block (all linears W8A8, streamed attention-map quantiser in both
    attentions) vs the simulation oracle with the same recipe (BlockRef(attn_map=...), materialised map)."""
    from test_gpu_block import make_block, rel_err
    from wan import ops
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    dim, ffn, heads, grid, lc = 512, 1024, 4, (2, 6, 8), 64
    blk = make_block(dim, ffn, heads, 0)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    n_tok = grid[0] * grid[1] * grid[2]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n_tok, dim, generator=g)
    e0 = torch.randn(1, 6, dim, generator=g) * 0.3
    ctx = torch.randn(lc, dim, generator=g)
    freqs = wr.rope_freqs(dim // heads)
    ref = wr.block_from_state(sd, heads, quant=True, attn_map=(8, False), cross_attn_map=(8, True))(x, e0, grid, n_tok, ctx, freqs)
    ref_fp_map = wr.block_from_state(sd, heads, quant=True)(x, e0, grid, n_tok, ctx, freqs)
    hb = WanAttentionBlockWithHipKernel.from_float(blk.to(DEV), attn_map=(8, False), cross_attn_map=(8, True))
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), n_tok, _FpSrc(ctx.to(DEV), torch.bfloat16)).float().cpu()
    err = rel_err(out, ref)
    print(f"block with quantised attention map: rel err vs recipe oracle {err:.2e}; recipe vs FP-map oracle {rel_err(ref, ref_fp_map):.2e}")
    assert err < 1e-2
