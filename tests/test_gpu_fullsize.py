"""Parity at BASELINE.json's full sizes (cfg-B: L = 32760 tokens, C = 1536, F = 8960, 12 heads), where the CPU oracle cannot
cover every element in seconds: size-independent properties over the WHOLE output (integer checksums of checksums, per-row
quantiser invariants, softmax partition of unity, key-permutation invariance) plus the oracle / an fp32 definition on a
random sample of rows."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref as kr

pytestmark = pytest.mark.gpu
DEV = "cuda"
L, C, F, H = 32760, 1536, 8960, 12


def _rand_i8(shape, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.randint(-128, 128, shape, dtype=torch.int8, device=DEV, generator=g)


@pytest.mark.parametrize("N,K", [(C, C), (F, C), (C, F)])
def test_gemm_accumulators_full_size_checksums_and_sampled_rows(N, K):
    import viditq_extension.qgemm as qgemm

    a, w = _rand_i8((L, K), 1), _rand_i8((N, K), 2)
    acc = qgemm.w8a8_o32(a, w)
    assert acc.shape == (L, N) and acc.dtype == torch.int32
    # checksum of checksums, exact in float64 (|values| < 2^53): row sums and column sums of A.W^T
    a64, w64 = a.double(), w.double()
    np.testing.assert_array_equal(acc.double().sum(1).cpu().numpy(), (a64 @ w64.sum(0)).cpu().numpy())
    np.testing.assert_array_equal(acc.double().sum(0).cpu().numpy(), (w64 @ a64.sum(0)).cpu().numpy())
    # sampled rows against the oracle, bit for bit (first / last rows of the ragged last m-tile included)
    rows = np.unique(np.concatenate([np.random.default_rng(0).integers(0, L, 48), [0, 255, 256, L - 249, L - 1]]))
    ref = kr.w8a8_o32(a[rows].cpu().numpy(), w.cpu().numpy())
    np.testing.assert_array_equal(acc[rows].cpu().numpy(), ref)


@pytest.mark.parametrize("cols,dtype", [(C, torch.float32), (C, torch.bfloat16), (F, torch.bfloat16)])
def test_quantize_full_size_row_invariants_and_sampled_rows(cols, dtype):
    import viditq_extension.fused as fused

    g = torch.Generator(device=DEV).manual_seed(cols)
    x = (torch.randn(L, cols, device=DEV, generator=g) * torch.exp(torch.randn(cols, device=DEV, generator=g))).to(dtype)
    x[17] = 0  # eps branch
    scale = torch.zeros(L, dtype=torch.float32, device=DEV)
    ssum = torch.zeros(L, dtype=torch.float32, device=DEV)
    q = fused.quant_sum(x, ssum, scale)
    xf = x.float()
    amax = xf.abs().amax(1)
    want = torch.maximum(amax / torch.full_like(amax, 127.0), torch.full_like(amax, 1e-6))  # IEEE division
    assert torch.equal(scale, want)
    qi = q.int()
    assert int(qi.abs().max()) <= 127
    live = amax > 1e-3
    assert bool((qi.abs().amax(1)[live] == 127).all())          # the row maximum maps to +-127
    assert int(qi[17].abs().max()) == 0
    err = (xf - qi.float() * scale[:, None]).abs()
    # round-to-nearest: half a step, plus the fp32 roundings of x/s and q*s themselves
    assert bool((err <= scale[:, None] * 0.5 + xf.abs() * 3e-7).all())
    np.testing.assert_allclose(ssum.cpu().numpy(), (qi.sum(1).double() * scale.double()).cpu().numpy(), rtol=1e-6, atol=1e-6)
    rows = np.unique(np.concatenate([np.random.default_rng(1).integers(0, L, 40), [0, 17, L - 1]]))
    oq, oscale, _ = kr.quant_sum(xf[rows].cpu().numpy())
    np.testing.assert_array_equal(q[rows].cpu().numpy(), oq)
    np.testing.assert_array_equal(scale[rows].cpu().numpy(), oscale)


def test_attention_full_size_properties_and_sampled_queries():
    from wan import ops

    g = torch.Generator(device=DEV).manual_seed(3)
    q = torch.randn(L, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    k = torch.randn(L, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(L, H * 128, device=DEV, generator=g).to(torch.bfloat16)
    o = ops.attention(q, k, v, H)
    assert o.shape == q.shape and bool(torch.isfinite(o.float()).all())
    # partition of unity: with V constant along the keys the output is that constant, whatever the scores
    vc = torch.randn(1, H * 128, device=DEV, generator=g).to(torch.bfloat16).expand(L, -1).contiguous()
    oc = ops.attention(q, k, vc, H).float()
    assert float((oc - vc.float()).abs().max()) <= 2.0 ** -7 * float(vc.float().abs().max())  # one bf16 rounding of O
    # key order does not matter (only the fp32 summation order changes)
    perm = torch.randperm(L, device=DEV, generator=g)
    op = ops.attention(q, k[perm].contiguous(), v[perm].contiguous(), H).float()
    assert float((op - o.float()).abs().max()) < 2e-2
    # sampled queries against the fp32 definition softmax(q k^T / sqrt(d)) v over all 32760 keys
    rows = torch.from_numpy(np.unique(np.concatenate([np.random.default_rng(2).integers(0, L, 60), [0, 255, 256, L - 1]]))).to(DEV)
    qs = q[rows].float().view(-1, H, 128).transpose(0, 1)                    # [H, S, d]
    kk, vv = k.float().view(L, H, 128).transpose(0, 1), v.float().view(L, H, 128).transpose(0, 1)
    ref = torch.softmax(qs @ kk.transpose(1, 2) / 128 ** 0.5, dim=-1) @ vv   # [H, S, d]
    ref = ref.transpose(0, 1).reshape(len(rows), H * 128)
    got = o[rows].float()
    assert float((got - ref).abs().max()) < 3e-2
    assert float((got - ref).norm() / ref.norm()) < 1e-2


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE config 4 / 5: Wan2.1-T2V-14B at 1280x720x81f under Ulysses = 8.  One rank holds 75600 / 8 = 9450 tokens (a ragged
# count: 36 full 256-row tiles + 234 rows), C = 5120, F = 13824, and attends with its 40 / 8 = 5 heads over all 75600 keys.
LR, C14, F14, HR, LK14 = 9450, 5120, 13824, 5, 75600


@pytest.mark.parametrize("N,K,w4", [(C14, C14, False), (F14, C14, False), (C14, F14, False), (F14, C14, True), (C14, F14, True)])
def test_config4_per_rank_gemm_checksums_and_sampled_rows(N, K, w4):
    import viditq_extension.qgemm as qgemm

    a = _rand_i8((LR, K), 11)
    if w4:
        g = torch.Generator(device=DEV).manual_seed(12)
        w = torch.randint(0, 16, (N, K), dtype=torch.int8, device=DEV, generator=g)
        acc = qgemm.w4a8_o32(a, qgemm.pack_w4(w, bias=0))
    else:
        w = _rand_i8((N, K), 12)
        acc = qgemm.w8a8_o32(a, w)
    assert acc.shape == (LR, N) and acc.dtype == torch.int32
    a64, w64 = a.double(), w.double()
    np.testing.assert_array_equal(acc.double().sum(1).cpu().numpy(), (a64 @ w64.sum(0)).cpu().numpy())
    np.testing.assert_array_equal(acc.double().sum(0).cpu().numpy(), (w64 @ a64.sum(0)).cpu().numpy())
    rows = np.unique(np.concatenate([np.random.default_rng(4).integers(0, LR, 40), [0, 255, 256, 9215, 9216, LR - 1]]))
    np.testing.assert_array_equal(acc[rows].cpu().numpy(), kr.w8a8_o32(a[rows].cpu().numpy(), w.cpu().numpy()))


def test_config4_per_rank_quantize_13824_columns():
    import viditq_extension.fused as fused

    g = torch.Generator(device=DEV).manual_seed(13)
    x = (torch.randn(LR, F14, device=DEV, generator=g) * torch.exp(torch.randn(F14, device=DEV, generator=g))).to(torch.bfloat16)
    scale, ssum = torch.zeros(LR, device=DEV), torch.zeros(LR, device=DEV)
    q = fused.quant_sum(x, ssum, scale)
    amax = x.float().abs().amax(1)
    assert torch.equal(scale, torch.maximum(amax / torch.full_like(amax, 127.0), torch.full_like(amax, 1e-6)))
    assert bool((q.int().abs().amax(1) == 127).all())
    rows = np.unique(np.concatenate([np.random.default_rng(5).integers(0, LR, 24), [0, LR - 1]]))
    oq, oscale, _ = kr.quant_sum(x[rows].float().cpu().numpy())
    np.testing.assert_array_equal(q[rows].cpu().numpy(), oq)
    np.testing.assert_array_equal(scale[rows].cpu().numpy(), oscale)


def test_config4_per_rank_attention_5_heads_75600_keys():
    """The Ulysses-8 attention call of one rank: all 75600 queries and keys of its 5 heads (split-KV chosen by the library's own
    policy), sampled queries against the fp32 definition and the partition-of-unity property over the whole output."""
    from wan import ops

    g = torch.Generator(device=DEV).manual_seed(6)
    q = torch.randn(LK14, HR * 128, device=DEV, generator=g).to(torch.bfloat16)
    k = torch.randn(LK14, HR * 128, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(LK14, HR * 128, device=DEV, generator=g).to(torch.bfloat16)
    o = ops.attention(q, k, v, HR)
    assert bool(torch.isfinite(o.float()).all())
    vc = torch.randn(1, HR * 128, device=DEV, generator=g).to(torch.bfloat16).expand(LK14, -1).contiguous()
    oc = ops.attention(q, k, vc, HR).float()
    assert float((oc - vc.float()).abs().max()) <= 2.0 ** -7 * float(vc.float().abs().max())
    rows = torch.from_numpy(np.unique(np.concatenate([np.random.default_rng(7).integers(0, LK14, 40), [0, 255, LK14 - 1]]))).to(DEV)
    qs = q[rows].float().view(-1, HR, 128).transpose(0, 1)
    kk, vv = k.float().view(LK14, HR, 128).transpose(0, 1), v.float().view(LK14, HR, 128).transpose(0, 1)
    ref = (torch.softmax(qs @ kk.transpose(1, 2) / 128 ** 0.5, dim=-1) @ vv).transpose(0, 1).reshape(len(rows), HR * 128)
    got = o[rows].float()
    assert float((got - ref).abs().max()) < 3e-2 and float((got - ref).norm() / ref.norm()) < 1e-2


# ---------------------------------------------------------------------------------------------------------------------
# The attention-map quantiser at the headline size.  Its column parameters are statistics over ALL 32760 queries
# of a head, so the small-size oracle comparisons (tests/test_gpu_attn_qk8.py, Lq <= 1000) say nothing about 512-tile rings, the
# ragged last tile (32760 = 511 x 64 + 56) or 128 workgroups per head meeting in the column-maximum atomics.
def _column_maxima_and_rows(qf, kf, rows, chunk=4096):
    """fp32 definition, streamed: per (head, key) maximum of softmax(q k^T / sqrt(d)) over all queries, and the map rows of the
    sampled queries.  qf [Lq, H, d], kf [Lk, H, d] fp32 on the GPU -> colmax [H, Lk], p_rows [H, S, Lk]."""
    Lq, Hn, d = qf.shape
    kt = kf.permute(1, 2, 0).contiguous()  # [H, d, Lk]
    colmax = torch.zeros(Hn, kf.shape[0], device=qf.device)
    for a in range(0, Lq, chunk):
        p = torch.softmax(qf[a:a + chunk].transpose(0, 1) @ kt / d ** 0.5, dim=-1)  # [H, c, Lk]
        colmax = torch.maximum(colmax, p.amax(dim=1))
        del p
    p_rows = torch.softmax(qf[rows].transpose(0, 1) @ kt / d ** 0.5, dim=-1)
    return colmax, p_rows


@pytest.mark.parametrize("form,sym", [("bf16", False), ("qk8", False), ("bf16", True)])
def test_attention_map_quant_full_size_sampled_queries(form, sym):
    """wanq_attention_map_quant_fwd / _qk8_fwd at L = 32760 x 12 heads against the oracle's column quantiser
    (oracle/wan_ref.py::attn_map_fake_quant, pinned by the reference's QuantizedAttentionMapOpenSORA) on sampled query rows.  The
    oracle function is given the sampled rows of the fp32 map plus one row holding the streamed column maxima, so its column
    parameters are those of the whole map (softmax values are >= 0: the clamped column minimum is 0 either way)."""
    from oracle import wan_ref as wr
    from viditq_extension import fused
    from wan import ops

    d, bits = 128, 8
    g = torch.Generator(device=DEV).manual_seed(11)
    q = (torch.randn(L, H * d, device=DEV, generator=g) * 1.5).to(torch.bfloat16)
    k = (torch.randn(L, H * d, device=DEV, generator=g) * 1.5).to(torch.bfloat16)
    v = torch.randn(L, H * d, device=DEV, generator=g).to(torch.bfloat16)
    k[69] *= 3.0  # a dominant key column
    rows = torch.from_numpy(np.unique(np.concatenate([np.random.default_rng(5).integers(0, L, 40), [0, 255, 256, L - 57, L - 1]]))).to(DEV)
    if form == "qk8":
        ident = torch.zeros(L, d // 2, 2, device=DEV)
        ident[..., 0] = 1.0  # rotary = identity, no RMSNorm weight: the kernel only quantises
        q8, k8 = ops.rmsnorm_rope_q8(q, None, ident, d, False), ops.rmsnorm_rope_q8(k, None, ident, d, True)
        qf = q8.codes.float().view(L, H, d) * q8.scales[0, :, :L].t().unsqueeze(-1)  # the dequantised operands (codes bit-exact by
        kf = k8.codes.float().view(L, H, d) * k8.scales[0, :, :L].t().unsqueeze(-1)  # tests/test_gpu_attn_qk8.py)
        vd = v.clone()
        fused.fake_quant_cols_(vd, 8)  # attn.v of the full recipe (bit-exact by its own golden test)
        out = ops.attention_map_quant(q8, k8, vd, H, bits, sym)
    else:
        qf, kf, vd = q.float().view(L, H, d), k.float().view(L, H, d), v
        out = ops.attention_map_quant(q, k, v, H, bits, sym)
    assert out.shape == (L, H * d) and bool(torch.isfinite(out.float()).all())
    colmax, p_rows = _column_maxima_and_rows(qf, kf, rows)
    S = len(rows)
    pq = wr.attn_map_fake_quant(torch.cat([p_rows, colmax[:, None, :]], dim=1).cpu(), bits, sym)[:, :S].to(DEV)  # [H, S, Lk]; the oracle on the CPU
    vv = vd.float().view(L, H, d).transpose(0, 1)                                                   # [H, Lk, d]
    ref = (pq @ vv).transpose(0, 1).reshape(S, H * d)
    fp = (p_rows @ vv).transpose(0, 1).reshape(S, H * d)
    got = out[rows].float()
    levels = (2 ** (bits - 1) - 1) if sym else (2 ** bits - 1)
    step = float(colmax.max()) / levels * float(vd.float().abs().max())  # one code of the widest column
    err, noise = float((got - ref).norm() / ref.norm()), float((ref - fp).norm() / fp.norm())
    # bar: the small-size test's (P~ goes to the P.V MFMA as bf16; a code may flip at a .5 boundary because the GPU recomputes the
    # map exp2-based), and well inside the recipe's own distance from FP attention
    assert float((got - ref).abs().max()) < 3e-2 + 1.5 * step, float((got - ref).abs().max())
    assert err < 1.2e-2 and err < 0.5 * noise, (err, noise)  # measured: 6.4e-3 against a recipe noise of 2.2e-2 (bf16, asymmetric)


# ---------------------------------------------------------------------------------------------------------------------
# The two row-wise kernels of the headline step that the tests above do not reach at L = 32760: the q / k / v transform
# (LayerNorm + modulation, three channel masks, Hadamard rotation, per-token int8: rotate_kernel<12, ..., LN, MULTI>) and the
# FFN's GELU + quantise (quant_rows_wave_kernel, one wave per [8960] row).
def test_qkv_transform_full_size_sampled_rows():
    import viditq_extension.fused as fused
    from oracle import qdiff_ref as qr
    from qdiff.quarot import quarot_utils as qu

    g = torch.Generator(device=DEV).manual_seed(1536)
    x = torch.randn(L, C, device=DEV, generator=g) * 2 + 0.1
    x[:, 7] *= 20.0  # an outlier channel, as the synthetic model has them
    sh, sc = torch.randn(1, C, device=DEV, generator=g) * 0.2, torch.randn(1, C, device=DEV, generator=g) * 0.2
    pms = [(torch.rand(C, device=DEV, generator=g) + 0.5) * (torch.randint(0, 2, (C,), device=DEV, generator=g) * 2 - 1).float() for _ in range(3)]
    rot = qu.kernel_rotation_params(C, DEV)
    qs = [torch.empty(L, C, dtype=torch.int8, device=DEV) for _ in range(3)]
    scales, sums = [torch.zeros(L, device=DEV) for _ in range(3)], [torch.zeros(L, device=DEV) for _ in range(3)]
    fused.layernorm_rotate_quant_multi(qs, x, None, sh, sc, pms, rot, sums, scales, 1e-6)
    rows = np.unique(np.concatenate([np.random.default_rng(4).integers(0, L, 40), [0, 1, L - 2, L - 1]]))
    h = kr.layernorm_t2i(x[rows].cpu().numpy(), None, sh.cpu().numpy(), sc.cpu().numpy(), 1e-6, len(rows))
    for i in range(3):
        # whole output: the codes reach +-127 in every row and the row sums are those of the codes
        qi = qs[i].to(torch.int32)
        assert bool((qi.abs().amax(1) == 127).all())
        np.testing.assert_allclose(sums[i].cpu().numpy(), qi.sum(1).cpu().numpy().astype(np.float64) * scales[i].cpu().numpy().astype(np.float64),
                                   rtol=1e-6, atol=1e-6)
        # sampled rows: the oracle's fp64 rotation of the fp64 LayerNorm (the kernel's transform is fp32: a code moves only at a .5 tie)
        ref = qr.matmul_hadU(h * pms[i].double().cpu().numpy()).astype(np.float32)
        oq, oscale = qr.dynamic_quantize_sym(ref)
        np.testing.assert_allclose(scales[i][rows].cpu().numpy(), oscale, rtol=1e-5)
        d = np.abs(qs[i][rows].cpu().numpy().astype(np.int32) - oq)
        assert d.max() <= 1 and (d != 0).mean() < 3e-3, (d.max(), (d != 0).mean())


def test_gelu_quantise_full_size_sampled_rows():
    import viditq_extension.fused as fused

    g = torch.Generator(device=DEV).manual_seed(8960)
    x = (torch.randn(L, F, device=DEV, generator=g) * torch.exp(0.5 * torch.randn(F, device=DEV, generator=g))).to(torch.bfloat16)
    x[17] = 0  # GELU(0) = 0: the eps branch
    scale, ssum = torch.zeros(L, device=DEV), torch.zeros(L, device=DEV)
    q = fused.gelu_quant_sum(x, ssum, scale)
    qi = q.to(torch.int32)
    live = torch.ones(L, dtype=torch.bool, device=DEV)
    live[17] = False
    assert bool((qi.abs().amax(1)[live] == 127).all()) and int(qi[17].abs().max()) == 0 and scale[17].item() == pytest.approx(1e-6)
    np.testing.assert_allclose(ssum.cpu().numpy(), qi.sum(1).cpu().numpy().astype(np.float64) * scale.cpu().numpy().astype(np.float64), rtol=1e-6, atol=1e-6)
    rows = np.unique(np.concatenate([np.random.default_rng(6).integers(0, L, 40), [0, 16, 17, 18, L - 1]]))
    oq, oscale, _ = kr.gelu_quant_sum(x[rows].float().cpu().numpy())
    # the kernel's GELU is the exp / rcp form (relative error < 3e-6 against tanhf): scales to 1e-5, codes move at ties
    np.testing.assert_allclose(scale[rows].cpu().numpy(), oscale, rtol=1e-5)
    d = np.abs(q[rows].cpu().numpy().astype(np.int32) - oq.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())


# ---------------------------------------------------------------------------------------------------------------------
# The GEMM's fused epilogues at the headline shapes (the accumulator test above stores raw int32): the 16-bit store loop, the
# GELU form, and the fp32 + gate + residual form that rewrites the residual stream IN PLACE with its lines prefetched by LDS-DMA
# (the path whose missing barrier once survived every small test).  Whole output against an fp64 evaluation of the
# epilogue formula on the exact integer product (int8 values and their dot products are exact in fp64), sampled rows against the
# oracle's left-to-right fp32 evaluation (K/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu:416-441).
@pytest.mark.parametrize("N,K,form", [(C, C, "bf16"), (F, C, "bf16_gelu"), (C, C, "f32_gate_res"), (C, F, "f32_gate_res")])
def test_gemm_epilogues_full_size_whole_output(N, K, form):
    import viditq_extension.qgemm as qgemm

    g = torch.Generator(device=DEV).manual_seed(N * 3 + K)
    a, w = _rand_i8((L, K), 5), _rand_i8((N, K), 6)
    sa = torch.rand(L, device=DEV, generator=g) * 0.02 + 1e-3
    asum = a.float().sum(1) * sa
    sw = torch.rand(N, device=DEV, generator=g) * 0.01 + 1e-4
    zp = torch.randint(-20, 140, (N,), device=DEV, generator=g).float()
    bias = torch.randn(N, device=DEV, generator=g)
    acc64 = a.double() @ w.double().t()  # exact
    y64 = acc64 * sa.double()[:, None] * sw.double()[None, :] + asum.double()[:, None] * (zp.double() * sw.double())[None, :] + bias.double()[None, :]
    del acc64
    rows = np.unique(np.concatenate([np.random.default_rng(9).integers(0, L, 24), [0, 255, 256, L - 249, L - 1]]))
    yo = kr.w8a8_epilogue(kr.w8a8_o32(a[rows].cpu().numpy(), w.cpu().numpy()), sa[rows].cpu().numpy(), sw.cpu().numpy(), bias.cpu().numpy(),
                          asum[rows].cpu().numpy(), zp.cpu().numpy())
    if form == "f32_gate_res":
        gate = torch.randn(N, device=DEV, generator=g)
        x = torch.randn(L, N, device=DEV, generator=g)
        want = x.double() + y64 * gate.double()[None, :]
        xo = x[rows].cpu().numpy()
        out = qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.float32, gate=gate, residual=x, out=x)  # in place
        assert out.data_ptr() == x.data_ptr()
        scale = float(want.abs().max())
        assert float((out.double() - want).abs().max()) < 4e-6 * scale  # a few fp32 roundings of values up to `scale`
        np.testing.assert_allclose(out[rows].cpu().numpy(), xo + yo * gate.cpu().numpy()[None, :], rtol=0, atol=4e-6 * scale)
    else:
        gelu = form == "bf16_gelu"
        out = qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, gelu=gelu)
        want = torch.nn.functional.gelu(y64, approximate="tanh") if gelu else y64
        scale = float(want.abs().max())
        # one bf16 rounding (2^-9 relative) on top of the fp32 epilogue
        err = (out.double() - want).abs()
        assert float((err - want.abs() * 2.0 ** -8).max()) < 4e-6 * scale, float(err.max())
        ref = kr.gelu_tanh(yo) if gelu else yo
        np.testing.assert_allclose(out[rows].float().cpu().numpy(), ref, rtol=2.0 ** -8, atol=4e-6 * scale)


# ---------------------------------------------------------------------------------------------------------------------
# RMSNorm + RoPE over the headline latent grid (21 x 30 x 52 = 32760 positions) and the cross-attention launch (32760 queries x
# 512 text keys, with and without key masking): whole outputs against the oracle's definitions evaluated on the GPU in fp32 /
# fp64 (oracle/wan_ref.py is plain torch: the same functions the small-size tests call on the CPU).
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_rmsnorm_rope_full_size_whole_output(dtype):
    from oracle import wan_ref as wr
    from wan import ops

    d, grid = 128, (21, 30, 52)
    assert grid[0] * grid[1] * grid[2] == L
    g = torch.Generator(device=DEV).manual_seed(21)
    x = (torch.randn(L, C, device=DEV, generator=g) * 1.7).to(dtype)
    w = torch.rand(C, device=DEV, generator=g) + 0.5
    freqs = wr.rope_freqs(d)
    table = ops.rope_table(freqs, grid, DEV)
    assert table.shape == (L, d // 2, 2)
    ref = wr.rope_apply(wr.rms_norm(x.float(), w, 1e-6).view(L, H, d), grid, freqs.to(DEV)).reshape(L, C)
    y = ops.rmsnorm_rope_(x.clone(), w, table, d, eps=1e-6)
    if dtype == torch.bfloat16:  # one bf16 rounding of the result
        assert float(((y.float() - ref).abs() - ref.abs() * 2.0 ** -8).max()) < 1e-5
    else:
        assert float((y - ref).abs().max()) < 4e-6 * float(ref.abs().max())


@pytest.mark.parametrize("k_len", [512, 77])
def test_cross_attention_full_size_whole_output(k_len):
    from wan import ops

    d, Lc = 128, 512
    g = torch.Generator(device=DEV).manual_seed(512 + k_len)
    q = torch.randn(L, H * d, device=DEV, generator=g).to(torch.bfloat16)
    k = torch.randn(Lc, H * d, device=DEV, generator=g).to(torch.bfloat16)
    v = torch.randn(Lc, H * d, device=DEV, generator=g).to(torch.bfloat16)
    o = ops.attention(q, k, v, H, k_len=None if k_len == Lc else k_len)
    qs = q.float().view(L, H, d).transpose(0, 1)                                   # [H, L, d]
    kk, vv = k[:k_len].float().view(k_len, H, d).transpose(0, 1), v[:k_len].float().view(k_len, H, d).transpose(0, 1)
    ref = (torch.softmax(qs @ kk.transpose(1, 2) / d ** 0.5, dim=-1) @ vv).transpose(0, 1).reshape(L, H * d)
    got = o.float()
    assert bool(torch.isfinite(got).all())
    assert float((got - ref).abs().max()) < 3e-2 and float((got - ref).norm() / ref.norm()) < 1e-2  # the flash-attention bar (bf16 P, bf16 O)


# ---------------------------------------------------------------------------------------------------------------------
# The remaining row-wise / column-wise kernels of the step and of the calibration pass at L = 32760.
def test_layernorm_modulate_quant_full_size_sampled_rows():
    """norm3 / norm2 -> quantise ([32760, 1536] fp32 residual stream -> int8 + scale + sum): rowwise_kernel<..., LN>."""
    import viditq_extension.fused as fused

    g = torch.Generator(device=DEV).manual_seed(77)
    x = torch.randn(L, C, device=DEV, generator=g) * 2 + 0.3
    x[:, 11] *= 25.0
    w = torch.randn(C, device=DEV, generator=g)
    sh, sc = torch.randn(1, C, device=DEV, generator=g) * 0.3, torch.randn(1, C, device=DEV, generator=g) * 0.3
    q = torch.empty(L, C, dtype=torch.int8, device=DEV)
    scale, ssum = torch.zeros(L, device=DEV), torch.zeros(L, device=DEV)
    fused.layernorm_nobias_t2i_quant_sum_fuse(q, x, w, sh, sc, ssum, scale, 1e-6)
    qi = q.to(torch.int32)
    assert bool((qi.abs().amax(1) == 127).all())
    np.testing.assert_allclose(ssum.cpu().numpy(), qi.sum(1).cpu().numpy().astype(np.float64) * scale.cpu().numpy().astype(np.float64), rtol=1e-6, atol=1e-6)
    rows = np.unique(np.concatenate([np.random.default_rng(8).integers(0, L, 40), [0, L - 1]]))
    oq, oscale, _ = kr.layernorm_t2i_quant_sum(x[rows].cpu().numpy(), w.cpu().numpy(), sh.cpu().numpy(), sc.cpu().numpy(), 1e-6, len(rows))
    np.testing.assert_allclose(scale[rows].cpu().numpy(), oscale, rtol=1e-5)
    d = np.abs(q[rows].cpu().numpy().astype(np.int32) - oq.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())


@pytest.mark.parametrize("cols,dtype", [(C, torch.float32), (F, torch.bfloat16)])
def test_calibration_column_absmax_full_size_exact(cols, dtype):
    """A8 at the size of one hook call of the headline run (W/get_calib_data_wanx.py:262-267): a maximum is order independent, so
    the whole result is exact; the running buffer keeps what it already held."""
    import viditq_extension.fused as fused

    g = torch.Generator(device=DEV).manual_seed(cols + 1)
    x = (torch.randn(L, cols, device=DEV, generator=g) * torch.exp(torch.randn(cols, device=DEV, generator=g))).to(dtype)
    running = torch.full((cols,), 0.25, device=DEV)
    running[5] = 1e9
    fused.col_absmax_(running, x)
    want = torch.maximum(x.float().abs().amax(0), torch.full((cols,), 0.25, device=DEV))
    want[5] = 1e9
    assert torch.equal(running, want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_v_fake_quant_full_size_whole_output(dtype):
    """attn.v at [32760, 12 x 128]: the per-(head, channel) DynamicQuantizer over all tokens, whole output bit for bit against the
    oracle function."""
    import viditq_extension.fused as fused
    from oracle import wan_ref as wr

    g = torch.Generator(device=DEV).manual_seed(33)
    v = (torch.randn(L, C, device=DEV, generator=g) * torch.exp(torch.randn(C, device=DEV, generator=g))).to(dtype)
    v[:, 3] = 0
    # the oracle runs on the CPU: torch's fp32 division on the GPU is not the IEEE quotient (evaluated there, the same function
    # differs from its CPU result on 1.8e6 of the 5e7 elements; tools/probes/vq_fullsize_diag.py), the kernel's is
    ref = wr.v_fake_quant(v.float().cpu().view(L, H, 128), 8).reshape(L, C)
    out, colmax = fused.fake_quant_cols_(v.clone(), 8)
    assert torch.equal(colmax, v.float().abs().amax(0))
    assert torch.equal(out.float().cpu(), ref.to(dtype).float()) and float(out[:, 3].abs().max()) == 0


def test_int8_qk_attention_full_size_sampled_queries():
    """attn.qk at the headline size: RMSNorm + RoPE + per-(token, head) int8 codes for q and k over the 21 x 30 x 52 grid (codes
    against the oracle chain on sampled rows), then int8 Q.K^T attention over all 32760 keys against the fp32 definition on the
    kernel's own codes (sampled queries)."""
    from oracle import qdiff_ref as qr
    from oracle import wan_ref as wr
    from wan import ops

    d, grid = 128, (21, 30, 52)
    g = torch.Generator(device=DEV).manual_seed(88)
    xq = (torch.randn(L, C, device=DEV, generator=g) * torch.exp(0.5 * torch.randn(C, device=DEV, generator=g))).to(torch.bfloat16)
    xk = (torch.randn(L, C, device=DEV, generator=g) * torch.exp(0.5 * torch.randn(C, device=DEV, generator=g))).to(torch.bfloat16)
    v = torch.randn(L, C, device=DEV, generator=g).to(torch.bfloat16)
    w = torch.rand(C, device=DEV, generator=g) + 0.5
    freqs = wr.rope_freqs(d)
    table = ops.rope_table(freqs, grid, DEV)
    q8 = ops.rmsnorm_rope_q8(xq, w, table, d, False)
    k8 = ops.rmsnorm_rope_q8(xk, w, table, d, True)
    rows = torch.from_numpy(np.unique(np.concatenate([np.random.default_rng(3).integers(0, L, 40), [0, 255, 256, L - 1]]))).to(DEV)
    for x, c8 in ((xq, q8), (xk, k8)):
        full = wr.rope_apply(wr.rms_norm(x.float(), w, 1e-6).view(L, H, d), grid, freqs.to(DEV))  # [L, H, d]
        oq, oscale = qr.dynamic_quantize_sym(full[rows].reshape(-1, d).cpu().numpy())
        scale = c8.scales[0, :, :L].t()[rows].reshape(-1).cpu().numpy()
        np.testing.assert_allclose(scale, oscale.reshape(-1), rtol=2e-5)
        dq = np.abs(c8.codes[rows].cpu().numpy().astype(np.int32).reshape(-1, d) - oq.astype(np.int32))
        assert dq.max() <= 1 and (dq != 0).mean() < 5e-3, (dq.max(), (dq != 0).mean())
    out = ops.attention_qk8(q8, k8, v, H)
    assert bool(torch.isfinite(out.float()).all())
    qf = q8.codes[rows].float().view(-1, H, d) * q8.scales[0, :, :L].t()[rows].unsqueeze(-1)
    kf = k8.codes.float().view(L, H, d) * k8.scales[0, :, :L].t().unsqueeze(-1)
    s = qf.transpose(0, 1) @ kf.permute(1, 2, 0) / d ** 0.5                       # [H, S, L]
    ref = (torch.softmax(s, dim=-1) @ v.float().view(L, H, d).transpose(0, 1)).transpose(0, 1).reshape(len(rows), C)
    got = out[rows].float()
    assert float((got - ref).abs().max()) < 3e-2 and float((got - ref).norm() / ref.norm()) < 1e-2


# ---------------------------------------------------------------------------------------------------------------------
# One WHOLE kernel-mode block of the headline workload -- L = 32760 tokens over the 21 x 30 x 52 grid, C = 1536, 12 heads,
# F = 8960, 512 context tokens, the configuration bench.py measures (quant_configs/w8a8_all_linears.yaml: every Linear W8A8,
# ViDiT-Q scale + rotate on self_attn q / k / v) -- against the simulation-mode oracle on a sample of rows.  Everything in the
# block but the self-attention keys and values is row-local, so the oracle computes k and v for all tokens and the rest for the
# sampled rows only (oracle/wan_ref.py::BlockRef.rows, checked against the full block in tests/test_oracle_golden.py).
def _headline_block():
    """The kernel-mode block of the headline configuration with its oracle ingredients (state dict, ViDiT masks / rotations, inputs)."""
    from oracle import qdiff_ref as qr
    from oracle import wan_ref as wr
    from qdiff import config as qcfg
    from qdiff.base.quant_model import quant_layer_refactor_
    from qdiff.utils import apply_func_to_submodules
    from wan import calib
    from wan.modules.model import WanAttentionBlock
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel

    grid, lc = (21, 30, 52), 512
    torch.manual_seed(7)
    blk = WanAttentionBlock("t2v_cross_attn", C, F, H, cross_attn_norm=True)
    for m in blk.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    blk.norm3.weight.data.uniform_(0.5, 1.5)
    blk.norm3.bias.data.normal_(std=0.1)
    for nm in (blk.self_attn.norm_q, blk.self_attn.norm_k, blk.cross_attn.norm_q, blk.cross_attn.norm_k):
        nm.weight.data.uniform_(0.5, 1.5)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(8)
    x = torch.randn(L, C, generator=g)
    x[:, 9] *= 15.0  # an outlier channel
    e0 = torch.randn(1, 6, C, generator=g) * 0.3
    ctx = torch.randn(lc, C, generator=g)
    freqs = wr.rope_freqs(C // H)
    act_mask = torch.rand(C, generator=g) * 3 + 0.2

    cfg = qcfg.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                       "viditq": {"alpha": 0.5665, "layer_name_regex": r"self_attn\.(q|k|v)$"}, "remain_fp_regex": None})
    blk = blk.to(DEV)
    apply_func_to_submodules(blk, torch.nn.Linear, quant_layer_refactor_, name=None, parent_module=None, quant_config=cfg,
                             full_name=None, remain_fp_regex=cfg.remain_fp_regex)
    gen = torch.Generator().manual_seed(11)
    vidit = {}
    for name in ("q", "k", "v"):
        lin = getattr(blk.self_attn, name)
        assert type(lin).__name__ == "ViDiTQuantizedLinear"
        calib.init_rotation_and_channel_mask_(lin, "x", {"x": act_mask[None]}, gen)
        vidit["self_attn." + name] = (lin.channel_mask.cpu(), torch.from_numpy(qr.hadamard_from_signs(lin.rotation_signs.numpy())))
    assert type(blk.self_attn.o).__name__ == "QuantizedLinear" and type(blk.ffn[2]).__name__ == "QuantizedLinear"
    hb = WanAttentionBlockWithHipKernel.from_float(blk, None)
    assert hb.self_attn.q.quantized and hb.self_attn.q.rot[0] == 12 and hb.self_attn.o.quantized and hb.ffn2.quantized
    return hb, sd, vidit, x, e0, ctx, freqs, grid


def _to_dev(obj):
    if torch.is_tensor(obj):
        return obj.to(DEV)
    if isinstance(obj, dict):
        return {k: _to_dev(v) for k, v in obj.items()}
    if isinstance(obj, (tuple, list)):
        return type(obj)(_to_dev(v) for v in obj)
    return obj


def _spectral_concentration(err):
    """Share of the error's energy in its largest singular direction (tests/test_model_golden.py: code-flip noise is unstructured,
    a wrong additive term of the dequantisation equation is rank one); eigenvalues of E^T E, fp64, on the device."""
    ev = torch.linalg.eigvalsh(err.double().t() @ err.double())
    return float(ev[-1] / ev.sum())


def test_headline_block_full_size_sampled_rows_vs_simulation_oracle():
    from oracle import wan_ref as wr
    from wan import ops
    from wan.quant_wanx_hip import _FpSrc

    hb, sd, vidit, x, e0, ctx, freqs, grid = _headline_block()
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), L, _FpSrc(ctx.to(DEV), torch.bfloat16))
    assert bool(torch.isfinite(out).all())

    rows = np.unique(np.concatenate([np.random.default_rng(12).integers(0, L, 28), [0, 255, 256, L - 249, L - 1]]))
    ref_q = wr.block_from_state(sd, H, quant=True, vidit=vidit).rows(x, e0, grid, L, ctx, freqs, rows)
    ref_fp = wr.block_from_state(sd, H, quant=False).rows(x, e0, grid, L, ctx, freqs, rows)
    got = out[torch.from_numpy(rows).to(DEV)].float().cpu()
    err = float((got.double() - ref_q.double()).norm() / ref_q.double().norm())
    noise = float((ref_q.double() - ref_fp.double()).norm() / ref_fp.double().norm())
    print(f"headline block, {len(rows)} rows: rel err vs fake-quant oracle {err:.2e}; fake-quant vs fp {noise:.2e}")
    assert err < 1e-2 and err < 0.5 * noise + 5e-3, (err, noise)  # the bars of tests/test_gpu_block.py's small-size block tests


def test_headline_block_WHOLE_output_vs_simulation_oracle_evaluated_on_the_gpu():
    """All 32760 x 1536 outputs of the headline block against oracle/wan_ref.py::BlockRef -- the oracle's own torch code, run on the
    GPU (quantiser divisions through fp64 so that they are the IEEE quotients the CPU evaluation has, Linears in fp64, exact
    softmax attention in query chunks; the note at the top of oracle/wan_ref.py).  The sampled-rows test above stays as the
    anchor to the CPU evaluation.  Reported and bounded: relative L2 error over the whole tensor (bar 5e-3 = the measured code-flip
    floor of the recipe, tests/test_model_golden.py), largest element error against the output range, and the shape of the error
    (energy share of its largest singular direction; a wrong zero-point / bias / scale term is rank one)."""
    from oracle import wan_ref as wr
    from wan import ops
    from wan.quant_wanx_hip import _FpSrc

    hb, sd, vidit, x, e0, ctx, freqs, grid = _headline_block()
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), L, _FpSrc(ctx.to(DEV), torch.bfloat16)).float()
    del hb
    ref = wr.block_from_state(_to_dev(sd), H, quant=True, vidit=_to_dev(vidit))(x.to(DEV), e0.to(DEV), grid, L, ctx.to(DEV), freqs.to(DEV))
    ref_fp = wr.block_from_state(_to_dev(sd), H, quant=False)(x.to(DEV), e0.to(DEV), grid, L, ctx.to(DEV), freqs.to(DEV))
    assert ref.shape == out.shape == (L, C)
    d = out.double() - ref.double()
    err = float(d.norm() / ref.double().norm())
    noise = float((ref.double() - ref_fp.double()).norm() / ref_fp.double().norm())
    rng_ = float(ref.max() - ref.min())
    worst = float(d.abs().max()) / rng_
    row_err = d.norm(dim=1) / ref.double().norm(dim=1)
    shape = _spectral_concentration(d)
    print(f"headline block, whole output [{L}, {C}]: rel L2 {err:.2e} (fake-quant vs fp {noise:.2e}); max |err| / range {worst:.2e}; "
          f"worst row {float(row_err.max()):.2e}, median row {float(row_err.median()):.2e}; error shape {shape:.3f}")
    assert err < 5e-3 and err < noise, (err, noise)  # below the flip floor, and below the quantisation effect it reproduces
    assert worst < 2e-3 and float(row_err.max()) < 1e-2, (worst, float(row_err.max()))
    assert shape < 0.05, shape
    # the shape measure proves its own sensitivity: a bias error of 2e-3 of the output's rms on a third of the channels (rank one:
    # 1 x delta b; 1.2e-3 in relative error, i.e. BELOW the flip floor and invisible to the first measure) must fail it
    fault = torch.zeros(C, device=DEV, dtype=torch.float64)
    fault[::3] = 2e-3 * float(ref.double().pow(2).mean().sqrt())
    assert float((d + fault).norm() / ref.double().norm()) < 5e-3 and _spectral_concentration(d + fault) > 0.05


# ---------------------------------------------------------------------------------------------------------------------
# The 8960-column ViDiT transform (rotate140_kernel: ffn.2's input under the reference's shipped config.yaml) and the fused
# CFG + scheduler update, at the headline sizes.
def test_rotate_8960_full_size_sampled_rows():
    import viditq_extension.fused as fused
    from oracle import qdiff_ref as qr
    from qdiff.quarot import quarot_utils as qu

    g = torch.Generator(device=DEV).manual_seed(140)
    x = (torch.randn(L, F, device=DEV, generator=g) * torch.exp(0.7 * torch.randn(F, device=DEV, generator=g))).clamp_min(-0.17).to(torch.bfloat16)
    x[5] = 0  # an all-zero token: the eps rule
    pm = (torch.rand(F, device=DEV, generator=g) + 0.5) * (torch.randint(0, 2, (F,), device=DEV, generator=g) * 2 - 1).float()
    rot = qu.kernel_rotation_params(F, DEV)
    assert rot[0] == 140
    scale, ssum = torch.zeros(L, device=DEV), torch.zeros(L, device=DEV)
    q = fused.rotate_quant(x, pm, rot, ssum, scale)
    qi = q.to(torch.int32)
    live = torch.ones(L, dtype=torch.bool, device=DEV)
    live[5] = False
    assert bool((qi.abs().amax(1)[live] == 127).all()) and int(qi[5].abs().max()) == 0 and scale[5].item() == pytest.approx(1e-6)
    np.testing.assert_allclose(ssum.cpu().numpy(), qi.sum(1).cpu().numpy().astype(np.float64) * scale.cpu().numpy().astype(np.float64), rtol=1e-6, atol=1e-6)
    rows = np.unique(np.concatenate([np.random.default_rng(14).integers(0, L, 24), [0, 4, 5, 6, L - 1]]))
    ref = qr.matmul_hadU(x[rows].double().cpu().numpy() * pm.double().cpu().numpy()).astype(np.float32)
    oq, oscale = qr.dynamic_quantize_sym(ref)
    np.testing.assert_allclose(scale[rows].cpu().numpy(), oscale, rtol=2e-6)
    d = np.abs(q[rows].cpu().numpy().astype(np.int32) - oq)
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())


def test_rotate_13824_config4_per_rank_size_sampled_rows():
    """The 13824-column transform (rotate108_kernel: the 14B ffn.2 input; repo-defined, the reference asserts on this width) at the
    per-rank size of BASELINE configs 4 / 5, [9450, 13824] bf16: row invariants over the whole output, sampled rows against the
    oracle's H_108 (x) H_128 evaluated in float64."""
    import viditq_extension.fused as fused
    from oracle import qdiff_ref as qr
    from qdiff.quarot import quarot_utils as qu

    R, N = 9450, 13824
    g = torch.Generator(device=DEV).manual_seed(108)
    x = (torch.randn(R, N, device=DEV, generator=g) * torch.exp(0.7 * torch.randn(N, device=DEV, generator=g))).clamp_min(-0.17).to(torch.bfloat16)
    x[5] = 0
    pm = (torch.rand(N, device=DEV, generator=g) + 0.5) * (torch.randint(0, 2, (N,), device=DEV, generator=g) * 2 - 1).float()
    rot = qu.kernel_rotation_params(N, DEV)
    assert rot[0] == 108
    scale, ssum = torch.zeros(R, device=DEV), torch.zeros(R, device=DEV)
    q = fused.rotate_quant(x, pm, rot, ssum, scale)
    qi = q.to(torch.int32)
    live = torch.ones(R, dtype=torch.bool, device=DEV)
    live[5] = False
    assert bool((qi.abs().amax(1)[live] == 127).all()) and int(qi[5].abs().max()) == 0 and scale[5].item() == pytest.approx(1e-6)
    np.testing.assert_allclose(ssum.cpu().numpy(), qi.sum(1).cpu().numpy().astype(np.float64) * scale.cpu().numpy().astype(np.float64), rtol=1e-6, atol=1e-6)
    rows = np.unique(np.concatenate([np.random.default_rng(108).integers(0, R, 24), [0, 4, 5, 6, R - 1]]))
    ref = qr.matmul_hadU(x[rows].double().cpu().numpy() * pm.double().cpu().numpy(), strict=False).astype(np.float32)
    oq, oscale = qr.dynamic_quantize_sym(ref)
    np.testing.assert_allclose(scale[rows].cpu().numpy(), oscale, rtol=2e-6)
    d = np.abs(q[rows].cpu().numpy().astype(np.int32) - oq)
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())
    # orthogonality at full size: the transform preserves every row's norm (fp output, all rows)
    out = torch.empty(R, N, device=DEV)
    fused.rotate_quant(x, None, rot, None, None, out_fp=out, quantize=False)
    n_in, n_out = x.float().norm(dim=1), out.norm(dim=1)
    assert float(((n_out - n_in).abs() / n_in.clamp_min(1e-6))[live].max()) < 2e-5


@pytest.mark.parametrize("solver", ["unipc", "dpm++", "euler"])
def test_fused_step_full_latent_equals_plain_scheduler(solver):
    """CFG combine + scheduler update as one kernel on the headline latent [16, 21, 60, 104], 30 steps, against the plain
    scheduler classes (pinned by the reference's own fm_solvers files in tests/test_schedulers_cpu.py)."""
    from wan.utils.fm_solvers import FlowDPMSolverMultistepScheduler, FlowMatchScheduler
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    from wan.utils.fused_step import FusedStep, _takes_timestep

    mk = {"unipc": lambda: FlowUniPCMultistepScheduler(1000, shift=1.0), "dpm++": lambda: FlowDPMSolverMultistepScheduler(1000),
          "euler": lambda: FlowMatchScheduler(1000)}[solver]
    a, b = mk(), mk()
    a.set_timesteps(30, device=DEV, shift=5.0)
    b.set_timesteps(30, device=DEV, shift=5.0)
    g = torch.Generator(device=DEV).manual_seed(2)
    x = torch.randn(16, 21, 60, 104, device=DEV, generator=g)
    xa, xb = x.clone(), x.clone()
    f = FusedStep(b, 5.0, like=x)
    for t in a.timesteps:
        c, u = torch.randn(x.shape, device=DEV, generator=g), torch.randn(x.shape, device=DEV, generator=g)
        noise = u + 5.0 * (c - u)
        xa = a.step(noise, t, xa) if _takes_timestep(a) else a.step(noise, xa)
        xb = f.step(c, u, xb, t)
        assert float((xa - xb).abs().max() / xa.abs().max()) < 2e-6
    assert f.n_launch == 30


def test_14b_block_full_sequence_sampled_rows_vs_simulation_oracle():
    """One kernel-mode block with the 14B dimensions (C = 5120, 40 heads, F = 13824) on the WHOLE 1280x720x81f sequence
    (75600 tokens over the 21 x 45 x 80 grid: 295 full 256-row tiles + 80 rows), every Linear W8A8 -- the single-GPU form of
    BASELINE config 4 -- against the simulation oracle on sampled rows (plain fake-quant Linears: the fp64 rotation of 75600 x 5120
    rows would take the host minutes; the ViDiT layers are covered at this width by tests/test_gpu_block.py and at L = 32760 above)."""
    from oracle import wan_ref as wr
    from wan import ops
    from wan.modules.model import WanAttentionBlock
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    Lb, Cb, Fb, Hb, grid, lc = 75600, 5120, 13824, 40, (21, 45, 80), 512
    assert grid[0] * grid[1] * grid[2] == Lb
    torch.manual_seed(14)
    blk = WanAttentionBlock("t2v_cross_attn", Cb, Fb, Hb, cross_attn_norm=True)
    for m in blk.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    blk.norm3.weight.data.uniform_(0.5, 1.5)
    blk.norm3.bias.data.normal_(std=0.1)
    for nm in (blk.self_attn.norm_q, blk.self_attn.norm_k, blk.cross_attn.norm_q, blk.cross_attn.norm_k):
        nm.weight.data.uniform_(0.5, 1.5)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(15)
    x = torch.randn(Lb, Cb, generator=g)
    x[:, 9] *= 15.0
    e0 = torch.randn(1, 6, Cb, generator=g) * 0.3
    ctx = torch.randn(lc, Cb, generator=g)
    freqs = wr.rope_freqs(Cb // Hb)

    hb = WanAttentionBlockWithHipKernel.from_float(blk.to(DEV))
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), Lb, _FpSrc(ctx.to(DEV), torch.bfloat16))
    assert bool(torch.isfinite(out).all())
    rows = np.unique(np.concatenate([np.random.default_rng(16).integers(0, Lb, 16), [0, 255, 256, Lb - 81, Lb - 1]]))
    ref_q = wr.block_from_state(sd, Hb, quant=True).rows(x, e0, grid, Lb, ctx, freqs, rows)
    ref_fp = wr.block_from_state(sd, Hb, quant=False).rows(x, e0, grid, Lb, ctx, freqs, rows)
    got = out[torch.from_numpy(rows).to(DEV)].float().cpu()
    err = float((got.double() - ref_q.double()).norm() / ref_q.double().norm())
    noise = float((ref_q.double() - ref_fp.double()).norm() / ref_fp.double().norm())
    print(f"14B block, 75600 tokens, {len(rows)} rows: rel err vs fake-quant oracle {err:.2e}; fake-quant vs fp {noise:.2e}")
    assert err < 1e-2 and err < 0.5 * noise + 5e-3, (err, noise)


def test_14b_block_WHOLE_output_vs_simulation_oracle_evaluated_on_the_gpu():
    """The 14B-dimension block on the whole 75600-token sequence (the test above), every output element against the simulation
    oracle evaluated on the GPU (plain fake-quant Linears, as above).  Same three measures and bars as the headline block."""
    from oracle import wan_ref as wr
    from wan import ops
    from wan.modules.model import WanAttentionBlock
    from wan.quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc

    Lb, Cb, Fb, Hb, grid, lc = 75600, 5120, 13824, 40, (21, 45, 80), 512
    torch.manual_seed(14)
    blk = WanAttentionBlock("t2v_cross_attn", Cb, Fb, Hb, cross_attn_norm=True)
    for m in blk.modules():
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.xavier_uniform_(m.weight)
            torch.nn.init.normal_(m.bias, std=0.05)
    blk.norm3.weight.data.uniform_(0.5, 1.5)
    blk.norm3.bias.data.normal_(std=0.1)
    for nm in (blk.self_attn.norm_q, blk.self_attn.norm_k, blk.cross_attn.norm_q, blk.cross_attn.norm_k):
        nm.weight.data.uniform_(0.5, 1.5)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(15)
    x = torch.randn(Lb, Cb, generator=g)
    x[:, 9] *= 15.0
    e0 = torch.randn(1, 6, Cb, generator=g) * 0.3
    ctx = torch.randn(lc, Cb, generator=g)
    freqs = wr.rope_freqs(Cb // Hb)
    hb = WanAttentionBlockWithHipKernel.from_float(blk.to(DEV))
    out = hb(x.to(DEV).clone(), e0.to(DEV), ops.rope_table(freqs, grid, DEV), Lb, _FpSrc(ctx.to(DEV), torch.bfloat16)).float()
    del hb, blk
    torch.cuda.empty_cache()
    ref = wr.block_from_state(_to_dev(sd), Hb, quant=True)(x.to(DEV), e0.to(DEV), grid, Lb, ctx.to(DEV), freqs.to(DEV))
    ref_fp = wr.block_from_state(_to_dev(sd), Hb, quant=False)(x.to(DEV), e0.to(DEV), grid, Lb, ctx.to(DEV), freqs.to(DEV))
    d = out.double() - ref.double()
    err = float(d.norm() / ref.double().norm())
    noise = float((ref.double() - ref_fp.double()).norm() / ref_fp.double().norm())
    worst = float(d.abs().max()) / float(ref.max() - ref.min())
    row_err = d.norm(dim=1) / ref.double().norm(dim=1)
    shape = _spectral_concentration(d)
    print(f"14B block, whole output [{Lb}, {Cb}]: rel L2 {err:.2e} (fake-quant vs fp {noise:.2e}); max |err| / range {worst:.2e}; "
          f"worst row {float(row_err.max()):.2e}, median row {float(row_err.median()):.2e}; error shape {shape:.3f}")
    assert err < 5e-3 and err < noise, (err, noise)  # below the flip floor, and below the quantisation effect it reproduces
    assert worst < 2e-3 and float(row_err.max()) < 1e-2, (worst, float(row_err.max()))
    assert shape < 0.05, shape


def test_fp_block_fused_glue_full_size_against_the_torch_expressions(monkeypatch):
    """One FP block at the headline size (32760 tokens, 1.3B width) under bf16 autocast, its row-wise glue on the fused kernels
    (wan/modules/model.py::fused_fp, what fp_generate.py and the calibration passes run) against the reference's torch expressions
    (WANQ_FP_FUSED=0): the LayerNorm + modulate output the q / k / v Linears (and the calibration hooks) see to fp32 rounding, the
    block output to the 16-bit roundings of the Linears and of the attention operands."""
    from wan.configs import seq_len_for
    from wan.modules.model import WanAttentionBlock, rope_params

    torch.manual_seed(0)
    with torch.device(DEV):
        blk = WanAttentionBlock("t2v_cross_attn", C, F, H, (-1, -1), True, True, 1e-6).eval()
    g = torch.Generator(device=DEV).manual_seed(7)
    with torch.no_grad():
        for p in blk.parameters():
            if p.dim() == 2:
                torch.nn.init.xavier_uniform_(p, generator=g)
    shape = (16, 21, 60, 104)
    grid, seq_len = (21, 30, 52), seq_len_for(shape)
    assert seq_len == L
    x = torch.randn(1, L, C, device=DEV, generator=g)
    e = torch.randn(1, 6, C, device=DEV, generator=g) * 0.1
    ctx = torch.randn(1, 512, C, device=DEV, generator=g)
    d = C // H
    freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)), rope_params(1024, 2 * (d // 6))], dim=1)

    def run(flag):
        monkeypatch.setenv("WANQ_FP_FUSED", flag)
        seen = {}
        h = blk.self_attn.q.register_forward_pre_hook(lambda m, a: seen.__setitem__("q_in", a[0].detach().clone()))
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            out = blk(x, e, [L], [grid], freqs, ctx, None)
        h.remove()
        return out.float(), seen["q_in"]

    o1, q1 = run("1")
    o0, q0 = run("0")
    assert q1.dtype == torch.float32 and q1.shape == q0.shape == (1, L, C)
    assert float((q1 - q0).abs().max()) <= 2e-5 * float(q0.abs().max())
    assert float((o1 - o0).norm() / o0.norm()) < 5e-3 and bool(torch.isfinite(o1).all())
