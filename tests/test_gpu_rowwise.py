"""GPU parity: HIP rowwise kernels (through the C ABI) vs the oracle and the golden vectors.
int8 codes of the pure quantise step: bit-exact.  LayerNorm-fused variants: codes within 1 LSB on a
tiny fraction of elements (fp32 reduction order), scales to 1e-5 relative."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref as kr
from oracle import qdiff_ref as qr

pytestmark = pytest.mark.gpu

DEV = "cuda"


def fused():
    import viditq_extension.fused as f

    return f


def t(a, dtype=None):
    x = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return x.to(dtype) if dtype is not None else x


def run_quant(x_t, vec_dtype=torch.float32, fn="quant_sum"):
    rows = x_t.numel() // x_t.shape[-1]
    scale = torch.zeros(rows, dtype=vec_dtype, device=DEV)
    ssum = torch.zeros(rows, dtype=vec_dtype, device=DEV)
    q = getattr(fused(), fn)(x_t, ssum, scale)
    torch.cuda.synchronize()
    return q.cpu().numpy(), scale.float().cpu().numpy(), ssum.float().cpu().numpy()


@pytest.mark.parametrize("name", ["a2_dynamic_7x64", "a2_dynamic_32x1536", "a2_dynamic_5x5120"])
def test_quant_rows_bit_exact_vs_reference_golden(golden, name):
    """The 'bit-exact int8 quantize step': fp32 input -> same codes and same fp32 delta as the
    reference's DynamicQuantizer (golden captured from qdiff)."""
    g = golden(name)
    q, scale, ssum = run_quant(t(g["x"]))
    assert np.array_equal(q, g["q"].astype(np.int8))
    assert np.array_equal(scale, g["delta"])
    np.testing.assert_allclose(ssum, g["q"].astype(np.int64).sum(1) * g["delta"].astype(np.float64), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(1, 8), (3, 64), (5, 1152), (130, 1536), (37, 2048), (9, 2056), (33, 4096),
                                       (17, 5120), (12, 8960), (6, 13824), (2, 16384)])
def test_quant_rows_bit_exact_vs_oracle(dtype, rows, cols):
    g = torch.Generator().manual_seed(rows * 100003 + cols)
    x = torch.randn(rows, cols, generator=g) * torch.exp(torch.randn(cols, generator=g))
    if rows > 2:
        x[1] = 0
        x[2, ::7] *= 30
    x = x.to(dtype)
    q, scale, ssum = run_quant(x.to(DEV))
    oq, oscale, osum = kr.quant_sum(x.float().numpy())
    assert np.array_equal(scale, oscale)
    assert np.array_equal(q, oq)
    np.testing.assert_allclose(ssum, osum, rtol=1e-6, atol=1e-6)


def test_quant_rows_ties_and_near_ties():
    """Inputs placed on and next to .5 boundaries exercise the exact-division fallback."""
    rows, cols = 64, 1536
    rng = np.random.default_rng(7)
    scale = rng.uniform(0.003, 3.0, size=(rows, 1)).astype(np.float32)
    k = rng.integers(-126, 126, size=(rows, cols)).astype(np.float32)
    x = ((k + 0.5) * scale).astype(np.float32)
    x = np.nextafter(x, x + rng.choice([-1, 0, 1], size=x.shape).astype(np.float32)).astype(np.float32)
    x[:, 0] = 127 * scale[:, 0]  # fixes absmax so that delta ~= scale
    q, s, _ = run_quant(t(x))
    oq, os_, _ = kr.quant_sum(x)
    assert np.array_equal(s, os_)
    assert np.array_equal(q, oq)


def test_quant_sum_fp16_buffers_kbench(golden):
    """Reference-style call: fp16 input, fp16 scale/sum buffers (K/bench/bench_quant_kernel.py:8-26)."""
    g = golden("kbench_quant")
    q, scale, ssum = run_quant(t(g["x"]), torch.float16)
    assert np.array_equal(q, g["q"])
    assert np.array_equal(scale, g["scale"].astype(np.float16).astype(np.float32))
    np.testing.assert_allclose(ssum, g["sum"].astype(np.float32), rtol=2e-3, atol=2e-2)
    gq, gscale, _ = run_quant(t(g["x"]), torch.float32, "gelu_quant_sum")
    np.testing.assert_allclose(gscale, g["gelu_scale"], rtol=1e-5)
    d = np.abs(gq.astype(np.int32) - g["gelu_q"].astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


def test_quant_sum_static_reads_amax():
    x = torch.randn(16, 1536, generator=torch.Generator().manual_seed(3))
    amax = (x.abs().amax(dim=1) * 1.5).to(DEV)
    ssum = torch.zeros(16, device=DEV)
    q = fused().quant_sum_static(x.to(DEV), ssum, amax)
    torch.cuda.synchronize()
    sc = (amax.cpu().numpy() / np.float32(127)).astype(np.float32)
    oq = np.clip(np.round(x.numpy() / sc[:, None]), -128, 127).astype(np.int8)
    assert np.array_equal(q.cpu().numpy(), oq)
    np.testing.assert_allclose(amax.cpu().numpy(), x.abs().amax(dim=1).numpy() * 1.5)  # untouched


@pytest.mark.parametrize("cols", [64, 1536, 5120])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_layernorm_family_vs_oracle(cols, dtype):
    B, T = 2, 19
    g = torch.Generator().manual_seed(cols)
    x = (torch.randn(B * T, cols, generator=g) * 2 + 0.3).to(dtype)
    w = torch.randn(cols, generator=g).to(dtype)
    sh = torch.randn(B, cols, generator=g).to(dtype)
    sc = torch.randn(B, cols, generator=g).to(dtype)
    xn, wn, shn, scn = (a.float().numpy() for a in (x, w, sh, sc))
    f = fused()
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float16 else dict(rtol=2e-5, atol=2e-5)

    out = torch.empty_like(x, device=DEV)
    f.layernorm_nobias(out, x.to(DEV), w.to(DEV), 1e-5)
    np.testing.assert_allclose(out.float().cpu().numpy(), kr.layernorm_nobias(xn, wn, 1e-5), **tol)
    f.layernorm_nobias_t2i_fuse(out, x.to(DEV), w.to(DEV), sh.to(DEV), sc.to(DEV), 1e-5)
    ref = kr.layernorm_t2i(xn, wn, shn, scn, 1e-5, T)
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=tol["rtol"], atol=tol["atol"] * 4)

    q = torch.empty(B * T, cols, dtype=torch.int8, device=DEV)
    scale = torch.zeros(B * T, dtype=torch.float32, device=DEV)
    ssum = torch.zeros(B * T, dtype=torch.float32, device=DEV)
    f.layernorm_nobias_t2i_quant_sum_fuse(q, x.to(DEV), w.to(DEV), sh.to(DEV), sc.to(DEV), ssum, scale, 1e-5)
    oq, oscale, osum = kr.layernorm_t2i_quant_sum(xn, wn, shn, scn, 1e-5, T)
    np.testing.assert_allclose(scale.cpu().numpy(), oscale, rtol=1e-5)
    d = np.abs(q.cpu().numpy().astype(np.int32) - oq.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3
    # the sum is exactly consistent with the codes the kernel itself produced
    np.testing.assert_allclose(ssum.cpu().numpy(), q.cpu().numpy().astype(np.int64).sum(1) * scale.cpu().numpy().astype(np.float64),
                               rtol=1e-6, atol=1e-6)
    f.layernorm_nobias_quant_sum_fuse(q, x.to(DEV), w.to(DEV), ssum, scale, 1e-5)
    oq, oscale, _ = kr.quant_sum(kr.layernorm_nobias(xn, wn, 1e-5).astype(np.float32))
    np.testing.assert_allclose(scale.cpu().numpy(), oscale, rtol=1e-5)
    f.layernorm_nobias_quant_nosum_fuse(q, x.to(DEV), w.to(DEV), scale, 1e-5)
    d = np.abs(q.cpu().numpy().astype(np.int32) - oq.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


def test_layernorm_t2i_quant_kbench(golden):
    """K/bench/bench_layer_norm_kernel.py:47-49 ground truth with the reference's fp16 buffers."""
    g = golden("kbench_layernorm")
    B, T, C = g["x"].shape
    x = t(g["x"]).view(-1, C)
    q = torch.empty(B * T, C, dtype=torch.int8, device=DEV)
    scale = torch.zeros(B * T, dtype=torch.float16, device=DEV)
    ssum = torch.zeros(B * T, dtype=torch.float16, device=DEV)
    fused().layernorm_nobias_t2i_quant_sum_fuse(q, x, t(g["weight"]), t(g["shift"]), t(g["scale_msa"]), ssum, scale, 1e-5)
    np.testing.assert_allclose(scale.float().cpu().numpy(), g["q_scale"], rtol=1e-3)
    d = np.abs(q.cpu().numpy().astype(np.int32) - g["q"].astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3
    np.testing.assert_allclose(ssum.float().cpu().numpy(), g["q_sum"], rtol=5e-3, atol=0.3)


def test_layernorm_module_matches_reference_call_pattern():
    from viditq_extension.nn import LayerNormGeneral, QuantParams

    B, L, C = 1, 50, 1536
    x = torch.randn(B, L, C, generator=torch.Generator().manual_seed(1)).half().to(DEV)
    shift = torch.randn(B, 1, C).half().to(DEV)
    scale = torch.randn(B, 1, C).half().to(DEV)
    qp = QuantParams(B * L, has_sum_input=True, device=DEV)
    ln = LayerNormGeneral(C, act_sum=True, eps=1e-6).to(DEV)
    out = ln(x, shift, scale, qp)
    assert out.shape == x.shape and out.dtype == torch.int8
    oq, oscale, _ = kr.layernorm_t2i_quant_sum(x.float().cpu().numpy().reshape(-1, C), None, shift.float().cpu().numpy().reshape(B, C),
                                               scale.float().cpu().numpy().reshape(B, C), 1e-6, L)
    d = np.abs(out.cpu().numpy().reshape(-1, C).astype(np.int32) - oq.astype(np.int32))
    assert d.max() <= 1 and (d != 0).mean() < 2e-3


@pytest.mark.parametrize("dtypes", [(torch.float16, torch.float16, torch.float16), (torch.float16, torch.float32, torch.float32)])
def test_gate_residual(dtypes):
    yd, gd, rd = dtypes
    B, T, C = 2, 33, 1536
    g = torch.Generator().manual_seed(11)
    y = torch.randn(B * T, C, generator=g).to(yd)
    gate = torch.randn(B, C, generator=g).to(gd)
    res = torch.randn(B * T, C, generator=g).to(rd)
    out = fused().gate_residual_fuse(y.to(DEV), gate.to(DEV), res.to(DEV), out_dtype=rd)
    ref = kr.gate_residual(y.float().numpy(), gate.float().numpy(), res.float().numpy(), T)
    tol = 2e-3 if rd == torch.float16 else 1e-6
    np.testing.assert_allclose(out.float().cpu().numpy(), ref, rtol=tol, atol=tol)


def test_col_absmax_matches_calibration_golden(golden):
    g = golden("a8_calib")
    C = g["calls"].shape[-1]
    running = torch.zeros(C, device=DEV)
    for c in g["calls"]:
        fused().col_absmax_(running, t(c).view(-1, C))
    got = running.cpu().numpy()
    assert np.array_equal(got, g["stacked"].max(axis=0))  # max is order independent: exact
    assert np.array_equal(qr.calib_act_mask(got[None]), g["act_mask"])


@pytest.mark.parametrize("rows,cols,dtype", [(1, 8, torch.float32), (777, 1536, torch.float16), (4680, 8960, torch.bfloat16),
                                             (100, 520, torch.float32)])
def test_col_absmax_vs_oracle(rows, cols, dtype):
    x = (torch.randn(rows, cols, generator=torch.Generator().manual_seed(rows)) * 3).to(dtype)
    running = torch.full((cols,), 0.5, device=DEV)
    fused().col_absmax_(running, x.to(DEV))
    ref = np.maximum(qr.calib_channel_absmax(x.float().numpy()), 0.5)
    assert np.array_equal(running.cpu().numpy(), ref)


@pytest.mark.parametrize("name", ["a1_static_16x64", "a1_static_12x1536"])
def test_weight_stats_and_static_quant_vs_golden(golden, name):
    g = golden(name)
    w = t(g["w"])
    lo, hi, am = fused().row_minmax(w)
    assert np.array_equal(lo.cpu().numpy(), g["w"].min(1)) and np.array_equal(hi.cpu().numpy(), g["w"].max(1))
    assert np.array_equal(am.cpu().numpy(), np.abs(g["w"]).max(1))
    for tag, bits in (("a8", 8), ("a4", 4)):
        n = 2 ** bits
        q8, dq = fused().weight_quant(w, t(g[f"{tag}_delta"]), t(g[f"{tag}_zp"]), -n - 1, n, want_int8=False, want_dequant=True)
        assert q8 is None and np.array_equal(dq.cpu().numpy(), g[f"{tag}_dequant"])
        q8, _ = fused().weight_quant(w, t(g[f"{tag}_delta"]), t(g[f"{tag}_zp"]), -128, 127)
        assert np.array_equal(q8.cpu().numpy(), g[f"{tag}_q"].astype(np.int8))
