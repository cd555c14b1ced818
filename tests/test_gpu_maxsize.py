"""Maximum sizes: operands past 2^31 ELEMENTS and past 2^32 BYTES through the C ABI, where a 32-bit row * cols product, a 32-bit byte
offset or a 32-bit LDS-DMA offset would wrap.  No Wan configuration of BASELINE.json gets there (the largest single tensor of the path
is the 14B ffn activation on one GPU, 75600 x 13824 = 1.05e9 elements), so this is the guard band: every row-wise kernel, the
calibration reduction, both GEMM kernels (the ping-pong kernel below its 2^32-byte operand bound and the general kernel the dispatcher
falls back to above it) and the attention kernel past the longest sequence of the configs.  Operands are generated on the GPU (a few
GB each); the oracle sees the rows on either side of the 2^31-element and 2^32-byte marks, the first and the last row, bit for bit
where the kernels are bit-exact and with the kernels' usual bars elsewhere."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref as kr
from oracle import qdiff_ref as qr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rows_around_marks(rows, cols, elem_bytes, extra=()):
    """first / last row, the rows that contain element 2^31 and byte 2^32 (and their neighbours), a few random ones"""
    marks = [0, rows - 1, *extra]
    for m in (2 ** 31 // cols, 2 ** 32 // (cols * elem_bytes), 2 ** 32 // cols):
        marks += [m - 1, m, m + 1]
    marks += list(np.random.default_rng(rows).integers(0, rows, 12))
    return np.unique([m for m in marks if 0 <= m < rows])


def _randn_rows(rows, cols, dtype, seed, chunk=32768):
    """randn * per-column spread, generated chunk by chunk straight into the target dtype (no fp32 image of the whole tensor)"""
    g = torch.Generator(device=DEV).manual_seed(seed)
    spread = torch.exp(0.5 * torch.randn(cols, device=DEV, generator=g))
    x = torch.empty(rows, cols, dtype=dtype, device=DEV)
    for r0 in range(0, rows, chunk):
        n = min(chunk, rows - r0)
        x[r0:r0 + n] = (torch.randn(n, cols, device=DEV, generator=g) * spread).to(dtype)
    return x


def test_quantise_rows_past_2_31_elements_and_2_32_bytes():
    """wanq_quant_rows on [270000, 8960] bf16: 2.42e9 elements, 4.84e9 bytes in, 2.42e9 bytes of codes out."""
    import viditq_extension.fused as fused

    R, N = 270000, 8960
    x = _randn_rows(R, N, torch.bfloat16, 1)
    x[R - 2] = 0
    scale, ssum = torch.zeros(R, device=DEV), torch.zeros(R, device=DEV)
    q = fused.quant_sum(x, ssum, scale)
    rows = _rows_around_marks(R, N, 2, extra=(R - 2,))
    oq, oscale, osum = kr.quant_sum(x[rows].float().cpu().numpy())
    np.testing.assert_array_equal(q[rows].cpu().numpy(), oq)
    np.testing.assert_array_equal(scale[rows].cpu().numpy(), oscale)
    # every row: the row maximum maps to +-127 (chunked: no int32 image of the whole tensor)
    for r0 in range(0, R, 65536):
        top = q[r0:r0 + 65536].to(torch.int16).abs().amax(1)
        live = torch.ones_like(top, dtype=torch.bool)
        if r0 <= R - 2 < r0 + 65536:
            live[R - 2 - r0] = False
        assert bool((top[live] == 127).all())


def test_layernorm_modulate_quantise_and_gate_residual_past_2_32_bytes():
    """wanq_layernorm_rows (fp32 in, modulated, int8 out) and wanq_gate_residual (bf16 * gate + fp32 residual, in place) on
    [440000, 5120]: 2.25e9 elements, 9.0e9 bytes of fp32."""
    import viditq_extension.fused as fused

    R, N = 440000, 5120
    x = _randn_rows(R, N, torch.float32, 2)
    g = torch.Generator(device=DEV).manual_seed(3)
    shift, mscale = torch.randn(1, N, device=DEV, generator=g) * 0.1, torch.randn(1, N, device=DEV, generator=g) * 0.1
    q = torch.empty(R, N, dtype=torch.int8, device=DEV)
    scale, ssum = torch.zeros(R, device=DEV), torch.zeros(R, device=DEV)
    fused.layernorm_nobias_t2i_quant_sum_fuse(q, x, None, shift, mscale, ssum, scale, 1e-6)
    rows = _rows_around_marks(R, N, 4)
    oq, oscale, _ = kr.layernorm_t2i_quant_sum(x[rows].cpu().numpy(), None, shift.cpu().numpy(), mscale.cpu().numpy(), 1e-6, len(rows))
    d = np.abs(q[rows].cpu().numpy().astype(np.int32) - oq)
    assert d.max() <= 1 and (d != 0).mean() < 2e-3, (d.max(), (d != 0).mean())    # fp32 reduction order
    np.testing.assert_allclose(scale[rows].cpu().numpy(), oscale, rtol=2e-6)
    del q
    y = _randn_rows(R, N, torch.bfloat16, 4)
    gate = torch.randn(1, N, device=DEV, generator=g)
    before = x[rows].clone()
    fused.gate_residual_into_(x, y, gate)
    want = kr.gate_residual(y[rows].float().cpu().numpy(), gate.cpu().numpy(), before.cpu().numpy(), len(rows))
    np.testing.assert_allclose(x[rows].cpu().numpy(), want, rtol=1e-6, atol=1e-6)  # one fp32 fma against mul + add (the bar of test_gpu_rowwise)


def test_rotate_quantise_8960_and_calibration_absmax_past_2_31_elements():
    """wanq_rotate_quant_rows (H_140 (x) H_64 after the per-channel multiplier) and wanq_col_absmax on [250000, 8960] bf16."""
    import viditq_extension.fused as fused
    from qdiff.quarot import quarot_utils as qu

    R, N = 250000, 8960
    x = _randn_rows(R, N, torch.bfloat16, 5)
    g = torch.Generator(device=DEV).manual_seed(6)
    pm = (torch.rand(N, device=DEV, generator=g) + 0.5) * (torch.randint(0, 2, (N,), device=DEV, generator=g) * 2 - 1).float()
    rot = qu.kernel_rotation_params(N, DEV)
    scale, ssum = torch.zeros(R, device=DEV), torch.zeros(R, device=DEV)
    q = fused.rotate_quant(x, pm, rot, ssum, scale)
    rows = _rows_around_marks(R, N, 2)
    ref = qr.matmul_hadU(x[rows].double().cpu().numpy() * pm.double().cpu().numpy()).astype(np.float32)
    oq, oscale = qr.dynamic_quantize_sym(ref)
    np.testing.assert_allclose(scale[rows].cpu().numpy(), oscale, rtol=2e-6)
    d = np.abs(q[rows].cpu().numpy().astype(np.int32) - oq)
    assert d.max() <= 1 and (d != 0).mean() < 3e-3, (d.max(), (d != 0).mean())
    del q
    # calibration: exact (a maximum has no rounding); the tensor's own maximum sits in the LAST row, past every mark
    x[R - 1, 77] = 1000.0
    run = torch.zeros(N, device=DEV)
    fused.col_absmax_(run, x)
    want = torch.zeros(N, device=DEV)
    for r0 in range(0, R, 65536):
        want = torch.maximum(want, x[r0:r0 + 65536].abs().amax(0).float())
    assert torch.equal(run, want) and run[77].item() == 1000.0


@pytest.mark.parametrize("M,K,N,kernel", [(270000, 8960, 256, "ping-pong (M*K = 2.42e9 < 2^32)"), (500000, 8960, 128, "general (M*K = 4.48e9 >= 2^32)")])
def test_gemm_activations_past_2_31_and_2_32_bytes(M, K, N, kernel):
    """int8 GEMM with an activation operand of 2.4 GB (past 2^31 bytes: the ping-pong kernel's LDS-DMA offsets are 32-bit UNSIGNED) and of
    4.5 GB (past 2^32: the dispatcher must leave the ping-pong kernel, gemm_pp_eligible): accumulators of the rows around the marks,
    bit for bit, and the bf16 epilogue on the same rows."""
    import viditq_extension.qgemm as qgemm

    g = torch.Generator(device=DEV).manual_seed(M)
    a = torch.empty(M, K, dtype=torch.int8, device=DEV)
    for r0 in range(0, M, 65536):
        n = min(65536, M - r0)
        a[r0:r0 + n] = torch.randint(-128, 128, (n, K), dtype=torch.int8, device=DEV, generator=g)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=DEV, generator=g)
    acc = qgemm.w8a8_o32(a, w)
    rows = _rows_around_marks(M, K, 1, extra=(M - 129, M - 128, 255, 256))
    ref = kr.w8a8_o32(a[rows].cpu().numpy(), w.cpu().numpy())
    np.testing.assert_array_equal(acc[rows].cpu().numpy(), ref)
    # column checksum over ALL rows, exact in int64: sum_m acc[m, n] = (sum_m a[m, :]) . w[n, :]
    colsum_a = torch.zeros(K, dtype=torch.int64, device=DEV)
    for r0 in range(0, M, 65536):
        colsum_a += a[r0:r0 + 65536].sum(0, dtype=torch.int64)
    want = (w.to(torch.float64) @ colsum_a.to(torch.float64)).to(torch.int64)       # |values| < 2^53: exact
    assert torch.equal(acc.sum(0, dtype=torch.int64), want)
    del acc
    sa = torch.rand(M, device=DEV, generator=g) * 0.01 + 1e-3
    sw = torch.rand(N, device=DEV, generator=g) * 0.01 + 1e-3
    bias = torch.randn(N, device=DEV, generator=g)
    out = qgemm.w8a8_linear(a, w, sa, sw, bias=bias, out_dtype=torch.bfloat16)
    want = torch.from_numpy(ref).to(DEV).float() * sa[torch.from_numpy(rows).to(DEV)][:, None] * sw[None, :] + bias[None, :]
    got = out[rows].float()
    assert float(((got - want).abs() / want.abs().clamp_min(1e-3)).max()) <= 2.0 ** -8   # one bf16 rounding


def test_attention_longer_than_any_configured_sequence():
    """111600 tokens (1280 x 720 x 121 frames; BASELINE's longest is 75600), 2 heads: finite, partition of unity over all keys, sampled
    queries against the fp32 definition."""
    from wan import ops

    L, H = 111600, 2
    g = torch.Generator(device=DEV).manual_seed(9)
    q, k, v = (torch.randn(L, H * 128, device=DEV, generator=g).to(torch.bfloat16) for _ in range(3))
    o = ops.attention(q, k, v, H)
    assert bool(torch.isfinite(o.float()).all())
    vc = torch.randn(1, H * 128, device=DEV, generator=g).to(torch.bfloat16).expand(L, -1).contiguous()
    oc = ops.attention(q, k, vc, H).float()
    assert float((oc - vc.float()).abs().max()) <= 2.0 ** -7 * float(vc.float().abs().max())
    rows = torch.from_numpy(np.unique(np.concatenate([np.random.default_rng(9).integers(0, L, 40), [0, 255, 256, L - 1]]))).to(DEV)
    qs = q[rows].float().view(-1, H, 128).transpose(0, 1)
    kk, vv = k.float().view(L, H, 128).transpose(0, 1), v.float().view(L, H, 128).transpose(0, 1)
    ref = (torch.softmax(qs @ kk.transpose(1, 2) / 128 ** 0.5, dim=-1) @ vv).transpose(0, 1).reshape(len(rows), H * 128)
    got = o[rows].float()
    assert float((got - ref).abs().max()) < 3e-2 and float((got - ref).norm() / ref.norm()) < 1e-2
