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
