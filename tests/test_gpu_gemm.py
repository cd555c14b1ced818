"""GPU parity of the int8-MFMA GEMM (C ABI wanq_gemm_w8a8) vs the oracle: int32 accumulators bit-exact,
fp epilogue within one ulp of the output dtype."""
import numpy as np
import pytest
import torch

from oracle import kernel_ref as kr

pytestmark = pytest.mark.gpu
DEV = "cuda"


def qgemm():
    import viditq_extension.qgemm as m

    return m


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize("M,N,K", [(1, 8, 16), (7, 24, 48), (128, 128, 128), (130, 136, 144), (257, 384, 1536),
                                   (1000, 1536, 1536), (333, 8960, 1536), (300, 1536, 8960), (512, 5120, 5120),
                                   # persistent kernel: one K-tile, two K-tiles (ragged M and N), nk = 32 (LDS-DMA issued at the top)
                                   (600, 264, 128), (520, 256, 256), (700, 512, 4096)])
def test_w8a8_o32_bit_exact(M, N, K):
    rng = np.random.default_rng(M * 7 + N + K)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    w = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    out = qgemm().w8a8_o32(t(a), t(w)).cpu().numpy()
    assert np.array_equal(out, kr.w8a8_o32(a, w))


def test_w8a8_o32_extreme_values_no_overflow():
    """K = 13824 of -128 * -128: 2.26e8 < 2^31, the largest accumulator the 14B model can produce."""
    M, N, K = 40, 16, 13824
    a = np.full((M, K), -128, np.int8)
    w = np.full((N, K), -128, np.int8)
    w[1] = 127
    out = qgemm().w8a8_o32(t(a), t(w)).cpu().numpy()
    assert np.array_equal(out, kr.w8a8_o32(a, w))


def test_identity_and_asymmetric_operand_layout():
    """A = I check with an asymmetric weight catches a transposed C write or swapped operand roles."""
    n = 128
    a = np.eye(n, dtype=np.int8)
    w = (np.arange(n)[:, None] * 3 + np.arange(n)[None, :] * 5) % 200 - 100
    w = w.astype(np.int8)
    out = qgemm().w8a8_o32(t(a), t(w)).cpu().numpy()
    assert np.array_equal(out, w.T.astype(np.int32))


def test_kbench_gemm_golden(golden):
    """K/bench/bench_gemm.py:27-29 ground truth, reference buffer dtypes (fp16 scales, int16 zp)."""
    g = golden("kbench_gemm")
    args = [t(g[k]) for k in ("a", "w", "bias", "sa", "sw", "a_sum", "zp")]
    y = qgemm().w8a8_of16_bias_weight_asym(*args)
    assert y.dtype == torch.float16 and tuple(y.shape) == g["y_asym"].shape
    d = np.abs(y.float().cpu().numpy() - g["y_asym_f32"])
    assert (d <= np.maximum(np.abs(g["y_asym_f32"]) * 2 ** -10, 1e-3)).all()  # within fp16 rounding of the fp32 truth
    ys = qgemm().w8a8_of16_bias_weight_sym(t(g["a"]), t(g["w"]), t(g["bias"]), t(g["sa"]), t(g["sw"]))
    assert np.abs(ys.float().cpu().numpy() - g["y_sym"].astype(np.float32)).max() <= 1.0
    yq = qgemm().w8a8_of16_nobias_weight_sym_qserve(t(g["a"]), t(g["w"]), t(g["sa"]), t(g["sw"]))
    ref = kr.w8a8_epilogue(g["acc"], g["sa"], g["sw"])
    np.testing.assert_allclose(yq.float().cpu().numpy(), ref, rtol=1e-3, atol=1e-2)
    assert np.array_equal(qgemm().w8a8_o32(t(g["a"]), t(g["w"])).cpu().numpy(), g["acc"])


@pytest.mark.parametrize("out_dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("M,N,K", [(200, 136, 192), (4680, 1536, 1536)])
def test_epilogue_variants_fp32_vectors(out_dtype, M, N, K):
    rng = np.random.default_rng(5)
    a = rng.integers(-127, 128, size=(M, K), dtype=np.int8)
    w = rng.integers(-128, 128, size=(N, K), dtype=np.int8)
    sa = rng.uniform(0.005, 0.02, M).astype(np.float32)
    sw = rng.uniform(0.001, 0.003, N).astype(np.float32)
    zp = rng.integers(-20, 20, N).astype(np.float32)
    bias = rng.normal(size=N).astype(np.float32)
    asum = (a.sum(1) * sa).astype(np.float32)
    gate = rng.normal(size=N).astype(np.float32)
    res = rng.normal(size=(M, N)).astype(np.float32)
    eps = {torch.float16: 2 ** -10, torch.bfloat16: 2 ** -7, torch.float32: 2e-6}[out_dtype]
    acc = kr.w8a8_o32(a, w)

    def close(y, ref):
        y = y.float().cpu().numpy()
        assert (np.abs(y - ref) <= np.abs(ref) * eps + 1e-4).all()

    base = kr.w8a8_epilogue(acc, sa, sw, bias, asum, zp)
    y = qgemm().w8a8_linear(t(a), t(w), t(sa), t(sw), t(bias), t(asum), t(zp), out_dtype=out_dtype)
    close(y, base)
    y = qgemm().w8a8_linear(t(a), t(w), t(sa), t(sw), t(bias), t(asum), t(zp), out_dtype=out_dtype, gelu=True)
    close(y, kr.gelu_tanh(base))
    res_t = t(res).to(out_dtype)
    y = qgemm().w8a8_linear(t(a), t(w), t(sa), t(sw), t(bias), t(asum), t(zp), out_dtype=out_dtype, gate=t(gate), residual=res_t)
    close(y, res_t.float().cpu().numpy() + base * gate[None, :])
    # in-place residual update (out aliases residual), as the block uses it
    buf = res_t.clone()
    qgemm().w8a8_linear(t(a), t(w), t(sa), t(sw), t(bias), t(asum), t(zp), out_dtype=out_dtype, gate=t(gate), residual=buf, out=buf)
    assert torch.equal(buf, y)


@pytest.mark.parametrize("M,N,K", [(32760, 1536, 256), (32760, 1536, 1536), (9450, 5120, 384)])
def test_fp32_gate_residual_in_place_many_tiles(M, N, K):
    """The fp32 + gate + residual epilogue of the persistent kernel prefetches the residual lines by LDS-DMA into the stage the
    NEXT tile's first K-tile is loaded into: several tiles per workgroup, a short K (the store loop dominates, waves drift
    apart), a ragged last m-tile, in place, repeated -- against the epilogue recomputed in torch from the exact accumulators."""
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=DEV, generator=g)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=DEV, generator=g)
    sa = torch.rand(M, device=DEV, generator=g) * 0.01 + 1e-3
    asum = a.float().sum(1) * sa
    sw = torch.rand(N, device=DEV, generator=g) * 0.01 + 1e-3
    zp = torch.randn(N, device=DEV, generator=g).round()
    bias = torch.randn(N, device=DEV, generator=g)
    gate = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g)
    acc = qgemm().w8a8_o32(a, w).double()
    ref = res.double() + gate.double()[None, :] * (acc * sa.double()[:, None] * sw.double()[None, :] +
                                                   asum.double()[:, None] * (zp.double() * sw.double())[None, :] + bias.double()[None, :])
    scale = ref.abs().max().item()
    for _ in range(5):
        buf = res.clone()
        qgemm().w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.float32, gate=gate, residual=buf, out=buf)
        err = (buf.double() - ref).abs().max().item()
        assert torch.isfinite(buf).all() and err < 1e-5 * scale, err


def test_qlinear_module_from_linear_and_forward():
    """W8A8OF16LinearDynamicInputScale.from_linear + forward == fake-quant linear on the same codes."""
    from viditq_extension.nn import LayerNormGeneral, QuantParams, W8A8OF16LinearDynamicInputScale  # noqa: F401
    import viditq_extension.fused as fused

    g = torch.Generator().manual_seed(0)
    lin = torch.nn.Linear(1536, 384)
    lin.weight.data = torch.randn(384, 1536, generator=g) * 0.05
    lin = lin.half().to(DEV)
    ql = W8A8OF16LinearDynamicInputScale.from_linear(lin, weight_sym=False)
    x = torch.randn(2, 77, 1536, generator=g).half().to(DEV)
    qp = QuantParams(2 * 77, has_sum_input=True, device=DEV)
    xq = fused.quant_sum(x, qp.sum_input, qp.scale_input)
    y = ql(xq, qp)
    assert y.shape == (2, 77, 384) and y.dtype == torch.float16
    ref = kr.fake_quant_linear_from_int(xq.cpu().numpy().reshape(-1, 1536).astype(np.int32), qp.scale_input.float().cpu().numpy(),
                                        ql.weight.cpu().numpy().astype(np.int32), ql.scale_weight.float().cpu().numpy(),
                                        ql.zp_weight.float().cpu().numpy(), ql.bias.float().cpu().numpy())
    np.testing.assert_allclose(y.float().cpu().numpy().reshape(-1, 384), ref, rtol=5e-3, atol=5e-3)
    # and close to the fp16 linear itself (quantisation error only)
    yfp = torch.nn.functional.linear(x, lin.weight, lin.bias)
    assert (y - yfp).abs().max().item() < 0.05 * yfp.abs().max().item()


def test_shape_errors_raise_runtime_error():
    a = torch.zeros(8, 24, dtype=torch.int8, device=DEV)
    w = torch.zeros(16, 24, dtype=torch.int8, device=DEV)
    with pytest.raises(RuntimeError, match="K=24"):
        qgemm().w8a8_o32(a, w)
    with pytest.raises(RuntimeError, match="shape"):
        qgemm().w8a8_o32(torch.zeros(8, 32, dtype=torch.int8, device=DEV), w)


def _pack_w4_ref(q, bias):
    """The library's packed layout restated in numpy (include/wanq_hip.h): per 32 codes 16 bytes = (P0a, P1a, P0b, P1b);
    for a 16-code half e: P0 byte i = u[i] | u[4+i] << 4, P1 byte i = u[8+i] | u[12+i] << 4, u = e + bias."""
    N, K = q.shape
    u = (q.astype(np.int32) + bias).reshape(N, K // 16, 16)
    p0 = u[:, :, 0:4] | (u[:, :, 4:8] << 4)
    p1 = u[:, :, 8:12] | (u[:, :, 12:16] << 4)
    return np.concatenate([p0, p1], axis=2).reshape(N, K // 2).astype(np.uint8)


def test_w4_pack_unpack_roundtrip_and_w4a8_equation():
    """4-bit codes survive pack/unpack exactly; the QServe W4A8 entry point reproduces its equation
    (oracle/kernel_ref.py::w4a8_of16, K/csrc/qgemm/w4a8/w4a8_per_channel_gemm_cuda_qserve.cu:580-587)."""
    rng = np.random.default_rng(44)
    N, K, M = 136, 256, 77
    q = rng.integers(-8, 8, size=(N, K), dtype=np.int8)
    packed = qgemm().pack_w4(t(q), bias=8)
    assert packed.shape == (N, K // 2) and packed.dtype == torch.uint8
    assert np.array_equal(packed.cpu().numpy(), _pack_w4_ref(q, 8))
    assert np.array_equal(qgemm().unpack_w4(packed, bias=8).cpu().numpy(), q)

    u4 = rng.integers(0, 16, size=(N, K), dtype=np.int8)
    assert np.array_equal(qgemm().pack_w4(t(u4), bias=0).cpu().numpy(), _pack_w4_ref(u4, 0))
    a = rng.integers(-127, 128, size=(M, K), dtype=np.int8)
    ws = rng.uniform(0.01, 0.03, N).astype(np.float16)
    zw = rng.integers(0, 16, N).astype(np.float32)
    w_sz = (ws.astype(np.float32) * zw).astype(np.float16)
    asc = rng.uniform(0.005, 0.02, M).astype(np.float16)
    a_ssum = (a.sum(1) * asc.astype(np.float32)).astype(np.float16)
    out = torch.zeros(M, N, dtype=torch.float16, device=DEV)
    qgemm().w4a8_of16_nobias_weight_asym_qserve(t(a), qgemm().pack_w4(t(u4), bias=0), t(ws), t(asc), t(w_sz), t(a_ssum), out)
    refy = kr.w4a8_of16(a, u4, ws, asc, w_sz, a_ssum)
    np.testing.assert_allclose(out.float().cpu().numpy(), refy, rtol=4e-3, atol=4e-2)


@pytest.mark.parametrize("M,N,K", [(77, 136, 256), (130, 128, 32), (512, 256, 128), (1000, 520, 1536), (2048, 1536, 8960),
                                   (700, 264, 13824), (300, 5120, 13824)])
def test_w4a8_accumulators_bit_exact(M, N, K):
    """int8 activations x packed 4-bit weights, nibbles expanded in registers (both kernels: 128x128 for small / ragged
    shapes, persistent 256x256 from M = 512): int32 accumulators == the integer product on the unpacked codes, exactly."""
    rng = np.random.default_rng(M + N + K)
    a = rng.integers(-128, 128, size=(M, K), dtype=np.int8)
    u4 = rng.integers(0, 16, size=(N, K), dtype=np.int8)
    if K == 13824:  # extreme values: |acc| up to 13824 * 128 * 15 = 2.65e7, still exact in int32
        a[:2] = -128
        u4[:3] = 15
    acc = qgemm().w4a8_o32(t(a), qgemm().pack_w4(t(u4), bias=0))
    assert np.array_equal(acc.cpu().numpy(), kr.w8a8_o32(a, u4))


def test_w4a8_epilogue_matches_w8a8_on_unpacked_codes():
    """Same epilogue as the W8 kernel (gelu / gate + residual / bias), signed codes stored with bias 8 and zp - 8."""
    rng = np.random.default_rng(9)
    M, N, K = 777, 384, 512
    a = t(rng.integers(-127, 128, size=(M, K), dtype=np.int8))
    q = rng.integers(-8, 8, size=(N, K), dtype=np.int8)
    sa, sw = t(rng.uniform(0.005, 0.02, M).astype(np.float32)), t(rng.uniform(0.01, 0.03, N).astype(np.float32))
    asum = (a.float().sum(1) * sa).contiguous()
    zp = t(rng.integers(0, 16, N).astype(np.float32))
    bias = t(rng.normal(size=N).astype(np.float32))
    gate, res = t(rng.normal(size=N).astype(np.float32)), t(rng.normal(size=(M, N)).astype(np.float32))
    packed = qgemm().pack_w4(t(q), bias=8)
    for kw in (dict(out_dtype=torch.bfloat16, gelu=True), dict(out_dtype=torch.float32, gate=gate, residual=res),
               dict(out_dtype=torch.float16)):
        y8 = qgemm().w8a8_linear(a, t(q), sa, sw, bias, asum, zp, **kw)
        y4 = qgemm().w8a8_linear(a, packed, sa, sw, bias, asum, zp - 8.0, w4=True, **kw)
        # acc_u + (zp - 8) * sum  vs  acc_q + zp * sum: the same value through different fp32 roundings -> 1 ulp of the output type
        # (the +8 bias makes acc_u larger than acc_q and cancels in the zero-point term: ~1e-6 of the largest term in fp32)
        tol = 1e-4 if kw["out_dtype"] == torch.float32 else 1e-2
        assert torch.allclose(y4.float(), y8.float(), rtol=tol, atol=tol * float(y8.float().abs().max())), (y4.float() - y8.float()).abs().max()


@pytest.mark.parametrize("M,N,K,kw", [(4096, 1536, 8960, dict(out_dtype=torch.bfloat16, gelu=True)),
                                      (9450, 1024, 13824, dict(out_dtype=torch.float32, gate=True)),
                                      (2048, 520, 512, dict(out_dtype=torch.float16)), (33000, 264, 256, dict(out_dtype=torch.bfloat16))])
def test_w4a8_expanded_once_at_large_m_is_bit_equal_to_the_in_kernel_expansion(M, N, K, kw):
    """From WANQ_W4_UNPACK_ROWS rows on, w8a8_linear(w4=True) expands the packed weights once per launch and runs the W8 ping-pong
    kernel; the in-register expansion of wanq_gemm_w4a8 (forced by a threshold of 0) gives the same bits."""
    mod = qgemm()
    rng = np.random.default_rng(M + N)
    a = t(rng.integers(-127, 128, size=(M, K), dtype=np.int8))
    q = rng.integers(-8, 8, size=(N, K), dtype=np.int8)
    sa, sw = t(rng.uniform(0.005, 0.02, M).astype(np.float32)), t(rng.uniform(0.01, 0.03, N).astype(np.float32))
    asum = (a.float().sum(1) * sa).contiguous()
    zp = t(rng.integers(0, 16, N).astype(np.float32)) - 8.0
    bias = t(rng.normal(size=N).astype(np.float32))
    kw = dict(kw)
    if kw.pop("gate", False):
        kw.update(gate=t(rng.normal(size=N).astype(np.float32)), residual=t(rng.normal(size=(M, N)).astype(np.float32)))
    packed = mod.pack_w4(t(q), bias=8)
    keep = mod._W4_UNPACK_ROWS
    try:
        mod._W4_UNPACK_ROWS = 2048
        y_once = mod.w8a8_linear(a, packed, sa, sw, bias, asum, zp, w4=True, **kw)
        mod._W4_UNPACK_ROWS = 0
        y_tile = mod.w8a8_linear(a, packed, sa, sw, bias, asum, zp, w4=True, **kw)
    finally:
        mod._W4_UNPACK_ROWS = keep
    assert torch.equal(y_once, y_tile)
    assert torch.equal(packed, mod.pack_w4(t(q), bias=8))  # the weights at rest are untouched


# ---------------------------------------------------------------------------------------------------------------------------
# The ping-pong persistent kernel (csrc/gemm_w8a8_pp.hip) against the other two kernels in ONE process
# (wanq_gemm_select_kernel): same accumulators; the two persistent kernels share the epilogue expression -> bit-identical.
def _select(which):
    from viditq_extension import _C

    prev = _C.lib.wanq_gemm_select_kernel(which)
    assert prev >= 0
    return prev


@pytest.fixture
def kernel_select():
    yield _select
    _select(0)


PP_SHAPES = [  # (M, N, K): two / three / odd numbers of K-tiles, ragged M and N, several tiles per workgroup, one workgroup only
    (512, 256, 256), (777, 264, 384), (2100, 520, 640), (4680, 1536, 1536), (33000, 1536, 256), (9450, 1024, 1152), (600, 4096, 512)]


@pytest.mark.parametrize("M,N,K", PP_SHAPES)
def test_pingpong_kernel_bit_equal_to_the_other_kernels(kernel_select, M, N, K):
    g = torch.Generator(device=DEV).manual_seed(M * 3 + N + K)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=DEV, generator=g)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=DEV, generator=g)
    sa = torch.rand(M, device=DEV, generator=g) * 0.01 + 1e-3
    asum = a.float().sum(1) * sa
    sw = torch.rand(N, device=DEV, generator=g) * 0.01 + 1e-3
    zp = torch.randn(N, device=DEV, generator=g).round() * 3
    bias = torch.randn(N, device=DEV, generator=g)
    gate = torch.randn(N, device=DEV, generator=g)
    res = torch.randn(M, N, device=DEV, generator=g)

    def run_all():
        outs = [qgemm().w8a8_o32(a, w)]
        for od in (torch.bfloat16, torch.float16, torch.float32):
            outs.append(qgemm().w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=od))
            outs.append(qgemm().w8a8_linear(a, w, sa, sw, None, None, None, out_dtype=od))  # symmetric weights, no bias
        outs.append(qgemm().w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, gelu=True))
        outs.append(qgemm().w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.float32, gate=gate, residual=res))
        # fp16 vectors + int16 zero points (the reference's buffer dtypes): the kernel's non-prefetched scale path
        outs.append(qgemm().w8a8_linear(a, w, sa.half(), sw.half(), bias.half(), asum.half(), zp.to(torch.int16), out_dtype=torch.float16))
        buf = res.clone()  # in place, as the block calls it
        qgemm().w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.float32, gate=gate, residual=buf, out=buf)
        outs.append(buf)
        return outs

    kernel_select(0)
    y_pp = run_all()
    kernel_select(2)
    y_v2 = run_all()
    kernel_select(1)
    y_v1 = run_all()
    assert np.array_equal(y_pp[0].cpu().numpy(), kr.w8a8_o32(a.cpu().numpy(), w.cpu().numpy()))
    assert torch.equal(y_pp[0], y_v1[0])
    for i, (p_, v2, v1) in enumerate(zip(y_pp, y_v2, y_v1)):
        assert torch.equal(p_, v2), f"output {i}: ping-pong != persistent"
        # the 128 x 128 kernel sums the epilogue's three terms in another order: equal to one unit of the output type
        eps = {torch.int32: 0, torch.float16: 2 ** -10, torch.bfloat16: 2 ** -7, torch.float32: 4e-6}[p_.dtype]
        d = (p_.double() - v1.double()).abs()
        assert (d <= v1.double().abs() * eps + 1e-4).all(), f"output {i}: ping-pong vs 128x128 kernel"


def test_pingpong_kernel_repeated_launches_are_deterministic(kernel_select):
    """Race screen of the LDS ring: 20 launches of a several-tiles-per-workgroup problem, every output equal to the first."""
    M, N, K = 32760, 1536, 1536
    g = torch.Generator(device=DEV).manual_seed(11)
    a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device=DEV, generator=g)
    w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=DEV, generator=g)
    kernel_select(0)
    first = qgemm().w8a8_o32(a, w)
    for _ in range(20):
        assert torch.equal(qgemm().w8a8_o32(a, w), first)
    kernel_select(2)
    assert torch.equal(qgemm().w8a8_o32(a, w), first)
