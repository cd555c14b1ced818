"""CPU oracle: numpy restatement of the reference's fake-quant ("simulation") path.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported by the product
(wan2.1-quantization_amd/); only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg use it, and only as the checker / reported baseline.

Parity status: PINNED.  Every function here is checked bit-for-bit (integer codes)
or to <=1 ulp (fp32 params) against golden vectors produced by importing the
reference's own `qdiff` package in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz, tests/test_oracle_golden.py).

Citations: Q/ = /root/reference/ViDiT-Q/quant_utils/qdiff/,
           W/ = /root/reference/ViDiT-Q/examples/Wan2.1/.

All arithmetic is IEEE fp32 with round-to-nearest-even exactly where the reference
(torch CPU/GPU fp32) has it; np.round == torch.round (half to even).
"""
import numpy as np

F32 = np.float32


# ------------------------------------------------------------------ A2 dynamic per-token
def dynamic_quant_params_sym(x, n_bits=8):
    """Per-row delta of DynamicQuantizer (sym).  Q/base/base_quantizer.py:116-128.

    delta = absmax / (2**(b-1) - 1); entries below eps=1e-6 are set to eps (the
    reference reaches this through its assert/except branch)."""
    x = np.asarray(x, dtype=F32)
    assert x.ndim == 2 and not np.isnan(x).any()  # :112-113
    n_levels = F32(2 ** (n_bits - 1) - 1)  # :32
    delta = (np.abs(x).max(axis=1) / n_levels).astype(F32)
    delta = np.where(delta < F32(1e-6), F32(1e-6), delta).astype(F32)
    return delta


def dynamic_quantize_sym(x, n_bits=8):
    """q = clamp(rne(x / delta), -n-1, n), zero_point = 0.  Q/base/base_quantizer.py:154-157.

    Returns (q as int32, delta fp32[T])."""
    x = np.asarray(x, dtype=F32)
    delta = dynamic_quant_params_sym(x, n_bits)
    n = 2 ** (n_bits - 1) - 1
    q = np.round((x / delta[:, None]).astype(F32))
    q = np.clip(q, -n - 1, n)
    return q.astype(np.int32), delta


def dynamic_fake_quant_sym(x, n_bits=8):
    """DynamicQuantizer.forward: (q + 0) * delta.  Q/base/base_quantizer.py:159-162."""
    q, delta = dynamic_quantize_sym(x, n_bits)
    return (q.astype(F32) * delta[:, None]).astype(F32)


def dynamic_quant_params_asym(x, n_bits=8):
    """Per-row (delta, zero_point) of DynamicQuantizer, asymmetric branch.  Q/base/base_quantizer.py:130-149:
    x_max clipped to >= 0, x_min to <= 0, delta = (x_max - x_min) / (2**b - 1) floored at 1e-8, zp = rne(x_min / delta) + 2**b / 2."""
    x = np.asarray(x, dtype=F32)
    assert x.ndim == 2 and not np.isnan(x).any()
    n_levels = 2 ** n_bits
    hi = np.maximum(x.max(axis=1), F32(0)).astype(F32)
    lo = np.minimum(x.min(axis=1), F32(0)).astype(F32)
    delta = ((hi - lo) / F32(n_levels - 1)).astype(F32)
    delta = np.where(delta < F32(1e-8), F32(1e-8), delta).astype(F32)
    zp = (np.round((lo / delta).astype(F32)) + F32(n_levels / 2)).astype(F32)
    return delta, zp


def dynamic_quantize_asym(x, n_bits=8):
    """x_int = rne(x / delta) - zp clamped to [-2**b - 1, 2**b].  Q/base/base_quantizer.py:154-157.  -> (q int32, delta, zp)."""
    x = np.asarray(x, dtype=F32)
    delta, zp = dynamic_quant_params_asym(x, n_bits)
    n = 2 ** n_bits
    q = np.clip(np.round((x / delta[:, None]).astype(F32)) - zp[:, None], -n - 1, n)
    return q.astype(np.int32), delta, zp


def dynamic_fake_quant_asym(x, n_bits=8):
    """DynamicQuantizer.forward, asymmetric: (q + zp) * delta.  Q/base/base_quantizer.py:159-162."""
    q, delta, zp = dynamic_quantize_asym(x, n_bits)
    return ((q.astype(F32) + zp[:, None]) * delta[:, None]).astype(F32)


# ------------------------------------------------------------------ A7 mixed-precision dynamic (list of bit-widths)
def mixed_dynamic_quantize(x, n_bits, sym):
    """MixedPrecisionDynamicQuantizer.quantize at the ACTIVE bit-width `n_bits` (Q/base/mixed_precision_quantizer.py:135-175;
    `bitwidth_refactor(i)` :182-186 only selects `n_bits`, n_levels is recomputed per call :139).  Unlike DynamicQuantizer:
    symmetric -- delta = absmax / (2**(b-1) - 1) with NO eps floor (:141-146), so an all-zero row gives 0 / 0 = NaN codes, exactly as
    the reference (pinned with its NaNs by tests/golden/a7_mixed_dynamic.npz); asymmetric -- floor 1e-6 (:158-165; DynamicQuantizer
    has 1e-8).  -> (x_quant fp32 with the reference's NaNs, delta fp32[T], zero_point fp32[T])."""
    x = np.asarray(x, dtype=F32)
    assert x.ndim == 2
    with np.errstate(all="ignore"):
        if sym:
            n = 2 ** (n_bits - 1) - 1
            delta = (np.abs(x).max(axis=1) / F32(n)).astype(F32)
            zp = np.zeros_like(delta)
        else:
            n = 2 ** n_bits
            hi = np.maximum(x.max(axis=1), F32(0)).astype(F32)
            lo = np.minimum(x.min(axis=1), F32(0)).astype(F32)
            delta = ((hi - lo) / F32(n - 1)).astype(F32)
            delta = np.where(delta < F32(1e-6), F32(1e-6), delta).astype(F32)
            zp = (np.round((lo / delta).astype(F32)) + F32(n / 2)).astype(F32)
        q = np.round((x / delta[:, None]).astype(F32)) - zp[:, None]
        q = np.where(np.isnan(q), q, np.clip(q, -n - 1, n)).astype(F32)  # torch.clamp keeps NaN
    return q, delta, zp


def mixed_dynamic_fake_quant(x, n_bits, sym):
    """MixedPrecisionDynamicQuantizer.forward: (x_quant + zero_point) * delta.  Q/base/mixed_precision_quantizer.py:177-180."""
    q, delta, zp = mixed_dynamic_quantize(x, n_bits, sym)
    with np.errstate(all="ignore"):
        return ((q + zp[:, None]) * delta[:, None]).astype(F32)


# ------------------------------------------------------------------ A16 fake-quant with a precomputed delta
def fake_quant_with_delta(x, delta, n_bits=8, mixed_precision=None):
    """DynamicQuantizer.forward_with_quant_params (Q/base/base_quantizer.py:164-206; symmetric quantisers only, :167): the fake-quant
    step of the reference's block-wise attention-map quantisers with a PRECOMPUTED delta of x's own shape.  delta < 1e-6 -> 1e-6
    (:181-189).  Plain: d = delta / (2 n + 1), n = 2**(b-1) - 1, y = clamp(rne(x / d), 0, 2 n + 1) * d (:196-199).  mixed_precision
    (integer bit-widths, x's shape): levels 2**bits - 1, codes clipped from ABOVE only, 0-bit elements masked to zero (:174-178,
    :191-195, :203-204).  -> (y fp32, the floored delta)."""
    x = np.asarray(x, dtype=F32)
    d0 = np.asarray(delta, dtype=F32).copy()
    d0[d0 < F32(1e-6)] = F32(1e-6)
    if mixed_precision is not None:
        bits = np.asarray(mixed_precision).astype(np.int64)
        nl = (2 ** bits - 1).astype(np.int64)
        zero = nl == 0
        nl = np.where(zero, 255, nl)
        d = (d0 / nl.astype(F32)).astype(F32)
        xi = np.round((x / d).astype(F32))
        xq = np.where(xi > nl, nl.astype(F32), xi).astype(F32)
        return (xq * d * (~zero)).astype(F32), d0
    n = 2 ** (n_bits - 1) - 1
    d = (d0 / F32(2 * n + 1)).astype(F32)
    xq = np.clip(np.round((x / d).astype(F32)), 0, 2 * n + 1).astype(F32)
    return (xq * d).astype(F32), d0


# ------------------------------------------------------------------ A1 static per-channel
def static_quant_params(w, n_bits=8, sym=False):
    """StaticQuantizer.init_quant_params.  Q/base/base_quantizer.py:70-99.

    asym: max+ = max(row,0), min- = min(row,0), delta=(max+ - min-)/(2**b - 1),
          zp = rne(min-/delta) + 2**b/2.
    sym : delta = absmax/(2**(b-1)-1), zp = 0."""
    w = np.asarray(w, dtype=F32)
    assert w.ndim == 2
    if sym:
        n_levels = F32(2 ** (n_bits - 1) - 1)
        delta = (np.abs(w).max(axis=1) / n_levels).astype(F32)
        zp = np.zeros_like(delta)
    else:
        n_levels = 2 ** n_bits
        x_max = np.maximum(w.max(axis=1), F32(0))
        x_min = np.minimum(w.min(axis=1), F32(0))
        delta = ((x_max - x_min).astype(F32) / F32(n_levels - 1)).astype(F32)
        zp = (np.round((x_min / delta).astype(F32)) + F32(n_levels / 2)).astype(F32)
    return delta, zp


def _static_levels(n_bits, sym):
    return (2 ** (n_bits - 1) - 1) if sym else 2 ** n_bits


def static_quantize(w, delta, zp, n_bits=8, sym=False):
    """x_int = rne(x/delta) - zp; clamp(-n_levels-1, n_levels).  Q/base/base_quantizer.py:61-68.
    (The clamp is looser than the bit-width -- SURVEY D9; kept as is.)"""
    w = np.asarray(w, dtype=F32)
    n = _static_levels(n_bits, sym)
    q = np.round((w / delta[:, None]).astype(F32)) - zp[:, None]
    return np.clip(q, -n - 1, n).astype(np.int32)


def static_fake_quant(w, n_bits=8, sym=False, params=None):
    """StaticQuantizer.forward: (q + zp) * delta.  Q/base/base_quantizer.py:56-59."""
    delta, zp = params if params is not None else static_quant_params(w, n_bits, sym)
    q = static_quantize(w, delta, zp, n_bits, sym)
    return ((q.astype(F32) + zp[:, None]) * delta[:, None]).astype(F32), delta, zp


def mixed_static_quant_params(w, bit_list, sym=False):
    """MixedPrecisionStaticQuantizer.init_quant_params: one (delta, zp) per bit-width.
    Q/base/mixed_precision_quantizer.py:79-125."""
    ds, zs = zip(*[static_quant_params(w, b, sym) for b in bit_list])
    return np.stack(ds), np.stack(zs)


def export_int8_weight(w, delta, zp):
    """quantize_and_save_weight_: fp16 weight, fp16 delta/zp -> int8.  W/wan/quant_wanx_cuda.py:39-53.
    torch evaluates fp16 (w/scale) and the subtraction in fp16 with RNE; numpy float16 does the same."""
    w16 = np.asarray(w, dtype=F32).astype(np.float16)
    s16 = delta.astype(np.float16)
    z16 = zp.astype(np.float16)
    q = np.round((w16 / s16[:, None]).astype(np.float16)) - z16[:, None]
    return np.clip(q.astype(np.float16), -128, 127).astype(np.int8), s16, z16


# ------------------------------------------------------------------ A3 QuantizedLinear
def quantized_linear(x, w_dequant, bias, n_bits_a=8):
    """QuantizedLinear.forward (fp32, no autocast).  Q/base/quant_layer.py:57-74.
    x: [B, T, C]; w_dequant: the fake-quantized weight the ctor stored (:38-39)."""
    B, T, C = x.shape
    xq = dynamic_fake_quant_sym(x.reshape(B * T, C), n_bits_a).reshape(B, T, C)
    y = xq.astype(F32) @ w_dequant.astype(F32).T
    if bias is not None:
        y = y + bias.astype(F32)
    return y.astype(F32)


# ------------------------------------------------------------------ A5 Hadamard
def _is_pow2(n):
    return n > 0 and (n & (n - 1)) == 0


def paley1(q):
    """Paley type-I Hadamard matrix of order q+1 (q prime, q = 3 mod 4), normalised the way the
    reference's literal tables are: first column +1, first row (+1,-1,...,-1), core[i][j] = chi(i-j)
    off the diagonal and +1 on it.  Reproduces get_had12/20/60/108/140
    (Q/quarot/quarot_utils.py:269ff; checked in tests/test_oracle_golden.py against products of them)."""
    chi = -np.ones(q, dtype=np.int64)
    chi[[(a * a) % q for a in range(1, q)]] = 1
    chi[0] = 0
    i = np.arange(q)
    H = np.ones((q + 1, q + 1), dtype=np.int64)
    H[0, 1:] = -1
    H[1:, 1:] = chi[(i[:, None] - i[None, :]) % q] + np.eye(q, dtype=np.int64)
    return H


_PALEY = {12: 11, 20: 19, 60: 59, 108: 107, 140: 139}


def had_k(n, strict=True):
    """get_hadK: pick the non-power-of-two factor K and its table.  Q/quarot/quarot_utils.py:100-155.
    Same precedence order as the reference; tables we cannot construct raise NotImplementedError.
    strict (the reference): the FIRST K that divides n must leave a power-of-two co-factor, else AssertionError -- 13824 = 144 x 96
    dies at :110-112 although its K = 108 branch (:118-121) would fit.
    strict=False (REPO-DEFINED, not the reference): a K whose co-factor is not a power of two is skipped and the reference's own
    order continues: 13824 -> K = 108 (Paley-107, the reference's get_had108 table) x H_128.  Sizes the reference accepts are
    unaffected.  Pinned by tests/golden/a5_hadamard_13824.npz (made with the reference's table and butterfly)."""
    for K in (172, 156, 144, 140, 108, 60, 52, 36, 28, 40, 20, 12):
        if n % K == 0:
            if not _is_pow2(n // K):
                assert not strict, f"{n} = {K} x non-power-of-two"
                continue
            if K in _PALEY:
                return paley1(_PALEY[K]), K
            if K == 40:  # the reference's had40 equals [[H20,H20],[H20,-H20]]
                return np.kron(np.array([[1, 1], [1, -1]]), paley1(19)), K
            raise NotImplementedError(f"Hadamard order {K} table not constructed here")
    assert _is_pow2(n)
    return None, 1


def matmul_hadU(X, strict=True):
    """(hadK (x) H_{n/K}) X / sqrt(n) along the last axis.  Q/quarot/quarot_utils.py:158-179.  strict: see had_k.

    The reference's butterfly loop is the natural-order Walsh-Hadamard transform on each of the K
    contiguous blocks, followed by hadK across blocks; the divisor is fp32 sqrt (`torch.tensor(n).sqrt()`)."""
    X = np.asarray(X)
    n = X.shape[-1]
    hadK, K = had_k(n, strict)
    m = n // K
    v = X.reshape(-1, K, m).astype(X.dtype, copy=True)
    h = 1
    while h < m:  # stage with stride h pairs element bit log2(h)
        v = v.reshape(-1, K, m // (2 * h), 2, h)
        a, b = v[..., 0, :], v[..., 1, :]
        v = np.stack([a + b, a - b], axis=-2).reshape(-1, K, m)
        h *= 2
    if K > 1:
        v = np.einsum("kj,bjm->bkm", hadK.astype(X.dtype), v)
    div = np.sqrt(F32(n))  # fp32 value
    return (v.reshape(X.shape) / X.dtype.type(div)).astype(X.dtype)


def hadamard_from_signs(signs):
    """random_hadamard_matrix with the +-1 draw made explicit: R = hadU(diag(s)) in fp64.
    Q/quarot/quarot_utils.py:186-192.  Row i of R is s_i * hadU(e_i), so x @ R == hadU(x * s)."""
    s = np.asarray(signs, dtype=np.float64)
    return matmul_hadU(np.diag(s))


# ------------------------------------------------------------------ A4 ViDiT layer
def vidit_channel_mask(w, act_mask, alpha):
    """get_channel_mask: w_absmax_in**alpha / act_absmax**(1-alpha).  Q/viditq/viditq_quant_layer.py:30-35."""
    w = np.asarray(w, dtype=F32)
    wm = np.abs(w).max(axis=0)
    return ((np.abs(wm) ** F32(alpha)) / (np.abs(np.asarray(act_mask, F32)) ** F32(1 - alpha))).astype(F32)


def vidit_weight(w, channel_mask, R, n_bits=8, sym=False):
    """update_quantized_weight_rotated_and_scaled: W1 = Q(W/mask); W2 = Q((W1.double() @ R).float()).
    Q/viditq/viditq_quant_layer.py:40-50.  Returns (w_final fp32, delta, zp) of the SECOND quantisation."""
    w = np.asarray(w, dtype=F32)
    w1, _, _ = static_fake_quant((w / channel_mask[None, :]).astype(F32), n_bits, sym)
    w2in = (w1.astype(np.float64) @ R).astype(F32)
    return static_fake_quant(w2in, n_bits, sym)


def vidit_act_transform(x, channel_mask, R):
    """x * mask -> (x.double() @ R).to(fp32).  Q/viditq/viditq_quant_layer.py:62-63."""
    xs = (np.asarray(x, F32) * channel_mask.astype(F32)).astype(F32)
    return (xs.astype(np.float64) @ R).astype(F32)


def vidit_linear(x, w_final, bias, channel_mask, R, n_bits_a=8):
    """ViDiTQuantizedLinear.forward (fp32).  Q/viditq/viditq_quant_layer.py:52-73."""
    B, T, C = x.shape
    xr = vidit_act_transform(x.reshape(B * T, C), channel_mask, R)
    xq = dynamic_fake_quant_sym(xr, n_bits_a)
    y = xq @ w_final.astype(F32).T
    if bias is not None:
        y = y + bias.astype(F32)
    return y.astype(F32).reshape(B, T, -1)


# ------------------------------------------------------------------ A8 calibration
def calib_channel_absmax(x):
    """SaveActivationHook default branch: reshape([-1,C]).abs().max(dim=0).  W/get_calib_data_wanx.py:262-263."""
    x = np.asarray(x, dtype=F32)
    return np.abs(x.reshape(-1, x.shape[-1])).max(axis=0)


def calib_act_mask(stacked):
    """init_rotation_and_channel_mask_: max over calls, floor 1e-3.  W/ptq_wanx.py:336-341."""
    m = np.asarray(stacked, F32).max(axis=0)
    return np.where(m < F32(1e-3), F32(1e-3), m).astype(F32)
