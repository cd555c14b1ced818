"""CPU oracle for the kernel-mode ("hardware") operators: numpy restatement of the equations of
the reference's CUDA extension `viditq_extension` (K/ = /root/reference/ViDiT-Q/kernels/).

TEST INFRASTRUCTURE ONLY (see oracle/qdiff_ref.py header).

Parity status: the CUDA extension cannot be built or run here (inline PTX, needs nvcc + an NVIDIA
GPU), so these functions restate (i) the arithmetic read from K/csrc/fused/fused.cu and
K/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu and (ii) the closed-form ground truths the reference's own
kernel checks use (K/bench/bench_gemm.py:27-29, bench_quant_kernel.py:8-11,24-26,
bench_layer_norm_kernel.py:15-16,34-36,47-49).  They are PINNED against tests/golden/kbench_*.npz,
which evaluate those bench formulas with torch-CPU on the benches' input distributions.

Rounding convention (SURVEY Appendix A, section 7 "hard parts"): the bench ground truth and the
simulation path quantise with a DIVISION, q = rne(x / (amax/127)); the CUDA kernel multiplies by
127/amax.  The build follows the division form (it is the one the reference checks against and the
one that is bit-identical to qdiff's DynamicQuantizer), plus qdiff's eps rule for all-zero rows.
"""
import numpy as np

from .qdiff_ref import F32, dynamic_quantize_sym


def _f(x):
    return np.asarray(x).astype(F32)


# ------------------------------------------------------------------ N10 quant_sum
def quant_sum(x):
    """fused.quant_sum: per-row dynamic int8 quant + dequantised row sum.  K/csrc/fused/fused.cu:30-131.
    Returns (q int8[T,C], scale fp32[T], sum fp32[T]); callers round scale/sum to the storage dtype."""
    q, scale = dynamic_quantize_sym(_f(x), 8)
    s = (q.sum(axis=1).astype(F32) * scale).astype(F32)
    return q.astype(np.int8), scale, s


def gelu_tanh(x):
    """tanh-GELU in fp32: 0.5 x (1 + tanh(0.79788456 (x + 0.044715 x^3))).  K/csrc/fused/fused.cu:22-26."""
    x = _f(x)
    inner = F32(0.79788456) * (x + F32(0.044715) * x * x * x)
    return (F32(0.5) * x * (F32(1) + np.tanh(inner).astype(F32))).astype(F32)


def gelu_quant_sum(x):
    """fused.gelu_quant_sum.  K/csrc/fused/fused.cu:134-232 (GELU evaluated in fp32 here)."""
    return quant_sum(gelu_tanh(x))


# ------------------------------------------------------------------ N7 / N8 layernorm family
def layernorm_nobias(x, gamma, eps):
    """LN without bias: (x-mean)*rstd*gamma, fp32 statistics.  K/csrc/fused/fused.cu:259-290.
    Computed in fp64 here and rounded once, so HIP results are compared with a tolerance."""
    x64 = np.asarray(x, np.float64)
    mean = x64.mean(axis=-1, keepdims=True)
    var = ((x64 - mean) ** 2).mean(axis=-1, keepdims=True)
    y = (x64 - mean) / np.sqrt(var + eps)
    if gamma is not None:
        y = y * np.asarray(gamma, np.float64)
    return y


def layernorm_t2i(x, gamma, shift, scale, eps, rows_per_batch):
    """LN -> *(1+scale[b]) + shift[b].  K/csrc/fused/fused.cu:293-307 (fp32 here, half2 there) and
    W/wan/modules/model.py:327 (`norm1(x).float()*(1+e[1])+e[0]`, fp32)."""
    y = layernorm_nobias(x, gamma, eps)
    b = np.arange(y.shape[0]) // rows_per_batch
    return y * (1.0 + np.asarray(scale, np.float64)[b]) + np.asarray(shift, np.float64)[b]


def layernorm_t2i_quant_sum(x, gamma, shift, scale, eps, rows_per_batch):
    """fused.layernorm_nobias_t2i_quant_sum_fuse.  K/csrc/fused/fused.cu:234-380,854-915."""
    return quant_sum(layernorm_t2i(x, gamma, shift, scale, eps, rows_per_batch).astype(F32))


# ------------------------------------------------------------------ N9 gate residual
def gate_residual(x, gate, residual, rows_per_batch):
    """fused.gate_residual_fuse: x*gate[b] + residual.  K/csrc/fused/fused.cu:382-483,917-961."""
    b = np.arange(np.asarray(x).shape[0]) // rows_per_batch
    return (_f(x) * _f(gate)[b] + _f(residual)).astype(F32)


# ------------------------------------------------------------------ N1 / N2 / N3 int8 GEMM
def w8a8_o32(a, w):
    """qgemm.w8a8_o32: int32 accumulators of A[M,K] . W[N,K]^T.  K/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu:780-838."""
    return (np.asarray(a, np.int32) @ np.asarray(w, np.int32).T).astype(np.int32)


def w8a8_epilogue(acc, sa, sw, bias=None, a_sum=None, zp=None):
    """fp32 epilogue, evaluated left to right as the kernel does:
    acc*sA*sB (+ sumA*zp*sB) (+ bias).  K/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu:416-441."""
    y = (acc.astype(F32) * _f(sa)[:, None]).astype(F32) * _f(sw)[None, :]
    if zp is not None:
        y = y + ((_f(a_sum)[:, None] * _f(zp)[None, :]).astype(F32) * _f(sw)[None, :]).astype(F32)
    if bias is not None:
        y = y + _f(bias)[None, :]
    return y.astype(F32)


def w8a8_of16_bias_weight_asym(a, w, bias, sa, sw, a_sum, zp):
    """qgemm.w8a8_of16_bias_weight_asym -> fp16.  K/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu:624-704."""
    return w8a8_epilogue(w8a8_o32(a, w), sa, sw, bias, a_sum, zp).astype(np.float16)


def w8a8_of16_bias_weight_sym(a, w, bias, sa, sw):
    """qgemm.w8a8_of16_bias_weight_sym -> fp16.  K/csrc/qgemm/w8a8/w8a8_gemm_cuda.cu:707-777."""
    return w8a8_epilogue(w8a8_o32(a, w), sa, sw, bias).astype(np.float16)


def w4a8_of16(a, u4, wscale, ascale, w_sz, a_ssum):
    """qgemm.w4a8_of16_nobias_weight_asym_qserve equation with UNSIGNED 4-bit weights u in [0,15]:
    y = acc*sW*sA - (sW*zW)*sumA.  K/csrc/qgemm/w4a8/w4a8_per_channel_gemm_cuda_qserve.cu:580-587."""
    acc = (np.asarray(a, np.int32) @ np.asarray(u4, np.int32).T).astype(F32)
    y = (acc * _f(wscale)[None, :]) * _f(ascale)[:, None] - _f(w_sz)[None, :] * _f(a_ssum)[:, None]
    return y.astype(F32)


# ------------------------------------------------------------------ link between the two modes
def fake_quant_linear_from_int(q_a, delta_a, q_w, delta_w, zp_w, bias):
    """What sim mode computes with the same integer codes: F.linear(q_a*da, (q_w+zp)*dw) + bias
    (Q/base/quant_layer.py:72, Q/base/base_quantizer.py:56-59) == the asym kernel epilogue with
    sumA = da * sum_k q_a.  Used to check the two modes agree."""
    xa = q_a.astype(np.float64) * np.asarray(delta_a, np.float64)[:, None]
    ww = (q_w.astype(np.float64) + np.asarray(zp_w, np.float64)[:, None]) * np.asarray(delta_w, np.float64)[:, None]
    y = xa @ ww.T
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y
