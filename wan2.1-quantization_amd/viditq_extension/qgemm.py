"""`viditq_extension.qgemm` -- same functions as the reference pybind module
(ViDiT-Q/kernels/csrc/qgemm/pybind.cpp:5-12), executed by the int8-MFMA kernel in csrc/gemm_w8a8.hip.

Widenings: any M (ragged last tile in-kernel), N % 8 == 0, K % 16 == 0 (reference: M,N % 128, K % 64, and a
C assert that aborts the process otherwise); per-token / per-channel vectors fp16 or fp32; zero point
int16 or fp32; output fp16 (reference), bf16 or fp32; launched on the CURRENT stream (reference: legacy
default stream).
"""
import os

import torch

from . import _C

_VEC = (torch.float16, torch.float32)

_timer = None


def set_timer(t):
    """bench.py hook: when set to a list, every GEMM launch appends (start_event, end_event, int8 ops), the
    events bracketing exactly that launch on the current stream."""
    global _timer
    _timer = t


def _check_i8(name, t, rows=None, cols=None):
    _C.check_gpu(name, t)
    _C.check_contig(name, t)
    _C.check_dtype(name, t, torch.int8)
    if t.dim() != 2:
        raise RuntimeError(f"Tensor {name} must have dimension number (2)")
    if rows is not None:
        _C.check_shape(name, t, rows, cols)


def _check_vec(name, t, n, dtypes=_VEC):
    if t is None:
        return
    _C.check_gpu(name, t)
    _C.check_contig(name, t)
    _C.check_dtype(name, t, *dtypes)
    _C.check_shape(name, t, n)


# Packed 4-bit weights at large M: the int8 matrix cores are the bound, not the weight bytes, so the fastest W4A8 product on this
# machine is the W8 ping-pong kernel on codes expanded ONCE per launch (wanq_unpack_w4: 1.5 N K bytes of traffic, 1-6 % of the
# product's time from 32760 down to 9450 rows) instead of once per tile inside the kernel (wanq_gemm_w4a8's in-register expansion
# costs ~20 % there).  The nibbles u = q + 8 go in as they are: the 8 rides in the zero point both ways (include/wanq_hip.h), the
# int32 accumulators and the epilogue are the same, outputs bit-equal (tests/test_gpu_gemm.py).  Weights stay packed at rest; the
# expanded copy lives for the launch.  WANQ_W4_UNPACK_ROWS: minimum M (0 = never).
_W4_UNPACK_ROWS = int(os.environ.get("WANQ_W4_UNPACK_ROWS", "2048"))


def w8a8_linear(input, weight, scale_input, scale_weight, bias=None, input_sum=None, zp_weight=None,
                out_dtype=torch.float16, gelu=False, gate=None, residual=None, out=None, w4=False):
    """General entry: y = epilogue(int8 input[M,K] @ int8 weight[N,K]^T); see include/wanq_hip.h.

    gate (fp32 [N]) + residual ([M,N], out dtype): y = residual + y * gate (may alias `out`).
    w4=True: `weight` is uint8 [N, K/2], unsigned 4-bit codes in the pack_w4 layout (wanq_gemm_w4a8; at M >= WANQ_W4_UNPACK_ROWS and
    K % 128 == 0 expanded once and multiplied by the W8 ping-pong kernel, see _W4_UNPACK_ROWS)."""
    _check_i8("input", input)
    M, K = input.shape
    if w4:
        _C.check_gpu("weight", weight)
        _C.check_contig("weight", weight)
        _C.check_dtype("weight", weight, torch.uint8)
        N = weight.shape[0]
        _C.check_shape("weight", weight, N, K // 2)
        if K % 32:
            raise RuntimeError(f"w4a8: K={K} must be a multiple of 32")
    else:
        _check_i8("weight", weight)
        N = weight.shape[0]
        _C.check_shape("weight", weight, N, K)
    _check_vec("scale_input", scale_input, M)
    _check_vec("scale_weight", scale_weight, N)
    _check_vec("bias", bias, N)
    _check_vec("input_sum", input_sum, M)
    _check_vec("zp_weight", zp_weight, N, (torch.int16, torch.float32))
    if input_sum is not None and input_sum.dtype != scale_input.dtype:
        raise RuntimeError("input_sum and scale_input must share a dtype")
    if bias is not None and bias.dtype != scale_weight.dtype:
        raise RuntimeError("bias and scale_weight must share a dtype")
    if zp_weight is not None and input_sum is None:
        raise RuntimeError("asymmetric weights (zp_weight) need input_sum")
    epi = _C.EPI_GELU if gelu else 0
    if gate is not None or residual is not None:
        if gate is None or residual is None:
            raise RuntimeError("gate and residual must be given together")
        _check_vec("gate", gate, N, (torch.float32,))
        _C.check_gpu("residual", residual)
        _C.check_contig("residual", residual)
        _C.check_dtype("residual", residual, out_dtype)
        _C.check_shape("residual", residual, M, N)
        epi |= _C.EPI_GATE_RES
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=input.device)
    else:
        _C.check_gpu("out", out)
        _C.check_contig("out", out)
        _C.check_dtype("out", out, out_dtype)
        _C.check_shape("out", out, M, N)
    _C.check_same_device(input, weight, scale_input, scale_weight, bias, input_sum, zp_weight, gate, residual, out)
    with torch.cuda.device(input.device):
        if _timer is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if w4 and _W4_UNPACK_ROWS and M >= _W4_UNPACK_ROWS and K % 128 == 0 and K >= 256:
            weight, w4 = unpack_w4(weight, bias=0, out=_w4_scratch_for(weight)), False  # inside the timed region: part of this product's cost
        _C.call("wanq_gemm_w4a8" if w4 else "wanq_gemm_w8a8", _C.ptr(input), _C.ptr(weight), _C.ptr(out), _C.dt(out_dtype), _C.ptr(scale_input),
                _C.ptr(input_sum), _C.dt(scale_input), _C.ptr(scale_weight), _C.ptr(bias), _C.dt(scale_weight),
                _C.ptr(zp_weight), _C.dt(zp_weight) if zp_weight is not None else _C.F32, _C.ptr(gate),
                _C.ptr(residual), epi, M, N, K, _C.stream())
        if _timer is not None:
            ev1.record()
            _timer.append((ev0, ev1, 2 * M * N * K))
    return out


def w8a8_of16_bias_weight_asym(input, weight, bias, scale_input, scale_weight, input_sum, zp_weight):
    """fp16( acc*sA*sB + sumA*zp*sB + bias )."""
    return w8a8_linear(input, weight, scale_input, scale_weight, bias, input_sum, zp_weight)


def w8a8_of16_bias_weight_sym(input, weight, bias, scale_input, scale_weight):
    """fp16( acc*sA*sB + bias )."""
    return w8a8_linear(input, weight, scale_input, scale_weight, bias)


def w8a8_of16_nobias_weight_sym_qserve(input, weight, scale_input, scale_weight):
    """fp16( acc*sW*sA ), no bias."""
    return w8a8_linear(input, weight, scale_input, scale_weight)


def w8a8_o32(input, weight):
    """Raw int32 accumulators."""
    _check_i8("input", input)
    M, K = input.shape
    _check_i8("weight", weight)
    N = weight.shape[0]
    _C.check_shape("weight", weight, N, K)
    _C.check_same_device(input, weight)
    out = torch.empty((M, N), dtype=torch.int32, device=input.device)
    with torch.cuda.device(input.device):
        _C.call("wanq_gemm_w8a8", _C.ptr(input), _C.ptr(weight), _C.ptr(out), _C.I32, None, None, _C.F32, None, None,
                _C.F32, None, _C.F32, None, None, 0, M, N, K, _C.stream())
    return out


# ---- W4 -----------------------------------------------------------------------------------------------------------
def pack_w4(codes, bias=8):
    """int8 codes [N, K] (signed [-8,7] with bias 8, or unsigned 0..15 with bias 0) -> uint8 [N, K/2] in the library's
    packed layout (include/wanq_hip.h: 32 codes -> 16 bytes arranged as int8-MFMA operand nibbles)."""
    _check_i8("codes", codes)
    N, K = codes.shape
    out = torch.empty((N, K // 2), dtype=torch.uint8, device=codes.device)
    with torch.cuda.device(codes.device):
        _C.call("wanq_pack_w4", _C.ptr(codes), _C.ptr(out), int(bias), N, K, _C.stream())
    return out


_w4_scratch = {}


def _w4_scratch_for(packed):
    """ONE int8 [N, K] buffer per (device, stream, N, K) for the large-M W4A8 path (a fresh 70-MB tensor per launch at the
    14B FFN shapes was twice the packed weights' footprint and invisible to the memory accounting).  Launches on one stream are
    ordered, so the GEMM that reads the expansion is enqueued before the next expansion that overwrites it."""
    N, K2 = packed.shape
    key = (packed.device, torch.cuda.current_stream(packed.device).cuda_stream, N, K2 * 2)
    buf = _w4_scratch.get(key)
    if buf is None:
        buf = _w4_scratch[key] = torch.empty((N, K2 * 2), dtype=torch.int8, device=packed.device)
    return buf


def w4_scratch_bytes(device=None):
    """Bytes held by the large-M W4A8 path's expansion buffers (reported beside the sharded weights by bench.py / --dit_fsdp)."""
    return sum(b.numel() for k, b in _w4_scratch.items() if device is None or k[0] == torch.device(device))


def unpack_w4(packed, bias=8, out=None):
    """uint8 [N, K/2] -> int8 codes [N, K] (into `out` when given)."""
    _C.check_gpu("packed", packed)
    _C.check_contig("packed", packed)
    _C.check_dtype("packed", packed, torch.uint8)
    N, K2 = packed.shape
    if out is None:
        out = torch.empty((N, K2 * 2), dtype=torch.int8, device=packed.device)
    else:
        _check_i8("out", out, N, K2 * 2)
    with torch.cuda.device(packed.device):
        _C.call("wanq_unpack_w4", _C.ptr(packed), _C.ptr(out), int(bias), N, K2 * 2, _C.stream())
    return out


def w4a8_o32(input, kernel):
    """Raw int32 accumulators of int8 input [M, K] x unsigned 4-bit codes (packed uint8 [N, K/2])."""
    _check_i8("input", input)
    M, K = input.shape
    N = kernel.shape[0]
    _C.check_dtype("kernel", kernel, torch.uint8)
    _C.check_shape("kernel", kernel, N, K // 2)
    out = torch.empty((M, N), dtype=torch.int32, device=input.device)
    with torch.cuda.device(input.device):
        _C.call("wanq_gemm_w4a8", _C.ptr(input), _C.ptr(kernel), _C.ptr(out), _C.I32, None, None, _C.F32, None, None,
                _C.F32, None, _C.F32, None, None, 0, M, N, K, _C.stream())
    return out


def w4a8_of16_nobias_weight_asym_qserve(in_feats, kernel, wscales, ascales, w_szs, a_ssums, out_feats):
    """Reference signature (ViDiT-Q/kernels/csrc/qgemm/pybind.cpp:12): writes out_feats (fp16 [M, N]).
    kernel: UNSIGNED 4-bit codes packed two per byte, uint8 [N, K/2] in THIS library's layout (pack_w4(..., bias=0));
    y = acc * wscales[n] * ascales[m] - w_szs[n] * a_ssums[m]  with w_szs = scale*zero
    (w4a8_per_channel_gemm_cuda_qserve.cu:580-587).  The nibbles are expanded in registers inside the GEMM."""
    vec = ascales.dtype
    zp = (-(w_szs.float() / wscales.float())).contiguous()  # asym epilogue: + a_ssum * zp * wscale
    w8a8_linear(in_feats, kernel, ascales, wscales.to(vec), None, a_ssums, zp.float(), out_dtype=out_feats.dtype, out=out_feats,
                w4=True)
