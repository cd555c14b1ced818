"""`viditq_extension.fused` -- same functions as the reference pybind module
(ViDiT-Q/kernels/csrc/fused/pybind.cpp:5-99), executed by libwanq_hip (csrc/rowwise.hip).

Differences from the reference, all widenings:
  * input / output may be fp16, bf16 or fp32 (reference: fp16 only); per-token vectors fp16 or fp32;
  * any hidden size that is a multiple of 8 up to 16384 (reference: %128, <= 4096 threads*4 / 8192);
  * statistics, modulation and the quantisation division are fp32 (reference: half2 modulate,
    multiply by 127/amax) -- the int8 codes equal qdiff's DynamicQuantizer bit for bit;
  * all-zero rows give scale 1e-6 and zeros (reference: inf/NaN scale).
"""
import torch

from . import _C

_FP = (torch.float16, torch.bfloat16, torch.float32)
_VEC = (torch.float16, torch.float32)


def _rows_cols(name, t):
    _C.check_gpu(name, t)
    _C.check_contig(name, t)
    _C.check_dtype(name, t, *_FP)
    cols = t.shape[-1]
    return t.numel() // cols, cols


def _check_vec(name, t, rows):
    _C.check_gpu(name, t)
    _C.check_contig(name, t)
    _C.check_dtype(name, t, *_VEC)
    _C.check_shape(name, t, rows)


def _quant_rows(input, sum_output, scaling, act, static_amax):
    rows, cols = _rows_cols("input", input)
    _check_vec("scaling", scaling, rows)
    if sum_output is not None:
        _check_vec("sum_output", sum_output, rows)
        if sum_output.dtype != scaling.dtype:
            raise RuntimeError("sum_output and scaling must share a dtype")
    _C.check_same_device(input, sum_output, scaling)
    out = torch.empty(input.shape, dtype=torch.int8, device=input.device)
    with torch.cuda.device(input.device):
        _C.call("wanq_quant_rows", _C.ptr(input), _C.dt(input), _C.ptr(out), _C.ptr(scaling), _C.ptr(sum_output),
                _C.dt(scaling), rows, cols, act, static_amax, _C.stream(),
                hbm=("gelu_quant_sum" if act else "quant_sum", rows * cols * (input.element_size() + 1) + 2 * rows * scaling.element_size()))
    return out


def quant_sum(input, sum_output, scaling):
    """int8 = quant(input); writes per-row scale into `scaling`, dequantised row sum into `sum_output`."""
    return _quant_rows(input, sum_output, scaling, 0, 0)


def quant_sum_levels(input, sum_output, scaling, n_levels, floor=1e-6):
    """quant_sum at a narrower symmetric range: codes rne(x / scale), scale = max(absmax / n_levels, floor), n_levels = 2^(b-1) - 1
    (the activation quantisers at n_bits < 8 and MixedPrecisionDynamicQuantizer, whose symmetric branch has floor = 0)."""
    rows, cols = _rows_cols("input", input)
    _check_vec("scaling", scaling, rows)
    if sum_output is not None:
        _check_vec("sum_output", sum_output, rows)
        if sum_output.dtype != scaling.dtype:
            raise RuntimeError("sum_output and scaling must share a dtype")
    _C.check_same_device(input, sum_output, scaling)
    out = torch.empty(input.shape, dtype=torch.int8, device=input.device)
    with torch.cuda.device(input.device):
        _C.call("wanq_quant_rows_levels", _C.ptr(input), _C.dt(input), _C.ptr(out), _C.ptr(scaling), _C.ptr(sum_output),
                _C.dt(scaling), rows, cols, int(n_levels), float(floor), _C.stream())
    return out


def quant_sum_static(input, sum_output, scaling):
    """Like quant_sum, but `scaling` is an INPUT holding the per-row absmax (reference fused.cu:84-86)."""
    return _quant_rows(input, sum_output, scaling, 0, 1)


def gelu_quant_sum(input, sum_output, scaling):
    """tanh-GELU then quant_sum."""
    return _quant_rows(input, sum_output, scaling, 1, 0)


def _layernorm(output, input, weight, shift_msa, scale_msa, sum_output, scaling, epsilon, quant):
    rows, cols = _rows_cols("input", input)
    _C.check_gpu("output", output)
    _C.check_contig("output", output)
    if output.numel() != input.numel():
        raise RuntimeError("Tensor output must have as many elements as input")
    batch, mod_stride, mod_dtype = 1, 0, _C.F32
    mods = [weight, shift_msa, scale_msa]
    present = [m for m in mods if m is not None]
    if present:
        md = present[0].dtype if all(m.dtype == present[0].dtype for m in present) else torch.float32
        _C.check_dtype("weight/shift/scale", present[0], *_FP)
        mod_dtype = _C.dt(md)
        if weight is not None:
            _C.check_gpu("weight", weight)
            _C.check_shape("weight", weight, cols)
            weight = weight.to(md).contiguous()
        if shift_msa is not None or scale_msa is not None:
            ref = shift_msa if shift_msa is not None else scale_msa
            if ref.dim() != 2 or ref.shape[1] != cols:
                raise RuntimeError(f"shift_msa/scale_msa must have shape (batch, {cols})")
            batch = ref.shape[0]
            if rows % batch:
                raise RuntimeError("rows must be a multiple of the modulation batch size")

            def prep(name, m):
                if m is None:
                    return None
                _C.check_gpu(name, m)
                _C.check_shape(name, m, batch, cols)
                m = m.to(md)
                return m if m.stride(1) == 1 else m.contiguous()

            shift_msa, scale_msa = prep("shift_msa", shift_msa), prep("scale_msa", scale_msa)
            strides = {m.stride(0) for m in (shift_msa, scale_msa) if m is not None}
            if len(strides) > 1:
                shift_msa, scale_msa = shift_msa.contiguous(), scale_msa.contiguous()
            mod_stride = (shift_msa if shift_msa is not None else scale_msa).stride(0)
    q = scale_p = sum_p = out_fp = None
    vec_dt, out_dt = _C.F32, _C.F32
    if quant:
        _C.check_dtype("output", output, torch.int8)
        _check_vec("scaling", scaling, rows)
        if sum_output is not None:
            _check_vec("sum_output", sum_output, rows)
            if sum_output.dtype != scaling.dtype:
                raise RuntimeError("sum_output and scaling must share a dtype")
        q, scale_p, sum_p, vec_dt = _C.ptr(output), _C.ptr(scaling), _C.ptr(sum_output), _C.dt(scaling)
    else:
        _C.check_dtype("output", output, *_FP)
        out_fp, out_dt = _C.ptr(output), _C.dt(output)
    _C.check_same_device(input, output, weight, shift_msa, scale_msa, sum_output, scaling)
    with torch.cuda.device(input.device):
        _C.call("wanq_layernorm_rows", _C.ptr(input), _C.dt(input), _C.ptr(weight), _C.ptr(shift_msa), _C.ptr(scale_msa),
                mod_dtype, mod_stride, rows // batch, float(epsilon), out_fp, out_dt, q, scale_p, sum_p, vec_dt,
                rows, cols, _C.stream(),
                hbm=("layernorm_quant" if quant else "layernorm", rows * cols * (input.element_size() + output.element_size()) + 8 * rows * bool(quant)))


def layernorm_nobias(out, input, weight, epsilon):
    _layernorm(out, input, weight, None, None, None, None, epsilon, False)


def layernorm_nobias_quant_nosum_fuse(out, input, weight, scaling, epsilon):
    _layernorm(out, input, weight, None, None, None, scaling, epsilon, True)


def layernorm_nobias_quant_sum_fuse(output, input, weight, sum_output, scaling, epsilon):
    _layernorm(output, input, weight, None, None, sum_output, scaling, epsilon, True)


def layernorm_nobias_t2i_fuse(output, input, weight, shift_msa, scale_msa, epsilon):
    _layernorm(output, input, weight, shift_msa, scale_msa, None, None, epsilon, False)


def layernorm_nobias_t2i_quant_sum_fuse(output, input, weight, shift_msa, scale_msa, sum_output, scaling, epsilon):
    _layernorm(output, input, weight, shift_msa, scale_msa, sum_output, scaling, epsilon, True)


def gate_residual_fuse(input, gate_msa, residual, out_dtype=None):
    """input * gate_msa[b] + residual.  input/residual: [batch*tokens, hidden]; gate_msa: [batch, hidden]."""
    rows, cols = _rows_cols("input", input)
    _rows_cols("residual", residual)
    _C.check_gpu("gate_msa", gate_msa)
    _C.check_dtype("gate_msa", gate_msa, *_FP)
    if gate_msa.dim() != 2 or gate_msa.shape[1] != cols:
        raise RuntimeError(f"Tensor gate_msa must have shape (batch, {cols})")
    if gate_msa.stride(1) != 1:
        raise RuntimeError("Tensor gate_msa must be contiguous at the last dimension")
    batch = gate_msa.shape[0]
    if residual.numel() != input.numel() or rows % batch:
        raise RuntimeError("input, residual and gate_msa shapes do not agree")
    _C.check_same_device(input, gate_msa, residual)
    out = torch.empty(input.shape, dtype=out_dtype or input.dtype, device=input.device)
    with torch.cuda.device(input.device):
        _C.call("wanq_gate_residual", _C.ptr(input), _C.dt(input), _C.ptr(gate_msa), _C.dt(gate_msa), gate_msa.stride(0),
                _C.ptr(residual), _C.dt(residual), _C.ptr(out), _C.dt(out), rows, cols, rows // batch, _C.stream())
    return out


def gate_residual_into_(residual, input, gate_msa):
    """residual <- input * gate_msa[b] + residual, in place (fp32 stream, bf16/fp16 update)."""
    rows, cols = _rows_cols("input", input)
    _rows_cols("residual", residual)
    _C.check_gpu("gate_msa", gate_msa)
    _C.check_dtype("gate_msa", gate_msa, *_FP)
    batch = gate_msa.shape[0]
    with torch.cuda.device(input.device):
        _C.call("wanq_gate_residual", _C.ptr(input), _C.dt(input), _C.ptr(gate_msa), _C.dt(gate_msa), gate_msa.stride(0),
                _C.ptr(residual), _C.dt(residual), _C.ptr(residual), _C.dt(residual), rows, cols, rows // batch, _C.stream())
    return residual


# ---- calibration / PTQ reductions (no counterpart in the reference extension; they replace torch
#      reductions in get_calib_data_wanx.py:262-263 and qdiff/base/base_quantizer.py:70-90)
def col_absmax_(running_max, x):
    """running_max[c] = max(running_max[c], max_r |x[r,c]|) in place; running_max: fp32 [C]."""
    rows, cols = _rows_cols("x", x)
    _C.check_gpu("running_max", running_max)
    _C.check_dtype("running_max", running_max, torch.float32)
    _C.check_contig("running_max", running_max)
    _C.check_shape("running_max", running_max, cols)
    _C.check_same_device(x, running_max)
    with torch.cuda.device(x.device):
        _C.call("wanq_col_absmax", _C.ptr(x), _C.dt(x), _C.ptr(running_max), rows, cols, _C.stream(),
                hbm=("col_absmax", rows * cols * x.element_size() + 4 * cols))
    return running_max


def fake_quant_with_delta(x, delta, n_bits=8, bits=None):
    """DynamicQuantizer.forward_with_quant_params as one elementwise launch: x fake-quantised with a precomputed fp32 `delta` of its
    own shape on the unsigned range 2^b - 1, or with a per-element int32 bit-width map `bits` (0 = masked to zero)."""
    _C.check_gpu("x", x)
    _C.check_contig("x", x)
    _C.check_dtype("x", x, *_FP)
    _C.check_gpu("delta", delta)
    _C.check_contig("delta", delta)
    _C.check_dtype("delta", delta, torch.float32)
    _C.check_shape("delta", delta, *x.shape)
    if bits is not None:
        _C.check_gpu("bits", bits)
        _C.check_contig("bits", bits)
        _C.check_dtype("bits", bits, torch.int32)
        _C.check_shape("bits", bits, *x.shape)
    _C.check_same_device(x, delta, bits)
    n = x.numel()
    if n % 8:
        raise RuntimeError(f"fake_quant_with_delta: {n} elements, must be a multiple of 8")
    out = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _C.call("wanq_fake_quant_with_delta", _C.ptr(x), _C.dt(x), _C.ptr(delta), _C.ptr(bits), _C.ptr(out), _C.dt(out), int(n_bits), n,
                _C.stream())
    return out


def fake_quant_cols_(x, n_bits=8, colmax=None):
    """In place: every column of x [rows, cols] fake-quantised with its own dynamic symmetric scale over all rows (the v recipe of
    the reference's quantized attention: per (head, channel) over all tokens).  Returns (x, colmax)."""
    rows, cols = _rows_cols("x", x)
    if colmax is None:
        colmax = torch.zeros(cols, dtype=torch.float32, device=x.device)
        col_absmax_(colmax, x)
    with torch.cuda.device(x.device):
        _C.call("wanq_fake_quant_cols", _C.ptr(x), _C.dt(x), _C.ptr(colmax), _C.ptr(x), _C.dt(x), int(n_bits), rows, cols, _C.stream())
    return x, colmax


def row_minmax(w):
    """(min, max, absmax) per row of a 2-D weight, fp32."""
    rows, cols = _rows_cols("w", w)
    o = torch.empty(3, rows, dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        _C.call("wanq_row_minmax", _C.ptr(w), _C.dt(w), _C.ptr(o[0]), _C.ptr(o[1]), _C.ptr(o[2]), rows, cols, _C.stream())
    return o[0], o[1], o[2]


def weight_quant(w, delta, zero_point, qmin, qmax, want_int8=True, want_dequant=False):
    """Static per-row quantisation with given fp32 params; returns (int8 codes | None, fp32 fake-quant | None)."""
    rows, cols = _rows_cols("w", w)
    for n, t in (("delta", delta), ("zero_point", zero_point)):
        _C.check_gpu(n, t)
        _C.check_dtype(n, t, torch.float32)
        _C.check_contig(n, t)
        _C.check_shape(n, t, rows)
    q8 = torch.empty(w.shape, dtype=torch.int8, device=w.device) if want_int8 else None
    dq = torch.empty(w.shape, dtype=torch.float32, device=w.device) if want_dequant else None
    with torch.cuda.device(w.device):
        _C.call("wanq_weight_quant", _C.ptr(w), _C.dt(w), _C.ptr(delta), _C.ptr(zero_point), int(qmin), int(qmax),
                _C.ptr(q8), _C.ptr(dq), rows, cols, _C.stream())
    return q8, dq


def weight_export_f16(w, delta_f16, zp_f16):
    """int8 codes of `w` by the reference's half-precision export equation (quantize_and_save_weight_,
    ViDiT-Q/examples/Wan2.1/wan/quant_wanx_cuda.py:39-53): clamp(round(f16(w) / delta) - zp, -128, 127)."""
    rows, cols = _rows_cols("w", w)
    for n, t in (("delta_f16", delta_f16), ("zp_f16", zp_f16)):
        _C.check_gpu(n, t)
        _C.check_dtype(n, t, torch.float16)
        _C.check_contig(n, t)
        _C.check_shape(n, t, rows)
    q8 = torch.empty(w.shape, dtype=torch.int8, device=w.device)
    with torch.cuda.device(w.device):
        _C.call("wanq_weight_export_f16", _C.ptr(w), _C.dt(w), _C.ptr(delta_f16), _C.ptr(zp_f16), _C.ptr(q8), rows, cols,
                _C.stream())
    return q8


# ---- ViDiT activation transform fused with the quantiser (no counterpart in the reference extension: its kernel mode
#      skips the transform altogether; simulation mode does x*mask -> x.double() @ R in torch)
def _rotation_args(premul, rotation, cols, device):
    """rotation: None (no transform: not accepted by the rotate entry points) or (had_k, _) from
    qdiff.quarot.quarot_utils.kernel_rotation_params; the +-1 table across the 128-wide blocks is a fixed function of the
    width and lives inside the library, so only had_k = cols / 128 crosses the boundary."""
    had_k = 0 if rotation is None else rotation[0]
    if premul is not None:
        _C.check_gpu("premul", premul)
        _C.check_dtype("premul", premul, torch.float32)
        _C.check_contig("premul", premul)
        _C.check_shape("premul", premul, cols)
    return int(had_k)


def rotate_quant(input, premul, rotation, sum_output, scaling, out_fp=None, quantize=True):
    """y = hadU(input * premul) -> int8 codes (+ scale / sum) and / or the fp result in `out_fp`.
    rotation: (had_k, _) from qdiff.quarot.quarot_utils.kernel_rotation_params."""
    rows, cols = _rows_cols("input", input)
    had_k = _rotation_args(premul, rotation, cols, input.device)
    q = None
    if quantize:
        _check_vec("scaling", scaling, rows)
        if sum_output is not None:
            _check_vec("sum_output", sum_output, rows)
        q = torch.empty(input.shape, dtype=torch.int8, device=input.device)
    if out_fp is not None:
        _rows_cols("out_fp", out_fp)
    with torch.cuda.device(input.device):
        _C.call("wanq_rotate_quant_rows", _C.ptr(input), _C.dt(input), _C.ptr(premul), had_k,
                _C.ptr(out_fp), _C.dt(out_fp) if out_fp is not None else _C.F32, _C.ptr(q),
                _C.ptr(scaling) if quantize else None, _C.ptr(sum_output) if quantize else None,
                _C.dt(scaling) if quantize else _C.F32, rows, cols, _C.stream(),
                hbm=("rotate_quant", rows * cols * (input.element_size() + bool(quantize) + (out_fp.element_size() if out_fp is not None else 0))))
    return q


def layernorm_rotate_quant(output, input, weight, shift_msa, scale_msa, premul, rotation, sum_output, scaling, epsilon):
    """layernorm_nobias_t2i_quant_sum_fuse with the ViDiT transform between the modulation and the quantiser."""
    rows, cols = _rows_cols("input", input)
    had_k = _rotation_args(premul, rotation, cols, input.device)
    _C.check_gpu("output", output)
    _C.check_dtype("output", output, torch.int8)
    _C.check_contig("output", output)
    _check_vec("scaling", scaling, rows)
    if sum_output is not None:
        _check_vec("sum_output", sum_output, rows)
    batch, mod_stride = 1, 0
    mods = [m for m in (weight, shift_msa, scale_msa) if m is not None]
    for m in mods:
        _C.check_gpu("weight/shift/scale", m)
        _C.check_dtype("weight/shift/scale", m, torch.float32)
    if weight is not None:
        _C.check_shape("weight", weight, cols)
        weight = weight.contiguous()
    ref = shift_msa if shift_msa is not None else scale_msa
    if ref is not None:
        batch = ref.shape[0]
        mod_stride = ref.stride(0)
        for m in (shift_msa, scale_msa):
            if m is not None and (m.dim() != 2 or m.shape != ref.shape or m.stride(1) != 1 or m.stride(0) != mod_stride):
                raise RuntimeError("shift_msa / scale_msa must be [batch, cols] views with equal strides")
    with torch.cuda.device(input.device):
        _C.call("wanq_layernorm_rotate_quant_rows", _C.ptr(input), _C.dt(input), _C.ptr(weight), _C.ptr(shift_msa),
                _C.ptr(scale_msa), _C.F32, mod_stride, rows // batch, float(epsilon), _C.ptr(premul), had_k,
                _C.ptr(output), _C.ptr(scaling), _C.ptr(sum_output), _C.dt(scaling), rows, cols, _C.stream(),
                hbm=("layernorm_rotate_quant", rows * cols * (input.element_size() + 1) + 8 * rows))


def layernorm_rotate_quant_multi(outputs, input, weight, shift_msa, scale_msa, premuls, rotation, sum_outputs, scalings, epsilon):
    """layernorm_rotate_quant for up to three consumers of the same normalised row (self-attention q / k / v): `input` is
    read and normalised once; consumer t gets hadU(LN(input) * premuls[t]) quantised into outputs[t] / scalings[t] /
    sum_outputs[t].  All consumers share `rotation` (same width); each has its own premul = channel_mask * signs."""
    n = len(outputs)
    if not (1 <= n <= 3 and len(premuls) == n and len(sum_outputs) == n and len(scalings) == n):
        raise RuntimeError("layernorm_rotate_quant_multi: 1..3 sets, lists of equal length")
    rows, cols = _rows_cols("input", input)
    had_k = 0
    for t in range(n):
        had_k = _rotation_args(premuls[t], rotation, cols, input.device)
        _C.check_gpu("output", outputs[t])
        _C.check_dtype("output", outputs[t], torch.int8)
        _C.check_contig("output", outputs[t])
        _C.check_shape("output", outputs[t], rows, cols)
        _check_vec("scaling", scalings[t], rows)
        _check_vec("sum_output", sum_outputs[t], rows)
        if scalings[t].dtype != scalings[0].dtype or sum_outputs[t].dtype != scalings[0].dtype:
            raise RuntimeError("layernorm_rotate_quant_multi: scale / sum vectors must share one dtype")
    batch, mod_stride = 1, 0
    for m in (weight, shift_msa, scale_msa):
        if m is not None:
            _C.check_gpu("weight/shift/scale", m)
            _C.check_dtype("weight/shift/scale", m, torch.float32)
    if weight is not None:
        _C.check_shape("weight", weight, cols)
        weight = weight.contiguous()
    ref = shift_msa if shift_msa is not None else scale_msa
    if ref is not None:
        batch = ref.shape[0]
        mod_stride = ref.stride(0)
        for m in (shift_msa, scale_msa):
            if m is not None and (m.dim() != 2 or m.shape != ref.shape or m.stride(1) != 1 or m.stride(0) != mod_stride):
                raise RuntimeError("shift_msa / scale_msa must be [batch, cols] views with equal strides")
    with torch.cuda.device(input.device):
        _C.call("wanq_layernorm_rotate_quant_rows_multi", _C.ptr(input), _C.dt(input), _C.ptr(weight), _C.ptr(shift_msa),
                _C.ptr(scale_msa), _C.F32, mod_stride, rows // batch, float(epsilon), n, _C.ptr_array(premuls),
                had_k, _C.ptr_array(outputs), _C.ptr_array(scalings), _C.ptr_array(sum_outputs), _C.dt(scalings[0]), rows, cols,
                _C.stream(), hbm=(f"layernorm_rotate_quant_x{n}", rows * cols * (input.element_size() + n) + 8 * rows * n))
    return outputs

