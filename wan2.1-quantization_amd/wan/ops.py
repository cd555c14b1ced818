"""Functional wrappers over libwanq_hip for the DiT block (no module state)."""
import math

import torch

from viditq_extension import _C, fused, qgemm  # noqa: F401


def rmsnorm_rope_(x, weight, rope, head_dim, rows_per_batch=None, eps=1e-6, out=None):
    """In place by default: x[rows, C] <- RMSNorm_C(x) * weight, then rotary per head from `rope`
    (fp32 [positions, head_dim/2, 2]).  Either of weight / rope may be None."""
    _C.check_gpu("x", x)
    _C.check_contig("x", x)
    rows, cols = x.numel() // x.shape[-1], x.shape[-1]
    if weight is not None:
        _C.check_dtype("weight", weight, torch.float32)
        _C.check_shape("weight", weight, cols)
    positions = 0
    if rope is not None:
        _C.check_dtype("rope", rope, torch.float32)
        _C.check_contig("rope", rope)
        if rope.dim() != 3 or rope.shape[1] != head_dim // 2 or rope.shape[2] != 2:
            raise RuntimeError(f"rope must have shape (positions, {head_dim // 2}, 2)")
        positions = rope.shape[0]
    out = x if out is None else out
    with torch.cuda.device(x.device):
        _C.call("wanq_rmsnorm_rope", _C.ptr(x), _C.dt(x), _C.ptr(weight), _C.ptr(rope), _C.ptr(out), _C.dt(out), rows, cols,
                head_dim, rows_per_batch or rows, positions, float(eps), _C.stream(),
                hbm=("rmsnorm_rope" if rope is not None else "rmsnorm", rows * cols * (x.element_size() + out.element_size()) +
                     (min(rows, positions) * head_dim * 4 if rope is not None else 0)))
    return out


def rmsnorm_rope_scatter(x, weight, rope, head_dim, out, head_map, rows_per_batch=None, eps=1e-6):
    """rmsnorm_rope_ with the store scattered per head: head h of row r -> out.view(-1)[head_map[h, 0] + r * head_map[h, 1] + ...]
    (elements).  `out` is a flat buffer holding the Ulysses send images (SeqParallel.packed_layout); x is left untouched."""
    _C.check_gpu("x", x)
    _C.check_contig("x", x)
    _C.check_gpu("out", out)
    _C.check_contig("out", out)
    _C.check_gpu("head_map", head_map)
    _C.check_dtype("head_map", head_map, torch.int64)
    _C.check_contig("head_map", head_map)
    rows, cols = x.shape
    _C.check_shape("head_map", head_map, cols // head_dim, 2)
    if out.numel() < rows * cols:
        raise RuntimeError("rmsnorm_rope_scatter: out is smaller than the row-major image")
    if weight is not None:
        _C.check_dtype("weight", weight, torch.float32)
        _C.check_shape("weight", weight, cols)
    positions = 0
    if rope is not None:
        _C.check_dtype("rope", rope, torch.float32)
        _C.check_contig("rope", rope)
        if rope.dim() != 3 or rope.shape[1] != head_dim // 2 or rope.shape[2] != 2:
            raise RuntimeError(f"rope must have shape (positions, {head_dim // 2}, 2)")
        positions = rope.shape[0]
    with torch.cuda.device(x.device):
        _C.call("wanq_rmsnorm_rope_scatter", _C.ptr(x), _C.dt(x), _C.ptr(weight), _C.ptr(rope), _C.ptr(out), _C.dt(out),
                _C.ptr(head_map), rows, cols, head_dim, rows_per_batch or rows, positions, float(eps), _C.stream(),
                hbm=("rmsnorm_rope_scatter", 2 * rows * cols * x.element_size() + (min(rows, positions) * head_dim * 4 if rope is not None else 0)))
    return out


class Q8Rows:
    """Per-(token, head) int8 form of a q or k tensor for the int8 Q.K^T attention: codes int8 [rows, C] and the fp32 scale
    planes [2, H, stride] (delta, -12582912 * delta) that wanq_rmsnorm_rope_q8 writes."""

    def __init__(self, rows, cols, head_dim, device, pad_to=1):
        self.rows, self.cols, self.heads = rows, cols, cols // head_dim
        self.stride = -(-rows // pad_to) * pad_to
        self.codes = torch.empty(rows, cols, dtype=torch.int8, device=device)
        self.scales = torch.zeros(2, self.heads, self.stride, dtype=torch.float32, device=device)

    @classmethod
    def from_exchange(cls, codes, planes, head_dim, for_keys):
        """After the Ulysses head exchange: codes int8 [L, w] (this rank's heads, all tokens) and the token-major scale pairs
        [L, heads * 2] that travelled with them -> the [2, heads, stride] planes the attention kernel reads (key planes padded to
        a multiple of 64 rows with zeros, as wanq_rmsnorm_rope_q8 leaves them)."""
        L, w = codes.shape
        self = cls.__new__(cls)
        self.rows, self.cols, self.heads = L, w, w // head_dim
        self.stride = -(-L // 64) * 64 if for_keys else L
        self.codes = codes
        self.scales = torch.zeros(2, self.heads, self.stride, dtype=torch.float32, device=codes.device)
        self.scales[:, :, :L] = planes.view(L, self.heads, 2).permute(2, 1, 0)
        return self


def rmsnorm_rope_q8(x, weight, rope, head_dim, for_keys, eps=1e-6, want_fp=False):
    """RMSNorm_C(x) * weight -> rotary -> per-(token, head) int8 quantise: returns Q8Rows (and the bf16 row when want_fp).
    for_keys: pad the scale planes' row stride to a multiple of 64 (the attention kernel fetches key scales per 64-key tile)."""
    _C.check_gpu("x", x)
    _C.check_contig("x", x)
    rows, cols = x.shape
    if head_dim != 128:
        raise RuntimeError("the int8 Q.K^T attention is implemented for head_dim 128")
    positions = 0
    if rope is not None:
        _C.check_dtype("rope", rope, torch.float32)
        _C.check_contig("rope", rope)
        positions = rope.shape[0]
    if weight is not None:
        _C.check_dtype("weight", weight, torch.float32)
        _C.check_shape("weight", weight, cols)
    q8 = Q8Rows(rows, cols, head_dim, x.device, 64 if for_keys else 1)
    out = torch.empty(rows, cols, dtype=torch.bfloat16, device=x.device) if want_fp else None
    with torch.cuda.device(x.device):
        _C.call("wanq_rmsnorm_rope_q8", _C.ptr(x), _C.dt(x), _C.ptr(weight), _C.ptr(rope), _C.ptr(out),
                _C.BF16, _C.ptr(q8.codes), _C.ptr(q8.scales), q8.stride, rows, cols, head_dim, rows, positions, float(eps),
                _C.stream(), hbm=("rmsnorm_rope_q8", rows * cols * (x.element_size() + 1 + (2 if want_fp else 0)) +
                                  8 * rows * (cols // head_dim) + (min(rows, positions) * head_dim * 4 if rope is not None else 0)))
    return (q8, out) if want_fp else q8


def attention_qk8(q8, k8, v, num_heads, k_len=None, out=None, splits=None):
    """softmax((q8 . k8) * delta_q * delta_k / sqrt(d)) v with the score matrix on the int8 matrix cores
    (csrc/attention.hip, QK8); q8 / k8: Q8Rows, v bf16 [Lk, C] -> bf16 [Lq, C]."""
    Lq, C = q8.codes.shape
    d = C // num_heads
    _C.check_gpu("v", v)
    _C.check_dtype("v", v, torch.bfloat16)
    if v.dim() != 2 or v.shape[1] != C or v.stride(1) != 1 or v.shape[0] != k8.codes.shape[0]:
        raise RuntimeError(f"Tensor v must be [{k8.codes.shape[0]}, {C}] with unit column stride")
    Lk = k8.codes.shape[0] if k_len is None else min(int(k_len), k8.codes.shape[0])
    if out is None:
        out = torch.empty(Lq, C, dtype=torch.bfloat16, device=v.device)
    if splits is None:
        splits = attention_splits(Lq, Lk, num_heads, v.device)
    ws, nbytes = None, 0
    if splits > 1:
        nbytes = _C.lib.wanq_attention_split_workspace(Lq, num_heads, d, int(splits))
        ws = torch.empty(nbytes // 4, dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        if _attn_timer is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        _C.call("wanq_attention_qk8_fwd", _C.ptr(q8.codes), _C.ptr(q8.scales), q8.stride, _C.ptr(k8.codes), _C.ptr(k8.scales),
                k8.stride, _C.ptr(v), _C.ptr(out), _C.BF16, Lq, Lk, num_heads, d, q8.codes.stride(0), k8.codes.stride(0),
                v.stride(0), out.stride(0), 1.0 / math.sqrt(d), max(1, int(splits)), _C.ptr(ws), nbytes, _C.stream())
        if _attn_timer is not None:
            ev1.record()
            _attn_timer.append((ev0, ev1, 4 * Lq * Lk * d * num_heads))
    return out


def attention_map_quant(q, k, v, num_heads, n_bits=8, sym=False, k_len=None, out=None, q_len=None):
    """softmax(q k^T / sqrt(d)) with every KEY column of the map fake-quantised over all queries, then @ v -- the reference's
    `attn.attn_map` recipe, group 'row' (Q/base/quant_attn.py:118-173), streamed in three passes (csrc/attn_map.hip).
    q [Lq, C], k / v [Lk, C] bf16 -> [Lq, C].  q_len: number of REAL query rows (the rest is sequence padding): a column's
    quantisation step is taken over the real queries only, as the reference's map holds no padding rows; padded output rows
    are zero."""
    if isinstance(q, Q8Rows):  # the reference's full recipe: int8 q / k (attn.qk) under the map quantiser
        return _attention_map_quant_q8(q, k, v, num_heads, n_bits, sym, k_len, out, q_len)
    if q_len is not None and int(q_len) < q.shape[0]:
        n = int(q_len)
        if out is None:
            out = torch.empty(q.shape[0], q.shape[1], dtype=q.dtype, device=q.device)
        out[n:].zero_()
        if n > 0:
            attention_map_quant(q[:n], k, v, num_heads, n_bits, sym, k_len, out[:n])
        return out
    Lq, C = q.shape
    d = C // num_heads
    for n, t in (("q", q), ("k", k), ("v", v)):
        _C.check_gpu(n, t)
        _C.check_dtype(n, t, torch.bfloat16)
        if t.dim() != 2 or t.shape[1] != C or t.stride(1) != 1:
            raise RuntimeError(f"Tensor {n} must be [tokens, {C}] with unit column stride")
    if k.shape[0] != v.shape[0]:
        raise RuntimeError("k and v must have the same number of tokens")
    Lk = k.shape[0] if k_len is None else min(int(k_len), k.shape[0])
    if out is None:
        out = torch.empty(Lq, C, dtype=q.dtype, device=q.device)
    nbytes = _C.lib.wanq_attention_map_workspace(Lq, Lk, num_heads)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=q.device)
    with torch.cuda.device(q.device):
        _C.call("wanq_attention_map_quant_fwd", _C.ptr(q), _C.ptr(k), _C.ptr(v), _C.ptr(out), _C.dt(q), Lq, Lk, num_heads, d,
                q.stride(0), k.stride(0), v.stride(0), out.stride(0), 1.0 / math.sqrt(d), int(n_bits), 1 if sym else 0, _C.ptr(ws),
                nbytes, _C.stream())
    return out


def _attention_map_quant_q8(q8, k8, v, num_heads, n_bits, sym, k_len, out, q_len):
    """attention_map_quant with q and k as Q8Rows (per-(token, head) int8 codes + scale planes): S runs on the int8 matrix cores
    in all three passes -- together with a fake-quantised v the reference's whole recipe (W/models/quant_opensora.py:431-476)."""
    if not isinstance(k8, Q8Rows):
        raise RuntimeError("attention_map_quant: q and k must both be Q8Rows or both be bf16 tensors")
    rows, C = q8.codes.shape
    Lq = rows if q_len is None else min(int(q_len), rows)
    d = C // num_heads
    _C.check_gpu("v", v)
    _C.check_dtype("v", v, torch.bfloat16)
    if v.dim() != 2 or v.shape[1] != C or v.stride(1) != 1 or v.shape[0] != k8.codes.shape[0]:
        raise RuntimeError(f"Tensor v must be [{k8.codes.shape[0]}, {C}] with unit column stride")
    Lk = k8.codes.shape[0] if k_len is None else min(int(k_len), k8.codes.shape[0])
    if out is None:
        out = torch.empty(rows, C, dtype=torch.bfloat16, device=v.device)
    if Lq < rows:
        out[Lq:].zero_()  # padding rows of the sequence: not queries of the map (see attention_map_quant)
    if Lq == 0:
        return out
    nbytes = _C.lib.wanq_attention_map_workspace(Lq, Lk, num_heads)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=v.device)
    with torch.cuda.device(v.device):
        _C.call("wanq_attention_map_quant_qk8_fwd", _C.ptr(q8.codes), _C.ptr(q8.scales), q8.stride, _C.ptr(k8.codes), _C.ptr(k8.scales),
                k8.stride, _C.ptr(v), _C.ptr(out), _C.BF16, Lq, Lk, num_heads, d, q8.codes.stride(0), k8.codes.stride(0), v.stride(0),
                out.stride(0), 1.0 / math.sqrt(d), int(n_bits), 1 if sym else 0, _C.ptr(ws), nbytes, _C.stream())
    return out


def rope_table(freqs, grid, device):
    """(cos, sin) table fp32 [f*h*w, d/2, 2] for one (f,h,w) grid from the model's complex freqs [1024, d/2]
    -- the `freqs_i` of rope_apply (wan/modules/model.py:56-61), built once per grid in float64."""
    f, h, w = grid
    c = freqs.shape[1]
    parts = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    fi = torch.cat([parts[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
                    parts[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                    parts[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, c)
    return torch.view_as_real(fi).to(torch.float32).contiguous().to(device)


_attn_timer = None


def set_attention_timer(t):
    """bench.py hook: a list collecting (start_event, end_event, flops) per attention launch."""
    global _attn_timer
    _attn_timer = t


_N_CU = {}


def attention_splits(Lq, Lk, num_heads, device, ncu=None):
    """Key splits for wanq_attention_fwd_split: > 1 only when (query blocks x heads) leaves much of the last round of CUs idle
    and the key sequence is long enough to share (e.g. 3 heads x 128 blocks on 256 CUs: 1.5 rounds cost 2 -> split 2)."""
    if ncu is None:
        if device not in _N_CU:
            _N_CU[device] = torch.cuda.get_device_properties(device).multi_processor_count
        ncu = _N_CU[device]
    blocks, tiles = -(-Lq // 256) * num_heads, -(-Lk // 64)
    if tiles < 32:
        return 1
    best, cost = 1, float(-(-blocks // ncu))
    for s in (2, 3, 4):
        c = -(-(blocks * s) // ncu) / s * 1.03  # partial write-out + merge
        if c < cost * 0.93:
            best, cost = s, c
    return best


def attention(q, k, v, num_heads, k_len=None, out=None, splits=None):
    """softmax(q k^T / sqrt(d)) v for one sample on the HIP flash-attention kernel (csrc/attention.hip).
    q [Lq, C], k/v [Lk, C] bf16, token-major (row stride may exceed C: column slices of a packed buffer are
    fine) -> [Lq, C].  k_len masks key padding (flash_attention(..., k_lens), wan/modules/attention.py:78-80).
    splits: None = attention_splits() decides; 1 = one workgroup per (query block, head); n = split-KV.
    fp32 operands (the kernel-mode block built with act_dtype=float32, a parity-test configuration) are rounded to bf16 here:
    bf16 operands and a bf16 P are the kernel's contract, as they are flash_attn's in the reference."""
    if q.dtype == torch.float32 and k.dtype == torch.float32 and v.dtype == torch.float32:
        q, k, v = q.to(torch.bfloat16), k.to(torch.bfloat16), v.to(torch.bfloat16)
    Lq, C = q.shape
    d = C // num_heads
    for n, t in (("q", q), ("k", k), ("v", v)):
        _C.check_gpu(n, t)
        _C.check_dtype(n, t, torch.bfloat16)
        if t.dim() != 2 or t.shape[1] != C or t.stride(1) != 1:
            raise RuntimeError(f"Tensor {n} must be [tokens, {C}] with unit column stride")
    if k.shape[0] != v.shape[0]:
        raise RuntimeError("k and v must have the same number of tokens")
    Lk = k.shape[0] if k_len is None else min(int(k_len), k.shape[0])
    if out is None:
        out = torch.empty(Lq, C, dtype=q.dtype, device=q.device)
    if splits is None:
        splits = attention_splits(Lq, Lk, num_heads, q.device)
    with torch.cuda.device(q.device):
        if _attn_timer is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if splits <= 1:
            _C.call("wanq_attention_fwd", _C.ptr(q), _C.ptr(k), _C.ptr(v), _C.ptr(out), _C.dt(q), Lq, Lk, num_heads, d,
                    q.stride(0), k.stride(0), v.stride(0), out.stride(0), 1.0 / math.sqrt(d), _C.stream())
        else:
            nbytes = _C.lib.wanq_attention_split_workspace(Lq, num_heads, d, int(splits))
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=q.device)
            _C.call("wanq_attention_fwd_split", _C.ptr(q), _C.ptr(k), _C.ptr(v), _C.ptr(out), _C.dt(q), Lq, Lk, num_heads, d,
                    q.stride(0), k.stride(0), v.stride(0), out.stride(0), 1.0 / math.sqrt(d), int(splits), _C.ptr(ws), nbytes,
                    _C.stream())
        if _attn_timer is not None:
            ev1.record()
            _attn_timer.append((ev0, ev1, 4 * Lq * Lk * d * num_heads))
    return out


# ---------------------------------------------------------------- the fp32 ends of a DiT pass (csrc/embed_head.hip)
_ACT = {None: 0, "gelu_tanh": 1, "silu": 2}
_T_KIND = {torch.float32: 0, torch.int64: 1, torch.float64: 2, torch.int32: 3}


def _f32(name, t, *shape):
    _C.check_gpu(name, t)
    _C.check_dtype(name, t, torch.float32)
    _C.check_contig(name, t)
    if shape:
        _C.check_shape(name, t, *shape)
    return t


def linear_f32(x, weight, bias=None, in_act=None, out_act=None, rows=None):
    """act_out(act_in(x) @ weight.T + bias) in fp32 on the matrix cores; x [x_rows, K], weight [N, K].  rows > x_rows computes the
    missing input rows as zeros (text_embedding's padding to text_len, reference model.py:600-605)."""
    _f32("x", x)
    _f32("weight", weight)
    if x.dim() != 2 or weight.dim() != 2 or x.shape[1] != weight.shape[1]:
        raise RuntimeError(f"linear_f32: x {tuple(x.shape)} against weight {tuple(weight.shape)}")
    n, k = weight.shape
    if bias is not None:
        _f32("bias", bias, n)
    rows = x.shape[0] if rows is None else int(rows)
    out = torch.empty(rows, n, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _C.call("wanq_linear_f32", _C.ptr(x), x.shape[0], _C.ptr(weight), _C.ptr(bias), _C.ptr(out), rows, n, k, _ACT[in_act],
                _ACT[out_act], _C.stream())
    return out


def time_sinusoid(t, dim):
    """sinusoidal_embedding_1d(dim, t).float() (reference model.py:18-28; float64 angles on the device) -> fp32 [len(t), dim]."""
    _C.check_gpu("t", t)
    _C.check_contig("t", t)
    t = t.reshape(-1)
    if t.dtype not in _T_KIND:
        t = t.to(torch.float64)
    out = torch.empty(t.numel(), dim, dtype=torch.float32, device=t.device)
    with torch.cuda.device(t.device):
        _C.call("wanq_time_sinusoid", _C.ptr(t), _T_KIND[t.dtype], _C.ptr(out), t.numel(), dim, _C.stream())
    return out


def patch_embed(latent, weight, bias, out_rows=None):
    """Conv3d(kernel == stride) patch embedding of latent [C, F, H, W] with the convolution's own weight [N, C, pt, ph, pw]:
    -> ([out_rows, N] tokens in (f, h, w) order, rows past the token count zero; the token grid)."""
    _f32("latent", latent)
    _f32("weight", weight)
    if latent.dim() != 4 or weight.dim() != 5 or weight.shape[1] != latent.shape[0]:
        raise RuntimeError(f"patch_embed: latent {tuple(latent.shape)} against weight {tuple(weight.shape)}")
    c, f, h, w = latent.shape
    n, _, pt, ph, pw = weight.shape
    if bias is not None:
        _f32("bias", bias, n)
    grid = (f // pt, h // ph, w // pw)
    tokens = grid[0] * grid[1] * grid[2]
    out_rows = tokens if out_rows is None else int(out_rows)
    out = torch.empty(out_rows, n, dtype=torch.float32, device=latent.device)
    with torch.cuda.device(latent.device):
        _C.call("wanq_patch_embed", _C.ptr(latent), _C.ptr(weight), _C.ptr(bias), _C.ptr(out), c, f, h, w, pt, ph, pw, n, out_rows,
                _C.stream())
    return out, grid


def head(x, modulation, e, weight, bias, eps, latent_shape=None, patch=None):
    """Head.forward (reference model.py:391-399) on x [rows, K]: LayerNorm * (1 + modulation[1] + e) + modulation[0] + e, Linear.
    latent_shape = (C_out, F, H, W) with patch = (pt, ph, pw) also unpatchifies (model.py:633-656) and returns that latent;
    otherwise [rows, N]."""
    _f32("x", x)
    rows, k = x.shape
    _f32("modulation", modulation, 2, k)
    _f32("e", e, k)
    _f32("weight", weight)
    n = weight.shape[0]
    _C.check_shape("weight", weight, n, k)
    if bias is not None:
        _f32("bias", bias, n)
    if latent_shape is None:
        out, geo, un = torch.empty(rows, n, dtype=torch.float32, device=x.device), (0,) * 7, 0
    else:
        out, geo, un = torch.empty(*latent_shape, dtype=torch.float32, device=x.device), (*latent_shape, *patch), 1
    with torch.cuda.device(x.device):
        _C.call("wanq_head_fwd", _C.ptr(x), _C.ptr(modulation), _C.ptr(e), _C.ptr(weight), _C.ptr(bias), _C.ptr(out), rows, k, n,
                float(eps), un, *[int(g) for g in geo], _C.stream())
    return out
