"""CPU oracle for the DiT block: torch-CPU fp32 restatement of the reference's SIMULATION path -- the FP
WanAttentionBlock (W/wan/modules/model.py:293-370) with every block Linear replaced by qdiff's
QuantizedLinear / ViDiTQuantizedLinear fake-quant forward (Q/base/quant_layer.py:57-74,
Q/viditq/viditq_quant_layer.py:52-73), fp32 everywhere (no autocast), exact softmax attention.

TEST INFRASTRUCTURE ONLY.  Also the `cpu_baseline` of bench.py (kind "port"): it is what the reference's
fake-quant path costs on host cores.

Parity status: the quantizers used here are the golden-pinned ones of oracle/qdiff_ref.py (restated with
torch ops so the block runs in reasonable time; tests/test_oracle_golden.py::test_torch_quantizers_match_numpy
ties the two together).  The surrounding FP block is restated from the reference source; the reference's own
wan package cannot be imported here (needs diffusers/flash_attn/xfuser...), and attention has no fixture in
the reference, so the block level is pinned by construction + the fp32 softmax definition only.
"""
import math

import torch
import torch.nn.functional as F

# Evaluating this oracle ON THE GPU (whole-output block parity at the headline sizes, tests/test_gpu_fullsize.py): the functions
# below take tensors of any device.  Two things differ there and are handled here so that the GPU evaluation is the same
# function as the CPU one: torch's fp32 division on the GPU is not the correctly rounded quotient (tools/probes/vq_fullsize_diag.py:
# 1.8e6 of 5e7 codes differ), so the quantisers' divisions go through fp64 and are rounded once to fp32 (`_div`; on the CPU the
# plain fp32 division IS the IEEE quotient and stays as the reference writes it); and the Linears' fp32 matmuls run in fp64 there
# (the sum of fp32 products, rounded once: closer to the exact value than any fp32 summation order, which is all the CPU result is).


def _div(a, b):
    if a.is_cuda:
        b = torch.as_tensor(b, device=a.device)
        return (a.double() / b.double()).float()
    return a / b


def _linear(x, w, b):
    if x.is_cuda:
        return F.linear(x.double(), w.double(), None if b is None else b.double()).float()
    return F.linear(x, w, b)


# ---------------------------------------------------------------- quantizers (torch restatement of qdiff_ref)
def dyn_fake_quant(x, n_bits=8):
    """DynamicQuantizer.forward, sym (Q/base/base_quantizer.py:116-128,154-161).  x: [T, C] fp32."""
    n = 2 ** (n_bits - 1) - 1
    delta = _div(x.abs().amax(dim=1, keepdim=True), float(n))
    delta = torch.where(delta < 1e-6, torch.full_like(delta, 1e-6), delta)
    return torch.clamp(torch.round(_div(x, delta)), -n - 1, n) * delta


def dyn_fake_quant_asym(x, n_bits=8, eps=1e-8):
    """DynamicQuantizer.forward, asymmetric branch (Q/base/base_quantizer.py:130-149,154-161).  x: [T, C] fp32."""
    n = 2 ** n_bits
    hi = x.amax(dim=1, keepdim=True).clamp_min(0.0)
    lo = x.amin(dim=1, keepdim=True).clamp_max(0.0)
    delta = _div(hi - lo, float(n - 1))
    delta = torch.where(delta < eps, torch.full_like(delta, eps), delta)
    zp = torch.round(_div(lo, delta)) + n / 2
    return (torch.clamp(torch.round(_div(x, delta)) - zp, -n - 1, n) + zp) * delta


def static_params(w, n_bits=8, sym=False):
    """StaticQuantizer.init_quant_params (Q/base/base_quantizer.py:70-99)."""
    if sym:
        return _div(w.abs().amax(dim=1), float(2 ** (n_bits - 1) - 1)).unsqueeze(1), torch.zeros(w.shape[0], 1, device=w.device)
    n_levels = 2 ** n_bits
    hi = w.amax(dim=1).clamp_min(0.0)
    lo = w.amin(dim=1).clamp_max(0.0)
    delta = _div(hi - lo, float(n_levels - 1))
    zp = torch.round(_div(lo, delta)) + n_levels / 2
    return delta.unsqueeze(1), zp.unsqueeze(1)


def static_fake_quant(w, n_bits=8, sym=False):
    """StaticQuantizer.forward (Q/base/base_quantizer.py:56-68)."""
    delta, zp = static_params(w, n_bits, sym)
    n = (2 ** (n_bits - 1) - 1) if sym else 2 ** n_bits
    q = torch.clamp(torch.round(_div(w, delta)) - zp, -n - 1, n)
    return (q + zp) * delta


def hadamard_rotation(signs, strict=True):
    """x -> x.double() @ R for R = random_hadamard_matrix with the sign draw `signs` (Q/quarot/quarot_utils.py:186-192), evaluated
    WITHOUT the dense matrix: row i of R is s_i * hadU(e_i), so x @ R == hadU(x * s) (qdiff_ref.hadamard_from_signs); hadU is
    qdiff_ref.matmul_hadU's restatement of quarot_utils.matmul_hadU (:158-179) in float64, on the tensor's device.  For widths where
    the dense fp64 matrix is impractical (13824 x 13824 = 1.5 GB).  strict=False: the repo-defined K for 13824 (qdiff_ref.had_k)."""
    from oracle import qdiff_ref as qr

    s64 = torch.as_tensor(signs, dtype=torch.float64)

    def apply(x):
        n = x.shape[-1]
        hadK, K = qr.had_k(n, strict)
        m = n // K
        v = (x.double() * s64.to(x.device)).reshape(-1, K, m)
        h = 1
        while h < m:
            v = v.reshape(-1, K, m // (2 * h), 2, h)
            a, b = v[..., 0, :], v[..., 1, :]
            v = torch.stack([a + b, a - b], dim=-2).reshape(-1, K, m)
            h *= 2
        if K > 1:
            v = torch.matmul(torch.from_numpy(hadK.astype("float64")).to(x.device), v)
        import numpy as np
        return v.reshape(x.shape) / float(np.sqrt(np.float32(n)))  # fp32 sqrt, as the reference

    return apply


def _rotate(x, rotation):
    """x.double() @ R for a dense fp64 R, or the same product through hadamard_rotation's callable."""
    return rotation(x) if callable(rotation) else x.double() @ rotation


class FakeQuantLinear:
    """QuantizedLinear / ViDiTQuantizedLinear forward on a 2-D input.  `rotation`: the fp64 matrix R, or hadamard_rotation(signs)."""

    def __init__(self, weight, bias, w_bits=8, a_bits=8, w_sym=False, channel_mask=None, rotation=None, a_sym=True):
        self.a_sym = a_sym
        self.bias = None if bias is None else bias.float()
        self.mask, self.R = channel_mask, rotation
        w = weight.float()
        if channel_mask is None:
            self.weight = static_fake_quant(w, w_bits, w_sym)  # quant_layer.py:38-39
        else:  # viditq_quant_layer.py:40-50: quantise, rotate, quantise again
            w1 = static_fake_quant(_div(w, channel_mask.reshape(1, -1)), w_bits, w_sym)
            self.weight = static_fake_quant(_rotate(w1, rotation).float(), w_bits, w_sym)
        self.a_bits = a_bits

    def __call__(self, x):
        if self.mask is not None:  # viditq_quant_layer.py:62-63
            x = _rotate(x * self.mask.reshape(1, -1), self.R).float()
        return _linear(dyn_fake_quant(x, self.a_bits) if self.a_sym else dyn_fake_quant_asym(x, self.a_bits), self.weight, self.bias)


class FpLinear:
    def __init__(self, weight, bias):
        self.weight, self.bias = weight.float(), None if bias is None else bias.float()

    def __call__(self, x):
        return _linear(x, self.weight, self.bias)


# ---------------------------------------------------------------- FP pieces of the block
def layer_norm(x, eps, weight=None, bias=None):
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)  # WanLayerNorm (model.py:92-102)


def rms_norm(x, weight, eps):
    return x * torch.rsqrt(x.pow(2).mean(dim=-1, keepdim=True) + eps) * weight  # WanRMSNorm (model.py:73-89)


def rope_freqs(head_dim, max_len=1024, theta=10000):
    """WanModel.freqs (model.py:523-529) + rope_params (:31-40)."""
    def rp(dim):
        inv = 1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64) / dim)
        ang = torch.outer(torch.arange(max_len, dtype=torch.float64), inv)
        return torch.polar(torch.ones_like(ang), ang)

    d = head_dim
    return torch.cat([rp(d - 4 * (d // 6)), rp(2 * (d // 6)), rp(2 * (d // 6))], dim=1)


def rope_apply(x, grid, freqs):
    """rope_apply (model.py:43-70) for one sample; x [L, n, d] fp32, float64 complex rotation."""
    f, h, w = grid
    n_tok, c = f * h * w, x.shape[-1] // 2
    parts = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    fi = torch.cat([parts[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), parts[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                    parts[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(n_tok, 1, c)
    xc = torch.view_as_complex(x[:n_tok].double().reshape(n_tok, x.shape[1], c, 2))
    return torch.cat([torch.view_as_real(xc * fi).flatten(2), x[n_tok:].double()]).float()


def attention(q, k, v, k_len=None):
    """softmax(q k^T / sqrt(d)) v per head, fp32 (what flash_attention computes, attention.py:24-130).
    q [Lq, n, d], k/v [Lk, n, d]; keys at or beyond k_len are masked (k_lens, attention.py:78-80)."""
    if k_len is not None:
        k, v = k[:k_len], v[:k_len]
    Lq, n, _ = q.shape
    step = max(1, (1 << 29) // max(1, n * k.shape[0]))  # query chunks of <= 2 GiB of scores: rows are independent, same result
    if Lq > step:
        return torch.cat([attention(q[i:i + step], k, v) for i in range(0, Lq, step)])
    s = torch.einsum("qnd,knd->nqk", q, k) / math.sqrt(q.shape[-1])
    return torch.einsum("nqk,knd->qnd", torch.softmax(s, dim=-1), v)


def qk_fake_quant(x, n_bits=8):
    """The reference's q / k quantisation for quantized attention: DynamicQuantizer over head_dim for every (token, head) --
    `self.q_quantizer(q.reshape([-1, N_dim]))` (W/models/quant_opensora.py:431-436; quantizers built from
    quant_config.attn.qk, Q/base/quant_attn.py:168-174).  x [L, n, d] fp32 -> fake-quantised fp32."""
    L, n, d = x.shape
    return dyn_fake_quant(x.reshape(L * n, d), n_bits).reshape(L, n, d)


def v_fake_quant(v, n_bits=8):
    """The reference's v quantisation for quantized attention: DynamicQuantizer over ALL TOKENS for every (head, channel) --
    `self.v_quantizer(v.permute([0,1,3,2]).reshape([-1, N_token]))` (W/models/quant_opensora.py:438-440).
    v [L, n, d] fp32 -> fake-quantised fp32."""
    L, n, d = v.shape
    return dyn_fake_quant(v.reshape(L, n * d).t().contiguous(), n_bits).t().reshape(L, n, d)


def attention_qk_quant(q, k, v, k_len=None, n_bits=8):
    """attention() on fake-quantised q and k (post-RoPE, pre-scale, as the reference places the quantizers)."""
    return attention(qk_fake_quant(q, n_bits), qk_fake_quant(k, n_bits), v, k_len)


def attn_map_fake_quant(p, n_bits=8, sym=False):
    """QuantizedAttentionMapOpenSORA.forward, group 'row' (Q/base/quant_attn.py:166-173): the post-softmax map [n, Lq, Lk] is
    permuted so that every KEY column is one quantisation group (all queries share its parameters) and goes through the
    DynamicQuantizer (Q/base/base_quantizer.py:101-162, restated here with both of its forms)."""
    n, Lq, Lk = p.shape
    x = p.permute(0, 2, 1).reshape(n * Lk, Lq)
    if sym:
        n_levels = 2 ** (n_bits - 1) - 1
        delta = x.abs().max(dim=1)[0] / n_levels
        zp = torch.zeros_like(delta)
        delta[delta < 1e-6] = 1e-6
    else:
        n_levels = 2 ** n_bits
        x_max = x.max(dim=1)[0].clamp(min=0.0)
        x_min = x.min(dim=1)[0].clamp(max=0.0)
        delta = (x_max - x_min) / (n_levels - 1)
        delta[delta < 1e-8] = 1e-8
        zp = torch.round(x_min / delta) + n_levels / 2
    delta, zp = delta[:, None], zp[:, None]
    xq = torch.clamp(torch.round(x / delta) - zp, -n_levels - 1, n_levels)
    return ((xq + zp) * delta).reshape(n, Lk, Lq).permute(0, 2, 1)


def attention_map_quant(q, k, v, k_len=None, n_bits=8, sym=False):
    """attention() with the post-softmax map fake-quantised per key column before `attn @ v` (W/models/quant_opensora.py:459-476;
    the map stays fp32 here, the reference casts it to the model dtype first)."""
    if k_len is not None:
        k, v = k[:k_len], v[:k_len]
    s = torch.einsum("qnd,knd->nqk", q, k) / math.sqrt(q.shape[-1])
    return torch.einsum("nqk,knd->qnd", attn_map_fake_quant(torch.softmax(s, dim=-1), n_bits, sym), v)


class BlockRef:
    """One WanAttentionBlock in simulation mode.  `lin` maps 'self_attn.q' ... 'ffn.2' to callables."""

    def __init__(self, lin, norm_w, modulation, num_heads, eps=1e-6, norm3=None, qk_bits=None, cross_qk_bits=None, v_bits=None,
                 cross_v_bits=None, attn_map=None, cross_attn_map=None):
        self.lin, self.norm_w, self.mod, self.n, self.eps, self.norm3 = lin, norm_w, modulation.float(), num_heads, eps, norm3
        self.qk_bits, self.cross_qk_bits = qk_bits, cross_qk_bits  # None = FP attention (the reference's Wan wiring)
        self.v_bits, self.cross_v_bits = v_bits, cross_v_bits
        self.attn_map, self.cross_attn_map = attn_map, cross_attn_map  # (n_bits, sym) of the attention-map quantiser, or None

    def __call__(self, x, e0, grid, seq_len, context, freqs):
        """x [L, C], e0 [1, 6, C], context [Lc, C] -> x' [L, C]   (model.py:293-370 for B = 1)."""
        L, C = x.shape
        n, d = self.n, C // self.n
        e = (self.mod + e0.float()).chunk(6, dim=1)
        e = [t.reshape(1, C) for t in e]
        h = layer_norm(x, self.eps) * (1 + e[1]) + e[0]
        q = rms_norm(self.lin["self_attn.q"](h), self.norm_w["self_attn.norm_q"], self.eps).view(L, n, d)
        k = rms_norm(self.lin["self_attn.k"](h), self.norm_w["self_attn.norm_k"], self.eps).view(L, n, d)
        v = self.lin["self_attn.v"](h).view(L, n, d)
        q, k = rope_apply(q, grid, freqs), rope_apply(k, grid, freqs)
        if self.qk_bits:
            q, k = qk_fake_quant(q, self.qk_bits), qk_fake_quant(k, self.qk_bits)
        if self.v_bits:
            v = torch.cat([v_fake_quant(v[:seq_len], self.v_bits), v[seq_len:]])
        o = (attention_map_quant(q, k, v, seq_len, *self.attn_map) if self.attn_map else attention(q, k, v, seq_len)).reshape(L, C)
        x = x + self.lin["self_attn.o"](o) * e[2]
        h = layer_norm(x, self.eps, *(self.norm3 or (None, None)))
        q = rms_norm(self.lin["cross_attn.q"](h), self.norm_w["cross_attn.norm_q"], self.eps).view(L, n, d)
        k = rms_norm(self.lin["cross_attn.k"](context), self.norm_w["cross_attn.norm_k"], self.eps).view(-1, n, d)
        v = self.lin["cross_attn.v"](context).view(-1, n, d)
        if self.cross_qk_bits:
            q, k = qk_fake_quant(q, self.cross_qk_bits), qk_fake_quant(k, self.cross_qk_bits)
        if self.cross_v_bits:
            v = v_fake_quant(v, self.cross_v_bits)
        o = attention_map_quant(q, k, v, None, *self.cross_attn_map) if self.cross_attn_map else attention(q, k, v)
        x = x + self.lin["cross_attn.o"](o.reshape(L, C))
        h = layer_norm(x, self.eps) * (1 + e[4]) + e[3]
        y = self.lin["ffn.2"](F.gelu(self.lin["ffn.0"](h), approximate="tanh"))
        return x + y * e[5]


    def rows(self, x, e0, grid, seq_len, context, freqs, rows):
        """__call__(...)[rows] without computing the other rows' outputs: only the self-attention keys and values need every
        token (LayerNorm, the per-token dynamic quantisers, the Linears, cross-attention and the FFN are all row-local), so a
        headline-size block (L = 32760) is checked on a sample of rows in seconds.  FP attention only (the Wan wiring);
        tests/test_oracle_golden.py checks it against __call__ at a small size."""
        assert not (self.qk_bits or self.v_bits or self.attn_map or self.cross_qk_bits or self.cross_v_bits or self.cross_attn_map)
        L, C = x.shape
        n, d = self.n, C // self.n
        rows = torch.as_tensor(rows, dtype=torch.long)
        S = len(rows)
        e = [t.reshape(1, C) for t in (self.mod + e0.float()).chunk(6, dim=1)]
        h = layer_norm(x, self.eps) * (1 + e[1]) + e[0]
        k = rope_apply(rms_norm(self.lin["self_attn.k"](h), self.norm_w["self_attn.norm_k"], self.eps).view(L, n, d), grid, freqs)
        v = self.lin["self_attn.v"](h).view(L, n, d)
        q_all = torch.zeros(L, n, d)
        q_all[rows] = rms_norm(self.lin["self_attn.q"](h[rows]), self.norm_w["self_attn.norm_q"], self.eps).view(S, n, d)
        q = rope_apply(q_all, grid, freqs)[rows]  # the rotation depends on the row's position in the grid
        xr = x[rows] + self.lin["self_attn.o"](attention(q, k, v, seq_len).reshape(S, C)) * e[2]
        h = layer_norm(xr, self.eps, *(self.norm3 or (None, None)))
        q = rms_norm(self.lin["cross_attn.q"](h), self.norm_w["cross_attn.norm_q"], self.eps).view(S, n, d)
        k = rms_norm(self.lin["cross_attn.k"](context), self.norm_w["cross_attn.norm_k"], self.eps).view(-1, n, d)
        v = self.lin["cross_attn.v"](context).view(-1, n, d)
        xr = xr + self.lin["cross_attn.o"](attention(q, k, v).reshape(S, C))
        h = layer_norm(xr, self.eps) * (1 + e[4]) + e[3]
        return xr + self.lin["ffn.2"](F.gelu(self.lin["ffn.0"](h), approximate="tanh")) * e[5]


LINEARS = ("self_attn.q", "self_attn.k", "self_attn.v", "self_attn.o", "cross_attn.q", "cross_attn.k", "cross_attn.v",
           "cross_attn.o", "ffn.0", "ffn.2")


def block_from_state(sd, num_heads, eps=1e-6, quant=True, w_bits=8, a_bits=8, vidit=None, qk_bits=None, cross_qk_bits=None,
                     v_bits=None, cross_v_bits=None, attn_map=None, cross_attn_map=None, a_sym=True):
    """Build a BlockRef from a WanAttentionBlock state dict (CPU tensors).
    vidit: optional {linear name: (channel_mask fp32 [K], rotation fp64 [K,K])}."""
    lin = {}
    for name in LINEARS:
        w, b = sd[name + ".weight"], sd.get(name + ".bias")
        if quant:
            cm, R = (vidit or {}).get(name, (None, None))
            lin[name] = FakeQuantLinear(w, b, w_bits, a_bits, False, cm, R, a_sym)
        else:
            lin[name] = FpLinear(w, b)
    norm_w = {k: sd[k + ".weight"].float() for k in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k")}
    norm3 = (sd["norm3.weight"].float(), sd["norm3.bias"].float()) if "norm3.weight" in sd else None
    return BlockRef(lin, norm_w, sd["modulation"], num_heads, eps, norm3, qk_bits, cross_qk_bits, v_bits, cross_v_bits, attn_map, cross_attn_map)
