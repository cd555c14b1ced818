"""Kernel-mode ("hardware") Wan DiT block on MI355X: every Linear of the block runs as an int8-MFMA GEMM with
its producer (LayerNorm+modulate+quant, attention-output quant, GELU+quant) and consumer (gate*y+residual)
fused around it.

Counterpart of ViDiT-Q/examples/Wan2.1/wan/quant_wanx_cuda.py (WanAttentionBlockWithCudaKernel :112-310,
WanSelfAttentionWithCudaKernel :331-474, WANT2VCrossAttentionWithCudaKernel :477-517, the commented FFN path
:520-564), with these deliberate differences:
  * all ten Linears of a block are quantized (the reference wires q/k/v only: use_kernel=[True,False,False]);
  * the residual stream stays fp32 and per-token scales fp32 (simulation-path semantics; the reference casts
    to fp16);
  * no host padding to 128 rows (ragged tile in-kernel), no per-call clones of the scale buffers, no
    torch.cuda.synchronize() inside the block, everything on the current stream (D8);
  * RMSNorm + RoPE is one fp32 kernel instead of float64 complex math.
"""
import torch
import torch.nn as nn

from viditq_extension import fused, qgemm

from . import ops
from .modules.model import WanModel


class HipLinearW8A8(nn.Module):
    """Integer weight + per-output-channel fp32 (delta, zero_point) + fp32 bias.
    weight_dequant = (code + zero_point) * delta  (StaticQuantizer.forward, qdiff/base/base_quantizer.py:56-59).
    w_bits 8: `weight` int8 [N, K].  w_bits 4: `weight` uint8 [N, K/2], the codes + 8 as nibbles in the library's packed layout
    (include/wanq_hip.h); they stay packed in HBM and are expanded in registers inside the GEMM (wanq_gemm_w4a8), the +8 being
    folded into the zero point the epilogue uses (`zp_gemm` = zero_point - 8).
    Optional ViDiT activation transform: `act_premul` fp32 [K] (= channel_mask * rotation signs) and `rot`
    (had_k, hadk) -- the producer kernel applies hadU(x * premul) before quantising this layer's input.
    `act_quantizer`: None for the activation quantiser the fused producers implement (8-bit symmetric per token, eps 1e-6: every Wan
    configuration and the reference's kernels, fused.cu:330-370); otherwise the layer's own qdiff quantiser object (asymmetric,
    below 8 bits, or the mixed-precision class), and the block takes the UNFUSED path for this layer: fp32 activation -> that
    quantiser's HIP kernels -> int8 GEMM -> the zero point's rank-one term (WanAttentionBlockWithHipKernel._linear_any_act)."""
    quantized = True
    act_quantizer = None

    def __init__(self, in_features, out_features, bias=True, sym=False, w_bits=8):
        super().__init__()
        assert w_bits in (4, 8), "integer storage exists for 8-bit and packed 4-bit weights"
        self.in_features, self.out_features, self.w_bits = in_features, out_features, w_bits
        if w_bits == 4:
            assert in_features % 32 == 0, "packed 4-bit weights need in_features % 32 == 0"
            self.register_buffer("weight", torch.empty(out_features, in_features // 2, dtype=torch.uint8))
        else:
            self.register_buffer("weight", torch.empty(out_features, in_features, dtype=torch.int8))
        self.register_buffer("scale_weight", torch.empty(out_features, dtype=torch.float32))
        self.register_buffer("zp_weight", None if sym else torch.empty(out_features, dtype=torch.float32))
        self.register_buffer("zp_gemm", torch.empty(out_features, dtype=torch.float32) if w_bits == 4 else None)
        self.register_buffer("bias", torch.empty(out_features, dtype=torch.float32) if bias else None)
        self.register_buffer("act_premul", None)
        self.rot = None

    @staticmethod
    def quant_params(w, n_bits=8, sym=False):
        """delta, zero_point of StaticQuantizer.init_quant_params (base_quantizer.py:70-90)."""
        from qdiff.base.base_quantizer import static_params

        return static_params(w, n_bits, sym)

    def _set_codes(self, codes, delta, zero_point):
        """codes: int8 [N, K] in [-2^(b-1), 2^(b-1)-1]."""
        if self.w_bits == 4:
            self.weight.copy_(qgemm.pack_w4(codes.clamp(-8, 7).contiguous(), bias=8))
            self.zp_gemm.copy_((zero_point if zero_point is not None else torch.zeros_like(delta)) - 8.0)
        else:
            self.weight.copy_(codes)
        self.scale_weight.copy_(delta)
        if self.zp_weight is not None:
            self.zp_weight.copy_(zero_point)

    def w_rowsum(self):
        """sum_k w_dq[n, k] = (sum_k code + K * zero_point) * delta, fp32 [N]: the channel factor of an asymmetric activation
        quantiser's rank-one term (x_dq = (q + zp_a) * s_a).  From the integer codes the layer holds; cached until they change."""
        r = self.__dict__.get("_w_rowsum")
        if r is None:
            codes = qgemm.unpack_w4(self.weight, bias=8) if self.w_bits == 4 else self.weight
            zp = self.zp_weight if self.zp_weight is not None else torch.zeros_like(self.scale_weight)
            r = (codes.sum(dim=1, dtype=torch.float32) + self.in_features * zp) * self.scale_weight
            self.__dict__["_w_rowsum"] = r
        return r

    def refresh_zp_gemm(self):
        """After scale_weight / zp_weight were loaded from a checkpoint."""
        self.__dict__.pop("_w_rowsum", None)
        if self.w_bits == 4:
            self.zp_gemm.copy_((self.zp_weight if self.zp_weight is not None else torch.zeros_like(self.scale_weight)) - 8.0)

    @classmethod
    def from_float(cls, weight, bias=None, n_bits=8, sym=False, delta=None, zero_point=None):
        weight = weight.detach().float().contiguous()
        m = cls(weight.shape[1], weight.shape[0], bias is not None, sym, n_bits).to(weight.device)
        if delta is None:
            delta, zero_point = cls.quant_params(weight, n_bits, sym)
        delta, zero_point = delta.float().contiguous().view(-1), zero_point.float().contiguous().view(-1)
        codes, _ = fused.weight_quant(weight, delta, zero_point, -128, 127)
        m._set_codes(codes, delta, None if sym else zero_point)
        if bias is not None:
            m.bias.copy_(bias.detach().float())
        return m

    @classmethod
    def from_quantized(cls, ql):
        """From a qdiff QuantizedLinear (any variant): same integer codes, same parameters, same transform."""
        wq = ql.w_quantizer
        m = cls(ql.in_features, ql.out_features, ql.bias is not None, wq.sym, wq.n_bits).to(ql.fp_module.weight.device)
        m._set_codes(ql.int_weight, wq.delta.reshape(-1).float(), None if wq.sym else wq.zero_point.reshape(-1).float())
        if ql.bias is not None:
            m.bias.copy_(ql.bias.detach().float())
        premul, rot = ql._act_transform()
        m.act_premul, m.rot = premul, rot
        aq = ql.a_quantizer
        if not aq.sym or aq.n_bits != 8 or aq._sym_floor != 1e-6:
            m.__dict__["act_quantizer"] = aq  # (not a submodule: the quantiser belongs to the qdiff layer)
        return m

    @property
    def act_key(self):
        """Layers with the same key can share one quantised copy of their input."""
        return None if self.act_premul is None and self.rot is None else id(self)

    def forward(self, a_q, a_scale, a_sum, out_dtype=torch.bfloat16, gelu=False, gate=None, residual=None, out=None):
        if self.w_bits == 4:
            return qgemm.w8a8_linear(a_q, self.weight, a_scale, self.scale_weight, self.bias, a_sum, self.zp_gemm,
                                     out_dtype=out_dtype, gelu=gelu, gate=gate, residual=residual, out=out, w4=True)
        return qgemm.w8a8_linear(a_q, self.weight, a_scale, self.scale_weight, self.bias,
                                 a_sum if self.zp_weight is not None else None, self.zp_weight, out_dtype=out_dtype,
                                 gelu=gelu, gate=gate, residual=residual, out=out)


class HipLinearFp(nn.Module):
    """A Linear the quant config leaves FP (remain_fp_regex): bf16 GEMM through torch (hipBLASLt), as the
    reference's kernel-mode block keeps nn.Linear for those (quant_wanx_cuda.py:360,510)."""
    quantized = False
    act_key = "fp"

    def __init__(self, weight, bias, dtype=torch.bfloat16):
        super().__init__()
        self.register_buffer("weight", weight.detach().to(dtype).contiguous())
        self.register_buffer("bias", None if bias is None else bias.detach().to(dtype).contiguous())


def _to_hip_linear(lin, n_bits, sym, act_dtype):
    from qdiff.base.quant_layer import QuantizedLinear

    if isinstance(lin, QuantizedLinear):
        if lin.quant_mode and lin.w_quantizer is not None and lin.a_quantizer is not None:
            return HipLinearW8A8.from_quantized(lin)
        lin = lin.fp_module
    if n_bits is None:
        return HipLinearFp(lin.weight.data, lin.bias.data if lin.bias is not None else None, act_dtype)
    return HipLinearW8A8.from_float(lin.weight.data, lin.bias.data if lin.bias is not None else None, n_bits, sym)


class _LnSrc:
    """Lazy activation = LayerNorm(x) (* gamma) * (1 + scale) + shift, materialised per consumer format."""

    def __init__(self, blk, x, gamma, shift, scale):
        self.blk, self.x, self.gamma, self.shift, self.scale, self.cache = blk, x, gamma, shift, scale, {}

    def int8(self, lin):
        key = lin.act_key
        if key not in self.cache:
            x, rows, C = self.x, self.x.shape[0], self.x.shape[1]
            q = torch.empty(rows, C, dtype=torch.int8, device=x.device)
            qs = torch.empty(2, rows, dtype=torch.float32, device=x.device)
            if key is None:
                fused.layernorm_nobias_t2i_quant_sum_fuse(q, x, self.gamma, self.shift, self.scale, qs[1], qs[0], self.blk.eps)
            else:
                fused.layernorm_rotate_quant(q, x, self.gamma, self.shift, self.scale, lin.act_premul, lin.rot, qs[1], qs[0], self.blk.eps)
            self.cache[key] = (q, qs[0], qs[1])
        return self.cache[key]

    def _same_rotation(self, a, b):
        """Same Hadamard order and the same K' x K' sign matrix (a fixed function of the order: compared once, cached)."""
        key = (id(a), id(b))
        memo = self.blk.__dict__.setdefault("_rot_memo", {})
        if key not in memo:
            (ka, ha), (kb, hb) = a.rot, b.rot
            memo[key] = ka == kb and ((ha is None and hb is None) or
                                      (ha is not None and hb is not None and ha.shape == hb.shape and bool(torch.equal(ha, hb))))
        return memo[key]

    def prefetch(self, lins):
        """Materialise the int8 form for several quantized consumers in ONE pass over x (q / k / v of a block share
        this LayerNorm but carry their own ViDiT mask and rotation signs).  Falls back to the lazy per-consumer path
        whenever the consumers do not all rotate with the same Hadamard parameters."""
        todo = []
        for lin in lins:
            if not getattr(lin, "quantized", False) or lin.act_key is None or lin.act_key in self.cache or lin.rot is None or \
                    lin.act_quantizer is not None:
                continue
            if lin.act_key not in [l.act_key for l in todo]:
                todo.append(lin)
        if len(todo) < 2 or not all(self._same_rotation(todo[0], l) for l in todo[1:]):
            return
        todo = todo[:3]
        x, rows, C = self.x, self.x.shape[0], self.x.shape[1]
        qs = [torch.empty(rows, C, dtype=torch.int8, device=x.device) for _ in todo]
        vec = torch.empty(len(todo), 2, rows, dtype=torch.float32, device=x.device)
        fused.layernorm_rotate_quant_multi(qs, x, self.gamma, self.shift, self.scale, [l.act_premul for l in todo], todo[0].rot,
                                           [vec[i, 1] for i in range(len(todo))], [vec[i, 0] for i in range(len(todo))], self.blk.eps)
        for i, lin in enumerate(todo):
            self.cache[lin.act_key] = (qs[i], vec[i, 0], vec[i, 1])

    def fp(self):
        if "fp" not in self.cache:
            out = torch.empty(self.x.shape, dtype=self.blk.act_dtype, device=self.x.device)
            fused.layernorm_nobias_t2i_fuse(out, self.x, self.gamma, self.shift, self.scale, self.blk.eps)
            self.cache["fp"] = out
        return self.cache["fp"]

    def fp32(self):
        """The normalised, modulated row in fp32 (what a simulation-mode quantiser sees: model.py:327)."""
        if "fp32" not in self.cache:
            out = torch.empty(self.x.shape, dtype=torch.float32, device=self.x.device)
            fused.layernorm_nobias_t2i_fuse(out, self.x, self.gamma, self.shift, self.scale, self.blk.eps)
            self.cache["fp32"] = out
        return self.cache["fp32"]


class _FpSrc:
    """An activation that already exists in bf16 (attention output, FFN hidden, text context).  `gelu`: the tensor is the FFN's
    PRE-activation and the quantiser applies the tanh-GELU itself (gelu_quant_sum -- the reference's own split,
    K/csrc/fused/fused.cu gelu_quant_sum behind a 16-bit GEMM output, W/models/quant_opensora_cuda.py:402); only offered to a
    consumer that quantises without a rotation."""

    def __init__(self, t, fp_dtype=None, gelu=False):
        self.t, self.cache, self.fp_dtype, self.gelu = t, {}, fp_dtype, gelu
        # `derived`: what consumers computed from this source and may reuse while the source lives (the text context is the same
        # tensor in every denoising step: a block's cross-attention k / v of it are kept here, keyed by the block)
        self.derived = None

    def int8(self, lin):
        key = lin.act_key
        if key not in self.cache:
            qs = torch.empty(2, self.t.shape[0], dtype=torch.float32, device=self.t.device)
            if key is None:
                q = (fused.gelu_quant_sum if self.gelu else fused.quant_sum)(self.t, qs[1], qs[0])
            else:
                q = fused.rotate_quant(self.t, lin.act_premul, lin.rot, qs[1], qs[0])
            self.cache[key] = (q, qs[0], qs[1])
        return self.cache[key]

    def fp32(self):
        if "fp32" not in self.cache:
            t = self.t.float()
            self.cache["fp32"] = torch.nn.functional.gelu(t, approximate="tanh") if self.gelu else t
        return self.cache["fp32"]

    def fp(self):
        assert not self.gelu, "a pre-activation source has no floating-point view"
        if self.fp_dtype is not None and self.t.dtype != self.fp_dtype:
            if "fp" not in self.cache:
                self.cache["fp"] = self.t.to(self.fp_dtype)
            return self.cache["fp"]
        return self.t


_N_CU = {}
_FORCE_CHUNK_UNIT = None  # tests: pipeline the exchange even when a tiny model's heads do not fill a GPU round


def _head_chunks(heads, seq_len, device, max_chunks=4):
    """Split a rank's `heads` for the pipelined Ulysses exchange.  The attention kernel launches ceil(L/256) workgroups per
    head, one per CU at a time, so a chunk should fill whole rounds of the GPU: with L = 32760 (128 workgroups per head,
    256 CUs) chunks are multiples of 2 heads -- 6 heads -> (2, 2, 2), never (3, 3), which would run 4 half-empty rounds
    instead of 3 full ones.  Small chunks first and last: those are the exchanges nothing hides."""
    if device not in _N_CU:
        _N_CU[device] = torch.cuda.get_device_properties(device).multi_processor_count if device.type == "cuda" else 256
    per_head = -(-seq_len // 256)
    unit = _FORCE_CHUNK_UNIT or max(1, _N_CU[device] // per_head)   # heads per full round of the GPU (>= 1)
    units, rem = divmod(heads, unit)
    n = min(max_chunks - (1 if rem else 0), units)
    if n <= 0:
        return [(0, heads)]
    sizes = [units // n * unit] * n
    for i in range(units % n):               # spare whole units go to the middle chunks
        sizes[(n // 2 + i) % n] += unit
    if rem:
        sizes.append(rem)                    # a last, smaller chunk: its partial round exists either way
    out, a = [], 0
    for sz in sizes:
        out.append((a, a + sz))
        a += sz
    return out if len(out) > 1 else [(0, heads)]


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.q = self.k = self.v = self.o = None
        self.register_buffer("norm_q_weight", torch.ones(dim))
        self.register_buffer("norm_k_weight", torch.ones(dim))


class WanAttentionBlockWithHipKernel(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, eps=1e-6, act_dtype=torch.bfloat16, attn_qk8=False, cross_attn_qk8=False,
                 attn_v_bits=None, cross_attn_v_bits=None, attn_map=None, cross_attn_map=None):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads, self.head_dim, self.eps = dim, ffn_dim, num_heads, dim // num_heads, eps
        self.act_dtype = act_dtype
        # quant_config.attn.qk / cross_attn.qk (8-bit symmetric): q and k of that attention leave RMSNorm + RoPE as
        # per-(token, head) int8 codes and Q.K^T runs on the int8 matrix cores (Q/base/quant_attn.py:168-174)
        self.attn_qk8, self.cross_attn_qk8 = bool(attn_qk8), bool(cross_attn_qk8)
        # quant_config.attn.v / cross_attn.v: v fake-quantised per (head, channel) over all tokens before the attention
        # (W/models/quant_opensora.py:438-440); P.V itself stays bf16 -- the reference's recipe has no integer P either
        self.attn_v_bits, self.cross_attn_v_bits = attn_v_bits, cross_attn_v_bits
        self.attn_map, self.cross_attn_map = attn_map, cross_attn_map  # (n_bits, sym) of the attention-map quantiser, or None
        self.self_attn, self.cross_attn = _Attn(dim), _Attn(dim)
        self.ffn0 = self.ffn2 = None
        self.register_buffer("modulation", torch.zeros(1, 6, dim))
        self.register_buffer("norm3_weight", torch.ones(dim))
        self.register_buffer("norm3_bias", torch.zeros(dim))
        self.register_buffer("ones_gate", torch.ones(dim))

    @classmethod
    def from_float(cls, blk, n_bits=8, sym=False, act_dtype=torch.bfloat16, attn_qk8=False, cross_attn_qk8=False,
                   attn_v_bits=None, cross_attn_v_bits=None, attn_map=None, cross_attn_map=None):
        """Build from a WanAttentionBlock (wan/modules/model.py) whose Linears are either plain nn.Linear
        (quantized here with plain per-channel W8 when n_bits is given, kept FP when n_bits is None) or qdiff
        QuantizedLinear variants (their codes / parameters / ViDiT transform are taken over as they are)."""
        m = cls(blk.dim, blk.ffn_dim, blk.num_heads, blk.eps, act_dtype, attn_qk8, cross_attn_qk8, attn_v_bits,
                cross_attn_v_bits, attn_map, cross_attn_map).to(blk.modulation.device)
        for name in ("self_attn", "cross_attn"):
            src, dst = getattr(blk, name), getattr(m, name)
            for l in "qkvo":
                setattr(dst, l, _to_hip_linear(getattr(src, l), n_bits, sym, act_dtype))
            dst.norm_q_weight.copy_(src.norm_q.weight.data.float())
            dst.norm_k_weight.copy_(src.norm_k.weight.data.float())
        m.ffn0 = _to_hip_linear(blk.ffn[0], n_bits, sym, act_dtype)
        m.ffn2 = _to_hip_linear(blk.ffn[2], n_bits, sym, act_dtype)
        m.modulation.copy_(blk.modulation.data.float())
        if isinstance(blk.norm3, nn.LayerNorm) and blk.norm3.weight is not None:
            m.norm3_weight.copy_(blk.norm3.weight.data.float())
            m.norm3_bias.copy_(blk.norm3.bias.data.float())
        return m

    @staticmethod
    def _quant(x):
        qs = torch.empty(2, x.shape[0], dtype=torch.float32, device=x.device)
        q = fused.quant_sum(x, qs[1], qs[0])
        return q, qs[0], qs[1]

    def _linear(self, lin, src, out_dtype=None, gelu=False, gate=None, residual=None):
        """y = lin(src) with the producer / epilogue fusion the layer's kind allows.  With `residual` (the fp32
        stream) the result is residual + y*gate written in place."""
        out_dtype = out_dtype or self.act_dtype
        if lin.quantized and lin.act_quantizer is not None:
            return self._linear_any_act(lin, src, out_dtype, gelu, gate, residual)
        if lin.quantized:
            q, s, ssum = src.int8(lin)
            if residual is not None:
                return lin(q, s, ssum, torch.float32, gate=gate, residual=residual, out=residual)
            return lin(q, s, ssum, out_dtype, gelu=gelu)
        y = torch.nn.functional.linear(src.fp(), lin.weight, lin.bias)
        if gelu:
            y = torch.nn.functional.gelu(y, approximate="tanh")
        if residual is not None:
            fused.gate_residual_into_(residual, y, gate.view(1, -1))
            return residual
        return y

    def _linear_any_act(self, lin, src, out_dtype, gelu, gate, residual):
        """A quantised Linear whose activation quantiser is not the fused producers' (asymmetric, below 8 bits, the mixed-precision
        class: Q/base/base_quantizer.py:130-149, mixed_precision_quantizer.py:126-186 -- no Wan configuration, and the reference's
        kernel mode has no such path at all): the layer's own qdiff quantiser on the fp32 activation (its HIP kernels: row statistics
        + quantise, the ViDiT transform written out in fp32 first), the int8 GEMM with its plain fp32 epilogue, the asymmetric zero
        point's rank-one term (zp_a s_a)[token] x rowsum(w_dq)[channel], then GELU / gate + residual unfused."""
        aq = lin.act_quantizer
        q, s, ssum = aq.quantize_int8(src.fp32(), lin.act_premul, lin.rot)
        y = lin(q, s, ssum, torch.float32)
        if not aq.sym:
            y = torch.addcmul(y, (aq.zero_point.reshape(-1).float() * s).unsqueeze(1), lin.w_rowsum().unsqueeze(0))
        if gelu:
            y = torch.nn.functional.gelu(y, approximate="tanh")
        if residual is not None:
            fused.gate_residual_into_(residual, y, gate.view(1, -1))
            return residual
        return y.to(out_dtype)

    @staticmethod
    def _vq(v, n_bits, k_len):
        """attn.v / cross_attn.v of the quant config: v fake-quantised in place, per column over the first k_len rows."""
        if n_bits:
            fused.fake_quant_cols_(v if k_len is None or k_len >= v.shape[0] else v[:k_len], n_bits)
        return v

    def _context_kv(self, ctx):
        """Cross-attention k (after its RMSNorm; Q8Rows under cross_attn.qk) and v (after cross_attn.v's fake-quant) of the text
        context.  Neither depends on the timestep or on the latent (W/wan/modules/model.py:178-200: k = norm_k(k(context)),
        v = v(context)), and the sampling loop hands every step the same context tensor (W/wan/text2video.py:248-269), so a
        source that carries a `derived` dict (QuantWanModel.forward keeps one per live context tensor) gets them computed once
        per block and reused: 2 GEMMs + 1-2 rowwise launches per block and pass leave the step, bit-identical results."""
        memo = ctx.derived
        if memo is not None and id(self) in memo:
            return memo[id(self)]
        ca, d = self.cross_attn, self.head_dim
        k = self._linear(ca.k, ctx)
        v = self._vq(self._linear(ca.v, ctx), self.cross_attn_v_bits, None)
        if self.cross_attn_qk8:
            k = ops.rmsnorm_rope_q8(k, ca.norm_k_weight, None, d, True, eps=self.eps)
        else:
            ops.rmsnorm_rope_(k, ca.norm_k_weight, None, d, eps=self.eps)
        if memo is not None:
            memo[id(self)] = (k, v)
        return k, v

    def _self_attention_qk8_ulysses(self, q, h, rope, seq_len, sp):
        """attn.qk under Ulysses.  The q / k quantiser works on (token, head) rows (Q/base/quant_attn.py:168-174 on the reshape of
        W/models/quant_opensora.py:431-436), and a rank holds whole (token, head) rows on BOTH sides of the head exchange -- so
        the rows are quantised where RMSNorm + RoPE produces them (same kernel, same codes as on one GPU) and the all-to-all moves
        the int8 codes and the two fp32 scale planes: half the xGMI bytes of the bf16 exchange, bit-equal to the single-rank
        result.  Pipelined over head chunks like the bf16 path (chunk 0 of q under the k GEMM, of k under the v GEMM, the rest
        and the way back under the attention of the previous chunk)."""
        sa, d, P = self.self_attn, self.head_dim, sp.size
        lp, C = q.shape
        H = C // d
        chunks = [(a * d, b * d) for a, b in _head_chunks(H // P, lp * P, q.device)]

        def token_major(s8):  # scale planes [2, H, stride] -> [lp, H * 2]: a head's (delta, constant) pair travels with its codes
            return s8.scales[:, :, :lp].permute(2, 1, 0).reshape(lp, 2 * H)

        def send(s8, planes, ch):
            return (sp.scatter_heads(s8.codes, async_op=True, cols=ch),
                    sp.scatter_heads(planes, async_op=True, cols=(2 * ch[0] // d, 2 * ch[1] // d)))

        q8 = ops.rmsnorm_rope_q8(q, sa.norm_q_weight, rope, d, False, eps=self.eps)
        qp = token_major(q8)
        pend = [[send(q8, qp, chunks[0])]]
        k8 = ops.rmsnorm_rope_q8(self._linear(sa.k, h), sa.norm_k_weight, rope, d, True, eps=self.eps)
        kp = token_major(k8)
        pend[0].append(send(k8, kp, chunks[0]))
        v = self._linear(sa.v, h)
        pend[0].append(sp.scatter_heads(v, async_op=True, cols=chunks[0]))
        for ch in chunks[1:]:
            pend.append([send(q8, qp, ch), send(k8, kp, ch), sp.scatter_heads(v, async_op=True, cols=ch)])
        o = torch.empty(lp, C, dtype=self.act_dtype, device=q.device)
        back = []
        for (c0, c1), (wq, wk, wv) in zip(chunks, pend):
            qc = ops.Q8Rows.from_exchange(wq[0].wait(), wq[1].wait(), d, False)
            kc = ops.Q8Rows.from_exchange(wk[0].wait(), wk[1].wait(), d, True)
            vc = self._vq(wv.wait(), self.attn_v_bits, seq_len)
            if self.attn_map is not None:
                oc = ops.attention_map_quant(qc, kc, vc, (c1 - c0) // d, self.attn_map[0], self.attn_map[1], seq_len, q_len=seq_len)
            else:
                oc = ops.attention_qk8(qc, kc, vc, (c1 - c0) // d, seq_len)
            back.append(sp.gather_heads(oc, async_op=True, out=o, cols=(c0, c1)))
        for b in back:
            b.wait()
        return o

    def forward(self, x, e0, rope, seq_len, ctx, sp=None, e=None):
        """x: fp32 [L, C] residual stream (this rank's token shard under sequence parallelism), updated IN PLACE.
        e0: fp32 [1, 6, C].  rope: fp32 [pos, d/2, 2] for the local tokens.  seq_len: number of real (unpadded)
        tokens of the WHOLE sequence.  ctx: _FpSrc of the bf16 text context [Lc, C] (shared by all blocks so that
        its plain int8 copy is made once per pass).  sp: wan.distributed.SeqParallel or None.  e: this block's
        `modulation + e0` when the caller has it already (the model adds all blocks' modulations in one launch per pass)."""
        H, d = self.num_heads, self.head_dim
        if e is None:
            e = self.modulation + e0  # [1, 6, C] fp32
        sa, ca = self.self_attn, self.cross_attn

        # ---- self attention: LN*(1+e1)+e0 -> q,k,v -> RMSNorm+RoPE -> attention -> o (+gate, +residual)
        h = _LnSrc(self, x, None, e[:, 0], e[:, 1])
        h.prefetch([sa.q, sa.k, sa.v])  # one pass over x for the three ViDiT-transformed int8 inputs
        q = self._linear(sa.q, h)
        if self.attn_qk8 and sp is not None and sp.size > 1:
            o = self._self_attention_qk8_ulysses(q, h, rope, seq_len, sp)
        elif self.attn_qk8:
            q8 = ops.rmsnorm_rope_q8(q, sa.norm_q_weight, rope, d, False, eps=self.eps)
            k8 = ops.rmsnorm_rope_q8(self._linear(sa.k, h), sa.norm_k_weight, rope, d, True, eps=self.eps)
            v = self._vq(self._linear(sa.v, h), self.attn_v_bits, seq_len)
            if self.attn_map is not None:  # q / k / v quantisers AND the map quantiser: the reference's whole recipe
                o = ops.attention_map_quant(q8, k8, v, H, self.attn_map[0], self.attn_map[1], seq_len, q_len=seq_len)
            else:
                o = ops.attention_qk8(q8, k8, v, H, seq_len)
        elif sp is None or sp.size == 1:
            ops.rmsnorm_rope_(q, sa.norm_q_weight, rope, d, eps=self.eps)
            k = self._linear(sa.k, h)
            ops.rmsnorm_rope_(k, sa.norm_k_weight, rope, d, eps=self.eps)
            v = self._vq(self._linear(sa.v, h), self.attn_v_bits, seq_len)
            if self.attn_map is not None:
                o = ops.attention_map_quant(q, k, v, H, self.attn_map[0], self.attn_map[1], seq_len, q_len=seq_len)
            else:
                o = ops.attention(q, k, v, H, seq_len)
        else:
            # Ulysses, pipelined over head chunks: this rank's H/P heads are split in two; the exchange of chunk 1 (and the
            # way back of chunk 0) flies under the attention of the other chunk, so about half of the all-to-all time of a
            # block hides behind its 2-6 ms of attention.  The collectives run in issue order on the group's own stream.
            # RMSNorm+RoPE writes q and k straight into the send images of the chunks ([P, Lp, w] each): no pack pass.
            lp, C = q.shape
            chunks = [(a * d, b * d) for a, b in _head_chunks(H // sp.size, lp * sp.size, q.device)]
            _, hmap, where = sp.packed_layout(lp, C, d, chunks, q.device)
            qs = ops.rmsnorm_rope_scatter(q, sa.norm_q_weight, rope, d, torch.empty(lp * C, dtype=q.dtype, device=q.device), hmap, eps=self.eps)
            pend = [[sp.scatter_packed(qs, lp, *where[0], async_op=True)]]        # chunk 0 of q flies under the k GEMM
            k = self._linear(sa.k, h)
            ks = ops.rmsnorm_rope_scatter(k, sa.norm_k_weight, rope, d, torch.empty(lp * C, dtype=k.dtype, device=k.device), hmap, eps=self.eps)
            pend[0].append(sp.scatter_packed(ks, lp, *where[0], async_op=True))   # ... of k under the v GEMM
            v = self._linear(sa.v, h)
            pend[0].append(sp.scatter_heads(v, async_op=True, cols=chunks[0]))
            for cw, wh in zip(chunks[1:], where[1:]):
                pend.append([sp.scatter_packed(qs, lp, *wh, async_op=True), sp.scatter_packed(ks, lp, *wh, async_op=True),
                             sp.scatter_heads(v, async_op=True, cols=cw)])
            o = torch.empty_like(q)
            back = []
            for (c0, c1), (wq, wk, wv) in zip(chunks, pend):
                # (v of this rank's heads, all tokens: the per-(head, channel) statistics are local after the exchange)
                # (after the exchange a rank holds ALL tokens of its heads: the attention-map quantiser's per-key-column statistics
                # over all queries are local, like v's per-(head, channel) statistics)
                qc, kc, vc = wq.wait(), wk.wait(), self._vq(wv.wait(), self.attn_v_bits, seq_len)
                if self.attn_map is not None:
                    oc = ops.attention_map_quant(qc, kc, vc, (c1 - c0) // d, self.attn_map[0], self.attn_map[1], seq_len, q_len=seq_len)
                else:
                    oc = ops.attention(qc, kc, vc, (c1 - c0) // d, seq_len)
                back.append(sp.gather_heads(oc, async_op=True, out=o, cols=(c0, c1)))
            for b in back:
                b.wait()
        self._linear(sa.o, _FpSrc(o), gate=e[0, 2].contiguous(), residual=x)

        # ---- cross attention: LN_affine -> q; k,v from the text context
        h = _LnSrc(self, x, self.norm3_weight, self.norm3_bias.view(1, -1), None)
        q = self._linear(ca.q, h)
        k, v = self._context_kv(ctx)
        # cross_attn.attn_map under sequence parallelism: a rank's queries are a token shard here (the cross-attention is not
        # exchanged), but a key column's quantisation step is a maximum over ALL queries.  The queries of all ranks are gathered
        # (rank order == sequence order), every rank evaluates the whole map's statistics on the 512 keys and keeps its own rows:
        # the numerics of N = 1 bit for bit, for one all-gather of q per block -- an optional recipe, not the headline path.
        gather_q = self.cross_attn_map is not None and sp is not None and sp.size > 1
        if gather_q:
            q = sp.all_gather_rows(q)
        if self.cross_attn_qk8:
            q8 = ops.rmsnorm_rope_q8(q, ca.norm_q_weight, None, d, False, eps=self.eps)
            if self.cross_attn_map is not None:
                o = ops.attention_map_quant(q8, k, v, H, self.cross_attn_map[0], self.cross_attn_map[1], q_len=seq_len)
            else:
                o = ops.attention_qk8(q8, k, v, H)
        else:
            ops.rmsnorm_rope_(q, ca.norm_q_weight, None, d, eps=self.eps)
            if self.cross_attn_map is not None:
                o = ops.attention_map_quant(q, k, v, H, self.cross_attn_map[0], self.cross_attn_map[1], q_len=seq_len)
            else:
                o = ops.attention(q, k, v, H)
        if gather_q:
            o = sp.shard_rows(o).contiguous()
        self._linear(ca.o, _FpSrc(o), gate=self.ones_gate, residual=x)

        # ---- FFN: LN*(1+e4)+e3 -> GEMM -> GELU + quantise -> GEMM (+gate, +residual).  The GELU runs in ffn.2's quantiser when
        # that is a plain per-token quantiser (the memory-bound pass has the vector slack: 439 + 218 us against 494 + 199 us with
        # the GELU in the GEMM epilogue, whose store loop is vector-bound; cfg-B, tools/probes/ffn_gelu_placement.py); with a
        # rotation in front of ffn.2, or a floating-point ffn.2, it stays in ffn.0's epilogue.
        h = _LnSrc(self, x, None, e[:, 3], e[:, 4])
        late_gelu = self.ffn0.quantized and self.ffn2.quantized and self.ffn2.act_key is None
        hid = self._linear(self.ffn0, h, gelu=not late_gelu)
        self._linear(self.ffn2, _FpSrc(hid, gelu=late_gelu), gate=e[0, 5].contiguous(), residual=x)
        return x
