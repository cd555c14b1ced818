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
    """int8 weight [N,K] + per-output-channel fp32 (delta, zero_point) + fp32 bias.
    weight_dequant = (code + zero_point) * delta  (StaticQuantizer.forward, qdiff/base/base_quantizer.py:56-59)."""

    def __init__(self, in_features, out_features, bias=True, sym=False):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.register_buffer("weight", torch.empty(out_features, in_features, dtype=torch.int8))
        self.register_buffer("scale_weight", torch.empty(out_features, dtype=torch.float32))
        self.register_buffer("zp_weight", None if sym else torch.empty(out_features, dtype=torch.float32))
        self.register_buffer("bias", torch.empty(out_features, dtype=torch.float32) if bias else None)

    @staticmethod
    def quant_params(w, n_bits=8, sym=False):
        """delta, zero_point of StaticQuantizer.init_quant_params (base_quantizer.py:70-90); the row statistics
        come from the HIP reduction, the [N]-sized arithmetic is plain fp32 (IEEE: bit-identical to CPU torch)."""
        lo, hi, am = fused.row_minmax(w)
        # NB: on the GPU torch evaluates `tensor / python_scalar` as a multiply by the reciprocal (1 ulp off the
        # IEEE quotient now and then); the golden vectors are IEEE divisions, so divide by a TENSOR.
        if sym:
            return am / torch.full_like(am, float(2 ** (n_bits - 1) - 1)), torch.zeros_like(am)
        n_levels = 2 ** n_bits
        hi, lo = hi.clamp_min(0.0), lo.clamp_max(0.0)
        delta = (hi - lo) / torch.full_like(hi, float(n_levels - 1))
        return delta, torch.round(lo / delta) + n_levels / 2

    @classmethod
    def from_float(cls, weight, bias=None, n_bits=8, sym=False, delta=None, zero_point=None):
        assert n_bits == 8, "int8 storage: W4 goes through the packed path"
        weight = weight.detach().float().contiguous()
        m = cls(weight.shape[1], weight.shape[0], bias is not None, sym).to(weight.device)
        if delta is None:
            delta, zero_point = cls.quant_params(weight, n_bits, sym)
        delta, zero_point = delta.float().contiguous().view(-1), zero_point.float().contiguous().view(-1)
        codes, _ = fused.weight_quant(weight, delta, zero_point, -128, 127)
        m.weight.copy_(codes)
        m.scale_weight.copy_(delta)
        if not sym:
            m.zp_weight.copy_(zero_point)
        if bias is not None:
            m.bias.copy_(bias.detach().float())
        return m

    def forward(self, a_q, a_scale, a_sum, out_dtype=torch.bfloat16, gelu=False, gate=None, residual=None, out=None):
        return qgemm.w8a8_linear(a_q, self.weight, a_scale, self.scale_weight, self.bias,
                                 a_sum if self.zp_weight is not None else None, self.zp_weight, out_dtype=out_dtype,
                                 gelu=gelu, gate=gate, residual=residual, out=out)


class _Attn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.q = self.k = self.v = self.o = None
        self.register_buffer("norm_q_weight", torch.ones(dim))
        self.register_buffer("norm_k_weight", torch.ones(dim))


class WanAttentionBlockWithHipKernel(nn.Module):
    def __init__(self, dim, ffn_dim, num_heads, eps=1e-6, act_dtype=torch.bfloat16):
        super().__init__()
        self.dim, self.ffn_dim, self.num_heads, self.head_dim, self.eps = dim, ffn_dim, num_heads, dim // num_heads, eps
        self.act_dtype = act_dtype
        self.self_attn, self.cross_attn = _Attn(dim), _Attn(dim)
        self.ffn0 = self.ffn2 = None
        self.register_buffer("modulation", torch.zeros(1, 6, dim))
        self.register_buffer("norm3_weight", torch.ones(dim))
        self.register_buffer("norm3_bias", torch.zeros(dim))
        self.register_buffer("ones_gate", torch.ones(dim))

    @classmethod
    def from_float(cls, blk, n_bits=8, sym=False, act_dtype=torch.bfloat16):
        """Build from an FP WanAttentionBlock (wan/modules/model.py)."""
        m = cls(blk.dim, blk.ffn_dim, blk.num_heads, blk.eps, act_dtype).to(blk.modulation.device)
        for name in ("self_attn", "cross_attn"):
            src, dst = getattr(blk, name), getattr(m, name)
            for l in "qkvo":
                lin = getattr(src, l)
                setattr(dst, l, HipLinearW8A8.from_float(lin.weight.data, lin.bias.data if lin.bias is not None else None, n_bits, sym))
            dst.norm_q_weight.copy_(src.norm_q.weight.data.float())
            dst.norm_k_weight.copy_(src.norm_k.weight.data.float())
        m.ffn0 = HipLinearW8A8.from_float(blk.ffn[0].weight.data, blk.ffn[0].bias.data, n_bits, sym)
        m.ffn2 = HipLinearW8A8.from_float(blk.ffn[2].weight.data, blk.ffn[2].bias.data, n_bits, sym)
        m.modulation.copy_(blk.modulation.data.float())
        if isinstance(blk.norm3, nn.LayerNorm) and blk.norm3.weight is not None:
            m.norm3_weight.copy_(blk.norm3.weight.data.float())
            m.norm3_bias.copy_(blk.norm3.bias.data.float())
        return m

    # -- producers ---------------------------------------------------------------------------------
    def _ln_quant(self, x, gamma, shift, scale):
        rows = x.shape[0]
        q = torch.empty(rows, self.dim, dtype=torch.int8, device=x.device)
        qs = torch.empty(2, rows, dtype=torch.float32, device=x.device)
        fused.layernorm_nobias_t2i_quant_sum_fuse(q, x, gamma, shift, scale, qs[1], qs[0], self.eps)
        return q, qs[0], qs[1]

    @staticmethod
    def _quant(x):
        qs = torch.empty(2, x.shape[0], dtype=torch.float32, device=x.device)
        q = fused.quant_sum(x, qs[1], qs[0])
        return q, qs[0], qs[1]

    def forward(self, x, e0, rope, seq_len, ctx_q, sp=None):
        """x: fp32 [L, C] residual stream (this rank's token shard under sequence parallelism), updated IN PLACE.
        e0: fp32 [1, 6, C].  rope: fp32 [pos, d/2, 2] for the local tokens.  seq_len: number of real (unpadded)
        tokens of the WHOLE sequence.  ctx_q: (int8 [Lc, C], scale [Lc], sum [Lc]) text context.
        sp: wan.distributed.SeqParallel or None."""
        H, d, dt = self.num_heads, self.head_dim, self.act_dtype
        e = self.modulation + e0  # [1, 6, C] fp32
        sa, ca = self.self_attn, self.cross_attn

        # ---- self attention: LN*(1+e1)+e0 -> int8 -> q,k,v GEMMs -> RMSNorm+RoPE -> attention -> int8 -> o GEMM (+gate, +res)
        hq, hs, hsum = self._ln_quant(x, None, e[:, 0], e[:, 1])
        q = sa.q(hq, hs, hsum, dt)
        ops.rmsnorm_rope_(q, sa.norm_q_weight, rope, d, eps=self.eps)
        if sp is None or sp.size == 1:
            k = sa.k(hq, hs, hsum, dt)
            ops.rmsnorm_rope_(k, sa.norm_k_weight, rope, d, eps=self.eps)
            v = sa.v(hq, hs, hsum, dt)
            o = ops.attention(q, k, v, H, seq_len)
        else:  # Ulysses: the q / k all-to-alls fly under the k / v GEMMs
            wq = sp.scatter_heads(q, async_op=True)
            k = sa.k(hq, hs, hsum, dt)
            ops.rmsnorm_rope_(k, sa.norm_k_weight, rope, d, eps=self.eps)
            wk = sp.scatter_heads(k, async_op=True)
            v = sa.v(hq, hs, hsum, dt)
            wv = sp.scatter_heads(v, async_op=True)
            o = ops.attention(wq.wait(), wk.wait(), wv.wait(), H // sp.size, seq_len)
            o = sp.gather_heads(o)
        oq, os_, osum = self._quant(o)
        sa.o(oq, os_, osum, torch.float32, gate=e[0, 2].contiguous(), residual=x, out=x)

        # ---- cross attention: LN_affine -> int8 -> q GEMM; k,v from the (pre-quantized) text context
        hq, hs, hsum = self._ln_quant(x, self.norm3_weight, self.norm3_bias.view(1, -1), None)
        q = ca.q(hq, hs, hsum, dt)
        ops.rmsnorm_rope_(q, ca.norm_q_weight, None, d, eps=self.eps)
        k = ca.k(*ctx_q, dt)
        ops.rmsnorm_rope_(k, ca.norm_k_weight, None, d, eps=self.eps)
        v = ca.v(*ctx_q, dt)
        o = ops.attention(q, k, v, H)
        oq, os_, osum = self._quant(o)
        ca.o(oq, os_, osum, torch.float32, gate=self.ones_gate, residual=x, out=x)

        # ---- FFN: LN*(1+e4)+e3 -> int8 -> GEMM+GELU -> int8 -> GEMM (+gate, +res)
        hq, hs, hsum = self._ln_quant(x, None, e[:, 3], e[:, 4])
        h = self.ffn0(hq, hs, hsum, dt, gelu=True)
        hq, hs, hsum = self._quant(h)
        self.ffn2(hq, hs, hsum, torch.float32, gate=e[0, 5].contiguous(), residual=x, out=x)
        return x


class QuantWanModelHip(nn.Module):
    """Kernel-mode model: FP embedders / head of a WanModel + WanAttentionBlockWithHipKernel blocks.
    forward has WanModel.forward's signature (batch of ONE sample, as every T2V call site uses it)."""

    def __init__(self, fp_model: WanModel, n_bits=8, sym=False, act_dtype=torch.bfloat16, keep_fp_blocks=False):
        super().__init__()
        self.cfg = dict(fp_model.config)
        self.fp = fp_model  # embedders + head are used as they are (FP, remain_fp_regex: config.yaml:8)
        self.blocks = nn.ModuleList()
        for i, blk in enumerate(fp_model.blocks):
            self.blocks.append(WanAttentionBlockWithHipKernel.from_float(blk, n_bits, sym, act_dtype))
        # the FP blocks are detached from the FP model (its forward is only used for embed/head); they are kept
        # on request as the FP reference for quality metrics
        self.fp_blocks = fp_model.blocks if keep_fp_blocks else None
        fp_model.blocks = nn.ModuleList()
        if not keep_fp_blocks:
            torch.cuda.empty_cache()
        self._rope_cache = {}

    def _rope(self, grid, device):
        if grid not in self._rope_cache:
            self._rope_cache[grid] = ops.rope_table(self.fp.freqs, grid, device)
        return self._rope_cache[grid]

    @torch.no_grad()
    def forward(self, x, t, context, seq_len, sp=None):
        """WanModel.forward signature for ONE sample; with `sp` (SeqParallel) the token sequence is sharded
        over the group as usp_dit_forward does (xdit_context_parallel.py:131-142): seq_len must be a multiple
        of sp.size (WanT2V.generate pads it so, text2video.py:170-172)."""
        assert len(x) == 1, "kernel-mode forward takes one sample (cond and uncond are separate passes)"
        with torch.autocast("cuda", enabled=False):
            h, e, e0, ctx, seq_lens, grids = self.fp.embed(x, t, context, seq_len)
            h = h[0].float()  # [seq_len, C] fp32 residual stream
            rope = self._rope(grids[0], h.device)
            if sp is not None and sp.size > 1:
                assert seq_len % sp.size == 0
                lp = seq_len // sp.size
                h = sp.shard_rows(h)
                rope = rope[sp.rank * lp:(sp.rank + 1) * lp]  # may be shorter than lp on the last rank: pads stay unrotated
            h = h.contiguous()
            cq = WanAttentionBlockWithHipKernel._quant(ctx[0].float().contiguous())
            for blk in self.blocks:
                blk(h, e0.float(), rope, seq_lens[0], cq, sp)
            out = self.fp.head(h.unsqueeze(0), e)
            if sp is not None and sp.size > 1:
                out = sp.all_gather_rows(out[0]).unsqueeze(0)
            return [u.float() for u in self.fp.unpatchify(out, grids)]
