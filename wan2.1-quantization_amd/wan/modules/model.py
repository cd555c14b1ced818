"""Wan2.1 DiT backbone (FP).  Same parameter names and forward semantics as
ViDiT-Q/examples/Wan2.1/wan/modules/model.py (so Wan2.1 checkpoints and the reference's layer-name regexes
apply unchanged), without diffusers / flash_attn: attention goes through wan.ops.attention, the rotary
table is built once per grid, and the self-attention q path is the correct one (the reference's committed
WanSelfAttention.forward drops norm_q and the head view -- SURVEY D1; semantics follow
wan/distributed/xdit_context_parallel.py:162-170)."""
import contextlib
import json
import math
import os

import torch
import torch.nn as nn

from viditq_extension import fused

from .. import ops

__all__ = ["WanModel", "WanAttentionBlock", "WanSelfAttention", "WanT2VCrossAttention", "WanRMSNorm", "WanLayerNorm",
           "sinusoidal_embedding_1d", "rope_params", "rope_apply"]


def fused_fp(x):
    """The FP blocks' row-wise glue on the GPU (LayerNorm + modulate, RMSNorm + rotary embedding, gate + residual) runs as the library's
    fused kernels -- the ones kernel mode uses, fp32 arithmetic -- instead of eight to twelve torch launches each: the FP pass is what
    fp_generate.py and the calibration passes of get_calib_data_wanx.py run (BASELINE config 3), and at the headline size 40 % of it was
    that glue (profiles/r05_fp_*).  Results differ from the torch expressions by fp32 rounding order, and by the one 16-bit rounding
    torch's RMSNorm makes between the normalisation and the weight under autocast (.type_as(x)).  WANQ_FP_FUSED=0, or a CPU tensor (the
    golden-fixture tests of tests/test_model_golden.py): the torch expressions of the reference."""
    return x.is_cuda and os.environ.get("WANQ_FP_FUSED", "1") != "0"


_ROPE_TABLES = {}


def _rope_table(freqs, grid, device):
    # WanModel.freqs is a fixed function of the head dimension (rope_params); two samples of the table guard a caller's own table
    key = (freqs.shape[1], complex(freqs[1, 0]), complex(freqs[-1, -1]), tuple(grid), str(device))
    if key not in _ROPE_TABLES:
        if len(_ROPE_TABLES) > 8:
            _ROPE_TABLES.clear()
        _ROPE_TABLES[key] = ops.rope_table(freqs, grid, device)
    return _ROPE_TABLES[key]


def sinusoidal_embedding_1d(dim, position):
    """float64 sinusoid, cos half first (reference model.py:18-28)."""
    half = dim // 2
    pos = position.to(torch.float64)
    ang = torch.outer(pos, torch.pow(10000, -torch.arange(half, dtype=torch.float64, device=pos.device) / half))
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def rope_params(max_seq_len, dim, theta=10000):
    """complex128 [max_seq_len, dim/2] (reference model.py:31-40)."""
    inv = 1.0 / torch.pow(theta, torch.arange(0, dim, 2, dtype=torch.float64) / dim)
    ang = torch.outer(torch.arange(max_seq_len, dtype=torch.float64), inv)
    return torch.polar(torch.ones_like(ang), ang)


def rope_apply(x, grid, freqs, offset=0):
    """Reference-semantics rotary embedding in float64 for ONE sample: x [L, n, d]; rows beyond f*h*w are
    passed through (reference model.py:43-70).  Used by the FP torch path; the HIP path uses
    ops.rmsnorm_rope_ with ops.rope_table.  `offset`: x holds rows offset .. offset + L of the sequence (a rank's token shard
    under Ulysses: the reference slices the frequency table per rank, wan/distributed/xdit_context_parallel.py:52-58)."""
    f, h, w = grid
    n_tok = f * h * w
    c = x.shape[-1] // 2
    parts = freqs.split([c - 2 * (c // 3), c // 3, c // 3], dim=1)
    fi = torch.cat([parts[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1), parts[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                    parts[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(n_tok, 1, c)
    real = max(0, min(x.shape[0], n_tok - offset))  # rows of x that are real tokens
    fi = fi[offset:offset + real].to(x.device)
    xc = torch.view_as_complex(x[:real].to(torch.float64).reshape(real, x.shape[1], c, 2))
    out = torch.view_as_real(xc * fi).flatten(2)
    return torch.cat([out, x[real:].to(torch.float64)]).float()


class WanRMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.dim, self.eps = dim, eps
        self.weight = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        xf = x.float()
        return (xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + self.eps)).type_as(x) * self.weight


class WanLayerNorm(nn.LayerNorm):
    def __init__(self, dim, eps=1e-6, elementwise_affine=False):
        super().__init__(dim, elementwise_affine=elementwise_affine, eps=eps)

    def forward(self, x):
        return super().forward(x.float()).type_as(x)


class WanSelfAttention(nn.Module):
    def __init__(self, dim, num_heads, window_size=(-1, -1), qk_norm=True, eps=1e-6):
        super().__init__()
        assert dim % num_heads == 0
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.window_size, self.qk_norm, self.eps = window_size, qk_norm, eps
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()
        self.norm_k = WanRMSNorm(dim, eps=eps) if qk_norm else nn.Identity()

    def forward(self, x, seq_lens, grid_sizes, freqs, sp=None):
        """sp: wan.distributed.parallel.SeqParallel or None.  Under Ulysses x is this rank's token shard [1, L/P, C]: q, k, v are
        projected and rotated locally (the rank's slice of the rotary table), exchanged head-scatter / sequence-gather, the
        attention runs on H/P heads over the whole sequence, and the output comes back (usp_attn_forward,
        W/wan/distributed/xdit_context_parallel.py:149-192)."""
        b, s, n, d = x.shape[0], x.shape[1], self.num_heads, self.head_dim
        par = sp is not None and sp.size > 1
        fuse = fused_fp(x) and self.qk_norm
        if fuse:  # RMSNorm + rotary embedding of q and k: one kernel each, in place on the projections
            q, k, v = self.q(x), self.k(x), self.v(x)
        else:
            q = self.norm_q(self.q(x)).view(b, s, n, d)
            k = self.norm_k(self.k(x)).view(b, s, n, d)
            v = self.v(x).view(b, s, n, d)
        outs = []
        for i in range(b):
            off = sp.rank * s if par else 0
            if fuse:
                tab = _rope_table(freqs, grid_sizes[i], x.device)[off:off + s]  # rows past the grid pass through (no table row)
                qi, ki = q[i].contiguous(), k[i].contiguous()
                if tab.shape[0]:
                    ops.rmsnorm_rope_(qi, self.norm_q.weight.float(), tab, d, eps=self.norm_q.eps)
                    ops.rmsnorm_rope_(ki, self.norm_k.weight.float(), tab, d, eps=self.norm_k.eps)
                else:
                    ops.rmsnorm_rope_(qi, self.norm_q.weight.float(), None, d, eps=self.norm_q.eps)
                    ops.rmsnorm_rope_(ki, self.norm_k.weight.float(), None, d, eps=self.norm_k.eps)
                qi, ki, vi = qi.to(torch.bfloat16), ki.to(torch.bfloat16), v[i].to(torch.bfloat16).contiguous()
            else:
                qi = rope_apply(q[i], grid_sizes[i], freqs, off).to(torch.bfloat16).flatten(1)
                ki = rope_apply(k[i], grid_sizes[i], freqs, off).to(torch.bfloat16).flatten(1)
                vi = v[i].to(torch.bfloat16).flatten(1)
            if par:
                wq, wk, wv = (sp.scatter_heads(t_, async_op=True) for t_ in (qi, ki, vi))
                oi = ops.attention(wq.wait(), wk.wait(), wv.wait(), n // sp.size, int(seq_lens[i]))
                outs.append(sp.gather_heads(oi))
            else:
                outs.append(ops.attention(qi, ki, vi, n, int(seq_lens[i])))
        return self.o(torch.stack(outs))


class WanT2VCrossAttention(WanSelfAttention):
    def forward(self, x, context, context_lens):
        b, n = x.shape[0], self.num_heads
        if fused_fp(x) and self.qk_norm:
            q, k = self.q(x).contiguous(), self.k(context).contiguous()
            for i in range(b):
                ops.rmsnorm_rope_(q[i], self.norm_q.weight.float(), None, self.head_dim, eps=self.norm_q.eps)
                ops.rmsnorm_rope_(k[i], self.norm_k.weight.float(), None, self.head_dim, eps=self.norm_k.eps)
        else:
            q = self.norm_q(self.q(x))
            k = self.norm_k(self.k(context))
        v = self.v(context)
        outs = [ops.attention(q[i].to(torch.bfloat16), k[i].to(torch.bfloat16), v[i].to(torch.bfloat16), n,
                              None if context_lens is None else int(context_lens[i])) for i in range(b)]
        return self.o(torch.stack(outs))


class WanAttentionBlock(nn.Module):
    def __init__(self, cross_attn_type, dim, ffn_dim, num_heads, window_size=(-1, -1), qk_norm=True,
                 cross_attn_norm=False, eps=1e-6):
        super().__init__()
        assert cross_attn_type == "t2v_cross_attn", "only the T2V backbone is in scope"
        self.dim, self.ffn_dim, self.num_heads, self.eps = dim, ffn_dim, num_heads, eps
        self.norm1 = WanLayerNorm(dim, eps)
        self.self_attn = WanSelfAttention(dim, num_heads, window_size, qk_norm, eps)
        self.norm3 = WanLayerNorm(dim, eps, elementwise_affine=True) if cross_attn_norm else nn.Identity()
        self.cross_attn = WanT2VCrossAttention(dim, num_heads, (-1, -1), qk_norm, eps)
        self.norm2 = WanLayerNorm(dim, eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)

    def forward(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens, sp=None):
        """x [B, L, C] fp32 residual stream (a token shard under Ulysses: everything but the self-attention is token-local),
        e [B, 6, C] fp32 (reference model.py:293-370)."""
        if fused_fp(x):
            return self._forward_fused(x, e, seq_lens, grid_sizes, freqs, context, context_lens, sp)
        with torch.autocast("cuda", enabled=False):
            e = (self.modulation.float() + e.float()).chunk(6, dim=1)
        y = self.self_attn(self.norm1(x).float() * (1 + e[1]) + e[0], seq_lens, grid_sizes, freqs, sp)
        x = x + y.float() * e[2]
        x = x + self.cross_attn(self.norm3(x), context, context_lens).float()
        y = self.ffn(self.norm2(x).float() * (1 + e[4]) + e[3])
        return x + y.float() * e[5]

    def _forward_fused(self, x, e, seq_lens, grid_sizes, freqs, context, context_lens, sp):
        """The same block with LayerNorm + modulate and gate + residual as one kernel each (fp32 in, fp32 out: the Linears' forward
        hooks of the calibration pass see fp32 inputs, as in the torch form)."""
        B, L, C = x.shape
        with torch.autocast("cuda", enabled=False):
            e = (self.modulation.float() + e.float()).contiguous()  # [B, 6, C]
        x2 = x.float().reshape(B * L, C)
        if not x2.is_contiguous():
            x2 = x2.contiguous()

        def ln_mod(src, shift, scale):
            h = torch.empty_like(src)
            fused.layernorm_nobias_t2i_fuse(h, src, None, e[:, shift], e[:, scale], self.eps)
            return h.view(B, L, C)

        def add(src, y, gate):  # src + y * gate
            y2 = y.reshape(B * L, C)
            return fused.gate_residual_fuse(y2 if y2.is_contiguous() else y2.contiguous(), gate, src, out_dtype=torch.float32)

        y = self.self_attn(ln_mod(x2, 0, 1), seq_lens, grid_sizes, freqs, sp)
        x2 = add(x2, y, e[:, 2])
        ones = e.new_ones(1, C).expand(B, C)
        x2 = add(x2, self.cross_attn(self.norm3(x2.view(B, L, C)), context, context_lens), ones)
        y = self.ffn(ln_mod(x2, 3, 4))
        return add(x2, y, e[:, 5]).view(B, L, C)


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps=1e-6):
        super().__init__()
        self.dim, self.out_dim, self.patch_size, self.eps = dim, out_dim, patch_size, eps
        self.norm = WanLayerNorm(dim, eps)
        self.head = nn.Linear(dim, math.prod(patch_size) * out_dim)
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)

    def forward(self, x, e):
        with torch.autocast("cuda", enabled=False):
            e = (self.modulation.float() + e.float().unsqueeze(1)).chunk(2, dim=1)
            return self.head(self.norm(x.float()) * (1 + e[1]) + e[0])


class WanModel(nn.Module):
    """T2V diffusion backbone.  forward(x: list[C,F,H,W], t: [B], context: list[L,C], seq_len) -> list."""

    def __init__(self, model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, dim=2048, ffn_dim=8192,
                 freq_dim=256, text_dim=4096, out_dim=16, num_heads=16, num_layers=32, window_size=(-1, -1),
                 qk_norm=True, cross_attn_norm=True, eps=1e-6, _skip_init=False, _device=None):
        """_skip_init / _device: build the module tree WITHOUT drawing initial values, parameters left uninitialised on `_device` --
        for callers that load a full state dict next (QuantWanModel.from_float / from_pretrained: drawing 1.4e9 -- or 1.4e10 --
        uniform numbers on the host only to overwrite them was half of ptq_wanx.py's wall time)."""
        super().__init__()
        assert model_type == "t2v", "only the T2V backbone is in scope (SURVEY section 2.1)"
        self.config = dict(model_type=model_type, patch_size=tuple(patch_size), text_len=text_len, in_dim=in_dim, dim=dim,
                           ffn_dim=ffn_dim, freq_dim=freq_dim, text_dim=text_dim, out_dim=out_dim, num_heads=num_heads,
                           num_layers=num_layers, window_size=tuple(window_size), qk_norm=qk_norm,
                           cross_attn_norm=cross_attn_norm, eps=eps)
        for k, v in self.config.items():
            setattr(self, k, v)
        with torch.device("meta") if _skip_init else contextlib.nullcontext():
            self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
            self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
            self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
            self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
            self.blocks = nn.ModuleList([WanAttentionBlock("t2v_cross_attn", dim, ffn_dim, num_heads, window_size, qk_norm,
                                                           cross_attn_norm, eps) for _ in range(num_layers)])
            self.head = Head(dim, out_dim, self.patch_size, eps)
        d = dim // num_heads
        assert dim % num_heads == 0 and d % 2 == 0
        self.freqs = torch.cat([rope_params(1024, d - 4 * (d // 6)), rope_params(1024, 2 * (d // 6)),
                                rope_params(1024, 2 * (d // 6))], dim=1)
        if _skip_init:
            self.to_empty(device=_device or "cpu")
        else:
            self.init_weights()

    def _patch_embed(self, u):
        """patch_embedding(u) for kernel == stride (model.py:580-582 of the reference calls the Conv3d): the convolution is a
        Linear over non-overlapping patches, evaluated as such -- one GEMM instead of MIOpen's solver search, which on this
        stack lands on a 3-ms naive kernel per call and may pick different solvers in different processes (the two-rank
        rehearsal caught ranks disagreeing at 5e-4).  u [C, F, H, W] -> ([L, dim] tokens in (f, h, w) order, grid)."""
        c, f, h, w = u.shape
        pt, ph, pw = self.patch_size
        grid = (f // pt, h // ph, w // pw)
        patches = u.view(c, grid[0], pt, grid[1], ph, grid[2], pw).permute(1, 3, 5, 0, 2, 4, 6).reshape(grid[0] * grid[1] * grid[2], -1)
        return torch.nn.functional.linear(patches, self.patch_embedding.weight.flatten(1), self.patch_embedding.bias), grid

    # ---- pieces shared with the kernel-mode model (wan/quant_wanx_hip.py)
    def embed(self, x, t, context, seq_len):
        dev = self.patch_embedding.weight.device
        x = [self._patch_embed(u.to(self.patch_embedding.weight.dtype)) for u in x]
        grid_sizes = [g for _, g in x]
        x = [u.unsqueeze(0) for u, _ in x]
        seq_lens = [u.size(1) for u in x]
        assert max(seq_lens) <= seq_len
        x = torch.cat([torch.cat([u, u.new_zeros(1, seq_len - u.size(1), u.size(2))], dim=1) for u in x])
        with torch.autocast("cuda", enabled=False):
            e = self.time_embedding(sinusoidal_embedding_1d(self.freq_dim, t.to(dev)).float())
            e0 = self.time_projection(e).unflatten(1, (6, self.dim))
        context = self.text_embedding(torch.stack([
            torch.cat([u, u.new_zeros(self.text_len - u.size(0), u.size(1))]) for u in context]).to(dev))
        return x, e, e0, context, seq_lens, grid_sizes

    def unpatchify(self, x, grid_sizes):
        c, out = self.out_dim, []
        for u, g in zip(x, grid_sizes):
            u = u[: math.prod(g)].view(*g, *self.patch_size, c)
            u = torch.einsum("fhwpqrc->cfphqwr", u)
            out.append(u.reshape(c, *[i * j for i, j in zip(g, self.patch_size)]))
        return out

    def forward(self, x, t, context, seq_len, sp=None):
        """sp: SeqParallel or None.  Ulysses as the reference patches it onto the FP model (usp_dit_forward,
        W/wan/distributed/xdit_context_parallel.py:66-146; W/wan/text2video.py:89-100): the token sequence (padded to a multiple
        of P by seq_len) is cut into P contiguous shards after the embedding, every block runs on its shard, the head output is
        all-gathered in rank order and unpatchified on every rank."""
        x, e, e0, context, seq_lens, grid_sizes = self.embed(x, t, context, seq_len)
        x = x.float()
        par = sp is not None and sp.size > 1
        if par:
            assert x.shape[0] == 1 and x.shape[1] % sp.size == 0, "Ulysses: batch 1, seq_len a multiple of the degree"
            x = sp.shard_rows(x[0]).unsqueeze(0).contiguous()
        for block in self.blocks:
            x = block(x, e0, seq_lens, grid_sizes, self.freqs, context, None, sp if par else None)
        x = self.head(x, e)
        if par:
            x = sp.all_gather_rows(x[0].contiguous()).unsqueeze(0)
        return [u.float() for u in self.unpatchify(x, grid_sizes)]

    def init_weights(self):
        """Xavier-uniform Linear / zero bias, N(0, .02) for the text/time MLPs (reference model.py:658-680);
        the head is zero there -- callers that need a non-trivial synthetic model re-draw it."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
        nn.init.xavier_uniform_(self.patch_embedding.weight.flatten(1))
        for seq in (self.text_embedding, self.time_embedding):
            for m in seq.modules():
                if isinstance(m, nn.Linear):
                    nn.init.normal_(m.weight, std=0.02)
        nn.init.zeros_(self.head.head.weight)

    @classmethod
    def from_pretrained(cls, checkpoint_dir, **overrides):
        """Load a Wan2.1 checkpoint directory: config.json + *.safetensors (diffusers layout)."""
        from safetensors.torch import load_file

        cfg = json.load(open(os.path.join(checkpoint_dir, "config.json")))
        keys = ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim",
                "num_heads", "num_layers", "window_size", "qk_norm", "cross_attn_norm", "eps")
        kw = {k: cfg[k] for k in keys if k in cfg}
        kw.update(overrides)
        model = cls(_skip_init=True, **kw)  # every parameter comes from the checkpoint (strict load below)
        sd = {}
        for f in sorted(os.listdir(checkpoint_dir)):
            if f.endswith(".safetensors"):
                sd.update(load_file(os.path.join(checkpoint_dir, f)))
        model.load_state_dict(sd, strict=True)
        return model
