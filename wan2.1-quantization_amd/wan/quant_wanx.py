"""QuantWanModel: WanModel + the quantization API the four entry points call.

Counterpart of ViDiT-Q/examples/Wan2.1/wan/quant_wanx.py:28-228 (same method names):
    quant_layer_refactor / convert_quant, save_quant_param_dict, load_quant_param_dict, set_init_done,
    bitwidth_refactor, quantize_and_save_weight(save_path), hardware_forward_refactor(load_path, seq_len).
Simulation mode = the qdiff drop-in layers inside the unmodified WanAttentionBlock (they already compute in int8);
kernel mode = every block replaced by WanAttentionBlockWithHipKernel (fused producers / epilogues around the GEMMs)."""
import logging
import math

import torch
import torch.nn as nn

from qdiff.base.quant_layer import QuantizedLinear
from qdiff.base.quant_model import QuantModel

from .modules.model import WanModel
from .quant_wanx_hip import WanAttentionBlockWithHipKernel, _FpSrc
from . import ops

logger = logging.getLogger(__name__)


class QuantWanModel(WanModel, QuantModel):
    def __init__(self, quant_config=None, **model_kwargs):
        WanModel.__init__(self, **model_kwargs)
        self.q_cfg = quant_config
        self.quant_param_dict = {}
        self.hip_blocks = None
        self._fsdp = None
        self._rope_cache = {}

    @classmethod
    def from_pretrained(cls, checkpoint_dir, quant_config=None, **overrides):
        fp = WanModel.from_pretrained(checkpoint_dir, **overrides)
        return cls.from_float(fp, quant_config)

    @classmethod
    def from_float(cls, fp_model, quant_config):
        """A quantisable copy of `fp_model` on its device: the module tree is built without drawing initial values (every parameter is
        overwritten by the strict load that follows) and straight on that device (no host round trip of the weights)."""
        dev = next(fp_model.parameters()).device
        m = cls(quant_config, _skip_init=True, _device=dev, **fp_model.config)
        m.load_state_dict(fp_model.state_dict(), strict=True)
        return m

    def convert_quant(self, quant_config=None):
        if quant_config is not None:
            self.q_cfg = quant_config
        self.quant_layer_refactor()

    # ---- kernel mode -----------------------------------------------------------------------------------
    def quantize_and_save_weight(self, save_path=None, reference_format=False):
        """Integer state dict of every quantized Linear (reference quant_wanx.py:137-185): `<name>.weight` int8,
        `<name>.scale_weight`, `<name>.zp_weight`, `<name>.bias`; fp_module / fp_weight entries are dropped; ViDiT /
        SmoothQuant / QuaRot layers also carry `<name>.act_premul` (= channel_mask * rotation signs).

        reference_format=False (default): parameters fp32 -- the precision the simulation path computes with.
        reference_format=True: byte-for-byte what the reference's `int_weight.pt` holds, so that its loader
        (W8A8OF16LinearDynamicInputScale buffers, K/viditq_extension/nn/qlinear.py:23-58) can read it: fp16 `scale_weight`,
        fp16 `zp_weight` (integer-valued: the loader copies it into an int16 buffer), fp16 bias, every other floating tensor
        fp16 (the kernel-mode blocks are `.half()`), `blocks.N.norm{1,2}.weight` = ones, and int8 codes by the reference's
        own half-precision equation (quantize_and_save_weight_, quant_wanx_cuda.py:39-53) on the HIP export kernel.  For a
        plain QuantizedLinear that equation is applied to the FP weight, exactly as the reference does; for a transformed
        layer it is applied to the layer's transformed weight (the reference applies it to the RAW weight with parameters
        fitted on the transformed one -- SURVEY D3 -- and its kernel path has no activation transform to go with it)."""
        from viditq_extension import fused

        sd = {}
        skip = set()
        f16 = torch.float16
        for name, mod in self.named_modules():
            if isinstance(mod, QuantizedLinear) and mod.w_quantizer is not None and mod.quant_mode:
                wq = mod.w_quantizer
                premul, _ = mod._act_transform()
                if reference_format:
                    if wq.n_bits != 8:
                        raise NotImplementedError(f"{name}: the reference's int_weight.pt format is W8 only (n_bits={wq.n_bits})")
                    s16 = wq.delta.reshape(-1).to(f16).contiguous()
                    z16 = (torch.zeros_like(s16) if wq.sym else wq.zero_point.reshape(-1).to(f16)).contiguous()
                    src = mod.fp_module.weight.data if premul is None else mod.weight.data
                    sd[name + ".weight"] = fused.weight_export_f16(src.detach().contiguous(), s16, z16)
                    sd[name + ".scale_weight"] = s16
                    sd[name + ".zp_weight"] = z16
                    if mod.bias is not None:
                        sd[name + ".bias"] = mod.bias.detach().to(f16).clone()
                else:
                    # 8-bit: int8 [N, K]; 4-bit: uint8 [N, K/2], the packed nibbles exactly as the GEMM reads them
                    sd[name + ".weight"] = mod._codes.clone()
                    sd[name + ".scale_weight"] = wq.delta.reshape(-1).float().clone()
                    if not wq.sym:
                        sd[name + ".zp_weight"] = wq.zero_point.reshape(-1).float().clone()
                    if mod.bias is not None:
                        sd[name + ".bias"] = mod.bias.detach().float().clone()
                if premul is not None:
                    sd[name + ".act_premul"] = premul.clone()
                skip.add(name)
        for k, v in self.state_dict().items():
            owner = k.rsplit(".", 1)[0]
            if any(owner == s or owner.startswith(s + ".") for s in skip):
                continue
            v = v.detach().clone()
            sd[k] = v.to(f16) if reference_format and v.is_floating_point() and k.startswith("blocks.") else v
        if reference_format:
            for i in range(len(self.blocks)):  # the fused LayerNorm kernel takes a weight: vanilla LN = ones (quant_wanx.py:173-177)
                sd[f"blocks.{i}.norm1.weight"] = torch.ones(self.dim, dtype=f16)
                sd[f"blocks.{i}.norm2.weight"] = torch.ones(self.dim, dtype=f16)
        if save_path:
            torch.save(sd, save_path)
        return sd

    def hardware_forward_refactor(self, load_path=None, seq_len=None, act_dtype=torch.bfloat16):
        """Switch forward() to kernel mode: every block becomes a WanAttentionBlockWithHipKernel (reference
        quant_wanx.py:188-228).  The blocks are built from the model's current layers (which fixes which Linears are
        quantized and which carry a rotation); with `load_path` the integer checkpoint is then LOADED into them, as the
        reference does with `int_weight.pt`: codes, scales, zero points, biases and activation pre-multipliers of every
        quantized Linear come from the file (either format of quantize_and_save_weight), a quantized Linear without its keys
        or with a shape mismatch is an error, and the number of tensors taken is logged."""
        qk8, vb, amap = {}, {}, {}
        for key in ("attn", "cross_attn"):  # quant_config.attn.qk / cross_attn.qk (Q/base/quant_attn.py:19-29,130-143)
            sub = self.q_cfg.get(key, None) if self.q_cfg is not None else None
            qk = sub.get("qk", None) if sub is not None else None
            if qk is not None:
                if qk.get("n_bits", 8) != 8 or not qk.get("sym", True):
                    raise NotImplementedError(f"{key}.qk: the int8 Q.K^T kernel implements symmetric 8-bit q / k")
                qk8[key] = True
            vq = sub.get("v", None) if sub is not None else None
            if vq is not None:  # v fake-quant per (head, channel) over all tokens (W/models/quant_opensora.py:438-440)
                if not vq.get("sym", True):
                    raise NotImplementedError(f"{key}.v: symmetric v quantisation is implemented (the reference's DynamicQuantizer default)")
                vb[key] = int(vq.get("n_bits", 8))
            am = sub.get("attn_map", None) if sub is not None else None
            if am is not None:  # post-softmax map, one dynamic group per KEY column (Q/base/quant_attn.py:166-173, group 'row')
                if am.get("group", "row") not in ("row", "column"):  # the OpenSORA class calls it 'row', the CogVideoX class 'column': same code
                    raise NotImplementedError(
                        f"{key}.attn_map.group = {am.get('group')!r}: the 'block' mode of the reference is tied to CogVideoX's 13x30x45 grid "
                        "and to per-head reorder tables (Q/base/quant_attn.py:176-236); 'row' is implemented (streamed, csrc/attn_map.hip)")
                amap[key] = (int(am.get("n_bits", 8)), bool(am.get("sym", False)))
        # the fp32 ends of a pass (csrc/embed_head.hip: patch / time / text embeddings, head) read these parameters as fp32, as the
        # reference computes them (amp.autocast(dtype=torch.float32) around the time MLPs and the head, model.py:592-597,396-399):
        # say so here, once, instead of a dtype refusal from inside the first forward
        ends = [("patch_embedding", self.patch_embedding), ("time_embedding.0", self.time_embedding[0]), ("time_embedding.2", self.time_embedding[2]),
                ("time_projection.1", self.time_projection[1]), ("text_embedding.0", self.text_embedding[0]), ("text_embedding.2", self.text_embedding[2]),
                ("head.head", self.head.head)]
        bad = [f"{n}.weight is {m.weight.dtype}" for n, m in ends if m.weight.dtype != torch.float32]
        if bad or self.head.modulation.dtype != torch.float32:
            raise TypeError("kernel mode keeps the embeddings and the head in fp32 (load the checkpoint in fp32, or call .float() on them "
                            "before hardware_forward_refactor): " + ", ".join(bad + ([f"head.modulation is {self.head.modulation.dtype}"]
                                                                                   if self.head.modulation.dtype != torch.float32 else [])))
        self.__dict__.pop("_ctx_cache", None)
        self.hip_blocks = nn.ModuleList([WanAttentionBlockWithHipKernel.from_float(
            b, None, False, act_dtype, attn_qk8=qk8.get("attn", False), cross_attn_qk8=qk8.get("cross_attn", False),
            attn_v_bits=vb.get("attn"), cross_attn_v_bits=vb.get("cross_attn"), attn_map=amap.get("attn"),
            cross_attn_map=amap.get("cross_attn")) for b in self.blocks])
        if load_path:
            sd = torch.load(load_path, map_location="cpu", weights_only=True)
            taken = 0
            for i, hb in enumerate(self.hip_blocks):
                for owner, attr, key in [(hb.self_attn, l, f"self_attn.{l}") for l in "qkvo"] + \
                                        [(hb.cross_attn, l, f"cross_attn.{l}") for l in "qkvo"] + \
                                        [(hb, "ffn0", "ffn.0"), (hb, "ffn2", "ffn.2")]:
                    lin = getattr(owner, attr)
                    if not lin.quantized:
                        continue
                    base = f"blocks.{i}.{key}"
                    for buf, k, required in (("weight", "weight", True), ("scale_weight", "scale_weight", True),
                                             ("zp_weight", "zp_weight", lin.zp_weight is not None),
                                             ("bias", "bias", lin.bias is not None),
                                             ("act_premul", "act_premul", lin.act_premul is not None)):
                        dst = getattr(lin, buf)
                        if f"{base}.{k}" not in sd:
                            if required:
                                raise KeyError(f"{load_path}: missing {base}.{k}")
                            continue
                        src = sd[f"{base}.{k}"]
                        if dst is None:
                            # the reference's format always carries `zp_weight` (quant_wanx_cuda.py:39-53 writes one per layer):
                            # for a symmetric layer it is all zeros and there is nothing to load
                            if buf == "zp_weight" and not bool(src.float().abs().max() > 0):
                                continue
                            raise KeyError(f"{load_path}: {base}.{k} present but the model's layer has no {buf}")
                        if tuple(src.shape) != tuple(dst.shape) or (buf == "weight" and src.dtype != dst.dtype):
                            raise ValueError(f"{load_path}: {base}.{k} is {tuple(src.shape)} {src.dtype}, expected {tuple(dst.shape)} {dst.dtype}")
                        dst.copy_(src.to(dst.dtype))
                        taken += 1
                    lin.refresh_zp_gemm()
            logger.info("loaded %d tensors of the integer checkpoint %s into the kernel-mode blocks", taken, load_path)
        for i, hb in enumerate(self.hip_blocks):
            hb.block_index = i
        self.__dict__.pop("_mod_all", None)
        return self

    def shard_blocks(self, group=None):
        """`--dit_fsdp` (reference wan/distributed/fsdp.py:10-32, text2video.py:106-107): shard the kernel-mode blocks' integer
        weights over the ranks of `group`; forward() then all-gathers one block ahead of the one it is computing."""
        from .distributed.fsdp import ShardedBlocks

        assert self.hip_blocks is not None, "shard_blocks applies to kernel mode (call hardware_forward_refactor first)"
        self.__dict__.pop("_ctx_cache", None)
        self._fsdp = ShardedBlocks(self.hip_blocks, group)
        return self._fsdp

    def software_forward(self):
        self.hip_blocks = None
        return self

    def _rope(self, grid, device):
        if grid not in self._rope_cache:
            self._rope_cache[grid] = ops.rope_table(self.freqs, grid, device)
        return self._rope_cache[grid]

    def _context_source(self, raw, embed):
        """The blocks' view of one text context, with what they derive from it kept across the sampling loop.  Everything a pass
        computes from the context alone -- text_embedding, its plain int8 copy, cross_attn.k (+ RMSNorm) and cross_attn.v -- is
        independent of the timestep and the latent, and text2video.py hands every step the SAME context tensors (cond, uncond), so
        it is kept per live tensor: the entry holds the tensor itself (its storage cannot be recycled under the cache) and its
        in-place version counter; a different tensor, or the same one after an in-place write, recomputes.  `embed` is the callable
        that runs text_embedding (only on a miss).  At most four entries; `context_cache = False` (bench.py --no-context-cache)
        turns it off, anything that rebuilds the kernel-mode blocks (hardware_forward_refactor, shard_blocks) empties it.  Never
        used under graph capture (wan/graph.py turns it off for its warm-up and capture as well): a captured graph must hold the
        text_embedding / cross_attn.k / .v launches itself -- it would otherwise bake in pointers to tensors that only this cache
        owns (freed by the fifth other context or a rebuild of the blocks) and ignore later in-place updates of its static
        context buffers."""
        src = lambda: _FpSrc(embed().float().contiguous(), self.hip_blocks[0].act_dtype)  # noqa: E731
        if not getattr(self, "context_cache", True) or (raw.is_cuda and torch.cuda.is_current_stream_capturing()):
            return src()
        cache = self.__dict__.setdefault("_ctx_cache", [])
        for ent in cache:
            if ent[0] is raw and ent[1] == raw._version:
                return ent[2]
        cq = src()
        cq.derived = {}
        cache.append((raw, raw._version, cq))
        del cache[:-4]
        return cq

    def _modulations(self, device):
        """All kernel-mode blocks' modulation tables as one [blocks, 6, C] tensor (so that `modulation + e0` of reference
        model.py:322-324 is one launch per pass, not one per block); rebuilt when a block's table was written or moved."""
        if torch.cuda.is_current_stream_capturing():
            # under graph capture the table must live in the graph's own memory pool: a tensor this cache owns would be freed by the
            # next rebuild (a modulation write, .to(), hardware_forward_refactor) while the captured graph still points at it
            return torch.cat([hb.modulation.to(device) for hb in self.hip_blocks]).contiguous()
        key = (device, tuple(hb.modulation._version for hb in self.hip_blocks), tuple(hb.modulation.data_ptr() for hb in self.hip_blocks))
        ent = self.__dict__.get("_mod_all")
        if ent is None or ent[0] != key:
            ent = (key, torch.cat([hb.modulation.to(device) for hb in self.hip_blocks]).contiguous())
            self.__dict__["_mod_all"] = ent
        return ent[1]

    def _embed_hip(self, xi, ti, seq_len):
        """The embeddings in front of the blocks (reference model.py:580-597) on csrc/embed_head.hip, all fp32: patch embedding
        gathered straight from the latent (padded to seq_len rows with zeros), sinusoid -> time_embedding -> time_projection.
        -> (h [seq_len, dim], e [dim], e0 [1, 6, dim], token grid)."""
        pe, te, tp = self.patch_embedding, self.time_embedding, self.time_projection
        dev = pe.weight.device
        h, grid = ops.patch_embed(xi.to(dev, torch.float32).contiguous(), pe.weight, pe.bias, out_rows=seq_len)
        assert math.prod(grid) <= seq_len
        e = ops.linear_f32(ops.time_sinusoid(ti.to(dev), self.freq_dim), te[0].weight, te[0].bias, out_act="silu")
        e = ops.linear_f32(e, te[2].weight, te[2].bias)
        e0 = ops.linear_f32(e, tp[1].weight, tp[1].bias, in_act="silu")
        return h, e[0], e0.view(1, 6, self.dim), grid

    def _text_embed_hip(self, ci):
        """text_embedding (reference model.py:600-605): the context padded to text_len rows with zeros BEFORE the MLP."""
        tx = self.text_embedding
        c = ci.to(tx[0].weight.device, torch.float32).contiguous()
        assert c.shape[0] <= self.text_len
        return ops.linear_f32(ops.linear_f32(c, tx[0].weight, tx[0].bias, out_act="gelu_tanh", rows=self.text_len), tx[2].weight, tx[2].bias)

    @torch.no_grad()
    def forward(self, x, t, context, seq_len, sp=None):
        if self.hip_blocks is None:  # simulation mode: the fake-quant Linears are token-local, the FP model's Ulysses path applies
            return WanModel.forward(self, x, t, context, seq_len, sp)
        outs = []
        hd = self.head
        for xi, ci, ti in zip(x, context, t.reshape(-1, 1)):
            with torch.autocast("cuda", enabled=False):
                h, e, e0, grid = self._embed_hip(xi, ti, seq_len)
                L0 = math.prod(grid)
                rope = self._rope(grid, h.device)
                if sp is not None and sp.size > 1:
                    lp = seq_len // sp.size
                    h = sp.shard_rows(h).contiguous()
                    rope = rope[sp.rank * lp:(sp.rank + 1) * lp]
                cq = self._context_source(ci, lambda: self._text_embed_hip(ci))
                e_all = self._modulations(e0.device) + e0  # every block's modulation + e0 (reference model.py:322-324), one launch per pass
                if getattr(self, "_fsdp", None) is not None:
                    self._fsdp.run(lambda blk: blk(h, e0, rope, L0, cq, sp, e=e_all[blk.block_index:blk.block_index + 1]))
                else:
                    for i, blk in enumerate(self.hip_blocks):
                        blk(h, e0, rope, L0, cq, sp, e=e_all[i:i + 1])
                hw, hb, mod = hd.head.weight, hd.head.bias, hd.modulation.view(2, self.dim)
                if sp is not None and sp.size > 1:  # the head is token-local: on the shard, then gathered (xdit_context_parallel.py:138-142)
                    out = sp.all_gather_rows(ops.head(h, mod, e, hw, hb, hd.eps)).unsqueeze(0)
                    outs.append(self.unpatchify(out, [grid])[0].float())
                else:  # LayerNorm + modulation + Linear + unpatchify in one launch
                    latent = (self.out_dim, *[g * p for g, p in zip(grid, self.patch_size)])
                    outs.append(ops.head(h[:L0], mod, e, hw, hb, hd.eps, latent_shape=latent, patch=self.patch_size))
        return outs
