"""QuantWanModel: WanModel + the quantization API the four entry points call.

Counterpart of ViDiT-Q/examples/Wan2.1/wan/quant_wanx.py:28-228 (same method names):
    quant_layer_refactor / convert_quant, save_quant_param_dict, load_quant_param_dict, set_init_done,
    bitwidth_refactor, quantize_and_save_weight(save_path), hardware_forward_refactor(load_path, seq_len).
Simulation mode = the qdiff drop-in layers inside the unmodified WanAttentionBlock (they already compute in int8);
kernel mode = every block replaced by WanAttentionBlockWithHipKernel (fused producers / epilogues around the GEMMs)."""
import logging

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
        self._rope_cache = {}

    @classmethod
    def from_pretrained(cls, checkpoint_dir, quant_config=None, **overrides):
        fp = WanModel.from_pretrained(checkpoint_dir, **overrides)
        m = cls(quant_config, **fp.config)
        m.load_state_dict(fp.state_dict())
        return m

    @classmethod
    def from_float(cls, fp_model, quant_config):
        m = cls(quant_config, **fp_model.config)
        m.load_state_dict(fp_model.state_dict())
        return m.to(next(fp_model.parameters()).device)

    def convert_quant(self, quant_config=None):
        if quant_config is not None:
            self.q_cfg = quant_config
        self.quant_layer_refactor()

    # ---- kernel mode -----------------------------------------------------------------------------------
    def quantize_and_save_weight(self, save_path=None):
        """Integer state dict of every quantized Linear (reference quant_wanx.py:137-185): `<name>.weight` int8,
        `<name>.scale_weight`, `<name>.zp_weight`, `<name>.bias`; fp_module / fp_weight entries are dropped.
        Parameters are stored fp32 (the reference casts delta / zero_point to fp16, losing the precision the
        simulation path has); ViDiT layers also carry `<name>.act_premul` (= channel_mask * rotation signs)."""
        sd = {}
        skip = set()
        for name, mod in self.named_modules():
            if isinstance(mod, QuantizedLinear) and mod.w_quantizer is not None and mod.quant_mode:
                wq = mod.w_quantizer
                sd[name + ".weight"] = mod.int_weight.clone()
                sd[name + ".scale_weight"] = wq.delta.reshape(-1).float().clone()
                if not wq.sym:
                    sd[name + ".zp_weight"] = wq.zero_point.reshape(-1).float().clone()
                if mod.bias is not None:
                    sd[name + ".bias"] = mod.bias.detach().float().clone()
                premul, _ = mod._act_transform()
                if premul is not None:
                    sd[name + ".act_premul"] = premul.clone()
                skip.add(name)
        for k, v in self.state_dict().items():
            owner = k.rsplit(".", 1)[0]
            if any(owner == s or owner.startswith(s + ".") for s in skip):
                continue
            sd[k] = v.detach().clone()
        if save_path:
            torch.save(sd, save_path)
        return sd

    def hardware_forward_refactor(self, load_path=None, seq_len=None, act_dtype=torch.bfloat16):
        """Switch forward() to kernel mode: every block becomes a WanAttentionBlockWithHipKernel built from the
        block's current layers (reference quant_wanx.py:188-228 loads `int_weight.pt` into freshly constructed
        blocks; here the quantized layers in memory are the source of truth, `load_path` is accepted for call
        compatibility and verified against them when given)."""
        self.hip_blocks = nn.ModuleList([WanAttentionBlockWithHipKernel.from_float(b, None, False, act_dtype) for b in self.blocks])
        if load_path:
            sd = torch.load(load_path, map_location="cpu", weights_only=True)
            for i, hb in enumerate(self.hip_blocks):
                k = f"blocks.{i}.self_attn.q.weight"
                if k in sd and hb.self_attn.q.quantized:
                    assert torch.equal(sd[k], hb.self_attn.q.weight.cpu()), f"{load_path} does not match the in-memory model at {k}"
        return self

    def software_forward(self):
        self.hip_blocks = None
        return self

    def _rope(self, grid, device):
        if grid not in self._rope_cache:
            self._rope_cache[grid] = ops.rope_table(self.freqs, grid, device)
        return self._rope_cache[grid]

    @torch.no_grad()
    def forward(self, x, t, context, seq_len, sp=None):
        if self.hip_blocks is None:
            assert sp is None or sp.size == 1, "sequence parallelism is wired for kernel mode"
            return WanModel.forward(self, x, t, context, seq_len)
        outs = []
        for xi, ci, ti in zip(x, context, t.reshape(-1, 1)):
            with torch.autocast("cuda", enabled=False):
                h, e, e0, ctx, seq_lens, grids = self.embed([xi], ti, [ci], seq_len)
                h = h[0].float()
                rope = self._rope(grids[0], h.device)
                if sp is not None and sp.size > 1:
                    lp = seq_len // sp.size
                    h = sp.shard_rows(h)
                    rope = rope[sp.rank * lp:(sp.rank + 1) * lp]
                h = h.contiguous()
                cq = _FpSrc(ctx[0].float().contiguous(), self.hip_blocks[0].act_dtype)
                for blk in self.hip_blocks:
                    blk(h, e0.float(), rope, seq_lens[0], cq, sp)
                out = self.head(h.unsqueeze(0), e)
                if sp is not None and sp.size > 1:
                    out = sp.all_gather_rows(out[0]).unsqueeze(0)
                outs.append(self.unpatchify(out, grids)[0].float())
        return outs
