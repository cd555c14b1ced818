"""Model-surgery helpers with the reference's names and calling convention
(ViDiT-Q/quant_utils/qdiff/base/quant_model.py:15-175): they are passed to apply_func_to_submodules."""
import logging
import re

import torch
import torch.nn as nn

from ..utils import apply_func_to_submodules
from .base_quantizer import BaseQuantizer
from ..quarot.quarot_quant_layer import QuarotQuantizedLinear
from ..smooth_quant.sq_quant_layer import SQQuantizedLinear
from ..viditq.viditq_quant_layer import ViDiTQuantizedLinear
from .quant_layer import QuantizedLinear

logger = logging.getLogger(__name__)


def pick_layer_type(quant_config, full_name):
    """smooth_quant < quarot < viditq precedence, each gated by its layer_name_regex (reference :17-53)."""
    layer_type = QuantizedLinear
    for key, cls in (("smooth_quant", SQQuantizedLinear), ("quarot", QuarotQuantizedLinear), ("viditq", ViDiTQuantizedLinear)):
        sub = quant_config.get(key, None)
        if sub is not None and re.search(re.compile(sub.layer_name_regex), full_name):
            layer_type = cls
    return layer_type


def quant_layer_refactor_(submodule, name, parent_module, quant_config, full_name, remain_fp_regex):
    """Replace one nn.Linear by its quantized counterpart unless remain_fp_regex matches (reference :15-74)."""
    if isinstance(submodule, QuantizedLinear):
        return
    layer_type = pick_layer_type(quant_config, full_name)
    if remain_fp_regex is not None and re.compile(remain_fp_regex).search(full_name):
        logger.info("remain %s as FP due to fp_regex", full_name)
        return
    q = layer_type(submodule.in_features, submodule.out_features, submodule.bias is not None, submodule.weight.device,
                   quant_config, submodule)
    q.module_name = full_name
    for qz in (q.w_quantizer, q.a_quantizer):
        if qz is not None:
            qz.module_name = full_name
    setattr(parent_module, name, q)


def bitwidth_refactor_(submodule, name, parent_module, quant_config, full_name):
    """mixed_precision.{weight,act}.layer_name_regex: list index 0 => FP16 (quant off), index k => n_bits[k-1]
    (reference :76-105).  Unlike the reference the weight is re-quantised at the new width."""
    mp = quant_config.mixed_precision
    for kind, quantizer in (("weight", submodule.w_quantizer), ("act", submodule.a_quantizer)):
        for idx, rgx in enumerate(mp[kind].layer_name_regex):
            if len(rgx) == 0 or not re.search(re.compile(rgx), full_name):
                continue
            if idx == 0:
                submodule.quant_mode = False
                logger.info("[Mixed Precision] %s %s -> FP16", full_name, kind)
            else:
                quantizer.bitwidth_refactor(idx - 1)
                if kind == "weight":
                    submodule.refresh()
                logger.info("[Mixed Precision] %s %s -> %d bit", full_name, kind, quantizer.bitwidth_list[idx - 1])


def save_quant_param_dict_(submodule, full_name, parent_module, model):
    """One entry per quantizer: delta, zero_point, the parent's channel_mask; rotation_matrix is stored as None like the
    reference (:161-172) -- plus `rotation_signs`, the [C] +-1 vector from which the rotation is rebuilt exactly."""
    d = {"delta": submodule.delta, "zero_point": submodule.zero_point}
    if getattr(parent_module, "uses_mask", False):
        d["channel_mask"] = parent_module.channel_mask
    if getattr(parent_module, "uses_rotation", False):
        d["rotation_matrix"] = None
        d["rotation_signs"] = parent_module.rotation_signs
    model.quant_param_dict[full_name] = d


def load_quant_param_dict_(submodule, full_name, parent_module, quant_param_dict, model):
    """Restore a quantizer and re-derive its layer's weight (reference :138-159).  When the dict carries no
    rotation_signs (a file written by the reference) a fresh rotation is drawn, as the reference does."""
    entry = quant_param_dict[full_name]
    dev = parent_module.fp_module.weight.device
    submodule.delta = None if entry["delta"] is None else entry["delta"].to(dev)
    submodule.zero_point = None if entry["zero_point"] is None else entry["zero_point"].to(dev)
    is_weight_q = submodule is parent_module.w_quantizer
    if getattr(parent_module, "uses_mask", False) and entry.get("channel_mask") is not None:
        parent_module.channel_mask = entry["channel_mask"].to(dev)
        parent_module._premul = parent_module._rot = None
    if getattr(parent_module, "uses_rotation", False):
        if entry.get("rotation_signs") is not None:
            parent_module.rotation_signs = entry["rotation_signs"]
            parent_module._premul = parent_module._rot = None
        elif parent_module.rotation_signs is None:
            parent_module.get_rotation_matrix()
    if is_weight_q:
        if isinstance(parent_module, ViDiTQuantizedLinear):
            parent_module.update_quantized_weight_rotated_and_scaled()
        elif isinstance(parent_module, QuarotQuantizedLinear):
            parent_module.update_quantized_weight_rotated()
        elif isinstance(parent_module, SQQuantizedLinear):
            parent_module.update_quantized_weight_scaled()
        else:
            parent_module.refresh()
    model.quant_param_dict[full_name] = entry


def set_init_done_(submodule):
    submodule.init_done = True


class QuantModel(nn.Module):
    """Mixin-style template (reference :182-233): subclasses call these on themselves."""

    def quant_layer_refactor(self):
        apply_func_to_submodules(self, class_type=nn.Linear, function=quant_layer_refactor_, name=None, parent_module=None,
                                 quant_config=self.q_cfg, full_name=None, remain_fp_regex=self.q_cfg.get("remain_fp_regex", None))

    def save_quant_param_dict(self):
        self.quant_param_dict = {}
        apply_func_to_submodules(self, class_type=BaseQuantizer, function=save_quant_param_dict_, full_name=None,
                                 parent_module=None, model=self)
        return self.quant_param_dict

    def load_quant_param_dict(self, quant_param_dict):
        if not hasattr(self, "quant_param_dict"):
            self.quant_param_dict = {}
        apply_func_to_submodules(self, class_type=BaseQuantizer, function=load_quant_param_dict_, full_name=None,
                                 parent_module=None, quant_param_dict=quant_param_dict, model=self)

    def set_init_done(self):
        apply_func_to_submodules(self, class_type=BaseQuantizer, function=set_init_done_)

    def bitwidth_refactor(self):
        apply_func_to_submodules(self, class_type=QuantizedLinear, function=bitwidth_refactor_, name=None, parent_module=None,
                                 quant_config=self.q_cfg, full_name=None)
