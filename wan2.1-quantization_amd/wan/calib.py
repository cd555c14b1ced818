"""PTQ calibration: per-input-channel absmax of every nn.Linear input, reduced on the device.

The reference registers a forward hook on every Linear that does `x.reshape(-1,C).abs().max(dim=0)` per call and
appends the result to a Python list (get_calib_data_wanx.py:240-275), stacks the list at the end (:443-449) and
gathers pickled dicts across ranks (:455-470); ptq then takes `.max(dim=0)` over the stack (ptq_wanx.py:336).
Here each hook folds its call into ONE running fp32 [C] vector with the HIP column-absmax kernel, ranks are
joined with all_reduce(MAX), and the saved file keeps the reference's format {layer_name: Tensor[N, C]} with
N = 1 (so `calib_data[name].max(dim=0)[0]` gives the same mask)."""
import torch
import torch.distributed as dist
import torch.nn as nn

from viditq_extension import fused


class SaveActivationHook:
    def __init__(self):
        self.running = None
        self.calls = 0
        self.hook_handle = None

    def __call__(self, module, module_in, module_out):
        x = module_in[0]
        c = x.shape[-1]
        if self.running is None:
            self.running = torch.zeros(c, dtype=torch.float32, device=x.device)
        x2 = x.reshape(-1, c)
        if x2.dtype not in (torch.float16, torch.bfloat16, torch.float32):
            x2 = x2.float()
        fused.col_absmax_(self.running, x2.contiguous())
        self.calls += 1

    @property
    def outputs(self):
        return [self.running]


def add_hooks(model, class_type=nn.Linear):
    hooks = {}
    for name, mod in model.named_modules():
        if isinstance(mod, class_type):
            h = SaveActivationHook()
            h.hook_handle = mod.register_forward_hook(h)
            hooks[name] = h
    return hooks


def gather_and_save_activation(hooks, save_path=None, group=None):
    """{clean layer name: [1, C] fp32 on CPU}; all ranks reduce with MAX first."""
    out = {}
    for name, h in hooks.items():
        if h.hook_handle is not None:
            h.hook_handle.remove()
        if h.running is None:
            continue
        r = h.running
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(r, op=dist.ReduceOp.MAX, group=group)
        out[name.replace("_fsdp_wrapped_module.", "")] = r.unsqueeze(0).cpu()
    if save_path and (not dist.is_initialized() or dist.get_rank() == 0):
        torch.save(out, save_path)
    return out


def init_rotation_and_channel_mask_(module, full_name, calib_data, generator=None):
    """ptq_wanx.py:334-344: act_mask = max over calls, floor 1e-3, then mask -> rotation -> re-quantised weight."""
    act_mask = calib_data[full_name].max(dim=0)[0].to(module.fp_module.weight.device)
    act_mask = torch.where(act_mask < 1e-3, torch.full_like(act_mask, 1e-3), act_mask)
    if getattr(module, "uses_mask", False):
        module.get_channel_mask(act_mask)
    if getattr(module, "uses_rotation", False):
        module.get_rotation_matrix(generator)
    if module.uses_mask and module.uses_rotation:
        module.update_quantized_weight_rotated_and_scaled()
    elif module.uses_mask:
        module.update_quantized_weight_scaled()
    elif module.uses_rotation:
        module.update_quantized_weight_rotated()
