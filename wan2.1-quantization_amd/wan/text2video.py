"""WanT2V: the text-to-video denoising pipeline around the DiT (interface of
ViDiT-Q/examples/Wan2.1/wan/text2video.py:28-318: ctor arguments, generate() arguments, shape arithmetic).

In scope here is the denoising loop -- cond pass, uncond pass, CFG, scheduler update (text2video.py:248-269).
The T5 text encoder and the VAE are out of scope (they run once per video, are not quantized, and no
weights exist in this environment): text conditioning comes from `encode_text` (a file of precomputed T5 states,
or a deterministic synthetic embedding of the right shape), and generate() returns the final LATENT
[16, (F-1)/4+1, H/8, W/8] instead of decoded frames."""
import hashlib
import logging
import math
import os

import torch

from .configs import latent_shape, seq_len_for
from .modules.model import WanModel
from .utils.fm_solvers import FlowDPMSolverMultistepScheduler, FlowMatchScheduler
from .utils.fused_step import FusedStep
from .utils.two_pass import TwoPassStreams
from .utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

logger = logging.getLogger(__name__)


def synth_wan_model(model_kwargs, device, seed=0):
    """Random-init backbone (xavier Linears as WanModel.init_weights, small random biases, non-zero head)."""
    torch.manual_seed(seed)
    with torch.device(device):
        m = WanModel(**model_kwargs)
    g = torch.Generator(device=device).manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Linear) and mod.bias is not None:
            mod.bias.data.normal_(std=0.02, generator=g)
    torch.nn.init.xavier_uniform_(m.head.head.weight, generator=g)
    return m.eval()


class WanT2V:
    def __init__(self, config, checkpoint_dir=None, device_id=0, rank=0, t5_fsdp=False, dit_fsdp=False, use_usp=False,
                 t5_cpu=False, model=None, plan=None, context_file=None):
        self.device = torch.device(f"cuda:{device_id}")
        self.config, self.rank = config, rank
        self.num_train_timesteps = config["num_train_timesteps"]
        self.param_dtype = config["param_dtype"]
        self.vae_stride, self.patch_size = config["vae_stride"], config["patch_size"]
        self.sp_size = plan.sp_degree if plan is not None else 1
        self.plan = plan
        self.context_file = context_file
        if dit_fsdp and hasattr(model, "shard_blocks") and getattr(model, "hip_blocks", None) is not None \
                and getattr(model, "_fsdp", None) is None and plan is not None and plan.world > 1:
            model.shard_blocks(None)  # kernel-mode blocks: integer weights 1/P per rank (wan/distributed/fsdp.py)
        elif dit_fsdp or t5_fsdp:
            logger.info("--dit_fsdp / --t5_fsdp: sharding applies to the kernel-mode DiT blocks (QuantWanModel.shard_blocks); "
                        "the FP model and the text encoder stay replicated (bf16 14B = 28 GB << 288 GB HBM)")
        if model is not None:
            self.model = model
        elif checkpoint_dir and os.path.exists(os.path.join(checkpoint_dir, "config.json")):
            self.model = WanModel.from_pretrained(checkpoint_dir).to(self.device).eval()
        else:
            from .configs import MODEL_KEYS

            logger.warning("no checkpoint at %r: using a random-init model of the configured architecture", checkpoint_dir)
            self.model = synth_wan_model({k: config[k] for k in MODEL_KEYS}, self.device)
        self.sample_neg_prompt = config.get("sample_neg_prompt", "")

    def encode_text(self, prompt):
        """[L_txt <= 512, 4096] text states.  File of precomputed T5 outputs ({prompt: tensor}) if given, else a
        deterministic synthetic embedding seeded by the prompt (N(0, 0.1), the scale of T5 states)."""
        if self.context_file:
            d = torch.load(self.context_file, map_location=self.device, weights_only=True)
            if prompt in d:
                return d[prompt].to(self.device).float()
        seed = int.from_bytes(hashlib.sha256(prompt.encode()).digest()[:4], "little")
        g = torch.Generator(device=self.device).manual_seed(seed)
        n = max(8, min(self.config["text_len"], len(prompt.split()) * 2 + 8))
        return torch.randn(n, self.config["text_dim"], generator=g, device=self.device) * 0.1

    def generate(self, input_prompt, size=(1280, 720), frame_num=81, shift=5.0, sample_solver="unipc", sampling_steps=50,
                 guide_scale=5.0, n_prompt="", seed=-1, offload_model=True, step_callback=None):
        """Returns the denoised latent [16, F', H/8, W/8] (fp32)."""
        target_shape = latent_shape(size, frame_num, self.vae_stride, self.model.in_dim)
        seq_len = seq_len_for(target_shape, self.patch_size, self.sp_size)
        if n_prompt == "":
            n_prompt = self.sample_neg_prompt
        seed = seed if seed >= 0 else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        seed_g = torch.Generator(device=self.device).manual_seed(seed)
        context, context_null = self.encode_text(input_prompt), self.encode_text(n_prompt)
        latent = torch.randn(*target_shape, dtype=torch.float32, device=self.device, generator=seed_g)

        if sample_solver not in ("unipc", "dpm++", "euler"):
            raise NotImplementedError(f"Unsupported solver {sample_solver}")
        if sample_solver == "unipc":    # the reference's default (text2video.py:215-222)
            sched = FlowUniPCMultistepScheduler(self.num_train_timesteps, shift=1.0)
        elif sample_solver == "dpm++":  # text2video.py:223-232: sigmas from get_sampling_sigmas(steps, shift)
            sched = FlowDPMSolverMultistepScheduler(self.num_train_timesteps, shift=1.0)
        else:                           # first-order flow-matching update
            sched = FlowMatchScheduler(self.num_train_timesteps, shift=1.0)
        sched.set_timesteps(sampling_steps, device=self.device, shift=shift)
        # guidance + scheduler update as one kernel per step (wan/utils/fused_step.py) on the GPU
        fused = FusedStep(sched, guide_scale, latent) if latent.is_cuda else None
        plan = self.plan
        sp = plan.sp if plan is not None else None
        kw = {"sp": sp} if sp is not None and sp.size > 1 else {}
        # one rank, kernel mode: the two passes of a step on one HIP stream or on two, whichever steps 2 - 5 of this loop measure faster
        # (wan/utils/two_pass.py: bit-equal latents either way; WANQ_PASS_STREAMS=1 / 2 fixes the order)
        two = TwoPassStreams(self.device, enabled=None if (latent.is_cuda and not kw and (plan is None or plan.cfg_degree == 1)
                                                            and getattr(self.model, "hip_blocks", None) is not None) else False)
        with torch.no_grad(), torch.autocast("cuda", dtype=self.param_dtype):
            for i, t in enumerate(sched.timesteps):
                ts = t.reshape(1)
                if plan is not None and plan.cfg_degree == 2:
                    mine = self.model([latent], ts, [context if plan.cfg_index == 0 else context_null], seq_len, **kw)[0]
                    cond, uncond = plan.gather_cfg(mine)
                else:
                    cond, uncond = two(lambda c: self.model([latent], ts, [c], seq_len, **kw)[0], latent, [context, context_null])
                if fused is not None:
                    latent = fused.step(cond.float(), uncond.float(), latent, t)
                else:
                    noise_pred = uncond + guide_scale * (cond - uncond)
                    latent = sched.step(noise_pred, latent) if sample_solver == "euler" else sched.step(noise_pred, t, latent)
                if step_callback is not None:
                    step_callback(i, latent)
        return latent
