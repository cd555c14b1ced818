#!/usr/bin/env python3
"""FP (bf16-autocast) generation -- entry point 1/4 (ViDiT-Q/examples/Wan2.1/fp_generate.py).
    torchrun --nproc_per_node=N fp_generate.py --task t2v-1.3B --size 832*480 --ckpt_dir ... [--ulysses_size P]
Writes the denoised latent of every prompt to <output_dir>/fp_latent_<i>.pt (VAE decode is out of scope)."""
import logging
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from wan import cli  # noqa: E402
from wan.configs import SIZE_CONFIGS  # noqa: E402
from wan.text2video import WanT2V  # noqa: E402


def generate(args):
    cfg = cli.model_config(args)
    rank, world, local, plan = cli.setup_distributed(args, cfg["num_heads"])
    cli.init_logging(rank)
    t2v = WanT2V(cfg, args.ckpt_dir, device_id=local, rank=rank, t5_fsdp=args.t5_fsdp, dit_fsdp=args.dit_fsdp,
                 t5_cpu=args.t5_cpu, plan=plan, context_file=args.context_file)
    os.makedirs(args.output_dir, exist_ok=True)
    for i, prompt in enumerate(cli.read_prompts(args)):
        t0 = time.perf_counter()
        latent = t2v.generate(prompt, size=SIZE_CONFIGS[args.size], frame_num=args.frame_num, shift=args.sample_shift,
                              sample_solver=args.sample_solver, sampling_steps=args.sample_steps,
                              guide_scale=args.sample_guide_scale, seed=args.base_seed, offload_model=args.offload_model)
        torch.cuda.synchronize()
        logging.info("prompt %d: %d steps in %.2f s", i, args.sample_steps, time.perf_counter() - t0)
        if rank == 0:
            torch.save(latent.cpu(), args.save_file or os.path.join(args.output_dir, f"fp_latent_{i}.pt"))
    return 0


if __name__ == "__main__":
    sys.exit(generate(cli.validate_args(cli.build_parser("FP generation").parse_args())))
