#!/usr/bin/env python3
"""Calibration data collection -- entry point 2/4 (ViDiT-Q/examples/Wan2.1/get_calib_data_wanx.py): run the FP model
with a forward hook on every nn.Linear that keeps the per-input-channel absmax (HIP reduction, running max on the
device), and save {layer_name: [1, C_in]} to quant_config.calib_data.save_path."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from qdiff import config as qcfg  # noqa: E402
from wan import calib, cli  # noqa: E402
from wan.configs import SIZE_CONFIGS  # noqa: E402
from wan.text2video import WanT2V  # noqa: E402


def main(args):
    cfg = cli.model_config(args)
    rank, world, local, plan = cli.setup_distributed(args, cfg["num_heads"])
    cli.init_logging(rank)
    quant_config = qcfg.load(args.quant_config)
    save_path = args.calib_data or quant_config.calib_data.save_path
    # --ulysses_size P (get_calib_data_wanx.py:455-473 runs under the reference's USP patch): the P ranks of a sequence-parallel
    # group run ONE prompt on token shards (each hook sees its shard's rows); without it every rank runs its own prompts.  Either
    # way the per-channel maxima of all ranks are MAX-reduced at the end, which is what the reference's concatenation + ptq's
    # .max(dim=0) amounts to.
    sharded = args.ulysses_size > 1
    t2v = WanT2V(cfg, args.ckpt_dir, device_id=local, rank=rank, plan=plan if sharded else None, context_file=args.context_file)
    hooks = calib.add_hooks(t2v.model, torch.nn.Linear)
    logging.info("hooked %d Linear layers", len(hooks))
    prompts = cli.read_prompts(args)
    if sharded:
        mine = prompts  # world = ulysses_size (x 2 with CFG parallelism): one group, every prompt
    else:
        mine = prompts[rank::world] or prompts[:1]  # prompts are data-parallel over ranks; the masks are MAX-reduced
    for prompt in mine:
        t2v.generate(prompt, size=SIZE_CONFIGS[args.size], frame_num=args.frame_num, shift=args.sample_shift,
                     sample_solver=args.sample_solver, sampling_steps=args.sample_steps, guide_scale=args.sample_guide_scale,
                     seed=args.base_seed, offload_model=args.offload_model)
    os.makedirs(os.path.dirname(os.path.abspath(save_path)), exist_ok=True)
    data = calib.gather_and_save_activation(hooks, save_path)
    logging.info("saved calibration data of %d layers to %s", len(data), save_path)
    return 0


if __name__ == "__main__":
    sys.exit(main(cli.validate_args(cli.build_parser("calibration", quant=True).parse_args())))
