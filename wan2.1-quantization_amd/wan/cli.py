"""Shared command-line surface of the four entry points (flags of ViDiT-Q/examples/Wan2.1/fp_generate.py:75-199 plus
--quant_config, ptq_wanx.py:206-210)."""
import argparse
import logging
import os
import random
import sys

import torch
import torch.distributed as dist

from .configs import SIZE_CONFIGS, WAN_CONFIGS

EXAMPLE_PROMPT = {"t2v-1.3B": "Two anthropomorphic cats in comfy boxing gear and bright gloves fight intensely on a spotlighted stage.",
                  "t2v-14B": "Two anthropomorphic cats in comfy boxing gear and bright gloves fight intensely on a spotlighted stage."}


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected")


def build_parser(description, quant=False):
    p = argparse.ArgumentParser(description=description)
    p.add_argument("--task", type=str, default="t2v-14B", choices=list(WAN_CONFIGS.keys()))
    p.add_argument("--size", type=str, default="1280*720", choices=list(SIZE_CONFIGS.keys()))
    p.add_argument("--frame_num", type=int, default=None, help="4n+1 frames")
    p.add_argument("--ckpt_dir", type=str, default=None, help="Wan2.1 checkpoint directory (config.json + *.safetensors)")
    p.add_argument("--offload_model", type=str2bool, default=None)
    p.add_argument("--ulysses_size", type=int, default=1)
    p.add_argument("--ring_size", type=int, default=1, help="accepted; ring attention is not implemented (must be 1)")
    p.add_argument("--t5_fsdp", action="store_true", default=False)
    p.add_argument("--t5_cpu", action="store_true", default=False)
    p.add_argument("--dit_fsdp", action="store_true", default=False)
    p.add_argument("--save_file", type=str, default=None)
    p.add_argument("--prompt", type=str, default=None)
    p.add_argument("--prompt_file", type=str, default=None, help="one prompt per line (the reference hard-codes test_prompts.txt)")
    p.add_argument("--use_prompt_extend", action="store_true", default=False, help="accepted; prompt extension is out of scope")
    p.add_argument("--base_seed", type=int, default=-1)
    p.add_argument("--sample_solver", type=str, default="unipc", choices=["unipc", "dpm++", "euler"])
    p.add_argument("--sample_steps", type=int, default=None)
    p.add_argument("--sample_shift", type=float, default=None)
    p.add_argument("--sample_guide_scale", type=float, default=5.0)
    p.add_argument("--context_file", type=str, default=None, help="torch file {prompt: T5 states [L,4096]} (T5 itself is out of scope)")
    p.add_argument("--cfg_parallel", type=str2bool, default=True, help="run cond / uncond passes on different GPUs when possible")
    p.add_argument("--output_dir", type=str, default="./outputs")
    p.add_argument("--num_layers", type=int, default=None, help=argparse.SUPPRESS)  # test hook: truncate the backbone
    if quant:
        p.add_argument("--quant_config", type=str, default=None, help="Quant config file name.")
        p.add_argument("--calib_data", type=str, default=None, help="overrides quant_config.calib_data.save_path")
        p.add_argument("--quant_params", type=str, default=None, help="checkpoint/quant_params.pth")
        p.add_argument("--hardware", type=str2bool, default=True, help="kernel mode (if_hardware, quant_generate.py:372)")
    return p


def validate_args(args):
    assert args.task in WAN_CONFIGS, f"Unsupport task: {args.task}"
    if args.sample_steps is None:
        args.sample_steps = 50
    if args.sample_shift is None:
        args.sample_shift = 5.0
    if args.frame_num is None:
        args.frame_num = 81
    assert args.ring_size == 1, "ring attention is out of scope (SURVEY 2.3); use --ulysses_size"
    args.base_seed = args.base_seed if args.base_seed >= 0 else random.randint(0, sys.maxsize)
    return args


def init_logging(rank):
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.ERROR,
                        format="[%(asctime)s] %(levelname)s: %(message)s", handlers=[logging.StreamHandler(sys.stdout)], force=True)


def setup_distributed(args, num_heads):
    """env:// process group (backend nccl == RCCL on ROCm) and the cfg x ulysses plan (fp_generate.py:221-259)."""
    from .distributed.parallel import ParallelPlan

    rank, world, local = int(os.getenv("RANK", 0)), int(os.getenv("WORLD_SIZE", 1)), int(os.getenv("LOCAL_RANK", 0))
    # Rehearsal of the multi-rank control flow on a ONE-GPU box (RCCL refuses two ranks on one device): every rank uses cuda:0,
    # the rendezvous is gloo and the collectives are staged through host memory (tools/one_gpu_rehearsal.py: test scaffolding
    # outside this package).  Refused with a non-zero exit on a box with more than one GPU; never a product mode.
    from .distributed import enter_one_gpu_rehearsal, stage_rehearsal_collectives

    rehearse = enter_one_gpu_rehearsal("WANQ_REHEARSE_ON_ONE_GPU", world)
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    if rehearse:
        dist.init_process_group(backend="gloo", init_method="env://", rank=rank, world_size=world)
        stage_rehearsal_collectives()
    elif world > 1:
        dist.init_process_group(backend="nccl", init_method="env://", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    if world > 1:
        seed = [args.base_seed] if rank == 0 else [None]
        dist.broadcast_object_list(seed, src=0)
        args.base_seed = seed[0]
    if args.ulysses_size > 1:
        assert num_heads % args.ulysses_size == 0, f"`num_heads` must be divisible by `ulysses_size`."
        cfg = world // args.ulysses_size
        assert cfg in (1, 2) and cfg * args.ulysses_size == world, "world size must be ulysses_size or 2 x ulysses_size"
        plan = ParallelPlan(world, rank, cfg, args.ulysses_size)
    else:
        plan = ParallelPlan(world, rank, *ParallelPlan.choose(world, num_heads, args.cfg_parallel))
    return rank, world, local, plan


def model_config(args):
    cfg = dict(WAN_CONFIGS[args.task])
    if args.num_layers:
        cfg["num_layers"] = args.num_layers
    return cfg


def read_prompts(args):
    if args.prompt_file and os.path.exists(args.prompt_file):
        lines = [l.strip() for l in open(args.prompt_file) if l.strip()]
        if lines:
            return lines
    return [args.prompt or EXAMPLE_PROMPT[args.task]]
