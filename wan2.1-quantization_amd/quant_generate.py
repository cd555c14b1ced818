#!/usr/bin/env python3
"""Quantized generation -- entry point 4/4 (ViDiT-Q/examples/Wan2.1/quant_generate.py): load the quant params, switch
the DiT to kernel mode (int8-MFMA GEMMs with fused producers, HIP flash attention; --hardware false = simulation-mode
layers inside the unmodified block) and run the denoising loop, optionally sequence / CFG parallel."""
import logging
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from qdiff import config as qcfg  # noqa: E402
from qdiff.utils import seed_everything  # noqa: E402
from wan import cli  # noqa: E402
from wan.configs import SIZE_CONFIGS  # noqa: E402
from wan.quant_wanx import QuantWanModel  # noqa: E402
from wan.text2video import WanT2V  # noqa: E402


def main(args):
    cfg = cli.model_config(args)
    rank, world, local, plan = cli.setup_distributed(args, cfg["num_heads"])
    cli.init_logging(rank)
    seed_everything(args.base_seed)
    quant_config = qcfg.load(args.quant_config)
    fp = WanT2V(cfg, args.ckpt_dir, device_id=local, rank=rank).model
    model = QuantWanModel.from_float(fp, quant_config)
    del fp
    model.quant_layer_refactor()
    qp = args.quant_params or os.path.join(args.output_dir, "checkpoint", "quant_params.pth")
    params = torch.load(qp, map_location="cpu", weights_only=True)
    params = {k.replace("_fsdp_wrapped_module.", ""): v for k, v in params.items()}  # quant_generate.py:385-388
    model.load_quant_param_dict(params)
    if quant_config.get("mixed_precision", None) is not None:
        model.bitwidth_refactor()
    model.set_init_done()
    if args.hardware:
        iw = os.path.join(args.output_dir, "checkpoint", "int_weight.pt")
        # the reference exports the integer checkpoint and loads it into the kernel-mode blocks (quant_generate.py:397-409);
        # ptq_wanx.py already wrote it next to quant_params.pth: load it when it is there
        model.hardware_forward_refactor(iw if os.path.exists(iw) else None)
        if args.dit_fsdp and world > 1:  # FULL_SHARD of the DiT blocks (wan/distributed/fsdp.py): integer weights over all ranks
            sh = model.shard_blocks(None)
            logging.info("dit_fsdp: %.1f MB of block weights per rank (of %.1f MB)", sh.bytes_per_rank() / 1e6,
                         len(sh.blocks) * sh.full_bytes / 1e6)
    # (simulation mode, --hardware false, shards through the FP model's Ulysses path: its fake-quant Linears are token-local)
    t2v = WanT2V(cfg, device_id=local, rank=rank, model=model.eval(), plan=plan, context_file=args.context_file)
    os.makedirs(args.output_dir, exist_ok=True)
    for i, prompt in enumerate(cli.read_prompts(args)):
        t0 = time.perf_counter()
        latent = t2v.generate(prompt, size=SIZE_CONFIGS[args.size], frame_num=args.frame_num, shift=args.sample_shift,
                              sample_solver=args.sample_solver, sampling_steps=args.sample_steps,
                              guide_scale=args.sample_guide_scale, seed=args.base_seed, offload_model=args.offload_model)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        logging.info("prompt %d: %d steps in %.2f s (%.3f steps/s, %s)", i, args.sample_steps, dt, args.sample_steps / dt, plan.describe())
        if rank == 0:
            torch.save(latent.cpu(), args.save_file or os.path.join(args.output_dir, f"quant_latent_{i}.pt"))
    return 0


if __name__ == "__main__":
    sys.exit(main(cli.validate_args(cli.build_parser("quantized generation", quant=True).parse_args())))
