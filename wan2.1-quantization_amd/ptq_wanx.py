#!/usr/bin/env python3
"""Post-training quantization -- entry point 3/4 (ViDiT-Q/examples/Wan2.1/ptq_wanx.py): replace the Linears per the
quant config, fit channel masks / rotations from the calibration data, and save `checkpoint/quant_params.pth`
(+ `checkpoint/int_weight.pt`, the integer state dict)."""
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from qdiff import config as qcfg  # noqa: E402
from qdiff.base.quant_layer import QuantizedLinear  # noqa: E402
from qdiff.utils import apply_func_to_submodules, seed_everything  # noqa: E402
from wan import calib, cli  # noqa: E402
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
    if any(quant_config.get(k, None) is not None for k in ("viditq", "smooth_quant", "quarot")):
        calib_data = torch.load(args.calib_data or quant_config.calib_data.save_path, weights_only=True)
        g = torch.Generator().manual_seed(args.base_seed)
        apply_func_to_submodules(model, class_type=QuantizedLinear,
                                 function=lambda m, full_name, calib_data: calib.init_rotation_and_channel_mask_(m, full_name, calib_data, g)
                                 if (m.uses_mask or m.uses_rotation) else None, full_name="", calib_data=calib_data)
    if quant_config.get("mixed_precision", None) is not None:
        model.bitwidth_refactor()
    model.set_init_done()
    params = model.save_quant_param_dict()
    if rank == 0:
        out = os.path.join(args.output_dir, "checkpoint")
        os.makedirs(out, exist_ok=True)
        torch.save({k: {kk: (vv.cpu() if torch.is_tensor(vv) else vv) for kk, vv in v.items()} for k, v in params.items()},
                   args.quant_params or os.path.join(out, "quant_params.pth"))
        model.quantize_and_save_weight(os.path.join(out, "int_weight.pt"))
        logging.info("saved quant params of %d quantizers to %s", len(params), out)
    return 0


if __name__ == "__main__":
    sys.exit(main(cli.validate_args(cli.build_parser("PTQ", quant=True).parse_args())))
