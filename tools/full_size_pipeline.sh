#!/bin/bash
# The four entry points at the headline size (Wan2.1-T2V-1.3B architecture, synthetic weights, 832x480x81f, all 30 blocks):
# FP generate -> calibration -> PTQ (masks, rotations, int8 weights) -> quantized generate in kernel mode; prints the latent
# deviation of the quantized run from the FP run after N UniPC steps.  Usage: tools/full_size_pipeline.sh <outdir> [steps]
set -e
OUT=${1:?output directory}; STEPS=${2:-10}
PKG=$(dirname "$0")/../wan2.1-quantization_amd
QC=$PKG/quant_configs/w8a8_all_linears.yaml
COMMON="--task t2v-1.3B --size 832*480 --frame_num 81 --sample_steps $STEPS --base_seed 42 --output_dir $OUT"
mkdir -p "$OUT"
t0=$(date +%s); python $PKG/fp_generate.py $COMMON; echo "fp_generate: $(( $(date +%s) - t0 )) s"
t0=$(date +%s); python $PKG/get_calib_data_wanx.py $COMMON --quant_config $QC --calib_data $OUT/calib.pth; echo "get_calib_data_wanx: $(( $(date +%s) - t0 )) s"
t0=$(date +%s); python $PKG/ptq_wanx.py $COMMON --quant_config $QC --calib_data $OUT/calib.pth; echo "ptq_wanx: $(( $(date +%s) - t0 )) s"
t0=$(date +%s); python $PKG/quant_generate.py $COMMON --quant_config $QC; echo "quant_generate: $(( $(date +%s) - t0 )) s"
python - "$OUT" <<'PY'
import sys, torch
fp = torch.load(sys.argv[1] + "/fp_latent_0.pt", weights_only=True).float()
q = torch.load(sys.argv[1] + "/quant_latent_0.pt", weights_only=True).float()
mse = (fp - q).pow(2).mean().item(); rng = (fp.max() - fp.min()).item()
import math
print(f"final latent {tuple(fp.shape)}: rel L2 {((fp - q).norm() / fp.norm()).item():.3e}, PSNR {10 * math.log10(rng * rng / mse):.1f} dB (quantized kernel mode vs FP, after the full sampling loop)")
PY
