#!/bin/bash
# BASELINE config 3 at full size: PTQ calibration of Wan2.1-T2V-1.3B (synthetic weights), 832x480x81f, 30 sampling steps through
# the activation hooks (get_calib_data_wanx.py), then ptq_wanx.py (masks, rotations, integer weights).  Prints wall times.
set -e
OUT=${1:-/tmp/calib_full}; STEPS=${2:-30}
PKG=$(dirname "$0")/../wan2.1-quantization_amd
QC=$PKG/quant_configs/w8a8_all_linears.yaml
COMMON="--task t2v-1.3B --size 832*480 --frame_num 81 --sample_steps $STEPS --base_seed 42 --output_dir $OUT"
mkdir -p "$OUT"
t0=$(date +%s.%N); python $PKG/get_calib_data_wanx.py $COMMON --quant_config $QC --calib_data $OUT/calib.pth | grep -v "^\[" || true
t1=$(date +%s.%N); python $PKG/ptq_wanx.py $COMMON --quant_config $QC --calib_data $OUT/calib.pth | grep -v "^\[" || true
t2=$(date +%s.%N)
python3 - "$OUT" "$STEPS" "$t0" "$t1" "$t2" <<'PY'
import sys, torch
out, steps, t0, t1, t2 = sys.argv[1], int(sys.argv[2]), *map(float, sys.argv[3:6])
cd = torch.load(out + "/calib.pth", weights_only=True)
n = len(cd); k = next(iter(cd)); calls = cd[k].shape[0]
print(f"get_calib_data_wanx: {t1 - t0:.1f} s wall for {steps} steps x 2 passes (process start, model build and {steps * 2} hooked FP passes: "
      f"{n} layers x {calls} recorded calls of per-channel absmax); ptq_wanx: {t2 - t1:.1f} s wall")
PY
