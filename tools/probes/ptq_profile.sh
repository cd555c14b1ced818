#!/bin/bash
# Where ptq_wanx.py's wall time goes at the headline size (BASELINE config 3, second half): cProfile, top functions by cumulative time.
# Needs the calibration file of tools/calib_full_size.sh (made here with 1 step when absent).
set -o pipefail
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; D=/tmp/ptq_prof; mkdir -p $D
PKG=wan2.1-quantization_amd; QC=$PKG/quant_configs/w8a8_all_linears.yaml
COMMON="--task t2v-1.3B --size 832*480 --frame_num 81 --sample_steps 1 --base_seed 42 --output_dir $D"
[ -f $D/calib.pth ] || python $PKG/get_calib_data_wanx.py $COMMON --quant_config $QC --calib_data $D/calib.pth > $D/calib.log 2>&1 || { tail -5 $D/calib.log; exit 1; }
python -m cProfile -o $D/ptq.prof $PKG/ptq_wanx.py $COMMON --quant_config $QC --calib_data $D/calib.pth > $D/ptq.log 2>&1 || { tail -5 $D/ptq.log; exit 1; }
python - > "$OUT/${1:-r05}_ptq_profile.txt" <<'PY'
import pstats
p = pstats.Stats("/tmp/ptq_prof/ptq.prof"); p.sort_stats("cumulative").print_stats(45)
PY
head -80 "$OUT/${1:-r05}_ptq_profile.txt"
