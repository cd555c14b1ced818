#!/bin/bash
# Per-kernel time of the hooked FP passes of calibration (BASELINE config 3) at the headline size: rocprofv3 --kernel-trace --stats over
# get_calib_data_wanx.py with 2 sampling steps (4 hooked passes); top kernels by total time -> gpurun_out/<tag>_fp_pass_kernels.txt
set -o pipefail
TAG=${1:-r05_fp}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; export TMPDIR=/tmp
PKG=wan2.1-quantization_amd
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof" -o fp -- python3 $PKG/get_calib_data_wanx.py --task t2v-1.3B --size '832*480' --frame_num 81 --sample_steps 2 --base_seed 42 --output_dir /tmp/fp_prof --quant_config $PKG/quant_configs/w8a8_all_linears.yaml --calib_data /tmp/fp_prof/calib.pth > "$OUT/${TAG}_run.log" 2>&1 || { tail -5 "$OUT/${TAG}_run.log"; exit 1; }
S=$(find "$OUT/${TAG}_prof" -name "*kernel_stats.csv" | head -1)
python3 - "$S" > "$OUT/${TAG}_fp_pass_kernels.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"4 hooked FP passes (2 steps x cond / uncond), all kernels: {tot / 1e6:.1f} ms = {tot / 4e6:.1f} ms per pass")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f"{float(r['TotalDurationNs']) / 4e6:8.2f} ms/pass {100 * float(r['TotalDurationNs']) / tot:5.1f}%  {int(r['Calls']) // 4:5d} calls/pass  {r['Name'][:150]}")
PY
cat "$OUT/${TAG}_fp_pass_kernels.txt"
find "$OUT/${TAG}_prof" -name "*kernel_trace.csv" -delete
