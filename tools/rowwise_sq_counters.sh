#!/bin/bash
# SQ counters of the row-wise kernels (tools/rowwise_multi_bench.py + tools/rowwise_once.py), separate --pmc passes.
#   tools/rowwise_sq_counters.sh <tag>  ->  gpurun_out/<tag>_rowwise_sq.csv
set -o pipefail
TAG=${1:-r03_x}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD"
P3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
i=0
for DRV in tools/rowwise_multi_bench.py tools/rowwise_once.py; do
for P in "$P1" "$P2" "$P3"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-include-regex "rotate_kernel|rowwise_kernel|rmsnorm_rope" --output-format csv -d "$OUT/${TAG}_rw_pass$i" -o pmc -- python3 $DRV \
    > "$OUT/${TAG}_rw_pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/${TAG}_rw_pass$i.log"; }
done
done
python3 - "$OUT" "$TAG" > "$OUT/${TAG}_rowwise_sq.csv" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for d in sorted(glob.glob(os.path.join(out, f"{tag}_rw_pass*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = defaultdict(lambda: defaultdict(float)); names = {}
        for r in csv.DictReader(open(f)):
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"]); names[int(r["Dispatch_Id"])] = r["Kernel_Name"].replace("void wanq::", "").split("(")[0]
        for did, cs in per.items():
            for c, v in cs.items():
                acc[names[did]][c].append(v)
cols = sorted({c for k in acc.values() for c in k})
w = csv.writer(sys.stdout); w.writerow(["kernel", "dispatches"] + cols + ["cycles_per_valu_inst", "wait_frac", "issue_stall_frac", "valu_active_frac"])
for k, cs in sorted(acc.items()):
    med = {c: sorted(v)[len(v) // 2] for c, v in cs.items()}
    g = lambda c: med.get(c, float("nan"))
    wc = g("SQ_WAVE_CYCLES") * 4
    w.writerow([k, len(next(iter(cs.values())))] + [f"{med.get(c, float('nan')):.4g}" for c in cols] +
               [f"{wc / g('SQ_INSTS_VALU'):.2f}", f"{g('SQ_WAIT_ANY') * 4 / wc:.3f}", f"{g('SQ_WAIT_INST_ANY') * 4 / wc:.3f}", f"{g('SQ_ACTIVE_INST_VALU') * 4 / wc:.3f}"])
PY
cat "$OUT/${TAG}_rowwise_sq.csv"
