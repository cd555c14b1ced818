#!/bin/bash
# HBM-side fetch and time of the ffn.0 GEMM (32760 x 8960 x 1536: the 13.7 MB weight panel streams through a 4-MiB L2) against the
# tile-walk parameter GROUP_M (m-tiles per L2 panel; large = weight-stationary order), + the SQ counters of the three block shapes.
# Separate --pmc passes, never with a trace domain.  usage (GPU box): bash tools/gemm_group_m_probe.sh <out file>
OUT=${1:-gpurun_out/gemm_group_m.txt}
export TMPDIR=/tmp
D=$PWD/gpurun_out/gm_probe; rm -rf "$D"; mkdir -p "$D"
cat > "$D/one.py" <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.getcwd(), "wan2.1-quantization_amd"))
import viditq_extension.qgemm as qgemm
M, N, K = 32760, 8960, 1536
a = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda"); w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda")
sa = torch.rand(M, device="cuda") * 0.01; asum = torch.rand(M, device="cuda"); sw = torch.rand(N, device="cuda") * 0.01
zp = torch.randn(N, device="cuda"); bias = torch.randn(N, device="cuda")
f = lambda: qgemm.w8a8_linear(a, w, sa, sw, bias, asum, zp, out_dtype=torch.bfloat16, gelu=True)
for _ in range(3): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
print("TIME_US", s.elapsed_time(e) * 100)
PY
echo "ffn.0 (32760 x 8960 x 1536, bf16 + GELU epilogue): algorithmic read 64.1 MB" > "$OUT"
for G in 2 4 8 16 128; do
  T=$(WANQ_GEMM_GROUP_M=$G python3 "$D/one.py" | grep TIME_US | awk '{print $2}')
  WANQ_GEMM_GROUP_M=$G timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex gemm_w8a8 --output-format csv -d "$D/f$G" -o pmc -- python3 "$D/one.py" > "$D/f$G.log" 2>&1 || { tail -3 "$D/f$G.log"; exit 1; }
  F=$(find "$D/f$G" -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$G" "$T" >> "$OUT" <<'PY'
import csv, sys
rows = [float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "FETCH_SIZE"]
v = sorted(rows)[len(rows) // 2] * 1024 * 2 / 1e6
print(f"GROUP_M={int(sys.argv[2]):4d}: FETCH_SIZE x2 = {v:8.1f} MB ({v / 64.1:5.1f}x algorithmic)   time {float(sys.argv[3]):7.1f} us")
PY
done
echo >> "$OUT"; echo "SQ counters, default GROUP_M, one launch each of C x C / ffn.0 / ffn.2 (tools/gemm_once.py, last dispatch of each shape):" >> "$OUT"
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  TAG=$(echo $C | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-include-regex gemm_w8a8 --output-format csv -d "$D/$TAG" -o pmc -- python3 tools/gemm_once.py > "$D/$TAG.log" 2>&1 || { tail -3 "$D/$TAG.log"; echo "counter set $C failed" >> "$OUT"; continue; }
  F=$(find "$D/$TAG" -name "*counter_collection.csv" | head -1)
  python3 - "$F" >> "$OUT" <<'PY'
import csv, sys, collections
d = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(d)
for name, i in zip(("C x C", "ffn.0", "ffn.2"), (ids[2], ids[5], ids[8])):
    print(f"  {name:6s} " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(d[i].items())))
PY
done
cat "$OUT"
