#!/bin/bash
# SQ counters of the attention kernels (north star: "rocprof MFMA-utilisation"): three separate --pmc passes (8 SQ slots per
# pass; never combined with a trace domain) over tools/attn_once.py = one cfg-B self-attention (32760 x 32760 x 12 heads) and one
# cross-attention (32760 x 512) launch.   tools/attn_sq_counters.sh <tag> [code]   ->  gpurun_out/<tag>_attention_sq.csv
set -o pipefail
TAG=${1:-r03_x}
CODE=${2:-unknown}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"
P3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="GRBM_GUI_ACTIVE"
P5="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES"
P6="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5" "$P6"; do
  i=$((i + 1))
  timeout -k 10 300 rocprofv3 --pmc $P --kernel-include-regex "attn_fwd" --output-format csv -d "$OUT/${TAG}_sq_pass$i" -o pmc -- python3 tools/attn_once.py \
    > "$OUT/${TAG}_sq_pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$OUT/${TAG}_sq_pass$i.log"; [ $i -lt 4 ] && exit 1; rm -rf "$OUT/${TAG}_sq_pass$i"; }
done
python3 tools/attn_sq_summary.py "$OUT" "$TAG" "$CODE" > "$OUT/${TAG}_attention_sq.csv" && cat "$OUT/${TAG}_attention_sq.csv"
