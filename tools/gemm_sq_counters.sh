#!/bin/bash
# SQ counters of the int8 GEMM kernels (north star: "rocprof MFMA-utilisation"): separate --pmc passes (never combined with a trace
# domain) over tools/gemm_block_shapes.py = the ten GEMMs of one cfg-B block, once with the ping-pong kernel and once with the
# round-3 persistent kernel (WANQ_GEMM_PP=0).   tools/gemm_sq_counters.sh <tag> [code]   ->  gpurun_out/<tag>_gemm_sq.csv
set -o pipefail
TAG=${1:-r04_x}
CODE=${2:-unknown}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES"
P2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU"
P3="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P4="GRBM_GUI_ACTIVE"
for V in pp v2; do
  if [ $V = v2 ]; then export WANQ_GEMM_PP=0; else unset WANQ_GEMM_PP; fi
  i=0
  for P in "$P1" "$P2" "$P3" "$P4"; do
    i=$((i + 1))
    timeout -k 10 300 rocprofv3 --pmc $P --kernel-include-regex "gemm_w8a8" --output-format csv -d "$OUT/${TAG}_${V}_sq_pass$i" -o pmc -- python3 tools/gemm_block_shapes.py \
      > "$OUT/${TAG}_${V}_sq_pass$i.log" 2>&1 || { echo "$V pass $i failed"; tail -5 "$OUT/${TAG}_${V}_sq_pass$i.log"; exit 1; }
  done
done
python3 tools/gemm_sq_summary.py "$OUT" "$TAG" "$CODE" > "$OUT/${TAG}_gemm_sq.csv" && cat "$OUT/${TAG}_gemm_sq.csv"
