#!/bin/bash
# Everything profiles/rNN_<tag>_* is made from, in one gpurun call:  tools/profile_round.sh r02_b <code tag>
#   1. full pytest -m gpu                      -> gpurun_out/<tag>_tests.log
#   2. bench.py (default flags)                -> gpurun_out/<tag>_bench_line.txt
#   3. rocprofv3 --kernel-trace --stats of a 2-step bench.py  -> gpurun_out/<tag>_prof/  (+ step breakdown)
#   4. PMC passes (FETCH_SIZE, WRITE_SIZE separately; never together with a trace domain) over the GEMM and attention
#      launches of one cfg-B block            -> gpurun_out/<tag>_{gemm,attention}_traffic.{csv,json}
# Steps are joined so that a failing GPU step stops the rest.
set -o pipefail
TAG=${1:-r02_x}
CODE=${2:-unknown}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout -k 10 1000 python -m pytest tests -m gpu -x -q > "$OUT/${TAG}_tests.log" 2>&1
  rc=$?; echo "pytest rc=$rc" >> "$OUT/${TAG}_tests.log"; tail -4 "$OUT/${TAG}_tests.log"
  [ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 600 python bench.py > "$OUT/${TAG}_bench_line.txt" 2> "$OUT/${TAG}_bench.err" || { tail -5 "$OUT/${TAG}_bench.err"; exit 1; }
cat "$OUT/${TAG}_bench_line.txt"
# per-kernel statistics and the step breakdown are taken with the two passes of a step on ONE stream (--pass-streams 1): on two streams
# (what the default, auto, may pick: wan/utils/two_pass.py) kernels of the two passes share the GPU and a kernel's traced duration is no
# longer its own time
timeout -k 10 600 python bench.py --pass-streams 1 --no-cpu-baseline --no-quality > "$OUT/${TAG}_bench_line_one_stream.txt" 2>> "$OUT/${TAG}_bench.err" || { tail -5 "$OUT/${TAG}_bench.err"; exit 1; }
MS=$(python3 -c "import json,sys; print(json.loads(open('$OUT/${TAG}_bench_line_one_stream.txt').read().strip().splitlines()[-1])['ms_per_step'])")
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${TAG}_prof" -o "$TAG" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-quality --pass-streams 1 > "$OUT/${TAG}_prof_bench.log" 2>&1 || { tail -5 "$OUT/${TAG}_prof_bench.log"; exit 1; }
TRACE=$(find "$OUT/${TAG}_prof" -name "*kernel_trace.csv" | head -1)
python3 tools/step_breakdown.py "$TRACE" --ms "$MS" --title "$TAG, $CODE" > "$OUT/${TAG}_step_breakdown.txt" && head -14 "$OUT/${TAG}_step_breakdown.txt"
for W in gemm attention; do
  if [ $W = gemm ]; then DRV=tools/gemm_block_shapes.py; RE=gemm_w8a8; SUM=tools/gemm_traffic_summary.py; else DRV=tools/attn_once.py; RE=attn_fwd; SUM=tools/attn_traffic_summary.py; fi
  for CNT in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $CNT --kernel-include-regex $RE --output-format csv -d "$OUT/${TAG}_pmc_${W}_${CNT}" -o pmc -- python3 $DRV > "$OUT/${TAG}_pmc_${W}_${CNT}.log" 2>&1 || { tail -5 "$OUT/${TAG}_pmc_${W}_${CNT}.log"; exit 1; }
  done
  F=$(find "$OUT/${TAG}_pmc_${W}_FETCH_SIZE" -name "*counter_collection.csv" | head -1)
  Wf=$(find "$OUT/${TAG}_pmc_${W}_WRITE_SIZE" -name "*counter_collection.csv" | head -1)
  python3 $SUM "$F" "$Wf" "$OUT/${TAG}_${W}_traffic" "$CODE" || exit 1
done
# the trace is large: keep the stats and the breakdown, drop the per-dispatch rows from what travels back
find "$OUT/${TAG}_prof" -name "*kernel_trace.csv" -size +40M -delete
echo "profile_round $TAG done"
