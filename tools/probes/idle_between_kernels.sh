#!/bin/bash
# How much of a TIMED step (no event pairs around the launches) has no kernel running, with the two passes of a step on one stream and on
# two: rocprofv3 --kernel-trace of `bench.py --no-instrumented-repeat`, union of the kernels' intervals inside the last step window.
# usage (GPU box): bash tools/probes/idle_between_kernels.sh <out file>
OUT=${1:-gpurun_out/idle_between_kernels.txt}
export TMPDIR=/tmp
D=$PWD/gpurun_out/idle_probe; rm -rf "$D"; mkdir -p "$D"
: > "$OUT"
for S in 1 2; do
  timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d "$D/s$S" -o t -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-quality --no-instrumented-repeat --pass-streams $S > "$D/s$S.log" 2>&1 || { tail -5 "$D/s$S.log"; exit 1; }
  MS=$(python3 -c "import json; print(json.loads([l for l in open('$D/s$S.log') if l.startswith('{')][-1])['ms_per_step'])")
  TRACE=$(find "$D/s$S" -name "*kernel_trace.csv" | head -1)
  echo "== --pass-streams $S: ${MS} ms per step (under the tracer)" >> "$OUT"
  python3 tools/step_breakdown.py "$TRACE" --ms "$MS" --title "pass-streams $S" | head -8 >> "$OUT"
  rm -f "$TRACE"
done
cat "$OUT"
