#!/bin/bash
# alternating runs on one box: one stream vs two streams for the two passes of a step
for r in 1 2 3; do
  for s in 1 2; do
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-quality --pass-streams $s 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass-streams $s: %.1f ms/step  (attention %.0f us, gemm %.1f us)  instrumented %.1f' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_second_kernel']['avg_launch_us'], d['instrumented_ms_per_step']))"
  done
done
