#!/bin/bash
# alternating whole-step A/B of two full libraries on one box
for i in 1 2 3; do
  for v in before pk; do
    WANQ_LIB=wan2.1-quantization_amd/lib/variants/lib_full_$v.so python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-quality 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', round(d['ms_per_step'],2), 'attn', round(d['roofline']['avg_launch_us'],1), 'gemm', round(d['roofline_second_kernel']['avg_launch_us'],2), round(d['roofline_second_kernel']['frac'],4))"
  done
done
