#!/bin/bash
# background sampler: power / sclk / temperature of GPU 0 every 0.5 s -> $1  (kill it by PID when done)
out=$1
while true; do
  echo "$(date +%s.%N) $(rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n')" >> "$out"
  sleep 0.5
done
