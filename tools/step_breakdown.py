#!/usr/bin/env python3
"""Per-kernel time inside the LAST denoising step of a `rocprofv3 --kernel-trace --output-format csv` run of bench.py
(the trace also holds model construction, calibration and warm-up: the window is the last `--ms` milliseconds of the trace,
= the step time the bench line reports).  usage: step_breakdown.py <..._kernel_trace.csv> --ms 482 [--title "..."]"""
import argparse
import csv
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--ms", type=float, required=True)
ap.add_argument("--title", default="")
a = ap.parse_args()
rows = []
with open(a.csv) as fh:
    for r in csv.DictReader(fh):
        rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
end = max(r[2] for r in rows)
t0 = end - int(a.ms * 1e6)
agg = defaultdict(lambda: [0, 0])
busy = 0
for n, s, e in rows:
    if e <= t0:
        continue
    s = max(s, t0)
    agg[n][0] += 1
    agg[n][1] += e - s
    busy += e - s
# union of the kernels' intervals (on two streams kernels overlap: the sum of durations is then more than the time the GPU is busy)
iv = sorted((max(s, t0), e) for _, s, e in rows if e > t0)
union, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        union += (cur_e - cur_s) if cur_e is not None else 0
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += (cur_e - cur_s) if cur_e is not None else 0
print(f"Per-kernel time inside the last denoising step ({a.title}): window = last {a.ms:.0f} ms of the trace; GPU busy {busy / 1e6:.1f} ms of it"
      f" (sum of kernel durations; union of their intervals {union / 1e6:.1f} ms, i.e. {a.ms - union / 1e6:.1f} ms with no kernel running).\n")
print(f"{'kernel':86s}{'calls':>6s}{'total ms':>10s}{'share':>7s}{'avg us':>10s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{n[:84]:86s}{c:6d}{t / 1e6:10.2f}{100.0 * t / busy:6.1f}%{t / c / 1e3:10.1f}")
