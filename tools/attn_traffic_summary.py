#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/attn_once.py -> profiles/*_attention_traffic.{csv,json} (same conventions as
tools/gemm_traffic_summary.py: FETCH_SIZE KiB x2 on gfx950, WRITE_SIZE KiB).

usage: attn_traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out prefix> [code tag]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_once import H, LAUNCHES  # noqa: E402
from gemm_traffic_summary import counter  # noqa: E402


def main():
    fetch, write = counter(sys.argv[1], "FETCH_SIZE"), counter(sys.argv[2], "WRITE_SIZE")
    assert len(fetch) == len(write) == len(LAUNCHES)
    rows, tot_hbm, tot_alg = [], 0.0, 0.0
    for (name, Lq, Lk), f, w in zip(LAUNCHES, fetch, write):
        alg_r, alg_w = (Lq + 2 * Lk) * H * 128 * 2, Lq * H * 128 * 2
        fr, wr = f * 1024 * 2, w * 1024
        rows.append([name, Lq, Lk, H, round(alg_r / 1e6, 1), round(fr / 1e6, 1), round(alg_w / 1e6, 1), round(wr / 1e6, 1)])
        tot_hbm += fr + wr
        tot_alg += alg_r + alg_w
    pre = sys.argv[3]
    with open(pre + ".csv", "w", newline="") as fh:
        wtr = csv.writer(fh)
        wtr.writerow(["launch", "Lq", "Lk", "heads", "algorithmic_read_MB", "FETCH_SIZE_x2_MB", "algorithmic_write_MB", "WRITE_SIZE_MB"])
        wtr.writerows(rows)
    summary = {"kernel": "attn_fwd16_kernel<false, false> (the default bf16 kernel; WANQ_ATTN_M16=0 would run attn_fwd_kernel)", "launches": len(LAUNCHES), "hbm_bytes_per_launch": tot_hbm / len(LAUNCHES),
               "algorithmic_bytes_per_launch": tot_alg / len(LAUNCHES),
               # which bench.py workload the launches belong to (bench.py reports the figure only for that workload) and which
               # code they were taken on
               "workload": "t2v-1.3B 832*480 81f n1", "code": sys.argv[4] if len(sys.argv) > 4 else "unknown",
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/attn_once.py, FETCH_SIZE x2 "
                         "(gfx950), KiB units; one self-attention + one cross-attention launch of a cfg-B block"}
    json.dump(summary, open(pre + ".json", "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
