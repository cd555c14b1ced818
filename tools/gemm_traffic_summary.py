#!/usr/bin/env python3
"""rocprofv3 counter CSVs of tools/gemm_block_shapes.py -> per-launch HBM traffic table + the per-launch average that
bench.py reports as roofline.traffic.  FETCH_SIZE is in KiB and needs the x2 gfx950 correction
(/opt/skills/guides/MI355X_MICROARCH.md, HBM / rocprofv3 section); WRITE_SIZE is in KiB.

usage: gemm_traffic_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out prefix> [code tag]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_block_shapes import LAUNCHES  # noqa: E402


def counter(path, name):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == name]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [float(r["Counter_Value"]) for r in rows]


def main():
    fetch = counter(sys.argv[1], "FETCH_SIZE")
    write = counter(sys.argv[2], "WRITE_SIZE")
    assert len(fetch) == len(write) == len(LAUNCHES), (len(fetch), len(write))
    esz = {"torch.bfloat16": 2, "torch.float16": 2, "torch.float32": 4}
    out_rows, tot_hbm, tot_alg = [], 0.0, 0.0
    for (name, M, N, K, odt, gelu, res), f, w in zip(LAUNCHES, fetch, write):
        e = esz[str(odt)]
        alg_r = M * K + N * K + (M * N * e if res else 0)
        alg_w = M * N * e
        fr, wr = f * 1024 * 2, w * 1024
        out_rows.append([name, M, N, K, str(odt).split(".")[1], int(gelu), int(res), round(alg_r / 1e6, 1), round(fr / 1e6, 1),
                         round(alg_w / 1e6, 1), round(wr / 1e6, 1)])
        tot_hbm += fr + wr
        tot_alg += alg_r + alg_w
    pre = sys.argv[3]
    with open(pre + ".csv", "w", newline="") as fh:
        wtr = csv.writer(fh)
        wtr.writerow(["launch", "M", "N", "K", "out", "gelu", "gate_res", "algorithmic_read_MB", "FETCH_SIZE_x2_MB",
                      "algorithmic_write_MB", "WRITE_SIZE_MB"])
        wtr.writerows(out_rows)
    summary = {"kernel": "gemm_w8a8_pp_kernel (fp32 + gate + residual, 16-bit) / gemm_w8a8_big_kernel", "launches": len(LAUNCHES),
               "hbm_bytes_per_launch": tot_hbm / len(LAUNCHES), "algorithmic_bytes_per_launch": tot_alg / len(LAUNCHES),
               # which bench.py workload the launches belong to (bench.py reports the figure only for that workload) and which
               # code they were taken on
               "workload": "t2v-1.3B 832*480 81f n1", "code": sys.argv[4] if len(sys.argv) > 4 else "unknown",
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over tools/gemm_block_shapes.py, "
                         "FETCH_SIZE x2 (gfx950), KiB units; the ten GEMMs of one cfg-B block"}
    json.dump(summary, open(pre + ".json", "w"), indent=1)
    print(json.dumps(summary))


if __name__ == "__main__":
    main()
