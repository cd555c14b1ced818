#!/usr/bin/env python3
"""Merge the --pmc passes of tools/attn_sq_counters.sh into one row per attention dispatch + derived ratios.
usage: attn_sq_summary.py <gpurun_out dir> <tag> <code>"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict


def main():
    out, tag, code = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = defaultdict(OrderedDict)  # (kernel, ordinal within pass) -> counter -> value
    for d in sorted(glob.glob(os.path.join(out, f"{tag}_sq_pass*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = defaultdict(lambda: defaultdict(float))  # dispatch id -> counter -> sum over dimensions
            names = {}
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    did = int(r["Dispatch_Id"])
                    per[did][r["Counter_Name"]] += float(r["Counter_Value"])
                    names[did] = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void wanq::", "")
            for k, did in enumerate(sorted(per)):
                for c, v in per[did].items():
                    rows[(names[did], k)][c] = v
    cols = []
    for r in rows.values():
        for c in r:
            if c not in cols:
                cols.append(c)
    derived = ["mfma_busy_frac = VALU_MFMA_BUSY / (4 * BUSY_CYCLES-per-CU-normalised: see note)", ]
    print(f"# SQ counters, attention kernels of tools/attn_once.py (dispatch 0 = self-attention 32760x32760x12, 1 = cross 32760x512x12); code {code}")
    print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs")
    print("# derived: mfma_per_valu = SQ_INSTS_MFMA / SQ_INSTS_VALU; wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES;")
    print("#          mfma_busy_per_simd_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs / 8 XCD-replicated count): see DESIGN 3.2 for the reading")
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "dispatch"] + cols + ["mfma_per_valu", "wait_frac", "issue_stall_frac", "active_frac", "lds_conflict_frac"])
    for (name, k), r in sorted(rows.items(), key=lambda kv: kv[0][1]):
        g = lambda c: r.get(c, float("nan"))  # noqa: E731
        wc = g("SQ_WAVE_CYCLES")
        w.writerow([name, k] + [f"{r.get(c, float('nan')):.4g}" for c in cols] +
                   [f"{g('SQ_INSTS_MFMA') / g('SQ_INSTS_VALU'):.4f}", f"{g('SQ_WAIT_ANY') / wc:.4f}", f"{g('SQ_WAIT_INST_ANY') / wc:.4f}",
                    f"{g('SQ_ACTIVE_INST_ANY') / wc:.4f}", f"{g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.4f}"])


if __name__ == "__main__":
    main()
