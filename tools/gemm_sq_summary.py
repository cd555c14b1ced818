#!/usr/bin/env python3
"""Merge the --pmc passes of tools/gemm_sq_counters.sh into one row per GEMM dispatch and kernel variant + derived ratios.
usage: gemm_sq_summary.py <gpurun_out dir> <tag> <code>"""
import csv
import glob
import os
import sys
from collections import OrderedDict, defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
NAMES = ["self_attn.q", "self_attn.k", "self_attn.v", "self_attn.o", "cross_attn.q", "cross_attn.k", "cross_attn.v", "cross_attn.o", "ffn.0", "ffn.2"]
OPS = {"self_attn.q": 2 * 32760 * 1536 * 1536, "cross_attn.k": 2 * 512 * 1536 * 1536, "ffn.0": 2 * 32760 * 8960 * 1536}


def main():
    out, tag, code = sys.argv[1], sys.argv[2], sys.argv[3]
    print(f"# SQ counters of the ten GEMMs of one cfg-B block (tools/gemm_block_shapes.py, uniform random int8), code {code}")
    print("# variant pp = gemm_w8a8_pp_kernel (ping-pong persistent), v2 = round-3 persistent kernel (WANQ_GEMM_PP=0)")
    print("# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs;")
    print("# SQ_BUSY_CYCLES is summed over the 8 XCDs' SQs (x 4 shader engines each): mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)")
    w = csv.writer(sys.stdout)
    head = None
    for var in ("pp", "v2"):
        rows = defaultdict(OrderedDict)
        for d in sorted(glob.glob(os.path.join(out, f"{tag}_{var}_sq_pass*"))):
            if not os.path.isdir(d):
                continue
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                per = defaultdict(lambda: defaultdict(float))
                names = {}
                with open(f) as fh:
                    for r in csv.DictReader(fh):
                        did = int(r["Dispatch_Id"])
                        per[did][r["Counter_Name"]] += float(r["Counter_Value"])
                        names[did] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void wanq::", "").split("(")[0]
                for k, did in enumerate(sorted(per)):
                    rows[k]["kernel"] = names[did]
                    for c, v in per[did].items():
                        rows[k][c] = v
        cols = []
        for r in rows.values():
            for c in r:
                if c not in cols and c != "kernel":
                    cols.append(c)
        if head is None:
            head = cols
            w.writerow(["variant", "launch", "kernel"] + cols + ["mfma_busy", "wait_frac", "issue_stall_frac", "mfma_per_valu", "lds_conflict_frac"])
        for k in sorted(rows):
            r = rows[k]
            g = lambda c: r.get(c, float("nan"))  # noqa: E731
            gui = g("GRBM_GUI_ACTIVE") / 8.0
            w.writerow([var, NAMES[k] if k < len(NAMES) else k, r["kernel"]] + [f"{r.get(c, float('nan')):.5g}" for c in head] +
                       [f"{g('SQ_VALU_MFMA_BUSY_CYCLES') / (gui * 1024):.4f}", f"{g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.4f}",
                        f"{g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.4f}", f"{g('SQ_INSTS_MFMA') / g('SQ_INSTS_VALU'):.4f}",
                        f"{g('SQ_LDS_BANK_CONFLICT') / max(1.0, g('SQ_LDS_IDX_ACTIVE')):.4f}"])


if __name__ == "__main__":
    main()
