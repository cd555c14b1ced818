#!/usr/bin/env python3
"""Every `file:line` citation of the reference in include/wanq_hip.h (and, with --all, in the product's docstrings and the oracle) must
name a file that exists under the reference tree and a line range inside it.  Path shorthand as in SURVEY.md: Q/ = ViDiT-Q/quant_utils/
qdiff/, K/ = ViDiT-Q/kernels/, W/ = ViDiT-Q/examples/Wan2.1/ (XF/ = inside a vendored tarball: skipped).  Needs the reference tree
(--ref, default /root/reference); exits 0 with a note when it is absent (the GPU box)."""
import argparse
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PREFIX = {"Q/": "ViDiT-Q/quant_utils/qdiff/", "K/": "ViDiT-Q/kernels/", "W/": "ViDiT-Q/examples/Wan2.1/"}
CITE = re.compile(r"(?<![A-Za-z0-9_./\-])((?:[A-Za-z0-9_\-]+/)*[A-Za-z0-9_\-]+\.(?:py|cu|cuh|cpp|h|yaml|yml|sh)):(\d+)(?:-(\d+))?")
MORE = re.compile(r",\s?(\d+)(?:-(\d+))?(?![\d.])")
_INDEX = {}


def ref_index(ref):
    """basename -> paths (relative to the reference root), built once"""
    if ref not in _INDEX:
        idx = {}
        for d, _, names in os.walk(ref):
            for n in names:
                idx.setdefault(n, []).append(os.path.relpath(os.path.join(d, n), ref))
        _INDEX[ref] = idx
    return _INDEX[ref]


def candidates(ref, path):
    """files of the reference a citation can mean: the shorthand prefixes expand to one path; anything else (wan/quant_wanx_cuda.py,
    model.py ...) is matched as a path SUFFIX, and a citation is good when some match has the cited lines.  None = the name does not
    occur in the reference at all but does in this repository (a citation of our own file: not checked here)."""
    for short, full in PREFIX.items():
        if path.startswith(short):
            return [os.path.join(ref, full + path[len(short):])]
    if path.startswith("ViDiT-Q/"):
        return [os.path.join(ref, path)]
    hits = [os.path.join(ref, r) for r in ref_index(ref).get(os.path.basename(path), []) if ("/" + r).endswith("/" + path)]
    if not hits and glob.glob(os.path.join(ROOT, "**", os.path.basename(path)), recursive=True):
        return None
    return hits


def check_file(ref, src):
    bad, n = [], 0
    text = open(src, errors="replace").read()
    cites = []
    for m in CITE.finditer(text):
        cites.append((m.group(0), m.group(1), int(m.group(2)), int(m.group(3) or m.group(2))))
        for more in MORE.finditer(text, m.end()):      # "file.py:22-26, 69-857" / ":313-328,395-404": further ranges of the same file
            if more.start() != (cites[-1][4] if len(cites[-1]) > 4 else m.end()):
                break
            cites[-1] = cites[-1][:4] + (more.end(),)
            cites.append((m.group(1) + ":" + more.group(0).lstrip(", "), m.group(1), int(more.group(1)), int(more.group(2) or more.group(1)), more.end()))
    for shown, path, a, b, *_ in cites:
        cands = candidates(ref, path)
        if cands is None:
            continue
        n += 1
        cands = [c for c in cands if os.path.isfile(c)]
        if not cands:
            bad.append(f"{os.path.relpath(src, ROOT)}: {shown}: no such file in the reference")
            continue
        lines = max(sum(1 for _ in open(c, errors="replace")) for c in cands)
        if not (1 <= a <= b <= lines):
            bad.append(f"{os.path.relpath(src, ROOT)}: {shown}: the file has {lines} lines")
    return n, bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--all", action="store_true", help="also the product package, the oracle and the golden generators")
    a = ap.parse_args()
    if not os.path.isdir(a.ref):
        print(f"{a.ref} absent: nothing checked")
        return 0
    files = [os.path.join(ROOT, "include", "wanq_hip.h")]
    if a.all:
        for pat in ("wan2.1-quantization_amd/**/*.py", "wan2.1-quantization_amd/csrc/*", "oracle/*.py", "tests/golden/*.py", "tests/*.py",
                    "DESIGN.md", "INTEGRATION.md", "README.md", "bench.py"):
            files += [f for f in glob.glob(os.path.join(ROOT, pat), recursive=True) if os.path.isfile(f)]
    total, bad = 0, []
    for f in files:
        n, b = check_file(a.ref, f)
        total += n
        bad += b
    print(f"{total} citations in {len(files)} files, {len(bad)} bad")
    for b in bad:
        print("  " + b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
