"""The documents cite files (profiles, tools, tests, fixtures) as evidence: every cited path must exist in the tree."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "profiles/HISTORY.md", "profiles/PARITY_NOTES.md", "tools/README.md",
        "tests/golden/README.md"]
PKG = "wan2.1-quantization_amd"


def _expand(p):
    m = re.search(r"\{([^{}]*)\}", p)
    if not m:
        return [p]
    return [q for alt in m.group(1).split(",") for q in _expand(p[:m.start()] + alt + p[m.end():])]


def _exists(rel):
    rel = rel.split("::")[0].rstrip(".,")
    for base in (ROOT, os.path.join(ROOT, PKG)):          # quant_configs/..., wan/..., csrc/... are cited relative to the package
        if glob.glob(os.path.join(base, rel)) or glob.glob(os.path.join(base, rel + "*")):
            return True
    return False


def test_every_cited_path_exists():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for m in re.finditer(r"`([A-Za-z0-9_./*{},\-]+)`", text):
            t = m.group(1)
            if "rNN" in t or "<" in t:
                continue                                   # placeholders
            if re.match(r"^r0\d_[A-Za-z0-9_.*{},\-]+$", t):
                t = "profiles/" + t                        # profile files are cited by bare name
            elif not re.match(r"^(profiles|tools|tests|oracle|include|quant_configs|csrc)/", t):
                continue
            missing += [(doc, p) for p in _expand(t) if not _exists(p)]
    assert not missing, missing


def test_reference_citations_name_real_files_and_lines():
    """Every `file:line` citation of the reference -- the header's "replaces ..." notes, the product's and the oracle's docstrings, the
    documents -- names a file of the reference tree and lines inside it (tools/check_citations.py).  Needs /root/reference: skipped on a
    box without it (the GPU box)."""
    import subprocess
    import sys

    import pytest

    if not os.path.isdir("/root/reference"):
        pytest.skip("the reference tree is not on this box")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_citations.py"), "--all"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert int(r.stdout.split()[0]) > 300
