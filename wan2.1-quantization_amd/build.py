#!/usr/bin/env python3
"""Build libwanq_hip.so (the C-ABI hot-path library) for gfx950 with hipcc, in-tree.

    python wan2.1-quantization_amd/build.py [--force] [-j N]

Each csrc/*.hip is compiled to build/<name>.o (skipped when the object is newer than the source and
every header), then linked into lib/libwanq_hip.so.  No GPU is needed: hipcc cross-compiles.

Every freshly compiled object goes through tools/isa_lint.py (disassembly of its gfx950 code object: the packed-fp32
op_sel hazard of DESIGN.md 3.5, no spill inside the ping-pong GEMM's K loops, its store / LDS-DMA counts); an error fails
the build before the library is linked.  WANQ_BUILD_LINT=0 skips it (A/B builds of deliberately different kernels).
"""
import argparse
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libwanq_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wno-unused-result"]


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def _compile(src, obj):
    cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {os.path.basename(src)}:\n{r.stdout}\n{r.stderr}")
    return os.path.basename(src)


def _lint(objs, verbose):
    import importlib.util
    spec = importlib.util.spec_from_file_location("wanq_isa_lint", os.path.join(HERE, "..", "tools", "isa_lint.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    errors, lines, _ = lint.lint_objects(objs)
    if verbose:
        for ln in lines:
            print(f"[wanq build] isa_lint {ln}", flush=True)
    if errors:
        for o in objs:  # so that the next build compiles (and lints) them again
            if os.path.exists(o):
                os.remove(o)
        raise RuntimeError("isa_lint failed (tools/isa_lint.py):\n" + "\n".join(errors))


def build(force=False, jobs=4, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + sorted(
        glob.glob(os.path.join(HERE, "..", "include", "*.h")))
    todo, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ, os.path.splitext(os.path.basename(s))[0] + ".o")
        objs.append(o)
        if force or not _newer(o, [s, *hdrs]):
            todo.append((s, o))
    if todo:
        with cf.ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
            for name in ex.map(lambda so: _compile(*so), todo):
                if verbose:
                    print(f"[wanq build] compiled {name}", flush=True)
        if os.environ.get("WANQ_BUILD_LINT", "1") != "0":
            _lint([o for _, o in todo], verbose)
    if todo or not _newer(LIB, objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[wanq build] linked {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("-j", type=int, default=4)
    a = ap.parse_args()
    try:
        print(build(a.force, a.j))
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
