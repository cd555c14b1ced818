#!/usr/bin/env python3
"""Build libwanq_hip.so (the C-ABI hot-path library) for gfx950 with hipcc, in-tree.

    python wan2.1-quantization_amd/build.py [--force] [-j N]

Each csrc/*.hip is compiled to build/<name>.o (skipped when the object is newer than the source and
every header), then linked into lib/libwanq_hip.so.  No GPU is needed: hipcc cross-compiles.
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
