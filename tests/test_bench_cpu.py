"""bench.py's own launcher, on the CPU: `python bench.py --gpus N` with no WORLD_SIZE must start the N ranks itself (as
children of a parent that never touches the GPU) and exit non-zero when a rank fails.  Here every rank fails -- there is
no GPU in this container and the hot path has no CPU fallback -- which is exactly the failure the parent has to report."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_reports_a_failed_rank():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without a GPU (the GPU form is tests/test_gpu_sp_rehearsal.py, launcher 'self')")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr  # raised by the RANKS (children); the parent made no GPU call
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no JSON line from a failed run


def test_bench_refuses_a_world_size_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
