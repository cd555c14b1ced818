import os
import sys

from .parallel import ParallelPlan, SeqParallel  # noqa: F401


def enter_one_gpu_rehearsal(env_var, world):
    """True when this process must rehearse the multi-rank control flow on a ONE-GPU box (every rank on cuda:0, gloo rendezvous,
    collectives staged through host memory by tools/one_gpu_rehearsal.py -- test scaffolding that lives outside this package).
    The switch is `env_var`=1 in the environment; it is REFUSED (exit status 4, one line) unless exactly one GPU is visible, so
    that a variable left set on a real node cannot silently turn an RCCL run into gloo + host copies."""
    if world <= 1 or os.environ.get(env_var) != "1":
        return False
    import torch

    n = torch.cuda.device_count()
    if n != 1:
        print(f"{env_var}=1 is the ONE-GPU rehearsal switch (gloo + host-staged collectives, never a measurement); this box shows "
              f"{n} GPUs: refused.  Unset it to run on RCCL.", file=sys.stderr, flush=True)
        sys.exit(4)
    path = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tools", "one_gpu_rehearsal.py"))
    if not os.path.exists(path):
        print(f"{env_var}=1: the rehearsal scaffolding {path} is not part of this installation", file=sys.stderr, flush=True)
        sys.exit(4)
    print(f"*** {env_var}=1: ONE-GPU REHEARSAL -- {world} ranks share cuda:0, gloo rendezvous, collectives staged through host "
          f"memory.  NOT a measurement, NOT the product path. ***", file=sys.stderr, flush=True)
    return True


def stage_rehearsal_collectives():
    """Load tools/one_gpu_rehearsal.py by path and install its host-staged collectives (after init_process_group("gloo"))."""
    import importlib.util

    path = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", "tools", "one_gpu_rehearsal.py"))
    spec = importlib.util.spec_from_file_location("wanq_one_gpu_rehearsal", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.stage_collectives_through_host()
