#!/usr/bin/env python3
"""Golden vectors for the two samplers of the denoising loop, from the reference's OWN scheduler files.

Run ONLY in the build container (where /root/reference exists):    python tests/golden/make_golden_schedulers.py

ViDiT-Q/examples/Wan2.1/wan/utils/fm_solvers_unipc.py (FlowUniPCMultistepScheduler) and fm_solvers.py
(FlowDPMSolverMultistepScheduler, get_sampling_sigmas) are loaded as stand-alone modules -- the `wan` package itself needs
flash_attn / xfuser / easydict -- with the stand-in of tests/golden/gen/diffusers/ for the configuration mixins they inherit
(plumbing only; every line of arithmetic is the reference's).  Each scheduler is driven exactly as text2video.py:215-269 drives
it (UniPC: set_timesteps(N, shift=5); DPM++: sigmas = get_sampling_sigmas(N, shift) through retrieve_timesteps' path) on seeded
random model outputs; timesteps, sigmas and every step's latent are stored in tests/golden/sched_*.npz."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ViDiT-Q/examples/Wan2.1/wan/utils"
sys.path.insert(0, os.path.join(HERE, "gen"))


def load(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def drive(sched, timesteps, shape, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g)
    xs, outs = [x.clone()], []
    for t in timesteps:
        out = torch.randn(shape, generator=g)
        outs.append(out)
        x = sched.step(out, t, x, return_dict=False)[0]
        xs.append(x.clone())
    return torch.stack(xs).numpy(), torch.stack(outs).numpy()


def main():
    torch.set_grad_enabled(False)
    uni, dpm = load("fm_solvers_unipc"), load("fm_solvers")
    shape = (4, 2, 6, 5)
    for n in (3, 10, 50):  # 3: every lower-order rule fires; 50: the CLI default
        s = uni.FlowUniPCMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)  # text2video.py:216-219
        s.set_timesteps(n, device="cpu", shift=5.0)
        xs, outs = drive(s, s.timesteps, shape, 100 + n)
        np.savez_compressed(os.path.join(HERE, f"sched_unipc_{n}.npz"), timesteps=s.timesteps.numpy(), sigmas=s.sigmas.numpy(), x=xs, model_out=outs)
        d = dpm.FlowDPMSolverMultistepScheduler(num_train_timesteps=1000, shift=1, use_dynamic_shifting=False)  # text2video.py:224-232
        sig = dpm.get_sampling_sigmas(n, 5.0)
        d.set_timesteps(device="cpu", sigmas=sig)  # what retrieve_timesteps(scheduler, device, sigmas=...) calls
        xs, outs = drive(d, d.timesteps, shape, 200 + n)
        np.savez_compressed(os.path.join(HERE, f"sched_dpmpp_{n}.npz"), timesteps=d.timesteps.numpy(), sigmas=d.sigmas.numpy(),
                            sampling_sigmas=np.asarray(sig), x=xs, model_out=outs)
        print(n, "unipc t[:3]", s.timesteps[:3].tolist(), "dpm++ t[:3]", d.timesteps[:3].tolist())


if __name__ == "__main__":
    main()
