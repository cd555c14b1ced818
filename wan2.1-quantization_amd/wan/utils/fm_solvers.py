"""Flow-matching sampling schedule and the per-step latent update for the denoising loop.

Timestep schedule = FlowUniPCMultistepScheduler.set_timesteps (ViDiT-Q/examples/Wan2.1/wan/utils/
fm_solvers_unipc.py:182-211): sigmas_train = 1 - linspace(1, 1/1000, 1000)[::-1] with
constructor shift 1, then sigma = linspace(sigma_max, sigma_min, N+1)[:-1], sigma' = s*sigma/(1+(s-1)*sigma),
t = int64(sigma' * 1000), final sigma 0.

Update rule: first-order flow-matching step x <- x + (sigma_next - sigma) * v  (the order-1 predictor that
the reference's multistep UniPC/DPM++ solvers reduce to at their first step; the higher-order
predictor-corrector of fm_solvers_unipc.py is a host-side elementwise refinement on a 2 MB latent and is
listed as 'next' in SURVEY 8f.3).
"""
import numpy as np
import torch


class FlowMatchScheduler:
    def __init__(self, num_train_timesteps=1000, shift=1.0):
        self.num_train_timesteps = num_train_timesteps
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sigmas = 1.0 - alphas
        self.sigmas_train = shift * sigmas / (1 + (shift - 1) * sigmas)
        self.sigma_max, self.sigma_min = float(self.sigmas_train[0]), float(self.sigmas_train[-1])  # descending: 0.999 .. 0 (reference :131-132)
        self.timesteps, self.sigmas = None, None
        self._i = 0

    def set_timesteps(self, num_inference_steps, device=None, shift=5.0):
        s = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1)[:-1]
        s = shift * s / (1 + (shift - 1) * s)
        self.timesteps = torch.from_numpy((s * self.num_train_timesteps).astype(np.int64)).to(device)
        self.sigmas = [float(v) for v in np.concatenate([s, [0.0]])]  # host floats: no device sync per step
        self._i = 0

    def step(self, model_output, sample):
        d = self.sigmas[self._i + 1] - self.sigmas[self._i]
        self._i += 1
        return torch.add(sample, model_output, alpha=d)
