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
        return sample + d * model_output


def get_sampling_sigmas(sampling_steps, shift):
    """sigma = linspace(1, 0, N+1)[:N] shifted (reference fm_solvers.py:22-26): the `--sample_solver dpm++` schedule."""
    sigma = np.linspace(1, 0, sampling_steps + 1)[:sampling_steps]
    return shift * sigma / (1 + (shift - 1) * sigma)


class FlowDPMSolverMultistepScheduler:
    """`--sample_solver dpm++`: DPM-Solver++ (2M, midpoint), data-prediction form, for flow-matching models
    (ViDiT-Q/examples/Wan2.1/wan/utils/fm_solvers.py:69-857 -- diffusers' DPMSolverMultistepScheduler with the flow
    parameterisation alpha = 1 - sigma; constructor defaults solver_order 2, algorithm_type "dpmsolver++", solver_type
    "midpoint", lower_order_final, final_sigmas_type "zero").  Host restatement like the UniPC one: float64 scalars, the update
    is linear in latent-sized tensors.  Restated lines: convert_model_output (flow_prediction) :341-395, first-order update
    :415-484, second-order multistep update :486-594, step control flow :706-800; sigma schedule :22-26 + :226-290.
    Parity: PINNED -- tests/golden/sched_dpmpp_{3,10,50}.npz hold trajectories produced by the reference's own fm_solvers.py
    (tests/golden/make_golden_schedulers.py loads the file stand-alone with a stand-in for the diffusers configuration mixins);
    tests/test_schedulers_cpu.py compares timesteps exactly and every latent to 2e-5, besides the closed-form properties."""

    def __init__(self, num_train_timesteps=1000, solver_order=2, shift=1.0, lower_order_final=True):
        assert solver_order in (1, 2), "orders 1 and 2 are implemented (the reference's default is 2)"
        self.num_train_timesteps, self.order, self.lower_order_final = num_train_timesteps, solver_order, lower_order_final
        self.timesteps = self.sigmas = None

    def set_timesteps(self, num_inference_steps, device=None, shift=5.0, sigmas=None):
        s = np.asarray(get_sampling_sigmas(num_inference_steps, shift) if sigmas is None else sigmas, dtype=np.float64)
        self.timesteps = torch.from_numpy((s * self.num_train_timesteps).astype(np.int64)).to(device)  # :271-277
        self.sigmas = [float(np.float32(v)) for v in s] + [0.0]  # the reference keeps sigmas in fp32 (:272-275)
        self._i, self._m, self._lower = 0, [], 0

    @staticmethod
    def _lam(sigma):
        import math
        if sigma <= 0.0:
            return math.inf
        if sigma >= 1.0:  # the dpm++ schedule starts at sigma = 1: alpha = 0, log(0) = -inf as torch evaluates it
            return -math.inf
        return math.log(1.0 - sigma) - math.log(sigma)

    def step(self, model_output, timestep=None, sample=None):
        import math
        i, n = self._i, len(self.timesteps)
        lower_final = i == n - 1                                  # final_sigmas_type == "zero" (:746-749)
        lower_second = i == n - 2 and self.lower_order_final and n < 15
        m0 = sample - self.sigmas[i] * model_output               # x0 prediction (:381-384)
        self._m.append(m0)
        if len(self._m) > self.order:
            self._m.pop(0)
        st, s0 = self.sigmas[i + 1], self.sigmas[i]
        at = 1.0 - st
        h = self._lam(st) - self._lam(s0)
        em1 = math.expm1(-h) if math.isfinite(h) else -1.0        # exp(-h) - 1
        if self.order == 1 or self._lower < 1 or lower_final:
            prev = (st / s0) * sample - (at * em1) * m0           # first order (:465-468)
        else:
            # second order needs two model outputs; `lower_second` only matters for order 3 (:779-784)
            h0 = self._lam(s0) - self._lam(self.sigmas[i - 1])
            r0 = h0 / h
            d1 = (1.0 / r0) * (m0 - self._m[-2])
            prev = (st / s0) * sample - (at * em1) * m0 - (0.5 * at * em1) * d1   # midpoint (:550-553)
        if self._lower < self.order:
            self._lower += 1
        self._i += 1
        return prev
