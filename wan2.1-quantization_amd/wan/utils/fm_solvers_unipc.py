"""FlowUniPCMultistepScheduler: UniPC (B(h) = expm1(h), "bh2"; data-prediction form; order 2 with order-1 warm-up and
lower-order final step) for flow-matching models -- the default `--sample_solver unipc` of the reference
(ViDiT-Q/examples/Wan2.1/wan/utils/fm_solvers_unipc.py:20-800, itself derived from diffusers' UniPC).

Host-side restatement: the step coefficients are Python floats (float64), the update is a handful of elementwise torch
ops on the 2 MB latent.  Interface: set_timesteps(n, device, shift) / timesteps / step(model_output, timestep, sample).
Parity: PINNED -- tests/golden/sched_unipc_{3,10,50}.npz hold trajectories produced by the reference's own file
(tests/golden/make_golden_schedulers.py loads it stand-alone with a stand-in for the diffusers configuration mixins it inherits;
all arithmetic is the reference's); tests/test_schedulers_cpu.py compares timesteps exactly and every latent of the trajectory to
2e-5.  Also tested: an order-1 step equals the flow-matching Euler step; a constant velocity field is integrated exactly; the
schedule equals FlowMatchScheduler's."""
import math

import numpy as np
import torch


class FlowUniPCMultistepScheduler:
    def __init__(self, num_train_timesteps=1000, solver_order=2, shift=1.0, lower_order_final=True):
        assert solver_order in (1, 2), "orders 1 and 2 are implemented (the reference uses 2)"
        self.num_train_timesteps, self.order, self.lower_order_final = num_train_timesteps, solver_order, lower_order_final
        alphas = np.linspace(1, 1 / num_train_timesteps, num_train_timesteps)[::-1].copy()
        sig = 1.0 - alphas
        sig = shift * sig / (1 + (shift - 1) * sig)                       # reference :107-113
        self.sigma_max, self.sigma_min = float(sig[0]), float(sig[-1])  # descending: 0.999 .. 0 (reference :131-132)
        self.timesteps = self.sigmas = None

    def set_timesteps(self, num_inference_steps, device=None, shift=5.0):
        s = np.linspace(self.sigma_max, self.sigma_min, num_inference_steps + 1)[:-1]
        s = shift * s / (1 + (shift - 1) * s)                             # reference :182-193
        self.timesteps = torch.from_numpy((s * self.num_train_timesteps).astype(np.int64)).to(device)
        self.sigmas = [float(v) for v in s] + [0.0]                       # final_sigmas_type == "zero" (:198-199)
        self.num_inference_steps = num_inference_steps
        self._i = 0
        self._m = []            # data predictions of the last `order` steps (newest last)
        self._lower = 0         # warm-up counter (lower_order_nums)
        self._last_sample = None
        self._this_order = 1

    @staticmethod
    def _lam(sigma):
        a = 1.0 - sigma                                                   # _sigma_to_alpha_sigma_t (:272-273)
        if sigma <= 0.0:
            return math.inf
        return math.log(a) - math.log(sigma)

    def _coeffs(self, i_t, i_s0, i_prev, order):
        """Scalars shared by predictor and corrector for the move sigma[i_s0] -> sigma[i_t]."""
        st, s0 = self.sigmas[i_t], self.sigmas[i_s0]
        at = 1.0 - st
        h = self._lam(st) - self._lam(s0)
        hh = -h
        h_phi_1 = math.expm1(hh) if math.isfinite(hh) else -1.0            # e^{-h} - 1
        B_h = h_phi_1                                                     # bh2
        rk = None
        if order == 2:
            rk = (self._lam(self.sigmas[i_prev]) - self._lam(s0)) / h
        return st, s0, at, hh, h_phi_1, B_h, rk

    def _predict(self, x, order):
        """multistep_uni_p_bh_update (:354-488), predict_x0 branch."""
        i = self._i
        st, s0, at, hh, h_phi_1, B_h, rk = self._coeffs(i + 1, i, i - 1, order)
        m0 = self._m[-1]
        x_t = (st / s0) * x - (at * h_phi_1) * m0
        if order == 2:
            d1 = (self._m[-2] - m0) / rk
            x_t = x_t - (at * B_h * 0.5) * d1                             # rhos_p = 0.5 for order 2 (:459-460)
        return x_t

    def _correct(self, model_t, x_last, order):
        """multistep_uni_c_bh_update (:490-630), predict_x0 branch.  `order` is the order the predictor used."""
        i = self._i
        st, s0, at, hh, h_phi_1, B_h, rk = self._coeffs(i, i - 1, i - 2, order)
        m0 = self._m[-1]
        x_t = (st / s0) * x_last - (at * h_phi_1) * m0
        d1_t = model_t - m0
        if order == 1:
            return x_t - (at * B_h * 0.5) * d1_t                          # rhos_c = [0.5] (:607-608)
        # order 2: solve [[1, 1], [rk, 1]] rho = b,  b_k = h_phi_k * k! / B_h (:590-611)
        h_phi_k = h_phi_1 / hh - 1.0
        b0 = h_phi_k / B_h
        h_phi_k2 = h_phi_k / hh - 0.5
        b1 = h_phi_k2 * 2.0 / B_h
        det = 1.0 - rk
        rho0 = (b0 - b1) / det
        rho1 = (b1 - rk * b0) / det
        d1 = (self._m[-2] - m0) / rk
        return x_t - (at * B_h) * (rho0 * d1 + rho1 * d1_t)

    def step(self, model_output, timestep=None, sample=None):
        """One scheduler update (reference step(), :659-745).  model_output: the flow (velocity) prediction."""
        i = self._i
        x0_pred = sample - self.sigmas[i] * model_output                  # convert_model_output, flow_prediction (:303-307)
        if i > 0 and self._last_sample is not None:
            sample = self._correct(x0_pred, self._last_sample, self._this_order)
        self._m.append(x0_pred)
        if len(self._m) > self.order:
            self._m.pop(0)
        this_order = min(self.order, len(self.timesteps) - i) if self.lower_order_final else self.order
        self._this_order = min(this_order, self._lower + 1)
        self._last_sample = sample
        prev = self._predict(sample, self._this_order)
        if self._lower < self.order:
            self._lower += 1
        self._i += 1
        return prev
