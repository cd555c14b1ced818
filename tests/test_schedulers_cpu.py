"""Host-side samplers (CPU): schedule values per SURVEY Appendix C, Euler consistency, exactness on linear fields."""
import numpy as np
import torch

from wan.utils.fm_solvers import FlowMatchScheduler
from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler


def test_timestep_schedule_matches_reference_recipe():
    s = FlowUniPCMultistepScheduler(1000, shift=1.0)
    s.set_timesteps(30, shift=5.0)
    e = FlowMatchScheduler(1000, shift=1.0)
    e.set_timesteps(30, shift=5.0)
    assert torch.equal(s.timesteps, e.timesteps) and np.allclose(s.sigmas, e.sigmas)
    # sigmas_train = 1 - linspace(1, 1/1000, 1000)[::-1]; sigma' = 5 s / (1 + 4 s); t = int64(1000 sigma'); last sigma 0
    sig = np.linspace(0.999, 0.0, 31)[:-1]
    sig = 5 * sig / (1 + 4 * sig)
    assert np.allclose(s.sigmas[:-1], sig) and s.sigmas[-1] == 0.0
    assert s.timesteps[0].item() == int(sig[0] * 1000) and len(s.timesteps) == 30


def test_first_unipc_step_equals_euler_step():
    g = torch.Generator().manual_seed(0)
    x, v = torch.randn(4, 5, generator=g, dtype=torch.float64), torch.randn(4, 5, generator=g, dtype=torch.float64)
    s = FlowUniPCMultistepScheduler(1000)
    s.set_timesteps(10, shift=5.0)
    e = FlowMatchScheduler(1000)
    e.set_timesteps(10, shift=5.0)
    torch.testing.assert_close(s.step(v, s.timesteps[0], x), e.step(v, x), rtol=1e-12, atol=1e-12)


def test_constant_velocity_field_is_integrated_exactly():
    """x(sigma) = x0 + sigma (eps - x0): with the true velocity v = eps - x0 every solver order lands on x0 at sigma 0."""
    g = torch.Generator().manual_seed(1)
    x0, eps = torch.randn(3, 7, generator=g, dtype=torch.float64), torch.randn(3, 7, generator=g, dtype=torch.float64)
    for order in (1, 2):
        s = FlowUniPCMultistepScheduler(1000, solver_order=order)
        s.set_timesteps(8, shift=5.0)
        x = (1 - s.sigmas[0]) * x0 + s.sigmas[0] * eps
        for t in s.timesteps:
            x = s.step(eps - x0, t, x)
        torch.testing.assert_close(x, x0, rtol=1e-9, atol=1e-9)


def test_unipc_order2_beats_euler_on_a_curved_field():
    """dx/dsigma = a(sigma) with the model returning a velocity that depends on sigma: the multistep corrector must help."""
    def run(sched, n):
        sched.set_timesteps(n, shift=1.0)
        x = torch.tensor([1.0], dtype=torch.float64)
        for i, t in enumerate(sched.timesteps):
            sig = sched.sigmas[i]
            v = torch.tensor([np.cos(3 * sig)], dtype=torch.float64) * x  # dx/dsigma = cos(3 sigma) x
            x = sched.step(v, t, x) if isinstance(sched, FlowUniPCMultistepScheduler) else sched.step(v, x)
        return x.item()

    s_hi = FlowUniPCMultistepScheduler(1000)
    ref = run(s_hi, 2000)
    exact = np.exp(-np.sin(3 * s_hi.sigmas[0]) / 3)  # x(0) = x(s0) exp(-(sin(3 s0))/3)
    assert abs(ref - exact) < 1e-5
    e_err = abs(run(FlowMatchScheduler(1000), 20) - exact)
    u_err = abs(run(FlowUniPCMultistepScheduler(1000), 20) - exact)
    assert u_err < 0.2 * e_err
