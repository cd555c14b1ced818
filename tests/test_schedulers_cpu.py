"""Host-side samplers (CPU): schedule values per SURVEY Appendix C, Euler consistency, exactness on linear fields."""
import numpy as np
import pytest
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


def test_dpmpp_first_step_constant_field_and_convergence():
    """FlowDPMSolverMultistepScheduler (--sample_solver dpm++): the schedule of get_sampling_sigmas (reference
    fm_solvers.py:22-26: starts at sigma = 1, where alpha = 0), the first step is the flow-matching Euler step, a constant
    velocity field is integrated exactly, and the 2M update converges faster than Euler on a curved field."""
    from wan.utils.fm_solvers import FlowDPMSolverMultistepScheduler, FlowMatchScheduler, get_sampling_sigmas

    n = 10
    s = FlowDPMSolverMultistepScheduler(1000)
    s.set_timesteps(n, shift=5.0)
    sig = get_sampling_sigmas(n, 5.0)
    assert sig[0] == 1.0 and len(s.sigmas) == n + 1 and s.sigmas[-1] == 0.0
    np.testing.assert_allclose(s.sigmas[:-1], sig.astype(np.float32), rtol=0, atol=0)
    assert int(s.timesteps[0]) == 1000
    x = torch.randn(4, 5, generator=torch.Generator().manual_seed(0))
    v = torch.randn(4, 5, generator=torch.Generator().manual_seed(1))
    y = s.step(v, s.timesteps[0], x)
    torch.testing.assert_close(y, x + (s.sigmas[1] - s.sigmas[0]) * v, rtol=1e-6, atol=1e-6)
    # constant velocity: x(sigma) = x1 + (sigma - 1) v exactly, for every step
    s.set_timesteps(n, shift=5.0)
    z = x.clone()
    for t in s.timesteps:
        z = s.step(v, t, z)
    torch.testing.assert_close(z, x - v, rtol=1e-5, atol=1e-5)

    # curved field: x' = dx/dsigma = a * x  ->  x(0) = x(1) * exp(-a); velocity model v(x) = a x
    def run(make, steps):
        sc = make()
        sc.set_timesteps(steps, shift=1.0)
        z = torch.ones(1, dtype=torch.float64)
        for t in sc.timesteps:
            vv = 0.7 * z
            z = sc.step(vv, t, z) if "timestep" in sc.step.__code__.co_varnames else sc.step(vv, z)
        return abs(float(z) - np.exp(-0.7))

    e2 = [run(lambda: FlowDPMSolverMultistepScheduler(1000), k) for k in (20, 40)]
    e1 = [run(lambda: FlowMatchScheduler(1000), k) for k in (20, 40)]
    # (the reference keeps its sigmas in fp32, which puts a ~4e-5 floor under the second-order error here)
    assert e2[0] < e1[0] * 0.05 and e2[1] < e1[1] * 0.05 and max(e2) < 1e-4, (e1, e2)


def test_fused_step_equals_scheduler_for_all_solvers():
    """FusedStep (CFG combine + scheduler update as ONE linear combination per step, coefficients derived by running the
    scheduler's own step() on symbolic linear forms) reproduces the plain `noise = u + g (c - u); x = sched.step(...)` loop for
    UniPC, DPM++ and Euler.  The GPU kernel is replaced by its definition here (out[o] = sum_i coef[o][i] in[i])."""
    import wan.utils.fused_step as fs
    from wan.utils.fm_solvers import FlowDPMSolverMultistepScheduler, FlowMatchScheduler
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

    def definition(coef, ins, outs):
        c = torch.as_tensor(coef, dtype=torch.float32).reshape(len(outs), len(ins))
        for o in range(len(outs)):
            outs[o].copy_(sum(c[o, i] * ins[i] for i in range(len(ins))))

    saved = fs.lincomb
    fs.lincomb = definition
    try:
        for mk in (lambda: FlowUniPCMultistepScheduler(1000, shift=1.0), lambda: FlowDPMSolverMultistepScheduler(1000),
                   lambda: FlowMatchScheduler(1000)):
            a, b = mk(), mk()
            a.set_timesteps(9, device="cpu", shift=5.0)
            b.set_timesteps(9, device="cpu", shift=5.0)
            g = torch.Generator().manual_seed(0)
            x = torch.randn(16, 3, 8, 8, generator=g)
            xa, xb = x.clone(), x.clone()
            f = fs.FusedStep(b, 5.0, like=x)
            for t in a.timesteps:
                c, u = torch.randn(x.shape, generator=g), torch.randn(x.shape, generator=g)
                noise = u + 5.0 * (c - u)
                xa = a.step(noise, t, xa) if fs._takes_timestep(a) else a.step(noise, xa)
                xb = f.step(c, u, xb, t)
                assert float((xa - xb).abs().max() / xa.abs().max()) < 1e-5
            assert f.n_launch == 9
    finally:
        fs.lincomb = saved


# ---------------------------------------------------------------------------------------------------------------------
# Pinned: the reference's own scheduler files driven on seeded model outputs (tests/golden/make_golden_schedulers.py)
@pytest.mark.parametrize("n", [3, 10, 50])
def test_unipc_vs_reference_golden(n):
    """FlowUniPCMultistepScheduler (bh2, order 2, lower_order_final) == the reference's, step by step: timesteps exact, sigmas and
    every latent of the trajectory to fp32 rounding."""
    import os

    import numpy as np
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"sched_unipc_{n}.npz"))
    s = FlowUniPCMultistepScheduler(1000, shift=1.0)
    s.set_timesteps(n, device="cpu", shift=5.0)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps"])
    np.testing.assert_allclose(np.asarray(s.sigmas, dtype=np.float64), g["sigmas"].astype(np.float64), rtol=2e-7, atol=1e-9)
    x = torch.from_numpy(g["x"][0])
    for i, t in enumerate(s.timesteps):
        x = s.step(torch.from_numpy(g["model_out"][i]), t, x)
        ref = g["x"][i + 1]
        assert float(np.abs(x.numpy() - ref).max()) <= 2e-5 * float(np.abs(ref).max()), (n, i)


@pytest.mark.parametrize("n", [3, 10, 50])
def test_dpmpp_vs_reference_golden(n):
    """FlowDPMSolverMultistepScheduler (dpmsolver++, midpoint, order 2) == the reference's on the `--sample_solver dpm++` schedule
    (sigmas from get_sampling_sigmas, text2video.py:223-232)."""
    import os

    import numpy as np
    from wan.utils.fm_solvers import FlowDPMSolverMultistepScheduler, get_sampling_sigmas

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"sched_dpmpp_{n}.npz"))
    np.testing.assert_allclose(get_sampling_sigmas(n, 5.0), g["sampling_sigmas"], rtol=1e-15)
    s = FlowDPMSolverMultistepScheduler(1000)
    s.set_timesteps(n, device="cpu", shift=5.0)
    assert np.array_equal(s.timesteps.numpy(), g["timesteps"])
    np.testing.assert_allclose(np.asarray(s.sigmas, dtype=np.float64), g["sigmas"].astype(np.float64), rtol=2e-7, atol=1e-9)
    x = torch.from_numpy(g["x"][0])
    for i, t in enumerate(s.timesteps):
        x = s.step(torch.from_numpy(g["model_out"][i]), t, x)
        ref = g["x"][i + 1]
        assert float(np.abs(x.numpy() - ref).max()) <= 2e-5 * float(np.abs(ref).max()), (n, i)


def test_two_pass_helper_is_sequential_without_a_gpu():
    """wan/utils/two_pass.py on a CPU tensor: no streams exist, the two passes run back to back in order (the reference's order)."""
    import torch

    from wan.utils.two_pass import TwoPassStreams

    two = TwoPassStreams("cpu")
    assert not two.enabled and two.streams is None
    seen = []
    outs = two(lambda c: (seen.append(c), torch.full((2,), float(c)))[1], torch.zeros(2), [3, 5])
    assert seen == [3, 5] and [o[0].item() for o in outs] == [3.0, 5.0]
    assert two.tune(lambda c: torch.zeros(1), torch.zeros(2), [3, 5]) is None and two.describe() == "one stream"
    with pytest.raises(ValueError):
        TwoPassStreams("cpu", mode="3")
