"""Step-level host path: the fused CFG + scheduler kernel and the HIP-graph replay of the DiT passes."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_lincomb_kernel_vs_definition():
    from wan.utils.fused_step import lincomb

    g = torch.Generator(device=DEV).manual_seed(0)
    for n_out, n_in, numel in [(1, 1, 4), (3, 6, 16 * 21 * 60 * 104), (4, 8, 1000)]:
        ins = [torch.randn(numel, device=DEV, generator=g) for _ in range(n_in)]
        outs = [torch.empty(numel, device=DEV) for _ in range(n_out)]
        coef = torch.randn(n_out, n_in)  # host coefficients: they travel by value with the launch
        lincomb(coef.numpy(), ins, outs)
        for o in range(n_out):
            ref = sum(coef[o, i].double() * ins[i].double() for i in range(n_in))
            assert float((outs[o].double() - ref).abs().max()) < 1e-5 * float(ref.abs().max() + 1)
    # an output may alias an input element for element
    a, b = torch.randn(4096, device=DEV, generator=g), torch.randn(4096, device=DEV, generator=g)
    want = 2.0 * a - 0.5 * b
    lincomb([[2.0, -0.5]], [a, b], [a])
    torch.testing.assert_close(a, want, rtol=1e-6, atol=1e-6)
    with pytest.raises(RuntimeError):
        lincomb([[1.0]], [torch.zeros(6, device=DEV)], [torch.zeros(6, device=DEV)])  # numel % 4


@pytest.mark.parametrize("solver", ["unipc", "dpm++", "euler"])
def test_fused_step_on_gpu_equals_plain_scheduler(solver):
    from wan.utils.fm_solvers import FlowDPMSolverMultistepScheduler, FlowMatchScheduler
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    from wan.utils.fused_step import FusedStep, _takes_timestep

    mk = {"unipc": lambda: FlowUniPCMultistepScheduler(1000, shift=1.0), "dpm++": lambda: FlowDPMSolverMultistepScheduler(1000),
          "euler": lambda: FlowMatchScheduler(1000)}[solver]
    a, b = mk(), mk()
    a.set_timesteps(12, device=DEV, shift=5.0)
    b.set_timesteps(12, device=DEV, shift=5.0)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(16, 3, 60, 104, device=DEV, generator=g)
    xa, xb = x.clone(), x.clone()
    f = FusedStep(b, 5.0, like=x)
    for t in a.timesteps:
        c, u = torch.randn(x.shape, device=DEV, generator=g), torch.randn(x.shape, device=DEV, generator=g)
        noise = u + 5.0 * (c - u)
        xa = a.step(noise, t, xa) if _takes_timestep(a) else a.step(noise, xa)
        xb = f.step(c, u, xb, t)
        assert float((xa - xb).abs().max() / xa.abs().max()) < 2e-6
    assert f.n_launch == 12


def test_fused_step_with_the_cpu_running_steps_ahead_of_the_gpu():
    """In the real loop the CPU enqueues a whole step (~1400 launches, 0.5 s of GPU time) and is back in FusedStep.step long
    before the GPU gets there.  Whatever carries the coefficients must not be reused before the launch has consumed it (an
    earlier form refreshed ONE pinned host buffer per step with an async copy: steps then ran with their successor's
    coefficients -- found by the headline-size entry-point test).  Here ~30 ms of queued GPU work precede every fused step and
    nothing synchronises until the end."""
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    from wan.utils.fused_step import FusedStep

    a, b = FlowUniPCMultistepScheduler(1000, shift=1.0), FlowUniPCMultistepScheduler(1000, shift=1.0)
    a.set_timesteps(8, device=DEV, shift=5.0)
    b.set_timesteps(8, device=DEV, shift=5.0)
    g = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(16, 21, 60, 104, device=DEV, generator=g)
    cs = [torch.randn(x.shape, device=DEV, generator=g) for _ in range(8)]
    us = [torch.randn(x.shape, device=DEV, generator=g) for _ in range(8)]
    big = torch.randn(8192, 8192, device=DEV, dtype=torch.bfloat16)
    xa = x.clone()
    for i, t in enumerate(a.timesteps):
        xa = a.step(us[i] + 5.0 * (cs[i] - us[i]), t, xa)
    torch.cuda.synchronize()
    f, xb = FusedStep(b, 5.0, like=x), x.clone()
    for i, t in enumerate(b.timesteps):
        for _ in range(24):
            big @ big  # ~1.3 ms each: the GPU falls behind, the CPU does not wait
        xb = f.step(cs[i], us[i], xb, t)
    torch.cuda.synchronize()
    assert float((xa - xb).abs().max() / xa.abs().max()) < 2e-6


def test_graph_replay_of_the_dit_passes_is_bit_equal_to_eager():
    """The conditional + unconditional pass of a kernel-mode model captured into one HIP graph: replays with new latents /
    timesteps reproduce the eager outputs bit for bit."""
    from qdiff import config as qcfg
    from wan.configs import seq_len_for
    from wan.graph import GraphedPasses
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    quant_config = qcfg.load(os.path.join(root, "wan2.1-quantization_amd", "quant_configs", "w8a8_plain.yaml"))
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=512, ffn_dim=1024, num_heads=4, num_layers=2, text_dim=64, freq_dim=64).eval()
    g = torch.Generator(device=DEV).manual_seed(2)
    torch.nn.init.xavier_uniform_(fp.head.head.weight, generator=g)
    model = QuantWanModel.from_float(fp, quant_config)
    model.quant_layer_refactor()
    model.set_init_done()
    model.hardware_forward_refactor()
    shape = (16, 3, 20, 18)
    seq_len = seq_len_for(shape)
    ctx = [torch.randn(24, 64, device=DEV, generator=g) * 0.1 for _ in range(2)]
    lat0 = torch.randn(shape, device=DEV, generator=g)
    gp = GraphedPasses(model, lat0, ctx, seq_len)
    for step in range(3):
        lat = torch.randn(shape, device=DEV, generator=g)
        t = torch.tensor([900 - 300 * step], device=DEV)
        eager = [model([lat], t, [c], seq_len)[0].clone() for c in ctx]
        outs = gp(lat, t)
        torch.cuda.synchronize()
        for e, o in zip(eager, outs):
            assert torch.equal(e, o)


def test_graph_replay_survives_cache_eviction_and_sees_context_updates():
    """The captured passes hold their own cross_attn.k / .v launches (the per-context cache is off during capture): a replay is
    still right after more other contexts ran eagerly than the cache keeps, after the blocks' cache was dropped, and after an
    in-place update of the graph's static context buffer."""
    from qdiff import config as qcfg
    from wan.configs import seq_len_for
    from wan.graph import GraphedPasses
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    quant_config = qcfg.load(os.path.join(root, "wan2.1-quantization_amd", "quant_configs", "w8a8_plain.yaml"))
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=512, ffn_dim=1024, num_heads=4, num_layers=2, text_dim=64, freq_dim=64).eval()
    g = torch.Generator(device=DEV).manual_seed(5)
    torch.nn.init.xavier_uniform_(fp.head.head.weight, generator=g)
    model = QuantWanModel.from_float(fp, quant_config)
    model.quant_layer_refactor()
    model.set_init_done()
    model.hardware_forward_refactor()
    shape = (16, 3, 20, 18)
    seq_len = seq_len_for(shape)
    ctx = [torch.randn(24, 64, device=DEV, generator=g) * 0.1 for _ in range(2)]
    lat = torch.randn(shape, device=DEV, generator=g)
    t = torch.tensor([700], device=DEV)
    gp = GraphedPasses(model, lat, ctx, seq_len)
    assert getattr(model, "context_cache", True)  # the capture restored the setting
    want = [model([lat], t, [c], seq_len)[0].clone() for c in ctx]
    # six other contexts through the eager path (the cache keeps four), with large temporaries in between
    for i in range(6):
        other = torch.randn(24, 64, device=DEV, generator=g)
        model([lat], t, [other], seq_len)
        del other
        junk = torch.randn(1 << 22, device=DEV)
        del junk
    model.__dict__.pop("_ctx_cache", None)
    torch.cuda.empty_cache()
    outs = gp(lat, t)
    torch.cuda.synchronize()
    for w_, o in zip(want, outs):
        assert torch.equal(w_, o)
    # in-place update of the graph's own context buffer
    new_ctx = torch.randn(24, 64, device=DEV, generator=g) * 0.1
    gp.ctx[0].copy_(new_ctx)
    outs = gp(lat, t)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], model([lat], t, [new_ctx], seq_len)[0])
    assert torch.equal(outs[1], want[1])


@pytest.mark.parametrize("dims", [(512, 1024, 4, 2, (16, 3, 20, 18)), (1536, 8960, 12, 2, (16, 5, 60, 104))])
def test_two_pass_streams_are_bit_equal_to_one_stream(dims):
    """The conditional and the unconditional pass of a step issued on two HIP streams (wan/utils/two_pass.py: what WanT2V.generate and
    bench.py do on one rank) against the same passes back to back on one stream, a sampling loop of several steps with the fused CFG +
    scheduler update in between: every latent bit-equal -- the kernels of one pass share no mutable state with those of the other, so
    running beside each other must not change a bit (cf. tests/test_gpu_corun.py).  Second case: two blocks at the 1.3B width, 7800
    tokens, the headline quant config (ViDiT on q / k / v)."""
    from qdiff import config as qcfg
    from qdiff.base.quant_layer import QuantizedLinear
    from wan import calib
    from wan.configs import seq_len_for
    from wan.modules.model import WanModel
    from wan.quant_wanx import QuantWanModel
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler
    from wan.utils.fused_step import FusedStep
    from wan.utils.two_pass import TwoPassStreams

    dim, ffn, heads, layers, shape = dims
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    quant_config = qcfg.load(os.path.join(root, "wan2.1-quantization_amd", "quant_configs", "w8a8_all_linears.yaml"))
    torch.manual_seed(0)
    with torch.device(DEV):
        fp = WanModel(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=layers, text_dim=64, freq_dim=64).eval()
    g = torch.Generator(device=DEV).manual_seed(2)
    torch.nn.init.xavier_uniform_(fp.head.head.weight, generator=g)
    seq_len = seq_len_for(shape)
    ctx = [torch.randn(24, 64, device=DEV, generator=g) * 0.1 for _ in range(2)]
    lat0 = torch.randn(shape, device=DEV, generator=g)
    model = QuantWanModel.from_float(fp, quant_config)
    model.quant_layer_refactor()
    hooks = calib.add_hooks(fp)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        fp([lat0], torch.tensor([900], device=DEV), [ctx[0]], seq_len)
    data = calib.gather_and_save_activation(hooks)
    gen = torch.Generator().manual_seed(0)
    for name, mod in model.named_modules():
        if isinstance(mod, QuantizedLinear) and (mod.uses_mask or mod.uses_rotation):
            calib.init_rotation_and_channel_mask_(mod, name, data, gen)
    model.set_init_done()
    model.hardware_forward_refactor()

    def loop(two):
        sched = FlowUniPCMultistepScheduler(1000, shift=1.0)
        sched.set_timesteps(6, device=DEV, shift=5.0)
        fused = FusedStep(sched, 5.0, lat0)
        lat, lats = lat0, []
        for t in sched.timesteps:
            cond, uncond = two(lambda c: model([lat], t.reshape(1), [c], seq_len)[0], lat, ctx)
            lat = fused.step(cond, uncond, lat, t)
            lats.append(lat.clone())
        torch.cuda.synchronize()
        return lats

    one = loop(TwoPassStreams(DEV, enabled=False))
    two = TwoPassStreams(DEV, mode="2")
    both = loop(two)
    assert two.enabled and two.decided and two.calls == 6
    for a, b in zip(one, both):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    # and again: the second loop's first step is already on two streams (the caches are filled)
    for a, b in zip(one, loop(two)):
        assert torch.equal(a, b)
    # auto (the default): step 1 on one stream, steps 2 - 5 timed alternately on one stream and on two, then whichever was faster -- the
    # latents do not depend on which
    auto = TwoPassStreams(DEV, mode="auto")
    assert not auto.decided
    for a, b in zip(one, loop(auto)):
        assert torch.equal(a, b)
    assert auto.decided and auto.tuned is not None and min(auto.tuned) > 0 and auto.enabled == (auto.tuned[1] < 0.99 * auto.tuned[0])
    assert ("two HIP streams" in auto.describe()) == auto.enabled
