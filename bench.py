#!/usr/bin/env python3
"""Headline benchmark: denoising steps/s of the W8A8 Wan2.1-T2V-1.3B DiT at 832x480x81f on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = conditional DiT pass + unconditional DiT pass + classifier-free-guidance combine + scheduler update
(ViDiT-Q/examples/Wan2.1/wan/text2video.py:248-269) on synthetic latents / text context / random-init weights
of the named architecture, all resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12        # HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured with a float4 copy)
BF16_MFMA_PEAK = 2.5e15   # dense bf16 FLOP/s (same guide; a bare register-only MFMA stream sustains 1.8e15 here: tools/probes)
INT8_MFMA_PEAK = 5.03e15  # dense int8 op/s: 2048 op/clk/SIMD x 4 SIMD x 256 CU x 2.4 GHz (MI355X_MICROARCH.md, Matrix cores)


def synth_model(name, device, seed=0):
    """Random-init Wan backbone of the named architecture (xavier Linear as init_weights, small random biases,
    non-zero head) -- SURVEY 8(d) synthetic-input recipe."""
    from wan.configs import model_kwargs
    from wan.modules.model import WanModel

    torch.manual_seed(seed)
    with torch.device(device):
        m = WanModel(**model_kwargs(name))
    g = torch.Generator(device=device).manual_seed(seed)
    for mod in m.modules():
        if isinstance(mod, torch.nn.Linear) and mod.bias is not None:
            mod.bias.data.normal_(std=0.02, generator=g)
    torch.nn.init.xavier_uniform_(m.head.head.weight, generator=g)
    return m.eval()


class GemmTimer(list):
    """Collects (start_event, end_event, int8_ops) for every W8A8 GEMM launch in the timed region."""

    def summary(self):
        if not self:
            return None
        t = sum(s.elapsed_time(e) for s, e, _ in self) * 1e-3
        ops = float(sum(o for _, _, o in self))
        return dict(launches=len(self), seconds=t, ops=ops)


def pmc_traffic(which, workload_key):
    """(HBM bytes per launch, source) for `which` in {"gemm", "attention"} from the newest committed
    profiles/*_<which>_traffic.json (written by tools/{gemm,attn}_traffic_summary.py from separate rocprofv3 --pmc passes:
    counters cannot be read inside this process).  The summary names the workload it was taken on; when that is not the
    workload being timed now the figure does not apply and (None, reason) is returned."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{which}_traffic.json")))
    if not files:
        return None, "no committed PMC summary"
    with open(files[-1]) as fh:
        d = json.load(fh)
    src = "profiles/" + os.path.basename(files[-1])
    if d.get("workload", "t2v-1.3B 832*480 81f n1") != workload_key:
        return None, f"{src} was taken on '{d.get('workload', 't2v-1.3B 832*480 81f n1')}', not on this run's '{workload_key}'"
    return d["hbm_bytes_per_launch"], src + (f" ({d['code']})" if "code" in d else "")


def cpu_baseline(cfg, L, n_vidit_per_block=3, rows=1024, reps=3):
    """Reference fake-quant path (oracle/wan_ref.py, torch-CPU fp32; ViDiT layers with the reference's fp64 rotation
    GEMM, Q/viditq/viditq_quant_layer.py:62-63) on a bounded sample of the SAME workload: `rows` query tokens of one DiT
    block at sequence length L (all ten fake-quant Linears on those tokens, fp32 attention of rows x L keys), scaled by
    L/rows x blocks x 2 passes to one denoising step.  Beside it `cfg_a`: one WHOLE block of cfg-A (9 frames, L = 4680:
    BASELINE config 1, the reference's own CPU-runnable case), all tokens, scaled by blocks x 2 passes only."""
    from oracle import qdiff_ref as qr
    from oracle import wan_ref as wr

    cores_available = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    cores = max(1, min(cores_available, 64))  # torch's intra-op pool gains nothing past 64 threads on these GEMM sizes
    torch.set_num_threads(cores)
    C, Fd, H = cfg["dim"], cfg["ffn_dim"], cfg["num_heads"]
    g = torch.Generator().manual_seed(0)
    sd = {}
    for n in wr.LINEARS:
        o, i = (Fd, C) if n == "ffn.0" else (C, Fd) if n == "ffn.2" else (C, C)
        sd[n + ".weight"] = torch.randn(o, i, generator=g) * (2.0 / (i + o)) ** 0.5
        sd[n + ".bias"] = torch.zeros(o)
    for n in ("self_attn.norm_q", "self_attn.norm_k", "cross_attn.norm_q", "cross_attn.norm_k"):
        sd[n + ".weight"] = torch.ones(C)
    sd["norm3.weight"], sd["norm3.bias"], sd["modulation"] = torch.ones(C), torch.zeros(C), torch.randn(1, 6, C, generator=g) / C ** 0.5
    vidit = None
    if n_vidit_per_block:  # self-attention q / k / v carry the ViDiT scale + rotate, as in the GPU workload
        import numpy as np
        vidit = {}
        for n in ("self_attn.q", "self_attn.k", "self_attn.v"):
            signs = (torch.randint(0, 2, (C,), generator=g) * 2 - 1).double().numpy()
            R = torch.from_numpy(np.ascontiguousarray(qr.hadamard_from_signs(signs)))
            vidit[n] = (torch.rand(C, generator=g) + 0.5, R)
    blk = wr.block_from_state(sd, H, quant=True, vidit=vidit)
    d = C // H
    e0 = torch.randn(1, 6, C, generator=g) * 0.1
    ctx = torch.randn(512, C, generator=g)
    freqs = wr.rope_freqs(d)

    def median_time(fn):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2]

    # ---- (1) the timed workload's sequence length: a row slice against full-length keys / values
    x = torch.randn(rows, C, generator=g)
    kf = torch.randn(L, H, d, generator=g)
    vf = torch.randn(L, H, d, generator=g)
    grid = (1, 16, rows // 16)

    def slice_pass():
        e = [t.reshape(1, C) for t in (blk.mod + e0).chunk(6, dim=1)]
        h = wr.layer_norm(x, blk.eps) * (1 + e[1]) + e[0]
        q = wr.rms_norm(blk.lin["self_attn.q"](h), blk.norm_w["self_attn.norm_q"], blk.eps).view(rows, H, d)
        k = wr.rms_norm(blk.lin["self_attn.k"](h), blk.norm_w["self_attn.norm_k"], blk.eps).view(rows, H, d)
        v = blk.lin["self_attn.v"](h).view(rows, H, d)
        q, k = wr.rope_apply(q, grid, freqs), wr.rope_apply(k, grid, freqs)
        kf[:rows], vf[:rows] = k, v
        o = wr.attention(q, kf, vf).reshape(rows, C)
        y = x + blk.lin["self_attn.o"](o) * e[2]
        h = wr.layer_norm(y, blk.eps, *blk.norm3)
        q = wr.rms_norm(blk.lin["cross_attn.q"](h), blk.norm_w["cross_attn.norm_q"], blk.eps).view(rows, H, d)
        kc = wr.rms_norm(blk.lin["cross_attn.k"](ctx), blk.norm_w["cross_attn.norm_k"], blk.eps).view(-1, H, d)
        vc = blk.lin["cross_attn.v"](ctx).view(-1, H, d)
        y = y + blk.lin["cross_attn.o"](wr.attention(q, kc, vc).reshape(rows, C))
        h = wr.layer_norm(y, blk.eps) * (1 + e[4]) + e[3]
        return y + blk.lin["ffn.2"](torch.nn.functional.gelu(blk.lin["ffn.0"](h), approximate="tanh")) * e[5]

    t_slice = median_time(slice_pass)
    step_s = t_slice * (L / rows) * cfg["num_layers"] * 2
    # ---- (2) cfg-A: one whole block, every token (grid 3 x 30 x 52 = 4680 tokens)
    grid_a = (3, 30, 52)
    La = grid_a[0] * grid_a[1] * grid_a[2]
    xa = torch.randn(La, C, generator=g)
    t_blk = median_time(lambda: blk(xa, e0, grid_a, La, ctx, freqs))
    step_a = t_blk * cfg["num_layers"] * 2
    vd = f"ViDiT scale + fp64 rotation on {n_vidit_per_block} of them" if n_vidit_per_block else "no ViDiT layers"
    return dict(value=1.0 / step_s, unit="steps/s", cores=cores, cores_available=cores_available, kind="port",
                sample=f"{rows}-token row slice of one fake-quant DiT block at L={L} (10 fake-quant Linears, {vd}, + fp32 "
                       f"attention {rows}x{L}x{H} heads), median of {reps}: {t_slice:.3f}s, scaled x{L / rows:.1f} x{cfg['num_layers']} blocks x2 passes",
                cfg_a={"value": 1.0 / step_a, "unit": "steps/s", "workload": f"832*480 9f (L={La})",
                       "sample": f"one whole fake-quant block on all {La} tokens, median of {reps}: {t_blk:.3f}s, scaled x{cfg['num_layers']} blocks x2 passes"})


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: run `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N ... bench.py <same arguments>` as a CHILD process (never an exec: the parent stays a plain process
    that has made no GPU call), pass its output through and return its exit status (non-zero if any rank failed)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this image
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="t2v-1.3B")
    ap.add_argument("--size", default="832*480")
    ap.add_argument("--frames", type=int, default=81)
    ap.add_argument("--guide", type=float, default=5.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-quality", action="store_true")
    ap.add_argument("--quant-config", dest="quant_config", default="w8a8_all_linears.yaml", help="file under quant_configs/")
    ap.add_argument("--no-cfg-parallel", action="store_true", help="pure Ulysses over all GPUs (needs heads %% N == 0)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("WANQ_BENCH_GRAPH", "0")), choices=[0, 1],
                    help="1 (single GPU only): the two DiT passes of a step are replayed from a captured HIP graph "
                         "(wan/graph.py); 0 (default): every kernel is launched eagerly -- measured the same (DESIGN.md 5): the host runs ahead")
    ap.add_argument("--pass-streams", dest="pass_streams", default=os.environ.get("WANQ_PASS_STREAMS", "auto"), choices=["auto", "1", "2"],
                    help="one rank without cfg parallelism: the conditional and the unconditional pass of a step back to back on one stream, as "
                         "the reference runs them (1), or on two HIP streams (2: wan/utils/two_pass.py; bit-equal results, one pass's kernels fill "
                         "the other's launch boundaries -- faster by 0 - 3 %% at cfg-B depending on the box, slower at the 14B shapes).  auto "
                         "(default, what WanT2V.generate does): five untimed evaluations before the warm-up steps -- one to fill the caches, then "
                         "one stream / two streams / one / two, timed -- pick by the faster sample of each; the choice and its two timings are in "
                         "config.launch")
    ap.add_argument("--no-instrumented-repeat", dest="no_repeat", action="store_true",
                    help="diagnostic (kernel traces of the timed steps themselves): skip the instrumented repeat of the K steps; the line then "
                         "carries no roofline objects")
    ap.add_argument("--no-context-cache", dest="no_context_cache", action="store_true",
                    help="recompute cross_attn.k / cross_attn.v (+ RMSNorm) of the text context in every DiT pass, as round 2 did: they "
                         "do not depend on the timestep, and by default they are computed once per context tensor and kept "
                         "(QuantWanModel._context_source); the A/B of the two is in profiles/")
    ap.add_argument("--dit-fsdp", dest="dit_fsdp", action="store_true",
                    help="N > 1: shard the kernel-mode blocks' integer weights over all ranks, gathered one block ahead (wan/distributed/fsdp.py)")
    ap.add_argument("--preset", default=os.environ.get("WANQ_BENCH_PRESET", ""), choices=["", "14B-ulysses", "14B-w4a8-fsdp"],
                    help="14B-ulysses = BASELINE config 4: --model t2v-14B --size 1280*720 --no-cfg-parallel (Ulysses degree = N); "
                         "14B-w4a8-fsdp = BASELINE config 5: the same + --quant-config w4a8_mixed.yaml --dit-fsdp "
                         "(also selectable with WANQ_BENCH_PRESET in the environment)")
    args = ap.parse_args()
    if args.preset in ("14B-ulysses", "14B-w4a8-fsdp"):
        args.model, args.size, args.no_cfg_parallel = "t2v-14B", "1280*720", True
    if args.preset == "14B-w4a8-fsdp":
        args.quant_config, args.dit_fsdp = "w4a8_mixed.yaml", True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started the way the N = 1 run is started (`python bench.py --gpus N`): this process has not touched the GPU yet
        # and never will -- it starts the N ranks as children, relays rank 0's JSON line and exits with their status
        sys.exit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (start it as `python bench.py --gpus N`, or under "
                 f"torch.distributed.run with --nproc-per-node equal to --gpus)")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the quantized hot path has no CPU fallback)"
    # Rehearsal of the N > 1 control flow on a ONE-GPU box (RCCL refuses two ranks on one device): all ranks share cuda:0,
    # rendezvous over gloo, and the collectives are staged through host memory (tools/one_gpu_rehearsal.py).  Never a measurement;
    # refused (exit 4) when the box shows more than one GPU, so the variable cannot turn a real N > 1 run into gloo by accident.
    from wan.distributed import enter_one_gpu_rehearsal, stage_rehearsal_collectives
    rehearse = enter_one_gpu_rehearsal("WANQ_BENCH_REHEARSE_ON_ONE_GPU", world)
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if rehearse:
        dist.init_process_group("gloo", init_method="env://")
        stage_rehearsal_collectives()
    elif world > 1:
        try:
            dist.init_process_group("nccl", init_method="env://", device_id=dev)
            dist.barrier()  # the communicator is created lazily: make a failure show here, with its reason, not inside the step
        except Exception as e:  # one line, non-zero status (launch_ranks relays it); never a re-exec, never a silent fallback
            print(f"bench.py rank {rank}/{world}: RCCL process group failed to initialise: {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}",
                  file=sys.stderr, flush=True)
            sys.exit(3)

    from viditq_extension import qgemm
    from wan.configs import SIZE_CONFIGS, WAN_CONFIGS, latent_shape, seq_len_for
    from wan.distributed.parallel import ParallelPlan
    from wan.utils.fm_solvers_unipc import FlowUniPCMultistepScheduler

    cfg = WAN_CONFIGS[args.model]
    # N GPUs = cfg-parallel degree (cond / uncond pass on different GPUs) x Ulysses degree (token sequence sharded)
    plan = ParallelPlan(world, rank, *ParallelPlan.choose(world, cfg["num_heads"], not args.no_cfg_parallel))
    shape = latent_shape(SIZE_CONFIGS[args.size], args.frames)
    seq_len = seq_len_for(shape, sp_size=plan.sp_degree)
    torch.backends.cuda.matmul.allow_tf32 = False

    g = torch.Generator(device=dev).manual_seed(42)
    latent0 = torch.randn(shape, generator=g, device=dev)
    ctx_c = torch.randn(512, cfg["text_dim"], generator=g, device=dev) * 0.1
    ctx_u = torch.randn(512, cfg["text_dim"], generator=g, device=dev) * 0.1
    total = args.steps + args.warmup
    sched = FlowUniPCMultistepScheduler(cfg["num_train_timesteps"], shift=1.0)  # the reference's default solver
    sched.set_timesteps(max(total + args.steps, 30), device=dev, shift=5.0)  # + the instrumented repeat of the K steps

    # ---- the reference's flow on the synthetic model: FP model -> quant_layer_refactor (config) -> calibration pass
    #      (per-channel absmax hooks) -> channel masks + rotations (ptq) -> kernel mode (quant_generate, if_hardware)
    from qdiff import config as qcfg
    from qdiff.base.quant_layer import QuantizedLinear
    from wan import calib
    from wan.quant_wanx import QuantWanModel

    quant_config = qcfg.load(os.path.join(ROOT, "wan2.1-quantization_amd", "quant_configs", args.quant_config))
    fp = synth_model(args.model, dev, seed=0)
    model = QuantWanModel.from_float(fp, quant_config)
    model.quant_layer_refactor()
    hooks = calib.add_hooks(fp)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        fp([latent0], sched.timesteps[0:1], [ctx_c], seq_len_for(shape))
    # BASELINE config 3's hot reduction, timed: the SAME hooked FP pass once more (the running per-channel maxima are idempotent
    # under a repeat of the same input), wall time of the pass + a HIP event pair around every wanq_col_absmax launch
    # (W/get_calib_data_wanx.py:262-267: one per-channel absmax of every Linear input per call)
    from viditq_extension import _C as wanq_C
    calib_report = None
    if rank == 0:
        ctimer = {}
        wanq_C.set_call_timer(ctimer)
        torch.cuda.synchronize()
        tc0 = time.perf_counter()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            fp([latent0], sched.timesteps[0:1], [ctx_c], seq_len_for(shape))
        torch.cuda.synchronize()
        calib_ms = (time.perf_counter() - tc0) * 1e3
        wanq_C.set_call_timer(None)
        ev = ctimer.get("col_absmax", [])
        if ev:
            k_s = sum(a.elapsed_time(b) for a, b, _ in ev) * 1e-3
            k_b = float(sum(n for _, _, n in ev))
            calib_report = {"what": "one hooked FP (bf16 autocast) DiT pass at the timed workload's size: per-channel absmax of every Linear input "
                                    "(wanq_col_absmax, running maxima kept on the device)",
                            "ms_per_pass": calib_ms, "absmax_launches": len(ev), "absmax_bytes": k_b, "absmax_ms": k_s * 1e3,
                            "absmax_TBps": k_b / k_s / 1e12, "frac_of_8TBps": k_b / k_s / HBM_PEAK}
    calib_data = calib.gather_and_save_activation(hooks)
    gen = torch.Generator().manual_seed(0)
    n_vidit = 0
    for name, mod in model.named_modules():
        if isinstance(mod, QuantizedLinear) and (mod.uses_mask or mod.uses_rotation):
            calib.init_rotation_and_channel_mask_(mod, name, calib_data, gen)
            n_vidit += 1
    if quant_config.get("mixed_precision", None) is not None:  # BASELINE config 5: per-layer bit-widths by regex (quant_generate.py does the same)
        model.bitwidth_refactor()
    model.set_init_done()
    model.hardware_forward_refactor()
    model.context_cache = not args.no_context_cache
    n_quant = sum(1 for m in model.modules() if isinstance(m, QuantizedLinear))
    n_w4 = sum(1 for b in model.hip_blocks for m in b.modules() if getattr(m, "w_bits", 8) == 4)
    hb0 = model.hip_blocks[0]
    amaps = [(nm, am) for nm, am in (("self", getattr(hb0, "attn_map", None)), ("cross", getattr(hb0, "cross_attn_map", None))) if am]
    attn_desc = "; ".join(f"{nm}: attention map {am[0]}-bit {'sym' if am[1] else 'asym'} per key column (three streamed passes, csrc/attn_map.hip)"
                          for nm, am in amaps) + "; other attention bf16" if amaps else \
        "bf16" if not (hb0.attn_qk8 or hb0.cross_attn_qk8 or hb0.attn_v_bits or hb0.cross_attn_v_bits) else ", ".join(
        [f"{nm}: " + " + ".join((["int8 Q.K^T (q, k per (token, head))"] if qk else []) + ([f"v {vb}-bit per (head, channel)"] if vb else []))
         for nm, qk, vb in (("self", hb0.attn_qk8, hb0.attn_v_bits), ("cross", hb0.cross_attn_qk8, hb0.cross_attn_v_bits)) if qk or vb]) + "; P.V bf16"
    sharded = None
    if args.dit_fsdp and world > 1:
        sharded = model.shard_blocks(None)  # all ranks hold the same model: FULL_SHARD over the world
    if world > 1 or args.no_quality:
        del fp
        torch.cuda.empty_cache()

    from wan.utils.fused_step import FusedStep
    fused = FusedStep(sched, args.guide, latent0)  # CFG combine + UniPC update: one kernel per step
    graphed = None
    use_graph = [bool(args.graph) and world == 1]
    if use_graph[0]:
        from wan.graph import GraphedPasses
        graphed = GraphedPasses(model, latent0, [ctx_c, ctx_u], seq_len)

    from wan.utils.two_pass import TwoPassStreams
    two = TwoPassStreams(dev, enabled=plan.cfg_degree == 1 and world == 1 and not use_graph[0], mode=args.pass_streams)

    def step(latent, i):
        t = sched.timesteps[i:i + 1]
        if graphed is not None and use_graph[0]:
            cond, uncond = graphed(latent, t)
        elif two.enabled:
            cond, uncond = two(lambda c: model([latent], t, [c], seq_len, plan.sp)[0], latent, [ctx_c, ctx_u])
        elif plan.cfg_degree == 2:  # my half of the GPUs runs ONE of the two passes; a 2 MB all-gather joins them
            mine = model([latent], t, [ctx_c if plan.cfg_index == 0 else ctx_u], seq_len, plan.sp)[0]
            cond, uncond = plan.gather_cfg(mine)
        else:
            cond = model([latent], t, [ctx_c], seq_len, plan.sp)[0]
            uncond = model([latent], t, [ctx_u], seq_len, plan.sp)[0]
        return fused.step(cond, uncond, latent, sched.timesteps[i])

    latent = latent0
    # untimed, before the warm-up steps: the helper's first call (one stream: it fills the caches both passes read) and, in auto mode, the
    # measurement that picks the order of the two passes -- the warm-up and the timed steps then run ONE schedule
    two.tune(lambda c: model([latent0], sched.timesteps[0:1], [c], seq_len, plan.sp)[0], latent0, [ctx_c, ctx_u])
    launch_desc = two.describe()
    for i in range(args.warmup):
        latent = step(latent, i)
    from wan import ops as wan_ops
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    a2a0 = (plan.sp.a2a_calls, plan.sp.a2a_bytes_sent)
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        latent = step(latent, i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    a2a = ((plan.sp.a2a_calls - a2a0[0]) / args.steps, (plan.sp.a2a_bytes_sent - a2a0[1]) / args.steps)
    if world > 1:
        tmax = torch.tensor([dt], device="cpu" if rehearse else dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    assert torch.isfinite(latent).all(), "non-finite latent after the timed steps"

    # ---- per-kernel timing for the roofline objects: the same K steps once more, with a HIP event pair around every GEMM /
    # attention launch on the launch stream (about 1400 pairs per step, which is why they stay out of the headline region
    # above; `instrumented_ms_per_step` shows what they cost).  share_of_step = kernel time / wall time of THIS region.
    use_graph[0] = False  # launches must be eager to be bracketed by events
    two.enabled = False   # ... and on ONE stream: beside another pass's kernels an event pair times the sharing, not the kernel
    timer = GemmTimer()
    atimer = GemmTimer()
    qgemm.set_timer(timer)
    wan_ops.set_attention_timer(atimer)
    htimer = {}
    wanq_C.set_call_timer(htimer)  # the HBM-bound row-wise kernels of the step (an event pair around each launch)
    torch.cuda.synchronize()
    tp0 = time.perf_counter()
    for i in range(total, total + (0 if args.no_repeat else args.steps)):
        latent = step(latent, i)
    torch.cuda.synchronize()
    dt_prof = max(time.perf_counter() - tp0, 1e-9)
    qgemm.set_timer(None)
    wan_ops.set_attention_timer(None)
    wanq_C.set_call_timer(None)

    gs = timer.summary()
    out = {
        "metric": "denoising steps/sec Wan2.1-%s %s %sx%df" % (args.model.split("-")[-1], "W4A8-mixed" if n_w4 else "W8A8", args.size.replace("*", "x"), args.frames), "value": args.steps / dt, "unit": "steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "instrumented_ms_per_step": dt_prof / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
        "config": {"workload": f"Wan2.1-{args.model} DiT, {n_quant} Linears W8A8{f' of which {n_w4} W4A8 (packed 4-bit weights)' if n_w4 else ''} (W asym per-channel static, A sym per-token dynamic; "
                               f"ViDiT-Q scale+rotate alpha=0.5665 on {n_vidit} self-attn q/k/v layers), "
                               f"{args.size}x{args.frames}f, latent {list(shape)}, L={seq_len}, cond+uncond+CFG+scheduler per step",
                   "quant_config": args.quant_config, "attention": attn_desc, "parallelism": plan.describe(),
                   "context_kv": "cross_attn.k / .v of the text context computed once per context tensor (step-invariant), reused by every step"
                                 if model.context_cache else "cross_attn.k / .v of the text context recomputed in every pass",
                   "dit_fsdp": None if sharded is None else {"ranks": sharded.P, "block_weight_MB_per_rank": round(sharded.bytes_per_rank() / 1e6, 1),
                                                             "of_MB": round(len(sharded.blocks) * sharded.full_bytes / 1e6, 1),
                                                             "w4_unpack_scratch_MB": round(qgemm.w4_scratch_bytes(dev) / 1e6, 1)},
                   "launch": "hip graph replay of the two DiT passes + 1 fused CFG/scheduler kernel" if graphed is not None else
                             f"eager launches, the two passes of a step on {launch_desc}"
                             + " + 1 fused CFG/scheduler kernel",
                   "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                   # Ulysses exchange of THIS rank inside the timed region (wan/distributed/parallel.py counts what it hands to
                   # all_to_all_single): bytes leaving the GPU over xGMI per step, and the number of all-to-alls
                   "a2a_per_step": int(a2a[0]), "a2a_bytes_per_step": int(a2a[1])},
    }
    # Two MFMA-bound kernels carry the step: the bf16 flash attention (the dominant one at L = 32760) and the int8 GEMM
    # behind every W8A8 Linear.  `roofline` is whichever took more of the timed region; the other is reported beside it.
    # `traffic`: HBM bytes per launch from the committed PMC passes (profiles/*_traffic.json: FETCH_SIZE x2 + WRITE_SIZE,
    # one counter per rocprofv3 pass, averaged over the same launch mix); counters cannot be read inside this process.
    lines = []
    wkey = f"{args.model} {args.size} {args.frames}f n{world}"
    g_traffic, g_src = pmc_traffic("gemm", wkey)
    a_traffic, a_src = pmc_traffic("attention", wkey)
    if attn_desc != "bf16" and a_traffic is not None:
        a_traffic, a_src = None, a_src + " was taken on the bf16 attention kernel, not on this run's attention configuration"
    if gs:
        ach = gs["ops"] / gs["seconds"]
        lines.append({"bound": "mfma", "kernel": "gemm_w8a8_pp_kernel (ping-pong persistent, W8) / gemm_w8a8_big_kernel (W4, GELU or 16-bit residual forms) / gemm_w8a8_kernel (small shapes): int8 MFMA, every quantised linear",
                      "achieved": ach / 1e12, "peak": INT8_MFMA_PEAK / 1e12, "unit": "TOP/s", "frac": ach / INT8_MFMA_PEAK,
                      "traffic": g_traffic, "traffic_source": g_src, "traffic_unit": "HBM bytes per launch (PMC)", "launches": gs["launches"],
                      "avg_launch_us": gs["seconds"] / gs["launches"] * 1e6, "share_of_step": gs["seconds"] / dt_prof})
    asum = atimer.summary()
    if asum:
        ach = asum["ops"] / asum["seconds"]
        lines.append({"bound": "mfma", "kernel": "attn_fwd16_kernel (bf16 MFMA 16x16x32 flash attention, self + cross; WANQ_ATTN_M16=0: attn_fwd_kernel, 32x32x16)" if attn_desc == "bf16" else f"attn_fwd_kernel ({attn_desc})", "achieved": ach / 1e12,
                      "peak": BF16_MFMA_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK,
                      "traffic": a_traffic, "traffic_source": a_src, "traffic_unit": "HBM bytes per launch (PMC)",
                      "launches": asum["launches"], "avg_launch_us": asum["seconds"] / asum["launches"] * 1e6,
                      "share_of_step": asum["seconds"] / dt_prof})
    # whole-step int8 matrix work against the int8 MFMA roofline (north star: "throughput ... as fraction of the int8-MFMA roofline"):
    # every W8A8 / W4A8 GEMM of the step, plus the Q.K^T half of the attention FLOPs where that runs on the int8 matrix cores
    # (attn.qk configs).  With bf16 attention (the reference's Wan wiring) most of the step's time is not int8 work at all.
    if gs:
        qk_share = sum(1 for b_ in model.hip_blocks for f_ in (b_.attn_qk8,) if f_) / max(1, len(model.hip_blocks))
        int8_ops = gs["ops"] / args.steps + (0.5 * asum["ops"] / args.steps * qk_share if (asum and qk_share) else 0.0)
        out["int8_step"] = {"int8_ops_per_step": int8_ops, "achieved_TOPs": int8_ops * args.steps / dt / 1e12,
                            "frac_of_int8_mfma_peak": int8_ops * args.steps / dt / INT8_MFMA_PEAK,
                            "what": "all int8 MFMA work of a step / the step's wall time / 5.03 POP/s"}
    lines.sort(key=lambda r: -r["share_of_step"])
    if lines:
        out["roofline"] = lines[0]
    if len(lines) > 1:
        out["roofline_second_kernel"] = lines[1]
    # HBM-bound kernels of the step (north star: "HBM GB/s against gfx950 peak"): algorithmic bytes per launch (what the wrapper
    # hands over: rows x cols x (input + output element sizes) + the per-token vectors / the rotary table) / the launch's
    # duration between its own HIP events, against the 8 TB/s spec
    hb = []
    names = {"gelu_quant_sum": "quant_rows_wave_kernel<BF16,18,GELU> (GELU + per-token quantise of the FFN hidden)",
             "quant_sum": "rowwise_kernel<..., false> (per-token quantise: attention output -> o, cross-attention output)",
             "layernorm_quant": "rowwise_kernel<..., LN> (LayerNorm + modulate + per-token quantise)",
             "layernorm_rotate_quant_x3": "rotate_kernel<12,1,4,LN,MULTI> (LayerNorm + modulate + ViDiT scale / rotate + quantise for q, k, v)",
             "layernorm_rotate_quant": "rotate_kernel<..., LN> (LayerNorm + modulate + ViDiT scale / rotate + quantise, one consumer)",
             "rotate_quant": "rotate_kernel / rotate140_kernel (ViDiT scale / rotate + quantise)",
             "rmsnorm_rope": "rmsnorm_rope_kernel (RMSNorm + RoPE on q / k, in place)",
             "rmsnorm_rope_scatter": "rmsnorm_rope_kernel<..., SC> (RMSNorm + RoPE stored into the Ulysses send images)",
             "rmsnorm_rope_q8": "rmsnorm_rope_kernel<..., Q8> (RMSNorm + RoPE + per-(token, head) int8 codes for the int8 Q.K^T attention)", "rmsnorm": "rmsnorm_rope_kernel (RMSNorm only: cross-attention q / k)",
             "layernorm": "rowwise_kernel<..., LN> (LayerNorm, fp output)"}
    for tag, ev in htimer.items():
        secs = sum(a.elapsed_time(b) for a, b, _ in ev) * 1e-3
        nbytes = float(sum(n for _, _, n in ev))
        if secs > 0:
            hb.append({"kernel": names.get(tag, tag), "entry": tag, "launches_per_step": len(ev) / args.steps, "bytes_per_launch": nbytes / len(ev),
                       "avg_us": secs / len(ev) * 1e6, "TBps": nbytes / secs / 1e12, "frac": nbytes / secs / HBM_PEAK,
                       "share_of_step": secs / dt_prof})
    hb.sort(key=lambda r: -r["share_of_step"])
    if hb:
        out["roofline_hbm"] = hb
    if calib_report is not None:
        out["calibration"] = calib_report
    if rank == 0 and world == 1 and not args.no_quality:
        # deviation of the quantized DiT output from the FP (bf16-autocast) output of the same synthetic model
        t = sched.timesteps[0:1]
        with torch.no_grad():
            yq = model([latent0], t, [ctx_c], seq_len_for(shape))[0]
            with torch.autocast("cuda", dtype=torch.bfloat16):
                yf = fp([latent0], t, [ctx_c], seq_len_for(shape))[0]
        mse = (yq - yf).pow(2).mean().item()
        rng = (yf.max() - yf.min()).item()
        out["quality"] = {"tensor": "DiT output latent (noise_pred)", "rel_l2_vs_fp": ((yq - yf).norm() / yf.norm()).item(),
                          "psnr_db_vs_fp": 10 * torch.log10(torch.tensor(rng * rng / mse)).item()}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # reported at N = 1 only
        out["cpu_baseline"] = cpu_baseline(cfg, seq_len, n_vidit_per_block=n_vidit // max(1, cfg["num_layers"]))
    if rank == 0:
        if rehearse:
            out["data"] = "REHEARSAL on one GPU (not a measurement)"
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
