"""The four entry points end to end on a truncated 1.3B backbone (2 blocks) at a tiny video size:
fp_generate -> get_calib_data_wanx -> ptq_wanx -> quant_generate (kernel mode and simulation mode)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "wan2.1-quantization_amd")


def run(script, *args, cwd):
    cmd = [sys.executable, os.path.join(PKG, script), "--task", "t2v-1.3B", "--size", "832*480", "--frame_num", "5", "--num_layers", "2",
           "--sample_steps", "2", "--base_seed", "42", "--output_dir", str(cwd), *args]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=cwd, timeout=600)
    assert r.returncode == 0, f"{script} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    return r.stdout


@pytest.mark.parametrize("config", ["config.yaml", "w8a8_all_linears.yaml"])
def test_four_entry_points_chain(tmp_path, config):
    qc = os.path.join(PKG, "quant_configs", config)
    calib = str(tmp_path / "calib.pth")
    run("fp_generate.py", cwd=tmp_path)
    fp = torch.load(tmp_path / "fp_latent_0.pt", weights_only=True)
    assert fp.shape == (16, 2, 60, 104) and torch.isfinite(fp).all()

    run("get_calib_data_wanx.py", "--quant_config", qc, "--calib_data", calib, cwd=tmp_path)
    cd = torch.load(calib, weights_only=True)
    assert "blocks.0.self_attn.q" in cd and cd["blocks.0.self_attn.q"].shape == (1, 1536) and cd["blocks.1.ffn.2"].shape == (1, 8960)
    assert (cd["blocks.0.self_attn.q"] > 0).all()

    run("ptq_wanx.py", "--quant_config", qc, "--calib_data", calib, cwd=tmp_path)
    qp = torch.load(tmp_path / "checkpoint" / "quant_params.pth", weights_only=True)
    e = qp["blocks.0.self_attn.q.w_quantizer"]
    assert e["delta"].shape == (1536, 1) and e["channel_mask"].shape == (1536,) and e["rotation_matrix"] is None
    iw = torch.load(tmp_path / "checkpoint" / "int_weight.pt", weights_only=True)
    assert iw["blocks.0.self_attn.q.weight"].dtype == torch.int8 and "blocks.0.self_attn.q.fp_module.weight" not in iw
    n_q = sum(1 for k in qp if k.endswith("w_quantizer"))
    assert n_q == (6 if config == "config.yaml" else 20)  # q,k,v of 2 blocks vs all ten Linears of 2 blocks

    run("quant_generate.py", "--quant_config", qc, cwd=tmp_path)
    hw = torch.load(tmp_path / "quant_latent_0.pt", weights_only=True)
    run("quant_generate.py", "--quant_config", qc, "--hardware", "false", "--save_file", str(tmp_path / "sim.pt"), cwd=tmp_path)
    sim = torch.load(tmp_path / "sim.pt", weights_only=True)
    assert hw.shape == fp.shape and torch.isfinite(hw).all()
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    print(f"{config}: kernel-mode vs fp {rel(hw, fp):.3e}; simulation-mode vs fp {rel(sim, fp):.3e}; kernel vs simulation {rel(hw, sim):.3e}")
    assert rel(hw, fp) < 0.05 and rel(sim, fp) < 0.05 and rel(hw, sim) < 0.03
    if config == "config.yaml":  # the other solver of the reference's CLI (fm_solvers.py:69): runs, finite, a different trajectory
        run("quant_generate.py", "--quant_config", qc, "--sample_solver", "dpm++", "--save_file", str(tmp_path / "dpm.pt"), cwd=tmp_path)
        dpm = torch.load(tmp_path / "dpm.pt", weights_only=True)
        assert dpm.shape == fp.shape and torch.isfinite(dpm).all() and not torch.equal(dpm, hw)


@pytest.mark.parametrize("config", ["w4a8_mixed.yaml", "w4a8_mixed_viditq.yaml"])
def test_four_entry_points_chain_mixed_precision(tmp_path, config):
    """BASELINE config 5's recipe through the entry scripts on the truncated backbone: FFN weights 4 bit (packed in `int_weight.pt`
    and in HBM), the rest 8, `bitwidth_refactor` driven by the config's regex lists (ptq_wanx.py / quant_generate.py); the second
    config adds the ViDiT mask + rotation on q / k / v AND on the 4-bit ffn.0 / ffn.2 (8960-wide rotation at these dims)."""
    qc = os.path.join(PKG, "quant_configs", config)
    calib = str(tmp_path / "calib.pth")
    run("fp_generate.py", cwd=tmp_path)
    fp = torch.load(tmp_path / "fp_latent_0.pt", weights_only=True)
    run("get_calib_data_wanx.py", "--quant_config", qc, "--calib_data", calib, cwd=tmp_path)
    run("ptq_wanx.py", "--quant_config", qc, "--calib_data", calib, cwd=tmp_path)
    iw = torch.load(tmp_path / "checkpoint" / "int_weight.pt", weights_only=True)
    assert iw["blocks.0.ffn.0.weight"].dtype == torch.uint8 and tuple(iw["blocks.0.ffn.0.weight"].shape) == (8960, 1536 // 2)  # packed nibbles
    assert iw["blocks.1.ffn.2.weight"].dtype == torch.uint8 and iw["blocks.0.self_attn.q.weight"].dtype == torch.int8
    qp = torch.load(tmp_path / "checkpoint" / "quant_params.pth", weights_only=True)
    rotated = [k for k, e in qp.items() if k.endswith("w_quantizer") and e.get("channel_mask") is not None]
    assert len(rotated) == (10 if "viditq" in config else 0)  # q, k, v, ffn.0, ffn.2 of two blocks
    run("quant_generate.py", "--quant_config", qc, cwd=tmp_path)
    hw = torch.load(tmp_path / "quant_latent_0.pt", weights_only=True)
    run("quant_generate.py", "--quant_config", qc, "--hardware", "false", "--save_file", str(tmp_path / "sim.pt"), cwd=tmp_path)
    sim = torch.load(tmp_path / "sim.pt", weights_only=True)
    rel = lambda a, b: ((a - b).norm() / b.norm()).item()  # noqa: E731
    print(f"{config}: kernel-mode vs fp {rel(hw, fp):.3e}; simulation-mode vs fp {rel(sim, fp):.3e}; kernel vs simulation {rel(hw, sim):.3e}")
    assert hw.shape == fp.shape and torch.isfinite(hw).all()
    assert rel(hw, fp) < 0.2 and rel(sim, fp) < 0.2 and rel(hw, sim) < 0.08  # 4-bit FFN weights: the recipe itself is ~1e-1 from FP


def test_four_entry_points_at_headline_size(tmp_path):
    """The same chain at BASELINE config 2's size -- all 30 blocks, 832x480x81f (L = 32760), W8A8 on all 300 Linears with the
    ViDiT transform on q / k / v -- for 3 UniPC steps: the quantized kernel-mode latent after the whole loop stays within 5e-2
    of the FP run's, artefacts have the full-size shapes, and kernel mode loads the exported integer checkpoint."""
    qc = os.path.join(PKG, "quant_configs", "w8a8_all_linears.yaml")
    calib = str(tmp_path / "calib.pth")

    def run_full(script, *args):
        cmd = [sys.executable, os.path.join(PKG, script), "--task", "t2v-1.3B", "--size", "832*480", "--frame_num", "81", "--sample_steps", "3",
               "--base_seed", "42", "--output_dir", str(tmp_path), *args]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, timeout=900)
        assert r.returncode == 0, f"{script} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
        return r.stdout + r.stderr

    run_full("fp_generate.py")
    fp = torch.load(tmp_path / "fp_latent_0.pt", weights_only=True).float()
    assert fp.shape == (16, 21, 60, 104) and torch.isfinite(fp).all()
    run_full("get_calib_data_wanx.py", "--quant_config", qc, "--calib_data", calib)
    cd = torch.load(calib, weights_only=True)
    assert len([k for k in cd if k.startswith("blocks.")]) == 300 and cd["blocks.29.ffn.2"].shape[-1] == 8960
    run_full("ptq_wanx.py", "--quant_config", qc, "--calib_data", calib)
    iw = torch.load(tmp_path / "checkpoint" / "int_weight.pt", weights_only=True)
    assert iw["blocks.29.ffn.0.weight"].shape == (8960, 1536) and iw["blocks.29.ffn.0.weight"].dtype == torch.int8
    log = run_full("quant_generate.py", "--quant_config", qc)
    q = torch.load(tmp_path / "quant_latent_0.pt", weights_only=True).float()
    rel = ((q - fp).norm() / fp.norm()).item()
    print(f"headline size, 3 steps: quantized kernel mode vs FP rel L2 {rel:.3e}")
    assert q.shape == fp.shape and torch.isfinite(q).all() and rel < 5e-2
    assert "int_weight" in log  # the kernel-mode blocks were loaded from the exported checkpoint


def test_quant_generate_two_ranks_ulysses_and_dit_fsdp(tmp_path):
    """The entry script itself under two ranks (one-GPU rehearsal: gloo rendezvous, host-staged collectives): `quant_generate.py
    --ulysses_size 2 --dit_fsdp` -- sequence-parallel kernel-mode blocks with their integer weights sharded over the two ranks --
    reproduces the single-rank latent of the same checkpoint bit for bit."""
    import socket

    qc = os.path.join(PKG, "quant_configs", "w8a8_all_linears.yaml")
    calib = str(tmp_path / "calib.pth")
    run("fp_generate.py", cwd=tmp_path)
    run("get_calib_data_wanx.py", "--quant_config", qc, "--calib_data", calib, cwd=tmp_path)
    run("ptq_wanx.py", "--quant_config", qc, "--calib_data", calib, cwd=tmp_path)
    run("quant_generate.py", "--quant_config", qc, "--save_file", str(tmp_path / "one.pt"), cwd=tmp_path)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(PKG, "quant_generate.py"), "--task", "t2v-1.3B", "--size", "832*480", "--frame_num", "5",
           "--num_layers", "2", "--sample_steps", "2", "--base_seed", "42", "--output_dir", str(tmp_path), "--quant_config", qc,
           "--ulysses_size", "2", "--dit_fsdp", "--save_file", str(tmp_path / "two.pt")]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, timeout=600, env=dict(os.environ, OMP_NUM_THREADS="4", WANQ_REHEARSE_ON_ONE_GPU="1"))
    assert r.returncode == 0, f"two-rank quant_generate failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"
    assert "dit_fsdp:" in r.stdout + r.stderr and "cfg1xsp2" in r.stdout + r.stderr
    one = torch.load(tmp_path / "one.pt", weights_only=True)
    two = torch.load(tmp_path / "two.pt", weights_only=True)
    assert torch.equal(one, two)


def test_fp_generate_and_calibration_two_ranks_ulysses(tmp_path):
    """`fp_generate.py --ulysses_size 2` and `get_calib_data_wanx.py --ulysses_size 2` under two ranks (one-GPU rehearsal): the FP
    latent and the calibration file against the single-rank runs of the same scripts (bf16 GEMMs on token shards: close, not
    bit-equal), and the sharded calibration file feeding ptq_wanx.py."""
    import socket

    qc = os.path.join(PKG, "quant_configs", "w8a8_all_linears.yaml")
    run("fp_generate.py", "--save_file", str(tmp_path / "fp_one.pt"), cwd=tmp_path)
    run("get_calib_data_wanx.py", "--quant_config", qc, "--calib_data", str(tmp_path / "calib_one.pth"), cwd=tmp_path)

    def two_ranks(script, *extra):
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(PKG, script), "--task", "t2v-1.3B", "--size", "832*480", "--frame_num", "5",
               "--num_layers", "2", "--sample_steps", "2", "--base_seed", "42", "--output_dir", str(tmp_path), "--ulysses_size", "2", *extra]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp_path, timeout=600,
                           env=dict(os.environ, OMP_NUM_THREADS="4", WANQ_REHEARSE_ON_ONE_GPU="1"))
        assert r.returncode == 0, f"two-rank {script} failed:\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}"

    two_ranks("fp_generate.py", "--save_file", str(tmp_path / "fp_two.pt"))
    one, two = torch.load(tmp_path / "fp_one.pt", weights_only=True), torch.load(tmp_path / "fp_two.pt", weights_only=True)
    assert one.shape == two.shape and float((one - two).norm() / one.norm()) < 2e-2
    two_ranks("get_calib_data_wanx.py", "--quant_config", qc, "--calib_data", str(tmp_path / "calib_two.pth"))
    c1, c2 = torch.load(tmp_path / "calib_one.pth", weights_only=True), torch.load(tmp_path / "calib_two.pth", weights_only=True)
    assert set(c1) == set(c2) and len(c1) > 20
    worst = max(float(((c1[k] - c2[k]).abs().max() / c1[k].abs().max().clamp_min(1e-6))) for k in c1)
    assert worst < 5e-2, worst
    run("ptq_wanx.py", "--quant_config", qc, "--calib_data", str(tmp_path / "calib_two.pth"), cwd=tmp_path)
