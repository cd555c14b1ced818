#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/*.npz.

Run ONLY in the build container (where /root/reference exists):

    python tests/golden/make_golden.py

It imports the reference's fake-quant package `qdiff`
(/root/reference/ViDiT-Q/quant_utils/qdiff) on CPU -- with the omegaconf stand-in
in tests/golden/gen/ -- feeds it seeded inputs and stores inputs + outputs as
small .npz fixtures.  The fixtures (data only) are committed; the reference
itself never travels.  The oracle (oracle/*.py) and the HIP kernels are both
checked against these files.

Which reference entry point produced which file:

  a2_dynamic_*.npz    DynamicQuantizer.quantize/forward     Q/base/base_quantizer.py:101-162
  a1_static_*.npz     StaticQuantizer.{init_quant_params,quantize,forward}  base_quantizer.py:43-99
  a7_mixed_*.npz      MixedPrecisionStaticQuantizer         Q/base/mixed_precision_quantizer.py:56-125
  a3_qlinear.npz      QuantizedLinear.forward               Q/base/quant_layer.py:14-74
  a5_hadamard_*.npz   matmul_hadU / random_hadamard_matrix  Q/quarot/quarot_utils.py:158-192
  a4_viditq.npz       ViDiTQuantizedLinear                  Q/viditq/viditq_quant_layer.py:8-73
  a4_smoothquant_1536.npz / a4_quarot_1536.npz   SQQuantizedLinear / QuarotQuantizedLinear   Q/smooth_quant/sq_quant_layer.py, Q/quarot/quarot_quant_layer.py
  a8_calib.npz        SaveActivationHook default branch     W/get_calib_data_wanx.py:262-267,443-449 ; W/ptq_wanx.py:334-344
  a6_surgery.npz      quant_layer_refactor_ / save_quant_param_dict_ / bitwidth_refactor_ on a Wan-named toy tree   Q/base/quant_model.py:15-172
  a16_qkv_attn.npz    DynamicQuantizer with the q / k / v reshapes of quantized attention   W/models/quant_opensora.py:431-440
  a12_intweight.npz   quantize_and_save_weight_ equation    W/wan/quant_wanx_cuda.py:39-53 (4-line equation applied to a1's delta/zp)
  kbench_*.npz        closed-form ground truths of K/bench/bench_gemm.py:27-29,
                      bench_quant_kernel.py:8-11,24-26, bench_layer_norm_kernel.py:15-16,34-36,47-49
                      (formulas evaluated with torch-CPU fp32 on the benches' own input distributions)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/ViDiT-Q/quant_utils"
sys.path.insert(0, os.path.join(HERE, "gen"))  # omegaconf stand-in
sys.path.insert(0, REF)

from omegaconf import OmegaConf  # noqa: E402  (the stand-in)
from qdiff.base.base_quantizer import DynamicQuantizer, StaticQuantizer  # noqa: E402
from qdiff.base.mixed_precision_quantizer import MixedPrecisionStaticQuantizer  # noqa: E402
from qdiff.base.quant_layer import QuantizedLinear  # noqa: E402
from qdiff.quarot import quarot_utils  # noqa: E402
from qdiff.viditq.viditq_quant_layer import ViDiTQuantizedLinear  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(1)  # keep fp32 reductions in a fixed order


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        v = np.asarray(v)
        if v.dtype == np.int32 and v.size and v.min() >= -128 and v.max() <= 127:
            v = v.astype(np.int8)  # integer codes are stored narrow to keep fixtures small
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB  keys={list(out)}")


def outlier_acts(g, T, C):
    """N(0,1) x lognormal per-channel gains, a few x20 outlier channels."""
    x = torch.randn(T, C, generator=g)
    gain = torch.exp(0.5 * torch.randn(C, generator=g))
    x = x * gain
    idx = torch.randperm(C, generator=g)[: max(1, C // 200)]
    x[:, idx] *= 20.0
    return x


# ----------------------------------------------------------------------------- A2
def gen_a2():
    g = torch.Generator().manual_seed(1234)
    for T, C in [(7, 64), (32, 1536), (5, 5120)]:
        x = outlier_acts(g, T, C)
        x[0] = 0.0  # all-zero row -> delta < eps branch (base_quantizer.py:122-128)
        if T > 2:
            x[1] = torch.rand(C, generator=g) * 1e-7  # |x| tiny -> eps branch again
            x[2] = -x[2].abs()  # all-negative row
        # rows that produce exact .5 ties: absmax 127 => delta == 1, values k+0.5
        if T > 4:
            x[3] = torch.arange(C, dtype=torch.float32).remainder(250) - 124.5
            x[3, 0] = 127.0
        q = DynamicQuantizer(OmegaConf.create({"n_bits": 8, "sym": True}))
        q.module_name = "golden"
        xq = q.quantize(x.clone())
        deq = q.forward(x.clone())
        save(f"a2_dynamic_{T}x{C}", x=x, q=xq.to(torch.int32), delta=q.delta.reshape(-1), dequant=deq)


# ----------------------------------------------------------------------------- A1 / A7 / A12
def weights(g, N, K):
    w = torch.randn(N, K, generator=g) * 0.05
    w[0] = w[0].abs() + 0.01  # all positive row: x_min clamps to 0 (base_quantizer.py:85-86)
    w[1] = -w[1].abs() - 0.01  # all negative row: x_max clamps to 0 (:80-81)
    w[2, :] *= 8.0
    return w


class _CpuCuda:
    """The sym static branch calls `.to("cuda")` (base_quantizer.py:75); map it to a no-op here."""

    def __enter__(self):
        self._to = torch.Tensor.to

        def to(t, *a, **k):
            a = tuple("cpu" if (isinstance(v, str) and v == "cuda") else v for v in a)
            return self._to(t, *a, **k)

        torch.Tensor.to = to

    def __exit__(self, *e):
        torch.Tensor.to = self._to


def gen_a1():
    g = torch.Generator().manual_seed(4321)
    for N, K in [(16, 64), (12, 1536)]:
        w = weights(g, N, K)
        arrs = {"w": w}
        for tag, cfg in [("a8", {"n_bits": 8, "sym": False}), ("a4", {"n_bits": 4, "sym": False}),
                         ("s8", {"n_bits": 8, "sym": True})]:
            q = StaticQuantizer(OmegaConf.create(cfg))
            with _CpuCuda():
                wq = q.quantize(w.clone())
                q.init_done = True
                deq = q.forward(w.clone())
            arrs[f"{tag}_q"] = wq.to(torch.int32)
            arrs[f"{tag}_delta"] = q.delta.reshape(-1)
            arrs[f"{tag}_zp"] = q.zero_point.reshape(-1)
            arrs[f"{tag}_dequant"] = deq
            if tag == "a8":
                # A12: W/wan/quant_wanx_cuda.py:39-53 applied to these params
                fp_w = w.to(torch.float16)
                scale = q.delta.view(-1).to(torch.float16)
                zp = q.zero_point.view(-1).to(torch.float16)
                int_w = torch.clamp(torch.round(fp_w / scale.view(-1, 1)) - zp.view(-1, 1), -128, 127).to(torch.int8)
                arrs["a12_int_weight"] = int_w
                arrs["a12_scale_f16"] = scale
                arrs["a12_zp_f16"] = zp
        save(f"a1_static_{N}x{K}", **arrs)

    # A7 mixed precision, n_bits list [4, 8], both lists materialised by init_quant_params
    w = weights(g, 24, 128)
    cfg = OmegaConf.create({"n_bits": [4, 8], "i_bitwidth": 1, "sym": False})
    q = MixedPrecisionStaticQuantizer(cfg)
    deq8 = q.forward(w.clone())
    q.init_done = True
    dl, zl = q.delta_list.clone(), q.zero_point_list.clone()
    q.bitwidth_refactor(0)
    q.n_levels = 2 ** q.n_bits  # reference leaves n_levels stale after refactor; 4-bit clamp uses 2**4
    deq4 = q.forward(w.clone())
    save("a7_mixed_24x128", w=w, delta_list=dl.squeeze(-1), zp_list=zl.squeeze(-1), dequant8=deq8, dequant4=deq4)


# ----------------------------------------------------------------------------- A3
def gen_a3():
    g = torch.Generator().manual_seed(77)
    lin = torch.nn.Linear(64, 48)
    lin.weight.data = torch.randn(48, 64, generator=g) * 0.1
    lin.bias.data = torch.randn(48, generator=g) * 0.1
    cfg = OmegaConf.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True}})
    ql = QuantizedLinear(64, 48, True, "cpu", cfg, lin)
    ql.a_quantizer.module_name = "golden"
    x = outlier_acts(g, 33, 64).reshape(1, 33, 64)
    y = ql(x)
    save("a3_qlinear", x=x, w=lin.weight.data, b=lin.bias.data, w_dequant=ql.weight.data,
         w_delta=ql.w_quantizer.delta.reshape(-1), w_zp=ql.w_quantizer.zero_point.reshape(-1), y=y)


# ----------------------------------------------------------------------------- A5 / A4
def hadamard_from_signs(s):
    """random_hadamard_matrix (quarot_utils.py:186-192) with the sign draw made explicit."""
    Q = torch.diag(s.to(torch.float64))
    return quarot_utils.matmul_hadU(Q)


def gen_a5_a4():
    g = torch.Generator().manual_seed(99)
    for n in [96, 1536, 5120, 8960]:
        _, K = quarot_utils.get_hadK(n)
        s = (torch.randint(0, 2, (n,), generator=g) * 2 - 1).to(torch.float64)
        x = torch.randn(4, n, generator=g, dtype=torch.float64)
        hx = quarot_utils.matmul_hadU(x)  # fp64 rows
        arrs = dict(signs=s, K=np.int64(K), x=x, hadU_x=hx)
        if n <= 1536:
            R = hadamard_from_signs(s)
            arrs["xR"] = x @ R
            arrs["orth_err"] = (R @ R.T - torch.eye(n, dtype=torch.float64)).abs().max()
            if n == 96:
                arrs["R"] = R
            else:
                arrs["R_rows"] = R[[0, 1, 777, n - 1]]
        save(f"a5_hadamard_{n}", **arrs)

    # 13824 (14B ffn.2 in_features) cannot be rotated by the reference
    try:
        quarot_utils.get_hadK(13824)
        raise SystemExit("expected get_hadK(13824) to assert")
    except AssertionError:
        pass

    # A4: ViDiT layer, in=96 (K=12 x 8), out=48
    n, out = 96, 48
    lin = torch.nn.Linear(n, out)
    lin.weight.data = torch.randn(out, n, generator=g) * 0.1
    lin.weight.data[:, 5] *= 6.0
    lin.bias.data = torch.randn(out, generator=g) * 0.1
    cfg = OmegaConf.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                            "viditq": {"alpha": 0.5665, "layer_name_regex": ""}})
    vl = ViDiTQuantizedLinear(n, out, True, "cpu", cfg, lin)
    vl.a_quantizer.module_name = "golden"
    w_first = vl.weight.data.clone()
    act_mask = outlier_acts(g, 64, n).abs().max(dim=0)[0]
    act_mask[3] = 1e-5
    act_mask = torch.where(act_mask < 1e-3, torch.tensor(1e-3), act_mask)  # ptq_wanx.py:340-341
    vl.get_channel_mask(act_mask)
    s = (torch.randint(0, 2, (n,), generator=g) * 2 - 1).to(torch.float64)
    vl.rotation_matrix = hadamard_from_signs(s)  # get_rotation_matrix() with explicit signs, on CPU
    vl.update_quantized_weight_rotated_and_scaled()
    x = outlier_acts(g, 17, n).reshape(1, 17, n)
    y = vl(x)
    save("a4_viditq", x=x, w=lin.weight.data, b=lin.bias.data, act_mask=act_mask, signs=s,
         channel_mask=vl.channel_mask, w_first=w_first, w_final=vl.weight.data,
         w_delta=vl.w_quantizer.delta.reshape(-1), w_zp=vl.w_quantizer.zero_point.reshape(-1), y=y)


def gen_a4_1536():
    """ViDiT layer at the real Wan hidden size (in=1536 = 12 x 128), small out/tokens to keep the fixture small."""
    g = torch.Generator().manual_seed(1536)
    n, out = 1536, 24
    lin = torch.nn.Linear(n, out)
    lin.weight.data = torch.randn(out, n, generator=g) * 0.05
    lin.weight.data[:, 11] *= 5.0
    lin.bias.data = torch.randn(out, generator=g) * 0.1
    cfg = OmegaConf.create({"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True},
                            "viditq": {"alpha": 0.5665, "layer_name_regex": ""}})
    vl = ViDiTQuantizedLinear(n, out, True, "cpu", cfg, lin)
    vl.a_quantizer.module_name = "golden"
    act_mask = outlier_acts(g, 32, n).abs().max(dim=0)[0]
    act_mask = torch.where(act_mask < 1e-3, torch.tensor(1e-3), act_mask)
    vl.get_channel_mask(act_mask)
    s = (torch.randint(0, 2, (n,), generator=g) * 2 - 1).to(torch.float64)
    vl.rotation_matrix = hadamard_from_signs(s)
    vl.update_quantized_weight_rotated_and_scaled()
    x = outlier_acts(g, 9, n).reshape(1, 9, n)
    # the transformed activation and its integer codes, as forward computes them (viditq_quant_layer.py:62-68)
    xt = torch.matmul((x * vl.channel_mask.reshape(1, 1, n)).double(), vl.rotation_matrix).to(torch.float32).reshape(9, n)
    q = vl.a_quantizer.quantize(xt.clone())
    y = vl(x)
    save("a4_viditq_1536", x=x, w=lin.weight.data, b=lin.bias.data, act_mask=act_mask, signs=s, channel_mask=vl.channel_mask,
         w_final=vl.weight.data, w_delta=vl.w_quantizer.delta.reshape(-1), w_zp=vl.w_quantizer.zero_point.reshape(-1),
         x_rot=xt, x_q=q.to(torch.int32), x_delta=vl.a_quantizer.delta.reshape(-1), y=y)


# ----------------------------------------------------------------------------- A8
def gen_a8():
    g = torch.Generator().manual_seed(5)
    C = 192
    calls = [outlier_acts(g, 40, C).reshape(2, 20, C) for _ in range(3)]
    calls[1][..., 7] = 0.0
    calls[0][..., 7] = 1e-4
    calls[2][..., 7] = -2e-4
    # hook default branch, get_calib_data_wanx.py:262-263
    outs = [c.reshape([-1, C]).abs().max(dim=0)[0] for c in calls]
    stacked = torch.stack(outs, dim=0)  # :448
    act_mask = stacked.max(dim=0)[0]  # ptq_wanx.py:336
    act_mask = torch.where(act_mask < 1e-3, torch.tensor(1e-3), act_mask)  # :340-341
    save("a8_calib", calls=torch.stack(calls), stacked=stacked, act_mask=act_mask)


# ----------------------------------------------------------------------------- kernel bench formulas
def gen_kbench():
    g = torch.Generator().manual_seed(2024)
    # bench_gemm.py:7-29 (smaller M,N,K; same distributions)
    M, N, K = 200, 96, 192
    a = torch.randint(-80, 80, (M, K), generator=g, dtype=torch.int8)
    w = torch.randint(-80, 80, (N, K), generator=g, dtype=torch.int8)
    zp = torch.randint(-10, 10, (N,), generator=g, dtype=torch.int16)
    sa = (0.01 * torch.rand(M, generator=g) + 0.005).to(torch.float16)
    sw = (0.1 * torch.rand(N, generator=g) + 0.1).to(torch.float16)
    bias = (torch.rand(N, generator=g) * 200).to(torch.float16)
    a_sum = (sa.view(-1, 1).float() * a.float()).sum(dim=1).to(torch.float16)
    acc = a.to(torch.int64) @ w.to(torch.int64).T
    y32 = (acc.float() * sa.view(-1, 1).float() * sw.view(1, -1).float()
           + a_sum.view(-1, 1).float() * zp.float().view(1, -1) * sw.view(1, -1).float() + bias.float())
    y_sym32 = acc.float() * sa.view(-1, 1).float() * sw.view(1, -1).float() + bias.float()
    save("kbench_gemm", a=a, w=w, zp=zp, sa=sa, sw=sw, bias=bias, a_sum=a_sum, acc=acc.to(torch.int32),
         y_asym=y32.to(torch.float16), y_asym_f32=y32, y_sym=y_sym32.to(torch.float16))

    # bench_quant_kernel.py:8-11,24-26
    T, C = 24, 1152
    x = torch.randn(T, C, generator=g).to(torch.float16)
    scale = x.abs().max(dim=1).values.float() / 127.0
    q = torch.round(x.float() / scale.view(-1, 1))
    s = (q.to(torch.float16).sum(dim=1) * scale.view(-1)).to(torch.float16)
    gx = torch.nn.functional.gelu(x.float(), approximate="tanh")
    gscale = gx.abs().max(dim=1).values / 127.0
    gq = torch.round(gx / gscale.view(-1, 1))
    save("kbench_quant", x=x, scale=scale, q=q.to(torch.int32), sum=s, gelu_scale=gscale, gelu_q=gq.to(torch.int32))

    # bench_layer_norm_kernel.py:15-16,34-36,47-49
    B, T, C = 2, 8, 1536
    x = torch.randn(1, T, C, generator=g).to(torch.float16).repeat(B, 1, 1)
    wt = torch.randn(C, generator=g).to(torch.float16)
    shift = torch.randn(B, C, generator=g).to(torch.float16)
    scl = torch.randn(B, C, generator=g).to(torch.float16)
    y = torch.nn.functional.layer_norm(x.float(), (C,), weight=wt.float(), eps=1e-5)
    y_t2i = y * (1 + scl.float().view(B, 1, C)) + shift.float().view(B, 1, C)
    sc = y_t2i.view(-1, C).abs().max(dim=1, keepdim=True).values / 127.0
    qy = torch.round(y_t2i.view(-1, C) / sc)
    sm = qy.sum(dim=1) * sc.view(-1)
    save("kbench_layernorm", x=x, weight=wt, shift=shift, scale_msa=scl, ln=y, ln_t2i=y_t2i,
         q_scale=sc.view(-1), q=qy.to(torch.int32), q_sum=sm)


def gen_a16():
    """q / k / v of the reference's quantized attention: its DynamicQuantizer applied with the exact reshapes of
    examples/Wan2.1/models/quant_opensora.py:431-440 -- q and k per (token, head) over head_dim, v per (head, channel) over all
    tokens -- on [B, heads, tokens, head_dim] tensors (the layout of that file)."""
    g = torch.Generator().manual_seed(16)
    BS, H, N, D = 1, 3, 37, 128
    q = torch.randn(BS, H, N, D, generator=g) * torch.exp(0.5 * torch.randn(D, generator=g))
    k = torch.randn(BS, H, N, D, generator=g) * 2.0
    v = torch.randn(BS, H, N, D, generator=g) * torch.exp(torch.randn(H, 1, D, generator=g))
    v[:, 1, :, 7] = 0  # an all-zero (head, channel): the eps rule
    out = {}
    for bits in (8, 4):
        cfg = OmegaConf.create({"n_bits": bits, "sym": True})
        qq, kq, vq = DynamicQuantizer(cfg), DynamicQuantizer(cfg), DynamicQuantizer(cfg)
        for z in (qq, kq, vq):
            z.module_name = "golden"  # only read by the eps branch's log line (the owning layer sets it in the reference)
        out[f"q{bits}"] = qq(q.reshape([-1, D])).reshape([BS, H, N, D])                      # :432
        out[f"k{bits}"] = kq(k.reshape([-1, D])).reshape([BS, H, N, D])                      # :436
        out[f"v{bits}"] = vq(v.permute([0, 1, 3, 2]).reshape([-1, N])).reshape([BS, H, D, N]).permute([0, 1, 3, 2])  # :438-440
    save("a16_qkv_attn", q=q, k=k, v=v, **out)


def gen_a4_sq_quarot():
    """The other two transform layers of the reference at in_features 1536 (one Wan model dimension): SmoothQuant
    (SQQuantizedLinear, Q/smooth_quant/sq_quant_layer.py:6-68: channel mask only) and QuaRot (QuarotQuantizedLinear,
    Q/quarot/quarot_quant_layer.py:7-69: rotation only).  QuaRot's weight update moves the weight `.to("cuda")` (:38): on this
    CPU-only container that one call is redirected to the CPU for the duration of the update."""
    from qdiff.quarot.quarot_quant_layer import QuarotQuantizedLinear
    from qdiff.smooth_quant.sq_quant_layer import SQQuantizedLinear

    g = torch.Generator().manual_seed(44)
    n, out = 1536, 24
    base = {"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True}}

    def lin_():
        lin = torch.nn.Linear(n, out)
        lin.weight.data = torch.randn(out, n, generator=g) * 0.05
        lin.weight.data[:, 11] *= 8.0
        lin.bias.data = torch.randn(out, generator=g) * 0.1
        return lin

    # SmoothQuant
    lin = lin_()
    sq = SQQuantizedLinear(n, out, True, "cpu", OmegaConf.create(dict(base, smooth_quant={"alpha": 0.5, "layer_name_regex": ""})), lin)
    sq.a_quantizer.module_name = "golden"
    act_mask = outlier_acts(g, 64, n).abs().max(dim=0)[0]
    act_mask = torch.where(act_mask < 1e-3, torch.tensor(1e-3), act_mask)
    sq.get_channel_mask(act_mask)
    sq.update_quantized_weight_scaled()
    x = outlier_acts(g, 9, n).reshape(1, 9, n)
    save("a4_smoothquant_1536", x=x, w=lin.weight.data, b=lin.bias.data, act_mask=act_mask, channel_mask=sq.channel_mask,
         w_final=sq.weight.data, w_delta=sq.w_quantizer.delta.reshape(-1), w_zp=sq.w_quantizer.zero_point.reshape(-1), y=sq(x))
    # QuaRot
    lin = lin_()
    qr_ = QuarotQuantizedLinear(n, out, True, "cpu", OmegaConf.create(dict(base, quarot={"layer_name_regex": ""})), lin)
    qr_.a_quantizer.module_name = "golden"
    s = (torch.randint(0, 2, (n,), generator=g) * 2 - 1).to(torch.float64)
    qr_.rotation_matrix = hadamard_from_signs(s)
    real_to = torch.Tensor.to
    torch.Tensor.to = lambda self, *a, **k: real_to(self, *[("cpu" if (isinstance(v, str) and v == "cuda") else v) for v in a], **k)
    try:
        qr_.update_quantized_weight_rotated()
    finally:
        torch.Tensor.to = real_to
    x = outlier_acts(g, 9, n).reshape(1, 9, n)
    y = qr_(x)
    xr = torch.matmul(x.double(), qr_.rotation_matrix).to(x.dtype).reshape(9, n)
    a_q = DynamicQuantizer(OmegaConf.create(base["act"]))
    a_q.module_name = "golden"
    save("a4_quarot_1536", x=x, w=lin.weight.data, b=lin.bias.data, signs=s, w_final=qr_.weight.data,
         w_delta=qr_.w_quantizer.delta.reshape(-1), w_zp=qr_.w_quantizer.zero_point.reshape(-1), y=y, x_q=a_q.quantize(xr), x_delta=a_q.delta.reshape(-1))


def gen_a6_surgery():
    """Model surgery and the quant_param_dict, by the reference's own functions driven with QuantWanModel's keyword arguments
    (W/wan/quant_wanx.py:85-133 -> Q/base/quant_model.py:15-172) on a toy tree that carries Wan's module names.  Stored: which
    Linear became which class under the shipped config.yaml regexes, the key / shape map of save_quant_param_dict after the ViDiT
    initialisation, and -- under a mixed-precision config -- every layer's weight bit-width and quant_mode."""
    import json

    import torch.nn as nn
    from qdiff.base.base_quantizer import BaseQuantizer
    from qdiff.base.quant_model import bitwidth_refactor_, quant_layer_refactor_, save_quant_param_dict_, set_init_done_
    from qdiff.utils import apply_func_to_submodules

    def tree():
        torch.manual_seed(6)

        class Attn(nn.Module):
            def __init__(self):
                super().__init__()
                self.q, self.k, self.v, self.o = (nn.Linear(256, 256) for _ in range(4))

        class Block(nn.Module):
            def __init__(self):
                super().__init__()
                self.self_attn, self.cross_attn = Attn(), Attn()
                self.ffn = nn.Sequential(nn.Linear(256, 512), nn.GELU(approximate="tanh"), nn.Linear(512, 256))

        class Head(nn.Module):
            def __init__(self):
                super().__init__()
                self.head = nn.Linear(256, 64)

        class Toy(nn.Module):
            def __init__(self):
                super().__init__()
                self.text_embedding = nn.Sequential(nn.Linear(64, 256), nn.GELU(approximate="tanh"), nn.Linear(256, 256))
                self.time_embedding = nn.Sequential(nn.Linear(64, 256), nn.SiLU(), nn.Linear(256, 256))
                self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(256, 1536))
                self.blocks = nn.ModuleList([Block(), Block()])
                self.head = Head()
                self.quant_param_dict = {}

        return Toy()

    def refactor(model, cfg):
        apply_func_to_submodules(model, class_type=nn.Linear, function=quant_layer_refactor_, name=None, parent_module=None,
                                 quant_config=cfg, full_name=None, remain_fp_regex=cfg.remain_fp_regex)

    def classes(model):
        return {n: type(m).__name__ for n, m in model.named_modules() if isinstance(m, nn.Linear) or hasattr(m, "w_quantizer")}

    out = {}
    # (1) the shipped Wan config: W8A8, ViDiT on every quantized layer, FP everywhere but self_attn q / k / v
    import yaml
    with open("/root/reference/ViDiT-Q/examples/Wan2.1/quant_configs/config.yaml") as fh:
        wan_cfg = OmegaConf.create(yaml.safe_load(fh))
    m = tree()
    refactor(m, wan_cfg)
    out["wan_config_classes"] = classes(m)
    g = torch.Generator().manual_seed(7)
    for n, mod in m.named_modules():
        if type(mod).__name__ == "ViDiTQuantizedLinear":  # what ptq_wanx.py:334-344 does per layer, on the CPU
            mod.get_channel_mask(torch.rand(mod.in_features, generator=g) + 0.5)
            mod.rotation_matrix = hadamard_from_signs(torch.randint(0, 2, (mod.in_features,), generator=g).double() * 2 - 1)
            mod.update_quantized_weight_rotated_and_scaled()
    apply_func_to_submodules(m, class_type=BaseQuantizer, function=save_quant_param_dict_, full_name=None, parent_module=None, model=m)
    out["wan_config_param_dict"] = {k: {kk: (None if vv is None else list(vv.shape)) for kk, vv in v.items()} for k, v in m.quant_param_dict.items()}
    # (2) mixed precision: weight bit-width list, regex lists with index 0 = FP16
    mp_cfg = OmegaConf.create({"remain_fp_regex": r"text_embedding|time_embedding|time_projection|head\.head",
                               "weight": {"n_bits": [4, 8], "i_bitwidth": 1, "sym": False}, "act": {"n_bits": 8, "sym": True},
                               "mixed_precision": {"weight": {"layer_name_regex": [r"cross_attn\.o", "ffn", ""]},
                                                   "act": {"layer_name_regex": ["", ""]}}})
    m2 = tree()
    refactor(m2, mp_cfg)
    from qdiff.base.quant_layer import QuantizedLinear as QL
    apply_func_to_submodules(m2, class_type=QL, function=bitwidth_refactor_, name=None, parent_module=None, quant_config=mp_cfg, full_name=None)
    out["mixed_classes"] = classes(m2)
    out["mixed_bits"] = {n: {"w_bits": int(mod.w_quantizer.n_bits), "quant_mode": bool(mod.quant_mode)} for n, mod in m2.named_modules()
                         if hasattr(mod, "w_quantizer")}
    save("a6_surgery", json=np.array(json.dumps(out, sort_keys=True)))


if __name__ == "__main__":
    gen_a4_sq_quarot()
    gen_a6_surgery()
    gen_a16()
    gen_a2()
    gen_a1()
    gen_a3()
    gen_a5_a4()
    gen_a4_1536()
    gen_a8()
    gen_kbench()
