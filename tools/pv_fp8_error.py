#!/usr/bin/env python3
"""Host evaluation (no GPU) of an fp8 P.V for the flash-attention kernel: VERDICT r4 item 8 / SURVEY 8(f)1 back end.

The candidate: O = P~ . V~ on v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 x e4m3 -> fp32, twice the bf16 rate per clock,
MI355X_MICROARCH.md "Matrix cores"), P~ = e4m3(2^k * exp2(s - m)) with the running reference m of the lazy rescale (P <= 2^6 between
rescales) and a constant power-of-two scale 2^k riding in the instruction's block scale, V~ = e4m3(V) with unit scale; row sums from
the SAME rounded P~ (as the shipped kernel sums its bf16-rounded P); fp32 accumulation.

This script computes, on the seven shapes of tests/test_gpu_attn_qk8.py::test_attention_qk8_vs_fp32_definition (ragged, masked,
cross-like, long-key and the cfg-A size) with the tests' data (randn q, k, v), the relative Frobenius error of O against the fp32
definition softmax(Q K^T / sqrt(d)) V for: bf16 P.V (what ships), e4m3 P with bf16 V, bf16 P with e4m3 V, and e4m3 x e4m3 -- the only
one the 2x instruction can run.  The build rule was: a kernel only if rel-Frobenius <= 2e-2.
    python tools/pv_fp8_error.py"""
import math
import sys

import torch

torch.set_grad_enabled(False)
SHAPES = [(256, 256, 2, None), (300, 333, 3, None), (513, 700, 2, 650), (64, 40, 4, None), (1000, 4096, 2, None), (520, 4100, 1, 4000),
          (4680, 4680, 12, None)]
E4M3 = torch.float8_e4m3fn


def rnd(x, fmt, scale=1.0):
    if fmt == "f32":
        return x
    if fmt == "bf16":
        return x.to(torch.bfloat16).float()
    return (x * scale).clamp(-448.0, 448.0).to(E4M3).float() / scale


def attention(q, k, v, klen, p_fmt, v_fmt, p_scale, tile=64, thr=6.0):
    """Tile-wise online softmax as csrc/attention.hip runs it (64-key tiles, lazy rescale when a row's maximum grows by > 2^thr),
    with P and V rounded to the given formats in front of the P.V product; fp32 everywhere else."""
    Lq, H, d = q.shape
    Lk = k.shape[0] if klen is None else klen
    c = math.log2(math.e) / math.sqrt(d)
    out = torch.empty(Lq, H, d)
    for h in range(H):
        s = (q[:, h].double() @ k[:Lk, h].double().T).float() * c      # exp2 domain
        vv = rnd(v[:Lk, h], v_fmt)
        m = torch.zeros(Lq)
        o = torch.zeros(Lq, d)
        l = torch.zeros(Lq)
        for j0 in range(0, Lk, tile):
            st = s[:, j0:j0 + tile]
            mx = st.amax(dim=1)
            if j0 == 0:
                m = mx.clone()
            else:
                grow = mx - m
                need = grow > thr
                if need.any():  # (the kernel votes per wave of 32 queries; per row is the same arithmetic on the rows that move)
                    delta = torch.where(need, grow.clamp_min(0), torch.zeros_like(grow))
                    alpha = torch.exp2(-delta)
                    o, l, m = o * alpha[:, None], l * alpha, m + delta
            p = rnd(torch.exp2(st - m[:, None]), p_fmt, p_scale)
            o = o + p @ vv[j0:j0 + tile]
            l = l + p.sum(dim=1)
        out[:, h] = o / l[:, None]
    return out


def main():
    print("rel-Frobenius error of O against the fp32 definition (randn q, k, v; head_dim 128; 64-key tiles, lazy rescale 2^6)")
    print(f"{'Lq x Lk x H':>18s} {'bf16 P.V':>10s} {'e4m3 P, bf16 V':>15s} {'bf16 P, e4m3 V':>15s} {'e4m3 x e4m3 (k=2)':>18s} {'(k=0)':>9s}")
    worst = 0.0
    for (Lq, Lk, H, klen) in SHAPES:
        g = torch.Generator().manual_seed(Lq + Lk)
        q, k, v = (torch.randn(L, H, 128, generator=g).to(torch.bfloat16).float() for L in (Lq, Lk, Lk))
        if (Lq, Lk) == (4680, 4680):  # the cfg-A size: four of the twelve heads bound the host time
            q, k, v, H = q[:, :4], k[:, :4], v[:, :4], 4
        ref = attention(q, k, v, klen, "f32", "f32", 1.0)
        row = []
        for pf, vf, ps in (("bf16", "bf16", 1.0), ("e4m3", "bf16", 4.0), ("bf16", "e4m3", 1.0), ("e4m3", "e4m3", 4.0), ("e4m3", "e4m3", 1.0)):
            o = attention(q, k, v, klen, pf, vf, ps)
            row.append(((o - ref).norm() / ref.norm()).item())
        worst = max(worst, row[3])
        print(f"{Lq:>6d} x {Lk:>5d} x {H:<2d} {row[0]:10.2e} {row[1]:15.2e} {row[2]:15.2e} {row[3]:18.2e} {row[4]:9.2e}", flush=True)
    print(f"worst e4m3 x e4m3: {worst:.2e}  (build rule: <= 2e-2)")
    return 0 if worst <= 2e-2 else 1


if __name__ == "__main__":
    sys.exit(main())
