"""Kernels of this library running BESIDE each other (another stream) must not change each other's results.  Regression test of
profiles/r04_z_corun_corruption.txt: with the attention-map passes on a side stream, RMSNorm+RoPE on the main stream returned one
wrong element per 16-byte chunk in the last 16 lanes of a row in ~0.4 % of launches (a packed multiply reading 0 while the wave's other load was returning)."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "wan2.1-quantization_amd"))

pytestmark = pytest.mark.gpu


def test_rmsnorm_rope_and_attention_beside_the_attention_map_kernels_are_bit_stable():
    from wan import ops

    H, L, d = 4, 270, 128
    g = torch.Generator().manual_seed(3)
    q_raw, k_raw, v = (torch.randn(L, H * d, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    w = (1.0 + 0.1 * torch.randn(H * d, generator=g)).cuda()
    rope = torch.randn(L, d // 2, 2, generator=g).cuda()
    rope = rope / rope.norm(dim=-1, keepdim=True)

    # the q / k / v transform (LayerNorm + modulation + ViDiT scale / rotate + quantise, rotate.hip) and the 8960-wide transform of
    # ffn.2 (rotate140.hip) are the other kernels with packed fp32 ops of the op_sel form that failed (tools/isa_lint.py keeps
    # them away from load destinations with loads in flight; this keeps them in the hardware regression too: ADVICE r4)
    import viditq_extension.fused as fused
    from qdiff.quarot import quarot_utils as qu

    C, F = 1536, 8960
    xs = (torch.randn(L, C, generator=g) * 2 + 0.1).cuda()
    sh, sc = (torch.randn(1, C, generator=g) * 0.2).cuda(), (torch.randn(1, C, generator=g) * 0.2).cuda()
    pms = [((torch.rand(C, generator=g) + 0.5) * (torch.randint(0, 2, (C,), generator=g) * 2 - 1)).cuda() for _ in range(3)]
    rotC, rotF = qu.kernel_rotation_params(C, "cuda"), qu.kernel_rotation_params(F, "cuda")
    xf = torch.randn(64, F, generator=g).to(torch.bfloat16).cuda()
    pmF = ((torch.rand(F, generator=g) + 0.5) * (torch.randint(0, 2, (F,), generator=g) * 2 - 1)).cuda()

    def transforms():
        qs = [torch.empty(L, C, dtype=torch.int8, device="cuda") for _ in range(3)]
        scales, sums = [torch.zeros(L, device="cuda") for _ in range(3)], [torch.zeros(L, device="cuda") for _ in range(3)]
        fused.layernorm_rotate_quant_multi(qs, xs, None, sh, sc, pms, rotC, sums, scales, 1e-6)
        sF, uF = torch.zeros(64, device="cuda"), torch.zeros(64, device="cuda")
        qF = fused.rotate_quant(xf, pmF, rotF, uF, sF)
        return torch.cat([t.float().flatten() for t in qs + scales + sums + [qF, sF, uF]])

    def sequence(plain):
        q, k = q_raw.clone(), k_raw.clone()
        t = transforms() if plain else None
        ops.rmsnorm_rope_(q, w, rope, d)
        ops.rmsnorm_rope_(k, w, rope, d)
        o = ops.attention(q, k, v, H, L) if plain else ops.attention_map_quant(q, k, v, H, 8, False, L, q_len=L)
        return q, k, o, t

    q0, k0, o0, t0 = (t.clone() for t in sequence(True))
    m0 = sequence(False)[2].clone()
    side, bad = torch.cuda.Stream(), 0
    for _ in range(300):  # 3000 sequences: ~12 corrupted ones at the rate measured before the fix
        with torch.cuda.stream(side):
            beside = [sequence(False) for _ in range(10)]
        mine = [sequence(True) for _ in range(10)]
        torch.cuda.synchronize()
        bad += sum(int(not (torch.equal(q, q0) and torch.equal(k, k0) and torch.equal(o, o0) and torch.equal(t, t0))) for q, k, o, t in mine)
        bad += sum(int(not (torch.equal(q, q0) and torch.equal(k, k0) and torch.equal(o, m0))) for q, k, o, _ in beside)
    assert bad == 0, f"{bad} of 6000 sequences differ"
