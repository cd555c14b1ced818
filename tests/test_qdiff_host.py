"""CPU tests of the qdiff host logic: config objects, the module visitor, layer-type selection by regex, the
Hadamard construction against the golden products of the reference's tables, kernel rotation parameters."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from qdiff import config as qcfg
from qdiff.quarot import quarot_utils as qu
from qdiff.utils import apply_func_to_submodules

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config_objects_and_shipped_yaml():
    cfg = qcfg.load(os.path.join(ROOT, "wan2.1-quantization_amd", "quant_configs", "config.yaml"))
    assert cfg.weight.n_bits == 8 and cfg.weight.sym is False and cfg.act.sym is True
    assert abs(cfg.viditq.alpha - 0.5665) < 1e-9 and cfg.viditq.layer_name_regex == ""
    assert cfg.get("smooth_quant", None) is None
    mp = qcfg.create({"n_bits": [4, 8], "i_bitwidth": 1})
    assert isinstance(mp.n_bits, qcfg.ListConfig) and not isinstance(mp.n_bits, list) and mp.n_bits[1] == 8


def test_remain_fp_regex_of_shipped_config_selects_only_self_attn_qkv():
    import re

    cfg = qcfg.load(os.path.join(ROOT, "wan2.1-quantization_amd", "quant_configs", "config.yaml"))
    rgx = re.compile(cfg.remain_fp_regex)
    quantized = [n for n in ["blocks.0.self_attn.q", "blocks.12.self_attn.k", "blocks.29.self_attn.v", "blocks.0.self_attn.o",
                             "blocks.3.cross_attn.q", "blocks.3.ffn.0", "blocks.3.ffn.2", "text_embedding.0", "head.head",
                             "time_projection.1"] if not rgx.search(n)]
    assert quantized == ["blocks.0.self_attn.q", "blocks.12.self_attn.k", "blocks.29.self_attn.v"]


def test_apply_func_to_submodules_names_and_replacement():
    m = nn.Sequential(nn.Linear(4, 4), nn.Sequential(nn.Linear(4, 4), nn.ReLU()))
    seen = {}
    apply_func_to_submodules(m, nn.Linear, lambda sub, full_name: full_name, return_d=seen, full_name=None)
    assert seen == {"0": "0", "1.0": "1.0"}

    def swap(sub, name, parent_module, full_name):
        setattr(parent_module, name, nn.Identity())

    apply_func_to_submodules(m, nn.Linear, swap, name=None, parent_module=None, full_name=None)
    assert isinstance(m[0], nn.Identity) and isinstance(m[1][0], nn.Identity)


def test_layer_type_selection():
    from qdiff.base.quant_model import pick_layer_type

    base = {"weight": {"n_bits": 8, "sym": False}, "act": {"n_bits": 8, "sym": True}}
    assert pick_layer_type(qcfg.create(base), "blocks.0.ffn.0").__name__ == "QuantizedLinear"
    c = qcfg.create(dict(base, viditq={"alpha": 0.5, "layer_name_regex": "self_attn"}, smooth_quant={"alpha": 0.5, "layer_name_regex": ""}))
    assert pick_layer_type(c, "blocks.0.self_attn.q").__name__ == "ViDiTQuantizedLinear"
    assert pick_layer_type(c, "blocks.0.ffn.0").__name__ == "SQQuantizedLinear"


@pytest.mark.parametrize("n", [96, 1536, 5120, 8960, 13824])
def test_torch_hadamard_matches_reference_products(golden, n):
    g = golden(f"a5_hadamard_{n}")
    _, K = qu.get_hadK(n)
    assert K == int(g["K"])
    hx = qu.matmul_hadU(torch.from_numpy(g["x"]).double())  # (13824: repo-defined K = 108, the reference's own table; see get_hadK)
    np.testing.assert_allclose(hx.numpy(), g["hadU_x"], rtol=0, atol=1e-12)
    if "xR" in g:
        R = qu.random_hadamard_matrix(n, "cpu", torch.from_numpy(g["signs"]))
        np.testing.assert_allclose(torch.from_numpy(g["x"]) @ R, g["xR"], rtol=0, atol=1e-12)


def test_hadamard_size_rules():
    with pytest.raises(AssertionError):
        qu.get_hadK(13824, strict=True)  # the reference to the letter asserts
    H, K = qu.get_hadK(13824)            # repo-defined: its own K = 108 branch
    assert K == 108 and torch.equal(H @ H.T, 108 * torch.eye(108, dtype=torch.float64))
    k, h = qu.kernel_rotation_params(13824, "cpu")  # 108 x 128: its own kernel (csrc/rotate108.hip)
    assert k == 108 and h.shape == (108, 108)
    k, h = qu.kernel_rotation_params(8960, "cpu")  # 140 x 64: its own kernel (csrc/rotate140.hip), Paley matrix of order 140
    assert k == 140 and h.shape == (140, 140) and torch.equal(h @ h.T, 140 * torch.eye(140))
    assert qu.kernel_rotation_params(140 * 32, "cpu") is None  # any other block < 128: no fused kernel
    k, h = qu.kernel_rotation_params(1536, "cpu")
    assert k == 12 and h.shape == (12, 12) and torch.equal(h @ h.T, 12 * torch.eye(12))
    k, h = qu.kernel_rotation_params(4096, "cpu")
    assert k == 32 and torch.equal(h @ h.T, 32 * torch.eye(32))
    # (H_K' (x) H_128) is the same operator as the reference's (hadK (x) H_m)
    x = torch.randn(3, 4096, dtype=torch.float64, generator=torch.Generator().manual_seed(0))
    ref = qu.matmul_hadU(x)
    v = x.view(3, 32, 128)
    H128 = torch.from_numpy(qu.sylvester(128)).double()
    mine = torch.einsum("kj,bjm->bkm", h.double(), v @ H128.T).reshape(3, 4096) / torch.tensor(4096).sqrt().item()
    torch.testing.assert_close(mine, ref, rtol=0, atol=1e-12)
