"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol the header
declares, refuses bad arguments with a message instead of crashing, and the Python mirror of the
reference's extension modules exposes the reference's names."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wanq_hip.h")
PKG = os.path.join(ROOT, "wan2.1-quantization_amd")
LIB = os.path.join(ROOT, "wan2.1-quantization_amd", "lib", "libwanq_hip.so")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wanq_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import importlib.util

        spec = importlib.util.spec_from_file_location("wanq_build", os.path.join(ROOT, "wan2.1-quantization_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build(verbose=False)
    return ctypes.CDLL(LIB)


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 9
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/wanq_hip.h but not exported"


def test_binding_covers_header(lib):
    from viditq_extension import _C

    assert set(_C.PROTOTYPES) | {"wanq_last_error", "wanq_abi_version"} == set(declared_symbols())


def test_bad_arguments_are_refused_with_message(lib):
    lib.wanq_last_error.restype = ctypes.c_char_p
    rc = lib.wanq_quant_rows(None, 0, None, None, None, 0, ctypes.c_int64(4), 64, 0, 0, None)
    assert rc == 1 and b"non-NULL" in lib.wanq_last_error()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = lib.wanq_quant_rows(p, 0, p, p, None, 0, ctypes.c_int64(4), 60, 0, 0, None)  # cols % 8 != 0
    assert rc == 2 and b"multiple of 8" in lib.wanq_last_error()
    rc = lib.wanq_gemm_w8a8(p, p, p, 0, p, None, 0, p, None, 0, None, 0, None, None, 0, ctypes.c_int64(8), 12, 64, None)
    assert rc == 2 and b"N=12" in lib.wanq_last_error()
    rc = lib.wanq_gemm_w8a8(p, p, p, 0, p, None, 0, p, None, 0, None, 0, None, None, 0, ctypes.c_int64(8), 16, 24, None)
    assert rc == 2 and b"K=24" in lib.wanq_last_error()


def test_round2_entry_points_refuse_bad_arguments(lib):
    """Argument checks of the entry points added in ABI version 2 run on the host, before any launch: return code + message."""
    lib.wanq_last_error.restype = ctypes.c_char_p
    i64 = ctypes.c_int64
    buf = ctypes.create_string_buffer(256)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.wanq_abi_version() == 6
    # the narrow-range quantiser (ABI 6): int8 codes hold at most 127 levels, the floor is not negative
    lib.wanq_quant_rows_levels.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           i64, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
    rc = lib.wanq_quant_rows_levels(p, 2, p, p, None, 2, i64(4), 64, 128, 1e-6, None)
    assert rc == 1 and b"n_levels=128" in lib.wanq_last_error()
    rc = lib.wanq_quant_rows_levels(p, 2, p, p, None, 2, i64(4), 64, 31, -1.0, None)
    assert rc == 1 and b"floor" in lib.wanq_last_error()
    rc = lib.wanq_quant_rows_levels(p, 2, p, p, None, 2, i64(4), 60, 31, 0.0, None)
    assert rc == 2 and b"cols=60" in lib.wanq_last_error()
    # W4A8: K must hold whole 32-code groups
    rc = lib.wanq_gemm_w4a8(p, p, p, 0, p, None, 0, p, None, 0, None, 0, None, None, 0, i64(8), 16, 48, None)
    assert rc == 2 and b"K=48" in lib.wanq_last_error()
    # the 8960 transform: had_k = 140 only goes with cols = 8960; the LayerNorm form has no such variant
    rc = lib.wanq_rotate_quant_rows(p, 2, None, 140, p, 2, None, None, None, 2, i64(4), 1536, None)
    assert rc == 2 and b"8960" in lib.wanq_last_error()
    rc = lib.wanq_layernorm_rotate_quant_rows(p, 2, None, None, None, 2, i64(0), i64(4), ctypes.c_float(1e-6), None, 140, p, p, p, 2, i64(4), 8960, None)
    assert rc == 2 and b"8960" in lib.wanq_last_error()
    # scattered RMSNorm+RoPE store needs its head map
    rc = lib.wanq_rmsnorm_rope_scatter(p, 1, p, None, p, 1, None, i64(4), 256, 128, i64(4), i64(0), ctypes.c_float(1e-6), None)
    assert rc == 1 and b"NULL" in lib.wanq_last_error()
    # column fake-quant: bit-width range
    rc = lib.wanq_fake_quant_cols(p, 2, p, p, 2, 9, i64(4), 64, None)
    assert rc == 1 and b"n_bits=9" in lib.wanq_last_error()
    # fused step: at most 4 outputs of at most 8 inputs, 16-byte aligned tensors of numel % 4 == 0
    arr = (ctypes.c_void_p * 1)(p)
    coef = (ctypes.c_float * 1)(1.0)
    rc = lib.wanq_lincomb(5, 1, coef, arr, arr, i64(16), None)
    assert rc == 1 and b"n_out=5" in lib.wanq_last_error()
    rc = lib.wanq_lincomb(1, 1, coef, arr, arr, i64(6), None)
    assert rc == 2 and b"numel=6" in lib.wanq_last_error()
    # int8 Q.K^T attention: head_dim 128 only
    rc = lib.wanq_attention_qk8_fwd(p, p, i64(64), p, p, i64(64), p, p, 1, i64(64), i64(64), 2, 64, i64(128), i64(128), i64(128), i64(128),
                                    ctypes.c_float(0.1), 1, None, i64(0), None)
    assert rc != 0 and lib.wanq_last_error()


def test_reference_module_surface():
    import viditq_extension.fused as fused
    import viditq_extension.qgemm as qgemm
    from viditq_extension.nn.base import QuantParams  # noqa: F401
    from viditq_extension.nn.layernorm import LayerNormGeneral  # noqa: F401
    from viditq_extension.nn.qlinear import W8A8OF16LinearDynamicInputScale  # noqa: F401

    # K/csrc/fused/pybind.cpp:57-99 and K/csrc/qgemm/pybind.cpp:5-12
    for n in ["quant_sum", "quant_sum_static", "gelu_quant_sum", "layernorm_nobias", "layernorm_nobias_quant_nosum_fuse",
              "layernorm_nobias_quant_sum_fuse", "layernorm_nobias_t2i_fuse", "layernorm_nobias_t2i_quant_sum_fuse",
              "gate_residual_fuse"]:
        assert callable(getattr(fused, n))
    for n in ["w8a8_of16_bias_weight_asym", "w8a8_of16_bias_weight_sym", "w8a8_o32", "w8a8_of16_nobias_weight_sym_qserve"]:
        assert callable(getattr(qgemm, n))


def test_no_cpu_fallback():
    import viditq_extension.fused as fused
    import viditq_extension.qgemm as qgemm

    with pytest.raises(RuntimeError, match="must be on the GPU"):
        fused.quant_sum(torch.zeros(4, 64, dtype=torch.float16), torch.zeros(4, dtype=torch.float16), torch.zeros(4, dtype=torch.float16))
    with pytest.raises(RuntimeError, match="must be on the GPU"):
        qgemm.w8a8_o32(torch.zeros(4, 64, dtype=torch.int8), torch.zeros(8, 64, dtype=torch.int8))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "wan2.1-quantization_amd")
    for d, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_gemm_kernel_selection_is_a_host_side_setting(lib):
    """wanq_gemm_select_kernel (ABI 4) only records which kernel later GEMM calls launch: no GPU needed, bad values refused."""
    prev = lib.wanq_gemm_select_kernel(2)
    assert prev in (0, 1, 2)
    assert lib.wanq_gemm_select_kernel(1) == 2
    assert lib.wanq_gemm_select_kernel(7) == -1 and lib.wanq_gemm_select_kernel(-1) == -1
    assert lib.wanq_gemm_select_kernel(prev) == 1
    lib.wanq_attention_select_form.restype = ctypes.c_int64
    lib.wanq_attention_select_form.argtypes = [ctypes.c_int64]
    prev = lib.wanq_attention_select_form(0)              # which form of the bf16 attention kernel later calls launch (-1 = start-up value)
    assert prev == -1 and lib.wanq_attention_select_form(1 << 40) == 0 and lib.wanq_attention_select_form(-5) == 1 << 40
    assert lib.wanq_attention_select_form(-1) == -1


def test_rehearsal_switch_is_refused_unless_exactly_one_gpu_is_visible(monkeypatch, capsys):
    """The one-GPU rehearsal (gloo + host-staged collectives, tools/one_gpu_rehearsal.py) cannot be tripped on a box that is not a
    one-GPU box: this container shows 0 GPUs, a real node 8 -- both are refused with exit status 4 and one line."""
    import torch
    from wan.distributed import enter_one_gpu_rehearsal

    monkeypatch.delenv("WANQ_REHEARSE_ON_ONE_GPU", raising=False)
    assert enter_one_gpu_rehearsal("WANQ_REHEARSE_ON_ONE_GPU", 2) is False       # not requested
    monkeypatch.setenv("WANQ_REHEARSE_ON_ONE_GPU", "1")
    assert enter_one_gpu_rehearsal("WANQ_REHEARSE_ON_ONE_GPU", 1) is False       # a single rank has nothing to rehearse
    for n in (0, 8):
        monkeypatch.setattr(torch.cuda, "device_count", lambda n=n: n)
        with pytest.raises(SystemExit) as ex:
            enter_one_gpu_rehearsal("WANQ_REHEARSE_ON_ONE_GPU", 2)
        assert ex.value.code == 4 and f"shows {n} GPUs: refused" in capsys.readouterr().err
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert enter_one_gpu_rehearsal("WANQ_REHEARSE_ON_ONE_GPU", 2) is True
    assert "ONE-GPU REHEARSAL" in capsys.readouterr().err
    # the scaffolding itself lives outside the product package
    assert not os.path.exists(os.path.join(PKG, "wan", "distributed", "rehearsal.py"))
    assert os.path.exists(os.path.join(ROOT, "tools", "one_gpu_rehearsal.py"))


def test_wanq_lib_override_is_confined_to_the_package_lib_directory(tmp_path):
    """WANQ_LIB (whole-step A/B runs of kernel variants) loads files under wan2.1-quantization_amd/lib/ only; any other path is an
    ImportError, not a warning."""
    code = "import sys; sys.path.insert(0, %r); import viditq_extension._C" % PKG
    foreign = tmp_path / "libwanq_hip.so"
    foreign.write_bytes(b"")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, WANQ_LIB=str(foreign)), capture_output=True, text=True)
    assert r.returncode != 0 and "refused" in r.stderr
    r = subprocess.run([sys.executable, "-W", "always", "-c", code], env=dict(os.environ, WANQ_LIB=os.path.join(PKG, "lib", "libwanq_hip.so")),
                       capture_output=True, text=True)
    assert r.returncode == 0 and "WANQ_LIB overrides" in r.stderr


def test_header_is_plain_c_and_a_c_program_links_against_the_library(lib, tmp_path):
    """The boundary is a C ABI: include/wanq_hip.h compiles as pedantic C99 and as C++11, and a C program with no torch and no Python
    in it links against lib/libwanq_hip.so, reads the ABI version and gets a refusal (code + message) for a null pointer -- no GPU
    is touched by either call."""
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "wanq_hip.h"\n'
                   "int main(void) {\n"
                   "  int rc = wanq_quant_rows(NULL, WANQ_F32, NULL, NULL, NULL, WANQ_F32, 4, 64, 0, 0, NULL);\n"
                   '  printf("%d %d %d\\n", wanq_abi_version(), rc, (int)strlen(wanq_last_error()));\n'
                   "  return 0;\n}\n")
    inc = os.path.join(ROOT, "include")
    for cc, std in (("gcc", "-std=c99"), ("g++", "-std=c++11")):
        r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-fsyntax-only", "-x", "c" if cc == "gcc" else "c++", str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    exe = tmp_path / "abi"
    libdir = os.path.dirname(LIB)
    r = subprocess.run(["gcc", "-std=c99", "-I", inc, str(src), "-o", str(exe), "-L", libdir, "-lwanq_hip", f"-Wl,-rpath,{libdir}",
                        "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    ver, rc, msg_len = map(int, r.stdout.split())
    assert ver == lib.wanq_abi_version() and rc != 0 and msg_len > 0
