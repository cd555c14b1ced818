"""ctypes binding of libwanq_hip.so (the C ABI declared in include/wanq_hip.h).

The product path has NO fallback: if the HIP library is missing, importing this module raises, and
every operator refuses non-GPU tensors.  torch is used only for device memory and streams.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# WANQ_LIB: another build of the same library, for whole-step A/B runs of kernel variants.  Honoured ONLY for a file inside this
# package's own lib/ directory (lib/variants/*.so from tools/probes/attn_variants_build.sh) -- any other path is refused, so that a
# stray environment variable cannot make the product load a foreign binary -- and announced with a warning.
_LIBDIR = os.path.join(os.path.dirname(_HERE), "lib")
LIB_PATH = os.path.join(_LIBDIR, "libwanq_hip.so")
if os.environ.get("WANQ_LIB"):
    import warnings

    _over = os.path.realpath(os.environ["WANQ_LIB"])
    if os.path.commonpath([_over, os.path.realpath(_LIBDIR)]) != os.path.realpath(_LIBDIR):
        raise ImportError(f"WANQ_LIB={os.environ['WANQ_LIB']} is outside {_LIBDIR}: refused (the override exists for A/B builds under lib/variants/ only)")
    LIB_PATH = _over
    warnings.warn(f"WANQ_LIB overrides the hot-path library: loading {LIB_PATH} (A/B and diagnostic builds only)", RuntimeWarning)

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"libwanq_hip.so not found at {LIB_PATH}: build it with "
        f"`python wan2.1-quantization_amd/build.py` (hipcc --offload-arch=gfx950). "
        "There is no CPU or PyTorch fallback for the quantized hot path.")

lib = ctypes.CDLL(LIB_PATH)

F16, BF16, F32, I32, I16 = 0, 1, 2, 3, 4
EPI_GELU, EPI_GATE_RES = 1, 2
ABI_VERSION = 6

_vp, _i, _i64, _f = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> argtypes, exactly include/wanq_hip.h
PROTOTYPES = {
    "wanq_quant_rows": [_vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp],
    "wanq_quant_rows_levels": [_vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _f, _vp],
    "wanq_layernorm_rows": [_vp, _i, _vp, _vp, _vp, _i, _i64, _i64, _f, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _vp],
    "wanq_gate_residual": [_vp, _i, _vp, _i, _i64, _vp, _i, _vp, _i, _i64, _i, _i64, _vp],
    "wanq_gemm_w8a8": [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i64, _i, _i, _vp],
    "wanq_gemm_w4a8": [_vp, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i, _i64, _i, _i, _vp],
    "wanq_gemm_select_kernel": [_i],
    "wanq_lincomb": [_i, _i, _vp, _vp, _vp, _i64, _vp],
    "wanq_linear_f32": [_vp, _i64, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp],
    "wanq_time_sinusoid": [_vp, _i, _vp, _i, _i, _vp],
    "wanq_patch_embed": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i64, _vp],
    "wanq_head_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _f, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "wanq_col_absmax": [_vp, _i, _vp, _i64, _i, _vp],
    "wanq_row_minmax": [_vp, _i, _vp, _vp, _vp, _i64, _i, _vp],
    "wanq_weight_quant": [_vp, _i, _vp, _vp, _i, _i, _vp, _vp, _i64, _i, _vp],
    "wanq_weight_export_f16": [_vp, _i, _vp, _vp, _vp, _i64, _i, _vp],
    "wanq_rmsnorm_rope": [_vp, _i, _vp, _vp, _vp, _i, _i64, _i, _i, _i64, _i64, _f, _vp],
    "wanq_fake_quant_cols": [_vp, _i, _vp, _vp, _i, _i, _i64, _i, _vp],
    "wanq_fake_quant_with_delta": [_vp, _i, _vp, _vp, _vp, _i, _i, _i64, _vp],
    "wanq_rmsnorm_rope_scatter": [_vp, _i, _vp, _vp, _vp, _i, _vp, _i64, _i, _i, _i64, _i64, _f, _vp],
    "wanq_rmsnorm_rope_q8": [_vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i64, _i, _i, _i64, _i64, _f, _vp],
    "wanq_attention_qk8_fwd": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i, _i64, _i64, _i, _i, _i64, _i64, _i64, _i64, _f, _i, _vp, _i64, _vp],
    "wanq_rotate_quant_rows": [_vp, _i, _vp, _i, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _vp],
    "wanq_layernorm_rotate_quant_rows": [_vp, _i, _vp, _vp, _vp, _i, _i64, _i64, _f, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _vp],
    "wanq_layernorm_rotate_quant_rows_multi": [_vp, _i, _vp, _vp, _vp, _i, _i64, _i64, _f, _i, _vp, _i, _vp, _vp, _vp, _i, _i64, _i, _vp],
    "wanq_pack_w4": [_vp, _vp, _i, _i64, _i, _vp],
    "wanq_unpack_w4": [_vp, _vp, _i, _i64, _i, _vp],
    "wanq_attention_fwd": [_vp, _vp, _vp, _vp, _i, _i64, _i64, _i, _i, _i64, _i64, _i64, _i64, _f, _vp],
    "wanq_attention_fwd_split": [_vp, _vp, _vp, _vp, _i, _i64, _i64, _i, _i, _i64, _i64, _i64, _i64, _f, _i, _vp, _i64, _vp],
    "wanq_attention_split_workspace": [_i64, _i, _i, _i],
    "wanq_attention_select_form": [_i64],
    "wanq_attention_map_workspace": [_i64, _i64, _i],
    "wanq_attention_map_quant_fwd": [_vp, _vp, _vp, _vp, _i, _i64, _i64, _i, _i, _i64, _i64, _i64, _i64, _f, _i, _i, _vp, _i64, _vp],
    "wanq_attention_map_quant_qk8_fwd": [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _i, _i64, _i64, _i, _i, _i64, _i64, _i64, _i64, _f, _i, _i,
                                         _vp, _i64, _vp],
}
for _name, _args in PROTOTYPES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header and library out of sync
    _fn.argtypes = _args
    _fn.restype = ctypes.c_int
lib.wanq_last_error.restype = ctypes.c_char_p
lib.wanq_attention_split_workspace.restype = ctypes.c_int64
lib.wanq_attention_select_form.restype = ctypes.c_int64
lib.wanq_attention_map_workspace.restype = ctypes.c_int64
lib.wanq_abi_version.restype = ctypes.c_int
if lib.wanq_abi_version() != ABI_VERSION:
    raise ImportError(f"libwanq_hip.so ABI {lib.wanq_abi_version()} != binding ABI {ABI_VERSION}; rebuild")

_DT = {torch.float16: F16, torch.bfloat16: BF16, torch.float32: F32, torch.int32: I32, torch.int16: I16}


def dt(t_or_dtype):
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    try:
        return _DT[d]
    except KeyError:
        raise RuntimeError(f"unsupported dtype {d}") from None


def ptr(t):
    return None if t is None else t.data_ptr()


def ptr_array(tensors):
    """Host array of device pointers (`T* const*` arguments); NULL for None entries."""
    return (ctypes.c_void_p * len(tensors))(*[None if t is None else t.data_ptr() for t in tensors])


def stream():
    return torch.cuda.current_stream().cuda_stream


_call_timer = None


def set_call_timer(t):
    """bench.py hook: when set to a dict, every launch whose wrapper states its algorithmic HBM bytes (`hbm=(tag, bytes)`) appends
    (start_event, end_event, bytes) to t[tag], the events bracketing exactly that launch on the current stream."""
    global _call_timer
    _call_timer = t


def call(name, *args, hbm=None):
    timed = _call_timer is not None and hbm is not None
    if timed:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError(lib.wanq_last_error().decode() or f"{name} failed with code {rc}")
    if timed:
        e.record()
        _call_timer.setdefault(hbm[0], []).append((s, e, int(hbm[1])))


# --- argument checks in the spirit of the reference's TORCH_CHECK macros (K/csrc/utils.cuh:4-22):
#     violations surface as RuntimeError, never as a process abort.
def check_gpu(name, t):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"Tensor {name} must be on the GPU (no CPU fallback exists for this operator)")


def check_contig(name, t):
    if not t.is_contiguous():
        raise RuntimeError(f"Tensor {name} must be contiguous")


def check_dtype(name, t, *dtypes):
    if t.dtype not in dtypes:
        raise RuntimeError(f"Tensor {name} must have dtype in {dtypes}, got {t.dtype}")


def check_shape(name, t, *shape):
    if tuple(t.shape) != tuple(shape):
        raise RuntimeError(f"Tensor {name} must have shape {tuple(shape)}, got {tuple(t.shape)}")


def check_same_device(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"tensors on different devices: {dev} vs {t.device}")
