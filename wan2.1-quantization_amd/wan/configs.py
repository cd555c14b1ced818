"""Model / size configurations (values of ViDiT-Q/examples/Wan2.1/wan/configs/{wan_t2v_1_3B,wan_t2v_14B,
shared_config,__init__}.py; plain dicts instead of EasyDict)."""
import torch

_SHARED = dict(model_type="t2v", patch_size=(1, 2, 2), text_len=512, in_dim=16, out_dim=16, freq_dim=256,
               text_dim=4096, window_size=(-1, -1), qk_norm=True, cross_attn_norm=True, eps=1e-6,
               param_dtype=torch.bfloat16, num_train_timesteps=1000, sample_fps=16, vae_stride=(4, 8, 8))

WAN_CONFIGS = {
    "t2v-1.3B": dict(_SHARED, dim=1536, ffn_dim=8960, num_heads=12, num_layers=30),
    "t2v-14B": dict(_SHARED, dim=5120, ffn_dim=13824, num_heads=40, num_layers=40),
}

SIZE_CONFIGS = {"720*1280": (720, 1280), "1280*720": (1280, 720), "480*832": (480, 832), "832*480": (832, 480),
                "1024*1024": (1024, 1024)}
MAX_AREA_CONFIGS = {k: v[0] * v[1] for k, v in SIZE_CONFIGS.items()}
SUPPORTED_SIZES = {"t2v-14B": ("720*1280", "1280*720", "480*832", "832*480"), "t2v-1.3B": ("480*832", "832*480")}

MODEL_KEYS = ("model_type", "patch_size", "text_len", "in_dim", "dim", "ffn_dim", "freq_dim", "text_dim", "out_dim",
              "num_heads", "num_layers", "window_size", "qk_norm", "cross_attn_norm", "eps")


def model_kwargs(name):
    cfg = WAN_CONFIGS[name]
    return {k: cfg[k] for k in MODEL_KEYS}


def latent_shape(size, frame_num, vae_stride=(4, 8, 8), z_dim=16):
    """target_shape of WanT2V.generate (wan/text2video.py:166-168): size = (W, H) pixels."""
    w, h = size
    return (z_dim, (frame_num - 1) // vae_stride[0] + 1, h // vae_stride[1], w // vae_stride[2])


def seq_len_for(target_shape, patch_size=(1, 2, 2), sp_size=1):
    """ceil(h*w/(ph*pw) * f / sp) * sp (wan/text2video.py:170-172)."""
    import math

    _, f, h, w = target_shape
    return math.ceil((h * w) / (patch_size[1] * patch_size[2]) * f / sp_size) * sp_size
