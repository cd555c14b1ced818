"""QuantizedLinear: the base of the SmoothQuant / QuaRot / ViDiT variants (smooth_quant/, quarot/, viditq/), with everything they share.

Interface of ViDiT-Q/quant_utils/qdiff/base/quant_layer.py:8-74 (+ smooth_quant/sq_quant_layer.py,
quarot/quarot_quant_layer.py, viditq/viditq_quant_layer.py): ctor signature, attributes `fp_module`, `fp_weight`,
`w_quantizer`, `a_quantizer`, `quant_mode`, `use_kernel`, `module_name`, `channel_mask`, `rotation_matrix`, methods
`get_channel_mask`, `get_rotation_matrix`, `update_quantized_weight_*`.

What differs: `weight.data` still holds the fake-quantised (dequantised) weight for inspection / serialisation,
but forward() does not use it.  It uses int8 codes of that same weight and runs
    per-token quantise (optionally fused with x*mask -> Hadamard rotate)  ->  int8-MFMA GEMM + dequant epilogue.
The rotation is stored as its +-1 sign vector (`rotation_signs`); `rotation_matrix` materialises on demand."""
import torch
import torch.nn as nn

from viditq_extension import qgemm

from ..config import ListConfig
from ..quarot import quarot_utils
from .base_quantizer import DynamicQuantizer, StaticQuantizer
from .mixed_precision_quantizer import MixedPrecisionDynamicQuantizer, MixedPrecisionStaticQuantizer


class QuantizedLinear(nn.Linear):
    uses_mask = False
    uses_rotation = False

    def __init__(self, in_features, out_features, bias, device, quant_config, fp_module):
        super().__init__(in_features, out_features, bias, device)
        self.fp_module, self.q_cfg = fp_module, quant_config
        self.w_quantizer = self.a_quantizer = None
        self.channel_mask = None
        self.rotation_signs = None
        self._premul = self._rot = None
        self.register_buffer("_codes", None, persistent=False)  # int8 [N,K], or uint8 [N,K/2] packed nibbles (4-bit weights)
        self._zp_gemm = None
        if quant_config.get("weight", None) is not None:
            wq = quant_config["weight"]
            self.w_quantizer = (MixedPrecisionStaticQuantizer if isinstance(wq["n_bits"], ListConfig) else StaticQuantizer)(wq)
            self._requantize(fp_module.weight.data)
            self.w_quantizer.init_done = True
        else:
            self.weight.data = fp_module.weight.data
        self.fp_weight = fp_module.weight
        self.bias = fp_module.bias
        if quant_config.get("act", None) is not None:
            aq = quant_config["act"]
            self.a_quantizer = (MixedPrecisionDynamicQuantizer if isinstance(aq["n_bits"], ListConfig) else DynamicQuantizer)(aq)
        self.use_kernel = False
        self.quant_mode = True
        self.module_name = None

    # ---- weight side --------------------------------------------------------------------------------
    def _requantize(self, w):
        """weight.data <- fake-quant(w) and int_weight <- its integer codes, with the quantizer's current state."""
        codes, deq = self.w_quantizer.codes_and_dequant(w.detach().float())
        self.weight.data = deq
        self.int_weight = codes

    @property
    def packed_w4(self):
        return self.w_quantizer is not None and self.w_quantizer.n_bits == 4 and self.in_features % 32 == 0

    @property
    def int_weight(self):
        """int8 codes [N, K].  4-bit layers keep them PACKED in HBM (two per byte, qgemm.pack_w4 layout); this view unpacks."""
        c = self._codes
        if c is not None and c.dtype == torch.uint8:
            return qgemm.unpack_w4(c, bias=8)
        return c

    @int_weight.setter
    def int_weight(self, codes):
        if codes is not None and self.packed_w4:
            self._codes = qgemm.pack_w4(codes.contiguous(), bias=8)
        else:
            self._codes = codes
        self._zp_gemm = None

    def refresh(self):
        """Re-derive the quantized weight after the quantizer changed (bitwidth_refactor, load): the SAME derivation the
        layer's PTQ step uses -- scale / rotate / double quantisation for the SmoothQuant / QuaRot / ViDiT variants once
        their mask / rotation exist (as load_quant_param_dict_ dispatches) -- so that the integer codes always match the
        transform forward() applies to the activations."""
        if self.w_quantizer is None:
            return
        if self.uses_mask and self.uses_rotation and self.channel_mask is not None and self.rotation_signs is not None:
            self.update_quantized_weight_rotated_and_scaled()
        elif self.uses_rotation and not self.uses_mask and self.rotation_signs is not None:
            self.update_quantized_weight_rotated()
        elif self.uses_mask and not self.uses_rotation and self.channel_mask is not None:
            self.update_quantized_weight_scaled()
        elif (self.uses_mask and self.channel_mask is not None) or (self.uses_rotation and self.rotation_signs is not None):
            raise NotImplementedError(f"{type(self).__name__}: transform is only partly initialised (mask / rotation)")
        else:
            self.w_quantizer.init_done = True
            self._requantize(self.fp_module.weight.data.float())

    # ---- activation side ----------------------------------------------------------------------------
    def _act_transform(self):
        """(premul fp32 [C] or None, (had_k, hadk) or None) for the fused quantiser; cached."""
        if self._premul is None and (self.channel_mask is not None or self.rotation_signs is not None):
            dev = self.fp_module.weight.device
            pm = torch.ones(self.in_features, device=dev)
            if self.uses_mask and self.channel_mask is not None:
                pm = pm * self.channel_mask.float().to(dev)
            if self.uses_rotation and self.rotation_signs is not None:
                pm = pm * self.rotation_signs.float().to(dev)
                rot = quarot_utils.kernel_rotation_params(self.in_features, dev)
                if rot is None:
                    raise NotImplementedError(f"no fused rotation for in_features={self.in_features}")
                self._rot = rot
            self._premul = pm.contiguous()
        return self._premul, self._rot

    def forward(self, x, *args, **kwargs):
        """x: [B, N_token, C] (or [tokens, C]) on the GPU."""
        if not self.quant_mode or self.w_quantizer is None or self.a_quantizer is None:
            return self.fp_module(x, *args, **kwargs)
        shape = x.shape
        x2 = x.reshape(-1, shape[-1])
        premul, rot = self._act_transform()
        q, scale, ssum = self.a_quantizer.quantize_int8(x2, premul, rot)
        wq = self.w_quantizer
        zp = None if wq.sym else wq.zero_point.reshape(-1).float().contiguous()
        out_dtype = x.dtype if x.dtype in (torch.float16, torch.bfloat16, torch.float32) else torch.float32
        w4 = self._codes.dtype == torch.uint8
        if w4:  # nibbles are stored as code + 8: fold the 8 into the zero point of the epilogue
            if self._zp_gemm is None:
                self._zp_gemm = ((zp if zp is not None else torch.zeros_like(wq.delta.reshape(-1).float())) - 8.0).contiguous()
            zp = self._zp_gemm
        y = qgemm.w8a8_linear(q, self._codes, scale, wq.delta.reshape(-1).float().contiguous(),
                              None if self.bias is None else self.bias.detach().float().contiguous(),
                              ssum if zp is not None else None, zp, out_dtype=out_dtype, w4=w4)
        if not self.a_quantizer.sym:
            # asymmetric activations x_dq = (q + zp_a) * s_a: the zero point's share of x_dq . w_dq^T is the rank-one term
            # (zp_a s_a)[token] x rowsum(w_dq)[channel]  (Q/base/base_quantizer.py:130-162; no Wan configuration uses this branch)
            t_a = (self.a_quantizer.zero_point.reshape(-1).float() * scale).to(y.dtype)
            y = torch.addcmul(y, t_a.unsqueeze(1), self.weight.data.float().sum(dim=1).to(y.dtype).unsqueeze(0))
        return y.view(*shape[:-1], self.out_features)

    # ---- PTQ hooks shared by the variants ------------------------------------------------------------
    def get_channel_mask(self, act_mask):
        """w_absmax_in**alpha / act_absmax**(1-alpha) (viditq_quant_layer.py:30-35, sq_quant_layer.py:27-32)."""
        wm = self.fp_module.weight.data.float().abs().amax(dim=0)
        self.channel_mask = (wm.abs() ** self.alpha) / (act_mask.to(wm.device).float().abs() ** (1 - self.alpha))
        assert not torch.isnan(self.channel_mask).any() and not torch.isinf(self.channel_mask).any(), "bad channel mask"
        self._premul = self._rot = None

    def get_rotation_matrix(self, generator=None):
        """Draw the random +-1 diagonal (quarot_utils.random_hadamard_matrix draws it from the global RNG)."""
        self.rotation_signs = quarot_utils.random_hadamard_signs(self.in_features, generator)
        self._premul = self._rot = None

    @property
    def rotation_matrix(self):
        if self.rotation_signs is None:
            return None
        return quarot_utils.random_hadamard_matrix(self.in_features, self.fp_module.weight.device, self.rotation_signs)

    def _rotate_weight(self, w):
        """(w.double() @ R).float() evaluated as hadU(w * signs) in float64 (row i of R = sign_i * hadU(e_i))."""
        s = self.rotation_signs.to(w.device)
        return quarot_utils.matmul_hadU(w.double() * s).float()
