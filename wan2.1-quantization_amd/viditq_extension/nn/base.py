"""QuantParams: the per-token scratch shared between an activation-quantising producer kernel and the
int8 GEMMs that consume it.  Mirrors ViDiT-Q/kernels/viditq_extension/nn/base.py:3-26 (same attribute
names: `scale_input`, `sum_input`, `has_sum_input`), with the storage dtype selectable: fp16 is what the
reference allocates, fp32 keeps the simulation path's scale precision (qdiff deltas are fp32)."""
import torch


class QuantParams:
    __slots__ = ("has_sum_input", "scale_input", "sum_input")

    def __init__(self, seq_len, has_sum_input=False, device="cuda", dtype=torch.float16):
        if dtype not in (torch.float16, torch.float32):
            raise ValueError("QuantParams dtype must be float16 or float32")
        self.has_sum_input = bool(has_sum_input)
        n = 2 if self.has_sum_input else 1
        # one allocation, two views: scale and sum of a token end up in the same pages
        buf = torch.empty(n, int(seq_len), dtype=dtype, device=device)
        self.scale_input = buf[0]
        self.sum_input = buf[1] if self.has_sum_input else None

    @property
    def seq_len(self):
        return self.scale_input.shape[0]

    def like(self, seq_len=None):
        """A fresh buffer set with the same options (used when several producers are in flight)."""
        return QuantParams(seq_len or self.seq_len, self.has_sum_input, self.scale_input.device, self.scale_input.dtype)
