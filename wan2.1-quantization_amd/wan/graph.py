"""HIP-graph capture of the DiT passes of a denoising step.

The kernel-mode forward of a fixed (latent shape, context length, seq_len) launches the same ~700 kernels with the same
addresses every time -- shapes are static and every temporary comes from torch's allocator -- so the two passes of a step
(conditional + unconditional) are captured once into a hipGraph and replayed: the host then spends one graph launch per step
instead of ~1400 kernel launches (Python + ctypes + torch allocator calls), and the inter-kernel gaps shrink to the device's
own dependent-launch latency.  The scheduler update stays outside (its coefficients and buffers change per step; it is one
wanq_lincomb launch, wan/utils/fused_step.py).  Single-rank only: the sequence-parallel path issues collectives.

The model's per-context cache (QuantWanModel._context_source: cross_attn.k / .v of a text context kept across steps) is OFF while
the graph warms up and captures: the graph recomputes k / v from its own static context buffers on every replay, so it depends on
no tensor the cache owns and `graph.ctx[i].copy_(new_context)` takes effect at the next replay."""
import torch


class GraphedPasses:
    def __init__(self, model, latent, contexts, seq_len, warmup=2):
        """model: QuantWanModel in kernel mode.  latent: [C,F,H,W] fp32; contexts: list of [L_txt, D] tensors, one pass each."""
        assert latent.is_cuda and model.hip_blocks is not None, "graph capture is for the kernel-mode model on the GPU"
        self.model, self.seq_len = model, seq_len
        self.latent = latent.clone()
        self.t = torch.zeros(1, dtype=torch.int64, device=latent.device)
        self.ctx = [c.clone() for c in contexts]
        cache_was = getattr(model, "context_cache", True)
        model.context_cache = False
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):  # lazy initialisation (rope tables, function attributes, split-KV workspaces) happens here
                    for c in self.ctx:
                        model([self.latent], self.t, [c], seq_len)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph), torch.no_grad():
                self.outs = [model([self.latent], self.t, [c], seq_len)[0] for c in self.ctx]
        finally:
            model.context_cache = cache_was

    def __call__(self, latent, t):
        """-> list of model outputs (static buffers: consume them before the next call)."""
        self.latent.copy_(latent)
        self.t.copy_(t.reshape(1))
        self.graph.replay()
        return self.outs
