"""The two DiT passes of a denoising step (conditional, unconditional: W/wan/text2video.py:255-258) on TWO HIP streams.

The reference runs them back to back on one stream.  They are independent until the guidance combine, and every kernel of a pass is
sized to fill the GPU, so issuing them on two streams does not make the GPU run two passes "in parallel": what it does is let one
pass's next kernel start on the CUs the other's kernel has already left -- the ramp-up of a launch, the drain of its last workgroups
and half-empty last rounds (ffn.0's persistent GEMM: 17.5 tiles per workgroup) are where a one-stream step idles; between kernels it
hardly does (1.4 ms of 425: profiles/r05_s_idle_between_kernels.txt).  Measured: 427.5 -> 413.7 ms per step (1.033x), 422.0 -> 415.6,
438.7 -> 433.6 on three boxes, alternating runs (profiles/r05_q_*, r05_r_*, r05_zz_bench_line*.txt); results bit-equal to the one-stream order (tests/test_gpu_step.py): every kernel
of the path is deterministic and none shares mutable state with a kernel of the other pass.

Rules this helper keeps: the first step it sees runs on ONE stream (it fills what both passes later only read: the per-context
cross-attention k / v, the rotary table, the blocks' modulation table); each pass runs wholly inside its stream's context, so its
temporaries come from that stream's allocator pool; the latent is recorded on both side streams and the outputs on the caller's.
One rank without CFG parallelism only (under sequence parallelism the passes' collectives would interleave on one communicator)."""
import os

import torch


class TwoPassStreams:
    def __init__(self, device, enabled=None):
        if enabled is None:
            enabled = os.environ.get("WANQ_PASS_STREAMS", "2") != "1"
        self.enabled = bool(enabled) and torch.device(device).type == "cuda"
        self.streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)] if self.enabled else None
        self.warm = False

    def __call__(self, run_pass, latent, contexts):
        """run_pass(context) -> output tensor, called once per context; returns the outputs in order."""
        if not self.enabled or not self.warm or len(contexts) != 2:
            self.warm = True
            return [run_pass(c) for c in contexts]
        cur = torch.cuda.current_stream(latent.device)
        outs = []
        for st, c in zip(self.streams, contexts):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(run_pass(c))
            latent.record_stream(st)
        for st, o in zip(self.streams, outs):
            cur.wait_stream(st)
            o.record_stream(cur)
        return outs
