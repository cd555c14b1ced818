"""The two DiT passes of a denoising step (conditional, unconditional: W/wan/text2video.py:255-258) on ONE or on TWO HIP streams.

The reference runs them back to back on one stream.  They are independent until the guidance combine, and every kernel of a pass is
sized to fill the GPU, so issuing them on two streams does not make the GPU run two passes "in parallel": what it does is let one
pass's next kernel start on the CUs the other's kernel has already left -- the ramp-up of a launch, the drain of its last workgroups
and half-empty last rounds (ffn.0's persistent GEMM: 17.5 tiles per workgroup) are where a one-stream step idles; between kernels it
hardly does (1.4 ms of 425: profiles/r05_s_idle_between_kernels.txt).  What it costs: workgroups of two kernels interleave on the
XCDs, so two heads' K / V (and two GEMMs' panels) share each L2 and the MALL.  Which side wins depends on the box and on the size:
cfg-B (1.3B, L = 32760) 1.033x / 1.015x / 1.012x / 1.000x on four boxes, alternating runs; the 14B shapes on one GPU (L = 75600:
twice the K / V per head) 0.96x (profiles/r05_q_*, r05_r_*, r05_u_*, r05_v_*).  So the order is CHOSEN BY MEASUREMENT at the start of
a sampling loop (`mode="auto"`, the default): call 1 runs on one stream and fills what both passes later only read (per-context
cross-attention k / v, rotary table, modulation table); calls 2 - 5 are timed alternately on one stream and on two (HIP events on
the caller's stream; the only host synchronisation this helper ever makes, once, before call 6); two streams are kept when the
faster of their two samples was at least 1 % below the faster of the one-stream samples.  Results do not depend on the choice:
latents are bit-equal either way (tests/test_gpu_step.py) -- every kernel of the path is deterministic and none shares mutable state
with a kernel of the other pass.

Rules the two-stream order keeps: each pass runs wholly inside its stream's context, so its temporaries come from that stream's
allocator pool; the latent is recorded on both side streams and the outputs on the caller's.  One rank without CFG parallelism only
(under sequence parallelism the passes' collectives would interleave on one communicator).
WANQ_PASS_STREAMS = auto (default) | 1 (always one stream: the reference's order) | 2 (always two)."""
import os

import torch


class TwoPassStreams:
    def __init__(self, device, enabled=None, mode=None):
        """enabled=False: one stream, whatever the environment says (callers that cannot use two: FP / simulation mode, N > 1)."""
        mode = (mode or os.environ.get("WANQ_PASS_STREAMS", "auto")).lower()
        if mode not in ("auto", "1", "2"):
            raise ValueError(f"WANQ_PASS_STREAMS / mode must be auto, 1 or 2 (got {mode!r})")
        usable = torch.device(device).type == "cuda" and enabled is not False and mode != "1"
        self.mode = mode if usable else "1"
        self.enabled = usable            # two streams may be used
        self.decided = self.mode != "auto"
        self.streams = [torch.cuda.Stream(device), torch.cuda.Stream(device)] if usable else None
        self.calls = 0
        self.tuned = None                # (ms on one stream, ms on two) once measured
        self._ev = []

    def _one(self, run_pass, contexts):
        return [run_pass(c) for c in contexts]

    def _two(self, run_pass, latent, contexts):
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

    def _timed(self, fn):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        out = fn()
        b.record()
        self._ev.append((a, b))
        return out

    def __call__(self, run_pass, latent, contexts):
        """run_pass(context) -> output tensor, called once per context; returns the outputs in order."""
        self.calls += 1
        if not self.enabled or len(contexts) != 2 or self.calls == 1:
            return self._one(run_pass, contexts)          # (call 1: fills the caches both passes read)
        if self.decided:
            return self._two(run_pass, latent, contexts)
        k = self.calls - 2                                # 0 .. 3: one, two, one, two
        if k < 4:
            return self._timed((lambda: self._one(run_pass, contexts)) if k % 2 == 0 else (lambda: self._two(run_pass, latent, contexts)))
        self._decide()
        return self._two(run_pass, latent, contexts) if self.enabled else self._one(run_pass, contexts)

    def _decide(self):
        self._ev[-1][1].synchronize()                     # the one host wait: the last timed call has finished
        ms = [a.elapsed_time(b) for a, b in self._ev]
        t1, t2 = min(ms[0::2]), min(ms[1::2])             # the faster of two samples each (a sample can only be slowed, not sped up)
        self.tuned, self._ev = (t1, t2), []
        self.decided = True
        self.enabled = t2 < 0.99 * t1

    def tune(self, run_pass, latent, contexts):
        """Make the choice now, on five untimed evaluations of the two passes (bench.py: before its warm-up steps, so that the timed
        region runs one schedule).  Returns (ms one stream, ms two streams) or None when there was nothing to choose."""
        if not self.enabled or self.decided:
            if self.calls == 0 and self.enabled:
                self(run_pass, latent, contexts)          # still fill the caches on one stream
            return self.tuned
        while len(self._ev) < 4 and not self.decided:
            self(run_pass, latent, contexts)
        if not self.decided:
            self._decide()
        return self.tuned

    def describe(self):
        if self.streams is None:
            return "one stream"
        how = f"measured {self.tuned[0]:.1f} ms on one stream, {self.tuned[1]:.1f} ms on two" if self.tuned else ("forced" if self.mode != "auto" else "not measured yet")
        return ("two HIP streams" if self.enabled else "one stream") + f" ({how}; wan/utils/two_pass.py)"
