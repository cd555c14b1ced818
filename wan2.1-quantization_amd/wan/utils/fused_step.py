"""One kernel per denoising step for everything that is not the DiT: classifier-free guidance + the scheduler update
(ViDiT-Q/examples/Wan2.1/wan/text2video.py:260-269).

Every operation of the UniPC / DPM++ / Euler update is `tensor +- tensor` or `scalar * tensor`, so each result of a step is a
linear combination of a handful of latent-sized tensors: the current sample, the two model outputs and the scheduler's own
history.  FusedStep runs the scheduler's UNCHANGED step() on symbolic linear forms (LinForm: a float64 coefficient vector over
those tensors) to get the coefficients, and evaluates all results of the step -- next sample, the new history entries -- with
ONE launch of wanq_lincomb.  The (at most 32) coefficients of a step travel BY VALUE in the kernel arguments of that launch:
a coefficient table in device memory refreshed by an async copy per step raced as soon as the CPU ran a step ahead of the GPU
(the next step's copy overwrote the table under the previous launch; csrc/stepops.hip header, regression test
test_fused_step_with_the_cpu_running_steps_ahead_of_the_gpu).  The launch therefore cannot be replayed from a captured HIP
graph with changing coefficients: the scheduler update stays OUTSIDE the captured DiT passes (wan/graph.py)."""
import numpy as np
import torch

from viditq_extension import _C


class LinForm:
    """sum_k c[k] * basis[k] with float64 coefficients."""
    __array_priority__ = 1000

    def __init__(self, c):
        self.c = np.asarray(c, dtype=np.float64)

    @classmethod
    def unit(cls, k, n):
        c = np.zeros(n)
        c[k] = 1.0
        return cls(c)

    def __add__(self, o):
        return LinForm(self.c + o.c)

    def __sub__(self, o):
        return LinForm(self.c - o.c)

    def __neg__(self):
        return LinForm(-self.c)

    def __mul__(self, s):
        return LinForm(self.c * float(s))

    __rmul__ = __mul__

    def __truediv__(self, s):
        return LinForm(self.c / float(s))

    def is_unit(self):
        nz = np.flatnonzero(self.c)
        return len(nz) == 1 and self.c[nz[0]] == 1.0, (int(nz[0]) if len(nz) == 1 else -1)


def lincomb(coef, ins, outs):
    """outs[o] = sum_i coef[o, i] * ins[i]; fp32 contiguous tensors of equal numel; coef: host fp32 [n_out, n_in] (numpy / torch
    CPU / nested lists) -- it is copied into the kernel arguments by the call."""
    import ctypes
    c = np.ascontiguousarray(np.asarray(coef, dtype=np.float32).reshape(len(outs), len(ins)))
    n = ins[0].numel()
    for t in list(ins) + list(outs):
        _C.check_gpu("tensor", t)
        _C.check_contig("tensor", t)
        _C.check_dtype("tensor", t, torch.float32)
        if t.numel() != n:
            raise RuntimeError("lincomb: tensors must have the same number of elements")
    with torch.cuda.device(ins[0].device):
        _C.call("wanq_lincomb", len(outs), len(ins), c.ctypes.data_as(ctypes.c_void_p), _C.ptr_array(ins), _C.ptr_array(outs), n,
                _C.stream())


class FusedStep:
    """Drop-in for `noise = uncond + g * (cond - uncond); latent = sched.step(noise, t, latent)`.

        fs = FusedStep(sched, guide_scale, like=latent)
        latent = fs.step(cond, uncond, latent)          # one kernel

    Works with any scheduler whose step() is linear in its tensor arguments and history and keeps that history in the
    attributes `_m` (list) and, optionally, `_last_sample` (FlowUniPCMultistepScheduler, FlowDPMSolverMultistepScheduler,
    FlowMatchScheduler).  Buffers are allocated once (`like`) and rotated, so the addresses a captured graph saw stay valid."""

    MAX_IN = 8

    def __init__(self, sched, guide_scale, like):
        self.sched, self.g = sched, float(guide_scale)
        # output pool: rotated so that an output never aliases a live history tensor
        self.pool = [torch.empty_like(like, dtype=torch.float32) for _ in range(8)]
        self.n_launch = 0

    def _free_buffers(self, live, k):
        ids = {t.data_ptr() for t in live}
        out = [b for b in self.pool if b.data_ptr() not in ids]
        assert len(out) >= k, "FusedStep: buffer pool exhausted"
        return out[:k]

    def step(self, cond, uncond, sample, timestep=None):
        s = self.sched
        hist = list(getattr(s, "_m", []))
        last = getattr(s, "_last_sample", None)
        basis = [sample, cond, uncond] + hist + ([last] if last is not None else [])
        n = len(basis)
        assert n <= self.MAX_IN
        unit = [LinForm.unit(k, n) for k in range(n)]
        # ---- the scheduler's own arithmetic on linear forms
        if hasattr(s, "_m"):
            s._m = unit[3:3 + len(hist)]
        if last is not None:
            s._last_sample = unit[3 + len(hist)]
        noise = unit[2] + self.g * (unit[1] - unit[2])
        try:
            prev = s.step(noise, timestep, unit[0]) if _takes_timestep(s) else s.step(noise, unit[0])
        except Exception:
            if hasattr(s, "_m"):
                s._m = hist
            if last is not None:
                s._last_sample = last
            raise
        # ---- which results are new tensors, which are old ones carried over
        results = [("prev", prev)]
        new_hist = list(getattr(s, "_m", []))
        for idx, f in enumerate(new_hist):
            results.append((("m", idx), f))
        if getattr(s, "_last_sample", None) is not None:
            results.append(("last", s._last_sample))
        todo, resolved = [], {}
        for key, f in results:
            isu, k = f.is_unit()
            if isu:
                resolved[key] = basis[k]
            else:
                todo.append((key, f))
        assert len(todo) <= 4
        outs = self._free_buffers(basis, len(todo))
        coef = np.zeros((4, self.MAX_IN), dtype=np.float32)
        for o, (_, f) in enumerate(todo):
            coef[o, :n] = f.c.astype(np.float32)
        # the coefficients go by value with the launch: the CPU may be steps ahead of the GPU
        lincomb(coef[:len(todo), :n], [b.contiguous() for b in basis], outs)
        self.n_launch += 1
        for (key, _), t in zip(todo, outs):
            resolved[key] = t
        if hasattr(s, "_m"):
            s._m = [resolved[("m", idx)] for idx in range(len(new_hist))]
        if getattr(s, "_last_sample", None) is not None:
            s._last_sample = resolved["last"]
        return resolved["prev"]


def _takes_timestep(s):
    import inspect
    return "timestep" in inspect.signature(s.step).parameters
