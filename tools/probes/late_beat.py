#!/usr/bin/env python3
"""Runner of tools/probes/late_beat/victim.hip (build: hipcc -O3 --offload-arch=gfx950 -fPIC -shared victim.hip -o
wan2.1-quantization_amd/lib/variants/liblate_beat.so).  The micro-victim copies a load's four destination registers right behind
the s_waitcnt that declares the load complete; this script runs it on the main stream with and without the attention-map sequence
on a side stream and counts copies that are not the table's values, by register (dword of the load) and 16-lane group."""
import collections
import ctypes
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "wan2.1-quantization_amd"))
from wan import ops  # noqa: E402

lib = ctypes.CDLL(os.path.join(ROOT, "wan2.1-quantization_amd", "lib", "variants", "liblate_beat.so"))
lib.late_beat_victim.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]


def attacker():
    H, L, d = 4, 270, 128
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(L, H * d, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    return lambda: ops.attention_map_quant(q, k, v, H, 8, False, L, q_len=L)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 300
    rows, per = 270, 40
    table = torch.randn(rows, 64, 8, device="cuda") + 3.0  # never 0, never the -7 sentinel
    want = table[:, :, :4].cpu()
    att, side = attacker(), torch.cuda.Stream()
    names = {0: "A, B in flight, vmcnt(1) (hipcc's form)", 1: "A alone, vmcnt(0)", 2: "A, B in flight, vmcnt(0)", 3: "as 0, second copy 8+ cycles later",
             4: "the RoPE stage's own sequence: loads, address register overwritten, vmcnt(1), two v_pk_mul_f32 with op_sel",
             5: "as 4 + s_nop 3 behind the wait", 6: "as 4, multiplier NOT in the loads' address register (no overwrite of v8)",
             7: "as 4, the multiply that reads A's last register FIRST behind the wait",
             8: "as 4, A's registers pre-filled with 5.0 (a stale read shows as 5.0, not 0)",
             12: "as 4, the second packed multiply WITHOUT op_sel (lo = v8 * v12, hi = v9 * v13)",
             13: "as 12 with v_pk_fma_f32 (+ 0) in place of v_pk_mul_f32",
             10: "as 4, the RESULT registers pre-filled with 9.0 (a lost write shows as 9.0)",
             11: "as 10, two plain v_mul_f32 in place of the second packed multiply",
             9: "as 4 + a plain v_mov of A's last register right behind the multiply, returned as 'A.x' (dword 0)"}
    for beside in (False, True):
        for mode in ((4, 12, 13) if "--pk4" in sys.argv else (4,) if "--pk0" in sys.argv else (10, 11) if "--pk3" in sys.argv else (8, 9) if "--pk2" in sys.argv else (4, 5, 6, 7) if "--pk" in sys.argv else (0, 1, 2, 3, 4)):
            early = torch.empty(per, rows, 64, 8, device="cuda")
            second = torch.empty(per, rows, 64, 4, device="cuda")
            where, vals, launches, bad_launches, bad_second = collections.Counter(), collections.Counter(), 0, 0, 0
            for _ in range(iters):
                if beside:
                    with torch.cuda.stream(side):
                        junk = [att() for _ in range(6)]
                for i in range(per):
                    rc = lib.late_beat_victim(mode, table.data_ptr(), early[i].data_ptr(), second[i].data_ptr(), rows, torch.cuda.current_stream().cuda_stream)
                    assert rc == 0
                torch.cuda.synchronize()
                launches += per
                e = early[:, :, :, :4].cpu()
                if mode == 9:  # dword 0 carries the plain v_mov copy of A's last register: compare it with A.w, then put A.x back
                    mov_bad = (e[..., 0] != want[..., 3])
                    prod_bad = (e[..., 3] != want[..., 3])
                    vals["product wrong, v_mov right"] += int((prod_bad & ~mov_bad).sum())
                    vals["product wrong, v_mov wrong"] += int((prod_bad & mov_bad).sum())
                    vals["product right, v_mov wrong"] += int((~prod_bad & mov_bad).sum())
                    e[..., 0] = want[..., 0]
                d = (e != want).nonzero()
                if len(d):
                    bad_launches += len(d[:, 0].unique())
                    for _, r, lane, dw in d.tolist():
                        where[(dw, lane // 16)] += 1
                    for x in e[e != want].tolist()[:50]:
                        vals["sentinel -7" if x == -7.0 else ("0" if x == 0.0 else "5.0" if x == 5.0 else "9.0" if x == 9.0 else "other")] += 1
                if mode == 3:
                    bad_second += int((second.cpu() != want).any(dim=(1, 2, 3)).sum())
            print(f"attention-map beside: {beside!s:5}  mode {mode} [{names[mode]}]: {launches} launches, {bad_launches} with a wrong copy; "
                  f"(dword, 16-lane group) -> count: {dict(sorted(where.items()))}; wrong values: {dict(vals)}"
                  + (f"; launches whose SECOND copy is wrong: {bad_second}" if mode == 3 else ""), flush=True)


if __name__ == "__main__":
    main()
