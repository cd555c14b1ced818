#!/usr/bin/env python3
"""ISA lint of the compiled gfx950 kernels (no GPU needed): the co-run hazard of DESIGN.md 3.5 as a build-time check.

What it looks for.  On gfx950 a packed fp32 operation (`v_pk_mul_f32`, `v_pk_fma_f32`, `v_pk_add_f32`; `v_pk_mov_b32` is held to
the same rule) whose `op_sel` feeds the HIGH register of a 64-bit source pair to the LOW lane returned 0 for lanes 48-63 when that
pair had been written by a vector-memory load and ANOTHER register-returning load of the wave was still in flight
(`profiles/r04_z_corun_corruption.txt`: `s_waitcnt vmcnt(1)` had retired the pair's own load, the wave's second
`global_load_dwordx4` was still landing).  The fix in `csrc/attn_prep.hip` is an operand-tied `s_waitcnt vmcnt(0)`; nothing stops
a compiler bump or a packed-op edit from re-creating the pattern elsewhere, so every code object is checked:

  ERROR   a packed op reads `v[n:n+1]` through `op_sel` high -> low, some register of the pair may hold the result of a
          vector-memory load (not overwritten since by a vector / LDS instruction), and a register-returning vector-memory load may
          still be outstanding at that point (forward may-analysis over the kernel's control-flow graph: the vmcnt queue holds
          loads, stores, atomics and LDS-DMA in issue order, `s_waitcnt vmcnt(N)` keeps the N youngest).

Second check: the ping-pong GEMM's LDS ring rests on hand-counted `vmcnt` immediates that assume exact numbers of
store / LDS-DMA instructions per tile.  The lint holds every `gemm_w8a8_pp_kernel` instantiation to (i) no scratch instruction
inside a K loop (innermost loop with MFMAs), (ii) only the `vmcnt` immediates the schedule uses, (iii) a recorded per-instantiation count of global stores and LDS-DMA
instructions (`PP_SIGNATURE`): a change there is not an error of the code but a demand to re-audit `csrc/gemm_w8a8_pp.hip`'s counted
waits (header comment, `PP_WAITVM`, the first burst behind an epilogue) and then to update the table.

Inputs: object files with a gfx950 offload bundle (default: every `wan2.1-quantization_amd/build/*.o`), or `--asm FILE` with
ISA text (`llvm-objdump -d` or `hipcc -S` syntax).  Exit status 1 when an ERROR is found.
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("WANQ_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(HERE, "..", "wan2.1-quantization_amd")

PACKED = ("v_pk_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mov_b32")
VMEM_PREFIX = ("global_", "buffer_", "scratch_", "flat_", "tbuffer_")
QUEUE_CAP = 64  # vmcnt is a 6-bit counter
STATE_CAP = 24  # distinct vmcnt queues kept per basic block before they are merged conservatively


class Inst:
    __slots__ = ("addr", "mn", "ops", "mods", "text", "label", "target")

    def __init__(self, addr, mn, ops, mods, text, label=None, target=None):
        self.addr, self.mn, self.ops, self.mods, self.text, self.label, self.target = addr, mn, ops, mods, text, label, target


def disassemble_object(path):
    """gfx950 ISA text of the offload bundle inside a host object file (hipcc -c output)."""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return ""  # a host-only translation unit
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                        f"--input={fat}", f"--output={co}"], check=True, capture_output=True)
        return subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout


_RE_FUNC_OBJDUMP = re.compile(r"^([0-9a-f]+) <([^>]+)>:\s*$")
_RE_LABEL = re.compile(r"^([.\w$]+):")
_RE_ADDR = re.compile(r"//\s*([0-9A-Fa-f]+):")
_RE_TARGET = re.compile(r"<([^>+]+)(?:\+0x([0-9a-fA-F]+))?>\s*$")


def parse(text):
    """-> {function name: [Inst]} from objdump or -S text."""
    funcs, cur, name, pending_label, base = {}, None, None, None, {}
    for raw in text.splitlines():
        line = raw.rstrip()
        m = _RE_FUNC_OBJDUMP.match(line)
        if m:
            name = m.group(2)
            base[name] = int(m.group(1), 16)
            cur = funcs.setdefault(name, [])
            continue
        if not line.startswith(("\t", " ")):  # -S syntax: labels and directives start in column 0
            m = _RE_LABEL.match(line)
            if m and not line.startswith(";"):
                lab = m.group(1)
                if not lab.startswith(".L"):  # a function symbol
                    name = lab
                    cur = funcs.setdefault(name, [])
                    pending_label = None
                else:
                    pending_label = lab
            continue
        body = line.strip()
        if not body or body.startswith((";", ".", "//")):
            continue
        if cur is None:
            name = "<snippet>"
            cur = funcs.setdefault(name, [])
        addr = None
        ma = _RE_ADDR.search(body)
        if ma:
            addr = int(ma.group(1), 16)
        target = None
        mt = _RE_TARGET.search(body)
        if mt and mt.group(1) in base:
            target = base[mt.group(1)] + int(mt.group(2) or "0", 16)
        code = re.split(r"\s*//|\s+;", body, maxsplit=1)[0].strip()
        parts = code.split(None, 1)
        mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        # operands are comma separated up to the first modifier token (a word without a leading register / literal shape after a space)
        ops, mods = [], ""
        depth, tok, i = 0, "", 0
        while i < len(rest):
            ch = rest[i]
            if ch == "[":
                depth += 1
            elif ch == "]":
                depth -= 1
            if ch == "," and depth == 0:
                ops.append(tok.strip())
                tok = ""
            else:
                tok += ch
            i += 1
        if tok.strip():
            ops.append(tok.strip())
        if ops:  # the last operand may carry trailing modifiers: "v[12:13] op_sel:[0,1] op_sel_hi:[0,0]", "off offset:16", "s[2:3] sc0"
            last = ops[-1].split(None, 1)
            ops[-1] = last[0]
            mods = last[1] if len(last) > 1 else ""
        if mn.startswith(("s_branch", "s_cbranch")) and target is None and ops and not ma:
            target = ops[0]  # -S syntax: a label
        cur.append(Inst(addr, mn, ops, mods, code, pending_label, target))
        pending_label = None
    return {k: v for k, v in funcs.items() if v}


_RE_VREG = re.compile(r"^(v|a)(?:(\d+)|\[(\d+):(\d+)\])$")


def regs(op):
    """set of ('v'|'a', n) named by a vector-register operand; empty for scalars / literals / `off`."""
    m = _RE_VREG.match(op.strip().lstrip("-|").rstrip("|"))
    if not m:
        return frozenset()
    if m.group(2) is not None:
        return frozenset({(m.group(1), int(m.group(2)))})
    return frozenset((m.group(1), n) for n in range(int(m.group(3)), int(m.group(4)) + 1))


def is_vmem(i):
    return i.mn.startswith(VMEM_PREFIX) and ("_load" in i.mn or "_store" in i.mn or "_atomic" in i.mn)


def is_lds_dma(i):
    return "_load_lds_" in i.mn or re.search(r"(^|\s)lds(\s|$)", i.mods) is not None


def vmem_dest(i):
    """registers a vector-memory instruction returns data into."""
    if not is_vmem(i) or is_lds_dma(i) or "_store" in i.mn:
        return frozenset()
    if "_atomic" in i.mn and "sc0" not in i.mods.split() and "glc" not in i.mods.split():
        return frozenset()
    return regs(i.ops[0]) if i.ops else frozenset()


_NO_VDEST = ("s_", "ds_write", "ds_store", "v_cmp", "v_cmpx", "v_readfirstlane", "v_readlane", "v_nop", "ds_nop", "ds_gws", "ds_append", "ds_consume")


def valu_dests(i):
    """vector registers a non-VMEM instruction overwrites."""
    if i.mn.startswith(_NO_VDEST) or not i.ops:
        return frozenset()
    d = regs(i.ops[0])
    if i.mn.startswith(("v_swap_b32", "v_permlane16_swap", "v_permlane32_swap")) and len(i.ops) > 1:
        d = d | regs(i.ops[1])
    return d


def op_sel(i):
    m = re.search(r"op_sel:\[([0-9,]+)\]", i.mods)
    return [int(x) for x in m.group(1).split(",")] if m else []


def hi_to_lo_pairs(i):
    """source pairs whose HIGH register feeds the LOW lane of a packed op."""
    if not i.mn.startswith(PACKED):
        return []
    sel, out = op_sel(i), []
    for k, s in enumerate(sel):
        if s == 1 and 1 + k < len(i.ops):
            r = regs(i.ops[1 + k])
            if len(r) == 2:
                out.append((i.ops[1 + k], r))
    return out


def waitcnt_vm(i):
    if not i.mn.startswith("s_waitcnt"):
        return None
    m = re.search(r"vmcnt\((\d+)\)", i.text)
    return int(m.group(1)) if m else None


def blocks_of(insts):
    """basic blocks as (start, end) index ranges + successor lists."""
    index_of = {}
    for n, i in enumerate(insts):
        if i.addr is not None:
            index_of[i.addr] = n
        if i.label:
            index_of[i.label] = n
    leaders = {0}
    for n, i in enumerate(insts):
        if i.mn.startswith(("s_branch", "s_cbranch")):
            if i.target in index_of:
                leaders.add(index_of[i.target])
            if n + 1 < len(insts):
                leaders.add(n + 1)
        elif i.mn.startswith(("s_endpgm", "s_setpc", "s_swappc")) and n + 1 < len(insts):
            leaders.add(n + 1)
        if i.label and not i.mn.startswith("s_"):
            leaders.add(n)
        elif i.label:
            leaders.add(n)
    order = sorted(leaders)
    blocks, succ = [], {}
    for k, s in enumerate(order):
        e = order[k + 1] if k + 1 < len(order) else len(insts)
        blocks.append((s, e))
    start_to_block = {s: k for k, (s, _) in enumerate(blocks)}
    for k, (s, e) in enumerate(blocks):
        last = insts[e - 1]
        out = []
        if last.mn.startswith("s_branch"):
            if last.target in index_of:
                out.append(start_to_block[index_of[last.target]])
        elif last.mn.startswith("s_cbranch"):
            if last.target in index_of:
                out.append(start_to_block[index_of[last.target]])
            if e < len(insts):
                out.append(start_to_block[e])
        elif last.mn.startswith(("s_endpgm", "s_setpc", "s_swappc")):
            pass
        elif e < len(insts):
            out.append(start_to_block[e])
        succ[k] = out
    return blocks, succ


def lint_function(name, insts):
    """-> (errors, stats).  State per program point: {vmcnt queue (tuple of bool: register-returning load?) : tainted registers}."""
    blocks, succ = blocks_of(insts)
    in_states = [dict() for _ in blocks]
    in_states[0] = {(): frozenset()}
    work, errors, seen_err, collapsed = [0], [], set(), set()
    stats = {"packed_hi_to_lo": 0, "on_load_destinations": 0}
    counted = set()
    while work:
        b = work.pop()
        s, e = blocks[b]
        states = dict(in_states[b])
        for n in range(s, e):
            i = insts[n]
            pairs = hi_to_lo_pairs(i)
            if pairs and n not in counted:
                counted.add(n)
                stats["packed_hi_to_lo"] += 1
            new_states = {}
            for q, taint in states.items():
                if pairs:
                    for op, r in pairs:
                        if r & taint:
                            if ("t", n) not in counted:
                                counted.add(("t", n))
                                stats["on_load_destinations"] += 1
                            if any(q) and n not in seen_err:
                                seen_err.add(n)
                                errors.append((name, i, op, sum(q)))
                vm = waitcnt_vm(i)
                if vm is not None:
                    q = q[len(q) - vm:] if vm < len(q) else q
                    if vm == 0:
                        q = ()
                elif is_vmem(i):
                    d = vmem_dest(i)
                    q = (q + (bool(d),))[-QUEUE_CAP:]
                    taint = taint | d
                else:
                    d = valu_dests(i)
                    if d:
                        taint = taint - d
                new_states[q] = new_states.get(q, frozenset()) | taint
            states = new_states
        for t in succ[b]:
            changed = False
            tgt = in_states[t]
            for q, taint in states.items():
                if t in collapsed:  # everything folds into the block's single conservative entry state
                    (top, cur_t), = tgt.items()
                    if len(q) > len(top) or not taint <= cur_t:
                        tgt.clear()
                        tgt[(True,) * min(QUEUE_CAP, max(len(q), len(top)))] = cur_t | taint
                        changed = True
                    continue
                if q in tgt:
                    if not taint <= tgt[q]:
                        tgt[q] = tgt[q] | taint
                        changed = True
                elif len(tgt) < STATE_CAP:
                    tgt[q] = taint
                    changed = True
                else:  # too many distinct queues: collapse the block's entry to ONE conservative state (monotone: terminates)
                    n_top = min(QUEUE_CAP, max(len(q), max(len(k) for k in tgt)))
                    top = (True,) * n_top
                    merged_t = taint
                    for v in tgt.values():
                        merged_t = merged_t | v
                    tgt.clear()
                    tgt[top] = merged_t
                    collapsed.add(t)
                    changed = True
            if changed and t not in work:
                work.append(t)
    return errors, stats


# ---- the ping-pong GEMM: (global stores, LDS-DMA instructions, register-returning loads) per instantiation of gemm_w8a8_pp_kernel.
# A mismatch means the epilogue / prefetch code changed shape: re-audit the counted waits of csrc/gemm_w8a8_pp.hip (the 16 / 32 stores
# of a full tile behind vmcnt(20) / vmcnt(36), the eight scale pieces behind vmcnt(12), four pieces per PP_ISSUE_*), then update.
PP_VMCNT_ALLOWED = {0, 4, 8, 10, 12, 20, 36}
PP_SIGNATURE = None  # filled in from the shipped build (see _pp_signature_table below)


def pp_signature(insts):
    st = sum(1 for i in insts if is_vmem(i) and "_store" in i.mn and not i.mn.startswith("scratch_"))
    dma = sum(1 for i in insts if is_vmem(i) and is_lds_dma(i))
    ld = sum(1 for i in insts if vmem_dest(i))
    return (st, dma, ld)


def mfma_loops(insts):
    """innermost natural loops that contain MFMA instructions (the K loops), each as a list of instruction index ranges.  A loop is
    taken from a branch to a lower address: its body is the header plus every block that reaches the latch without passing the
    header (hipcc may place a loop's latch behind unrelated epilogue blocks, so address ranges will not do)."""
    blocks, succ = blocks_of(insts)
    pred = {k: [] for k in range(len(blocks))}
    for k, out in succ.items():
        for t in out:
            pred[t].append(k)
    # dominators (bit sets), entry = block 0; blocks never reached from the entry keep the full set and are ignored
    nb = len(blocks)
    full = (1 << nb) - 1
    dom = [full] * nb
    dom[0] = 1
    changed = True
    while changed:
        changed = False
        for k in range(1, nb):
            d = full
            for q in pred[k]:
                d &= dom[q]
            d |= 1 << k
            if d != dom[k]:
                dom[k] = d
                changed = True
    loops = []
    for k, (s, e) in enumerate(blocks):
        for t in succ[k]:
            if dom[k] != full and (dom[k] >> t) & 1:  # back edge k -> t: the target dominates the source
                body, stack = {t, k}, [k]
                while stack:
                    x = stack.pop()
                    if x == t:
                        continue
                    for q in pred[x]:
                        if q not in body:
                            body.add(q)
                            stack.append(q)
                if any(i.mn.startswith("v_mfma") for b_ in body for i in insts[blocks[b_][0]:blocks[b_][1]]):
                    loops.append(frozenset(body))
    loops = list(set(loops))
    inner = [l for l in loops if not any(o < l for o in loops)]
    return [[blocks[b_] for b_ in sorted(l)] for l in inner]


def lint_pp(name, insts, table):
    errs = []
    for ranges in mfma_loops(insts):
        scratch = [i for (m, n) in ranges for i in insts[m:n] if i.mn.startswith("scratch_")]
        if scratch:
            errs.append(f"{name}: {len(scratch)} scratch instruction(s) inside the K loop (a spill reload waits for the whole vmcnt queue "
                        f"and drains the LDS-DMA ring every K-tile): {scratch[0].text}")
    imms = {waitcnt_vm(i) for i in insts} - {None}
    if not imms <= PP_VMCNT_ALLOWED:
        errs.append(f"{name}: vmcnt immediates {sorted(imms - PP_VMCNT_ALLOWED)} outside the schedule's set {sorted(PP_VMCNT_ALLOWED)}")
    sig = pp_signature(insts)
    key = re.sub(r"^_ZN4wanq\d+_GLOBAL__N_1", "", name)
    if table is not None:
        want = table.get(key)
        if want is None:
            errs.append(f"{name}: no recorded (stores, LDS-DMA, loads) signature; measured {sig}: audit the counted waits, then add it to tools/isa_lint_pp_signature.json")
        elif tuple(want) != sig:
            errs.append(f"{name}: (stores, LDS-DMA, loads) = {sig}, recorded {tuple(want)}: the epilogue / prefetch changed shape -- re-audit the "
                        "counted vmcnt waits of csrc/gemm_w8a8_pp.hip, then update tools/isa_lint_pp_signature.json")
    return errs, (key, sig)


def lint_text(text, pp_table=None, what=""):
    funcs = parse(text)
    all_err, report, pp_sigs = [], [], {}
    for name, insts in funcs.items():
        errs, stats = lint_function(name, insts)
        for (fn, i, op, nload) in errs:
            all_err.append(f"{what}{fn}: `{i.text}` reads the high register of load destination {op} into the low lane with up to {nload} "
                           f"register-returning load(s) still outstanding" + (f" (0x{i.addr:x})" if i.addr is not None else ""))
        if stats["packed_hi_to_lo"]:
            report.append((name, stats))
        if "gemm_w8a8_pp_kernel" in name:
            e2, (key, sig) = lint_pp(name, insts, pp_table)
            all_err += [what + x for x in e2]
            pp_sigs[key] = sig
    return all_err, report, pp_sigs


def default_objects():
    return sorted(glob.glob(os.path.join(PKG, "build", "*.o")))


def load_pp_table():
    import json
    p = os.path.join(HERE, "isa_lint_pp_signature.json")
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def lint_objects(paths, verbose=False, pp_table="default"):
    table = load_pp_table() if pp_table == "default" else pp_table
    errors, lines, sigs = [], [], {}
    for p in paths:
        text = disassemble_object(p)
        errs, report, pp_sigs = lint_text(text, table if "gemm_w8a8_pp" in os.path.basename(p) else None, what=os.path.basename(p) + ": ")
        errors += errs
        sigs.update(pp_sigs)
        n_hi = sum(s["packed_hi_to_lo"] for _, s in report)
        n_ld = sum(s["on_load_destinations"] for _, s in report)
        lines.append(f"{os.path.basename(p):22s} packed ops with op_sel high->low: {n_hi:5d}   of them on load destinations: {n_ld:4d}   "
                     f"errors: {len(errs)}")
        if verbose:
            for name, s in report:
                lines.append(f"    {name}: {s}")
    return errors, lines, sigs


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("objects", nargs="*", help="host objects with a gfx950 bundle (default: wan2.1-quantization_amd/build/*.o)")
    ap.add_argument("--asm", help="lint an ISA text file instead (objdump or -S syntax)")
    ap.add_argument("--write-pp-signature", action="store_true", help="record the ping-pong GEMM's store / LDS-DMA counts after an audit")
    ap.add_argument("-v", "--verbose", action="store_true")
    a = ap.parse_args()
    if a.asm:
        with open(a.asm) as f:
            errs, report, _ = lint_text(f.read())
        for name, s in report:
            print(f"{name}: {s}")
    else:
        objs = a.objects or default_objects()
        if not objs:
            print("isa_lint: no objects (run wan2.1-quantization_amd/build.py first)", file=sys.stderr)
            return 2
        errs, lines, sigs = lint_objects(objs, a.verbose, pp_table=None if a.write_pp_signature else "default")
        print("\n".join(lines))
        if a.write_pp_signature:
            import json
            with open(os.path.join(HERE, "isa_lint_pp_signature.json"), "w") as f:
                json.dump({k: list(v) for k, v in sorted(sigs.items())}, f, indent=1)
            print(f"recorded {len(sigs)} ping-pong GEMM signatures")
    for e in errs:
        print("ERROR " + e)
    print(f"isa_lint: {len(errs)} error(s)")
    return 1 if errs else 0


if __name__ == "__main__":
    sys.exit(main())
