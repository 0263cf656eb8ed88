#!/usr/bin/env python3
"""Lint for kernels whose global loads are inline asm with explicit vmcnt waits (conv_thin4_mfma_kernel, conv_bf3*_kernel,
wgrad_bf3_kernel, tools/ubench/wgrad1x1_direct.hip): hipcc does not know that the destination registers of an asm load are in
flight, so nothing stops it from reading (copying, spilling) or overwriting them before the matching asm wait -- the
register-allocation accidents DESIGN.md 3.6 / 3.11 describe, each of which gave wrong results on the GPU and clean code on
paper.  This walks the ISA of a kernel in layout order and keeps the queue of asm loads in flight: an asm `s_waitcnt vmcnt(N)`
retires all but the youngest N (vector memory operations return in order); compiler-issued vector memory instructions enter the
queue too (they count in vmcnt); a compiler-issued `s_waitcnt vmcnt(N)` retires likewise.  Every instruction outside an asm
block that reads or writes a register of a load still in the queue is reported.

    python tools/isa_inflight_check.py <file.s> [kernel-name-substring ...]        exit code 1 if anything is reported

Layout order is not execution order: a loop's back edge re-enters its header with the queue of the loop's end, which this walk
does not model (the header is visited once, with the queue of the code above it).  For the kernels this is used on, whose loop
bodies end with the same loads in flight as their prologues, the two coincide."""
import re
import sys

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")


def regs_of(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.update((m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def check_kernel(name, lines):
    queue = []          # (line number, frozenset of destination registers or empty for stores / compiler loads, is_asm)
    in_asm = False
    bad = []
    for ln, raw in lines:
        l = raw.split(";")[0].strip()
        if "#ASMSTART" in raw:
            in_asm = True
            continue
        if "#ASMEND" in raw:
            in_asm = False
            continue
        if not l or l.endswith(":") or l.startswith("."):
            continue
        op = l.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                queue = queue[len(queue) - n:] if n < len(queue) else queue
                if n == 0:
                    queue = []
            continue
        is_vmem = op.startswith(("global_load", "global_store", "global_atomic", "buffer_", "flat_", "scratch_"))
        if in_asm:
            if is_vmem and "load" in op and "_lds_" not in op:          # (LDS-DMA loads have no destination register)
                dst = l.split(None, 1)[1].split(",")[0]
                queue.append((ln, frozenset(regs_of(dst)), True))
            elif is_vmem:
                queue.append((ln, frozenset(), True))
            continue
        inflight = set().union(*[q[1] for q in queue if q[2]]) if queue else set()
        if inflight:
            used = regs_of(l.split(None, 1)[1]) if " " in l else set()
            hit = used & inflight
            if hit:
                bad.append((ln, l, sorted(hit)[:4]))
        if is_vmem:
            queue.append((ln, frozenset(), False))
    return bad


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    txt = open(path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(r"^[_A-Za-z][\w$.]*:\s*(;.*)?$", l) and not l.startswith(".L")]
    total = 0
    nker = 0
    for k, s in enumerate(starts):
        name = txt[s].split(":")[0]
        if pats and not any(p in name for p in pats):
            continue
        e = starts[k + 1] if k + 1 < len(starts) else len(txt)
        body = [(i + 1, txt[i]) for i in range(s, e)]
        if not any("#ASMSTART" in b[1] for b in body):
            continue
        nker += 1
        bad = check_kernel(name, body)
        print(f"{name[:110]}: {len(bad)} instruction(s) touch a register with an asm load in flight")
        for ln, l, hit in bad[:8]:
            print(f"    line {ln}: {l}    <- {hit}")
        total += len(bad)
    print(f"{nker} kernel(s) with asm blocks checked, {total} finding(s)")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
