"""Do two builds of a kernel file contain the same machine code for the same kernels?

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include --cuda-device-only -S file.hip -o new.s
    python tools/isa_diff.py old.s new.s

Compares the instruction streams function by function (labels renumbered, comments and assembler directives dropped) and
prints the kernels that differ or exist on one side only.  Used in round 4 to show what the `bool GRP` template parameter
(csrc/dip_group.h: grouped multi-instance launches; GRP = true kernels are new and skipped here) did to the solo kernels:
127 of 143 instruction streams identical, the other 16 (streaming / helper kernels whose pointers lost `__restrict__` on
the way through DIP_GRP_PTR) differ by <= 5 instructions of scheduling.
"""
import re
import sys


def functions(path):
    out, name, cur = {}, None, []
    for line in open(path, errors="replace"):
        s = line.rstrip("\n")
        m = re.match(r"^(_Z[\w$.]*|[A-Za-z_][\w$.]*):\s*(;.*)?$", s)
        if m and not s.startswith(".L"):
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if name is None:
            continue
        t = s.split(";")[0].strip()
        if not t:
            continue
        if t.startswith(".Lfunc_end"):
            name = None
            continue
        if t.startswith(".") and not t.startswith(".L"):      # directives (.p2align, .loc, ...)
            continue
        cur.append(t)
    return out


def normalise(body):
    labels = {}
    res = []
    for t in body:
        m = re.match(r"^(\.L[\w$.]+):$", t)
        if m:
            labels.setdefault(m.group(1), f".L{len(labels)}")
    for t in body:
        res.append(re.sub(r"\.L[\w$.]+", lambda m: labels.get(m.group(0), m.group(0)), t))
    return res


def canon(name):
    """Key of a kernel that survives the `bool GRP` parameter (csrc/dip_group.h): (function name, template arguments
    without a trailing GRP = false).  Returns (key, is_grouped)."""
    m = re.match(r"^_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:
        return name, False
    n = int(m.group(1))
    base = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    args = ""
    if rest.startswith("I"):
        mm = re.match(r"I(.*?)EEv", rest)
        args = mm.group(1) if mm else rest
    grouped = False
    if "DipGrp8DipNoGrp" in rest or "conditional" in rest:
        if args.endswith("Lb1E"):
            grouped = True
        elif args.endswith("Lb0E"):
            args = args[:-4]
    return f"{base}<{args}>", grouped


def main():
    a, b = functions(sys.argv[1]), functions(sys.argv[2])
    skip = ("__hip_cuid", "amdhsa.")
    a = {canon(k)[0]: v for k, v in a.items() if not k.startswith(skip)}
    grouped = [k for k in b if canon(k)[1]]
    b = {canon(k)[0]: v for k, v in b.items() if not k.startswith(skip) and k not in grouped}
    same = diff = 0
    for k in sorted(set(a) | set(b)):
        if k not in a:
            print("only in new:", k)
            continue
        if k not in b:
            print("ONLY IN OLD:", k)
            diff += 1
            continue
        na, nb = normalise(a[k]), normalise(b[k])
        if na == nb:
            same += 1
        else:
            diff += 1
            first = next((i for i, (x, y) in enumerate(zip(na, nb)) if x != y), min(len(na), len(nb)))
            print(f"DIFFERENT: {k}: {len(na)} vs {len(nb)} instructions, first difference at {first}")
    print(f"{same} functions identical, {diff} different; {len(grouped)} grouped kernels (new)")
    return 1 if diff else 0


if __name__ == "__main__":
    sys.exit(main())
