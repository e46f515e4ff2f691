#!/usr/bin/env python3
"""Instruction counts of one kernel from hipcc's assembly listing, by class and by basic block.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only X.hip -o X.s
    python3 tools/isa_count.py X.s <substring of the mangled kernel name> [--blocks N]

Classes: mfma, pk (v_pk_*: two fp32 lane-operations per lane = 2 issue slots on the SIMD-32 vector ALU), valu (every other
v_* that is not a matrix instruction), salu (s_* but waits / nops / branches / barriers), lds (ds_*), vmem (global_/buffer_/
flat_/scratch_), nop (s_nop, with the idle cycles its operand asks for), wait (s_waitcnt), other.  `valu_slots` = valu + 2 * pk
is what bench.py prices against the vector issue rate (78.6 T lane-operations/s).  The N largest basic blocks (label to label)
are listed separately: the hot loop of a kernel is normally the largest one.  No GPU needed."""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_pk_"):
        return "pk"
    if op.startswith("v_"):
        return "valu"
    if op == "s_nop":
        return "nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier") or op.startswith("s_cbranch") or op.startswith("s_branch") or op in ("s_endpgm", "s_setprio", "s_sleep"):
        return "other"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.split("_")[0] in ("global", "buffer", "flat", "scratch"):
        return "vmem"
    return "other"


def kernel_lines(path, sub):
    """-> (name, [lines]) of the first kernel whose mangled name contains `sub`"""
    name, body = None, []
    for line in open(path):
        s = line.rstrip("\n")
        if name is None:
            m = re.match(r"^(_Z\w+):", s)
            if m and sub in m.group(1):
                name = m.group(1)
            continue
        if s.startswith(".Lfunc_end"):
            break
        body.append(s)
    if name is None:
        sys.exit(f"isa_count: no kernel matching {sub!r} in {path}")
    return name, body


def count(lines):
    c = collections.Counter()
    for s in lines:
        t = s.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        cls = classify(op)
        c[cls] += 1
        if cls == "nop":
            m = re.match(r"s_nop\s+(\d+)", t)
            c["nop_cycles"] += int(m.group(1)) + 1 if m else 1
        c["op:" + op] += 1
    c["valu_slots"] = c["valu"] + 2 * c["pk"]
    return c


def blocks(lines):
    out, cur, label = [], [], "<entry>"
    for s in lines:
        t = s.strip()
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            out.append((label, cur))
            label, cur = m.group(1), []
        else:
            cur.append(s)
    out.append((label, cur))
    return out


def show(title, c, top_ops=0):
    keys = ("mfma", "valu", "pk", "valu_slots", "salu", "lds", "vmem", "nop", "nop_cycles", "wait", "other")
    print(f"{title}: " + " ".join(f"{k}={c[k]}" for k in keys if c[k]))
    if top_ops:
        ops = sorted(((n, k[3:]) for k, n in c.items() if k.startswith("op:")), reverse=True)[:top_ops]
        print("    " + ", ".join(f"{k} x{n}" for n, k in ops))


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    nblocks = int(sys.argv[sys.argv.index("--blocks") + 1]) if "--blocks" in sys.argv else 3
    name, body = kernel_lines(sys.argv[1], sys.argv[2])
    print(name)
    show("whole kernel", count(body), top_ops=12)
    bl = sorted(blocks(body), key=lambda b: -sum(count(b[1])[k] for k in ("mfma", "valu", "pk", "salu", "lds", "vmem")))[:nblocks]
    for label, ls in bl:
        show(f"block {label}", count(ls), top_ops=10)


if __name__ == "__main__":
    main()
