#!/usr/bin/env python3
"""Instruction accounting of a kernel's loops from `hipcc -S --cuda-device-only` output.

usage: isa_stats.py <file.s> <kernel-name-substring> [--loop N] [--dump]

Lists the kernel's loops (backward branches), and for loop N (default: the longest loop nested
inside another one, i.e. the SGD iteration) the basic blocks with their instruction counts by class:
VALU, of which DPP / division sequence / transcendental; SALU; s_nop; LDS; branches.  A lone wavefront
issues roughly one instruction per 4 cycles, so the block sums along a path are its cost."""
import argparse
import re
from collections import Counter


def classify(ins):
    op = ins.split()[0]
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("kernel")
    ap.add_argument("--loop", type=int, default=-1)
    ap.add_argument("--dump", action="store_true")
    a = ap.parse_args()
    s = open(a.file).read()
    m = re.search(r'^(_Z[A-Za-z0-9_]*' + re.escape(a.kernel) + r'[A-Za-z0-9_]*):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M)
    if not m:
        raise SystemExit("kernel not found")
    items = []          # ("label", name) | ("ins", text)
    for l in m.group(2).split("\n"):
        l = l.split(";")[0].strip()
        if not l:
            continue
        if l.endswith(":"):
            items.append(("label", l[:-1]))
        elif not l.startswith("."):
            items.append(("ins", l))
    pos = {}
    n = 0
    for kind, x in items:
        if kind == "label":
            pos[x] = n
        else:
            n += 1
    ins = [x for k, x in items if k == "ins"]
    loops = []
    for i, l in enumerate(ins):
        if l.startswith(("s_cbranch", "s_branch")):
            tgt = l.split()[-1]
            if tgt in pos and pos[tgt] <= i:
                loops.append((pos[tgt], i))
    loops.sort(key=lambda ab: ab[0])
    print(f"kernel {m.group(1)}: {len(ins)} instructions")
    for k, (b, e) in enumerate(loops):
        depth = sum(1 for (b2, e2) in loops if b2 <= b and e <= e2) - 1
        print(f"  loop {k}: [{b}, {e}] len {e - b + 1} depth {depth}")
    if a.loop >= 0:
        b, e = loops[a.loop]
    else:
        inner = [(b, e) for (b, e) in loops if any(b2 <= b and e <= e2 and (b2, e2) != (b, e) for (b2, e2) in loops)]
        b, e = max(inner or loops, key=lambda ab: ab[1] - ab[0])
    print(f"loop [{b}, {e}]:")
    rev = {}
    for name, p in pos.items():
        rev.setdefault(p, []).append(name)
    blocks = []
    cur = None
    for i in range(b, e + 1):
        if i in rev or cur is None:
            cur = {"name": ",".join(rev.get(i, ["(entry)"])), "ins": []}
            blocks.append(cur)
        cur["ins"].append(ins[i])
        if ins[i].startswith(("s_cbranch", "s_branch")) and i < e:
            cur = {"name": "(fallthrough)", "ins": []}
            blocks.append(cur)
    tot = Counter()
    for blk in blocks:
        if not blk["ins"]:
            continue
        c = Counter(classify(x) for x in blk["ins"])
        extra = Counter()
        for x in blk["ins"]:
            op = x.split()[0]
            if "dpp" in x or "row_sh" in x or "wave_sh" in x:
                extra["dpp"] += 1
            if op.startswith(("v_div_", "v_rcp")):
                extra["div"] += 1
            if op.startswith("v_cndmask"):
                extra["cndmask"] += 1
            if op.startswith("v_mov"):
                extra["mov"] += 1
        tot.update(c)
        tot.update({"x_" + k: v for k, v in extra.items()})
        last = blk["ins"][-1] if blk["ins"][-1].startswith(("s_cbranch", "s_branch")) else ""
        print(f"  {blk['name']:<28} {len(blk['ins']):4d}  " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) +
              "  | " + " ".join(f"{k}={v}" for k, v in sorted(extra.items())) + (f"  -> {last}" if last else ""))
        if a.dump:
            for x in blk["ins"]:
                print("        " + x)
    print("  total:", dict(tot))


if __name__ == "__main__":
    main()
