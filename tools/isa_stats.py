#!/usr/bin/env python3
"""Instruction mix of a kernel's hottest loop from hipcc -S output.
usage: isa_stats.py <file.s> <kernel-name-substring>"""
import re
import sys
from collections import Counter


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r'^(_Z[A-Za-z0-9_]*' + re.escape(key) + r'[A-Za-z0-9_]*):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M)
    body = m.group(2)
    lines = []
    for l in body.split('\n'):
        l = l.split(';')[0].strip()
        if not l or (l.startswith('.') and not l.endswith(':')):
            continue
        lines.append(l)
    labels = {}
    idx = 0
    instrs = []
    for l in lines:
        if l.endswith(':'):
            labels[l[:-1]] = idx
        else:
            instrs.append(l)
            idx += 1
    # backward branches = loops
    loops = []
    for i, l in enumerate(instrs):
        if l.startswith(('s_cbranch', 's_branch')):
            tgt = l.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i))
    loops.sort(key=lambda ab: ab[1] - ab[0], reverse=True)
    print(f"kernel {m.group(1)}: {len(instrs)} instructions; loops (start,end,len):",
          [(a, b, b - a) for a, b in loops[:6]])
    for a, b in loops[:int(sys.argv[3]) if len(sys.argv) > 3 else 2]:
        seg = instrs[a:b + 1]
        cat = Counter()
        ops = Counter()
        for l in seg:
            op = l.split()[0]
            ops[op] += 1
            if op.startswith('v_readlane') or op.startswith('v_writelane'):
                cat['sgpr-spill lane ops'] += 1
            elif op.startswith('v_'):
                cat['VALU'] += 1
            elif op.startswith('s_waitcnt') or op.startswith('s_nop'):
                cat['wait/nop'] += 1
            elif op.startswith('s_'):
                cat['SALU'] += 1
            elif op.startswith('ds_'):
                cat['LDS'] += 1
            elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                cat['VMEM'] += 1
            else:
                cat['other'] += 1
        print(f"loop [{a},{b}] len {b - a + 1}:", dict(cat))
        print("  top ops:", ops.most_common(28))


if __name__ == "__main__":
    main()
