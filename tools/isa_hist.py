#!/usr/bin/env python3
"""Instruction-class histogram of a range of a kernel's ISA, in the classes of tools/microbench/issue_table.hip,
and (with --table) the SIMD time that range needs at 1 / 2 / 4 / 8 wavefronts per SIMD according to the measured
per-class issue costs.

usage: isa_hist.py <file.s> <kernel-name-substring> --blocks .LBB24_71,.LBB24_75 [--table profiles/r03_issue_table.json]
       isa_hist.py <file.s> <kernel-name-substring> --range 1570:2740

A block is the run of instructions from its label up to the next label.  The classes:
  fast      v_fma/v_mul/v_add/v_sub/v_mov/v_xor/v_add_u32 ... in VOP1/VOP2 form with VGPR / literal / inline operands
  slow      every other single-pass VALU form: VOP3 encodings, an SGPR source, v_max/v_min, shifts, converts,
            compares, v_div_scale/fmas/fixup, packed fp32 (v_pk_*), DPP moves / adds, v_cndmask with an SGPR-pair mask
  cnd_vcc   v_cndmask_b32 (plain or DPP) selecting on VCC
  trans     v_rcp_f32 and the other quarter-rate ops
  bperm     ds_bpermute / other LDS
  salu, nop, branch, wait, vmem
"""
import argparse
import json
import re
from collections import Counter

FAST_OPS = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_xor_b32", "v_add_u32",
            "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_fmac_f32", "v_mac_f32"}
TRANS_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def classify(ins):
    parts = ins.replace(",", " ").split()
    op = parts[0]
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "bperm"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if not op.startswith("v_"):
        return "other"
    base = re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", op)
    is_dpp = ("dpp" in op) or any(p.startswith(("row_", "wave_", "quad_perm", "row_bcast")) for p in parts)
    if base in TRANS_OPS:
        return "trans"
    if base.startswith("v_cndmask"):
        return "cnd_vcc" if parts[-1] == "vcc" or (is_dpp and "vcc" in parts) else "slow"
    if is_dpp or base.startswith("v_pk_"):
        return "slow"
    if base in FAST_OPS and not op.endswith("_e64"):
        # an SGPR / VCC / EXEC source moves it to the slow class
        if any(re.fullmatch(r"-?\|?(s\d+|s\[\d+:\d+\]|vcc(_lo|_hi)?|exec(_lo|_hi)?|m0)\|?", p) for p in parts[2:]):
            return "slow"
        return "fast"
    return "slow"


def kernel_items(path, name):
    s = open(path).read()
    m = re.search(r'^(_Z[A-Za-z0-9_]*' + re.escape(name) + r'[A-Za-z0-9_]*):[^\n]*\n(.*?)^\.Lfunc_end\d+:', s, re.S | re.M)
    if not m:
        raise SystemExit("kernel not found")
    items = []
    for l in m.group(2).split("\n"):
        l = l.split(";")[0].strip()
        if not l:
            continue
        if l.endswith(":"):
            items.append(("label", l[:-1]))
        elif not l.startswith("."):
            items.append(("ins", l))
    return m.group(1), items


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("file")
    ap.add_argument("kernel")
    ap.add_argument("--blocks", default="")
    ap.add_argument("--range", default="")
    ap.add_argument("--table", default="")
    ap.add_argument("--ops", action="store_true", help="also list the opcodes of each class")
    a = ap.parse_args()
    name, items = kernel_items(a.file, a.kernel)
    ins, pos = [], {}
    for kind, x in items:
        if kind == "label":
            pos[x] = len(ins)
        else:
            ins.append(x)
    starts = sorted(set(pos.values()))
    sel = []
    if a.range:
        b, e = [int(v) for v in a.range.split(":")]
        sel = ins[b:e + 1]
    for lab in [b for b in a.blocks.split(",") if b]:
        b = pos[lab]
        e = min([s for s in starts if s > b] + [len(ins)])
        sel += ins[b:e]
    hist = Counter(classify(x) for x in sel)
    ops = {}
    for x in sel:
        ops.setdefault(classify(x), Counter())[x.split()[0]] += 1
    print(f"{name}: {len(sel)} instructions")
    for k, v in hist.most_common():
        print(f"  {k:8s} {v:5d}" + ("   " + " ".join(f"{o}:{n}" for o, n in ops[k].most_common(12)) if a.ops else ""))
    if a.table:
        t = json.load(open(a.table))
        rows = {(r["class"], r["form"]): r for r in t["rows"]}

        def cost(cls, form, shape):
            return rows[(cls, form)][shape]["ns_per_op_per_simd"]

        def class_cost(k, shape):
            if shape == "lone":
                # a wavefront alone on its SIMD: 4.44 cycles per instruction of any class (v_rcp_f32: 8.44) at ~2.3 GHz
                return rows[("v_rcp_f32", "D")]["cu_1wave"]["wave_cycles_per_op"] / 2.3 if k == "trans" else 4.44 / 2.3
            if k == "cnd_vcc":
                # IN SITU (a select between other instructions): from the V_SEG position chain, 2 v_add_f32 + 2 selects per
                # round.  Back to back, VCC selects cost 9.6 ns each (row "v_cndmask_b32 (VCC)"): the kernels never do that.
                return (cost("seg_fwd_xy<10>, per round", "D", shape) - 2 * cost("v_add_f32", "D", shape)) / 2
            rep = {"fast": ("v_mul_f32", "D"), "slow": ("v_cndmask_b32 (SGPR pair)", "D"), "trans": ("v_rcp_f32", "D"),
                   "bperm": ("ds_bpermute_b32", "I")}
            return cost(*rep.get(k, ("s_nop 0", "D")), shape)

        print("  SIMD time of this range by the measured per-class costs (ns; salu / branch / wait / vmem priced as s_nop;")
        print("  'lone' = one wavefront per SIMD, issue-bound; chip_N = N wavefronts per SIMD, the SIMD's own time per wavefront):")
        for shape in ("lone", "chip_2", "chip_4", "chip_8"):
            parts = {k: v * class_cost(k, shape) for k, v in hist.items()}
            print(f"    {shape:6s}: {sum(parts.values()):8.1f} ns  ("
                  + ", ".join(f"{k} {parts[k]:.0f}" for k, _ in hist.most_common()) + ")")

if __name__ == "__main__":
    main()
