#!/usr/bin/env python3
"""Instruction table of a kernel's loops from an ISA listing (hipcc -S): for every innermost loop (a backward branch), the number of
wave-instructions per class and per workgroup barrier, priced with the cycles per wave64 instruction measured by tools/ubench.hip
(v_fma_f32 2.0, f64 add / mul / fma 3.7, v_cvt_f64_f32 3.7; transcendentals 8).  Usage: isa_loop_table.py file.s 'kernel-name-substring' ['second substring']"""
import re
import sys

# cycles a SIMD's VALU is busy per wave64 instruction (tools/ubench.hip)
def classify(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")): return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")): return "vmem_st"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_"):
        if "_f64" in op and not op.startswith("v_cvt"): return "valu_f64"
        if op.startswith("v_cvt") and "f64" in op: return "valu_cvt64"
        if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log")): return "valu_trans"
        return "valu_f32"
    return "other"

COST = {"valu_f32": 2.0, "valu_f64": 3.7, "valu_cvt64": 3.7, "valu_trans": 8.0, "lds": 0.0, "vmem_ld": 0.0, "vmem_st": 0.0, "salu": 0.0,
        "waitcnt": 0.0, "barrier": 0.0, "mfma": 16.0, "other": 0.0}


def main():
    path, keys = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and l.rstrip().split(":")[0] and all(k in l for k in keys) and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: labels[m.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # innermost only
    inner = loops if "--all" in sys.argv else [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    print(f"kernel at line {start + 1}, {end - start} lines, {len(inner)} innermost loops")
    for a, b in inner:
        cnt = {}
        for l in body[a:b + 1]:
            m = re.match(r"\s+([a-z_0-9]+)", l)
            if not m or l.strip().startswith((";", ".")): continue
            c = classify(m.group(1))
            cnt[c] = cnt.get(c, 0) + 1
        nb = max(cnt.get("barrier", 0), 1)
        valu = sum(cnt.get(k, 0) * COST[k] for k in cnt)
        total = sum(cnt.values())
        if total < 40: continue
        per = {k: round(v / nb, 1) for k, v in sorted(cnt.items()) if k not in ("waitcnt", "other")}
        print(f"  loop lines {start + a + 1}-{start + b + 1}: {total} instructions, {cnt.get('barrier', 0)} barriers; per barrier: {per}; VALU busy cycles per barrier {valu / nb:.0f}")


if __name__ == "__main__":
    main()
