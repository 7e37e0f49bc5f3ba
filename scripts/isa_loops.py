#!/usr/bin/env python3
"""Where the spills are: loops of a gfx950 ISA listing (hipcc -S --cuda-device-only) with their instruction mix.
usage: isa_loops.py file.s [kernel-substring]   — a loop = a backward branch; nested loops are listed innermost first."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else "hpt_path_kernel"
# slice out the kernel body
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*%s.*:" % want, l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end + 1]
label_at = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        label_at[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in label_at and label_at[m.group(2)] <= i:
        loops.append((label_at[m.group(2)], i))
loops.sort(key=lambda ab: ab[1] - ab[0])


def mix(a, b):
    c = {"valu": 0, "salu": 0, "vmem_ld": 0, "vmem_st": 0, "scr_ld": 0, "scr_st": 0, "lds": 0, "smem": 0, "wait": 0, "call": 0, "f64": 0}
    for l in body[a:b + 1]:
        s = l.strip()
        op = s.split(" ")[0] if s else ""
        if op.startswith("scratch_load"): c["scr_ld"] += 1
        elif op.startswith("scratch_store"): c["scr_st"] += 1
        elif op.startswith("global_load") or op.startswith("buffer_load") or op.startswith("flat_load"): c["vmem_ld"] += 1
        elif op.startswith("global_store") or op.startswith("global_atomic") or op.startswith("flat_store") or op.startswith("flat_atomic"): c["vmem_st"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"): c["smem"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_swappc") or op.startswith("s_setpc"): c["call"] += 1
        elif op.startswith("v_"):
            c["valu"] += 1
            if "_f64" in op: c["f64"] += 1
        elif op.startswith("s_"): c["salu"] += 1
    return c


print("kernel body: %d lines, %d loops; whole body: %s" % (len(body), len(loops), mix(0, len(body) - 1)))
for a, b in loops:
    inner = [(x, y) for (x, y) in loops if a <= x and y <= b and (x, y) != (a, b)]
    print("loop lines %6d..%6d (%5d lines, %d nested)  %s" % (a, b, b - a + 1, len(inner), mix(a, b)))
