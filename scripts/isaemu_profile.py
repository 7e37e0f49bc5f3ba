#!/usr/bin/env python3
"""What a bench kernel's wavefront actually executes, counted in the gfx950 interpreter of tests/isaemu (no GPU, no timing): the dynamic instruction mix of the shipped binary on a
crop of the workload's parity fixture — how much of the stream is arithmetic, how much is the register allocator's traffic (scratch spills / reloads of VGPRs, v_writelane /
v_readlane spills of SGPRs), the VALU lane utilisation (active lanes per vector instruction / 64: the GPU's SQ_ACTIVE_INST_VALU x lanes counter measures the same thing).

    python scripts/isaemu_profile.py <crop> [workload ...]        -> profiles/r05_isaemu_instruction_mix.md"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import hash_rd, load_case   # noqa: E402
from tests.isaemu import run as R   # noqa: E402
from tests.wavemu import emu as w   # noqa: E402
K = R.kernel_symbol
WORKLOADS = {"bunny": ("b8", "measured", K(False, False, 3, 4, 0, True, False, True), w.K_MEASURED_STEAL), "killeroo": ("cfg1", "basic", K(False, False, 1, 4, 0, True, False, True), w.K_BASIC_STEAL),
             "anim": ("anim", "basic_i", K(False, True, 1, 4, 0, True, False, True), w.K_STEAL), "soup": ("env", "basic", K(False, False, 1, 3, 0, True, False, True), w.K_BASIC_STEAL),
             "metal": ("metal", "lean", K(False, False, 61, 4, 0, True, False, True), w.K_LEAN_STEAL)}


def klass(op):
    if op.startswith("scratch_load"):
        return "scratch reloads (VGPR spills)"
    if op.startswith("scratch_store"):
        return "scratch stores (VGPR spills)"
    if op.startswith(("v_readlane", "v_writelane")):
        return "v_readlane / v_writelane (SGPR spills)"
    if op.startswith(("global_", "flat_")):
        return "global / flat memory"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "s_waitcnt / s_nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_getpc")):
        return "branches / calls"
    if op.startswith("s_load"):
        return "scalar loads"
    if op.startswith("s_"):
        return "scalar ALU"
    if op.startswith(("v_mov", "v_pk_mov", "v_cndmask")):
        return "vector moves / selects"
    if op.startswith("v_cmp"):
        return "vector compares"
    if "f64" in op:
        return "vector f64"
    return "vector arithmetic"


def main():
    n = int(sys.argv[1])
    names = sys.argv[2:] or list(WORKLOADS)
    rows, classes = {}, []
    for name in names:
        case, unit, sym, kid = WORKLOADS[name]
        s = load_case(case)
        rd = hash_rd(s, seed=3)
        rd.x_start += (rd.x_count - n) // 2; rd.y_start += (rd.y_count - n) // 2; rd.x_count = rd.y_count = n
        counts, lanes = {}, [0, 0]

        def tr(wv, ins, counts=counts, lanes=lanes):
            k = klass(ins.op)
            counts[k] = counts.get(k, 0) + 1
            if ins.op.startswith("v_") and not ins.op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
                lanes[0] += 1; lanes[1] += bin(wv.exec).count("1")
        f, info = R.BinaryRender(s, R.code_object_for(unit, sym), sym, kid).render(s.camera, rd, trace=tr)
        total = sum(counts.values())
        rows[name] = (counts, total, lanes[1] / (64.0 * max(lanes[0], 1)), info["samples"])
        for k in counts:
            if k not in classes:
                classes.append(k)
        print(name, total, "instructions", file=sys.stderr)
    order = ["vector arithmetic", "vector f64", "vector moves / selects", "vector compares", "scratch reloads (VGPR spills)", "scratch stores (VGPR spills)", "v_readlane / v_writelane (SGPR spills)",
             "global / flat memory", "LDS", "scalar ALU", "scalar loads", "branches / calls", "s_waitcnt / s_nop"]
    out = ["# r05 — what the bench kernels' wavefronts execute (gfx950 interpreter, `scripts/isaemu_profile.py %d`: one wave over a %d x %d crop of each workload's parity fixture; counts, not time)" % (n, n, n), "",
           "| instruction class | " + " | ".join(names) + " |", "|---|" + "---:|" * len(names)]
    for k in order:
        out.append("| %s | " % k + " | ".join("%.1f %%" % (100.0 * rows[nm][0].get(k, 0) / rows[nm][1]) for nm in names) + " |")
    out.append("| **wave-instructions per camera sample** | " + " | ".join("%d" % (rows[nm][1] // max(rows[nm][3], 1)) for nm in names) + " |")
    out.append("| **VALU lane utilisation** (active lanes / 64) | " + " | ".join("%.0f %%" % (100.0 * rows[nm][2]) for nm in names) + " |")
    open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_isaemu_instruction_mix.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
