#!/usr/bin/env python3
"""Where a bench kernel's wavefront spends its instructions and where its lanes idle, by SOURCE FUNCTION — counted in the gfx950 interpreter of tests/isaemu on a
kernel object built with line tables (no GPU; the kernels are VALU-issue bound, so issued vector instructions are the time and active lanes / 64 the utilisation).

    python scripts/isaemu_lineprofile.py <workload> [crop] [spp]     -> profiles/r06_lineprofile_<workload>.md

The translation unit is compiled as the Makefile compiles it plus -gline-tables-only (into /tmp/isaemu_g/), every executed instruction is attributed to the chain of
inlined frames llvm-symbolizer reports for its address, and two tables come out: by SECTION of the persistent loop (the frame directly under hpt_path_kernel) and by the
innermost function."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import hash_rd, load_case   # noqa: E402
from tests.isaemu import run as R   # noqa: E402
from tests.isaemu import gfx950 as g   # noqa: E402
from tests.wavemu import emu as w   # noqa: E402
K = R.kernel_symbol
WORKLOADS = {"bunny": ("b8", "measured", K(False, False, 3, 4, 0, True, False, True), w.K_MEASURED_STEAL), "killeroo": ("cfg1", "basic", K(False, False, 1, 4, 0, True, False, True), w.K_BASIC_STEAL),
             "anim": ("anim", "basic_i", K(False, True, 1, 4, 0, True, False, True), w.K_STEAL), "soup": ("env", "basic", K(False, False, 1, 3, 0, True, False, True), w.K_BASIC_STEAL),
             "metal": ("metal", "lean", K(False, False, 61, 4, 0, True, False, True), w.K_LEAN_STEAL)}
SYMBOLIZER = "/opt/rocm/lib/llvm/bin/llvm-symbolizer"


def build_g(unit):
    out = "/tmp/isaemu_g"
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, "hpt_kernels_%s.o" % unit)
    src = os.path.join(ROOT, "pbrt-v2_amd", "csrc", "hpt_kernels_%s.hip" % unit)
    hdrs = [os.path.join(ROOT, "pbrt-v2_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "pbrt-v2_amd", "csrc")) if f.endswith(".h")]
    if os.path.exists(obj) and all(os.path.getmtime(obj) > os.path.getmtime(h) for h in hdrs + [src]):
        return obj
    mk = subprocess.run(["make", "-n", "-B", "-C", os.path.join(ROOT, "pbrt-v2_amd"), "build/hpt_kernels_%s.o" % unit], capture_output=True, text=True, check=True).stdout
    cmd = next(l for l in mk.split("\n") if "hipcc" in l and "hpt_kernels_%s.hip" % unit in l).split()
    cmd = [c for c in cmd]
    cmd[cmd.index("-o") + 1] = obj
    cmd.insert(1, "-gline-tables-only")
    print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd, cwd=os.path.join(ROOT, "pbrt-v2_amd"))
    return obj


def symbolize(co, addrs):
    """{addr: [innermost frame, ..., outermost]} (function names without template arguments)"""
    inp = "\n".join("0x%x" % a for a in addrs) + "\n"
    out = subprocess.run([SYMBOLIZER, "--obj=" + co, "--inlines", "--functions=short", "--output-style=LLVM"], input=inp, capture_output=True, text=True, check=True).stdout
    frames, res, i = [], {}, 0
    blocks = out.split("\n\n")
    for a, b in zip(addrs, blocks):
        ls = [l for l in b.split("\n") if l.strip()]
        names = [re.sub(r"<.*", "", ls[k]).strip() for k in range(0, len(ls) - 1, 2)]
        lines = [ls[k + 1].strip() for k in range(0, len(ls) - 1, 2)]
        res[a] = (names, lines)
    return res


def main():
    name = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    case, unit, sym, kid = WORKLOADS[name]
    s = load_case(case)
    rd = hash_rd(s, seed=3)
    if len(sys.argv) > 3:
        rd.spp = int(sys.argv[3])
    rd.x_start += (rd.x_count - n) // 2; rd.y_start += (rd.y_count - n) // 2; rd.x_count = rd.y_count = n
    co = R.code_object(build_g(unit), out_dir="/tmp/isaemu_g")
    br = R.BinaryRender(s, co, sym, kid)
    stat = {}        # addr -> [issues, valu issues, active lanes of the valu issues]

    def tr(wv, ins):
        e = stat.get(ins.addr)
        if e is None:
            e = stat[ins.addr] = [0, 0, 0, ins.op]
        e[0] += 1
        if ins.op.startswith("v_") and not ins.op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            e[1] += 1; e[2] += bin(wv.exec).count("1")
    f, info = br.render(s.camera, rd, trace=tr)
    addrs = sorted(stat)
    symb = symbolize(co, addrs)
    total = sum(e[0] for e in stat.values()); tv = sum(e[1] for e in stat.values()); tl = sum(e[2] for e in stat.values())
    by_sec, by_fn, by_line, by_src, by_walk = {}, {}, {}, {}, {}
    for a in addrs:
        names, lines = symb.get(a, (["?"], ["?"]))
        # the section: the frame directly under the kernel (outermost is the kernel itself); code of the kernel body proper: "(loop)"
        chain = [x for x in names if x]
        if chain and chain[-1].startswith("hpt_path_kernel"):
            sec = chain[-2] if len(chain) > 1 else "(kernel loop)"
            sub = chain[-3] if len(chain) > 2 else sec
        else:
            sec = chain[-1] if chain else "?"       # an out-of-line device function (wave_kd_run, irreg_eval, ...)
            sub = chain[-2] if len(chain) > 1 else sec
        e = stat[a]
        src = re.sub(r"^.*/", "", lines[0]) if lines else "?"
        src = re.sub(r":\d+$", "", src)                        # file:line (the column dropped)
        # the line of the WALK this instruction belongs to: the frame of traverse_steal (inlined callees count at their call site)
        if "traverse_steal" in chain:
            wl = re.sub(r"^.*/", "", lines[chain.index("traverse_steal")]) if len(lines) > chain.index("traverse_steal") else "?"
            wl = re.sub(r":\d+$", "", wl)
            r = by_walk.setdefault(wl, [0, 0, 0])
            r[0] += e[0]; r[1] += e[1]; r[2] += e[2]
        for d, k in ((by_sec, sec), (by_fn, chain[0] if chain else "?"), (by_line, (sec, sub)), (by_src, (src, chain[0] if chain else "?"))):
            r = d.setdefault(k, [0, 0, 0])
            r[0] += e[0]; r[1] += e[1]; r[2] += e[2]
    out = ["# r06 — `%s`: issued instructions and VALU lane utilisation by source function (gfx950 interpreter, one wave, %d x %d crop of fixture `%s`, %d spp, %d camera samples; `scripts/isaemu_lineprofile.py`)" % (name, n, n, case, rd.spp, info["samples"]), "",
           "%d wave-instructions (%d per camera sample), %d of them vector (%.1f %%), VALU lane utilisation %.1f %%.  `share` = of the VALU instructions issued (the kernels are VALU-issue bound); `lanes` = active lanes / 64 of those; `lost` = share x (1 - lanes): the part of the whole kernel's VALU issue slots spent on idle lanes there." % (total, total // max(info["samples"], 1), tv, 100.0 * tv / total, 100.0 * tl / (64.0 * tv)), "",
           "## by section of the persistent loop (frame directly under `hpt_path_kernel`; out-of-line device functions by their own name)", "", "| section | all instr. | VALU share | lanes | lost |", "|---|---:|---:|---:|---:|"]
    for k, r in sorted(by_sec.items(), key=lambda kv: -kv[1][1]):
        if r[1] * 1000 < tv: continue
        out.append("| `%s` | %.1f %% | %.1f %% | %.0f %% | %.1f %% |" % (k, 100.0 * r[0] / total, 100.0 * r[1] / tv, 100.0 * r[2] / (64.0 * max(r[1], 1)), 100.0 * (r[1] - r[2] / 64.0) / tv))
    out += ["", "## by (section, function inlined directly into it)", "", "| section / function | VALU share | lanes | lost |", "|---|---:|---:|---:|"]
    for k, r in sorted(by_line.items(), key=lambda kv: -kv[1][1]):
        if r[1] * 200 < tv: continue
        out.append("| `%s` / `%s` | %.1f %% | %.0f %% | %.1f %% |" % (k[0], k[1], 100.0 * r[1] / tv, 100.0 * r[2] / (64.0 * max(r[1], 1)), 100.0 * (r[1] - r[2] / 64.0) / tv))
    out += ["", "## by innermost function", "", "| function | VALU share | lanes | lost |", "|---|---:|---:|---:|"]
    for k, r in sorted(by_fn.items(), key=lambda kv: -kv[1][1]):
        if r[1] * 200 < tv: continue
        out.append("| `%s` | %.1f %% | %.0f %% | %.1f %% |" % (k, 100.0 * r[1] / tv, 100.0 * r[2] / (64.0 * max(r[1], 1)), 100.0 * (r[1] - r[2] / 64.0) / tv))
    out += ["", "## by source line (innermost frame; the 40 heaviest)", "", "| file:line (function) | VALU share | lanes | lost |", "|---|---:|---:|---:|"]
    for k, r in sorted(by_src.items(), key=lambda kv: -kv[1][1])[:40]:
        out.append("| `%s` (`%s`) | %.2f %% | %.0f %% | %.2f %% |" % (k[0], k[1], 100.0 * r[1] / tv, 100.0 * r[2] / (64.0 * max(r[1], 1)), 100.0 * (r[1] - r[2] / 64.0) / tv))
    out += ["", "## the walk (`traverse_steal`) by its own source line — inlined callees (`trav_node4`, `tri_test`, `trav_begin` ...) counted at their call site; `all` = share of ALL issued instructions", "",
            "| line of hpt_kernels_impl.h | all instr. | VALU share | lanes |", "|---|---:|---:|---:|"]
    def lno(k):
        m = re.search(r":(\d+)$", k)
        return int(m.group(1)) if m else 0
    for k, r in sorted(by_walk.items(), key=lambda kv: lno(kv[0])):
        if r[0] * 500 < total: continue
        out.append("| `%s` | %.2f %% | %.2f %% | %.0f %% |" % (k, 100.0 * r[0] / total, 100.0 * r[1] / tv, 100.0 * r[2] / (64.0 * max(r[1], 1))))
    p = os.path.join(ROOT, "profiles", "r06_lineprofile_%s.md" % name)
    open(p, "w").write("\n".join(out) + "\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
