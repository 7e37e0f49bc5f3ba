#!/usr/bin/env python3
"""Kernel resource table of the path kernels (hipcc -Rpass-analysis=kernel-resource-usage, gfx950, the build's own per-file flags) ->
profiles/<tag>_kernel_resources.md.  usage: scripts/kres_all.py r03"""
import concurrent.futures
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PK = os.path.join(ROOT, "pbrt-v2_amd")
tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
FLAGS = "--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -munsafe-fp-atomics".split()
ILP = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]
TUS = {"hpt_kernels_basic": ILP, "hpt_kernels_basic_i": [], "hpt_kernels_measured": ILP + ["-mllvm", "-greedy-regclass-priority-trumps-globalness=1"], "hpt_kernels_measured_i": [],
       "hpt_kernels_all": [], "hpt_kernels_all_i": [], "hpt_kernels_ext": ILP + ["-mllvm", "-vgpr-regalloc=basic"], "hpt_kernels_ext_i": ["-mllvm", "-vgpr-regalloc=basic"], "hpt_kernels_lean": ILP}   # (= FLAGS_* of pbrt-v2_amd/Makefile, round 5)


def run(tu):
    cmd = ["/opt/rocm/bin/hipcc", *FLAGS, *TUS[tu], "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(PK, "csrc", tu + ".hip"), "-o", "/dev/null"]
    return tu, subprocess.run(cmd, capture_output=True, text=True, cwd=PK).stderr


rows = []
with concurrent.futures.ThreadPoolExecutor(8) as ex:
    for tu, err in ex.map(run, TUS):
        cur = {}
        for line in err.splitlines():
            m = re.search(r"remark: .*?Function Name: (\S+)", line)
            if m:
                cur = {"tu": tu, "name": m.group(1)}
                rows.append(cur)
                continue
            for key, pat in (("vgprs", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("sgprs", r" SGPRs: (\d+)"),
                             ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)")):
                m = re.search(pat, line)
                if m and cur is not None:
                    cur[key] = int(m.group(1))
out = os.path.join(ROOT, "profiles", "%s_kernel_resources.md" % tag)
with open(out, "w") as f:
    f.write("# %s — kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950; the build's own flags per translation unit)\n\n" % tag)
    f.write("hpt_path_kernel<COUNT, INST, MATS, WAVES, EE, PHASED, DL, STEAL, WIN>; WIN = the window samplers' kernels (Sampler \"halton\"); MATS 1 = matte + plastic, 3 = + measured, 15 = + metal + substrate, 31 = + extension set;\n"
            "MATS 61 = the lean extension set; TOP (last argument) = the walk from the top-level tree;\n"
            "the instance-free kernels of MATS 1, 3, 31 and 61 are scheduled with -amdgpu-sched-strategy=max-ilp (MATS 3 also with -greedy-regclass-priority-trumps-globalness); round 5: the two extension units (MATS 31) are allocated by -vgpr-regalloc=basic.  Demangled with c++filt.\n\n")
    f.write("| kernel | VGPRs | spilled VGPRs | spilled SGPRs | scratch B/lane | occupancy (waves/SIMD) |\n|---|---:|---:|---:|---:|---:|\n")
    names = [r["name"] for r in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for r, d in zip(rows, dem):
        if "hpt_path_kernel" not in d:
            continue
        d = re.sub(r"^void hpt::", "", d).replace("(hpt::PathKernelArgs)", "")
        f.write("| `%s` | %s | %s | %s | %s | %s |\n" % (d, r.get("vgprs", ""), r.get("spill", ""), r.get("sspill", ""), r.get("scratch", ""), r.get("occ", "")))
print(open(out).read())
