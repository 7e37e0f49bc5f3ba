#!/usr/bin/env python3
"""EVERY path kernel of the build, from its BINARY, on the CPU: for each hpt_path_kernel<...> in the code objects of pbrt-v2_amd/build/hpt_kernels_*.o the gfx950 interpreter of
tests/isaemu renders a crop of a fixture the kernel's template arguments fit (material set, instances, integrator, sampler) and compares the film with the oracle's.
profiles/r05_isaemu_all_kernels.txt.  The last GPU run of round 5 validated this build but for one kernel; this is the check of the library that ships.

    python scripts/isaemu_all_kernels.py <crop size> [unit ...]"""
import glob
import importlib
import os
import re
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import util   # noqa: E402
from tests.isaemu import gfx950 as g, run as R   # noqa: E402
from oracle import orc   # noqa: E402  (the checker)
film = importlib.import_module("pbrt-v2_amd.film")
abi = importlib.import_module("pbrt-v2_amd.abi")


def template_args(sym):
    m = re.match(r"^_ZN3hpt15hpt_path_kernelILb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EEEvNS_14PathKernelArgsE$", sym)
    if not m:
        return None
    v = [int(x) for x in m.groups()]
    return dict(count=v[0], inst=v[1], mats=v[2], waves=v[3], ee=v[4], phased=v[5], dl=v[6], steal=v[7], win=v[8], top=v[9])


def fixture_for(t):
    """a golden fixture whose scene / integrator / sampler the instantiation can render"""
    ext = t["mats"] in (31, 61)
    if t["win"]:
        return ("hdl" if t["dl"] else ("hanim" if t["inst"] and not ext else "hk"))
    if t["dl"]:
        if t["inst"]:
            return "aquaddl" if t["mats"] == 31 else "dlanim"
        return "specdl" if t["mats"] == 31 else "dlb" if t["mats"] == 3 else "dl1"
    if t["inst"]:
        return "aquad" if t["mats"] == 31 else "anim"
    return {1: "cfg1", 3: "b8", 15: "metal", 31: "tex", 61: "metal"}[t["mats"]]


def main():
    n = int(sys.argv[1])
    units = sys.argv[2:] or [os.path.basename(p)[len("hpt_kernels_"):-2] for p in sorted(glob.glob(os.path.join(ROOT, "pbrt-v2_amd", "build", "hpt_kernels_*.o")))]
    print("# unit, hpt_path_kernel<COUNT, INST, MATS, WAVES, EE, PHASED, DL, STEAL, WIN, TOP>, fixture, wave-instructions, seconds, RGB rmse against the oracle, pixels off by > 1e-2, camera samples / oracle's, bad samples")
    bad = total = 0
    cache = {}
    for unit in units:
        co = R.code_object(unit)
        syms = subprocess.run([g.READELF, "-sW", co], check=True, capture_output=True, text=True).stdout
        kernels = sorted(set(f.split()[-1] for f in syms.splitlines() if " FUNC " in f and "hpt_path_kernel" in f))
        for sym in kernels:
            t = template_args(sym)
            if t is None:
                continue
            name = fixture_for(t)
            if name not in cache:
                s = util.load_case(name)
                rd = abi.copy_struct(s.render)
                rd.seed = 3
                if rd.sampler_mode == abi.HPT_SAMPLER_MT_REPLAY:
                    rd.sampler_mode = abi.HPT_SAMPLER_LD_HASH
                windowed = abi.sampler_kind(rd.sampler_mode) == abi.HPT_SAMPLER_HALTON_HASH
                m = 32 if windowed else n
                al = 32 if windowed else 1
                rd.x_start = (rd.x_start + (rd.x_count - m) // 2) // al * al; rd.y_start = (rd.y_start + (rd.y_count - m) // 2) // al * al
                rd.x_count = rd.y_count = m
                fo, so = orc.OracleScene(s).render(s.camera, rd)
                cache[name] = (s, rd, fo, so)
            s, rd, fo, so = cache[name]
            flags = (1 if (t["steal"]) else 0) | (2 if t["dl"] else 0) | (4 if t["top"] else 0) | (8 if t["win"] else 0) | (16 if t["mats"] == 3 else 0) | (32 if t["inst"] else 0) | (64 if t["count"] else 0) \
                | (128 if t["mats"] == 31 else 0)
            label = "<%d, %d, %2d, %d, %2d, %d, %d, %d, %d, %d>" % (t["count"], t["inst"], t["mats"], t["waves"], t["ee"], t["phased"], t["dl"], t["steal"], t["win"], t["top"])
            t0 = time.time()
            total += 1
            try:
                f, info = R.BinaryRender(s, co, sym, -flags).render(s.camera, rd)
            except g.EmuError as e:
                if "tree too deep" in str(e) or "no LDS rows" in str(e):      # (kernel_residency's refusal: the device falls back to another configuration on such a scene)
                    total -= 1
                    print("%-10s %s %-7s not launched: %s" % (unit, label, name, str(e)[:200])); sys.stdout.flush(); continue
                bad += 1
                print("%-10s %s %-7s ERROR %s" % (unit, label, name, str(e)[:240])); sys.stdout.flush(); continue
            a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
            rmse, off = float(film.rmse(a, b)), int((np.abs(a - b).max(axis=2) > 1e-2).sum())
            ok = rmse < (5e-4 if t["win"] else 2e-4) and off == 0 and info["bad"] == 0 and info["samples"] == int(so[0]) and np.array_equal(f[..., 3], fo[..., 3])
            bad += 0 if ok else 1
            print("%-10s %s %-7s %9d instructions %6.1fs rmse %.3g, %d px off, samples %d/%d, bad %d%s" % (unit, label, name, info["instructions"], time.time() - t0, rmse, off, info["samples"], int(so[0]), info["bad"],
                  "" if ok else "   <-- WRONG"))
            sys.stdout.flush()
    print("isaemu: %d of %d kernels differ from the oracle" % (bad, total))


if __name__ == "__main__":
    main()
