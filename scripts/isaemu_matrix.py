#!/usr/bin/env python3
"""scripts/gpu_matrix.py on the CPU, on the BINARIES: the instantiations of the path kernel an instanced scene can run — configuration 3 / 5 / 6 (what the library ships since round 6), serial visit / top-level walk,
production / instrumented, direct lighting — taken out of the code objects the build ships (pbrt-v2_amd/build/hpt_kernels_*.o), executed by the gfx950 interpreter of tests/isaemu
over a crop of each fixture, against the oracle.  profiles/r05_isaemu_matrix.txt.

    python scripts/isaemu_matrix.py <crop size> [case ...]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import hash_rd, load_case, with_instance_copies   # noqa: E402
from tests.isaemu import gfx950 as g, run as R   # noqa: E402
from tests.wavemu import emu as w   # noqa: E402
from oracle import orc   # noqa: E402  (the checker)
film = importlib.import_module("pbrt-v2_amd.film")
abi = importlib.import_module("pbrt-v2_amd.abi")

K = R.kernel_symbol
# (label, unit, symbol, the tests/wavemu kernel id that prepares the same launch)
PATH_KERNELS = [("cfg 3", "ext_i", K(False, True, 31, 4, 0, True, False, False), w.K_LOCKSTEP), ("cfg 5", "ext_i", K(False, True, 31, 4, 0, True, False, True), w.K_STEAL),
                ("cfg 6", "ext_i", K(False, True, 31, 3, 0, True, False, True), w.K_STEAL), ("count", "ext_i", K(True, True, 31, 4, 0, True, False, True), w.K_STEAL_COUNT)]
TOP_KERNELS = [("cfg 5 top", "ext_i", K(False, True, 31, 4, 0, True, False, True, False, True), w.K_STEAL_TOP), ("cfg 6 top", "ext_i", K(False, True, 31, 3, 0, True, False, True, False, True), w.K_STEAL_TOP)]
DL_KERNELS = [("dl", "ext_i", K(False, True, 31, 3, 0, True, True, True), w.K_DL)]
DL_TOP = [("dl top", "ext_i", K(False, True, 31, 3, 0, True, True, True, False, True), w.K_DL_TOP)]


def main():
    n = int(sys.argv[1])
    cases = sys.argv[2:] or ["aquad", "oinst", "oinst64", "abi8dl", "aquaddl", "anim", "tex"]
    print("# fixture, kernel (unit), wave-instructions executed, seconds, RGB rmse against the oracle, pixels off by > 1e-2, weights equal, camera samples completed / in the job, bad samples")
    bad = total = 0
    for name in cases:
        s = with_instance_copies(load_case("oinst"), 2, 58, start=(-40.0, 0.0, -30.0), step=(-0.9, 0.0, -0.7)) if name == "oinst64" else load_case(name)
        rd = hash_rd(s, seed=3)
        rd.x_start += (rd.x_count - n) // 2; rd.y_start += (rd.y_count - n) // 2; rd.x_count = rd.y_count = n
        fo, so = orc.OracleScene(s).render(s.camera, rd)
        b = film.xyzw_to_rgb(fo)
        inst = len(s.instances) > 0
        ks = (PATH_KERNELS + (TOP_KERNELS if inst else [])) if rd.integrator == abi.HPT_INTEGRATOR_PATH else (DL_KERNELS + (DL_TOP if inst else []))
        for label, unit, sym, kid in ks:
            t = time.time()
            total += 1
            try:
                f, info = R.BinaryRender(s, R.code_object_for(os.environ.get("ISAEMU_UNIT_" + unit.upper(), unit), sym), sym, kid).render(s.camera, rd)      # (ISAEMU_UNIT_EXT_I=<object>: another build of the unit)
            except g.EmuError as e:
                bad += 1
                print("%-8s %-10s ERROR %s" % (name, label, str(e)[:300])); sys.stdout.flush(); continue
            a = film.xyzw_to_rgb(f)
            rmse, off = float(film.rmse(a, b)), int((np.abs(a - b).max(axis=2) > 1e-2).sum())
            ok = rmse < 1e-4 and off == 0 and info["bad"] == 0 and info["samples"] == int(so[0]) and np.array_equal(f[..., 3], fo[..., 3])
            bad += 0 if ok else 1
            print("%-8s %-10s (%s) %9d instructions %6.1fs rmse %.3g, %d px off, weights %s, samples %d/%d, bad %d%s" % (name, label, unit, info["instructions"], time.time() - t, rmse, off,
                  bool(np.array_equal(f[..., 3], fo[..., 3])), info["samples"], int(so[0]), info["bad"], "" if ok else "   <-- WRONG"))
            sys.stdout.flush()
    print("isaemu matrix: %d of %d combinations wrong" % (bad, total))


if __name__ == "__main__":
    main()
