#!/usr/bin/env python3
"""Every instantiation of the path kernel an instanced scene can run, against the oracle: configuration 0 / 5 / 6 x serial visit / top-level walk
(HPT_TOP) x production / instrumented (count_work) build, one line per combination that is WRONG and a summary.  Round 4: the instanced
extension-set kernels came out wrong in one instantiation or another after every change to the kernel source (profiles/r04_ab.md, runs B2-F);
this is the check a build of those kernels has to pass.

    python scripts/gpu_matrix.py [case ...]        (HPT_LIB selects a variant library)
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import hash_rd, load_case, with_instance_copies   # noqa: E402
hpt = importlib.import_module("pbrt-v2_amd.hpt")
abi = importlib.import_module("pbrt-v2_amd.abi")
film = importlib.import_module("pbrt-v2_amd.film")
from oracle import orc   # noqa: E402  (the checker)


def main():
    cases = sys.argv[1:] or ["aquad", "oinst", "oinst64", "abi8dl", "aquaddl", "anim", "tex", "b8"]
    bad = total = 0
    for name in cases:
        s = with_instance_copies(load_case("oinst"), 2, 58, start=(-40.0, 0.0, -30.0), step=(-0.9, 0.0, -0.7)) if name == "oinst64" else load_case(name)
        rd = hash_rd(s, seed=3)
        fo, so = orc.OracleScene(s).render(s.camera, rd)
        b = film.xyzw_to_rgb(fo)
        path = rd.integrator == abi.HPT_INTEGRATOR_PATH
        inst = len(s.instances) > 0
        for top in (("0", "1") if inst else ("0",)):
            for cw in (0, 1):
                for cfg in (("0", "5", "6") if (path and not cw) else ("-",)):
                    os.environ["HPT_TOP"] = top
                    if cfg != "-":
                        os.environ["HPT_TUNE"] = cfg
                    else:
                        os.environ.pop("HPT_TUNE", None)
                    r = abi.copy_struct(rd)
                    r.count_work = cw
                    try:
                        f, st = hpt.DeviceScene(s).render(s.camera, r)
                    except hpt.HptError as e:      # (HPT_E_INTERNAL: sample conservation, or a check of the debug build — the message names it)
                        total += 1; bad += 1
                        print("ERROR %-8s top %s count %d cfg %s: %s" % (name, top, cw, cfg, e)); sys.stdout.flush()
                        continue
                    a = film.xyzw_to_rgb(f)
                    rmse = float(film.rmse(a, b))
                    ok = rmse < 1e-3 and np.array_equal(f[..., 3], fo[..., 3]) and st.bad_samples == 0
                    total += 1
                    if not ok:
                        bad += 1
                        print("WRONG %-8s top %s count %d cfg %s (ran %d): rmse %.3g, %d px off by > 1e-2, %d bad samples, weights equal %s"
                              % (name, top, cw, cfg, st.tune_cfg, rmse, int((np.abs(a - b).max(axis=2) > 1e-2).sum()), st.bad_samples, bool(np.array_equal(f[..., 3], fo[..., 3]))))
                        sys.stdout.flush()
    os.environ.pop("HPT_TOP", None); os.environ.pop("HPT_TUNE", None)
    print("matrix: %d of %d combinations wrong (lib %s)" % (bad, total, os.environ.get("HPT_LIB", "libhpt.so")))


if __name__ == "__main__":
    main()
