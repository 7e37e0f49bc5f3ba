#!/usr/bin/env python3
"""EVERY golden fixture of tests/golden, whole frame, through the path kernel SOURCE on the CPU scheduler of tests/wavemu (debug checks armed), against the oracle:
the kernels hpt_render_device would launch for the fixture (free-running and lock step + stealing; the walk from the top-level tree where there are instances;
the direct-lighting and window-sampler instantiations), the film's filter one-pass and two-pass.  profiles/r05_wavemu_all_fixtures.txt.

    python scripts/wavemu_all.py [case ...]
"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util   # noqa: E402
from tests.wavemu import emu as w   # noqa: E402
from oracle import orc   # noqa: E402  (the checker)
film = importlib.import_module("pbrt-v2_amd.film")
abi = importlib.import_module("pbrt-v2_amd.abi")


def main():
    names = sys.argv[1:] or (util.CASES + ["envmap", "envmap_dl", "acam"] + list(util.DL_CASES) + list(util.FILTER_CASES) + list(util.COMBO_CASES) + list(util.RANDOM_CASES)
                             + list(util.STRATIFIED_CASES) + list(util.HALTON_CASES) + list(util.ADAPTIVE_CASES) + list(util.BESTCANDIDATE_CASES)
                             + [n for n in util.R2_CASES if n != "merl"] + list(util.R2_VIEW_CASES))
    print("# fixture, kernel id (tests/wavemu/emu.py), film pass, seconds, RGB rmse / max difference against the oracle, pixels whose weights differ, camera samples completed / oracle's, bad samples, rendezvous")
    bad = total = 0
    for name in names:
        s = util.load_case(name)
        rd = abi.copy_struct(s.render)
        rd.seed = 3
        flt = getattr(s, "filter", None)
        cm = getattr(s, "camera_motion", None)
        kind = abi.sampler_kind(rd.sampler_mode)
        tbl = util.sample_table() if kind == abi.HPT_SAMPLER_BESTCANDIDATE_HASH else None
        if rd.sampler_mode == abi.HPT_SAMPLER_MT_REPLAY:
            rd.sampler_mode = abi.HPT_SAMPLER_LD_HASH
        fo, so = orc.OracleScene(s).render(s.camera, rd, flt=flt, cam_motion=cm, sample_table=tbl)
        b = film.xyzw_to_rgb(fo)
        ws = w.WaveScene(s)
        dl = rd.integrator != abi.HPT_INTEGRATOR_PATH
        inst = len(s.instances) > 0
        windowed = kind in (abi.HPT_SAMPLER_HALTON_HASH, abi.HPT_SAMPLER_ADAPTIVE_HASH, abi.HPT_SAMPLER_BESTCANDIDATE_HASH)
        if windowed:
            ks = [w.K_DL_WIN if dl else w.K_WIN]
        elif dl:
            ks = [w.K_DL] + ([w.K_DL_TOP] if inst else [])
        else:
            ks = [w.K_FREE, w.K_STEAL] + ([w.K_STEAL_TOP] if inst else [])
        adaptive = kind == abi.HPT_SAMPLER_ADAPTIVE_HASH
        for k in ks:
            for two_pass in ((False, True) if (flt is not None and not windowed) else (False,)):
                t = time.time()
                total += 1
                try:
                    f, info = ws.render(s.camera, rd, k, flt=flt, two_pass=two_pass, cam_motion=cm, sample_table=tbl)
                except w.WaveEmuError as e:
                    bad += 1
                    print("%-9s kernel %2d ERROR %s" % (name, k, e)); sys.stdout.flush(); continue
                a = film.xyzw_to_rgb(f)
                rmse = float(film.rmse(a, b))
                wdiff = int((f[..., 3] != fo[..., 3]).sum()) if flt is None else int((np.abs(f[..., 3] - fo[..., 3]) > 1e-5 * np.abs(fo[..., 3]).max()).sum())
                ok = rmse < (2e-3 if adaptive else 1e-4) and (wdiff == 0 or (adaptive and wdiff < 2e-3 * f[..., 3].size)) and info["bad"] == int(so[5]) \
                    and (info["samples"] == int(so[0]) or (adaptive and abs(info["samples"] - int(so[0])) <= 2e-3 * int(so[0])))
                bad += 0 if ok else 1
                print("%-9s kernel %2d %-8s %5.1fs rmse %.3g maxdiff %.3g weights differ %d samples %d/%d bad %d rendezvous %d%s" % (
                    name, k, "two-pass" if two_pass else "one-pass", time.time() - t, rmse, float(np.abs(a - b).max()), wdiff, info["samples"], int(so[0]), info["bad"], info["rendezvous"],
                    "" if ok else "   <-- DIFFERS"))
                sys.stdout.flush()
    print("wavemu: %d of %d renders differ from the oracle" % (bad, total))


if __name__ == "__main__":
    main()
