#!/usr/bin/env python3
"""Round 5: what a cross-product mode of the triangle test (hpt_device.h, HPT_CROSS_MODE) does to the hits and to the films.
200 k aggregate-test rays per case against the oracle (same primitive? t / b1 / b2 how many ulps apart?), then the small renders against the oracle.
usage: HPT_LIB=build/variants/libhpt_x1.so gpu_r05_cross.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.util import hash_rd, load_case, random_rays   # noqa: E402
hpt = importlib.import_module("pbrt-v2_amd.hpt")
film = importlib.import_module("pbrt-v2_amd.film")
from oracle import orc   # noqa: E402  (the checker)


def ulps(a, b):
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


for name in ("cfg1", "b8", "env", "anim"):
    s = load_case(name)
    rays = random_rays(s, 200000, seed=5)
    ho, po = orc.OracleScene(s).intersect(rays)
    d = hpt.DeviceScene(s)
    hd, pd = d.intersect(rays)
    same = (po == pd) & (po >= 0)
    u = [ulps(ho[same, k], hd[same, k]) for k in range(3)]
    diff = (po != pd)
    both = diff & (po >= 0) & (pd >= 0)
    print("%-5s same prim %.5f of %d rays; where the same: t identical %.4f, max ulps t %d b1 %d b2 %d; prim differs on %d rays (%d hit on both sides, max ulps of t there %s)"
          % (name, float((po == pd).mean()), len(po), float((u[0] == 0).mean()), int(u[0].max()), int(u[1].max()), int(u[2].max()), int(diff.sum()), int(both.sum()),
             int(ulps(ho[both, 0], hd[both, 0]).max()) if both.any() else "-"))
    rd = hash_rd(s, seed=3)
    fo, _ = orc.OracleScene(s).render(s.camera, rd)
    f, st = d.render(s.camera, rd)
    print("      render rmse vs oracle %.3g, weights equal %s, bad %d" % (float(film.rmse(film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo))), bool(np.array_equal(f[..., 3], fo[..., 3])), st.bad_samples))
    sys.stdout.flush()
