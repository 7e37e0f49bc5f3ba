#!/usr/bin/env python3
"""Round 6, run Y: tests/test_gpu_parity.py::test_shipped_anim_moving_reflection_scene_end_to_end failed once (RMSE 4.9e-3) on the build that branched
per texture lookup.  The same comparison per tuning configuration, through the library (not pbrt_hip), with where the differences are."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
abi = importlib.import_module("pbrt-v2_amd.abi")
hpt = importlib.import_module("pbrt-v2_amd.hpt")
film = importlib.import_module("pbrt-v2_amd.film")
from oracle import orc  # noqa: E402  (the checker)

blob = sys.argv[1]
s = abi.Scene.load(blob)
rd = abi.copy_struct(s.render)
rd.sampler_mode, rd.seed = abi.HPT_SAMPLER_LD_HASH, 0
fo, so = orc.OracleScene(s).render(s.camera, rd)
want = film.xyzw_to_rgb(fo)
d = hpt.DeviceScene(s)
for cfg in ("", "3", "5", "6", "0"):
    if cfg:
        os.environ["HPT_TUNE"] = cfg
    f, st = d.render(s.camera, rd)
    got = film.xyzw_to_rgb(f)
    diff = np.abs(got - want).max(axis=2)
    ys, xs = np.nonzero(diff > 1e-2)
    print("cfg %-2s -> ran %d: rmse %.3g, weights equal %s, bad %d, pixels off > 1e-2: %d, rows %s cols %s, max %.3g" % (
        cfg or "-", st.tune_cfg, film.rmse(got, want), np.array_equal(f[..., 3], fo[..., 3]), st.bad_samples, len(ys),
        (ys.min(), ys.max()) if len(ys) else None, (xs.min(), xs.max()) if len(xs) else None, float(diff.max())), flush=True)
