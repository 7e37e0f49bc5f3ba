#!/usr/bin/env python3
"""GPU debug: BVH4 build + animated instances + the instrumented (COUNT) kernel (run C: 3 parity tests failed on `anim` under count_work)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import load_case, hash_rd, abi
film = importlib.import_module("pbrt-v2_amd.film")
hpt = importlib.import_module("pbrt-v2_amd.hpt")
from oracle import orc
for name, seed in (("anim", 5), ("anim", 7), ("anim", 11), ("cfg1", 5)):
    s = load_case(name)
    rd = hash_rd(s, seed=seed)
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    for env in ({}, {"HPT_LEAF_Q": "0", "HPT_LEAF_BLOCK_Q": "0"}, {"HPT_RETRACE_MIN": "65"}, {"HPT_NO_XF_CACHE": "1"}, {"HPT_BVH4_CAP": "0"}, {"HPT_BVH4_CAP": "8"}):
        for k, v in env.items():
            os.environ[k] = v
        for count in (0, 1):
            for cfg in ("5", "6"):
                os.environ["HPT_TUNE"] = cfg
                rd.count_work = count
                f, st = hpt.DeviceScene(s).render(s.camera, rd)
                d = np.abs(film.xyzw_to_rgb(f) - film.xyzw_to_rgb(fo)).sum(axis=2)
                worst = np.argsort(d.ravel())[::-1][:3]
                print(name, seed, env, "worst pixels", [(int(i % d.shape[1]), int(i // d.shape[1]), float(d.ravel()[i])) for i in worst])
                print(name, env, "count", count, "cfg", st.tune_cfg, "rmse %.3g" % film.rmse(film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)),
                      "weights equal", np.array_equal(f[..., 3], fo[..., 3]), "rays", int(st.closest_rays), int(so[1]), int(st.shadow_rays), int(so[2]), flush=True)
        for k in env:
            del os.environ[k]
