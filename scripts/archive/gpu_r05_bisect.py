#!/usr/bin/env python3
"""Round 5: the wrong films that 'move with the code generator' (VERDICT r04 item 1 ii), reproduced — which RUN-TIME path do they need?
For every (case, configuration, top) given, render against the oracle once as configured and once per run-time switch (no rebuild: the same
binary, one code path taken out at a time), and say what the wrong pixels look like.
usage: HPT_LIB=... gpu_r05_bisect.py case:cfg:top [...]"""
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

SWITCHES = [{}, {"HPT_NO_XF_CACHE": "1"}, {"HPT_REGEN_MIN": "1"}, {"HPT_RETRACE_MIN": "65"}, {"HPT_RETRACE_MAX": "1"}, {"HPT_LEAF_Q": "0", "HPT_LEAF_BLOCK_Q": "0"},
            {"HPT_XCD_QUEUE": "0"}, {"HPT_CHUNK": "1"}, {"HPT_CHUNK": "64"}, {"HPT_BVH4_CAP": "6"}, {"HPT_TOP": "1"}]


def describe(f, fo):
    a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
    d = np.abs(a - b)
    wrong = d.max(axis=2) > 1e-2
    n = int(wrong.sum())
    out = "rmse %.3g, %d px wrong" % (float(film.rmse(a, b)), n)
    if n:
        aw, bw = a[wrong], b[wrong]
        for c, nm in enumerate("RGB"):
            out += ", %s: dev==0 in %d, dev NaN %d, |d|>1e-2 in %d" % (nm, int((aw[:, c] == 0).sum()), int(np.isnan(aw[:, c]).sum()), int((d[wrong][:, c] > 1e-2).sum()))
        ys, xs = np.nonzero(wrong)
        out += "; rows %d..%d cols %d..%d; first %s dev %s orc %s" % (ys.min(), ys.max(), xs.min(), xs.max(), (int(ys[0]), int(xs[0])), aw[0], bw[0])
        # does the set of wrong pixels follow the 8x8 micro-tiles (= the 64 lanes of a wave at the first bounce)?
        t = np.zeros(((f.shape[0] + 7) // 8, (f.shape[1] + 7) // 8), int)
        np.add.at(t, (ys // 8, xs // 8), 1)
        out += "; 8x8 tiles touched %d, full %d" % (int((t > 0).sum()), int((t == 64).sum()))
    return out


def main():
    for spec in sys.argv[1:]:
        name, cfg, top = spec.split(":")
        s = with_instance_copies(load_case("oinst"), 2, 58, start=(-40.0, 0.0, -30.0), step=(-0.9, 0.0, -0.7)) if name == "oinst64" else load_case(name)
        rd = hash_rd(s, seed=3)
        fo, _ = orc.OracleScene(s).render(s.camera, rd)
        for sw in SWITCHES:
            env = dict(sw)
            env.setdefault("HPT_TOP", top)
            env["HPT_TUNE"] = cfg
            for k, v in env.items():
                os.environ[k] = v
            try:
                f, st = hpt.DeviceScene(s).render(s.camera, rd)
                print("%-8s cfg %s top %s %-44s: %s, bad %d, weights equal %s" % (name, cfg, env["HPT_TOP"], sw or "as configured", describe(f, fo), st.bad_samples, bool(np.array_equal(f[..., 3], fo[..., 3]))))
            except hpt.HptError as e:
                print("%-8s cfg %s top %s %-44s: ERROR %s" % (name, cfg, top, sw, e))
            sys.stdout.flush()
            for k in env:
                os.environ.pop(k, None)


if __name__ == "__main__":
    main()
