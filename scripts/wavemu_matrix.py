#!/usr/bin/env python3
"""scripts/gpu_matrix.py without a GPU: the fixtures on which one instantiation or another of the instanced extension-set kernels came out wrong on the GPU
(rounds 4 / 5), rendered by the SAME kernel source on the CPU scheduler of tests/wavemu (64 fibers per wave, the debug build's checks armed) and compared with
the oracle.  The source is right where the binary was wrong: profiles/r05_wavemu_matrix.txt.

    python scripts/wavemu_matrix.py <crop size> [case ...]        (HPT_WAVEMU_SAN=1 + LD_PRELOAD of the sanitizer runtimes: scripts/wavemu_sanitize.sh)
"""
import os
sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import sys
sys.path.insert(0, sys_path_root)
import importlib
import time

import numpy as np
from tests.util import load_case, hash_rd, with_instance_copies
from tests.wavemu import emu as w
from oracle import orc
film = importlib.import_module("pbrt-v2_amd.film")
abi = importlib.import_module("pbrt-v2_amd.abi")
print("# the path kernel source on the CPU scheduler of tests/wavemu against the oracle: fixture, kernel id (tests/wavemu/emu.py), seconds, RGB rmse / max difference, film weights equal, camera samples completed / in the job, bad samples, LDS rows, rows for ordinary BVH4 entries, rendezvous (cross-lane operations) executed")
n = int(sys.argv[1]); cases = sys.argv[2:] or ["aquad", "oinst", "oinst64", "abi8dl", "aquaddl", "anim", "tex", "b8"]
for name in cases:
    s = with_instance_copies(load_case("oinst"), 2, 58, start=(-40.0, 0.0, -30.0), step=(-0.9, 0.0, -0.7)) if name == "oinst64" else load_case(name)
    rd = hash_rd(s, seed=3)
    if n < rd.x_count:
        rd.x_count = rd.y_count = n; rd.x_start = (rd.xres - n) // 2; rd.y_start = (rd.yres - n) // 2
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    b = film.xyzw_to_rgb(fo)
    ws = w.WaveScene(s)
    path = rd.integrator == abi.HPT_INTEGRATOR_PATH
    inst = len(s.instances) > 0
    ks = ([w.K_FREE, w.K_LOCKSTEP, w.K_STEAL, w.K_STEAL_COUNT] + ([w.K_STEAL_TOP] if inst else [])) if path else ([w.K_DL] + ([w.K_DL_TOP] if inst else []))
    for k in ks:
        t = time.time()
        try:
            f, info = ws.render(s.camera, rd, k)
        except w.WaveEmuError as e:
            print("%-8s kernel %2d ERROR %s" % (name, k, e)); sys.stdout.flush(); continue
        a = film.xyzw_to_rgb(f)
        print("%-8s kernel %2d %5.1fs rmse %.3g maxdiff %.3g weights %s samples %d/%d bad %d rows %d cap %d rendezvous %d" % (name, k, time.time() - t, float(film.rmse(a, b)), float(np.abs(a - b).max()),
              bool(np.array_equal(f[..., 3], fo[..., 3])), info["samples"], int(so[0]), info["bad"], info["lds_rows"], info["cap_normal"], info["rendezvous"]))
        sys.stdout.flush()
