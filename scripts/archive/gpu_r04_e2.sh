#!/bin/bash
# Round 4, run E2 (debug): oinst / 64 instances at configuration 6 under the top-level walk render garbage — which run-time switch changes that?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_e2; mkdir -p $O
python - > $O/dbg.txt 2>&1 <<'PY'
import os, subprocess, sys
code = '''
import os, sys, importlib
sys.path.insert(0, ".")
import numpy as np
from tests.util import load_case, hash_rd
hpt = importlib.import_module("pbrt-v2_amd.hpt")
film = importlib.import_module("pbrt-v2_amd.film")
from oracle import orc
s = load_case(sys.argv[1]); rd = hash_rd(s, seed=3)
fo, so = orc.OracleScene(s).render(s.camera, rd)
d = hpt.DeviceScene(s)
f, st = d.render(s.camera, rd)
a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
bad = np.argwhere(np.abs(a - b).max(axis=2) > 1e-2)
print(sys.argv[1], "cfg", st.tune_cfg, "rmse", float(film.rmse(a, b)), "bad px", len(bad), bad[:6].tolist(), "max", float(np.abs(a).max()), "w eq", bool(np.array_equal(f[...,3], fo[...,3])), "badsamples", st.bad_samples)
'''
envs = [{"HPT_TUNE": "6"}, {"HPT_TUNE": "6", "HPT_TOP": "0"}, {"HPT_TUNE": "5"}, {"HPT_TUNE": "6", "HPT_NO_XF_CACHE": "1"}, {"HPT_TUNE": "6", "HPT_REGEN_MIN": "1"},
        {"HPT_TUNE": "6", "HPT_RETRACE_MAX": "0"}, {"HPT_TUNE": "6", "HPT_LEAF_Q": "0", "HPT_LEAF_BLOCK_Q": "0"}, {"HPT_TUNE": "6", "HPT_BVH4_CAP": "0"},
        {"HPT_TUNE": "6", "HPT_QUADRIC_LINEAR": "1"}, {"HPT_TUNE": "0"}, {"HPT_TUNE": "0", "HPT_QUADRIC_LINEAR": "1"}, {"HPT_TUNE": "0", "HPT_NO_XF_CACHE": "1"}, {"HPT_TUNE": "3"}]
for c in ("oinst", "aquad"):
    for e in envs:
        p = subprocess.run([sys.executable, "-c", code, c], env=dict(os.environ, **e), capture_output=True, timeout=120)
        print(c, e, "rc", p.returncode, p.stdout.decode()[-300:].strip(), p.stderr.decode()[-80:].strip().replace("\n", " | "))
        sys.stdout.flush()
PY
cat $O/dbg.txt | cut -c1-420
