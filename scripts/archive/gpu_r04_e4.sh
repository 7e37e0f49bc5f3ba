#!/bin/bash
# Round 4, run E4 (debug): what is wrong in the wrong films — aquad at configuration 5 (serial visit), oinst at 5 (serial) and at 6 (top-level walk)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_e4; mkdir -p $O
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
s = load_case(sys.argv[1]); rd = hash_rd(s, seed=3); rd.spp = 1
for md in (1, 2, 3):
    rd.maxdepth = md
    fo, so = orc.OracleScene(s).render(s.camera, rd)
    d = hpt.DeviceScene(s)
    f, st = d.render(s.camera, rd)
    a, b = film.xyzw_to_rgb(f), film.xyzw_to_rgb(fo)
    bad = np.abs(a - b).max(axis=2) > 1e-3
    print(sys.argv[1], "maxdepth", md, "cfg", st.tune_cfg, "bad px", int(bad.sum()), "of", bad.size, "mean dev/orc on bad", float(a[bad].mean()), float(b[bad].mean()), "dev>orc", int((a[bad].sum(axis=-1) > b[bad].sum(axis=-1)).sum()), "badsamples", st.bad_samples)
    if md == 1 and bad.sum():
        ys, xs = np.nonzero(bad)
        for k in range(0, min(len(ys), 4000), 800):
            print("   px", int(ys[k]), int(xs[k]), "dev", a[ys[k], xs[k]].tolist(), "orc", b[ys[k], xs[k]].tolist())
'''
for c, cfg, top in (("aquad", "5", "0"), ("oinst", "5", "0"), ("oinst", "6", "1")):
    p = subprocess.run([sys.executable, "-c", code, c], env=dict(os.environ, HPT_TUNE=cfg, HPT_TOP=top), capture_output=True, timeout=300)
    print(p.stdout.decode()[-3000:], p.stderr.decode()[-300:])
    sys.stdout.flush()
PY
cat $O/dbg.txt | cut -c1-330
