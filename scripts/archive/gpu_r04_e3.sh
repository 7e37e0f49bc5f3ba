#!/bin/bash
# Round 4, run E3 (debug): three rebuilds of the instanced extension-set kernels (quadric test in the leaf loop / regeneration batching compiled out /
# max-ILP scheduling) on the two wrong films of run E2: aquad at configuration 5 (serial instance visit), oinst at 6 under the top-level walk
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_e3; mkdir -p $O
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
print(sys.argv[1], "cfg", st.tune_cfg, "rmse %.3g" % float(film.rmse(a, b)), "bad px", len(bad), "badsamples", st.bad_samples)
'''
libs = [None, "nq", "nr", "ilp"]
for lib in libs:
    for c in ("oinst", "aquad"):
        for cfg in ("5", "6", "0"):
            for top in ("0", "1"):
                e = dict(os.environ, HPT_TUNE=cfg, HPT_TOP=top)
                if lib: e["HPT_LIB"] = os.path.abspath("pbrt-v2_amd/build/variants/libhpt_%s.so" % lib)
                p = subprocess.run([sys.executable, "-c", code, c], env=e, capture_output=True, timeout=120)
                print(lib, c, "cfg", cfg, "top", top, "rc", p.returncode, p.stdout.decode()[-120:].strip(), p.stderr.decode()[-60:].strip().replace("\n", " | "))
                sys.stdout.flush()
PY
cat $O/dbg.txt | cut -c1-200
