#!/bin/bash
# Round 3, run H: where the wall time of `pbrt_hip bunny.pbrt` goes (HPT_TIMING stage lines), host tree build forked / serial.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r03_h; mkdir -p $O
python - > $O/e2e.txt 2>&1 <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import bench
ROOT = os.getcwd()
exe = os.path.join(ROOT, "pbrt-v2_amd", "host", "_build", "pbrt_hip")
for wl, spp in (("bunny", 64), ("killeroo", 64)):
    with tempfile.TemporaryDirectory() as tmp:
        sf = os.path.join(tmp, "s.pbrt")
        open(sf, "w").write(bench.ref_scene_text(wl, 1920, 1080, spp, 8, os.path.join(tmp, "o.pfm"), renderer="hip"))
        for env in ({}, {}, {}, {"HPT_NO_WARMUP": "1"}, {"HPT_NO_PRELOAD": "1"}, {}, {}, {"HPT_NO_WARMUP": "1", "HPT_NO_PRELOAD": "1"}, {}):
            t = time.time()
            p = subprocess.run([exe, "--quiet", sf], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, HPT_TIMING="1", **env))
            print(wl, env, "wall %.3f s" % (time.time() - t))
            print("\n".join(l for l in p.stderr.decode().splitlines() if l.startswith("hpt")))
PY
tail -60 $O/e2e.txt
