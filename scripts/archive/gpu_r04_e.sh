#!/bin/bash
# Round 4, run E: the build with both instance walks (TOP template flag; serial visit up to four instances): the instanced cases under HPT_TOP=0 / 1,
# production and instrumented, each in a process of its own (run D: the instrumented extension-set kernel faulted on aquad / oinst); then the suite; then
# the workloads' kernel times, anim also with HPT_TOP=1.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_e; mkdir -p $O
python - > $O/dbg.txt 2>&1 <<'PY'
import os, subprocess, sys
code = '''
import os, sys, importlib
sys.path.insert(0, ".")
from tests.util import load_case, hash_rd
hpt = importlib.import_module("pbrt-v2_amd.hpt")
s = load_case(sys.argv[1]); rd = hash_rd(s, seed=3); rd.count_work = int(sys.argv[2])
d = hpt.DeviceScene(s)
f, st = d.render(s.camera, rd)
print(sys.argv[1], "count", sys.argv[2], "ok cfg", st.tune_cfg, round(st.kernel_ms, 3), st.camera_samples, st.closest_rays, st.shadow_rays)
'''
for c in ("aquad", "oinst", "anim", "abi8dl", "aquaddl"):
    for top in ("0", "1"):
        for cw in ("1", "0"):
            p = subprocess.run([sys.executable, "-c", code, c, cw], env=dict(os.environ, HPT_TOP=top), capture_output=True, timeout=120)
            print(c, "top", top, "count", cw, "rc", p.returncode, p.stdout.decode()[-100:].strip(), p.stderr.decode()[-100:].strip().replace("\n", " | "))
            sys.stdout.flush()
PY
cat $O/dbg.txt | cut -c1-260
timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|Abort" $O/pytest_gpu.txt | tail -8
timeout 900 python scripts/ab_knobs.py --workloads killeroo,anim,bunny,soup,metal,killeroo-dl --knob HPT_TOP --values 0,1 --frames 3 > $O/ab.jsonl 2> $O/ab.err; cat $O/ab.jsonl; tail -2 $O/ab.err
HPT_QUADRIC_LINEAR=1 timeout 600 python scripts/ab_knobs.py --workloads killeroo,bunny,anim --knob HPT_TOP --values 0 --frames 3 > $O/ab_linear.jsonl 2> $O/ab_linear.err; cat $O/ab_linear.jsonl
