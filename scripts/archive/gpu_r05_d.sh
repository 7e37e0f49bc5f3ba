#!/bin/bash
# Round 5, GPU call D: the production library of call A (kept as libhpt_prodfail.so: aquad / oinst wrong at configuration 6, serial visit) under ROCr's scratch-memory
# switches — is the spill memory of these 1-1.7 KB/lane kernels taken away or shared while they run? — and twice in one process (is the wrong film the same film?).
O=gpurun_out/r05d; mkdir -p $O
L=$PWD/pbrt-v2_amd/build/variants/libhpt_prodfail.so
run() { echo "== $*"; env "$@" HPT_LIB=$L timeout 300 python scripts/gpu_matrix.py aquad oinst 2>&1 | tail -4 | cut -c1-200; }
{
run X=1
run HSA_NO_SCRATCH_RECLAIM=1
run HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0
run HSA_SCRATCH_SINGLE_LIMIT=4294967295
run AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3
run GPU_MAX_HW_QUEUES=1
run HSA_ENABLE_SDMA=0
} > $O/scratch_env.txt 2>&1
cat $O/scratch_env.txt
HPT_LIB=$L timeout 300 python - > $O/determinism.txt 2>&1 <<'PY'
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from tests.util import hash_rd, load_case
hpt = importlib.import_module("pbrt-v2_amd.hpt")
os.environ["HPT_TUNE"] = "6"; os.environ["HPT_TOP"] = "0"
for name in ("aquad", "oinst"):
    s = load_case(name); rd = hash_rd(s, seed=3)
    d = hpt.DeviceScene(s)
    films = [d.render(s.camera, rd)[0] for _ in range(4)]
    d2 = hpt.DeviceScene(s)
    films.append(d2.render(s.camera, rd)[0])
    for i, f in enumerate(films[1:], 1):
        df = np.abs(f - films[0])
        print(name, "render", i, "vs render 0: identical" if np.array_equal(f, films[0]) else "differs in %d pixels, max |d| %.3g" % (int((df.max(axis=2) > 0).sum()), float(df.max())))
PY
cat $O/determinism.txt
