#!/bin/bash
# Round 6, GPU call Y — ABI 9 (spherical / cylindrical / planar texture mappings): the GPU suite, smoke, the textured workloads against the library of the commit
# before (build/variants/head_tree: the WHOLE library from HEAD's sources with HEAD's bench.py and Python package), then every workload and the driver's command
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06y; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
# (the end-to-end test that failed once in run Y2: the same frame through the library, configuration by configuration)
mkdir -p /tmp/amr && cp oracle/_ref/scenes/anim-moving-reflection.pbrt /tmp/amr/ && ln -sfn $ROOT/oracle/_ref/scenes/textures /tmp/amr/textures
( cd /tmp/amr && PBRT_RENDERER_HIP=1 HPT_DUMP_SCENE=/tmp/amr/scene.hpts HPT_HOST_BVH=1 $ROOT/pbrt-v2_amd/host/_build/pbrt_hip --quiet --outfile /tmp/amr/unused.pfm anim-moving-reflection.pbrt ) > $O/amr_debug.txt 2>&1
timeout 900 python scripts/gpu_r06_y_debug.py /tmp/amr/scene.hpts >> $O/amr_debug.txt 2>&1; tail -6 $O/amr_debug.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" >> $O/smoke.txt; tail -3 $O/smoke.txt
# (the old library reads 80-byte texture records: it runs from its own checkout — build/variants/head_tree = HEAD's bench.py, Python package and libhpt.so + the two fixtures)
for i in 1 2; do for v in head default; do
  D=$ROOT; [ $v = head ] && D=$V/head_tree
  for w in metal killeroo; do
    ( cd $D && timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null ) | line "$w $v" | tee -a $O/ab.txt
  done
done; done
for w in anim bunny soup; do timeout 900 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w default" | tee -a $O/ab.txt; done
( time timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.out 2> $O/bench_driver_cmd.err ) 2> $O/bench_driver_cmd.time; tail -c 400 $O/bench_driver_cmd.out; tail -3 $O/bench_driver_cmd.time
cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null

