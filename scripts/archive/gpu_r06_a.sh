#!/bin/bash
# Round 6, GPU call A — the pilot the verdict asked for, on the round-5 kernels + the first changes of the round (kd_step without its rare-lane blocks; the refill
# without 64-bit divisions; COOPERATIVE LEAVES: the parked leaves' triangles as (ray, triangle) pairs over all 64 lanes):
#  1. what a film atomic costs and how WRITE_SIZE counts it (scripts/calib/calib_atomic.hip)
#  2. GPU suite on the new default build
#  3. same-box A/B: default | base (no cooperative leaves) | nofilm (timing only: no film atomics) | leaf-phase thresholds of the default
#  4. wave clocks and lane-weighted clocks per loop section (-DHPT_PHASE_TIMERS=1: ptbase = before the cooperative leaves, pt = with them) on bunny, killeroo, anim
#  5. WRITE_SIZE / FETCH_SIZE of bunny and killeroo at HPT_CHUNK=1 (one atomic set per sample) against 64 (one per pixel)
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06a; mkdir -p $O
V=$ROOT/pbrt-v2_amd/build/variants
Q="--no-cpu-baseline --no-extra --no-pmc --no-work --no-verify"
line() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['kernel']['avg_ms'], d['kernel']['tune_cfg'], d['kernel']['vgprs'], d['kernel']['scratch_B'])"; }
bash scripts/calib/run_atomic.sh > $O/calib_atomic.txt 2>&1; tail -12 $O/calib_atomic.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.txt 2>&1; echo "pytest rc $?" >> $O/pytest_gpu.txt; tail -4 $O/pytest_gpu.txt | cut -c1-300
for i in 1 2; do
  for v in default base nofilm; do
    L=$V/libhpt_$v.so; [ $v = default ] && L=$ROOT/pbrt-v2_amd/libhpt.so
    [ -f $L ] || continue
    for w in bunny killeroo; do
      HPT_LIB=$L timeout 600 python bench.py --workload $w --steps 4 --warmup 2 $Q 2>/dev/null | line "$w $v" | tee -a $O/ab.txt
    done
  done
done
for lq in "1 8" "2 8" "2 2" "4 2" "8 8"; do set -- $lq
  for w in bunny killeroo; do
    HPT_LEAF_Q=$1 HPT_LEAF_BLOCK_Q=$2 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 $Q 2>/dev/null | line "$w default leaf_q=$1 block_q=$2" | tee -a $O/ab_leafq.txt
  done
done
for v in ptbase pt; do for w in bunny killeroo anim; do
  HPT_PHASE_TIMERS=1 HPT_LIB=$V/libhpt_$v.so timeout 600 python bench.py --workload $w --steps 1 --warmup 1 $Q > $O/${v}_$w.out 2> $O/${v}_$w.err
  echo "== $v $w"; grep "hpt phase\|hpt walk" $O/${v}_$w.err | tail -3
done; done 2>&1 | tee $O/pt.txt
cd /tmp
for w in bunny killeroo; do for c in 1 64; do for k in WRITE_SIZE FETCH_SIZE; do
  HPT_CHUNK=$c timeout 600 rocprofv3 --kernel-trace --pmc $k --output-format csv -d $O/pmc_${w}_chunk${c}_$k -o p -- python $ROOT/bench.py --workload $w --steps 1 --warmup 0 $Q > $O/pmc_${w}_chunk${c}_$k.log 2>&1
  python - <<PY
import csv, glob
best = {}
for f in glob.glob("$O/pmc_${w}_chunk${c}_$k/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hpt_path_kernel" in r["Kernel_Name"]:
            d = best.setdefault(r["Dispatch_Id"], [0.0, r["Kernel_Name"][:60]]); d[0] += float(r["Counter_Value"])
if best:
    v = max(best.values())
    print("$w chunk $c $k max-over-dispatches %.4g KiB = %.2f GB  (%d dispatches)" % (v[0], v[0] * 1024 / 1e9, len(best)))
PY
done; done; done 2>&1 | tee $O/pmc_chunk.txt
