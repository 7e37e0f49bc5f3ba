#!/bin/bash
# Round 2, GPU call F: suite; bench default; filtered bunny (new gather kernel vs broadcast kernel); soup with device / host BVH
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=gpurun_out/r02f; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu.log | tail -25
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("headline", d["value"], d["kernel"]["tune_cfg"][:1], "rmse", d.get("rmse_vs_oracle"), "roofline", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], "e2e", d.get("end_to_end", {}).get("wall_s"))
for w in d.get("workloads", []):
    print(w["workload"], w["value"], w["kernel"]["tune_cfg"][:1], "rmse", w.get("rmse_vs_oracle"), "setup", w["setup_s"])
PY
for k in lds bcast; do
  HPT_GATHER_KERNEL=$k timeout 600 python bench.py --workload bunny --filter gaussian --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/bench_gauss_$k.json 2> $O/bench_gauss_$k.err
  python -c "
import json; d=json.load(open('$O/bench_gauss_$k.json')); print('gaussian $k', d['value'], d['kernel']['avg_ms'], d.get('rmse_vs_oracle'))"
done
HPT_BVH_BUILD=sah timeout 600 python bench.py --workload soup --spp 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $O/bench_soup_sah.json 2>/dev/null
python -c "
import json; d=json.load(open('$O/bench_soup_sah.json')); print('soup host SAH', d['value'], d['setup_s'])"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace_gauss -o t -- python $ROOT/bench.py --workload bunny --filter gaussian --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-verify > $ROOT/$O/trace_gauss.log 2>&1; cd $ROOT
grep -h "gather\|path_kernel" $O/trace_gauss/*kernel_stats.csv | cut -c1-200 | head -5
echo done > $O/done
