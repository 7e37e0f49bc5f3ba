#!/bin/bash
# Round 2, GPU call D: the whole suite (round-2 features on the device, multi-GPU in the library) + bench default + metal workload
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r02d}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; grep -E "^FAILED|^ERROR|passed|failed" $O/pytest_gpu.log | tail -25
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("headline", d["value"], d["kernel"]["tune_cfg"], "rmse", d.get("rmse_vs_oracle"), "roofline", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], "e2e", d.get("end_to_end", {}).get("wall_s"))
for w in d.get("workloads", []):
    print(w["workload"], w["value"], w["kernel"]["tune_cfg"], "rmse", w.get("rmse_vs_oracle"), "roofline", w.get("roofline", {}).get("frac"))
PY
timeout 900 python bench.py --workload metal --spp 16 --steps 2 --warmup 1 --no-extra > $O/bench_metal.json 2> $O/bench_metal.err; echo "metal rc=$?"; tail -2 $O/bench_metal.err; python -c "
import json; d=json.load(open('$O/bench_metal.json')); print('metal', d['value'], d['kernel'], 'rmse', d.get('rmse_vs_oracle'), d['cpu_baseline']['value'])"
echo done > $O/done
