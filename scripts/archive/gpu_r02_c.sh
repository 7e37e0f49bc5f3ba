#!/bin/bash
# Round 2, GPU call C: the whole GPU suite + the default bench line (headline + killeroo / anim / soup as written + CPU baselines + pbrt_hip end to end)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/${TAG:-r02c}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("headline", d["value"], d["kernel"]["tune_cfg"], "rmse", d.get("rmse_vs_oracle"), "roofline", d["roofline"]["frac"], "cpu", d["cpu_baseline"]["kind"], d["cpu_baseline"]["value"], d.get("cpu_baseline_port", {}).get("value"), "e2e", d.get("end_to_end"))
for w in d.get("workloads", []):
    print(w["workload"], w["value"], w["kernel"]["tune_cfg"], "rmse", w.get("rmse_vs_oracle"), "roofline", w.get("roofline", {}).get("frac"))
PY
echo done > $O/done
