#!/bin/bash
# GPU call L: full GPU test suite, smoke, default bench (all workloads) after the retrace change
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02l; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; grep -n "passed\|failed" $O/pytest_gpu.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 3000 $O/bench_default.json
