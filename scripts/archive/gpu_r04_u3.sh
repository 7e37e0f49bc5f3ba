#!/bin/bash
# Round 4, run U3: the full GPU suite twice more (run T's two failures did not reproduce in 5 partial sessions + 1680 stress renders).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_u3; mkdir -p $O
for i in 1 2; do
  timeout 900 python -m pytest tests -q -m gpu > $O/pytest_$i.txt 2>&1; echo "session $i rc=$?"; tail -1 $O/pytest_$i.txt; grep "^E  *AssertionError\|^E  *assert" $O/pytest_$i.txt | head -6 | cut -c1-400
done
