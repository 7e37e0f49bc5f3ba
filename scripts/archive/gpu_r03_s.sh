#!/bin/bash
# Round 3, run S: instruction-cache counters of the path kernel (the killeroo kernel is 149 KB of code against a 64 KB instruction cache shared by
# two CUs): SQC_ICACHE_* / SQ_IFETCH* on killeroo, bunny and metal, one rocprofv3 --pmc pass each (kernel trace only).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r03_s; mkdir -p $O
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_WAVES_[A-Z_]*\|SQ_BUSY_CU[A-Z_]*" | sort -u | tr '\n' ' ' > $O/counters.txt; cat $O/counters.txt; echo
for w in killeroo bunny metal; do
B="python $ROOT/bench.py --workload $w --no-cpu-baseline --no-verify --no-extra --no-pmc --no-work --steps 1 --warmup 1"
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/ic_$w -o p -- $B > $O/ic_$w.log 2>&1
echo "$w rc=$?"
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = 0
for f in glob.glob("$O/ic_$w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "hpt_path_kernel" in r["Kernel_Name"] and "true, false, true" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print("$w", dict(acc))
PY
done
