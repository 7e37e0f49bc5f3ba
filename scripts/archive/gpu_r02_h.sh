#!/bin/bash
# GPU call H: PMC counters of the gather kernel (bunny, gaussian filter)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
ROOT=$PWD
O=$ROOT/gpurun_out/r02h; mkdir -p $O
B="python $ROOT/bench.py --workload bunny --filter gaussian --no-cpu-baseline --no-verify --no-extra --steps 1 --warmup 0"
cd /tmp
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o p -- $B > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"; }
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
pass lat SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
pass fetch FETCH_SIZE
cd $ROOT
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r02h/pmc_*/')):
    f = glob.glob(d + '*counter_collection.csv')
    if not f: continue
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if 'gather' in r['Kernel_Name']:
            acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    print(d, {k: (v, n[k]) for k, v in acc.items()}, 'lds', [r.get('LDS_Block_Size') for r in csv.DictReader(open(f[0])) if 'gather' in r['Kernel_Name']][:1])
PY
