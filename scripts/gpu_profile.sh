#!/bin/bash
# rocprofv3 evidence for one bench workload: kernel trace + stats, then PMC passes (each in its own
# run, with --kernel-trace only, as the MI355X guide prescribes).  Output under gpurun_out/prof_*/.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
export TMPDIR=/tmp
W=${1:-bunny}
O=$ROOT/gpurun_out/prof_${W}_$(date +%H%M%S)
mkdir -p $O
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
B="python $ROOT/bench.py --workload $W --no-cpu-baseline --no-verify --no-extra --no-pmc --no-work ${BENCH_EXTRA:-}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- $B --steps 3 --warmup 1 > $O/trace.log 2>&1
pass() { # name counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -o p -- $B --steps 1 --warmup 0 > $O/pmc_$n.log 2>&1
  echo "pmc $n rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
if [ "${PROF_SHORT:-0}" = "1" ]; then   # HBM bytes, L2 hit rate and lane utilisation only (a workload whose set-up dominates the run: the 4 M-triangle soup)
  pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
  pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
  find $O -name "*.csv" | head -40
  for f in $(find $O/trace -name "*kernel_stats.csv"); do cat $f; done
  echo done > $O/done
  exit 0
fi
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
pass sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES GRBM_GUI_ACTIVE
if [ "${PROF_EXTRA:-0}" = "1" ]; then
  pass ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL GRBM_GUI_ACTIVE
  pass lat SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_FLAT GRBM_GUI_ACTIVE
  pass tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
fi
find $O -name "*.csv" | head -40
for f in $(find $O/trace -name "*kernel_stats.csv"); do cat $f; done
echo done > $O/done
