#!/bin/bash
# Film-atomic calibration (round 6): timing + WRITE_SIZE / FETCH_SIZE per dispatch of scripts/calib/calib_atomic.hip.  Output: gpurun_out/calib_atomic/
cd "$(dirname "$0")/../.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/calib_atomic; mkdir -p $O
B=$ROOT/scripts/calib/_build/calib_atomic
$B > $O/timing.txt 2>&1; cat $O/timing.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum --output-format csv -d $O/wrreq -o p -- $B > $O/wrreq.log 2>&1
python $ROOT/scripts/calib/summarize_atomic.py $O
