#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (run on the GPU box).  Output: gpurun_out/calib/
cd "$(dirname "$0")/../.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/calib; mkdir -p $O
B=$ROOT/scripts/calib/_build/calib_fetch
$B > $O/known_bytes.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -o p -- $B > $O/fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -o p -- $B > $O/write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/rdreq -o p -- $B > $O/rdreq.log 2>&1
python $ROOT/scripts/calib/summarize_calib.py $O
