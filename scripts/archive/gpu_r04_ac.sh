#!/bin/bash
# Round 4, run AC: the N > 1 path of bench.py once more on the FINAL build (two ranks on the one device, host-staged film exchange; the numbers mean nothing).
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_ac; mkdir -p $O
HPT_BENCH_ONE_DEVICE=1 timeout 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench_n2.txt 2> $O/bench_n2.err
echo "rc=$?"; tail -n 1 $O/bench_n2.txt | cut -c1-900; tail -2 $O/bench_n2.err | cut -c1-300
