#!/bin/bash
# Round 4, run J: dry run of bench.py's N > 1 path on the one-GPU box — two and four ranks, all on device 0, gloo for the barrier, the library's
# host-staged film exchange (HPT_BENCH_ONE_DEVICE=1).  Checks the launch contract, the sharding, the exchange and the final line; the numbers mean nothing.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_j; mkdir -p $O
for n in 2 4; do
  HPT_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 2 --warmup 1 > $O/bench_n$n.txt 2> $O/bench_n$n.err
  echo "n=$n rc=$?"; tail -n 1 $O/bench_n$n.txt | cut -c1-1500; tail -3 $O/bench_n$n.err | cut -c1-300
done
