#!/bin/bash
# Round 6, GPU call S — the driver's multi-GPU command on the one-GPU box (HPT_BENCH_ONE_DEVICE=1: every rank on device 0, host-staged film exchange): the launch contract, the
# sharding, the exchange and the line (rccl_ranks, exchange_ms, north_star_scaling).  The numbers mean nothing; the line is what is checked.  Also: RCCL with ONE rank (the real transport).
cd "$(dirname "$0")/.."
ROOT=$PWD; export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06s; mkdir -p $O
for n in 2 4; do
  HPT_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 2 --warmup 1 > $O/bench_n$n.txt 2> $O/bench_n$n.err
  echo "n=$n rc=$?"; tail -n 1 $O/bench_n$n.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print({k: d.get(k) for k in ('value', 'n_gpus', 'scaling', 'rccl_ranks', 'exchange_ms', 'film_exchange', 'north_star_scaling')})
for w in d.get('workloads', []): print(w.get('workload')[:60], w.get('value'), w.get('scaling'), w.get('rccl_ranks'), w.get('exchange_ms'))"
done
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q 2>&1 | tail -3
