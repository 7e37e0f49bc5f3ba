#!/bin/bash
# Round 5, GPU call Q: dry run of bench.py's N > 1 path (two and four ranks on the one device, host-staged exchange) after this round's changes to the timed loop
# (two films, copy stream, film on the host); the new single-frame host-transport test
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r05q; mkdir -p $O
for n in 2 4; do
  HPT_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 2 --warmup 1 > $O/bench_n$n.txt 2> $O/bench_n$n.err
  echo "n=$n rc=$?"; tail -n 1 $O/bench_n$n.txt | cut -c1-900; tail -3 $O/bench_n$n.err | cut -c1-300
done
timeout 900 python -m pytest tests/test_gpu_multi.py -m gpu -q 2>&1 | tail -4
