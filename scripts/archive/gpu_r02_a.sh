#!/bin/bash
# Round 2, GPU call A: parity tests (incl. the 1080p crop tests), smoke, the full default bench line, FETCH_SIZE calibration,
# and an experiment: does RCCL accept two ranks on ONE device (needed for a single-GPU test of the RCCL film gather)?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -3 $O/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
timeout 600 scripts/calib/run_calib.sh > $O/calib.log 2>&1; tail -12 $O/calib.log
cat > /tmp/nccl2.py <<'EOT'
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=int(os.environ["RANK"]), world_size=2)
t = torch.ones(4, device="cuda") * (dist.get_rank() + 1)
dist.all_reduce(t); torch.cuda.synchronize()
print("rank", dist.get_rank(), "allreduce on one device ok:", t.tolist())
dist.destroy_process_group()
EOT
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 /tmp/nccl2.py > $O/nccl_same_device.log 2>&1; echo "nccl same device rc=$?"; tail -5 $O/nccl_same_device.log
echo done > $O/done
