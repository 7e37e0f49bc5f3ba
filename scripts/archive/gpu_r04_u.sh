#!/bin/bash
# Round 4, run U: run T's two failures (film weights of env / ms differ between kernel configurations) — how often, which configuration, by how much.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/r04_u; mkdir -p $O
timeout 600 python scripts/stress_cfgs.py env,ms,k8 30 > $O/stress.txt 2>&1; tail -40 $O/stress.txt | cut -c1-400
