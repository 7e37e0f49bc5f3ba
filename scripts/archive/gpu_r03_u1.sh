cd /root/repo; export TMPDIR=/tmp; O=gpurun_out/r03_u; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.txt 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR\|^E  " $O/pytest_gpu.txt | tail -30
