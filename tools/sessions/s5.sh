#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s5; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
grep -E "passed|failed|error" $O/gpu_tests.log | tail -3; tail -30 $O/gpu_tests.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; tail -5 $O/bench.err
python tools/bench_diag_frame_sweep.py > $O/diag_sweep.txt 2>&1; tail -5 $O/diag_sweep.txt
