#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s8; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
grep -E "passed|failed|error" $O/gpu_tests.log | tail -3; grep -B5 -A30 "Error\|FAILED" $O/gpu_tests.log | head -60
timeout 300 python tools/bench_sweep_sizes.py > $O/sweep_sizes.txt 2>&1; grep "'ell_sweep': 1, 'ell_sweep_split': 1" $O/sweep_sizes.txt
