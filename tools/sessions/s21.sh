#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s21; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
grep -E "passed|failed|error" $O/gpu_tests.log | tail -3; grep -B5 -A30 "Error\|FAILED" $O/gpu_tests.log | head -60
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/profile_round.sh r03 > $O/profile.log 2>&1; tail -3 $O/profile.log
timeout 600 python tools/bench_baseline_configs.py > $O/baseline_configs.jsonl 2>&1; cat $O/baseline_configs.jsonl | cut -c1-220
timeout 300 python tools/bench_diag_frame_sweep.py > $O/diag_sweep.txt 2>&1; tail -2 $O/diag_sweep.txt | cut -c1-400
