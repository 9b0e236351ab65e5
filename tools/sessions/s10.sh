#!/bin/bash
# final-ish validation of the round: full GPU tests, smoke, refreshed tool numbers, then the profile round
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s10; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
grep -E "passed|failed|error" $O/gpu_tests.log | tail -3
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 600 python tools/bench_baseline_configs.py > $O/baseline_configs.jsonl 2>&1; cat $O/baseline_configs.jsonl | cut -c1-200
timeout 300 python tools/bench_sweep_sizes.py > $O/sweep_sizes.txt 2>&1; grep "'ell_sweep': 1, 'ell_sweep_split': 1" $O/sweep_sizes.txt
timeout 300 python tools/bench_diag_frame_sweep.py > $O/diag_sweep.txt 2>&1; tail -2 $O/diag_sweep.txt | cut -c1-600
timeout 300 python tools/soak_resident.py > $O/soak.txt 2>&1; tail -4 $O/soak.txt
python bench.py --dry-ranks 2>/dev/null | tail -1
