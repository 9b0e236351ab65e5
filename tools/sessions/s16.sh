#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s16; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
grep -E "passed|failed|error" $O/gpu_tests.log | tail -3; grep -B5 -A30 "Error\|FAILED" $O/gpu_tests.log | head -60
timeout 300 python bench.py --steps 40 --warmup 4 --repeats 3 --no-cpu-baseline --no-end-to-end > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(d["value"], d["repeat_rhs_evals_per_s"], d["roofline"]["frac"], d["dense_kernels_same_model"]["rhs_evals_per_s"], d["dense_kernels_same_model"]["frac"], d["dense_complex"]["rhs_evals_per_s"], d["projected_strong_scaling"]["cfg3"])
print(d["cfg5"]["mfma_work_list_route"]["ms_per_step"], d["cfg4"]["ms_per_step"], d["single_trajectory"]["rhs_evals_per_s"])
PY
timeout 300 python tools/bench_splitk.py 64 128 256 512 1024 2048 4096 2>&1 | tail -8
