#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_headline.py -x -q -m gpu 2>&1 | tail -5
for ns in 1 2 4 8; do
  echo "== stage_streams $ns"
  timeout 300 python bench.py --steps 40 --warmup 4 --repeats 2 --no-cpu-baseline --no-end-to-end --no-single --no-configs --no-projection --opt stage_streams=$ns > $O/b_$ns.json 2> $O/b_$ns.err
  python - <<PY
import json
d=json.load(open("$O/b_$ns.json"))
print(d["value"], d["repeat_rhs_evals_per_s"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("dense_kernels_same_model",{}).get("rhs_evals_per_s"), d.get("dense_complex",{}).get("rhs_evals_per_s"), d["max_norm_deviation"])
PY
done
