#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s19; mkdir -p $O
for b in c d c d; do for v in 0 1 3 2; do build/probes/gp_$b 4096 $v 1 | sed 's/executed.*= / /'; done; build/probes/gp_$b 512 0 8 | sed 's/executed.*= / /'; done
timeout 900 python -m pytest tests/test_gpu_steps_kernel.py -x -q -m gpu 2>&1 | tail -5
for opt in "steps_kernel=0" "steps_kernel=1" "steps_kernel_skew=0"; do
  echo "== $opt"
  MIDYN_STEPS_DEBUG=1 timeout 300 python bench.py --steps 40 --warmup 4 --repeats 3 --no-cpu-baseline --no-end-to-end --no-single --no-configs --no-projection --opt $opt > $O/b.json 2> $O/b.err
  python - <<PY
import json
d=json.load(open("$O/b.json"))
print(d["value"], d["repeat_rhs_evals_per_s"], d["ms_per_step"], d.get("dense_kernels_same_model",{}).get("rhs_evals_per_s"), d.get("dense_complex",{}).get("rhs_evals_per_s"), d["max_norm_deviation"])
PY
  grep "steps kernel" $O/b.err | head -3
done
