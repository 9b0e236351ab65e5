#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s15; mkdir -p $O
run() {  # run <tag> <ns> [env...]
  tag=$1; ns=$2; shift 2
  env "$@" timeout 300 python bench.py --steps 40 --warmup 4 --repeats 2 --no-cpu-baseline --no-end-to-end --no-single --no-configs --no-projection --opt stage_streams=$ns > $O/b_$tag.json 2> $O/b_$tag.err
  python - <<PY
import json
d=json.load(open("$O/b_$tag.json"))
print("$tag", d["value"], d["repeat_rhs_evals_per_s"], d["ms_per_step"])
PY
}
run q8_ns4 4 GPU_MAX_HW_QUEUES=8
run q8_ns2 2 GPU_MAX_HW_QUEUES=8
run q8_ns8 8 GPU_MAX_HW_QUEUES=8
run q4_ns2 2 FOO=1
run q4_ns1 1 FOO=1
run dbg_ns4 4 AMD_SERIALIZE_KERNEL=0 HIP_FORCE_DEV_KERNARG=1
