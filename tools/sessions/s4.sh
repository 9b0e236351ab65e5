#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s4; mkdir -p $O
{
for args in "512 0 8" "512 4 4" "512 4 2" "512 5 4" "512 5 2" "1024 0 4" "1024 4 2" "1024 4 1" "256 0 8" "256 4 4" "256 5 4" "256 5 2" "4096 4 1"; do timeout 120 ./build/probes/gp $args; done
} > $O/gp.txt 2>&1
cat $O/gp.txt
timeout 1800 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
tail -5 $O/gpu_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
