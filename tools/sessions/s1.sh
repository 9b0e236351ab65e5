#!/bin/bash
# GPU session 1 (round 3): sweep-split probes, GEMM probe baselines, new headline tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s1; mkdir -p $O
{
for args in "128 2 0" "128 2 1" "64 4 0" "64 4 1" "32 4 1" "16 4 1"; do timeout 120 ./build/probes/ssp_a0 $args; done
echo "--- ablate 1 (no operator pass)"
for args in "128 2 0" "128 2 1" "64 4 1"; do timeout 120 ./build/probes/ssp_a1 $args; done
} > $O/ssp.txt 2>&1
{
for args in "4096 0" "4096 1" "4096 2" "4096 3" "512 0 8" "512 0 4" "2048 0 2" "1024 0 4"; do timeout 120 ./build/probes/gp $args; done
} > $O/gp.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_headline.py -x -q -m gpu > $O/headline.log 2>&1
tail -5 $O/headline.log
cat $O/ssp.txt $O/gp.txt
