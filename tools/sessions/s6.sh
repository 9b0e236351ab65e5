#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s6; mkdir -p $O
{
for b in gp gp_s1 gp_s2 gp_s3; do echo "--- $b"; timeout 120 ./build/probes/$b 4096 8;  timeout 120 ./build/probes/$b 4096 8; done
} > $O/gp.txt 2>&1
cat $O/gp.txt
