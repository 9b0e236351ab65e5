#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s2; mkdir -p $O
export SP2_DIRECT=1
{
for v in p0 p2 p3; do echo "--- $v"; timeout 120 ./build/probes/sp2_$v 128 2; done
echo "--- order 1"; timeout 120 ./build/probes/sp2_p0 128 1;  timeout 120 ./build/probes/sp2_p2 128 1
echo "--- 256 / 1024 instances"; timeout 120 ./build/probes/sp2_p0 256 2; timeout 120 ./build/probes/sp2_p0 1024 2
} > $O/sp2.txt 2>&1
cat $O/sp2.txt
