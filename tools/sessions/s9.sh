#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s9; mkdir -p $O
{ timeout 120 ./build/probes/gp 4096 0; timeout 120 ./build/probes/gp 4096 9; } > $O/gp.txt 2>&1
cat $O/gp.txt
