#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s7; mkdir -p $O
export SP_DIRECT=1
{
timeout 120 ./build/probes/sp 128 2; timeout 120 ./build/probes/sp 256 2; timeout 120 ./build/probes/sp 1024 2
} > $O/sp.txt 2>&1
cat $O/sp.txt
