#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s7; mkdir -p $O
export SP_DIRECT=1
{
for v in g0 g1; do echo "--- $v"; timeout 120 ./build/probes/sp_$v 128 2; timeout 120 ./build/probes/sp_$v 128 1; timeout 120 ./build/probes/sp_$v 1024 2; done
} > $O/sp.txt 2>&1
cat $O/sp.txt
