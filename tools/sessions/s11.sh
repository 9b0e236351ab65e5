#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 4096 512; do
  sp=1; [ $n = 512 ] && sp=8
  echo "== N $n splits $sp"
  timeout 120 build/probes/gp_base $n 0 $sp
  timeout 120 build/probes/gp_exact $n 0 $sp
  timeout 120 build/probes/gp_exact $n 10 $sp
done
timeout 120 build/probes/gp_exact 4096 1 1
timeout 120 build/probes/gp_exact 2048 10 2
timeout 120 build/probes/gp_exact 2048 0 2
