#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power\|mclk" | head -8
echo "--- under load (variant 0, N 4096, looped)"
( for i in 1 2 3 4 5 6 7 8 9 10 11 12; do build/probes/gp_exact 4096 0 1 > /dev/null; done ) &
L=$!
sleep 4
for i in 1 2 3 4; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Socket Power\|Average" | head -4; sleep 1.5; done
wait $L
echo "--- dense 4M (variant 3)"
( for i in 1 2 3 4 5 6; do build/probes/gp_exact 4096 3 1 > /dev/null; done ) &
L=$!
sleep 4
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|Socket Power\|Average" | head -4; sleep 1.5; done
wait $L
