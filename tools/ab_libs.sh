#!/bin/bash
# A/B of builds of libmidyn.so on one box: tools/ab_libs.sh ROUNDS lib1.so lib2.so ...  (alternating processes of tools/bench_cfg5_variants.py)
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    echo -n "$(basename $lib): "
    MIDYN_LIB_AB=$PWD/$lib timeout 250 python tools/bench_cfg5_variants.py flip no_exchange nothing 2>&1 | grep "^{" | head -1
  done
done
