#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in a b c a b c; do
  echo "== build $b"
  for v in 0 1 3 2; do build/probes/gp_$b 4096 $v 1 | sed 's/executed.*= / /'; done
  build/probes/gp_$b 512 0 8 | sed 's/executed.*= / /'
done
