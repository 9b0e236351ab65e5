#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in e p p e e p p e; do echo -n "$b: "; build/probes/gp_$b 4096 0 1 | sed 's/check.*launch, //;s/executed.*= / /'; done
for b in e p p e; do echo -n "$b dense: "; build/probes/gp_$b 4096 1 1 | sed 's/check.*launch, //;s/executed.*= / /'; done
for b in e p p e; do echo -n "$b 3M: "; build/probes/gp_$b 4096 2 1 | sed 's/check.*launch, //;s/executed.*= / /'; done
