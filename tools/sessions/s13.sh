#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for v in clk NOBARRIER NOVMWAIT ONLYVMWAIT; do echo "== v2 $v"; timeout 60 build/probes/gp_$v 4096 10 1 | grep variant | sed 's/check max.d. = [0-9.e+-]*//'; done
