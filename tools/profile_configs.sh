#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel trace of the secondary configs (cfg 4, cfg 5, sweeps).
# usage: tools/profile_configs.sh <tag>     output: $GRAFT_REPO_ROOT/gpurun_out/<tag>_configs/summary.md
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_configs
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats -o configs -- python $R/tools/bench_configs.py cfg4 cfg4sweep cfg5 > $O/configs.jsonl 2> $O/prof.log
python $R/tools/summarize_rocprof.py $O $O/summary.md "tools/bench_configs.py cfg4 cfg4sweep cfg5 (round 1, final code)" > /dev/null 2>&1
rm -rf $O/stats
head -40 $O/summary.md | cut -c1-180
