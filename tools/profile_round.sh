#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes.
# usage: tools/profile_round.sh <tag>      outputs under $GRAFT_REPO_ROOT/gpurun_out/<tag>/
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
python $R/tools/bench_configs.py > $O/configs.jsonl 2> $O/configs.err
B="python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-end-to-end"
rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B > $O/prof_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_mfma.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_lds -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $O/prof_lds.log 2>&1
tail -c 3000 $O/bench.json
