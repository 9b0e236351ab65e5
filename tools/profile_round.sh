#!/bin/bash
# Runs on the GPU box (via gpurun): bench + rocprofv3 kernel trace + separate PMC passes (one counter set per run,
# never combined with tracing domains other than --kernel-trace/--stats).
# usage: tools/profile_round.sh <tag>      outputs under $GRAFT_REPO_ROOT/gpurun_out/<tag>/
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench.json 2> $O/bench.err
cp $R/bench_detail.json $O/bench_detail.json     # the full result the compact stdout line (bench.json) was reduced from
# (the traced command times the same 40 steps behind the same 16 warm-up steps as the default run: the first dozen launches
# of the dominant kernel run 10-20 % slower while the clocks settle, and a 10-step region would be mostly that)
B="python $R/bench.py --steps 40 --warmup 16 --repeats 1 --no-cpu-baseline --no-end-to-end --no-projection"
P="python $R/bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-end-to-end --no-projection --no-variants"
# (rocprofv3 of ROCm 7.2 sometimes dies inside a PMC pass, and segfaults at exit AFTER writing its database: a pass
# counts when its .db exists; up to six attempts)
pass() {  # pass <dir> <log> <command...>
    local d=$1 l=$2; shift 2
    for attempt in 1 2 3 4 5 6; do
        rm -rf $O/$d
        "$@" > $O/$l 2>&1
        if compgen -G "$O/$d/*.db" > /dev/null || compgen -G "$O/$d/*/*.db" > /dev/null; then return 0; fi
        echo "profile_round: $d attempt $attempt produced no database" >&2
    done
    return 1
}
pass stats prof_stats.log rocprofv3 --kernel-trace --stats -d $O/stats -o bench -- $B
pass pmc_fetch prof_fetch.log rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o bench -- $P
pass pmc_write prof_write.log rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o bench -- $P
pass pmc_mfma prof_mfma.log rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o bench -- $P
pass pmc_lds prof_lds.log rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc_lds -o bench -- $P
python $R/tools/summarize_rocprof.py $O $O/rocprof_bench.md "round $TAG: bench.py (cfg 3 headline + dense_complex + cfg 2 + cfg 4 + cfg 5 legs)" > /dev/null
python $R/tools/bench_splitk.py 64 128 256 512 1024 2048 4096 > $O/splitk.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; tail -5 $O/prof_stats.log; du -sh $O; ls $O
tail -c 600 $O/bench.json
