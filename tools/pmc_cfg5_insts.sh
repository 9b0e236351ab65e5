#!/bin/bash
# cfg 5 shard: instruction counters of ell_flip_duo_kernel<2, 2, 1024> for one variant of tools/bench_cfg5_variants.py per process
# (flip = default, no_exchange, nothing = skeleton, device_out ...): where the instructions of a series term are.
#   usage (GPU box): bash tools/pmc_cfg5_insts.sh <tag> <variant> [<variant> ...]     -> gpurun_out/<tag>/cfg5_insts.md
TAG=${1:-r06}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
echo "# cfg 5 shard: instructions per wave and series term of ell_flip_duo_kernel<2, 2, 1024> by variant (tools/pmc_cfg5_insts.sh)" > $O/cfg5_insts.md
echo >> $O/cfg5_insts.md
echo "| variant | vector | scalar | LDS | scalar memory | all | SQ_WAIT_ANY / SQ_WAVE_CYCLES | SQ_ACTIVE_INST_VALU x 4 / SQ_WAVE_CYCLES x (1/4 waves)|" >> $O/cfg5_insts.md
echo "|---|---|---|---|---|---|---|---|" >> $O/cfg5_insts.md
for V in "$@"; do
  for attempt in 1 2 3 4; do
    rm -rf $O/ci_$V
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $O/ci_$V -o v -- python $R/tools/bench_cfg5_variants.py $V > $O/ci_$V.log 2>&1
    if compgen -G "$O/ci_$V/*.db" > /dev/null || compgen -G "$O/ci_$V/*/*.db" > /dev/null; then break; fi
  done
  python - <<PY >> $O/cfg5_insts.md 2>&1
import glob, sqlite3
dbs = glob.glob("$O/ci_$V/*.db") + glob.glob("$O/ci_$V/*/*.db")
if not dbs:
    print("| $V | (no database) | | | | | | |")
else:
    con = sqlite3.connect(dbs[0])
    try:
        rows = list(con.execute("select dispatch_id, counter_name, sum(counter_value) from pmc_events where name like '%flip_duo%' group by dispatch_id, counter_name"))
    except sqlite3.Error:
        rows = list(con.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name like '%flip_duo%' group by dispatch_id, counter_name"))
    per = {}
    for did, cname, v in rows:
        per.setdefault(cname, []).append(v)
    m = {k: sum(x) / len(x) for k, x in per.items()}
    wt = 4096 * 150.0
    tot = sum(m.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
    print("| $V | %.0f | %.0f | %.0f | %.0f | %.0f | %.3f | %.3f |" % (m.get("SQ_INSTS_VALU", 0) / wt, m.get("SQ_INSTS_SALU", 0) / wt, m.get("SQ_INSTS_LDS", 0) / wt,
          m.get("SQ_INSTS_SMEM", 0) / wt, tot / wt, m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1), m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / max(m.get("SQ_WAVE_CYCLES", 1), 1)))
PY
done
find $O -name "*.db" -delete
cat $O/cfg5_insts.md
