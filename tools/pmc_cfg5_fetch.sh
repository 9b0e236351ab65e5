#!/bin/bash
# (MIDYN_LIB_AB=<path of another libmidyn.so> measures that build.)
# FETCH_SIZE of every dispatch of the cfg 5 shard kernel (complete / without exchange / skeleton launches of
# tools/bench_cfg5_variants.py), one PMC pass, no tracing domain.   usage (GPU box): bash tools/pmc_cfg5_fetch.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05z
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for attempt in 1 2 3; do
  rm -rf $O/pmc_fetch5
  rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch5 -o v -- python $R/tools/bench_cfg5_variants.py ${VARIANTS:-flip no_exchange nothing} > $O/pmc_fetch5.log 2>&1
  if compgen -G "$O/pmc_fetch5/*.db" > /dev/null || compgen -G "$O/pmc_fetch5/*/*.db" > /dev/null; then break; fi
done
python - <<PY > $O/pmc_fetch5.txt 2>&1
import glob, sqlite3
db = (glob.glob("$O/pmc_fetch5/*.db") + glob.glob("$O/pmc_fetch5/*/*.db"))[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(pmc_events)")]
print(cols)
key = "dispatch_id" if "dispatch_id" in cols else cols[0]
for row in con.execute(f"select {key}, name, counter_name, sum(counter_value) from pmc_events where name like '%flip_duo%' group by {key} order by {key}"):
    print(row[0], row[2], "%.1f MB (x2-corrected)" % (row[3] * 1024 * 2 / 1e6))
PY
find $O -name "*.db" -delete
tail -2 $O/pmc_fetch5.log; cat $O/pmc_fetch5.txt
