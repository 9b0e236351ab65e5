#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for b in e f e f; do echo "== $b"; for v in 0 1 3 2; do build/probes/gp_$b 4096 $v 1 | sed 's/executed.*= / /'; done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc -o x -- $GRAFT_REPO_ROOT/build/probes/gp_f 4096 0 1 > /tmp/pmc.log 2>&1
python - <<'PY'
import sqlite3, glob
db=glob.glob('/tmp/pmc/*.db')+glob.glob('/tmp/pmc/*/*.db')
con=sqlite3.connect(db[0])
for r in con.execute("select name, counter_name, count(*), sum(counter_value) from pmc_events group by name, counter_name"):
    if 'zgemm' in r[0]: print(r[0][:60], r[1], r[2], r[3])
PY
