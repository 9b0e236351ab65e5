#!/bin/bash
# cfg 5 shard, DEFAULT launches of ell_flip_duo_kernel<2, 2, 1024> only (tools/bench_cfg5_variants.py flip: no ablation launch in the
# process, so the per-dispatch means are unmixed -- VERDICT round 5 item 2b): three separate PMC passes, no tracing domain.
#   usage (GPU box): bash tools/pmc_cfg5_default.sh [tag]      -> gpurun_out/<tag>/cfg5_pmc.md
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # run <dir> <counters...>
    local d=$1; shift
    for attempt in 1 2 3 4; do
        rm -rf $O/$d
        rocprofv3 --pmc "$@" -d $O/$d -o v -- python $R/tools/bench_cfg5_variants.py flip > $O/$d.log 2>&1
        if compgen -G "$O/$d/*.db" > /dev/null || compgen -G "$O/$d/*/*.db" > /dev/null; then return 0; fi
    done
    return 1
}
run c5_fetch FETCH_SIZE
run c5_write WRITE_SIZE
run c5_wait SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY
run c5_insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU
python - <<PY > $O/cfg5_pmc.md 2>&1
import glob, sqlite3
print("# cfg 5 shard: counters of the DEFAULT launches of ell_flip_duo_kernel<2, 2, 1024> only (tools/pmc_cfg5_default.sh)")
print()
print("128 instances, 20 Magnus-2 steps, 150 series terms per instance, one launch per solve; `tools/bench_cfg5_variants.py flip` under "
      "rocprofv3 --pmc, one counter set per process, no tracing domain; FETCH_SIZE / WRITE_SIZE in units of 32 B x 2 (gfx950 correction of "
      "MI355X_MICROARCH.md) summed over the chip per dispatch.")
print()
print("| counter | dispatches | mean per dispatch | min | max |")
print("|---|---|---|---|---|")
vals = {}
for d in ("c5_fetch", "c5_write", "c5_wait", "c5_insts"):
    dbs = glob.glob("$O/%s/*.db" % d) + glob.glob("$O/%s/*/*.db" % d)
    if not dbs:
        print("| (%s: no database) | | | | |" % d)
        continue
    con = sqlite3.connect(dbs[0])
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    pm = [t for t in tabs if t.startswith("pmc_events") or t == "counters_collection"]
    rows = []
    try:
        rows = list(con.execute("select dispatch_id, counter_name, sum(counter_value) from pmc_events where name like '%flip_duo%' group by dispatch_id, counter_name"))
    except sqlite3.Error:
        try:
            rows = list(con.execute("select dispatch_id, counter_name, sum(value) from counters_collection where kernel_name like '%flip_duo%' group by dispatch_id, counter_name"))
        except sqlite3.Error as exc:
            print("| (%s: %s; tables %s) | | | | |" % (d, exc, ", ".join(tabs[:12])))
    per = {}
    for did, cname, v in rows:
        per.setdefault(cname, []).append(v)
    for cname, xs in sorted(per.items()):
        scale = 32.0 * 2 if cname in ("FETCH_SIZE", "WRITE_SIZE") else 1.0     # KB -> bytes: rocprofv3 reports 32-B units? see the note below
        vals[cname] = sum(xs) / len(xs)
        print("| %s | %d | %.4g | %.4g | %.4g |" % (cname, len(xs), sum(xs) / len(xs), min(xs), max(xs)))
print()
if "FETCH_SIZE" in vals:
    print("FETCH_SIZE x 1024 B x 2 = %.1f MB per launch;" % (vals["FETCH_SIZE"] * 1024 * 2 / 1e6), end=" ")
if "WRITE_SIZE" in vals:
    print("WRITE_SIZE x 1024 B x 2 = %.1f MB per launch (published payload: 3 vectors x 32 KB x 256 workgroups x 150 terms = 3.8 GB; algorithmic bytes ~10 MB)." % (vals["WRITE_SIZE"] * 1024 * 2 / 1e6))
if "SQ_WAIT_ANY" in vals and "SQ_WAVE_CYCLES" in vals:
    print("SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f; SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = %.3f." % (vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"], vals.get("SQ_ACTIVE_INST_ANY", float("nan")) / vals["SQ_WAVE_CYCLES"]))
if "SQ_INSTS_VALU" in vals:
    tot = sum(vals.get(k, 0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM"))
    print("Instructions per launch (all waves): vector %.3g, scalar %.3g, LDS %.3g, scalar memory %.3g = %.3g; per wave and series term (4096 waves x 150 terms): %.0f." % (
        vals["SQ_INSTS_VALU"], vals.get("SQ_INSTS_SALU", 0), vals.get("SQ_INSTS_LDS", 0), vals.get("SQ_INSTS_SMEM", 0), tot, tot / (4096 * 150)))
PY
find $O -name "*.db" -delete
cat $O/cfg5_pmc.md
