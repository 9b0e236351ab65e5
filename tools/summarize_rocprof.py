#!/usr/bin/env python
"""Turn rocprofv3 (rocpd sqlite) outputs into the text summaries committed under profiles/.

usage: python tools/summarize_rocprof.py <dir with stats/ pmc_*/ sub-dirs> <out.md> [title]
Reads <dir>/stats/*.db (--kernel-trace --stats run) and every <dir>/pmc_*/*.db (--pmc runs, one
counter set per run as MI355X_MICROARCH.md prescribes).  FETCH_SIZE is reported raw (KiB) AND
corrected x2 (gfx950: 128-B requests are tallied as 64 B for wide coalesced reads).

When <dir>/prof_stats.log holds the bench line that the traced command printed (bench.py's ONE JSON line), the
dominant kernel's dispatches are split by that line's `roofline.launch_sequence` -- warm-up, the timed back-to-back
repetitions, the per-launch-event pass -- and the timed repetition is compared with the SAME run's `ms_per_step` and
`roofline.frac` (the whole-trace average mixes launches that are bracketed by event records with back-to-back ones).
"""
import glob
import json
import os
import sqlite3
import statistics
import sys
from collections import defaultdict


def short(name):
    name = name.split("(")[0]
    for a, b in (("midyn::", ""), ("void ", "")):
        name = name.replace(a, b)
    return name[:90]


def kernel_stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), "
                       "max(grid_x), max(workgroup_x) from kernels group by name order by sum(duration) desc").fetchall()
    med = {}
    for (name,) in con.execute("select distinct name from kernels").fetchall():
        d = [r[0] for r in con.execute("select duration from kernels where name = ?", (name,)).fetchall()]
        med[name] = statistics.median(d)
    con.close()
    return rows, med


def kernel_dispatches(db, wanted):
    """(start, end) of every dispatch whose shortened name starts with `wanted`, in start order."""
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select name, start, end from kernels order by start").fetchall()
    except sqlite3.OperationalError:      # a rocpd schema without start / end in the kernels view
        rows = []
    con.close()
    key = wanted.replace(" ", "")
    return [(s, e) for name, s, e in rows if short(name).replace(" ", "").startswith(key)]


def bench_line(log):
    """bench.py's JSON line out of the log of the traced command (None when absent)."""
    try:
        with open(log, errors="replace") as f:
            for ln in f:
                i = ln.find('{"metric"')
                if i >= 0:
                    try:
                        return json.loads(ln[i:])
                    except ValueError:
                        continue
    except OSError:
        pass
    return None


def phase_report(db, line):
    """Markdown lines: the dominant kernel's dispatches by phase of bench.py, against the same run's bench line."""
    roof = (line or {}).get("roofline") or {}
    seq = roof.get("launch_sequence")
    kname = (roof.get("kernel") or "").split(" (")[0]
    if not seq or not kname:
        return []
    disp = kernel_dispatches(db, kname)
    want = sum(c for _, c in seq)
    out = ["## dominant kernel by phase of the traced bench.py run", "",
           f"`{kname}`: {len(disp)} dispatches in the trace, `launch_sequence` of the same run accounts for {want}."]
    if len(disp) < want:
        out += ["", "(fewer dispatches than the sequence: no per-phase split)", ""]
        return out
    if len(disp) > want:
        out += ["", f"(the first {want} dispatches are the sequence; the other {len(disp) - want} belong to later A/B legs of "
                    "bench.py that run the same kernel on other lists -- e.g. skip_zero_blocks=0 -- and are left out here)"]
        disp = disp[:want]
    flops = roof.get("executed_mfma_flops_per_launch")
    peak = roof.get("peak")
    out += ["",
            "| phase | launches | avg us | median us | min us | max us | span / launch us | frac of peak (avg) | (median) | (span) |",
            "|---|---|---|---|---|---|---|---|---|---|"]
    pos = 0
    timed = None
    for phase, count in seq:
        part = disp[pos:pos + count]
        pos += count
        if not part:
            continue
        dur = [e - s for s, e in part]
        avg, med = sum(dur) / len(dur), statistics.median(dur)
        span = (part[-1][1] - part[0][0]) / len(part)     # first start -> last end, per launch (includes the gaps)

        def frac(ns):
            return f"{flops / (ns * 1e-9) / 1e12 / peak:.4f}" if flops and peak else ""

        out.append(f"| {phase} | {len(part)} | {avg / 1e3:.2f} | {med / 1e3:.2f} | {min(dur) / 1e3:.2f} | "
                   f"{max(dur) / 1e3:.2f} | {span / 1e3:.2f} | {frac(avg)} | {frac(med)} | {frac(span)} |")
        if phase == "timed_rep0":
            timed = (avg, med, span)
    if timed:
        avg, med, span = timed
        ms_step = line.get("ms_per_step")
        out += ["",
                f"Same run's bench line: `ms_per_step` = {ms_step} (host clock around the K steps), "
                f"`roofline.avg_launch_ms` = {roof.get('avg_launch_ms')} (HIP events around the same region / 4K), "
                f"`roofline.frac` = {roof.get('frac')}.",
                f"Trace, timed repetition: 4 x avg = {4 * avg / 1e6:.4f} ms, 4 x median = {4 * med / 1e6:.4f} ms, "
                f"4 x span per launch = {4 * span / 1e6:.4f} ms per step.",
                "(kernel durations exclude the gaps between dispatches; the span and the HIP-event / host figures include "
                "them -- under `--kernel-trace` every dispatch carries the profiler's completion signal)", ""]
    return out


def pmc_stats(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, counter_name, count(*), avg(counter_value), sum(counter_value) "
                       "from pmc_events group by name, counter_name").fetchall()
    con.close()
    return rows


def main():
    root, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(root)
    lines = [f"# rocprofv3 summary: {title}", ""]
    line = bench_line(os.path.join(root, "prof_stats.log"))
    for db in sorted(glob.glob(os.path.join(root, "stats", "*.db")) + glob.glob(os.path.join(root, "stats", "*", "*.db"))):
        rows, med = kernel_stats(db)
        total = sum(r[2] for r in rows) or 1
        lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats`)", "",
                  "| kernel | calls | total ms | avg us | median us | min us | max us | % | vgpr | agpr | sgpr | lds B | scratch | grid | wg |",
                  "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
        for r in rows:
            lines.append(f"| `{short(r[0])}` | {r[1]} | {r[2] / 1e6:.3f} | {r[3] / 1e3:.2f} | {med[r[0]] / 1e3:.2f} | "
                         f"{r[4] / 1e3:.2f} | {r[5] / 1e3:.2f} | {100 * r[2] / total:.1f} | {r[6]} | {r[7]} | {r[8]} | "
                         f"{r[9]} | {r[10]} | {r[11]} | {r[12]} |")
        lines.append("")
        lines += phase_report(db, line)
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        for db in sorted(glob.glob(os.path.join(d, "*.db")) + glob.glob(os.path.join(d, "*", "*.db"))):
            rows = pmc_stats(db)
            if not rows:
                continue
            lines += [f"## counters ({os.path.basename(d)}; separate `rocprofv3 --pmc` pass)", "",
                      "| kernel | counter | dispatches | avg per dispatch | note |", "|---|---|---|---|---|"]
            by_kernel = defaultdict(dict)
            for name, cname, n, avg, tot in rows:
                by_kernel[name][cname] = avg
                note = ""
                if cname == "FETCH_SIZE":
                    note = f"KiB; x2 gfx950 correction -> {2 * avg * 1024 / 1e6:.1f} MB/dispatch"
                if cname == "WRITE_SIZE":
                    note = f"KiB -> {avg * 1024 / 1e6:.1f} MB/dispatch (uncalibrated)"
                lines.append(f"| `{short(name)}` | {cname} | {n} | {avg:.4g} | {note} |")
            for name, c in by_kernel.items():
                if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
                    lines.append(f"| `{short(name)}` | MFMA busy / SQ busy | | "
                                 f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f} | ratio of summed counters |")
            lines.append("")
    # machine-readable HBM traffic per dispatch (read side x2 as the gfx950 note prescribes)
    traffic = {}
    for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
        for db in sorted(glob.glob(os.path.join(d, "*.db")) + glob.glob(os.path.join(d, "*", "*.db"))):
            for name, cname, n, avg, tot in pmc_stats(db):
                if cname in ("FETCH_SIZE", "WRITE_SIZE"):
                    t = traffic.setdefault(short(name), {})
                    if cname == "FETCH_SIZE":
                        t["fetch_bytes"] = 2.0 * avg * 1024
                    else:
                        t["write_bytes"] = avg * 1024
                    t["dispatches"] = n
    if traffic:
        with open(os.path.splitext(out)[0] + ".traffic.json", "w") as f:
            json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, {title}; FETCH_SIZE x2 (gfx950)",
                       "kernels": traffic}, f, indent=1)
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
