"""The ONE stdout line of bench.py.

The driver keeps the last 8 018 characters of bench.py's stdout and parses the last line as JSON; round 5's line had
grown to 31 KB and was not parsed.  `compact_line(out)` reduces the full result dict to the
contract fields + `roofline` + `cpu_baseline` + one small object per other configuration / SURVEY section 8 row, and
`emit(out, ...)` writes everything else to `bench_detail.json` (beside bench.py, and under gpurun_out/ when that directory
exists) and to stderr.  tests/test_host_logic.py holds the size and key-set assertions (MAX_LINE_BYTES)."""
import json
import os
import sys

MAX_LINE_BYTES = 6144          # the driver's stdout tail holds 8 018 characters; stay well inside it
DETAIL_NAME = "bench_detail.json"

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "config")
ROOFLINE_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                 "executed_flops_per_launch", "frac_survey_8d")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _get(d, *path, default=None):
    for p in path:
        if not isinstance(d, dict) or p not in d:
            return default
        d = d[p]
    return d


def _sig(x, digits=4):
    """floats to `digits` significant digits (the detail file keeps full precision)"""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None                       # no NaN / Infinity tokens on the line
    return float("%.*g" % (digits, x))


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 3] + "..."


def _pair(leg, value_key, unit=None, roof=("roofline",), **more):
    """{value, unit, frac, bound} of one leg + a few named scalars"""
    if not isinstance(leg, dict):
        return None
    if "error" in leg and len(leg) == 1:
        return {"error": _short(leg["error"], 120)}
    r = _get(leg, *roof, default={}) or {}
    o = {"value": _sig(leg.get(value_key)), "unit": unit, "frac": _sig(r.get("frac")), "bound": _short(r.get("bound"), 24) if r.get("bound") else None}
    for name, key in more.items():
        v = _get(leg, *key) if isinstance(key, tuple) else leg.get(key)
        if v is not None:
            o[name] = _sig(v)
    cb = leg.get("cpu_baseline")
    if isinstance(cb, dict) and cb.get("value") is not None:
        o["cpu"] = {"value": _sig(cb.get("value")), "unit": _short(cb.get("unit"), 24), "cores": cb.get("cores"),
                    "kind": _short(cb.get("kind"), 12)}
    return o


def compact_line(out, detail=DETAIL_NAME):
    """The driver-facing reduction of bench.py's full result dict `out` (a dict that json.dumps to < MAX_LINE_BYTES)."""
    c = {k: (_short(out[k], 120) if isinstance(out[k], str) else out[k]) for k in CONTRACT_KEYS if k in out}
    c["dtype"] = "f64"                                          # complex128 = pairs of f64; the arithmetic type
    cfg = dict(out.get("config") or {})
    c["config"] = {k: (_short(v, 200 if k == "workload" else 40) if isinstance(v, str) else v) for k, v in cfg.items()
                   if isinstance(v, (str, int, float, bool)) or v is None}
    for k in ("solve_wall_clock_s", "stream_ms_per_step", "repeat_spread", "max_norm_deviation"):
        if k in out:
            c[k] = _sig(out[k])
    r = out.get("roofline")
    if isinstance(r, dict):
        rr = dict(r)
        rr["kernel"] = str(r.get("kernel", "")).split(" (")[0]
        rr.setdefault("executed_flops_per_launch", r.get("executed_mfma_flops_per_launch"))
        c["roofline"] = {k: (_short(rr[k], 60) if isinstance(rr.get(k), str) else _sig(rr.get(k), 5)) for k in ROOFLINE_KEYS}
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c["cpu_baseline"] = {k: (_short(cb[k], 200 if k == "sample" else 24) if isinstance(cb[k], str) else _sig(cb[k]))
                             for k in CPU_KEYS if k in cb}
    t3 = out.get("cfg3_three_numbers")
    if isinstance(t3, dict):
        c["cfg3_three_numbers"] = {k: t3[k] for k in ("structured_model_default_route", "same_planes_no_zero_blocks",
                                                      "general_complex_operators") if k in t3}
    dc = out.get("dense_complex")
    if isinstance(dc, dict):
        c["dense_complex"] = {"value": _sig(dc.get("rhs_evals_per_s")), "unit": "RHS evals/s", "frac": _sig(dc.get("frac")),
                              "avg_launch_ms": _sig(dc.get("avg_launch_ms"))}
    e2e = out.get("end_to_end_solve")
    if isinstance(e2e, dict):
        c["end_to_end_solve"] = {"solve_s": e2e.get("solve_s"), "value": _sig(e2e.get("rhs_evals_per_s_end_to_end")),
                                 "unit": "RHS evals/s"}
    # ---- one {value, frac} object per other BASELINE configuration / SURVEY section 8 row ---------------------------------
    st = out.get("single_trajectory")
    if isinstance(st, dict):
        c["cfg2"] = _pair({**st, "roofline": out.get("roofline_single_trajectory_default_route") or {}}, "rhs_evals_per_s",
                          "RHS evals/s", us_per_evaluation="us_per_evaluation")
        if c["cfg2"] is not None:
            c["cfg2"]["stream_kernel_hbm_frac"] = _sig(_get(out, "roofline_single_trajectory", "frac"))
    if "cfg4" in out:
        c["cfg4"] = _pair(out["cfg4"], "ms_per_step", "ms per scipy_expm step", us_per_product="us_per_product",
                          products_per_step="products_per_step", diag_frame_ms_per_step=("diag_frame_run", "ms_per_step"))
    if "cfg5" in out:
        c["cfg5"] = _pair(out["cfg5"], "ms_per_step", "ms per Magnus-2 step, 128-instance shard",
                          us_per_term=("roofline", "us_per_term"), us_per_term_device_out=("roofline", "us_per_term_device_out"),
                          us_per_instance_step="us_per_instance_step",
                          solve_s="solve_s", first_call_s="solve_s_first_call_of_a_grid", kernel_ms=("roofline", "avg_launch_ms"),
                          plan_solve_s=("plan", "solve_s"),
                          host_ms_one_shot=("roofline", "host_side_ms_one_shot"), host_ms_plan=("roofline", "host_side_ms_plan"))
        if c["cfg5"] is not None and "roofline" in out["cfg5"]:
            c["cfg5"]["kernel"] = _short(str(_get(out, "cfg5", "roofline", "kernel", default="")).split(" (")[0], 60)
    s5 = out.get("sharded_cfg5")
    if isinstance(s5, dict):
        c["sharded_cfg5"] = ({"error": _short(s5["error"], 120)} if "error" in s5 else
                             {"value": _sig(s5.get("instance_steps_per_s")), "unit": "instance-steps/s",
                              "solve_s": s5.get("solve_s_max_over_ranks"), "instances_per_gpu": s5.get("instances_per_gpu")})
    de = out.get("dense_expm")
    if isinstance(de, dict):
        c["dense_expm"] = ({"error": _short(de["error"], 120)} if "error" in de else
                           {k: _pair(v, "kernel_ms", "ms per expm (kernels)") for k, v in de.items() if isinstance(v, dict)})
    if "lindblad_rk4_unvectorized" in out:
        c["f2"] = _pair(out["lindblad_rk4_unvectorized"], "rhs_evals_per_s", "RHS evals/s (n=1024 Lindblad)")
    if "parallel_in_time" in out:
        c["f3"] = _pair(out["parallel_in_time"], "steps_per_s_parallel_in_time", "steps/s")
    pt = out.get("perturbative")
    if isinstance(pt, dict):
        c["f4"] = ({"error": _short(pt["error"], 120)} if "error" in pt else
                   {k: _pair(v, "solve_s", "s per 1000-step solve", frac_unpadded=("roofline", "frac_unpadded"))
                    for k, v in pt.items() if isinstance(v, dict)})
    ps = out.get("projected_strong_scaling")
    if isinstance(ps, dict):
        c["projected_strong_scaling"] = {k: {g: _sig(_get(v, g, "efficiency")) for g in ("2", "4", "8") if g in v}
                                         for k, v in ps.items() if isinstance(v, dict)}
    bc = out.get("stack_broadcast")
    if isinstance(bc, dict) and out.get("n_gpus", 1) > 1:
        c["stack_broadcast"] = {k: _sig(v) if not isinstance(v, str) else _short(v, 60) for k, v in bc.items()
                                if isinstance(v, (int, float, str))}
    c["detail"] = detail
    return c


def emit(out, json_out, root, rank=0):
    """Rank 0: full dict -> bench_detail.json (+ gpurun_out/) and stderr, compact dict -> the LAST stdout line."""
    if rank != 0:
        return None
    text = json.dumps(out)
    written = []
    for d in (root, os.path.join(root, "gpurun_out")):
        if d != root and not os.path.isdir(d):
            continue
        try:
            with open(os.path.join(d, DETAIL_NAME), "w") as f:
                f.write(text + "\n")
            written.append(os.path.join(d, DETAIL_NAME))
        except OSError:
            pass
    print("bench.py detail (%d bytes, also in %s):" % (len(text), ", ".join(written) or "nowhere: not writable"),
          file=sys.stderr)
    print(text, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(out), separators=(",", ":"))
    if len(line) >= MAX_LINE_BYTES:         # never hand the driver a line it cannot hold: drop the optional objects
        c = compact_line(out)
        for k in ("projected_strong_scaling", "f4", "f3", "f2", "dense_expm", "dense_complex", "sharded_cfg5", "cfg4", "cfg2",
                  "cfg5", "end_to_end_solve"):
            c.pop(k, None)
            line = json.dumps(c, separators=(",", ":"))
            if len(line) < MAX_LINE_BYTES:
                break
    print(line, file=json_out, flush=True)
    return line
