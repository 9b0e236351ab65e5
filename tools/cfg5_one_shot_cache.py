"""One-shot midyn_expm_solve of the cfg 5 shard with and without the plan kept in the stack (ctx option expm_plan_cache), and the
explicit plan beside them: wall clock per solve, interleaved, minimum of 9.      python tools/cfg5_one_shot_cache.py   (on the GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd import workloads as W  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points  # noqa: E402

ctx = qd.default_context(0)
cfg = W.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
ops, static, fim, _ = bench.build_diag_frame_stack(cfg)
stack = qd.Stack(ctx, ops, static, fim)
sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(2))
y0 = cfg["y0"].reshape(-1, 1)
count = 128
tables = [bench.sweep_table(W, sched.times, first, count, 8, cfg["carrier"], cfg["t_final"])[0] for first in (0, 128)]
plan = qd.ExpmPlan(stack, sched.times, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)


def one_shot(t):
    return stack.expm_solve(sched.times, t, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)


best = {"one_shot_cache_off": 1e9, "one_shot_cache_on": 1e9, "plan": 1e9}
ref = [one_shot(t).copy() for t in tables]
for rnd in range(9):
    for tag in best:
        t = tables[rnd & 1]
        if tag == "plan":
            plan.solve(t)
            ctx.synchronize()
            t0 = time.perf_counter()
            r = plan.solve(t)
            dt = time.perf_counter() - t0
        else:
            with ctx.options(expm_plan_cache=1 if tag.endswith("on") else 0):
                one_shot(t)                  # (the option change retired the plan: this call makes it, the timed one finds it)
                ctx.synchronize()
                t0 = time.perf_counter()
                r = one_shot(t)
                dt = time.perf_counter() - t0
                hit = int(ctx.counters("expm_plan_cache")["launches"])
                assert hit == (1 if tag.endswith("on") else 0), (tag, hit)
        assert np.array_equal(r, ref[rnd & 1]), tag
        best[tag] = min(best[tag], dt)
print({k: round(v * 1e6, 1) for k, v in best.items()}, "us per 128-instance, 20-step solve")
