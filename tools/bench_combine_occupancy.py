"""The strong-scaling shards of cfg 3 on the combine + apply route: ctx option combine_occupancy = 1 (one wave per SIMD, at
most 4 waves split the list of a (row group, instance block) pair) against 2 (two waves per SIMD, up to 8 waves per pair,
accumulators added through LDS as a tree).  Prints us per batched evaluation, the workgroup shape and the largest
difference between the two (different summation order: rounding only).

    python tools/bench_combine_occupancy.py [instances ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402
from bench import build_model_stack  # noqa: E402
from qiskit_dynamics_amd import workloads  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points  # noqa: E402

ctx = qd.default_context()
cfg = workloads.schrodinger_config()
ops, static, frame_im, perm = build_model_stack(cfg)
stack = qd.Stack(ctx, ops, static, frame_im)
stack.set_permutation(perm)
sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
S = 42
rows = sched.step_rows[:S]
nr = int(rows.max()) + 1
y0 = cfg["y0"].reshape(-1, 1)
sizes = [int(x) for x in sys.argv[1:]] or [128, 256, 384, 512, 1024, 2048, 4096]
for B in sizes:
    amps = np.array([workloads.sweep_parameters(b, 8)[0] for b in range(B)])
    phs = np.array([workloads.sweep_parameters(b, 8)[1] for b in range(B)])
    table = workloads.gaussian_coefficient_table(sched.times[:nr], amps, phs, cfg["carrier"], 5.0)
    res, outs = {}, {}
    for occ in (1, 2):
        ctx.set_option("combine_occupancy", occ)
        ctx.set_option("combine", 2)
        best = None
        for _ in range(3):
            p = qd.Rk4Plan(stack, sched.times[:nr], table, rows, sched.step_h[:S], y0, B, True)
            p.run(0, 2)
            ctx.synchronize()
            ctx.timer_start()
            p.run(2, S)
            ms = ctx.timer_stop()
            outs[occ] = p.fetch()
            p.close()
            us = ms / (4 * (S - 2)) * 1e3
            best = us if best is None else min(best, us)
        shape = ctx.counters("combine_shape")
        res[occ] = (round(best, 1), int(shape["launches"]), int(shape["ms"]), int(ctx.counters("combine_wave")["launches"]))
    ctx.set_option("combine_occupancy", 2)
    ctx.set_option("combine", 1)
    diff = float(np.max(np.abs(outs[1] - outs[2])))
    print(f"{B:5d} instances: occupancy 1 {res[1][0]:7.1f} us (pairs/workgroup {res[1][1]}, splits {res[1][2]}, {res[1][3]} instances/wave);  "
          f"occupancy 2 {res[2][0]:7.1f} us (pairs/workgroup {res[2][1]}, splits {res[2][2]}, {res[2][3]} instances/wave);  max|diff| {diff:.1e}", flush=True)
