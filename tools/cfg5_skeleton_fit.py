"""cfg 5 shard kernel (ell_flip_duo_kernel<2,2,1024>, 128 instances): is the time that remains with the exchange and all slots
switched off (ablate 13, 2.5 us per term at 20 steps x 9 terms) paid per LAUNCH, per STEP or per TERM?  The same sweep with
10 / 20 / 40 steps of the same length and with 20 steps of half the length (fewer terms per step); least squares for
kernel_us = a + b steps + c terms, complete kernel and skeleton.       python tools/cfg5_skeleton_fit.py   (GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd import workloads  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points  # noqa: E402

ctx = qd.default_context(0)
count = 128
rows = []
for t_final, max_dt in ((5.0, 0.25), (2.5, 0.25), (10.0, 0.25), (2.5, 0.125), (5.0, 0.125), (5.0, 0.5)):
    cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
    ops, static, fim, _ = bench.build_diag_frame_stack(cfg)
    stack = qd.Stack(ctx, ops, static, fim)
    t_span = [2.5 - t_final / 2 if t_final <= 5.0 else 0.0, 2.5 + t_final / 2 if t_final <= 5.0 else t_final]
    sched = FixedStepSchedule(t_span, None, max_dt, _magnus_points(2))
    y0 = cfg["y0"].reshape(-1, 1)
    table, _, _ = bench.sweep_table(workloads, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])

    def run():
        return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)

    res = {}
    for tag, opts in (("complete", {}), ("skeleton", dict(ablate=13)), ("no_exchange", dict(ablate=1))):
        best = 1e9
        for rnd in range(4):
            with ctx.options(**opts):
                cs = bench.profile_pass(ctx, run, ("rk4_resident",))
                terms = ctx.counters("sweep_series")["launches"]
            best = min(best, cs["rk4_resident"]["ms"] * 1e3)
        res[tag] = best
    steps = len(sched.step_h)
    rows.append((steps, terms, res))
    print(f"steps {steps:3d} terms {int(terms):4d}: " + "  ".join(f"{k} {v:8.1f} us ({v / terms:5.2f} per term)" for k, v in res.items()), flush=True)
a = np.array([[1.0, r[0], r[1]] for r in rows])
for tag in ("complete", "skeleton", "no_exchange"):
    y = np.array([r[2][tag] for r in rows])
    sol, resid, _, _ = np.linalg.lstsq(a, y, rcond=None)
    print(f"{tag}: kernel_us = {sol[0]:.1f} + {sol[1]:.2f} x steps + {sol[2]:.3f} x terms   (max residual {np.abs(a @ sol - y).max():.1f} us)", flush=True)
