"""Batched RHS evaluation vs batch size (the strong-scaling shards of cfg 3): split-K reduced inside the launch
(default) against the separate reduction kernel (splitk_inlaunch=0); results must be bit-identical."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402
from bench import build_frame_basis_stack, build_model_stack  # noqa: E402
from qiskit_dynamics_amd import workloads  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points  # noqa: E402

ctx = qd.default_context()
cfg = workloads.schrodinger_config()
if os.environ.get("MIDYN_DENSE_STACK"):      # the reference's eigenvector order: scattered exact zeros, dense kernels
    ops, static, frame_im = build_frame_basis_stack(cfg)
    stack = qd.Stack(ctx, ops, static, frame_im)
else:                                        # as HamiltonianModel uploads it: grouped by symmetry sector, work lists
    ops, static, frame_im, perm = build_model_stack(cfg)
    stack = qd.Stack(ctx, ops, static, frame_im)
    stack.set_permutation(perm)
sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
S = 22
rows = sched.step_rows[:S]
nr = int(rows.max()) + 1
y0 = cfg["y0"].reshape(-1, 1)
sizes = [int(x) for x in sys.argv[1:]] or [64, 128, 256, 512, 1024, 2048, 4096]
for B in sizes:
    amps = np.array([workloads.sweep_parameters(b, 8)[0] for b in range(B)])
    phs = np.array([workloads.sweep_parameters(b, 8)[1] for b in range(B)])
    table = workloads.gaussian_coefficient_table(sched.times[:nr], amps, phs, cfg["carrier"], 5.0)
    res, ref = {}, None
    for tag, flag in (("inlaunch", 1), ("reduce_kernel", 0)):
        ctx.set_option("splitk_inlaunch", flag)
        p = qd.Rk4Plan(stack, sched.times[:nr], table, rows, sched.step_h[:S], y0, B, True)
        p.run(0, 2)
        ctx.synchronize()
        ctx.timer_start()
        p.run(2, S)
        ms = ctx.timer_stop()
        out = p.fetch()
        p.close()
        ref = out if ref is None else ref
        res[tag] = (ms / (4 * (S - 2)) * 1e3, bool(np.array_equal(out, ref)))
    ctx.set_option("splitk_inlaunch", 1)
    print(B, {k: (round(v[0], 1), v[1]) for k, v in res.items()}, "us per batched eval;",
          f"{B / res['inlaunch'][0] * 1e6:.0f} evals/s", flush=True)
