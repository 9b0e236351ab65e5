"""Per-launch COMBINE + APPLY kernel at MID sizes (n = 256 / 512 / 1024 random dense imaginary-plane stacks, 8 operators): us per
batched evaluation against the matrix-pipe time of its executed flops, per sweep size.  (Above n_pad = 256 no one-launch
kernel exists: the shape heuristic of launch_combine is what fills the chip.)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points  # noqa: E402

ctx = qd.default_context()
ctx.set_option("combine_sweep", 0)
rng = np.random.default_rng(1)
S = 14
sched = FixedStepSchedule([0.0, 0.1], None, 0.005, _rk4_points)
rows = sched.step_rows[:S]
nr = int(rows.max()) + 1
for n in (256, 512, 1024):
    ops = np.array([1j * rng.uniform(-1, 1, (n, n)) * 0.01 for _ in range(8)])
    st = qd.Stack(ctx, ops, None, None)
    y0 = np.zeros((n, 1), complex)
    y0[0] = 1
    for B in (512, 1024, 2048, 4096, 8192, 16384):
        if n * B > 1024 * 16384:
            continue
        table = rng.uniform(-1, 1, (B, nr, 8))
        p = qd.Rk4Plan(st, sched.times[:nr], table, rows, sched.step_h[:S], y0, B, True)
        p.run(0, 2)
        ctx.synchronize()
        ctx.timer_start()
        p.run(2, S)
        us = ctx.timer_stop() / (4 * (S - 2)) * 1e3
        shape = ctx.counters("combine_shape")
        p.close()
        fmas = 10.0 * n * n * B                                   # 8 plane slots + 2 apply FMAs per element
        pipe_us = fmas / (1024 * 16 * 2.3e9) * 1e6                # 1024 SIMDs x 16 FMA per clock at 2.3 GHz
        print(f"n {n:5d} B {B:6d}: {us:8.1f} us per evaluation, pipe {pipe_us:7.1f} us ({pipe_us / us:.2f}); pairs per workgroup "
              f"{int(shape['launches'])}, list splits {int(shape['ms'])}", flush=True)
    st.close()
