"""Per-launch time of rhs_combine_kernel variants at the headline size (n = 1024, 4096 instances, dense lists): how much do
the static operator's rows (the C input of the combining MFMAs, read 16 lanes per address) cost beside the planes?"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points  # noqa: E402

ctx = qd.default_context()
rng = np.random.default_rng(1)
n, B, S = 1024, 4096, 12


def ops(kinds):
    out = []
    for kd in kinds:
        a = (rng.uniform(-1, 1, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))) * 0.01
        out.append(a.real + 0j if kd == "r" else (1j * a.imag if kd == "i" else a))
    return np.array(out)


sched = FixedStepSchedule([0.0, 0.1], None, 0.005, _rk4_points)
rows = sched.step_rows[:S]
nr = int(rows.max()) + 1
y0 = np.zeros((n, 1), complex)
y0[0] = 1
for kinds, stat in (("iiiiiiii", None), ("iiiiiiii", "i"), ("iiiiiiii", "c"), ("iiii", None), ("iiii", "i"), ("iiii", "c"), ("cccc", None),
                    ("cccc", "c"), ("ii", "i"), ("cc", "c"), ("iiiiiicc", "i"), ("cccccccc", None), ("cccccccc", "c"),
                    ("i" * 12, None), ("i" * 16, None), ("i" * 16, "i"), ("r" * 12, "c"),
                    ("c" * 9, None), ("c" * 12, "c"), ("c" * 8 + "iii", None), ("c" * 13, None)):     # (round 5: a third group beside the other kind; 13: GEMM)
    st = qd.Stack(ctx, ops(kinds), None if stat is None else ops(stat)[0], None)
    table = rng.uniform(-1, 1, (B, nr, len(kinds)))
    res = {}
    for comb in (2, 0):
        ctx.set_option("combine", comb)
        p = qd.Rk4Plan(st, sched.times[:nr], table, rows, sched.step_h[:S], y0, B, True)
        p.run(0, 2)
        ctx.synchronize()
        ctx.timer_start()
        p.run(2, S)
        res[comb] = ctx.timer_stop() / (4 * (S - 2))
        p.close()
    ctx.set_option("combine", 1)
    info = ctx.counters("combine_info")
    print(f"{kinds:16s} static {stat}: combine {res[2]:.3f} ms, GEMM route {res[0]:.3f} ms per batched evaluation", flush=True)
    st.close()
