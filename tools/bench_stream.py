"""A/B of rhs_stream_kernel variants on cfg 2 (n=1024, k=8): avg launch time via HIP events."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_frame_basis_stack
ctx = qd.default_context()
cfg = workloads.schrodinger_config()
ops, static, frame_im = build_frame_basis_stack(cfg)
stack = qd.Stack(ctx, ops, static, frame_im)
sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
S = 100
rows = sched.step_rows[:S]; nr = int(rows.max()) + 1
amps, phs = workloads.sweep_parameters(0, 8)
table = workloads.gaussian_coefficient_table(sched.times[:nr], amps[None], phs[None], cfg["carrier"], 5.0)
y0 = cfg["y0"].reshape(-1, 1)
nbytes = 16 * stack.n_segments * 1024 * 1024 + 32 * 1024
for rnd in range(2):
  for planes in (0, 1):
    ctx.set_option("stream_planes", planes)
    for v in (0, 1, 2, 3, 4):
        ctx.set_option("stream_variant", v)
        p = qd.Rk4Plan(stack, sched.times[:nr], table, rows, sched.step_h[:S], y0, 1, True)
        p.run(0, 10); ctx.synchronize()
        ctx.reset_counters(); ctx.set_option("profile", 1)
        p.run(10, S); ctx.synchronize()
        c = ctx.counters("rhs_stream"); ctx.set_option("profile", 0)
        ms = c["ms"] / c["launches"]
        print(f"planes {planes} variant {v}: {ms*1e3:.2f} us  algorithmic {nbytes/ms/1e6:.0f} GB/s", flush=True)
        p.close()
