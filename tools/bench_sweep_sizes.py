"""cfg 5 model (n = 4096, diagonal frame, Magnus-2 scipy_expm, 20 steps) at small sweep sizes: one workgroup per instance,
four workgroups per instance (ell_sweep_split) and the launch-per-product route."""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads as W
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points
from bench import build_diag_frame_stack, sweep_table
ctx = qd.default_context()
cfg = W.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
ops, static, fim, _ = build_diag_frame_stack(cfg)
stack = qd.Stack(ctx, ops, static, fim)
sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(2))
y0 = cfg["y0"].reshape(-1, 1)
for count in (1, 2, 8, 32, 64, 128):
    table, _, _ = sweep_table(W, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])
    for opts in ({"ell_sweep": 1, "ell_sweep_split": 0}, {"ell_sweep": 1, "ell_sweep_split": 1}, {"ell_sweep": 0, "ell_sweep_split": 0}):
        for k_, v_ in opts.items(): ctx.set_option(k_, v_)
        run = lambda: stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)
        run(); ctx.synchronize()
        t0 = time.perf_counter(); ctx.timer_start(); ys = run(); dev = ctx.timer_stop(); wall = time.perf_counter() - t0
        print(count, opts, f"device {dev/20:.4f} ms per step, wall {wall/20*1e3:.4f}", flush=True)
ctx.set_option("ell_sweep", 1); ctx.set_option("ell_sweep_split", 1)
