"""8/9-qubit chains in the diagonal frame (n = 256 / 512), scipy_expm sweeps: the one-launch sweep kernel (256- / 512-thread
workgroups) against the work-list MFMA route."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads as W
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points
from bench import build_diag_frame_stack, sweep_table, ALL_CLASSES
ctx = qd.default_context()
for nq in (8, 9):
    cfg = W.schrodinger_config(n_qubits=nq, n_drives=8, t_final=5.0, max_dt=0.25)
    ops, static, fim, _ = build_diag_frame_stack(cfg)
    stack = qd.Stack(ctx, ops, static, fim)
    y0 = cfg["y0"].reshape(-1, 1)
    for order in (1, 2):
        sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(order))
        for count in (8, 64, 512, 4096):
            table, _, _ = sweep_table(W, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])
            res = {}
            for flag in (1, 0):
                ctx.set_option("ell_sweep", flag)
                run = lambda: stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, order, y0, count, True)
                run(); ctx.synchronize()
                ctx.reset_counters(); ctx.set_option("profile", 1); run(); ctx.set_option("profile", 0)
                # kernel time (HIP events around the launches): the wall clock of the call is dominated by the pageable
                # PCIe copies of the coefficient table and the results, the same on both routes
                res[flag] = sum(ctx.counters(c)["ms"] for c in ALL_CLASSES) / len(sched.step_h)
            print(f"n={2**nq} order {order} {count:5d} instances: sweep kernel {res[1]:.4f} ms of kernels per step, work-list route {res[0]:.4f}", flush=True)
ctx.set_option("ell_sweep", 1)
