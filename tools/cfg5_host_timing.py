"""Where the wall clock of a cfg 5 shard solve goes on the host (MIDYN_TIMING=1 prints the marks of expm_action_solve).
    MIDYN_TIMING=1 python tools/cfg5_host_timing.py          (on the GPU box)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points
ctx = qd.default_context(0)
cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
ops, static, fim, _ = bench.build_diag_frame_stack(cfg)
stack = qd.Stack(ctx, ops, static, fim)
sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(2))
y0 = cfg["y0"].reshape(-1, 1)
count = 128
table, _, _ = bench.sweep_table(workloads, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])
def run():
    return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)
for i in range(4):
    t0 = time.perf_counter(); ys = run(); t1 = time.perf_counter()
    print("solve wall %.1f us" % ((t1 - t0) * 1e6), ys.shape, file=sys.stderr, flush=True)
# ... and the same shard after a 1024-instance solve of the same stack (the order of bench.py's projection leg)
table_all, _, _ = bench.sweep_table(workloads, sched.times, 0, 1024, 8, cfg["carrier"], cfg["t_final"])
stack.expm_solve(sched.times, table_all, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, 1024, True)
print("--- after the 1024-instance solve", file=sys.stderr, flush=True)
for i in range(4):
    t0 = time.perf_counter(); ys = run(); t1 = time.perf_counter()
    print("solve wall %.1f us" % ((t1 - t0) * 1e6), ys.shape, file=sys.stderr, flush=True)
# ... and through the plan object (midyn_expm_plan_*): tables, buffers and the result block's route made once
print("--- plan object: create once, then run + fetch per solve", file=sys.stderr, flush=True)
t0 = time.perf_counter()
plan = qd.ExpmPlan(stack, sched.times, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)
print("plan create %.1f us" % ((time.perf_counter() - t0) * 1e6), file=sys.stderr, flush=True)
for direct in (1, 0):
    with ctx.options(expm_direct_out=direct):
        for i in range(5):
            t0 = time.perf_counter(); ctx.timer_start(); plan.run(table); t1 = time.perf_counter(); ys2 = plan.fetch(); ev = ctx.timer_stop(); t2 = time.perf_counter()
            print("direct_out=%d  plan run %.1f us  fetch %.1f us  total %.1f us  (stream %.1f us)  equal %s" % (
                direct, (t1 - t0) * 1e6, (t2 - t1) * 1e6, (t2 - t0) * 1e6, ev * 1e3, np.array_equal(ys2, ys)), file=sys.stderr, flush=True)
        for i in range(3):
            t0 = time.perf_counter(); ys3 = run(); t1 = time.perf_counter()
            print("direct_out=%d  one-shot solve wall %.1f us" % (direct, (t1 - t0) * 1e6), file=sys.stderr, flush=True)
