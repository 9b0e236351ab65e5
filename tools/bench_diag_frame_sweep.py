#!/usr/bin/env python
"""The cfg-3 sweep (10-qubit chain, 4096 instances, RK4) in the DIAGONAL frame diag(H_d) instead of the full
frame H_d: same physics (results agree out of the frame), but the operators stay in the computational basis
and are block sparse, so the RHS contraction runs on the work-list kernels.
Timed like bench.py: coefficient table resident, RK4 steps of the whole batch through an Rk4Plan."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.rotating_frame import RotatingFrame
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

ctx = qd.default_context()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
warm, steps = 4, 40
cfg = workloads.schrodinger_config()
frame = RotatingFrame(np.diag(cfg["h_d"]).real.copy())
static = -1j * cfg["h_d"] - np.diag(frame.frame_diag)
stack = qd.Stack(ctx, -1j * cfg["ops"], static, frame.frame_diag_imag)
sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
rows = sched.step_rows[:warm + steps]
nr = int(rows.max()) + 1
amps = np.array([workloads.sweep_parameters(b, 8)[0] for b in range(B)])
phs = np.array([workloads.sweep_parameters(b, 8)[1] for b in range(B)])
table = workloads.gaussian_coefficient_table(sched.times[:nr], amps, phs, cfg["carrier"], 5.0)
y0 = cfg["y0"].reshape(-1, 1)
out = {}
for blocks in (1, 0):
    ctx.set_option("skip_zero_blocks", blocks)
    p = qd.Rk4Plan(stack, sched.times[:nr], table, rows, sched.step_h[:warm + steps], y0, B, True)
    p.run(0, warm); ctx.synchronize()
    t0 = time.perf_counter(); p.run(warm, warm + steps); ctx.synchronize(); dt = time.perf_counter() - t0
    y = p.fetch()[:, :, 0]; p.close()
    out["work_lists" if blocks else "dense_kernels"] = {
        "ms_per_step": round(dt / steps * 1e3, 3), "rhs_evals_per_s": round(4 * B * steps / dt),
        "max_norm_deviation": float(np.max(np.abs(np.linalg.norm(y, axis=1) - 1)))}
    if blocks: ref = y
    else: out["max_abs_difference_between_routes"] = float(np.max(np.abs(y - ref)))
ctx.set_option("skip_zero_blocks", 1)
# the same steps through midyn_rk4_solve: ONE launch of ell_sweep_rk4_kernel (kernel time by HIP events; the call itself
# also uploads the table and copies the results back)
n_st = warm + steps
save = np.full(n_st, -1, dtype=np.int32); save[-1] = 1
run = lambda: stack.rk4_solve(sched.times[:nr], table, rows, sched.step_h[:n_st], save, 2, y0, B, True)
run(); ctx.synchronize()
ctx.reset_counters(); ctx.set_option("profile", 1); ys = run(); ctx.set_option("profile", 0)
c = ctx.counters("rk4_resident")
if c["launches"]:
    out["one_launch_sweep_kernel"] = {"kernel_ms_per_step": round(c["ms"] / n_st, 4),
                                      "rhs_evals_per_s_in_the_kernel": round(4 * B * n_st / (c["ms"] * 1e-3)),
                                      "element_form": int(ctx.counters("sweep_split")["ms"]),
                                      "max_abs_difference_to_work_lists": float(np.max(np.abs(ys[:, -1, :, 0] - ref)))}
    # the same kernel WITH operator elements (option ell_sweep_flip = 0): interleaved, minimum of three
    best = {}
    for rnd in range(3):
        for flag in (1, 0):
            with ctx.options(ell_sweep_flip=flag, profile=1):
                ctx.reset_counters()
                run()
                ms = ctx.counters("rk4_resident")["ms"]
                form = int(ctx.counters("sweep_split")["ms"])
            best[form] = min(best.get(form, 1e9), ms)
    out["one_launch_sweep_kernel"]["rhs_evals_per_s_by_element_form"] = {f: round(4 * B * n_st / (ms * 1e-3)) for f, ms in best.items()}
print(json.dumps({"what": f"cfg3 model in the diagonal frame diag(H_d), {B} instances, RK4 (block-sparse stack), "
                          f"{steps} timed steps, inputs resident", **out}))
