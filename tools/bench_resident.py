"""Single trajectory of the cfg 2 model (n = 1024, 8 drives, RK4): register-resident kernel vs the launch-per-stage
streaming route -- per-evaluation time and the difference between the two final states."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points
from bench import build_model_stack, build_frame_basis_stack

ctx = qd.default_context()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = workloads.schrodinger_config(n_qubits=nq, n_drives=min(8, nq))
if os.environ.get("MIDYN_DENSE_STACK"):
    ops, static, frame_im = build_frame_basis_stack(cfg); perm = None
else:
    ops, static, frame_im, perm = build_model_stack(cfg)
stack = qd.Stack(ctx, ops, static, frame_im)
stack.set_permutation(perm)
sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
S = 400
rows = sched.step_rows[:S]; nr = int(rows.max()) + 1
k = ops.shape[0]
amps, phs = workloads.sweep_parameters(0, k)
table = workloads.gaussian_coefficient_table(sched.times[:nr], amps[None], phs[None], cfg["carrier"], 5.0)
y0 = cfg["y0"].reshape(-1, 1)
out = {}
for tag, flag in (("resident", 1), ("per-stage", 0), ("resident again", 1)):
    ctx.set_option("resident_rk4", flag)
    p = qd.Rk4Plan(stack, sched.times[:nr], table, rows, sched.step_h[:S], y0, 1, True)
    p.run(0, 20); ctx.synchronize()
    t0 = time.perf_counter(); p.run(20, S); ctx.synchronize(); dt = time.perf_counter() - t0
    out[tag] = p.fetch(); p.close()
    print(f"{tag:15s}: {dt / (4 * (S - 20)) * 1e6:7.2f} us per RHS evaluation   |y| = {np.linalg.norm(out[tag]):.15f}", flush=True)
ctx.set_option("resident_rk4", 1)
print("max |resident - per-stage| =", float(np.max(np.abs(out["resident"] - out["per-stage"]))),
      " resident run-to-run:", float(np.max(np.abs(out["resident"] - out["resident again"]))))
