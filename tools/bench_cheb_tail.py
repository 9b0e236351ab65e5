"""The two truncation rules of the Chebyshev series of the expm action (ctx option cheb_tail: 1 = dropped terms below 2^-53,
0 = every |J_k| >= 1e-18 kept), A/B on the BASELINE configurations that use it: cfg 5 shard (128 instances, n = 4096, Magnus 2,
20 steps) and cfg 4 (N = 4096 vectorised Lindbladian, Magnus 1, 100 steps): series terms, kernel time, and the difference
between the two results.                      python tools/bench_cheb_tail.py     (GPU box)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd import workloads  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points  # noqa: E402

ctx = qd.default_context(0)
cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
ops, static, fim, _ = bench.build_diag_frame_stack(cfg)
stack = qd.Stack(ctx, ops, static, fim)
sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(2))
y0 = cfg["y0"].reshape(-1, 1)
count = 128
table, _, _ = bench.sweep_table(workloads, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])


def run5():
    return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)


res = {}
for tail in (0, 1, 0, 1):
    with ctx.options(cheb_tail=tail):
        run5()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            ys = run5()
            best = min(best, time.perf_counter() - t0)
        cs = bench.profile_pass(ctx, run5, ("rk4_resident",))
        terms = ctx.counters("sweep_series")["launches"]
    res[tail] = ys
    print(json.dumps({"config": "cfg5 shard", "cheb_tail": tail, "terms_per_instance": int(terms), "kernel_ms": round(cs["rk4_resident"]["ms"], 4),
                      "solve_ms": round(best * 1e3, 3), "max_norm_deviation": float(np.abs(np.linalg.norm(ys[:, -1, :, 0], axis=1) - 1).max())}), flush=True)
print(json.dumps({"config": "cfg5 shard", "max_abs_difference_between_the_rules": float(np.abs(res[0] - res[1]).max())}), flush=True)

# cfg 4 through the public Solver (bench.py's leg builds it the same way)
c4 = workloads.lindblad_config()
amps, phases = workloads.sweep_parameters(0, len(c4["ops"]))
sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph) for a, nu, ph in zip(amps, c4["carrier"], phases)]
solver = qd.Solver(static_hamiltonian=c4["h_d"], hamiltonian_operators=c4["ops"], static_dissipators=c4["static_dissipators"],
                   vectorized=True)
y4 = c4["rho0"].flatten(order="F")
out4 = {}
for tail in (0, 1, 0, 1):
    with ctx.options(cheb_tail=tail):
        solver.solve(t_span=c4["t_span"], y0=y4, signals=sigs, method="scipy_expm", max_dt=c4["max_dt"])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r = solver.solve(t_span=c4["t_span"], y0=y4, signals=sigs, method="scipy_expm", max_dt=c4["max_dt"])
            best = min(best, time.perf_counter() - t0)
    out4[tail] = np.asarray(r.y)
    rho = out4[tail][-1].reshape(64, 64, order="F")
    print(json.dumps({"config": "cfg4", "cheb_tail": tail, "solve_ms": round(best * 1e3, 3), "trace_deviation": float(abs(np.trace(rho) - 1.0)),
                      "route": getattr(r, "route", None)}), flush=True)
print(json.dumps({"config": "cfg4", "max_abs_difference_between_the_rules": float(np.abs(out4[0] - out4[1]).max())}), flush=True)
