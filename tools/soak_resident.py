"""Soak of the sentinel-ring kernels: long solves (tens of thousands of exchange rounds), resident route against the
launch-per-product route, repeated; any stale or torn exchange would show as a difference far above rounding."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads as W

ctx = qd.default_context()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
# dense resident kernel: cfg 2 model, 20000 RK4 steps = 80000 rounds on 128 workgroups
cfg = W.schrodinger_config(t_final=100.0, max_dt=0.005)
amps, phases = W.sweep_parameters(2, len(cfg["ops"]))
sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 50.0) ** 2) / 800.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)]
solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
ref = None
for rep in range(reps + 1):
    flag = 0 if rep == 0 else 1
    ctx.set_option("resident_rk4", flag)
    t0 = time.perf_counter()
    r = solver.solve(t_span=[0.0, 100.0], y0=cfg["y0"], signals=sigs, method="RK4", max_dt=0.005)
    dt = time.perf_counter() - t0
    if ref is None:
        ref = r.y[-1]
    print(f"cfg2 20000 steps resident={flag}: {dt:.3f} s, |y|-1 = {abs(np.linalg.norm(r.y[-1]) - 1):.2e}, max|diff to per-stage| = {np.max(np.abs(r.y[-1] - ref)):.3e}", flush=True)
ctx.set_option("resident_rk4", 1)
# lane-per-row kernel: cfg 4 Lindbladian, 2000 steps x 36 terms = 72000 rounds on 64 workgroups
cfg = W.lindblad_config(t_final=100.0)
m = qd.LindbladModel(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                     hamiltonian_signals=[qd.Signal(1.0, nu) for nu in cfg["carrier"]],
                     static_dissipators=cfg["static_dissipators"], vectorized=True)
y0 = cfg["rho0"].flatten(order="F")
ref = None
for rep in range(reps + 1):
    flag = 0 if rep == 0 else 1
    ctx.set_option("resident_rk4", flag)
    t0 = time.perf_counter()
    r = qd.solve_lmde(m, [0.0, 100.0], y0, method="scipy_expm", max_dt=0.05)
    dt = time.perf_counter() - t0
    if ref is None:
        ref = r.y[-1]
    rho = r.y[-1].reshape(64, 64, order="F")
    print(f"cfg4 2000 steps resident={flag}: {dt:.3f} s, trace-1 = {abs(np.trace(rho) - 1):.2e}, max|diff to per-launch| = {np.max(np.abs(r.y[-1] - ref)):.3e}", flush=True)
ctx.set_option("resident_rk4", 1)
