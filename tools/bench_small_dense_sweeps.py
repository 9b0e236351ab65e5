"""A handful of instances of the cfg 2 / 3 model (n = 1024, full frame), RK4, 1000 steps through Solver.solve: the loop of
register-resident single trajectories (default) against the batched stage (resident_rk4 = 0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads as W

ctx = qd.default_context()
nq = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = W.schrodinger_config(n_qubits=nq, n_drives=min(8, nq))
solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
print("n =", 2**nq)
for nb in (2, 4, 8, 12):
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, len(cfg["ops"]))
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    out = {}
    for flag in (1, 0):
        ctx.set_option("resident_rk4", flag)
        solver.solve(t_span=[0.0, 0.05], y0=cfg["y0"], signals=sweeps, method="RK4", max_dt=0.005)
        t0 = time.perf_counter()
        r = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sweeps, method="RK4", max_dt=0.005)
        out[flag] = (time.perf_counter() - t0, np.stack([x.y[-1] for x in r]))
    ctx.set_option("resident_rk4", 1)
    print(f"{nb:2d} instances x 1000 steps: resident loop {out[1][0]*1e3:.1f} ms, batched stage {out[0][0]*1e3:.1f} ms, "
          f"max |diff| {np.max(np.abs(out[1][1] - out[0][1])):.1e}", flush=True)
