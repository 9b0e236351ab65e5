"""Lane-per-row resident kernel: per-product time for RK4 on the chain in its diagonal frame and for the cfg 4
Lindbladian's Chebyshev action, resident on / off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads as W

ctx = qd.default_context()
for nq in (8, 10):
    cfg = W.schrodinger_config(n_qubits=nq, n_drives=min(8, nq), t_final=5.0, max_dt=0.005)
    amps, phases = W.sweep_parameters(1, len(cfg["ops"]))
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=np.diag(cfg["h_d"]).real.copy())
    for flag in (1, 0, 1):
        ctx.set_option("resident_rk4", flag)
        solver.solve(t_span=[0.0, 0.05], y0=cfg["y0"], signals=sigs, method="RK4", max_dt=0.005)
        t0 = time.perf_counter()
        r = solver.solve(t_span=[0.0, 2.0], y0=cfg["y0"], signals=sigs, method="RK4", max_dt=0.005)
        dt = time.perf_counter() - t0
        print(f"chain diag frame n={2**nq} RK4 resident={flag}: {dt / 1600 * 1e6:.2f} us per evaluation (wall, incl. host)", flush=True)
ctx.set_option("resident_rk4", 1)
cfg = W.lindblad_config()
m = qd.LindbladModel(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                     hamiltonian_signals=[qd.Signal(1.0, nu) for nu in cfg["carrier"]],
                     static_dissipators=cfg["static_dissipators"], vectorized=True)
y0 = cfg["rho0"].flatten(order="F")
for flag in (1, 0, 1):
    ctx.set_option("resident_rk4", flag)
    qd.solve_lmde(m, [0.0, 0.5], y0, method="scipy_expm", max_dt=0.05)
    t0 = time.perf_counter()
    r = qd.solve_lmde(m, [0.0, 5.0], y0, method="scipy_expm", max_dt=0.05)
    dt = time.perf_counter() - t0
    print(f"cfg4 scipy_expm resident={flag}: {dt / 100 * 1e3:.4f} ms per step (wall, incl. host)", flush=True)
ctx.set_option("resident_rk4", 1)
print(m.stack.block_info())
