"""Small-system Magnus/expm sweep: B instances of a 2-qubit vectorised Lindblad model (N=16)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
ctx = qd.default_context()
for nq, B in ((2, 1024), (3, 512), (4, 128)):
    cfg = workloads.lindblad_config(n_qubits=nq, n_drives=nq, n_diss=nq, gamma=1e-2, t_final=1.0, max_dt=0.01)
    s = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                  static_dissipators=cfg["static_dissipators"], vectorized=True)
    sigs = []
    for b in range(B):
        amps, phs = workloads.sweep_parameters(b, nq)
        sigs.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
                     for a, nu, ph in zip(amps, cfg["carrier"], phs)])
    y0 = cfg["rho0"].flatten(order="F")
    s.solve(t_span=[0, 0.05], y0=y0, signals=sigs[:2], method="scipy_expm", max_dt=0.01)
    t0 = time.perf_counter()
    res = s.solve(t_span=[0, 1.0], y0=y0, signals=sigs, method="scipy_expm", max_dt=0.01)
    dt = time.perf_counter() - t0
    tr = max(abs(np.trace(r.y[-1].reshape(2**nq, 2**nq, order="F")) - 1) for r in res)
    print(f"{nq} qubits (N={4**nq}), B={B}, 100 Magnus-1 steps: {dt:.3f} s  = {B*100/dt:.0f} instance-steps/s  trace err {tr:.1e}", flush=True)
