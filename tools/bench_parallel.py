"""Parallel-in-time (row f3) vs sequential fixed-step solves of ONE trajectory: wall-clock per solve.
    python tools/bench_parallel.py            (on the GPU box; PYTHONPATH=.)"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd import workloads as W  # noqa: E402


def run(nq, nsteps, methods):
    cfg = W.schrodinger_config(nq) if nq > 2 else None
    if cfg is None:
        c1 = W.config1()
        h_d, ops = c1["h_d"], c1["ops"]
    else:
        h_d, ops = cfg["h_d"], cfg["ops"]
    n = h_d.shape[0]
    solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=h_d)
    sigs = [qd.Signal(lambda t, j=j: 0.2 * np.cos(0.3 * t + j) + 0j, 4.0 + 0.1 * j, 0.0) for j in range(len(ops))]
    y0 = np.zeros(n, dtype=complex)
    y0[0] = 1.0
    t_final = 1.0
    out = {"n": n, "steps": nsteps}
    ref = None
    for name, kw in methods:
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r = solver.solve(t_span=[0.0, t_final], y0=y0, signals=sigs, method=name, max_dt=t_final / nsteps, **kw)
            best = min(best, time.perf_counter() - t0)
        tag = name + ("" if not kw else f"_m{kw['magnus_order']}")
        out[tag + "_s"] = round(best, 4)
        if ref is None:
            ref = r.y[-1]
        out[tag + "_diff"] = float(np.max(np.abs(r.y[-1] - ref)))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    M = [("scipy_expm", {"magnus_order": 1}), ("hip_expm_parallel", {"magnus_order": 1}),
         ("RK4", {}), ("hip_RK4_parallel", {})]
    run(2, 2000, M)
    run(5, 2000, M)
    run(6, 2000, M)
    run(7, 1000, M)
    run(8, 500, M)
    run(9, 200, M)
