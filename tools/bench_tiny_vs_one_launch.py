import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
rng = np.random.default_rng(7)
for levels, sites in ((3, 2), (2, 4), (4, 2), (5, 2), (2, 5)):
    h_d, ops, freqs = workloads.transmon_chain(levels, sites)
    n = h_d.shape[0]
    solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=h_d)
    ctx = solver.model._ctx
    steps, dt = 200, 0.005
    tf = steps * dt
    for B in (256, 4096, 32768):
        lists = [[qd.DiscreteSignal(tf / 20, rng.uniform(0.2, 1.0) * np.hanning(22)[1:-1], carrier_freq=f, phase=rng.uniform(0, 6)) for f in freqs] for _ in range(B)]
        y0 = np.zeros(n, complex); y0[0] = 1
        out = {}
        for tiny in (1, 0):
            ctx.set_option("tiny_rk4", tiny)
            devs = []
            for _ in range(4):
                r = solver.solve(t_span=[0, tf], y0=y0, signals=lists, method="RK4", max_dt=dt)
                devs.append(r[0].wall_s)
            out[tiny] = (min(devs[1:]), r[0].y[-1].copy())
        ctx.set_option("tiny_rk4", 1)
        print(f"n {n:3d} k {len(ops)} B {B:6d}: tiny kernel {out[1][0]*1e3:8.3f} ms, without {out[0][0]*1e3:8.3f} ms  diff {np.max(np.abs(out[1][1]-out[0][1])):.1e}", flush=True)
