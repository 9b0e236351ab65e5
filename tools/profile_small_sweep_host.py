"""Host side of a list-mode solve of a SMALL system (3 transmons, n = 27, 4096 instances x 200 RK4 steps): cProfile of
Solver.solve -- the device part is 2.3 ms, what the caller waits for is the Python around it."""
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import qiskit_dynamics_amd as qd                     # noqa: E402
from qiskit_dynamics_amd import workloads            # noqa: E402

h_d, ops, freqs = workloads.transmon_chain(3, 3)
solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=h_d)
rng = np.random.default_rng(7)
steps, dt, B = 200, 0.005, 4096
t_final = steps * dt
n_smp = 20
lists = [[qd.DiscreteSignal(t_final / n_smp, rng.uniform(0.2, 1.0) * np.hanning(n_smp + 2)[1:-1], carrier_freq=f,
                            phase=rng.uniform(0, 2 * np.pi)) for f in freqs] for _ in range(B)]
y0 = np.zeros(27, dtype=complex)
y0[0] = 1.0
for rep in range(3):
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    prof.enable()
    res = solver.solve(t_span=[0.0, t_final], y0=y0, signals=lists, method="RK4", max_dt=dt)
    prof.disable()
    print(f"solve {rep}: wall {time.perf_counter() - t0:.4f} s, device part {res[0].wall_s:.4f} s")
s = io.StringIO()
pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue())
