"""cProfile of the model build of the 10-qubit (cfg 2 / 3) Solver: where the host time of a single solve goes."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads as W
ctx = qd.default_context()
cfg = W.schrodinger_config()
qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
s = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
pr.disable()
print("build", time.perf_counter() - t0)
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(18); print(st.getvalue()[:3500])
