"""Where the wall clock of the complete cfg-3 solve goes (the `end_to_end_solve` leg of bench.py) beyond its 4000 batched
evaluations: cProfile of `Solver.solve` in list mode (4096 instances x 1000 RK4 steps, DiscreteSignal pulses + carrier).

    python tools/profile_end_to_end.py [--instances 4096] [--steps 1000]
"""
import argparse
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--top", type=int, default=30)
    args = ap.parse_args()
    t_final, max_dt = 5.0, 0.005
    t_final = max_dt * args.steps
    cfg = workloads.schrodinger_config(10, 8, t_final, max_dt)
    k = len(cfg["ops"])
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    lists = []
    for b in range(args.instances):
        amps, phs = workloads.sweep_parameters(b, k)
        sl = [qd.Signal(lambda t, a=a: a * np.exp(-((t - t_final / 2) ** 2) / 2.0), nu, ph)
              for a, nu, ph in zip(amps, cfg["carrier"], phs)]
        lists.append([qd.DiscreteSignal.from_Signal(sg, dt=0.05, n_samples=int(round(t_final / 0.05))) for sg in sl])
    for rep in range(2):        # the first solve also builds the stack layouts
        prof = cProfile.Profile()
        t0 = time.perf_counter()
        prof.enable()
        res = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=lists, method="RK4", max_dt=max_dt)
        prof.disable()
        wall = time.perf_counter() - t0
        print(f"solve {rep}: wall {wall:.3f} s, device wall_s of the batch {res[0].wall_s:.3f} s, nfev {res[0].nfev}")
    s = io.StringIO()
    pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(args.top)
    print(s.getvalue())


if __name__ == "__main__":
    main()
