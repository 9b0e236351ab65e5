"""Sweeps of SMALL systems through the public Solver (list mode): chains of q three-level transmons (n = 3^q) or qubits
(n = 2^q), one drive per site, rotating frame of the static Hamiltonian -- the sizes most pulse-level simulations have.
Prints, per shape: the route the product took (kernel classes with launches), RHS evaluations per second over the device
part of the solve (`OdeResult.wall_s`) and over the whole call.

    python tools/bench_small_sweeps.py [--instances 4096] [--steps 200]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import qiskit_dynamics_amd as qd                     # noqa: E402
from qiskit_dynamics_amd import workloads            # noqa: E402

CLASSES = ("rhs_stream", "rhs_gemm", "zgemm", "gen_eval", "elementwise", "rhs_blocks", "rhs_blocks_gemm", "rk4_resident",
           "rhs_combine")


def chain(levels, sites, seed=0):
    return workloads.transmon_chain(levels, sites, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--method", default="RK4")
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--shapes", default="3x2,3x3,3x4,3x5,2x4,2x6,2x7,2x8")
    args = ap.parse_args()
    dt = args.dt
    t_final = dt * args.steps
    rng = np.random.default_rng(7)
    print(f"{'model':>8} {'n':>5} {'k':>3} {'inst':>6} {'steps':>5} | {'device s':>9} {'call s':>8} {'M evals/s (device)':>19} | route")
    for shape in args.shapes.split(","):
        levels, sites = (int(v) for v in shape.split("x"))
        h_d, ops, freqs = chain(levels, sites)
        n = h_d.shape[0]
        solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=h_d)
        ctx = solver.model._ctx
        n_smp = max(4, int(round(t_final / 0.05)))
        lists = []
        for b in range(args.instances):
            lists.append([qd.DiscreteSignal(t_final / n_smp, rng.uniform(0.2, 1.0) * np.hanning(n_smp + 2)[1:-1],
                                            carrier_freq=f, phase=rng.uniform(0, 2 * np.pi)) for f in freqs])
        y0 = np.zeros(n, dtype=complex)
        y0[0] = 1.0
        def timed(reps):
            devs, calls = [], []
            for _ in range(reps):
                t0 = time.perf_counter()
                r = solver.solve(t_span=[0.0, t_final], y0=y0, signals=lists, method=args.method, max_dt=dt)
                calls.append(time.perf_counter() - t0)
                devs.append(r[0].wall_s)
            return r, devs, calls

        timed(1)                                   # builds the stack layouts
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        res, _, _ = timed(1)
        ctx.synchronize()
        route = {c: ctx.counters(c)["launches"] for c in CLASSES}
        kernel_ms = ctx.counters("rhs_combine")["ms"] if route.get("rhs_combine", 0) == 1 else float("nan")
        ctx.set_option("profile", 0)
        route = {c: v for c, v in route.items() if v}
        res, devs, calls = timed(args.repeats)
        ctx.set_option("combine_sweep", 0)         # the per-launch kernels of the same formulation
        timed(1)
        _, devs0, _ = timed(args.repeats)
        ctx.set_option("combine_sweep", 1)
        dev, call, dev0 = min(devs), min(calls), min(devs0)
        dev_norm = max(abs(np.linalg.norm(r.y[-1]) - 1.0) for r in res[:: max(1, args.instances // 16)])
        evals = args.instances * 4 * args.steps
        print(f"{shape:>8} {n:5d} {len(ops):3d} {args.instances:6d} {args.steps:5d} | {dev:9.4f} {call:8.3f} {evals / dev / 1e6:19.2f} | "
              f"us/stage {dev / (4 * args.steps) * 1e6:7.2f} (median {sorted(devs)[len(devs) // 2] / (4 * args.steps) * 1e6:7.2f}; "
              f"per-launch route {dev0 / (4 * args.steps) * 1e6:7.2f}; one-launch KERNEL alone {kernel_ms / (4 * args.steps) * 1e3:7.2f}) {route}  |norm-1| {dev_norm:.1e}")


if __name__ == "__main__":
    main()
