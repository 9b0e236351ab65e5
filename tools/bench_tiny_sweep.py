"""Sweeps of SMALL systems under RK4 (row a9/a15): the persistent tiny_rk4_kernel (whole solve in one
launch, one wave per instance) vs the batched per-stage path.   python tools/bench_tiny_sweep.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd import workloads as W  # noqa: E402

ctx = qd.default_context()
for nq, batch, steps in ((2, 4096, 2000), (3, 4096, 2000), (4, 2048, 1000), (5, 1024, 1000), (2, 1, 2000)):
    cfg = W.schrodinger_config(nq, n_drives=min(nq, 3))
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    rng = np.random.default_rng(nq)
    sigs = [[qd.DiscreteSignal(dt=0.1, samples=rng.uniform(0.1, 1, 20) * np.exp(1j * rng.uniform(0, 1, 20)),
                               carrier_freq=nu, phase=rng.uniform(0, 1)) for nu in cfg["carrier"]] for _ in range(batch)]
    y0 = cfg["y0"]
    out = {"n": 2**nq, "instances": batch, "steps": steps}
    res = {}
    for tag, flag in (("tiny_kernel", 1), ("batched_stages", 0)):
        ctx.set_option("tiny_rk4", flag)
        kw = dict(t_span=[0.0, 2.0], y0=y0, signals=sigs if batch > 1 else sigs[0], method="RK4", max_dt=2.0 / steps)
        solver.solve(**kw)
        t0 = time.perf_counter()
        r = solver.solve(**kw)
        out[tag + "_s"] = round(time.perf_counter() - t0, 4)
        res[tag] = np.array([x.y[-1] for x in (r if isinstance(r, list) else [r])])
    ctx.set_option("tiny_rk4", 1)
    out["max_diff"] = float(np.max(np.abs(res["tiny_kernel"] - res["batched_stages"])))
    out["instance_steps_per_s_tiny"] = round(batch * steps / out["tiny_kernel_s"], 1)
    print(json.dumps(out), flush=True)
