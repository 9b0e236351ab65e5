#!/usr/bin/env python
"""End-to-end wall clock of the five BASELINE.json configs through the public API on ONE GPU
(model build, signal evaluation, PCIe and result unpacking included; SURVEY 8(d) inputs).  cfg 3 is
`end_to_end_solve` of bench.py (4096 instances x 1000 steps) and is not repeated here unless asked for."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads

ctx = qd.default_context()
which = sys.argv[1:] or ["cfg1", "cfg2", "cfg4", "cfg5"]


def gauss(a, t_final):
    return lambda t, a=a: a * np.exp(-((t - t_final / 2) ** 2) / 2.0)


def report(what, build_s, solve_s, extra):
    print(json.dumps({"what": what, "model_build_s": round(build_s, 3), "solve_s": round(solve_s, 4), **extra}), flush=True)


def timed(fn):
    ctx.synchronize()
    t0 = time.perf_counter()
    r = fn()
    ctx.synchronize()
    return r, time.perf_counter() - t0


if "cfg1" in which:
    c1 = workloads.config1()
    sigs = [qd.Signal(1.0, 5.0), qd.Signal(lambda t: np.exp(-((t - 5.0) ** 2) / 8.0), 5.0)]
    hm, tb = timed(lambda: qd.HamiltonianModel(static_operator=c1["h_d"], operators=c1["ops"], signals=sigs, rotating_frame=c1["h_d"]))
    fn = lambda: qd.solve_lmde(hm, c1["t_span"], c1["y0"], method="RK4", max_dt=c1["max_dt"])
    fn()
    r, ts = timed(fn)
    report("cfg1: 2 qubits, RK4, 1000 steps", tb, ts, {"norm_deviation": float(abs(np.linalg.norm(r.y[-1]) - 1))})

if "cfg2" in which:
    cfg = workloads.schrodinger_config()
    amps, phases = workloads.sweep_parameters(0, 8)
    sigs = [qd.Signal(gauss(a, 5.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver, tb = timed(lambda: qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"]))
    fn = lambda: solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sigs, method="RK4", max_dt=0.005)
    fn()
    r, ts = timed(fn)
    report("cfg2: 10 qubits (n=1024), 8 drives, RK4, 1000 steps, 1 trajectory", tb, ts,
           {"rhs_evals_per_s": round(4000 / ts), "norm_deviation": float(abs(np.linalg.norm(r.y[-1]) - 1))})
    del solver

if "cfg4" in which:
    cfg = workloads.lindblad_config()
    amps, phases = workloads.sweep_parameters(0, 6)
    sigs = [qd.Signal(gauss(a, 5.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    y0 = cfg["rho0"].flatten(order="F")
    for tag, frame in (("no frame", None), ("diagonal frame", np.diag(cfg["h_d"]).real.copy())):
        solver, tb = timed(lambda: qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                                             static_dissipators=cfg["static_dissipators"], rotating_frame=frame, vectorized=True))
        fn = lambda: solver.solve(t_span=cfg["t_span"], y0=y0, signals=sigs, method="scipy_expm", max_dt=cfg["max_dt"])
        fn()
        r, ts = timed(fn)
        rho = r.y[-1].reshape(64, 64, order="F")
        report(f"cfg4: 6-qubit vectorised Lindblad (N=4096), scipy_expm, 100 steps, {tag}", tb, ts,
               {"ms_per_step": round(ts * 10, 3), "trace": float(abs(np.trace(rho)))})
        del solver

if "cfg5" in which:
    cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
    solver, tb = timed(lambda: qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                                         rotating_frame=np.diag(cfg["h_d"]).real.copy()))
    sweeps = []
    for b in range(1024):
        amps, phases = workloads.sweep_parameters(b, 8)
        sweeps.append([qd.Signal(gauss(a, 5.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    fn = lambda: solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sweeps, method="scipy_expm", max_dt=0.25, magnus_order=2)
    fn()
    r, ts = timed(fn)
    report("cfg5: 12 qubits (n=4096), diagonal frame, Magnus-2 scipy_expm, 20 steps, ALL 1024 instances on one GPU", tb, ts,
           {"ms_per_instance_step": round(ts / (1024 * 20) * 1e3, 4),
            "max_norm_deviation": float(max(abs(np.linalg.norm(x.y[-1]) - 1) for x in r))})
    del solver
