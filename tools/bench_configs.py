#!/usr/bin/env python
"""Secondary measurements for BASELINE.json configs 2, 4, 5 and the expm kernel alone.
Prints one JSON object per line.  (bench.py is the contract benchmark; this feeds profiles/.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd import workloads  # noqa: E402

ctx = qd.default_context()
which = sys.argv[1:] or ["expm", "cfg4", "cfg5"]


def timed(fn):
    ctx.synchronize()
    t0 = time.perf_counter()
    out = fn()
    ctx.synchronize()
    return out, time.perf_counter() - t0


if "expm_ab" in which:
    for n in (1024, 2048):
        rng = np.random.default_rng(n)
        a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        a = a - a.conj().T
        a *= 2.0 / np.linalg.norm(a, 1)
        for ft in (0, 64, 128):
            ctx.set_option("force_tile", ft)
            ctx.expm(a)
            ctx.reset_counters(); ctx.set_option("profile", 1)
            ctx.expm(a)
            c = ctx.counters("zgemm"); ctx.set_option("profile", 0)
            print(json.dumps({"what": "expm A/B", "n": n, "force_tile": ft, "zgemm_ms": round(c["ms"], 3),
                              "tflops": round(8.0 * n**3 * c["launches"] / (c["ms"] * 1e-3) / 1e12, 2)}), flush=True)
        ctx.set_option("force_tile", 0)

if "expm" in which:
    for n in (1024, 2048, 4096):
        rng = np.random.default_rng(n)
        a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        a = a - a.conj().T
        a *= 2.0 / np.linalg.norm(a, 1)
        ctx.expm(a[:64, :64])  # warm
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        (e, info), dt = timed(lambda: ctx.expm(a, return_info=True))
        c = ctx.counters("zgemm")
        ctx.set_option("profile", 0)
        s = int(info[0, 0])
        flops = 8.0 * n**3 * (6 + s)
        unit = float(np.linalg.norm(e.conj().T @ e - np.eye(n)) / n) if n <= 2048 else None
        print(json.dumps({"what": "expm", "n": n, "norm1": 2.0, "squarings": s, "wall_s_incl_pcie": round(dt, 4),
                          "zgemm_launches": c["launches"], "zgemm_ms": round(c["ms"], 3),
                          "tflops_in_zgemm": round(flops / (c["ms"] * 1e-3) / 1e12, 2),
                          "unitarity_per_n": unit}), flush=True)

def all_counters():
    return {k_: ctx.counters(k_) for k_ in ("rhs_stream", "rhs_gemm", "zgemm", "gen_eval", "elementwise", "rhs_blocks",
                                            "rhs_blocks_gemm")}


def run_modes(label, fn, nsteps, extra):
    """Time `fn` with the expm-action path (default for few columns) and with the dense expm path."""
    modes = (("action", 1, 1), ("action, dense kernels (skip_zero_blocks=0)", 1, 0), ("dense_expm", 0, 1))
    for mode, flag, blocks in modes:
        ctx.set_option("expm_action", flag)
        ctx.set_option("skip_zero_blocks", blocks)
        fn()  # warm (allocations, lazy norms)
        r, dt = timed(fn)   # wall clock without the two event records per launch of the profiling pass
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        fn()
        cs = all_counters()
        ctx.set_option("profile", 0)
        dev_ms = sum(c["ms"] for c in cs.values())
        out = {"what": label, "propagation": mode, "steps": nsteps, "wall_ms_per_step": round(dt * 1e3 / nsteps, 3),
               "device_ms_per_step": round(dev_ms / nsteps, 3),
               "launches_per_step": {k_: c["launches"] / nsteps for k_, c in cs.items() if c["launches"]}}
        out.update(extra(r))
        print(json.dumps(out), flush=True)
    ctx.set_option("expm_action", 1)
    ctx.set_option("skip_zero_blocks", 1)


if "cfg4" in which:
    t0 = time.time()
    cfg = workloads.lindblad_config()  # 6 qubits, N = 4096
    amps, phases = workloads.sweep_parameters(0, 6)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], vectorized=True)
    build_s = time.time() - t0
    nsteps = 5

    def extra4(r):
        rho = r.y[-1].reshape(64, 64, order="F")
        return {"model_build_s": round(build_s, 1), "trace": float(abs(np.trace(rho))),
                "hermiticity": float(np.linalg.norm(rho - rho.conj().T))}

    run_modes("cfg4 (6-qubit vectorised Lindblad, N=4096, scipy_expm m=1, no frame), 1 trajectory",
              lambda: solver.solve(t_span=[0.0, nsteps * cfg["max_dt"]], y0=cfg["rho0"].flatten(order="F"),
                                   signals=sigs, method="scipy_expm", max_dt=cfg["max_dt"]), nsteps, extra4)
    # second run of SURVEY 8(d) cfg 4: diagonal rotating frame diag(H_d)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], rotating_frame=np.diag(cfg["h_d"]).real.copy(),
                       vectorized=True)
    run_modes("cfg4 with the diagonal frame diag(H_d) (N=4096, scipy_expm m=1), 1 trajectory",
              lambda: solver.solve(t_span=[0.0, nsteps * cfg["max_dt"]], y0=cfg["rho0"].flatten(order="F"),
                                   signals=sigs, method="scipy_expm", max_dt=cfg["max_dt"]), nsteps, extra4)
    del solver

if "cfg4sweep" in which:
    # sweeps of the cfg-4 model (N = 4096 superoperators): every Taylor term is one MFMA contraction over the
    # tile lists for all instances
    cfg = workloads.lindblad_config()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], static_dissipators=cfg["static_dissipators"], vectorized=True)
    y0 = cfg["rho0"].flatten(order="F")
    for nb in (16, 64, 256):
        sweeps = []
        for b in range(nb):
            amps, phases = workloads.sweep_parameters(b, 6)
            sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)])
        res = {}
        for blocks, nst in ((1, 100), (0, 4)):
            ctx.set_option("skip_zero_blocks", blocks)
            fn = lambda: solver.solve(t_span=[0.0, nst * 0.05], y0=y0, signals=sweeps, method="scipy_expm", max_dt=0.05)
            fn()
            ctx.synchronize(); t0 = time.perf_counter(); r = fn(); ctx.synchronize(); dt = time.perf_counter() - t0
            rho = r[-1].y[-1].reshape(64, 64, order="F")
            res["work_lists" if blocks else "dense_kernels"] = {"steps": nst, "ms_per_step": round(dt / nst * 1e3, 3), "ms_per_instance_step": round(dt / nst / nb * 1e3, 4), "trace": float(abs(np.trace(rho)))}
        print(json.dumps({"what": f"cfg4 model (N=4096 vectorised Lindbladian, no frame), sweep of {nb} instances, scipy_expm m=1 (expm action)", **res}), flush=True)
    ctx.set_option("skip_zero_blocks", 1)

if "cfg5" in which:
    t0 = time.time()
    cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
    frame = np.diag(cfg["h_d"]).real.copy()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
    build_s = time.time() - t0
    nsteps = 2

    def sig_list(b):
        amps, phases = workloads.sweep_parameters(b, 8)
        return [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
                for a, nu, ph in zip(amps, cfg["carrier"], phases)]

    def extra5(r):
        rr = r if isinstance(r, list) else [r]
        return {"model_build_s": round(build_s, 1),
                "max_norm_deviation": float(max(abs(np.linalg.norm(x.y[-1]) - 1.0) for x in rr))}

    run_modes("cfg5 (12-qubit Schrodinger n=4096, diagonal frame, Magnus-2 expm), 1 instance",
              lambda: solver.solve(t_span=[0.0, nsteps * 0.25], y0=cfg["y0"], signals=sig_list(0), method="scipy_expm",
                                   max_dt=0.25, magnus_order=2), nsteps, extra5)
    # the per-GPU shard of the 1024-instance sweep on 8 GPUs: 128 instances in one batched solve
    shard = [sig_list(b) for b in range(128)]
    ctx.set_option("expm_action", 1)
    for blocks, nst in ((1, 20), (0, 2)):  # the whole 20-step solve on the work-list kernels; 2 steps on the dense ones
        ctx.set_option("skip_zero_blocks", blocks)
        fn = lambda: solver.solve(t_span=[0.0, nst * 0.25], y0=cfg["y0"], signals=shard, method="scipy_expm",
                                  max_dt=0.25, magnus_order=2)
        fn()
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        r, dt = timed(fn)
        cs = all_counters()
        ctx.set_option("profile", 0)
        ctx.reset_counters()
        r, dt_np = timed(fn)  # wall clock without per-launch event timing
        print(json.dumps({"what": "cfg5 shard: 128 instances (1024-instance sweep / 8 GPUs), Magnus-2, expm action",
                          "skip_zero_blocks": blocks, "steps": nst, "wall_ms_per_step": round(dt_np * 1e3 / nst, 2),
                          "wall_ms_per_instance_step": round(dt_np * 1e3 / nst / 128, 4),
                          "device_ms_per_step": round(sum(c["ms"] for c in cs.values()) / nst, 2),
                          "launches_per_step": {k_: c["launches"] / nst for k_, c in cs.items() if c["launches"]},
                          **extra5(r)}), flush=True)
    ctx.set_option("skip_zero_blocks", 1)
    del solver

if "lind1024" in which:
    # 10-qubit open system, non-vectorised: n = 1024, 8 drives, 4 static sigma^- dissipators, RK4
    t0 = time.time()
    cfg = workloads.lindblad_config(n_qubits=10, n_drives=8, n_diss=4, gamma=1e-3, t_final=5.0, max_dt=0.005)
    amps, phases = workloads.sweep_parameters(0, 8)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    m = qd.LindbladModel(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], hamiltonian_signals=sigs,
                         static_dissipators=cfg["static_dissipators"], rotating_frame=np.diag(cfg["h_d"]).real.copy(),
                         vectorized=False)
    build_s = time.time() - t0
    nsteps = 5
    ctx.reset_counters(); ctx.set_option("profile", 1)
    r, dt = timed(lambda: qd.solve_lmde(m, [0.0, nsteps * 0.005], cfg["rho0"], method="RK4", max_dt=0.005))
    cz, cg = ctx.counters("zgemm"), ctx.counters("gen_eval")
    ctx.set_option("profile", 0)
    rho = r.y[-1]
    print(json.dumps({"what": "10-qubit non-vectorised Lindblad (n=1024, 4 dissipators), RK4", "steps": nsteps,
                      "model_build_s": round(build_s, 1), "wall_s": round(dt, 3),
                      "ms_per_rhs_eval_device": round((cz["ms"] + cg["ms"]) / (4 * nsteps), 3),
                      "zgemm_per_eval": cz["launches"] / (4 * nsteps),
                      "zgemm_tflops": round(8 * 1024.0**3 * cz["launches"] / (cz["ms"] * 1e-3) / 1e12, 2),
                      "trace": float(abs(np.trace(rho))), "hermiticity": float(np.linalg.norm(rho - rho.conj().T))}),
          flush=True)

if "lindsweep" in which:
    # sweep of small open systems, non-vectorised (n x n density matrices): every instance advances in the same
    # batched launches (two generator evaluations + 2 + 2 n_diss batched zgemms per RHS evaluation)
    for nq, B, nsteps in ((4, 256, 1000), (6, 256, 400), (8, 64, 200)):
        cfg = workloads.lindblad_config(n_qubits=nq, n_drives=min(nq, 8), n_diss=nq, gamma=1e-2, t_final=nsteps * 0.005,
                                        max_dt=0.005)
        solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                           static_dissipators=cfg["static_dissipators"], rotating_frame=np.diag(cfg["h_d"]).real.copy(),
                           vectorized=False)
        sweeps = []
        for b in range(B):
            amps, phases = workloads.sweep_parameters(b, len(cfg["ops"]))
            sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
                           for a, nu, ph in zip(amps, cfg["carrier"], phases)])
        solver.solve(t_span=[0, 4 * 0.005], y0=cfg["rho0"], signals=sweeps[:2], method="RK4", max_dt=0.005)
        res, dt = timed(lambda: solver.solve(t_span=[0, nsteps * 0.005], y0=cfg["rho0"], signals=sweeps, method="RK4",
                                             max_dt=0.005))
        rho = res[-1].y[-1]
        print(json.dumps({"what": f"{nq}-qubit non-vectorised Lindblad sweep (n={2**nq}, {nq} dissipators), RK4",
                          "instances": B, "steps": nsteps, "wall_s": round(dt, 3),
                          "us_per_instance_rhs_eval": round(dt / (4.0 * nsteps * B) * 1e6, 3),
                          "trace": float(abs(np.trace(rho))),
                          "hermiticity": float(np.linalg.norm(rho - rho.conj().T))}), flush=True)
        del solver

if "unitary1024" in which:
    from bench import build_frame_basis_stack
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points
    cfg = workloads.schrodinger_config()
    ops, static, frame_im = build_frame_basis_stack(cfg)
    stack = qd.Stack(ctx, ops, static, frame_im)
    sched = FixedStepSchedule([0.0, 0.1], None, 0.005, _rk4_points)
    amps, phs = workloads.sweep_parameters(0, 8)
    table = workloads.gaussian_coefficient_table(sched.times, amps[None], phs[None], cfg["carrier"], 5.0)
    y0 = np.eye(1024, dtype=complex)
    for flag in (1, 0):
        ctx.set_option("combine_first", flag)
        stack.rk4_solve(sched.times, table, sched.step_rows[:2], sched.step_h[:2], sched.step_save[:2], 2, y0, 1, True)
        u, dt = timed(lambda: stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                              sched.n_save, y0, 1, True))
        uu = u[0, -1]
        print(json.dumps({"what": "cfg2 model, unitary propagator (y0 = I, m = 1024), 20 RK4 steps",
                          "combine_first": flag, "wall_s_incl_pcie": round(dt, 4),
                          "ms_per_rhs_eval": round(dt / 80 * 1e3, 3),
                          "unitarity": float(np.linalg.norm(uu.conj().T @ uu - np.eye(1024)))}), flush=True)
    ctx.set_option("combine_first", 1)
