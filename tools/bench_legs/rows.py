"""bench.py legs: SURVEY section 8 rows a11 (dense expm), f2 (non-vectorised Lindblad), f3 (parallel in time), f4 (perturbative)."""
import os
import time

import numpy as np

from .common import (ALL_CLASSES, CFG5_SWEEP, FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, LDS_PEAK_GBS, MAX_DT, N_DRIVES, N_QUBITS, ROOT, SWEEP,  # noqa: F401
                     T_FINAL, ZGEMM_NOTE, _mfma_roofline, build_diag_frame_stack, build_frame_basis_stack, build_model_stack,
                     measured_traffic, profile_pass, sweep_table)


def leg_dense_expm(qd, ctx):
    """Row a11: the dense matrix exponential `midyn_expm` (scaling and squaring of a Taylor polynomial on the MFMA zgemm;
    the reference calls scipy.linalg.expm, solvers/fixed_step_solvers.py:22,104) at the sizes BASELINE names: n = 1024 and
    n = 4096, complex128, anti-Hermitian matrices of 1-norm 5 (a Magnus step of cfg 4 has 2.7) -- wall clock of the call
    (PCIe both ways included), kernel time and executed flops from the library's counters, scipy on this host beside it."""
    import scipy.linalg
    from threadpoolctl import threadpool_limits

    out = {}
    rng = np.random.default_rng(11)
    for n in (1024, 4096):
        a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        a = a - a.conj().T
        a *= 5.0 / np.abs(a).sum(axis=0).max()
        ctx.expm(a)                                           # warm-up: workspace, clocks
        t0 = time.perf_counter()
        e, info = ctx.expm(a, return_info=True)
        wall = time.perf_counter() - t0
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        try:
            ctx.expm(a)
            ctx.synchronize()
            cz, ce = ctx.counters("zgemm"), ctx.counters("elementwise")
            flops = ctx.executed_flops("zgemm")
        finally:
            ctx.set_option("profile", 0)
        threads = min(os.cpu_count() or 8, 64)
        with threadpool_limits(limits=threads):
            t0 = time.perf_counter()
            ref = scipy.linalg.expm(a)
            cpu_s = time.perf_counter() - t0
        err = float(np.abs(e - ref).sum(axis=0).max() / np.abs(ref).sum(axis=0).max())
        unit = float(np.abs(e.conj().T @ e - np.eye(n)).max()) if n <= 1024 else None
        products = int(cz["launches"])
        out[f"n{n}"] = {
            "workload": f"expm of one {n} x {n} complex128 anti-Hermitian matrix, ||A||_1 = 5",
            "wall_s_host_to_host": round(wall, 4), "kernel_ms": round(cz["ms"] + ce["ms"], 3), "matrix_products": products,
            "squarings": int(info[0][0]), "elementwise_ms": round(ce["ms"], 3),
            "expm_per_s_kernels": round(1e3 / (cz["ms"] + ce["ms"]), 2),
            "rel_1norm_error_vs_scipy": err, "unitarity_defect": unit,
            "roofline": _mfma_roofline("zgemm_seg_kernel<64, 64, 2, 2, 16, 4> (3M dense complex product)", flops, cz["ms"],
                                       ZGEMM_NOTE, launches=products),
            "cpu_baseline": {"value": round(1.0 / cpu_s, 4), "unit": "expm/s", "s_per_expm": round(cpu_s, 3), "cores": threads,
                             "kind": "reference", "sample": f"scipy.linalg.expm (what the reference calls) of the same matrix "
                                                            f"on this host, {threads} BLAS threads: {cpu_s:.2f} s"},
            "pcie_note": "wall_s_host_to_host includes 2 x 16 n^2 bytes over PCIe and the host-side padding copy; "
                         "inside midyn_expm_solve the matrices never leave the device"}
    return out


def leg_lindblad_rk4(qd, ctx, workloads, n_qubits=10, instances=64, steps=5):
    """Row f2: NON-vectorised Lindblad RK4 (LindbladCollection.evaluate_rhs, models/operator_collections.py:451-567:
    (A + B) rho + rho (A - B) + sum gamma L rho L^+ with n x n products, no n^2 x n^2 superoperator) -- the 10-qubit chain
    (n = 1024) with 4 static dissipators in the frame of its static Hamiltonian, a sweep of density-matrix trajectories
    through the product Solver (list mode)."""
    from oracle import dynamics_oracle as orc
    from threadpoolctl import threadpool_limits

    cfg = workloads.lindblad_config(n_qubits=n_qubits, n_drives=8, n_diss=4, gamma=1e-3, t_final=5.0, max_dt=0.005)
    n = cfg["h_d"].shape[0]
    t0 = time.perf_counter()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], rotating_frame=cfg["h_d"], vectorized=False)
    build_s = time.perf_counter() - t0
    sweeps = []
    for b in range(instances):
        amps, phases = workloads.sweep_parameters(b, 8)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    t_span = [2.4, 2.4 + steps * cfg["max_dt"]]
    solver.solve(t_span=[2.4, 2.4 + cfg["max_dt"]], y0=cfg["rho0"], signals=sweeps[:2], method="RK4", max_dt=cfg["max_dt"])
    t0 = time.perf_counter()
    res = solver.solve(t_span=t_span, y0=cfg["rho0"], signals=sweeps, method="RK4", max_dt=cfg["max_dt"])
    wall = time.perf_counter() - t0
    dev_s = res[0].wall_s
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        solver.solve(t_span=[2.4, 2.4 + cfg["max_dt"]], y0=cfg["rho0"], signals=sweeps[:4], method="RK4", max_dt=cfg["max_dt"])
        ctx.synchronize()
        cz, cg, ce = ctx.counters("zgemm"), ctx.counters("gen_eval"), ctx.counters("elementwise")
        flops = ctx.executed_flops("zgemm")
    finally:
        ctx.set_option("profile", 0)
    evals = instances * 4 * steps
    rho = res[-1].y[-1]
    # CPU: the oracle's matrix-form RHS (lindblad_rhs) of the same model in the same frame, one instance, a few evaluations
    h_d, h_ops, n_static, l_ops, d, basis = orc.lindblad_model_build(cfg["h_d"], cfg["ops"], cfg["static_dissipators"], None,
                                                                      cfg["h_d"])
    amps, phases = workloads.sweep_parameters(0, 8)
    rho_f = basis.conj().T @ cfg["rho0"] @ basis
    threads = min(os.cpu_count() or 8, 64)
    n_cpu = 4
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        for i in range(n_cpu):
            t = 2.4 + 0.0025 * i
            c = workloads.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], cfg["t_final"])[0]
            orc.lindblad_rhs(h_d, h_ops, n_static, l_ops, c, None, d, t, rho_f)
        cpu_s = (time.perf_counter() - t0) / n_cpu
    return {
        "workload": f"{n_qubits}-qubit (n = {n}) Lindblad master equation, 8 drives, 4 static sigma^- dissipators, frame of H_d, "
                    f"vectorized=False, RK4 max_dt 0.005: {instances} density-matrix trajectories x {steps} steps",
        "rhs_evals_per_s": round(evals / dev_s, 2), "ms_per_instance_evaluation": round(dev_s / evals * 1e3, 4),
        "solve_s_device_call": round(dev_s, 3), "solve_s_whole_call": round(wall, 3), "model_build_s": round(build_s, 2),
        "trace_deviation": float(abs(np.trace(rho) - 1.0)), "hermiticity": float(np.linalg.norm(rho - rho.conj().T)),
        "kernel_ms_per_evaluation": {"zgemm": round(cz["ms"] / 16, 4), "gen_eval": round(cg["ms"] / 16, 4),
                                     "elementwise": round(ce["ms"] / 16, 4)},
        "products_per_evaluation": round(cz["launches"] / 16, 2),
        "roofline": _mfma_roofline("zgemm_seg_kernel (n x n products of the non-vectorised Lindblad right-hand side)", flops,
                                   cz["ms"], ZGEMM_NOTE + "; profiled pass: 4 instances x 1 step = 16 evaluations",
                                   launches=int(cz["launches"])),
        "cpu_baseline": {"value": round(1.0 / cpu_s, 3), "unit": "RHS evals/s", "cores": threads, "kind": "port",
                         "sample": f"{n_cpu} evaluations of oracle.lindblad_rhs (the reference's matrix form, NumPy matmul) of the "
                                   f"same model on this host, {threads} BLAS threads: {cpu_s * 1e3:.1f} ms each"}}


def leg_parallel_in_time(qd, ctx, workloads, n_qubits=4, steps=1000):
    """Row f3: parallel-in-time propagation (fixed_step_lmde_solver_parallel_template_jax, solvers/fixed_step_solvers.py:
    524-613): all step propagators by batched launches, then a binary-tree product.  cfg 4's model class -- the vectorised
    Lindbladian of the qubit chain with sigma^- dissipators, scipy_expm Magnus order 1 -- at 4 qubits (N = 256), 1000 steps.
    (At cfg 4's own N = 4096 the 1000 step propagators are 268 GB: that size keeps the sequential expm action.)"""
    import scipy.linalg
    from oracle import dynamics_oracle as orc
    from threadpoolctl import threadpool_limits

    cfg = workloads.lindblad_config(n_qubits=n_qubits, n_drives=n_qubits, n_diss=min(4, n_qubits), gamma=1e-3, t_final=5.0,
                                    max_dt=5.0 / steps)
    n = cfg["h_d"].shape[0]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], vectorized=True)
    amps, phases = workloads.sweep_parameters(0, n_qubits)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)]

    y0_vec = cfg["rho0"].flatten(order="F")      # (a vectorised model takes the column-stacked density matrix)

    def solve(method):
        best, r_ = 1e9, None
        for _ in range(3):
            t0 = time.perf_counter()
            r_ = solver.solve(t_span=cfg["t_span"], y0=y0_vec, signals=sigs, method=method, max_dt=cfg["max_dt"])
            best = min(best, time.perf_counter() - t0)
        return best, r_

    t_par, r_par = solve("hip_expm_parallel")
    t_seq, r_seq = solve("scipy_expm")
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        solver.solve(t_span=cfg["t_span"], y0=y0_vec, signals=sigs, method="hip_expm_parallel", max_dt=cfg["max_dt"])
        ctx.synchronize()
        cz, cg, ce = ctx.counters("zgemm"), ctx.counters("gen_eval"), ctx.counters("elementwise")
        flops = ctx.executed_flops("zgemm")
    finally:
        ctx.set_option("profile", 0)
    # CPU: the reference's sequential loop (generator by tensordot, scipy.linalg.expm, matvec) on the superoperators, 20 steps
    s_d, s_ops = orc.vectorized_lindblad_stack(cfg["h_d"], cfg["ops"], cfg["static_dissipators"], None)
    threads = 8
    n_cpu = 20

    def gen(t):
        c = workloads.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], cfg["t_final"])[0]
        return orc.generator_evaluate(s_d, s_ops, c, None, None, t)

    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        y = cfg["rho0"].flatten(order="F")
        for i in range(n_cpu):
            y = scipy.linalg.expm(orc.magnus_terms(gen, 2.4 + i * cfg["max_dt"], cfg["max_dt"], 1)) @ y
        cpu_s = (time.perf_counter() - t0) / n_cpu
    rho = r_par.y[-1].reshape(n, n, order="F")
    return {
        "workload": f"{n_qubits}-qubit vectorised Lindbladian (N = {n * n}), {n_qubits} drives, {len(cfg['static_dissipators'])} "
                    f"dissipators, no frame, scipy_expm magnus_order 1, {steps} steps, ONE trajectory",
        "steps_per_s_parallel_in_time": round(steps / t_par, 1), "solve_s_parallel_in_time": round(t_par, 4),
        "solve_s_sequential_device_route": round(t_seq, 4), "route": getattr(r_par, "route", None),
        "max_abs_difference_between_the_routes": float(np.max(np.abs(r_par.y[-1] - r_seq.y[-1]))),
        "trace_deviation": float(abs(np.trace(rho) - 1.0)),
        "kernel_ms": {"zgemm": round(cz["ms"], 3), "gen_eval": round(cg["ms"], 3), "elementwise": round(ce["ms"], 3)},
        "roofline": _mfma_roofline("zgemm_seg_kernel (batched N x N products: Taylor blocks and squarings of every step's expm, "
                                   "then the tree of step propagators)", flops, cz["ms"], ZGEMM_NOTE, launches=int(cz["launches"])),
        "cpu_baseline": {"value": round(1.0 / cpu_s, 2), "unit": "steps/s", "cores": threads, "kind": "port",
                         "sample": f"{n_cpu} steps of the reference's sequential loop with the NumPy oracle (tensordot generator, "
                                   f"scipy.linalg.expm, matvec) on the N = {n * n} superoperators, {threads} BLAS threads: "
                                   f"{cpu_s * 1e3:.1f} ms per step"}}


def leg_perturbative(qd, ctx):
    """Row f4: MagnusSolver / DysonSolver (solvers/perturbative_solvers/magnus_solver.py:107-129, perturbation/
    array_polynomial.py:524-544) on the two-transmon model of the reference's own test (dim 25, two drives; tests/
    bench_perturbative_vs_oracle.py holds the same model): 1000 steps of dt = 0.01, y0 = identity."""
    from oracle import dynamics_oracle as orc

    w_c, w_t = 2 * np.pi * 5.033, 2 * np.pi * 4.067
    alpha_c, alpha_t, jc = 2 * np.pi * (-0.33534), 2 * np.pi * (-0.33834), 2 * np.pi * 0.002
    dim = 5
    a = np.diag(np.sqrt(np.arange(1, dim)), 1)
    num = np.diag(np.arange(dim)).astype(float)
    i1, i2 = np.eye(dim), np.eye(dim**2)
    a0, a1 = np.kron(a, i1), np.kron(i1, a)
    n0, n1 = np.kron(num, i1), np.kron(i1, num)
    h0 = w_c * n0 + 0.5 * alpha_c * n0 @ (n0 - i2) + w_t * n1 + 0.5 * alpha_t * n1 @ (n1 - i2) + jc * (a0 @ a1.T + a0.T @ a1)
    hdc, hdt = 2 * np.pi * (a0 + a0.T), 2 * np.pi * (a1 + a1.T)
    sig_w = 0.399128 / 0.2
    gauss = qd.Signal(lambda t: np.exp(-((t - 3.5 * sig_w) ** 2) / (2 * sig_w**2)), carrier_freq=5.0)
    dt, n_steps = 0.01, 1000
    y0 = np.eye(dim**2, dtype=complex)
    out = {}
    for name, cls, order in (("magnus", qd.MagnusSolver, 3), ("dyson", qd.DysonSolver, 4)):
        t0 = time.perf_counter()
        sol = cls(operators=[-1j * hdc, -1j * hdt], rotating_frame=-1j * h0, dt=dt, carrier_freqs=[5.0, 5.0],
                  chebyshev_orders=[1, 1], expansion_order=order, integration_method="DOP853", atol=1e-10, rtol=1e-10)
        build_s = time.perf_counter() - t0
        sol.solve(t0=0.0, n_steps=64, y0=y0, signals=[gauss, gauss])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            yf = sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=[gauss, gauss]).y[-1]
            best = min(best, time.perf_counter() - t0)
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        try:
            sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=[gauss, gauss])
            ctx.synchronize()
            cz, ce = ctx.counters("zgemm"), ctx.counters("elementwise")
            flops = ctx.executed_flops("zgemm")
            per_block = int(ctx.counters("expansion_pack")["launches"])
        finally:
            ctx.set_option("profile", 0)
        with ctx.options(expansion_pack=0):          # one step per padded block (rounds 1-5), beside it
            y_one = sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=[gauss, gauss]).y[-1]
            best_one = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=[gauss, gauss])
                best_one = min(best_one, time.perf_counter() - t0)
        m = sol.model
        coeffs = m.approximate_signals([gauss, gauss], 0.0, n_steps)
        labels = np.array([list(lab) + [-1] * (order - len(lab)) for lab in m.monomial_labels])
        d, basis = orc.frame_setup(-1j * h0)
        n_cpu = 50
        t0 = time.perf_counter()
        orc.perturbative_solve(name, m.array_coefficients, labels, m.Udt, d, basis, coeffs[:, :n_cpu], y0, 0.0, n_cpu, dt)
        cpu_s = (time.perf_counter() - t0) / n_cpu
        out[name] = {
            "workload": f"{cls.__name__} expansion_order {order}, two 5-level transmons (dim 25), {len(m.monomial_labels)} expansion "
                        f"terms, {n_steps} steps of dt {dt}, y0 = identity",
            "steps_per_s": round(n_steps / best, 1), "solve_s": round(best, 5), "model_build_s": round(build_s, 2),
            "unitarity_defect": float(np.abs(yf.conj().T @ yf - np.eye(dim**2)).max()),
            "kernel_ms": {"zgemm": round(cz["ms"], 3), "elementwise": round(ce["ms"], 3)},
            "steps_per_padded_block": per_block,
            "one_step_per_block": {"solve_s": round(best_one, 5), "max_abs_difference": float(np.max(np.abs(y_one - yf)))},
            "roofline": _mfma_roofline("zgemm_seg_kernel (the polynomial of all steps as ONE product mono[T][M] x terms[M][n_pad^2], "
                                       "batched expm and Udt products, tree of step maps)", flops, cz["ms"],
                                       ZGEMM_NOTE + "; dim 25 pads to 64: the launches are latency-bound at this size, the "
                                       "roofline fraction says so", launches=int(cz["launches"]),
                                       # the cubic part (expm, Udt and tree products) of the EXECUTED flops that is not padding:
                                       # blocks of 64 / steps_per_padded_block rows hold 25 x 25 matrices
                                       frac_unpadded=round(flops / (cz["ms"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS
                                                           * per_block * (dim**2 / 64.0) ** 3, 4) if cz["ms"] > 0 else None),
            "cpu_baseline": {"value": round(1.0 / cpu_s, 1), "unit": "steps/s", "cores": 1, "kind": "port",
                             "sample": f"{n_cpu} steps of oracle.perturbative_solve (the reference's per-step loop: array polynomial by "
                                       f"tensordot, scipy.linalg.expm, matmul) on this host: {cpu_s * 1e3:.2f} ms per step"}}
    return out
