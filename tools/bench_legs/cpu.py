"""bench.py legs: the CPU baselines (the NumPy oracle on this host; the only legs that import oracle/)."""
import os
import time

import numpy as np

from .common import (ALL_CLASSES, CFG5_SWEEP, FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, LDS_PEAK_GBS, MAX_DT, N_DRIVES, N_QUBITS, ROOT, SWEEP,  # noqa: F401
                     T_FINAL, ZGEMM_NOTE, _mfma_roofline, build_diag_frame_stack, build_frame_basis_stack, build_model_stack,
                     measured_traffic, profile_pass, sweep_table)


def leg_cpu_baseline(workloads, cfg, static, ops, frame_im, amps, phs):
    from oracle import dynamics_oracle as orc
    from threadpoolctl import threadpool_info, threadpool_limits

    a_d, a = static, ops
    d = 1j * frame_im
    best = None
    for threads in sorted({8, 32, os.cpu_count() or 8}):      # short probe: which BLAS width is fastest here
        if threads > (os.cpu_count() or 8):
            continue
        with threadpool_limits(limits=threads):
            t0c = time.perf_counter()

            def rhs(t, y):
                c = workloads.gaussian_coefficient_table(np.array([t]), amps[0], phs[0], cfg["carrier"], T_FINAL)[0]
                return orc.generator_rhs(a_d, a, c, d, None, t, y)

            orc.rk4_solve(rhs, [0.0, 10 * MAX_DT], cfg["y0"], MAX_DT)
            rate = 40 / (time.perf_counter() - t0c)
        if best is None or rate > best[0]:
            best = (rate, threads)
    threads = best[1]
    n_inst = 4
    n_steps = int(min(200, max(20, best[0] * 15 / (4 * n_inst))))   # ~15 s of CPU work
    with threadpool_limits(limits=threads):
        t0c = time.perf_counter()
        for b in range(n_inst):
            def rhs(t, y, b=b):
                c = workloads.gaussian_coefficient_table(np.array([t]), amps[b], phs[b], cfg["carrier"], T_FINAL)[0]
                return orc.generator_rhs(a_d, a, c, d, None, t, y)

            orc.rk4_solve(rhs, [0.0, n_steps * MAX_DT], cfg["y0"], MAX_DT)
        cpu_s = time.perf_counter() - t0c
    best = (n_inst * n_steps * 4 / cpu_s, threads, cpu_s)
    return {
        "value": round(best[0], 1), "unit": "RHS evals/s", "cores": best[1], "kind": "port",
        "cores_for_blas3": min(os.cpu_count() or 8, 64),
        "sample": f"{n_inst} instances x {n_steps} RK4 steps ({n_inst * n_steps * 4} RHS evals) of the same "
                  f"model with the NumPy oracle (tensordot + matvec); best of BLAS thread counts 8/32/all on a "
                  f"{os.cpu_count()}-CPU host: {best[1]} threads, {best[2]:.1f} s",
        "host": {"cpu_count": os.cpu_count(), "numpy": np.__version__,
                 "blas": [f"{i.get('internal_api')} {i.get('version')} ({i.get('threading_layer') or i.get('user_api')})"
                          for i in threadpool_info()],
                 "OPENBLAS_NUM_THREADS": os.environ.get("OPENBLAS_NUM_THREADS")}}


def leg_cpu_configs(workloads, threads, want4=True, want5=True):
    """CPU baselines of cfg 4 / cfg 5 beside the device numbers: ONE step of the reference's algorithm
    (solvers/fixed_step_solvers.py:80-108,321-363: dense generator(s) by tensordot, Magnus term, scipy.linalg.expm, one
    matrix-vector product) with the NumPy oracle on this host -- exactly linear in steps (and instances)."""
    import scipy.linalg
    import scipy.sparse as sp
    from oracle import dynamics_oracle as orc
    from threadpoolctl import threadpool_limits

    out = {}
    with threadpool_limits(limits=threads):
        if want4:
            cfg = workloads.lindblad_config()
            n = cfg["h_d"].shape[0]
            eye = sp.identity(n, format="csr")

            def vcomm(a):        # -i (I (x) A - A^T (x) I), oracle.vec_commutator built sparse (set-up only, not timed)
                a = sp.csr_matrix(a)
                return (-1j * (sp.kron(eye, a) - sp.kron(a.T, eye))).toarray()

            def vdiss(l):        # conj(L) (x) L - (I (x) L^+L + (L^+L)^T (x) I) / 2, oracle.vec_dissipator
                l = sp.csr_matrix(l)
                ldl = l.conj().T @ l
                return (sp.kron(l.conj(), l) - 0.5 * (sp.kron(eye, ldl) + sp.kron(ldl.T, eye))).toarray()

            s_d = vcomm(cfg["h_d"]) + sum(vdiss(l) for l in cfg["static_dissipators"])
            s_ops = np.stack([vcomm(o) for o in cfg["ops"]])
            amps, phases = workloads.sweep_parameters(0, len(cfg["ops"]))
            h, t0 = cfg["max_dt"], cfg["t_final"] / 2

            def gen(t):
                c = workloads.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], cfg["t_final"])[0]
                return orc.generator_evaluate(s_d, s_ops, c, None, None, t)

            y = cfg["rho0"].flatten(order="F")
            t1 = time.perf_counter()
            omega = orc.magnus_terms(gen, t0, h, 1)
            t2 = time.perf_counter()
            prop = scipy.linalg.expm(omega)
            t3 = time.perf_counter()
            y = prop @ y
            t4 = time.perf_counter()
            out["cfg4"] = {"value": round(1.0 / (t4 - t1), 4), "unit": "steps/s", "s_per_step": round(t4 - t1, 2), "cores": threads,
                           "kind": "port", "sample": "1 scipy_expm step (Magnus order 1) of the N = 4096 superoperator model "
                           "with the NumPy oracle: generator by tensordot over the dense (7, 4096, 4096) stack %.1f s, "
                           "scipy.linalg.expm %.1f s, matvec %.3f s; 100 steps per solve" % (t2 - t1, t3 - t2, t4 - t3),
                           "trace_after_the_step": float(abs(np.trace(y.reshape(n, n, order="F"))))}
            del s_d, s_ops, prop, omega
        if want5:
            cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
            ops, static, fim, _ = build_diag_frame_stack(cfg)
            d = 1j * fim
            amps, phases = workloads.sweep_parameters(0, 8)

            def gen5(t):
                c = workloads.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], cfg["t_final"])[0]
                return orc.generator_evaluate(static, ops, c, d, None, t)

            t1 = time.perf_counter()
            omega = orc.magnus_terms(gen5, 2.5, cfg["max_dt"], 2)
            t2 = time.perf_counter()
            prop = scipy.linalg.expm(omega)
            t3 = time.perf_counter()
            y = prop @ cfg["y0"]
            t4 = time.perf_counter()
            out["cfg5"] = {"value": round(1.0 / (t4 - t1), 4), "unit": "instance-steps/s", "s_per_instance_step": round(t4 - t1, 2),
                           "cores": threads, "kind": "port",
                           "sample": "1 instance x 1 scipy_expm step (Magnus order 2) of the n = 4096 model with the NumPy "
                                     "oracle: two dense generators + commutator %.1f s, scipy.linalg.expm %.1f s, matvec "
                                     "%.3f s; 1024 instances x 20 steps per sweep" % (t2 - t1, t3 - t2, t4 - t3),
                           "norm_after_the_step": float(np.linalg.norm(y))}
    return out
