"""Random LINDBLAD models through the public interface (Solver.solve, single and list mode; reference:
solvers/solver_classes.py:384-590, models/lindblad_model.py:100-212,410-538, models/operator_collections.py:451-567,851-1061)
against the NumPy oracle -- the open-system twin of tools/fuzz_solver.py, which draws Hamiltonian models only.

Per case: dimension 2 .. 24 (superoperator dimension 4 .. 576), static Hamiltonian or none, 0 .. 3 Hamiltonian operators with
signals, 0 .. 3 static dissipators, 0 .. 2 dissipator operators with (real) signals -- at least one group of each side is
drawn so that the model is neither empty nor closed --, no / diagonal / full rotating frame, vectorized or not, a single solve
or a sweep of 2 .. 40 instances (own signals; own or shared initial density matrix), forwards or backwards, optional t_eval,
RK4 (both forms) or scipy_expm with Magnus order 1 .. 3, sequential or parallel in time (vectorised form: the reference raises for the matrix form,
solvers/solver_functions.py:334-337 -- checked too).  The oracle side: oracle.lindblad_model_build -> frame-basis groups;
vectorised: vectorized_lindblad_stack + vectorized_frame_diag through solve_generator_model(kind="lindblad_vec");
matrix form: lindblad_rhs through rk4_solve with the frame-basis maps of solver_functions.py:376-450.
Instances 0, middle, last are compared at 1e-9; trace and hermiticity of the final states are printed.

    python tools/fuzz_lindblad.py --cases 60 --seed 0          (GPU box; one line per case; exits 1 on a mismatch)
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def crand(rng, *shape):
    return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)


def run_case(qd, orc, seed, verbose=True):
    rng = np.random.default_rng(70_000 + seed)
    n = int(rng.choice([rng.integers(2, 6), rng.integers(6, 13), rng.integers(13, 25)], p=[0.4, 0.4, 0.2]))
    vectorized = bool(rng.integers(0, 2))
    has_static = bool(rng.integers(0, 2))
    k_h = int(rng.integers(0, 4))
    if not has_static and k_h == 0:
        k_h = 1
    n_s = int(rng.integers(0, 4))
    k_l = int(rng.integers(0, 3))
    if n_s == 0 and k_l == 0:
        n_s = 1
    frame_kind = ["none", "diag", "full"][int(rng.integers(0, 3))]
    batch = int(rng.choice([0, 2, 5, 17, 40], p=[0.25, 0.2, 0.25, 0.2, 0.1]))      # 0: a single (non-list) solve
    if not vectorized and n > 12:
        batch = min(batch, 5)
    method = "RK4" if not vectorized else ["RK4", "scipy_expm"][int(rng.integers(0, 2))]
    mo = int(rng.integers(1, 4))
    backwards = rng.random() < 0.25
    shared_y0 = bool(rng.integers(0, 2))
    no_signals = k_h == 0 and k_l == 0
    if no_signals:
        shared_y0 = False                       # (list mode then comes from the initial states alone)

    def herm(scale=1.0):
        a = crand(rng, n, n)
        return (a + a.conj().T) / 2 * (scale / np.sqrt(n))

    h_static = herm(2.0) if has_static else None
    h_ops = np.array([herm() for _ in range(k_h)]) if k_h else None
    n_stat = np.array([crand(rng, n, n) * (0.4 / np.sqrt(n)) for _ in range(n_s)]) if n_s else None
    l_ops = np.array([crand(rng, n, n) * (0.5 / np.sqrt(n)) for _ in range(k_l)]) if k_l else None
    frame = {"none": None, "diag": rng.normal(size=n), "full": herm(2.0)}[frame_kind]
    span = 0.4
    t_span = [span, 0.0] if backwards else [0.0, span]
    t_eval = None if rng.integers(0, 2) else sorted(rng.uniform(0, span, 2), reverse=backwards)
    max_dt = 0.01 if method == "RK4" else 0.05
    # a quarter of the vectorised cases take the parallel-in-time form of their method (row f3:
    # solvers/fixed_step_solvers.py:222-258,524-613; the same linear map per step, the products re-associated -- compared
    # with the oracle's sequential loop at the same 1e-9)
    parallel = vectorized and rng.random() < 0.25
    dev_method = {"RK4": "hip_RK4_parallel", "scipy_expm": "hip_expm_parallel"}[method] if parallel else method

    def make_sigs():
        """(signals as Solver.solve takes them, ham coefficient function, dissipator coefficient function)"""
        ha, hnu, hph = rng.uniform(0.3, 1.0, k_h), rng.uniform(0, 3, k_h), rng.uniform(-3, 3, k_h)
        hw = rng.uniform(0.5, 2.0, k_h)
        da, dw = rng.uniform(0.2, 1.0, k_l), rng.uniform(0.5, 2.0, k_l)
        hs = [qd.Signal(lambda t, a=ha[j], w=hw[j]: a * np.cos(w * t) + 0j, hnu[j], hph[j]) for j in range(k_h)]
        ds = [qd.Signal(lambda t, a=da[j], w=dw[j]: a * (1.0 + 0.5 * np.sin(w * t)) + 0j, 0.0) for j in range(k_l)]

        def hc(t):
            return np.array([np.real(ha[j] * np.cos(hw[j] * t) * np.exp(1j * (2 * np.pi * hnu[j] * t + hph[j]))) for j in range(k_h)])

        def dc(t):
            return np.array([da[j] * (1.0 + 0.5 * np.sin(dw[j] * t)) for j in range(k_l)])

        if k_l:
            sig = (hs if k_h else None, ds)
        else:
            sig = hs if k_h else None          # (a model without operators takes no signals: the reference raises for [])
        return sig, hc, dc

    def make_rho():
        a = crand(rng, n, n)
        rho = a @ a.conj().T
        return rho / np.trace(rho)

    nb = max(batch, 1)
    per = [make_sigs() for _ in range(nb)]
    rhos = [make_rho() for _ in range(1 if shared_y0 else nb)]

    def y_of(rho):
        return rho.flatten(order="F") if vectorized else rho

    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, static_dissipators=n_stat,
                       dissipator_operators=l_ops, rotating_frame=frame, vectorized=vectorized)
    kw = dict(method=dev_method, max_dt=max_dt, t_eval=t_eval)
    if method == "scipy_expm":
        kw["magnus_order"] = mo
    t0 = time.perf_counter()
    if batch == 0:
        res = [solver.solve(t_span=t_span, y0=y_of(rhos[0]), signals=per[0][0], **kw)]
    else:
        y0_arg = y_of(rhos[0]) if shared_y0 else [y_of(r) for r in rhos]
        res = solver.solve(t_span=t_span, y0=y0_arg, signals=None if no_signals else [p[0] for p in per], **kw)
        assert isinstance(res, list) and len(res) == nb
    dt_dev = time.perf_counter() - t0

    extra = ""
    if not vectorized and seed % 7 == 0:      # the matrix form has no generator: LMDE methods must refuse it, as the reference does
        try:
            solver.solve(t_span=t_span, y0=y_of(rhos[0]), signals=per[0][0], method="scipy_expm", max_dt=max_dt)
            return False, f"seed {seed:5d} FAIL scipy_expm accepted a non-vectorised Lindblad model"
        except Exception:        # noqa: BLE001 -- any error type the product raises where the reference raises QiskitError
            extra = " [scipy_expm refused]"

    h_d, hf, nf, lf, d, basis = orc.lindblad_model_build(h_static, h_ops, n_stat, l_ops, frame)
    worst = 0.0
    tr_dev = herm_dev = 0.0
    for b in sorted({0, nb // 2, nb - 1}):
        _, hc, dc = per[b]
        rho0 = rhos[0 if shared_y0 else b]
        if vectorized:
            s_d, s = orc.vectorized_lindblad_stack(h_d, hf, nf, lf)
            dd = None if d is None else orc.vectorized_frame_diag(d)

            def cf(t, hc=hc, dc=dc):
                return np.concatenate([hc(t), dc(t)])

            _, y = orc.solve_generator_model(s_d, s, dd, basis, cf, t_span, rho0.flatten(order="F"), method=method,
                                             max_dt=max_dt, t_eval=t_eval, magnus_order=mo, kind="lindblad_vec")
            fin = np.asarray(res[b].y)[-1].reshape(n, n, order="F")
        else:
            y0f = orc.y0_into_frame_basis(rho0, basis, "lindblad")

            def rhs(t, r, hc=hc, dc=dc):
                return orc.lindblad_rhs(h_d, hf, nf, lf, hc(t) if k_h else None, dc(t) if k_l else None, d, t, r)

            _, y = orc.rk4_solve(rhs, t_span, y0f, max_dt, t_eval)
            y = orc.results_out_of_frame_basis(y, basis, "lindblad", 2)
            fin = np.asarray(res[b].y)[-1]
        got = np.asarray(res[b].y)
        if got.shape != y.shape:
            return False, f"seed {seed:5d} FAIL shape {got.shape} vs oracle {y.shape}"
        worst = max(worst, float(np.abs(got - y).max()))
        tr_dev = max(tr_dev, abs(np.trace(fin) - 1.0))
        herm_dev = max(herm_dev, float(np.abs(fin - fin.conj().T).max()))
    ok = worst < 1e-9
    line = (f"seed {seed:5d} {'ok  ' if ok else 'FAIL'} n={n:2d} {'vec' if vectorized else 'mat'} Hd={int(has_static)} kh={k_h} "
            f"ns={n_s} kl={k_l} frame={frame_kind:4s} B={batch:2d} {dev_method}{mo if method == 'scipy_expm' else ''} "
            f"{'bwd' if backwards else 'fwd'} t_eval={'y' if t_eval is not None else 'n'} y0={'shared' if shared_y0 else 'own'}: "
            f"oracle {worst:.1e} trace {tr_dev:.1e} herm {herm_dev:.1e}{extra} {dt_dev:.2f}s")
    if verbose:
        print(line, flush=True)
    return ok, line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import qiskit_dynamics_amd as qd
    from oracle import dynamics_oracle as orc

    bad = []
    for c in range(args.cases):
        ok, line = run_case(qd, orc, args.seed + c)
        if not ok:
            bad.append(args.seed + c)
            print(line, flush=True)
    print(f"{args.cases - len(bad)} of {args.cases} cases agree; failing seeds: {bad}", flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
