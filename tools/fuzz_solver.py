"""Random Hamiltonian models through the PUBLIC interface (Solver.solve in list mode, solvers/solver_classes.py:384-590 of the
reference) against the NumPy oracle: the sizes tests/test_gpu_parity.py::test_randomised_models_vs_oracle does not reach --
dimension 2 .. 200, 1 .. 8 Hermitian drive operators, static Hamiltonian or none, no / diagonal / full rotating frame
(the full frame goes through eigh and the sector ordering of the model build), sweeps of 1 .. 600 instances whose signals are
DiscreteSignal pulses with a carrier (the coefficient table is then evaluated ON THE DEVICE, row f1) or analytic envelopes
(host table), per-instance or shared initial states (vectors or 3 columns), forwards / backwards, a t_eval, RK4 or scipy_expm
with Magnus order 1 .. 3.  Instances 0, middle, last are compared with the oracle's solve at 1e-9.

    python tools/fuzz_solver.py --cases 60 --seed 0          (GPU box; one line per case; exits 1 on a mismatch)
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
    rng = np.random.default_rng(50_000 + seed)
    n = int(rng.choice([rng.integers(2, 17), rng.integers(17, 65), rng.integers(65, 201)], p=[0.3, 0.4, 0.3]))
    k = int(rng.integers(1, 9))
    has_static = bool(rng.integers(0, 2))
    frame_kind = ["none", "diag", "full"][int(rng.integers(0, 3))]
    m = [None, 3][int(rng.choice([0, 1], p=[0.8, 0.2]))]
    batch = int(rng.choice([1, 2, 17, 64, 300, 600], p=[0.1, 0.15, 0.25, 0.2, 0.2, 0.1]))
    method = ["RK4", "scipy_expm"][int(rng.choice([0, 1], p=[0.6, 0.4]))]
    mo = int(rng.integers(1, 4))
    backwards = bool(rng.integers(0, 2))
    discrete = bool(rng.integers(0, 2))
    shared_y0 = bool(rng.integers(0, 2))
    real_ops = rng.random() < 0.3                    # real-symmetric Hamiltonians: purely imaginary generators

    def herm(scale=1.0):
        a = crand(rng, n, n)
        if real_ops:
            a = a.real + 0j
        return (a + a.conj().T) / 2 * (scale / np.sqrt(n))

    h_static = herm(2.0) if has_static else None
    h_ops = np.array([herm() for _ in range(k)])
    frame = {"none": None, "diag": rng.normal(size=n), "full": herm(2.0)}[frame_kind]
    span = 0.5
    t_span = [span, 0.0] if backwards else [0.0, span]
    t_eval = None if rng.integers(0, 2) else sorted(rng.uniform(0, span, 2), reverse=backwards)
    max_dt = 0.02 if method == "RK4" else 0.06
    dt_s, ns = 0.07, 9                               # discrete pulses: 9 samples of 0.07 (the last steps run past the pulse)

    def make_sigs():
        if discrete:
            smp = rng.uniform(0.2, 1.0, (k, ns)) * np.exp(1j * rng.uniform(0, 1, (k, ns)))
            nus, phs = rng.uniform(0, 3, k), rng.uniform(-3, 3, k)
            sigs = [qd.DiscreteSignal(dt=dt_s, samples=smp[j], carrier_freq=nus[j], phase=phs[j]) for j in range(k)]

            def coeff(t):
                return np.array([orc.signal_sum_value(np.array([orc.discrete_envelope(smp[j], dt_s, 0.0, t)]), [nus[j]], [phs[j]], t)
                                 for j in range(k)])
        else:
            amps, nus, phs = rng.uniform(-1, 1, k), rng.uniform(0, 2, k), rng.uniform(-3, 3, k)
            sigs = [qd.Signal(lambda t, a=a: a * np.cos(0.7 * t) + 0j, nu, ph) for a, nu, ph in zip(amps, nus, phs)]

            def coeff(t):
                return np.array([orc.signal_sum_value(np.array([a * np.cos(0.7 * t) + 0j]), [nu], [ph], t)
                                 for a, nu, ph in zip(amps, nus, phs)])
        return sigs, coeff

    def make_y0():
        y = crand(rng, n) if m is None else crand(rng, n, m)
        return y / np.linalg.norm(y)

    sig_sets = [make_sigs() for _ in range(batch)]
    y0s = [make_y0()] * batch if shared_y0 else [make_y0() for _ in range(batch)]
    t0 = time.perf_counter()
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    kwargs = dict(method=method, max_dt=max_dt, t_eval=t_eval)
    if method == "scipy_expm":
        kwargs["magnus_order"] = mo
    if batch > 1:
        res = solver.solve(t_span=t_span, y0=y0s[0] if shared_y0 else y0s, signals=[s for s, _ in sig_sets], **kwargs)
    else:
        res = solver.solve(t_span=t_span, y0=y0s[0], signals=sig_sets[0][0], **kwargs)
    res = res if isinstance(res, list) else [res]
    wall = time.perf_counter() - t0
    assert len(res) == batch
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
    err = 0.0
    for b in sorted({0, batch // 2, batch - 1}):
        t_ref, y_ref = orc.solve_generator_model(a_d, a, d, basis, sig_sets[b][1], t_span, y0s[b], method, max_dt,
                                                 t_eval=t_eval, magnus_order=mo)
        assert np.array_equal(np.asarray(res[b].t), np.asarray(t_ref)), (res[b].t, t_ref)
        assert res[b].y.shape == y_ref.shape, (res[b].y.shape, y_ref.shape)
        err = max(err, float(np.max(np.abs(res[b].y - y_ref)) / (1.0 + np.max(np.abs(y_ref)))))
    ok = err < 1e-9
    if verbose or not ok:
        route = getattr(res[0], "route", "")
        print(f"seed {seed:5d} {'ok  ' if ok else 'FAIL'} n={n:3d} k={k} st={int(has_static)} frame={frame_kind:4s} real={int(real_ops)} "
              f"B={batch:3d} sharedy0={int(shared_y0)} m={m} discrete={int(discrete)} bw={int(backwards)} t_eval={int(t_eval is not None)} "
              f"{method}{mo if method != 'RK4' else ''}: oracle {err:.1e} [{route}] {wall:.2f}s", flush=True)
    return ok


def run_pauli_case(qd, orc, seed, verbose=True):
    """Models built from random PAULI STRINGS on 8 .. 10 qubits in the computational basis with a diagonal frame (or none and
    no diagonal): very sparse stacks -- the one-launch sweep kernels with their element forms: strings of X only (one signed
    magnitude and one flip mask per slot: no operator elements at all, ell_flip_duo_kernel / ell_sweep_kernel<.., 3>), X and Z
    (one magnitude, signs per row: packed), Y too (imaginary planes), equal or different magnitudes inside an operator."""
    rng = np.random.default_rng(90_000 + seed)
    nq = int(rng.choice([8, 9, 10], p=[0.4, 0.4, 0.2]))
    n = 2**nq
    rows = np.arange(n)
    kind = ["x", "xz", "xyz"][int(rng.choice([0, 1, 2], p=[0.5, 0.25, 0.25]))]
    k = int(rng.integers(1, 5))
    framed = bool(rng.integers(0, 2))
    batch = int(rng.choice([1, 2, 5, 33, 128], p=[0.1, 0.2, 0.3, 0.25, 0.15]))
    method = ["RK4", "scipy_expm"][int(rng.choice([0, 1], p=[0.3, 0.7]))]
    mo = int(rng.integers(1, 3))
    same_mag = bool(rng.integers(0, 2))
    used = set()

    def string():
        # one Pauli letter per qubit: X flips, Y flips with the factor i (-1)^(column bit), Z gives the sign (-1)^(bit)
        while True:
            letters = rng.choice(4, size=nq, p={"x": [0.6, 0.4, 0.0, 0.0], "xz": [0.5, 0.3, 0.0, 0.2], "xyz": [0.4, 0.25, 0.15, 0.2]}[kind])
            if rng.random() < 0.4:
                letters[nq - 1] = 1                  # (X on the top qubit: crosses the halves of the two-workgroup kernels)
            xm = sum(1 << q for q in range(nq) if letters[q] == 1)
            ym = sum(1 << q for q in range(nq) if letters[q] == 2)
            zm = sum(1 << q for q in range(nq) if letters[q] == 3)
            if (xm | ym) == 0 or (xm, ym, zm) in used:          # (diagonal strings belong to the frame)
                continue
            used.add((xm, ym, zm))
            cols = rows ^ (xm | ym)
            par = lambda v: np.array([bin(int(x)).count("1") & 1 for x in v])
            val = (1j) ** bin(ym).count("1") * (-1.0) ** par(cols & ym) * (-1.0) ** par(rows & zm)
            mat = np.zeros((n, n), dtype=complex)
            mat[rows, cols] = val
            return mat

    def operator(scale):
        op = np.zeros((n, n), dtype=complex)
        for t in range(int(rng.integers(1, 4))):
            mag = 1.0 if same_mag else 0.5 + 0.25 * t
            op += mag * (-1.0) ** int(rng.integers(0, 2)) * string()
        return scale * op

    h_ops = np.array([operator(2 * np.pi * 0.03) for _ in range(k)])
    diag = 2 * np.pi * rng.uniform(0.0, 0.4, n) if framed else np.zeros(n)
    h_static = np.diag(diag).astype(complex) + (operator(2 * np.pi * 0.005) if rng.integers(0, 2) else 0.0)
    frame = diag.copy() if framed else None
    span = 0.3
    t_span = [0.0, span]
    max_dt = 0.02 if method == "RK4" else 0.06
    sig_sets = []
    for _ in range(batch):
        amps, nus, phs = rng.uniform(-1, 1, k), rng.uniform(0, 1, k), rng.uniform(-3, 3, k)
        sigs = [qd.Signal(lambda t, a=a: a * np.cos(0.7 * t) + 0j, nu, ph) for a, nu, ph in zip(amps, nus, phs)]

        def coeff(t, amps=amps, nus=nus, phs=phs):
            return np.array([orc.signal_sum_value(np.array([a * np.cos(0.7 * t) + 0j]), [nu], [ph], t) for a, nu, ph in zip(amps, nus, phs)])
        sig_sets.append((sigs, coeff))
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    ctx = qd.default_context()
    t0 = time.perf_counter()
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    kwargs = dict(method=method, max_dt=max_dt)
    if method == "scipy_expm":
        kwargs["magnus_order"] = mo
    if batch > 1:
        res = solver.solve(t_span=t_span, y0=y0, signals=[s for s, _ in sig_sets], **kwargs)
    else:
        res = [solver.solve(t_span=t_span, y0=y0, signals=sig_sets[0][0], **kwargs)]
    wall = time.perf_counter() - t0
    split = ctx.counters("sweep_split")
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
    err = 0.0
    for b in sorted({0, batch - 1}):
        _, y_ref = orc.solve_generator_model(a_d, a, d, basis, sig_sets[b][1], t_span, y0, method, max_dt, magnus_order=mo)
        err = max(err, float(np.max(np.abs(res[b].y - y_ref)) / (1.0 + np.max(np.abs(y_ref)))))
    ok = err < 1e-9
    if verbose or not ok:
        print(f"seed {seed:5d} {'ok  ' if ok else 'FAIL'} pauli nq={nq} kind={kind:3s} k={k} framed={int(framed)} same_mag={int(same_mag)} "
              f"B={batch:3d} {method}{mo if method != 'RK4' else ''}: oracle {err:.1e} "
              f"[last sweep launch: {int(split['launches'])} workgroup(s) per instance, element form {int(split['ms'])}] {wall:.2f}s", flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pauli", action="store_true", help="models built from random Pauli strings (very sparse stacks)")
    args = ap.parse_args()
    import qiskit_dynamics_amd as qd
    from oracle import dynamics_oracle as orc

    qd.default_context()
    bad = []
    for s in range(args.seed, args.seed + args.cases):
        try:
            if not (run_pauli_case if args.pauli else run_case)(qd, orc, s):
                bad.append(s)
        except Exception as exc:
            print(f"seed {s:5d} EXC  {type(exc).__name__}: {exc}", flush=True)
            bad.append(s)
    print(f"{args.cases - len(bad)} of {args.cases} cases agree; failing seeds: {bad}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
