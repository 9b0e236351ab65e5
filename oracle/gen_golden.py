"""Generate tests/golden/*.npz by running the REAL reference (container only).

Usage (from the repo root, in the build container where /root/reference exists):
    python -m oracle.gen_golden

Each .npz stores the inputs AND the reference's outputs for one scenario family, so the parity
tests need neither the reference nor this script at run time.  Scenario list = SURVEY.md App. B
(re-stated here on seeded inputs; the reference's unittest files themselves cannot run without
ddt/qiskit).  Results that involve a non-diagonal rotating frame are stored OUT of the frame
basis (`in_frame_basis=False`), which is invariant under the eigenvector gauge of `eigh`.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle.ref_shim import load_reference  # noqa: E402

load_reference()

from qiskit_dynamics.models import GeneratorModel, HamiltonianModel, LindbladModel  # noqa: E402
from qiskit_dynamics.models import RotatingFrame  # noqa: E402
from qiskit_dynamics.models.operator_collections import OperatorCollection  # noqa: E402
from qiskit_dynamics.signals import DiscreteSignal, Signal, SignalList  # noqa: E402
from qiskit_dynamics.solvers.fixed_step_solvers import (  # noqa: E402
    RK4_solver, get_fixed_step_sizes, scipy_expm_solver)
from qiskit_dynamics.solvers.solver_classes import Solver  # noqa: E402
from qiskit_dynamics.solvers.solver_functions import solve_lmde  # noqa: E402

from qiskit_dynamics_amd import workloads  # noqa: E402  (pure-numpy input builders)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def crand(rng, *shape):
    return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)


def herm(rng, n):
    a = crand(rng, n, n)
    return a + a.conj().T


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays")


# ---------------------------------------------------------------------------------------------
def gen_collection():
    """a1/a2: OperatorCollection.evaluate / evaluate_rhs (test_operator_collections.py:60-94)."""
    rng = np.random.default_rng(342)
    n, k, m = 16, 6, 3
    ops = crand(rng, k, n, n)
    static = crand(rng, n, n)
    coeffs = rng.uniform(-1, 1, (5, k))
    y1 = crand(rng, n)
    ym = crand(rng, n, m)
    out = {}
    for tag, st, op in (("full", static, ops), ("nostatic", None, ops)):
        coll = OperatorCollection(static_operator=st, operators=op)
        out[f"{tag}_eval"] = np.array([coll.evaluate(c) for c in coeffs])
        out[f"{tag}_rhs1"] = np.array([coll.evaluate_rhs(c, y1) for c in coeffs])
        out[f"{tag}_rhsm"] = np.array([coll.evaluate_rhs(c, ym) for c in coeffs])
    coll = OperatorCollection(static_operator=static, operators=None)
    out["staticonly_eval"] = coll.evaluate(None)
    out["staticonly_rhs1"] = coll.evaluate_rhs(None, y1)
    # Pauli KAT, test/dynamics/arraylias/test_alias.py:92-102
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    coll = OperatorCollection(operators=np.array([x, 1j * z]))
    out["pauli_eval"] = coll.evaluate(np.array([1.0, 2.0]))
    save("collection", ops=ops, static=static, coeffs=coeffs, y1=y1, ym=ym, **out)


# ---------------------------------------------------------------------------------------------
def gen_signals():
    """a8: Signal / DiscreteSignal / SignalSum / SignalList values (test_signals.py:386-430)."""
    t = np.linspace(-0.7, 2.3, 41)
    out = {"t": t}
    s_const = Signal(0.37, carrier_freq=1.3, phase=0.4)
    out["const"] = s_const(t)
    amp, t0, sig, nu, phi = 0.8, 0.9, 0.35, 4.75, -1.1
    s_gauss = Signal(lambda tt: amp * np.exp(-((tt - t0) ** 2) / (2 * sig**2)), nu, phi)
    out["gauss_params"] = np.array([amp, t0, sig, nu, phi])
    out["gauss"] = s_gauss(t)
    samples = np.array([1.0 + 2.0j, 2.0 + 1.0j, 3.0 + 0.0j, -0.5j])
    d = DiscreteSignal(dt=0.5, samples=samples, start_time=0.25, carrier_freq=0.9, phase=0.2)
    out["disc_samples"] = samples
    out["disc_params"] = np.array([0.5, 0.25, 0.9, 0.2])
    out["disc"] = d(t)
    out["disc_env"] = d.envelope(t)
    # exact bin edges
    edges = np.array([0.25, 0.75, 1.25, 1.75, 2.25, 0.2499999, 2.2499999])
    out["disc_edges_t"] = edges
    out["disc_edges"] = d(edges)
    ssum = s_const + s_gauss
    out["sum"] = ssum(t)
    sl = SignalList([s_const, s_gauss, d, ssum, 1.5])
    out["list"] = sl(t)  # (T, 5)
    out["list_scalar_t"] = sl(0.613)
    # signal algebra (signals/signals.py:838-1121): values only, the composite types are not pinned
    out["prod_const_gauss"] = (s_const * s_gauss)(t)
    out["prod_disc_gauss"] = (d * s_gauss)(t)
    out["prod_gauss_gauss"] = (s_gauss * s_gauss)(t)
    out["prod_scalar"] = (2.5 * s_gauss)(t)
    out["neg_gauss"] = (-s_gauss)(t)
    out["diff_gauss_disc"] = (s_gauss - d)(t)
    out["conj_gauss_complex"] = s_gauss.conjugate().complex_value(t)
    dsig = DiscreteSignal.from_Signal(s_gauss, dt=0.1, n_samples=20, start_time=0.0)
    out["from_signal"] = dsig(t)
    out["from_signal_samples"] = np.asarray(dsig.samples)
    dsig2 = DiscreteSignal.from_Signal(s_gauss, dt=0.1, n_samples=20, start_time=0.0, sample_carrier=True)
    out["from_signal_carrier"] = dsig2(t)
    out["flatten_sum"] = ssum.flatten()(t)
    # DiscreteSignalSum (signals/signals.py:612-777), add_samples (:411-439), SignalList.flatten (:805-814)
    from qiskit_dynamics.signals import DiscreteSignalSum
    rng = np.random.default_rng(99)
    dss_samples = rng.normal(size=(6, 3)) + 1j * rng.normal(size=(6, 3))
    dss = DiscreteSignalSum(dt=0.4, samples=dss_samples, start_time=-0.3, carrier_freq=np.array([0.5, 1.5, -0.7]),
                            phase=np.array([0.1, -0.2, 0.3]))
    out["dss_samples"] = dss_samples
    out["dss"] = dss(t)
    out["dss_complex"] = dss.complex_value(t)
    out["dss_item1"] = dss[1](t)
    out["dss_slice"] = dss[np.array([0, 2])](t)
    dss2 = DiscreteSignalSum.from_SignalSum(ssum, dt=0.1, n_samples=20, start_time=0.0)
    out["dss_from_sum"] = dss2(t)
    out["dss_from_sum_samples"] = np.asarray(dss2.samples)
    dss3 = DiscreteSignalSum.from_SignalSum(ssum, dt=0.1, n_samples=20, start_time=0.0, sample_carrier=True)
    out["dss_from_sum_carrier"] = dss3(t)
    d_add = DiscreteSignal(dt=0.5, samples=samples, start_time=0.25, carrier_freq=0.9, phase=0.2)
    d_add.add_samples(6, [0.5, -0.25j])
    out["add_samples"] = d_add(t)
    out["add_samples_samples"] = np.asarray(d_add.samples)
    out["list_flatten"] = SignalList([s_const, ssum, d, dss]).flatten()(t)
    save("signals", **out)


# ---------------------------------------------------------------------------------------------
def _gm_cases():
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    y = np.array([[0, -1j], [1j, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    return x, y, z


def gen_generator_model():
    """a3-a7: GeneratorModel / HamiltonianModel evaluate + evaluate_rhs
    (test_generator_model.py:171-221,267-347,370-397,495-505,615-674;
    test_hamiltonian_model.py:99-139,201-287)."""
    x, y, z = _gm_cases()
    out = {}
    # analytic 2x2 KAT
    gm = GeneratorModel(operators=[-1j * x / 2 * 2, -1j * z / 2], static_operator=None,
                        signals=[Signal(1.0, 1.0 / 3), Signal(1.0, 2.0 / 3)])
    out["kat_ops"] = np.array([-1j * x, -1j * z / 2])
    out["kat_carrier"] = np.array([1.0 / 3, 2.0 / 3])
    out["kat_eval_t2"] = gm.evaluate(2.0)
    out["kat_rhs_t2"] = gm.evaluate_rhs(2.0, np.array([0.2, 0.5]))

    for tag, seed, n, k in (("r5", 30493, 5, 3), ("r10", 94818, 10, 5)):
        rng = np.random.default_rng(seed)
        ops = crand(rng, k, n, n)
        static = crand(rng, n, n)
        fr = crand(rng, n, n)
        frame_op = fr - fr.conj().T  # anti-Hermitian
        carr = rng.uniform(0.2, 2.0, k)
        phases = rng.uniform(-np.pi, np.pi, k)
        amps = rng.uniform(-1, 1, k)
        sigs = [Signal(a, c, p) for a, c, p in zip(amps, carr, phases)]
        times = np.array([0.0, 1.0, 1.123, np.pi])
        y1 = crand(rng, n)
        ym = crand(rng, n, 2)
        out[f"{tag}_ops"], out[f"{tag}_static"], out[f"{tag}_frame"] = ops, static, frame_op
        out[f"{tag}_carrier"], out[f"{tag}_phase"], out[f"{tag}_amp"] = carr, phases, amps
        out[f"{tag}_times"], out[f"{tag}_y1"], out[f"{tag}_ym"] = times, y1, ym
        for ftag, frame in (("fr", frame_op), ("diag", np.diag(frame_op).copy()), ("nofr", None)):
            for stag, st in (("st", static), ("nost", None)):
                m = GeneratorModel(static_operator=st, operators=ops, signals=sigs,
                                   rotating_frame=frame)
                key = f"{tag}_{ftag}_{stag}"
                out[key + "_coeffs"] = np.array([m.signals(t) for t in times])
                out[key + "_eval"] = np.array([m.evaluate(t) for t in times])
                out[key + "_rhs1"] = np.array([m.evaluate_rhs(t, y1) for t in times])
                out[key + "_rhsm"] = np.array([m.evaluate_rhs(t, ym) for t in times])
                if ftag == "diag":
                    m.in_frame_basis = True
                    out[key + "_eval_fb"] = np.array([m.evaluate(t) for t in times])
                    out[key + "_rhs1_fb"] = np.array([m.evaluate_rhs(t, y1) for t in times])
        # Hamiltonian model, Hermitian frame R + R^dagger (test_hamiltonian_model.py:201-287)
        hops = np.array([herm(rng, n) for _ in range(k)])
        hstatic = herm(rng, n)
        hframe = herm(rng, n)
        out[f"{tag}_hops"], out[f"{tag}_hstatic"], out[f"{tag}_hframe"] = hops, hstatic, hframe
        hm = HamiltonianModel(static_operator=hstatic, operators=hops, signals=sigs,
                              rotating_frame=hframe)
        out[f"{tag}_ham_eval"] = np.array([hm.evaluate(t) for t in times])
        out[f"{tag}_ham_rhs1"] = np.array([hm.evaluate_rhs(t, y1) for t in times])
        out[f"{tag}_ham_rhsm"] = np.array([hm.evaluate_rhs(t, ym) for t in times])
        out[f"{tag}_ham_static_getter"] = hm.static_operator
        out[f"{tag}_ham_ops_getter"] = hm.operators
        hm2 = HamiltonianModel(static_operator=hstatic, operators=hops, signals=sigs,
                               rotating_frame=hstatic)
        out[f"{tag}_ham_selfframe_eval"] = np.array([hm2.evaluate(t) for t in times])
        out[f"{tag}_ham_selfframe_rhs1"] = np.array([hm2.evaluate_rhs(t, y1) for t in times])
    save("generator_model", **out)


# ---------------------------------------------------------------------------------------------
def gen_fixed_step():
    """a9-a11: step-size rule, RK4, scipy_expm m=1,2,3 (test_fixed_step_solvers.py:56-389)."""
    out = {}
    cases = [
        ([0.0, 1.0], None, 0.1), ([0.0, 1.0], None, 0.3), ([0.0, 1.0], [0.0, 0.25, 0.9, 1.0], 0.1),
        ([1.0, 0.0], None, 0.1), ([1.0, 0.0], [0.75, 0.3], 0.07), ([0.0, 0.05], None, 0.1),
        ([0.0, 5.0], None, 0.005), ([0.0, 10.0], None, 0.01), ([0.0, 1.5], [0.5, 0.5, 1.0], 0.5),
        ([0.0, 0.3], None, 0.1), ([0.0, 0.7], None, 0.1),
    ]
    for i, (ts, te, mdt) in enumerate(cases):
        tl, hl, nl = get_fixed_step_sizes(ts, te, mdt)
        out[f"sizes{i}_tspan"] = np.array(ts)
        out[f"sizes{i}_teval"] = np.array([] if te is None else te)
        out[f"sizes{i}_has_teval"] = np.array(te is not None)
        out[f"sizes{i}_maxdt"] = np.array(mdt)
        out[f"sizes{i}_t"], out[f"sizes{i}_h"], out[f"sizes{i}_n"] = tl, hl, nl
    out["n_sizes"] = np.array(len(cases))

    rng = np.random.default_rng(5213)
    n = 5
    a = crand(rng, n, n)
    g0 = a - a.conj().T
    b = crand(rng, n, n)
    g1 = b - b.conj().T
    y0 = crand(rng, n)
    ym = np.eye(n, dtype=complex)
    out["g0"], out["g1"], out["y0"] = g0, g1, y0

    def gen(t):
        return g0 + np.cos(1.3 * t) * g1

    def rhs(t, y):
        return gen(t) @ y

    for tag, ts, te, mdt in (("fw", [0.0, 1.0], None, 0.1),
                             ("te", [0.0, 1.0], [0.0, 0.33, 0.71, 1.0], 0.05),
                             ("bw", [1.0, 0.0], [0.8, 0.2], 0.1)):
        r = RK4_solver(rhs, ts, y0, max_dt=mdt, t_eval=te)
        out[f"rk4_{tag}_t"], out[f"rk4_{tag}_y"] = np.asarray(r.t), np.asarray(r.y)
        r = RK4_solver(rhs, ts, ym, max_dt=mdt, t_eval=te)
        out[f"rk4m_{tag}_y"] = np.asarray(r.y)
        for mo in (1, 2, 3):
            r = scipy_expm_solver(gen, ts, y0, max_dt=mdt, t_eval=te, magnus_order=mo)
            out[f"expm{mo}_{tag}_t"], out[f"expm{mo}_{tag}_y"] = np.asarray(r.t), np.asarray(r.y)
            r = scipy_expm_solver(gen, ts, ym, max_dt=mdt, t_eval=te, magnus_order=mo)
            out[f"expm{mo}m_{tag}_y"] = np.asarray(r.y)
    save("fixed_step", **out)


# ---------------------------------------------------------------------------------------------
def gen_solve_lmde():
    """a12 + cfg 1 + down-scaled cfg 2: solve_lmde end to end
    (test_solver_functions.py:46-217,247-356; test_solver_functions_interface.py:164-395)."""
    out = {}
    # cfg 1 in full (n=4, 1000 RK4 steps) -- BASELINE.json configs[0]
    c1 = workloads.config1()
    sigs = [Signal(1.0, 5.0), Signal(lambda t: np.exp(-((t - 5.0) ** 2) / 8.0), 5.0)]
    hm = HamiltonianModel(static_operator=c1["h_d"], operators=c1["ops"], signals=sigs,
                          rotating_frame=c1["h_d"])
    r = solve_lmde(hm, c1["t_span"], c1["y0"], method="RK4", max_dt=c1["max_dt"],
                   t_eval=[0.0, 2.5, 5.0, 7.5, 10.0])
    out["cfg1_t"], out["cfg1_y"] = np.asarray(r.t), np.asarray(r.y)
    for mo in (1, 2):
        r = solve_lmde(hm, c1["t_span"], c1["y0"], method="scipy_expm", max_dt=0.05,
                       magnus_order=mo)
        out[f"cfg1_expm{mo}_y"] = np.asarray(r.y)

    # random 7x7 framed model with a DiscreteSignal (test_solver_functions.py:76-115)
    rng = np.random.default_rng(3093)
    n, k = 7, 3
    hops = np.array([herm(rng, n) for _ in range(k)])
    hstatic = herm(rng, n)
    hframe = herm(rng, n)
    samples = crand(rng, 5)
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    sigs = [Signal(0.5, 1.0, 0.3), DiscreteSignal(dt=0.1, samples=samples, carrier_freq=1.0),
            Signal(lambda t: 0.3 * np.cos(t) + 0 * 1j, 0.0)]
    out["r7_hops"], out["r7_hstatic"], out["r7_hframe"] = hops, hstatic, hframe
    out["r7_samples"], out["r7_y0"] = samples, y0
    hm = HamiltonianModel(static_operator=hstatic, operators=hops, signals=sigs,
                          rotating_frame=hframe)
    r = solve_lmde(hm, [0.0, 0.5], y0, method="RK4", max_dt=1e-3, t_eval=[0.1, 0.3, 0.5])
    out["r7_rk4_t"], out["r7_rk4_y"] = np.asarray(r.t), np.asarray(r.y)
    r = solve_lmde(hm, [0.0, 0.5], np.eye(n, dtype=complex), method="RK4", max_dt=1e-3)
    out["r7_rk4_unitary"] = np.asarray(r.y)
    for mo in (1, 2, 3):
        r = solve_lmde(hm, [0.0, 0.5], y0, method="scipy_expm", max_dt=1e-2, magnus_order=mo,
                       t_eval=[0.1, 0.3, 0.5])
        out[f"r7_expm{mo}_y"] = np.asarray(r.y)
    r = solve_lmde(hm, [0.5, 0.0], y0, method="RK4", max_dt=1e-3)
    out["r7_rk4_backwards"] = np.asarray(r.y)
    hm.in_frame_basis = True
    r = solve_lmde(hm, [0.0, 0.5], y0, method="RK4", max_dt=1e-3)
    out["r7_rk4_in_fb"] = np.asarray(r.y)  # gauge dependent: informational only
    save("solve_lmde", **out)

    # down-scaled cfg 2/3: 6 qubits (n=64), k=6, RK4 + sweep of 4 instances through Solver
    cfg = workloads.schrodinger_config(n_qubits=6, n_drives=6, t_final=1.0, max_dt=0.01)
    out = {}
    solver = Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                    rotating_frame=cfg["h_d"])
    sig_lists = []
    for b in range(4):
        amps, phases = workloads.sweep_parameters(b, 6)
        sig_lists.append([
            Signal(lambda t, a=a: a * np.exp(-((t - cfg["t_final"] / 2) ** 2) / (2 * 1.0**2)),
                   nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    res = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sig_lists, method="RK4",
                       max_dt=cfg["max_dt"])
    out["sweep_y_final"] = np.array([r.y[-1] for r in res])
    tt = np.linspace(0, 1, 7)
    out["sweep_tt"] = tt
    out["sweep_coeffs"] = np.array([SignalList(s)(tt) for s in sig_lists])
    # single RHS evaluations of instance 0 on the in-frame-basis model (diagonal-frame twin is
    # gauge free: use frame = diag(H_d) for the per-eval golden)
    hm = HamiltonianModel(static_operator=cfg["h_d"], operators=cfg["ops"], signals=sig_lists[0],
                          rotating_frame=np.diag(cfg["h_d"]).real.copy())
    rng = np.random.default_rng(64)
    yv = crand(rng, 64)
    out["diag_yv"] = yv
    out["diag_rhs"] = np.array([hm.evaluate_rhs(t, yv) for t in (0.0, 0.37, 1.0)])
    out["diag_eval_t037"] = hm.evaluate(0.37)
    save("cfg2_small", **out)


# ---------------------------------------------------------------------------------------------
def gen_lindblad():
    """a13/a14: vectorised LindbladModel (test_lindblad_model.py:281-425,461-523;
    test_operator_collections.py:550-706; test_rotating_frame.py:440-488)."""
    rng = np.random.default_rng(9848)
    n = 4
    hops = np.array([herm(rng, n) for _ in range(2)])
    hstatic = herm(rng, n)
    nstat = crand(rng, 2, n, n)
    lops = crand(rng, 2, n, n)
    hframe = herm(rng, n)
    hsig = [Signal(0.7, 1.1, 0.2), Signal(lambda t: 0.4 * np.sin(2 * t) + 0j, 0.6)]
    dsig = [Signal(0.3, 0.0), Signal(lambda t: 0.2 + 0.1 * np.cos(t) + 0j, 0.0)]
    rho = crand(rng, n, n)
    rho = rho @ rho.conj().T
    rho /= np.trace(rho)
    times = np.array([0.0, 0.4, 1.7])
    out = dict(hops=hops, hstatic=hstatic, nstat=nstat, lops=lops, hframe=hframe, rho=rho,
               times=times)
    for ftag, frame in (("fr", hframe), ("diag", np.diag(hframe).real.copy()), ("nofr", None)):
        for vec in (True, False):
            m = LindbladModel(static_hamiltonian=hstatic, hamiltonian_operators=hops,
                              hamiltonian_signals=hsig, static_dissipators=nstat,
                              dissipator_operators=lops, dissipator_signals=dsig,
                              rotating_frame=frame, vectorized=vec)
            key = f"{ftag}_{'vec' if vec else 'mat'}"
            yin = rho.flatten(order="F") if vec else rho
            out[key + "_rhs"] = np.array([m.evaluate_rhs(t, yin) for t in times])
            if vec:
                out[key + "_eval"] = np.array([m.evaluate(t) for t in times])
                out[key + "_hcoeffs"] = np.array([m.signals[0](t) for t in times])
                out[key + "_dcoeffs"] = np.array([m.signals[1](t) for t in times])
                r = solve_lmde(m, [0.0, 0.6], yin, method="scipy_expm", max_dt=0.02,
                               t_eval=[0.2, 0.6])
                out[key + "_expm_y"] = np.asarray(r.y)
                r = solve_lmde(m, [0.0, 0.6], yin, method="RK4", max_dt=0.002)
                out[key + "_rk4_y"] = np.asarray(r.y)
    # subsets of operator groups (presence patterns)
    m = LindbladModel(hamiltonian_operators=hops, hamiltonian_signals=hsig,
                      static_dissipators=nstat, vectorized=True)
    out["pat_hs_eval"] = m.evaluate(0.4)
    m = LindbladModel(static_hamiltonian=hstatic, dissipator_operators=lops,
                      dissipator_signals=dsig, vectorized=True, rotating_frame=hframe)
    out["pat_sd_fr_eval"] = m.evaluate(0.4)
    out["pat_sd_fr_rhs"] = m.evaluate_rhs(0.4, rho.flatten(order="F"))

    # down-scaled cfg 4: 3 qubits, N=64, 3 drives, 2 static sigma^- dissipators
    cfg = workloads.lindblad_config(n_qubits=3, n_drives=3, n_diss=2, gamma=1e-2, t_final=1.0,
                                    max_dt=0.05)
    amps, phases = workloads.sweep_parameters(0, 3)
    sigs = [Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    for ftag, frame in (("nofr", None), ("diag", np.diag(cfg["h_d"]).real.copy())):
        s = Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                   static_dissipators=cfg["static_dissipators"], rotating_frame=frame,
                   vectorized=True)
        r = s.solve(t_span=cfg["t_span"], y0=cfg["rho0"].flatten(order="F"), signals=sigs,
                    method="scipy_expm", max_dt=cfg["max_dt"])
        out[f"cfg4s_{ftag}_y"] = np.asarray(r.y)
    save("lindblad", **out)


# ---------------------------------------------------------------------------------------------
def gen_solver_list():
    """a15: Solver list mode == individual solves (test_solver_classes.py:1389-1599)."""
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    out = {}
    s = Solver(hamiltonian_operators=[x], static_hamiltonian=5 * z, rotating_frame=5 * z)
    y0 = np.array([0.0, 1.0], dtype=complex)
    res = s.solve(t_span=[0.0, 0.4232], y0=y0, signals=[[Signal(1.0, 5.0)], [Signal(0.5, 5.0)]],
                  method="RK4", max_dt=0.001)
    out["ham_list_y"] = np.array([r.y for r in res])
    res = s.solve(t_span=[[0.0, 0.4232], [0.0, 1.23]], y0=y0, signals=[Signal(1.0, 5.0)],
                  method="RK4", max_dt=0.001)
    out["ham_tspan_list_y_final"] = np.array([r.y[-1] for r in res])
    res = s.solve(t_span=[0.0, 0.4232], y0=[y0, np.array([1.0, 0.0], dtype=complex)],
                  signals=[Signal(1.0, 5.0)], method="scipy_expm", max_dt=0.01)
    out["ham_y0_list_expm_y"] = np.array([r.y for r in res])
    sl = Solver(hamiltonian_operators=[x], static_hamiltonian=5 * z, rotating_frame=5 * z,
                static_dissipators=[0.01 * x], vectorized=True)
    rho0 = np.array([[0.0, 0.0], [0.0, 1.0]], dtype=complex)
    res = sl.solve(t_span=[0.0, 0.4232], y0=rho0.flatten(order="F"),
                   signals=[[Signal(1.0, 5.0)], [Signal(0.5, 5.0)]], method="scipy_expm",
                   max_dt=0.01)
    out["lind_list_y"] = np.array([r.y for r in res])
    save("solver_list", **out)


# ---------------------------------------------------------------------------------------------
def gen_rotating_frame():
    """a5/a6/a14: RotatingFrame maps (test_rotating_frame.py scenarios re-stated on seeded inputs)."""
    rng = np.random.default_rng(515)
    n = 4
    f = herm(rng, n)
    y = crand(rng, n, 2)
    op = crand(rng, n, n)
    sup = crand(rng, n * n, n * n)
    t = 0.731
    out = {"f": f, "y": y, "op": op, "sup": sup, "t": np.array(t)}
    rf = RotatingFrame(f)
    out["state_into"] = rf.state_into_frame(t, y)
    out["state_out"] = rf.state_out_of_frame(t, y)
    out["op_into"] = rf.operator_into_frame(t, op)
    out["op_out"] = rf.operator_out_of_frame(t, op)
    out["gen_into"] = rf.generator_into_frame(t, op)
    out["gen_out"] = rf.generator_out_of_frame(t, op)
    out["vec_map"] = rf.vectorized_map_into_frame(t, sup)
    # diagonal frame: the frame-basis flags are gauge free
    d = rng.normal(size=n)
    rd = RotatingFrame(d)
    out["d"] = d
    out["diag_state_into"] = rd.state_into_frame(t, y)
    out["diag_gen_into"] = rd.generator_into_frame(t, op, operator_in_frame_basis=True, return_in_frame_basis=True)
    out["diag_gen_out"] = rd.generator_out_of_frame(t, op, operator_in_frame_basis=True, return_in_frame_basis=True)
    out["diag_vec_map"] = rd.vectorized_map_into_frame(t, sup, operator_in_frame_basis=True, return_in_frame_basis=True)
    out["none_gen_out"] = RotatingFrame(None).generator_out_of_frame(t, op)
    save("rotating_frame", **out)


# ---------------------------------------------------------------------------------------------
def _labels_array(labels):
    """Multiset labels -> (M, max_order) int array of sorted indices, padded with -1."""
    lists = [sorted(list(lab)) for lab in labels]
    width = max(len(x) for x in lists)
    return np.array([x + [-1] * (width - len(x)) for x in lists], dtype=np.int64)


def gen_perturbative():
    """f4: DysonSolver / MagnusSolver (test_dyson_magnus_solvers.py:75-330): Chebyshev coefficients
    of the signals, the expansion terms, and the propagated states."""
    from oracle.ref_shim import load_reference_perturbation

    load_reference_perturbation()
    from qiskit_dynamics.solvers.perturbative_solvers import DysonSolver, MagnusSolver

    out = {}
    # ---- the qubit of the reference test (:86-140), shorter horizon
    r = 0.2
    sig_w = 0.399128 / r
    t_c = 3.5 * sig_w

    def gauss(t):
        return 1.0 * np.exp(-((t - t_c) ** 2) / (2 * sig_w**2))

    gauss_signal = Signal(gauss, carrier_freq=5.0)
    dt = 0.0125
    h_ops = 2 * np.pi * r * np.array([[[0.0, 1.0], [1.0, 0.0]]]) / 2
    h_static = 2 * np.pi * 5.0 * np.array([[1.0, 0.0], [0.0, -1.0]]) / 2
    rng = np.random.default_rng(21342)
    y_rand = crand(rng, 2, 2)
    out["q1_params"] = np.array([r, sig_w, t_c, dt, 5.0])
    out["q1_ops"] = -1j * h_ops
    out["q1_frame"] = -1j * h_static
    out["q1_y_rand"] = y_rand
    for name, cls, order in (("dyson", DysonSolver, 6), ("magnus", MagnusSolver, 3)):
        sol = cls(operators=-1j * h_ops, rotating_frame=-1j * h_static, dt=dt, carrier_freqs=[5.0],
                  chebyshev_orders=[1], expansion_order=order, integration_method="DOP853", atol=1e-12,
                  rtol=1e-12)
        poly = sol.model.expansion_polynomial
        out[f"q1_{name}_labels"] = _labels_array(poly.monomial_labels)
        out[f"q1_{name}_terms"] = np.asarray(poly.array_coefficients)
        out[f"q1_{name}_udt"] = np.asarray(sol.model.Udt)
        out[f"q1_{name}_cheb_t0"] = np.asarray(sol.model.approximate_signals([gauss_signal], 0.0, 120))
        out[f"q1_{name}_cheb_t1"] = np.asarray(sol.model.approximate_signals([gauss_signal], 3.1, 50))
        out[f"q1_{name}_y_eye"] = sol.solve(t0=0.0, n_steps=120, y0=np.eye(2, dtype=complex),
                                            signals=[gauss_signal]).y[-1]
        out[f"q1_{name}_y_rand_t1"] = sol.solve(t0=3.1, n_steps=50, y0=y_rand, signals=[gauss_signal]).y[-1]
        out[f"q1_{name}_y_vec"] = sol.solve(t0=0.0, n_steps=30, y0=y_rand[:, 0], signals=[gauss_signal]).y[-1]
        c = np.asarray(sol.model.approximate_signals([gauss_signal], 3.1, 3))[:, 1]
        out[f"q1_{name}_eval_c"] = c
        out[f"q1_{name}_eval"] = np.asarray(sol.model.evaluate(c))
    # ---- a 3-level transmon with two drives: mixed Chebyshev orders, one real-only envelope, extra labels
    dim = 3
    a = np.diag(np.sqrt(np.arange(1, dim)), 1)
    num = np.diag(np.arange(dim)).astype(float)
    h0 = 2 * np.pi * 4.9 * num + np.pi * (-0.33) * num @ (num - np.eye(dim))
    hd1 = 2 * np.pi * 0.05 * (a + a.T)
    hd2 = 2 * np.pi * 0.02 * num
    sig_a = Signal(lambda t: 0.8 * np.exp(-((t - 1.0) ** 2) / 0.5) * np.exp(0.3j * t), carrier_freq=4.9, phase=0.2)
    sig_b = Signal(lambda t: 0.4 * np.cos(0.7 * t) + 0j, carrier_freq=0.0)
    sig_c = Signal(lambda t: 0.5 * np.exp(-((t - 0.7) ** 2) / 0.3) + 0j, carrier_freq=4.95, phase=-0.4)
    out["t3_ops"] = np.array([-1j * hd1, -1j * hd2])
    out["t3_frame"] = -1j * h0
    y3 = crand(rng, 3, 2)
    out["t3_y0"] = y3
    for name, cls in (("dyson", DysonSolver), ("magnus", MagnusSolver)):
        sol = cls(operators=[-1j * hd1, -1j * hd2], rotating_frame=-1j * h0, dt=0.02, carrier_freqs=[4.9, 0.0],
                  chebyshev_orders=[1, 0], expansion_order=2, expansion_labels=[[0, 0, 1], [0, 1, 4]],
                  include_imag=[True, False], integration_method="DOP853", atol=1e-12, rtol=1e-12)
        poly = sol.model.expansion_polynomial
        out[f"t3_{name}_labels"] = _labels_array(poly.monomial_labels)
        out[f"t3_{name}_terms"] = np.asarray(poly.array_coefficients)
        out[f"t3_{name}_udt"] = np.asarray(sol.model.Udt)
        out[f"t3_{name}_cheb"] = np.asarray(sol.model.approximate_signals([sig_a, sig_b], 0.1, 60))
        res = sol.solve(t0=0.1, n_steps=60, y0=[np.eye(3, dtype=complex), y3],
                        signals=[[sig_a, sig_b], [sig_c, sig_b]])
        out[f"t3_{name}_y_list0"] = res[0].y[-1]
        out[f"t3_{name}_y_list1"] = res[1].y[-1]
        out[f"t3_{name}_t"] = np.asarray(res[0].t)
    save("perturbative", **out)


# ---------------------------------------------------------------------------------------------
def lab_frame_signals(n_drives, carrier, instance):
    """Constant-envelope drives of sweep instance `instance` (amplitude, carrier, phase per drive)."""
    amps, phases = workloads.sweep_parameters(instance, n_drives)
    return [(float(a), float(nu), float(ph)) for a, nu, ph in zip(amps, carrier, phases)]


def gen_lab_frame():
    """Large ||h G||: models WITHOUT a rotating frame, `scipy_expm` with Magnus orders 1 and 2
    (fixed_step_solvers.py:80-108,308-403 through solve_lmde / Solver list mode).  These are the inputs on which
    the device takes the Chebyshev expm action, the work-list kernels (8 qubits: block-sparse computational
    basis), the persistent small-system kernel (3 qubits) and the device-built Lindblad superoperators."""
    out = {}
    # (a) 8-qubit chain, n = 256, lab frame: a sweep of 3 instances
    cfg = workloads.schrodinger_config(n_qubits=8, n_drives=4, t_final=1.0, max_dt=0.05)
    rng = np.random.default_rng(808)
    y0 = crand(rng, 256)
    y0 /= np.linalg.norm(y0)
    out["q8_y0"] = y0
    solver = Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"])
    sig_lists = [[Signal(a, nu, ph) for a, nu, ph in lab_frame_signals(4, cfg["carrier"], b)] for b in range(3)]
    for mo in (1, 2):
        res = solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sig_lists, method="scipy_expm", max_dt=0.05,
                           magnus_order=mo)
        out[f"q8_expm{mo}_y"] = np.array([r.y[-1] for r in res])
    # the same model in the diagonal frame (block sparse, small norm): the frame handling on the same inputs
    solver_d = Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                      rotating_frame=np.diag(cfg["h_d"]).real.copy())
    res = solver_d.solve(t_span=[0.0, 0.2], y0=y0, signals=sig_lists, method="scipy_expm", max_dt=0.05, magnus_order=1)
    out["q8_diag_expm1_y"] = np.array([r.y[-1] for r in res])
    # RK4 on the same block-sparse stacks (lab frame and diagonal frame)
    res = solver.solve(t_span=[0.0, 0.1], y0=y0, signals=sig_lists, method="RK4", max_dt=0.002)
    out["q8_rk4_y"] = np.array([r.y[-1] for r in res])
    res = solver_d.solve(t_span=[0.0, 0.1], y0=y0, signals=sig_lists, method="RK4", max_dt=0.002)
    out["q8_diag_rk4_y"] = np.array([r.y[-1] for r in res])
    # (b) 3-qubit chain, n = 8, lab frame, sweep of 6 instances
    cfg3 = workloads.schrodinger_config(n_qubits=3, n_drives=3, t_final=1.0, max_dt=0.04)
    y3 = crand(rng, 8)
    y3 /= np.linalg.norm(y3)
    out["q3_y0"] = y3
    solver3 = Solver(static_hamiltonian=cfg3["h_d"], hamiltonian_operators=cfg3["ops"])
    sig3 = [[Signal(a, nu, ph) for a, nu, ph in lab_frame_signals(3, cfg3["carrier"], b)] for b in range(6)]
    for mo in (1, 2):
        res = solver3.solve(t_span=[0.0, 0.4], y0=y3, signals=sig3, method="scipy_expm", max_dt=0.04, magnus_order=mo)
        out[f"q3_expm{mo}_y"] = np.array([r.y[-1] for r in res])
    # (c) 3-qubit vectorised Lindbladian (N = 64), no frame, weak dissipation
    lc = workloads.lindblad_config(n_qubits=3, n_drives=3, n_diss=3, gamma=1e-2, t_final=1.0, max_dt=0.05)
    sigl = [Signal(a, nu, ph) for a, nu, ph in lab_frame_signals(3, lc["carrier"], 0)]
    m = LindbladModel(static_hamiltonian=lc["h_d"], hamiltonian_operators=lc["ops"], hamiltonian_signals=sigl,
                      static_dissipators=lc["static_dissipators"], vectorized=True)
    rho0 = crand(rng, 8, 8)
    rho0 = rho0 @ rho0.conj().T
    rho0 /= np.trace(rho0)
    out["l3_rho0"] = rho0
    r = solve_lmde(m, [0.0, 0.2], rho0.flatten(order="F"), method="scipy_expm", max_dt=0.05)
    out["l3_expm1_y"] = np.asarray(r.y[-1])
    # (d) sweep of non-vectorised open systems through Solver list mode (RK4 on n x n density matrices):
    # 3 instances of a 3-qubit chain in the diagonal frame, static + one dynamic dissipator, own initial states
    frame3 = np.diag(lc["h_d"]).real.copy()
    sm = lc["static_dissipators"]
    solver_nv = Solver(static_hamiltonian=lc["h_d"], hamiltonian_operators=lc["ops"], static_dissipators=sm[:2],
                       dissipator_operators=sm[2:3], rotating_frame=frame3, vectorized=False)
    rhos, sweeps = [], []
    for b in range(3):
        x = crand(rng, 8, 8)
        x = x @ x.conj().T
        rhos.append(x / np.trace(x))
        sweeps.append(([Signal(a, nu, ph) for a, nu, ph in lab_frame_signals(3, lc["carrier"], b)],
                       [Signal(0.5 + 0.25 * b, 0.0)]))
    out["nv_rho0"] = np.array(rhos)
    res = solver_nv.solve(t_span=[0.0, 0.2], y0=rhos, signals=sweeps, method="RK4", max_dt=0.01)
    out["nv_rk4_y"] = np.array([r.y[-1] for r in res])
    # (e) LindbladModel.from_hamiltonian (lindblad_model.py:214-260) on framed Hamiltonian models
    rngf = np.random.default_rng(2140)
    hs, hops = herm(rngf, 4), np.array([herm(rngf, 4), herm(rngf, 4)])
    frm = herm(rngf, 4)
    ldis = crand(rngf, 1, 4, 4)
    rho = herm(rngf, 4)
    out["fh_hs"], out["fh_hops"], out["fh_frame"], out["fh_l"], out["fh_rho"] = hs, hops, frm, ldis, rho
    for tag, frame in (("nofr", None), ("fr", frm), ("diag", np.diag(frm).real.copy())):
        hm = HamiltonianModel(static_operator=hs, operators=hops, signals=[Signal(0.5, 1.0), Signal(0.3, 2.0, 0.4)],
                              rotating_frame=frame)
        for vec in (False, True):
            lm = LindbladModel.from_hamiltonian(hm, static_dissipators=ldis, vectorized=vec)
            yin = rho.flatten(order="F") if vec else rho
            out[f"fh_{tag}_{'vec' if vec else 'mat'}_rhs"] = np.array([lm.evaluate_rhs(t, yin) for t in (0.0, 0.3)])
    save("lab_frame", **out)


# ---------------------------------------------------------------------------------------------
def gen_interface():
    """a12: y0 into the frame basis / results out of it for Hamiltonian, Lindblad and vectorised Lindblad models
    (test_solver_functions_interface.py:164-395: X drive, Z static, Y dissipator, frame 1.2 X - 3.132 Y,
    y0 = (3.43, 1.31)), and the Solver sanity scenario `vectorised == non-vectorised` (test_solver_classes.py:461-697).
    Models evaluated at t = 231.232 (the reference test's time) and solved end to end; results are stored OUT of the
    frame basis (gauge invariant)."""
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    y = np.array([[0, -1j], [1j, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    frame = 1.2 * x - 3.132 * y
    y0 = np.array([3.43, 1.31], dtype=complex)
    rho0 = np.outer(y0, y0.conj()) / np.vdot(y0, y0)
    t = 231.232
    out = {"frame": frame, "y0": y0, "rho0": rho0, "t": np.array(t)}
    for ftag, fr in (("nofr", None), ("fr", frame)):
        hm = HamiltonianModel(operators=[x], signals=[Signal(1.0, 5.0)], static_operator=z, rotating_frame=fr)
        out[f"{ftag}_ham_eval"] = hm(t)
        out[f"{ftag}_ham_rhs"] = hm(t, y0)
        for method, kw in (("RK4", dict(max_dt=1e-3)), ("scipy_expm", dict(max_dt=1e-2)),
                           ("scipy_expm", dict(max_dt=1e-2, magnus_order=2))):
            hm2 = HamiltonianModel(operators=[x], signals=[Signal(1.0, 5.0)], static_operator=z, rotating_frame=fr)
            r = solve_lmde(hm2, t_span=[0.0, 1.1], y0=y0, method=method, t_eval=[0.3, 1.1], **kw)
            out[f"{ftag}_ham_{method}_{kw.get('magnus_order', 1)}_y"] = np.array(r.y)
        for vec in (False, True):
            lm = LindbladModel(hamiltonian_operators=[x], hamiltonian_signals=[Signal(1.0, 5.0)], static_hamiltonian=z,
                               static_dissipators=[y], rotating_frame=fr, vectorized=vec)
            yin = rho0.flatten(order="F") if vec else rho0
            out[f"{ftag}_lind_{'vec' if vec else 'mat'}_rhs"] = lm(t, yin)
            if vec:
                out[f"{ftag}_lind_vec_eval"] = lm(t)
            lm2 = LindbladModel(hamiltonian_operators=[x], hamiltonian_signals=[Signal(1.0, 5.0)], static_hamiltonian=z,
                                static_dissipators=[y], rotating_frame=fr, vectorized=vec)
            r = solve_lmde(lm2, t_span=[0.0, 0.7], y0=yin, method="RK4", max_dt=1e-3) if not vec else \
                solve_lmde(lm2, t_span=[0.0, 0.7], y0=yin, method="scipy_expm", max_dt=1e-2)
            out[f"{ftag}_lind_{'vec' if vec else 'mat'}_y"] = np.array(r.y)
    # Solver: weak 0.01 X dissipator, vectorised and not, pi-pulse style drive (test_solver_classes.py:461-697)
    for vec in (False, True):
        s = Solver(hamiltonian_operators=[x / 2], static_hamiltonian=5 * z, rotating_frame=5 * z,
                   static_dissipators=[0.01 * x], vectorized=vec)
        rho = np.array([[0.0, 0.0], [0.0, 1.0]], dtype=complex)
        r = s.solve(t_span=[0.0, 1.0], y0=rho.flatten(order="F") if vec else rho, signals=[Signal(1.0, 5.0 / np.pi)],
                    method="RK4" if not vec else "scipy_expm", max_dt=1e-3 if not vec else 1e-2)
        out[f"solver_{'vec' if vec else 'mat'}_y"] = np.array(r.y)
    save("interface", **out)


# ---------------------------------------------------------------------------------------------
def gen_adaptive():
    """Callers of the RHS either side of the hot path: solve_ode / solve_lmde with scipy's adaptive methods
    (solvers/scipy_solve_ivp.py:31-84 via solver_functions.py:200-207, 349-364) on the random framed 7 x 7 model of
    test_solver_functions.py:76-115 and on the Lindblad models of the interface scenario."""
    rng = np.random.default_rng(3093)
    n, k = 7, 3
    hops = np.array([herm(rng, n) for _ in range(k)])
    hstatic = herm(rng, n)
    hframe = herm(rng, n)
    samples = crand(rng, 5)
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    sigs = [Signal(0.5, 1.0, 0.3), DiscreteSignal(dt=0.1, samples=samples, carrier_freq=1.0),
            Signal(lambda t: 0.3 * np.cos(t) + 0 * 1j, 0.0)]
    out = {"r7_hops": hops, "r7_hstatic": hstatic, "r7_hframe": hframe, "r7_samples": samples, "r7_y0": y0}
    for method in ("RK45", "RK23", "DOP853", "BDF"):   # (the real-embedded LSODA / Radau wrappers fail inside the reference)
        hm = HamiltonianModel(static_operator=hstatic, operators=hops, signals=sigs, rotating_frame=hframe)
        tol = 1e-10 if method in ("RK45", "DOP853") else 1e-7
        r = solve_lmde(hm, [0.0, 0.5], y0, method=method, t_eval=[0.1, 0.3, 0.5], atol=tol, rtol=tol)
        out[f"r7_{method}_y"] = np.asarray(r.y)
    hm = HamiltonianModel(static_operator=hstatic, operators=hops, signals=sigs, rotating_frame=hframe)
    r = solve_lmde(hm, [0.0, 0.3], np.eye(n, dtype=complex), method="DOP853", atol=1e-10, rtol=1e-10)
    out["r7_DOP853_unitary"] = np.asarray(r.y)
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    yy = np.array([[0, -1j], [1j, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    frame = 1.2 * x - 3.132 * yy
    v = np.array([3.43, 1.31], dtype=complex)
    rho0 = np.outer(v, v.conj()) / np.vdot(v, v)
    out["rho0"], out["frame"] = rho0, frame
    for vec in (False, True):
        lm = LindbladModel(hamiltonian_operators=[x], hamiltonian_signals=[Signal(1.0, 5.0)], static_hamiltonian=z,
                           static_dissipators=[yy], rotating_frame=frame, vectorized=vec)
        yin = rho0.flatten(order="F") if vec else rho0
        r = solve_lmde(lm, [0.0, 0.7], yin, method="DOP853", atol=1e-10, rtol=1e-10)
        out[f"lind_{'vec' if vec else 'mat'}_DOP853_y"] = np.asarray(r.y)
    save("adaptive", **out)


# ---------------------------------------------------------------------------------------------
if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "perturbative":
        gen_perturbative()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "lab_frame":
        gen_lab_frame()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "interface":
        gen_interface()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "adaptive":
        gen_adaptive()
        sys.exit(0)
    gen_collection()
    gen_signals()
    gen_generator_model()
    gen_fixed_step()
    gen_solve_lmde()
    gen_lindblad()
    gen_solver_list()
    gen_rotating_frame()
    gen_perturbative()
    gen_lab_frame()
    gen_interface()
    gen_adaptive()
