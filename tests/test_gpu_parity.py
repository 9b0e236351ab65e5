"""Parity of the HIP path (through the C-ABI of libmidyn.so) against
  (1) golden vectors captured from the real reference (tests/golden/*.npz) and
  (2) the CPU oracle (oracle/dynamics_oracle.py) on seeded inputs, up to BASELINE.json sizes.

Tolerances (fp64): single evaluations max|d| <= 1e-12 (1 + max|ref|); fixed-step
solves <= 1e-9; expm ||E - E_ref||_1 / ||E_ref||_1 <= 1e-12 and unitarity <= 1e-12 * n.
All tests need a real MI355X: run with `pytest -m gpu`.
"""
import numpy as np
import pytest
import scipy.linalg

from conftest import assert_close

pytestmark = pytest.mark.gpu

EVAL_TOL = 1e-12
SOLVE_TOL = 1e-9


@pytest.fixture(scope="module")
def qd():
    import qiskit_dynamics_amd as q

    q.default_context()  # raises loudly when libmidyn / the GPU is missing
    return q


def crand(rng, *shape):
    return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)


# ------------------------------------------------------------------------------------------------
# MFMA zgemm: layout, transposition, padding, both tile configurations
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tile", [0, 64, 128])
@pytest.mark.parametrize("shape", [(5, 7, 3), (64, 64, 64), (130, 70, 200), (256, 384, 128),
                                   (1, 1, 1), (16, 300, 17)])
def test_zgemm(qd, shape, tile):
    m, n, k = shape
    rng = np.random.default_rng(m * 1000 + n * 10 + k)
    a = crand(rng, m, k)
    b = crand(rng, k, n)  # asymmetric on purpose (catches row/col swaps)
    ctx = qd.default_context()
    ctx.set_option("force_tile", tile)
    try:
        c = ctx.zgemm(a, b)
    finally:
        ctx.set_option("force_tile", 0)
    assert_close(c, a @ b, 1e-13)


def test_zgemm_identity_asymmetric(qd):
    """A = I with an asymmetric B: detects an output transpose (guide rule: A=I check)."""
    n = 96
    b = np.arange(n * n, dtype=float).reshape(n, n) + 1j * np.arange(n * n, dtype=float).reshape(n, n)[::-1]
    c = qd.default_context().zgemm(np.eye(n, dtype=complex), b)
    assert_close(c, b, 0)


# ------------------------------------------------------------------------------------------------
# a1/a2 operator collection
# ------------------------------------------------------------------------------------------------
def test_collection_golden(qd, golden):
    g = golden("collection")
    ctx = qd.default_context()
    for tag, st in (("full", g["static"]), ("nostatic", None)):
        stack = qd.Stack(ctx, g["ops"], st, None)
        for i, c in enumerate(g["coeffs"]):
            assert_close(stack.eval_generator(c, 0.0), g[f"{tag}_eval"][i], EVAL_TOL)
            assert_close(stack.eval_rhs(c, 0.0, g["y1"]), g[f"{tag}_rhs1"][i], EVAL_TOL)
            assert_close(stack.eval_rhs(c, 0.0, g["ym"]), g[f"{tag}_rhsm"][i], EVAL_TOL)
    stack = qd.Stack(ctx, None, g["static"], None)
    assert_close(stack.eval_generator(None, 0.0), g["staticonly_eval"], 0)
    assert_close(stack.eval_rhs(None, 0.0, g["y1"]), g["staticonly_rhs1"], EVAL_TOL)
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    stack = qd.Stack(ctx, np.array([x, 1j * z]), None, None)
    assert_close(stack.eval_generator(np.array([1.0, 2.0]), 0.0), g["pauli_eval"], 0)


def test_zero_plane_skipping_is_exact(qd):
    """Purely real / purely imaginary / zero operators take the reduced MFMA paths; results must
    match the dense path."""
    rng = np.random.default_rng(7)
    n, k, m = 48, 4, 5
    ops = np.zeros((k, n, n), dtype=complex)
    ops[0] = rng.normal(size=(n, n))              # real only
    ops[1] = 1j * rng.normal(size=(n, n))         # imaginary only
    ops[2] = 0.0                                  # exactly zero
    ops[3] = crand(rng, n, n)
    static = 1j * rng.normal(size=(n, n))
    c = rng.uniform(-1, 1, k)
    y = crand(rng, n, m)
    ref = (np.tensordot(c, ops, axes=1) + static) @ y
    ctx = qd.default_context()
    stack = qd.Stack(ctx, ops, static, None)
    assert stack.n_active_segments == 4
    for skip in (1, 0):
        ctx.set_option("skip_zero_planes", skip)
        try:
            assert_close(stack.eval_rhs(c, 0.3, y), ref, EVAL_TOL)
            assert_close(stack.eval_rhs(c, 0.3, y[:, 0]), ref[:, 0], EVAL_TOL)
        finally:
            ctx.set_option("skip_zero_planes", 1)


# ------------------------------------------------------------------------------------------------
# a3-a7 generator / Hamiltonian model
# ------------------------------------------------------------------------------------------------
def _sigs(qd, g, tag):
    return [qd.Signal(a, c, p) for a, c, p in zip(g[f"{tag}_amp"], g[f"{tag}_carrier"], g[f"{tag}_phase"])]


@pytest.mark.parametrize("tag", ["r5", "r10"])
def test_generator_model_golden(qd, golden, tag):
    g = golden("generator_model")
    ops, static, frame = g[f"{tag}_ops"], g[f"{tag}_static"], g[f"{tag}_frame"]
    times, y1, ym = g[f"{tag}_times"], g[f"{tag}_y1"], g[f"{tag}_ym"]
    sigs = _sigs(qd, g, tag)
    for ftag, fr in (("fr", frame), ("diag", np.diag(frame).copy()), ("nofr", None)):
        for stag, st in (("st", static), ("nost", None)):
            key = f"{tag}_{ftag}_{stag}"
            m = qd.GeneratorModel(static_operator=st, operators=ops, signals=sigs, rotating_frame=fr)
            for i, t in enumerate(times):
                assert_close(m.signals(t), g[key + "_coeffs"][i], 1e-15)
                assert_close(m.evaluate(t), g[key + "_eval"][i], EVAL_TOL)
                assert_close(m(t), g[key + "_eval"][i], EVAL_TOL)
                assert_close(m.evaluate_rhs(t, y1), g[key + "_rhs1"][i], EVAL_TOL)
                assert_close(m(t, ym), g[key + "_rhsm"][i], EVAL_TOL)
            if ftag == "diag":
                m.in_frame_basis = True
                for i, t in enumerate(times):
                    assert_close(m.evaluate(t), g[key + "_eval_fb"][i], EVAL_TOL)
                    assert_close(m.evaluate_rhs(t, y1), g[key + "_rhs1_fb"][i], EVAL_TOL)
    hm = qd.HamiltonianModel(static_operator=g[f"{tag}_hstatic"], operators=g[f"{tag}_hops"],
                             signals=sigs, rotating_frame=g[f"{tag}_hframe"])
    for i, t in enumerate(times):
        assert_close(hm.evaluate(t), g[f"{tag}_ham_eval"][i], EVAL_TOL)
        assert_close(hm.evaluate_rhs(t, y1), g[f"{tag}_ham_rhs1"][i], EVAL_TOL)
        assert_close(hm.evaluate_rhs(t, ym), g[f"{tag}_ham_rhsm"][i], EVAL_TOL)
    assert_close(hm.static_operator, g[f"{tag}_ham_static_getter"], 1e-12)
    assert_close(hm.operators, g[f"{tag}_ham_ops_getter"], 1e-12)
    hm2 = qd.HamiltonianModel(static_operator=g[f"{tag}_hstatic"], operators=g[f"{tag}_hops"],
                              signals=sigs, rotating_frame=g[f"{tag}_hstatic"])
    for i, t in enumerate(times):
        assert_close(hm2.evaluate(t), g[f"{tag}_ham_selfframe_eval"][i], EVAL_TOL)
        assert_close(hm2.evaluate_rhs(t, y1), g[f"{tag}_ham_selfframe_rhs1"][i], EVAL_TOL)


def test_generator_kat(qd, golden):
    g = golden("generator_model")
    m = qd.GeneratorModel(operators=g["kat_ops"],
                          signals=[qd.Signal(1.0, f) for f in g["kat_carrier"]])
    assert_close(m.evaluate(2.0), g["kat_eval_t2"], 1e-15)
    assert_close(m.evaluate_rhs(2.0, np.array([0.2, 0.5])), g["kat_rhs_t2"], 1e-15)


def test_model_errors(qd):
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    with pytest.raises(qd.DynamicsError):
        qd.GeneratorModel()
    with pytest.raises(qd.DynamicsError):
        qd.HamiltonianModel(operators=[x + 1j * np.triu(np.ones((2, 2)), 1)])
    m = qd.HamiltonianModel(operators=[x])
    with pytest.raises(qd.DynamicsError):
        m.evaluate(0.1)  # no signals
    with pytest.raises(qd.DynamicsError):
        m.signals = [qd.Signal(1.0), qd.Signal(1.0)]
    with pytest.raises(qd.DynamicsError):
        qd.GeneratorModel(operators=[x], array_library="jax")
    with pytest.raises(qd.DynamicsError):
        qd.RotatingFrame(np.array([[0, 1], [0, 0]], dtype=complex))


# ------------------------------------------------------------------------------------------------
# a9-a11 fixed-step solvers on a time-dependent 5x5 generator
# ------------------------------------------------------------------------------------------------
CASES = {"fw": ([0.0, 1.0], None, 0.1), "te": ([0.0, 1.0], [0.0, 0.33, 0.71, 1.0], 0.05),
         "bw": ([1.0, 0.0], [0.8, 0.2], 0.1)}


@pytest.mark.parametrize("tag", list(CASES))
def test_fixed_step_golden(qd, golden, tag):
    g = golden("fixed_step")
    ts, te, mdt = CASES[tag]
    m = qd.GeneratorModel(static_operator=g["g0"], operators=[g["g1"]],
                          signals=[qd.Signal(lambda t: np.cos(1.3 * t) + 0j)])
    r = qd.solve_lmde(m, ts, g["y0"], method="RK4", max_dt=mdt, t_eval=te)
    assert_close(r.t, g[f"rk4_{tag}_t"], 0)
    assert_close(r.y, g[f"rk4_{tag}_y"], SOLVE_TOL)
    r = qd.solve_lmde(m, ts, np.eye(5, dtype=complex), method="hip_RK4", max_dt=mdt, t_eval=te)
    assert_close(r.y, g[f"rk4m_{tag}_y"], SOLVE_TOL)
    for mo in (1, 2, 3):
        r = qd.solve_lmde(m, ts, g["y0"], method="scipy_expm", max_dt=mdt, t_eval=te, magnus_order=mo)
        assert_close(r.t, g[f"expm{mo}_{tag}_t"], 0)
        assert_close(r.y, g[f"expm{mo}_{tag}_y"], SOLVE_TOL)
        r = qd.solve_lmde(m, ts, np.eye(5, dtype=complex), method="hip_expm", max_dt=mdt, t_eval=te,
                          magnus_order=mo)
        assert_close(r.y, g[f"expm{mo}m_{tag}_y"], SOLVE_TOL)


def test_solver_error_surface(qd):
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    m = qd.HamiltonianModel(operators=[x], signals=[qd.Signal(1.0)])
    y0 = np.array([1.0, 0.0], dtype=complex)
    with pytest.raises(qd.DynamicsError):
        qd.solve_lmde(m, [0, 1], y0, method="jax_odeint", max_dt=0.1)
    with pytest.raises(qd.DynamicsError):
        qd.solve_lmde(m, [0, 1], y0, method="scipy_expm", max_dt=0.1, magnus_order=4)
    with pytest.raises(qd.DynamicsError):
        qd.solve_lmde(lambda t: x, [0, 1], y0, method="RK4", max_dt=0.1)
    lm = qd.LindbladModel(hamiltonian_operators=[x], hamiltonian_signals=[qd.Signal(1.0)],
                          static_dissipators=[0.1 * x], vectorized=False)
    with pytest.raises(qd.DynamicsError):
        qd.solve_lmde(lm, [0, 1], np.eye(2, dtype=complex) / 2, method="scipy_expm", max_dt=0.1)
    with pytest.raises(qd.DynamicsError):
        qd.solve_lmde(m, [0, 1], np.ones(3, dtype=complex), method="RK4", max_dt=0.1)
    with pytest.raises(ValueError):
        qd.solve_lmde(m, [0, 1], y0, method="RK4", max_dt=0.1, t_eval=[0.5, 1.5])


# ------------------------------------------------------------------------------------------------
# a12 solve_lmde end to end
# ------------------------------------------------------------------------------------------------
def test_cfg1_golden(qd, golden):
    """BASELINE.json configs[0] shape (2 qubits, 3 ops, RK4) -- in full, 1000 steps."""
    from qiskit_dynamics_amd import workloads

    g = golden("solve_lmde")
    c1 = workloads.config1()
    sigs = [qd.Signal(1.0, 5.0), qd.Signal(lambda t: np.exp(-((t - 5.0) ** 2) / 8.0), 5.0)]
    hm = qd.HamiltonianModel(static_operator=c1["h_d"], operators=c1["ops"], signals=sigs,
                             rotating_frame=c1["h_d"])
    r = qd.solve_lmde(hm, c1["t_span"], c1["y0"], method="RK4", max_dt=c1["max_dt"],
                      t_eval=[0.0, 2.5, 5.0, 7.5, 10.0])
    assert_close(r.t, g["cfg1_t"], 0)
    assert_close(r.y, g["cfg1_y"], SOLVE_TOL)
    for mo in (1, 2):
        r = qd.solve_lmde(hm, c1["t_span"], c1["y0"], method="scipy_expm", max_dt=0.05, magnus_order=mo)
        assert_close(r.y, g[f"cfg1_expm{mo}_y"], SOLVE_TOL)


def test_random_framed_model_golden(qd, golden):
    g = golden("solve_lmde")
    sigs = [qd.Signal(0.5, 1.0, 0.3), qd.DiscreteSignal(dt=0.1, samples=g["r7_samples"], carrier_freq=1.0),
            qd.Signal(lambda t: 0.3 * np.cos(t) + 0 * 1j, 0.0)]
    hm = qd.HamiltonianModel(static_operator=g["r7_hstatic"], operators=g["r7_hops"], signals=sigs,
                             rotating_frame=g["r7_hframe"])
    r = qd.solve_lmde(hm, [0.0, 0.5], g["r7_y0"], method="RK4", max_dt=1e-3, t_eval=[0.1, 0.3, 0.5])
    assert_close(r.t, g["r7_rk4_t"], 0)
    assert_close(r.y, g["r7_rk4_y"], SOLVE_TOL)
    r = qd.solve_lmde(hm, [0.0, 0.5], np.eye(7, dtype=complex), method="RK4", max_dt=1e-3)
    assert_close(r.y, g["r7_rk4_unitary"], SOLVE_TOL)
    for mo in (1, 2, 3):
        r = qd.solve_lmde(hm, [0.0, 0.5], g["r7_y0"], method="scipy_expm", max_dt=1e-2,
                          magnus_order=mo, t_eval=[0.1, 0.3, 0.5])
        assert_close(r.y, g[f"r7_expm{mo}_y"], SOLVE_TOL)
    r = qd.solve_lmde(hm, [0.5, 0.0], g["r7_y0"], method="RK4", max_dt=1e-3)
    assert_close(r.y, g["r7_rk4_backwards"], SOLVE_TOL)
    assert hm.in_frame_basis is False  # solve does not leave the model mutated


def _sweep_signals(qd, cfg, b, k, t_final):
    from qiskit_dynamics_amd import workloads

    amps, phases = workloads.sweep_parameters(b, k)
    return [qd.Signal(lambda t, a=a: a * np.exp(-((t - t_final / 2) ** 2) / (2 * 1.0**2)), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]


def test_cfg2_small_golden(qd, golden):
    """Down-scaled cfg 2/3 (6 qubits, n=64, k=6): Solver sweep of 4 instances, one batched solve."""
    from qiskit_dynamics_amd import workloads

    g = golden("cfg2_small")
    cfg = workloads.schrodinger_config(n_qubits=6, n_drives=6, t_final=1.0, max_dt=0.01)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       rotating_frame=cfg["h_d"])
    sig_lists = [_sweep_signals(qd, cfg, b, 6, 1.0) for b in range(4)]
    assert_close(np.array([qd.SignalList(s)(g["sweep_tt"]) for s in sig_lists]), g["sweep_coeffs"], 0)
    res = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sig_lists, method="RK4",
                       max_dt=cfg["max_dt"])
    assert isinstance(res, list) and len(res) == 4
    assert_close(np.array([r.y[-1] for r in res]), g["sweep_y_final"], SOLVE_TOL)
    # list result == individual solves (Solver list-mode contract)
    one = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sig_lists[2], method="RK4",
                       max_dt=cfg["max_dt"])
    assert_close(one.y[-1], res[2].y[-1], 1e-12)
    hm = qd.HamiltonianModel(static_operator=cfg["h_d"], operators=cfg["ops"], signals=sig_lists[0],
                             rotating_frame=np.diag(cfg["h_d"]).real.copy())
    for i, t in enumerate((0.0, 0.37, 1.0)):
        assert_close(hm.evaluate_rhs(t, g["diag_yv"]), g["diag_rhs"][i], EVAL_TOL)
    assert_close(hm.evaluate(0.37), g["diag_eval_t037"], EVAL_TOL)


# ------------------------------------------------------------------------------------------------
# a13/a14 vectorised Lindblad
# ------------------------------------------------------------------------------------------------
def _lind_signals(qd):
    hsig = [qd.Signal(0.7, 1.1, 0.2), qd.Signal(lambda t: 0.4 * np.sin(2 * t) + 0j, 0.6)]
    dsig = [qd.Signal(0.3, 0.0), qd.Signal(lambda t: 0.2 + 0.1 * np.cos(t) + 0j, 0.0)]
    return hsig, dsig


@pytest.mark.parametrize("ftag", ["fr", "diag", "nofr"])
def test_lindblad_golden(qd, golden, ftag):
    g = golden("lindblad")
    frame = {"fr": g["hframe"], "diag": np.diag(g["hframe"]).real.copy(), "nofr": None}[ftag]
    hsig, dsig = _lind_signals(qd)
    rho = g["rho"]
    for vec in (True, False):
        m = qd.LindbladModel(static_hamiltonian=g["hstatic"], hamiltonian_operators=g["hops"],
                             hamiltonian_signals=hsig, static_dissipators=g["nstat"],
                             dissipator_operators=g["lops"], dissipator_signals=dsig,
                             rotating_frame=frame, vectorized=vec)
        key = f"{ftag}_{'vec' if vec else 'mat'}"
        yin = rho.flatten(order="F") if vec else rho
        for i, t in enumerate(g["times"]):
            assert_close(m.evaluate_rhs(t, yin), g[key + "_rhs"][i], EVAL_TOL)
            if vec:
                assert_close(m.evaluate(t), g[key + "_eval"][i], EVAL_TOL)
        if vec:
            r = qd.solve_lmde(m, [0.0, 0.6], yin, method="scipy_expm", max_dt=0.02, t_eval=[0.2, 0.6])
            assert_close(r.y, g[key + "_expm_y"], SOLVE_TOL)
            r = qd.solve_lmde(m, [0.0, 0.6], yin, method="RK4", max_dt=0.002)
            assert_close(r.y, g[key + "_rk4_y"], SOLVE_TOL)
        else:
            with pytest.raises(NotImplementedError):
                m.evaluate(0.1)
            r = qd.solve_lmde(m, [0.0, 0.6], yin, method="RK4", max_dt=0.002)
            assert_close(r.y[-1], g[f"{ftag}_vec_rk4_y"][-1].reshape(4, 4, order="F"), SOLVE_TOL)


@pytest.mark.parametrize("ftag", ["diag", "nofr"])
def test_lindblad_unvectorized_sweep_matches_vectorized(qd, golden, ftag):
    """Row f2, sweep form: the instances advance in the same batched launches (own Hamiltonian and
    dissipator coefficients, own initial state). Checked against the vectorised solve of the same
    instances (pinned to the reference by test_lindblad_golden) and against instance-by-instance solves."""
    g = golden("lindblad")
    frame = {"diag": np.diag(g["hframe"]).real.copy(), "nofr": None}[ftag]
    rng = np.random.default_rng(11)
    nb = 5
    sweeps, y0s = [], []
    for b in range(nb):
        hs = [qd.Signal(0.5 + 0.1 * b, 1.1, 0.2 * b), qd.Signal(lambda t, b=b: (0.3 + 0.05 * b) * np.sin(2 * t) + 0j, 0.6)]
        ds = [qd.Signal(0.1 * (b + 1), 0.0), qd.Signal(lambda t, b=b: 0.2 + 0.02 * b * np.cos(t) + 0j, 0.0)]
        sweeps.append((hs, ds))
        a = rng.normal(size=(4, 4)) + 1j * rng.normal(size=(4, 4))
        rho = a @ a.conj().T
        y0s.append(rho / np.trace(rho))
    kw = dict(static_hamiltonian=g["hstatic"], hamiltonian_operators=g["hops"], static_dissipators=g["nstat"],
              dissipator_operators=g["lops"], rotating_frame=frame)
    mat = qd.Solver(vectorized=False, **kw)
    vec = qd.Solver(vectorized=True, **kw)
    for y0 in (y0s, y0s[0]):
        rm = mat.solve(t_span=[0.0, 0.6], y0=y0, signals=sweeps, method="RK4", max_dt=0.002, t_eval=[0.0, 0.3, 0.6])
        yv = [y.flatten(order="F") for y in y0] if isinstance(y0, list) else y0.flatten(order="F")
        rv = vec.solve(t_span=[0.0, 0.6], y0=yv, signals=sweeps, method="RK4", max_dt=0.002, t_eval=[0.0, 0.3, 0.6])
        assert len(rm) == nb
        for b in range(nb):
            assert rm[b].y.shape == (3, 4, 4)
            assert_close(rm[b].y, np.stack([v.reshape(4, 4, order="F") for v in rv[b].y]), SOLVE_TOL)
        one = mat.solve(t_span=[0.0, 0.6], y0=(y0[3] if isinstance(y0, list) else y0), signals=sweeps[3], method="RK4",
                        max_dt=0.002, t_eval=[0.0, 0.3, 0.6])
        assert_close(rm[3].y, one.y, 1e-13)


def test_lindblad_patterns_and_cfg4_small(qd, golden):
    from qiskit_dynamics_amd import workloads

    g = golden("lindblad")
    hsig, dsig = _lind_signals(qd)
    m = qd.LindbladModel(hamiltonian_operators=g["hops"], hamiltonian_signals=hsig,
                         static_dissipators=g["nstat"], vectorized=True)
    assert_close(m.evaluate(0.4), g["pat_hs_eval"], EVAL_TOL)
    m = qd.LindbladModel(static_hamiltonian=g["hstatic"], dissipator_operators=g["lops"],
                         dissipator_signals=dsig, vectorized=True, rotating_frame=g["hframe"])
    assert_close(m.evaluate(0.4), g["pat_sd_fr_eval"], EVAL_TOL)
    assert_close(m.evaluate_rhs(0.4, g["rho"].flatten(order="F")), g["pat_sd_fr_rhs"], EVAL_TOL)
    cfg = workloads.lindblad_config(n_qubits=3, n_drives=3, n_diss=2, gamma=1e-2, t_final=1.0, max_dt=0.05)
    amps, phases = workloads.sweep_parameters(0, 3)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    for ftag, frame in (("nofr", None), ("diag", np.diag(cfg["h_d"]).real.copy())):
        s = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                      static_dissipators=cfg["static_dissipators"], rotating_frame=frame, vectorized=True)
        r = s.solve(t_span=cfg["t_span"], y0=cfg["rho0"].flatten(order="F"), signals=sigs,
                    method="scipy_expm", max_dt=cfg["max_dt"])
        assert_close(r.y, g[f"cfg4s_{ftag}_y"], SOLVE_TOL)
        rho_t = r.y[-1].reshape(8, 8, order="F")
        assert abs(np.trace(rho_t) - 1.0) < 1e-10  # trace preservation


# ------------------------------------------------------------------------------------------------
# a15 Solver list mode
# ------------------------------------------------------------------------------------------------
def test_solver_list_golden(qd, golden):
    g = golden("solver_list")
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    s = qd.Solver(hamiltonian_operators=[x], static_hamiltonian=5 * z, rotating_frame=5 * z)
    y0 = np.array([0.0, 1.0], dtype=complex)
    res = s.solve(t_span=[0.0, 0.4232], y0=y0, signals=[[qd.Signal(1.0, 5.0)], [qd.Signal(0.5, 5.0)]],
                  method="RK4", max_dt=0.001)
    assert_close(np.array([r.y for r in res]), g["ham_list_y"], SOLVE_TOL)
    res = s.solve(t_span=[[0.0, 0.4232], [0.0, 1.23]], y0=y0, signals=[qd.Signal(1.0, 5.0)],
                  method="RK4", max_dt=0.001)
    assert_close(np.array([r.y[-1] for r in res]), g["ham_tspan_list_y_final"], SOLVE_TOL)
    res = s.solve(t_span=[0.0, 0.4232], y0=[y0, np.array([1.0, 0.0], dtype=complex)],
                  signals=[qd.Signal(1.0, 5.0)], method="scipy_expm", max_dt=0.01)
    assert_close(np.array([r.y for r in res]), g["ham_y0_list_expm_y"], SOLVE_TOL)
    sl = qd.Solver(hamiltonian_operators=[x], static_hamiltonian=5 * z, rotating_frame=5 * z,
                   static_dissipators=[0.01 * x], vectorized=True)
    rho0 = np.array([[0.0, 0.0], [0.0, 1.0]], dtype=complex)
    res = sl.solve(t_span=[0.0, 0.4232], y0=rho0.flatten(order="F"),
                   signals=[[qd.Signal(1.0, 5.0)], [qd.Signal(0.5, 5.0)]], method="scipy_expm", max_dt=0.01)
    assert_close(np.array([r.y for r in res]), g["lind_list_y"], SOLVE_TOL)
    single = s.solve(t_span=[0.0, 0.4232], y0=y0, signals=[qd.Signal(1.0, 5.0)], method="RK4", max_dt=0.001)
    assert not isinstance(single, list)
    with pytest.raises(qd.DynamicsError):
        s.solve(t_span=[[0, 1], [0, 1], [0, 1]], y0=[y0, y0], signals=[qd.Signal(1.0, 5.0)],
                method="RK4", max_dt=0.1)


# ------------------------------------------------------------------------------------------------
# a11 expm against scipy (the dependency the reference calls) + invariants
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,scale", [(3, 1e-3), (8, 0.1), (16, 0.7), (40, 1.9), (64, 6.0), (100, 60.0),
                                     (256, 3.0)])
def test_expm_vs_scipy(qd, n, scale):
    rng = np.random.default_rng(n)
    a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    a = (a - a.conj().T)
    a *= scale / np.linalg.norm(a, 1)
    e, info = qd.default_context().expm(a, return_info=True)
    ref = scipy.linalg.expm(a)
    assert np.linalg.norm(e - ref, 1) / np.linalg.norm(ref, 1) < 1e-12
    assert np.linalg.norm(e.conj().T @ e - np.eye(n)) < 1e-12 * n
    einv = qd.default_context().expm(-a)
    assert np.linalg.norm(e @ einv - np.eye(n)) < 1e-12 * n


def test_expm_general_and_batch(qd):
    rng = np.random.default_rng(5)
    a = rng.normal(size=(3, 20, 20)) * 0.3 + 1j * rng.normal(size=(3, 20, 20)) * 0.1
    e = qd.default_context().expm(a)
    for i in range(3):
        ref = scipy.linalg.expm(a[i])
        assert np.linalg.norm(e[i] - ref, 1) / np.linalg.norm(ref, 1) < 1e-12
    z = np.zeros((4, 4), dtype=complex)
    assert_close(qd.default_context().expm(z), np.eye(4), 0)


# ------------------------------------------------------------------------------------------------
# BASELINE.json sizes: cfg 2 (n=1024, k=8) against the oracle, cfg 3 through invariants
# ------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg2(qd):
    from qiskit_dynamics_amd import workloads
    from oracle import dynamics_oracle as orc

    cfg = workloads.schrodinger_config()  # 10 qubits, n=1024, k=8
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])
    stack = qd.Stack(qd.default_context(), a, a_d, np.ascontiguousarray(d.imag))
    return cfg, (a_d, a, d, basis), stack


def test_cfg2_full_size_rhs_and_generator(qd, cfg2):
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    cfg, (a_d, a, d, basis), stack = cfg2
    rng = np.random.default_rng(1024)
    y = crand(rng, 1024)
    amps, phases = workloads.sweep_parameters(0, 8)
    for t in (0.0, 1.234, 5.0):
        c = workloads.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], 5.0)[0]
        assert_close(stack.eval_rhs(c, t, y), orc.generator_rhs(a_d, a, c, d, None, t, y), EVAL_TOL)
    c = workloads.gaussian_coefficient_table(np.array([2.5]), amps, phases, cfg["carrier"], 5.0)[0]
    assert_close(stack.eval_generator(c, 2.5), orc.generator_evaluate(a_d, a, c, d, None, 2.5), EVAL_TOL)
    ym = crand(rng, 1024, 96)
    assert_close(stack.eval_rhs(c, 2.5, ym), orc.generator_rhs(a_d, a, c, d, None, 2.5, ym), EVAL_TOL)


def _table_for(cfg, instances, times, k=8):
    from qiskit_dynamics_amd import workloads

    amps = np.array([workloads.sweep_parameters(b, k)[0] for b in instances])
    phs = np.array([workloads.sweep_parameters(b, k)[1] for b in instances])
    return workloads.gaussian_coefficient_table(times, amps, phs, cfg["carrier"], cfg["t_final"]), amps, phs


def test_cfg2_single_trajectory_rk4_vs_oracle(qd, cfg2):
    """cfg 2: one trajectory, streaming kernel, 40 RK4 steps against the oracle."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg, (a_d, a, d, basis), stack = cfg2
    t_span = [2.4, 2.6]
    sched = FixedStepSchedule(t_span, None, cfg["max_dt"], _rk4_points)
    table, amps, phs = _table_for(cfg, [0], sched.times)
    y0 = np.zeros((1024, 1), dtype=complex)
    y0[3, 0] = 1.0
    ys = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                         sched.n_save, y0, 1, True)

    def rhs(t, y):
        c = workloads.gaussian_coefficient_table(np.array([t]), amps[0], phs[0], cfg["carrier"], 5.0)[0]
        return orc.generator_rhs(a_d, a, c, d, None, t, y)

    _, yref = orc.rk4_solve(rhs, t_span, y0[:, 0], cfg["max_dt"])
    assert_close(ys[0, -1, :, 0], yref[-1], SOLVE_TOL)


def test_cfg3_batched_sweep_vs_oracle_and_invariants(qd, cfg2):
    """cfg 3 shape: 256 instances (2 x 128-column MFMA tiles on the full 9216-deep K loop),
    20 RK4 steps; three instances checked against the oracle, all of them for norm conservation,
    and the batched result must equal the single-trajectory (streaming-kernel) result."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg, (a_d, a, d, basis), stack = cfg2
    B = 256
    t_span = [2.45, 2.55]
    sched = FixedStepSchedule(t_span, None, cfg["max_dt"], _rk4_points)
    table, amps, phs = _table_for(cfg, range(B), sched.times)
    y0 = np.zeros((1024, 1), dtype=complex)
    y0[0, 0] = 1.0
    ys = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                         sched.n_save, y0, B, True)
    final = ys[:, -1, :, 0]
    norms = np.linalg.norm(final, axis=1)
    assert np.max(np.abs(norms - 1.0)) < 1e-9  # RK4 on anti-Hermitian generator, tiny steps
    for b in (0, 101, 255):
        def rhs(t, y, b=b):
            c = workloads.gaussian_coefficient_table(np.array([t]), amps[b], phs[b], cfg["carrier"], 5.0)[0]
            return orc.generator_rhs(a_d, a, c, d, None, t, y)

        _, yref = orc.rk4_solve(rhs, t_span, y0[:, 0], cfg["max_dt"])
        assert_close(final[b], yref[-1], SOLVE_TOL)
    one = stack.rk4_solve(sched.times, table[7:8], sched.step_rows, sched.step_h, sched.step_save,
                          sched.n_save, y0, 1, True)
    assert_close(one[0, -1, :, 0], final[7], 1e-12)
    # linearity in y0: solve(a*y0) == a*solve(y0)
    ys2 = stack.rk4_solve(sched.times, table[:64], sched.step_rows, sched.step_h, sched.step_save,
                          sched.n_save, (0.3 - 0.4j) * y0, 64, True)
    assert_close(ys2[:, -1, :, 0], (0.3 - 0.4j) * final[:64], 1e-12)


def test_rk4_plan_matches_solve(qd, cfg2):
    """The bench hook (device-resident plan, chunked runs) gives the same states as rk4_solve."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg, _, stack = cfg2
    B = 128
    sched = FixedStepSchedule([0.0, 0.05], None, cfg["max_dt"], _rk4_points)
    table, _, _ = _table_for(cfg, range(B), sched.times)
    y0 = np.zeros((1024, 1), dtype=complex)
    y0[0, 0] = 1.0
    ref = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                          sched.n_save, y0, B, True)[:, -1]
    plan = qd.Rk4Plan(stack, sched.times, table, sched.step_rows, sched.step_h, y0, B, True)
    plan.run(0, 4)
    plan.run(4, plan.nsteps)
    stack.ctx.synchronize()
    assert_close(plan.fetch(), ref, 1e-13)
    plan.close()
    # every MFMA tile configuration gives the same states (different summation orders only)
    for tile in (64, 128):
        stack.ctx.set_option("force_tile", tile)
        try:
            got = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                  sched.n_save, y0, B, True)[:, -1]
        finally:
            stack.ctx.set_option("force_tile", 0)
        assert_close(got, ref, 1e-12)


ADOPT_SCRIPT = r"""
import sys
import numpy as np
import torch                      # imported FIRST: libmidyn then binds to torch's HIP runtime
sys.path.insert(0, sys.argv[1])
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import _lib

def crand(rng, *shape):
    return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)

rng = np.random.default_rng(3)
n, k = 20, 3
ops = crand(rng, k, n, n)
ops[1] = 1j * rng.normal(size=(n, n))      # single-plane operator: flags must travel too
static = crand(rng, n, n)
frame_im = rng.normal(size=n)
ctx = qd.default_context()
assert "torch" in _lib.HIP_RUNTIME, _lib.HIP_RUNTIME
dev = torch.device("cuda", ctx.device)
nbytes = qd.Stack.packed_bytes(n, k, True)
buf0 = torch.empty(nbytes, dtype=torch.uint8, device=dev)
s0 = qd.Stack(ctx, ops, static, frame_im, dev_buffer_ptr=buf0.data_ptr())
ctx.synchronize()
torch.cuda.synchronize(dev)
buf1 = buf0.clone()
torch.cuda.synchronize(dev)
s1 = qd.Stack(ctx, None, None, None, dev_buffer_ptr=buf1.data_ptr(), _adopt=(n, k, 1, 1))
assert s1.segment_modes == s0.segment_modes == [0, 0, 2, 0], (s0.segment_modes, s1.segment_modes)
c = rng.uniform(-1, 1, k)
y = crand(rng, n, 3)
assert np.array_equal(s1.eval_rhs(c, 0.4, y), s0.eval_rhs(c, 0.4, y))
assert np.array_equal(s1.eval_generator(c, 0.4), s0.eval_generator(c, 0.4))
e = np.exp(1j * frame_im * 0.4)
ref = (np.tensordot(c, ops, axes=1) + static) * (e.conj()[:, None] * e[None, :])
assert np.max(np.abs(s1.eval_generator(c, 0.4) - ref)) < 1e-12
# a world-size-1 NCCL (= RCCL) broadcast of the packed buffer through torch.distributed
import os, torch.distributed as dist
import socket
with socket.socket() as _s:
    _s.bind(("127.0.0.1", 0)); _port = _s.getsockname()[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(_port))
dist.init_process_group("nccl", rank=0, world_size=1)
from qiskit_dynamics_amd.distributed import broadcast_stack
s2, keep = broadcast_stack(ctx, ops, static, frame_im, n, k, src=0)
assert np.array_equal(s2.eval_rhs(c, 0.4, y), s0.eval_rhs(c, 0.4, y))
# the sharded form of Solver.solve list mode on the same (one-rank) NCCL group: all-gather of CUDA tensors
from qiskit_dynamics_amd.distributed import solve_sweep
x = np.array([[0, 1], [1, 0]], dtype=complex); z = np.diag([1.0, -1.0]).astype(complex)
solver = qd.Solver(static_hamiltonian=5 * z, hamiltonian_operators=[x], rotating_frame=5 * z)
sweep = [[qd.Signal(a, 5.0)] for a in (1.0, 0.5, 0.25)]
y0 = np.array([0.0, 1.0], dtype=complex)
got = solve_sweep(solver, [0.0, 0.4], y0, sweep, method="RK4", max_dt=0.001)
want = solver.solve(t_span=[0.0, 0.4], y0=y0, signals=sweep, method="RK4", max_dt=0.001)
assert len(got) == 3 and all(np.array_equal(g_.y, w_.y) and np.array_equal(g_.t, w_.t) for g_, w_ in zip(got, want))
dist.destroy_process_group()
print("ADOPT_OK")
"""


def test_stack_adopt_roundtrip(tmp_path):
    """Multi-GPU plumbing on one GPU, in a fresh process that imports torch first (as under
    torchrun): build the packed stack inside a torch buffer (rank 0), copy the bytes (what the RCCL
    broadcast does), adopt the copy (other ranks), and run a world-size-1 NCCL broadcast."""
    import subprocess
    import sys

    from conftest import ROOT

    script = tmp_path / "adopt.py"
    script.write_text(ADOPT_SCRIPT)
    p = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ADOPT_OK" in p.stdout, p.stdout[-2000:] + p.stderr[-4000:]


def test_lindblad_unvectorized_large(qd):
    """Row f2 at a size where the superoperator cannot exist (n=256 -> N=65536): device n x n zgemm
    RHS against the oracle formula, batched rho input, and RK4 invariants (trace, Hermiticity)."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(256)
    n = 256

    def herm(k=1.0):
        a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        return k * (a + a.conj().T) / np.sqrt(n)

    h_static, h_ops = herm(), np.array([herm(0.5), herm(0.3)])
    n_stat = (rng.normal(size=(2, n, n)) + 1j * rng.normal(size=(2, n, n))) * 0.05 / np.sqrt(n)
    l_ops = (rng.normal(size=(1, n, n)) + 1j * rng.normal(size=(1, n, n))) * 0.05 / np.sqrt(n)
    frame = np.diag(h_static).real.copy()
    hsig = [qd.Signal(0.7, 0.3, 0.2), qd.Signal(lambda t: 0.4 * np.sin(2 * t) + 0j, 0.1)]
    dsig = [qd.Signal(lambda t: 0.5 + 0.1 * np.cos(t) + 0j, 0.0)]
    m = qd.LindbladModel(static_hamiltonian=h_static, hamiltonian_operators=h_ops, hamiltonian_signals=hsig,
                         static_dissipators=n_stat, dissipator_operators=l_ops, dissipator_signals=dsig,
                         rotating_frame=frame, vectorized=False)
    a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    rho = a @ a.conj().T
    rho /= np.trace(rho)
    h_d, ho, ns, lo, d, basis = orc.lindblad_model_build(h_static, h_ops, n_stat, l_ops, frame)
    t = 0.37
    hc = np.array([s(t) for s in hsig])
    dc = np.array([s(t) for s in dsig])
    ref = orc.lindblad_rhs(h_d, ho, ns, lo, hc, dc, d, t, rho)
    assert_close(m.evaluate_rhs(t, rho), ref, EVAL_TOL)
    both = m.evaluate_rhs(t, np.stack([rho, 2 * rho]))
    assert_close(both[1], 2 * ref, EVAL_TOL)
    r = qd.solve_lmde(m, [0.0, 0.05], rho, method="RK4", max_dt=0.01)
    rf = r.y[-1]
    assert abs(np.trace(rf) - 1.0) < 1e-10
    assert np.linalg.norm(rf - rf.conj().T) < 1e-10

    def rhs(tt, y):
        return orc.lindblad_rhs(h_d, ho, ns, lo, np.array([s(tt) for s in hsig]),
                                np.array([s(tt) for s in dsig]), d, tt, y)

    _, yref = orc.rk4_solve(rhs, [0.0, 0.05], rho, 0.01)
    assert_close(rf, yref[-1], SOLVE_TOL)


def test_single_instance_many_columns_fast_path(qd, cfg2):
    """B = 1 with m >= 8 columns forms C(t) once and runs ONE zgemm per stage; it must agree with the
    per-segment contraction."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points
    from qiskit_dynamics_amd import workloads

    cfg, _, stack = cfg2
    sched = FixedStepSchedule([1.0, 1.02], None, cfg["max_dt"], _rk4_points)
    amps, phs = workloads.sweep_parameters(3, 8)
    table = workloads.gaussian_coefficient_table(sched.times, amps[None], phs[None], cfg["carrier"], 5.0)
    rng = np.random.default_rng(1)
    y0 = crand(rng, 1024, 40)
    outs = []
    for flag in (1, 0):
        stack.ctx.set_option("combine_first", flag)
        try:
            outs.append(stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                        sched.n_save, y0, 1, True)[0, -1])
        finally:
            stack.ctx.set_option("combine_first", 1)
    assert_close(outs[0], outs[1], 1e-12)


def test_expm_lindbladian_vs_scipy(qd):
    """Non-normal generator with a large norm (4-qubit vectorised Lindbladian, no frame, h = 0.05):
    the Taylor/squaring device expm against scipy's Pade, plus trace preservation of the map."""
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.models import vec_commutator, vec_dissipator

    cfg = workloads.lindblad_config(n_qubits=4, n_drives=4, n_diss=3, gamma=5e-2)
    n = 16
    lind = vec_commutator(cfg["h_d"] + 0.7 * cfg["ops"][0] - 0.4 * cfg["ops"][2]) \
        + np.sum(vec_dissipator(cfg["static_dissipators"]), axis=0)
    a = 0.05 * lind
    e, info = qd.default_context().expm(a, return_info=True)
    ref = scipy.linalg.expm(a)
    assert info[0, 0] >= 3  # several squarings were needed
    assert np.linalg.norm(e - ref, 1) / np.linalg.norm(ref, 1) < 1e-12
    vec_id = np.eye(n).flatten(order="F")
    assert np.max(np.abs(vec_id @ e - vec_id)) < 1e-12  # tr(rho) preserved: vec(I)^T E = vec(I)^T


def test_planar_single_plane_kernel(qd):
    """Stacks whose operators are all purely real / purely imaginary run the planar two-tiles-per-
    barrier kernel (M, N multiples of 128): mixed kinds, odd segment count, split-K and many
    instances with different coefficients, against NumPy and against the generic kernel."""
    rng = np.random.default_rng(128)
    n, k = 128, 4
    ops = np.zeros((k, n, n), dtype=complex)
    ops[0] = rng.normal(size=(n, n))
    ops[1] = 1j * rng.normal(size=(n, n))
    ops[2] = 1j * rng.normal(size=(n, n))
    ops[3] = rng.normal(size=(n, n))
    static = 1j * rng.normal(size=(n, n))       # 5 active single-plane segments (odd)
    frame_im = rng.normal(size=n)
    ctx = qd.default_context()
    stack = qd.Stack(ctx, ops, static, frame_im)
    assert stack.segment_modes == [2, 1, 2, 2, 1]
    c = rng.uniform(-1, 1, k)
    y = crand(rng, n, 128)
    e = np.exp(1j * frame_im * 0.7)
    ref = np.conj(e)[:, None] * ((np.tensordot(c, ops, axes=1) + static) @ (e[:, None] * y))
    ctx.set_option("plane_kernel", 1)          # opt-in variant (default off: measured slower)
    try:
        got = stack.eval_rhs(c, 0.7, y)
    finally:
        ctx.set_option("plane_kernel", 0)
    assert_close(got, ref, EVAL_TOL)
    assert_close(stack.eval_rhs(c, 0.7, y), got, 1e-13)
    # sweep with per-instance coefficients: 256 instances, 3 RK4 steps, both kernels
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    sched = FixedStepSchedule([0.0, 0.03], None, 0.01, _rk4_points)
    B = 256
    table = rng.uniform(-1, 1, (B, len(sched.times), k))
    y0 = crand(rng, n, 1)
    outs = []
    for flag in (1, 0):
        ctx.set_option("plane_kernel", flag)
        try:
            outs.append(stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                        sched.n_save, y0, B, True)[:, -1, :, 0])
        finally:
            ctx.set_option("plane_kernel", 0)
    assert_close(outs[0], outs[1], 1e-12)
    single = stack.rk4_solve(sched.times, table[5:6], sched.step_rows, sched.step_h, sched.step_save,
                             sched.n_save, y0, 1, True)[0, -1, :, 0]
    assert_close(outs[0][5], single, 1e-12)


@pytest.mark.parametrize("seed", range(12))
def test_randomised_models_vs_oracle(qd, seed):
    """Random small models (ragged sizes, with/without static operator, frames of all kinds, vector and
    matrix states, sweeps with per-instance signals and initial states, forwards/backwards, t_eval)
    through the public Solver / solve_lmde API against the oracle."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(2, 40))
    k = int(rng.integers(0, 5))
    has_static = bool(rng.integers(0, 2)) or k == 0
    frame_kind = ["none", "diag", "full"][int(rng.integers(0, 3))]
    m = [None, 1, 3][int(rng.integers(0, 3))]          # None: vector state
    batch = [1, 2, 5][int(rng.integers(0, 3))]
    method = ["RK4", "scipy_expm"][int(rng.integers(0, 2))]
    backwards = bool(rng.integers(0, 2))

    def herm():
        a = crand(rng, n, n)
        return (a + a.conj().T) / 2

    h_static = herm() if has_static else None
    h_ops = np.array([herm() for _ in range(k)]) if k else None
    frame = {"none": None, "diag": rng.normal(size=n), "full": herm()}[frame_kind]
    t_span = [0.4, 0.0] if backwards else [0.0, 0.4]
    t_eval = None if rng.integers(0, 2) else (sorted(rng.uniform(0, 0.4, 3), reverse=backwards))
    max_dt = 0.01 if method == "RK4" else 0.05
    mo = int(rng.integers(1, 4))

    def make_sigs():
        amps, nus, phs = rng.uniform(-1, 1, k), rng.uniform(0, 2, k), rng.uniform(-3, 3, k)
        return ([qd.Signal(lambda t, a=a: a * np.cos(0.7 * t) + 0j, nu, ph) for a, nu, ph in zip(amps, nus, phs)],
                (amps, nus, phs))

    def make_y0():
        y = crand(rng, n) if m is None else crand(rng, n, m)
        return y / np.linalg.norm(y)

    sig_sets = [make_sigs() for _ in range(batch)]
    y0s = [make_y0() for _ in range(batch)]
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    kwargs = dict(method=method, max_dt=max_dt, t_eval=t_eval)
    if method == "scipy_expm":
        kwargs["magnus_order"] = mo
    if k == 0:
        res = solver.solve(t_span=t_span, y0=y0s if batch > 1 else y0s[0], **kwargs)
    else:
        res = solver.solve(t_span=t_span, y0=y0s if batch > 1 else y0s[0],
                           signals=[s for s, _ in sig_sets] if batch > 1 else sig_sets[0][0], **kwargs)
    res = res if isinstance(res, list) else [res]
    assert len(res) == batch
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
    for b in range(batch):
        amps, nus, phs = sig_sets[b][1]

        def coeff(t, amps=amps, nus=nus, phs=phs):
            return np.array([orc.signal_sum_value(np.array([a_ * np.cos(0.7 * t) + 0j]), [nu], [ph], t)
                             for a_, nu, ph in zip(amps, nus, phs)])

        t_ref, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff, t_span, y0s[b],
                                                 "RK4" if method == "RK4" else "scipy_expm", max_dt,
                                                 t_eval=t_eval, magnus_order=mo)
        assert_close(res[b].t, t_ref, 0)
        assert_close(res[b].y, y_ref, SOLVE_TOL)


def test_batched_expm_solve_equals_individual(qd):
    """A sweep solved with the Magnus/expm method advances all instances in batched launches; the
    batch must reproduce every individual solve (shared squaring count included) and the oracle."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(77)
    n, k, B = 12, 2, 40

    def herm():
        a = crand(rng, n, n)
        return (a + a.conj().T) / 2

    h_static, h_ops = herm(), np.array([herm(), herm()])
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=np.diag(h_static).real)
    amps = rng.uniform(0.1, 3.0, (B, k))          # very different norms -> different natural squarings
    sig_lists = [[qd.Signal(lambda t, a=a: a * np.cos(t) + 0j, 0.3 * j) for j, a in enumerate(amps[b])]
                 for b in range(B)]
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    res = solver.solve(t_span=[0.0, 0.3], y0=y0, signals=sig_lists, method="scipy_expm", max_dt=0.1,
                       magnus_order=2, t_eval=[0.1, 0.3])
    for b in (0, 17, 39):
        one = solver.solve(t_span=[0.0, 0.3], y0=y0, signals=sig_lists[b], method="scipy_expm", max_dt=0.1,
                           magnus_order=2, t_eval=[0.1, 0.3])
        assert_close(res[b].y, one.y, 1e-12)
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, np.diag(h_static).real)
    b = 23

    def coeff(t):
        return np.array([orc.signal_sum_value(np.array([a_ * np.cos(t) + 0j]), [0.3 * j], [0.0], t)
                         for j, a_ in enumerate(amps[b])])

    _, yref = orc.solve_generator_model(a_d, a, d, basis, coeff, [0.0, 0.3], y0, "scipy_expm", 0.1,
                                        t_eval=[0.1, 0.3], magnus_order=2)
    assert_close(res[b].y, yref, SOLVE_TOL)
    # batched expm entry point
    mats = np.array([1j * herm() * s for s in (0.01, 0.5, 3.0, 20.0)])
    e = qd.default_context().expm(mats)
    for i in range(4):
        ref = scipy.linalg.expm(mats[i])
        assert np.linalg.norm(e[i] - ref, 1) / np.linalg.norm(ref, 1) < 1e-12


def test_in_kernel_rk_epilogue_without_split(qd, cfg2):
    """The headline path (>= 256 tiles) runs the RK4 epilogue inside the contraction kernel; smaller
    test batches take the split-K + reduce route, so force the in-kernel route here and compare."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg, _, stack = cfg2
    B = 256
    sched = FixedStepSchedule([1.0, 1.02], None, cfg["max_dt"], _rk4_points)
    table, _, _ = _table_for(cfg, range(B), sched.times)
    y0 = np.zeros((1024, 1), dtype=complex)
    y0[5, 0] = 1.0
    outs = {}
    for name, opts in (("split", {}), ("nosplit128", {"split_k": 0}), ("nosplit64", {"split_k": 0, "force_tile": 64}),
                       ("dense3m", {"split_k": 0, "skip_zero_planes": 0}),
                       ("dense4m", {"split_k": 0, "skip_zero_planes": 0, "complex_3m": 0})):
        for k_, v_ in opts.items():
            stack.ctx.set_option(k_, v_)
        try:
            outs[name] = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                         sched.n_save, y0, B, True)[:, -1]
        finally:
            for k_, v_ in (("split_k", 1), ("force_tile", 0), ("skip_zero_planes", 1), ("complex_3m", 1)):
                stack.ctx.set_option(k_, v_)
    for name in outs:
        assert_close(outs[name], outs["split"], 1e-12)


# ---- row f1: coefficient table evaluated on the device --------------------------------------------
def _device_table(qd, instances, times):
    from qiskit_dynamics_amd import _lib
    from qiskit_dynamics_amd.signals import discrete_term_arrays

    arrays = discrete_term_arrays(instances)
    assert arrays is not None
    tab = _lib.SignalTable(_lib.default_context(), len(instances), len(instances[0]), times, *arrays)
    return tab


def test_device_signal_table_matches_reference_values(qd, golden):
    """DiscreteSignal values captured from the reference (incl. times exactly on sample edges)."""
    g = golden("signals")
    dt, t0, nu, phi = g["disc_params"]
    d = qd.DiscreteSignal(dt=dt, samples=g["disc_samples"], start_time=t0, carrier_freq=nu, phase=phi)
    for times, want in ((g["t"], g["disc"]), (g["disc_edges_t"], g["disc_edges"])):
        got = _device_table(qd, [[d]], times).fetch()[0, :, 0]
        assert_close(got, want, 1e-15 * 8)
        # the piecewise-constant sample picked must be the same one: zero exactly where the reference is
        np.testing.assert_array_equal(got == 0.0, want == 0.0)
    ds = qd.DiscreteSignal(dt=0.1, samples=g["from_signal_samples"], start_time=0.0,
                           carrier_freq=g["gauss_params"][3], phase=g["gauss_params"][4])
    got = _device_table(qd, [[ds]], g["t"]).fetch()[0, :, 0]
    assert_close(got, g["from_signal"], 1e-15 * 8)


def test_device_signal_table_sample_index_is_exact(qd):
    """carrier 0, integer samples: the table must equal the host table BIT FOR BIT, in particular at
    times that are float multiples of dt (NumPy's fmod-based floor_divide decides the bin)."""
    rng = np.random.default_rng(5)
    for dt, t0 in ((0.1, 0.0), (0.1, -0.3), (1.0 / 3, 0.7), (0.222, 1e-3), (2.0**-4, -1.0)):
        ns = 57
        smp = np.arange(1, ns + 1) + 1j * np.arange(ns, 0, -1)
        d = qd.DiscreteSignal(dt=dt, samples=smp, start_time=t0)
        k_ = np.arange(-5, ns + 6)
        times = np.concatenate([t0 + k_ * dt, t0 + dt * k_.astype(float) * (1 + 2.0**-52), (t0 + k_ * dt) - 1e-17,
                                np.nextafter(t0 + k_ * dt, -np.inf), np.nextafter(t0 + k_ * dt, np.inf),
                                rng.uniform(t0 - 1, t0 + ns * dt + 1, 300)])
        got = _device_table(qd, [[d]], times).fetch()[0, :, 0]
        want = qd.SignalList([d]).table(times)[:, 0]
        np.testing.assert_array_equal(got, want)


def test_device_signal_table_random_sums(qd):
    """Random SignalSums of DiscreteSignals + constants, many instances: device vs host table."""
    rng = np.random.default_rng(11)
    B, k = 37, 5
    times = np.sort(rng.uniform(-1.0, 12.0, 211))
    inst = []
    for _ in range(B):
        sigs = []
        for j in range(k):
            terms = []
            for _t in range(rng.integers(1, 4)):
                ns = int(rng.integers(1, 40))
                terms.append(qd.DiscreteSignal(dt=rng.uniform(0.05, 0.7), samples=rng.normal(size=ns) + 1j * rng.normal(size=ns),
                                               start_time=rng.uniform(-0.5, 2.0), carrier_freq=rng.uniform(-6, 6),
                                               phase=rng.uniform(-3, 3)))
            if j == 1:
                terms.append(qd.Signal(rng.normal(), 0.0, rng.uniform(-1, 1)))
            sig = terms[0]
            for t_ in terms[1:]:
                sig = sig + t_
            sigs.append(sig if j != 3 else 0.75)
        inst.append(sigs)
    got = _device_table(qd, inst, times).fetch()
    want = np.stack([qd.SignalList(s).table(times) for s in inst])
    assert got.shape == want.shape == (B, len(times), k)
    assert np.max(np.abs(got - want)) <= 2e-14, np.max(np.abs(got - want))


def test_solver_sweep_with_device_table(qd, monkeypatch):
    """Solver.solve list mode over DiscreteSignal sweeps: device-evaluated table vs host table, for the
    RK4, expm and non-vectorised Lindblad routes."""
    from qiskit_dynamics_amd import solvers as S
    from qiskit_dynamics_amd import workloads as W

    rng = np.random.default_rng(3)
    c1 = W.config1()
    B = 6
    sig_lists = []
    for _ in range(B):
        sig_lists.append([qd.DiscreteSignal(dt=0.05, samples=rng.uniform(0.2, 1.0, 24) * np.exp(1j * rng.uniform(0, 1, 24)),
                                            carrier_freq=c1["carrier"][j] if "carrier" in c1 else 5.0, phase=rng.uniform(0, 3))
                          for j in range(len(c1["ops"]))])
    created = []
    orig = S.SignalTable

    def spy(*a, **kw):
        tab = orig(*a, **kw)
        created.append(tab)
        return tab

    solver = qd.Solver(static_hamiltonian=c1["h_d"], hamiltonian_operators=c1["ops"], rotating_frame=c1["h_d"])
    y0 = np.eye(c1["h_d"].shape[0], dtype=complex)[:, 0]
    for method, kw in (("RK4", {}), ("scipy_expm", {"magnus_order": 2})):
        monkeypatch.setattr(S, "DEVICE_SIGNAL_TABLE_MIN", 1 << 62)
        host = solver.solve(t_span=[0.0, 1.0], y0=y0, signals=sig_lists, method=method, max_dt=0.01, **kw)
        monkeypatch.setattr(S, "DEVICE_SIGNAL_TABLE_MIN", 0)
        monkeypatch.setattr(S, "SignalTable", spy)
        n0 = len(created)
        dev = solver.solve(t_span=[0.0, 1.0], y0=y0, signals=sig_lists, method=method, max_dt=0.01, **kw)
        monkeypatch.setattr(S, "SignalTable", orig)
        assert len(created) == n0 + 1, "the device table route was not taken"
        for a, b in zip(host, dev):
            assert_close(b.y, a.y, 1e-12)
    # non-vectorised Lindblad (host-side coefficient reads from a device table)
    lops = [np.array([[0, 1], [0, 0]], dtype=complex)]
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    ls = qd.Solver(static_hamiltonian=2 * np.pi * 2.5 * z, hamiltonian_operators=[2 * np.pi * 0.1 * x],
                   static_dissipators=[0.1 * lops[0]], rotating_frame=2 * np.pi * 2.5 * z, vectorized=False)
    sigs = [[qd.DiscreteSignal(dt=0.1, samples=rng.uniform(0.2, 1, 10), carrier_freq=5.0, phase=rng.uniform(0, 1))]
            for _ in range(4)]
    rho0 = np.diag([1.0, 0.0]).astype(complex)
    monkeypatch.setattr(S, "DEVICE_SIGNAL_TABLE_MIN", 1 << 62)
    host = ls.solve(t_span=[0.0, 1.0], y0=rho0, signals=sigs, method="RK4", max_dt=0.01)
    monkeypatch.setattr(S, "DEVICE_SIGNAL_TABLE_MIN", 0)
    dev = ls.solve(t_span=[0.0, 1.0], y0=rho0, signals=sigs, method="RK4", max_dt=0.01)
    for a, b in zip(host, dev):
        assert_close(b.y, a.y, 1e-12)


# ---- row f3: parallel-in-time propagation ---------------------------------------------------------
@pytest.mark.parametrize("seed", range(8))
def test_parallel_in_time_vs_sequential_and_oracle(qd, seed):
    """``hip_RK4_parallel`` / ``hip_expm_parallel`` (step propagators formed together, tree products,
    fixed_step_solvers.py:524-613) against the sequential device methods and the oracle: vector, matrix
    and square (propagator) states, frames, t_eval (several intervals), backwards, sweeps."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(2, 70))
    k = int(rng.integers(1, 4))
    frame_kind = ["none", "diag", "full"][seed % 3]
    m = [None, 2, "square"][int(rng.integers(0, 3))]
    batch = [1, 3][int(rng.integers(0, 2))]
    backwards = bool(rng.integers(0, 2))
    mo = 1 + seed % 3
    par_method, seq_method = (("hip_RK4_parallel", "RK4") if seed % 2 == 0 else ("hip_expm_parallel", "scipy_expm"))

    def herm():
        a = crand(rng, n, n)
        return (a + a.conj().T) / 2

    h_static = herm()
    h_ops = np.array([herm() for _ in range(k)])
    frame = {"none": None, "diag": rng.normal(size=n), "full": herm()}[frame_kind]
    t_span = [0.3, 0.0] if backwards else [0.0, 0.3]
    t_eval = [None, sorted(rng.uniform(0, 0.3, 4), reverse=backwards),
              np.linspace(t_span[0], t_span[1], 31)][int(rng.integers(0, 3))]
    max_dt = 0.004 if seq_method == "RK4" else 0.02

    def make_sigs():
        amps, nus, phs = rng.uniform(-1, 1, k), rng.uniform(0, 2, k), rng.uniform(-3, 3, k)
        return ([qd.Signal(lambda t, a=a: a * np.cos(0.7 * t) + 0j, nu, ph) for a, nu, ph in zip(amps, nus, phs)],
                (amps, nus, phs))

    def make_y0():
        if m == "square":
            return np.linalg.qr(crand(rng, n, n))[0]
        y = crand(rng, n) if m is None else crand(rng, n, m)
        return y / np.linalg.norm(y)

    sig_sets = [make_sigs() for _ in range(batch)]
    y0s = [make_y0() for _ in range(batch)]
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    kw = dict(max_dt=max_dt, t_eval=t_eval)
    if seq_method == "scipy_expm":
        kw["magnus_order"] = mo
    args = dict(t_span=t_span, y0=y0s if batch > 1 else y0s[0],
                signals=[s for s, _ in sig_sets] if batch > 1 else sig_sets[0][0])
    par = solver.solve(method=par_method, **args, **kw)
    seq = solver.solve(method=seq_method, **args, **kw)
    par = par if isinstance(par, list) else [par]
    seq = seq if isinstance(seq, list) else [seq]
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
    for b in range(batch):
        assert_close(par[b].t, seq[b].t, 0)
        assert_close(par[b].y, seq[b].y, SOLVE_TOL)
        amps, nus, phs = sig_sets[b][1]

        def coeff(t, amps=amps, nus=nus, phs=phs):
            return np.array([orc.signal_sum_value(np.array([a_ * np.cos(0.7 * t) + 0j]), [nu], [ph], t)
                             for a_, nu, ph in zip(amps, nus, phs)])

        t_ref, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff, t_span, y0s[b], seq_method, max_dt,
                                                 t_eval=t_eval, magnus_order=mo)
        assert_close(par[b].t, t_ref, 0)
        assert_close(par[b].y, y_ref, SOLVE_TOL)


def test_parallel_in_time_many_steps_and_lindblad(qd):
    """More steps than one chunk holds (intervals split across chunks), the jax_* aliases of the
    reference, the 128-tile product path, and a vectorised Lindblad model."""
    from qiskit_dynamics_amd import workloads as W

    c1 = W.config1()
    solver = qd.Solver(static_hamiltonian=c1["h_d"], hamiltonian_operators=c1["ops"], rotating_frame=c1["h_d"])
    sigs = [qd.Signal(lambda t: 0.3 * np.exp(-((t - 2.0) ** 2)) + 0j, 5.0, 0.1 * j) for j in range(len(c1["ops"]))]
    n = c1["h_d"].shape[0]
    y0 = np.eye(n, dtype=complex)
    t_eval = [0.0, 1.3, 2.0, 4.0]
    # 4 / 0.0008 = 5000 steps > 4096 per chunk at this size
    par = solver.solve(t_span=[0.0, 4.0], y0=y0, signals=sigs, method="jax_RK4_parallel", max_dt=0.0008, t_eval=t_eval)
    seq = solver.solve(t_span=[0.0, 4.0], y0=y0, signals=sigs, method="RK4", max_dt=0.0008, t_eval=t_eval)
    assert_close(par.y, seq.y, SOLVE_TOL)
    par = solver.solve(t_span=[0.0, 4.0], y0=y0[:, 0], signals=sigs, method="jax_expm_parallel", max_dt=0.0009,
                       magnus_order=2, t_eval=t_eval)
    seq = solver.solve(t_span=[0.0, 4.0], y0=y0[:, 0], signals=sigs, method="scipy_expm", max_dt=0.0009,
                       magnus_order=2, t_eval=t_eval)
    assert_close(par.y, seq.y, SOLVE_TOL)
    # n = 128: products on the 128x128 tile with per-problem offsets
    cfg = W.schrodinger_config(7)
    s7 = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    sig7 = [qd.Signal(lambda t, j=j: 0.2 * np.cos(0.3 * t + j) + 0j, 4.0 + 0.1 * j, 0.0) for j in range(len(cfg["ops"]))]
    y7 = np.zeros(128, dtype=complex)
    y7[3] = 1.0
    for pm, sm, kw in (("hip_RK4_parallel", "RK4", {}), ("hip_expm_parallel", "scipy_expm", {"magnus_order": 3})):
        par = s7.solve(t_span=[0.0, 0.5], y0=y7, signals=sig7, method=pm, max_dt=0.01, t_eval=[0.0, 0.2, 0.5], **kw)
        seq = s7.solve(t_span=[0.0, 0.5], y0=y7, signals=sig7, method=sm, max_dt=0.01, t_eval=[0.0, 0.2, 0.5], **kw)
        assert_close(par.y, seq.y, SOLVE_TOL)
        assert abs(np.linalg.norm(par.y[-1]) - 1.0) < 1e-8
    # vectorised Lindblad (N = 16)
    lc = W.lindblad_config(2, n_drives=2, n_diss=2, gamma=0.05)
    ls = qd.Solver(static_hamiltonian=lc["h_d"], hamiltonian_operators=lc["ops"], static_dissipators=lc["static_dissipators"],
                   rotating_frame=lc["h_d"], vectorized=True)
    lsig = [qd.Signal(lambda t: 0.5 * np.sin(t) ** 2 + 0j, nu, 0.0) for nu in lc["carrier"]]
    rho0 = lc["rho0"].flatten(order="F")  # plain arrays are not reshaped (solver_classes.py:781-790)
    par = ls.solve(t_span=[0.0, 1.0], y0=rho0, signals=lsig, method="hip_expm_parallel", max_dt=0.01, magnus_order=1)
    seq = ls.solve(t_span=[0.0, 1.0], y0=rho0, signals=lsig, method="scipy_expm", max_dt=0.01, magnus_order=1)
    assert_close(par.y, seq.y, SOLVE_TOL)
    with pytest.raises(qd.DynamicsError):
        qd.Solver(static_hamiltonian=lc["h_d"], hamiltonian_operators=lc["ops"], static_dissipators=lc["static_dissipators"],
                  vectorized=False).solve(t_span=[0.0, 1.0], y0=lc["rho0"], signals=lsig, method="hip_expm_parallel",
                                          max_dt=0.01)


def test_planar_stream_kernel_matches_interleaved(qd, cfg2):
    """Single-plane stacks: the one-column kernel streams only the non-zero planes (half the bytes);
    the skipped products are exact zeros, so it must agree with the interleaved kernel to rounding
    (different summation order across lanes) and with the oracle."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg, ref, stack = cfg2
    assert all(m in (1, 2, 3) for m in stack.segment_modes)
    rng = np.random.default_rng(2)
    y = crand(rng, 1024)
    c = rng.normal(size=8)
    outs = {}
    for planes in (1, 0):
        stack.ctx.set_option("stream_planes", planes)
        for variant in (0, 1, 2, 3, 4):
            stack.ctx.set_option("stream_variant", variant)
            outs[(planes, variant)] = stack.eval_rhs(c, 0.37, y)
    stack.ctx.set_option("stream_planes", 1)
    stack.ctx.set_option("stream_variant", 0)
    for key, val in outs.items():
        assert_close(val, outs[(0, 0)], 1e-13)
    sched = FixedStepSchedule([0.0, 0.1], None, cfg["max_dt"], _rk4_points)
    table, _, _ = _table_for(cfg, range(1), sched.times)
    y0 = np.zeros((1024, 1), dtype=complex)
    y0[7, 0] = 1.0
    res = {}
    for planes in (1, 0):
        stack.ctx.set_option("stream_planes", planes)
        res[planes] = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                      sched.n_save, y0, 1, True)
    stack.ctx.set_option("stream_planes", 1)
    assert_close(res[1], res[0], 1e-13)


# ---- row f4: perturbative Dyson / Magnus solvers ------------------------------------------------------
@pytest.mark.parametrize("kind", ["dyson", "magnus"])
def test_perturbative_solvers_golden(qd, golden, kind):
    """DysonSolver / MagnusSolver through the device (one GEMM for the polynomial of all steps, batched
    expm, tree product) against final states captured from the reference (its sequential host loop)."""
    g = golden("perturbative")
    r, sig_w, t_c, dt, nu = g["q1_params"]
    cls = qd.DysonSolver if kind == "dyson" else qd.MagnusSolver
    sol = cls(operators=g["q1_ops"], rotating_frame=g["q1_frame"], dt=dt, carrier_freqs=[nu], chebyshev_orders=[1],
              expansion_order=6 if kind == "dyson" else 3, integration_method="DOP853", atol=1e-12, rtol=1e-12)
    gauss = qd.Signal(lambda t: 1.0 * np.exp(-((t - t_c) ** 2) / (2 * sig_w**2)), carrier_freq=nu)
    res = sol.solve(t0=0.0, n_steps=120, y0=np.eye(2, dtype=complex), signals=[gauss])
    assert_close(res.t, np.array([0.0, 120 * dt]), 1e-15)
    assert_close(res.y[0], np.eye(2), 0)
    assert_close(res.y[-1], g[f"q1_{kind}_y_eye"], 1e-9)
    res = sol.solve(t0=3.1, n_steps=50, y0=g["q1_y_rand"], signals=[gauss])
    assert_close(res.y[-1], g[f"q1_{kind}_y_rand_t1"], 1e-9)
    res = sol.solve(t0=0.0, n_steps=30, y0=g["q1_y_rand"][:, 0], signals=[gauss])
    assert res.y[-1].shape == (2,)
    assert_close(res.y[-1], g[f"q1_{kind}_y_vec"], 1e-9)
    if kind == "dyson":
        assert_close(sol.model.evaluate(g["q1_dyson_eval_c"]), g["q1_dyson_eval"], 1e-10)
    # transmon, list mode (two signal sets, two initial states of different shapes)
    sol = cls(operators=g["t3_ops"], rotating_frame=g["t3_frame"], dt=0.02, carrier_freqs=[4.9, 0.0],
              chebyshev_orders=[1, 0], expansion_order=2, expansion_labels=[[0, 0, 1], [0, 1, 4]],
              include_imag=[True, False], integration_method="DOP853", atol=1e-12, rtol=1e-12)
    sig_a = qd.Signal(lambda t: 0.8 * np.exp(-((t - 1.0) ** 2) / 0.5) * np.exp(0.3j * t), carrier_freq=4.9, phase=0.2)
    sig_b = qd.Signal(lambda t: 0.4 * np.cos(0.7 * t) + 0j, carrier_freq=0.0)
    sig_c = qd.Signal(lambda t: 0.5 * np.exp(-((t - 0.7) ** 2) / 0.3) + 0j, carrier_freq=4.95, phase=-0.4)
    res = sol.solve(t0=0.1, n_steps=60, y0=[np.eye(3, dtype=complex), g["t3_y0"]],
                    signals=[[sig_a, sig_b], [sig_c, sig_b]])
    assert isinstance(res, list) and len(res) == 2
    assert_close(res[0].t, g[f"t3_{kind}_t"], 1e-15)
    assert_close(res[0].y[-1], g[f"t3_{kind}_y_list0"], 1e-9)
    assert_close(res[1].y[-1], g[f"t3_{kind}_y_list1"], 1e-9)
    with pytest.raises(qd.DynamicsError, match="same length as the operators"):
        sol.solve(t0=0.0, n_steps=5, y0=np.eye(3, dtype=complex), signals=[sig_a])


def test_perturbative_solvers_vs_direct_solution(qd):
    """As test_dyson_magnus_solvers.py:222-246: the perturbative solvers reproduce the direct solution
    of the same model (here: the device RK4 with a small step), many steps (> one chunk of 64 rows)."""
    r = 0.2
    sig_w = 0.399128 / r
    t_c = 3.5 * sig_w
    gauss = qd.Signal(lambda t: np.exp(-((t - t_c) ** 2) / (2 * sig_w**2)), carrier_freq=5.0)
    dt = 0.0125
    n_steps = int((7 * sig_w) // dt) // 3
    h_ops = 2 * np.pi * r * np.array([[[0.0, 1.0], [1.0, 0.0]]]) / 2
    h_static = 2 * np.pi * 5.0 * np.array([[1.0, 0.0], [0.0, -1.0]]) / 2
    direct = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=h_static).solve(
        t_span=[0.0, dt * n_steps], y0=np.eye(2, dtype=complex), signals=[gauss], method="RK4", max_dt=dt / 8).y[-1]
    dys = qd.DysonSolver(operators=-1j * h_ops, rotating_frame=-1j * h_static, dt=dt, carrier_freqs=[5.0],
                         chebyshev_orders=[1], expansion_order=6, integration_method="DOP853", atol=1e-10, rtol=1e-10)
    mag = qd.MagnusSolver(operators=-1j * h_ops, rotating_frame=-1j * h_static, dt=dt, carrier_freqs=[5.0],
                          chebyshev_orders=[1], expansion_order=3, integration_method="DOP853", atol=1e-10,
                          rtol=1e-10)
    for sol in (dys, mag):
        yf = sol.solve(t0=0.0, n_steps=n_steps, y0=np.eye(2, dtype=complex), signals=[gauss]).y[-1]
        assert np.max(np.abs(yf - direct)) < 1e-6, np.max(np.abs(yf - direct))


@pytest.mark.parametrize("kind", ["antiherm", "general"])
def test_expm_adaptive_degree(qd, kind):
    """The Taylor degree is chosen from the 1-norm (schemes 2,4,6,9,12,16 + squarings): every scheme at
    the edge of its range, and the automatic choice across the thresholds, against scipy."""
    ctx = qd.default_context()
    rng = np.random.default_rng(77)
    n = 24
    a0 = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    if kind == "antiherm":
        a0 = a0 - a0.conj().T
    a0 /= np.linalg.norm(a0, 1)
    thetas = {2: 8.0e-6, 4: 1.5e-3, 6: 1.6e-2, 9: 0.1, 12: 0.3, 16: 0.75}
    try:
        for deg, theta in thetas.items():
            ctx.set_option("expm_degree", deg)
            for scale in (theta, 3.7 * theta):           # at the edge, and with two squarings
                a = a0 * scale
                e, info = ctx.expm(a, return_info=True)
                ref = scipy.linalg.expm(a)
                err = np.linalg.norm(e - ref, 1) / np.linalg.norm(ref, 1)
                assert err < 3e-15 * (1 + int(info[0, 0])) + 2e-16, (deg, scale, err, info)
    finally:
        ctx.set_option("expm_degree", 0)
    for scale in (0.0, 1e-9, 7e-6, 9e-6, 1e-3, 2e-3, 0.015, 0.02, 0.09, 0.12, 0.29, 0.35, 0.7, 0.8, 1.4, 3.3, 11.0):
        a = a0 * scale
        e, info = ctx.expm(a, return_info=True)
        ref = scipy.linalg.expm(a)
        err = np.linalg.norm(e - ref, 1) / np.linalg.norm(ref, 1)
        assert err < 3e-15 * (1 + int(info[0, 0])) + 2e-16, (scale, err, info)


@pytest.mark.parametrize("seed", range(6))
def test_expm_action_matches_dense_expm(qd, seed):
    """States with few columns take the expm ACTION path (Taylor series of matrix-vector products on the
    batched contraction kernel, commutator-free Magnus-2); it must agree with the dense expm path and
    the oracle: frames / no frame (large norms -> scaling), sweeps and single instances, 1-2 columns."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(4200 + seed)
    n = [6, 33, 70, 130][seed % 4]
    k = int(rng.integers(1, 4))
    frame_kind = ["none", "full", "diag"][seed % 3]
    batch = [1, 4][seed % 2]
    m = [None, 2][(seed // 2) % 2]
    mo = 1 + seed % 2

    def herm(scale=1.0):
        a = crand(rng, n, n)
        return scale * (a + a.conj().T) / 2

    h_static = herm(6.0 if frame_kind == "none" else 2.0)   # no frame: ||h G|| of order 10 -> several scalings
    h_ops = np.array([herm() for _ in range(k)])
    frame = {"none": None, "diag": rng.normal(size=n), "full": h_static}[frame_kind]
    t_span, t_eval, max_dt = [0.0, 0.5], [0.0, 0.2, 0.5], 0.05

    def make_sigs():
        amps, nus, phs = rng.uniform(-1, 1, k), rng.uniform(0, 2, k), rng.uniform(-3, 3, k)
        return ([qd.Signal(lambda t, a=a: a * np.cos(0.7 * t) + 0j, nu, ph) for a, nu, ph in zip(amps, nus, phs)],
                (amps, nus, phs))

    def make_y0():
        y = crand(rng, n) if m is None else crand(rng, n, m)
        return y / np.linalg.norm(y)

    sig_sets = [make_sigs() for _ in range(batch)]
    y0s = [make_y0() for _ in range(batch)]
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    args = dict(t_span=t_span, y0=y0s if batch > 1 else y0s[0],
                signals=[s for s, _ in sig_sets] if batch > 1 else sig_sets[0][0], method="scipy_expm",
                max_dt=max_dt, t_eval=t_eval, magnus_order=mo)
    ctx = qd.default_context()
    act = solver.solve(**args)
    ctx.set_option("expm_action", 0)
    try:
        dense = solver.solve(**args)
    finally:
        ctx.set_option("expm_action", 1)
    act = act if isinstance(act, list) else [act]
    dense = dense if isinstance(dense, list) else [dense]
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
    for b in range(batch):
        assert_close(act[b].y, dense[b].y, 1e-11)
        amps, nus, phs = sig_sets[b][1]

        def coeff(t, amps=amps, nus=nus, phs=phs):
            return np.array([orc.signal_sum_value(np.array([a_ * np.cos(0.7 * t) + 0j]), [nu], [ph], t)
                             for a_, nu, ph in zip(amps, nus, phs)])

        _, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff, t_span, y0s[b], "scipy_expm", max_dt,
                                             t_eval=t_eval, magnus_order=mo)
        assert_close(act[b].y, y_ref, SOLVE_TOL)
        assert abs(np.linalg.norm(act[b].y[-1]) - 1.0) < 1e-10


@pytest.mark.parametrize("n,batch,m", [(63, 1, None), (64, 3, None), (65, 64, None), (127, 65, None), (128, 130, None),
                                       (129, 2, 2), (191, 33, 3), (200, 128, None), (256, 257, None), (70, 1, 64)])
def test_tile_boundary_shapes_rk4_and_expm(qd, n, batch, m):
    """Sizes that straddle the 64/128 tile and split-K boundaries (n, B*m on both sides of multiples of
    64/128; 1, few, many instances): batched RK4 and expm solves through Solver.solve vs the oracle."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(n * 1000 + batch)
    k = 3

    def herm():
        a = crand(rng, n, n)
        return (a + a.conj().T) / 2

    h_static = herm()
    h_ops = np.array([herm() for _ in range(k)])
    frame = h_static if n % 2 else rng.normal(size=n)
    params = [(rng.uniform(-1, 1, k), rng.uniform(0, 2, k), rng.uniform(-3, 3, k)) for _ in range(batch)]
    sigs = [[qd.Signal(lambda t, a=a: a * np.cos(0.7 * t) + 0j, nu, ph) for a, nu, ph in zip(*p)] for p in params]
    y0s = []
    for _ in range(batch):
        y = crand(rng, n) if m is None else crand(rng, n, m)
        y0s.append(y / np.linalg.norm(y))
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
    check = sorted(set([0, batch // 2, batch - 1]))
    for method, kw, t_span in (("RK4", {"max_dt": 0.01}, [0.0, 0.05]),
                               ("scipy_expm", {"max_dt": 0.02, "magnus_order": 2}, [0.0, 0.06])):
        res = solver.solve(t_span=t_span, y0=y0s if batch > 1 else y0s[0], signals=sigs if batch > 1 else sigs[0],
                           method=method, **kw)
        res = res if isinstance(res, list) else [res]
        assert len(res) == batch
        for b in check:
            amps, nus, phs = params[b]

            def coeff(t, amps=amps, nus=nus, phs=phs):
                return np.array([orc.signal_sum_value(np.array([a_ * np.cos(0.7 * t) + 0j]), [nu], [ph], t)
                                 for a_, nu, ph in zip(amps, nus, phs)])

            _, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff, t_span, y0s[b], method, kw["max_dt"],
                                                 magnus_order=kw.get("magnus_order", 1))
            assert_close(res[b].y, y_ref, SOLVE_TOL)


@pytest.mark.parametrize("kind,batch,n_steps", [("magnus", 37, 50), ("dyson", 37, 50), ("magnus", 10, 500),
                                                ("dyson", 3, 2000)])
def test_perturbative_sweep_batched_over_instances(qd, golden, kind, batch, n_steps):
    """List-mode sweeps of the perturbative solvers: all (instance, step) pairs are rows of one table cut
    into device chunks (several instances per chunk; chunks that split an instance) -- against the oracle's
    step loop instance by instance."""
    from oracle import dynamics_oracle as orc

    g = golden("perturbative")
    r, sig_w, t_c, dt, nu = g["q1_params"]
    cls = qd.DysonSolver if kind == "dyson" else qd.MagnusSolver
    sol = cls(operators=g["q1_ops"], rotating_frame=g["q1_frame"], dt=dt, carrier_freqs=[nu], chebyshev_orders=[1],
              expansion_order=6 if kind == "dyson" else 3, integration_method="DOP853", atol=1e-12, rtol=1e-12)
    rng = np.random.default_rng(batch * 7 + n_steps)
    amps, cents, phs = rng.uniform(0.3, 1.0, batch), rng.uniform(0.2, 3.0, batch), rng.uniform(-1, 1, batch)
    sigs = [[qd.Signal(lambda t, a=a, c=c: a * np.exp(-((t - c) ** 2) / 0.8), carrier_freq=nu, phase=p)]
            for a, c, p in zip(amps, cents, phs)]
    y0s = [crand(rng, 2, 2) for _ in range(batch)]
    res = sol.solve(t0=0.05, n_steps=n_steps, y0=y0s, signals=sigs)
    assert len(res) == batch
    d, basis = orc.frame_setup(g["q1_frame"])
    terms, labels, udt = g[f"q1_{kind}_terms"], g[f"q1_{kind}_labels"], g[f"q1_{kind}_udt"]
    for b in sorted(set([0, 1, batch // 2, batch - 1])):
        a, c, p = amps[b], cents[b], phs[b]
        cv = lambda t, a=a, c=c, p=p: a * np.exp(-((t - c) ** 2) / 0.8) * np.exp(1j * (2 * np.pi * nu * t + p))
        coeffs = orc.signal_list_envelope_dct([cv], [nu], [1], 0.05, dt, n_steps)
        want = orc.perturbative_solve(kind, terms, labels, udt, d, basis, coeffs, y0s[b], 0.05, n_steps, dt)
        assert_close(res[b].y[-1], want, 1e-9)
        assert_close(res[b].y[0], y0s[b], 0)


def test_cfg3_full_length_solve_vs_oracle(qd, cfg2):
    """BASELINE cfg 3 at FULL length: t_span [0, 5], max_dt 0.005 -> 1000 RK4 steps (4000 RHS evaluations per
    instance) of the n = 1024, k = 8 model, 128 instances in one batched device solve; one instance is
    integrated by the oracle over the same 1000 steps and compared at the final time, all of them for norm
    conservation.  (The full sweep differs only in the number of columns.)"""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg, (a_d, a, d, basis), stack = cfg2
    B = 128
    sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _rk4_points)
    assert len(sched.step_h) == 1000
    table, amps, phs = _table_for(cfg, range(B), sched.times)
    y0 = np.zeros((1024, 1), dtype=complex)
    y0[0, 0] = 1.0
    ys = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, y0, B, True)
    final = ys[:, -1, :, 0]
    assert np.max(np.abs(np.linalg.norm(final, axis=1) - 1.0)) < 1e-8
    from threadpoolctl import threadpool_limits

    b = 77

    def rhs(t, y):
        c = workloads.gaussian_coefficient_table(np.array([t]), amps[b], phs[b], cfg["carrier"], 5.0)[0]
        return orc.generator_rhs(a_d, a, c, d, None, t, y)

    with threadpool_limits(limits=8):   # the oracle's matvecs run best on a few BLAS threads (see bench.py)
        _, yref = orc.rk4_solve(rhs, cfg["t_span"], y0[:, 0], cfg["max_dt"])
    assert_close(final[b], yref[-1], SOLVE_TOL)


def test_cfg4_full_size_vectorised_lindblad_vs_matrix_form(qd):
    """BASELINE cfg 4 at FULL size (6 qubits, N = 4096 superoperator, 4 dissipators, scipy_expm, no
    frame): the vectorised device propagation against an independent CPU evaluation of the same steps
    that never forms the superoperator -- expm(h L(t_mid)) rho as a scaled Taylor series of the n x n
    matrix form of the Lindbladian (oracle.lindblad_rhs) -- plus trace / Hermiticity / positivity."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    cfg = workloads.lindblad_config()
    assert cfg["h_d"].shape == (64, 64)
    amps, phases = workloads.sweep_parameters(0, 6)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], vectorized=True)
    assert solver.model.stack.n == 4096
    h = cfg["max_dt"]
    n_steps = 3
    t0 = 2.4
    res = solver.solve(t_span=[t0, t0 + n_steps * h], y0=cfg["rho0"].flatten(order="F"), signals=sigs,
                       method="scipy_expm", max_dt=h)
    rho_dev = res.y[-1].reshape(64, 64, order="F")
    h_d, h_ops, n_static, l_ops, d, basis = orc.lindblad_model_build(cfg["h_d"], cfg["ops"],
                                                                     cfg["static_dissipators"], None, None)
    rho = cfg["rho0"].astype(complex)
    for st in range(n_steps):
        t_mid = t0 + st * h + h / 2          # Magnus order 1: Omega = h L(t + h/2)
        coeffs = np.array([s(t_mid) for s in sigs])
        scal = 64                            # ||h L|| ~ 10 -> ||h L / 64|| < 0.2
        for _ in range(scal):
            term = rho
            acc = rho.copy()
            for j in range(1, 16):
                term = orc.lindblad_rhs(h_d, h_ops, n_static, l_ops, coeffs, None, d, t_mid, term) * (h / (scal * j))
                acc = acc + term
            rho = acc
    assert_close(rho_dev, rho, SOLVE_TOL)
    assert abs(np.trace(rho_dev) - 1.0) < 1e-12
    assert np.linalg.norm(rho_dev - rho_dev.conj().T) < 1e-12
    assert np.min(np.linalg.eigvalsh((rho_dev + rho_dev.conj().T) / 2)) > -1e-12


def test_cfg5_full_size_magnus2_vs_cpu_action(qd):
    """BASELINE cfg 5 at FULL size (12 qubits, n = 4096, k = 8, diagonal rotating frame, scipy_expm with
    magnus_order 2, max_dt 0.25): two sweep instances, one step, in one batched device solve -- against
    a CPU evaluation of expm(Omega_2) y0 that uses the oracle's generators G(t1), G(t2) and a
    commutator-free Taylor series (never forming Omega or its exponential) -- plus norm conservation."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
    frame = np.diag(cfg["h_d"]).real.copy()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
    assert solver.model.stack.n == 4096

    def sig_list(b):
        amps, phases = workloads.sweep_parameters(b, 8)
        return [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
                for a, nu, ph in zip(amps, cfg["carrier"], phases)]

    sweeps = [sig_list(0), sig_list(1)]
    t0, h = 2.25, 0.25
    res = solver.solve(t_span=[t0, t0 + h], y0=cfg["y0"], signals=sweeps, method="scipy_expm", max_dt=h,
                       magnus_order=2)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], frame)
    c1, c2 = 0.5 - np.sqrt(3) / 6, 0.5 + np.sqrt(3) / 6
    for b in range(2):
        g1 = orc.generator_evaluate(a_d, a, np.array([s(t0 + c1 * h) for s in sweeps[b]]), d, basis, t0 + c1 * h)
        g2 = orc.generator_evaluate(a_d, a, np.array([s(t0 + c2 * h) for s in sweeps[b]]), d, basis, t0 + c2 * h)

        def omega(v):   # fixed_step_solvers.py:348-363 applied to a vector
            u1, u2 = g1 @ v, g2 @ v
            return (h / 2) * (u1 + u2) + (np.sqrt(3) / 12) * h * h * (g2 @ u1 - g1 @ u2)

        y = cfg["y0"].astype(complex)
        for _ in range(4):                      # ||Omega|| ~ 0.05: four scalings, Taylor degree 12
            term, acc = y, y.copy()
            for j in range(1, 13):
                term = omega(term) / (4 * j)
                acc = acc + term
            y = acc
        assert_close(res[b].y[-1], y, SOLVE_TOL)
        assert abs(np.linalg.norm(res[b].y[-1]) - 1.0) < 1e-12


def test_large_device_table_sweep_small_system(qd):
    """4096 instances of a 3-qubit model, 2000 RK4 steps, DiscreteSignal pulses: a 393 MB coefficient
    table evaluated on the device and handed to the solver device-to-device.  (Regression: that copy
    used to run unordered on the null stream, so the first stages could read a half-filled table --
    wrong by 1e-5 for late instances, not reproducibly.)  Both RK4 routes -- the persistent tiny-system
    kernel and the batched per-stage contraction -- against each other and against the oracle."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(3, n_drives=3)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    batch, steps = 4096, 2000
    rng = np.random.default_rng(3)
    raw = [[(rng.uniform(0.1, 1, 20) * np.exp(1j * rng.uniform(0, 1, 20)), nu, rng.uniform(0, 1)) for nu in cfg["carrier"]]
           for _ in range(batch)]
    sigs = [[qd.DiscreteSignal(dt=0.1, samples=s, carrier_freq=nu, phase=ph) for s, nu, ph in inst] for inst in raw]
    res = {}
    try:
        for tag, flag in (("tiny", 1), ("batched", 0)):
            ctx.set_option("tiny_rk4", flag)
            r = solver.solve(t_span=[0.0, 2.0], y0=cfg["y0"], signals=sigs, method="RK4", max_dt=2.0 / steps)
            res[tag] = np.array([x.y[-1] for x in r])
    finally:
        ctx.set_option("tiny_rk4", 1)
    assert_close(res["batched"], res["tiny"], 1e-12)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])
    for b in (0, 2811, batch - 1):
        def coeff(t, b=b):
            return np.array([orc.signal_sum_value(np.array([orc.discrete_envelope(s, 0.1, 0.0, t)]), [nu], [ph], t)
                             for s, nu, ph in raw[b]])

        _, yref = orc.solve_generator_model(a_d, a, d, basis, coeff, [0.0, 2.0], cfg["y0"], "RK4", 2.0 / steps)
        assert_close(res["tiny"][b], yref[-1], SOLVE_TOL)
        assert_close(res["batched"][b], yref[-1], SOLVE_TOL)


@pytest.mark.parametrize("batch,m", [(2, 1), (3, 1), (5, 1), (8, 1), (1, 2), (2, 3), (1, 8), (4, 2)])
def test_multi_column_streaming_kernel(qd, cfg2, batch, m):
    """2..8 state columns at n = 1024 take the multi-column streaming kernel (operator rows read once for
    all columns, own coefficients per instance): against the MFMA path and the oracle, RK4 and expm action."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg, (a_d, a, d, basis), stack = cfg2
    rng = np.random.default_rng(batch * 10 + m)
    sched = FixedStepSchedule([2.4, 2.42], None, cfg["max_dt"], _rk4_points)
    table, amps, phs = _table_for(cfg, range(batch), sched.times)
    y0 = crand(rng, batch, 1024, m)
    outs = {}
    try:
        for flag in (1, 2, 0):   # 1: planar multi-column kernel (single-plane stack), 2: interleaved one, 0: MFMA
            stack.ctx.set_option("multi_stream", 1 if flag else 0)
            stack.ctx.set_option("stream_planes", 0 if flag == 2 else 1)
            outs[flag] = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                         sched.n_save, y0, batch, False)
    finally:
        stack.ctx.set_option("multi_stream", 1)
        stack.ctx.set_option("stream_planes", 1)
    assert_close(outs[1], outs[0], 1e-12)
    assert_close(outs[2], outs[0], 1e-12)
    b = batch - 1

    def rhs(t, y):
        c = workloads.gaussian_coefficient_table(np.array([t]), amps[b], phs[b], cfg["carrier"], 5.0)[0]
        return orc.generator_rhs(a_d, a, c, d, None, t, y)

    _, yref = orc.rk4_solve(rhs, [2.4, 2.42], y0[b], cfg["max_dt"])
    assert_close(outs[1][b, -1], yref[-1], SOLVE_TOL)
    # single evaluation with m columns (one instance)
    c = rng.normal(size=8)
    ym = crand(rng, 1024, max(m, 2))
    try:
        stack.ctx.set_option("multi_stream", 1)
        got = stack.eval_rhs(c, 0.37, ym)
        stack.ctx.set_option("multi_stream", 0)
        want = stack.eval_rhs(c, 0.37, ym)
    finally:
        stack.ctx.set_option("multi_stream", 1)
    assert_close(got, want, 1e-12)
    assert_close(got, orc.generator_rhs(a_d, a, c, d, None, 0.37, ym), EVAL_TOL)


def test_repeated_solves_are_bitwise_identical(qd, monkeypatch):
    """No atomics, no unordered copies: the same solve must give the same bits every time (a data race
    shows up as irreproducibility long before it shows up as a wrong answer).  Sweeps with a device
    coefficient table through the batched RK4 stages, the tiny kernel, the expm action, the
    parallel-in-time method, and a mid-size model through split-K and the multi-column kernel."""
    from qiskit_dynamics_amd import solvers as S
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    monkeypatch.setattr(S, "DEVICE_SIGNAL_TABLE_MIN", 0)
    rng = np.random.default_rng(17)

    def sweep(cfg, batch):
        return [[qd.DiscreteSignal(dt=0.05, samples=rng.uniform(0.1, 1, 40) * np.exp(1j * rng.uniform(0, 1, 40)),
                                   carrier_freq=nu, phase=rng.uniform(0, 1)) for nu in cfg["carrier"]]
                for _ in range(batch)]

    cases = []
    cfg = W.schrodinger_config(3, n_drives=3)
    s3 = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    sig3 = sweep(cfg, 1500)
    for method, kw, opts in (("RK4", {}, {"tiny_rk4": 1}), ("RK4", {}, {"tiny_rk4": 0}),
                             ("scipy_expm", {"magnus_order": 2}, {"tiny_rk4": 0}),
                             ("scipy_expm", {"magnus_order": 1}, {"tiny_rk4": 1})):
        cases.append((s3, dict(t_span=[0.0, 2.0], y0=cfg["y0"], signals=sig3, method=method, max_dt=0.002, **kw), opts))
    cases.append((s3, dict(t_span=[0.0, 2.0], y0=cfg["y0"], signals=sig3[0], method="hip_expm_parallel", max_dt=0.002,
                           magnus_order=2), {}))
    cfg = W.schrodinger_config(8)
    s8 = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    for batch in (5, 200):
        cases.append((s8, dict(t_span=[0.0, 0.1], y0=cfg["y0"], signals=sweep(cfg, batch), method="RK4", max_dt=0.005), {}))
    for solver, kw, opts in cases:
        try:
            for k_, v_ in opts.items():
                ctx.set_option(k_, v_)
            runs = []
            for _ in range(3):
                r = solver.solve(**kw)
                r = r if isinstance(r, list) else [r]
                runs.append(np.array([x.y for x in r]))
        finally:
            ctx.set_option("tiny_rk4", 1)
        assert np.array_equal(runs[0], runs[1]) and np.array_equal(runs[0], runs[2]), (kw["method"], opts)


@pytest.mark.parametrize("n,method,mo", [(4, "RK4", 1), (27, "RK4", 1), (64, "scipy_expm", 2), (9, "scipy_expm", 1)])
def test_auto_parallel_in_time_routing(qd, monkeypatch, n, method, mo):
    """One small trajectory with many steps is routed to the parallel-in-time implementation of the SAME
    method (all step propagators at once + tree product); the numbers must agree with the sequential
    route to rounding, and the switch must bring the sequential route back."""
    from qiskit_dynamics_amd import solvers as S

    rng = np.random.default_rng(n)

    def herm(s_=1.0):
        a_ = crand(rng, n, n)
        return s_ * (a_ + a_.conj().T) / 2

    h0 = np.diag(rng.normal(size=n) * 20.0).astype(complex)
    ops = np.array([herm(0.3) for _ in range(3)])
    solver = qd.Solver(static_hamiltonian=h0, hamiltonian_operators=ops, rotating_frame=h0)
    sigs = [qd.Signal(lambda t, j=j: 0.3 * np.cos(0.5 * t + j) + 0j, 3.0 + 0.2 * j, 0.1 * j) for j in range(3)]
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    kw = dict(t_span=[0.0, 1.0], y0=y0, signals=sigs, method=method, max_dt=1.0 / 600, t_eval=[0.0, 0.37, 1.0])
    if method == "scipy_expm":
        kw["magnus_order"] = mo
    calls = []
    orig = qd.Stack.parallel_solve

    def spy(self, *a_, **k_):
        calls.append(1)
        return orig(self, *a_, **k_)

    monkeypatch.setattr(qd.Stack, "parallel_solve", spy)
    auto = solver.solve(**kw)
    assert len(calls) == 1, "the parallel-in-time route was not taken"
    monkeypatch.setattr(S, "AUTO_PARALLEL_IN_TIME", False)
    seq = solver.solve(**kw)
    assert len(calls) == 1, "the switch did not disable the routing"
    assert_close(auto.t, seq.t, 0)
    assert_close(auto.y, seq.y, 1e-11)
    assert abs(np.linalg.norm(auto.y[-1]) - 1.0) < 1e-8


@pytest.mark.parametrize("n,magn", [(200, 9.0), (300, 25.0)])
def test_krylov_expm_action_matches_taylor(qd, n, magn):
    """One column, no rotating frame, ||h G||_1 of order 10..50: the expm action as a Chebyshev series (default
    for nearly skew-Hermitian generators), as an Arnoldi process (krylov = 2: CGS2 on the device, small expm of
    the Hessenberg block, Saad's estimate) and as the scaled Taylor series (~15 products per unit of norm),
    against each other, the dense expm and the oracle."""
    from oracle import dynamics_oracle as orc

    ctx = qd.default_context()
    rng = np.random.default_rng(n)
    evals = rng.uniform(-magn, magn, n) * 20.0          # spectral radius of h G about magn at h = 0.05
    q_, _ = np.linalg.qr(crand(rng, n, n))
    h_static = (q_ * evals) @ q_.conj().T
    h_static = (h_static + h_static.conj().T) / 2
    a_ = crand(rng, n, n)
    h_ops = np.array([(a_ + a_.conj().T) / 2 * 0.3])
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops)
    sig = [qd.Signal(lambda t: 0.5 * np.cos(t) + 0j, 1.0, 0.2)]
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    kw = dict(t_span=[0.0, 0.15], y0=y0, signals=sig, method="scipy_expm", max_dt=0.05)
    res, products = {}, {}
    try:
        for tag, opts in (("chebyshev", {}), ("krylov", {"krylov": 2}), ("taylor", {"krylov": 0, "chebyshev": 0}),
                          ("dense", {"expm_action": 0})):
            for k_, v_ in opts.items():
                ctx.set_option(k_, v_)
            ctx.reset_counters()
            ctx.set_option("profile", 1)
            res[tag] = solver.solve(**kw).y
            counters = {c: ctx.counters(c)["launches"] for c in ("rhs_stream", "zgemm")}
            ctx.set_option("profile", 0)
            ctx.set_option("krylov", 1)
            ctx.set_option("chebyshev", 1)
            ctx.set_option("expm_action", 1)
            products[tag] = counters["rhs_stream"]
            if tag == "chebyshev":   # the default: about rho + 10 rho^(1/3) + 10 products per step (rho = the norm
                assert counters["zgemm"] == 0, counters   # BOUND of h G, several times the spectral radius here)
            if tag == "krylov":
                assert counters["zgemm"] > 0 and counters["rhs_stream"] < 3 * 64, counters   # Arnoldi was used
            if tag == "taylor":
                assert counters["zgemm"] == 0 and counters["rhs_stream"] >= 3 * 64, counters
    finally:
        ctx.set_option("profile", 0)
        ctx.set_option("krylov", 1)
        ctx.set_option("chebyshev", 1)
        ctx.set_option("expm_action", 1)
    assert products["chebyshev"] < 0.85 * products["taylor"], products
    assert_close(res["chebyshev"], res["taylor"], 1e-11)
    assert_close(res["chebyshev"], res["dense"], 1e-11)
    assert_close(res["krylov"], res["dense"], 1e-11)
    assert_close(res["krylov"], res["taylor"], 1e-11)
    assert_close(res["krylov"], res["dense"], 1e-11)
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, None)
    coeff = lambda t: np.array([orc.signal_sum_value(np.array([0.5 * np.cos(t) + 0j]), [1.0], [0.2], t)])
    _, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff, [0.0, 0.15], y0, "scipy_expm", 0.05)
    assert_close(res["krylov"], y_ref, SOLVE_TOL)
    assert abs(np.linalg.norm(res["krylov"][-1]) - 1.0) < 1e-12


def test_dyson_magnus_two_transmon_reference_scenario(qd):
    """The reference's own acceptance scenario (test_dyson_magnus_solvers.py:142-330): two coupled 5-level
    transmons (dim 25), Gaussian drives on both, DysonSolver with expansion_order 6 (3002 terms) and
    MagnusSolver with order 3, 1000 steps of dt = 0.01; criterion = the reference's fidelity test against
    the direct solution, |1 - |<U_pert, U_direct>|^2 / dim^4| < 1e-6."""
    w_c, w_t = 2 * np.pi * 5.033, 2 * np.pi * 4.067
    alpha_c, alpha_t, coupling = 2 * np.pi * (-0.33534), 2 * np.pi * (-0.33834), 2 * np.pi * 0.002
    dim = 5
    a = np.diag(np.sqrt(np.arange(1, dim)), 1)
    num = np.diag(np.arange(dim)).astype(float)
    i1, i2 = np.eye(dim), np.eye(dim**2)
    a0, a1 = np.kron(a, i1), np.kron(i1, a)
    n0, n1 = np.kron(num, i1), np.kron(i1, num)
    h0 = (w_c * n0 + 0.5 * alpha_c * n0 @ (n0 - i2) + w_t * n1 + 0.5 * alpha_t * n1 @ (n1 - i2)
          + coupling * (a0 @ a1.T + a0.T @ a1))
    hdc, hdt = 2 * np.pi * (a0 + a0.T), 2 * np.pi * (a1 + a1.T)
    r = 0.2
    sig_w = 0.399128 / r
    gauss = qd.Signal(lambda t: np.exp(-((t - 3.5 * sig_w) ** 2) / (2 * sig_w**2)), carrier_freq=5.0)
    dt, n_steps = 0.01, 1000
    y0 = np.eye(dim**2, dtype=complex)
    direct = qd.Solver(static_hamiltonian=h0, hamiltonian_operators=[hdc, hdt], rotating_frame=h0).solve(
        t_span=[0.0, dt * n_steps], y0=y0, signals=[gauss, gauss], method="RK4", max_dt=dt / 20).y[-1]
    assert np.linalg.norm(direct.conj().T @ direct - y0) < 1e-8
    for cls, order in ((qd.DysonSolver, 6), (qd.MagnusSolver, 3)):
        sol = cls(operators=[-1j * hdc, -1j * hdt], rotating_frame=-1j * h0, dt=dt, carrier_freqs=[5.0, 5.0],
                  chebyshev_orders=[1, 1], expansion_order=order, integration_method="DOP853", atol=1e-10, rtol=1e-10)
        yf = sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=[gauss, gauss]).y[-1]
        infidelity = abs(1.0 - abs((yf.conj() * direct).sum()) ** 2 / dim**4)
        assert infidelity < 1e-6, (cls.__name__, infidelity)


# ------------------------------------------------------------------------------------------------
# block-sparse stacks: operators in a computational / diagonal-frame basis are dense arrays whose
# 16 x 16 blocks are almost all exactly zero; the work-list kernels read only the non-zero blocks
# ------------------------------------------------------------------------------------------------
def _chain_sweep(qd, nq, nb, t_final):
    from qiskit_dynamics_amd import workloads as W

    cfg = W.schrodinger_config(nq, min(nq, 8), t_final, 0.01)
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, len(cfg["ops"]))
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    return cfg, sweeps


@pytest.mark.usefixtures("per_launch_routes")
@pytest.mark.parametrize("nq,nb", [(8, 1), (8, 3), (8, 8), (8, 24), (9, 130)])
def test_block_sparse_routes_match_dense_routes(qd, nq, nb):
    """8/9-qubit chain in the diagonal frame (n = 256 / 512, 9 operators, ~9 % of the 16 x 16 blocks non-zero):
    one column and 2..8 columns (rhs_blocks_kernel), 64-column and 128-column MFMA tiles with work lists
    (SPARSE zgemm_seg_kernel) against the dense kernels on the same inputs, for RK4 and for the expm
    action (Magnus 1 and 2); the routes are checked through the launch counters."""
    ctx = qd.default_context()
    cfg, sweeps = _chain_sweep(qd, nq, nb, 0.2)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       rotating_frame=np.diag(cfg["h_d"]).real.copy())
    rng = np.random.default_rng(5)
    y0 = rng.normal(size=2**nq) + 1j * rng.normal(size=2**nq)
    y0 /= np.linalg.norm(y0)
    sig = sweeps if nb > 1 else sweeps[0]
    for method, kw in (("RK4", {}), ("scipy_expm", {"magnus_order": 1}), ("scipy_expm", {"magnus_order": 2})):
        out = {}
        for flag in (1, 0):
            ctx.set_option("skip_zero_blocks", flag)
            ctx.set_option("resident_rk4", 0)      # (one RK4 trajectory would otherwise take the resident kernel)
            ctx.reset_counters()
            ctx.set_option("profile", 1)
            try:
                r = solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sig, method=method, max_dt=0.01, **kw)
            finally:
                ctx.set_option("profile", 0)
                ctx.set_option("skip_zero_blocks", 1)
                ctx.set_option("resident_rk4", 1)
            blocks = ctx.counters("rhs_blocks")["launches"] + ctx.counters("rhs_blocks_gemm")["launches"]
            # (without block skipping a sweep of >= 256 state columns is a dense problem for the combine + apply kernel,
            # round 4; with it, these very sparse operators stay on the work lists)
            dense = (ctx.counters("rhs_stream")["launches"] + ctx.counters("rhs_gemm")["launches"]
                     + ctx.counters("rhs_combine")["launches"])
            assert (blocks > 0 and dense == 0) if flag else (blocks == 0 and dense > 0), (method, flag, blocks, dense)
            out[flag] = np.stack([x.y[-1] for x in r]) if nb > 1 else r.y[-1][None]
        assert_close(out[1], out[0], 1e-13)
        assert np.max(np.abs(np.linalg.norm(out[1], axis=1) - 1.0)) < 1e-8
        if nb > 8 and method == "RK4":
            # unsplit work lists: more than 64 entries per workgroup (the list vector is refilled in flight)
            ctx.set_option("force_splits", 1)
            try:
                r = solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sig, method=method, max_dt=0.01, **kw)
            finally:
                ctx.set_option("force_splits", 0)
            assert_close(np.stack([x.y[-1] for x in r]), out[0], 1e-13)


def test_block_sparse_against_oracle(qd):
    """The work-list kernels against the CPU restatement (not only against the dense device kernels):
    8-qubit chain, full evaluate_rhs at several times for 1 and 5 columns, and a short RK4 solve."""
    from oracle import dynamics_oracle as orc

    cfg, sweeps = _chain_sweep(qd, 8, 1, 0.1)
    frame = np.diag(cfg["h_d"]).real.copy()
    m = qd.HamiltonianModel(static_operator=cfg["h_d"], operators=cfg["ops"], signals=sweeps[0], rotating_frame=frame)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], frame)

    def coeffs(t):
        return np.array([np.real(s(t)) for s in sweeps[0]])

    rng = np.random.default_rng(2)
    for cols in (None, 5):
        y = rng.normal(size=(256,) if cols is None else (256, cols)) + 0j
        for t in (0.0, 0.37, 1.9):
            assert_close(m.evaluate_rhs(t, y), orc.generator_rhs(a_d, a, coeffs(t), d, basis, t, y, False), EVAL_TOL)
    r = qd.solve_lmde(m, [0.0, 0.1], cfg["y0"], method="RK4", max_dt=0.01)
    _, ref = orc.solve_generator_model(a_d, a, d, basis, coeffs, [0.0, 0.1], cfg["y0"], "RK4", 0.01)
    assert_close(r.y[-1], ref[-1], SOLVE_TOL)


def test_block_sparse_vectorised_lindblad(qd):
    """4-qubit vectorised Lindbladian without a frame (N = 256 Kronecker superoperators, block sparse):
    scipy_expm through the work-list kernels (Taylor action and Arnoldi) against the dense kernels."""
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.lindblad_config(n_qubits=4, n_drives=4, n_diss=4, gamma=1e-2, t_final=1.0, max_dt=0.05)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, 0.1 * a)
            for a, nu in zip((0.9, 0.5, 0.7, 0.3), cfg["carrier"])]
    m = qd.LindbladModel(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], hamiltonian_signals=sigs,
                         static_dissipators=cfg["static_dissipators"], vectorized=True)
    y0 = cfg["rho0"].flatten(order="F")
    out = {}
    for flag in (1, 0):
        for kry in (2, 0):   # 2: Arnoldi on every step (the automatic rule prefers the series for cheap block products)
            ctx.set_option("skip_zero_blocks", flag)
            ctx.set_option("krylov", kry)
            ctx.set_option("resident_rk4", 0)      # (the series would otherwise run inside ell_resident_kernel)
            ctx.reset_counters()
            ctx.set_option("profile", 1)
            try:
                r = qd.solve_lmde(m, [0.0, 1.0], y0, method="scipy_expm", max_dt=0.05)
            finally:
                ctx.set_option("profile", 0)
                ctx.set_option("skip_zero_blocks", 1)
                ctx.set_option("krylov", 1)
                ctx.set_option("resident_rk4", 1)
            assert (ctx.counters("rhs_blocks")["launches"] > 0) == bool(flag)
            out[(flag, kry)] = r.y[-1]
    for key in ((1, 0), (0, 2), (0, 0)):
        assert_close(out[(1, 2)], out[key], 1e-12)
    rho = out[(1, 2)].reshape(16, 16, order="F")
    assert abs(np.trace(rho) - 1.0) < 1e-12


@pytest.mark.usefixtures("per_launch_routes")
@pytest.mark.parametrize("nb", [24, 130])
def test_block_sparse_mfma_route_mixed_and_complex_planes(qd, nb):
    """The SPARSE MFMA instantiations other than the single-plane ones: (a) a sweep of 5-qubit vectorised
    Lindbladians without a frame (N = 1024; imaginary Hamiltonian part, real dissipator part: per-segment
    run-time plane modes), (b) a 9-qubit chain driven through Y operators in the lab frame, where every
    drive operator -iY is purely real and the static one purely imaginary (mixed), and (c) drives
    X + Y with both planes occupied (dense complex mode).  Work lists against the dense kernels."""
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()

    def both_routes(fn):
        out = {}
        for flag in (1, 0):
            ctx.set_option("skip_zero_blocks", flag)
            ctx.reset_counters()
            ctx.set_option("profile", 1)
            try:
                r = fn()
            finally:
                ctx.set_option("profile", 0)
                ctx.set_option("skip_zero_blocks", 1)
            assert (ctx.counters("rhs_blocks_gemm")["launches"] > 0) == bool(flag)
            out[flag] = np.stack([x.y[-1] for x in r])
        return out

    # (a) vectorised Lindblad sweep
    cfg = W.lindblad_config(n_qubits=5, n_drives=5, n_diss=5, gamma=1e-2, t_final=0.2, max_dt=0.05)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], vectorized=True)
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, 5)
        sweeps.append([qd.Signal(float(a), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    y0 = cfg["rho0"].flatten(order="F")
    for method, kw in (("RK4", {"max_dt": 0.01}), ("scipy_expm", {"max_dt": 0.05})):
        out = both_routes(lambda: solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sweeps, method=method, **kw))
        assert_close(out[1], out[0], 1e-12)
        tr = [abs(np.trace(v.reshape(32, 32, order="F")) - 1.0) for v in out[1]]
        assert max(tr) < 1e-10

    # (b), (c) Hamiltonians with Y and X + iY-type drives, lab frame (no rotating frame)
    nq = 9
    h_d, ops_x, nu = W.chain_hamiltonian(nq, 4)
    y_mat = np.array([[0, -1j], [1j, 0]])
    ops_y = np.stack([2 * np.pi * 0.02 * W.embed(y_mat, q, nq) / 2 for q in range(4)])
    rng = np.random.default_rng(8)
    y0 = rng.normal(size=2**nq) + 1j * rng.normal(size=2**nq)
    y0 /= np.linalg.norm(y0)
    for ops in (ops_y, ops_x + ops_y):
        solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops)
        sweeps = []
        for b in range(nb):
            amps, phases = W.sweep_parameters(b, 4)
            sweeps.append([qd.Signal(float(a), f, ph) for a, f, ph in zip(amps, nu[:4], phases)])
        out = both_routes(lambda: solver.solve(t_span=[0.0, 0.02], y0=y0, signals=sweeps, method="RK4", max_dt=0.002))
        assert_close(out[1], out[0], 1e-12)


@pytest.mark.usefixtures("per_launch_routes")
@pytest.mark.parametrize("bm", [16, 32, 64, 128])
@pytest.mark.parametrize("nb", [24, 130])
def test_block_sparse_every_panel_height(qd, bm, nb):
    """The automatic rule picks the row-panel height with the least listed work (16 rows for scattered
    patterns); `sparse_bm` pins each of the compiled tiles (16x64, 16x128, 32x64, 32x128, 64x64, 128x128 with
    work lists) so that all of them are checked against the dense kernels, RK4 and Magnus-2 action."""
    ctx = qd.default_context()
    cfg, sweeps = _chain_sweep(qd, 9, nb, 0.1)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       rotating_frame=np.diag(cfg["h_d"]).real.copy())
    rng = np.random.default_rng(6)
    y0 = rng.normal(size=512) + 1j * rng.normal(size=512)
    y0 /= np.linalg.norm(y0)
    for method, kw in (("RK4", {}), ("scipy_expm", {"magnus_order": 2})):
        out = {}
        for flag in (1, 0):
            ctx.set_option("skip_zero_blocks", flag)
            ctx.set_option("sparse_bm", bm if flag else 0)
            ctx.reset_counters()
            ctx.set_option("profile", 1)
            try:
                r = solver.solve(t_span=[0.0, 0.1], y0=y0, signals=sweeps, method=method, max_dt=0.01, **kw)
            finally:
                ctx.set_option("profile", 0)
                ctx.set_option("skip_zero_blocks", 1)
                ctx.set_option("sparse_bm", 0)
            assert (ctx.counters("rhs_blocks_gemm")["launches"] > 0) == bool(flag)
            out[flag] = np.stack([x.y[-1] for x in r])
        assert_close(out[1], out[0], 1e-13)


@pytest.mark.parametrize("n,with_hd,n_s,k_h,k_l,framed", [(5, True, 2, 2, 1, True), (8, False, 3, 0, 2, False),
                                                            (4, True, 0, 3, 0, True), (16, True, 4, 4, 2, False)])
def test_lindblad_superoperators_built_on_device(qd, n, with_hd, n_s, k_h, k_l, framed):
    """a13: the vectorised Lindblad stack assembled on the device from the n x n operators
    (midyn_stack_create_lindblad) against the stack uploaded from the host Kronecker formulas
    (models.vec_commutator / vec_dissipator), entry for entry through the generator evaluation, with every
    combination of absent parts."""
    from qiskit_dynamics_amd import models as M

    ctx = qd.default_context()
    rng = np.random.default_rng(100 * n + n_s)

    def herm(k):
        a = crand(rng, k, n, n)
        return (a + np.swapaxes(a.conj(), -1, -2)) / 2

    h_d = herm(1)[0] if with_hd else None
    h_ops = herm(k_h) if k_h else None
    n_static = crand(rng, n_s, n, n) if n_s else None
    l_ops = crand(rng, k_l, n, n) if k_l else None
    fim = rng.normal(size=n * n) if framed else None
    dev = qd.Stack.from_lindblad(ctx, h_d, h_ops, n_static, l_ops, fim)
    s_d = None
    if h_d is not None:
        s_d = M.vec_commutator(h_d)
    if n_static is not None:
        nd = np.sum(M.vec_dissipator(n_static), axis=0)
        s_d = nd if s_d is None else s_d + nd
    parts = ([M.vec_commutator(h_ops)] if h_ops is not None else []) + ([M.vec_dissipator(l_ops)] if l_ops is not None else [])
    s_ops = None if not parts else np.concatenate(parts, axis=0)
    host = qd.Stack(ctx, s_ops, s_d, fim)
    assert (dev.n, dev.k, dev.has_static, dev.has_frame) == (host.n, host.k, host.has_static, host.has_frame)
    for t in (0.0, 0.7):
        c = rng.normal(size=k_h + k_l)
        g_dev, g_host = dev.eval_generator(c, t), host.eval_generator(c, t)
        assert_close(g_dev, g_host, 1e-14)


def _count_products(ctx, fn):
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        r = fn()
    finally:
        ctx.set_option("profile", 0)
    n = sum(ctx.counters(c)["launches"] for c in ("rhs_stream", "rhs_gemm", "rhs_blocks", "rhs_blocks_gemm"))
    return r, n


def test_chebyshev_action_sweep_with_frame_and_fallbacks(qd):
    """Chebyshev expm action (default for nearly skew-Hermitian generators): (a) a sweep of 12 instances of a
    7-qubit chain without a frame (large ||hG||) and in the full frame, against the Taylor series
    (chebyshev = 0) with fewer products; (b) a strongly dissipative Lindbladian, where the stability check
    must reject the series (same products as with chebyshev = 0, same results); (c) the series forced on a
    small norm (chebyshev = 2)."""
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(7, 4, 1.0, 0.05)
    sweeps = []
    for b in range(12):
        amps, phases = W.sweep_parameters(b, 4)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.cos(0.3 * t) + 0j, nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    rng = np.random.default_rng(4)
    y0 = crand(rng, 128)
    y0 /= np.linalg.norm(y0)
    ctx.set_option("combine_sweep", 0)      # products are counted as launches here: a launch per series term, not the one-launch
    try:                                     # kernel of small sweeps (tests/test_gpu_combine.py covers that one)
        for frame, expect_fewer in ((None, True), (cfg["h_d"], None)):
            solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
            fn = lambda: solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sweeps, method="scipy_expm", max_dt=0.05)
            ctx.set_option("chebyshev", 1)
            r1, n1 = _count_products(ctx, fn)
            ctx.set_option("chebyshev", 0)
            r0, n0 = _count_products(ctx, fn)
            a1, a0 = np.stack([x.y[-1] for x in r1]), np.stack([x.y[-1] for x in r0])
            assert_close(a1, a0, 1e-12)
            assert np.max(np.abs(np.linalg.norm(a1, axis=1) - 1.0)) < 1e-12
            if expect_fewer:
                assert n1 < 0.6 * n0, (n1, n0)
            # (c) forced on whatever norm this is
            ctx.set_option("chebyshev", 2)
            r2, _ = _count_products(ctx, fn)
            assert_close(np.stack([x.y[-1] for x in r2]), a0, 1e-12)
        # Magnus 2 on the large-norm generator (sweep, with and without a frame; one trajectory on the streaming path)
        for frame, sig in ((None, sweeps), (np.diag(cfg["h_d"]).real.copy(), sweeps), (None, sweeps[0])):
            solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
            fn = lambda: solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sig, method="scipy_expm", max_dt=0.05,
                                      magnus_order=2)
            ctx.set_option("chebyshev", 1)
            r1, n1 = _count_products(ctx, fn)
            ctx.set_option("chebyshev", 0)
            r0, n0 = _count_products(ctx, fn)
            as_arr = lambda r: np.stack([x.y[-1] for x in r]) if isinstance(r, list) else r.y[-1][None]
            assert_close(as_arr(r1), as_arr(r0), 1e-12)
            if frame is None:
                assert n1 < 0.7 * n0, (n1, n0)
        # (b) strong dissipation: Hermitian part comparable to the norm -> the series must not be used
        lc = W.lindblad_config(n_qubits=3, n_drives=3, n_diss=3, gamma=40.0, t_final=1.0, max_dt=0.05)
        sig = [qd.Signal(0.4, nu, 0.1) for nu in lc["carrier"]]
        m = qd.LindbladModel(static_hamiltonian=lc["h_d"], hamiltonian_operators=lc["ops"], hamiltonian_signals=sig,
                             static_dissipators=lc["static_dissipators"], vectorized=True)
        yv = lc["rho0"].flatten(order="F")
        fn = lambda: qd.solve_lmde(m, [0.0, 0.2], yv, method="scipy_expm", max_dt=0.05)
        ctx.set_option("chebyshev", 1)
        r1, n1 = _count_products(ctx, fn)
        ctx.set_option("chebyshev", 0)
        r0, n0 = _count_products(ctx, fn)
        assert n1 == n0, (n1, n0)
        assert_close(r1.y[-1], r0.y[-1], 1e-14)
        ctx.set_option("expm_action", 0)
        rd = fn()
        assert_close(r1.y[-1], rd.y[-1], 1e-11)
    finally:
        ctx.set_option("chebyshev", 1)
        ctx.set_option("expm_action", 1)
        ctx.set_option("combine_sweep", 1)


@pytest.mark.parametrize("order", [1, 2])
def test_tiny_kernel_chebyshev_lab_frame(qd, order):
    """Small systems without a rotating frame have ||h G|| of several units (2 pi nu h per qubit): the
    persistent one-launch solver then runs the Chebyshev recurrence in registers (negative degree in its step
    table) for Magnus 1 and 2.  Against the Taylor series in the same kernel (chebyshev = 0), the dense expm
    route and the oracle; a sweep of 40 instances of a 3-qubit chain (n = 8)."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(3, 3, 1.0, 0.04)
    sweeps, params = [], []
    for b in range(40):
        amps, phases = W.sweep_parameters(b, 3)
        params.append((amps, phases))
        sweeps.append([qd.Signal(float(a), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"])
    rng = np.random.default_rng(12)
    y0 = crand(rng, 8)
    y0 /= np.linalg.norm(y0)
    fn = lambda: solver.solve(t_span=[0.0, 0.4], y0=y0, signals=sweeps, method="scipy_expm", max_dt=0.04,
                              magnus_order=order)
    out = {}
    try:
        for tag, opts in (("cheb", {"chebyshev": 1}), ("taylor", {"chebyshev": 0}), ("dense", {"expm_action": 0})):
            for k_, v_ in opts.items():
                ctx.set_option(k_, v_)
            out[tag] = np.stack([x.y[-1] for x in fn()])
            ctx.set_option("chebyshev", 1)
            ctx.set_option("expm_action", 1)
    finally:
        ctx.set_option("chebyshev", 1)
        ctx.set_option("expm_action", 1)
    assert_close(out["cheb"], out["taylor"], 1e-12)
    assert_close(out["cheb"], out["dense"], 1e-11)
    assert np.max(np.abs(np.linalg.norm(out["cheb"], axis=1) - 1.0)) < 1e-12
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], None)
    for b in (0, 39):
        amps, phases = params[b]
        coeff = lambda t: np.array([np.real(s_(t)) for s_ in sweeps[b]])
        _, yref = orc.solve_generator_model(a_d, a, d, basis, coeff, [0.0, 0.4], y0, "scipy_expm", 0.04, magnus_order=order)
        assert_close(out["cheb"][b], yref[-1], SOLVE_TOL)


def test_lab_frame_expm_routes_against_reference_golden(qd, golden):
    """The reference's own results (tests/golden/lab_frame.npz) for models WITHOUT a rotating frame, where the
    device takes its newest routes: Chebyshev expm action on the work-list kernels (8 qubits, lab frame), the
    persistent small-system kernel with the Chebyshev recurrence (3 qubits), the diagonal-frame twin, and a
    vectorised Lindbladian whose superoperators are assembled on the device -- Magnus orders 1 and 2."""
    from qiskit_dynamics_amd import workloads as W

    g = golden("lab_frame")

    def sigs(n_drives, carrier, b):
        amps, phases = W.sweep_parameters(b, n_drives)
        return [qd.Signal(float(a), float(nu), float(ph)) for a, nu, ph in zip(amps, carrier, phases)]

    cfg = W.schrodinger_config(n_qubits=8, n_drives=4, t_final=1.0, max_dt=0.05)
    sweeps = [sigs(4, cfg["carrier"], b) for b in range(3)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"])
    for mo in (1, 2):
        res = solver.solve(t_span=[0.0, 0.2], y0=g["q8_y0"], signals=sweeps, method="scipy_expm", max_dt=0.05,
                           magnus_order=mo)
        assert_close(np.stack([r.y[-1] for r in res]), g[f"q8_expm{mo}_y"], SOLVE_TOL)
        one = solver.solve(t_span=[0.0, 0.2], y0=g["q8_y0"], signals=sweeps[2], method="scipy_expm", max_dt=0.05,
                           magnus_order=mo)
        assert_close(one.y[-1], g[f"q8_expm{mo}_y"][2], SOLVE_TOL)
    solver_d = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                         rotating_frame=np.diag(cfg["h_d"]).real.copy())
    res = solver_d.solve(t_span=[0.0, 0.2], y0=g["q8_y0"], signals=sweeps, method="scipy_expm", max_dt=0.05)
    assert_close(np.stack([r.y[-1] for r in res]), g["q8_diag_expm1_y"], SOLVE_TOL)
    for slv, key in ((solver, "q8_rk4_y"), (solver_d, "q8_diag_rk4_y")):   # RK4 on the work-list kernels
        res = slv.solve(t_span=[0.0, 0.1], y0=g["q8_y0"], signals=sweeps, method="RK4", max_dt=0.002)
        assert_close(np.stack([r.y[-1] for r in res]), g[key], SOLVE_TOL)
        one = slv.solve(t_span=[0.0, 0.1], y0=g["q8_y0"], signals=sweeps[1], method="RK4", max_dt=0.002)
        assert_close(one.y[-1], g[key][1], SOLVE_TOL)

    cfg3 = W.schrodinger_config(n_qubits=3, n_drives=3, t_final=1.0, max_dt=0.04)
    solver3 = qd.Solver(static_hamiltonian=cfg3["h_d"], hamiltonian_operators=cfg3["ops"])
    sweeps3 = [sigs(3, cfg3["carrier"], b) for b in range(6)]
    for mo in (1, 2):
        res = solver3.solve(t_span=[0.0, 0.4], y0=g["q3_y0"], signals=sweeps3, method="scipy_expm", max_dt=0.04,
                            magnus_order=mo)
        assert_close(np.stack([r.y[-1] for r in res]), g[f"q3_expm{mo}_y"], SOLVE_TOL)

    lc = W.lindblad_config(n_qubits=3, n_drives=3, n_diss=3, gamma=1e-2, t_final=1.0, max_dt=0.05)
    m = qd.LindbladModel(static_hamiltonian=lc["h_d"], hamiltonian_operators=lc["ops"],
                         hamiltonian_signals=sigs(3, lc["carrier"], 0), static_dissipators=lc["static_dissipators"],
                         vectorized=True)
    r = qd.solve_lmde(m, [0.0, 0.2], g["l3_rho0"].flatten(order="F"), method="scipy_expm", max_dt=0.05)
    assert_close(r.y[-1], g["l3_expm1_y"], SOLVE_TOL)


def test_unvectorized_lindblad_sweep_against_reference_golden(qd, golden):
    """Row f2 in sweep form against the reference itself (tests/golden/lab_frame.npz, `nv_*`): Solver list
    mode with `vectorized=False`, per-instance Hamiltonian and dissipator signals and initial states -- the
    instances advance together in the batched launches of midyn_lindblad_rk4_solve."""
    from qiskit_dynamics_amd import workloads as W

    g = golden("lab_frame")
    lc = W.lindblad_config(n_qubits=3, n_drives=3, n_diss=3, gamma=1e-2, t_final=1.0, max_dt=0.05)
    sm = lc["static_dissipators"]
    solver = qd.Solver(static_hamiltonian=lc["h_d"], hamiltonian_operators=lc["ops"], static_dissipators=sm[:2],
                       dissipator_operators=sm[2:3], rotating_frame=np.diag(lc["h_d"]).real.copy(), vectorized=False)
    sweeps = []
    for b in range(3):
        amps, phases = W.sweep_parameters(b, 3)
        sweeps.append(([qd.Signal(float(a), float(nu), float(ph)) for a, nu, ph in zip(amps, lc["carrier"], phases)],
                       [qd.Signal(0.5 + 0.25 * b, 0.0)]))
    res = solver.solve(t_span=[0.0, 0.2], y0=list(g["nv_rho0"]), signals=sweeps, method="RK4", max_dt=0.01)
    assert_close(np.stack([r.y[-1] for r in res]), g["nv_rk4_y"], SOLVE_TOL)


def test_more_than_64_operators(qd):
    """A model with 70 drive operators + a static one (the MFMA contraction keeps a 64-entry segment table in one
    VGPR: longer lists run as chunked launches whose raw partial sums are reduced together).  Sweep of 12
    instances (MFMA route) and one trajectory (streaming route), RK4 and scipy_expm, against the oracle."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(70)
    n, k = 64, 70
    h_static = crand(rng, n, n)
    h_static = (h_static + h_static.conj().T) / 2
    a_ = crand(rng, k, n, n)
    h_ops = (a_ + np.swapaxes(a_.conj(), -1, -2)) / 2 * 0.2
    frame = crand(rng, n, n)
    frame = (frame + frame.conj().T) / 2
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    par = [(rng.uniform(0.2, 1.0, k), rng.uniform(0.0, 2.0, k), rng.uniform(-3, 3, k)) for _ in range(12)]
    sweeps = [[qd.Signal(float(a), float(f), float(p)) for a, f, p in zip(*pb)] for pb in par]
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)

    def coeff(b):
        amp, fr, ph = par[b]
        return lambda t: np.array([orc.signal_sum_value(np.array([amp[j] + 0j]), [fr[j]], [ph[j]], t) for j in range(k)])

    for method, kw in (("RK4", {"max_dt": 0.01}), ("scipy_expm", {"max_dt": 0.05})):
        res = solver.solve(t_span=[0.0, 0.1], y0=y0, signals=sweeps, method=method, **kw)
        one = solver.solve(t_span=[0.0, 0.1], y0=y0, signals=sweeps[3], method=method, **kw)
        for b in (0, 3, 11):
            _, yref = orc.solve_generator_model(a_d, a, d, basis, coeff(b), [0.0, 0.1], y0, method, kw["max_dt"])
            assert_close(res[b].y[-1], yref[-1], SOLVE_TOL)
        assert_close(one.y[-1], res[3].y[-1], 1e-11)


def test_lindblad_from_hamiltonian_golden(qd, golden):
    """`LindbladModel.from_hamiltonian` (lindblad_model.py:214-260) on Hamiltonian models without a frame, in a
    Hermitian frame and in a diagonal frame, vectorised and not, against the reference's RHS evaluations (the
    getters of a framed Hamiltonian model return the frame-subtracted static operator: reproduced as is)."""
    g = golden("lab_frame")
    for tag, frame in (("nofr", None), ("fr", g["fh_frame"]), ("diag", np.diag(g["fh_frame"]).real.copy())):
        hm = qd.HamiltonianModel(static_operator=g["fh_hs"], operators=g["fh_hops"],
                                 signals=[qd.Signal(0.5, 1.0), qd.Signal(0.3, 2.0, 0.4)], rotating_frame=frame)
        for vec in (False, True):
            lm = qd.LindbladModel.from_hamiltonian(hm, static_dissipators=g["fh_l"], vectorized=vec)
            yin = g["fh_rho"].flatten(order="F") if vec else g["fh_rho"]
            for i, t in enumerate((0.0, 0.3)):
                assert_close(lm.evaluate_rhs(t, yin), g[f"fh_{tag}_{'vec' if vec else 'mat'}_rhs"][i], EVAL_TOL)
        assert hm.in_frame_basis is False


def test_hermiticity_validation_on_device(qd):
    """Large operators with no / a diagonal frame are validated on the device (|| H - H^dagger ||_F from the
    uploaded -iH, tolerance 1e-10 as hamiltonian_model.py:98-104,196-222): same verdicts and the same errors as
    the host check, including the order static operator -> operators and the tolerance boundary."""
    from qiskit_dynamics_amd import _lib
    from qiskit_dynamics_amd.models import is_hermitian

    rng = np.random.default_rng(21)
    n = 1280
    a = crand(rng, n, n)
    h_d = (a + a.conj().T) / 2
    b = crand(rng, 2, n, n)
    h_ops = (b + np.swapaxes(b.conj(), -1, -2)) / 2
    frame = np.diag(h_d).real.copy()
    m = qd.HamiltonianModel(static_operator=h_d, operators=h_ops, rotating_frame=frame)
    d = m.stack.antiherm_defect()
    assert d.shape == (3,) and np.all(d < 1e-10)
    bad = h_d.copy()
    bad[7, 900] += 3e-10
    assert not is_hermitian(bad)
    with pytest.raises(_lib.DynamicsError, match="static_operator must be Hermitian"):
        qd.HamiltonianModel(static_operator=bad, operators=h_ops, rotating_frame=frame)
    ok = h_d.copy()
    ok[7, 900] += 5e-11          # inside the tolerance: accepted by both checks
    assert is_hermitian(ok)
    qd.HamiltonianModel(static_operator=ok, operators=h_ops, rotating_frame=None)
    bad_ops = h_ops.copy()
    bad_ops[1, 3, 4] += 1e-6
    with pytest.raises(_lib.DynamicsError, match="operators must be Hermitian"):
        qd.HamiltonianModel(static_operator=h_d, operators=bad_ops)
    with pytest.raises(_lib.DynamicsError, match="operators must be Hermitian"):
        qd.HamiltonianModel(operators=bad_ops, rotating_frame=frame)   # no user static operator, frame segment present
    qd.HamiltonianModel(static_operator=bad, operators=bad_ops, validate=False)
    # the defect itself against numpy
    m2 = qd.HamiltonianModel(static_operator=bad, operators=bad_ops, validate=False)
    ref = [np.linalg.norm(x.conj().T - x) for x in (bad, bad_ops[0], bad_ops[1])]
    assert np.allclose(m2.stack.antiherm_defect(), ref, rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("nq", [3, 8])
def test_chebyshev_action_backwards_in_time(qd, nq):
    """Negative step sizes through the Chebyshev action (persistent small-system kernel at 3 qubits, work-list
    kernels at 8): integrating forwards and then backwards over the same grid returns the initial state, and
    the backward solve agrees with the Taylor route (chebyshev = 0)."""
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(nq, 3, 1.0, 0.05)
    amps, phases = W.sweep_parameters(1, 3)
    sig = [qd.Signal(float(a), float(nu), float(ph)) for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"])
    rng = np.random.default_rng(nq)
    y0 = crand(rng, 2**nq)
    y0 /= np.linalg.norm(y0)
    for mo in (1, 2):
        fw = solver.solve(t_span=[0.0, 0.3], y0=y0, signals=sig, method="scipy_expm", max_dt=0.05, magnus_order=mo)
        bw = solver.solve(t_span=[0.3, 0.0], y0=fw.y[-1], signals=sig, method="scipy_expm", max_dt=0.05, magnus_order=mo)
        assert_close(bw.y[-1], y0, 1e-11)
        ctx.set_option("chebyshev", 0)
        try:
            bw0 = solver.solve(t_span=[0.3, 0.0], y0=fw.y[-1], signals=sig, method="scipy_expm", max_dt=0.05, magnus_order=mo)
        finally:
            ctx.set_option("chebyshev", 1)
        assert_close(bw.y[-1], bw0.y[-1], 1e-12)


@pytest.mark.parametrize("ftag", ["nofr", "fr"])
def test_interface_scenario_golden(qd, golden, ftag):
    """y0 into the frame basis / results out of it for Hamiltonian, Lindblad and vectorised Lindblad
    models (test_solver_functions_interface.py:164-395: X drive, Z static, Y dissipator, frame 1.2 X - 3.132 Y,
    y0 = (3.43, 1.31), evaluations at t = 231.232) and the Solver sanity scenario with a weak dissipator, vectorised
    and not (test_solver_classes.py:461-697) -- the device path against values captured from the reference."""
    g = golden("interface")
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    y = np.array([[0, -1j], [1j, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    frame = None if ftag == "nofr" else g["frame"]
    t = float(g["t"])
    hm = qd.HamiltonianModel(operators=[x], signals=[qd.Signal(1.0, 5.0)], static_operator=z, rotating_frame=frame)
    assert_close(hm(t), g[f"{ftag}_ham_eval"], 1e-10)        # |t F| ~ 800: phases to 1e-13 relative
    assert_close(hm(t, g["y0"]), g[f"{ftag}_ham_rhs"], 1e-10)
    for method, mo, dt in (("RK4", 1, 1e-3), ("scipy_expm", 1, 1e-2), ("scipy_expm", 2, 1e-2)):
        kw = {"magnus_order": mo} if method == "scipy_expm" else {}
        r = qd.solve_lmde(hm, t_span=[0.0, 1.1], y0=g["y0"], method=method, t_eval=[0.3, 1.1], max_dt=dt, **kw)
        assert_close(r.y, g[f"{ftag}_ham_{method}_{mo}_y"], SOLVE_TOL)
    rho0 = g["rho0"]
    for vec in (False, True):
        lm = qd.LindbladModel(hamiltonian_operators=[x], hamiltonian_signals=[qd.Signal(1.0, 5.0)], static_hamiltonian=z,
                              static_dissipators=[y], rotating_frame=frame, vectorized=vec)
        tag = "vec" if vec else "mat"
        yin = rho0.flatten(order="F") if vec else rho0
        assert_close(lm(t, yin), g[f"{ftag}_lind_{tag}_rhs"], 1e-10)
        if vec:
            assert_close(lm(t), g[f"{ftag}_lind_vec_eval"], 1e-10)
            r = qd.solve_lmde(lm, t_span=[0.0, 0.7], y0=yin, method="scipy_expm", max_dt=1e-2)
        else:
            r = qd.solve_lmde(lm, t_span=[0.0, 0.7], y0=yin, method="RK4", max_dt=1e-3)
        assert_close(r.y, g[f"{ftag}_lind_{tag}_y"], SOLVE_TOL)
    if ftag == "fr":
        for vec in (False, True):
            s = qd.Solver(hamiltonian_operators=[x / 2], static_hamiltonian=5 * z, rotating_frame=5 * z,
                          static_dissipators=[0.01 * x], vectorized=vec)
            rho = np.array([[0.0, 0.0], [0.0, 1.0]], dtype=complex)
            r = s.solve(t_span=[0.0, 1.0], y0=rho.flatten(order="F") if vec else rho,
                        signals=[qd.Signal(1.0, 5.0 / np.pi)], method="scipy_expm" if vec else "RK4",
                        max_dt=1e-2 if vec else 1e-3)
            assert_close(r.y, g[f"solver_{'vec' if vec else 'mat'}_y"], SOLVE_TOL)


def test_adaptive_scipy_methods_golden(qd, golden):
    """solve_lmde / solve_ode / Solver.solve with scipy's adaptive integrators calling the DEVICE right-hand side
    (reference: solvers/scipy_solve_ivp.py:31-84): the random framed 7 x 7 model with a DiscreteSignal
    (test_solver_functions.py:76-115) for RK45 / RK23 / DOP853 / BDF against values captured from the reference, a
    square (propagator) state, the real-embedded LSODA / Radau against DOP853, and both Lindblad forms."""
    g = golden("adaptive")
    sigs = [qd.Signal(0.5, 1.0, 0.3), qd.DiscreteSignal(dt=0.1, samples=g["r7_samples"], carrier_freq=1.0),
            qd.Signal(lambda t: 0.3 * np.cos(t) + 0 * 1j, 0.0)]
    hm = qd.HamiltonianModel(static_operator=g["r7_hstatic"], operators=g["r7_hops"], signals=sigs,
                             rotating_frame=g["r7_hframe"])
    y0 = g["r7_y0"]
    for method in ("RK45", "RK23", "DOP853", "BDF"):
        tol = 1e-10 if method in ("RK45", "DOP853") else 1e-7
        r = qd.solve_lmde(hm, [0.0, 0.5], y0, method=method, t_eval=[0.1, 0.3, 0.5], atol=tol, rtol=tol)
        assert r.route == "scipy_solve_ivp(device rhs)" and r.y.shape == (3, 7)
        assert_close(r.y, g[f"r7_{method}_y"], 5e-8 if tol < 1e-8 else 1e-5)
    r = qd.solve_ode(hm, [0.0, 0.3], np.eye(7, dtype=complex), method="DOP853", atol=1e-10, rtol=1e-10)
    assert_close(r.y[-1], g["r7_DOP853_unitary"][-1], 5e-8)     # (without t_eval every accepted step is returned)
    ref = g["r7_DOP853_y"]
    for method in ("LSODA", "Radau"):
        r = qd.solve_lmde(hm, [0.0, 0.5], y0, method=method, t_eval=[0.1, 0.3, 0.5], atol=1e-9, rtol=1e-9)
        assert_close(r.y, ref, 1e-6)
    with pytest.raises(qd.DynamicsError):
        qd.solve_lmde(hm, [0.0, 0.5], y0, method="DOP853", dense_output=True)
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    yy = np.array([[0, -1j], [1j, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    for vec in (False, True):
        lm = qd.LindbladModel(hamiltonian_operators=[x], hamiltonian_signals=[qd.Signal(1.0, 5.0)], static_hamiltonian=z,
                              static_dissipators=[yy], rotating_frame=g["frame"], vectorized=vec)
        yin = g["rho0"].flatten(order="F") if vec else g["rho0"]
        r = qd.solve_lmde(lm, [0.0, 0.7], yin, method="DOP853", atol=1e-10, rtol=1e-10)
        assert_close(r.y[-1], g[f"lind_{'vec' if vec else 'mat'}_DOP853_y"][-1], 5e-8)
    # Solver.solve list mode loops the instances for adaptive methods
    s = qd.Solver(static_hamiltonian=g["r7_hstatic"], hamiltonian_operators=g["r7_hops"], rotating_frame=g["r7_hframe"])
    res = s.solve(t_span=[0.0, 0.5], y0=[y0, y0], signals=sigs, method="DOP853", t_eval=[0.1, 0.3, 0.5], atol=1e-10,
                  rtol=1e-10)
    assert len(res) == 2
    assert_close(res[1].y, g["r7_DOP853_y"], 5e-8)


@pytest.mark.parametrize("kind", ["dyson", "magnus"])
def test_expansion_two_steps_per_padded_block(qd, golden, kind):
    """midyn_expansion_solve with n <= 32 (round 6, ctx option expansion_pack): the step matrices of steps i and i + nsteps / 2 share ONE
    padded 64 x 64 block (block diagonal: products and exponentials never mix the blocks), the batched expm / Udt products / tree run
    on half as many padded matrices, and an instance's product is (lower block) . (upper block).  Same results as one step per block
    (1e-12; the blocks' exponentials may take a different scaling than each alone) for an even number of steps -- packed, asserted
    through the counter -- and identical for an odd number (not packed); list mode with two instances included.  Reference:
    solvers/perturbative_solvers/perturbative_solver.py:172-219, array_polynomial.py:524-544."""
    g = golden("perturbative")
    ctx = qd.default_context()
    cls = qd.DysonSolver if kind == "dyson" else qd.MagnusSolver
    sol = cls(operators=g["t3_ops"], rotating_frame=g["t3_frame"], dt=0.02, carrier_freqs=[4.9, 0.0],
              chebyshev_orders=[1, 0], expansion_order=2, expansion_labels=[[0, 0, 1], [0, 1, 4]],
              include_imag=[True, False], integration_method="DOP853", atol=1e-12, rtol=1e-12)
    sig_a = qd.Signal(lambda t: 0.8 * np.exp(-((t - 1.0) ** 2) / 0.5) * np.exp(0.3j * t), carrier_freq=4.9, phase=0.2)
    sig_b = qd.Signal(lambda t: 0.4 * np.cos(0.7 * t) + 0j, carrier_freq=0.0)
    sig_c = qd.Signal(lambda t: 0.5 * np.exp(-((t - 0.7) ** 2) / 0.3) + 0j, carrier_freq=4.95, phase=-0.4)
    for n_steps, want in ((60, 2), (2, 2), (61, 1), (1, 1), (334, 2)):
        res = {}
        for pack in (1, 0):
            with ctx.options(expansion_pack=pack):
                r = sol.solve(t0=0.1, n_steps=n_steps, y0=[np.eye(3, dtype=complex), g["t3_y0"]], signals=[[sig_a, sig_b], [sig_c, sig_b]])
                per_block = int(ctx.counters("expansion_pack")["launches"])
            want_here = want if (pack and kind == "magnus") else 1        # (Dyson: no exponential per step, packing does not pay)
            assert per_block == want_here, (n_steps, pack, per_block)
            res[pack] = [x.y[-1] for x in r]
        for a, b in zip(res[1], res[0]):
            assert a.shape == b.shape
            if want == 1 or kind == "dyson":
                assert np.array_equal(a, b)
            else:
                assert_close(a, b, 1e-12)
    assert_close(res[1][0], sol.solve(t0=0.1, n_steps=334, y0=np.eye(3, dtype=complex), signals=[sig_a, sig_b]).y[-1], 1e-12)


@pytest.mark.parametrize("kind", ["dyson", "magnus"])
def test_expansion_coefficients_on_the_device_equal_the_monomial_table(qd, golden, kind):
    """midyn_expansion_solve_coeffs (round 6): the monomials c^I of the Chebyshev coefficients are formed on the device from the
    multiset labels (midyn_expansion_set_monomials) -- the first half of ArrayPolynomial.__call__ (perturbation/array_polynomial.py:
    524-528, :547-601) -- instead of arriving as a (nsteps x M) table.  Same bits as midyn_expansion_solve on the host's table
    (perturbative.compute_monomials associates the products the same way), for packed and unpacked blocks, several instances, a
    repeated shape (the offset tables of the last shape stay on the device) and changing shapes in between; the value of the table
    itself against the oracle's monomials."""
    from oracle import dynamics_oracle as orc

    g = golden("perturbative")
    ctx = qd.default_context()
    cls = qd.DysonSolver if kind == "dyson" else qd.MagnusSolver
    sol = cls(operators=g["t3_ops"], rotating_frame=g["t3_frame"], dt=0.02, carrier_freqs=[4.9, 0.0],
              chebyshev_orders=[1, 0], expansion_order=3, expansion_labels=[[0, 0, 1, 4]],
              include_imag=[True, False], integration_method="DOP853", atol=1e-10, rtol=1e-10)
    m = sol.model
    dev = m.device()
    sig_a = qd.Signal(lambda t: 0.8 * np.exp(-((t - 1.0) ** 2) / 0.5) * np.exp(0.3j * t), carrier_freq=4.9, phase=0.2)
    sig_b = qd.Signal(lambda t: 0.4 * np.cos(0.7 * t) + 0j, carrier_freq=0.0)
    sig_c = qd.Signal(lambda t: 0.5 * np.exp(-((t - 0.7) ** 2) / 0.3) + 0j, carrier_freq=4.95, phase=-0.4)
    rng = np.random.default_rng(5)
    y_shared = rng.standard_normal((3, 3)) + 1j * rng.standard_normal((3, 3))
    for n_steps in (60, 60, 61, 60, 2, 1, 334, 334):
        cs = np.stack([m.approximate_signals(sg, 0.1, n_steps) for sg in ([sig_a, sig_b], [sig_c, sig_b], [sig_a, sig_c])])
        mono = np.stack([m.monomial_table(c) for c in cs])
        labels = np.array([list(lab) + [-1] * (4 - len(lab)) for lab in m.monomial_labels])
        assert_close(mono[1].T, orc.monomials(labels, cs[1]), 1e-15)          # (left-to-right products there: last-bit differences)
        ys = np.stack([np.eye(3, dtype=complex), 1j * y_shared.T, y_shared])
        for shared in (False, True):
            y0 = y_shared if shared else ys
            a = dev.solve_coeffs(cs, y0, 3, shared)
            b = dev.solve(mono, y0, 3, shared)
            assert np.array_equal(a, b), (n_steps, shared, float(np.max(np.abs(a - b))))
            assert np.array_equal(a, dev.solve_coeffs(cs, y0, 3, shared))
    with pytest.raises(qd.DynamicsError):
        dev.solve_coeffs(cs[:, :-1], ys, 3, False)
    with pytest.raises(qd.DynamicsError):
        dev.set_monomials(2, m.monomial_labels)          # an index >= n_vars
    dev.set_monomials(cs.shape[1], m.monomial_labels)
