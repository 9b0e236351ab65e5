"""Edge cases of the hot path on the device against the CPU oracle: degenerate sizes (n = 1, k = 0, k = 1, one column
given as a matrix), zero-length and single-step intervals, t_eval at the endpoints, backwards with t_eval, sizes
around the tile boundaries, a sweep far larger than one launch tile, and the largest operator the tests can afford
(n = 8192) for a single evaluation.  `pytest -m gpu`.
"""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

SOLVE_TOL = 1e-9
EVAL_TOL = 1e-12


@pytest.fixture(scope="module")
def qd():
    import qiskit_dynamics_amd as q

    q.default_context()
    return q


def crand(rng, *shape):
    return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)


def herm(rng, n):
    a = crand(rng, n, n)
    return (a + a.conj().T) / 2


def _oracle_solve(orc, h_static, h_ops, frame, coeff, t_span, y0, method, max_dt, t_eval=None, magnus_order=1):
    a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
    return orc.solve_generator_model(a_d, a, d, basis, coeff, t_span, y0, method, max_dt, t_eval=t_eval,
                                     magnus_order=magnus_order)


@pytest.mark.parametrize("method", ["RK4", "scipy_expm"])
def test_one_dimensional_and_static_only_models(qd, method):
    """n = 1 (a scalar ODE), a model with a static operator only (k = 0, no signals), operators without a static
    part, a single operator -- each with and without a frame."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(11)
    kw = dict(method=method, max_dt=0.01 if method == "RK4" else 0.05)
    # n = 1
    h0, h1 = np.array([[0.7]], dtype=complex), np.array([[[0.3]]], dtype=complex)
    for frame in (None, np.array([0.4])):
        s = qd.Solver(static_hamiltonian=h0, hamiltonian_operators=h1, rotating_frame=frame)
        r = s.solve(t_span=[0.0, 1.0], y0=np.array([1.0 + 0.5j]), signals=[qd.Signal(1.0, 0.5)], **kw)
        _, ref = _oracle_solve(orc, h0, h1, frame, lambda t: np.array([np.cos(2 * np.pi * 0.5 * t)]), [0.0, 1.0],
                               np.array([1.0 + 0.5j]), method, kw["max_dt"])
        assert r.y.shape == (2, 1)
        assert_close(r.y, ref, SOLVE_TOL)
    # static operator only
    n = 6
    hs = herm(rng, n)
    y0 = crand(rng, n)
    for frame in (None, herm(rng, n), rng.normal(size=n)):
        s = qd.Solver(static_hamiltonian=hs, rotating_frame=frame)
        r = s.solve(t_span=[0.0, 0.5], y0=y0, **kw)
        _, ref = _oracle_solve(orc, hs, None, frame, None, [0.0, 0.5], y0, method, kw["max_dt"])
        assert_close(r.y, ref, SOLVE_TOL)
    # operators only, k = 1
    ho = np.array([herm(rng, n)])
    for frame in (None, herm(rng, n)):
        s = qd.Solver(hamiltonian_operators=ho, rotating_frame=frame)
        r = s.solve(t_span=[0.0, 0.5], y0=y0, signals=[qd.Signal(0.8, 1.5, 0.2)], **kw)
        _, ref = _oracle_solve(orc, None, ho, frame, lambda t: np.array([0.8 * np.cos(2 * np.pi * 1.5 * t + 0.2)]),
                               [0.0, 0.5], y0, method, kw["max_dt"])
        assert_close(r.y, ref, SOLVE_TOL)


def test_state_shapes_are_preserved(qd):
    """(n,), (n, 1), (n, m) and square states come back with a leading time axis and their own shape
    (fixed_step_solvers.py:447-457); a one-element list returns a list."""
    rng = np.random.default_rng(2)
    n = 5
    s = qd.Solver(static_hamiltonian=herm(rng, n), hamiltonian_operators=np.array([herm(rng, n)]))
    sig = [qd.Signal(0.3, 1.0)]
    for shape in ((n,), (n, 1), (n, 3), (n, n)):
        y0 = crand(rng, *shape)
        r = s.solve(t_span=[0.0, 0.2], y0=y0, signals=sig, method="RK4", max_dt=0.01)
        assert r.y.shape == (2,) + shape and r.t.shape == (2,)
        assert_close(r.y[0], y0, 0)
        r = s.solve(t_span=[0.0, 0.2], y0=y0, signals=sig, method="RK4", max_dt=0.01, t_eval=[0.05, 0.1, 0.2])
        assert r.y.shape == (3,) + shape
    r = s.solve(t_span=[0.0, 0.2], y0=[crand(rng, n)], signals=sig, method="RK4", max_dt=0.01)
    assert isinstance(r, list) and len(r) == 1
    r = s.solve(t_span=[[0.0, 0.2]], y0=crand(rng, n), signals=sig, method="RK4", max_dt=0.01)
    assert isinstance(r, list) and len(r) == 1


def test_degenerate_time_grids(qd):
    """Zero-length interval (the step rule gives one step of h = 0), an interval shorter than max_dt (one step),
    t_eval equal to the endpoints, t_eval with repeated points, backwards with t_eval -- times and states against the
    oracle (solver_utils.py:46-119, fixed_step_solvers.py:616-653)."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(3)
    n = 4
    hs, ho = herm(rng, n), np.array([herm(rng, n), herm(rng, n)])
    frame = herm(rng, n)
    y0 = crand(rng, n)
    sigs = [qd.Signal(0.5, 1.0, 0.1), qd.Signal(lambda t: 0.2 * np.sin(t) + 0j, 0.0)]

    def coeff(t):
        return np.array([0.5 * np.cos(2 * np.pi * t + 0.1), 0.2 * np.sin(t)])

    s = qd.Solver(static_hamiltonian=hs, hamiltonian_operators=ho, rotating_frame=frame)
    cases = [
        dict(t_span=[0.3, 0.3], t_eval=None, max_dt=0.1),
        dict(t_span=[0.0, 0.004], t_eval=None, max_dt=0.1),
        dict(t_span=[0.0, 0.5], t_eval=[0.0, 0.5], max_dt=0.02),
        dict(t_span=[0.0, 0.5], t_eval=[0.1, 0.1, 0.4], max_dt=0.02),
        dict(t_span=[0.5, 0.0], t_eval=[0.45, 0.2, 0.0], max_dt=0.02),
        dict(t_span=[0.0, 0.5], t_eval=[0.25], max_dt=10.0),
    ]
    for method, mo in (("RK4", 1), ("scipy_expm", 1), ("scipy_expm", 3)):
        for c in cases:
            kw = {"magnus_order": mo} if method == "scipy_expm" else {}
            r = s.solve(t_span=c["t_span"], y0=y0, signals=sigs, method=method, max_dt=c["max_dt"], t_eval=c["t_eval"],
                        **kw)
            t_ref, y_ref = _oracle_solve(orc, hs, ho, frame, coeff, c["t_span"], y0, method, c["max_dt"],
                                         t_eval=c["t_eval"], magnus_order=mo)
            assert_close(r.t, t_ref, 0)
            assert_close(r.y, y_ref, SOLVE_TOL)


@pytest.mark.parametrize("n", [63, 64, 65, 127, 129, 257])
def test_sizes_around_tile_boundaries_single_evaluations(qd, n):
    """evaluate / evaluate_rhs for dimensions just below, at and above the 64 / 128 padding steps, 1, 7 and 70
    columns, in a full frame, against the oracle."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(n)
    k = 3
    hs, ho = herm(rng, n), np.array([herm(rng, n) for _ in range(k)])
    frame = herm(rng, n)
    sigs = [qd.Signal(0.4 + 0.1 * j, 0.3 * j, 0.2 * j) for j in range(k)]
    m = qd.HamiltonianModel(static_operator=hs, operators=ho, signals=sigs, rotating_frame=frame)
    a_d, a, d, basis = orc.hamiltonian_model_build(hs, ho, frame)
    t = 0.37
    c = np.array([(0.4 + 0.1 * j) * np.cos(2 * np.pi * 0.3 * j * t + 0.2 * j) for j in range(k)])
    assert_close(m.evaluate(t), orc.generator_evaluate(a_d, a, c, d, basis, t, False), 1e-11)
    for cols in (None, 7, 70):
        y = crand(rng, n) if cols is None else crand(rng, n, cols)
        assert_close(m.evaluate_rhs(t, y), orc.generator_rhs(a_d, a, c, d, basis, t, y, False), 1e-11)


def test_sweep_larger_than_any_tile_small_system(qd):
    """5000 instances of a 3-qubit model (far more columns than one launch tile; persistent small-system kernel and
    the batched route agree) -- first, middle and last instance against the oracle."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(n_qubits=3, n_drives=3, t_final=1.0, max_dt=0.01)
    nb = 5000
    params = [W.sweep_parameters(b, 3) for b in range(nb)]
    sweeps = [[qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
               for a, nu, ph in zip(p[0], cfg["carrier"], p[1])] for p in params]
    s = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    out = {}
    for tiny in (1, 0):
        ctx.set_option("tiny_rk4", tiny)
        try:
            res = s.solve(t_span=[0.0, 0.3], y0=cfg["y0"], signals=sweeps, method="RK4", max_dt=0.01)
        finally:
            ctx.set_option("tiny_rk4", 1)
        out[tiny] = np.stack([r.y[-1] for r in res])
    assert_close(out[1], out[0], 1e-12)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])
    for b in (0, 2500, 4999):
        amps, phases = params[b]

        def coeff(t, amps=amps, phases=phases):
            return W.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], 1.0)[0]

        _, ref = orc.solve_generator_model(a_d, a, d, basis, coeff, [0.0, 0.3], cfg["y0"], "RK4", 0.01)
        assert_close(out[1][b], ref[-1], SOLVE_TOL)


def test_largest_affordable_operator_single_evaluation(qd):
    """n = 8192 (13 qubits), one operator + static operator, diagonal frame: one evaluate_rhs on the streaming
    kernel and one with 32 columns on the MFMA route against NumPy on the host (2 x 1 GiB of operators)."""
    rng = np.random.default_rng(8192)
    n = 8192
    d = rng.normal(size=n)

    def big_herm():
        a = rng.normal(size=(n, n)).astype(np.float32).astype(complex)
        a += 1j * np.triu(rng.normal(size=(n, n)).astype(np.float32), 1)
        a = np.triu(a)
        return a + np.triu(a, 1).conj().T

    hs, h1 = big_herm(), big_herm()
    m = qd.HamiltonianModel(static_operator=hs, operators=h1[None], signals=[qd.Signal(0.6, 0.25, 0.3)],
                            rotating_frame=d, validate=False)
    t = 0.4
    c = 0.6 * np.cos(2 * np.pi * 0.25 * t + 0.3)
    e = np.exp(-1j * d * t)                               # F = -i diag(d): exp(t F) = exp(-i d t)
    g = -1j * (hs - np.diag(d) + c * h1)
    for cols in (None, 32):
        y = crand(rng, n) if cols is None else crand(rng, n, cols)
        ey = (e * y.T).T
        ref = (e.conj() * (g @ ey).T).T                   # conj(e) o (C (e o y))
        out = m.evaluate_rhs(t, y)
        assert_close(out, ref, 1e-11)


def test_measurement_only_option_is_refused_without_the_debug_environment(qd, monkeypatch):
    """`resident_exchange_only` makes rk4_resident_kernel skip its arithmetic (bench.py's store -> poll floor): an
    ordinary caller of a shared context cannot switch it on -- the library refuses it unless MIDYN_DEBUG_OPTIONS=1 is
    in the process environment; switching it OFF is always allowed."""
    ctx = qd.default_context()
    monkeypatch.delenv("MIDYN_DEBUG_OPTIONS", raising=False)
    with pytest.raises(qd.DynamicsError, match="MIDYN_DEBUG_OPTIONS"):
        ctx.set_option("resident_exchange_only", 1)
    ctx.set_option("resident_exchange_only", 0)
    monkeypatch.setenv("MIDYN_DEBUG_OPTIONS", "1")
    try:
        ctx.set_option("resident_exchange_only", 1)
    finally:
        ctx.set_option("resident_exchange_only", 0)


def test_options_are_readable_and_scoped_changes_restore_the_previous_value(qd):
    """midyn_ctx_get_option / Context.options: a caller that changes an option for one solve puts back the value it FOUND --
    `combine_occupancy` defaults to 2, `combine_min_cols` to 256: restoring "1" (round 4's test helper) would have changed
    them for every later user of the shared context."""
    ctx = qd.default_context()
    defaults = {name: ctx.get_option(name) for name in ("combine", "combine_occupancy", "combine_min_cols", "skip_zero_blocks",
                                                        "complex_3m", "resident_spin_limit", "ell_sweep_split", "profile")}
    assert defaults["combine"] == 1 and defaults["combine_occupancy"] == 2 and defaults["profile"] == 0
    assert defaults["resident_spin_limit"] == 1 << 21
    with ctx.options(combine=0, combine_occupancy=1, combine_min_cols=512, complex_3m=2, profile=1):
        assert ctx.get_option("combine") == 0 and ctx.get_option("combine_occupancy") == 1
        assert ctx.get_option("combine_min_cols") == 512 and ctx.get_option("complex_3m") == 2 and ctx.get_option("profile") == 1
        with ctx.options(combine=2):
            assert ctx.get_option("combine") == 2
        assert ctx.get_option("combine") == 0
    assert {name: ctx.get_option(name) for name in defaults} == defaults
    with pytest.raises(qd.DynamicsError, match="unknown option"):
        ctx.get_option("no_such_option")
    with pytest.raises(qd.DynamicsError, match="unknown option"):
        with ctx.options(no_such_option=1):
            pass


@pytest.mark.gpu
def test_results_in_pinned_arrays_survive_later_solves(qd):
    """Large results come back in pinned host blocks that are recycled when collected: an array a caller still holds must keep
    its values through later solves, and two live results never share memory."""
    import gc

    from qiskit_dynamics_amd import _lib as L
    from qiskit_dynamics_amd import workloads

    ctx = qd.default_context()
    cfg = workloads.schrodinger_config(n_qubits=8, n_drives=4)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=np.diag(cfg["h_d"]).real.copy())
    stack = solver.model.stack
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points
    sched = FixedStepSchedule([0.0, 0.1], None, 0.01, _rk4_points)
    batch, k = 300, stack.k
    rng = np.random.default_rng(5)
    y0 = cfg["y0"].reshape(-1, 1)

    def solve(seed):
        table = np.random.default_rng(seed).uniform(-1, 1, (batch, len(sched.times), k))
        return stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, y0, batch, True)

    a = solve(1)
    assert a.nbytes >= L._PINNED_MIN_BYTES
    if stack.slot is None:          # (a stack with a row permutation returns a re-ordered copy of the pinned array)
        assert a.base is not None, "the result is not in a pinned block"
    keep = a.copy()
    b = solve(2)
    assert not np.shares_memory(a, b)
    del b
    gc.collect()
    c = solve(3)                    # (may reuse b's block, never a's)
    assert not np.shares_memory(a, c)
    assert np.array_equal(a, keep)
    assert np.array_equal(solve(1), keep)


@pytest.mark.parametrize("n,magn", [(200, 9.0), (128, 2.0)])
def test_chebyshev_series_ends_at_the_unit_roundoff(qd, n, magn):
    """Option cheb_tail (csrc/midyn_action.inc, cheb_plan): the Chebyshev series of the expm action ends where the dropped terms
    of a step sum to less than 2^-53 (default) instead of keeping every Bessel coefficient >= 1e-18 (cheb_tail = 0, rounds 2-5).
    The default must take FEWER products, agree with the old rule to a few ulps per step, and both must agree with the dense
    expm route and with the oracle's scipy.linalg.expm steps (solvers/fixed_step_solvers.py:80-108) at the solve tolerance."""
    from oracle import dynamics_oracle as orc

    ctx = qd.default_context()
    rng = np.random.default_rng(7 * n)

    def crand_(*shape):
        return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)

    evals = rng.uniform(-magn, magn, n) * 20.0
    q_, _ = np.linalg.qr(crand_(n, n))
    h_static = (q_ * evals) @ q_.conj().T
    h_static = (h_static + h_static.conj().T) / 2
    a_ = crand_(n, n)
    h_ops = np.array([(a_ + a_.conj().T) / 2 * 0.3])
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops)
    sig = [qd.Signal(lambda t: 0.5 * np.cos(t) + 0j, 1.0, 0.2)]
    y0 = crand_(n)
    y0 /= np.linalg.norm(y0)
    kw = dict(t_span=[0.0, 0.2], y0=y0, signals=sig, method="scipy_expm", max_dt=0.05)
    res, products = {}, {}
    for tag, opts in (("roundoff", dict(cheb_tail=1, chebyshev=2)), ("every_coefficient", dict(cheb_tail=0, chebyshev=2)),
                      ("dense", dict(expm_action=0))):
        with ctx.options(profile=1, **opts):
            ctx.reset_counters()
            res[tag] = np.asarray(solver.solve(**kw).y)
            products[tag] = ctx.counters("rhs_stream")["launches"]
    assert ctx.get_option("cheb_tail") == 1                       # (the default, restored)
    assert 0 < products["roundoff"] < products["every_coefficient"], products
    assert products["every_coefficient"] - products["roundoff"] <= 4 * 4, products      # a term or two per step, four steps
    assert np.abs(res["roundoff"] - res["every_coefficient"]).max() < 2e-15
    assert np.abs(res["roundoff"] - res["dense"]).max() < 1e-11
    g = -1j * h_static
    ops = -1j * h_ops

    def gen(t):
        return g + np.real(0.5 * np.cos(t) * np.exp(1j * (2 * np.pi * 1.0 * t + 0.2))) * ops[0]

    _, yo = orc.expm_solve(gen, [0.0, 0.2], y0, 0.05)
    assert np.abs(res["roundoff"] - yo).max() < 1e-9
    assert abs(np.linalg.norm(res["roundoff"][-1]) - 1.0) < 1e-13


def test_exchange_protocol_option_is_bit_identical(qd):
    """ctx option exchange_protocol (profiles/r06_exchange_protocol.md): 1 selects the conforming hand-off forms -- RELEASE publishes /
    ACQUIRE polls in rk4_resident_kernel (one trajectory, RK4) and ell_resident_kernel (vectorised Lindblad, scipy_expm), sc1 payload
    stores + agent-scope flags also between two workgroups that share an L2 in ell_flip_duo_kernel (the path partners on different
    XCDs always take; here forced, with 13 and 16 instances: instance-major and part-major workgroup order).  Same kernels, same
    arithmetic: the results are bit-identical, the one-launch kernels ran (counters) and no wait gave up."""
    from qiskit_dynamics_amd import workloads as W
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points

    ctx = qd.default_context()
    assert ctx.get_option("exchange_protocol") == 0
    gave_up = ctx.counters("resident_fallbacks")["launches"]
    # one trajectory, n = 1024: rk4_resident_kernel
    cfg = W.schrodinger_config(t_final=0.3, max_dt=0.005)
    amps, phases = W.sweep_parameters(1, len(cfg["ops"]))
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.15) ** 2) / 0.02), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    # vectorised Lindbladian, N = 4096: ell_resident_kernel
    cl = W.lindblad_config(t_final=0.5)
    lm = qd.LindbladModel(static_hamiltonian=cl["h_d"], hamiltonian_operators=cl["ops"],
                          hamiltonian_signals=[qd.Signal(1.0, nu) for nu in cl["carrier"]],
                          static_dissipators=cl["static_dissipators"], vectorized=True)
    rho0 = cl["rho0"].flatten(order="F")
    # flip-structured stack, n = 1024, Magnus 2: ell_flip_duo_kernel
    c5 = W.schrodinger_config(n_qubits=10, n_drives=8, t_final=1.0, max_dt=0.25)
    fr = RotatingFrame(np.diag(c5["h_d"]).real.copy())
    stack = qd.Stack(ctx, -1j * c5["ops"], -1j * c5["h_d"] - np.diag(fr.frame_diag), fr.frame_diag_imag)
    sched = FixedStepSchedule(c5["t_span"], None, c5["max_dt"], _magnus_points(2))
    y5 = c5["y0"].reshape(-1, 1)
    got = {}
    for proto in (0, 1):
        with ctx.options(exchange_protocol=proto, profile=1):
            ctx.reset_counters()
            a = solver.solve(t_span=[0.0, 0.3], y0=cfg["y0"], signals=sigs, method="RK4", max_dt=0.005).y[-1]
            assert ctx.counters("rk4_resident")["launches"] >= 1
            ctx.reset_counters()
            b = qd.solve_lmde(lm, [0.0, 0.5], rho0, method="scipy_expm", max_dt=0.05).y[-1]
            assert ctx.counters("rk4_resident")["launches"] >= 1
            c = []
            for count in (13, 16):
                amps_ = np.array([W.sweep_parameters(i, 8)[0] for i in range(count)])
                phs_ = np.array([W.sweep_parameters(i, 8)[1] for i in range(count)])
                table = W.gaussian_coefficient_table(sched.times, amps_, phs_, c5["carrier"], 1.0)
                ctx.reset_counters()
                c.append(stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y5, count, True))
                split = ctx.counters("sweep_split")
                assert (int(split["launches"]), int(split["ms"])) == (2, 3), split        # two workgroups per instance, flip masks
        got[proto] = (a, b, c[0], c[1])
    for x0, x1 in zip(got[0], got[1]):
        assert np.array_equal(x0, x1)
    assert abs(np.linalg.norm(got[0][0]) - 1.0) < 1e-10 and np.max(np.abs(np.linalg.norm(got[0][3][:, -1, :, 0], axis=1) - 1.0)) < 1e-12
    assert ctx.counters("resident_fallbacks")["launches"] == gave_up
    assert ctx.get_option("exchange_protocol") == 0
