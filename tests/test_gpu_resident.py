"""rk4_resident_kernel (csrc/midyn_resident.h): one RK4 trajectory with the operators held in registers and the stage
input exchanged through a polled ring in device memory -- against the CPU oracle, against the launch-per-stage route
on the same inputs, across the shapes that decide its layout (entries per chunk 2..16, one to sixteen operand chunks,
padding rows, symmetry sectors with and without padding, frames of every kind), saved states, backwards integration,
step ranges run in pieces, and the shapes that must NOT take it.  `pytest -m gpu`.
"""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

SOLVE_TOL = 1e-9


@pytest.fixture(scope="module")
def qd():
    import qiskit_dynamics_amd as q

    q.default_context()
    return q


def crand(rng, *shape):
    return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)


def herm(rng, n, real=False):
    a = rng.uniform(-1, 1, (n, n)).astype(complex) if real else crand(rng, n, n)
    return (a + a.conj().T) / 2


def _solve_both(qd, solver, **kw):
    """(resident result, per-stage result, resident launches, per-stage launches of the resident kernel)"""
    ctx = qd.default_context()
    out, launches = {}, {}
    for flag in (1, 0):
        ctx.set_option("resident_rk4", flag)
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        try:
            out[flag] = solver.solve(**kw)
        finally:
            ctx.set_option("profile", 0)
            ctx.set_option("resident_rk4", 1)
        launches[flag] = ctx.counters("rk4_resident")["launches"]
    return out[1], out[0], launches[1], launches[0]


CASES = [
    # n, k, frame kind, real operators, expected to take the resident kernel
    (33, 1, "none", False, True),        # smallest size above the tiny kernel; n_pad = 64: one chunk, 31 padding rows
    (64, 3, "full", False, True),        # 8 planes: NE = 8
    (100, 2, "full", False, True),       # n_pad = 128, 6 planes -> NE = 8, two chunks
    (130, 1, "diag", False, True),       # n_pad = 192: three chunks, static + 1 operator complex: NE = 4
    (200, 7, "none", True, True),        # real Hamiltonians: G = -iH purely imaginary, 8 planes with the static part
    (256, 3, "diag", True, True),        # NE = 4, four chunks
    (300, 0, "full", False, True),       # static operator only (k = 0)
    (512, 1, "full", False, True),       # NE = 4, eight chunks, 8 rows per workgroup
    (700, 1, "none", True, True),        # n_pad = 704: eleven chunks, NE = 2
    (96, 9, "full", False, False),       # 20 planes: more than 16 entries per chunk -> per-stage route
    (640, 3, "full", False, False),      # 8 planes x 10 chunks = 80 doubles per lane > 64 -> per-stage route
]


@pytest.mark.parametrize("n,k,frame_kind,real,expect", CASES)
def test_resident_kernel_against_oracle_and_per_stage_route(qd, n, k, frame_kind, real, expect):
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(1000 + n + k)
    hs = herm(rng, n, real) * (2.0 / np.sqrt(n))
    ho = np.array([herm(rng, n, real) for _ in range(k)]) * (2.0 / np.sqrt(n)) if k else None
    frame = {"none": None, "full": herm(rng, n) * (2.0 / np.sqrt(n)), "diag": rng.normal(size=n)}[frame_kind]
    sigs = [qd.Signal(lambda t, a=0.3 + 0.1 * j: a * np.exp(-((t - 0.3) ** 2)) + 0j, 0.4 * j, 0.2 * j) for j in range(k)]

    def coeff(t):
        return np.array([np.real(s(t)) for s in sigs])

    solver = qd.Solver(static_hamiltonian=hs, hamiltonian_operators=ho, rotating_frame=frame)
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    kw = dict(t_span=[0.0, 0.5], y0=y0, method="RK4", max_dt=0.01, t_eval=[0.0, 0.13, 0.5])
    if k:
        kw["signals"] = sigs
    res, per_stage, l_res, l_off = _solve_both(qd, solver, **kw)
    assert (l_res > 0) == expect and l_off == 0
    assert_close(res.y, per_stage.y, 1e-13)
    a_d, a, d, basis = orc.hamiltonian_model_build(hs, ho, frame)
    t_ref, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff if k else None, [0.0, 0.5], y0, "RK4", 0.01,
                                             t_eval=[0.0, 0.13, 0.5])
    assert_close(res.t, t_ref, 0)
    assert_close(res.y, y_ref, SOLVE_TOL)


def test_resident_kernel_backwards_one_step_and_zero_length(qd):
    """Integration backwards in time, a span shorter than max_dt (one step), a zero-length span (one step of h = 0)
    and every step saved -- the step tables the kernel reads on the device (fixed_step_solvers.py:616-653)."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(7)
    n = 80
    hs, ho, frame = herm(rng, n) * 0.3, np.array([herm(rng, n), herm(rng, n)]) * 0.3, herm(rng, n) * 0.3
    sigs = [qd.Signal(0.5, 1.0, 0.1), qd.Signal(lambda t: 0.2 * np.sin(t) + 0j, 0.0)]

    def coeff(t):
        return np.array([0.5 * np.cos(2 * np.pi * t + 0.1), 0.2 * np.sin(t)])

    solver = qd.Solver(static_hamiltonian=hs, hamiltonian_operators=ho, rotating_frame=frame)
    y0 = crand(rng, n)
    a_d, a, d, basis = orc.hamiltonian_model_build(hs, ho, frame)
    for t_span, t_eval, max_dt in (([0.5, 0.0], [0.45, 0.2, 0.0], 0.02), ([0.0, 0.004], None, 0.1),
                                   ([0.3, 0.3], None, 0.1), ([0.0, 0.2], list(np.linspace(0.0, 0.2, 21)), 0.01)):
        res, per_stage, l_res, _ = _solve_both(qd, solver, t_span=t_span, y0=y0, signals=sigs, method="RK4",
                                                max_dt=max_dt, t_eval=t_eval)
        assert l_res > 0
        t_ref, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff, t_span, y0, "RK4", max_dt, t_eval=t_eval)
        assert_close(res.t, t_ref, 0)
        assert_close(res.y, y_ref, SOLVE_TOL)
        assert_close(res.y, per_stage.y, 1e-13)


def test_resident_kernel_symmetry_sectors(qd):
    """Frames with conserved quantities: the parity-conserving chain (two aligned sectors, every workgroup polls only
    the other sector's chunks) and sectors of 100 / 70 / 30 states embedded on block boundaries (padding rows inside the
    device stack, operators that couple two of the three sectors) -- against the oracle."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    rng = np.random.default_rng(11)
    cfg = W.schrodinger_config(n_qubits=8, n_drives=8, t_final=1.0, max_dt=0.01)
    sel_frame = np.zeros((200, 200), dtype=complex)
    bounds = [(0, 100), (100, 170), (170, 200)]
    for lo, hi in bounds:
        sel_frame[lo:hi, lo:hi] = herm(rng, hi - lo) * 0.3
    sel_ops = np.zeros((2, 200, 200), dtype=complex)
    cpl = crand(rng, 100, 70) * 0.05
    sel_ops[0, 0:100, 100:170] = cpl
    sel_ops[0, 100:170, 0:100] = cpl.conj().T
    sel_ops[1, 170:200, 170:200] = herm(rng, 30) * 0.3
    cases = [(cfg["h_d"], cfg["ops"], cfg["h_d"]), (sel_frame, sel_ops, sel_frame)]
    for h_static, h_ops, frame in cases:
        k, n = len(h_ops), frame.shape[0]
        sigs = [qd.Signal(lambda t, a=0.3 + 0.1 * j: a * np.cos(0.9 * t) + 0j, 0.4 * j, 0.2 * j) for j in range(k)]
        solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
        assert solver.model.rotating_frame.sector_labels is not None
        y0 = crand(rng, n)
        y0 /= np.linalg.norm(y0)
        res, per_stage, l_res, _ = _solve_both(qd, solver, t_span=[0.0, 0.3], y0=y0, signals=sigs, method="RK4",
                                                max_dt=0.01)
        assert l_res > 0
        assert_close(res.y, per_stage.y, 1e-13)
        a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
        _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt: np.array([np.real(s(tt)) for s in sigs]),
                                           [0.0, 0.3], y0, "RK4", 0.01)
        assert_close(res.y[-1], ref[-1], SOLVE_TOL)


def test_resident_kernel_step_ranges_run_in_pieces(qd):
    """midyn_rk4_plan_run over [0, 7), [7, 8), [8, 40): every launch rebuilds the ring from the state -- equal to one
    launch over [0, 40) to the last bits, bit-reproducible run to run, and equal to the per-stage route to rounding."""
    from qiskit_dynamics_amd import workloads as W
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    ctx = qd.default_context()
    rng = np.random.default_rng(3)
    n, k = 192, 3
    ops = np.array([-1j * herm(rng, n) for _ in range(k)]) * 0.2
    static = -1j * herm(rng, n) * 0.2
    fim = rng.normal(size=n)
    stack = qd.Stack(ctx, ops, static, fim)
    sched = FixedStepSchedule([0.0, 0.4], None, 0.01, _rk4_points)
    nsteps = len(sched.step_h)
    nr = int(sched.step_rows.max()) + 1
    table = rng.uniform(-1, 1, (1, nr, k))
    y0 = crand(rng, n, 1)
    outs = {}
    for tag, flag, pieces in (("one", 1, [(0, nsteps)]), ("pieces", 1, [(0, 7), (7, 8), (8, nsteps)]),
                              ("again", 1, [(0, nsteps)]), ("per_stage", 0, [(0, nsteps)])):
        ctx.set_option("resident_rk4", flag)
        try:
            p = qd.Rk4Plan(stack, sched.times[:nr], table, sched.step_rows, sched.step_h, y0, 1, True)
            ctx.reset_counters()
            ctx.set_option("profile", 1)
            for lo, hi in pieces:
                p.run(lo, hi)
            ctx.synchronize()
            ctx.set_option("profile", 0)
            assert ctx.counters("rk4_resident")["launches"] == (len(pieces) if flag else 0)
            outs[tag] = p.fetch()
            p.close()
        finally:
            ctx.set_option("profile", 0)
            ctx.set_option("resident_rk4", 1)
    assert np.array_equal(outs["one"], outs["again"])          # run to run: bit-identical
    assert_close(outs["one"], outs["pieces"], 1e-15)           # (the phase product at a launch boundary may contract differently)
    assert_close(outs["one"], outs["per_stage"], 1e-13)
    stack.close()


def test_resident_kernel_cfg2_shape(qd):
    """BASELINE configs[1]: the 10-qubit model (n = 1024, 8 drives, rotating_frame = H_d), one trajectory, 40 steps:
    resident kernel vs the oracle and vs the per-stage route."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    cfg = W.schrodinger_config()
    k = len(cfg["ops"])
    amps, phases = W.sweep_parameters(3, k)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    res, per_stage, l_res, _ = _solve_both(qd, solver, t_span=[0.0, 0.2], y0=cfg["y0"], signals=sigs, method="RK4",
                                            max_dt=0.005)
    assert l_res > 0
    assert_close(res.y, per_stage.y, 1e-13)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])

    def coeff(t):
        return W.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], 5.0)[0]

    _, ref = orc.solve_generator_model(a_d, a, d, basis, coeff, [0.0, 0.2], cfg["y0"], "RK4", 0.005)
    assert_close(res.y[-1], ref[-1], SOLVE_TOL)


def _cfg2_solver_and_oracle(qd, b=3):
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    cfg = W.schrodinger_config()
    k = len(cfg["ops"])
    amps, phases = W.sweep_parameters(b, k)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    model = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])   # plain eigh, no symmetry sectors

    def coeff(t):
        return W.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], 5.0)[0]

    return cfg, sigs, solver, model, coeff


def test_resident_kernel_cfg2_active_pulse_window(qd):
    """BASELINE configs[1], 40 RK4 steps in t = [2.4, 2.6] -- the top of the Gaussian envelopes, where the drive
    operators carry their full weight (on [0, 0.2] the envelopes are at 4 % of their peak and a relative operator
    error of 1e-5 would pass the 1e-9 gate): the resident kernel (one launch, counter asserted) against the oracle
    from a dense random state."""
    from oracle import dynamics_oracle as orc

    cfg, sigs, solver, (a_d, a, d, basis), coeff = _cfg2_solver_and_oracle(qd)
    rng = np.random.default_rng(24)
    y0 = crand(rng, 1024)
    y0 /= np.linalg.norm(y0)
    res, per_stage, l_res, l_off = _solve_both(qd, solver, t_span=[2.4, 2.6], y0=y0, signals=sigs, method="RK4",
                                               max_dt=0.005)
    assert (l_res, l_off) == (1, 0)
    assert_close(res.y, per_stage.y, 1e-13)
    _, ref = orc.solve_generator_model(a_d, a, d, basis, coeff, [2.4, 2.6], y0, "RK4", 0.005)
    assert_close(res.y[-1], ref[-1], SOLVE_TOL)
    assert np.linalg.norm(ref[-1] - ref[0]) > 2e-3      # the drives have acted inside the window (10x the quiet one)


def test_resident_kernel_cfg2_all_1000_steps_one_launch_vs_oracle(qd):
    """BASELINE configs[1] exactly as bench.py times it: the product `Solver` of the 10-qubit model (n = 1024, 8
    drives, rotating_frame = H_d), ONE trajectory from e_0, RK4 with max_dt = 0.005 over t_span = [0, 5] -> all 1000
    steps (4000 RHS evaluations) in ONE launch of rk4_resident_kernel (counter asserted, no fallback), against
    oracle.solve_generator_model (fixed_step_solvers.py:43-77,406-459; plain eigh) over the same 1000 steps --
    about half a minute of CPU."""
    from threadpoolctl import threadpool_limits

    from oracle import dynamics_oracle as orc

    ctx = qd.default_context()
    cfg, sigs, solver, (a_d, a, d, basis), coeff = _cfg2_solver_and_oracle(qd, b=0)
    assert (cfg["t_span"], cfg["max_dt"]) == ([0.0, 5.0], 0.005)
    gave_up_before = ctx.counters("resident_fallbacks")["launches"]
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        res = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sigs, method="RK4", max_dt=cfg["max_dt"])
    finally:
        ctx.set_option("profile", 0)
    assert ctx.counters("rk4_resident")["launches"] == 1, "the 1000 steps did not run as one resident launch"
    for cls in ("rhs_stream", "rhs_gemm", "rhs_blocks", "rhs_blocks_gemm"):
        assert ctx.counters(cls)["launches"] == 0, f"a per-stage kernel ({cls}) ran"
    assert ctx.counters("resident_fallbacks")["launches"] == gave_up_before
    # bookkeeping fields of the result (scipy's names; the reference passes them through for its scipy methods)
    assert res.nfev == 4000 and res.device == f"hip:{ctx.device}" and 0.0 < res.wall_s < 60.0
    assert res.route == "sequential"
    with threadpool_limits(limits=8):   # the oracle's matvecs run best on a few BLAS threads (see bench.py)
        _, ref = orc.solve_generator_model(a_d, a, d, basis, coeff, cfg["t_span"], cfg["y0"], "RK4", cfg["max_dt"])
    assert_close(res.y[-1], ref[-1], SOLVE_TOL)
    assert abs(np.linalg.norm(res.y[-1]) - 1.0) < 1e-8
    assert np.linalg.norm(res.y[-1] - res.y[0]) > 0.03  # the pulses have moved the state


def _chain_diag_frame(qd, nq):
    """nq-qubit chain in the DIAGONAL frame diag(H_d): the operators stay in the computational basis, a handful of
    non-zeros per row (block-sparse stack with work lists)."""
    from qiskit_dynamics_amd import workloads as W

    cfg = W.schrodinger_config(n_qubits=nq, n_drives=nq, t_final=1.0, max_dt=0.01)
    amps, phases = W.sweep_parameters(1, len(cfg["ops"]))
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    frame = np.diag(cfg["h_d"]).real.copy()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
    return cfg, sigs, frame, solver


@pytest.mark.parametrize("nq", [8, 9])
def test_lane_per_row_kernel_rk4_and_chebyshev_action(qd, nq):
    """ell_resident_kernel (one lane per row, operator elements in registers) on the chain in its diagonal frame,
    n = 256 / 512: RK4 (MODE 0) and the Magnus-1 Chebyshev action of scipy_expm (MODE 1, the whole solve in one
    launch) against the oracle and against the launch-per-product routes, with saved states."""
    from oracle import dynamics_oracle as orc

    cfg, sigs, frame, solver = _chain_diag_frame(qd, nq)
    info = solver.model.stack.block_info()
    assert info["state"] == 1 and info["block_density"] < 0.25
    rng = np.random.default_rng(nq)
    y0 = crand(rng, 2**nq)
    y0 /= np.linalg.norm(y0)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], frame)

    def coeff(t):
        return np.array([np.real(s(t)) for s in sigs])

    for method, kw in (("RK4", dict(max_dt=0.01)), ("scipy_expm", dict(max_dt=0.05, magnus_order=1))):
        res, per_launch, l_res, l_off = _solve_both(qd, solver, t_span=[0.0, 0.3], y0=y0, signals=sigs, method=method,
                                                    t_eval=[0.0, 0.1, 0.3], **kw)
        assert l_res == 1 and l_off == 0, (method, l_res, l_off)
        assert_close(res.y, per_launch.y, 1e-13)
        t_ref, y_ref = orc.solve_generator_model(a_d, a, d, basis, coeff, [0.0, 0.3], y0, method, kw["max_dt"],
                                                 t_eval=[0.0, 0.1, 0.3], magnus_order=kw.get("magnus_order", 1))
        assert_close(res.t, t_ref, 0)
        assert_close(res.y, y_ref, SOLVE_TOL)


def test_lane_per_row_kernel_vectorised_lindbladian(qd):
    """The cfg 4 shape at 4 qubits (N = 256 superoperators built on the device, static dissipators, no frame and the
    diagonal frame of H_d): scipy_expm Magnus 1 of one density matrix in one launch against the launch-per-term route
    (which test_gpu_parity pins to the reference goldens), trace preserved."""
    from qiskit_dynamics_amd import workloads as W

    cfg = W.lindblad_config(n_qubits=4, n_drives=4, n_diss=4, gamma=1e-2, t_final=1.0, max_dt=0.05)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, 0.1 * a)
            for a, nu in zip((0.9, 0.5, 0.7, 0.3), cfg["carrier"])]
    ctx = qd.default_context()
    for frame in (None, np.diag(cfg["h_d"]).real.copy()):
        m = qd.LindbladModel(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], hamiltonian_signals=sigs,
                             static_dissipators=cfg["static_dissipators"], rotating_frame=frame, vectorized=True)
        y0 = cfg["rho0"].flatten(order="F")
        out = {}
        for flag in (1, 0):
            ctx.set_option("resident_rk4", flag)
            ctx.reset_counters()
            ctx.set_option("profile", 1)
            try:
                out[flag] = qd.solve_lmde(m, [0.0, 1.0], y0, method="scipy_expm", max_dt=0.05)
            finally:
                ctx.set_option("profile", 0)
                ctx.set_option("resident_rk4", 1)
            assert ctx.counters("rk4_resident")["launches"] == (1 if flag else 0)
        assert_close(out[1].y, out[0].y, 1e-12)
        rho = out[1].y[-1].reshape(16, 16, order="F")
        assert abs(np.trace(rho) - 1.0) < 1e-12


@pytest.mark.parametrize("nq,nb,order", [(8, 5, 2), (9, 3, 1), (10, 5, 1), (10, 24, 2), (11, 3, 2), (11, 2, 1), (10, 1, 2)])
def test_one_workgroup_per_instance_sweep_kernel(qd, nq, nb, order):
    """ell_sweep_kernel: sweeps (and one Magnus-2 trajectory) of the chain in its diagonal frame (n = 256 / 512 on 256- and
    512-thread workgroups; n = 1024 / 2048: one
    and two rows per thread), scipy_expm with magnus_order 1 / 2, ONE launch for all instances and steps -- first,
    middle and last instance against the oracle, all instances against the launch-per-product route, saved states
    included."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(n_qubits=nq, n_drives=min(8, nq), t_final=1.0, max_dt=0.05)
    k = len(cfg["ops"])
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, k)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    frame = np.diag(cfg["h_d"]).real.copy()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
    rng = np.random.default_rng(nq + nb)
    y0 = crand(rng, 2**nq)
    y0 /= np.linalg.norm(y0)
    sig = sweeps if nb > 1 else sweeps[0]
    out, launches, split, cross = {}, {}, {}, {}
    # 1: one workgroup per instance; 2: several, operand vectors all-gathered (ell_sweep_split = 3); 3: the default -- two
    # workgroups per instance with the exchange behind the local slots (ell_sweep_duo_kernel, n >= 512); 0: per-launch route
    for flag in (1, 2, 3, 0):
        with ctx.options(ell_sweep=1 if flag else 0, ell_sweep_split=3 if flag == 2 else 0, ell_sweep_duo=1 if flag == 3 else 0,
                         profile=1):
            ctx.reset_counters()
            r = solver.solve(t_span=[0.0, 0.4], y0=y0, signals=sig, method="scipy_expm", max_dt=0.05, magnus_order=order,
                             t_eval=[0.0, 0.15, 0.4])
            launches[flag] = ctx.counters("rk4_resident")["launches"]
            split[flag] = ctx.counters("sweep_split")["launches"]
            cross[flag] = ctx.counters("sweep_cross")
        out[flag] = np.stack([x.y for x in r]) if nb > 1 else r.y[None]
    assert launches[1] == 1 and launches[2] == 1 and launches[3] == 1 and launches[0] == 0, launches
    assert split[1] == 1 and split[2] == (2 if nq >= 11 else 1), split      # n = 2048: two workgroups of 1024 rows each
    assert split[3] == (2 if nq >= 9 else 1), split                          # (n = 256: too small to share)
    if nq >= 9:      # the chain couples the halves of the vector through the strings that flip the top qubit (qubit 0 is the
        assert cross[3]["launches"] == 2, cross[3]      # most significant bit): its drive and its XX coupling -- 2 slots
    assert_close(out[1], out[0], 1e-12)
    assert_close(out[2], out[0], 1e-12)
    assert_close(out[3], out[0], 1e-12)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], frame)
    for b in (sorted({0, nb // 2, nb - 1}) if nq <= 10 else [nb - 1]):     # (a 2048 x 2048 expm per step on the host)
        _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt, b=b: np.array([np.real(s(tt)) for s in sweeps[b]]),
                                           [0.0, 0.4], y0, "scipy_expm", 0.05, t_eval=[0.0, 0.15, 0.4], magnus_order=order)
        assert_close(out[1][b], ref, SOLVE_TOL)


def test_sweep_kernel_vectorised_lindbladian_sweep(qd):
    """A sweep of vectorised Lindbladians (5 qubits: N = 1024 superoperators built on the device, static dissipators,
    complex static part: slots of both planes) through ell_sweep_kernel<1>: ONE launch for all instances, against the
    launch-per-product route (pinned to the reference goldens in test_gpu_parity) -- all instances, saved states,
    trace preserved."""
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.lindblad_config(n_qubits=5, n_drives=5, n_diss=4, gamma=2e-2, t_final=1.0, max_dt=0.05)
    nb = 6
    sweeps = [[qd.Signal(lambda t, a=0.3 + 0.1 * j + 0.05 * b: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, 0.1 * j)
               for j, nu in enumerate(cfg["carrier"])] for b in range(nb)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], vectorized=True)
    y0 = cfg["rho0"].flatten(order="F")
    out = {}
    for flag in (1, 0):
        ctx.set_option("ell_sweep", flag)
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        try:
            r = solver.solve(t_span=[0.0, 1.0], y0=y0, signals=sweeps, method="scipy_expm", max_dt=0.05,
                             t_eval=[0.0, 0.5, 1.0])
        finally:
            ctx.set_option("profile", 0)
            ctx.set_option("ell_sweep", 1)
        assert ctx.counters("rk4_resident")["launches"] == (1 if flag else 0)
        out[flag] = np.stack([x.y for x in r])
    assert_close(out[1], out[0], 1e-12)
    for b in range(nb):
        assert abs(np.trace(out[1][b, -1].reshape(32, 32, order="F")) - 1.0) < 1e-12


def _banded_model(rng, n, scale):
    """A sparse Hermitian model that is not a qubit chain: random diagonal, three random off-diagonals at distances
    1, 7 and 64, two drive operators on other off-diagonals -- about ten non-zeros per row, complex entries."""
    def band(dist, amp):
        m = np.zeros((n, n), dtype=complex)
        v = amp * crand(rng, n - dist)
        m[np.arange(n - dist), np.arange(dist, n)] = v
        return m + m.conj().T

    h_static = np.diag(rng.normal(size=n)).astype(complex) + band(1, scale) + band(7, 0.5 * scale) + band(64, 0.3 * scale)
    h_ops = np.array([band(3, 0.4 * scale), band(32, 0.2 * scale)])
    return h_static, h_ops


@pytest.mark.parametrize("scale,max_dt", [(0.3, 0.05), (6.0, 0.5), (40.0, 2.0)])
def test_one_launch_expm_action_routes_large_norms_backwards_and_own_initial_states(qd, scale, max_dt):
    """The one-launch expm-action kernels where the series changes shape: small norms (short Taylor series), norms where
    the Chebyshev series replaces it, and norms beyond the Bessel table (rho > 128: the series is repeated) -- a
    complex banded model with n = 1024, integrated BACKWARDS, every instance with its OWN initial state, saved states.
    ell_resident_kernel<1> (one trajectory, order 1) and ell_sweep_kernel<1 | 2> (sweeps, and one trajectory of order
    2) against the launch-per-product routes; the smallest case also against the oracle."""
    from oracle import dynamics_oracle as orc

    ctx = qd.default_context()
    rng = np.random.default_rng(int(scale * 10))
    n = 1024
    h_static, h_ops = _banded_model(rng, n, scale)
    frame = np.diag(h_static).real.copy()
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
    nb = 3
    sweeps = [[qd.Signal(lambda t, a=0.5 + 0.2 * b + 0.1 * j: a * np.cos(0.7 * t) + 0j, 0.3 * j, 0.1 * b) for j in range(2)]
              for b in range(nb)]
    y0s = []
    for b in range(nb):
        y = crand(rng, n)
        y0s.append(y / np.linalg.norm(y))
    t_span, t_eval = [1.0, 0.0], [1.0, 0.5, 0.0]
    for order in (1, 2):
        for batch in (True, False):
            sig = sweeps if batch else sweeps[1]
            y0 = y0s if batch else y0s[1]
            out, launches = {}, {}
            for flag in (1, 0):
                ctx.set_option("ell_sweep", flag)
                ctx.set_option("resident_rk4", flag)
                ctx.reset_counters()
                ctx.set_option("profile", 1)
                try:
                    r = solver.solve(t_span=t_span, y0=y0, signals=sig, method="scipy_expm", max_dt=max_dt,
                                     magnus_order=order, t_eval=t_eval)
                finally:
                    ctx.set_option("profile", 0)
                    ctx.set_option("ell_sweep", 1)
                    ctx.set_option("resident_rk4", 1)
                launches[flag] = ctx.counters("rk4_resident")["launches"]
                out[flag] = np.stack([x.y for x in r]) if batch else r.y[None]
            assert launches[1] == 1 and launches[0] == 0, (order, batch, launches)
            assert_close(out[1], out[0], 1e-11)
            assert np.max(np.abs(np.linalg.norm(out[1][:, -1], axis=1) - 1.0)) < 1e-9
            if scale < 1.0 and not batch and order == 2:
                a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
                _, ref = orc.solve_generator_model(a_d, a, d, basis,
                                                   lambda tt: np.array([np.real(s(tt)) for s in sweeps[1]]), t_span,
                                                   y0s[1], "scipy_expm", max_dt, t_eval=t_eval, magnus_order=order)
                assert_close(out[1][0], ref, SOLVE_TOL)


def test_sweep_kernel_four_workgroups_per_instance_at_full_size(qd):
    """The cfg 5 model itself (12 qubits, n = 4096, diagonal frame, Magnus-2), 3 instances: FOUR workgroups per instance
    (ell_sweep_split = 3: the operand vectors are all-gathered through the sentinel ring; the default for stacks without a
    packed element form) -- against one workgroup per instance (the default here: direct element form) and against the
    launch-per-product work-list route, saved states included; norms preserved."""
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
    k = len(cfg["ops"])
    nb = 3
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, k)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       rotating_frame=np.diag(cfg["h_d"]).real.copy(), validate=False)
    out, split = {}, {}
    for tag, sweep, parts, duo, flip in (("four", 1, 3, 0, 1), ("one", 1, 1, 0, 1), ("two", 1, 1, 1, 1), ("two_elements", 1, 1, 1, 0),
                                         ("per_launch", 0, 1, 1, 1)):
        with ctx.options(ell_sweep=sweep, ell_sweep_split=parts, ell_sweep_duo=duo, ell_sweep_flip=flip, profile=1):
            ctx.reset_counters()
            r = solver.solve(t_span=[0.0, 1.0], y0=cfg["y0"], signals=sweeps, method="scipy_expm", max_dt=0.25,
                             magnus_order=2, t_eval=[0.0, 0.5, 1.0])
            assert ctx.counters("rk4_resident")["launches"] == (1 if sweep else 0)
            split[tag] = (ctx.counters("sweep_split")["launches"], ctx.counters("sweep_split")["ms"])
            if tag == "two":
                cross = ctx.counters("sweep_cross")
        out[tag] = np.stack([x.y for x in r])
    assert split["four"][0] == 4 and split["one"] == (1, 3), split     # (workgroups per instance, element form: 3 = flip masks only)
    # the default for this stack: two workgroups per instance, no operator elements (form 3: one flip mask per slot), TWO of the
    # 19 slots (the drive of the top qubit and its XX coupling) reach into the partner's half; with elements: form 2
    assert split["two"] == (2, 3) and (cross["launches"], cross["ms"]) == (2, 19), (split, cross)
    assert split["two_elements"] == (2, 2), split
    assert_close(out["two_elements"], out["per_launch"], 1e-12)
    assert_close(out["four"], out["per_launch"], 1e-12)
    assert_close(out["one"], out["per_launch"], 1e-12)
    assert_close(out["two"], out["per_launch"], 1e-12)
    assert np.max(np.abs(np.linalg.norm(out["four"][:, -1], axis=1) - 1.0)) < 1e-12


def _sweep_forms_model(kind, nq):
    """(H_d, drive operators, carriers) of chains in their diagonal frame whose ELL stacks take the three element forms of
    ell_sweep_kernel: "direct" -- XX couplings and X drives (one magnitude and one sign per slot, every slot full);
    "packed" -- XX + YY couplings (flip-flop terms: rows |00>, |11> have no entry: unused slots) and Y drives (-iY is real
    with both signs in every slot); "general" -- the XX chain with drive amplitudes that differ from row to row."""
    from qiskit_dynamics_amd import workloads as W

    h_d, ops, nu = W.chain_hamiltonian(nq, min(8, nq))
    if kind == "direct":
        return h_d, ops, nu[:len(ops)]
    yy = np.array([[0.0, -1j], [1j, 0.0]])
    if kind == "packed":
        for q in range(nq - 1):
            h_d = h_d + 2 * np.pi * 0.002 * W.embed_pair(yy, q, yy, q + 1, nq)
        ops = np.stack([2 * np.pi * 0.02 * W.embed(yy, j, nq) / 2 for j in range(len(ops))])
        return h_d, ops, nu[:len(ops)]
    scale = 1.0 + 0.25 * np.cos(np.arange(2**nq))         # Hermitian: D X D with a real diagonal D
    ops = np.stack([scale[:, None] * o * scale[None, :] for o in ops])
    return h_d, ops, nu[:len(ops)]


@pytest.mark.parametrize("kind,form,nq,order", [("direct", 2, 10, 2), ("direct", 2, 9, 1), ("packed", 1, 10, 2), ("packed", 1, 11, 1),
                                                ("general", 0, 10, 2), ("general", 0, 9, 1)])
def test_sweep_kernel_element_forms(qd, kind, form, nq, order):
    """The three element forms of ell_sweep_kernel (general 12-byte elements / packed column | sign with a zero slot /
    direct LDS addresses) are chosen from the stack's values; each against the oracle, and the packed forms against the
    general form of the same stack (option ell_sweep_packed = 0) and against the launch-per-product route."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    h_d, ops, carrier = _sweep_forms_model(kind, nq)
    k, nb = len(ops), 5
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, k)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, carrier, phases)])
    frame = np.diag(h_d).real.copy()
    solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=frame)
    rng = np.random.default_rng(nq + order)
    y0 = crand(rng, 2**nq)
    y0 /= np.linalg.norm(y0)
    out, forms = {}, {}
    parts = {}
    for tag, opts in (("default", {}), ("with_elements", {"ell_sweep_flip": 0}), ("one_workgroup", {"ell_sweep_duo": 0}),
                      ("one_workgroup_with_elements", {"ell_sweep_duo": 0, "ell_sweep_flip": 0}),
                      ("general", {"ell_sweep_packed": 0}), ("per_launch", {"ell_sweep": 0})):
        with ctx.options(profile=1, **opts):
            ctx.reset_counters()
            r = solver.solve(t_span=[0.0, 0.4], y0=y0, signals=sweeps, method="scipy_expm", max_dt=0.05, magnus_order=order,
                             t_eval=[0.0, 0.15, 0.4])
            assert ctx.counters("rk4_resident")["launches"] == (0 if tag == "per_launch" else 1)
            forms[tag] = ctx.counters("sweep_split")["ms"]
            parts[tag] = ctx.counters("sweep_split")["launches"]
        out[tag] = np.stack([x.y for x in r])
    # the "direct" stacks (XX couplings, X drives) also have ONE flip mask per slot: their two-workgroup kernel reads no operator
    # elements at all (form 3, ell_flip_duo_kernel); option ell_sweep_flip = 0 keeps the 4-byte elements
    assert forms["default"] == (3 if form == 2 else form) and forms["with_elements"] == form, forms
    assert forms["one_workgroup"] == (3 if form == 2 else form) and forms["general"] == 0, forms
    # the packed forms of a small sweep share an instance between two workgroups by default
    assert parts["default"] == (2 if form else 1) and parts["with_elements"] == parts["default"] and parts["one_workgroup"] == 1, parts
    assert forms["one_workgroup_with_elements"] == form and parts["one_workgroup_with_elements"] == 1, (forms, parts)
    assert_close(out["default"], out["with_elements"], 1e-12)
    assert_close(out["default"], out["one_workgroup"], 1e-12)
    assert_close(out["default"], out["one_workgroup_with_elements"], 1e-12)
    assert_close(out["default"], out["general"], 1e-12)
    assert_close(out["default"], out["per_launch"], 1e-12)
    a_d, a, d, basis = orc.hamiltonian_model_build(h_d, ops, frame)
    for b in (0, nb - 1):
        _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt, b=b: np.array([np.real(s(tt)) for s in sweeps[b]]),
                                           [0.0, 0.4], y0, "scipy_expm", 0.05, t_eval=[0.0, 0.15, 0.4], magnus_order=order)
        assert_close(out["default"][b], ref, SOLVE_TOL)


@pytest.mark.parametrize("nq,order,framed,seed", [(9, 2, True, 1), (10, 2, True, 2), (10, 1, True, 3), (11, 2, False, 4),
                                                  (12, 2, True, 5), (11, 1, False, 6), (9, 1, True, 7)])
def test_flip_kernel_random_x_strings(qd, nq, order, framed, seed):
    """Operators that are sums of RANDOM Pauli-X strings (flip masks anywhere: several crossing slots, partners in other waves,
    other threads, other row indices; several sets of crossing operands): every slot of the stack has one signed magnitude and
    one flip mask, so the two-workgroup kernel without operator elements takes the sweep (ell_flip_duo_kernel, element form 3)
    -- against the same kernel family with elements (ell_sweep_flip = 0), one workgroup per instance, the launch-per-product
    route and the oracle (reference: fixed_step_solvers.py:345-363)."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    rng = np.random.default_rng(100 + seed)
    n = 2**nq
    rows = np.arange(n)

    def xstring(mask):
        m = np.zeros((n, n))
        m[rows, rows ^ mask] = 1.0
        return m

    top = n >> 1
    k = 3
    used = set()

    def fresh_mask(force_top):
        while True:
            mask = int(rng.integers(1, n))
            mask = (mask | top) if force_top else mask
            if mask not in used:
                used.add(mask)
                return mask

    ops = []
    for j in range(k):
        op = np.zeros((n, n))
        for t in range(3):
            op += (0.3 + 0.2 * t + 0.05 * j) * (-1.0) ** t * xstring(fresh_mask(force_top=(t == 0)))
        ops.append(2 * np.pi * 0.02 * op)
    ops = np.stack(ops).astype(complex)
    diag = 2 * np.pi * rng.uniform(0.0, 0.5, n) if framed else np.zeros(n)    # (the diagonal goes into the frame)
    h_d = np.diag(diag).astype(complex)
    for t in range(2):
        h_d += 2 * np.pi * 0.004 * (t + 1) * xstring(fresh_mask(force_top=(t == 1)))
    carrier = rng.uniform(0.2, 0.5, k)
    nb = 5
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, k)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, carrier, phases)])
    frame = diag.copy() if framed else None
    solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=frame)
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    out, forms, parts = {}, {}, {}
    for tag, opts in (("default", {}), ("with_elements", {"ell_sweep_flip": 0}), ("one_workgroup", {"ell_sweep_duo": 0}),
                      ("one_workgroup_with_elements", {"ell_sweep_duo": 0, "ell_sweep_flip": 0}), ("per_launch", {"ell_sweep": 0})):
        with ctx.options(profile=1, **opts):
            ctx.reset_counters()
            r = solver.solve(t_span=[0.0, 0.4], y0=y0, signals=sweeps, method="scipy_expm", max_dt=0.05, magnus_order=order,
                             t_eval=[0.0, 0.15, 0.4])
            assert ctx.counters("rk4_resident")["launches"] == (0 if tag == "per_launch" else 1)
            forms[tag] = ctx.counters("sweep_split")["ms"]
            parts[tag] = ctx.counters("sweep_split")["launches"]
            if tag == "default":
                cross = ctx.counters("sweep_cross")
        out[tag] = np.stack([x.y for x in r])
    assert (forms["default"], parts["default"]) == (3, 2), (forms, parts)
    assert (forms["with_elements"], parts["with_elements"]) == (2, 2), (forms, parts)
    assert (forms["one_workgroup"], parts["one_workgroup"]) == (3, 1), (forms, parts)          # ell_sweep_kernel<.., 3>
    assert (forms["one_workgroup_with_elements"], parts["one_workgroup_with_elements"]) == (2, 1), (forms, parts)
    assert_close(out["default"], out["one_workgroup_with_elements"], 1e-12)
    assert cross["launches"] >= k + 1, cross        # every operator and the static part have a string that flips the top qubit
    assert_close(out["default"], out["with_elements"], 1e-12)
    assert_close(out["default"], out["one_workgroup"], 1e-12)
    assert_close(out["default"], out["per_launch"], 1e-12)
    if nq <= 10:        # (dense scipy expm per step on the CPU: seconds at n = 1024, minutes beyond)
        a_d, a, d, basis = orc.hamiltonian_model_build(h_d, ops, frame)
        for b in (0, nb - 1):
            _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt, b=b: np.array([np.real(s(tt)) for s in sweeps[b]]),
                                               [0.0, 0.4], y0, "scipy_expm", 0.05, t_eval=[0.0, 0.15, 0.4], magnus_order=order)
            assert_close(out["default"][b], ref, SOLVE_TOL)
    # the RK4 sweep of the same model: ell_sweep_rk4_kernel<.., 3> (no operator elements) against the kernel with elements, the
    # launch-per-stage route and (small sizes) the oracle
    rk, rk_forms = {}, {}
    for tag, opts in (("default", {}), ("with_elements", {"ell_sweep_flip": 0}), ("per_launch", {"ell_sweep": 0})):
        with ctx.options(profile=1, **opts):
            ctx.reset_counters()
            r = solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sweeps, method="RK4", max_dt=0.01)
            rk_forms[tag] = (ctx.counters("rk4_resident")["launches"], ctx.counters("sweep_split")["ms"])
        rk[tag] = np.stack([x.y for x in r])
    assert rk_forms["default"] == (1, 3) and rk_forms["with_elements"] == (1, 2) and rk_forms["per_launch"][0] == 0, rk_forms
    assert_close(rk["default"], rk["with_elements"], 1e-12)
    assert_close(rk["default"], rk["per_launch"], 1e-12)
    if nq <= 10:
        _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt: np.array([np.real(s(tt)) for s in sweeps[0]]),
                                           [0.0, 0.2], y0, "RK4", 0.01)
        assert_close(rk["default"][0], ref, SOLVE_TOL)


def test_resident_kernel_rows_without_any_operator(qd):
    """Half of the rows carry no operator element at all (n = 256, operators supported on the first 128 states only;
    the workgroups of rows 128..255 poll nothing and every chunk slot of theirs is unused): those components must
    simply keep their frame phase -- resident kernel against the oracle and the per-stage route, RK4."""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(21)
    n, m = 256, 128
    d = rng.normal(size=n)
    h_static = np.diag(d).astype(complex)
    h_static[:m, :m] += herm(rng, m) * 0.2
    h_ops = np.zeros((2, n, n), dtype=complex)
    h_ops[0, :m, :m] = herm(rng, m) * 0.2
    h_ops[1, :m, :m] = herm(rng, m) * 0.2
    sigs = [qd.Signal(0.7, 0.3, 0.1), qd.Signal(lambda t: 0.4 * np.cos(t) + 0j, 0.2)]
    solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=d)
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    res, per_stage, l_res, _ = _solve_both(qd, solver, t_span=[0.0, 0.4], y0=y0, signals=sigs, method="RK4", max_dt=0.01)
    assert l_res > 0
    assert np.all(np.isfinite(res.y))
    assert_close(res.y, per_stage.y, 1e-13)
    a_d, a, dd, basis = orc.hamiltonian_model_build(h_static, h_ops, d)
    _, ref = orc.solve_generator_model(a_d, a, dd, basis, lambda tt: np.array([np.real(s(tt)) for s in sigs]),
                                       [0.0, 0.4], y0, "RK4", 0.01)
    assert_close(res.y[-1], ref[-1], SOLVE_TOL)


@pytest.mark.parametrize("nq,nb", [(8, 6), (9, 3), (10, 17), (11, 2)])
def test_sweep_kernel_rk4(qd, nq, nb):
    """ell_sweep_rk4_kernel: RK4 sweeps of the chain in its diagonal frame (n = 256 .. 2048), ONE launch for all instances
    and steps, per-instance initial states, saved states -- against the oracle (first and last instance) and against
    the launch-per-stage work-list route (all instances)."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(n_qubits=nq, n_drives=min(8, nq), t_final=1.0, max_dt=0.01)
    k = len(cfg["ops"])
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, k)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, cfg["carrier"], phases)])
    frame = np.diag(cfg["h_d"]).real.copy()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
    rng = np.random.default_rng(nq * 7 + nb)
    y0s = []
    for b in range(nb):
        y = crand(rng, 2**nq)
        y0s.append(y / np.linalg.norm(y))
    out, launches = {}, {}
    for flag in (1, 0):
        ctx.set_option("ell_sweep", flag)
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        try:
            r = solver.solve(t_span=[0.0, 0.3], y0=y0s, signals=sweeps, method="RK4", max_dt=0.01, t_eval=[0.0, 0.11, 0.3])
        finally:
            ctx.set_option("profile", 0)
            ctx.set_option("ell_sweep", 1)
        launches[flag] = ctx.counters("rk4_resident")["launches"]
        out[flag] = np.stack([x.y for x in r])
    assert launches[1] == 1 and launches[0] == 0, launches
    assert_close(out[1], out[0], 1e-13)
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], frame)
    for b in ((0, nb - 1) if nq <= 10 else (nb - 1,)):        # (the host builds a dense 2048 x 2048 generator per evaluation)
        _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt, b=b: np.array([np.real(s(tt)) for s in sweeps[b]]),
                                           [0.0, 0.3], y0s[b], "RK4", 0.01, t_eval=[0.0, 0.11, 0.3])
        assert_close(out[1][b], ref, SOLVE_TOL)


def test_small_sweep_of_a_dense_model_runs_as_single_trajectories(qd):
    """Two instances of a dense 200-dimensional model in a full frame, RK4: midyn_rk4_solve runs them one after the
    other on the register-resident kernel (two launches) instead of the batched stage -- against the oracle and the
    batched route, per-instance initial states, saved states.  (From 1024 rows on the same holds up to 8 instances.)"""
    from oracle import dynamics_oracle as orc

    rng = np.random.default_rng(77)
    n, k, nb = 200, 2, 2
    hs = herm(rng, n) * 0.2
    ho = np.array([herm(rng, n) for _ in range(k)]) * 0.2
    frame = herm(rng, n) * 0.2
    sweeps = [[qd.Signal(lambda t, a=0.3 + 0.1 * j + 0.05 * b: a * np.cos(0.8 * t) + 0j, 0.4 * j, 0.2 * b) for j in range(k)]
              for b in range(nb)]
    y0s = []
    for b in range(nb):
        y = crand(rng, n)
        y0s.append(y / np.linalg.norm(y))
    solver = qd.Solver(static_hamiltonian=hs, hamiltonian_operators=ho, rotating_frame=frame)
    kw = dict(t_span=[0.0, 0.4], y0=y0s, signals=sweeps, method="RK4", max_dt=0.01, t_eval=[0.0, 0.17, 0.4])
    res, batched, l_res, l_off = _solve_both(qd, solver, **kw)
    assert l_res == nb and l_off == 0
    a_d, a, d, basis = orc.hamiltonian_model_build(hs, ho, frame)
    for b in range(nb):
        assert_close(res[b].y, batched[b].y, 1e-13)
        t_ref, y_ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt, b=b: np.array([np.real(s(tt)) for s in sweeps[b]]),
                                                 [0.0, 0.4], y0s[b], "RK4", 0.01, t_eval=[0.0, 0.17, 0.4])
        assert_close(res[b].t, t_ref, 0)
        assert_close(res[b].y, y_ref, SOLVE_TOL)


@pytest.mark.parametrize("route", ["dense_rk4", "ell_rk4", "ell_expm", "sweep_split", "sweep_duo"])
def test_one_launch_kernels_fall_back_when_a_wait_gives_up(qd, route):
    """The one-launch kernels wait for each other's data inside the launch.  With the spin limit forced to zero every wait
    whose first poll finds a word missing gives up (what a launch that is not co-resident after all, or a GPU shared with
    another process, would cause): the solve must not fail -- the step range is re-run on the per-launch route (counter
    resident_fallbacks) and the answer is the oracle's."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    rng = np.random.default_rng(11)
    if route == "dense_rk4":            # rk4_resident_kernel: dense 200 x 200 operators
        n, k = 200, 3
        h_d, ops = herm(rng, n), np.stack([herm(rng, n) for _ in range(k)])
        frame, method, kw = h_d, "RK4", dict(max_dt=0.01)
        carrier = np.array([1.0, 2.0, 3.0])
    else:                               # ELL stacks: the 9-qubit chain in its diagonal frame
        cfg = W.schrodinger_config(n_qubits={"sweep_split": 12, "sweep_duo": 10}.get(route, 9), n_drives=8, t_final=1.0, max_dt=0.05)
        h_d, ops, carrier = cfg["h_d"], cfg["ops"], cfg["carrier"]
        n, k = h_d.shape[0], len(ops)
        frame = np.diag(h_d).real.copy()
        method, kw = ("RK4", dict(max_dt=0.01)) if route == "ell_rk4" else ("scipy_expm", dict(max_dt=0.05))
        if route in ("sweep_split", "sweep_duo"):
            kw["magnus_order"] = 2
    nb = 3 if route in ("sweep_split", "sweep_duo") else 1
    sweeps = []
    for b in range(nb):
        amps, phases = W.sweep_parameters(b, k)
        sweeps.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph)
                       for a, nu, ph in zip(amps, carrier, phases)])
    solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=frame, validate=False)
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    sig = sweeps if nb > 1 else sweeps[0]
    t_span = [0.0, 0.3]
    runs, fallbacks = {}, {}
    for tag, limit in (("healthy", -1), ("gives_up", 0)):
        ctx.set_option("resident_spin_limit", limit)
        if route == "sweep_split":
            ctx.set_option("ell_sweep_split", 3)
        ctx.set_option("ell_sweep_duo", 1 if route == "sweep_duo" else 0)
        before = ctx.counters("resident_fallbacks")["launches"]
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        try:
            r = solver.solve(t_span=t_span, y0=y0, signals=sig, method=method, **kw)
        finally:
            ctx.set_option("profile", 0)
            ctx.set_option("resident_spin_limit", -1)
            ctx.set_option("ell_sweep_split", 1)
            ctx.set_option("ell_sweep_duo", 1)
        assert ctx.counters("rk4_resident")["launches"] >= 1, "the one-launch route was not taken"
        if route == "sweep_duo" and tag == "healthy":
            assert ctx.counters("sweep_split")["launches"] == 2
        fallbacks[tag] = ctx.counters("resident_fallbacks")["launches"] - before
        runs[tag] = np.stack([x.y[-1] for x in r]) if nb > 1 else r.y[-1][None]
    assert fallbacks["healthy"] == 0 and fallbacks["gives_up"] >= 1, fallbacks
    assert_close(runs["gives_up"], runs["healthy"], 1e-12)
    if route != "sweep_split":          # (n = 4096: the healthy route is pinned against the oracle elsewhere)
        a_d, a, d, basis = orc.hamiltonian_model_build(h_d, ops, frame)
        _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt: np.array([np.real(s(tt)) for s in sweeps[0]]),
                                           t_span, y0, method, kw["max_dt"], magnus_order=kw.get("magnus_order", 1))
        assert_close(runs["gives_up"][0], ref[-1], SOLVE_TOL)


@pytest.mark.parametrize("nq,count,order", [(10, 64, 2), (12, 16, 2), (10, 300, 1), (5, 7, 2)])
def test_expm_plan_equals_expm_solve(qd, nq, count, order):
    """midyn_expm_plan_* (csrc/midyn_sweep_plan.inc): a plan made ONCE for a model + time grid and run with several coefficient
    tables gives, for each table, exactly what midyn_expm_solve gives for it (the solve IS create + run + fetch + destroy on
    the sweep route) -- bit for bit, with the saved states written by the kernel straight into the pinned result block and with
    the device block + copy (option expm_direct_out = 0), for the two-workgroup flip kernel (10 and 12 qubits, 64 / 16
    instances), the one-workgroup kernel (300 instances) and a small dense model the sweep kernels do not take (5 qubits: the
    plan runs midyn_expm_solve itself).  Reference: solvers/fixed_step_solvers.py:80-108 called once per parameter set
    (solver_classes.py:556-590)."""
    from qiskit_dynamics_amd import workloads as W
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points

    ctx = qd.default_context()
    cfg = W.schrodinger_config(n_qubits=nq, n_drives=min(8, nq), t_final=2.0, max_dt=0.25)
    k = min(8, nq)
    fr = RotatingFrame(np.diag(cfg["h_d"]).real.copy())
    stack = qd.Stack(ctx, -1j * cfg["ops"], -1j * cfg["h_d"] - np.diag(fr.frame_diag), fr.frame_diag_imag)
    sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(order))
    rng = np.random.default_rng(nq * 100 + count)
    n = 2**nq
    y0 = (rng.normal(size=n) + 1j * rng.normal(size=n)).reshape(-1, 1)
    y0 /= np.linalg.norm(y0)

    def table(scale, first):
        amps = np.array([W.sweep_parameters(first + b, k)[0] for b in range(count)]) * scale
        phs = np.array([W.sweep_parameters(first + b, k)[1] for b in range(count)])
        return W.gaussian_coefficient_table(sched.times, amps, phs, cfg["carrier"][:k], 2.0)

    tables = [table(1.0, 0), table(2.5, 1000), table(0.1, 5)]       # (different norm bounds -> different series per run)
    want = [stack.expm_solve(sched.times, t, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, order, y0, count, True)
            for t in tables]
    assert np.max(np.abs(want[0] - want[1])) > 1e-3
    ctx.reset_counters()
    with ctx.options(profile=1):
        stack.expm_solve(sched.times, tables[0], sched.step_rows, sched.step_h, sched.step_save, sched.n_save, order, y0, count, True)
        on_sweep = ctx.counters("rk4_resident")["launches"] == 1
        split = ctx.counters("sweep_split")
    assert on_sweep == (nq >= 10)
    if nq >= 10:
        assert int(split["launches"]) == (2 if count <= 128 else 1), split
    plan = qd.ExpmPlan(stack, sched.times, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, order, y0, count, True)
    for direct in (1, 0):
        with ctx.options(expm_direct_out=direct):
            for t, w in zip(tables, want):
                got = plan.solve(t)
                assert got.shape == w.shape and np.array_equal(got, w), (direct, np.max(np.abs(got - w)))
            # run without fetch, then run again: the first launch is waited for, the second is what fetch returns
            plan.run(tables[1])
            plan.run(tables[2])
            assert np.array_equal(plan.fetch(), want[2])
            assert np.array_equal(stack.expm_solve(sched.times, tables[1], sched.step_rows, sched.step_h, sched.step_save,
                                                   sched.n_save, order, y0, count, True), want[1])
    with pytest.raises(qd.DynamicsError):
        plan.run(tables[0][:, :-1])
    with pytest.raises(qd.DynamicsError):
        plan.fetch()
    plan.close()
    assert np.max(np.abs(np.linalg.norm(want[1][:, -1, :, 0], axis=1) - 1.0)) < 1e-11


def test_one_shot_expm_solve_keeps_its_plan_in_the_stack(qd):
    """ctx option expm_plan_cache (round 6): the one-shot midyn_expm_solve on the sweep route keeps its plan -- frame phases, step
    tables, y0, exchange slots, result block -- in the stack, and the next call with the same shapes, time grid, step tables and
    initial states only uploads its coefficient table (the scan of solver_classes.py:556-590 through the one-shot entry point).
    Asserted: hit / miss by the counter; bit-equal results with the cache on and off for changing tables; a changed y0, time grid,
    instance count or ANY option change retires the plan; pageable and pinned result blocks in turn (the plan forgets the blocks
    it was shown: the one-shot caller promised nothing about them); a stack destroyed with a plan inside."""
    from qiskit_dynamics_amd import _lib
    from qiskit_dynamics_amd import workloads as W
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points

    ctx = qd.default_context()
    nq, count, order = 10, 64, 2
    cfg = W.schrodinger_config(n_qubits=nq, n_drives=8, t_final=2.0, max_dt=0.25)
    fr = RotatingFrame(np.diag(cfg["h_d"]).real.copy())
    stack = qd.Stack(ctx, -1j * cfg["ops"], -1j * cfg["h_d"] - np.diag(fr.frame_diag), fr.frame_diag_imag)
    sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(order))
    sched2 = FixedStepSchedule([0.0, 1.5], None, cfg["max_dt"], _magnus_points(order))
    rng = np.random.default_rng(77)
    n = 2**nq
    y0 = (rng.normal(size=n) + 1j * rng.normal(size=n)).reshape(-1, 1)
    y0 /= np.linalg.norm(y0)
    y0b = y0[::-1].copy()

    def table(sc, scale, first, cnt=count):
        amps = np.array([W.sweep_parameters(first + b, 8)[0] for b in range(cnt)]) * scale
        phs = np.array([W.sweep_parameters(first + b, 8)[1] for b in range(cnt)])
        return W.gaussian_coefficient_table(sc.times, amps, phs, cfg["carrier"][:8], 2.0)

    def solve(sc, t, y, cnt=count):
        r = stack.expm_solve(sc.times, t, sc.step_rows, sc.step_h, sc.step_save, sc.n_save, order, y, cnt, True)
        return r, int(ctx.counters("expm_plan_cache")["launches"])

    tabs = [table(sched, 1.0, 0), table(sched, 2.5, 1000), table(sched, 0.1, 5)]
    with ctx.options(expm_plan_cache=0):
        want = [solve(sched, t, y0) for t in tabs]
        assert all(h == 0 for _, h in want)
        want_b = solve(sched, tabs[1], y0b)[0]
        want_2 = solve(sched2, table(sched2, 1.0, 0), y0)[0]
        want_half = solve(sched, tabs[0][: count // 2], y0, count // 2)[0]
    want = [w for w, _ in want]
    # (the option change above retired whatever was cached: the first call makes a plan, the following ones find it)
    hits = []
    for i in (0, 1, 2, 1, 0):
        got, h = solve(sched, tabs[i], y0)
        assert np.array_equal(got, want[i]), i
        hits.append(h)
    assert hits == [0, 1, 1, 1, 1], hits
    got, h = solve(sched, tabs[1], y0b)                       # other initial states: a new plan
    assert h == 0 and np.array_equal(got, want_b)
    got, h = solve(sched, tabs[1], y0b)
    assert h == 1 and np.array_equal(got, want_b)
    got, h = solve(sched2, table(sched2, 1.0, 0), y0)         # other time grid
    assert h == 0 and np.array_equal(got, want_2)
    got, h = solve(sched, tabs[0][: count // 2], y0, count // 2)      # other instance count
    assert h == 0 and np.array_equal(got, want_half)
    got, h = solve(sched, tabs[0][: count // 2], y0, count // 2)
    assert h == 1 and np.array_equal(got, want_half)
    with ctx.options(expm_direct_out=0):                     # an option change retires the plan; so does changing it back
        got, h = solve(sched, tabs[0][: count // 2], y0, count // 2)
        assert h == 0 and np.array_equal(got, want_half)
    got, h = solve(sched, tabs[0][: count // 2], y0, count // 2)
    assert h == 0 and np.array_equal(got, want_half)
    # result blocks: the binding hands out page-locked blocks of the library (>= 1 MB); a pageable block in between takes the
    # device block + copy -- same bits, and nothing remembered about either
    lib, args = ctx.lib, stack._solve_args(sched.times, tabs[2], sched.step_rows, sched.step_h, sched.step_save, y0, count)
    times, r, tab, rows, hs, save, nsteps, y = args
    for kind in ("pageable", "pinned", "pageable"):
        out = np.empty((count, sched.n_save, n, 1), dtype=complex) if kind == "pageable" else _lib.result_array((count, sched.n_save, n, 1))
        assert _lib.is_pinned(out) == (kind == "pinned")
        ctx.check(lib.midyn_expm_solve(stack.handle, count, 1, r, _lib._ptr(times), _lib._ptr(tab), nsteps, _lib._ptr(rows), _lib._ptr(hs),
                                       _lib._ptr(save), sched.n_save, order, _lib._ptr(y), 1, _lib._ptr(out)))
        assert np.array_equal(stack._rows_out(out, 2), want[2]), kind
    stack.close()                                             # (with the plan inside)
    stack2 = qd.Stack(ctx, -1j * cfg["ops"], -1j * cfg["h_d"] - np.diag(fr.frame_diag), fr.frame_diag_imag)
    got = stack2.expm_solve(sched.times, tabs[0], sched.step_rows, sched.step_h, sched.step_save, sched.n_save, order, y0, count, True)
    assert int(ctx.counters("expm_plan_cache")["launches"]) == 0 and np.array_equal(got, want[0])
