"""Parity of the PRODUCTION routes at BASELINE shapes, directly against the CPU oracle (not against another
device route):

  * cfg 5 shard (n = 4096, k = 8, diagonal frame, Magnus-2, 128 instances, all 20 steps, dense random y0): the
    SPARSE MFMA work-list contraction (`zgemm_seg_kernel<..., SPARSE>`, counter "rhs_blocks_gemm");
  * cfg 4 sweep (N = 4096 vectorised Lindbladian, 64 instances): the same route through the Chebyshev action;
  * expm at n = 1024 / 2048 against scipy.linalg.expm (the function the reference calls,
    solvers/fixed_step_solvers.py:22,104);
  * 3M vs 4M complex products on badly scaled operators (entries spanning 1e-8 ... 1).

All of them need a real MI355X (`pytest -m gpu`).
"""
import numpy as np
import pytest
import scipy.linalg
import scipy.sparse as sp

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


def _gauss_signals(qd, cfg, b, k, t_mid):
    from qiskit_dynamics_amd import workloads

    amps, phases = workloads.sweep_parameters(b, k)
    return [qd.Signal(lambda t, a=a: a * np.exp(-((t - t_mid) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]


def _profiled(ctx, fn):
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        return fn()
    finally:
        ctx.set_option("profile", 0)


@pytest.mark.parametrize("route", ["default", "two_workgroups_with_elements", "one_workgroup_per_instance",
                                   "one_workgroup_with_elements", "mfma_work_lists"])
def test_cfg5_shard_vs_oracle(qd, route):
    """BASELINE cfg 5, the per-GPU shard of the 8-GPU run: 12 qubits (n = 4096), k = 8, diagonal rotating frame,
    scipy_expm with magnus_order = 2, max_dt = 0.25, T = 5 -> ALL 20 steps, 128 instances in ONE batched device
    solve, dense random y0.  Two routes, each asserted through the launch counters:

      * "default": what the product (and bench.py's cfg5 leg) runs -- ONE launch of ell_flip_duo_kernel<2,2,1024>: two
        workgroups per instance, 256 workgroups for the 128 instances, no operator elements at all (every slot of this stack has
        one signed magnitude and one flip mask, column = row ^ flip) (counter "rk4_resident" == 1 launch, "sweep_split" ==
        (2 workgroups per instance, element form 3 = flip masks), "sweep_cross" == (2 of the 19 slots reach across the halves:
        the drive and the XX coupling of the top qubit));
      * "two_workgroups_with_elements": option ell_sweep_flip = 0, ell_sweep_duo_kernel<2,2,1024,2> (4-byte elements from L2:
        what a stack with one magnitude but several flip masks per slot runs);
      * "one_workgroup_per_instance": option ell_sweep_duo = 0, ell_sweep_kernel<2,4,1024,3> (what a shard of more than 128
        instances runs: round 3's kernel without operator elements); "one_workgroup_with_elements": also ell_sweep_flip = 0,
        round 3's ell_sweep_kernel<2,4,1024,2>;
      * "mfma_work_lists": option ell_sweep = 0, the SPARSE MFMA work-list contraction ("rhs_blocks_gemm").

    Instances 0, 63 and 127 are compared with a CPU evaluation of the same 20 steps that uses the ORACLE's generators
    (oracle.generator_evaluate: G(t) = Delta(t) o (A_d + sum c_j A_j), dense, checked entry by entry against the
    CSR copy used for the matrix-vector products) and a commutator-free Taylor series of expm(Omega_2)
    (fixed_step_solvers.py:345-363 applied to a vector); all 128 instances are checked for norm conservation."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    ctx = qd.default_context()
    cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
    frame = np.diag(cfg["h_d"]).real.copy()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
    assert solver.model.stack.n == 4096
    nb, n = 128, 4096
    sweeps = [_gauss_signals(qd, cfg, b, 8, 2.5) for b in range(nb)]
    rng = np.random.default_rng(12)
    y0 = rng.normal(size=n) + 1j * rng.normal(size=n)
    y0 /= np.linalg.norm(y0)
    h, t_final = 0.25, 5.0
    gave_up_before = ctx.counters("resident_fallbacks")["launches"]
    duo = route in ("default", "two_workgroups_with_elements")
    with ctx.options(ell_sweep=0 if route == "mfma_work_lists" else 1, ell_sweep_duo=1 if duo else 0,
                     ell_sweep_flip=1 if route in ("default", "one_workgroup_per_instance") else 0):
        res = _profiled(ctx, lambda: solver.solve(t_span=[0.0, t_final], y0=y0, signals=sweeps, method="scipy_expm",
                                                  max_dt=h, magnus_order=2))
    if route != "mfma_work_lists":
        assert ctx.counters("rk4_resident")["launches"] == 1, "the one-launch sweep kernel did not take the solve"
        split = ctx.counters("sweep_split")
        if duo:
            form = 3 if route == "default" else 2
            assert (int(split["launches"]), int(split["ms"])) == (2, form), f"not the two-workgroup kernel of element form {form}: {split}"
            cross = ctx.counters("sweep_cross")
            assert (int(cross["launches"]), int(cross["ms"])) == (2, 19), cross
        else:
            form = 3 if route == "one_workgroup_per_instance" else 2
            assert (int(split["launches"]), int(split["ms"])) == (1, form), f"not ell_sweep_kernel<2,4,1024,{form}>: {split}"
        assert ctx.counters("rhs_blocks_gemm")["launches"] == 0
        assert ctx.counters("resident_fallbacks")["launches"] == gave_up_before, "the sweep kernel gave up"
    else:
        assert ctx.counters("rhs_blocks_gemm")["launches"] > 0, "the SPARSE MFMA work-list route did not run"
        assert ctx.counters("rk4_resident")["launches"] == 0
    assert ctx.counters("rhs_gemm")["launches"] == 0, "a dense contraction ran"
    assert all(r.route == "sequential" for r in res)
    assert all(r.nfev == 2 * 20 and r.device == f"hip:{ctx.device}" and r.wall_s > 0 for r in res)   # Magnus 2: 2 / step
    finals = np.stack([r.y[-1] for r in res])
    assert np.max(np.abs(np.linalg.norm(finals, axis=1) - 1.0)) < 1e-11

    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], frame)
    assert basis is None
    a_d_s = sp.csr_matrix(a_d)
    a_s = [sp.csr_matrix(x) for x in a]
    c1, c2 = 0.5 - np.sqrt(3) / 6, 0.5 + np.sqrt(3) / 6

    def gen_sparse(coeffs, t):
        c = a_d_s.copy()
        for cj, aj in zip(coeffs, a_s):
            c = c + cj * aj
        e = np.exp(d * t)
        return sp.diags(e.conj()) @ c @ sp.diags(e)

    # the CSR generator IS the oracle's generator (dense comparison at one instance / time)
    cchk = np.array([np.real(s(1.3)) for s in sweeps[5]])
    assert np.max(np.abs(gen_sparse(cchk, 1.3).toarray() - orc.generator_evaluate(a_d, a, cchk, d, None, 1.3))) < 1e-15

    for b in (0, 63, 127):
        y = y0.copy()
        for st in range(20):
            t0 = st * h
            t1, t2 = t0 + c1 * h, t0 + c2 * h
            g1 = gen_sparse(np.array([np.real(s(t1)) for s in sweeps[b]]), t1)
            g2 = gen_sparse(np.array([np.real(s(t2)) for s in sweeps[b]]), t2)

            def omega(v):
                u1, u2 = g1 @ v, g2 @ v
                return (h / 2) * (u1 + u2) + (np.sqrt(3) / 12) * h * h * (g2 @ u1 - g1 @ u2)

            for _ in range(4):          # ||Omega|| ~ 0.05-0.2: four scalings, Taylor degree 14
                term, acc = y, y.copy()
                for j in range(1, 15):
                    term = omega(term) / (4 * j)
                    acc = acc + term
                y = acc
        assert_close(res[b].y[-1], y, SOLVE_TOL)


def test_cfg5_two_full_size_steps_vs_the_oracles_magnus2_and_scipy_expm(qd):
    """BASELINE cfg 5 at full size against the REAL algorithm: the 128-instance shard on the product's
    default route (ONE launch of ell_flip_duo_kernel<2, 2, 1024>, asserted through the counters) over t_span = [2.25, 2.75] --
    two steps of max_dt = 0.25 in the middle of the pulse -- and instance 63 against
    oracle.solve_generator_model(..., "scipy_expm", magnus_order=2): the oracle's dense generators G(t1), G(t2) (n = 4096),
    oracle.magnus_terms (h (G1 + G2) / 2 + sqrt(3)/12 h^2 [G2, G1], two dense n^3 products) and scipy.linalg.expm of the 4096 x 4096
    Omega, as the reference does it (solvers/fixed_step_solvers.py:80-108, 345-363) -- no test-local series, no sparse copy.
    About 25 s of CPU.  Tolerance 1e-9 (SOLVE_TOL)."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    ctx = qd.default_context()
    cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
    frame = np.diag(cfg["h_d"]).real.copy()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
    nb, n, pick = 128, 4096, 63
    sweeps = [_gauss_signals(qd, cfg, b, 8, 2.5) for b in range(nb)]
    rng = np.random.default_rng(2026)
    y0 = rng.normal(size=n) + 1j * rng.normal(size=n)
    y0 /= np.linalg.norm(y0)
    t_span = [2.25, 2.75]
    gave_up_before = ctx.counters("resident_fallbacks")["launches"]
    res = _profiled(ctx, lambda: solver.solve(t_span=t_span, y0=y0, signals=sweeps, method="scipy_expm", max_dt=0.25,
                                              magnus_order=2))
    assert ctx.counters("rk4_resident")["launches"] == 1
    split = ctx.counters("sweep_split")
    assert (int(split["launches"]), int(split["ms"])) == (2, 3), f"not ell_flip_duo_kernel: {split}"
    assert ctx.counters("resident_fallbacks")["launches"] == gave_up_before
    assert all(r.nfev == 2 * 2 for r in res)
    finals = np.stack([r.y[-1] for r in res])
    assert np.max(np.abs(np.linalg.norm(finals, axis=1) - 1.0)) < 1e-12
    assert np.max(np.abs(finals[pick] - y0)) > 5e-4          # (the pulse is on: the state moves, by ~10 % of an entry)

    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], frame)
    assert basis is None
    sig = sweeps[pick]
    _, ys = orc.solve_generator_model(a_d, a, d, None, lambda t: np.array([np.real(s_(t)) for s_ in sig]), t_span, y0,
                                      method="scipy_expm", max_dt=0.25, magnus_order=2)
    assert_close(res[pick].y[-1], ys[-1], SOLVE_TOL)
    assert np.max(np.abs(res[pick].y[-1] - ys[-1])) < 1e-11      # (what two steps actually differ by is far below the bound)


@pytest.mark.parametrize("frame", ["no_frame", "diag_frame"])
def test_cfg4_default_route_all_steps_vs_oracle(qd, frame):
    """BASELINE cfg 4 exactly as bench.py's cfg4 leg times it: 6 qubits, N = 4096 superoperators built
    on the device, 4 static dissipators, scipy_expm with magnus_order = 1, max_dt = 0.05, t_span = [0, 5] -> ALL 100
    steps, one trajectory from the projector on e_0 -- once without a frame and once in the diagonal frame diag(H_d)
    (the "second run" of 8(d)).  The launch counters must show the product's default route: the whole solve in ONE
    launch of ell_resident_kernel<1, E> (counter "rk4_resident" == 1 launch, no per-product launches, no fallback).
    Compared with the ORACLE's matrix form of the Lindbladian (oracle.lindblad_rhs: n x n products, frame phases of
    models/lindblad_model.py:510-531) through the Magnus-1 step of fixed_step_solvers.py:345-349 applied to rho as a
    scaled Taylor series of expm(h L(t + h/2)) -- 100 steps x 16 scalings x 24 terms, about half a minute of CPU."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    ctx = qd.default_context()
    cfg = workloads.lindblad_config()
    frame_op = None if frame == "no_frame" else np.diag(cfg["h_d"]).real.copy()
    sigs = _gauss_signals(qd, cfg, 0, 6, 2.5)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], rotating_frame=frame_op, vectorized=True)
    assert solver.model.stack.n == 4096
    h = cfg["max_dt"]
    assert (h, cfg["t_span"]) == (0.05, [0.0, 5.0])
    gave_up_before = ctx.counters("resident_fallbacks")["launches"]
    res = _profiled(ctx, lambda: solver.solve(t_span=cfg["t_span"], y0=cfg["rho0"].flatten(order="F"), signals=sigs,
                                              method="scipy_expm", max_dt=h))
    assert ctx.counters("rk4_resident")["launches"] == 1, "not the one-launch ell_resident route"
    assert ctx.counters("rhs_blocks")["launches"] == 0 and ctx.counters("rhs_stream")["launches"] == 0
    assert ctx.counters("rhs_blocks_gemm")["launches"] == 0 and ctx.counters("zgemm")["launches"] == 0
    assert ctx.counters("resident_fallbacks")["launches"] == gave_up_before, "the one-launch kernel gave up"
    assert res.nfev == 100 and res.device == f"hip:{ctx.device}" and res.wall_s > 0   # Magnus 1: one G(t) per step
    rho_dev = res.y[-1].reshape(64, 64, order="F")

    h_d, h_ops, n_static, l_ops, d, basis = orc.lindblad_model_build(cfg["h_d"], cfg["ops"],
                                                                     cfg["static_dissipators"], None, frame_op)
    assert basis is None and (d is None) == (frame == "no_frame")
    from threadpoolctl import threadpool_limits

    n_steps = 100
    rho = cfg["rho0"].astype(complex)
    scal, degree = 16, 24                    # ||h L|| ~ 10 without a frame -> ||h L / 16|| < 0.7, 0.7^24 / 24! ~ 1e-28
    with threadpool_limits(limits=1):        # 64 x 64 products: BLAS threads only add overhead
        for st in range(n_steps):
            t_mid = st * h + h / 2           # Magnus order 1: Omega = h L(t + h/2)
            coeffs = np.array([np.real(s(t_mid)) for s in sigs])
            for _ in range(scal):
                term = rho
                acc = rho.copy()
                for j in range(1, degree + 1):
                    term = orc.lindblad_rhs(h_d, h_ops, n_static, l_ops, coeffs, None, d, t_mid, term) * (h / (scal * j))
                    acc = acc + term
                rho = acc
    assert_close(rho_dev, rho, SOLVE_TOL)
    assert abs(np.trace(rho_dev) - 1.0) < 1e-11
    assert np.linalg.norm(rho_dev - rho_dev.conj().T) < 1e-11
    assert np.min(np.linalg.eigvalsh((rho_dev + rho_dev.conj().T) / 2)) > -1e-11
    # the pulse has acted: the state is far from where it started (a test on a quiet window would pass a wrong operator)
    assert np.linalg.norm(rho_dev - cfg["rho0"]) > 0.05


@pytest.mark.usefixtures("per_launch_routes")
def test_cfg4_sweep_sparse_mfma_route_vs_oracle(qd):
    """BASELINE cfg 4 model (6 qubits, N = 4096 superoperators built on the device, 4 static dissipators, no
    frame) as a 64-instance sweep: scipy_expm (Magnus 1) over 3 steps through the SPARSE MFMA work-list route
    (asserted through the launch counters) against the oracle's matrix form of the Lindbladian
    (oracle.lindblad_rhs, n x n products, a scaled Taylor series of expm(h L) rho) for instances 0, 31, 63;
    trace / Hermiticity for all 64."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    ctx = qd.default_context()
    cfg = workloads.lindblad_config()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], vectorized=True)
    assert solver.model.stack.n == 4096
    nb = 64
    sweeps = [_gauss_signals(qd, cfg, b, 6, 2.5) for b in range(nb)]
    h, n_steps, t0 = cfg["max_dt"], 3, 2.4
    rng = np.random.default_rng(4)
    psi = rng.normal(size=64) + 1j * rng.normal(size=64)
    psi /= np.linalg.norm(psi)
    rho0 = 0.7 * np.outer(psi, psi.conj()) + 0.3 * np.eye(64) / 64      # dense, full-rank density matrix
    res = _profiled(ctx, lambda: solver.solve(t_span=[t0, t0 + n_steps * h], y0=rho0.flatten(order="F"),
                                              signals=sweeps, method="scipy_expm", max_dt=h))
    assert ctx.counters("rhs_blocks_gemm")["launches"] > 0, "the SPARSE MFMA work-list route did not run"
    assert ctx.counters("rhs_gemm")["launches"] == 0
    h_d, h_ops, n_static, l_ops, d, basis = orc.lindblad_model_build(cfg["h_d"], cfg["ops"],
                                                                     cfg["static_dissipators"], None, None)
    for b in range(nb):
        rho_dev = res[b].y[-1].reshape(64, 64, order="F")
        assert abs(np.trace(rho_dev) - 1.0) < 1e-12
        assert np.linalg.norm(rho_dev - rho_dev.conj().T) < 1e-12
    for b in (0, 31, 63):
        rho = rho0.astype(complex)
        for st in range(n_steps):
            t_mid = t0 + st * h + h / 2
            coeffs = np.array([np.real(s(t_mid)) for s in sweeps[b]])
            scal = 64
            for _ in range(scal):
                term, acc = rho, rho.copy()
                for j in range(1, 16):
                    term = orc.lindblad_rhs(h_d, h_ops, n_static, l_ops, coeffs, None, d, t_mid, term) * (h / (scal * j))
                    acc = acc + term
                rho = acc
        assert_close(res[b].y[-1].reshape(64, 64, order="F"), rho, SOLVE_TOL)


@pytest.mark.parametrize("n,scale,kind", [(1024, 2.5, "antiherm"), (1024, 0.04, "antiherm"), (1024, 1.2, "general"),
                                          (2048, 6.0, "antiherm"), (2048, 0.5, "general")])
def test_expm_vs_scipy_large(qd, n, scale, kind):
    """dev_expm (Taylor / Paterson-Stockmeyer scaling & squaring on the MFMA zgemm) against scipy.linalg.expm at
    the sizes of cfg 2/3 propagators and beyond: ||E - E_ref||_1 / ||E_ref||_1 <= 1e-12."""
    rng = np.random.default_rng(n + int(scale * 100))
    a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    if kind == "antiherm":
        a = a - a.conj().T
    a *= scale / np.linalg.norm(a, 1)
    e = qd.default_context().expm(a)
    ref = scipy.linalg.expm(a)
    assert np.linalg.norm(e - ref, 1) / np.linalg.norm(ref, 1) < 1e-12
    if kind == "antiherm":
        assert np.linalg.norm(e.conj().T @ e - np.eye(n)) < 1e-12 * n


def test_badly_scaled_operators_rhs_is_componentwise_accurate(qd):
    """Operators and states whose imaginary parts are 1e-8 of their real parts (entries spanning 1e-8 ... 1).
    BLAS zgemm, which the reference calls (operator_collections.py:124-134), forms a complex product with 4 real
    products, Im = Ar.Bi + Ai.Br: the small imaginary part of the result comes out with a relative error of a few
    ulps (componentwise bound).  The 3M product forms Im = (Ar+Ai)(Br+Bi) - Ar.Br - Ai.Bi: an absolute error of
    eps.|Ar.Br|, i.e. a RELATIVE error of about 1e-16 / 1e-8 in that imaginary part (normwise bound only).
    Decision taken from this test: direct evaluations (model.evaluate_rhs / evaluate, ctx.zgemm) use 4M and are
    componentwise accurate like the reference; 3M stays inside the solver loops and the expm pipeline, where the
    contract is the normwise solve tolerance.  `complex_3m` = 2 forces 3M everywhere, 0 disables it."""
    ctx = qd.default_context()
    n, k, nb = 256, 3, 96
    rng = np.random.default_rng(8)

    def skew(*shape):
        return rng.uniform(-1, 1, shape) + 1e-8j * rng.uniform(-1, 1, shape)

    ops = skew(k, n, n)
    static = skew(n, n)
    stack = qd.Stack(ctx, ops, static, None)
    y = skew(n, nb)
    coeffs = rng.uniform(-1, 1, k)
    g = static + np.tensordot(coeffs, ops, axes=1)
    # reference with exact-ish imaginary part: the four real products in long double
    gr, gi, yr, yi = (np.asarray(x, dtype=np.longdouble) for x in (g.real, g.imag, y.real, y.imag))
    ref_im = np.asarray(gr @ yi + gi @ yr, dtype=float)
    ref_re = np.asarray(gr @ yr - gi @ yi, dtype=float)
    im_scale = np.max(np.abs(ref_im))
    assert im_scale < 1e-6                                   # the imaginary part IS small
    out = stack.eval_rhs(coeffs, 0.0, y)
    assert np.max(np.abs(out.real - ref_re)) < 1e-12
    rel_4m = np.max(np.abs(out.imag - ref_im)) / im_scale
    assert rel_4m < 1e-12, f"default direct evaluation lost the small components: {rel_4m:.2e}"
    c = ctx.zgemm(g, y)
    assert np.max(np.abs(c.imag - ref_im)) / im_scale < 1e-12
    ctx.set_option("complex_3m", 2)            # 2: 3M also for direct evaluations
    try:
        out3 = stack.eval_rhs(coeffs, 0.0, y)
    finally:
        ctx.set_option("complex_3m", 1)
    ref = ref_re + 1j * ref_im
    assert np.max(np.abs(out3 - ref)) < 1e-12 * (1 + np.max(np.abs(ref)))     # normwise: fine
    rel_3m = np.max(np.abs(out3.imag - ref_im)) / im_scale
    assert rel_3m > 100 * rel_4m, (rel_3m, rel_4m)                            # componentwise: visibly worse


def test_strongly_damped_lindbladian_parallel_in_time_vs_sequential(qd, monkeypatch):
    """Parallel-in-time propagation of a STRONGLY dissipative vectorised Lindbladian (gamma ~ the Hamiltonian
    scale, state decays by orders of magnitude in the off-diagonals) against the sequential route and the oracle:
    recorded evidence for the routing rule (automatic routing is limited to HamiltonianModels; `.route` says
    which route ran)."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import solvers as S
    from qiskit_dynamics_amd import workloads as W

    cfg = W.lindblad_config(n_qubits=3, n_drives=3, n_diss=3, gamma=4.0, t_final=2.0, max_dt=0.002)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 1.0) ** 2) / 2.0), nu, 0.3 * a)
            for a, nu in zip((0.9, 0.5, 0.7), cfg["carrier"])]
    m = qd.LindbladModel(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], hamiltonian_signals=sigs,
                         static_dissipators=cfg["static_dissipators"], vectorized=True)
    rng = np.random.default_rng(1)
    psi = rng.normal(size=8) + 1j * rng.normal(size=8)
    psi /= np.linalg.norm(psi)
    y0 = np.outer(psi, psi.conj()).flatten(order="F")
    kw = dict(t_span=[0.0, 2.0], y0=y0, max_dt=0.002)
    for method, par in (("RK4", "hip_RK4_parallel"), ("scipy_expm", "hip_expm_parallel")):
        seq = qd.solve_lmde(m, method=method, **kw)
        assert seq.route == "sequential"            # dissipative model: no automatic re-routing
        par_res = qd.solve_lmde(m, method=par, **kw)
        assert par_res.route == "parallel_in_time"
        monkeypatch.setattr(S, "AUTO_PARALLEL_IN_TIME", "all")
        auto = qd.solve_lmde(m, method=method, **kw)
        monkeypatch.setattr(S, "AUTO_PARALLEL_IN_TIME", True)
        assert auto.route == "parallel_in_time(auto)"
        assert_close(par_res.y[-1], seq.y[-1], 1e-11)
        assert_close(auto.y[-1], seq.y[-1], 1e-11)
        rho = seq.y[-1].reshape(8, 8, order="F")
        assert abs(np.trace(rho) - 1.0) < 1e-11
        assert np.max(np.abs(rho[0, 1:])) < 0.2      # the damping is strong: coherences have decayed
    # and the oracle
    h_d, h_ops, n_static, l_ops, d, basis = orc.lindblad_model_build(cfg["h_d"], cfg["ops"],
                                                                     cfg["static_dissipators"], None, None)
    a_d, a = orc.vectorized_lindblad_stack(h_d, h_ops, n_static, l_ops)
    _, yref = orc.solve_generator_model(a_d, a, None, None, lambda t: np.array([np.real(s(t)) for s in sigs]),
                                        [0.0, 2.0], y0, "RK4", 0.002)
    assert_close(qd.solve_lmde(m, method="RK4", **kw).y[-1], yref[-1], SOLVE_TOL)


def test_shared_signal_sweep_is_folded_into_columns(qd):
    """A sweep over y0 only (all instances share their signals): solved as ONE problem with B columns; results
    equal the per-instance solves (solver_classes.py:568-586 would loop them)."""
    from qiskit_dynamics_amd import workloads as W

    cfg = W.schrodinger_config(n_qubits=5, n_drives=3, t_final=1.0, max_dt=0.01)
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, 0.2)
            for a, nu in zip((0.9, 0.5, 0.7), cfg["carrier"])]
    rng = np.random.default_rng(3)
    y0s = [crand(rng, 32) for _ in range(20)]
    for method, kw in (("RK4", {}), ("scipy_expm", {"magnus_order": 2})):
        many = solver.solve(t_span=[0.0, 0.5], y0=y0s, signals=sigs, method=method, max_dt=0.01,
                            t_eval=[0.1, 0.5], **kw)
        assert len(many) == 20 and many[0].y.shape == (2, 32)
        for b in (0, 7, 19):
            one = solver.solve(t_span=[0.0, 0.5], y0=y0s[b], signals=sigs, method=method, max_dt=0.01,
                               t_eval=[0.1, 0.5], **kw)
            assert_close(many[b].y, one.y, 1e-11)
    # matrix-valued states fold as well
    y0m = [crand(rng, 32, 3) for _ in range(4)]
    many = solver.solve(t_span=[0.0, 0.2], y0=y0m, signals=sigs, method="RK4", max_dt=0.01)
    one = solver.solve(t_span=[0.0, 0.2], y0=y0m[2], signals=sigs, method="RK4", max_dt=0.01)
    assert_close(many[2].y, one.y, 1e-11)
    assert many[0].route.endswith("+folded")
    # identical instances (same signals AND the same y0 object): one solve, replicated
    same = solver.solve(t_span=[[0.0, 0.2]] * 3, y0=y0m[2], signals=sigs, method="RK4", max_dt=0.01)
    assert len(same) == 3 and same[0].route.endswith("+replicated")
    assert same[1].y is not same[0].y and np.array_equal(same[1].y, same[0].y) and same[2].nfev == 4 * 20
    assert_close(same[1].y, one.y, 1e-11)
    # beyond FOLD_MAX_COLUMNS the instances stay instances
    from qiskit_dynamics_amd import solvers as S
    old_cap = S.FOLD_MAX_COLUMNS
    S.FOLD_MAX_COLUMNS = 8
    try:
        capped = solver.solve(t_span=[0.0, 0.2], y0=y0m, signals=sigs, method="RK4", max_dt=0.01)
    finally:
        S.FOLD_MAX_COLUMNS = old_cap
    assert "folded" not in capped[0].route
    assert_close(capped[2].y, one.y, 1e-11)


def test_stack_broadcast_over_the_c_abi_one_rank(qd):
    """midyn_comm_get_unique_id / midyn_comm_init_rank / midyn_stack_create_empty / midyn_stack_broadcast /
    midyn_stack_broadcast_from on one GPU (world size 1): librccl is resolved at first use, the communicator is bound
    to the context's device, the stack works after the in-place call and lazily built lists are rebuilt; the
    out-of-place call fills an empty stack, which then evaluates bit-identically.  The N > 1 use of the same entry
    points is bench.py --gpus N (driver's scaling run)."""
    from qiskit_dynamics_amd import _lib

    ctx = qd.default_context()
    rng = np.random.default_rng(3)
    n, k = 96, 3
    ops, static = crand(rng, k, n, n), crand(rng, n, n)
    fim = rng.normal(size=n)
    stack = qd.Stack(ctx, ops, static, fim)
    y = crand(rng, n, 5)
    c = rng.uniform(-1, 1, k)
    before = stack.eval_rhs(c, 0.3, y)
    uid = _lib.Comm.unique_id()
    assert len(uid) == 128 and _lib.RCCL_LIBRARY
    comm = _lib.Comm(ctx, 1, 0, uid)
    stack.broadcast(comm, 0)
    assert stack.n == n and stack.k == k and stack.n_active_segments == k + 1
    assert np.array_equal(stack.eval_rhs(c, 0.3, y), before)
    # a receiving-side stack: empty until a broadcast fills it -- the out-of-place broadcast of a one-rank
    # communicator is a copy through RCCL followed by the receiving side (stack_after_receive)
    empty = _lib.Stack.empty(ctx, n, k, True, True)
    assert empty.n == n and empty.n_active_segments == 0
    empty.broadcast_from(stack, comm, 0)
    assert empty.n_active_segments == k + 1 and empty.segment_modes == stack.segment_modes
    assert np.array_equal(empty.eval_rhs(c, 0.3, y), before)
    assert np.array_equal(empty.eval_generator(c, 0.3), stack.eval_generator(c, 0.3))
    with pytest.raises(qd.DynamicsError):
        stack.broadcast(comm, 3)                     # root out of range
    other = _lib.Stack.empty(ctx, n, k + 1, True, True)
    with pytest.raises(qd.DynamicsError):
        other.broadcast_from(stack, comm, 0)         # shapes differ
    comm.close()
    empty.close()
    other.close()


def test_receiving_rank_of_the_stack_broadcast_runs_the_headline_solve(qd):
    """What ranks 1..N-1 of `bench.py --gpus N` do, on ONE GPU and with data: the sector-grouped stack of the 10-qubit
    model (BASELINE configs[2]: n = 1024, k = 8, rotating_frame = H_d) is sent through ncclBroadcast into a stack
    from midyn_stack_create_empty (midyn_stack_broadcast_from, one-rank communicator); the receiving side re-derives
    plane flags, active-segment lists and -- lazily -- the tile work lists from the RECEIVED buffer.  The product
    Solver then runs the headline sweep (512 instances x 20 RK4 steps, active pulse window) on the stack it built and on
    the received one, on the default route (combine + apply: the layout of midyn_combine.h is re-packed from the received
    buffer) and on the MFMA work-list route (combine=0): same launches, same lists, `np.array_equal` final states;
    instance 100 against the oracle.  Reference seam:
    solvers/solver_classes.py:568-586 (the sequential instance loop the shards replace)."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import _lib, workloads
    from threadpoolctl import threadpool_limits

    ctx = qd.default_context()
    cfg = workloads.schrodinger_config()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    src = solver.model.stack
    assert src.n >= 1024 and src.slot is not None, "the sector-grouped stack was expected"
    nb, t_span = 512, [2.4, 2.5]
    sweeps = [_gauss_signals(qd, cfg, b, 8, 2.5) for b in range(nb)]
    rng = np.random.default_rng(77)
    y0 = crand(rng, 1024)
    y0 /= np.linalg.norm(y0)

    names = ("rhs_combine", "rhs_blocks_gemm", "rhs_gemm", "sparse_tile", "sparse_list", "combine_info", "combine_shape")

    def run(combine=1):
        ctx.set_option("combine", combine)
        try:
            res = _profiled(ctx, lambda: solver.solve(t_span=t_span, y0=y0, signals=sweeps, method="RK4",
                                                      max_dt=cfg["max_dt"]))
        finally:
            ctx.set_option("combine", 1)
        return np.stack([r.y[-1] for r in res]), {c: ctx.counters(c) for c in names}

    built, c_built = run()                       # default route: combine + apply on the sector lists
    assert c_built["rhs_combine"]["launches"] == 80 and c_built["rhs_blocks_gemm"]["launches"] == 0, c_built
    built_g, c_built_g = run(combine=0)          # the MFMA work-list route
    assert c_built_g["rhs_blocks_gemm"]["launches"] == 80 and c_built_g["rhs_gemm"]["launches"] == 0, c_built_g

    comm = _lib.Comm(ctx, 1, 0, _lib.Comm.unique_id())
    dst = _lib.Stack.empty(ctx, src.n, src.k, src.has_static, src.has_frame)
    assert dst.n_active_segments == 0
    dst.broadcast_from(src, comm, 0)
    comm.close()
    dst.set_embedding(src.slot)        # host-side index map: travels beside the buffer (distributed.broadcast_stack)
    assert dst.n_active_segments == src.n_active_segments and dst.segment_modes == src.segment_modes
    assert dst.block_info() == src.block_info()
    solver.model._stack = dst          # the solver of a receiving rank holds the received stack
    try:
        received, c_recv = run()
        received_g, c_recv_g = run(combine=0)
    finally:
        solver.model._stack = src
    for got, want in ((c_recv, c_built), (c_recv_g, c_built_g)):
        for c in ("rhs_combine", "rhs_blocks_gemm", "rhs_gemm"):
            assert got[c]["launches"] == want[c]["launches"], (c, got, want)
        for c in ("combine_info", "combine_shape"):
            assert got[c] == want[c], (c, got, want)
    assert c_recv_g["sparse_list"] == c_built_g["sparse_list"]
    assert (int(c_recv_g["sparse_tile"]["launches"]), int(c_recv_g["sparse_tile"]["ms"])) == (128, 128)
    assert np.array_equal(received, built), f"received stack: max|d| = {np.max(np.abs(received - built)):.3e}"
    assert np.array_equal(received_g, built_g)
    assert_close(built, built_g, 1e-13)
    dst.close()

    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])
    amps, phases = workloads.sweep_parameters(100, 8)

    def coeffs(t):
        return workloads.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], cfg["t_final"])[0]

    with threadpool_limits(limits=8):
        _, yref = orc.solve_generator_model(a_d, a, d, basis, coeffs, t_span, y0, "RK4", cfg["max_dt"])
    assert_close(received[100], yref[-1], SOLVE_TOL)


def test_abi_broadcast_probe_in_a_child_process(qd, monkeypatch):
    """distributed.abi_broadcast_probe: the C-ABI communicator + stack broadcast self-test that bench.py runs in a
    child process per rank before it puts its own context into the collective.  One rank here: the child builds
    the probe stack, broadcasts it, checks an evaluation and exits 0; a child that hangs is killed by PID after the
    time limit and reported as a failure (the caller then takes the torch.distributed route)."""
    from qiskit_dynamics_amd import _lib
    from qiskit_dynamics_amd.distributed import abi_broadcast_probe

    ctx = qd.default_context()
    ok, msg = abi_broadcast_probe(0, 1, ctx.device, _lib.Comm.unique_id(), timeout_s=120)
    assert ok, msg
    monkeypatch.setenv("MIDYN_PROBE_HANG", "1")
    ok, msg = abi_broadcast_probe(0, 1, ctx.device, _lib.Comm.unique_id(), timeout_s=3)
    assert not ok and "killed" in msg
    monkeypatch.delenv("MIDYN_PROBE_HANG")
    ok, msg = abi_broadcast_probe(0, 1, ctx.device, b"\0" * 16, timeout_s=60)       # malformed id: exits non-zero
    assert not ok and "exited" in msg


def test_bench_dry_ranks_mode_one_rank():
    """bench.py --dry-ranks (the multi-GPU plumbing check) with the one GPU of this box: communicator of one rank through
    the C-ABI, broadcast, evaluation of the broadcast stack."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dry-ranks"], capture_output=True, text=True,
                       timeout=300, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["mode"] == "dry-ranks" and line["rccl_ranks_seen"] == 1
    assert line["every_rank_evaluates_the_broadcast_stack_correctly"] is True


def test_event_timer_and_block_info(qd):
    """midyn_ctx_timer brackets launches on the library's stream; midyn_stack_block_info reports what the
    work-list kernels execute (checked against a host count of the non-zero 16x16 blocks)."""
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(n_qubits=9, n_drives=8, t_final=1.0, max_dt=0.01)
    frame = np.diag(cfg["h_d"]).real.copy()
    m = qd.HamiltonianModel(static_operator=cfg["h_d"], operators=cfg["ops"],
                            signals=[qd.Signal(1.0, 5.0)] * 8, rotating_frame=frame)
    info = m.stack.block_info()
    assert info["state"] == 1
    g = [-1j * (cfg["h_d"] - np.diag(np.diag(cfg["h_d"])))] + [-1j * o for o in cfg["ops"]]
    nz = 0
    for a in g:
        blocks = np.abs(a).reshape(32, 16, 32, 16).max(axis=(1, 3))
        nz += int(np.count_nonzero(blocks))
    assert info["nonzero_blocks"] == nz and info["blocks_per_side"] == 32
    assert abs(info["block_density"] - nz / (9 * 32 * 32)) < 1e-12
    assert info["tile_lists"][16]["listed_tiles"] == nz          # 16-row panels list exactly the non-zero blocks
    assert info["tile_lists"][128]["listed_tiles"] >= nz / 8
    ctx.timer_start()
    for _ in range(5):
        m.evaluate_rhs(0.1, cfg["y0"])
    ms = ctx.timer_stop()
    assert 0.0 < ms < 1000.0


@pytest.mark.usefixtures("per_launch_routes")
def test_paired_sparse_launches_are_bit_identical(qd):
    """The two independent products of a Magnus-2 level share ONE launch on the sparse MFMA route (ctx option
    pair_launch, zgemm_seg_pair_kernel): same arithmetic per product, so the solve must be bit-identical to the
    one-product-per-launch route -- with a diagonal frame (phased chain) and without a frame, split and unsplit."""
    ctx = qd.default_context()
    from qiskit_dynamics_amd import workloads as W

    cfg = W.schrodinger_config(n_qubits=9, n_drives=8, t_final=1.0, max_dt=0.05)
    rng = np.random.default_rng(6)
    y0 = crand(rng, 512)
    y0 /= np.linalg.norm(y0)
    for frame in (np.diag(cfg["h_d"]).real.copy(), None):
        solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=frame)
        for nb in (24, 130):
            sweeps = [_gauss_signals(qd, cfg, b, 8, 0.5) for b in range(nb)]
            runs = {}
            for pair in (1, 0):
                ctx.set_option("pair_launch", pair)
                try:
                    r = _profiled(ctx, lambda: solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sweeps,
                                                            method="scipy_expm", max_dt=0.05, magnus_order=2))
                finally:
                    ctx.set_option("pair_launch", 1)
                launches = ctx.counters("rhs_blocks_gemm")["launches"]
                assert launches > 0
                runs[pair] = (np.stack([x.y[-1] for x in r]), launches)
            assert runs[1][1] * 2 == runs[0][1], (runs[1][1], runs[0][1])      # half the launches
            assert np.array_equal(runs[1][0], runs[0][0])
            assert np.max(np.abs(np.linalg.norm(runs[1][0], axis=1) - 1.0)) < 1e-10


def test_bench_two_ranks_share_one_gpu():
    """`python bench.py --gpus 2` spawning its own ranks, with both ranks on THIS GPU (MIDYN_BENCH_SHARE_GPU: gloo
    rendezvous, every rank builds its stack -- RCCL cannot put two ranks on one device): the multi-rank flow of the
    benchmark with real kernels -- strong-scaling shards of the 4096-instance sweep, per-rank plans, barriers, MAX
    reduction, exactly one JSON line.  (The RCCL broadcast itself: test_stack_broadcast_over_the_c_abi_one_rank and
    MIDYN_BENCH_FORCE_DIST; a real multi-GPU run is the driver's.)"""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIDYN_BENCH_SHARE_GPU="1")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--repeats", "1", "--no-configs"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["config"]["global_instances"] == 4096 and out["config"]["instances_per_gpu"] == 2048
    assert out["max_norm_deviation"] < 1e-10 and out["value"] > 1e5
    assert out["roofline"]["frac"] <= 1.0
    assert len(lines[0]) < 6144                      # the N > 1 line too fits the driver's stdout tail


def test_bench_line_of_a_real_run_fits_the_drivers_stdout_tail():
    """`python bench.py` at N = 1 with the cfg 2 leg and the CPU baseline on (cfg 4 / cfg 5, the SURVEY-8 rows and the A/B variants off to
    keep the test short; their compact forms are covered on CPU by test_bench_line_is_compact_and_complete and by the two-rank cfg 5
    test below): the LAST stdout line is the compact object -- below 6 KB of the 8 018 characters the driver keeps
    (VERDICT round 5 item 1), with the contract fields, roofline and cpu_baseline -- and bench_detail.json beside the script
    holds the full result it was reduced from."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--repeats", "1",
                        "--no-rows", "--no-variants", "--no-end-to-end", "--no-configs", "--no-projection"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6144, (len(lines), len(lines[-1]))
    c = json.loads(lines[0])
    assert {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "cfg2", "detail"} <= set(c)
    assert c["n_gpus"] == 1 and c["steps"] == 4 and c["dtype"] == "f64" and c["vs_baseline"] is None
    assert {"kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "executed_flops_per_launch",
            "frac_survey_8d"} <= set(c["roofline"])
    assert 0.3 < c["roofline"]["frac"] <= 1.0 and c["roofline"]["kernel"].startswith("rhs_combine_kernel<0, 2, 0>")
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["value"] > 0
    assert c["value"] == pytest.approx(4096 * 4 * 1e3 / c["ms_per_step"], rel=1e-3)
    with open(os.path.join(root, c["detail"])) as f:
        full = json.load(f)
    assert full["value"] == c["value"] and "note" in full["roofline"] and "roofline_single_trajectory" in full


def test_bench_two_ranks_share_one_gpu_with_the_sharded_cfg5_leg():
    """The same two self-spawned ranks WITH the second sharded leg of `bench.py --gpus N`: BASELINE configs[4] (n = 4096,
    Magnus-2, 1024 instances in total), 512 instances per rank -- every rank builds the 2.4 GB stack (one GPU: no RCCL
    between them), solves its shard, the slowest rank's time is the leg's."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MIDYN_BENCH_SHARE_GPU="1")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--repeats", "1", "--no-projection", "--no-variants"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2
    assert len(lines[0]) < 6144
    assert "error" not in out["sharded_cfg5"] and out["sharded_cfg5"]["instances_per_gpu"] == 512     # (the compact line)
    with open(os.path.join(root, out["detail"])) as f:      # the full result of the same run
        leg = json.load(f)["sharded_cfg5"]
    assert "error" not in leg, leg
    assert leg["instances_total"] == 1024 and leg["instances_per_gpu"] == 512 and leg["n_gpus"] == 2
    assert leg["max_norm_deviation_rank0"] < 1e-10 and leg["instance_steps_per_s"] > 1e4
    assert out["sharded_cfg5"]["value"] == pytest.approx(leg["instance_steps_per_s"], rel=1e-3)


def test_c_program_drives_the_hot_path_through_the_c_abi(tmp_path):
    """tests/abi_solve.c: a C99 host (gcc, dlopen of ONE HIP runtime + librccl + libmidyn.so, no Python objects, no
    torch) creates a context and an operator stack, broadcasts it on a one-rank RCCL communicator, evaluates the RHS,
    and solves a sweep of driven qubits with midyn_rk4_solve and midyn_expm_solve (Magnus 2) against the closed-form
    solution -- the drop-in boundary exercised from the other side."""
    import os
    import subprocess

    from qiskit_dynamics_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert _lib.load() is not None
    rccl = _lib.preload_rccl()
    exe = tmp_path / "abi_solve"
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-o", str(exe), os.path.join(root, "tests", "abi_solve.c"),
                    "-ldl", "-lm"], check=True)
    p = subprocess.run([str(exe), _lib.HIP_RUNTIME, _lib.LIB_PATH, rccl], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "ABI_SOLVE_OK" in p.stdout, p.stdout + p.stderr
    assert "broadcast=1" in p.stdout


def test_symmetry_sectors_exact_zero_blocks_are_skipped(qd):
    """A frame operator that conserves parity (8-qubit chain, n = 256, rotating_frame = H_d): the frame basis is
    computed sector by sector, the device stack groups the frame-basis vectors by sector (internal permutation) and
    the batched contraction runs on tile work lists that skip the exactly-zero blocks.  Checked: the public API is in
    the reference's order (in_frame_basis I/O consistent with frame_basis), evaluations and a sweep agree with the
    oracle (plain eigh), the work-list route really runs and equals the dense kernels on the same stack to rounding,
    the one-column streaming kernel reads half of the planes."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    ctx = qd.default_context()
    cfg = W.schrodinger_config(n_qubits=8, n_drives=8, t_final=1.0, max_dt=0.01)
    n = 256
    sigs = _gauss_signals(qd, cfg, 0, 8, 0.5)
    m = qd.HamiltonianModel(static_operator=cfg["h_d"], operators=cfg["ops"], signals=sigs, rotating_frame=cfg["h_d"])
    assert m.rotating_frame.sector_labels is not None and m.stack.perm is not None and m.stack.n_api == m.stack.n
    info = m.stack.block_info()
    assert info["state"] == 1 and abs(info["block_density"] - 0.5) < 0.02 and abs(info["streamed_fraction"] - 0.5) < 0.02
    a_d, a, d, basis = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])
    rng = np.random.default_rng(3)
    t = 0.37
    c = np.array([np.real(s(t)) for s in sigs])
    y = crand(rng, n)
    ym = crand(rng, n, 20)
    assert_close(m.evaluate(t), orc.generator_evaluate(a_d, a, c, d, basis, t, False), 1e-11)
    assert_close(m.evaluate_rhs(t, y), orc.generator_rhs(a_d, a, c, d, basis, t, y, False), 1e-11)
    assert_close(m.evaluate_rhs(t, ym), orc.generator_rhs(a_d, a, c, d, basis, t, ym, False), 1e-11)
    # in-frame-basis I/O uses the model's own frame_basis in the reference's (ascending) order
    u = m.rotating_frame.frame_basis
    assert np.all(np.diff(m.rotating_frame.frame_diag.imag) <= 1e-12)          # d = -i * ascending eigenvalues
    m.in_frame_basis = True
    assert_close(u @ m.evaluate_rhs(t, u.conj().T @ y), orc.generator_rhs(a_d, a, c, d, basis, t, y, False), 1e-11)
    assert_close(u @ m.evaluate(t) @ u.conj().T, orc.generator_evaluate(a_d, a, c, d, basis, t, False), 1e-11)
    m.in_frame_basis = False
    # a sweep: work lists vs dense kernels (bit-identical) vs the oracle
    nb = 24
    sweeps = [_gauss_signals(qd, cfg, b, 8, 0.5) for b in range(nb)]
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    y0 = crand(rng, n)
    y0 /= np.linalg.norm(y0)
    runs = {}
    for flag in (1, 0):
        ctx.set_option("skip_zero_blocks", flag)
        try:
            res = _profiled(ctx, lambda: solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sweeps, method="RK4", max_dt=0.01))
        finally:
            ctx.set_option("skip_zero_blocks", 1)
        lists, dense = ctx.counters("rhs_blocks_gemm")["launches"], ctx.counters("rhs_gemm")["launches"]
        assert (lists > 0 and dense == 0) if flag else (lists == 0 and dense > 0)
        runs[flag] = np.stack([r.y[-1] for r in res])
    assert_close(runs[1], runs[0], 1e-13)      # (same products; tiles / split-K order differ at this small size)
    for b in (0, 11, 23):
        _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt, b=b: np.array([np.real(s(tt)) for s in sweeps[b]]),
                                           [0.0, 0.2], y0, "RK4", 0.01)
        assert_close(runs[1][b], ref[-1], SOLVE_TOL)
    # one trajectory (streaming kernel with column hulls) and the Magnus-2 action
    one = solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sweeps[5], method="RK4", max_dt=0.01)
    assert_close(one.y[-1], runs[1][5], 1e-12)
    ex = solver.solve(t_span=[0.0, 0.2], y0=y0, signals=sweeps[5], method="scipy_expm", max_dt=0.02, magnus_order=2)
    _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt: np.array([np.real(s(tt)) for s in sweeps[5]]),
                                       [0.0, 0.2], y0, "scipy_expm", 0.02, magnus_order=2)
    assert_close(ex.y[-1], ref[-1], SOLVE_TOL)


def test_symmetry_sectors_many_uneven_sectors_and_operators_without_selection_rules(qd):
    """(a) A frame that conserves the excitation number (6-qubit XX+YY chain with detunings: sectors of 1, 6, 15, 20,
    15, 6, 1 basis states) with drives X_j that connect neighbouring sectors only; (b) a block-diagonal frame with
    random DENSE operators that respect no selection rule (the grouping must then be harmless: dense kernels).
    Evaluations in and out of the frame basis and sweeps against the oracle (plain eigh)."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads as W

    rng = np.random.default_rng(17)
    nq, n = 6, 64
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    yp = np.array([[0, -1j], [1j, 0]], dtype=complex)
    z = np.diag([1.0, -1.0]).astype(complex)
    h_frame = np.zeros((n, n), dtype=complex)
    for q in range(nq):
        h_frame += 2 * np.pi * (1.0 + 0.07 * q) * W.embed(z, q, nq) / 2
    for q in range(nq - 1):
        h_frame += 2 * np.pi * 0.03 * (W.embed_pair(x, q, x, q + 1, nq) + W.embed_pair(yp, q, yp, q + 1, nq)) / 2
    drives = np.array([2 * np.pi * 0.05 * W.embed(x, q, nq) / 2 for q in range(4)])

    def herm(m_):
        a_ = crand(rng, m_, m_)
        return (a_ + a_.conj().T) / 2

    block_frame = np.zeros((80, 80), dtype=complex)
    block_frame[:30, :30] = herm(30)
    block_frame[30:, 30:] = herm(50)
    dense_ops = np.array([herm(80) for _ in range(3)])
    # (c) sectors of 100, 70 and 30 states (none a multiple of the 16-row blocks): the internal layout pads each sector
    # to a block boundary (n = 200 -> 212 rows on the device; the sector that would need the most padding goes last and gets none), operators that couple sectors 0 <-> 1 only
    sel_frame = np.zeros((200, 200), dtype=complex)
    bounds = [(0, 100), (100, 170), (170, 200)]
    for lo, hi in bounds:
        sel_frame[lo:hi, lo:hi] = herm(hi - lo)
    sel_ops = np.zeros((3, 200, 200), dtype=complex)
    for j, (lo, hi) in enumerate(bounds):
        sel_ops[j, lo:hi, lo:hi] = herm(hi - lo)
    cpl = crand(rng, 100, 70)
    sel_ops[0, 0:100, 100:170] = cpl
    sel_ops[0, 100:170, 0:100] = cpl.conj().T
    cases = [("excitation", h_frame, h_frame, drives, [1, 1, 6, 6, 15, 15, 20], 64),
             ("dense_ops", block_frame, herm(80), dense_ops, [30, 50], 82),
             ("padded_sectors", sel_frame, sel_frame, sel_ops, [30, 70, 100], 212)]
    for tag, frame, h_static, h_ops, sizes, n_internal in cases:
        k, dim = len(h_ops), frame.shape[0]
        sigs = [qd.Signal(lambda t, a=0.3 + 0.1 * j: a * np.cos(0.9 * t) + 0j, 0.4 * j, 0.2 * j) for j in range(k)]
        m = qd.HamiltonianModel(static_operator=h_static, operators=h_ops, signals=sigs, rotating_frame=frame)
        labels = m.rotating_frame.sector_labels
        assert labels is not None and sorted(np.bincount(labels).tolist()) == sizes, tag
        assert m.stack.slot is not None and m.stack.n_api == dim and m.stack.n == n_internal, tag
        if tag == "padded_sectors":
            # sectors occupy block rows 0-6, 7-11, 12-13 of the 16 x 16-block map: operator 0 fills sector 0's square
            # and the 0 <-> 1 coupling, operators 1 and 2 their own squares; the static operator cancels the frame
            info = m.stack.block_info()
            assert info["state"] == 1 and info["blocks_per_side"] == 16
            assert info["nonzero_blocks"] == (49 + 2 * 35) + 25 + 4, info
        a_d, a, d, basis = orc.hamiltonian_model_build(h_static, h_ops, frame)
        t = 0.83
        c = np.array([np.real(s(t)) for s in sigs])
        yv, ym = crand(rng, dim), crand(rng, dim, 9)
        assert_close(m.evaluate(t), orc.generator_evaluate(a_d, a, c, d, basis, t, False), 1e-11)
        assert_close(m.evaluate_rhs(t, yv), orc.generator_rhs(a_d, a, c, d, basis, t, yv, False), 1e-11)
        assert_close(m.evaluate_rhs(t, ym), orc.generator_rhs(a_d, a, c, d, basis, t, ym, False), 1e-11)
        u = m.rotating_frame.frame_basis
        m.in_frame_basis = True
        assert_close(u @ m.evaluate_rhs(t, u.conj().T @ ym), orc.generator_rhs(a_d, a, c, d, basis, t, ym, False), 1e-11)
        m.in_frame_basis = False
        solver = qd.Solver(static_hamiltonian=h_static, hamiltonian_operators=h_ops, rotating_frame=frame)
        y0s = [crand(rng, dim) for _ in range(12)]
        for method, kw in (("RK4", dict(max_dt=0.01)), ("scipy_expm", dict(max_dt=0.05, magnus_order=2))):
            res = solver.solve(t_span=[0.0, 0.3], y0=y0s, signals=sigs, method=method, **kw)
            for b in (0, 11):
                _, ref = orc.solve_generator_model(a_d, a, d, basis, lambda tt: np.array([np.real(s(tt)) for s in sigs]),
                                                   [0.0, 0.3], y0s[b], method, kw["max_dt"],
                                                   magnus_order=kw.get("magnus_order", 1))
                assert_close(res[b].y[-1], ref[-1], SOLVE_TOL)


SWEEP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["MIDYN_ROOT"])
import numpy as np
import torch.distributed as dist
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.distributed import init_process_group_from_env, shard_bounds, solve_sweep

rank, world = init_process_group_from_env(backend="gloo")     # both ranks compute on THE GPU (device 0), gloo carries the results
assert world == 2
cfg = workloads.schrodinger_config(n_qubits=5, n_drives=3, t_final=1.0, max_dt=0.01)
solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])


def signals_of(b):
    amps, phases = workloads.sweep_parameters(b, 3)
    return [qd.Signal(lambda t, a=a: a * np.exp(-((t - 0.5) ** 2) / 2.0), nu, ph) for a, nu, ph in zip(amps, cfg["carrier"], phases)]


rng = np.random.default_rng(77)
for n_inst, kw in ((5, dict(method="RK4", max_dt=0.01)), (1, dict(method="RK4", max_dt=0.01)),
                   (4, dict(method="scipy_expm", max_dt=0.05, magnus_order=2))):
    sweeps = [signals_of(b) for b in range(n_inst)]
    y_list = []
    for b in range(n_inst):
        v = rng.standard_normal(32) + 1j * rng.standard_normal(32)
        y_list.append(v / np.linalg.norm(v))
    t_span = [0.0, 0.3]
    # the one-rank answer, computed by THIS rank with the same Solver: one list-mode solve of all instances
    ref = solver.solve(t_span=t_span, y0=y_list, signals=sweeps, **kw)
    ref = ref if isinstance(ref, list) else [ref]
    lo, hi = shard_bounds(n_inst, rank, world)
    full = solve_sweep(solver, t_span, y_list, sweeps, gather="all", **kw)
    assert len(full) == n_inst
    for b in range(n_inst):
        assert np.array_equal(full[b].t, ref[b].t)
        # an instance's trajectory does not depend on which other instances share its batched launch: same bits
        assert np.array_equal(full[b].y, ref[b].y), (rank, n_inst, b, float(np.max(np.abs(full[b].y - ref[b].y))))
    at_root = solve_sweep(solver, t_span, y_list, sweeps, gather="root", root=1, **kw)
    if rank == 1:
        assert len(at_root) == n_inst and all(np.array_equal(at_root[b].y, ref[b].y) for b in range(n_inst))
    else:
        assert at_root is None
    off, own = solve_sweep(solver, t_span, y_list, sweeps, gather="none", **kw)
    assert off == lo and len(own) == hi - lo
    for i, r_ in enumerate(own):
        assert np.array_equal(r_.y, ref[lo + i].y)
    # a state shared by all instances
    shared = solve_sweep(solver, t_span, cfg["y0"], sweeps, gather="all", **kw)
    one = solver.solve(t_span=t_span, y0=cfg["y0"], signals=sweeps, **kw)
    one = one if isinstance(one, list) else [one]
    assert all(np.array_equal(shared[b].y, one[b].y) for b in range(n_inst))
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", flush=True)
'''


def test_solve_sweep_with_the_real_solver_on_two_ranks_sharing_the_gpu(tmp_path):
    """`distributed.solve_sweep` (the sharded Solver.solve list mode of SURVEY 8(e); reference loop:
    solvers/solver_classes.py:568-586) with the PRODUCT Solver under two ranks -- both on this GPU, gloo rendezvous, as
    bench.py's MIDYN_BENCH_SHARE_GPU mode does: uneven shards (5 instances = 3 + 2), fewer instances than ranks (1),
    per-instance and shared initial states, RK4 and Magnus-2 `scipy_expm`, gather = "all" / "root" / "none" -- every
    result `array_equal` to the one-rank list-mode solve of the same instances."""
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "sweep_worker.py"
    script.write_text(SWEEP_WORKER)
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MIDYN_ROOT=root)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                      text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=600)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            outs.append(p.communicate()[0] + "\n[timeout]")
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"rank {rank} ok" in out, f"rank {rank}:\n{out[-3000:]}"
