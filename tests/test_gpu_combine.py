"""The COMBINE + APPLY sweep contraction (csrc/midyn_combine.h, rhs_combine_kernel<NRE4, NIM4, STAT>) against the oracle and
against the MFMA GEMM routes (ctx option combine = 0), one test per kernel variant:

  * operators with real planes only / imaginary planes only / both / a mix of the three, 2 .. 8 of them, so that every
    (NRE4, NIM4) in {0, 1, 2}^2 \\ (0, 0) occurs, 9 .. 16 operators of ONE kind ((0, 3), (0, 4), (3, 0), (4, 0)), and 9 .. 12 of a
    kind beside planes of the other ((3, 1), (3, 2), (3, 3), (1, 3), (2, 3));
  * no static operator, a real one, an imaginary one, a complex one (STAT 0 .. 3: the C input of the combining MFMAs);
  * a frame diagonal (phases in the stage input and in the epilogue), ragged dimension (n = 96 -> 128 padded rows),
    300 instances (384 padded columns: eight waves split every list and sum through LDS as a tree; 32 instances per wave --
    the small-sweep variants -- and, under combine_occupancy = 1, 64 per wave with four waves per list);
  * RK4 (epilogues RK1..4) and the expm action of scipy_expm with magnus_order 1 and 2 (Taylor / Chebyshev epilogues).
Every variant is forced with ctx option combine = 2; the default (1) takes the kernel only where it is the faster formulation
(at least three quarters of its plane slots -- groups of four -- must hold a plane): asserted too.

The same 32 variants on the ONE-LAUNCH kernel of RK4 sweeps of small systems (csrc/midyn_combine_sweep.h,
combine_sweep_rk4_kernel<NRE4, NIM4, STAT, RT>: 16 instances per workgroup through all steps) at n_pad = 64 (two waves per
row tile), 128 and 256 (two tiles per wave), against the per-launch kernels and the oracle.

Plus: operators with exactly-zero blocks (lists shorter than the dense ones), matrix-valued states (several columns per
instance share a coefficient row), a stack with more operators than the kernels cover (falls back to the GEMM route).
The reference computes the same thing per instance as  (G_d + sum_j c_j G_j) y  (models/operator_collections.py:101-134).

All of them need a real MI355X (`pytest -m gpu`).
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


def _operators(rng, n, kinds):
    """One n x n generator per entry of `kinds`: 'r' real plane only, 'i' imaginary plane only, 'c' both."""
    ops = []
    for kind in kinds:
        a = crand(rng, n, n) * 0.3
        ops.append(a.real + 0j if kind == "r" else (1j * a.imag if kind == "i" else a))
    return np.array(ops)


def _solve(qd, stack, method, sched, table, y0, batch, shared, combine, magnus_order=1, one_launch=0, occupancy=2):
    """one_launch = 0 pins the per-launch kernels (RK4 sweeps of systems with n_pad <= 256 otherwise run on the one-launch kernel
    of midyn_combine_sweep.h, which has its own tests below)."""
    ctx = qd.default_context()
    ctx.set_option("combine", combine)
    ctx.set_option("combine_sweep", one_launch)
    ctx.set_option("combine_occupancy", occupancy)
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        if method == "RK4":
            ys = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, y0, batch, shared)
        else:
            ys = stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, magnus_order,
                                  y0, batch, shared)
    finally:
        ctx.set_option("profile", 0)
        ctx.set_option("combine", 1)
        ctx.set_option("combine_sweep", 1)
        ctx.set_option("combine_occupancy", 2)
    return ys, {c: ctx.counters(c) for c in ("rhs_combine", "rhs_gemm", "rhs_blocks_gemm", "combine_info", "combine_shape", "combine_wave",
                                             "combine_sweep")}


CASES = [
    # operator kinds, static kind -> (NRE4, NIM4, STAT)
    ("rr", None, (1, 0, 0)), ("rr", "r", (1, 0, 1)), ("rr", "i", (1, 0, 2)), ("rr", "c", (1, 0, 3)),
    ("rrrrr", None, (2, 0, 0)), ("rrrrr", "r", (2, 0, 1)), ("rrrrr", "i", (2, 0, 2)), ("rrrrrrrr", "c", (2, 0, 3)),
    ("iii", None, (0, 1, 0)), ("iii", "r", (0, 1, 1)), ("iiii", "i", (0, 1, 2)), ("i", "c", (0, 1, 3)),
    ("iiiiiiii", None, (0, 2, 0)), ("iiiii", "r", (0, 2, 1)), ("iiiiiiii", "i", (0, 2, 2)), ("iiiiii", "c", (0, 2, 3)),
    ("cc", None, (1, 1, 0)), ("ri", "r", (1, 1, 1)), ("cci", "i", (1, 1, 2)), ("cccc", "c", (1, 1, 3)),
    ("rrrrri", None, (2, 1, 0)), ("ccrrr", "r", (2, 1, 1)), ("rrrrrc", "i", (2, 1, 2)), ("ccccr", "c", (2, 1, 3)),
    ("iiiiir", None, (1, 2, 0)), ("cciii", "r", (1, 2, 1)), ("iiiiic", "i", (1, 2, 2)), ("cccci", "c", (1, 2, 3)),
    ("ccccc", None, (2, 2, 0)), ("cccccc", "r", (2, 2, 1)), ("cccccccc", "i", (2, 2, 2)), ("cccccccc", "c", (2, 2, 3)),
    # one plane kind alone: up to 16 operators (real-symmetric Hamiltonians with more than 8 drives / couplers)
    ("i" * 10, None, (0, 3, 0)), ("i" * 12, "i", (0, 3, 2)), ("i" * 13, "r", (0, 4, 1)), ("i" * 16, "c", (0, 4, 3)),
    ("r" * 9, None, (3, 0, 0)), ("r" * 11, "c", (3, 0, 3)), ("r" * 14, "i", (4, 0, 2)), ("r" * 16, "r", (4, 0, 1)),
    # both kinds, 9 .. 12 operators of a kind (round 5: a third plane group beside planes of the other kind)
    ("c" * 9, None, (3, 3, 0)), ("c" * 12, "c", (3, 3, 3)), ("c" * 10, "r", (3, 3, 1)),
    ("r" * 9 + "ii", "i", (3, 1, 2)), ("c" * 5 + "r" * 5, None, (3, 2, 0)), ("i" * 10 + "c", "c", (1, 3, 3)), ("c" * 8 + "iii", "r", (2, 3, 1)),
]
N_BOTH_KINDS = 32        # the first 32 cases: the (NRE4, NIM4) <= (2, 2) variants, which the one-launch kernels of small sweeps have too


@pytest.mark.parametrize("kinds,static_kind,variant", CASES, ids=["%d%d%d" % v for _, _, v in CASES])
def test_combine_kernel_variant_vs_oracle_and_gemm_route(qd, kinds, static_kind, variant):
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    ctx = qd.default_context()
    rng = np.random.default_rng(sum(variant) * 100 + len(kinds))
    n, batch = 96, 300
    ops = _operators(rng, n, kinds)
    static = None if static_kind is None else _operators(rng, n, static_kind)[0]
    fim = rng.normal(size=n)
    stack = qd.Stack(ctx, ops, static, fim)
    sched = FixedStepSchedule([0.0, 0.03], None, 0.01, _rk4_points)
    table = rng.uniform(-1, 1, (batch, len(sched.times), len(kinds)))
    y0 = crand(rng, batch, n, 1)
    comb, cc = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 2)      # 2: wherever the kernel applies
    assert cc["rhs_combine"]["launches"] == 12 and cc["rhs_gemm"]["launches"] == 0 and cc["rhs_blocks_gemm"]["launches"] == 0, cc
    assert int(cc["combine_info"]["ms"]) == 100 * variant[0] + 10 * variant[1] + variant[2], cc["combine_info"]
    # the default (combine = 1) takes the kernel only where it is the faster formulation (midyn_rk4.inc: plan_uses_combine)
    planes = sum(2 if kd == "c" else 1 for kd in kinds)          # at least three quarters of the plane slots must be in use
    nq = variant[0] + variant[1]                                 # (five and six groups: seven eighths)
    pays = 8 * planes >= 7 * 4 * nq if nq >= 5 else 4 * planes >= 3 * 4 * nq
    _, cd = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 1)
    assert (cd["rhs_combine"]["launches"] == 12) == bool(pays), (variant, planes, cd)
    # 48 pairs of (32 rows, 32 columns): eight waves split every list (LDS tree); stacks of up to two plane groups run their
    # 32-instance variant here (rhs_combine_small_kernel: a sweep too small for 64-instance waves to fill the chip) ...
    assert (int(cc["combine_shape"]["ms"]), int(cc["combine_wave"]["launches"])) == (8, 32), (cc["combine_shape"], cc["combine_wave"])
    # ... and their 64-instance variant under the one-wave-per-SIMD rule (combine_occupancy = 1: four waves per pair)
    wide, cw = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 2, occupancy=1)
    want_wave = 64 if variant[0] + variant[1] <= 2 else 32
    assert (int(cw["combine_shape"]["ms"]), int(cw["combine_wave"]["launches"])) == (4, want_wave), (cw["combine_shape"], cw["combine_wave"])
    assert cw["rhs_combine"]["launches"] == 12 and cw["rhs_gemm"]["launches"] + cw["rhs_blocks_gemm"]["launches"] == 0, cw
    assert_close(wide, comb, 1e-13)
    gemm, cg = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 0)
    assert cg["rhs_combine"]["launches"] == 0 and cg["rhs_gemm"]["launches"] + cg["rhs_blocks_gemm"]["launches"] == 12, cg
    assert_close(comb, gemm, 1e-13)
    d = 1j * fim
    for b in (0, 157, 299):
        def rhs(t, y, b=b):
            row = int(np.argmin(np.abs(np.asarray(sched.times) - t)))
            return orc.generator_rhs(static, ops, table[b, row], d, None, t, y)

        _, yref = orc.rk4_solve(rhs, [0.0, 0.03], y0[b, :, 0], 0.01)
        assert_close(comb[b, -1, :, 0], yref[-1], 1e-11)
    stack.close()


@pytest.mark.parametrize("magnus_order", [1, 2])
def test_combine_route_inside_the_expm_action(qd, magnus_order):
    """scipy_expm through the expm ACTION (Taylor / Chebyshev series of products, fixed_step_solvers.py:80-108,345-363): every
    product of the series runs on the combine kernel; against the GEMM route and, for two instances, the oracle's
    scipy.linalg.expm solve."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points

    ctx = qd.default_context()
    rng = np.random.default_rng(40 + magnus_order)
    n, k, batch = 96, 6, 280

    def antiherm():
        a = crand(rng, n, n) * 0.2
        return -1j * (a + a.conj().T) / 2

    ops = np.array([antiherm() for _ in range(k)])
    static = antiherm()
    fim = rng.normal(size=n)
    stack = qd.Stack(ctx, ops, static, fim)
    sched = FixedStepSchedule([0.0, 0.2], None, 0.05, _magnus_points(magnus_order))
    table = rng.uniform(-1, 1, (batch, len(sched.times), k))
    y0 = crand(rng, n, 1)
    y0 /= np.linalg.norm(y0)
    comb, cc = _solve(qd, stack, "scipy_expm", sched, table, y0, batch, True, 1, magnus_order)
    assert cc["rhs_combine"]["launches"] > 0 and cc["rhs_gemm"]["launches"] == 0 and cc["rhs_blocks_gemm"]["launches"] == 0, cc
    assert int(cc["combine_info"]["ms"]) == 223, cc["combine_info"]          # complex operators + complex static operator
    gemm, cg = _solve(qd, stack, "scipy_expm", sched, table, y0, batch, True, 0, magnus_order)
    assert cg["rhs_combine"]["launches"] == 0
    assert_close(comb, gemm, 1e-12)
    assert np.max(np.abs(np.linalg.norm(comb[:, -1, :, 0], axis=1) - 1.0)) < 1e-10
    d = 1j * fim
    times = np.asarray(sched.times)
    for b in (3, 279):
        def gen(t, b=b):
            return orc.generator_evaluate(static, ops, table[b, int(np.argmin(np.abs(times - t)))], d, None, t)

        _, yref = orc.expm_solve(gen, [0.0, 0.2], y0[:, 0], 0.05, None, magnus_order)
        assert_close(comb[b, -1, :, 0], yref[-1], SOLVE_TOL)
    stack.close()


def test_combine_lists_skip_exactly_zero_blocks_and_matrix_states(qd):
    """Operators that couple only the two halves of the basis (parity-like sectors, exactly-zero diagonal blocks): the combine
    lists hold half of the (32-row group, 16-column block) entries; states are n x 3 matrices per instance (the three
    columns of an instance share its coefficient row).  Against the GEMM route and the oracle."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    ctx = qd.default_context()
    rng = np.random.default_rng(8)
    n, k, batch, m = 256, 6, 100, 3
    ops = _operators(rng, n, "i" * k)
    half = np.arange(n) < n // 2
    ops[:, half[:, None] == half[None, :]] = 0.0          # couple only rows of one half with columns of the other
    stack = qd.Stack(ctx, ops, None, None)
    sched = FixedStepSchedule([0.0, 0.02], None, 0.01, _rk4_points)
    table = rng.uniform(-1, 1, (batch, len(sched.times), k))
    y0 = crand(rng, batch, n, m)
    comb, cc = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 1)
    assert cc["rhs_combine"]["launches"] == 8 and cc["rhs_gemm"]["launches"] == 0 and cc["rhs_blocks_gemm"]["launches"] == 0, cc
    assert cc["combine_info"]["launches"] == (n // 32) * (n // 16) // 2, cc["combine_info"]
    gemm, cg = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 0)
    assert cg["rhs_combine"]["launches"] == 0
    assert_close(comb, gemm, 1e-13)
    for b in (0, 99):
        def rhs(t, y, b=b):
            row = int(np.argmin(np.abs(np.asarray(sched.times) - t)))
            return orc.generator_rhs(None, ops, table[b, row], None, None, t, y)

        _, yref = orc.rk4_solve(rhs, [0.0, 0.02], y0[b], 0.01)
        assert_close(comb[b, -1], yref[-1], 1e-11)
    stack.close()


def test_more_operators_than_the_combine_kernels_cover_take_the_gemm_route(qd):
    """Thirteen COMPLEX operators (four groups of four of each kind; more than 12 per kind only go with one kind alone) and 17
    imaginary ones: the layout is not applicable and the sweep runs on the MFMA GEMM route as before; small sweeps (fewer
    than combine_min_cols state columns) likewise.  Ten imaginary operators at n = 40: the per-launch kernels have the variant,
    the one-launch kernel of small sweeps does not and hands the sweep over."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    ctx = qd.default_context()
    rng = np.random.default_rng(9)
    n = 64
    sched = FixedStepSchedule([0.0, 0.01], None, 0.01, _rk4_points)
    for kinds, batch, want_combine in (("c" * 13, 300, False), ("i" * 17, 300, False), ("iiii", 100, False), ("iiii", 300, True),
                                       ("c" * 9, 300, False), ("c" * 11, 300, True)):
        k = len(kinds)
        stack = qd.Stack(ctx, _operators(rng, n, kinds), None, None)
        table = rng.uniform(-1, 1, (batch, len(sched.times), k))
        _, cc = _solve(qd, stack, "RK4", sched, table, crand(rng, n, 1), batch, True, 1)
        assert (cc["rhs_combine"]["launches"] > 0) == want_combine, (k, batch, cc)
        assert (cc["rhs_gemm"]["launches"] + cc["rhs_blocks_gemm"]["launches"] > 0) != want_combine, (k, batch, cc)
        stack.close()
    stack = qd.Stack(ctx, _operators(rng, 40, "i" * 10), None, None)
    table = rng.uniform(-1, 1, (300, len(sched.times), 10))
    _, cc = _solve(qd, stack, "RK4", sched, table, crand(rng, 40, 1), 300, True, 1, one_launch=2)
    assert cc["rhs_combine"]["launches"] == 4 and int(cc["combine_info"]["ms"]) == 30, cc
    stack.close()


# ---- RK4 sweeps of small systems in ONE launch (csrc/midyn_combine_sweep.h) -----------------------------------------------------
SWEEP_SIZES = ((40, 37), (96, 300), (200, 70), (128, 16), (243, 33))     # (n, instances): n_pad 64 / 128 / 256 / 128 / 256


@pytest.mark.parametrize("idx", range(N_BOTH_KINDS), ids=["%d%d%d" % v for _, _, v in CASES[:N_BOTH_KINDS]])
def test_one_launch_sweep_variant_vs_per_launch_route_and_oracle(qd, idx):
    """Every (NRE4, NIM4, STAT) variant of combine_sweep_rk4_kernel, sizes rotating over the three workgroup shapes; a frame
    diagonal, per-instance initial states, steps of two different sizes and three saved states (fixed_step_solvers.py:406-459
    with a t_eval).  The per-launch kernels of the same formulation must agree to rounding, the oracle's RK4 to 1e-11."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    kinds, static_kind, variant = CASES[idx]
    ctx = qd.default_context()
    rng = np.random.default_rng(700 + idx)
    n, batch = SWEEP_SIZES[idx % len(SWEEP_SIZES)]
    ops = _operators(rng, n, kinds)
    static = None if static_kind is None else _operators(rng, n, static_kind)[0]
    fim = rng.normal(size=n) if idx % 3 else None
    stack = qd.Stack(ctx, ops, static, fim)
    t_eval = [0.0, 0.013, 0.03]
    sched = FixedStepSchedule([0.0, 0.03], t_eval, 0.01, _rk4_points)
    table = rng.uniform(-1, 1, (batch, len(sched.times), len(kinds)))
    y0 = crand(rng, batch, n, 1)
    one, c1 = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 2, one_launch=2)      # 2: wherever it applies
    n_pad = -(-n // 64) * 64
    rt = 2 if n_pad > 128 else 1
    waves = n_pad // (16 * rt) * (2 if n_pad == 64 else 1)
    assert c1["rhs_combine"]["launches"] == 1 and c1["rhs_gemm"]["launches"] == 0 and c1["rhs_blocks_gemm"]["launches"] == 0, c1
    assert (int(c1["combine_sweep"]["launches"]), int(c1["combine_sweep"]["ms"])) == (-(-batch // 16), 10 * waves + rt), c1["combine_sweep"]
    assert int(c1["combine_info"]["ms"]) == 100 * variant[0] + 10 * variant[1] + variant[2], c1["combine_info"]
    per, c0 = _solve(qd, stack, "RK4", sched, table, y0, batch, False, 2, one_launch=0)
    assert c0["rhs_combine"]["launches"] + c0["rhs_gemm"]["launches"] + c0["rhs_blocks_gemm"]["launches"] == 4 * len(sched.step_h), c0
    assert one.shape == per.shape == (batch, sched.n_save, n, 1)      # slots: t_span[0], the three t_eval points, t_span[1]
    assert_close(one, per, 1e-13)
    d = None if fim is None else 1j * fim
    times = np.asarray(sched.times)
    for b in (0, batch // 2, batch - 1):
        def rhs(t, y, b=b):
            return orc.generator_rhs(static, ops, table[b, int(np.argmin(np.abs(times - t)))], d, None, t, y)

        _, yref = orc.rk4_solve(rhs, [0.0, 0.03], y0[b, :, 0], 0.01, t_eval)
        assert_close(one[b, 1:-1, :, 0], yref, 1e-11)
    stack.close()


def test_one_launch_sweep_through_the_solver_with_shared_y0(qd):
    """The product Solver in list mode (solver_classes.py:556-590): a 4-level, 3-site chain (n = 64) in the frame of its
    static Hamiltonian, one y0 for the whole sweep, 45 instances; default options take the one-launch kernel; against the per-launch route and the oracle."""
    from oracle import dynamics_oracle as orc

    levels, sites = 4, 3
    a = np.diag(np.sqrt(np.arange(1, levels)), 1).astype(complex)
    num = a.conj().T @ a
    eye = np.eye(levels)

    def on(op, i):
        out = np.array([[1.0 + 0j]])
        for s_ in range(sites):
            out = np.kron(out, op if s_ == i else eye)
        return out

    h_d = sum((5.0 + 0.1 * i) * on(num, i) - 0.15 * on(num @ (num - eye), i) for i in range(sites))
    for i in range(sites - 1):
        hop = on(a.conj().T, i) @ on(a, i + 1)
        h_d = h_d + 0.02 * (hop + hop.conj().T)
    ops = [0.1 * (on(a, i) + on(a.conj().T, i)) for i in range(sites)]
    solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=h_d)
    ctx = solver.model._ctx
    rng = np.random.default_rng(5)
    batch = 45
    lists = [[qd.Signal(float(rng.uniform(0.3, 1.0)), 5.0 + 0.1 * i, float(rng.uniform(0, 6))) for i in range(sites)]
             for _ in range(batch)]
    y0 = np.zeros(levels**sites, dtype=complex)
    y0[0] = 1.0
    t_span, dt = [0.0, 0.5], 0.01
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        res = solver.solve(t_span=t_span, y0=y0, signals=lists, method="RK4", max_dt=dt)
        c1 = {c: ctx.counters(c) for c in ("rhs_combine", "combine_sweep", "combine_info")}
        ctx.set_option("combine_sweep", 0)
        ctx.reset_counters()
        ref = solver.solve(t_span=t_span, y0=y0, signals=lists, method="RK4", max_dt=dt)
        c0 = ctx.counters("rhs_combine")
    finally:
        ctx.set_option("profile", 0)
        ctx.set_option("combine_sweep", 1)
    assert c1["rhs_combine"]["launches"] == 1 and int(c1["combine_sweep"]["launches"]) == 3, c1
    assert c1["combine_info"]["launches"] <= (64 // 32) * (64 // 16), c1["combine_info"]
    assert c0["launches"] != 1
    for b in range(batch):
        assert_close(res[b].y[-1], ref[b].y[-1], 1e-12)
    a_d, a_ops, d, basis = orc.hamiltonian_model_build(h_d, np.array(ops), -1j * h_d)
    for b in (0, 44):
        def coeffs(t, b=b):
            return np.array([np.real(sg.complex_value(t)) for sg in lists[b]])

        _, yref = orc.solve_generator_model(a_d, a_ops, d, basis, coeffs, t_span, y0, "RK4", dt)
        assert_close(res[b].y[-1], yref[-1], SOLVE_TOL)
    assert np.max([abs(np.linalg.norm(r.y[-1]) - 1.0) for r in res]) < 1e-8


def test_one_launch_sweep_is_taken_where_it_is_faster(qd):
    """Default options (combine_sweep = 1): up to n_pad = 128 every RK4 sweep of a stack with a COMBINE layout runs in one launch;
    above, only when the workgroups of 16 instances fill the chip (midyn_rk4.inc: rk4_combine_sweep_one_launch): 33 and 2000
    instances at n = 200 stay on the per-launch kernels, 4096 instances take the one-launch kernel."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    ctx = qd.default_context()
    rng = np.random.default_rng(77)
    sched = FixedStepSchedule([0.0, 0.02], None, 0.01, _rk4_points)
    for n, batch, want in ((96, 33, True), (40, 300, True), (200, 33, False), (200, 2000, False), (200, 4096, True)):
        stack = qd.Stack(ctx, _operators(rng, n, "iii"), None, rng.normal(size=n))
        table = rng.uniform(-1, 1, (batch, len(sched.times), 3))
        _, cc = _solve(qd, stack, "RK4", sched, table, crand(rng, n, 1), batch, True, 1, one_launch=1)
        assert (cc["rhs_combine"]["launches"] == 1) == want, (n, batch, cc)
        stack.close()
    # 10 .. 16 rows: the persistent one-wave kernel of small systems (class rhs_stream) unless the sweep is large and has a few
    # operators -- then the MFMA kernel (midyn_rk4.inc: midyn_rk4_solve); same results
    stack = qd.Stack(ctx, _operators(rng, 16, "iiii"), None, rng.normal(size=16))
    results = {}
    for batch, want in ((100, False), (2048, True)):
        table = rng.uniform(-1, 1, (2048, len(sched.times), 4))[:batch].copy()
        y0 = crand(np.random.default_rng(5), 16, 1)
        results[batch], cc = _solve(qd, stack, "RK4", sched, table, y0, batch, True, 1, one_launch=1)
        assert (cc["rhs_combine"]["launches"] == 1) == want, (batch, cc)
        one_wave, _ = _solve(qd, stack, "RK4", sched, table, y0, batch, True, 1, one_launch=0)
        assert_close(results[batch], one_wave, 1e-13)
    stack.close()


# ---- scipy_expm (Magnus order 1) sweeps of small systems in ONE launch: combine_sweep_kernel<.., MODE 1> ------------------------
EXPM_CASES = [(40, 37, "iii", None, True), (96, 300, "cccc", "c", True), (200, 70, "iiiiii", "i", True),
              (128, 16, "rrc", "r", False), (243, 33, "ii", None, True), (64, 50, "cccccccc", "c", False)]


@pytest.mark.parametrize("n,batch,kinds,static_kind,framed", EXPM_CASES, ids=[f"n{c[0]}_{c[2]}_{c[3]}" for c in EXPM_CASES])
def test_one_launch_expm_sweep_vs_per_launch_route_and_oracle(qd, n, batch, kinds, static_kind, framed):
    """The expm ACTION of scipy_expm with magnus_order 1 (fixed_step_solvers.py:80-108,345-363) for sweeps of small systems: the
    whole solve as one launch (a stage = a term of the Chebyshev or scaled Taylor series of a step), at the three workgroup
    shapes, anti-Hermitian generators (Chebyshev series) and general ones (scaled Taylor), steps of two sizes, saved states.
    Against the per-launch kernels (same series: 1e-12) and the oracle's scipy.linalg.expm solve."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points

    ctx = qd.default_context()
    rng = np.random.default_rng(900 + n + len(kinds))
    anti = kinds[0] != "r"              # the cases that start with a real-plane operator are general (non-normal) generators

    def make(kind):
        g = _operators(rng, n, kind)[0] * (0.5 if anti else 0.2)
        if not anti:
            return g
        if kind == "i":                 # -i H with H real symmetric
            return 1j * (g.imag + g.imag.T) / 2
        return (g - g.conj().T) / 2

    ops = np.array([make(kd) for kd in kinds])
    static = None if static_kind is None else make(static_kind)
    fim = rng.normal(size=n) if framed else None
    stack = qd.Stack(ctx, ops, static, fim)
    t_eval = [0.0, 0.07, 0.2]
    sched = FixedStepSchedule([0.0, 0.2], t_eval, 0.05, _magnus_points(1))
    table = rng.uniform(-1, 1, (batch, len(sched.times), len(kinds)))
    y0 = crand(rng, batch, n, 1)
    one, c1 = _solve(qd, stack, "scipy_expm", sched, table, y0, batch, False, 2, one_launch=2)
    assert c1["rhs_combine"]["launches"] == 1 and c1["rhs_gemm"]["launches"] == 0 and c1["rhs_blocks_gemm"]["launches"] == 0, c1
    n_pad = -(-n // 64) * 64
    rt = 2 if n_pad > 128 else 1
    waves = n_pad // (16 * rt) * (2 if n_pad == 64 else 1)
    assert (int(c1["combine_sweep"]["launches"]), int(c1["combine_sweep"]["ms"])) == (-(-batch // 16), 10 * waves + rt), c1["combine_sweep"]
    per, c0 = _solve(qd, stack, "scipy_expm", sched, table, y0, batch, False, 2, one_launch=0)
    assert c0["rhs_combine"]["launches"] + c0["rhs_gemm"]["launches"] + c0["rhs_blocks_gemm"]["launches"] > len(sched.step_h), c0
    assert_close(one, per, 1e-12)
    d = None if fim is None else 1j * fim
    times = np.asarray(sched.times)
    for b in (0, batch - 1):
        def gen(t, b=b):
            return orc.generator_evaluate(static, ops, table[b, int(np.argmin(np.abs(times - t)))], d, None, t)

        _, yref = orc.expm_solve(gen, [0.0, 0.2], y0[b, :, 0], 0.05, t_eval, 1)
        assert_close(one[b, 1:-1, :, 0], yref, SOLVE_TOL)
    stack.close()


@pytest.mark.parametrize("method", ["RK4", "scipy_expm"])
def test_one_launch_sweep_of_a_vectorised_lindbladian(qd, method):
    """Open systems are small sweeps too: two three-level transmons with relaxation and dephasing, vectorised (N = 81 rows of the
    superoperator stack, models/lindblad_model.py:436-538), 40 instances with their own drive amplitudes and phases, frame of the
    static Hamiltonian.  The one-launch kernel (default options) against a launch per stage and against the NON-vectorised solver
    (n x n products, row f2: other kernels altogether), which is pinned to the reference's goldens elsewhere."""
    from qiskit_dynamics_amd import workloads

    h_d, ops, freqs = workloads.transmon_chain(3, 2)
    a = np.diag(np.sqrt(np.arange(1, 3)), 1).astype(complex)
    eye = np.eye(3)
    diss = [np.sqrt(0.02) * np.kron(a, eye), np.sqrt(0.03) * np.kron(eye, a), np.sqrt(0.01) * np.kron(a.conj().T @ a, eye)]
    kw = dict(static_hamiltonian=h_d, hamiltonian_operators=ops, static_dissipators=diss, rotating_frame=h_d)
    vec = qd.Solver(vectorized=True, **kw)
    mat = qd.Solver(vectorized=False, **kw)
    ctx = vec.model._ctx
    rng = np.random.default_rng(3)
    batch = 40
    lists = [[qd.Signal(float(rng.uniform(0.5, 3.0)), float(f), float(rng.uniform(0, 6))) for f in freqs] for _ in range(batch)]
    rho0 = np.zeros((9, 9), dtype=complex)
    rho0[0, 0] = 0.7
    rho0[4, 4] = 0.3
    rho0[0, 4] = rho0[4, 0] = 0.2
    t_span, dt = [0.0, 0.4], (0.002 if method == "RK4" else 0.01)
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        yv = rho0.flatten(order="F")          # (array input of a vectorised model: the column-stacked state)
        res = vec.solve(t_span=t_span, y0=yv, signals=lists, method=method, max_dt=dt)
        c1 = {c: ctx.counters(c) for c in ("rhs_combine", "combine_sweep")}
        ctx.set_option("combine_sweep", 0)
        ref = vec.solve(t_span=t_span, y0=yv, signals=lists, method=method, max_dt=dt)
    finally:
        ctx.set_option("profile", 0)
        ctx.set_option("combine_sweep", 1)
    assert c1["rhs_combine"]["launches"] == 1 and int(c1["combine_sweep"]["launches"]) == 3, c1
    for b in range(batch):
        assert_close(res[b].y[-1], ref[b].y[-1], 1e-12)
    assert max(abs(np.trace(r.y[-1].reshape(9, 9, order="F")) - 1.0) for r in res) < 1e-9
    if method == "RK4":
        rm = mat.solve(t_span=t_span, y0=rho0, signals=lists[:6], method="RK4", max_dt=dt)
        for b in range(6):
            assert_close(res[b].y[-1].reshape(9, 9, order="F"), rm[b].y[-1], SOLVE_TOL)


@pytest.mark.parametrize("method", ["RK4", "scipy_expm"])
def test_one_launch_sweep_backwards_in_time_and_matrix_states(qd, method):
    """t_span[1] < t_span[0] (negative steps, fixed_step_solvers.py:639-651) on the one-launch kernel against a launch per
    stage; matrix-valued states (several columns per instance) are not for that kernel and keep the per-launch kernels."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points, _rk4_points

    ctx = qd.default_context()
    rng = np.random.default_rng(31)
    n, batch, kinds = 70, 50, "iiic"
    ops = np.array([(g - g.conj().T) / 2 for g in _operators(rng, n, kinds)])
    static = _operators(rng, n, "c")[0]
    static = (static - static.conj().T) / 2
    stack = qd.Stack(ctx, ops, static, rng.normal(size=n))
    pts = _rk4_points if method == "RK4" else _magnus_points(1)
    sched = FixedStepSchedule([0.3, 0.1], [0.3, 0.22, 0.1], 0.02, pts)
    assert np.all(np.asarray(sched.step_h)[1:-1] < 0)
    table = rng.uniform(-1, 1, (batch, len(sched.times), len(kinds)))
    y0 = crand(rng, batch, n, 1)
    one, c1 = _solve(qd, stack, method, sched, table, y0, batch, False, 2, one_launch=1)
    per, c0 = _solve(qd, stack, method, sched, table, y0, batch, False, 2, one_launch=0)
    assert c1["rhs_combine"]["launches"] == 1 and c0["rhs_combine"]["launches"] != 1, (c1, c0)
    assert_close(one, per, 1e-12)
    y0m = crand(rng, batch, n, 3)
    _, cm = _solve(qd, stack, method, sched, table, y0m, batch, False, 2, one_launch=2)
    assert cm["rhs_combine"]["launches"] + cm["rhs_gemm"]["launches"] + cm["rhs_blocks_gemm"]["launches"] > 1, cm
    stack.close()
