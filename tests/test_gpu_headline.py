"""The HEADLINE route at the HEADLINE shape, directly against the CPU oracle.

`bench.py`'s `value` is produced by `rhs_combine_kernel<0, 2, 0>` (round 4: combine the 8 imaginary operator planes per
instance with MFMAs, apply the result to the state with vector FMAs; csrc/midyn_combine.h) on the sector-grouped stack that
the product `Solver(static_hamiltonian=H_d, hamiltonian_operators=ops, rotating_frame=H_d)` uploads for the 10-qubit model
(n = 1024, k = 8).  These tests run exactly that: the product Solver, the launch counters assert the combine route with the
sector lists (and that no GEMM contraction ran), the final states are compared

  * with `oracle.solve_generator_model` -- plain `eigh` of the whole frame operator, dense frame-basis operators, the
    reference's `U^+ G_d U - diag(d)` static part (models/generator_model.py:281-340, solvers/fixed_step_solvers.py:43-77)
    -- out of the frame basis, 1e-9;
  * with the MFMA GEMM routes on the same stack (`combine=0`: the work-list kernel zgemm_seg_kernel<128,128,...,SPARSE>,
    the default until round 3, and with `skip_zero_blocks=0` the dense kernels), 1e-13 -- and those two with each other
    bit for bit (`np.array_equal`) where neither splits K.

Same file: dense `midyn_expm` at n = 4096 (the size BASELINE configs[3]/[4] name) against `scipy.linalg.expm`, the
function the reference calls (solvers/fixed_step_solvers.py:22,104), ||E - E_ref||_1 / ||E_ref||_1 <= 1e-12.

All of them need a real MI355X (`pytest -m gpu`).
"""
import numpy as np
import pytest
import scipy.linalg

from conftest import assert_close

pytestmark = pytest.mark.gpu

SOLVE_TOL = 1e-9


@pytest.fixture(scope="module")
def qd():
    import qiskit_dynamics_amd as q

    q.default_context()
    return q


@pytest.fixture(scope="module")
def headline(qd):
    """The product Solver of BASELINE configs[1]/[2] and the oracle's independently built model."""
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads

    cfg = workloads.schrodinger_config()          # 10 qubits, n = 1024, k = 8, T = 5, max_dt = 0.005
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
    assert solver.model.stack.n == 1024
    assert solver.model.rotating_frame.sector_labels is not None, "parity sectors of H_d were not found"
    model = orc.hamiltonian_model_build(cfg["h_d"], cfg["ops"], cfg["h_d"])     # plain eigh, dense operators
    return cfg, solver, model


def _signals(qd, cfg, b, k=8):
    from qiskit_dynamics_amd import workloads

    amps, phases = workloads.sweep_parameters(b, k)
    return [qd.Signal(lambda t, a=a: a * np.exp(-((t - cfg["t_final"] / 2) ** 2) / 2.0), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]


def _oracle_final(cfg, model, b, t_span, y0):
    from oracle import dynamics_oracle as orc
    from qiskit_dynamics_amd import workloads
    from threadpoolctl import threadpool_limits

    a_d, a, d, basis = model
    amps, phases = workloads.sweep_parameters(b, 8)

    def coeffs(t):
        return workloads.gaussian_coefficient_table(np.array([t]), amps, phases, cfg["carrier"], cfg["t_final"])[0]

    with threadpool_limits(limits=8):
        _, yref = orc.solve_generator_model(a_d, a, d, basis, coeffs, t_span, y0, "RK4", cfg["max_dt"])
    return yref[-1]


def _run(qd, solver, sweeps, t_span, y0, max_dt, **options):
    """(final states, counters) of one product solve under ctx options (restored afterwards)."""
    ctx = qd.default_context()
    ctx.reset_counters()
    with ctx.options(profile=1, **options):      # every option back to the value it HAD, whatever its default is
        res = solver.solve(t_span=t_span, y0=y0, signals=sweeps, method="RK4", max_dt=max_dt)
    counts = {c: ctx.counters(c) for c in ("rhs_combine", "rhs_blocks_gemm", "rhs_gemm", "sparse_tile", "sparse_list",
                                           "combine_info", "combine_shape")}
    return np.stack([r.y[-1] for r in res]), counts


def test_headline_sweep_4096_instances_combine_route_vs_oracle_and_gemm_routes(qd, headline):
    """BASELINE configs[2] on one GPU, exactly as bench.py times it: all 4096 instances, 20 RK4 steps (80 batched
    evaluations) through the product Solver.  Asserted: every evaluation ran on rhs_combine_kernel<0, 2, 0> (8 imaginary
    planes in two MFMA groups, no static plane: the static operator in its own frame is exactly zero) over the parity-sector
    lists (half of the (32-row group, 16-column block) entries), 8 (row group, column block) pairs per workgroup, no list
    split, and no GEMM contraction was launched; instances 0, 2047 and 4095 equal the oracle (plain eigh) to 1e-9 out of the
    frame basis; ALL 4096 final states agree with the MFMA work-list route (combine=0) to 1e-13, and that route is
    bit-identical to the dense kernels on the same stack."""
    cfg, solver, model = headline
    nb = 4096
    t_span = [2.4, 2.5]
    sweeps = [_signals(qd, cfg, b) for b in range(nb)]
    rng = np.random.default_rng(1024)
    y0 = rng.normal(size=1024) + 1j * rng.normal(size=1024)
    y0 /= np.linalg.norm(y0)
    comb, c1 = _run(qd, solver, sweeps, t_span, y0, cfg["max_dt"])
    assert c1["rhs_combine"]["launches"] == 80, c1
    assert c1["rhs_gemm"]["launches"] == 0 and c1["rhs_blocks_gemm"]["launches"] == 0, "a GEMM contraction ran on the default route"
    assert int(c1["combine_info"]["ms"]) == 20, c1["combine_info"]                 # 100 NRE4 + 10 NIM4 + static planes
    n_pad = solver.model.stack.n_pad
    assert c1["combine_info"]["launches"] == (n_pad // 32) * (n_pad // 16) // 2, c1["combine_info"]   # parity sectors: half
    assert (int(c1["combine_shape"]["launches"]), int(c1["combine_shape"]["ms"])) == (8, 1), c1["combine_shape"]
    lists, c2 = _run(qd, solver, sweeps, t_span, y0, cfg["max_dt"], combine=0)
    assert c2["rhs_blocks_gemm"]["launches"] == 80 and c2["rhs_combine"]["launches"] == 0 and c2["rhs_gemm"]["launches"] == 0, c2
    assert (int(c2["sparse_tile"]["launches"]), int(c2["sparse_tile"]["ms"])) == (128, 128), c2["sparse_tile"]
    assert int(c2["sparse_list"]["ms"]) == 1, "the 4096-instance launch was split over K"
    dense, c0 = _run(qd, solver, sweeps, t_span, y0, cfg["max_dt"], combine=0, skip_zero_blocks=0)
    assert c0["rhs_blocks_gemm"]["launches"] == 0 and c0["rhs_gemm"]["launches"] == 80, c0
    assert np.array_equal(lists, dense), f"work lists vs dense kernels: max|d| = {np.max(np.abs(lists - dense)):.3e}"
    assert_close(comb, lists, 1e-13)
    assert np.max(np.abs(np.linalg.norm(comb, axis=1) - 1.0)) < 1e-10
    for b in (0, 2047, 4095):
        assert_close(comb[b], _oracle_final(cfg, model, b, t_span, y0), SOLVE_TOL)


def test_headline_shard_512_instances_full_length_vs_oracle(qd, headline):
    """The per-GPU shard of the 8-GPU run (512 instances) over ALL 1000 RK4 steps of cfg 3 through the product Solver: the
    combine route with EIGHT waves splitting the list of every (row group, 64-column block) pair and summing through LDS as
    a tree (256 pairs for 1024 SIMDs, two waves per SIMD); instance 300 against the oracle's 1000 steps, every instance for its norm, and the first
    20 steps against the MFMA work-list route (split-K) and the dense kernels, 1e-13."""
    cfg, solver, model = headline
    nb = 512
    sweeps = [_signals(qd, cfg, b) for b in range(nb)]
    full, c1 = _run(qd, solver, sweeps, cfg["t_span"], cfg["y0"], cfg["max_dt"])
    assert c1["rhs_combine"]["launches"] == 4000 and c1["rhs_gemm"]["launches"] == 0 and c1["rhs_blocks_gemm"]["launches"] == 0, c1
    assert (int(c1["combine_shape"]["launches"]), int(c1["combine_shape"]["ms"])) == (1, 8), c1["combine_shape"]
    assert np.max(np.abs(np.linalg.norm(full, axis=1) - 1.0)) < 1e-8
    assert_close(full[300], _oracle_final(cfg, model, 300, cfg["t_span"], cfg["y0"]), SOLVE_TOL)
    short, _ = _run(qd, solver, sweeps, [2.4, 2.5], cfg["y0"], cfg["max_dt"])
    lists, c2 = _run(qd, solver, sweeps, [2.4, 2.5], cfg["y0"], cfg["max_dt"], combine=0)
    assert c2["rhs_blocks_gemm"]["launches"] == 80 and c2["rhs_combine"]["launches"] == 0
    assert (int(c2["sparse_tile"]["launches"]), int(c2["sparse_tile"]["ms"])) == (128, 128), c2["sparse_tile"]
    dense, c0 = _run(qd, solver, sweeps, [2.4, 2.5], cfg["y0"], cfg["max_dt"], combine=0, skip_zero_blocks=0)
    assert c0["rhs_blocks_gemm"]["launches"] == 0 and c0["rhs_gemm"]["launches"] > 0
    assert_close(short, lists, 1e-13)
    assert_close(lists, dense, 1e-13)


@pytest.mark.parametrize("nb", [1024, 2048])
def test_headline_shards_of_2_and_4_gpus_combine_shapes(qd, headline, nb):
    """The shards of the 4- and 2-GPU runs (1024 / 2048 instances), 20 steps in the active pulse window: 1024 instances run
    four waves per (row group, column block) pair (list split 4, two pairs per workgroup), 2048 two waves per pair (four pairs
    per workgroup) -- eight waves per workgroup = two per SIMD, one workgroup per CU either way; both
    against the MFMA work-list route (1e-13) and instance nb - 1 against the oracle."""
    cfg, solver, model = headline
    sweeps = [_signals(qd, cfg, b) for b in range(nb)]
    rng = np.random.default_rng(nb)
    y0 = rng.normal(size=1024) + 1j * rng.normal(size=1024)
    y0 /= np.linalg.norm(y0)
    comb, c1 = _run(qd, solver, sweeps, [2.4, 2.5], y0, cfg["max_dt"])
    assert c1["rhs_combine"]["launches"] == 80 and c1["rhs_blocks_gemm"]["launches"] == 0, c1
    want = (2, 4) if nb == 1024 else (4, 2)
    assert (int(c1["combine_shape"]["launches"]), int(c1["combine_shape"]["ms"])) == want, c1["combine_shape"]
    lists, c2 = _run(qd, solver, sweeps, [2.4, 2.5], y0, cfg["max_dt"], combine=0)
    assert c2["rhs_blocks_gemm"]["launches"] == 80 and c2["rhs_combine"]["launches"] == 0
    assert_close(comb, lists, 1e-13)
    assert_close(comb[nb - 1], _oracle_final(cfg, model, nb - 1, [2.4, 2.5], y0), SOLVE_TOL)


@pytest.mark.parametrize("scale,kind", [(5.0, "antiherm"), (0.05, "antiherm"), (5.0, "general"), (0.05, "general")])
def test_expm_n4096_vs_scipy(qd, scale, kind):
    """Dense expm (Taylor / Paterson-Stockmeyer scaling & squaring on the MFMA zgemm) at n = 4096 -- the size of the
    cfg 4 superoperator and of the cfg 5 generators -- against scipy.linalg.expm: ||E - E_ref||_1 / ||E_ref||_1 <= 1e-12
    (SURVEY 8(d)), unitarity for anti-Hermitian input."""
    from threadpoolctl import threadpool_limits

    n = 4096
    rng = np.random.default_rng(4096 + int(scale * 100) + (kind == "general"))
    a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    if kind == "antiherm":
        a = a - a.conj().T
    a *= scale / np.linalg.norm(a, 1)
    e = qd.default_context().expm(a)
    with threadpool_limits(limits=32):
        ref = scipy.linalg.expm(a)
    assert np.linalg.norm(e - ref, 1) / np.linalg.norm(ref, 1) < 1e-12
    if kind == "antiherm":
        with threadpool_limits(limits=32):
            assert np.linalg.norm(e.conj().T @ e - np.eye(n)) < 1e-12 * n
