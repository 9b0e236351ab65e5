"""CPU-only tests of the host side of the product package: signals -> coefficient tables, the
fixed-step schedule (step-count rule, table rows, save slots), rotating-frame bookkeeping, list-mode
argument handling, and that the C-ABI library loads and exports every symbol of include/midyn.h.
No GPU compute is invoked here."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, assert_close

import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import _lib, solvers, workloads
from qiskit_dynamics_amd.distributed import shard_bounds


def test_signals_match_reference_golden(golden):
    g = golden("signals")
    t = g["t"]
    s_const = qd.Signal(0.37, carrier_freq=1.3, phase=0.4)
    assert_close(s_const(t), g["const"], 0)
    amp, t0, sig, nu, phi = g["gauss_params"]
    s_gauss = qd.Signal(lambda tt: amp * np.exp(-((tt - t0) ** 2) / (2 * sig**2)), nu, phi)
    assert_close(s_gauss(t), g["gauss"], 0)
    dt, st, cf, ph = g["disc_params"]
    d = qd.DiscreteSignal(dt=dt, samples=g["disc_samples"], start_time=st, carrier_freq=cf, phase=ph)
    assert_close(d.envelope(t), g["disc_env"], 0)
    assert_close(d(t), g["disc"], 0)
    assert_close(d(g["disc_edges_t"]), g["disc_edges"], 0)
    ssum = s_const + s_gauss
    assert_close(ssum(t), g["sum"], 0)
    sl = qd.SignalList([s_const, s_gauss, d, ssum, 1.5])
    assert_close(sl(t), g["list"], 0)          # bit-identical table
    assert_close(sl(0.613), g["list_scalar_t"], 0)
    assert_close(sl.table(t), g["list"], 0)
    assert sl.table(t).dtype == np.float64 and sl.table(t).flags.c_contiguous
    assert len(sl) == 5 and s_const.is_constant is False and qd.Signal(2.0).is_constant
    assert_close(qd.SignalList([qd.Signal(2.0), s_gauss]).drift, np.array([2.0, 0.0]), 0)
    with pytest.raises(qd.DynamicsError):
        qd.SignalList(["not a signal"])


def test_fixed_step_sizes_golden(golden):
    g = golden("fixed_step")
    for i in range(int(g["n_sizes"])):
        te = g[f"sizes{i}_teval"] if bool(g[f"sizes{i}_has_teval"]) else None
        t, h, n = solvers.get_fixed_step_sizes(g[f"sizes{i}_tspan"], te, float(g[f"sizes{i}_maxdt"]))
        assert_close(t, g[f"sizes{i}_t"], 0)
        assert_close(h, g[f"sizes{i}_h"], 0)
        assert np.array_equal(n, g[f"sizes{i}_n"])


def test_schedule_rows_follow_reference_time_arithmetic():
    sched = solvers.FixedStepSchedule([0.0, 1.0], [0.25, 0.9], 0.1, solvers._rk4_points)
    # replay the template loop (fixed_step_solvers.py:448-454) and check every table row
    t_list, h_list, n_list = solvers.get_fixed_step_sizes([0.0, 1.0], [0.25, 0.9], 0.1)
    s = 0
    for i, (t0, h, n) in enumerate(zip(t_list[:-1], h_list, n_list)):
        t = t0
        for j in range(int(n)):
            r = sched.step_rows[s]
            assert sched.times[r[0]] == t and sched.times[r[1]] == t + 0.5 * h and sched.times[r[2]] == t + h
            assert sched.step_h[s] == h
            assert sched.step_save[s] == (i + 1 if j == n - 1 else -1)
            t = t + h
            s += 1
    assert s == len(sched.step_h) == int(np.sum(n_list))
    assert sched.n_save == len(t_list)
    assert len(set(sched.times.tolist())) == len(sched.times)  # rows are distinct times
    # consecutive steps inside an interval share the boundary row (t+h of step s == t of step s+1)
    assert sched.step_rows[0][2] == sched.step_rows[1][0]
    t_out, y_out = sched.trim(np.arange(sched.n_save))
    assert_close(t_out, np.array([0.25, 0.9]), 0) and list(y_out) == [1, 2]
    # backwards, zero-length interval, Magnus points
    sb = solvers.FixedStepSchedule([1.0, 0.0], None, 0.3, solvers._magnus_points(2))
    assert np.all(sb.step_h < 0) and sb.step_rows.shape == (4, 3)
    c1 = 0.5 - np.sqrt(3) / 6
    assert sb.times[sb.step_rows[0][0]] == 1.0 + c1 * sb.step_h[0]
    s3 = solvers.FixedStepSchedule([0.0, 0.2], None, 0.1, solvers._magnus_points(3))
    assert s3.times[s3.step_rows[1][1]] == (0.0 + s3.step_h[0]) + 0.5 * s3.step_h[1]
    with pytest.raises(qd.DynamicsError):
        solvers._magnus_points(4)
    with pytest.raises(ValueError):
        solvers.FixedStepSchedule([0.0, 1.0], [1.5], 0.1, solvers._rk4_points)
    with pytest.raises(ValueError):
        solvers.FixedStepSchedule([0.0, 1.0], [0.5, 0.2], 0.1, solvers._rk4_points)


def test_rotating_frame_host():
    rng = np.random.default_rng(34233)
    n = 5
    a = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
    h = a + a.conj().T
    rf = qd.RotatingFrame(h)
    u, d = rf.frame_basis, rf.frame_diag
    assert_close(u @ np.diag(d) @ u.conj().T, -1j * h, 1e-12)   # F = -iH = U diag(d) U^+
    assert np.allclose(d.real, 0)
    rf2 = qd.RotatingFrame(-1j * h)                                # anti-Hermitian input kept
    assert_close(rf2.frame_diag, d, 1e-12)
    op = rng.normal(size=(n, n)) + 0j
    assert_close(rf.operator_out_of_frame_basis(rf.operator_into_frame_basis(op)), op, 1e-12)
    y = rng.normal(size=n) + 0j
    assert_close(rf.state_out_of_frame(0.7, rf.state_into_frame(0.7, y)), y, 1e-12)
    import scipy.linalg as sla
    f = -1j * h
    assert_close(rf.state_into_frame(0.7, y), sla.expm(-0.7 * f) @ y, 1e-12)
    assert_close(rf.operator_into_frame(0.3, op), sla.expm(-0.3 * f) @ op @ sla.expm(0.3 * f), 1e-11)
    assert_close(rf.generator_into_frame(0.3, op), sla.expm(-0.3 * f) @ op @ sla.expm(0.3 * f) - f, 1e-11)
    rd = qd.RotatingFrame(np.array([1.0, 2.0, -0.5]))             # 1-D Hermitian diagonal
    assert rd.frame_basis is None
    assert_close(rd.frame_diag, -1j * np.array([1.0, 2.0, -0.5]), 0)
    dv = rd.vectorized_frame_diag_imag()
    dd = rd.frame_diag.imag
    for r in range(3):
        for c in range(3):
            assert dv[r + 3 * c] == dd[r] - dd[c]
    assert qd.RotatingFrame(None).frame_diag is None
    assert_close(rf.vectorized_frame_basis, np.kron(u.conj(), u), 0)
    with pytest.raises(qd.DynamicsError):
        qd.RotatingFrame(a)


def test_list_mode_argument_handling():
    sig = qd.Signal(1.0, 5.0)
    (ts, ys, ss), multi = solvers._setup_args_lists([0, 1], np.zeros(2), [sig])
    assert not multi and len(ts) == len(ys) == len(ss) == 1
    (ts, ys, ss), multi = solvers._setup_args_lists([0, 1], np.zeros(2), [[sig], [sig], [sig]])
    assert multi and len(ts) == len(ys) == len(ss) == 3 and ys[0] is ys[2]
    (ts, ys, ss), multi = solvers._setup_args_lists([[0, 1], [0, 2]], [np.zeros(2), np.ones(2)], None)
    assert multi and ss == [None, None]
    (ts, ys, ss), multi = solvers._setup_args_lists([0, 1], np.zeros(2), ([sig], None))
    assert not multi and isinstance(ss[0], tuple)
    (ts, ys, ss), multi = solvers._setup_args_lists([0, 1], np.zeros(2), [([sig], None), ([sig], None)])
    assert multi and len(ss) == 2
    with pytest.raises(qd.DynamicsError):
        solvers._setup_args_lists([[0, 1], [0, 2], [0, 3]], [np.zeros(2), np.ones(2)], None)
    with pytest.raises(qd.DynamicsError):
        solvers._setup_args_lists([[[0, 1]]], np.zeros(2), None)


def test_workload_tables_are_bit_identical_to_signal_objects():
    cfg = workloads.schrodinger_config(n_qubits=3, n_drives=3, t_final=5.0)
    times = np.linspace(0, 5, 23)
    amps, phases = workloads.sweep_parameters(17, 3)
    sigs = [qd.Signal(lambda t, a=a: a * np.exp(-((t - 2.5) ** 2) / (2 * 1.0**2)), nu, ph)
            for a, nu, ph in zip(amps, cfg["carrier"], phases)]
    tab = workloads.gaussian_coefficient_table(times, amps, phases, cfg["carrier"], 5.0)
    assert_close(tab, qd.SignalList(sigs).table(times), 0)
    both = workloads.gaussian_coefficient_table(times, np.stack([amps, amps]), np.stack([phases, phases]),
                                                cfg["carrier"], 5.0)
    assert both.shape == (2, 23, 3) and np.array_equal(both[0], tab)
    h_d, ops, nu = workloads.chain_hamiltonian(3, 2)
    assert np.allclose(h_d, h_d.conj().T) and ops.shape == (2, 8, 8)


def test_shard_bounds():
    for n, w in ((4096, 8), (10, 3), (5, 8), (0, 2), (1024, 1)):
        cover = []
        for r in range(w):
            lo, hi = shard_bounds(n, r, w)
            assert 0 <= lo <= hi <= n
            cover += list(range(lo, hi))
        assert cover == list(range(n))
        sizes = [shard_bounds(n, r, w)[1] - shard_bounds(n, r, w)[0] for r in range(w)]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


def test_abi_library_loads_and_exports_every_header_symbol():
    header = open(os.path.join(ROOT, "include", "midyn.h")).read()
    declared = sorted(set(re.findall(r"\b(midyn_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed from include/midyn.h"
    assert sorted(_lib.ABI_SYMBOLS) == declared, "python binding and header disagree"
    assert os.path.exists(_lib.LIB_PATH), "libmidyn.so not built: run __graft_entry__.build()"
    assert _lib.load() is not None      # pre-loads ONE HIP runtime, then dlopens libmidyn.so
    assert _lib.HIP_RUNTIME is not None
    lib = ctypes.CDLL(_lib.LIB_PATH)    # a second handle on the same object: check the raw exports
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by libmidyn.so"
    nbytes = _lib.Stack.packed_bytes(1024, 8, 1)   # pure host arithmetic, no device needed
    assert nbytes >= 9 * 1024 * 1024 * 16
    assert _lib.Stack.packed_bytes(4, 2, 0) >= 2 * 64 * 64 * 16  # n padded to 64


def test_no_cpu_fallback_without_gpu():
    """Without a HIP device the product must fail loudly, not compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(qd.HipLibraryError):
        _lib.Context(0)
    x = np.array([[0, 1], [1, 0]], dtype=complex)
    with pytest.raises(qd.HipLibraryError):
        qd.HamiltonianModel(operators=[x], signals=[qd.Signal(1.0)])


def test_header_is_plain_c_and_symbols_resolve(tmp_path):
    """Compile a C99 consumer of include/midyn.h with gcc and resolve every entry point by dlsym."""
    import subprocess

    assert _lib.load() is not None
    exe = tmp_path / "abi_probe"
    src = os.path.join(ROOT, "tests", "abi_probe.c")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-o", str(exe), src, "-ldl"], check=True)
    p = subprocess.run([str(exe), _lib.HIP_RUNTIME, _lib.LIB_PATH], capture_output=True, text=True)
    assert p.returncode == 0 and "ABI_OK %d symbols" % len(_lib.ABI_SYMBOLS) in p.stdout, p.stdout + p.stderr
    # and the same header as C++
    cpp = tmp_path / "h.cpp"
    # (midyn_complex is `double _Complex` in both languages -- one function type for the C and the C++ view of every
    # entry point, which UBSan's function-type check of a C caller insists on -- with the layout of a (re, im) pair)
    cpp.write_text('#include "%s"\nint main() { midyn_complex z; __real__ z = 1.0; __imag__ z = 2.0;\n'
                   'static_assert(sizeof(midyn_complex) == 16, "(re, im) pair");\n'
                   'const double* p = reinterpret_cast<const double*>(&z); return p[0] == 1.0 && p[1] == 2.0 ? 0 : 1; }\n'
                   % os.path.join(ROOT, "include", "midyn.h"))
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-o", str(tmp_path / "h"), str(cpp)], check=True)
    assert subprocess.run([str(tmp_path / "h")]).returncode == 0


def test_signal_algebra_values_match_reference(golden):
    g = golden("signals")
    t = g["t"]
    s_const = qd.Signal(0.37, carrier_freq=1.3, phase=0.4)
    amp, t0, sig, nu, phi = g["gauss_params"]
    s_gauss = qd.Signal(lambda tt: amp * np.exp(-((tt - t0) ** 2) / (2 * sig**2)), nu, phi)
    dt, st, cf, ph = g["disc_params"]
    d = qd.DiscreteSignal(dt=dt, samples=g["disc_samples"], start_time=st, carrier_freq=cf, phase=ph)
    assert_close((s_const * s_gauss)(t), g["prod_const_gauss"], 1e-15)
    assert_close((d * s_gauss)(t), g["prod_disc_gauss"], 1e-15)
    assert_close((s_gauss * s_gauss)(t), g["prod_gauss_gauss"], 1e-15)
    assert_close((2.5 * s_gauss)(t), g["prod_scalar"], 1e-15)
    assert_close((-s_gauss)(t), g["neg_gauss"], 1e-15)
    assert_close((s_gauss - d)(t), g["diff_gauss_disc"], 1e-15)
    assert_close(s_gauss.conjugate().complex_value(t), g["conj_gauss_complex"], 1e-15)
    ds = qd.DiscreteSignal.from_Signal(s_gauss, dt=0.1, n_samples=20, start_time=0.0)
    assert_close(ds.samples, g["from_signal_samples"], 0)
    assert_close(ds(t), g["from_signal"], 1e-15)
    ds2 = qd.DiscreteSignal.from_Signal(s_gauss, dt=0.1, n_samples=20, start_time=0.0, sample_carrier=True)
    assert_close(ds2(t), g["from_signal_carrier"], 1e-15)
    assert_close((s_const + s_gauss).flatten()(t), g["flatten_sum"], 1e-15)
    assert (qd.Signal(2.0) * qd.Signal(3.0)).components[0].is_constant
    with pytest.raises(qd.DynamicsError):
        s_gauss * "x"
    # DiscreteSignalSum, add_samples, SignalList.flatten
    dss = qd.DiscreteSignalSum(dt=0.4, samples=g["dss_samples"], start_time=-0.3,
                               carrier_freq=np.array([0.5, 1.5, -0.7]), phase=np.array([0.1, -0.2, 0.3]))
    assert_close(dss(t), g["dss"], 1e-15)
    assert_close(dss.complex_value(t), g["dss_complex"], 1e-15)
    assert_close(dss[1](t), g["dss_item1"], 1e-15)
    assert_close(dss[np.array([0, 2])](t), g["dss_slice"], 1e-15)
    assert len(dss) == 3 and isinstance(dss[0], qd.DiscreteSignal)
    ssum = s_const + s_gauss
    dss2 = qd.DiscreteSignalSum.from_SignalSum(ssum, dt=0.1, n_samples=20, start_time=0.0)
    assert_close(dss2.samples, g["dss_from_sum_samples"], 0)
    assert_close(dss2(t), g["dss_from_sum"], 1e-15)
    dss3 = qd.DiscreteSignalSum.from_SignalSum(ssum, dt=0.1, n_samples=20, start_time=0.0, sample_carrier=True)
    assert_close(dss3(t), g["dss_from_sum_carrier"], 1e-15)
    d_add = qd.DiscreteSignal(dt=dt, samples=g["disc_samples"], start_time=st, carrier_freq=cf, phase=ph)
    d_add.add_samples(6, [0.5, -0.25j])
    assert_close(d_add.samples, g["add_samples_samples"], 0)
    assert_close(d_add(t), g["add_samples"], 1e-15)
    with pytest.raises(qd.DynamicsError):
        d_add.add_samples(2, [1.0])
    assert_close(qd.SignalList([s_const, ssum, d, dss]).flatten()(t), g["list_flatten"], 1e-15)
    # a DiscreteSignalSum is a sum of DiscreteSignals for the device table (row f1)
    from qiskit_dynamics_amd.signals import discrete_term_arrays
    tp, par, rng_, smp = discrete_term_arrays([[dss, d]])
    assert tp.tolist() == [0, 3, 4] and par.shape == (4, 4)


def test_discrete_term_arrays_layout():
    """Row f1 host side: CSR flattening of DiscreteSignal / constant terms; callable envelopes are
    not representable on the device."""
    from qiskit_dynamics_amd.signals import DiscreteSignal, Signal, discrete_term_arrays

    smp = np.array([1.0 + 2.0j, 2.0, -0.5j])
    d1 = DiscreteSignal(0.5, smp, start_time=0.25, carrier_freq=0.9, phase=0.2)
    d2 = DiscreteSignal(0.25, np.array([3.0, 4.0]), carrier_freq=1.5)
    inst = [[d1, d1 + d2, 1.5], [d2, Signal(0.3, 0.0, 0.7), d1]]
    term_ptr, params, ranges, samples = discrete_term_arrays(inst)
    assert term_ptr.tolist() == [0, 1, 3, 4, 5, 6, 7]
    assert params.shape == (7, 4) and ranges.shape == (7, 2)
    np.testing.assert_array_equal(params[0], [0.5, 0.25, 0.9, 0.2])
    np.testing.assert_array_equal(params[3], [0.0, 0.0, 0.0, 0.0])           # the constant 1.5
    np.testing.assert_array_equal(params[5], [0.0, 0.0, 0.0, 0.7])           # constant with a phase
    # d1 appears three times and shares ONE sample range
    assert ranges[0].tolist() == ranges[1].tolist() == ranges[6].tolist() == [0, 3]
    np.testing.assert_array_equal(samples[:3], smp)
    assert samples[ranges[3, 0]] == 1.5 and samples[ranges[5, 0]] == 0.3
    # a Python-callable envelope cannot go to the device
    assert discrete_term_arrays([[Signal(lambda t: t, 1.0)]]) is None
    # array-valued carrier (a SignalSum passed as one term is flattened first, so this is fine)
    assert discrete_term_arrays([[d1 + d2]]) is not None


# ---- row f4: host side of the perturbative solvers ----------------------------------------------------
def _load_golden(name):
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))


@pytest.mark.parametrize("kind", ["dyson", "magnus"])
def test_expansion_model_terms_match_reference(kind):
    """The expansion terms (own ODE formulation + series logarithm), their labels, Udt and the Chebyshev
    coefficients against arrays captured from the reference's DysonSolver / MagnusSolver."""
    from qiskit_dynamics_amd import Signal
    from qiskit_dynamics_amd.perturbative import ExpansionModel

    g = _load_golden("perturbative")
    r, sig_w, t_c, dt, nu = g["q1_params"]
    model = ExpansionModel(operators=g["q1_ops"], rotating_frame=g["q1_frame"], dt=dt, carrier_freqs=[nu],
                           chebyshev_orders=[1], expansion_method=kind, expansion_order=6 if kind == "dyson" else 3,
                           integration_method="DOP853", atol=1e-12, rtol=1e-12)
    want_labels = [tuple(int(i) for i in row if i >= 0) for row in g[f"q1_{kind}_labels"]]
    assert model.monomial_labels == want_labels
    np.testing.assert_allclose(model.Udt, g[f"q1_{kind}_udt"], atol=1e-14)
    np.testing.assert_allclose(model.array_coefficients, g[f"q1_{kind}_terms"], atol=2e-11)
    gauss = Signal(lambda t: 1.0 * np.exp(-((t - t_c) ** 2) / (2 * sig_w**2)), carrier_freq=nu)
    np.testing.assert_allclose(model.approximate_signals([gauss], 0.0, 120), g[f"q1_{kind}_cheb_t0"], atol=1e-13)
    np.testing.assert_allclose(model.approximate_signals([gauss], 3.1, 50), g[f"q1_{kind}_cheb_t1"], atol=1e-13)
    # transmon: two operators, one real-only envelope, extra labels
    model = ExpansionModel(operators=g["t3_ops"], rotating_frame=g["t3_frame"], dt=0.02, carrier_freqs=[4.9, 0.0],
                           chebyshev_orders=[1, 0], expansion_method=kind, expansion_order=2,
                           expansion_labels=[[0, 0, 1], [0, 1, 4]], include_imag=[True, False],
                           integration_method="DOP853", atol=1e-12, rtol=1e-12)
    want_labels = [tuple(int(i) for i in row if i >= 0) for row in g[f"t3_{kind}_labels"]]
    assert model.monomial_labels == want_labels
    np.testing.assert_allclose(model.Udt, g[f"t3_{kind}_udt"], atol=1e-13)
    np.testing.assert_allclose(model.array_coefficients, g[f"t3_{kind}_terms"], atol=2e-11)
    # monomials: products of the coefficient entries named by each label
    c = np.arange(1.0, 6.0).reshape(5, 1) * np.array([[1.0, -0.5]])
    mono = model.monomial_table(c)
    assert mono.shape == (2, len(want_labels))
    for j, lab in enumerate(want_labels):
        np.testing.assert_allclose(mono[:, j], np.prod(c[list(lab)], axis=0), rtol=1e-15)


def test_perturbative_solver_argument_errors():
    from qiskit_dynamics_amd.perturbative import DysonSolver, MagnusSolver, complete_labels

    for cls in (DysonSolver, MagnusSolver):
        with pytest.raises(qd.DynamicsError, match="carrier_freqs must have the same length"):
            cls(operators=np.array([[[1.0]], [[2.0]]]), rotating_frame=np.array([[1.0]]), dt=1.0,
                carrier_freqs=np.array([1.0]), chebyshev_orders=np.array([1, 1]))
        with pytest.raises(qd.DynamicsError, match="chebyshev_orders must have the same length"):
            cls(operators=np.array([[[1.0]], [[2.0]]]), rotating_frame=np.array([[1.0]]), dt=1.0,
                carrier_freqs=np.array([1.0, 1.0]), chebyshev_orders=np.array([1, 1, 1]))
    with pytest.raises(qd.DynamicsError):
        complete_labels(3, None, None)
    # closure under sub-multisets + canonical order
    assert complete_labels(3, 1, [[0, 0, 2]]) == [(0,), (1,), (2,), (0, 0), (0, 2), (0, 0, 2)]


def test_rotating_frame_maps_match_reference(golden):
    """RotatingFrame host maps (state / operator / generator into and out of the frame, the vectorised map)
    against values captured from the reference (models/rotating_frame.py:225-582)."""
    g = golden("rotating_frame")
    t = float(g["t"])
    rf = qd.RotatingFrame(g["f"])
    assert_close(rf.state_into_frame(t, g["y"]), g["state_into"], 1e-13)
    assert_close(rf.state_out_of_frame(t, g["y"]), g["state_out"], 1e-13)
    assert_close(rf.operator_into_frame(t, g["op"]), g["op_into"], 1e-13)
    assert_close(rf.operator_out_of_frame(t, g["op"]), g["op_out"], 1e-13)
    assert_close(rf.generator_into_frame(t, g["op"]), g["gen_into"], 1e-13)
    assert_close(rf.generator_out_of_frame(t, g["op"]), g["gen_out"], 1e-13)
    assert_close(rf.vectorized_map_into_frame(t, g["sup"]), g["vec_map"], 1e-12)
    rd = qd.RotatingFrame(g["d"])
    assert_close(rd.state_into_frame(t, g["y"]), g["diag_state_into"], 1e-14)
    assert_close(rd.generator_into_frame(t, g["op"], True, True), g["diag_gen_into"], 1e-14)
    assert_close(rd.generator_out_of_frame(t, g["op"], True, True), g["diag_gen_out"], 1e-14)
    assert_close(rd.vectorized_map_into_frame(t, g["sup"], True, True), g["diag_vec_map"], 1e-14)
    assert_close(qd.RotatingFrame(None).generator_out_of_frame(t, g["op"]), g["none_gen_out"], 0)
    # into then out of the frame is the identity
    assert_close(rf.generator_out_of_frame(t, rf.generator_into_frame(t, g["op"])), g["op"], 1e-12)


def test_rotating_frame_symmetry_sectors():
    """Frame operators whose non-zero pattern splits into connected components are diagonalised sector by sector
    (rotating_frame._eigh_by_sectors): same eigenvalues and ascending order as np.linalg.eigh (the reference's call,
    rotating_frame.py:102-107), unitary basis, but eigenvectors EXACTLY zero outside their sector -- so operators with
    a selection rule between the sectors get exactly-zero blocks in the frame basis."""
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame, SECTOR_MIN_DIM

    cfg = workloads.schrodinger_config(n_qubits=6, n_drives=6, t_final=1.0, max_dt=0.01)   # parity-conserving chain
    h = cfg["h_d"]
    n = h.shape[0]
    assert n >= SECTOR_MIN_DIM
    fr = RotatingFrame(h)
    labels, u, d = fr.sector_labels, fr.frame_basis, fr.frame_diag
    assert labels is not None and sorted(np.bincount(labels).tolist()) == [32, 32]
    w_ref = np.linalg.eigvalsh(h)
    assert np.all(np.diff(d.imag) <= 0) or np.all(np.diff((1j * d).real) >= 0)
    assert np.max(np.abs(np.sort((1j * d).real) - w_ref)) < 1e-12
    assert np.max(np.abs(u.conj().T @ u - np.eye(n))) < 1e-13
    assert np.max(np.abs(u.conj().T @ h @ u - np.diag((1j * d).real))) < 1e-12
    # exact zeros: every eigenvector lives on the computational states of its own parity only
    parity = np.array([bin(i).count("1") & 1 for i in range(n)])
    for a in range(n):
        support = parity[np.flatnonzero(u[:, a])]
        assert support.min() == support.max()
    # a parity-flipping drive has exactly-zero same-sector entries, the static part exactly-zero cross-sector ones
    x0 = fr.operator_into_frame_basis(-1j * cfg["ops"][0])
    st = fr.operator_into_frame_basis(-1j * h)
    same = labels[:, None] == labels[None, :]
    assert np.all(x0[same] == 0) and np.any(x0[~same] != 0)
    assert np.all(st[~same] == 0)
    # out of the frame basis nothing depends on how the eigenvectors were obtained
    y = np.random.default_rng(1).normal(size=n) + 0j
    w2, u2 = np.linalg.eigh(h)
    t = 0.7
    ref = u2 @ (np.exp(-1j * w2 * (-t)) * (u2.conj().T @ y))
    assert np.max(np.abs(fr.state_into_frame(t, y) - ref)) < 1e-12
    # one connected component, a small matrix or a diagonal frame: the plain path, no labels
    rng = np.random.default_rng(0)
    a = rng.normal(size=(40, 40)) + 1j * rng.normal(size=(40, 40))
    assert RotatingFrame(a + a.conj().T).sector_labels is None
    assert RotatingFrame(np.diag([1.0, -1.0, 2.0])).sector_labels is None
    assert RotatingFrame(np.arange(40.0)).sector_labels is None


def test_sector_layout_and_stack_embedding_host_side():
    """The internal layout of a stack with symmetry sectors (models._sector_slots): sectors contiguous in API order,
    each starting on a 16-row block boundary when the padding is affordable (the one that needs most goes last and
    gets none), plain grouping otherwise; and the row gather/scatter of the wrapper (Stack._rows_in/_rows_out) for a
    pure permutation and for a padded embedding.  No device calls."""
    from qiskit_dynamics_amd import _lib
    from qiskit_dynamics_amd.models import SECTOR_ALIGN, _sector_internal_dim, _sector_slots

    rng = np.random.default_rng(5)
    for sizes, n_int in (([512, 512], 1024), ([30, 50], 82), ([50, 30], 82), ([100, 70, 30], 212),
                         ([1, 1, 6, 6, 15, 15, 20], 64), ([3] * 40, 120)):
        labels = np.repeat(np.arange(len(sizes)), sizes)
        rng.shuffle(labels)
        slot = _sector_slots(labels)
        assert _sector_internal_dim(labels) == n_int and slot.max() + 1 == n_int
        assert np.unique(slot).size == slot.size
        padded = n_int > labels.size
        for c in range(len(sizes)):
            mine = slot[labels == c]
            assert np.all(np.diff(mine) == 1)                     # contiguous, API order kept inside the sector
            if padded:
                assert mine[0] % SECTOR_ALIGN == 0
    # the wrapper's row maps (no handle needed)
    for sizes in ([4, 4], [3, 5, 2]):
        labels = np.repeat(np.arange(len(sizes)), sizes)
        rng.shuffle(labels)
        n_api = labels.size
        starts = np.cumsum([0] + [-(-s // 4) * 4 for s in sizes])  # a layout padded to multiples of 4
        fill = list(starts[:-1])
        slot = np.empty(n_api, dtype=np.int64)
        for a, c in enumerate(labels):
            slot[a] = fill[c]
            fill[c] += 1
        st = _lib.Stack.__new__(_lib.Stack)
        st.handle = None
        st.n = int(starts[-1])
        st.set_embedding(slot)
        assert st.n_api == n_api and (st.perm is not None) == (st.n == n_api)
        y = rng.normal(size=(2, n_api, 3)) + 0j
        yi = st._rows_in(y, -2)
        assert yi.shape == (2, st.n, 3) and np.array_equal(yi[:, slot], y)
        mask = np.ones(st.n, dtype=bool)
        mask[slot] = False
        assert np.all(yi[:, mask] == 0)
        assert np.array_equal(st._rows_out(yi, 1), y)
        g = rng.normal(size=(st.n, st.n)) + 0j
        assert np.array_equal(st._rows_out(st._rows_out(g, 0), 1), g[np.ix_(slot, slot)])
    with pytest.raises(_lib.DynamicsError):
        st.set_embedding(np.array([0, 0, 1]))
    with pytest.raises(_lib.DynamicsError):
        st.set_embedding(np.arange(st.n + 1))


def test_result_bookkeeping_fields():
    """`OdeResult.nfev / wall_s / device` of the fixed-step HIP methods: nfev as scipy counts its own
    methods -- 4 RHS evaluations per RK4 step, the generator evaluations per step for the Magnus / expm methods
    (the reference passes scipy's fields through, solvers/scipy_solve_ivp.py:84)."""
    from qiskit_dynamics_amd import solvers as S

    class Ctx:
        device = 3

    class Model:
        _ctx = Ctx()

    assert S._result_extras(Model(), "RK4", 1, 1000, 0.25) == dict(nfev=4000, wall_s=0.25, device="hip:3")
    assert S._result_extras(Model(), "hip_RK4_parallel", 1, 10, 0.0)["nfev"] == 40
    for order in (1, 2, 3):
        assert S._result_extras(Model(), "scipy_expm", order, 20, 0.0)["nfev"] == order * 20
        assert S._result_extras(Model(), "hip_expm_parallel", order, 20, 0.0)["nfev"] == order * 20


def test_result_arrays_in_pinned_blocks_ownership(monkeypatch):
    """binding: large result arrays live in pinned blocks (hipHostMalloc) that return to a cache when the LAST view is collected
    and are handed out again -- never while a view is alive; small results are plain NumPy arrays.  (A fake allocator: no GPU.)"""
    import ctypes
    import gc

    from qiskit_dynamics_amd import _lib as L

    class FakeRuntime:
        def __init__(self):
            self.live, self.freed = {}, []

        def hipHostMalloc(self, pp, n, flags):
            b = ctypes.create_string_buffer(n)
            self.live[ctypes.addressof(b)] = b
            ctypes.cast(pp, ctypes.POINTER(ctypes.c_void_p))[0] = ctypes.addressof(b)
            return 0

        def hipHostFree(self, p):
            self.freed.append(p.value)
            return 0

    rt = FakeRuntime()
    monkeypatch.setattr(L, "_hiprt", rt)
    monkeypatch.setattr(L, "HIP_RUNTIME", "fake")
    monkeypatch.setattr(L, "_pinned_cache", {})
    monkeypatch.setattr(L, "_pinned_cached_bytes", 0)
    shape, nbytes = (16, 2, 4096, 1), 16 * 2 * 4096 * 16
    a = L.result_array(shape)
    assert a.shape == shape and a.dtype == np.complex128 and a.flags.c_contiguous and a.flags.writeable
    a[...] = 1 + 2j
    view = a[1:3]
    b = L.result_array(shape)
    assert not np.shares_memory(a, b)
    addr = a.ctypes.data
    del a
    gc.collect()
    assert view[0, 0, 0, 0] == 1 + 2j and not L._pinned_cache.get(nbytes)          # a view keeps the block
    del view
    gc.collect()
    assert L._pinned_cache[nbytes] == [addr]
    c = L.result_array(shape)
    assert c.ctypes.data == addr                                                    # handed out again
    assert L.result_array((3, 3)).base is None                                      # small: plain NumPy
    monkeypatch.setenv("MIDYN_PINNED_RESULTS", "0")
    assert L.result_array(shape).base is None
    monkeypatch.delenv("MIDYN_PINNED_RESULTS")
    monkeypatch.setattr(L, "_PINNED_CACHE_MAX", 0)                                  # a full cache: the block goes back to the runtime
    del c
    gc.collect()
    assert rt.freed == [addr]
    del b
    gc.collect()


def test_bench_line_is_compact_and_complete():
    """bench.py's ONE stdout line (tools/bench_legs/compact.py): the driver keeps 8 018 characters of stdout, so the line
    must stay below 6 KB whatever the legs put into the full result (round 5's 31 KB line was not parsed).  Inputs: the full
    result dicts of rounds 4 and 5 as committed under profiles/, and the round-5 one with every string blown up."""
    import io
    import json

    from tools.bench_legs import compact

    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "detail"}
    for name in ("r04_bench.json", "r05_bench_last.json"):
        with open(os.path.join(ROOT, "profiles", name)) as f:
            full = json.load(f)
        line = json.dumps(compact.compact_line(full), separators=(",", ":"))
        assert len(line) < compact.MAX_LINE_BYTES, (name, len(line))
        c = json.loads(line)
        assert need <= set(c), need - set(c)
        assert c["value"] == full["value"] and c["ms_per_step"] == full["ms_per_step"] and c["dtype"] == "f64"
        assert set(compact.ROOFLINE_KEYS) <= set(c["roofline"]) and c["roofline"]["bound"] == "mfma"
        assert c["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
        assert c["roofline"]["executed_flops_per_launch"] == pytest.approx(full["roofline"]["executed_mfma_flops_per_launch"], rel=1e-4)
        assert {"value", "unit", "cores", "kind"} <= set(c["cpu_baseline"])
        assert {"workload", "instances_per_gpu", "global_instances"} <= set(c["config"])
        assert "NaN" not in line and "Infinity" not in line
    for key in ("cfg2", "cfg4", "cfg5", "dense_expm", "f2", "f3", "f4", "cfg3_three_numbers", "projected_strong_scaling"):
        assert key in c, key                                   # (the round-5 dict has every leg)
    assert c["cfg5"]["frac"] == full["cfg5"]["roofline"]["frac"] and c["f4"]["magnus"]["value"] == full["perturbative"]["magnus"]["solve_s"]

    def blow_up(o):
        if isinstance(o, dict):
            return {k: blow_up(v) for k, v in o.items()}
        if isinstance(o, list):
            return [blow_up(v) for v in o] * 3
        return o * 40 if isinstance(o, str) else o

    big = blow_up(full)
    big["cfg4"] = {"error": "x" * 5000}
    big["max_norm_deviation"] = float("nan")
    out = io.StringIO()
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        line = compact.emit(big, out, tmp)
        with open(os.path.join(tmp, compact.DETAIL_NAME)) as f:
            assert json.load(f)["roofline"]["note"] == big["roofline"]["note"]      # nothing is lost: the detail file has it all
    assert out.getvalue() == line + "\n" and len(line) < compact.MAX_LINE_BYTES
    c = json.loads(line)
    assert need <= set(c) and c["max_norm_deviation"] is None and len(c["cfg4"]["error"]) <= 120
    assert "NaN" not in line
