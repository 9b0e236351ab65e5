"""CPU oracle: a plain-NumPy restatement of the qiskit-dynamics ODE-RHS hot path.

TEST INFRASTRUCTURE.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this module; the product package `qiskit_dynamics_amd` never does (its
compute path is the HIP library and it fails loudly when that library is missing).

Parity status: PINNED.  Every function below (incl. the row-f4 perturbative step at the end) is checked in `tests/test_oracle_golden.py` against
`tests/golden/*.npz`, which hold (inputs, outputs) captured from the real reference
(`/root/reference/qiskit_dynamics`, v0.6.0) executed in the build container through
`oracle/ref_shim.py` by `oracle/gen_golden.py`.  Exception: `expm_pade` (own restatement of the
published Al-Mohy--Higham scaling-and-squaring algorithm that `scipy.linalg.expm` 1.15.3
implements; the reference calls scipy at `solvers/fixed_step_solvers.py:22,104`) is pinned against
scipy itself, which is the very dependency the reference calls.

All citations are `file:line` relative to `/root/reference/qiskit_dynamics/`.
The restatement is functional (plain arrays in, plain arrays out); it shares no code with the
reference's class hierarchy.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg

# --------------------------------------------------------------------------------------------
# a8  signal coefficients
# --------------------------------------------------------------------------------------------


def signal_sum_value(envelope_values, carrier_freqs, phases, t):
    """Value of one model coefficient  Re[ sum_i f_i(t) exp(i(2 pi nu_i t + phi_i)) ].

    `SignalList` wraps every entry into a `SignalSum` (signals/signals.py:792-803,
    to_SignalSum :1085-1121) so a coefficient is always evaluated by
    `SignalSum.complex_value` (:574-577) followed by `real` (:153-155).

    envelope_values: (..., terms) complex/real envelope samples f_i(t);  carrier_freqs, phases:
    (terms,) ;  t: array of shape (...).
    """
    t = np.asarray(t)
    carrier_arg = 1j * 2 * np.pi * np.asarray(carrier_freqs)  # signals.py:129
    phase_arg = 1j * np.asarray(phases)  # signals.py:140
    exp_phases = np.exp(np.expand_dims(t, -1) * carrier_arg + phase_arg)  # signals.py:576
    return np.real(np.sum(np.asarray(envelope_values) * exp_phases, axis=-1))


def discrete_envelope(samples, dt, start_time, t):
    """Piecewise-constant envelope of a `DiscreteSignal` (signals/signals.py:293-311):
    sample index floor((t-t0)/dt) clipped to [-1, len]; both clip values hit a zero pad."""
    samples = np.asarray(samples)
    pad = np.zeros((1,) + samples.shape[1:], dtype=samples.dtype)
    padded = np.append(samples, pad, axis=0)
    idx = np.clip(np.array((np.asarray(t) - start_time) // dt, dtype=int), -1, len(samples))
    return padded[idx]


# --------------------------------------------------------------------------------------------
# a4  rotating frame construction
# --------------------------------------------------------------------------------------------


def enforce_anti_hermitian(mat, atol=1e-10, rtol=1e-10):
    """Hermitian -> -iH, anti-Hermitian unchanged, else error (models/rotating_frame.py:585-660)."""
    mat = np.asarray(mat)
    if mat.ndim == 1:
        if np.allclose(mat, mat.conj(), atol=atol, rtol=rtol):
            return -1j * mat
        if np.allclose(mat, -mat.conj(), atol=atol, rtol=rtol):
            return mat
    else:
        if np.allclose(mat, mat.conj().T, atol=atol, rtol=rtol):
            return -1j * mat
        if np.allclose(1j * mat, (1j * mat).conj().T, atol=atol, rtol=rtol):
            return mat
    raise ValueError("frame_operator must be either a Hermitian or anti-Hermitian matrix.")


def frame_setup(frame_operator):
    """Return (frame_diag d, frame_basis U or None) (models/rotating_frame.py:59-112).

    1-D input is taken as already diagonal (:87-95); 2-D input is diagonalised with
    `eigh(1j * F)` and `d = -1j * evals` (:102-107).  `None` -> (None, None).
    """
    if frame_operator is None:
        return None, None
    f = np.asarray(frame_operator)
    f = enforce_anti_hermitian(f)
    if f.ndim == 1:
        return f, None
    evals, basis = np.linalg.eigh(1j * f)
    return -1j * evals, basis


def into_frame_basis(op, basis):
    """U^dagger A U for one operator or a (k,n,n) stack (models/rotating_frame.py:163-190)."""
    if basis is None or op is None:
        return op
    return basis.conj().T @ (np.asarray(op) @ basis)


def out_of_frame_basis(op, basis):
    """U A U^dagger (models/rotating_frame.py:192-223)."""
    if basis is None or op is None:
        return op
    return basis @ (np.asarray(op) @ basis.conj().T)


# --------------------------------------------------------------------------------------------
# a3  model build (operators stored in the frame basis, frame subtracted from the static part)
# --------------------------------------------------------------------------------------------


def generator_model_build(static_operator, operators, frame_operator):
    """Frame-basis operator stack of a `GeneratorModel` (models/generator_model.py:125-178,319-365).

    Returns (A_d or None, A (k,n,n) or None, d or None, U or None).  With a frame the static part
    becomes  U^dagger G_d U - diag(d)  (generator_into_frame at t=0, rotating_frame.py:438-474);
    with a frame and no static operator it is  -diag(d)  (generator_model.py:329-334).
    """
    d, basis = frame_setup(frame_operator)
    if static_operator is None:
        a_d = None if d is None else np.diag(-d)
    else:
        a_d = np.asarray(static_operator, dtype=complex)
        if d is not None:
            a_d = into_frame_basis(a_d, basis) - np.diag(d)
    a = None
    if operators is not None:
        a = into_frame_basis(np.asarray(operators, dtype=complex), basis)
    return a_d, a, d, basis


def hamiltonian_model_build(static_operator, operators, frame_operator):
    """`HamiltonianModel` stores -i H (models/hamiltonian_model.py:97-120) then defers to
    `GeneratorModel`."""
    s = None if static_operator is None else -1j * np.asarray(static_operator, dtype=complex)
    o = None if operators is None else -1j * np.asarray(operators, dtype=complex)
    return generator_model_build(s, o, frame_operator)


# --------------------------------------------------------------------------------------------
# a1/a2  operator-stack sum and contraction
# --------------------------------------------------------------------------------------------


def collection_evaluate(a_d, a, coeffs):
    """C = G_d + sum_j c_j G_j  (models/operator_collections.py:101-122; the sum is
    `np.tensordot(coeffs, mats, axes=1)`, arraylias/register_functions/linear_combo.py:30-32)."""
    if a is not None and a_d is not None:
        return np.tensordot(coeffs, a, axes=1) + a_d
    if a is not None:
        return np.tensordot(coeffs, a, axes=1)
    if a_d is not None:
        return a_d
    raise ValueError("operator collection with neither static operator nor operators")


def collection_rhs(a_d, a, coeffs, y):
    """C . y  (models/operator_collections.py:124-134)."""
    return np.matmul(collection_evaluate(a_d, a, coeffs), y)


# --------------------------------------------------------------------------------------------
# a5-a7  rotating-frame generator / RHS evaluation
# --------------------------------------------------------------------------------------------


def state_phase(d, t, y):
    """exp(-d t) o y along axis 0 == `state_into_frame` in the frame basis
    (models/rotating_frame.py:225-257; `state_out_of_frame` is the same with -t, :259-284)."""
    return (np.exp(d * (-t)) * np.asarray(y).T).T


def generator_evaluate(a_d, a, coeffs, d, basis, t, in_frame_basis=True):
    """G(t) = Delta(t) o C(t), Delta_ab = conj(e_a) e_b, e = exp(d t)
    (models/generator_model.py:256-279 -> rotating_frame.py:286-370, lines :350-353);
    leaves the frame basis with U . U^dagger when `in_frame_basis` is False (:361-362)."""
    c = collection_evaluate(a_d, a, coeffs)
    if d is None:
        return c
    e = np.exp(d * t)
    out = c * (e.conj().reshape(-1, 1) * e)
    if not in_frame_basis:
        out = out_of_frame_basis(out, basis)
    return out


def generator_rhs(a_d, a, coeffs, d, basis, t, y, in_frame_basis=True):
    """G(t) y = exp(-d t) o ( C(t) ( exp(d t) o y ) )  (models/generator_model.py:281-316)."""
    y = np.asarray(y)
    if d is None:
        return collection_rhs(a_d, a, coeffs, y)
    out = y
    if not in_frame_basis and basis is not None:
        out = basis.conj().T @ out
    out = state_phase(d, -t, out)  # state_out_of_frame, return_in_frame_basis=True
    out = collection_rhs(a_d, a, coeffs, out)
    out = state_phase(d, t, out)  # state_into_frame, y_in_frame_basis=True
    if not in_frame_basis and basis is not None:
        out = basis @ out
    return out


# --------------------------------------------------------------------------------------------
# a13/a14  vectorised Lindblad
# --------------------------------------------------------------------------------------------


def vec_commutator(a):
    """Column-stacking matrix of X -> -i[A, X]:  -i (I (x) A - A^T (x) I)
    (models/model_utils.py:31-71); vectorised over a leading stack axis."""
    a = np.asarray(a)
    iden = np.eye(a.shape[-1])
    at = np.swapaxes(a, -1, -2)
    return -1j * (np.kron(iden, a) - np.kron(at, iden))


def vec_dissipator(l):
    """Column-stacking matrix of X -> L X L^+ - (L^+ L X + X L^+ L)/2:
    conj(L) (x) L - (I (x) L^+L + (L^+L)^T (x) I)/2   (models/model_utils.py:74-118)."""
    l = np.asarray(l)
    iden = np.eye(l.shape[-1])
    lconj = l.conj()
    ldagl = np.swapaxes(lconj, -1, -2) @ l
    return np.kron(lconj, iden) @ np.kron(iden, l) - 0.5 * (
        np.kron(iden, ldagl) + np.kron(np.swapaxes(ldagl, -1, -2), iden)
    )


def lindblad_model_build(static_hamiltonian, hamiltonian_operators, static_dissipators,
                         dissipator_operators, frame_operator):
    """Frame-basis operator groups of a `LindbladModel` (models/lindblad_model.py:100-212).

    The frame is subtracted from the static Hamiltonian (kept Hermitian: -i, frame, +i; :172-181)
    and all four groups are rotated into the frame basis (:183-199).
    Returns (H_d, H_ops, N_static, L_ops, d, U)."""
    d, basis = frame_setup(frame_operator)
    h_d = None
    if static_hamiltonian is not None:
        g = -1j * np.asarray(static_hamiltonian, dtype=complex)
        if d is not None:
            g = into_frame_basis(g, basis) - np.diag(d)
        h_d = 1j * g
    elif d is not None:
        h_d = 1j * np.diag(-d)

    def fb(x):
        return None if x is None else into_frame_basis(np.asarray(x, dtype=complex), basis)

    return h_d, fb(hamiltonian_operators), fb(static_dissipators), fb(dissipator_operators), d, basis


def vectorized_lindblad_stack(h_d, h_ops, n_static, l_ops):
    """Superoperator stack of `VectorizedLindbladCollection`
    (models/operator_collections.py:860-938): static = vec_comm(H_d) + sum_j vec_diss(N_j);
    operators = [vec_comm(H_j) ; vec_diss(L_j)] (coefficients concatenated ham-then-diss,
    :1051-1061).  Returns (S_d or None, S (k,N,N) or None)."""
    s_d = None
    if h_d is not None:
        s_d = vec_commutator(h_d)
    if n_static is not None:
        nd = np.sum(vec_dissipator(n_static), axis=0)
        s_d = nd if s_d is None else s_d + nd
    parts = []
    if h_ops is not None:
        parts.append(vec_commutator(h_ops))
    if l_ops is not None:
        parts.append(vec_dissipator(l_ops))
    s = None
    if parts:
        s = parts[0] if len(parts) == 1 else np.append(parts[0], parts[1], axis=0)
    return s_d, s


def lindblad_rhs(h_d, h_ops, n_static, l_ops, ham_coeffs, dis_coeffs, d, t, rho):
    """Non-vectorised Lindblad RHS in the frame basis: LindbladCollection.evaluate_rhs
    (models/operator_collections.py:451-567)  (A+B) rho + rho (A-B) + sum N rho N^+ + sum g L rho L^+
    with B = -iH, A = -1/2 sum N^+N - 1/2 sum g_j L_j^+L_j, wrapped by the frame maps of
    LindbladModel.evaluate_rhs (models/lindblad_model.py:510-531)."""
    rho = np.asarray(rho, dtype=complex)
    n = rho.shape[-1]
    if d is not None:
        e = np.exp(d * t)
        rho = rho * (e.reshape(n, 1) * e.conj())          # operator_out_of_frame = into_frame(-t)
    ham = None
    if h_d is not None or h_ops is not None:
        ham = -1j * collection_evaluate(h_d, h_ops, ham_coeffs)
    amat = np.zeros((n, n), dtype=complex)
    both = np.zeros_like(rho)
    if n_static is not None:
        amat = amat - 0.5 * np.sum(np.swapaxes(n_static.conj(), -1, -2) @ n_static, axis=0)
        both = both + np.sum(n_static @ (rho[..., None, :, :] @ np.swapaxes(n_static.conj(), -1, -2)), axis=-3)
    if l_ops is not None:
        amat = amat - 0.5 * np.tensordot(dis_coeffs, np.swapaxes(l_ops.conj(), -1, -2) @ l_ops, axes=1)
        mats = l_ops @ (rho[..., None, :, :] @ np.swapaxes(l_ops.conj(), -1, -2))
        both = both + np.tensordot(dis_coeffs, mats, axes=(-1, -3))
    hm = np.zeros((n, n), dtype=complex) if ham is None else ham
    out = (hm + amat) @ rho + rho @ (amat - hm) + both
    if d is not None:
        out = out * (e.conj().reshape(n, 1) * e)           # operator_into_frame
    return out


def vectorized_frame_diag(d):
    """Diagonal D with  vec(exp(-tF) X exp(tF)) = exp(-D t) o vec(X)  in column stacking:
    D[r + n c] = d_r - d_c.  Equivalent to the N x N Hadamard mask built at
    models/rotating_frame.py:568-577 and to the F-order reshapes at :322-331,:364-368."""
    d = np.asarray(d)
    return (d.reshape(-1, 1) - d.reshape(1, -1)).flatten(order="F")


def vectorized_frame_basis(basis):
    """kron(conj(U), U)  (models/rotating_frame.py:510-520)."""
    return np.kron(basis.conj(), basis)


# --------------------------------------------------------------------------------------------
# a9  fixed-step template, RK4
# --------------------------------------------------------------------------------------------


def merge_t_args(t_span, t_eval=None):
    """solvers/solver_utils.py:46-96."""
    if t_eval is None:
        return np.asarray(t_span, dtype=float)
    t_span = np.array(t_span, dtype=float)
    t_eval = np.array(t_eval, dtype=float)
    if t_eval.ndim > 1:
        raise ValueError("t_eval must be 1 dimensional.")
    if np.min(t_eval) < np.min(t_span) or np.max(t_eval) > np.max(t_span):
        raise ValueError("t_eval entries must lie in t_span.")
    direction = np.sign(t_span[1] - t_span[0])
    if np.any(direction * np.diff(t_eval) < 0.0):
        raise ValueError("t_eval must be ordered according to the direction of integration.")
    return np.append(np.append(t_span[0], t_eval), t_span[1])


def fixed_step_sizes(t_span, t_eval, max_dt):
    """Step-count rule (solvers/fixed_step_solvers.py:616-653): per interval
    n = int(|dt/max_dt|); n = 1 if 0; n += 1 if |dt/n|/max_dt > 1 + 1e-15; h = dt/n."""
    t_list = np.array(merge_t_args(t_span, t_eval))
    delta = np.diff(t_list)
    n_steps = np.abs(delta / max_dt).astype(int)
    for i, (dt_i, n_i) in enumerate(zip(delta, n_steps)):
        if n_i == 0:
            n_steps[i] = 1
        elif np.abs(dt_i / n_i) / max_dt > 1 + 1e-15:
            n_steps[i] = n_i + 1
    return t_list, np.array(delta / n_steps), n_steps


def fixed_step_template(take_step, t_span, y0, max_dt, t_eval=None):
    """Time loop (solvers/fixed_step_solvers.py:406-459) + `trim_t_results`
    (solvers/solver_utils.py:99-119).  Returns (t, y) with y.shape == (len(t), *y0.shape)."""
    y0 = np.asarray(y0)
    t_list, h_list, n_list = fixed_step_sizes(t_span, t_eval, max_dt)
    ys = [y0]
    for t0, h, n in zip(t_list, h_list, n_list):
        y = ys[-1]
        t = t0
        for _ in range(int(n)):
            y = take_step(t, y, h)
            t = t + h
        ys.append(y)
    ys = np.asarray(ys)
    if t_eval is not None:
        return t_list[1:-1], ys[1:-1]
    return t_list, ys


def rk4_step(rhs, t, y, h):
    """Classic RK4 exactly as written at solvers/fixed_step_solvers.py:62-73."""
    h2 = 0.5 * h
    th = t + h2
    k1 = rhs(t, y)
    k2 = rhs(th, y + h2 * k1)
    k3 = rhs(th, y + h2 * k2)
    k4 = rhs(t + h, y + h * k3)
    return y + (1.0 / 6) * h * (k1 + 2 * k2 + 2 * k3 + k4)


def rk4_solve(rhs, t_span, y0, max_dt, t_eval=None):
    return fixed_step_template(lambda t, y, h: rk4_step(rhs, t, y, h), t_span, y0, max_dt, t_eval)


# --------------------------------------------------------------------------------------------
# a10/a11  Magnus propagator and expm
# --------------------------------------------------------------------------------------------


def _comm(a, b):
    return a @ b - b @ a


def magnus_terms(generator, t0, h, order):
    """Omega_m for m = 1, 2, 3 (solvers/fixed_step_solvers.py:321-392)."""
    if order == 1:
        return generator(t0 + (h / 2)) * h
    if order == 2:
        c1 = 0.5 - np.sqrt(3) / 6
        c2 = 0.5 + np.sqrt(3) / 6
        p2 = np.sqrt(3) / 12
        g1 = generator(t0 + c1 * h)
        g2 = generator(t0 + c2 * h)
        return h * (g1 + g2) / 2 + p2 * (h**2) * _comm(g2, g1)
    if order == 3:
        d1 = 0.5 - np.sqrt(15) / 10
        d3 = 0.5 + np.sqrt(15) / 10
        c0 = np.sqrt(15) / 3
        c1 = 10.0 / 3
        g1 = generator(t0 + d1 * h)
        g2 = generator(t0 + 0.5 * h)
        g3 = generator(t0 + d3 * h)
        a1 = h * g2
        a2 = c0 * h * (g3 - g1)
        a3 = c1 * h * (g3 - 2 * g2 + g1)
        comm1 = _comm(a1, a2)
        comm2 = _comm(2 * a3 + comm1, a1) / 60
        return a1 + (a3 / 12) + _comm(-20 * a1 - a3 + comm1, a2 + comm2) / 240
    raise ValueError("Only magnus_order 1, 2, and 3 are supported.")


def expm_solve(generator, t_span, y0, max_dt, t_eval=None, magnus_order=1, expm=scipy.linalg.expm):
    """`scipy_expm_solver` (solvers/fixed_step_solvers.py:80-108): y <- expm(Omega_m) @ y."""

    def step(t, y, h):
        return expm(magnus_terms(generator, t, h, magnus_order)) @ y

    return fixed_step_template(step, t_span, y0, max_dt, t_eval)


# Pade coefficient tables and theta_m thresholds (Higham 2005 / Al-Mohy & Higham 2009, Table 3.1).
_PADE_B = {
    3: (120.0, 60.0, 12.0, 1.0),
    5: (30240.0, 15120.0, 3360.0, 420.0, 30.0, 1.0),
    7: (17297280.0, 8648640.0, 1995840.0, 277200.0, 25200.0, 1512.0, 56.0, 1.0),
    9: (17643225600.0, 8821612800.0, 2075673600.0, 302702400.0, 30270240.0, 2162160.0,
        110880.0, 3960.0, 90.0, 1.0),
    13: (64764752532480000.0, 32382376266240000.0, 7771770303897600.0, 1187353796428800.0,
         129060195264000.0, 10559470521600.0, 670442572800.0, 33522128640.0, 1323241920.0,
         40840800.0, 960960.0, 16380.0, 182.0, 1.0),
}
_THETA = {3: 1.495585217958292e-002, 5: 2.539398330063230e-001, 7: 9.504178996162932e-001,
          9: 2.097847961257068e+000, 13: 4.25}


def expm_choose(norm1):
    """(Pade order m, squarings s) from the 1-norm -- the order/scaling rule the HIP path uses.
    Classic Higham-2005 selection on ||A||_1 (scipy additionally sharpens the choice with
    d_p = ||A^p||^(1/p) estimates, so scipy may pick a smaller m/s; both are backward stable to
    unit roundoff, see DESIGN.md 'expm parity')."""
    for m in (3, 5, 7, 9):
        if norm1 <= _THETA[m]:
            return m, 0
    s = 0
    if norm1 > _THETA[13]:
        s = max(0, int(math.ceil(math.log2(norm1 / _THETA[13]))))
    return 13, s


def expm_pade(a, force_m=None, force_s=None):
    """Scaling-and-squaring Pade approximant r_m(A / 2^s)^(2^s)  (Higham 2005 Alg. 2.3).
    This is the algorithm the HIP `midyn_expm` implements (same m/s rule, same U/V splitting,
    LU solve with partial pivoting for (V-U) X = (V+U))."""
    a = np.asarray(a, dtype=complex)
    n = a.shape[0]
    m, s = expm_choose(np.linalg.norm(a, 1))
    if force_m is not None:
        m = force_m
    if force_s is not None:
        s = force_s
    a = a / (2.0**s)
    b = _PADE_B[m]
    iden = np.eye(n, dtype=complex)
    a2 = a @ a
    if m == 13:
        a4 = a2 @ a2
        a6 = a4 @ a2
        u = a @ (a6 @ (b[13] * a6 + b[11] * a4 + b[9] * a2)
                 + b[7] * a6 + b[5] * a4 + b[3] * a2 + b[1] * iden)
        v = (a6 @ (b[12] * a6 + b[10] * a4 + b[8] * a2)
             + b[6] * a6 + b[4] * a4 + b[2] * a2 + b[0] * iden)
    else:
        pows = [iden, a2]
        for _ in range(2, m // 2 + 1):
            pows.append(pows[-1] @ a2)
        usum = sum(b[2 * j + 1] * pows[j] for j in range(m // 2 + 1))
        v = sum(b[2 * j] * pows[j] for j in range(m // 2 + 1))
        u = a @ usum
    x = np.linalg.solve(v - u, v + u)
    for _ in range(s):
        x = x @ x
    return x


# --------------------------------------------------------------------------------------------
# a12  frame-basis I/O around a solve
# --------------------------------------------------------------------------------------------


def y0_into_frame_basis(y0, basis, kind):
    """solvers/solver_functions.py:376-415.  kind in {"generator", "lindblad_vec", "lindblad"}."""
    if basis is None:
        return y0
    if kind == "lindblad_vec":
        return vectorized_frame_basis(basis).conj().T @ y0
    if kind == "lindblad":
        return into_frame_basis(y0, basis)
    return basis.conj().T @ y0


def results_out_of_frame_basis(ys, basis, kind, y0_ndim):
    """solvers/solver_functions.py:418-450 (the transpose dance for 1-D states included)."""
    if basis is None:
        return ys
    if y0_ndim == 1:
        ys = ys.T
    if kind == "lindblad_vec":
        ys = vectorized_frame_basis(basis) @ ys
    elif kind == "lindblad":
        ys = out_of_frame_basis(ys, basis)
    else:
        ys = basis @ ys
    if y0_ndim == 1:
        ys = ys.T
    return ys


# --------------------------------------------------------------------------------------------
# whole-solve helpers used by tests / bench cpu_baseline
# --------------------------------------------------------------------------------------------


def solve_generator_model(a_d, a, d, basis, coeff_fn, t_span, y0, method="RK4", max_dt=None,
                          t_eval=None, magnus_order=1, in_frame_basis=False, kind="generator"):
    """`solve_lmde(model, ...)` for a model given by its frame-basis stack
    (solvers/solver_functions.py:220-373): y0 -> frame basis, integrate with the model in the
    frame basis, results -> out of the frame basis.  `coeff_fn(t)` returns the real (k,) vector."""
    y0 = np.asarray(y0, dtype=complex)
    if not in_frame_basis:
        y0 = y0_into_frame_basis(y0, basis, kind)

    def rhs(t, y):
        return generator_rhs(a_d, a, None if a is None else coeff_fn(t), d, None, t, y)

    def gen(t):
        return generator_evaluate(a_d, a, None if a is None else coeff_fn(t), d, None, t)

    if method == "RK4":
        t, y = rk4_solve(rhs, t_span, y0, max_dt, t_eval)
    elif method == "scipy_expm":
        t, y = expm_solve(gen, t_span, y0, max_dt, t_eval, magnus_order)
    else:
        raise ValueError(f"Method {method} not supported by the oracle.")
    if not in_frame_basis:
        y = results_out_of_frame_basis(y, basis, kind, y0.ndim)
    return t, y


# --------------------------------------------------------------------------------------------
# f4  perturbative (Dyson / Magnus expansion) solvers -- the RUN-TIME step; the expansion terms
#     themselves are inputs here (the product computes them in qiskit_dynamics_amd/perturbative.py
#     and is checked against the reference's terms directly, tests/golden/perturbative.npz)
# --------------------------------------------------------------------------------------------


def frame_state_map(d, basis, t, y, into=True):
    """`RotatingFrame.state_into_frame` (into) / `state_out_of_frame` for a state in the LAB basis
    (models/rotating_frame.py:225-284): U (exp(-/+ d t) o (U^+ y))."""
    y = np.asarray(y, dtype=complex)
    if d is None:
        return y
    yb = y if basis is None else basis.conj().T @ y
    yb = (np.exp(d * (-t if into else t)) * yb.T).T
    return yb if basis is None else basis @ yb


def chebyshev_dct(degree, dt):
    """DCT matrix and shifted Chebyshev points on [0, dt]
    (solvers/perturbative_solvers/expansion_model.py:518-551)."""
    from numpy.polynomial.chebyshev import chebpts1, chebvander

    order = degree + 1
    xcheb = chebpts1(order)
    shifted = 0.5 * (dt * xcheb + dt)
    mat = chebvander(xcheb, degree).T
    mat[0] /= order
    mat[1:] /= 0.5 * order
    return mat, shifted


def signal_envelope_dct(complex_value, reference_freq, degree, t0, dt, n_intervals):
    """Chebyshev coefficients of the envelope of one signal relative to `reference_freq` on each of
    `n_intervals` intervals (expansion_model.py:457-515): (degree+1, n_intervals) complex."""
    t_vals = t0 + np.arange(n_intervals) * dt
    phase_arg = -1j * 2 * np.pi * reference_freq
    final_shift = np.exp(-phase_arg * t_vals)
    mat, xcheb = chebyshev_dct(degree, dt)
    x_vals = np.add.outer(xcheb, t_vals)
    return (mat @ (complex_value(x_vals) * np.exp(phase_arg * x_vals))) * np.expand_dims(final_shift, 0)


def signal_list_envelope_dct(complex_values, reference_freqs, degrees, t0, dt, n_intervals, include_imag=None):
    """Real coefficient rows of all signals: Re rows, then Im rows when included
    (expansion_model.py:410-454)."""
    if include_imag is None:
        include_imag = [True] * len(complex_values)
    rows = []
    for cv, fr, dg, inc in zip(complex_values, reference_freqs, degrees, include_imag):
        c = signal_envelope_dct(cv, fr, dg, t0, dt, n_intervals)
        rows.append(c.real)
        if inc:
            rows.append(c.imag)
    return np.concatenate(rows, axis=0)


def monomials(labels, c):
    """c^I = prod_{i in I} c_i for each label row (padded with -1)
    (perturbation/array_polynomial.py:547-601); c is (n_vars,) or (n_vars, T)."""
    c = np.asarray(c)
    out = []
    for lab in np.asarray(labels):
        v = np.ones(c.shape[1:], dtype=c.dtype)
        for i in lab:
            if i >= 0:
                v = v * c[i]
        out.append(v)
    return np.asarray(out)


def array_polynomial_eval(terms, labels, c, constant_term=None):
    """sum_I c^I A_I (+ constant) (perturbation/array_polynomial.py:524-544)."""
    val = np.tensordot(np.asarray(terms), monomials(labels, c), axes=(0, 0))
    return val if constant_term is None else constant_term + val


def perturbative_solve(kind, terms, labels, udt, d, basis, cheb_coeffs, y0, t0, n_steps, dt):
    """`_perturbative_solve` (solvers/perturbative_solvers/perturbative_solver.py:172-192) with the
    Dyson step `(Udt + sum c^I Udt D_I) y` (dyson_solver.py:204-207; the terms already carry Udt,
    expansion_model.py:149-158) or the Magnus step `Udt expm(sum c^I O_I) y` (magnus_solver.py:122-125)."""
    n = np.asarray(udt).shape[0]
    u0 = frame_state_map(d, basis, t0, np.eye(n, dtype=complex), into=False)
    uf = frame_state_map(d, basis, t0 + n_steps * dt, np.eye(n, dtype=complex), into=True)
    y = u0 @ np.asarray(y0, dtype=complex)
    for k in range(n_steps):
        c = cheb_coeffs[:, k]
        if kind == "dyson":
            y = array_polynomial_eval(terms, labels, c, constant_term=udt) @ y
        else:
            y = udt @ scipy.linalg.expm(array_polynomial_eval(terms, labels, c)) @ y
    return uf @ y
