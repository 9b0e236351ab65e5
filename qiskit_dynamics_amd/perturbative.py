"""Perturbative (Dysolve-type) solvers: ``DysonSolver`` / ``MagnusSolver``.

Mirrors ``solvers/perturbative_solvers/{expansion_model,dyson_solver,magnus_solver,perturbative_solver}.py``
of the reference: the generator ``G(t) = sum_j Re[f_j(t) e^{i 2 pi nu_j t}] G_j`` is expanded, on every
step ``[t, t + dt]``, in Chebyshev coefficients of the envelopes; the step propagator is a truncated
Dyson series or the exponential of a truncated Magnus expansion, both POLYNOMIALS in those
coefficients with pre-computed matrix coefficients.

Split of the work
  * model construction (once; host, like the model build of rows a3/a4): the symmetric Dyson terms
    D_I(dt) are obtained by integrating  dD_I/dt = sum_{p in I} A_p(t) D_{I - p}(t)  (D_{} = 1) in the
    frame basis with SciPy's ``solve_ivp``; Magnus terms follow from the series logarithm.  This is
    an own formulation (the reference: ``perturbation/solve_lmde_perturbation.py`` +
    ``perturbation/dyson_magnus.py``); its results are checked against the reference's terms.
  * every solve (the hot step): the Chebyshev coefficients of the envelopes on the host (the signals are
    Python callables), then ``midyn_expansion_solve_coeffs`` on the device -- monomials of the
    coefficients, one GEMM that evaluates the polynomial for all steps, batched expm (Magnus), tree
    product of the step propagators.
"""
from __future__ import annotations

from itertools import combinations_with_replacement
from typing import List, Optional

import numpy as np
from numpy.polynomial.chebyshev import chebpts1, chebvander
from scipy.integrate import solve_ivp
from scipy.integrate._ivp.ivp import OdeResult

from . import _lib
from ._lib import DynamicsError
from .rotating_frame import RotatingFrame
from .signals import Signal, SignalList


# -------------------------------------------------------------------------------------------------
# Chebyshev approximation of signal envelopes (expansion_model.py:410-551)
# -------------------------------------------------------------------------------------------------
_DCT_CACHE = {}


def _construct_dct(degree: int, dt: float):
    key = (int(degree), float(dt))
    hit = _DCT_CACHE.get(key)
    if hit is None:
        order = degree + 1
        xcheb = chebpts1(order)
        shifted = 0.5 * (dt * xcheb + dt)
        mat = chebvander(xcheb, degree).T
        mat[0] /= order
        mat[1:] /= 0.5 * order
        if len(_DCT_CACHE) > 64:
            _DCT_CACHE.clear()
        hit = _DCT_CACHE[key] = (mat, shifted)
    return hit


def _envelope_in_reference_frame(signal: Signal, reference_freq: float, x_vals: np.ndarray) -> np.ndarray:
    """``signal.complex_value(x) * exp(-i 2 pi reference_freq x)``.  For a plain carrier signal the two phase factors are ONE
    factor ``exp(i (2 pi (nu - reference_freq) x + phase))`` -- and no factor at all when the expansion's reference frequency IS the
    signal's carrier (the usual case: expansion_model.py's ``carrier_freqs``); that halves the host time of a solve's signal
    evaluation and drops the rounding of two cancelling 2 pi nu x arguments."""
    if type(signal).complex_value is Signal.complex_value and np.ndim(signal.carrier_freq) == 0 and np.ndim(signal.phase) == 0:
        env = signal.envelope(x_vals)
        dnu, ph = float(signal.carrier_freq) - float(reference_freq), float(signal.phase)
        if dnu == 0.0 and ph == 0.0:
            return np.asarray(env, dtype=complex)
        return env * np.exp(1j * (2 * np.pi * dnu * x_vals + ph))
    return signal.complex_value(x_vals) * np.exp(-1j * 2 * np.pi * reference_freq * x_vals)


def signal_envelope_dct(signal: Signal, reference_freq: float, degree: int, t0: float, dt: float,
                        n_intervals: int) -> np.ndarray:
    """(degree+1, n_intervals) complex Chebyshev coefficients of the envelope of ``signal`` relative
    to ``reference_freq`` on consecutive intervals of length ``dt`` starting at ``t0``."""
    t_vals = t0 + np.arange(n_intervals) * dt
    final_phase_shift = np.exp((1j * 2 * np.pi * reference_freq) * t_vals)
    mat, xcheb = _construct_dct(degree, dt)
    x_vals = np.add.outer(xcheb, t_vals)
    shifted = _envelope_in_reference_frame(signal, reference_freq, x_vals)
    return (mat @ shifted) * np.expand_dims(final_phase_shift, axis=0)


def signal_list_envelope_dct(signals, reference_freqs, degrees, t0, dt, n_intervals, include_imag=None):
    """Real coefficient rows for all signals: real parts, then imaginary parts where included.  The SAME signal object with the
    same reference frequency and degree (two drives that share one pulse) is evaluated once."""
    if include_imag is None:
        include_imag = [True] * len(signals)
    rows, seen = [], {}
    for sig, freq, deg, inc in zip(signals, reference_freqs, degrees, include_imag):
        key = (id(sig), float(freq), int(deg))
        c = seen.get(key)
        if c is None:
            c = seen[key] = signal_envelope_dct(sig, freq, deg, t0, dt, n_intervals)
        rows.append(c.real)
        if inc:
            rows.append(c.imag)
    return np.concatenate(rows, axis=0)


# -------------------------------------------------------------------------------------------------
# index multisets (labels are sorted tuples of perturbation indices)
# -------------------------------------------------------------------------------------------------
def _clean_label(label) -> tuple:
    if isinstance(label, dict):
        label = [k for k, v in label.items() for _ in range(int(v))]
    label = tuple(sorted(int(i) for i in label))
    if any(i < 0 for i in label):
        raise DynamicsError("Only Multisets whose entries are non-negative integers are accepted.")
    return label


def _submultisets(label: tuple):
    """All non-empty proper sub-multisets J of ``label`` with their complements ``label - J``."""
    distinct = sorted(set(label))
    counts = [label.count(e) for e in distinct]
    out = []

    def rec(pos, chosen):
        if pos == len(distinct):
            sub = tuple(e for e, c in zip(distinct, chosen) for _ in range(c))
            if 0 < len(sub) < len(label):
                comp = tuple(e for e, c, full in zip(distinct, chosen, counts) for _ in range(full - c))
                out.append((sub, comp))
            return
        for c in range(counts[pos] + 1):
            rec(pos + 1, chosen + [c])

    rec(0, [])
    return out


def complete_labels(n_perturbations: int, expansion_order: Optional[int], expansion_labels) -> List[tuple]:
    """All multisets up to ``expansion_order`` plus ``expansion_labels``, closed under sub-multisets, in
    the canonical order (size, then lexicographic) of the reference
    (perturbation/perturbation_utils.py:31-96, multiset_utils.py:49-93)."""
    if expansion_order is None and expansion_labels is None:
        raise DynamicsError("At least one of expansion_order or expansion_labels must be specified.")
    labels = set()
    if expansion_order is not None:
        for size in range(1, int(expansion_order) + 1):
            labels.update(combinations_with_replacement(range(n_perturbations), size))
    for lab in expansion_labels or []:
        lab = _clean_label(lab)
        if lab and max(lab) >= n_perturbations:
            raise DynamicsError("expansion_labels refer to a perturbation index that does not exist.")
        if lab:
            labels.add(lab)
            labels.update(sub for sub, _ in _submultisets(lab))
    return sorted(labels, key=lambda x: (len(x), x))


def compute_monomials(labels: List[tuple], c: np.ndarray) -> np.ndarray:
    """``c^I = prod_{i in I} c_i`` for every label; ``c`` is (n_vars, T) -> (len(labels), T).  Higher
    orders reuse lower ones (array_polynomial.py:547-601)."""
    c = np.asarray(c, dtype=float)
    index = {lab: i for i, lab in enumerate(labels)}
    out = np.empty((len(labels),) + c.shape[1:], dtype=float)
    for i, lab in enumerate(labels):
        if len(lab) == 1:
            out[i] = c[lab[0]]
        else:
            rest = lab[1:]
            out[i] = c[lab[0]] * (out[index[rest]] if rest in index else np.prod(c[list(rest)], axis=0))
    return out


# -------------------------------------------------------------------------------------------------
# expansion terms
# -------------------------------------------------------------------------------------------------
def _dyson_terms(pert_funcs, n, labels, dt, integration_method, **kwargs) -> np.ndarray:
    """Symmetric Dyson terms D_I(dt) for the (sub-multiset closed, canonically ordered) ``labels``."""
    index = {lab: i + 1 for i, lab in enumerate(labels)}  # slot 0 holds D_{} = identity
    index[()] = 0
    n_pert = len(pert_funcs)
    dst = [[] for _ in range(n_pert)]
    src = [[] for _ in range(n_pert)]
    for lab in labels:
        for p in sorted(set(lab)):
            rest = list(lab)
            rest.remove(p)
            dst[p].append(index[lab])
            src[p].append(index[tuple(rest)])
    dst = [np.asarray(x, dtype=int) for x in dst]
    src = [np.asarray(x, dtype=int) for x in src]
    n_lab = len(labels)
    ident = np.eye(n, dtype=complex)

    def rhs(t, flat):
        d = np.empty((n_lab + 1, n, n), dtype=complex)
        d[0] = ident
        d[1:] = flat.reshape(n_lab, n, n)
        out = np.zeros((n_lab + 1, n, n), dtype=complex)
        for p in range(n_pert):
            if dst[p].size:
                out[dst[p]] += pert_funcs[p](t) @ d[src[p]]   # each label appears once per distinct p
        return out[1:].reshape(-1)

    sol = solve_ivp(rhs, (0.0, dt), np.zeros(n_lab * n * n, dtype=complex), method=integration_method or "DOP853",
                    **kwargs)
    if not sol.success:
        raise DynamicsError(f"integration of the perturbation terms failed: {sol.message}")
    return sol.y[:, -1].reshape(n_lab, n, n)


def _magnus_from_dyson(labels: List[tuple], dyson: np.ndarray) -> np.ndarray:
    """Symmetric Magnus terms from the series logarithm  Omega = sum_m (-1)^{m+1}/m (U - 1)^m,
    U - 1 = sum_I c^I D_I:  the coefficient of c^I in (U - 1)^m is the sum over ordered
    decompositions I = I_1 + ... + I_m into non-empty multisets of D_{I_1} ... D_{I_m}."""
    index = {lab: i for i, lab in enumerate(labels)}
    max_order = max(len(lab) for lab in labels)
    powers = [dyson]                      # powers[m-1][i] = coefficient of c^{labels[i]} in (U-1)^m
    subs = [_submultisets(lab) for lab in labels]
    for m in range(2, max_order + 1):
        prev = powers[-1]
        cur = np.zeros_like(dyson)
        for i, lab in enumerate(labels):
            if len(lab) < m:
                continue
            acc = np.zeros_like(dyson[0])
            for sub, comp in subs[i]:
                if len(comp) >= m - 1:
                    acc += dyson[index[sub]] @ prev[index[comp]]
            cur[i] = acc
        powers.append(cur)
    out = np.zeros_like(dyson)
    for m, pw in enumerate(powers, start=1):
        out += ((-1.0) ** (m + 1) / m) * pw
    return out


class ExpansionModel:
    """Perturbative expansion of an LMDE over one time step (expansion_model.py:44-233)."""

    def __init__(self, operators, rotating_frame, dt: float, carrier_freqs, chebyshev_orders,
                 expansion_method: str = "dyson", expansion_order: Optional[int] = None,
                 expansion_labels=None, integration_method: Optional[str] = None, include_imag=None,
                 context=None, **kwargs):
        if expansion_method not in ("dyson", "magnus"):
            raise DynamicsError("ExpansionModel only accepts expansion_method 'dyson' or 'magnus'.")
        operators = np.asarray([np.asarray(getattr(op, "data", op), dtype=complex) for op in operators])
        if len(operators) != len(carrier_freqs):
            raise DynamicsError("carrier_freqs must have the same length as operators.")
        if len(operators) != len(chebyshev_orders):
            raise DynamicsError("chebyshev_orders must have the same length as operators.")
        self._expansion_method = expansion_method
        self._operators = operators
        self._dt = float(dt)
        self._carrier_freqs = [float(f) for f in carrier_freqs]
        self._chebyshev_orders = [int(d) for d in chebyshev_orders]
        self._include_imag = [True] * len(operators) if include_imag is None else [bool(x) for x in include_imag]
        self._rotating_frame = rotating_frame if isinstance(rotating_frame, RotatingFrame) else RotatingFrame(
            rotating_frame)
        n = operators.shape[-1]
        frame = self._rotating_frame
        self._Udt = np.asarray(frame.state_out_of_frame(self._dt, np.eye(n, dtype=complex)))

        # perturbations cos(2 pi nu t) T_m(t) G_j(t) / sin(-2 pi nu t) T_m(t) G_j(t) in the frame, evaluated
        # in the frame BASIS (expansion_model.py:236-318 builds them in the lab basis)
        ops_fb = np.asarray([frame.operator_into_frame_basis(op) for op in operators])
        d_im = frame.frame_diag_imag if frame.frame_diag is not None else np.zeros(n)

        def cheb(t, deg):
            return np.cos(deg * np.arccos(np.clip(2.0 * t / self._dt - 1.0, -1.0, 1.0)))

        def make(op, freq, deg, sine):
            rad = 2 * np.pi * freq

            def func(t):
                ph = np.exp(1j * d_im * t)                    # e = exp(d t)
                in_frame = (np.conj(ph)[:, None] * op) * ph[None, :]
                return (cheb(t, deg) * (np.sin(-rad * t) if sine else np.cos(rad * t))) * in_frame

            return func

        pert = []
        for op, freq, deg, inc in zip(ops_fb, self._carrier_freqs, self._chebyshev_orders, self._include_imag):
            pert += [make(op, freq, k, False) for k in range(deg + 1)]
            if inc:
                pert += [make(op, freq, k, True) for k in range(deg + 1)]
        self._n_perturbations = len(pert)
        self._labels = complete_labels(len(pert), expansion_order, expansion_labels)
        dyson_fb = _dyson_terms(pert, n, self._labels, self._dt, integration_method, **kwargs)
        terms_fb = dyson_fb if expansion_method == "dyson" else _magnus_from_dyson(self._labels, dyson_fb)
        terms = np.asarray([frame.operator_out_of_frame_basis(x) for x in terms_fb])
        if expansion_method == "dyson":
            terms = self._Udt @ terms                      # expansion_model.py:149-152
            self._constant_term = self._Udt
        else:
            self._constant_term = None
        self._terms = np.ascontiguousarray(terms)
        self._ctx = context
        self._device = None

    # -- properties mirroring the reference -------------------------------------------------------
    @property
    def expansion_method(self):
        return self._expansion_method

    @property
    def dt(self):
        return self._dt

    @property
    def Udt(self):  # pylint: disable=invalid-name
        return self._Udt

    @property
    def operators(self):
        return self._operators

    @property
    def rotating_frame(self):
        return self._rotating_frame

    @property
    def monomial_labels(self) -> List[tuple]:
        """Index multisets of the expansion terms, as sorted tuples, in canonical order."""
        return list(self._labels)

    @property
    def array_coefficients(self) -> np.ndarray:
        return self._terms

    @property
    def constant_term(self):
        return self._constant_term

    def approximate_signals(self, signals, t0: float, n_steps: int) -> np.ndarray:
        sigs = list(signals) if not isinstance(signals, SignalList) else [s for s in signals]
        sigs = [s if isinstance(s, Signal) else Signal(s) for s in sigs]
        return signal_list_envelope_dct(sigs, self._carrier_freqs, self._chebyshev_orders, t0, self._dt, n_steps,
                                        self._include_imag)

    def monomial_table(self, coeffs: np.ndarray) -> np.ndarray:
        """(n_coeffs, T) Chebyshev coefficients -> (T, M) monomials, the device's ``mono`` rows."""
        return np.ascontiguousarray(compute_monomials(self._labels, coeffs).T)

    def evaluate(self, coeffs) -> np.ndarray:
        """Value of the expansion polynomial for ONE coefficient vector, on the device (one-step
        ``midyn_expansion_solve`` applied to the identity; Dyson only -- the Magnus polynomial is the
        exponent, which the device never materialises on the host)."""
        coeffs = np.asarray(coeffs, dtype=float).reshape(-1, 1)
        if self._expansion_method != "dyson":
            raise DynamicsError("evaluate() is available for the Dyson expansion; the Magnus step is "
                                "Udt expm(polynomial) and is applied by MagnusSolver.solve.")
        n = self._terms.shape[-1]
        return self.device().solve_coeffs(coeffs[None], np.eye(n, dtype=complex), 1, True)[0]

    def device(self) -> "_lib.Expansion":
        if self._device is None:
            ctx = self._ctx or _lib.default_context()
            magnus = self._expansion_method == "magnus"
            self._device = _lib.Expansion(ctx, self._terms, constant_term=self._constant_term,
                                          post=self._Udt if magnus else None, use_expm=magnus)
            self._device.set_monomials(self._n_perturbations, self._labels)
        return self._device


# -------------------------------------------------------------------------------------------------
# solvers
# -------------------------------------------------------------------------------------------------
def _nested_ndim(x) -> int:
    if isinstance(x, (list, tuple)):
        return 1 + _nested_ndim(x[0])
    if hasattr(x, "ndim"):
        return x.ndim
    return 0


def _scalar_to_list(x, name):
    nd = _nested_ndim(x)
    if nd > 1:
        raise DynamicsError(f"{name} must be either 0d or 1d.")
    return (list(x), True) if nd == 1 else ([x], False)


def _signals_to_list(signals):
    if signals is None:
        return [signals], False
    if isinstance(signals, list) and len(signals) and isinstance(signals[0], (list, SignalList)):
        return signals, True
    if isinstance(signals, SignalList) or isinstance(signals, list):
        return [signals], False
    raise DynamicsError("Signals specified in invalid format.")


class _PerturbativeSolver:
    """Common ``solve`` of DysonSolver / MagnusSolver (perturbative_solver.py:44-170)."""

    def __init__(self, model: ExpansionModel):
        self._model = model

    @property
    def model(self) -> ExpansionModel:
        return self._model

    def solve(self, t0, n_steps, y0, signals):
        """Solve from ``t0`` over ``n_steps`` steps of the model's ``dt``.  Any of the arguments may be
        a list (all lists of one length): the simulations that share ``n_steps`` and the y0 shape
        go to the device as one call."""
        t0s, m0 = _scalar_to_list(t0, "t0")
        nss, m1 = _scalar_to_list(n_steps, "n_steps")
        y0s, m2 = (y0, True) if isinstance(y0, list) else ([y0], False)
        sgs, m3 = _signals_to_list(signals)
        lists = [t0s, nss, y0s, sgs]
        names = ["t0", "n_steps", "y0", "signals"]
        flags = [m0, m1, m2, m3]
        count = max([len(x) for x, f in zip(lists, flags) if f] or [1])
        for x, f, nm in zip(lists, flags, names):
            if f and len(x) != count:
                raise DynamicsError("If one of " + ", ".join(names) + " is given as a list of valid inputs, "
                                    "then the others must specify only a single input, or a list of the same "
                                    f"length ({nm}).")
        t0s, nss, y0s, sgs = [x if f else x * count for x, f in zip(lists, flags)]
        model = self._model
        frame = model.rotating_frame
        n = model.array_coefficients.shape[-1]
        results = [None] * count
        groups = {}
        for i in range(count):
            if sgs[i] is None or len(sgs[i]) != len(model.operators):
                raise DynamicsError("Signals must be the same length as the operators in the model.")
            y = np.asarray(y0s[i], dtype=complex)
            if y.ndim not in (1, 2) or y.shape[0] != n:
                raise DynamicsError("Shape mismatch for initial state y0 and the operators of the model.")
            groups.setdefault((int(nss[i]), y.shape), []).append(i)
        ident = np.eye(n, dtype=complex)
        for (steps, shape), idxs in groups.items():
            dev = model.device()
            coeffs = np.empty((len(idxs), dev.n_vars, steps), dtype=np.float64)
            ys = []
            for j, i in enumerate(idxs):
                # (the monomials of the coefficients are formed on the device: midyn_expansion_solve_coeffs)
                coeffs[j] = model.approximate_signals(sgs[i], t0s[i], steps)
                u0 = np.asarray(frame.state_out_of_frame(t0s[i], ident))
                ys.append((u0 @ np.asarray(y0s[i], dtype=complex)).reshape(n, -1))
            finals = dev.solve_coeffs(coeffs, np.stack(ys), len(idxs), False)
            for j, i in enumerate(idxs):
                uf = np.asarray(frame.state_into_frame(t0s[i] + steps * model.dt, ident))
                yf = (uf @ finals[j]).reshape(shape)
                results[i] = OdeResult(t=np.array([t0s[i], t0s[i] + steps * model.dt]),
                                       y=np.array([np.asarray(y0s[i], dtype=complex), yf]))
        return results if any(flags) else results[0]


class DysonSolver(_PerturbativeSolver):
    """Truncated Dyson series step (dyson_solver.py:37-209)."""

    def __init__(self, operators, rotating_frame, dt, carrier_freqs, chebyshev_orders, expansion_order=None,
                 expansion_labels=None, integration_method=None, include_imag=None, **kwargs):
        super().__init__(ExpansionModel(
            operators=operators, rotating_frame=rotating_frame, dt=dt, carrier_freqs=carrier_freqs,
            chebyshev_orders=chebyshev_orders, expansion_method="dyson", expansion_order=expansion_order,
            expansion_labels=expansion_labels, integration_method=integration_method, include_imag=include_imag,
            **kwargs))


class MagnusSolver(_PerturbativeSolver):
    """Exponential of a truncated Magnus expansion per step (magnus_solver.py:38-129)."""

    def __init__(self, operators, rotating_frame, dt, carrier_freqs, chebyshev_orders, expansion_order=None,
                 expansion_labels=None, integration_method=None, include_imag=None, **kwargs):
        super().__init__(ExpansionModel(
            operators=operators, rotating_frame=rotating_frame, dt=dt, carrier_freqs=carrier_freqs,
            chebyshev_orders=chebyshev_orders, expansion_method="magnus", expansion_order=expansion_order,
            expansion_labels=expansion_labels, integration_method=integration_method, include_imag=include_imag,
            **kwargs))
