"""Fixed-step solvers on the device: ``solve_lmde`` / ``solve_ode`` dispatch and the ``Solver`` class.

Drop-in surface for the reference's ``solvers/solver_functions.py`` (``solve_lmde`` :220-373,
``solve_ode`` :129-217) and ``solvers/solver_classes.py`` (``Solver`` :177-377, ``solve`` :384-554,
list mode :556-590) restricted to the fixed-step methods of the hot path:

    method "RK4"  / "hip_RK4"        classic RK4          (fixed_step_solvers.py:43-77)
    method "scipy_expm" / "hip_expm" Magnus-1/2/3 + expm  (fixed_step_solvers.py:80-108,321-403)

Both spellings run the HIP kernels (there is no NumPy path in this package): user code written for
the reference keeps its method strings.  The time loop of ``fixed_step_solver_template`` (:406-459)
runs on the device; the host only (a) applies the step-count rule of ``get_fixed_step_sizes``
(:616-653), (b) evaluates every signal ONCE, array-vectorised over all evaluation times, into the
coefficient table S[B][R][k] (every time a fixed-step method touches is known up front), and
(c) changes basis of y0 / results (solver_functions.py:376-450).

A sweep (``Solver.solve`` with lists) whose instances share ``t_span``/``t_eval`` and the y0 shape is
ONE batched device solve: the reference's sequential loop over instances (solver_classes.py:568-586)
becomes the N dimension of the MFMA contraction.  The batched path is functional: it never mutates
the model's signals (the reference's ``_set_new_signals`` is not reentrant).
"""
from __future__ import annotations

import time
from typing import List, Optional, Tuple, Union

import numpy as np
from scipy.integrate._ivp.ivp import OdeResult

from ._lib import DynamicsError, SignalTable
from .models import BaseGeneratorModel, GeneratorModel, HamiltonianModel, LindbladModel
from .signals import Signal, SignalList, discrete_term_arrays

RK4_METHODS = ("RK4", "hip_RK4")
EXPM_METHODS = ("scipy_expm", "hip_expm")
# parallel-in-time LMDE methods; the reference names are accepted as aliases
RK4_PARALLEL_METHODS = ("hip_RK4_parallel", "jax_RK4_parallel")
EXPM_PARALLEL_METHODS = ("hip_expm_parallel", "jax_expm_parallel")
# scipy's adaptive integrators with the DEVICE right-hand side as their callback (a caller either side of the hot
# path; reference: solvers/scipy_solve_ivp.py:26-84).  COMPLEX methods integrate the complex state directly, the
# others a real embedding (Re, Im), exactly as the reference does.
SOLVE_IVP_COMPLEX_METHODS = ("RK45", "RK23", "BDF", "DOP853")
SOLVE_IVP_REAL_METHODS = ("LSODA", "Radau")
SOLVE_IVP_METHODS = SOLVE_IVP_COMPLEX_METHODS + SOLVE_IVP_REAL_METHODS
ODE_METHODS = list(RK4_METHODS + SOLVE_IVP_METHODS)
LMDE_METHODS = list(EXPM_METHODS + RK4_PARALLEL_METHODS + EXPM_PARALLEL_METHODS)


# -------------------------------------------------------------------------------------------------
# time grid (host)
# -------------------------------------------------------------------------------------------------
def merge_t_args(t_span, t_eval=None) -> np.ndarray:
    """``t_span`` with ``t_eval`` spliced in; same validation as the reference."""
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


def get_fixed_step_sizes(t_span, t_eval, max_dt) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(t_list, h_list, n_steps_list): fewest equal sub-steps per interval with |h| <= max_dt."""
    t_list = np.array(merge_t_args(t_span, t_eval))
    max_dt = np.array(max_dt)
    delta = np.diff(t_list)
    n_steps = np.abs(delta / max_dt).astype(int)
    for i, (dt_i, n_i) in enumerate(zip(delta, n_steps)):
        if n_i == 0:
            n_steps[i] = 1
        elif np.abs(dt_i / n_i) / max_dt > 1 + 1e-15:
            n_steps[i] = n_i + 1
    return t_list, np.array(delta / n_steps), n_steps


class FixedStepSchedule:
    """Everything the device loop needs about time: the distinct evaluation times, which table rows
    each step reads, step sizes and save slots.  ``points`` yields, for a step starting at t with
    size h, the times the method evaluates the model at (in the reference's floating-point order)."""

    def __init__(self, t_span, t_eval, max_dt, points):
        self.t_list, h_list, n_list = get_fixed_step_sizes(t_span, t_eval, max_dt)
        self.has_t_eval = t_eval is not None
        index = {}
        times: List[float] = []
        rows: List[List[int]] = []
        hs: List[float] = []
        save: List[int] = []

        def row_of(t):
            key = float(t)
            if key not in index:
                index[key] = len(times)
                times.append(key)
            return index[key]

        for i, (t0, h, n) in enumerate(zip(self.t_list[:-1], h_list, n_list)):
            t = t0
            for s in range(int(n)):
                pts = [row_of(p) for p in points(t, h)]
                rows.append((pts + [pts[-1]] * 3)[:3])
                hs.append(float(h))
                save.append(i + 1 if s == int(n) - 1 else -1)
                t = t + h
        if not times:
            times.append(float(self.t_list[0]))
        self.times = np.array(times, dtype=float)
        self.step_rows = np.array(rows, dtype=np.int32).reshape(-1, 3)
        self.step_h = np.array(hs, dtype=float)
        self.step_save = np.array(save, dtype=np.int32)
        self.n_save = len(self.t_list)

    def trim(self, y):
        """``trim_t_results``: with ``t_eval`` only the interior points are returned."""
        if self.has_t_eval:
            return self.t_list[1:-1], y[1:-1]
        return self.t_list, y


_SCHEDULE_CACHE: "dict" = {}
_SCHEDULE_CACHE_MAX = 16


def _cached_schedule(t_span, t_eval, max_dt, tag, points) -> FixedStepSchedule:
    """Schedules are pure functions of (t_span, t_eval, max_dt, method points): repeated solves of the
    same time grid (optimisation loops) reuse them instead of re-walking thousands of steps in Python."""
    try:
        key = (tuple(np.asarray(t_span, dtype=float).ravel().tolist()),
               None if t_eval is None else tuple(np.asarray(t_eval, dtype=float).ravel().tolist()),
               float(max_dt), tag)
    except (TypeError, ValueError):
        return FixedStepSchedule(t_span, t_eval, max_dt, points)
    sched = _SCHEDULE_CACHE.get(key)
    if sched is None:
        sched = FixedStepSchedule(t_span, t_eval, max_dt, points)
        if len(_SCHEDULE_CACHE) >= _SCHEDULE_CACHE_MAX:
            _SCHEDULE_CACHE.pop(next(iter(_SCHEDULE_CACHE)))
        _SCHEDULE_CACHE[key] = sched
    return sched


def _rk4_points(t, h):
    h2 = 0.5 * h
    return [t, t + h2, t + h]


def _magnus_points(order):
    if order == 1:
        return lambda t, h: [t + (h / 2)]
    if order == 2:
        c1 = 0.5 - np.sqrt(3) / 6
        c2 = 0.5 + np.sqrt(3) / 6
        return lambda t, h: [t + c1 * h, t + c2 * h]
    if order == 3:
        d1 = 0.5 - np.sqrt(15) / 10
        d3 = 0.5 + np.sqrt(15) / 10
        return lambda t, h: [t + d1 * h, t + 0.5 * h, t + d3 * h]
    raise DynamicsError("Only magnus_order 1, 2, and 3 are supported.")


# -------------------------------------------------------------------------------------------------
# model plumbing
# -------------------------------------------------------------------------------------------------
def _model_kind(model) -> str:
    if isinstance(model, LindbladModel):
        return "lindblad_vec" if model.vectorized else "lindblad"
    if isinstance(model, GeneratorModel):
        return "generator"
    raise DynamicsError(
        "The HIP solvers need a GeneratorModel / HamiltonianModel / LindbladModel instance; "
        "arbitrary Python callables cannot run on the device.")


def _signal_table(model, signals, times) -> np.ndarray:
    """(R, k) float64 table of ``signals`` (model's own signals when None) at ``times``."""
    if isinstance(model, LindbladModel):
        if signals is None:
            return model._signal_table(times)
        # the passed signals are validated and tabulated WITHOUT touching model.signals (functional path)
        return model._signal_table(times, signals if isinstance(signals, tuple) else (signals, None))
    n_ops = 0 if model._ops_fb is None else model._ops_fb.shape[0]
    sl = model.signals if signals is None else signals
    if sl is None:
        if n_ops:
            raise DynamicsError(
                f"{type(model).__name__} with non-empty operators must be evaluated signals.")
        return np.zeros((len(times), 0))
    if isinstance(sl, list):
        sl = SignalList(sl)
    if len(sl) != n_ops:
        raise DynamicsError("Signals needs to have the same length as operators.")
    return sl.table(times)


def _shape_y0(model, kind, y0):
    """Validate one initial state and return it as the (rows, m) matrix the device works on
    (still in the user's basis) plus the tag needed to undo the reshaping."""
    y0 = np.asarray(y0, dtype=complex)
    n = model.dim
    if kind == "lindblad":
        if y0.ndim != 2 or y0.shape != (n, n):
            raise DynamicsError("Shape mismatch for initial state y0 and LindbladModel.")
        return y0.flatten(order="F").reshape(-1, 1), "lindblad"
    rows = n * n if kind == "lindblad_vec" else n
    if y0.ndim not in (1, 2) or y0.shape[0] != rows:
        if kind == "lindblad_vec":
            raise DynamicsError("Shape mismatch for initial state y0 and LindbladModel in vectorized "
                                "evaluation mode.")
        raise DynamicsError("Shape mismatch for initial state y0 and HamiltonianModel.")
    if y0.ndim == 1:
        return y0.reshape(-1, 1), "vector"
    return y0, "matrix"


def _basis_matrix(model, kind):
    """Matrix V with (state in user basis) = V @ (state in frame basis), or None."""
    frame = model.rotating_frame
    if model.in_frame_basis or frame.frame_basis is None:
        return None
    if kind in ("lindblad", "lindblad_vec"):
        return frame.vectorized_frame_basis  # kron(conj(U), U): vec(U rho U^dagger) = V vec(rho)
    return frame.frame_basis


try:  # optional: keeps a host matvec from waking a whole BLAS thread pool (see _apply_basis)
    from threadpoolctl import threadpool_limits as _threadpool_limits
except ImportError:  # pragma: no cover
    _threadpool_limits = None


def _apply_basis(ctx, mat, cols):
    """mat @ cols for a (rows, ncols) block: one device zgemm for wide blocks (a sweep's states),
    a host matvec for a handful of columns.  The matvec runs on ONE BLAS thread: a pool of 64 OpenBLAS
    threads spins for milliseconds after the call, and under a CPU quota (16 CPUs on the measurement boxes) that
    throttles the thread which then launches thousands of kernels -- a 1000-step single-trajectory solve at
    n = 1024 took 100 ms instead of 61 ms."""
    if cols.shape[1] >= 16:
        return ctx.zgemm(mat, cols)
    if _threadpool_limits is None:
        return mat @ cols
    with _threadpool_limits(limits=1, user_api="blas"):
        return mat @ cols


def _prepare_y0_batch(model, kind, y0_list, shared):
    """y0 (shared) or list of y0 -> frame basis, device layout (rows, m) / (B, rows, m)."""
    shaped = [_shape_y0(model, kind, y0_list[0])] if shared else [_shape_y0(model, kind, y) for y in y0_list]
    tags = {t for _, t in shaped}
    shapes = {y.shape for y, _ in shaped}
    if len(tags) != 1 or len(shapes) != 1:
        raise DynamicsError("internal: batched instances must share the y0 shape")
    tag = shaped[0][1]
    v = _basis_matrix(model, kind)
    if shared:
        y = shaped[0][0]
        if v is not None:
            y = _apply_basis(model._ctx, v.conj().T, y)
        return np.ascontiguousarray(y), tag
    ys = np.stack([y for y, _ in shaped])  # (B, rows, m)
    if v is not None:
        b, rows, m = ys.shape
        cols = np.ascontiguousarray(ys.transpose(1, 0, 2).reshape(rows, b * m))
        cols = _apply_basis(model._ctx, v.conj().T, cols)
        ys = cols.reshape(rows, b, m).transpose(1, 0, 2)
    return np.ascontiguousarray(ys), tag


def _restore_batch(model, kind, tag, ys):
    """(B, P, rows, m) device result -> list of per-instance arrays in the user's layout, out of the
    frame basis when required (ONE basis-change product for the whole sweep)."""
    b, p, rows, m = ys.shape
    v = _basis_matrix(model, kind)
    if v is not None and m == 1 and tag == "vector" and b * p >= 16:
        # a sweep of state vectors: the states are the ROWS of one (B P, rows) block, so `ys v^T` leaves every instance's
        # (P, rows) result contiguous -- no transposition of the block before the product and none per instance after it
        # (0.18 s of the 4096-instance solve of BASELINE cfg 3)
        flat = model._ctx.zgemm(ys.reshape(b * p, rows), np.ascontiguousarray(v.T)).reshape(b, p, rows)
        return [flat[i] for i in range(b)]
    if v is not None:
        cols = np.ascontiguousarray(ys.transpose(2, 0, 1, 3).reshape(rows, b * p * m))
        cols = _apply_basis(model._ctx, v, cols)
        ys = cols.reshape(rows, b, p, m).transpose(1, 2, 0, 3)
    n = model.dim
    out = []
    for i in range(b):
        yi = ys[i]
        if tag == "lindblad":
            yi = np.stack([y[:, 0].reshape(n, n, order="F") for y in yi])
        elif tag == "vector":
            yi = yi[:, :, 0]
        out.append(np.ascontiguousarray(yi))
    return out


# Sweeps whose signals are all DiscreteSignals / constants get their coefficient table evaluated on
# the device once it has at least this many entries; smaller tables and
# Python-callable envelopes are evaluated on the host (row a8), bit-identically to the reference.
DEVICE_SIGNAL_TABLE_MIN = 1 << 16


def _signal_components(model, signals):
    """The k signal-likes behind one instance's coefficient vector, or None if there are none."""
    if isinstance(model, LindbladModel):
        if signals is None:
            ham, dis = model.signals
        else:
            ham, dis = signals if isinstance(signals, tuple) else (signals, None)
        out = []
        for part in (ham, dis):
            if part is not None:
                out += list(part)
        return out
    sl = model.signals if signals is None else signals
    return None if sl is None else list(sl)


def _batch_table(model, signals_list, times, batch, n_coeff):
    """Coefficient table of a batch: (B,R,k) ndarray (host evaluation, row a8) or a device
    ``SignalTable`` (row f1) when every signal is piecewise constant and the table is large."""
    shared_sig = all(s is signals_list[0] for s in signals_list)
    if n_coeff > 0 and not shared_sig and batch * len(times) * n_coeff >= DEVICE_SIGNAL_TABLE_MIN:
        comps = [_signal_components(model, s) for s in signals_list]
        if all(c is not None and len(c) == n_coeff for c in comps):
            arrays = discrete_term_arrays(comps)
            if arrays is not None:
                return SignalTable(model._ctx, batch, n_coeff, times, *arrays)
    if shared_sig:
        one = _signal_table(model, signals_list[0], times)
        return np.broadcast_to(one, (batch,) + one.shape)
    return np.stack([_signal_table(model, s, times) for s in signals_list])


# One small trajectory with many steps is launch-bound when advanced step by step.  The fixed-step
# methods are linear maps per step, so the same numbers (to rounding: products are re-associated) come
# out of the parallel-in-time route -- all step propagators at once, tree product (row f3) -- at a
# fraction of the latency.  "RK4" / "scipy_expm" solves of one (or a handful of) instances with at most
# this many rows and at least this many steps are routed there; AUTO_PARALLEL_IN_TIME = False keeps them
# sequential.
#
# The route taken is recorded in ``OdeResult.route`` ("sequential", "parallel_in_time(auto)", "parallel_in_time").
# Automatic routing is limited to HamiltonianModels (anti-Hermitian generators: every step propagator is unitary
# to rounding, so products of propagators are as accurate as stepping the state).  Dissipative models (a general
# GeneratorModel, a vectorised LindbladModel) are stepped sequentially unless AUTO_PARALLEL_IN_TIME == "all" or
# the user names a *_parallel method.
AUTO_PARALLEL_IN_TIME = True
FOLD_MAX_COLUMNS = 4096      # instances that share their signals become columns of one problem up to this many columns
AUTO_PARALLEL_MAX_ROWS = 128
AUTO_PARALLEL_MIN_STEPS = 256


def _auto_parallel_allowed(model) -> bool:
    if AUTO_PARALLEL_IN_TIME == "all":
        return True
    return bool(AUTO_PARALLEL_IN_TIME) and isinstance(model, HamiltonianModel)


def _solve_batch(model, t_span, y0_list, signals_list, method, t_eval=None, max_dt=None,
                 magnus_order=1, **unknown):
    """Solve ``len(y0_list)`` instances that share t_span/t_eval in one device call."""
    if unknown:
        raise DynamicsError(f"Unsupported solver options for the HIP fixed-step methods: {sorted(unknown)}")
    if max_dt is None:
        raise DynamicsError("max_dt must be specified for fixed-step methods.")
    kind = _model_kind(model)
    if method in EXPM_METHODS + EXPM_PARALLEL_METHODS + RK4_PARALLEL_METHODS:
        if kind == "lindblad":
            raise DynamicsError(
                "LMDE-specific methods with LindbladModel requires setting a vectorized=True.")
        if method in RK4_PARALLEL_METHODS:
            sched = _cached_schedule(t_span, t_eval, max_dt, "rk4", _rk4_points)
        else:
            sched = _cached_schedule(t_span, t_eval, max_dt, ("magnus", int(magnus_order)),
                                     _magnus_points(magnus_order))
    elif method in RK4_METHODS:
        sched = _cached_schedule(t_span, t_eval, max_dt, "rk4", _rk4_points)
    else:
        raise DynamicsError(f"Method {method} not supported by solve_lmde.")
    batch = len(y0_list)
    shared_y0 = all(y is y0_list[0] for y in y0_list)
    if kind == "lindblad":
        return _solve_batch_lindblad(model, sched, y0_list, signals_list, shared_y0)
    y0_dev, tag = _prepare_y0_batch(model, kind, y0_list, shared_y0)
    stack = model.stack
    # Instances that share their signals share the generator: they are just more COLUMNS of one problem
    # (the reference would loop them, solver_classes.py:568-586).  One coefficient table instead of B
    # copies of it on host and device, one generator evaluation per stage instead of nseg contractions.
    same_signals = batch > 1 and all(s is signals_list[0] for s in signals_list)
    n_inst, m_cols = batch, y0_dev.shape[-1]
    # ... identical instances (same signals AND same y0) are ONE solve whose result is replicated; otherwise the instances
    # are folded into columns while the folded problem stays within FOLD_MAX_COLUMNS (the output block is P x rows x
    # columns: an unbounded fold of matrix-valued y0 could not be chunked like the per-instance path)
    replicate = same_signals and shared_y0
    folded = same_signals and not shared_y0 and batch * m_cols <= FOLD_MAX_COLUMNS
    if replicate:
        signals_list, batch = signals_list[:1], 1
    elif folded:
        y0_dev = np.ascontiguousarray(y0_dev.transpose(1, 0, 2).reshape(y0_dev.shape[1], batch * m_cols))
        signals_list, batch, shared_y0 = signals_list[:1], 1, True
    table = _batch_table(model, signals_list, sched.times, batch, stack.k)
    # a few instances are still cheaper one after the other in parallel-in-time form (~2 ms each) than in
    # lock-step through thousands of launches (35-40 ms) or, for n <= 16, the persistent kernel (6-9 ms)
    max_batch = 3 if stack.n <= 16 else 16
    auto_parallel = (_auto_parallel_allowed(model) and batch <= max_batch and stack.n <= AUTO_PARALLEL_MAX_ROWS
                     and len(sched.step_h) >= AUTO_PARALLEL_MIN_STEPS and y0_dev.shape[-1] <= 64
                     and method in RK4_METHODS + EXPM_METHODS)
    route = "sequential"
    t_wall = time.perf_counter()
    if auto_parallel:
        route = "parallel_in_time(auto)"
        ys = stack.parallel_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save,
                                  0 if method in RK4_METHODS else magnus_order, y0_dev, batch, shared_y0)
    elif method in RK4_METHODS:
        ys = stack.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                             sched.n_save, y0_dev, batch, shared_y0)
    elif method in RK4_PARALLEL_METHODS + EXPM_PARALLEL_METHODS:
        route = "parallel_in_time"
        ys = stack.parallel_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                                  sched.n_save, 0 if method in RK4_PARALLEL_METHODS else magnus_order,
                                  y0_dev, batch, shared_y0)
    else:
        ys = stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save,
                              sched.n_save, magnus_order, y0_dev, batch, shared_y0)
    wall_s = time.perf_counter() - t_wall
    if folded:  # (1, P, rows, B*m) -> (B, P, rows, m)
        _, p, rows, _ = ys.shape
        ys = ys.reshape(p, rows, n_inst, m_cols).transpose(2, 0, 1, 3)
        route += "+folded"
    elif replicate:
        route += "+replicated"
    extras = _result_extras(model, method, magnus_order, len(sched.step_h), wall_s)
    results = []
    for y_b in _restore_batch(model, kind, tag, ys):     # (replicate: the ONE solved instance is restored ...)
        t_out, y_out = sched.trim(y_b)
        results.append(OdeResult(t=t_out, y=y_out, route=route, **extras))
    if replicate:                                        # (... and the restored arrays are copied per instance)
        one = results[0]
        results = [one] + [OdeResult(t=one.t.copy(), y=one.y.copy(), route=route, **extras) for _ in range(n_inst - 1)]
    return results


def _result_extras(model, method, magnus_order, n_steps, wall_s):
    """The bookkeeping fields of an ``OdeResult`` besides ``t`` / ``y`` / ``route``:

    ``nfev``    evaluations of the model per instance, as scipy counts them for its own methods (the reference passes
                scipy's fields through, solvers/scipy_solve_ivp.py:84; its fixed-step results carry none,
                solvers/fixed_step_solvers.py:445-459): 4 RHS evaluations per RK4 step; for the Magnus / expm methods
                the generator evaluations per step (1, 2, 3 quadrature points for magnus_order 1, 2, 3);
    ``wall_s``  wall-clock seconds of the ONE device call that solved the batch this instance belongs to (coefficient
                table H2D, device loop, results D2H) -- shared by the instances of a batch, not a per-instance share;
    ``device``  where it ran: ``"hip:<ordinal>"``."""
    if method in RK4_METHODS + RK4_PARALLEL_METHODS:
        per_step = 4
    else:
        per_step = int(magnus_order)
    ctx = getattr(model, "_ctx", None)
    return dict(nfev=int(per_step * n_steps), wall_s=float(wall_s), device=f"hip:{getattr(ctx, 'device', 0)}")


def _rotate_density(model, mats, into_frame_basis):
    """U^+ rho U (into) or U rho U^+ (out of the frame basis) for a stack of matrices."""
    basis = model.rotating_frame.frame_basis
    if model.in_frame_basis or basis is None:
        return mats
    ctx = model._ctx
    a, b = (basis.conj().T, basis) if into_frame_basis else (basis, basis.conj().T)
    n = basis.shape[0]
    if n < 128:
        if _threadpool_limits is None:
            return a @ mats @ b
        with _threadpool_limits(limits=1, user_api="blas"):  # see _apply_basis
            return a @ mats @ b
    return np.stack([ctx.zgemm(ctx.zgemm(a, m), b) for m in mats.reshape(-1, n, n)]).reshape(mats.shape)


def _solve_batch_lindblad(model, sched, y0_list, signals_list, shared_y0):
    """Non-vectorised Lindblad RK4: states are (n, n) density matrices, RHS by n x n zgemms."""
    n = model.dim
    batch = len(y0_list)
    mats = [np.asarray(y0_list[0], dtype=complex)] if shared_y0 else [np.asarray(y, dtype=complex) for y in y0_list]
    for m_ in mats:
        if m_.shape != (n, n):
            raise DynamicsError("Shape mismatch for initial state y0 and LindbladModel.")
    rho0 = _rotate_density(model, np.stack(mats), True)
    rho0 = rho0[0] if shared_y0 else rho0
    table = _batch_table(model, signals_list, sched.times, batch, model._lind.k)
    t_wall = time.perf_counter()
    ys = model._lind.rk4_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save,
                               rho0, batch, shared_y0)
    extras = _result_extras(model, "RK4", 1, len(sched.step_h), time.perf_counter() - t_wall)
    ys = _rotate_density(model, ys, False)
    results = []
    for b in range(batch):
        t_out, y_out = sched.trim(ys[b])
        results.append(OdeResult(t=t_out, y=y_out, route="sequential", **extras))
    return results


def _solve_ivp_device(model, t_span, y0, method, t_eval=None, signals=None, **kwargs) -> OdeResult:
    """``scipy.integrate.solve_ivp`` with the model's RHS evaluated on the device (scipy_solve_ivp.py:31-84 through
    solver_functions.py:200-207): y0 into the frame basis, the integrator runs on the host in the frame basis and calls
    ``G(t) y`` (``midyn_eval_rhs`` / ``midyn_lindblad_rhs``: kernels + one small PCIe round trip per evaluation),
    results out of the frame basis.  Keyword arguments (``atol``, ``rtol``, ``max_step``, ...) go to ``solve_ivp``."""
    from scipy.integrate import solve_ivp

    if kwargs.get("dense_output", False) is True:
        raise DynamicsError("dense_output not supported for solve_ivp.")
    kind = _model_kind(model)
    if kind == "lindblad":
        n = model.dim
        rho0 = np.asarray(y0, dtype=complex)
        if rho0.shape != (n, n):
            raise DynamicsError("Shape mismatch for initial state y0 and LindbladModel.")
        state0 = _rotate_density(model, rho0[None], True)[0]
        shape = (n, n)

        def rhs(t, rho):
            return model._lind.rhs(_signal_table(model, signals, np.array([t]))[0], t, rho[None])[0]
    else:
        state0, tag = _prepare_y0_batch(model, kind, [y0], True)      # (rows, m) in the frame basis
        shape = state0.shape
        stack = model.stack

        def rhs(t, y):
            coeffs = _signal_table(model, signals, np.array([t]))[0] if stack.k else None
            return stack.eval_rhs(coeffs, t, y)

    def flat(t, yf):
        return rhs(t, yf.reshape(shape)).ravel()

    embed_real = method in SOLVE_IVP_REAL_METHODS
    if embed_real:
        half = state0.size

        def fun(t, yr):
            out = flat(t, yr[:half] + 1j * yr[half:])
            return np.concatenate([out.real, out.imag])

        start = np.concatenate([state0.ravel().real, state0.ravel().imag])
    else:
        fun, start = flat, state0.ravel()
    t_wall = time.perf_counter()
    res = solve_ivp(fun, t_span=t_span, y0=start, t_eval=t_eval, method=method, **kwargs)
    ys = np.asarray(res.y)
    if embed_real:
        ys = ys[:ys.shape[0] // 2] + 1j * ys[ys.shape[0] // 2:]
    ys = ys.T.reshape((-1,) + tuple(shape))                            # (P, *shape): leading time axis
    if kind == "lindblad":
        y_out = _rotate_density(model, ys, False)
    else:
        y_out = _restore_batch(model, kind, tag, ys[None])[0]
    fields = dict(res)        # scipy's own fields (nfev, njev, nlu, status, message, success, ...) pass through
    fields.update(y=y_out, route="scipy_solve_ivp(device rhs)", wall_s=float(time.perf_counter() - t_wall),
                  device=f"hip:{getattr(getattr(model, '_ctx', None), 'device', 0)}")
    return OdeResult(**fields)


def solve_lmde(generator, t_span, y0, method: str = "RK4", t_eval=None, **kwargs) -> OdeResult:
    """Solve ``y' = G(t) y`` for a model instance with a fixed-step HIP method."""
    if not isinstance(generator, BaseGeneratorModel):
        raise DynamicsError(
            "solve_lmde on the HIP path requires a model instance (GeneratorModel, HamiltonianModel "
            "or LindbladModel); Python callables cannot be evaluated on the device.")
    if method in SOLVE_IVP_METHODS:   # ODE methods are accepted by solve_lmde (solver_functions.py:349-353)
        return _solve_ivp_device(generator, t_span, y0, method, t_eval=t_eval, **kwargs)
    if method not in RK4_METHODS + EXPM_METHODS + RK4_PARALLEL_METHODS + EXPM_PARALLEL_METHODS:
        raise DynamicsError(f"Method {method} not supported by solve_lmde.")
    return _solve_batch(generator, t_span, [y0], [None], method, t_eval=t_eval, **kwargs)[0]


def solve_ode(rhs, t_span, y0, method: str = "RK4", t_eval=None, **kwargs) -> OdeResult:
    """ODE-method entry point (solver_functions.py:129-217): the fixed-step RK4 of the hot path and scipy's adaptive
    methods with the device RHS (``"RK45"``, ``"RK23"``, ``"BDF"``, ``"DOP853"``, ``"LSODA"``, ``"Radau"``)."""
    if method not in RK4_METHODS + SOLVE_IVP_METHODS:
        raise DynamicsError(f"Method {method} not supported by solve_ode.")
    return solve_lmde(rhs, t_span, y0, method=method, t_eval=t_eval, **kwargs)


# -------------------------------------------------------------------------------------------------
# Solver
# -------------------------------------------------------------------------------------------------
def _nested_ndim(x) -> int:
    if isinstance(x, (list, tuple)):
        return 1 + _nested_ndim(x[0])
    if hasattr(x, "ndim"):
        return x.ndim
    return 0


def _t_span_to_list(t_span):
    nd = _nested_ndim(t_span)
    if nd > 2:
        raise DynamicsError("t_span must be either 1d or 2d.")
    if nd == 1:
        return [t_span], False
    return list(t_span), True


def _y0_to_list(y0):
    if isinstance(y0, list):
        return y0, True
    return [y0], False


def _signals_to_list(signals):
    if signals is None or isinstance(signals, tuple):
        return [signals], False
    if isinstance(signals, list) and len(signals) and isinstance(signals[0], (tuple, list, SignalList)):
        return signals, True
    if isinstance(signals, SignalList) or isinstance(signals, list):
        return [signals], False
    raise DynamicsError("Signals specified in invalid format.")


def _setup_args_lists(t_span, y0, signals):
    names = ["t_span", "y0", "signals"]
    lists, was_list = [], False
    for arg, fn in zip((t_span, y0, signals), (_t_span_to_list, _y0_to_list, _signals_to_list)):
        as_list, flag = fn(arg)
        lists.append(as_list)
        was_list = was_list or flag
    lens = [len(x) for x in lists]
    max_len = max(lens)
    for name, ln in zip(names, lens):
        if ln not in (1, max_len):
            big = names[lens.index(max_len)]
            raise DynamicsError(
                f"If one of t_span, y0, and signals is given as a list of valid inputs, then the "
                f"others must specify only a single input, or a list of the same length. {big} "
                f"specifies {max_len} inputs, but {name} is of length {ln}, which is incompatible.")
    lists = [x * max_len if ln == 1 else x for x, ln in zip(lists, lens)]
    return lists, was_list


class Solver:
    """Builds a :class:`HamiltonianModel` or :class:`LindbladModel` and solves it, including sweeps:
    each of ``t_span``, ``y0``, ``signals`` may be a single specification or a list (one common
    length); a list comes back iff any input was a list."""

    def __init__(self, static_hamiltonian=None, hamiltonian_operators=None, static_dissipators=None,
                 dissipator_operators=None, rotating_frame=None, in_frame_basis: bool = False,
                 array_library: Optional[str] = None, vectorized: Optional[bool] = None,
                 validate: bool = True, context=None):
        if static_dissipators is None and dissipator_operators is None:
            self._model = HamiltonianModel(
                static_operator=static_hamiltonian, operators=hamiltonian_operators,
                rotating_frame=rotating_frame, in_frame_basis=in_frame_basis,
                array_library=array_library, validate=validate, context=context)
        else:
            self._model = LindbladModel(
                static_hamiltonian=static_hamiltonian, hamiltonian_operators=hamiltonian_operators,
                static_dissipators=static_dissipators, dissipator_operators=dissipator_operators,
                rotating_frame=rotating_frame, in_frame_basis=in_frame_basis,
                array_library=array_library, vectorized=bool(vectorized), validate=validate,
                context=context)

    @property
    def model(self) -> Union[HamiltonianModel, LindbladModel]:
        return self._model

    def _normalize_signals(self, signals):
        if signals is None:
            return None
        if isinstance(self._model, LindbladModel) and isinstance(signals, (list, SignalList)):
            return (signals, None)
        return signals

    def solve(self, t_span, y0, signals=None, convert_results: bool = True, **kwargs):
        """As ``Solver.solve`` of the reference (solver_classes.py:384-554).  ``convert_results`` is
        accepted for signature compatibility: states are plain arrays here (the qiskit
        ``QuantumState``/``Operator`` wrappers belong to the provider shell, out of scope), so there
        is nothing to convert."""
        del convert_results
        (t_spans, y0s, sigs), multiple = _setup_args_lists(t_span, y0, signals)
        method = kwargs.pop("method", "RK4")
        t_eval = kwargs.get("t_eval", None)
        sigs = [self._normalize_signals(s) for s in sigs]
        n = len(t_spans)
        results: List[Optional[OdeResult]] = [None] * n
        # group instances that can share one device solve: same t_span and same y0 shape
        groups = {}
        for i in range(n):
            key = (tuple(np.asarray(t_spans[i], dtype=float).tolist()), np.asarray(y0s[i]).shape)
            groups.setdefault(key, []).append(i)
        if method in SOLVE_IVP_METHODS:
            # adaptive host integrators: each instance has its own step sequence (the reference loops them as well)
            for i in range(n):
                results[i] = _solve_ivp_device(self._model, t_spans[i], y0s[i], method, signals=sigs[i], **kwargs)
            return results if multiple else results[0]
        for (_tkey, _shape), idxs in groups.items():
            if t_eval is not None and _nested_ndim(t_eval) > 1:
                raise DynamicsError("t_eval must be 1 dimensional.")
            out = _solve_batch(self._model, t_spans[idxs[0]], [y0s[i] for i in idxs],
                               [sigs[i] for i in idxs], method, **kwargs)
            for i, r in zip(idxs, out):
                results[i] = r
        return results if multiple else results[0]
