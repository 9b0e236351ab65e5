"""Model classes whose evaluation runs on the MI355X through libmidyn.

Drop-in surface for the reference's ``GeneratorModel`` (models/generator_model.py:108-316),
``HamiltonianModel`` (models/hamiltonian_model.py:31-150) and ``LindbladModel``
(models/lindblad_model.py:38-538): same constructor keywords, ``evaluate(t)``, ``evaluate_rhs(t,
y)``, ``__call__``, ``signals`` setter, ``in_frame_basis``, ``dim``, ``rotating_frame`` and
operator getters.  ``array_library`` accepts ``None`` or ``"hip"``; this package has exactly one
compute path (the HIP library) and raises if it is unavailable.

Build time (host, once): -iH, frame diagonalisation, U^dagger . U of the operator stack, Kronecker
superoperators for the vectorised Lindblad model; the resulting frame-basis stack is uploaded once
and stays resident in HBM (``_lib.Stack``).  Per evaluation (device): the signal-weighted operator
sum, the frame phases and the contraction.
"""
from __future__ import annotations

from typing import List, Optional, Tuple, Union

import numpy as np

from . import _lib
from ._lib import DynamicsError
from .rotating_frame import RotatingFrame
from .signals import Signal, SignalList

_ACCEPTED_LIBS = (None, "hip")


def _check_library(array_library):
    if array_library not in _ACCEPTED_LIBS:
        raise DynamicsError(
            f"array_library={array_library!r} is not available in qiskit_dynamics_amd; the only "
            "compute path is array_library='hip' (libmidyn on MI355X).")


def is_hermitian(operator, tol: float = 1e-10) -> bool:
    """|| A^dagger - A ||_F < tol, accumulated over cache-sized tiles (the one-shot expression walks a transposed
    view of the whole matrix: 1.2 s per 4096 x 4096 operator, 10 of the 11 s of a 12-qubit model build)."""
    operator = np.asarray(operator)
    if operator.ndim != 2 or operator.shape[0] != operator.shape[1]:
        return False
    n, t = operator.shape[0], 256
    if n <= 2 * t:
        return bool(np.linalg.norm(operator.conj().T - operator) < tol)
    total = 0.0
    for i0 in range(0, n, t):
        for j0 in range(i0, n, t):
            d = operator[i0:i0 + t, j0:j0 + t] - operator[j0:j0 + t, i0:i0 + t].conj().T
            sq = float(np.vdot(d, d).real)
            total += sq if i0 == j0 else 2.0 * sq   # the (j, i) tile holds the same defect, conjugate transposed
            if not total < tol * tol:               # also leaves on NaN
                return False
    return True


def _as_signal_list(signals, n_ops, what="Signals"):
    if isinstance(signals, list):
        signals = SignalList(signals)
    if not isinstance(signals, SignalList):
        raise DynamicsError(f"{what} specified in unaccepted format.")
    if len(signals) != n_ops:
        raise DynamicsError(f"{what} needs to have the same length as operators.")
    return signals


def _kron_eye_left(a):
    """kron(I, a) for a (..., n, n): a on the diagonal blocks, written block by block (np.kron would multiply
    every entry of a with every entry of the identity: n^4 products, seconds at n = 64)."""
    n = a.shape[-1]
    out = np.zeros(a.shape[:-2] + (n, n, n, n), dtype=complex)
    for i in range(n):
        out[..., i, :, i, :] = a
    return out.reshape(a.shape[:-2] + (n * n, n * n))


def _kron_eye_right(b):
    """kron(b, I) for b (..., n, n): entry b[i, j] on the diagonal of block (i, j)."""
    n = b.shape[-1]
    out = np.zeros(b.shape[:-2] + (n, n, n, n), dtype=complex)
    for k in range(n):
        out[..., :, k, :, k] = b
    return out.reshape(b.shape[:-2] + (n * n, n * n))


def vec_commutator(a):
    """-i (I (x) A - A^T (x) I): column-stacking matrix of X -> -i[A, X]  (models/model_utils.py:31-71).
    Same entries as the Kronecker-product expression, assembled without its n^4 multiplications by the identity."""
    a = np.asarray(a, dtype=complex)
    return -1j * (_kron_eye_left(a) - _kron_eye_right(np.swapaxes(a, -1, -2)))


def vec_dissipator(l):
    """conj(L) (x) L - (I (x) L^+L + (L^+L)^T (x) I)/2: column-stacking Lindblad dissipator
    (models/model_utils.py:74-118).  (conj(L) (x) I)(I (x) L) = conj(L) (x) L entry for entry (every product
    of the matrix form has exactly one non-zero term), so the n^2 x n^2 matrix product is not needed."""
    l = np.asarray(l, dtype=complex)
    n = l.shape[-1]
    lc = l.conj()
    ldl = np.swapaxes(lc, -1, -2) @ l
    outer = np.einsum("...ij,...km->...ikjm", lc, l).reshape(l.shape[:-2] + (n * n, n * n))
    return outer - 0.5 * (_kron_eye_left(ldl) + _kron_eye_right(np.swapaxes(ldl, -1, -2)))


# Build-time basis changes U^+ A U (generator_model.py:319-365) are two n^3 zgemms per operator; from this dimension
# on they run on the device (componentwise-accurate 4M products) instead of host BLAS:
# the 10-qubit model of cfg 2/3 (9 operators of 1024 x 1024) builds in 0.1 s instead of 1.2 s.
DEVICE_BASIS_CHANGE_MIN_DIM = 512


def _into_frame_basis(ctx, frame, op):
    """``frame.operator_into_frame_basis(op)`` for (n, n) or (k, n, n) operators, on the device when large."""
    basis = frame.frame_basis
    if op is None or basis is None or basis.shape[0] < DEVICE_BASIS_CHANGE_MIN_DIM:
        return frame.operator_into_frame_basis(op)
    op = np.asarray(op, dtype=complex)
    uh = np.ascontiguousarray(basis.conj().T)
    mats = op.reshape(-1, op.shape[-2], op.shape[-1])
    out = np.stack([ctx.zgemm(uh, ctx.zgemm(m, basis)) for m in mats])
    return out.reshape(op.shape)


# Symmetry sectors are laid out contiguously on the device and every sector starts on a multiple of this many rows
# (the block map works on 16 x 16 blocks; a sector that straddled a block boundary would fill it from both sides).
# The padding rows are zero rows/columns of every operator.  Many small sectors would inflate the dimension, so the
# alignment is dropped (sectors contiguous, unpadded) when it would add more than SECTOR_MAX_PADDING of the dimension.
SECTOR_ALIGN = 16
SECTOR_MAX_PADDING = 0.125


def _sector_layout(labels):
    """(start row of each sector, internal dimension) for the internal layout: sectors in order of first appearance
    (the one with the most padding last), each aligned to SECTOR_ALIGN rows when that is affordable."""
    labels = np.asarray(labels)
    uniq, first, counts = np.unique(labels, return_index=True, return_counts=True)
    by_first = [int(i) for i in np.argsort(first)]
    for align in (SECTOR_ALIGN, 1):
        waste = [(-int(counts[i])) % align for i in by_first]
        last = by_first[int(np.argmax(waste))]           # the sector that would need the most padding goes last:
        order = [i for i in by_first if i != last] + [last]   # nothing follows it, so it needs none
        starts, pos = {}, 0
        for i in order:
            starts[int(uniq[i])] = pos
            pos += int(counts[i]) + (0 if i == last else (-int(counts[i])) % align)
        if pos <= labels.size * (1.0 + SECTOR_MAX_PADDING):
            break
    return starts, pos


def _sector_internal_dim(labels) -> int:
    return _sector_layout(labels)[1]


def _sector_slots(labels):
    """slot[a] = internal row of frame-basis vector a (API order = ascending eigenvalues): vectors of one sector are
    contiguous, in their API order, and every sector starts on a block boundary (padding rows stay zero)."""
    labels = np.asarray(labels)
    starts, _ = _sector_layout(labels)
    slot = np.empty(labels.size, dtype=np.int64)
    fill = dict(starts)
    for a, c in enumerate(labels):
        slot[a] = fill[int(c)]
        fill[int(c)] += 1
    return slot


class BaseGeneratorModel:
    """``model(t)`` -> generator matrix, ``model(t, y)`` -> RHS."""

    array_library = "hip"

    def __call__(self, time: float, y=None):
        return self.evaluate(time) if y is None else self.evaluate_rhs(time, y)


class GeneratorModel(BaseGeneratorModel):
    r"""LMDE generator :math:`G(t) = G_d + \sum_j s_j(t) G_j` evaluated on the device."""

    def __init__(self, static_operator=None, operators=None, signals=None, rotating_frame=None,
                 in_frame_basis: bool = False, array_library: Optional[str] = None, context=None):
        _check_library(array_library)
        if static_operator is None and operators is None:
            raise DynamicsError(
                f"{type(self).__name__} requires at least one of static_operator or operators to "
                "be specified at construction.")
        self._rotating_frame = RotatingFrame(rotating_frame)
        self._in_frame_basis = in_frame_basis
        frame = self._rotating_frame
        self._ctx = context or _lib.default_context()
        # operators into the frame basis; frame subtracted from the static part
        if static_operator is None:
            static_fb = None if frame.frame_diag is None else np.diag(-frame.frame_diag)
        else:
            static_fb = np.asarray(static_operator, dtype=complex)
            if static_fb.ndim != 2 or static_fb.shape[0] != static_fb.shape[1]:
                raise DynamicsError("static_operator must be a square matrix")
            if frame.frame_diag is not None:
                static_fb = frame.generator_minus_frame_in_basis(
                    static_fb, into_basis=lambda x: _into_frame_basis(self._ctx, frame, x))
        ops_fb = None
        if operators is not None:
            ops_fb = np.asarray(operators, dtype=complex)
            if ops_fb.ndim != 3 or ops_fb.shape[1] != ops_fb.shape[2]:
                raise DynamicsError("operators must be a (k, n, n) array or list of square matrices")
            ops_fb = _into_frame_basis(self._ctx, frame, ops_fb)
        self._static_fb = static_fb
        self._ops_fb = ops_fb
        self._dim = static_fb.shape[-1] if static_fb is not None else ops_fb.shape[-1]
        if frame.dim is not None and frame.dim != self._dim:
            raise DynamicsError("rotating frame dimension does not match the operators")
        # Frames with symmetry sectors: the device stack is stored with the frame-basis vectors grouped by sector
        # (a permutation of the reference's ascending-eigenvalue order), so that the exactly-zero blocks of operators
        # obeying a selection rule are contiguous and the work-list kernels skip them.  The permutation is internal to
        # the Stack wrapper: every array that crosses the C-ABI is permuted on the way in and back on the way out.
        slot = None
        labels = frame.sector_labels
        if labels is not None and type(self)._frame_diag_imag is GeneratorModel._frame_diag_imag:
            slot = _sector_slots(labels)

        def internal(x):
            """Frame-basis operator(s) in the internal layout: sectors contiguous and starting on block boundaries."""
            if x is None or slot is None:
                return x
            n_int = _sector_internal_dim(labels)
            out = np.zeros(x.shape[:-2] + (n_int, n_int), dtype=complex)
            out[..., slot[:, None], slot[None, :]] = x
            return out

        fim = self._frame_diag_imag()
        fim_int = fim
        if slot is not None and fim is not None:
            fim_int = np.zeros(_sector_internal_dim(labels))
            fim_int[slot] = fim
        self._stack = _lib.Stack(self._ctx, internal(ops_fb), internal(static_fb), fim_int)
        if slot is not None:
            self._stack.set_embedding(slot)
        self._signals = None
        self.signals = signals

    # frame diagonal seen by the device (overridden by the vectorised Lindblad model)
    def _frame_diag_imag(self):
        return self._rotating_frame.frame_diag_imag

    # -- properties -----------------------------------------------------------------------------
    @property
    def dim(self) -> int:
        return self._dim

    @property
    def rotating_frame(self) -> RotatingFrame:
        return self._rotating_frame

    @property
    def in_frame_basis(self) -> bool:
        return self._in_frame_basis

    @in_frame_basis.setter
    def in_frame_basis(self, value: bool):
        self._in_frame_basis = value

    @property
    def stack(self) -> "_lib.Stack":
        """The device-resident operator stack (frame basis)."""
        return self._stack

    @property
    def static_operator(self):
        if self._static_fb is None:
            return None
        if self._in_frame_basis:
            return self._static_fb
        return self._rotating_frame.operator_out_of_frame_basis(self._static_fb)

    @property
    def operators(self):
        if self._ops_fb is None:
            return None
        if self._in_frame_basis:
            return self._ops_fb
        return self._rotating_frame.operator_out_of_frame_basis(self._ops_fb)

    @property
    def signals(self) -> Optional[SignalList]:
        return self._signals

    @signals.setter
    def signals(self, signals):
        if signals is None:
            self._signals = None
        elif self._ops_fb is None:
            raise DynamicsError("Signals must be None if operators is None.")
        else:
            self._signals = _as_signal_list(signals, self._ops_fb.shape[0])

    # -- evaluation -----------------------------------------------------------------------------
    def _coefficients(self, time):
        if self._signals is None:
            if self._ops_fb is not None:
                raise DynamicsError(
                    f"{type(self).__name__} with non-empty operators must be evaluated signals.")
            return None
        return np.asarray(self._signals(time), dtype=float)

    def _basis(self):
        return self._rotating_frame.frame_basis

    def evaluate(self, time: float):
        """Generator matrix at ``time`` (in the rotating frame)."""
        g = self._stack.eval_generator(self._coefficients(time), time)
        basis = self._basis()
        if not self._in_frame_basis and basis is not None:
            g = self._ctx.zgemm(self._ctx.zgemm(basis, g), basis.conj().T)
        return g

    def evaluate_rhs(self, time: float, y):
        """``G(t) @ y`` for ``y`` of shape ``(n,)`` or ``(n, m)`` (states as columns)."""
        y = np.asarray(y, dtype=complex)
        basis = self._basis()
        rotate = (not self._in_frame_basis) and basis is not None
        if rotate:
            y = basis.conj().T @ y
        out = self._stack.eval_rhs(self._coefficients(time), time, y)
        if rotate:
            out = basis @ out
        return out


class HamiltonianModel(GeneratorModel):
    r"""Hamiltonian :math:`H(t) = H_d + \sum_j s_j(t) H_j`; evaluates the generator ``-i H`` in the
    rotating frame exactly like the reference (``model(t)`` returns ``-i e^{-tF}(H(t)-H_F)e^{tF}``)."""

    def __init__(self, static_operator=None, operators=None, signals=None, rotating_frame=None,
                 in_frame_basis: bool = False, array_library: Optional[str] = None,
                 validate: bool = True, context=None):
        # Hermiticity validation (hamiltonian_model.py:98-104).  Large operators that reach the device unchanged
        # (no frame or a diagonal one: no basis transform) are validated THERE -- || H - H^dagger ||_F from the
        # uploaded -iH, one pass at HBM rate instead of a transposed walk over host memory (0.35 s per 4096 x 4096
        # operator) -- with the same tolerance and the same errors.
        user_static = static_operator is not None
        if static_operator is None and operators is None:
            # hamiltonian_model.py:63-120 defers to the base class, which raises this (generator_model.py:125-140)
            raise DynamicsError(
                f"{type(self).__name__} requires at least one of static_operator or operators to be "
                "specified at construction.")
        dim = np.shape(static_operator if user_static else operators)[-1]
        frame_is_diagonal = rotating_frame is None or (
            isinstance(rotating_frame, RotatingFrame) and rotating_frame.frame_basis is None) or (
            not isinstance(rotating_frame, RotatingFrame) and np.ndim(rotating_frame) == 1)
        device_check = validate and dim > 1024 and frame_is_diagonal
        if static_operator is not None:
            if validate and not device_check and not is_hermitian(static_operator):
                raise DynamicsError("HamiltonianModel static_operator must be Hermitian.")
            static_operator = -1j * np.asarray(static_operator, dtype=complex)
        if operators is not None:
            if validate and not device_check and any(not is_hermitian(op) for op in operators):
                raise DynamicsError("HamiltonianModel operators must be Hermitian.")
            operators = -1j * np.asarray(operators, dtype=complex)
        super().__init__(static_operator=static_operator, operators=operators, signals=signals,
                         rotating_frame=rotating_frame, in_frame_basis=in_frame_basis,
                         array_library=array_library, context=context)
        if device_check:
            defect = self._stack.antiherm_defect()
            first_op = 1 if self._stack.has_static else 0
            if user_static and not defect[0] < 1e-10:
                raise DynamicsError("HamiltonianModel static_operator must be Hermitian.")
            if not np.all(defect[first_op:] < 1e-10):
                raise DynamicsError("HamiltonianModel operators must be Hermitian.")

    @property
    def static_operator(self):
        if self._static_fb is None:
            return None
        if self._in_frame_basis:
            return self._static_fb
        return 1j * self._rotating_frame.operator_out_of_frame_basis(self._static_fb)

    @property
    def operators(self):
        if self._ops_fb is None:
            return None
        if self._in_frame_basis:
            return 1j * self._ops_fb
        return 1j * self._rotating_frame.operator_out_of_frame_basis(self._ops_fb)


class LindbladModel(BaseGeneratorModel):
    """Lindblad master equation.  ``vectorized=True`` (needed by the LMDE/expm methods, as in the
    reference) turns it into a dim^2 generator model handled by the same device kernels;
    ``vectorized=False`` evaluates ``(A+B) rho + rho (A-B) + sum gamma L rho L^+`` with n x n MFMA
    zgemms on the device (``_lib.LindbladDevice``; no dim^2 x dim^2 superoperator is ever built)."""

    def __init__(self, static_hamiltonian=None, hamiltonian_operators=None, hamiltonian_signals=None,
                 static_dissipators=None, dissipator_operators=None, dissipator_signals=None,
                 rotating_frame=None, in_frame_basis: bool = False,
                 array_library: Optional[str] = None, vectorized: bool = False,
                 validate: bool = True, context=None):
        _check_library(array_library)
        if (static_hamiltonian is None and hamiltonian_operators is None
                and static_dissipators is None and dissipator_operators is None):
            raise DynamicsError(
                f"{type(self).__name__} requires at least one of static_hamiltonian "
                "hamiltonian_operators, static_dissipators, or dissipator_operators "
                "to be specified at construction.")
        if validate:
            if static_hamiltonian is not None and not is_hermitian(static_hamiltonian):
                raise DynamicsError("LinbladModel static_hamiltonian must be Hermitian.")
            if hamiltonian_operators is not None and any(
                    not is_hermitian(op) for op in hamiltonian_operators):
                raise DynamicsError("LindbladModel hamiltonian_operators must be Hermitian.")
        self._vectorized = vectorized
        self._rotating_frame = RotatingFrame(rotating_frame)
        self._in_frame_basis = in_frame_basis
        frame = self._rotating_frame

        def fb(x):
            if x is None:
                return None
            x = np.asarray(x, dtype=complex)
            if x.ndim == 2:
                x = x[None]
            return frame.operator_into_frame_basis(x)

        if static_hamiltonian is not None:
            g = -1j * np.asarray(static_hamiltonian, dtype=complex)
            if frame.frame_diag is not None:
                g = frame.generator_minus_frame_in_basis(g)
            h_d = 1j * g
        elif frame.frame_diag is not None:
            h_d = 1j * np.diag(-frame.frame_diag)
        else:
            h_d = None
        self._h_d = h_d
        self._h_ops = fb(hamiltonian_operators)
        self._n_static = fb(static_dissipators)
        self._l_ops = fb(dissipator_operators)
        for x in (self._h_d, self._h_ops, self._n_static, self._l_ops):
            if x is not None:
                self._dim = x.shape[-1]
                break
        self._ctx = context or _lib.default_context()
        self._hamiltonian_signals = None
        self._dissipator_signals = None
        self._lind = None
        if not vectorized:
            self._build_unvectorized()
            self.signals = (hamiltonian_signals, dissipator_signals)
            return
        # superoperator stack (column stacking): static = vec_comm(H_d) + sum vec_diss(N_j),
        # operators = [vec_comm(H_j) ; vec_diss(L_j)] -- assembled on the device from the n x n operators
        # (the host never holds the n^2 x n^2 arrays; `vec_commutator` / `vec_dissipator` above are the same
        # formulas on the host, kept for the getters and the tests)
        self._stack = _lib.Stack.from_lindblad(self._ctx, self._h_d, self._h_ops, self._n_static, self._l_ops,
                                               frame.vectorized_frame_diag_imag())
        self.signals = (hamiltonian_signals, dissipator_signals)

    @classmethod
    def from_hamiltonian(cls, hamiltonian, static_dissipators=None, dissipator_operators=None,
                         dissipator_signals=None, array_library: Optional[str] = None, vectorized: bool = False):
        """Construct from a :class:`HamiltonianModel` (models/lindblad_model.py:214-260): its operators are read
        out of the frame basis through the model's own getters and handed to the constructor together with its
        signals, rotating frame and ``in_frame_basis`` flag -- exactly what the reference does, including that
        ``static_operator`` of a framed Hamiltonian model is the static Hamiltonian with the frame already
        subtracted."""
        in_frame_basis = hamiltonian.in_frame_basis
        hamiltonian.in_frame_basis = False
        static_hamiltonian = hamiltonian.static_operator
        hamiltonian_operators = hamiltonian.operators
        hamiltonian.in_frame_basis = in_frame_basis
        return cls(static_hamiltonian=static_hamiltonian, hamiltonian_operators=hamiltonian_operators,
                   hamiltonian_signals=hamiltonian.signals, static_dissipators=static_dissipators,
                   dissipator_operators=dissipator_operators, dissipator_signals=dissipator_signals,
                   rotating_frame=hamiltonian.rotating_frame, in_frame_basis=hamiltonian.in_frame_basis,
                   array_library=array_library, vectorized=vectorized, context=getattr(hamiltonian, "_ctx", None))

    def _build_unvectorized(self):
        """Operator stacks of the non-vectorised RHS (operator_collections.py:451-567):
        left = A + B, right = A - B with B = -iH, A = -1/2 sum N^+N - 1/2 sum gamma_j L_j^+L_j; both
        share the coefficient vector (ham signals, dissipator signals)."""
        n = self._dim

        def ldl(x):
            return np.swapaxes(x.conj(), -1, -2) @ x

        a_static = None if self._n_static is None else -0.5 * np.sum(ldl(self._n_static), axis=0)
        left_static = right_static = None
        if self._h_d is not None or a_static is not None:
            hd = np.zeros((n, n), dtype=complex) if self._h_d is None else self._h_d
            az = np.zeros((n, n), dtype=complex) if a_static is None else a_static
            left_static, right_static = -1j * hd + az, 1j * hd + az
        lops, rops = [], []
        if self._h_ops is not None:
            lops.append(-1j * self._h_ops)
            rops.append(1j * self._h_ops)
        if self._l_ops is not None:
            half = -0.5 * ldl(self._l_ops)
            lops.append(half)
            rops.append(half)
        left_ops = np.concatenate(lops, axis=0) if lops else None
        right_ops = np.concatenate(rops, axis=0) if rops else None
        fim = self._rotating_frame.frame_diag_imag
        self._stack = _lib.Stack(self._ctx, left_ops, left_static, fim)
        self._stack_right = _lib.Stack(self._ctx, right_ops, right_static, fim)
        diss = [x for x in (self._n_static, self._l_ops) if x is not None]
        n_static = 0 if self._n_static is None else self._n_static.shape[0]
        n_dyn = 0 if self._l_ops is None else self._l_ops.shape[0]
        k_h = 0 if self._h_ops is None else self._h_ops.shape[0]
        self._lind = _lib.LindbladDevice(self._stack, self._stack_right, k_h, n_static, n_dyn,
                                         np.concatenate(diss, axis=0) if diss else None)

    # -- properties -----------------------------------------------------------------------------
    @property
    def dim(self) -> int:
        return self._dim

    @property
    def vectorized(self) -> bool:
        return self._vectorized

    @property
    def rotating_frame(self) -> RotatingFrame:
        return self._rotating_frame

    @property
    def in_frame_basis(self) -> bool:
        return self._in_frame_basis

    @in_frame_basis.setter
    def in_frame_basis(self, value: bool):
        self._in_frame_basis = value

    @property
    def stack(self):
        return self._stack

    def _out(self, x):
        if x is None or self._in_frame_basis:
            return x
        return self._rotating_frame.operator_out_of_frame_basis(x)

    @property
    def static_hamiltonian(self):
        return self._out(self._h_d)

    @property
    def hamiltonian_operators(self):
        return self._out(self._h_ops)

    @property
    def static_dissipators(self):
        return self._out(self._n_static)

    @property
    def dissipator_operators(self):
        return self._out(self._l_ops)

    @property
    def signals(self) -> Tuple[Optional[SignalList], Optional[SignalList]]:
        return (self._hamiltonian_signals, self._dissipator_signals)

    @signals.setter
    def signals(self, new_signals):
        self._hamiltonian_signals, self._dissipator_signals = self._checked_signals(new_signals)

    # -- evaluation -----------------------------------------------------------------------------
    def _coefficients(self, time):
        """Concatenated (ham, diss) coefficient vector, as _concatenate_coefficients does."""
        parts = []
        if self._hamiltonian_signals is not None:
            parts.append(np.asarray(self._hamiltonian_signals(time), dtype=float))
        elif self._h_ops is not None:
            raise DynamicsError(
                f"{type(self).__name__} with non-empty hamiltonian operators cannot be evaluated "
                "without hamiltonian signals.")
        if self._dissipator_signals is not None:
            parts.append(np.asarray(self._dissipator_signals(time), dtype=float))
        elif self._l_ops is not None:
            raise DynamicsError(
                f"{type(self).__name__} with non-empty dissipator operators cannot be evaluated "
                "without dissipator signals.")
        if not parts:
            return None
        return np.concatenate(parts, axis=-1)

    def _checked_signals(self, new_signals):
        """(ham SignalList | None, dis SignalList | None) from a user pair, validated like the setter."""
        ham, dis = new_signals
        if ham is not None:
            if self._h_ops is None:
                raise DynamicsError("Hamiltonian signals must be None if hamiltonian_operators is None.")
            ham = _as_signal_list(ham, self._h_ops.shape[0], "Hamiltonian signals")
        if dis is not None:
            if self._l_ops is None:
                raise DynamicsError("Dissipator signals must be None if dissipator_operators is None.")
            dis = _as_signal_list(dis, self._l_ops.shape[0], "Dissipator signals")
        return ham, dis

    def _signal_table(self, times, signals=None):
        """(R, k_h + k_d) table for an array of times (used by the solvers).  ``signals`` (a
        (ham, dis) pair) replaces the model's own for this call only; the model is not modified."""
        ham, dis = (self._hamiltonian_signals, self._dissipator_signals) if signals is None \
            else self._checked_signals(signals)
        if ham is None and self._h_ops is not None:
            raise DynamicsError(
                f"{type(self).__name__} with non-empty hamiltonian operators cannot be evaluated "
                "without hamiltonian signals.")
        if dis is None and self._l_ops is not None:
            raise DynamicsError(
                f"{type(self).__name__} with non-empty dissipator operators cannot be evaluated "
                "without dissipator signals.")
        parts = [sl.table(times) for sl in (ham, dis) if sl is not None]
        if not parts:
            return np.zeros((len(times), 0))
        return np.ascontiguousarray(np.concatenate(parts, axis=-1))

    def evaluate_hamiltonian(self, time: float):
        sig = None if self._hamiltonian_signals is None else self._hamiltonian_signals(time)
        if self._h_ops is not None and self._h_d is not None:
            ham = np.tensordot(sig, self._h_ops, axes=1) + self._h_d
        elif self._h_ops is not None:
            ham = np.tensordot(sig, self._h_ops, axes=1)
        elif self._h_d is not None:
            ham = self._h_d
        else:
            raise DynamicsError("LindbladModel has no Hamiltonian terms.")
        if self._rotating_frame.frame_diag is not None:
            ham = self._rotating_frame.operator_into_frame(
                time, ham, operator_in_frame_basis=True, return_in_frame_basis=self._in_frame_basis)
        return ham

    def evaluate(self, time: float):
        if not self._vectorized:
            raise NotImplementedError(
                "Non-vectorized Lindblad models cannot be represented without a given state.")
        g = self._stack.eval_generator(self._coefficients(time), time)
        vb = self._rotating_frame.vectorized_frame_basis
        if not self._in_frame_basis and vb is not None:
            g = self._ctx.zgemm(self._ctx.zgemm(vb, g), vb.conj().T)
        return g

    def evaluate_rhs(self, time: float, y):
        y = np.asarray(y, dtype=complex)
        coeffs = self._coefficients(time)
        basis = self._rotating_frame.frame_basis
        rotate = (not self._in_frame_basis) and basis is not None
        n = self._dim
        if self._vectorized:
            if rotate:
                y = self._rotating_frame.vectorized_frame_basis_adjoint @ y
            out = self._stack.eval_rhs(coeffs, time, y)
            if rotate:
                out = self._rotating_frame.vectorized_frame_basis @ out
            return out
        # non-vectorised: (n,n) or (l,n,n) density matrices, n x n zgemms on the device
        single = y.ndim == 2
        rho = y[None] if single else y
        if rho.ndim != 3 or rho.shape[1:] != (n, n):
            raise DynamicsError("Shape mismatch for initial state y0 and LindbladModel.")
        if rotate:
            rho = basis.conj().T @ rho @ basis
        res = self._lind.rhs(coeffs, time, rho)
        if rotate:
            res = basis @ res @ basis.conj().T
        return res[0] if single else res
