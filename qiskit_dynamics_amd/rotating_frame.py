"""Rotating frame bookkeeping (host side, build time only).

Everything here runs ONCE per model (diagonalisation, basis changes of the operator stack) or once
per solve (y0 / results basis change) -- SURVEY.md rows a3/a4/a12.  The per-evaluation frame work
(``exp(+-d t)`` phases, the Delta(t) Hadamard mask) is done on the device from ``frame_diag``;
inside a solve the model is always in the frame basis, so no ``U . U^dagger`` is ever in the loop.

Conventions follow the reference ``models/rotating_frame.py``: the frame operator F is
anti-Hermitian, a Hermitian input H means F = -iH (:585-660); a 1-D input is the diagonal (:87-95);
``frame_diag = -1j * eigh(1j F).evals`` and ``frame_basis = U`` (:102-107); vectorised frame basis
``kron(conj(U), U)`` (:518).
"""
from __future__ import annotations

import numpy as np

from ._lib import DynamicsError


def _to_anti_hermitian(mat, atol=1e-10, rtol=1e-10):
    mat = np.asarray(mat)
    if mat.ndim == 1:
        if np.allclose(mat.imag, 0.0, atol=atol, rtol=0) and np.allclose(mat, mat.conj(), atol=atol, rtol=rtol):
            return -1j * mat
        if np.allclose(mat, -mat.conj(), atol=atol, rtol=rtol):
            return mat
    elif mat.ndim == 2 and mat.shape[0] == mat.shape[1]:
        if np.allclose(mat, mat.conj().T, atol=atol, rtol=rtol):
            return -1j * mat
        if np.allclose(mat, -mat.conj().T, atol=atol, rtol=rtol):
            return mat
    raise DynamicsError("frame_operator must be either a Hermitian or anti-Hermitian matrix.")


# Frame operators of this dimension or more are examined for symmetry sectors (below: plain eigh, exactly the
# reference's call)
SECTOR_MIN_DIM = 32
SECTOR_MAX_COUNT = 64      # more connected components than this: not worth per-sector decompositions
SECTOR_MAX_SHARE = 0.75    # largest component above this share of the dimension: the zeros gained are too few


def _eigh_by_sectors(h):
    """``eigh`` of a Hermitian matrix, sector by sector.

    If the non-zero pattern of ``h`` splits the indices into several connected components (conserved quantities of
    the frame Hamiltonian: parity, excitation number, ...), ``h`` is block diagonal under a permutation and each block
    is diagonalised on its own.  Eigenvalues and eigenvectors are those of ``np.linalg.eigh`` (rotating_frame.py:102-107)
    up to rounding and the usual eigenvector gauge -- returned in the same ascending order -- but every eigenvector is
    EXACTLY zero outside its sector, where LAPACK on the full matrix leaves 1e-17 noise.  Operators that obey a
    selection rule between the sectors then have exactly-zero blocks in the frame basis, which the device skips
    (bit-identical to multiplying the zeros): the 10-qubit chain of BASELINE cfg 2/3 conserves parity, half of every
    frame-basis operator vanishes, and the batched RHS contraction takes 1.20 ms instead of 2.28 ms.  Also cheaper:
    two 512 x 512 decompositions, side by side on two threads, instead of one 1024 x 1024 (0.29 s vs 0.59 s).

    Returns (evals, basis, labels): ``labels[a]`` = sector of eigenvector a, or None when there is one sector only."""
    n = h.shape[0]
    if n < SECTOR_MIN_DIM:
        evals, basis = np.linalg.eigh(h)
        return evals, basis, None
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import connected_components

    n_comp, comp = connected_components(csr_matrix(h != 0), directed=False)
    # The sector path has to pay for itself: one component, dozens of tiny ones (a diagonal or nearly diagonal frame
    # operator given as a matrix would be n one-element "sectors": n Python-level eigh calls and a device embedding that
    # skips nothing), or one component that is almost everything -- plain eigh, as the reference does.
    if n_comp <= 1 or n_comp > SECTOR_MAX_COUNT or np.bincount(comp).max() > SECTOR_MAX_SHARE * n:
        evals, basis = np.linalg.eigh(h)
        return evals, basis, None
    evals = np.empty(n)
    basis = np.zeros((n, n), dtype=complex)
    labels = np.empty(n, dtype=np.int64)
    pos = 0
    members = [np.flatnonzero(comp == c) for c in range(n_comp)]
    big = [c for c in range(n_comp) if members[c].size >= 128]
    solved = {}
    if len(big) > 1:       # LAPACK releases the GIL: the large sectors are diagonalised side by side
        from concurrent.futures import ThreadPoolExecutor

        with ThreadPoolExecutor(max_workers=min(len(big), 8)) as pool:
            for c, res in zip(big, pool.map(lambda c_: np.linalg.eigh(h[np.ix_(members[c_], members[c_])]), big)):
                solved[c] = res
    for c in range(n_comp):
        idx = members[c]
        w, v = solved[c] if c in solved else np.linalg.eigh(h[np.ix_(idx, idx)])
        cols = np.arange(pos, pos + idx.size)
        basis[np.ix_(idx, cols)] = v
        evals[cols] = w
        labels[cols] = c
        pos += idx.size
    order = np.argsort(evals, kind="stable")      # the reference's (LAPACK's) ascending order
    return evals[order], np.ascontiguousarray(basis[:, order]), labels[order]


class RotatingFrame:
    """Frame operator F (anti-Hermitian) with its eigen-decomposition ``F = U diag(d) U^dagger``."""

    def __init__(self, frame_operator, atol: float = 1e-10, rtol: float = 1e-10):
        if isinstance(frame_operator, RotatingFrame):
            frame_operator = frame_operator.frame_operator
        self._frame_operator = frame_operator
        self._vec_basis = None
        self._sector_labels = None
        self._anti_hermitian = None
        if frame_operator is None:
            self._dim = None
            self._frame_diag = None
            self._frame_basis = None
            return
        f = _to_anti_hermitian(np.asarray(frame_operator), atol=atol, rtol=rtol)
        self._anti_hermitian = f
        if f.ndim == 1:
            self._frame_diag = f
            self._frame_basis = None
        else:
            evals, basis, sectors = _eigh_by_sectors(1j * f)
            self._frame_diag = -1j * evals
            self._frame_basis = basis
            self._sector_labels = sectors
        self._dim = len(self._frame_diag)

    # -- properties ---------------------------------------------------------------------------
    @property
    def dim(self):
        return self._dim

    @property
    def frame_operator(self):
        return self._frame_operator

    @property
    def frame_diag(self):
        return self._frame_diag

    @property
    def frame_basis(self):
        return self._frame_basis

    def generator_minus_frame_in_basis(self, generator, into_basis=None):
        """``U^+ G U - diag(d)`` -- the static generator of a model in this frame (generator_model.py:319-340) --
        evaluated as ``U^+ (G - F) U``: the same matrix, but the frame is subtracted BEFORE the basis change.  The
        reference subtracts diag(d) (entries ~ the frame's eigenvalues) from a product that carries rounding errors of
        that size; when the static operator IS the frame operator (the usual ``rotating_frame=H_d`` set-up) it is left
        with 1e-13 noise where the exact answer is zero, and so is every evaluation.  In this order the difference is
        formed exactly (``G - F`` is exactly zero then, and stays zero through the basis change), otherwise the two
        agree to rounding.  ``into_basis``: optional callable doing the basis change (the device one for large
        operators); default ``operator_into_frame_basis``."""
        g = np.asarray(generator, dtype=complex)
        f = self._anti_hermitian
        if f is None:
            return g
        if f.ndim == 1:
            return g - np.diag(f)
        fb = self.operator_into_frame_basis if into_basis is None else into_basis
        return fb(g - f)

    @property
    def sector_labels(self):
        """Symmetry sector of every frame-basis vector (None: one sector / diagonal frame); see _eigh_by_sectors."""
        return self._sector_labels

    @property
    def frame_basis_adjoint(self):
        return None if self._frame_basis is None else self._frame_basis.conj().T

    @property
    def frame_diag_imag(self):
        """Im(d): what the device needs (d is purely imaginary)."""
        return None if self._frame_diag is None else np.ascontiguousarray(self._frame_diag.imag)

    @property
    def vectorized_frame_basis(self):
        if self._frame_basis is None:
            return None
        if self._vec_basis is None:
            self._vec_basis = np.kron(self._frame_basis.conj(), self._frame_basis)
        return self._vec_basis

    @property
    def vectorized_frame_basis_adjoint(self):
        vb = self.vectorized_frame_basis
        return None if vb is None else vb.conj().T

    def vectorized_frame_diag_imag(self):
        """Im of D with vec(e^{-tF} X e^{tF}) = exp(-D t) o vec(X) (column stacking):
        D[r + n c] = d_r - d_c.  Lets the vectorised Lindblad model reuse the generator kernels."""
        if self._frame_diag is None:
            return None
        d = self._frame_diag.imag
        return np.ascontiguousarray((d.reshape(-1, 1) - d.reshape(1, -1)).flatten(order="F"))

    # -- basis changes (host, O(n^3) once) --------------------------------------------------------
    def state_into_frame_basis(self, y):
        y = np.asarray(y)
        return y if self._frame_basis is None else self._frame_basis.conj().T @ y

    def state_out_of_frame_basis(self, y):
        y = np.asarray(y)
        return y if self._frame_basis is None else self._frame_basis @ y

    def operator_into_frame_basis(self, op):
        if op is None or self._frame_basis is None:
            return None if op is None else np.asarray(op)
        return self._frame_basis.conj().T @ (np.asarray(op) @ self._frame_basis)

    def operator_out_of_frame_basis(self, op):
        if op is None or self._frame_basis is None:
            return None if op is None else np.asarray(op)
        return self._frame_basis @ (np.asarray(op) @ self._frame_basis.conj().T)

    # -- elementwise frame maps (host versions, used for results post-processing / tests) --------
    def state_into_frame(self, t, y, y_in_frame_basis=False, return_in_frame_basis=False):
        """exp(-tF) y."""
        y = np.asarray(y)
        if self._frame_operator is None:
            return y
        out = y if y_in_frame_basis else self.state_into_frame_basis(y)
        out = (np.exp(self._frame_diag * (-t)) * out.T).T
        return out if return_in_frame_basis else self.state_out_of_frame_basis(out)

    def state_out_of_frame(self, t, y, y_in_frame_basis=False, return_in_frame_basis=False):
        """exp(tF) y."""
        return self.state_into_frame(-t, y, y_in_frame_basis, return_in_frame_basis)

    def operator_into_frame(self, t, operator, operator_in_frame_basis=False,
                            return_in_frame_basis=False):
        """exp(-tF) A exp(tF)."""
        operator = np.asarray(operator)
        if self._frame_operator is None:
            return operator
        out = operator if operator_in_frame_basis else self.operator_into_frame_basis(operator)
        e = np.exp(self._frame_diag * t)
        out = out * (e.conj().reshape(self.dim, 1) * e)
        return out if return_in_frame_basis else self.operator_out_of_frame_basis(out)

    def operator_out_of_frame(self, t, operator, operator_in_frame_basis=False,
                              return_in_frame_basis=False):
        return self.operator_into_frame(-t, operator, operator_in_frame_basis, return_in_frame_basis)

    def generator_into_frame(self, t, operator, operator_in_frame_basis=False,
                             return_in_frame_basis=False):
        """exp(-tF) G exp(tF) - F."""
        operator = np.asarray(operator)
        if self._frame_operator is None:
            return operator
        out = operator if operator_in_frame_basis else self.operator_into_frame_basis(operator)
        e = np.exp(self._frame_diag * t)
        out = out * (e.conj().reshape(self.dim, 1) * e) - np.diag(self._frame_diag)
        return out if return_in_frame_basis else self.operator_out_of_frame_basis(out)

    def generator_out_of_frame(self, t, operator, operator_in_frame_basis=False,
                               return_in_frame_basis=False):
        """exp(tF) G exp(-tF) + F  (models/rotating_frame.py:476-508)."""
        operator = np.asarray(operator)
        if self._frame_operator is None:
            return operator
        out = operator if operator_in_frame_basis else self.operator_into_frame_basis(operator)
        e = np.exp(self._frame_diag * (-t))
        out = out * (e.conj().reshape(self.dim, 1) * e) + np.diag(self._frame_diag)
        return out if return_in_frame_basis else self.operator_out_of_frame_basis(out)

    def vectorized_map_into_frame(self, time, op, operator_in_frame_basis=False,
                                  return_in_frame_basis=False):
        """Vectorised (column-stacking) linear map of dimension dim**2 into the frame:
        ``(Delta-bar kron Delta) o op`` in the frame basis (models/rotating_frame.py:537-582; host
        version -- the device applies the same Hadamard factor through the vectorised frame diagonal)."""
        op = np.asarray(op)
        if self._frame_diag is None:
            return op
        if not operator_in_frame_basis and self._frame_basis is not None:
            op = self.vectorized_frame_basis_adjoint @ (op @ self.vectorized_frame_basis)
        expvals = np.exp(self._frame_diag * time)
        outer = (expvals.conj().reshape(self.dim, 1) * expvals).flatten()
        op = np.outer(outer.conj(), outer) * op
        if not return_in_frame_basis and self._frame_basis is not None:
            op = self.vectorized_frame_basis @ (op @ self.vectorized_frame_basis_adjoint)
        return op

