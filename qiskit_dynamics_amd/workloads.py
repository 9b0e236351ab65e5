"""Synthetic inputs for the five BASELINE.json configurations.

Pure NumPy, deterministic, no device code: the same builders feed the golden-vector generator
(`oracle/gen_golden.py`, which runs the real reference on them), the CPU oracle, the parity tests
and `bench.py`, so every leg sees bit-identical inputs.
"""
from __future__ import annotations

import numpy as np

_X = np.array([[0.0, 1.0], [1.0, 0.0]], dtype=complex)
_Z = np.array([[1.0, 0.0], [0.0, -1.0]], dtype=complex)
_SM = np.array([[0.0, 1.0], [0.0, 0.0]], dtype=complex)  # sigma^- = |0><1|
_I2 = np.eye(2, dtype=complex)


def embed(op, q, n_qubits):
    """op acting on qubit q of an n_qubits register (qubit 0 = leftmost Kronecker factor): the same
    matrix as the chain of Kronecker products with identities, written by index arithmetic (one pass
    over the 2^n rows per non-zero entry of `op` instead of n dense Kronecker products)."""
    return embed_pair(op, q, None, None, n_qubits)


def _embed_entries(op1, q1, op2, q2, n_qubits):
    """Non-zero entries (rows, cols, value) of (op1 on qubit q1) . (op2 on qubit q2), q1 != q2 (op2 None:
    just op1): entry (i, j) is op1[bit_q1(i), bit_q1(j)] * op2[bit_q2(i), bit_q2(j)] when all other
    bits of i and j agree."""
    idx = np.arange(2**n_qubits)
    if q1 >= n_qubits and op2 is None:   # no such qubit: the Kronecker chain is all identities
        yield idx, idx, 1.0 + 0j
        return
    s1 = n_qubits - 1 - q1
    s2 = None if op2 is None else n_qubits - 1 - q2
    two = [(0, 0)] if op2 is None else [(c, d_) for c in (0, 1) for d_ in (0, 1)]
    for a in (0, 1):
        for b in (0, 1):
            v1 = op1[a, b]
            if v1 == 0:
                continue
            for c, d_ in two:
                v = v1 if op2 is None else v1 * op2[c, d_]
                if v == 0:
                    continue
                sel = ((idx >> s1) & 1) == a
                if op2 is not None:
                    sel &= ((idx >> s2) & 1) == c
                rows = idx[sel]
                cols = (rows & ~(1 << s1)) | (b << s1)
                if op2 is not None:
                    cols = (cols & ~(1 << s2)) | (d_ << s2)
                yield rows, cols, v


def embed_pair(op1, q1, op2, q2, n_qubits):
    """Dense matrix of (op1 on qubit q1) . (op2 on qubit q2)."""
    dim = 2**n_qubits
    out = np.zeros((dim, dim), dtype=complex)
    for rows, cols, v in _embed_entries(op1, q1, op2, q2, n_qubits):
        out[rows, cols] = v
    return out


def chain_hamiltonian(n_qubits, n_drives, nu0=5.0, dnu=0.05, coupling=0.002, rabi=0.02):
    """Qubit chain of SURVEY 8(d): H_d = sum_q 2 pi nu_q Z_q/2 + sum_q 2 pi J X_q X_{q+1},
    drives H_j = 2 pi r X_j / 2.  Returns (H_d (n,n), H_ops (k,n,n), nu (n_qubits,)).
    The terms are accumulated entry by entry in the order and with the arithmetic of the dense
    expression `h_d += c * embed(...) / 2`, without its 2^n x 2^n temporaries."""
    dim = 2**n_qubits
    nu = nu0 + dnu * np.arange(n_qubits)
    h_d = np.zeros((dim, dim), dtype=complex)
    for q in range(n_qubits):
        for rows, cols, v in _embed_entries(_Z, q, None, None, n_qubits):
            h_d[rows, cols] += 2 * np.pi * nu[q] * v / 2
    for q in range(n_qubits - 1):
        for rows, cols, v in _embed_entries(_X, q, _X, q + 1, n_qubits):
            h_d[rows, cols] += 2 * np.pi * coupling * v
    ops = np.zeros((n_drives, dim, dim), dtype=complex)
    for j in range(n_drives):
        for rows, cols, v in _embed_entries(_X, j, None, None, n_qubits):
            ops[j][rows, cols] = 2 * np.pi * rabi * v / 2
    return h_d, ops, nu


def sweep_parameters(instance, n_drives, seed=355):
    """(amplitudes a_j ~ U(0,1), phases phi_j ~ U(-pi,pi)) of sweep instance `instance`."""
    rng = np.random.default_rng(seed + instance)
    return rng.uniform(0.0, 1.0, n_drives), rng.uniform(-np.pi, np.pi, n_drives)


def gaussian_coefficient_table(times, amps, phases, carrier, t_final, sigma=1.0):
    """Coefficient table of the Gaussian-envelope drive signals
        s_j(t) = Re[ a_j exp(-(t-T/2)^2/(2 sigma^2)) exp(i(2 pi nu_j t + phi_j)) ]
    written with the reference's arithmetic order (signals/signals.py:148-155,574-577) so that it
    is bit-identical to `SignalList([...])(times)` for the matching `Signal` objects.

    times (T,), amps/phases (..., k), carrier (k,)  ->  (..., T, k) float64.
    """
    times = np.asarray(times, dtype=float)
    amps = np.asarray(amps, dtype=float)
    phases = np.asarray(phases, dtype=float)
    env = np.exp(-((times - t_final / 2) ** 2) / (2 * sigma**2))  # (T,)
    carrier_arg = 1j * 2 * np.pi * np.asarray(carrier, dtype=float)  # (k,)
    lead = amps.shape[:-1]
    out = np.empty(lead + (times.size, amps.shape[-1]))
    flat_a = amps.reshape(-1, amps.shape[-1])
    flat_p = phases.reshape(-1, phases.shape[-1])
    flat_o = out.reshape(-1, times.size, amps.shape[-1])
    for b in range(flat_a.shape[0]):
        exp_ph = np.exp(times[:, None] * carrier_arg[None, :] + 1j * flat_p[b][None, :])
        flat_o[b] = np.real((flat_a[b][None, :] * env[:, None]) * exp_ph)
    return out


def config1():
    """cfg 1: 2 qubits, n=4, 1 drift + 2 drives, rotating_frame = H_d, RK4 max_dt=0.01, T=10."""
    z0, z1 = embed(_Z, 0, 2), embed(_Z, 1, 2)
    x0, x1 = embed(_X, 0, 2), embed(_X, 1, 2)
    h_d = 2 * np.pi * 5.0 * (z0 + z1) / 2 + 2 * np.pi * 0.02 * (x0 @ x1)
    ops = np.stack([2 * np.pi * 0.1 * x0 / 2, 2 * np.pi * 0.1 * x1 / 2])
    y0 = np.zeros(4, dtype=complex)
    y0[0] = 1.0
    return dict(h_d=h_d, ops=ops, y0=y0, t_span=[0.0, 10.0], max_dt=0.01, carrier=[5.0, 5.0])


def schrodinger_config(n_qubits=10, n_drives=8, t_final=5.0, max_dt=0.005):
    """cfg 2/3 (n_qubits=10, k=8) and their down-scaled test versions."""
    h_d, ops, nu = chain_hamiltonian(n_qubits, n_drives)
    y0 = np.zeros(2**n_qubits, dtype=complex)
    y0[0] = 1.0
    return dict(h_d=h_d, ops=ops, y0=y0, t_span=[0.0, t_final], max_dt=max_dt,
                carrier=nu[:n_drives].copy(), t_final=t_final)


def lindblad_config(n_qubits=6, n_drives=6, n_diss=4, gamma=1e-3, t_final=5.0, max_dt=0.05):
    """cfg 4 (n_qubits=6 -> N=4096) and down-scaled versions: chain + static sigma^- dissipators."""
    h_d, ops, nu = chain_hamiltonian(n_qubits, n_drives)
    diss = np.stack([np.sqrt(gamma) * embed(_SM, q, n_qubits) for q in range(n_diss)])
    dim = 2**n_qubits
    rho0 = np.zeros((dim, dim), dtype=complex)
    rho0[0, 0] = 1.0
    return dict(h_d=h_d, ops=ops, static_dissipators=diss, rho0=rho0, t_span=[0.0, t_final],
                max_dt=max_dt, carrier=nu[:n_drives].copy(), t_final=t_final)


def transmon_chain(levels, sites, seed=0):
    """A chain of `sites` oscillators truncated to `levels` levels (three: the transmon of pulse-level simulations):
    H_d = sum_i w_i N_i + alpha / 2 N_i (N_i - 1) + J sum_i (a_i^+ a_{i+1} + h.c.), one drive a_i + a_i^+ per site.
    Returns (h_d, drive operators, drive frequencies in GHz); n = levels ** sites."""
    rng = np.random.default_rng(seed)
    a = np.diag(np.sqrt(np.arange(1, levels)), 1).astype(complex)
    num = a.conj().T @ a
    eye = np.eye(levels)

    def on(op, i):
        out = np.array([[1.0 + 0j]])
        for s in range(sites):
            out = np.kron(out, op if s == i else eye)
        return out

    w = 2 * np.pi * (5.0 + 0.1 * rng.standard_normal(sites))
    alpha = -2 * np.pi * 0.3
    h_d = sum(w[i] * on(num, i) + 0.5 * alpha * on(num @ (num - eye), i) for i in range(sites))
    for i in range(sites - 1):
        hop = on(a.conj().T, i) @ on(a, i + 1)
        h_d = h_d + 2 * np.pi * 0.005 * (hop + hop.conj().T)
    ops = [2 * np.pi * 0.02 * (on(a, i) + on(a.conj().T, i)) for i in range(sites)]
    return h_d, ops, w / (2 * np.pi)
