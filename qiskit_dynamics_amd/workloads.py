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
    """op acting on qubit q of an n_qubits register (qubit 0 = leftmost Kronecker factor)."""
    out = np.array([[1.0 + 0j]])
    for i in range(n_qubits):
        out = np.kron(out, op if i == q else _I2)
    return out


def chain_hamiltonian(n_qubits, n_drives, nu0=5.0, dnu=0.05, coupling=0.002, rabi=0.02):
    """Qubit chain of SURVEY 8(d): H_d = sum_q 2 pi nu_q Z_q/2 + sum_q 2 pi J X_q X_{q+1},
    drives H_j = 2 pi r X_j / 2.  Returns (H_d (n,n), H_ops (k,n,n), nu (n_qubits,))."""
    dim = 2**n_qubits
    nu = nu0 + dnu * np.arange(n_qubits)
    h_d = np.zeros((dim, dim), dtype=complex)
    for q in range(n_qubits):
        h_d += 2 * np.pi * nu[q] * embed(_Z, q, n_qubits) / 2
    for q in range(n_qubits - 1):
        h_d += 2 * np.pi * coupling * (embed(_X, q, n_qubits) @ embed(_X, q + 1, n_qubits))
    ops = np.stack([2 * np.pi * rabi * embed(_X, j, n_qubits) / 2 for j in range(n_drives)])
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
