"""Row f4: DysonSolver / MagnusSolver on the two-transmon model of the reference's test
(test_dyson_magnus_solvers.py:142-219; dim 25, two drives) -- device solve vs the NumPy oracle loop.
    python tests/bench_perturbative_vs_oracle.py        (on the GPU box)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402
from oracle import dynamics_oracle as orc  # noqa: E402  (CPU comparison leg; lives under tests/: only tests, smoke()
#                                                and bench.py's cpu_baseline leg may use the oracle)

w_c, w_t = 2 * np.pi * 5.033, 2 * np.pi * 4.067
alpha_c, alpha_t, J = 2 * np.pi * (-0.33534), 2 * np.pi * (-0.33834), 2 * np.pi * 0.002
dim = 5
a = np.diag(np.sqrt(np.arange(1, dim)), 1)
N = np.diag(np.arange(dim)).astype(float)
I1, I2 = np.eye(dim), np.eye(dim**2)
a0, a1 = np.kron(a, I1), np.kron(I1, a)
N0, N1 = np.kron(N, I1), np.kron(I1, N)
H0 = w_c * N0 + 0.5 * alpha_c * N0 @ (N0 - I2) + w_t * N1 + 0.5 * alpha_t * N1 @ (N1 - I2) + J * (a0 @ a1.T + a0.T @ a1)
Hdc, Hdt = 2 * np.pi * (a0 + a0.T), 2 * np.pi * (a1 + a1.T)
r = 0.2
sig_w = 0.399128 / r
gauss = qd.Signal(lambda t: np.exp(-((t - 3.5 * sig_w) ** 2) / (2 * sig_w**2)), carrier_freq=5.0)
dt, n_steps = 0.01, 1000
y0 = np.eye(dim**2, dtype=complex)

direct = qd.Solver(static_hamiltonian=H0, hamiltonian_operators=[Hdc, Hdt], rotating_frame=H0).solve(
    t_span=[0.0, dt * n_steps], y0=y0, signals=[gauss, gauss], method="RK4", max_dt=dt / 20).y[-1]

for name, cls, order in (("magnus", qd.MagnusSolver, 3), ("dyson", qd.DysonSolver, 4)):
    t0 = time.perf_counter()
    sol = cls(operators=[-1j * Hdc, -1j * Hdt], rotating_frame=-1j * H0, dt=dt, carrier_freqs=[5.0, 5.0],
              chebyshev_orders=[1, 1], expansion_order=order, integration_method="DOP853", atol=1e-10, rtol=1e-10)
    t_build = time.perf_counter() - t0
    sol.solve(t0=0.0, n_steps=64, y0=y0, signals=[gauss, gauss])  # warm-up (allocations, upload)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        yf = sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=[gauss, gauss]).y[-1]
        best = min(best, time.perf_counter() - t0)
    fid = abs(1.0 - abs((yf.conj() * direct).sum()) ** 2 / dim**4)
    m = sol.model
    coeffs = m.approximate_signals([gauss, gauss], 0.0, n_steps)
    labels = np.array([list(lab) + [-1] * (order - len(lab)) for lab in m.monomial_labels])
    d, basis = orc.frame_setup(-1j * H0)
    t0 = time.perf_counter()
    y_cpu = orc.perturbative_solve(name, m.array_coefficients, labels, m.Udt, d, basis, coeffs[:, :50], y0, 0.0, 50, dt)
    t_cpu = (time.perf_counter() - t0) * n_steps / 50
    # list-mode sweep: 256 pulse amplitudes in one call (all instance-steps are rows of one device table)
    sweep = [[qd.Signal(lambda t, a=a: a * np.exp(-((t - 3.5 * sig_w) ** 2) / (2 * sig_w**2)), carrier_freq=5.0)] * 2
             for a in np.linspace(0.2, 1.0, 256)]
    sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=sweep[:4])
    t0 = time.perf_counter()
    rs = sol.solve(t0=0.0, n_steps=n_steps, y0=y0, signals=sweep)
    t_sweep = time.perf_counter() - t0
    t0 = time.perf_counter()
    mono = [m.monomial_table(m.approximate_signals(sg, 0.0, n_steps)) for sg in sweep]
    t_host = time.perf_counter() - t0
    print(json.dumps({"solver": name, "order": order, "dim": dim**2, "terms": len(m.monomial_labels), "steps": n_steps,
                      "sweep_256_instances_s": round(t_sweep, 3), "of_which_host_coefficients_s": round(t_host, 3),
                      "sweep_instance_steps_per_s": round(256 * n_steps / t_sweep, 1),
                      "model_build_s": round(t_build, 2), "device_solve_s": round(best, 4),
                      "steps_per_s": round(n_steps / best, 1), "infidelity_vs_direct_rk4": float(fid),
                      "numpy_oracle_solve_s_extrapolated": round(t_cpu, 2)}), flush=True)
