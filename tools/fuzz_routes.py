"""Random shapes through the DEFAULT routes of the sweep solvers (midyn_rk4_solve / midyn_expm_solve on a Stack) against
(a) the same solve with every round-4 route switched off (ctx options combine = 0, combine_sweep = 0: the MFMA GEMM / work
list kernels, one launch per product) and (b) the NumPy oracle for the first, a middle and the last instance.

What is drawn per case: the dimension (2 .. 400: the one-wave kernels, the one-launch sweep kernels with n_pad 64 / 128 /
256, the per-launch combine kernels), 1 .. 12 operators of random plane kinds (real / imaginary / complex planes, a few of
them block-sparse with exactly-zero blocks), the static operator's kind or none, a frame diagonal or none, 1 .. 4100
instances (ragged column blocks), shared or per-instance initial states with 1 or 3 columns, 2 .. 9 steps of two sizes with a
t_eval point, forwards or backwards in time, RK4 or scipy_expm with Magnus order 1 / 2.

    python tools/fuzz_routes.py --cases 150 --seed 0          (on the GPU box; prints one line per case, exits 1 on a mismatch)

The reference computes every instance as  y' = (G_d + sum_j c_j(t) G_j) y  (models/operator_collections.py:101-134,
solvers/fixed_step_solvers.py:43-108,321-403); the oracle restates that and is used here only as the checker.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

COUNTERS = ("rhs_combine", "rhs_gemm", "rhs_blocks_gemm", "combine_sweep", "rk4_resident")


def crand(rng, *shape):
    return rng.uniform(-1, 1, shape) + 1j * rng.uniform(-1, 1, shape)


def draw_case(rng):
    n = int(rng.choice([rng.integers(2, 17), rng.integers(17, 65), rng.integers(65, 129), rng.integers(129, 257),
                        rng.integers(257, 401)], p=[0.2, 0.25, 0.2, 0.2, 0.15]))
    k = int(rng.choice([rng.integers(1, 5), rng.integers(5, 9), rng.integers(9, 13)], p=[0.5, 0.35, 0.15]))
    pure = rng.random() < 0.4                      # all operators of one plane kind (Hamiltonian models in a real basis)
    kinds = [str(rng.choice(["r", "i"]))] * k if pure else [str(rng.choice(["r", "i", "c"])) for _ in range(k)]
    batch = int(rng.choice([1, 3, 16, 17, 100, 257, 1000, 2049, 4100], p=[0.05, 0.1, 0.1, 0.15, 0.2, 0.15, 0.15, 0.05, 0.05]))
    if n > 256 and batch > 1000:
        batch = 1000                                 # keeps the GEMM-route twin of a case within seconds
    return dict(n=n, kinds=kinds, static=[None, "r", "i", "c"][int(rng.integers(0, 4))], frame=bool(rng.integers(0, 2)),
                batch=batch, shared=bool(rng.integers(0, 2)), m=int(rng.choice([1, 3], p=[0.8, 0.2])),
                steps=int(rng.integers(2, 10)), backwards=bool(rng.integers(0, 2)),
                method=str(rng.choice(["RK4", "scipy_expm"], p=[0.65, 0.35])), magnus=int(rng.integers(1, 3)),
                sparse=rng.random() < 0.25)


def build_ops(rng, c):
    n = c["n"]
    scale = 0.6 / np.sqrt(n)

    def one(kind):
        a = crand(rng, n, n) * scale
        if c["method"] == "scipy_expm":              # anti-Hermitian generators: the norm check below applies
            a = -1j * (a + a.conj().T) / 2
        if c["sparse"] and n > 48:                   # exactly-zero off-diagonal blocks (two symmetry sectors)
            h = (n // 32) * 16 or n // 2
            a[:h, h:] = 0
            a[h:, :h] = 0
        return a.real + 0j if kind == "r" else (1j * a.imag if kind == "i" else a)

    ops = np.array([one(kd) for kd in c["kinds"]])
    static = None if c["static"] is None else one(c["static"])
    fim = rng.normal(size=n) if c["frame"] else None
    return ops, static, fim


def solve(qd, stack, c, sched, table, y0, default_routes):
    ctx = qd.default_context()
    ctx.set_option("combine", 1 if default_routes else 0)
    ctx.set_option("combine_sweep", 1 if default_routes else 0)
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        args = (sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save)
        if c["method"] == "RK4":
            ys = stack.rk4_solve(*args, y0, c["batch"], c["shared"])
        else:
            ys = stack.expm_solve(*args, c["magnus"], y0, c["batch"], c["shared"])
    finally:
        ctx.set_option("profile", 0)
        ctx.set_option("combine", 1)
        ctx.set_option("combine_sweep", 1)
    return ys, {name: int(ctx.counters(name)["launches"]) for name in COUNTERS}


def run_case(qd, orc, seed, verbose=True):
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points, _rk4_points

    rng = np.random.default_rng(seed)
    c = draw_case(rng)
    ops, static, fim = build_ops(rng, c)
    n, batch, k = c["n"], c["batch"], len(c["kinds"])
    h = 0.01 if c["method"] == "RK4" else 0.04
    span = c["steps"] * h * 0.93                     # a last step shorter than the others
    t_span = [span, 0.0] if c["backwards"] else [0.0, span]
    t_eval = [0.37 * span]
    points = _rk4_points if c["method"] == "RK4" else _magnus_points(c["magnus"])
    sched = FixedStepSchedule(t_span, t_eval, h, points)
    table = rng.uniform(-1, 1, (batch, len(sched.times), k))
    y0 = crand(rng, n, c["m"]) if c["shared"] else crand(rng, batch, n, c["m"])
    y0 /= np.linalg.norm(y0, axis=-2, keepdims=True)
    stack = qd.Stack(qd.default_context(), ops, static, fim)
    t0 = time.perf_counter()
    got, cd = solve(qd, stack, c, sched, table, y0, True)
    ref, cr = solve(qd, stack, c, sched, table, y0, False)
    stack.close()
    assert got.shape == ref.shape == (batch, sched.n_save, n, c["m"]), (got.shape, ref.shape)
    assert np.all(np.isfinite(got)), "non-finite result on the default route"
    scale = 1.0 + np.max(np.abs(ref))
    err_routes = float(np.max(np.abs(got - ref)) / scale)
    d = None if fim is None else 1j * fim
    times = np.asarray(sched.times)
    err_oracle = 0.0
    for b in sorted({0, batch // 2, batch - 1}):
        row = lambda t: table[b, int(np.argmin(np.abs(times - t)))]
        yb = (y0 if c["shared"] else y0[b])
        cols = []
        for col in range(c["m"]):
            if c["method"] == "RK4":
                _, yref = orc.rk4_solve(lambda t, y: orc.generator_rhs(static, ops, row(t), d, None, t, y), t_span, yb[:, col], h, t_eval)
            else:
                _, yref = orc.expm_solve(lambda t: orc.generator_evaluate(static, ops, row(t), d, None, t), t_span, yb[:, col], h,
                                         t_eval, c["magnus"])
            cols.append(np.asarray(yref))
        yref = np.stack(cols, axis=-1)               # (saved t_eval points, n, m)
        err_oracle = max(err_oracle, float(np.max(np.abs(got[b, 1:-1] - yref)) / (1.0 + np.max(np.abs(yref)))))
    ok = err_routes < 1e-11 and err_oracle < 1e-9
    if verbose or not ok:
        route = ",".join(f"{name}={v}" for name, v in cd.items() if v)
        print(f"seed {seed:5d} {'ok  ' if ok else 'FAIL'} n={n:3d} k={''.join(c['kinds']):<12s} st={c['static'] or '-'} fr={int(c['frame'])} "
              f"B={batch:4d} sh={int(c['shared'])} m={c['m']} steps={c['steps']} bw={int(c['backwards'])} sp={int(c['sparse'])} "
              f"{c['method']}{c['magnus'] if c['method'] != 'RK4' else ''}: routes {err_routes:.1e} oracle {err_oracle:.1e} "
              f"[{route}] {time.perf_counter() - t0:.2f}s", flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    import qiskit_dynamics_amd as qd
    from oracle import dynamics_oracle as orc

    qd.default_context()
    bad = []
    for s in range(args.seed, args.seed + args.cases):
        try:
            if not run_case(qd, orc, s):
                bad.append(s)
        except Exception as exc:                     # a refused shape or a crash is a finding too
            print(f"seed {s:5d} EXC  {type(exc).__name__}: {exc}", flush=True)
            bad.append(s)
    print(f"{args.cases - len(bad)} of {args.cases} cases agree; failing seeds: {bad}", flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
