"""Per-RHS-evaluation launch time of the batched path vs batch size (cfg 2/3 model), split-K on/off."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points
from bench import build_frame_basis_stack, build_model_stack
ctx = qd.default_context()
cfg = workloads.schrodinger_config()
if os.environ.get("MIDYN_DENSE_STACK"):      # reference eigenvector order: dense kernels
    ops, static, frame_im = build_frame_basis_stack(cfg)
    stack = qd.Stack(ctx, ops, static, frame_im)
else:                                        # as HamiltonianModel uploads it: grouped by symmetry sector
    ops, static, frame_im, perm = build_model_stack(cfg)
    stack = qd.Stack(ctx, ops, static, frame_im)
    stack.set_permutation(perm)
sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
S = 12
rows = sched.step_rows[:S]; nr = int(rows.max()) + 1
y0 = cfg["y0"].reshape(-1, 1)
sizes = [int(x) for x in sys.argv[1:]] or [2, 8, 64, 128, 256, 512, 1024, 2048]
for B in sizes:
    amps = np.array([workloads.sweep_parameters(b, 8)[0] for b in range(B)])
    phs = np.array([workloads.sweep_parameters(b, 8)[1] for b in range(B)])
    table = workloads.gaussian_coefficient_table(sched.times[:nr], amps, phs, cfg["carrier"], 5.0)
    res = {}
    ref = None
    for tag, opts in (("splitk", {"split_k": 1, "force_tile": 0, "multi_stream": 0}),
                      ("nosplit", {"split_k": 0, "force_tile": 0, "multi_stream": 0}),
                      ("splitk64", {"split_k": 1, "force_tile": 64, "multi_stream": 0}),
                      ("multi_stream", {"split_k": 1, "force_tile": 0, "multi_stream": 1})):
        for k_, v_ in opts.items():
            ctx.set_option(k_, v_)
        p = qd.Rk4Plan(stack, sched.times[:nr], table, rows, sched.step_h[:S], y0, B, True)
        p.run(0, 2); ctx.synchronize()
        import time
        t0 = time.perf_counter(); p.run(2, S); ctx.synchronize(); dt = time.perf_counter() - t0
        out = p.fetch(); p.close()
        if ref is None: ref = out
        err = float(np.max(np.abs(out - ref)))
        res[tag] = (dt / (4 * (S - 2)) * 1e6, err)
    ctx.set_option("split_k", 1); ctx.set_option("force_tile", 0); ctx.set_option("multi_stream", 1)
    print(B, {k: (round(v[0], 1), f"{v[1]:.1e}") for k, v in res.items()}, "us per batched eval;",
          f"{B / min(res['splitk'][0], res['multi_stream'][0]) * 1e6:.0f} evals/s", flush=True)
