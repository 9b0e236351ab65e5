"""128 x 128 tiles + split-K against 64 x 64 tiles (one or two workgroups per CU) for small shards of the cfg-3 sweep:
microseconds per batched RHS evaluation (HIP events).  Feeds the tile rule of launch_gemm."""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qiskit_dynamics_amd as qd
from bench import build_frame_basis_stack
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points
ctx = qd.default_context()
cfg = workloads.schrodinger_config()
ops, static, frame_im = build_frame_basis_stack(cfg)
stack = qd.Stack(ctx, ops, static, frame_im)
sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
S = 22; rows = sched.step_rows[:S]; nr = int(rows.max()) + 1
y0 = cfg["y0"].reshape(-1, 1)
for B in (64, 128, 192, 256, 384, 512, 1024):
    amps = np.array([workloads.sweep_parameters(b, 8)[0] for b in range(B)]); phs = np.array([workloads.sweep_parameters(b, 8)[1] for b in range(B)])
    table = workloads.gaussian_coefficient_table(sched.times[:nr], amps, phs, cfg["carrier"], 5.0)
    res = {}
    for ft, fs in ((128, 0), (0, 0), (64, 0), (128, 0), (0, 0)):   # 128: pinned big tile; 0: the launcher's rule; repeated (A/B/A/B)
        ctx.set_option("force_tile", ft)
        ctx.set_option("force_splits", fs)
        p = qd.Rk4Plan(stack, sched.times[:nr], table, rows, sched.step_h[:S], y0, B, True)
        p.run(0, 2); ctx.synchronize(); ctx.timer_start(); p.run(2, S); ms = ctx.timer_stop(); p.close()
        res.setdefault(ft, []).append(round(ms / (4 * (S - 2)) * 1e3, 1))
    ctx.set_option("force_tile", 0)
    ctx.set_option("force_splits", 0)
    print(B, res, flush=True)
