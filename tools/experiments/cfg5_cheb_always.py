"""cfg 5 shard with the Chebyshev series forced (ctx option chebyshev = 2) against the default choice: terms, kernel time, difference."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
import qiskit_dynamics_amd as qd
from qiskit_dynamics_amd import workloads
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points
ctx = qd.default_context(0)
cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
ops, static, fim, _ = bench.build_diag_frame_stack(cfg)
stack = qd.Stack(ctx, ops, static, fim)
sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(2))
y0 = cfg["y0"].reshape(-1, 1)
count = 128
table, _, _ = bench.sweep_table(workloads, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])
def run():
    return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)
res = {}
for tag, opts in (("default", {}), ("cheb_always", dict(chebyshev=2)), ("default", {}), ("cheb_always", dict(chebyshev=2)), ("work_list_default", dict(ell_sweep=0))):
    with ctx.options(**opts):
        run()
        cs = bench.profile_pass(ctx, run, ("rk4_resident", "rhs_blocks_gemm"))
        terms = ctx.counters("sweep_series")["launches"]
        ys = run()
    res[tag] = ys
    print(json.dumps({"variant": tag, "terms": int(terms), "kernel_ms": round(cs["rk4_resident"]["ms"], 4), "worklist_launches": int(cs["rhs_blocks_gemm"]["launches"]),
                      "norm_dev": float(np.abs(np.linalg.norm(ys[:, -1, :, 0], axis=1) - 1).max())}), flush=True)
print("max |cheb_always - default| =", float(np.abs(res["cheb_always"] - res["default"]).max()))
print("max |default - work list| =", float(np.abs(res["default"] - res["work_list_default"]).max()), " max |cheb_always - work list| =", float(np.abs(res["cheb_always"] - res["work_list_default"]).max()))
