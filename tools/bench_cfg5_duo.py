"""cfg 5 shard (n = 4096, Magnus 2, 20 steps): kernel time of the sweep kernels per series term -- one workgroup per instance
(ell_sweep_kernel), two with operator elements (ell_sweep_duo_kernel), two without (ell_flip_duo_kernel), and the two-workgroup kernels with parts switched off (ctx option `ablate`, results
wrong: 1 no exchange, 2 write-through stores although the partners share an XCD, 4 no local slots, 8 no crossing slots).
    python tools/bench_cfg5_duo.py [instances ...]          (on the GPU box)"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd import workloads  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points  # noqa: E402

ctx = qd.default_context(0)
cfg = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25)
ops, static, fim, _ = bench.build_diag_frame_stack(cfg)
stack = qd.Stack(ctx, ops, static, fim)
sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(2))
y0 = cfg["y0"].reshape(-1, 1)
for count in [int(x) for x in sys.argv[1:]] or [128, 64, 16]:
    table, _, _ = bench.sweep_table(workloads, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])

    def run():
        return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)

    ref = None
    fl = dict(ell_sweep_flip=0)
    for tag, opts in (("one_workgroup", dict(ell_sweep_duo=0)), ("duo", dict(fl)), ("flip", {}),
                      ("flip_write_through", dict(ablate=2)), ("flip_no_exchange", dict(ablate=1)),
                      ("flip_no_local_slots", dict(ablate=4)), ("flip_no_crossing_slots", dict(ablate=8)),
                      ("flip_no_ack_wait", dict(ablate=16)), ("flip_no_flag_poll", dict(ablate=32)), ("flip_nothing", dict(ablate=13)),
                      ("duo_write_through", dict(fl, ablate=2)), ("duo_no_exchange", dict(fl, ablate=1)),
                      ("duo_exchange_only", dict(fl, ablate=12)), ("duo_nothing", dict(fl, ablate=13))):
        with ctx.options(**opts):
            run()
            best = 1e9
            for _ in range(3):
                cs = bench.profile_pass(ctx, run, ("rk4_resident",))
                best = min(best, cs["rk4_resident"]["ms"])
            ys = run()
            terms = ctx.counters("sweep_series")["launches"]
            parts = ctx.counters("sweep_split")["launches"]
        if ref is None:
            ref = ys
        print(json.dumps({"instances": count, "variant": tag, "workgroups_per_instance": int(parts), "kernel_ms": round(best, 4),
                          "us_per_term": round(best * 1e3 / terms, 2), "ms_per_step": round(best / 20, 4),
                          "max_abs_diff_to_one_workgroup": float(np.max(np.abs(ys - ref)))}), flush=True)
