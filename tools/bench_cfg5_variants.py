"""cfg 5 shard (128 instances, n = 4096, Magnus 2, 20 steps): kernel time per series term of the two-workgroup kernels and
their ablations, five INTERLEAVED rounds over the variants, minimum per variant (a kernel measured right after a slower one
runs ~1 us per term slower for a few launches: sequential best-of-three comparisons drift).
    python tools/bench_cfg5_variants.py [variant ...]          (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("MIDYN_LIB_AB"):      # A/B of two builds: another libmidyn.so for this process
    import qiskit_dynamics_amd._lib as _l
    _l.LIB_PATH = os.environ["MIDYN_LIB_AB"]
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
count = 128
table, _, _ = bench.sweep_table(workloads, sched.times, 0, count, 8, cfg["carrier"], cfg["t_final"])


def run():
    return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 2, y0, count, True)


# ablate bits of ell_flip_duo_kernel (results wrong): 1 no exchange, 2 write-through on one XCD too, 4 no local slots, 8 no crossing
# slots, 16 no acknowledgement wait, 32 no flag polls, 64 crossing operands loaded but not applied, 128 / 256 loads after 5/16 / 3/4,
# 512 crossing operands waited for and unpacked, one add instead of the multiply-adds
variants = [("flip", {}), ("no_apply", dict(ablate=64)), ("no_poll", dict(ablate=32)), ("wait_only", dict(ablate=512)), ("no_crossing", dict(ablate=8)),
            ("no_exchange", dict(ablate=1)), ("nothing", dict(ablate=13)), ("write_through", dict(ablate=2)),
            ("no_local", dict(ablate=4)),
            # the kernel with its saved states written to DEVICE memory (option expm_direct_out = 0: the library then copies them to the
            # host) instead of straight into the pinned result block over the bus -- what the kernel alone costs per term
            ("device_out", dict(expm_direct_out=0)), ("device_out_no_exchange", dict(expm_direct_out=0, ablate=1)),
            ("device_out_nothing", dict(expm_direct_out=0, ablate=13)),
            ("with_elements", dict(ell_sweep_flip=0)), ("one_workgroup", dict(ell_sweep_duo=0)), ("one_workgroup_with_elements", dict(ell_sweep_duo=0, ell_sweep_flip=0))]
if len(sys.argv) > 1:
    variants = [v for v in variants if v[0] in sys.argv[1:]]
best = {}
for rnd in range(5):
    for tag, opts in variants:
        with ctx.options(**opts):
            cs = bench.profile_pass(ctx, run, ("rk4_resident",))
            terms = ctx.counters("sweep_series")["launches"]
        us = cs["rk4_resident"]["ms"] * 1e3 / terms
        best[tag] = min(best.get(tag, 1e9), us)
print({k: round(v, 2) for k, v in best.items()}, flush=True)
