"""General complex model on the combine + apply route: what do the static operator's rows (the C input of the combining MFMAs, one
`v_mov_b64_dpp row_newbcast` per MFMA) cost?  The SAME eight complex operator planes (rhs_combine_kernel<2, 2, *>) with a complex
static operator (STAT = 3: what `dense_complex` on the bench line runs), an imaginary one (STAT = 2), and none (STAT = 0), n = 1024,
4096 instances, per batched evaluation; executed flops = (4 (NRE4 + NIM4) MFMA-FMAs + 4 vector FMAs) x 2 per (row, kk, instance) -- the
static rows add none (they ride in as the accumulator input).   python tools/dense_complex_ablation.py [out.md]   """
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402
from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points  # noqa: E402

ctx = qd.default_context()
rng = np.random.default_rng(1)
n, B, S = 1024, 4096, 14
PEAK = 78.6


def ops(kinds):
    out = []
    for kd in kinds:
        a = (rng.uniform(-1, 1, (n, n)) + 1j * rng.uniform(-1, 1, (n, n))) * 0.01
        out.append(a.real + 0j if kd == "r" else (1j * a.imag if kd == "i" else a))
    return np.array(out)


sched = FixedStepSchedule([0.0, 0.1], None, 0.005, _rk4_points)
rows = sched.step_rows[:S]
nr = int(rows.max()) + 1
y0 = np.zeros((n, 1), complex)
y0[0] = 1
lines = ["# General complex operators on combine + apply: the cost of the static operator's rows (round 6)", "",
         "`tools/dense_complex_ablation.py`: n = 1024, 4096 instances, 8 complex operators = 8 real + 8 imaginary planes "
         "(`rhs_combine_kernel<2, 2, STAT>`), ms per batched RHS evaluation (HIP events around 48 launches, best of 3), executed "
         "TFLOP/s and fraction of 78.6.", "",
         "| planes | static operator | kernel | ms per evaluation | executed GFLOP | TFLOP/s | frac |", "|---|---|---|---|---|---|---|"]
res = {}
for kinds, stat in (("cccccccc", None), ("cccccccc", "i"), ("cccccccc", "c"), ("iiiiiiii", None), ("iiiiiiii", "i"), ("iiiiiiii", "c")):
    st = qd.Stack(ctx, ops(kinds), None if stat is None else ops(stat)[0], None)
    table = rng.uniform(-1, 1, (B, nr, len(kinds)))
    ctx.set_option("combine", 2)
    p = qd.Rk4Plan(st, sched.times[:nr], table, rows, sched.step_h[:S], y0, B, True)
    p.run(0, 2)
    ctx.synchronize()
    best = None
    for rep in range(3):
        ctx.timer_start()
        p.run(2, S)
        ms = ctx.timer_stop() / (4 * (S - 2))
        best = ms if best is None else min(best, ms)
    info = ctx.counters("combine_info")
    code = int(info["ms"])
    nre4, nim4, stt = code // 100, (code // 10) % 10, code % 10
    kinds_n = int(nre4 > 0 or (stt & 1)) + int(nim4 > 0 or (stt & 2))
    fl = (2.0 * 4 * (nre4 + nim4) + 2.0 * (2 if kinds_n == 1 else 4)) * info["launches"] * 16 * 32 * B
    p.close()
    st.close()
    ctx.set_option("combine", 1)
    res[(kinds, stat)] = best
    lines.append(f"| {'8 complex' if kinds[0] == 'c' else '8 imaginary'} | {'none' if stat is None else ('imaginary' if stat == 'i' else 'complex')} | "
                 f"`rhs_combine_kernel<{nre4}, {nim4}, {stt}>` | {best:.3f} | {fl / 1e9:.1f} | {fl / (best * 1e-3) / 1e12:.2f} | {fl / (best * 1e-3) / 1e12 / PEAK:.3f} |")
    print(lines[-1], flush=True)
c0, c3 = res[("cccccccc", None)], res[("cccccccc", "c")]
lines += ["", f"The sixteen `row_newbcast` moves per kk step (one per MFMA accumulator input: 4 rows x 2 tiles x 2 planes, shared by the two "
          f"instance groups of a wave) and the one extra 8-byte load per lane cost {c3 - c0:.3f} ms of {c3:.3f} ms = {100 * (c3 - c0) / c3:.1f} % of "
          "the general complex kernel: every one of them is a vector instruction on the SIMD's one fp64 pipe (measured_peaks."
          "fp64_mfma_and_vector_fma_share_a_pipe on the bench line).  Without a static operator the same planes run at "
          f"{res[('cccccccc', None)]:.3f} ms."]
print("\n".join(lines[-2:]))
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(lines) + "\n")
