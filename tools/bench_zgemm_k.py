"""The 3M dense product on 64 x 64 tiles (the kernel of midyn_expm's products): kernel time against K at M = N = 1024 and 4096 --
intercept = the fixed cost of a launch (ramp, first tile, recombination + store, tail), slope = the steady k loop.
    python tools/bench_zgemm_k.py [out.md]          (on the GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd  # noqa: E402

ctx = qd.default_context(0)
rng = np.random.default_rng(3)
lines = ["# zgemm_seg_kernel<64, 64, 2, 2, 16, 4> (3M): kernel time against K (round 6)", "",
         "`tools/bench_zgemm_k.py`: C[M][N] = A[M][K] B[K][N], complex128, 3M (6 real flops per complex multiply-add), HIP events around "
         "the launch (class zgemm, profile on), minimum of 7.", "",
         "| M = N | K | tiles 64 x 64 | us | TFLOP/s | frac |", "|---|---|---|---|---|---|"]
fit = {}
for mn in (1024, 4096):
    for k in (256, 512, 1024, 2048, 4096):
        if mn == 4096 and k > 2048:
            continue
        a = rng.standard_normal((mn, k)) + 1j * rng.standard_normal((mn, k))
        b = rng.standard_normal((k, mn)) + 1j * rng.standard_normal((k, mn))
        row = []
        for t3 in (0,):
            best = 1e9
            with ctx.options(complex_3m=2):
                ctx.zgemm(a, b)
                for _ in range(7):
                    ctx.reset_counters()
                    ctx.set_option("profile", 1)
                    try:
                        ctx.zgemm(a, b)
                        ctx.synchronize()
                        c = ctx.counters("zgemm")
                    finally:
                        ctx.set_option("profile", 0)
                    best = min(best, c["ms"] / max(c["launches"], 1))
            row.append(best * 1e3)
        fl = 6.0 * mn * mn * k
        fit.setdefault(mn, []).append((k, row[0]))
        lines.append(f"| {mn} | {k} | {(mn // 64) ** 2} | {row[0]:.1f} | {fl / row[0] / 1e6:.1f} | {fl / row[0] / 1e6 / 78.6:.3f} |")
        print(lines[-1], flush=True)
lines.append("")
for mn, pts in fit.items():
    ks = np.array([p[0] for p in pts], float)
    us = np.array([p[1] for p in pts], float)
    slope, icpt = np.polyfit(ks, us, 1)
    asym = 6.0 * mn * mn / slope / 1e6 / 78.6
    lines.append(f"* M = N = {mn}: time = {icpt:.1f} us + {slope * 1e3:.2f} us per 1000 of K: the steady k loop runs at {asym:.3f} of 78.6 TFLOP/s, "
                 f"a launch costs {icpt:.1f} us on top (at K = 1024: {100 * icpt / (icpt + slope * 1024):.0f} % of the launch).")
    print(lines[-1], flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "w") as f:
        f.write("\n".join(lines) + "\n")
