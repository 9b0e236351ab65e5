"""Dense midyn_expm (row a11) at n = 1024 / 2048: tile and split-K choices of its 3M products, A/B in one process.
    python tools/bench_expm_tiles.py          (on the GPU box; options through midyn_ctx_set_option)
Kernel time from the library's HIP-event counters (classes zgemm + elementwise, profile on), minimum of five calls per
variant, variants interleaved."""
import json, sys
import numpy as np
sys.path.insert(0, ".")
import qiskit_dynamics_amd as qd

ctx = qd.default_context(0)
VARIANTS = [("default", {}), ("splits2", {"force_splits": 2}), ("splits4", {"force_splits": 4}),
            ("4m", {"complex_3m": 0}), ("4m_splits4", {"complex_3m": 0, "force_splits": 4}),
            ("4m_tile64", {"complex_3m": 0, "force_tile": 64}), ("4m_tile64_splits2", {"complex_3m": 0, "force_tile": 64, "force_splits": 2})]
rng = np.random.default_rng(11)
for n in (1024, 2048):
    a = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    a = a - a.conj().T
    a *= 5.0 / np.abs(a).sum(axis=0).max()
    ref = ctx.expm(a)
    best = {name: (1e9, None) for name, _ in VARIANTS}
    for rnd in range(5):
        for name, opts in VARIANTS:
            with ctx.options(**opts):
                ctx.expm(a)
                ctx.reset_counters()
                ctx.set_option("profile", 1)
                try:
                    e = ctx.expm(a)
                    ctx.synchronize()
                    cz, ce = ctx.counters("zgemm"), ctx.counters("elementwise")
                    flops = ctx.executed_flops("zgemm")
                finally:
                    ctx.set_option("profile", 0)
            ms = cz["ms"]
            if ms < best[name][0]:
                best[name] = (ms, {"n": n, "variant": name, "zgemm_ms": round(ms, 4), "elementwise_ms": round(ce["ms"], 4),
                                   "launches": int(cz["launches"]), "executed_tflops": round(flops / ms / 1e9, 2),
                                   "useful_tflops_8n3": round(8.0 * n ** 3 * cz["launches"] / ms / 1e9, 2),
                                   "max_abs_diff_to_default": float(np.abs(e - ref).max())})
    for name, _ in VARIANTS:
        print(json.dumps(best[name][1]), flush=True)
