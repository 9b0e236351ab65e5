import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import qiskit_dynamics_amd as qd
ctx = qd.default_context()
rng = np.random.default_rng(3)
def crand(rng, *s): return rng.uniform(-1, 1, s) + 1j * rng.uniform(-1, 1, s)
for n in (4, 64, 100, 128, 300):
    a, b = crand(rng, n, n), crand(rng, n, n)
    for opt3m in (1, 0):
        ctx.set_option("complex_3m", 2 if opt3m else 0)
        print("zgemm n", n, "3m", opt3m, "err", np.max(np.abs(ctx.zgemm(a, b) - a @ b)), flush=True)
    ctx.set_option("complex_3m", 1)
    ar = a.real + 0j
    print("zgemm real A n", n, "err", np.max(np.abs(ctx.zgemm(ar, b) - ar @ b)), flush=True)
n, k = 100, 3
ops, static = crand(rng, k, n, n), crand(rng, n, n)
st = qd.Stack(ctx, ops, static, None)
c = rng.uniform(-1, 1, k)
for m in (1, 5, 32, 200):
    y = crand(rng, n, m)
    ref = (static + np.tensordot(c, ops, axes=1)) @ y
    print("eval_rhs m", m, "err", np.max(np.abs(st.eval_rhs(c, 0.0, y) - ref)), flush=True)
