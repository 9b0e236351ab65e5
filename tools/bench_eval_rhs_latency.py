import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
ctx = qd.default_context()
rng = np.random.default_rng(0)
def crand(*s): return rng.uniform(-1,1,s)+1j*rng.uniform(-1,1,s)
for n,k in ((4,2),(64,6),(1024,8)):
    ops=crand(k,n,n); st=crand(n,n); fr = rng.normal(size=n)
    stack = qd.Stack(ctx, ops, st, fr)
    c = rng.uniform(-1,1,k); y = crand(n)
    e = np.exp(1j*fr*0.3)
    ref = np.conj(e)*((np.tensordot(c,ops,axes=1)+st)@(e*y))
    out = stack.eval_rhs(c,0.3,y)
    t0=time.perf_counter()
    for i in range(200): out = stack.eval_rhs(c,0.3+1e-3*i,y)
    dt=(time.perf_counter()-t0)/200
    e = np.exp(1j*fr*(0.3+1e-3*199)); ref = np.conj(e)*((np.tensordot(c,ops,axes=1)+st)@(e*y))
    print(n, "eval_rhs us/call", round(dt*1e6,1), "err", float(np.max(np.abs(out-ref))), flush=True)
