import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
ctx = qd.default_context()
rng = np.random.default_rng(0)
def crand(*s): return rng.uniform(-1,1,s)+1j*rng.uniform(-1,1,s)
n,k=5,2
ops=crand(k,n,n); st=crand(n,n)
y1 = crand(n)
m = qd.GeneratorModel(static_operator=st-st.conj().T, operators=[ops[0]-ops[0].conj().T], signals=[qd.Signal(lambda t: np.cos(1.3*t)+0j)])
which = sys.argv[1]
if which == "nosplit":
    ctx.set_option("split_k", 0)
r = qd.solve_lmde(m, [0,0.1], np.eye(5,dtype=complex), method="RK4", max_dt=0.1); print(which, "rk4 m=5 ok", np.linalg.norm(r.y[-1]), flush=True)
