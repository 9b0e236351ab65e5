import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qiskit_dynamics_amd as qd
ctx = qd.default_context()
for name in ("mfma_f64", "mfma_f64_w1", "mfma_f64_w2", "mfma_f64_w2a16", "hbm_read", "mall_read"):
    print(name, round(ctx.microbench(name), 1))
