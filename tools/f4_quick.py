import sys, json
sys.path.insert(0, "/root/repo")
import qiskit_dynamics_amd as qd
from tools.bench_legs.rows import leg_perturbative
ctx = qd.default_context(0)
out = leg_perturbative(qd, ctx)
for k, v in out.items():
    print(k, v["solve_s"], v["kernel_ms"], v["steps_per_padded_block"], v["one_step_per_block"], v["roofline"]["frac"], v["unitarity_defect"])
