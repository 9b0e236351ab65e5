"""Random shapes through the DEFAULT routes of the sweep solvers (tools/fuzz_routes.py holds the generator and its description):
dimension 2 .. 400, 1 .. 12 operators of random plane kinds, static operator / frame diagonal or none, 1 .. 4100 instances,
shared or per-instance initial states with 1 or 3 columns, ragged step sizes with a t_eval point, forwards / backwards, RK4 and
scipy_expm with Magnus order 1 / 2.  Every case is solved on the default routes (one-launch sweep kernels, combine + apply,
one-wave kernels) and with those routes switched off (MFMA GEMM / work-list kernels, one launch per product): 1e-11 between the
two, and the first, a middle and the last instance against the NumPy oracle at 1e-9
(reference: models/operator_collections.py:101-134, solvers/fixed_step_solvers.py:43-108,321-403).

Needs a real MI355X (`pytest -m gpu`).  `python tools/fuzz_routes.py --cases 500 --seed 1000` runs more of the same.
"""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("block", range(4))
def test_random_shapes_default_routes_vs_reference_routes_and_oracle(block):
    import fuzz_routes
    import qiskit_dynamics_amd as qd
    from oracle import dynamics_oracle as orc

    qd.default_context()
    bad = [seed for seed in range(10 * block, 10 * block + 10) if not fuzz_routes.run_case(qd, orc, seed, verbose=False)]
    assert not bad, f"failing seeds: {bad} (python tools/fuzz_routes.py --seed <s> --cases 1)"


@pytest.mark.parametrize("block", range(3))
def test_random_lindblad_models_vs_oracle(block):
    """tools/fuzz_lindblad.py: random open-system models (vectorised and matrix form, every operator-group pattern, no / diagonal /
    full frame, single solves and sweeps, RK4 and scipy_expm with Magnus order 1 .. 3) through Solver.solve against the oracle's
    restatement of models/lindblad_model.py:100-212,410-538 and models/operator_collections.py:451-567,851-1061 at 1e-9."""
    import fuzz_lindblad
    import qiskit_dynamics_amd as qd
    from oracle import dynamics_oracle as orc

    qd.default_context()
    bad = [seed for seed in range(1000 + 12 * block, 1000 + 12 * block + 12)
           if not fuzz_lindblad.run_case(qd, orc, seed, verbose=False)[0]]
    assert not bad, f"failing seeds: {bad} (python tools/fuzz_lindblad.py --seed <s> --cases 1)"
