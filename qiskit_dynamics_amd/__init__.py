"""qiskit_dynamics_amd -- MI355X-native ODE-RHS hot path behind the qiskit-dynamics model/solver API.

Only what the hot path needs: host-side signals / rotating frame / model
build, the ctypes binding of libmidyn.so (HIP kernels for gfx950) and the fixed-step solvers.
Importing the package does not touch the GPU; the first model or context does, and it raises
``HipLibraryError`` if libmidyn.so or a HIP device is missing (there is no CPU fallback).
"""
from ._lib import DynamicsError, HipLibraryError, Context, Stack, Rk4Plan, ExpmPlan, default_context
from .signals import Signal, DiscreteSignal, SignalSum, DiscreteSignalSum, SignalList
from .rotating_frame import RotatingFrame
from .models import GeneratorModel, HamiltonianModel, LindbladModel
from .solvers import Solver, solve_lmde, solve_ode
from .perturbative import DysonSolver, MagnusSolver, ExpansionModel

__all__ = [
    "DynamicsError", "HipLibraryError", "Context", "Stack", "Rk4Plan", "ExpmPlan", "default_context",
    "Signal", "DiscreteSignal", "SignalSum", "DiscreteSignalSum", "SignalList", "RotatingFrame",
    "GeneratorModel", "HamiltonianModel", "LindbladModel", "Solver", "solve_lmde", "solve_ode",
    "DysonSolver", "MagnusSolver", "ExpansionModel",
]
__version__ = "0.1.0"
