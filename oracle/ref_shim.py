"""Dependency shim that lets the UNMODIFIED reference hot path be imported in the build container.

TEST INFRASTRUCTURE ONLY -- container only.  Nothing here is imported by the product package
(`qiskit_dynamics_amd`), by `bench.py`, or by any `-m gpu` test: `/root/reference` does not exist
on the GPU box.  The only consumer is `oracle/gen_golden.py`, which captures (inputs, outputs) of
the real reference into `tests/golden/*.npz`.

The reference (`/root/reference/qiskit_dynamics`, v0.6.0) imports two third-party packages that
are not installed here: `arraylias` (pure dispatch, no arithmetic) and `qiskit` (types/errors).
This module installs minimal stand-ins for both in `sys.modules` and registers `qiskit_dynamics`
/ `qiskit_dynamics.solvers` as bare namespace modules so that their package `__init__`s (which
pull in `backend/`, `pulse/`, `perturbative_solvers/` -> `multiset`) are not executed.  All hot
path ARITHMETIC is then executed by real NumPy / SciPy through the reference's own code.

No reference source is contained in this file.
"""
import importlib
import os
import sys
import types

import numpy as np
import scipy

REF = os.environ.get("QD_REFERENCE_ROOT", "/root/reference")


class LibraryError(Exception):
    pass


class _AliasedModule:
    """`alias()` / `alias(like=x)` result: attribute access dispatches on the first argument."""

    def __init__(self, alias, like=None, prefix=""):
        self._a, self._like, self._p = alias, like, prefix

    def __getattr__(self, name):
        path = f"{self._p}.{name}" if self._p else name
        tgt = self._a._static(path)
        if isinstance(tgt, types.ModuleType):
            return _AliasedModule(self._a, self._like, path)
        if self._like is not None:
            return self._a._function(self._a._lib_of(self._like), path)
        if tgt is not None and (not callable(tgt) or isinstance(tgt, type)):
            return tgt

        def dispatch(*args, **kw):
            lib = None
            if args:
                libs = self._a.infer_libs(args[0])
                lib = libs[0] if libs else None
            return self._a._function(lib, path)(*args, **kw)

        return dispatch


class _Alias:
    def __init__(self, base):
        self._base = base
        self._types = {"numpy": [np.ndarray, np.number, int, float, complex]}
        self._funcs, self._defaults, self._fallbacks = {}, {}, {}

    def _static(self, path):
        obj = self._base
        for p in path.split("."):
            obj = getattr(obj, p, None)
            if obj is None:
                return None
        return obj

    def register_type(self, t, lib):
        self._types.setdefault(lib, []).append(t)

    def registered_types(self):
        return tuple(t for ts in self._types.values() for t in ts)

    def registered_libs(self):
        return tuple(self._types.keys())

    def infer_libs(self, obj):
        if isinstance(obj, (list, tuple)):
            return self.infer_libs(obj[0]) if len(obj) else ()
        return tuple(l for l, ts in self._types.items() if isinstance(obj, tuple(ts)))

    def _lib_of(self, like):
        if isinstance(like, str):
            return like
        libs = self.infer_libs(like)
        return libs[0] if libs else None

    @staticmethod
    def _deco(table, key):
        def d(func):
            table[key] = func
            return func

        return d

    def register_function(self, func=None, lib=None, path=None):
        if func is not None:
            self._funcs[(lib, path)] = func
            return func
        return self._deco(self._funcs, (lib, path))

    def register_default(self, func=None, path=None):
        if func is not None:
            self._defaults[path] = func
            return func
        return self._deco(self._defaults, path)

    def register_fallback(self, func=None, path=None):
        if func is not None:
            self._fallbacks[path] = func
            return func
        return self._deco(self._fallbacks, path)

    def _function(self, lib, path):
        if lib is None:
            if path in self._defaults:
                return self._defaults[path]
            lib = "numpy"
        if (lib, path) in self._funcs:
            return self._funcs[(lib, path)]
        if lib == "numpy":
            f = self._static(path)
            if f is not None:
                return f
        if path in self._fallbacks:
            return self._fallbacks[path]
        raise LibraryError(f"no function {path} for lib {lib}")

    def __call__(self, like=None, path=None):
        if path is not None:
            return self._function(self._lib_of(like) if like is not None else None, path)
        return _AliasedModule(self, like)


class QiskitError(Exception):
    pass


class Operator:
    def __init__(self, data):
        self.data = np.asarray(data, dtype=complex)

    def __array__(self, dtype=None, copy=None):
        return self.data


def is_hermitian_matrix(mat, rtol=1e-5, atol=1e-8):
    mat = np.asarray(mat)
    return mat.ndim == 2 and bool(np.allclose(mat, mat.conj().T, rtol=rtol, atol=atol))


def _mod(name, **attrs):
    m = sys.modules.get(name) or types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _cls(name):
    return type(name, (), {})


def _bare_pkg(name, relpath):
    m = types.ModuleType(name)
    m.__path__ = [os.path.join(REF, relpath)]
    sys.modules[name] = m
    return m


_LOADED = False


def load_reference():
    """Install the stand-ins and import the reference hot-path modules. Idempotent."""
    global _LOADED
    if _LOADED:
        return sys.modules["qiskit_dynamics"]
    if not os.path.isdir(os.path.join(REF, "qiskit_dynamics")):
        raise RuntimeError(f"reference tree not found under {REF} (container-only tool)")
    import matplotlib

    matplotlib.use("Agg")

    al = types.ModuleType("arraylias")
    al.numpy_alias = lambda: _Alias(np)
    al.scipy_alias = lambda: _Alias(scipy)
    alx = types.ModuleType("arraylias.exceptions")
    alx.LibraryError = LibraryError
    al.exceptions = alx
    sys.modules["arraylias"] = al
    sys.modules["arraylias.exceptions"] = alx

    _mod("qiskit", QiskitError=QiskitError)
    _mod("qiskit.quantum_info", Operator=Operator, SuperOp=_cls("SuperOp"),
         DensityMatrix=_cls("DensityMatrix"))
    _mod("qiskit.quantum_info.operators", Operator=Operator)
    _mod("qiskit.quantum_info.operators.predicates", is_hermitian_matrix=is_hermitian_matrix)
    _mod("qiskit.pulse", Schedule=_cls("Schedule"), ScheduleBlock=_cls("ScheduleBlock"))
    _mod("qiskit.pulse.transforms", block_to_schedule=lambda x: x)
    _mod("qiskit.circuit", Gate=_cls("Gate"), QuantumCircuit=_cls("QuantumCircuit"))
    _mod("qiskit.quantum_info.operators.base_operator", BaseOperator=_cls("BaseOperator"))
    _mod("qiskit.quantum_info.operators.channel")
    _mod("qiskit.quantum_info.operators.channel.quantum_channel",
         QuantumChannel=_cls("QuantumChannel"))
    _mod("qiskit.quantum_info.states")
    _mod("qiskit.quantum_info.states.quantum_state", QuantumState=_cls("QuantumState"))

    qd = _bare_pkg("qiskit_dynamics", "qiskit_dynamics")
    importlib.import_module("qiskit_dynamics.arraylias")
    from qiskit_dynamics.arraylias import alias as _al  # noqa: E402

    for k in ("DYNAMICS_NUMPY_ALIAS", "DYNAMICS_SCIPY_ALIAS", "DYNAMICS_NUMPY", "DYNAMICS_SCIPY",
              "ArrayLike"):
        setattr(qd, k, getattr(_al, k))
    importlib.import_module("qiskit_dynamics.signals")
    importlib.import_module("qiskit_dynamics.models")
    _bare_pkg("qiskit_dynamics.solvers", "qiskit_dynamics/solvers")
    importlib.import_module("qiskit_dynamics.solvers.solver_functions")
    _mod("qiskit_dynamics.pulse", InstructionToSignals=None)
    from qiskit_dynamics.signals import Signal, DiscreteSignal  # noqa: E402
    from qiskit_dynamics.models import RotatingFrame  # noqa: E402

    qd.Signal, qd.DiscreteSignal, qd.RotatingFrame = Signal, DiscreteSignal, RotatingFrame
    importlib.import_module("qiskit_dynamics.solvers.solver_classes")
    _LOADED = True
    return qd


# ---------------------------------------------------------------------------------------------
# Row f4 (perturbative Dyson / Magnus solvers): the reference additionally imports the third-party
# `multiset` package (setup.py: multiset>=3.0.1), which is not installed here.  `Multiset` is only
# used as a LABEL type (which perturbation indices a term belongs to): a counter with multiset
# algebra.  The stand-in below implements the published semantics of multiset.Multiset that the
# reference uses; every number is computed by the reference's own NumPy/SciPy code.
# ---------------------------------------------------------------------------------------------
class Multiset:
    """Minimal stand-in for multiset.Multiset (mutable multiset: element -> multiplicity)."""

    def __init__(self, iterable=None):
        self._c = {}
        if iterable is None:
            return
        if isinstance(iterable, Multiset):
            self._c = dict(iterable._c)
        elif isinstance(iterable, dict):
            for k_, v_ in iterable.items():
                if v_ > 0:
                    self._c[k_] = int(v_)
        else:
            for e in iterable:
                self._c[e] = self._c.get(e, 0) + 1

    # -- container protocol
    def __len__(self):
        return sum(self._c.values())

    def __iter__(self):
        for e, m in self._c.items():
            for _ in range(m):
                yield e

    def __contains__(self, e):
        return e in self._c

    def __getitem__(self, e):
        return self._c.get(e, 0)

    def __bool__(self):
        return bool(self._c)

    def __eq__(self, other):
        if isinstance(other, Multiset):
            return self._c == other._c
        return NotImplemented

    def __ne__(self, other):
        r = self.__eq__(other)
        return r if r is NotImplemented else not r

    __hash__ = None

    def __repr__(self):
        return "Multiset(%r)" % (self._c,)

    def __str__(self):
        return "{%s}" % ", ".join(str(e) for e in self)

    # -- queries
    def distinct_elements(self):
        return self._c.keys()

    def items(self):
        return self._c.items()

    def multiplicities(self):
        return self._c.values()

    def get(self, e, default=0):
        return self._c.get(e, default)

    def copy(self):
        return Multiset(self)

    def issubset(self, other):
        other = other if isinstance(other, Multiset) else Multiset(other)
        return all(other[e] >= m for e, m in self._c.items())

    def issuperset(self, other):
        other = other if isinstance(other, Multiset) else Multiset(other)
        return other.issubset(self)

    def __le__(self, other):
        return self.issubset(other)

    def __lt__(self, other):
        return self.issubset(other) and len(self) < len(other)

    def __ge__(self, other):
        return self.issuperset(other)

    def __gt__(self, other):
        return self.issuperset(other) and len(self) > len(other)

    # -- algebra
    def combine(self, *others):
        out = Multiset(self)
        for o in others:
            for e, m in Multiset(o)._c.items():
                out._c[e] = out._c.get(e, 0) + m
        return out

    __add__ = combine

    def difference(self, *others):
        out = Multiset(self)
        for o in others:
            for e, m in Multiset(o)._c.items():
                if e in out._c:
                    left = out._c[e] - m
                    if left > 0:
                        out._c[e] = left
                    else:
                        del out._c[e]
        return out

    __sub__ = difference

    def union(self, *others):
        out = Multiset(self)
        for o in others:
            for e, m in Multiset(o)._c.items():
                out._c[e] = max(out._c.get(e, 0), m)
        return out

    __or__ = union

    def intersection(self, *others):
        out = Multiset(self)
        for o in others:
            o = Multiset(o)
            out._c = {e: min(m, o[e]) for e, m in out._c.items() if o[e] > 0}
        return out

    __and__ = intersection

    # -- mutation
    def add(self, e, multiplicity=1):
        self._c[e] = self._c.get(e, 0) + multiplicity

    def remove(self, e, multiplicity=None):
        if e not in self._c:
            raise KeyError(e)
        old = self._c[e]
        if multiplicity is None or multiplicity >= old:
            del self._c[e]
        else:
            self._c[e] = old - multiplicity
        return old

    def discard(self, e, multiplicity=None):
        if e in self._c:
            return self.remove(e, multiplicity)
        return 0

    def update(self, *others):
        for o in others:
            for e, m in Multiset(o)._c.items():
                self._c[e] = self._c.get(e, 0) + m


def load_reference_perturbation():
    """`load_reference()` + the perturbation package and the Dyson / Magnus solvers (row f4)."""
    qd = load_reference()
    if "qiskit_dynamics.solvers.perturbative_solvers" in sys.modules:
        return qd
    _mod("multiset", Multiset=Multiset, FrozenMultiset=Multiset)
    importlib.import_module("qiskit_dynamics.perturbation")
    importlib.import_module("qiskit_dynamics.solvers.perturbative_solvers")
    return qd
