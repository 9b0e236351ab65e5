"""ctypes binding of libmidyn.so (C-ABI declared in include/midyn.h).

There is NO CPU fallback: if the shared library is missing, or no HIP device is visible when a
context is requested, this module raises -- it never routes work to NumPy.
"""
from __future__ import annotations

import ctypes
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmidyn.so")

# every symbol include/midyn.h declares (__graft_entry__.build() and tests/test_host_logic.py check the .so exports all of them; tests/abi_probe.c compiles the header as plain C)
ABI_SYMBOLS = [
    "midyn_ctx_create", "midyn_ctx_destroy", "midyn_ctx_synchronize", "midyn_last_error",
    "midyn_ctx_set_option", "midyn_ctx_get_option", "midyn_stack_packed_bytes", "midyn_stack_create", "midyn_stack_create_lindblad",
    "midyn_stack_adopt", "midyn_stack_antiherm_defect",
    "midyn_stack_destroy", "midyn_stack_info", "midyn_stack_segment_modes", "midyn_eval_generator", "midyn_eval_rhs",
    "midyn_rk4_solve", "midyn_expm", "midyn_expm_solve", "midyn_zgemm", "midyn_rk4_plan_create",
    "midyn_rk4_plan_run", "midyn_rk4_plan_fetch", "midyn_rk4_plan_destroy", "midyn_expm_plan_create", "midyn_expm_plan_run",
    "midyn_expm_plan_fetch", "midyn_expm_plan_destroy", "midyn_get_counters",
    "midyn_reset_counters", "midyn_microbench", "midyn_lindblad_create", "midyn_lindblad_destroy",
    "midyn_lindblad_rhs", "midyn_lindblad_rk4_solve", "midyn_sigtable_create", "midyn_sigtable_data",
    "midyn_sigtable_fetch", "midyn_sigtable_destroy", "midyn_parallel_solve", "midyn_expansion_create",
    "midyn_expansion_destroy", "midyn_expansion_solve", "midyn_expansion_set_monomials", "midyn_expansion_solve_coeffs", "midyn_host_alloc", "midyn_host_free",
    "midyn_ctx_timer", "midyn_stack_block_info",
    "midyn_comm_get_unique_id", "midyn_comm_init_rank", "midyn_comm_destroy", "midyn_comm_count", "midyn_stack_create_empty",
    "midyn_stack_broadcast", "midyn_stack_broadcast_from",
]


class DynamicsError(Exception):
    """Raised where the reference raises ``QiskitError``."""


class HipLibraryError(RuntimeError):
    """libmidyn.so missing / failed / no GPU: the HIP path cannot run (and nothing else will)."""


_lib = None
_lock = threading.Lock()

_vp = ctypes.c_void_p
_ci = ctypes.c_int
_cd = ctypes.c_double
_cll = ctypes.c_longlong


HIP_RUNTIME = None  # path of the HIP runtime libmidyn's hip* symbols were bound to


def _preload_hip_runtime():
    """libmidyn.so has no DT_NEEDED on libamdhip64: exactly ONE HIP runtime must serve the process.
    If torch is already imported (torchrun / RCCL plumbing) use the runtime bundled with torch, else
    the system ROCm one.  Override with MIDYN_HIP_RUNTIME=torch|system|/path/to/libamdhip64.so."""
    global HIP_RUNTIME
    choice = os.environ.get("MIDYN_HIP_RUNTIME", "auto")
    cands = []
    if choice == "torch" or (choice == "auto" and "torch" in sys.modules):
        import torch

        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    elif choice not in ("auto", "system"):
        cands.append(choice)
    cands += ["/opt/rocm/lib/libamdhip64.so.7", "/opt/rocm/lib/libamdhip64.so", "libamdhip64.so.7",
              "libamdhip64.so"]
    errors = []
    for c in cands:
        if os.path.sep in c and not os.path.exists(c):
            continue
        try:
            ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            HIP_RUNTIME = c
            return
        except OSError as e:  # pragma: no cover
            errors.append(f"{c}: {e}")
    raise HipLibraryError("no HIP runtime (libamdhip64) could be loaded: " + "; ".join(errors))


RCCL_LIBRARY = None  # path of the librccl the C side resolves its nccl* entry points from


def preload_rccl():
    """Make ONE librccl global in the process before the first midyn_comm_* call: the one next to the HIP runtime
    in use (torch's bundled copy when libmidyn is bound to torch's runtime, the system ROCm one otherwise), so
    that RCCL and libmidyn share a HIP runtime.  Override with MIDYN_RCCL_LIB=/path/to/librccl.so."""
    global RCCL_LIBRARY
    if RCCL_LIBRARY is not None:
        return RCCL_LIBRARY
    load()
    cands = []
    if os.environ.get("MIDYN_RCCL_LIB"):
        cands.append(os.environ["MIDYN_RCCL_LIB"])
    if HIP_RUNTIME and os.path.sep in HIP_RUNTIME:
        d = os.path.dirname(HIP_RUNTIME)
        cands += [os.path.join(d, "librccl.so"), os.path.join(d, "librccl.so.1")]
    cands += ["/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"]
    errors = []
    for c in cands:
        if os.path.sep in c and not os.path.exists(c):
            continue
        try:
            ctypes.CDLL(c, mode=ctypes.RTLD_GLOBAL)
            RCCL_LIBRARY = c
            return c
        except OSError as e:  # pragma: no cover
            errors.append(f"{c}: {e}")
    raise HipLibraryError("librccl could not be loaded: " + "; ".join(errors))


def load():
    """Load libmidyn.so (once) and set the prototypes."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). qiskit_dynamics_amd has no CPU fallback."
            )
        _preload_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        P = ctypes.POINTER
        lib.midyn_last_error.restype = ctypes.c_char_p
        lib.midyn_last_error.argtypes = [_vp]
        lib.midyn_ctx_create.argtypes = [_ci, P(_vp)]
        lib.midyn_ctx_destroy.argtypes = [_vp]
        lib.midyn_ctx_synchronize.argtypes = [_vp]
        lib.midyn_ctx_set_option.argtypes = [_vp, ctypes.c_char_p, _cll]
        lib.midyn_ctx_get_option.argtypes = [_vp, ctypes.c_char_p, P(_cll)]
        lib.midyn_stack_packed_bytes.argtypes = [_ci, _ci, _ci, P(ctypes.c_size_t)]
        lib.midyn_stack_create.argtypes = [_vp, _ci, _ci, _vp, _vp, _vp, _vp, P(_vp)]
        lib.midyn_stack_adopt.argtypes = [_vp, _ci, _ci, _ci, _ci, _vp, P(_vp)]
        lib.midyn_stack_create_lindblad.argtypes = [_vp, _ci, _vp, _ci, _vp, _ci, _vp, _ci, _vp, _vp, P(_vp)]
        lib.midyn_stack_antiherm_defect.argtypes = [_vp, P(_cd)]
        lib.midyn_stack_destroy.argtypes = [_vp]
        lib.midyn_stack_info.argtypes = [_vp, P(_cll)]
        lib.midyn_stack_segment_modes.argtypes = [_vp, P(_ci)]
        lib.midyn_eval_generator.argtypes = [_vp, _vp, _cd, _vp]
        lib.midyn_eval_rhs.argtypes = [_vp, _vp, _cd, _vp, _ci, _vp]
        lib.midyn_rk4_solve.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _vp,
                                        _ci, _vp]
        lib.midyn_expm.argtypes = [_vp, _ci, _ci, _vp, _vp, _vp]
        lib.midyn_expm_solve.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _ci,
                                         _vp, _ci, _vp]
        lib.midyn_zgemm.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _vp]
        lib.midyn_rk4_plan_create.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci,
                                              P(_vp)]
        lib.midyn_rk4_plan_run.argtypes = [_vp, _ci, _ci]
        lib.midyn_rk4_plan_fetch.argtypes = [_vp, _vp]
        lib.midyn_rk4_plan_destroy.argtypes = [_vp]
        lib.midyn_expm_plan_create.argtypes = [_vp, _ci, _ci, _ci, _vp, _ci, _vp, _vp, _vp, _ci, _ci, _vp, _ci, ctypes.POINTER(_vp)]
        lib.midyn_expm_plan_run.argtypes = [_vp, _vp, _vp]
        lib.midyn_expm_plan_fetch.argtypes = [_vp, _vp]
        lib.midyn_expm_plan_destroy.argtypes = [_vp]
        lib.midyn_get_counters.argtypes = [_vp, ctypes.c_char_p, P(_cd)]
        lib.midyn_reset_counters.argtypes = [_vp]
        lib.midyn_microbench.argtypes = [_vp, ctypes.c_char_p, P(_cd)]
        lib.midyn_lindblad_create.argtypes = [_vp, _vp, _ci, _ci, _ci, _vp, P(_vp)]
        lib.midyn_lindblad_destroy.argtypes = [_vp]
        lib.midyn_lindblad_rhs.argtypes = [_vp, _vp, _cd, _vp, _ci, _vp]
        lib.midyn_lindblad_rk4_solve.argtypes = [_vp, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _vp, _ci, _vp]
        lib.midyn_parallel_solve.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _ci, _vp, _vp, _vp, _ci, _ci,
                                             _vp, _ci, _vp]
        lib.midyn_expansion_create.argtypes = [_vp, _ci, _ci, _vp, _vp, _vp, _ci, P(_vp)]
        lib.midyn_expansion_destroy.argtypes = [_vp]
        lib.midyn_expansion_solve.argtypes = [_vp, _ci, _ci, _vp, _ci, _vp, _ci, _vp]
        lib.midyn_host_alloc.argtypes = [ctypes.c_size_t, P(_vp)]
        lib.midyn_host_free.argtypes = [_vp]
        lib.midyn_expansion_set_monomials.argtypes = [_vp, _ci, _ci, _vp]
        lib.midyn_expansion_solve_coeffs.argtypes = [_vp, _ci, _ci, _vp, _ci, _vp, _ci, _vp]
        lib.midyn_sigtable_create.argtypes = [_vp, _ci, _ci, _ci, _vp, _vp, _vp, _vp, _vp, P(_vp)]
        lib.midyn_sigtable_data.argtypes = [_vp, P(_vp), _vp]
        lib.midyn_sigtable_fetch.argtypes = [_vp, _vp]
        lib.midyn_sigtable_destroy.argtypes = [_vp]
        lib.midyn_ctx_timer.argtypes = [_vp, _ci, P(_cd)]
        lib.midyn_stack_block_info.argtypes = [_vp, P(_cd)]
        lib.midyn_comm_get_unique_id.argtypes = [_vp]
        lib.midyn_comm_init_rank.argtypes = [_vp, _ci, _ci, _vp, P(_vp)]
        lib.midyn_comm_destroy.argtypes = [_vp, _vp]
        lib.midyn_comm_count.argtypes = [_vp, _vp, ctypes.POINTER(ctypes.c_int)]
        lib.midyn_stack_create_empty.argtypes = [_vp, _ci, _ci, _ci, _ci, P(_vp)]
        lib.midyn_stack_broadcast.argtypes = [_vp, _vp, _ci]
        lib.midyn_stack_broadcast_from.argtypes = [_vp, _vp, _vp, _ci]
        for name in ABI_SYMBOLS:
            if name != "midyn_last_error":
                getattr(lib, name).restype = _ci
        _lib = lib
        return lib


# ---- result arrays in pinned host memory -------------------------------------------------------------------------------
# A solve's results come back over PCIe into an array the binding allocates.  Into pageable memory the runtime stages the
# copy, and a fresh NumPy array takes its page faults inside it: the 8.4 MB of a cfg 5 shard (BASELINE configs[4]) cost
# 0.19-0.38 ms depending on where the allocator put the array -- of a 2.5 ms solve.  Large results therefore live in PINNED
# blocks (hipHostMalloc: the device writes them directly) that return to a small cache when the last view of the array is
# collected.  MIDYN_PINNED_RESULTS=0 (or a failed hipHostMalloc): plain np.empty.
_PINNED_MIN_BYTES = 1 << 20        # below: the copy is latency, not bandwidth
_PINNED_MAX_BYTES = 256 << 20      # above: page-locking that much for as long as the caller keeps the result is not ours to decide
_PINNED_CACHE_MAX = 1 << 30
_PINNED_LIVE_MAX = int(os.environ.get("MIDYN_PINNED_BUDGET_MB", "2048")) << 20   # page-locked bytes in the hands of callers at most: beyond, np.empty
_pinned_cache = {}          # nbytes -> [pointers]
_pinned_cached_bytes = 0
_pinned_live_bytes = 0
# RLock: _PinnedBlock.__del__ can run inside a garbage collection that starts while this thread holds the lock;
# the finalizer itself allocates nothing under it (the per-size lists are made in result_array)
_pinned_lock = threading.RLock()
_hiprt = None


class _LibraryHostBlocks:
    """Page-locked blocks through the library (``midyn_host_alloc`` / ``midyn_host_free``, include/midyn.h): the library then
    knows a result array by its address and hands the device address to the kernels that write results directly, without asking
    the HIP runtime about the pointer at every solve (60-80 us).  Method names: those of the runtime calls behind them."""

    def __init__(self):
        self.lib = load()

    def hipHostMalloc(self, pp, nbytes, flags):  # pylint: disable=invalid-name,unused-argument
        return self.lib.midyn_host_alloc(nbytes, pp)

    def hipHostFree(self, p):  # pylint: disable=invalid-name
        return self.lib.midyn_host_free(p)


def _hip_runtime():
    global _hiprt
    if _hiprt is None:
        _hiprt = _LibraryHostBlocks()
    return _hiprt


class _PinnedBlock:
    """Owns one pinned block; gives it back to the cache (or to the runtime) when collected."""

    __slots__ = ("ptr", "nbytes")

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def __del__(self):
        global _pinned_cached_bytes, _pinned_live_bytes
        try:
            with _pinned_lock:
                _pinned_live_bytes -= self.nbytes
                blocks = _pinned_cache.get(self.nbytes)         # (made by result_array before the block existed)
                if blocks is not None and _pinned_cached_bytes + self.nbytes <= _PINNED_CACHE_MAX:
                    blocks.append(self.ptr)
                    _pinned_cached_bytes += self.nbytes
                    return
            _hip_runtime().hipHostFree(_vp(self.ptr))
        except Exception:  # pylint: disable=broad-except
            pass               # (interpreter shutdown: the process is going away with its pinned memory)


def result_array(shape, dtype=np.complex128):
    """An uninitialised C-contiguous array for results the device writes: pinned host memory when large."""
    global _pinned_cached_bytes, _pinned_live_bytes
    dtype = np.dtype(dtype)
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if (nbytes < _PINNED_MIN_BYTES or nbytes > _PINNED_MAX_BYTES or os.environ.get("MIDYN_PINNED_RESULTS", "1") == "0"
            or HIP_RUNTIME is None):
        return np.empty(shape, dtype=dtype)
    ptr = None
    with _pinned_lock:
        # a scan that KEEPS its results must not page-lock the host: past the budget, results are ordinary arrays
        if _pinned_live_bytes + nbytes > _PINNED_LIVE_MAX:
            return np.empty(shape, dtype=dtype)
        _pinned_live_bytes += nbytes
        blocks = _pinned_cache.setdefault(nbytes, [])
        if blocks:
            ptr = blocks.pop()
            _pinned_cached_bytes -= nbytes
    if ptr is None:
        p = _vp()
        try:
            ok = _hip_runtime().hipHostMalloc(ctypes.byref(p), nbytes, 0) == 0 and bool(p.value)
        except (OSError, AttributeError, HipLibraryError):
            ok = False
        if not ok:
            with _pinned_lock:
                _pinned_live_bytes -= nbytes
            return np.empty(shape, dtype=dtype)
        ptr = p.value
    buf = (ctypes.c_char * nbytes).from_address(ptr)
    buf._midyn_block = _PinnedBlock(ptr, nbytes)      # lives exactly as long as the buffer every view of the array refers to
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def is_pinned(a) -> bool:
    """Does this array live in one of result_array's pinned blocks (device-writable host memory)?"""
    for _ in range(8):
        if a is None:
            return False
        if getattr(a, "_midyn_block", None) is not None:
            return True
        a = getattr(a, "base", None) if isinstance(a, np.ndarray) else getattr(a, "obj", None) if isinstance(a, memoryview) else None
    return False


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, SignalTable):
        return _vp(a.dev_ptr)
    return a.ctypes.data_as(_vp)


def c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class Context:
    """One HIP stream on one device (``midyn_ctx``)."""

    def __init__(self, device: int = 0):
        self.lib = load()
        h = _vp()
        st = self.lib.midyn_ctx_create(int(device), ctypes.byref(h))
        if st != 0:
            raise HipLibraryError(self.lib.midyn_last_error(None).decode())
        self.handle = h
        self.device = int(device)

    def check(self, status):
        if status != 0:
            raise DynamicsError(self.lib.midyn_last_error(self.handle).decode())

    def set_option(self, name: str, value: int):
        self.check(self.lib.midyn_ctx_set_option(self.handle, name.encode(), int(value)))

    def get_option(self, name: str) -> int:
        out = _cll(0)
        self.check(self.lib.midyn_ctx_get_option(self.handle, name.encode(), ctypes.byref(out)))
        return int(out.value)

    def options(self, **values):
        """Context manager: the given options for the duration of a `with` block, their PREVIOUS values afterwards."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            before = {name: self.get_option(name) for name in values}
            try:
                for name, val in values.items():
                    self.set_option(name, val)
                yield self
            finally:
                for name, val in before.items():
                    self.set_option(name, val)

        return scope()

    def synchronize(self):
        self.check(self.lib.midyn_ctx_synchronize(self.handle))

    def counters(self, name: str):
        out = (ctypes.c_double * 2)()
        self.check(self.lib.midyn_get_counters(self.handle, name.encode(), out))
        return {"launches": out[0], "ms": out[1]}

    def executed_flops(self, name: str) -> float:
        """Real flops the dense MFMA contraction launches of counter class `name` executed while `profile` was on."""
        out = (ctypes.c_double * 2)()
        self.check(self.lib.midyn_get_counters(self.handle, ("flops:" + name).encode(), out))
        return float(out[0])

    def microbench(self, name: str, full: bool = False):
        out = (ctypes.c_double * 4)()
        self.check(self.lib.midyn_microbench(self.handle, name.encode(), out))
        return [float(x) for x in out] if full else float(out[0])

    def reset_counters(self):
        self.check(self.lib.midyn_reset_counters(self.handle))

    def timer_start(self):
        """Record the start event of the HIP-event stopwatch on this context's stream."""
        self.check(self.lib.midyn_ctx_timer(self.handle, 0, None))

    def timer_stop(self) -> float:
        """Record the stop event, wait for it; elapsed milliseconds on the stream since ``timer_start``."""
        ms = _cd()
        self.check(self.lib.midyn_ctx_timer(self.handle, 1, ctypes.byref(ms)))
        return float(ms.value)

    def zgemm(self, a, b):
        a, b = c128(a), c128(b)
        m, k = a.shape
        k2, n = b.shape
        if k != k2:
            raise DynamicsError("zgemm: inner dimensions differ")
        c = np.empty((m, n), dtype=np.complex128)
        self.check(self.lib.midyn_zgemm(self.handle, m, n, k, _ptr(a), _ptr(b), _ptr(c)))
        return c

    def expm(self, a, return_info=False):
        a = c128(a)
        single = a.ndim == 2
        if single:
            a = a[None]
        batch, n, n2 = a.shape
        if n != n2:
            raise DynamicsError("expm: matrices must be square")
        out = np.empty_like(a)
        info = np.zeros((batch, 2), dtype=np.int64)
        self.check(self.lib.midyn_expm(self.handle, n, batch, _ptr(a), _ptr(out), _ptr(info)))
        out = out[0] if single else out
        return (out, info) if return_info else out

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle:
            self.lib.midyn_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


_default_ctx = {}


def default_context(device=None) -> Context:
    """Process-wide context for `device` (default: LOCAL_RANK, else 0)."""
    if device is None:
        device = int(os.environ.get("MIDYN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Stack:
    """Device-resident operator stack (``midyn_stack``).

    ops (k,n,n) | None, static (n,n) | None, frame_im (n,) | None -- all in the frame basis.
    """

    perm = None   # internal index permutation (set_permutation): internal position i holds API index perm[i]
    inv = None
    slot = None   # general embedding (set_embedding): API index a lives at internal position slot[a]
    n_api = None  # dimension seen by the callers of this wrapper (== n unless sectors are padded)

    def __init__(self, ctx: Context, ops, static, frame_im, dev_buffer_ptr=None, _adopt=None, _lindblad=None):
        self.ctx = ctx
        lib = ctx.lib
        h = _vp()
        if _lindblad is not None:
            # superoperators of the vectorised Lindblad model built on the device from the n x n operators
            h_d, h_ops, n_static, l_ops = _lindblad
            arrs = [None if x is None else c128(x) for x in (h_d, h_ops, n_static, l_ops)]
            n = next(x.shape[-1] for x in arrs if x is not None)
            counts = [0 if x is None else x.shape[0] for x in arrs[1:]]
            fr_a = None if frame_im is None else f64(frame_im)
            ctx.check(lib.midyn_stack_create_lindblad(ctx.handle, n, _ptr(arrs[0]), counts[0], _ptr(arrs[1]), counts[1],
                                                      _ptr(arrs[2]), counts[2], _ptr(arrs[3]), _ptr(fr_a),
                                                      ctypes.byref(h)))
        elif _adopt is not None:
            n, k, has_static, has_frame = _adopt
            ctx.check(lib.midyn_stack_adopt(ctx.handle, n, k, has_static, has_frame,
                                            _vp(dev_buffer_ptr), ctypes.byref(h)))
        else:
            ops_a = None if ops is None else c128(ops)
            st_a = None if static is None else c128(static)
            fr_a = None if frame_im is None else f64(frame_im)
            if ops_a is not None and ops_a.ndim != 3:
                raise DynamicsError("operators must be a (k,n,n) array")
            n = st_a.shape[-1] if st_a is not None else ops_a.shape[-1]
            k = 0 if ops_a is None else ops_a.shape[0]
            ctx.check(lib.midyn_stack_create(ctx.handle, n, k, _ptr(ops_a), _ptr(st_a), _ptr(fr_a),
                                             _vp(dev_buffer_ptr) if dev_buffer_ptr else None,
                                             ctypes.byref(h)))
        self.handle = h
        self._read_info()

    def set_permutation(self, perm):
        """The operators of this stack are stored in a permuted basis (internal position i = API index ``perm[i]``;
        models group the frame-basis vectors by symmetry sector).  States and generators that cross this wrapper are
        given / returned in API order: rows are permuted on the way in and back on the way out."""
        if perm is None:
            self.perm = self.inv = self.slot = None
            self.n_api = self.n
            return
        perm = np.asarray(perm, dtype=np.int64)
        if perm.shape != (self.n,) or not np.array_equal(np.sort(perm), np.arange(self.n)):
            raise DynamicsError("not a permutation of the stack's indices")
        self.set_embedding(np.argsort(perm))

    def set_embedding(self, slot):
        """General form of ``set_permutation``: API index a lives at internal position ``slot[a]`` of a stack that may
        be LARGER than the API dimension (symmetry sectors padded to block boundaries; the extra rows and columns of
        every operator are zero, so the padding rows of a state stay zero).  ``len(slot)`` is the API dimension."""
        slot = np.asarray(slot, dtype=np.int64)
        if slot.ndim != 1 or slot.size > self.n or np.unique(slot).size != slot.size or slot.min() < 0 \
                or slot.max() >= self.n:
            raise DynamicsError("not an embedding of the API indices into the stack's indices")
        self.slot = slot
        self.n_api = int(slot.size)
        if self.n_api == self.n:            # a pure permutation
            self.inv = slot                 # API index -> internal position
            self.perm = np.argsort(slot)    # internal position -> API index
        else:
            self.perm = self.inv = None

    def _rows_in(self, y, axis):
        if self.slot is None:
            return y
        if self.perm is not None:
            return np.ascontiguousarray(np.take(y, self.perm, axis=axis))
        axis = axis % y.ndim
        out = np.zeros(y.shape[:axis] + (self.n,) + y.shape[axis + 1:], dtype=y.dtype)
        idx = [slice(None)] * y.ndim
        idx[axis] = self.slot
        out[tuple(idx)] = y
        return out

    def _rows_out(self, y, axis):
        if self.slot is None:
            return y
        return np.ascontiguousarray(np.take(y, self.slot, axis=axis))

    def _read_info(self):
        ctx, lib, h = self.ctx, self.ctx.lib, self.handle
        info = (ctypes.c_longlong * 8)()
        ctx.check(lib.midyn_stack_info(h, info))
        (self.n, self.n_pad, self.k, self.has_static, self.has_frame, self.n_segments,
         self.n_active_segments, self.packed_mib) = [int(x) for x in info]
        if self.slot is None:
            self.n_api = self.n
        modes = (ctypes.c_int * max(self.n_segments, 1))()
        ctx.check(lib.midyn_stack_segment_modes(h, modes))
        self.segment_modes = [int(modes[i]) for i in range(self.n_segments)]

    @classmethod
    def empty(cls, ctx, n, k, has_static, has_frame):
        """Receiving side of ``broadcast``: a stack of the given shape with an allocated, unfilled buffer."""
        self = cls.__new__(cls)
        self.ctx = ctx
        h = _vp()
        ctx.check(ctx.lib.midyn_stack_create_empty(ctx.handle, int(n), int(k), int(bool(has_static)),
                                                   int(bool(has_frame)), ctypes.byref(h)))
        self.handle = h
        self._read_info()
        return self

    def broadcast(self, comm: "Comm", root: int = 0):
        """ONE RCCL broadcast of the packed stack from rank ``root`` (C-ABI ``midyn_stack_broadcast``)."""
        self.ctx.check(self.ctx.lib.midyn_stack_broadcast(self.handle, comm.handle, int(root)))
        self._read_info()

    def broadcast_from(self, src: "Stack | None", comm: "Comm", root: int = 0):
        """The same broadcast out of place (C-ABI ``midyn_stack_broadcast_from``): ``self`` (normally ``Stack.empty``)
        receives the packed buffer that rank ``root`` sends from ``src`` (``None`` off the root) and runs the receiving
        side.  The host-side index maps of ``src`` (``set_permutation`` / ``set_embedding``) are not part of the packed
        buffer: they travel separately (``distributed.broadcast_stack`` ships them with ``broadcast_object_list``)."""
        self.ctx.check(self.ctx.lib.midyn_stack_broadcast_from(self.handle, None if src is None else src.handle,
                                                               comm.handle, int(root)))
        self._read_info()

    def block_info(self) -> dict:
        """Block occupancy (work-list routes): fraction / number of non-zero 16x16 blocks and, per MFMA row-panel
        height, the listed fraction and number of (panel, K tile, operator) tiles."""
        out = (ctypes.c_double * 13)()
        self.ctx.check(self.ctx.lib.midyn_stack_block_info(self.handle, out))
        info = {"state": int(out[0]), "block_density": float(out[1]), "nonzero_blocks": int(out[2]),
                "blocks_per_side": int(out[3]), "streamed_fraction": float(out[12]), "tile_lists": {}}
        for t, bm in enumerate((64, 128, 32, 16)):
            info["tile_lists"][bm] = {"listed_fraction": float(out[4 + 2 * t]), "listed_tiles": int(round(out[5 + 2 * t]))}
        return info

    @classmethod
    def from_lindblad(cls, ctx, h_d, h_ops, n_static, l_ops, frame_im):
        """Vectorised Lindblad stack (dimension n^2) from the n x n operators in the frame basis: h_d (n,n) | None,
        h_ops (k_h,n,n) | None, n_static (n_s,n,n) | None, l_ops (k_l,n,n) | None, frame_im (n^2,) | None."""
        return cls(ctx, None, None, frame_im, _lindblad=(h_d, h_ops, n_static, l_ops))

    def antiherm_defect(self) -> np.ndarray:
        """|| A_seg + A_seg^dagger ||_F per segment (static operator first when present), computed on the device."""
        out = (ctypes.c_double * max(self.n_segments, 1))()
        self.ctx.check(self.ctx.lib.midyn_stack_antiherm_defect(self.handle, out))
        return np.array([out[i] for i in range(self.n_segments)])

    @staticmethod
    def packed_bytes(n, k, has_static):
        lib = load()
        b = ctypes.c_size_t()
        if lib.midyn_stack_packed_bytes(int(n), int(k), int(bool(has_static)), ctypes.byref(b)):
            raise DynamicsError(lib.midyn_last_error(None).decode())
        return int(b.value)

    # -- single evaluations -----------------------------------------------------------------
    def eval_generator(self, coeffs, t):
        out = np.empty((self.n, self.n), dtype=np.complex128)
        c = None if self.k == 0 else f64(coeffs)
        if c is not None and c.shape != (self.k,):
            raise DynamicsError("coefficient vector has the wrong length")
        self.ctx.check(self.ctx.lib.midyn_eval_generator(self.handle, _ptr(c), float(t), _ptr(out)))
        return self._rows_out(self._rows_out(out, 0), 1)

    def eval_rhs(self, coeffs, t, y):
        y = c128(y)
        if y.shape[0] != self.n_api or y.ndim > 2:
            raise DynamicsError("state has the wrong shape")
        y = self._rows_in(y, 0)
        m = 1 if y.ndim == 1 else y.shape[1]
        out = np.empty_like(y)
        c = None if self.k == 0 else f64(coeffs)
        if c is not None and c.shape != (self.k,):
            raise DynamicsError("coefficient vector has the wrong length")
        self.ctx.check(self.ctx.lib.midyn_eval_rhs(self.handle, _ptr(c), float(t), _ptr(y), m, _ptr(out)))
        return self._rows_out(out, 0)

    # -- solves -----------------------------------------------------------------------------
    def _solve_args(self, times, table, step_rows, step_h, step_save, y0, batch):
        times = f64(times)
        r = times.shape[0]
        if self.k > 0:
            if not isinstance(table, SignalTable):
                table = f64(table)
            if table.shape != (batch, r, self.k):
                raise DynamicsError(f"coefficient table must be (B,R,k)={(batch, r, self.k)}, got {table.shape}")
        else:
            table = None
        step_rows = i32(step_rows).reshape(-1, 3)
        step_h = f64(step_h)
        step_save = i32(step_save)
        nsteps = step_rows.shape[0]
        y0 = self._rows_in(c128(y0), -2)      # (n, m) or (B, n, m): rows into the internal order
        return times, r, table, step_rows, step_h, step_save, nsteps, y0

    def rk4_solve(self, times, table, step_rows, step_h, step_save, n_save, y0, batch, y0_shared):
        """y0: (n,m) if y0_shared else (B,n,m).  Returns (B, P, n, m)."""
        times, r, table, step_rows, step_h, step_save, nsteps, y0 = self._solve_args(
            times, table, step_rows, step_h, step_save, y0, batch)
        m = y0.shape[-1]
        out = result_array((batch, n_save, self.n, m))
        self.ctx.check(self.ctx.lib.midyn_rk4_solve(
            self.handle, batch, m, r, _ptr(times), _ptr(table), nsteps, _ptr(step_rows), _ptr(step_h),
            _ptr(step_save), n_save, _ptr(y0), int(bool(y0_shared)), _ptr(out)))
        return self._rows_out(out, 2)

    def expm_solve(self, times, table, step_rows, step_h, step_save, n_save, magnus_order, y0, batch,
                   y0_shared):
        times, r, table, step_rows, step_h, step_save, nsteps, y0 = self._solve_args(
            times, table, step_rows, step_h, step_save, y0, batch)
        m = y0.shape[-1]
        out = result_array((batch, n_save, self.n, m))
        self.ctx.check(self.ctx.lib.midyn_expm_solve(
            self.handle, batch, m, r, _ptr(times), _ptr(table), nsteps, _ptr(step_rows), _ptr(step_h),
            _ptr(step_save), n_save, int(magnus_order), _ptr(y0), int(bool(y0_shared)), _ptr(out)))
        return self._rows_out(out, 2)

    def parallel_solve(self, times, table, step_rows, step_h, step_save, n_save, method, y0, batch,
                       y0_shared):
        """Parallel-in-time propagation (row f3): ``method`` 0 = RK4 step propagators, 1..3 = expm of
        the Magnus expansion of that order."""
        times, r, table, step_rows, step_h, step_save, nsteps, y0 = self._solve_args(
            times, table, step_rows, step_h, step_save, y0, batch)
        m = y0.shape[-1]
        out = np.empty((batch, n_save, self.n, m), dtype=np.complex128)
        self.ctx.check(self.ctx.lib.midyn_parallel_solve(
            self.handle, batch, m, r, _ptr(times), _ptr(table), nsteps, _ptr(step_rows), _ptr(step_h),
            _ptr(step_save), n_save, int(method), _ptr(y0), int(bool(y0_shared)), _ptr(out)))
        return self._rows_out(out, 2)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle and self.ctx.handle:
            self.ctx.lib.midyn_stack_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class Comm:
    """RCCL communicator of one rank (one process per GPU), bound to the context's device; only used for the
    single broadcast of the operator stack.  ``unique_id``: the 128 bytes of ``Comm.unique_id()`` called on ONE
    rank and shipped to the others by the caller (any channel)."""

    def __init__(self, ctx: "Context", world: int, rank: int, unique_id: bytes):
        preload_rccl()
        if len(unique_id) != 128:
            raise DynamicsError("an RCCL unique id is 128 bytes")
        self.ctx = ctx
        h = _vp()
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        ctx.check(ctx.lib.midyn_comm_init_rank(ctx.handle, int(world), int(rank), buf, ctypes.byref(h)))
        self.handle = h
        self.world, self.rank = int(world), int(rank)

    @staticmethod
    def unique_id() -> bytes:
        preload_rccl()
        lib = load()
        buf = ctypes.create_string_buffer(128)
        if lib.midyn_comm_get_unique_id(buf):
            raise DynamicsError(lib.midyn_last_error(None).decode())
        return buf.raw

    def count(self) -> int:
        """Ranks of the communicator as RCCL reports them (ncclCommCount)."""
        n = ctypes.c_int(0)
        self.ctx.check(self.ctx.lib.midyn_comm_count(self.ctx.handle, self.handle, ctypes.byref(n)))
        return int(n.value)

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle and self.ctx.handle:
            self.ctx.lib.midyn_comm_destroy(self.ctx.handle, self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class SignalTable:
    """Coefficient table S[B][R][k] evaluated ON THE DEVICE from piecewise-constant samples and
    carriers (``midyn_sigtable_create``).  Accepted wherever the solve
    methods take a ``table``; ``fetch()`` copies it to the host."""

    def __init__(self, ctx: "Context", batch, k, times, term_ptr, term_params, sample_ptr, samples):
        self.ctx = ctx
        times = f64(times)
        term_ptr = np.ascontiguousarray(term_ptr, dtype=np.int64)
        term_params = f64(term_params).reshape(-1, 4)
        sample_ptr = np.ascontiguousarray(sample_ptr, dtype=np.int64).reshape(-1, 2)
        samples = c128(samples)
        if term_ptr.shape != (batch * k + 1,):
            raise DynamicsError("term_ptr must have B*k+1 entries")
        n_terms = int(term_ptr[-1])
        if term_params.shape[0] != n_terms or sample_ptr.shape[0] != n_terms:
            raise DynamicsError("term_params / sample_ptr must have one row per term")
        if n_terms and int((sample_ptr[:, 0] + sample_ptr[:, 1]).max()) > samples.shape[0]:
            raise DynamicsError("sample_ptr points past the end of samples")
        self.shape = (int(batch), int(times.shape[0]), int(k))
        h = _vp()
        ctx.check(ctx.lib.midyn_sigtable_create(
            ctx.handle, int(batch), int(k), self.shape[1], _ptr(times), _ptr(term_ptr), _ptr(term_params),
            _ptr(sample_ptr), _ptr(samples), ctypes.byref(h)))
        self.handle = h
        dev = ctypes.c_void_p()
        ctx.check(ctx.lib.midyn_sigtable_data(self.handle, ctypes.byref(dev), None))
        self.dev_ptr = dev.value

    def fetch(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float64)
        self.ctx.check(self.ctx.lib.midyn_sigtable_fetch(self.handle, _ptr(out)))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle and self.ctx.handle:
            self.ctx.lib.midyn_sigtable_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class Expansion:
    """Device-resident terms of a Dyson / Magnus expansion (row f4, ``midyn_expansion_*``)."""

    def __init__(self, ctx: "Context", terms, constant_term=None, post=None, use_expm=False):
        self.ctx = ctx
        terms = c128(terms)
        if terms.ndim != 3 or terms.shape[1] != terms.shape[2]:
            raise DynamicsError("expansion terms must be (M, n, n)")
        self.n_terms, self.n = int(terms.shape[0]), int(terms.shape[1])
        const = None if constant_term is None else c128(constant_term)
        post = None if post is None else c128(post)
        for arr in (const, post):
            if arr is not None and arr.shape != (self.n, self.n):
                raise DynamicsError("constant_term / post must be (n, n)")
        h = _vp()
        ctx.check(ctx.lib.midyn_expansion_create(ctx.handle, self.n, self.n_terms, _ptr(terms), _ptr(const),
                                                 _ptr(post), int(bool(use_expm)), ctypes.byref(h)))
        self.handle = h

    def solve(self, mono, y0, batch, y0_shared):
        """mono (B, nsteps, M) real; y0 (n, m) if shared else (B, n, m) -> final states (B, n, m)."""
        mono = f64(mono)
        if mono.ndim != 3 or mono.shape[0] != batch or mono.shape[2] != self.n_terms:
            raise DynamicsError(f"monomial table must be (B, nsteps, M) = ({batch}, *, {self.n_terms})")
        y0 = c128(y0)
        m = y0.shape[-1]
        out = np.empty((batch, self.n, m), dtype=np.complex128)
        self.ctx.check(self.ctx.lib.midyn_expansion_solve(self.handle, int(batch), int(mono.shape[1]), _ptr(mono),
                                                          int(m), _ptr(y0), int(bool(y0_shared)), _ptr(out)))
        return out

    def set_monomials(self, n_vars, labels):
        """labels: the M index multisets of the terms (tuples of coefficient indices < n_vars); afterwards ``solve_coeffs`` takes the
        Chebyshev coefficients themselves and the monomial table is formed on the device."""
        if len(labels) != self.n_terms or any(len(lab) == 0 for lab in labels):
            raise DynamicsError(f"one non-empty label per expansion term ({self.n_terms}) is required")
        order = max(len(lab) for lab in labels)
        tab = np.full((self.n_terms, order), -1, dtype=np.int32)
        for i, lab in enumerate(labels):
            tab[i, : len(lab)] = lab
        self.ctx.check(self.ctx.lib.midyn_expansion_set_monomials(self.handle, int(n_vars), int(order), _ptr(tab)))
        self.n_vars = int(n_vars)

    def solve_coeffs(self, coeffs, y0, batch, y0_shared):
        """coeffs (B, n_vars, nsteps) real; y0 (n, m) if shared else (B, n, m) -> final states (B, n, m)."""
        coeffs = f64(coeffs)
        if coeffs.ndim != 3 or coeffs.shape[0] != batch or coeffs.shape[1] != getattr(self, "n_vars", -1):
            raise DynamicsError(f"coefficient table must be (B, n_vars, nsteps) = ({batch}, {getattr(self, 'n_vars', '?')}, *); "
                                "call set_monomials first")
        y0 = c128(y0)
        m = y0.shape[-1]
        out = np.empty((batch, self.n, m), dtype=np.complex128)
        self.ctx.check(self.ctx.lib.midyn_expansion_solve_coeffs(self.handle, int(batch), int(coeffs.shape[2]), _ptr(coeffs),
                                                                 int(m), _ptr(y0), int(bool(y0_shared)), _ptr(out)))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle and self.ctx.handle:
            self.ctx.lib.midyn_expansion_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class Rk4Plan:
    """Device-resident RK4 state for benchmarking (inputs stay in HBM between timed runs)."""

    def __init__(self, stack: Stack, times, table, step_rows, step_h, y0, batch, y0_shared):
        self.stack = stack
        ctx = stack.ctx
        times, r, table, step_rows, step_h, _, nsteps, y0 = stack._solve_args(
            times, table, step_rows, step_h, np.zeros(0, dtype=np.int32), y0, batch)
        self.m = y0.shape[-1]
        self.batch = batch
        self.nsteps = nsteps
        h = _vp()
        ctx.check(ctx.lib.midyn_rk4_plan_create(
            stack.handle, batch, self.m, r, _ptr(times), _ptr(table), nsteps, _ptr(step_rows),
            _ptr(step_h), _ptr(y0), int(bool(y0_shared)), ctypes.byref(h)))
        self.handle = h

    def run(self, step_begin, step_end):
        self.stack.ctx.check(self.stack.ctx.lib.midyn_rk4_plan_run(self.handle, int(step_begin), int(step_end)))

    def fetch(self):
        out = np.empty((self.batch, self.stack.n, self.m), dtype=np.complex128)
        self.stack.ctx.check(self.stack.ctx.lib.midyn_rk4_plan_fetch(self.handle, _ptr(out)))
        return self.stack._rows_out(out, 1)

    def close(self):
        # the plan points into its stack and context: only destroy it while both are alive
        if (getattr(self, "handle", None) is not None and self.handle and self.stack.handle
                and self.stack.ctx.handle):
            self.stack.ctx.lib.midyn_rk4_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class ExpmPlan:
    """Device-resident state of a fixed-step Magnus / scipy_expm solve that is repeated with new signal parameters
    (midyn_expm_plan_*): the frame-phase table, step tables, y0, result block and exchange slots are made once;
    ``solve(table)`` uploads a coefficient table and runs; ``run(table)`` / ``fetch()`` split it (run returns when the
    launch is queued)."""

    def __init__(self, stack: Stack, times, step_rows, step_h, step_save, n_save, magnus_order, y0, batch, y0_shared):
        self.stack = stack
        ctx = stack.ctx
        times = f64(times)
        r = times.shape[0]
        step_rows = i32(step_rows).reshape(-1, 3)
        step_h = f64(step_h)
        step_save = i32(step_save)
        nsteps = step_rows.shape[0]
        y0 = stack._rows_in(c128(y0), -2)      # (n, m) or (B, n, m): rows into the internal order
        self.m = y0.shape[-1]
        self.batch, self.r, self.n_save = batch, r, int(n_save)
        self._out = None
        h = _vp()
        ctx.check(ctx.lib.midyn_expm_plan_create(
            stack.handle, batch, self.m, r, _ptr(times), nsteps, _ptr(step_rows), _ptr(step_h), _ptr(step_save),
            self.n_save, int(magnus_order), _ptr(y0), int(bool(y0_shared)), ctypes.byref(h)))
        self.handle = h

    def run(self, table):
        if self.stack.k > 0:
            if not isinstance(table, SignalTable):
                table = f64(table)
            if table.shape != (self.batch, self.r, self.stack.k):
                raise DynamicsError(f"coefficient table must be (B,R,k)={(self.batch, self.r, self.stack.k)}, got {table.shape}")
        else:
            table = None
        self._out = result_array((self.batch, self.n_save, self.stack.n, self.m))
        # (only a block of OUR pinned cache is offered for direct writes: the plan remembers accepted blocks by address)
        self.stack.ctx.check(self.stack.ctx.lib.midyn_expm_plan_run(self.handle, _ptr(table), _ptr(self._out) if is_pinned(self._out) else None))

    def fetch(self):
        if self._out is None:
            raise DynamicsError("ExpmPlan.fetch before run")
        out, self._out = self._out, None
        self.stack.ctx.check(self.stack.ctx.lib.midyn_expm_plan_fetch(self.handle, _ptr(out)))
        return self.stack._rows_out(out, 2)

    def solve(self, table):
        self.run(table)
        return self.fetch()

    def close(self):
        # the plan points into its stack and context: only destroy it while both are alive
        if (getattr(self, "handle", None) is not None and self.handle and self.stack.handle
                and self.stack.ctx.handle):
            self.stack.ctx.lib.midyn_expm_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass


class LindbladDevice:
    """Non-vectorised Lindblad RHS on the device (``midyn_lindblad``): two operator stacks (A+B, A-B)
    sharing one coefficient vector, plus the dissipators [static..., dynamic...], all in the frame basis."""

    def __init__(self, left: Stack, right: Stack, k_h: int, n_static: int, n_dyn: int, dissipators):
        self.left, self.right, self.ctx = left, right, left.ctx
        self.n = left.n
        self.k = left.k
        d = None if dissipators is None else c128(dissipators)
        h = _vp()
        self.ctx.check(self.ctx.lib.midyn_lindblad_create(left.handle, right.handle, int(k_h), int(n_static),
                                                         int(n_dyn), _ptr(d), ctypes.byref(h)))
        self.handle = h

    def rhs(self, coeffs, t, rho):
        rho = c128(rho)
        single = rho.ndim == 2
        r = rho[None] if single else rho
        if r.shape[1:] != (self.n, self.n):
            raise DynamicsError("Shape mismatch for initial state y0 and LindbladModel.")
        c = None if self.k == 0 else f64(coeffs)
        out = np.empty_like(r)
        self.ctx.check(self.ctx.lib.midyn_lindblad_rhs(self.handle, _ptr(c), float(t), _ptr(r), r.shape[0], _ptr(out)))
        return out[0] if single else out

    def rk4_solve(self, times, table, step_rows, step_h, step_save, n_save, rho0, batch, shared):
        times = f64(times)
        r = times.shape[0]
        if self.k == 0:
            table = None
        elif not isinstance(table, SignalTable):
            table = f64(table)
        if table is not None and table.shape != (batch, r, self.k):
            raise DynamicsError(f"coefficient table must be (B,R,k)={(batch, r, self.k)}, got {table.shape}")
        step_rows = i32(step_rows).reshape(-1, 3)
        step_h, step_save = f64(step_h), i32(step_save)
        rho0 = c128(rho0)
        out = np.empty((batch, n_save, self.n, self.n), dtype=np.complex128)
        self.ctx.check(self.ctx.lib.midyn_lindblad_rk4_solve(
            self.handle, batch, r, _ptr(times), _ptr(table), step_rows.shape[0], _ptr(step_rows), _ptr(step_h),
            _ptr(step_save), n_save, _ptr(rho0), int(bool(shared)), _ptr(out)))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle and self.ctx.handle:
            self.ctx.lib.midyn_lindblad_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # pylint: disable=broad-except
            pass
