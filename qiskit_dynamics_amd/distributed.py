"""Multi-GPU sweep sharding: one process per GPU, trajectories are independent.

The reference loops sweep instances sequentially in one process (solvers/solver_classes.py:568-586);
there is no communication in its path.  Here the B instances are split into contiguous shards, one
per rank; the ONLY collective is a single RCCL broadcast of the packed operator stack from rank 0
over xGMI at setup (`broadcast_stack`), after which every rank integrates its shard with zero
per-step traffic.  torch.distributed is used purely as the RCCL / rendezvous plumbing
(backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of `n_items` owned by `rank` (sizes differ by at most one)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def init_process_group_from_env(backend: str = "nccl"):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* (as torchrun sets them)."""
    import torch.distributed as dist

    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if backend == "nccl":
        # one process per GPU: every RCCL collective of this process must run on ITS device, not on
        # cuda:0 (two ranks on one device = RCCL "duplicate GPU" error or a hang)
        import torch

        dev = local_device()
        torch.cuda.set_device(dev)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        return rank, world
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def local_device() -> int:
    """Device ordinal of this rank: LOCAL_RANK (torchrun / bench.py's own spawner), modulo the visible devices."""
    import torch

    n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    local = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    return local % n_dev if n_dev else 0


def broadcast_stack(ctx, ops, static, frame_im, n, k, src: int = 0, perm=None, slot=None):
    """Build the packed device stack on `src`, RCCL-broadcast it, adopt it on the other ranks.

    `ops`/`static`/`frame_im` are only read on rank `src` (may be None elsewhere); n, k and the
    presence flags must be known on every rank.  `perm` (rank `src`): the internal index permutation of a
    stack whose arrays are grouped by symmetry sector (`Stack.set_permutation`), or `slot`, the general embedding of
    a stack whose sectors are also padded to block boundaries (`Stack.set_embedding`); it travels with the stack.
    Returns (Stack, torch tensor that owns the memory).
    """
    import torch
    import torch.distributed as dist

    from . import _lib

    rank = dist.get_rank()
    meta = torch.zeros(2, dtype=torch.int64)
    if rank == src:
        meta[0] = 1 if static is not None else 0
        meta[1] = 1 if frame_im is not None else 0
    dev = torch.device("cuda", ctx.device)
    meta = meta.to(dev)
    dist.broadcast(meta, src=src)
    has_static, has_frame = int(meta[0].item()), int(meta[1].item())
    nbytes = _lib.Stack.packed_bytes(n, k, has_static)
    buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    stack = None
    if rank == src:
        stack = _lib.Stack(ctx, ops, static, frame_im, dev_buffer_ptr=buf.data_ptr())
        ctx.synchronize()
    torch.cuda.synchronize(dev)
    dist.broadcast(buf, src=src)
    torch.cuda.synchronize(dev)
    if rank != src:
        stack = _lib.Stack(ctx, None, None, None, dev_buffer_ptr=buf.data_ptr(),
                           _adopt=(n, k, has_static, has_frame))
    box = [None if perm is None else np.asarray(perm).tolist(), None if slot is None else np.asarray(slot).tolist()]
    dist.broadcast_object_list(box, src=src)
    if box[0] is not None:
        stack.set_permutation(np.asarray(box[0], dtype=np.int64))
    elif box[1] is not None:
        stack.set_embedding(np.asarray(box[1], dtype=np.int64))
    return stack, buf


def gather_sweep_results(local: np.ndarray, n_total: int, device=None):
    """All-gather per-rank result blocks (shape (b_loc, ...)) into the full (n_total, ...) array on
    every rank.  Not on the timed path; used by tests and for returning sweep results.  `device` is the
    ordinal of the GPU this rank computes on (the solver context's); RCCL needs the staging tensors there."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    backend = dist.get_backend()
    if backend == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    else:
        dev = torch.device("cpu")
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    max_b = max(hi - lo for lo, hi in sizes)
    pad = np.zeros((max_b,) + local.shape[1:], dtype=np.complex128)
    pad[: local.shape[0]] = local
    t = torch.view_as_real(torch.from_numpy(pad)).contiguous().to(dev)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    full = np.concatenate([
        torch.view_as_complex(o.cpu().contiguous()).numpy()[: hi - lo] for o, (lo, hi) in zip(outs, sizes)
    ])
    return full


def gather_sweep_to_root(local: np.ndarray, n_total: int, root: int = 0, device=None):
    """Per-rank result blocks (b_loc, ...) concatenated on rank `root` only (None elsewhere): every rank ships its block
    once -- SURVEY 8(e) asks for per-rank buffers concatenated, not for a copy of everything everywhere."""
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    else:
        dev = torch.device("cpu")
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    max_b = max(hi - lo for lo, hi in sizes)
    pad = np.zeros((max_b,) + local.shape[1:], dtype=np.complex128)
    pad[: local.shape[0]] = local
    t = torch.view_as_real(torch.from_numpy(pad)).contiguous().to(dev)
    outs = [torch.empty_like(t) for _ in range(world)] if rank == root else None
    dist.gather(t, outs, dst=root)
    if rank != root:
        return None
    return np.concatenate([torch.view_as_complex(o.cpu().contiguous()).numpy()[: hi - lo] for o, (lo, hi) in zip(outs, sizes)])


def solve_sweep(solver, t_span, y0, signals, gather="all", root=0, **kwargs):
    """``solver.solve`` of a list-mode sweep, sharded over the ranks of the initialised process group
    (one process per GPU): rank r solves the contiguous shard ``shard_bounds(B, r, world)`` of the
    ``B = len(signals)`` instances with ONE batched device solve.  ``y0`` is one state shared by all
    instances or a list of B states; ``t_span`` is shared.  No per-step communication.

    ``gather``: what happens to the results (the reference's loop returns one list, solver_classes.py:568-586) --
      ``"all"``   every rank returns the full list of ``OdeResult``s in sweep order (an all-gather of every shard:
                  world x the result volume on the wire -- convenient, meant for small results and tests);
      ``"root"``  rank ``root`` returns the full list, the other ranks ``None`` (each shard travels once);
      ``"none"``  no communication at all: every rank returns ``(lo, results of its own shard)``.
    """
    import torch.distributed as dist
    from scipy.integrate._ivp.ivp import OdeResult

    if gather not in ("all", "root", "none"):
        raise ValueError('gather must be "all", "root" or "none"')
    if not isinstance(signals, list) or not signals or not isinstance(signals[0], (list, tuple)):
        raise ValueError("solve_sweep needs a list of per-instance signal lists")
    n_total = len(signals)
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_bounds(n_total, rank, world)
    y0_is_list = isinstance(y0, list)
    if y0_is_list and len(y0) != n_total:
        raise ValueError("y0 list and signals list must have the same length")
    local_t, local_y, res = None, None, []
    if hi > lo:
        res = solver.solve(t_span=t_span, y0=y0[lo:hi] if y0_is_list else y0, signals=signals[lo:hi], **kwargs)
        res = res if isinstance(res, list) else [res]
        local_t = np.asarray(res[0].t, dtype=float)
        local_y = np.stack([np.asarray(r.y, dtype=np.complex128) for r in res])
    if gather == "none":
        return lo, res
    # ranks with an empty shard (B < world) learn the result shape from the others
    ctx = getattr(getattr(solver, "model", None), "_ctx", None)
    device = getattr(ctx, "device", None)
    if dist.get_backend() == "nccl" and device is not None:
        import torch

        torch.cuda.set_device(int(device))  # all_gather_object stages on the current device
    shapes = [None] * world
    dist.all_gather_object(shapes, None if local_y is None else (local_y.shape[1:], local_t.tolist()))
    known = next(s_ for s_ in shapes if s_ is not None)
    if local_y is None:
        local_y = np.zeros((0,) + tuple(known[0]), dtype=np.complex128)
    if gather == "root":
        full = gather_sweep_to_root(local_y, n_total, root=root, device=device)
        if full is None:
            return None
    else:
        full = gather_sweep_results(local_y, n_total, device=device)
    t_out = np.asarray(known[1], dtype=float)
    return [OdeResult(t=t_out, y=full[b]) for b in range(n_total)]


# ---------------------------------------------------------------------------------------------------------------
# C-ABI broadcast self-test in a CHILD process.  A collective that goes wrong on a new machine usually does not
# fail, it hangs -- and a hung RCCL kernel poisons the process that launched it.  Before a rank puts its own
# context into `midyn_comm_init_rank` / `midyn_stack_broadcast` it can run the same calls on a small stack in a
# child (one per rank, same devices, its own ncclUniqueId): the child either exits 0 within the time limit or is
# killed by its PID, and the caller falls back to the torch.distributed broadcast without ever having touched
# the C-ABI communicator.
# ---------------------------------------------------------------------------------------------------------------
PROBE_N, PROBE_K = 96, 2


def _probe_arrays():
    rng = np.random.default_rng(20240)
    ops = rng.normal(size=(PROBE_K, PROBE_N, PROBE_N)) + 1j * rng.normal(size=(PROBE_K, PROBE_N, PROBE_N))
    static = rng.normal(size=(PROBE_N, PROBE_N)) + 1j * rng.normal(size=(PROBE_N, PROBE_N))
    return ops, static, rng.normal(size=PROBE_N)


def _probe_main(argv) -> int:
    """Child side: `python -m qiskit_dynamics_amd.distributed --probe RANK WORLD DEVICE UID_HEX`."""
    rank, world, device = int(argv[0]), int(argv[1]), int(argv[2])
    uid = bytes.fromhex(argv[3])
    if os.environ.get("MIDYN_PROBE_HANG"):          # test hook: behave like a hung collective
        import time

        time.sleep(3600)
    from . import _lib

    ctx = _lib.Context(device)
    comm = _lib.Comm(ctx, world, rank, uid)
    ops, static, frame_im = _probe_arrays()
    if rank == 0:
        stack = _lib.Stack(ctx, ops, static, frame_im)
    else:
        stack = _lib.Stack.empty(ctx, PROBE_N, PROBE_K, True, True)
    stack.broadcast(comm, 0)
    ctx.synchronize()
    c = np.array([0.3, -0.7])
    got = stack.eval_generator(c, 0.0)
    want = static + np.tensordot(c, ops, axes=1)
    err = float(np.max(np.abs(got - want)))
    comm.close()
    if not err < 1e-12:
        print(f"probe rank {rank}: broadcast stack differs from the source by {err:.3e}", flush=True)
        return 3
    return 0


def abi_broadcast_probe(rank: int, world: int, device: int, uid: bytes, timeout_s: float = 120.0):
    """Run the C-ABI communicator + stack broadcast self-test for this rank in a child process.  `uid`: an
    ncclUniqueId made for the probe on rank 0 (`_lib.Comm.unique_id()`) and shipped to every rank.  Returns
    (ok, message).  The child is killed by PID when it does not finish in `timeout_s`."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):   # the child is not a torch.distributed rank
        env.pop(var, None)
    cmd = [sys.executable, "-m", "qiskit_dynamics_amd.distributed", "--probe", str(rank), str(world), str(device),
           bytes(uid).hex()]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=root)
    try:
        out, _ = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return False, f"probe did not finish in {timeout_s:.0f} s (killed)"
    if proc.returncode != 0:
        tail = out.decode(errors="replace").strip().splitlines()[-3:]
        return False, f"probe exited with {proc.returncode}: " + " | ".join(tail)
    return True, "ok"


if __name__ == "__main__":
    import sys

    if len(sys.argv) >= 6 and sys.argv[1] == "--probe":
        sys.exit(_probe_main(sys.argv[2:]))
    sys.exit("usage: python -m qiskit_dynamics_amd.distributed --probe RANK WORLD DEVICE UID_HEX")
