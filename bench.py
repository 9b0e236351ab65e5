#!/usr/bin/env python
"""bench.py -- RHS evals/s of the 10-qubit (dim 1024) Schrodinger sweep on N MI355X (BASELINE.json metric).

Workload of `value` (BASELINE.json configs[1]/[2], SURVEY.md 8(d) cfg 3): chain Hamiltonian, n = 1024, k = 8 drive
operators + static operator, rotating frame = H_d (dense frame-basis operators), RK4 with max_dt = 0.005 on
t in [0, 5] (1000 steps, 4 RHS evaluations each), a sweep of 4096 signal instances.  A "step" is one RK4 step of
the whole sweep = 4 batched RHS evaluations of every instance.

    python bench.py --gpus N --steps K --warmup W

N = 1: all 4096 instances on one GPU.  N > 1 (default = BASELINE configs[2], "scaling": "strong"): the SAME
4096-instance sweep sharded 4096/N per GPU; `--weak` keeps 4096 instances per GPU instead.  `--gpus N` without a
launcher (no WORLD_SIZE in the environment) spawns the N ranks itself, one process per GPU; under torchrun /
torch.distributed.run it is one of the ranks.  It never prints an n_gpus = 1 line for --gpus 8: a world that does
not match --gpus, or fewer visible GPUs than ranks, is an error (non-zero exit).

The only collective of the path is ONE RCCL broadcast of the packed operator stack at setup (through the C-ABI,
`midyn_stack_broadcast`; `--torch-broadcast` uses torch.distributed's RCCL instead); its time is reported as
`broadcast_ms`, outside the timed region.  Timed region: inputs (stack, coefficient table, states) resident in
HBM; K steps enqueued on the library's HIP stream, bracketed by barrier + device synchronisation and by a HIP-event
pair on that stream; MAX over ranks.  Rank 0 prints ONE JSON line.

OUTPUT.  The LAST (and only) stdout line is a COMPACT object, below 6 KB (the driver keeps 8 018 characters of stdout):
the contract fields, `roofline`, `cpu_baseline`, `cfg3_three_numbers` and one {value, unit, frac, bound} object per other
configuration / SURVEY section 8 row (cfg2, cfg4, cfg5, sharded_cfg5, dense_expm, f2, f3, f4, projected_strong_scaling) --
tools/bench_legs/compact.py.  The FULL result (every object described below, notes, A/B legs, term decompositions) goes to
`bench_detail.json` beside this script (and to gpurun_out/ when that directory exists) and to stderr.

Objects of the full result besides the contract fields:
  roofline                    dominant kernel of `value` (fp64-MFMA batched RHS contraction): `achieved` = EXECUTED
                              MFMA flops / launch time, `frac` = achieved / 78.6 TFLOP/s (a hardware fraction, <= 1);
                              the SURVEY 8(d) "useful" figure is kept as `useful_tflops`
  dense_complex               the same sweep without exact-zero plane skipping (general complex operators)
  single_trajectory           cfg 2 = BASELINE configs[1], one trajectory: the register-resident RK4 kernel (whole step
                              range in one launch) with the per-stage route beside it
  roofline_single_trajectory  the per-stage HBM-bound streaming kernel of the same trajectory (resident_rk4=0):
                              `frac` = executed bytes / time / 8 TB/s
  cfg4, cfg5                  the vectorised-Lindblad / 12-qubit Magnus-2 configurations with their own rooflines
                              (executed work of the work lists AND the 8(d) dense-form price, labelled)
  sharded_cfg5                second sharded leg: the 1024-instance cfg-5 sweep, 1024/N per GPU
  diag_frame_rk4_sweep        the headline model set up in the diagonal frame diag(H_d) (sparse operators): the one-launch RK4 sweep
                              kernel without operator elements -- NOT `value`, which is BASELINE's full-frame configuration
  cpu_baseline                the NumPy oracle on the host cores (rank 0, N = 1), bounded sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tools.bench_legs.compact import emit  # noqa: E402  (the one stdout line: compact object; details -> bench_detail.json)
from tools.bench_legs.common import (ALL_CLASSES, CFG5_SWEEP, FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, LDS_PEAK_GBS, MAX_DT, N_DRIVES,  # noqa: E402,F401
                                     N_QUBITS, SWEEP, T_FINAL, build_diag_frame_stack, build_frame_basis_stack, build_model_stack,
                                     measured_traffic, profile_pass, sweep_table)
from tools.bench_legs.cfg4 import leg_cfg4, leg_cfg4_diag_frame  # noqa: E402
from tools.bench_legs.cfg5 import leg_cfg5  # noqa: E402
from tools.bench_legs.cpu import leg_cpu_baseline, leg_cpu_configs  # noqa: E402
from tools.bench_legs.rows import leg_dense_expm, leg_lindblad_rk4, leg_parallel_in_time, leg_perturbative  # noqa: E402
from tools.bench_legs.sweeps import leg_diag_frame_sweep, leg_small_sweeps  # noqa: E402



# -----------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` spawns its own ranks (one process per GPU)
# -----------------------------------------------------------------------------------------------------------------
def visible_gpus() -> int:
    """Number of HIP devices, probed in a child process (the launcher itself never touches the GPU)."""
    code = "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"
    try:
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        return int(out.stdout.strip().splitlines()[-1])
    except (subprocess.SubprocessError, ValueError, IndexError):
        return 0


def spawn_ranks(n_ranks: int, argv) -> int:
    """Start `n_ranks` copies of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
    127.0.0.1), wait for all of them; non-zero if any failed (the others are then terminated by PID)."""
    if os.environ.get("MIDYN_BENCH_STUB") or os.environ.get("MIDYN_BENCH_SHARE_GPU"):
        have = n_ranks          # plumbing tests: CPU stand-in solve, or all ranks on ONE GPU (see Dist)
    else:
        have = visible_gpus()
    if have < n_ranks:
        print(f"bench.py: --gpus {n_ranks} but only {have} GPU(s) visible; refusing to run a smaller job",
              file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    pending = set(range(n_ranks))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"bench.py: rank {r} exited with {code}; stopping the other ranks", file=sys.stderr, flush=True)
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


class Dist:
    """torch.distributed as rendezvous / barrier / MAX plumbing (nccl = RCCL on the GPU, gloo in the CPU test)."""

    def __init__(self, world, rank, local_rank, stub):
        self.world, self.rank, self.local_rank, self.stub = world, rank, local_rank, stub
        # MIDYN_BENCH_SHARE_GPU (test mode for a ONE-GPU box): all ranks compute on GPU 0 and rendezvous over gloo;
        # every rank builds its own stack (RCCL cannot put two ranks on one device).  Exercises the multi-rank flow of
        # this file -- sharding, per-rank plans, barriers, MAX reduction, the single JSON line -- with real kernels.
        self.share = bool(os.environ.get("MIDYN_BENCH_SHARE_GPU")) and not stub
        self.dist = None
        self.torch = None
        if world > 1 or os.environ.get("MIDYN_BENCH_FORCE_DIST"):
            import torch  # BEFORE the first libmidyn call: the library then binds to torch's HIP runtime
            import torch.distributed as dist

            from qiskit_dynamics_amd.distributed import init_process_group_from_env

            self.torch, self.dist = torch, dist
            if self.share:
                self.local_rank = 0
                torch.cuda.set_device(0)
            elif not stub:
                torch.cuda.set_device(local_rank)
            init_process_group_from_env(backend="gloo" if (stub or self.share) else "nccl")

    @property
    def active(self):
        return self.dist is not None

    def device(self):
        return self.torch.device("cpu") if (self.stub or self.share) else self.torch.device("cuda", self.local_rank)

    def barrier(self):
        if self.dist is not None:
            if not self.stub:
                self.torch.cuda.synchronize()
            self.dist.barrier()

    def max(self, x: float) -> float:
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.device())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def bcast_bytes(self, payload, src=0):
        box = [payload]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def shared_stack(qd, ctx, D, builder, n, k, route):
    """The operator stack on every rank: built on rank 0, ONE RCCL broadcast (route "abi": midyn_stack_broadcast
    on a communicator made from an ncclUniqueId shipped through the torch store; "torch": dist.broadcast of the
    packed buffer).  Returns (stack, keepalive, info)."""
    from qiskit_dynamics_amd import _lib
    from qiskit_dynamics_amd.distributed import broadcast_stack

    t0 = time.perf_counter()
    arrays, perm, build_error = (None, None, None), None, None
    if D.rank == 0 or D.share:
        try:
            built = builder()
            arrays, perm = tuple(built[:3]), built[3]
        except Exception as exc:  # pylint: disable=broad-except
            build_error = repr(exc)
    build_s = time.perf_counter() - t0

    def finish(stack_):
        stack_.set_permutation(perm)      # (None: no internal permutation)
        return stack_

    if not D.active:
        if build_error:
            raise RuntimeError(build_error)
        return finish(qd.Stack(ctx, *arrays)), None, {"route": "none (one rank)", "host_build_s": round(build_s, 2)}
    # every rank learns whether the host build on rank 0 worked BEFORE anybody enters a collective on the stack
    build_error = D.bcast_bytes(build_error)
    if build_error:
        raise RuntimeError("host model build failed on rank 0: " + build_error)
    if not D.share:
        perm = D.bcast_bytes(perm)        # the internal index permutation travels with the stack (a few KB)
    info = {"host_build_s": round(build_s, 2)}
    if D.share:
        info["route"] = "test mode MIDYN_BENCH_SHARE_GPU: every rank builds its own stack on the shared GPU (no collective)"
        return finish(qd.Stack(ctx, *arrays)), None, info
    if (route == "abi" and D.world > 1) or os.environ.get("MIDYN_BENCH_PROBE"):
        # a collective that goes wrong on a new machine hangs rather than fails: try the C-ABI communicator and
        # broadcast on a small stack in a CHILD process per rank first (killed by PID after a time limit); if any
        # rank's child fails, every rank takes the torch.distributed route without touching the C-ABI communicator
        from qiskit_dynamics_amd.distributed import abi_broadcast_probe

        t1 = time.perf_counter()
        uid = D.bcast_bytes(_lib.Comm.unique_id() if D.rank == 0 else None)
        ok, msg = abi_broadcast_probe(D.rank, D.world, ctx.device, uid,
                                      timeout_s=float(os.environ.get("MIDYN_BENCH_PROBE_TIMEOUT", "120")))
        all_ok = D.max(0.0 if ok else 1.0) == 0.0
        info["abi_probe"] = {"ok": all_ok, "s": round(time.perf_counter() - t1, 2)}
        if not all_ok:
            info["abi_probe"]["message"] = msg if not ok else "failed on another rank"
            route = "torch"
    if route == "abi":
        result, err = None, None
        try:
            uid = D.bcast_bytes(_lib.Comm.unique_id() if D.rank == 0 else None)
            comm = _lib.Comm(ctx, D.world, D.rank, uid)
            meta = D.bcast_bytes((arrays[1] is not None, arrays[2] is not None) if D.rank == 0 else None)
            stack = qd.Stack(ctx, *arrays) if D.rank == 0 else _lib.Stack.empty(ctx, n, k, meta[0], meta[1])
            ctx.synchronize()
            D.barrier()
            t1 = time.perf_counter()
            stack.broadcast(comm, 0)
            ctx.synchronize()
            bc_ms = (time.perf_counter() - t1) * 1e3
            result = (stack, comm, meta)
        except Exception as exc:  # pylint: disable=broad-except
            err = repr(exc)
        # the choice of route is collective: one failing rank sends every rank to the torch.distributed route
        if D.max(0.0 if result is not None else 1.0) == 0.0:
            info.update(route="midyn_stack_broadcast (C-ABI, RCCL ncclBroadcast)", broadcast_ms=round(D.max(bc_ms), 3),
                        bytes=_lib.Stack.packed_bytes(n, k, result[2][0]))
            return finish(result[0]), result[1], info
        info["abi_route_error"] = err or "failed on another rank"
    D.barrier()
    t1 = time.perf_counter()
    stack, keep = broadcast_stack(ctx, arrays[0], arrays[1], arrays[2], n, k, src=0)
    info.update(route="torch.distributed broadcast (RCCL)", broadcast_ms=round(D.max((time.perf_counter() - t1) * 1e3), 3))
    return finish(stack), keep, info


def run_dry_ranks(qd, ctx, D, json_out):
    """`bench.py --gpus N --dry-ranks`: the multi-GPU plumbing without the benchmark.  Every rank binds its own device,
    rank 0's ncclUniqueId travels through the torch store, every rank joins the C-ABI communicator
    (midyn_comm_init_rank), a 96-dimensional stack is broadcast from rank 0 (midyn_stack_broadcast) and evaluated on
    every rank against the host arithmetic.  One JSON line: rccl_ranks_seen = ncclCommCount on rank 0, MIN over ranks of
    'my evaluation of the broadcast stack is right'."""
    from qiskit_dynamics_amd import _lib
    from qiskit_dynamics_amd.distributed import PROBE_K, PROBE_N, _probe_arrays

    info = {"mode": "dry-ranks", "n_gpus": D.world, "device_of_rank0": ctx.device}
    ops, static, frame_im = _probe_arrays()
    t0 = time.perf_counter()
    if D.active and not D.share:
        uid = D.bcast_bytes(_lib.Comm.unique_id() if D.rank == 0 else None)
        comm = _lib.Comm(ctx, D.world, D.rank, uid)
        seen = comm.count()
        stack = qd.Stack(ctx, ops, static, frame_im) if D.rank == 0 else _lib.Stack.empty(ctx, PROBE_N, PROBE_K, True, True)
        D.barrier()
        stack.broadcast(comm, 0)
        ctx.synchronize()
    else:                                   # one rank (or the shared-GPU test mode): a communicator of one
        comm = _lib.Comm(ctx, 1, 0, _lib.Comm.unique_id())
        seen = comm.count()
        stack = qd.Stack(ctx, ops, static, frame_im)
        stack.broadcast(comm, 0)
    rng = np.random.default_rng(7)
    y = rng.normal(size=PROBE_N) + 1j * rng.normal(size=PROBE_N)
    c = np.array([0.3, -0.7])
    t = 0.4
    e = np.exp(1j * frame_im * t)
    ref = np.conj(e) * ((static + np.tensordot(c, ops, axes=1)) @ (e * y))
    err = float(np.max(np.abs(stack.eval_rhs(c, t, y) - ref)))
    ok = D.max(0.0 if err < 1e-12 else 1.0) == 0.0
    info.update(rccl_ranks_seen=int(seen), every_rank_evaluates_the_broadcast_stack_correctly=bool(ok),
                max_abs_error_rank0=err, seconds=round(time.perf_counter() - t0, 2))
    comm.close()
    if D.rank == 0:
        print(json.dumps(info), file=json_out, flush=True)
    D.close()


# -----------------------------------------------------------------------------------------------------------------
# CPU plumbing stub (tests/test_distributed_gloo.py): same launcher, sharding, barriers and MAX reduction, a
# stand-in for the device solve.  Never used with a GPU; prints a line marked "stub".
# -----------------------------------------------------------------------------------------------------------------
def run_stub(args, D, json_out):
    from qiskit_dynamics_amd.distributed import shard_bounds

    total = args.batch or SWEEP
    lo, hi = (D.rank * total, (D.rank + 1) * total) if args.weak else shard_bounds(total, D.rank, D.world)
    D.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (1 + D.rank))
    elapsed = D.max(time.perf_counter() - t0)
    n_inst = int(D.max(float(hi)))  # highest shard end == total instances (checks the all-reduce)
    if D.rank == 0:
        print(json.dumps({"metric": "stub", "stub": True, "n_gpus": D.world, "instances_total": n_inst,
                          "scaling": "weak" if args.weak else "strong", "shard_rank0": [lo, hi],
                          "elapsed_max_s": round(elapsed, 4)}), file=json_out, flush=True)
    D.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=16,
                    help="untimed steps before the K timed ones (the first ~60 launches of the contraction run 5-15 %% slower "
                         "while the clocks settle: profiles/r04_rocprof_bench.md, warm-up row)")
    ap.add_argument("--batch", type=int, default=0,
                    help="sweep instances: the TOTAL over all GPUs (default 4096), or per GPU with --weak")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --batch (4096) instances PER GPU")
    ap.add_argument("--torch-broadcast", action="store_true",
                    help="broadcast the stack with torch.distributed (RCCL) instead of the C-ABI midyn_stack_broadcast")
    ap.add_argument("--repeats", type=int, default=3, help="timed repetitions of the K steps (spread is reported)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the cfg-2 single-trajectory leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the cfg-4 / cfg-5 legs")
    ap.add_argument("--dense", action="store_true",
                    help="disable exact-zero plane skipping (time the general dense-complex path)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the end-to-end Solver.solve of the whole sweep (about 12 s)")
    ap.add_argument("--full-solve", action="store_true",
                    help="also time the end-to-end solve with Python-callable envelopes (host-evaluated coefficient table)")
    ap.add_argument("--complex-3m", type=int, default=-1, help="A/B: ctx option complex_3m (0 | 1 | 2)")
    ap.add_argument("--plane-kernel", action="store_true", help="A/B: planar two-tiles-per-barrier kernel (opt-in)")
    ap.add_argument("--ablate", type=int, default=0, help="profiling only: kernel ablation bits (results wrong)")
    ap.add_argument("--force-tile", type=int, default=0, help="0 auto | 64 | 128 (kernel A/B testing)")
    ap.add_argument("--opt", action="append", default=[], help="A/B: ctx option NAME=VALUE (repeatable)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the A/B legs of cfg 3 (GEMM route, no zero-block lists, general complex operators): PMC passes, so "
                         "that the per-dispatch counter averages of the dominant kernel are those of the headline launch only")
    ap.add_argument("--no-rows", action="store_true", help="skip the dense-expm / Lindblad / parallel-in-time / perturbative legs")
    ap.add_argument("--no-projection", action="store_true",
                    help="skip projected_strong_scaling (profiling: keeps shard-sized launches out of the per-kernel averages)")
    ap.add_argument("--dry-ranks", action="store_true",
                    help="multi-GPU plumbing check only: every rank binds its device, the C-ABI communicator is made from a "
                         "shipped ncclUniqueId, a small stack is broadcast and evaluated; prints rccl_ranks_seen (no bench)")
    args = ap.parse_args()
    stub = bool(os.environ.get("MIDYN_BENCH_STUB"))

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(spawn_ranks(args.gpus, sys.argv[1:]))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr, flush=True)
        raise SystemExit(2)
    # ONE JSON line on stdout: libraries (RCCL prints a version banner to stdout at communicator creation) get
    # stderr as their stdout; the line itself goes to a private duplicate of the original descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    D = Dist(world, rank, local_rank, stub)
    if stub:
        run_stub(args, D, json_out)
        return

    import qiskit_dynamics_amd as qd
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.distributed import shard_bounds
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    ctx = qd.default_context(D.local_rank)
    if args.dense:
        ctx.set_option("skip_zero_planes", 0)
    if args.force_tile:
        ctx.set_option("force_tile", args.force_tile)
    if args.ablate:
        ctx.set_option("ablate", args.ablate)
    if args.complex_3m >= 0:
        ctx.set_option("complex_3m", args.complex_3m)
    if args.plane_kernel:
        ctx.set_option("plane_kernel", 1)
    for item in args.opt:
        name, _, val = item.partition("=")
        ctx.set_option(name, int(val))

    if args.dry_ranks:
        run_dry_ranks(qd, ctx, D, json_out)
        return
    cfg = workloads.schrodinger_config(N_QUBITS, N_DRIVES, T_FINAL, MAX_DT)
    n = 2**N_QUBITS
    k = N_DRIVES
    host_arrays = {}

    def builder():
        host_arrays["v"] = build_model_stack(cfg)
        return host_arrays["v"]

    t_setup = time.perf_counter()
    stack, _keep, bcast = shared_stack(qd, ctx, D, builder, n, k, "torch" if args.torch_broadcast else "abi")
    setup_s = time.perf_counter() - t_setup

    # ---- the sweep of this rank ---------------------------------------------------------------------------------
    total_inst = (args.batch or SWEEP) * (world if args.weak else 1)
    lo, hi = shard_bounds(total_inst, rank, world)
    b_loc = hi - lo
    # schedule of the real 1000-step solve; the bench runs its first (warmup + repeats * steps) steps
    sched = FixedStepSchedule(cfg["t_span"], None, MAX_DT, _rk4_points)
    repeats = max(1, args.repeats)
    total = args.warmup + repeats * args.steps
    while total > len(sched.step_h) and repeats > 1:
        repeats -= 1
        total = args.warmup + repeats * args.steps
    if total > len(sched.step_h):
        raise SystemExit(f"warmup+steps must be <= {len(sched.step_h)}")
    rows = sched.step_rows[:total]
    n_rows = int(rows.max()) + 1
    times = sched.times[:n_rows]
    table, amps, phs = sweep_table(workloads, times, lo, b_loc, k, cfg["carrier"], T_FINAL)
    y0 = cfg["y0"].reshape(-1, 1)
    plan = qd.Rk4Plan(stack, times, table, rows, sched.step_h[:total], y0, b_loc, True)

    def sync_all():
        ctx.synchronize()
        D.barrier()

    plan.run(0, args.warmup)
    sync_all()
    samples = []          # (host seconds, HIP-event ms) of each repetition of K steps, MAX over ranks
    for rep in range(repeats):
        s0 = args.warmup + rep * args.steps
        sync_all()
        t0 = time.perf_counter()
        ctx.timer_start()
        plan.run(s0, s0 + args.steps)
        ev_ms = ctx.timer_stop()          # waits for the stop event = the stream is drained
        ctx.synchronize()
        el = time.perf_counter() - t0
        D.barrier()
        samples.append((D.max(el), D.max(ev_ms)))
    elapsed, event_ms = samples[0]        # the contract's "EXACTLY K steps": the first timed repetition
    final = plan.fetch()[:, :, 0]
    norm_dev = float(np.max(np.abs(np.linalg.norm(final, axis=1) - 1.0)))

    evals = total_inst * 4 * args.steps
    value = evals / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    rates = [total_inst * 4 * args.steps / s for s, _ in samples]

    # ---- roofline of the dominant kernel ------------------------------------------------------------------------
    # launch duration from the SAME back-to-back region as ms_per_step: HIP events on the stream the kernel runs on
    # around the K steps, divided by the 4K launches (+ a per-launch event pass for the launch count / class check)
    prof_steps = min(args.steps, 10)
    cnts = profile_pass(ctx, lambda: plan.run(total - prof_steps, total), ("rhs_gemm", "rhs_blocks_gemm", "rhs_combine"))
    on_combine = cnts["rhs_combine"]["launches"] > 0        # sweeps: combine (MFMA over the operator planes) + apply (VALU)
    on_lists = cnts["rhs_blocks_gemm"]["launches"] > 0      # symmetry sectors: the contraction runs on tile work lists
    cnt = cnts["rhs_combine"] if on_combine else (cnts["rhs_blocks_gemm"] if on_lists else cnts["rhs_gemm"])
    roofline = None
    n_launch_per_step = cnt["launches"] / prof_steps if cnt["launches"] else 0
    modes = stack.segment_modes if not args.dense else [0] * stack.n_segments
    useful = (4 * k + 10) * n * n * b_loc                          # SURVEY 8(d) cfg 3 "useful" flops per launch
    if cnt["launches"] > 0 and abs(n_launch_per_step - 4) < 1e-9:
        avg_ms = event_ms / (4 * args.steps)
        act_modes = [m for m in stack.segment_modes if m != 3]
        stack_um = act_modes[0] if act_modes and all(m == act_modes[0] for m in act_modes) else 3
        single_plane = (not args.dense) and all(m in (1, 2) for m in act_modes)
        extra = {}
        if on_combine:
            info = ctx.counters("combine_info")
            shape = ctx.counters("combine_shape")
            entries, code = info["launches"], int(info["ms"])
            nre4, nim4, stat = code // 100, (code // 10) % 10, code % 10
            kinds = int(nre4 > 0 or (stat & 1)) + int(nim4 > 0 or (stat & 2))
            pairs = entries * 16 * 32 * float(p_ld := (-(-b_loc // 128) * 128))   # listed (row, kk) positions x state columns
            mfma_flops = 2.0 * 4 * (nre4 + nim4) * pairs          # one MFMA-FMA per plane slot (padding slots included)
            valu_flops = 2.0 * (2 if kinds == 1 else 4) * pairs   # apply: 2 (one plane kind) or 4 vector FMAs
            executed = mfma_flops + valu_flops
            gemm_equiv = sum(4 if m in (1, 2) else 8 for m in modes if m != 3) * n * n * b_loc
            kname = "rhs_combine_kernel<%d, %d, %d>" % (nre4, nim4, stat)
            extra = {"combine_apply": {
                "listed_entries": int(entries), "listed_fraction": round(entries / ((n // 32) * (n // 16)), 4),
                "plane_groups_re_im": [nre4, nim4], "static_planes": stat,
                "row_group_x_column_block_pairs_per_workgroup": int(shape["launches"]), "list_splits": int(shape["ms"]),
                "mfma_flops_per_launch": mfma_flops, "vector_fma_flops_per_launch": valu_flops,
                "fmas_per_row_kk_instance": 4 * (nre4 + nim4) + (2 if kinds == 1 else 4),
                "dense_gemm_formulation_flops_per_launch": gemm_equiv,
                "why": "sum_j c_j[b] G_j is combined per instance first (one v_mfma_f64_16x16x4 per 4 operator planes: 16 rows x "
                       "16 instances x 4 planes, every FMA useful) and then applied to the state (2 vector FMAs per element for "
                       "purely imaginary generators) -- the reference's own order of operations (operator_collections.py:"
                       "101-134) -- instead of k + 1 GEMMs (2 MFMA-FMAs per plane and element); exactly-zero 16-column "
                       "blocks of a 32-row group (parity sectors) are not listed"}}
        elif on_lists:
            tile = ctx.counters("sparse_tile")
            lst = ctx.counters("sparse_list")
            bm, bn = int(tile["launches"]), int(tile["ms"])
            listed = lst["launches"]
            cols = -(-b_loc // bn) * bn
            executed = listed * bm * 16 * cols * (4 if single_plane else 8)   # listed (BM x 16) tiles x all columns
            blk = stack.block_info()
            kname = "zgemm_seg_kernel<%d, %d, 2, 4, 16, %d, 2, true>" % (bm, bn, stack_um)
            extra = {"work_lists": {"tile": [bm, bn], "listed_tiles": int(listed), "splits": int(lst["ms"]),
                                    "listed_fraction": round(blk["tile_lists"].get(bm, {}).get("listed_fraction", 0.0), 4),
                                    "why": "the frame operator H_d conserves parity: its eigenvectors are computed sector by "
                                           "sector (exactly zero outside their sector), so every frame-basis operator has "
                                           "exactly-zero blocks between sectors it does not couple; skipping them is "
                                           "bit-identical to multiplying the zeros (see dense_kernels_same_model)"}}
        else:
            use_3m = (args.dense or stack_um == 0) and args.complex_3m != 0
            per_seg = [(6 if use_3m else 8) if m == 0 else 4 for m in modes if m != 3]
            executed = sum(per_seg) * n * n * b_loc                # real MFMA flops the kernel executes per launch
            kname = ("zgemm_seg_kernel<64, 64, 2, 2, 16, 4," if (args.dense or stack_um == 0)
                     else "zgemm_seg_kernel<128, 128, 2, 4, 16, %d, 2, false>" % stack_um)
        tf = executed / (avg_ms * 1e-3) / 1e12
        traffic3 = measured_traffic(kname)
        roofline = {
            "kernel": kname.rstrip(",") + (" (sweep contraction: fp64 MFMA 16x16x4 over the operator planes + fp64 vector FMAs; "
                                           "both on the SIMD's one fp64 pipe, whose matrix peak and vector peak are the same "
                                           "78.6 TFLOP/s)" if on_combine else " (batched RHS contraction, fp64 MFMA 16x16x4)"),
            "bound": "mfma",
            "achieved": round(tf, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4),
            "traffic": traffic3[0], "traffic_source": traffic3[1],
            "avg_launch_ms": round(avg_ms, 4), "launches_timed": 4 * args.steps,
            "avg_launch_ms_per_launch_events": round(cnt["ms"] / cnt["launches"], 4),
            # dispatch order of THIS kernel in this process (what tools/summarize_rocprof.py slices a rocprofv3
            # --kernel-trace of the same command by): warm-up, the timed repetitions (back to back), then the pass that
            # brackets every launch with its own HIP events
            "launch_sequence": [["warmup", 4 * args.warmup]] + [["timed_rep%d" % r, 4 * args.steps] for r in range(repeats)]
                               + [["per_launch_events", 4 * prof_steps]],
            "executed_mfma_flops_per_launch": executed,
            "useful_flops_per_launch": useful, "useful_tflops": round(useful / (avg_ms * 1e-3) / 1e12, 3),
            "frac_survey_8d": round(useful / (avg_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
            "frac_survey_8d_note": "SURVEY 8(d) prices an instance-evaluation at (4k+10)n^2 = 44.0 MFLOP; this model's kernel "
                                   "executes %.1f MFLOP of them (purely imaginary operators: one real product per plane to "
                                   "combine, two to apply -- the GEMM formulation: 2 of 4 real products per plane --; parity "
                                   "sectors: half of the blocks; the static operator in its own frame is exactly zero), so "
                                   "the survey figure exceeds 1 -- frac counts what is executed" % (executed / b_loc / 1e6),
            "segment_plane_modes": modes, "active_segments": stack.n_active_segments,
            "zero_plane_skipping": not args.dense, **extra,
            "note": "achieved = EXECUTED real MFMA flops / launch time (launch time = HIP events around the K timed steps "
                    "on the library's stream / 4K launches).  Exact zeros are not multiplied: the operators of this model "
                    "are purely imaginary in the frame basis (4 instead of 8 real flops per complex MAC) and vanish "
                    "between parity sectors they do not couple (work lists); SURVEY 8(d) counts (4k+10)n^2 'useful' "
                    "flops per instance-evaluation (useful_tflops, may exceed the peak for that reason); see "
                    "dense_kernels_same_model and dense_complex for the same sweep without those savings",
        }

    def timed_variant(options, steps_=10):
        """ms per launch of the same sweep on a fresh plan created under `options` (routes are chosen at plan creation)."""
        with ctx.options(**options):         # (every option back to the value it had)
            pv = qd.Rk4Plan(stack, times, table, rows, sched.step_h[:total], y0, b_loc, True)
            w_steps = max(1, min(2, total - 1))
            d_steps = max(1, min(args.steps, steps_, total - w_steps))
            pv.run(0, w_steps)
            ctx.synchronize()
            ctx.timer_start()
            pv.run(w_steps, w_steps + d_steps)
            ms_ = ctx.timer_stop() / (4 * d_steps)
            pv.close()
        return ms_

    # ---- the same sweep without the symmetry-sector work lists, and as general dense-complex operators ----------
    dense = None
    same_model_dense = None
    gemm_route = None
    if not args.dense and roofline and world == 1 and on_combine and not args.no_variants:
        # the MFMA GEMM formulation of the same sweep (the default route until round 3): work-list tiles, k + 1 GEMMs
        avg_g = timed_variant({"combine": 0})
        ex_g = None
        lst_g = ctx.counters("sparse_list")
        tile_g = ctx.counters("sparse_tile")
        if lst_g["launches"] > 0:
            ex_g = lst_g["launches"] * tile_g["launches"] * 16 * (-(-b_loc // int(tile_g["ms"])) * int(tile_g["ms"])) * 4
        gemm_route = {"option": "combine=0", "avg_launch_ms": round(avg_g, 4), "rhs_evals_per_s": round(b_loc / (avg_g * 1e-3), 1),
                      "executed_tflops": round(ex_g / (avg_g * 1e-3) / 1e12, 3) if ex_g else None,
                      "frac": round(ex_g / (avg_g * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4) if ex_g else None,
                      "what": "zgemm_seg_kernel<128,128,...,SPARSE> on the sector work lists: k GEMMs whose results are scaled "
                              "and summed (16 MFMA-FMAs per element against 10 FMAs of combine + apply); same results to rounding"}
    if not args.dense and roofline and world == 1 and not args.no_variants:
        plan.close()
        if on_lists or on_combine:
            avg_k = timed_variant({"skip_zero_blocks": 0, "combine": 0} if on_combine else {"skip_zero_blocks": 0})
            ex_k = sum(4 if m in (1, 2) else 8 for m in modes if m != 3) * n * n * b_loc
            no_lists = None
            if on_combine:     # the default route of a model with the same planes but NO exactly-zero blocks
                avg_n = timed_variant({"skip_zero_blocks": 0})
                info_n = ctx.counters("combine_info")
                code_n = int(info_n["ms"])
                nq_n = code_n // 100 + (code_n // 10) % 10
                kinds_n = int(code_n // 100 > 0 or (code_n % 10) & 1) + int((code_n // 10) % 10 > 0 or (code_n % 10) & 2)
                ex_n = (2.0 * 4 * nq_n + 2.0 * (2 if kinds_n == 1 else 4)) * info_n["launches"] * 16 * 32 * (-(-b_loc // 128) * 128)
                no_lists = {"option": "skip_zero_blocks=0", "avg_launch_ms": round(avg_n, 4),
                            "rhs_evals_per_s": round(b_loc / (avg_n * 1e-3), 1),
                            "executed_tflops": round(ex_n / (avg_n * 1e-3) / 1e12, 3),
                            "frac": round(ex_n / (avg_n * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                            "what": "combine + apply over ALL (32-row, 16-column) entries: what a model with purely imaginary "
                                    "generators but no conserved quantity runs at"}
            same_model_dense = {"default_route_without_zero_block_lists": no_lists,
                                "avg_launch_ms": round(avg_k, 4), "rhs_evals_per_s": round(b_loc / (avg_k * 1e-3), 1),
                                "executed_tflops": round(ex_k / (avg_k * 1e-3) / 1e12, 3),
                                "frac": round(ex_k / (avg_k * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                                "what": "dense 128x128 MFMA GEMM kernel on the same stack (skip_zero_blocks=0, combine=0): "
                                        "multiplies the exactly-zero blocks too"}
        avg_d = timed_variant({"skip_zero_planes": 0, "skip_zero_blocks": 0})
        info_d = ctx.counters("combine_info")
        took_combine_d = on_combine and info_d["launches"] > 0 and int(info_d["ms"]) // 100 > 0
        if took_combine_d:
            code_d = int(info_d["ms"])
            nq_d = code_d // 100 + (code_d // 10) % 10
            ex_d = (2.0 * 4 * nq_d + 8.0) * info_d["launches"] * 16 * 32 * (-(-b_loc // 128) * 128)
        else:
            ex_d = 6 * stack.n_segments * n * n * b_loc   # 3M: 3 real MFMA products per complex product
        dense = {"one_wave_per_simd_64_instance_waves": "measured in round 5 and not kept: rhs_combine_wide_kernel<2, 2, 3> 3.27 ms per launch against 2.82 (profiles/r05_summary.md)",
                 "avg_launch_ms": round(avg_d, 4), "rhs_evals_per_s": round(b_loc / (avg_d * 1e-3), 1),
                 "useful_tflops": round(useful / (avg_d * 1e-3) / 1e12, 3),
                 "frac_survey_8d_useful": round(useful / (avg_d * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                 "executed_tflops": round(ex_d / (avg_d * 1e-3) / 1e12, 3),
                 "frac": round(ex_d / (avg_d * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                 "scheme": ("no structure exploited (every operator treated as complex, no zero planes / blocks skipped): "
                            "combine + apply, 8 real + 8 imaginary planes through the MFMAs, the static operator as their C "
                            "input, 4 vector FMAs per element (rhs_combine_kernel<2, 2, 3>)") if took_combine_d else
                           ("no structure exploited (general complex operators): 3M complex multiplication (3 real fp64 "
                            "MFMAs per complex product) inside the solver loop, dense 64x64 tiles")}
        if took_combine_d:
            avg_3m = timed_variant({"skip_zero_planes": 0, "skip_zero_blocks": 0, "combine": 0})
            ex_3m = 6 * stack.n_segments * n * n * b_loc
            dense["mfma_gemm_3m_route"] = {"option": "combine=0", "avg_launch_ms": round(avg_3m, 4),
                                           "rhs_evals_per_s": round(b_loc / (avg_3m * 1e-3), 1),
                                           "executed_tflops": round(ex_3m / (avg_3m * 1e-3) / 1e12, 3),
                                           "frac": round(ex_3m / (avg_3m * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4)}
    plan.close()
    # ---- what the N-GPU strong-scaling runs should show: the per-GPU shards of 2 / 4 / 8 ranks timed on THIS GPU --------
    projected = None
    if rank == 0 and world == 1 and not args.dense and not args.weak and total_inst % 8 == 0 and not args.no_projection:
        projected = {"what": "strong scaling of this sweep projected from one GPU: the shard of an N-GPU run (instances / N) timed "
                             "here with the same plan type (best of three stretches of K steps); efficiency = shard rate / full rate, both "
                             "as the best of their repetitions.  The stack broadcast "
                             "is outside the timed region of bench.py and of this projection",
                     "cfg3": {}}
        for n_r in (2, 4, 8):
            b_s = total_inst // n_r
            tab_s = table[:b_s]
            ps = qd.Rk4Plan(stack, times, tab_s, rows, sched.step_h[:total], y0, b_s, True)
            # (warm-up over the first steps, then the best of up to three back-to-back stretches of the same length: a shard's
            # launches are short, and the first ones after a change of kernel shape run at unsettled clocks)
            w_steps = max(0, min(args.warmup, total - 1))
            if w_steps:
                ps.run(0, w_steps)
            ctx.synchronize()
            d_steps = max(1, min(args.steps, (total - w_steps) // 3 or 1, total - w_steps))
            el_p = None
            for rep_p in range(3):
                lo = w_steps + rep_p * d_steps
                if lo + d_steps > total:
                    break
                t0p = time.perf_counter()
                ps.run(lo, lo + d_steps)
                ctx.synchronize()
                el_rep = time.perf_counter() - t0p
                el_p = el_rep if el_p is None else min(el_p, el_rep)
            ps.close()
            rate = b_s * 4 * d_steps / el_p
            projected["cfg3"][str(n_r)] = {"instances_per_gpu": b_s, "us_per_batched_evaluation": round(el_p / (4 * d_steps) * 1e6, 1),
                                          "projected_value": round(n_r * rate, 1), "efficiency": round(rate / max(rates), 4)}
    measured_peaks = None
    if rank == 0 and world == 1:
        co = ctx.microbench("fp64_coissue", full=True)
        sus = ctx.microbench("mfma_f64_sustained", full=True)
        sus0 = ctx.microbench("mfma_f64_sustained_zero", full=True)
        measured_peaks = {"mfma_f64_tflops": round(ctx.microbench("mfma_f64"), 1),
                          "mfma_f64_sustained_random_operands": {
                              "tflops": round(sus[0], 2), "frac_of_78.6": round(sus[0] / FP64_MFMA_PEAK_TFLOPS, 4),
                              "shader_clock_ghz": round(sus[1], 3), "ms": round(sus[2], 2),
                              "all_zero_operands": {"tflops": round(sus0[0], 2), "shader_clock_ghz": round(sus0[1], 3)},
                              "note": "bare v_mfma_f64_16x16x4 stream at the contraction kernels' cadence (2 waves per SIMD, 16 "
                                      "accumulator quads) for ~14 ms, operands with random mantissas and signs, with the shader "
                                      "clock it sustained (the chip clocks to its power budget; 78.6 TFLOP/s is quoted at "
                                      "2.4 GHz): what a kernel that did nothing but MFMAs on such data reaches; roofline.frac "
                                      "stays against 78.6"},
                          "hbm_read_gbs": round(ctx.microbench("hbm_read"), 0),
                          "mall_read_gbs": round(ctx.microbench("mall_read"), 0),
                          "fp64_mfma_and_vector_fma_share_a_pipe": {
                              "ms_interleaved": round(co[0], 3), "ms_mfma_only": round(co[1], 3),
                              "ms_vector_fma_only": round(co[2], 3),
                              "note": "interleaved = sum, not max: every fp64 VALU instruction in the contraction loop "
                                      "costs matrix-pipe time (tools/mfma_bank_probe.hip: so does every other VALU "
                                      "instruction, 3-7 cycles each: 96 v_mov_b32 per 64 MFMAs run at 0.93 of the peak)"}}

    scaling = "weak" if args.weak else "strong"
    out = {
        "metric": "RHS evals/sec, 10-qubit Schrodinger (dim 1024), 4096-param batch", "value": round(value, 1),
        "unit": "RHS evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": scaling,
        "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
        "config": {"workload": "cfg3: 10-qubit chain, n=1024, k=8 drives + static, rotating frame H_d, RK4 max_dt=0.005, "
                               "sweep of %d instances in total (%s scaling: %d per GPU)" % (total_inst, scaling, b_loc),
                   "instances_per_gpu": b_loc, "global_instances": total_inst,
                   "rhs_evals_per_step": 4 * total_inst, "parallelism": f"sweep-shard x{world}"},
        "repeat_rhs_evals_per_s": [round(r, 1) for r in rates],
        "repeat_spread": round((max(rates) - min(rates)) / max(rates), 4),
        "stream_ms_per_step": round(event_ms / args.steps, 4),
        "solve_wall_clock_s": round(ms_per_step * len(sched.step_h) / 1e3, 3),
        "solve_wall_clock_note": "1000 RK4 steps = 1000 x ms_per_step (fixed step, identical work per step)",
        "setup_s": round(setup_s, 2), "stack_broadcast": bcast, "max_norm_deviation": norm_dev,
    }
    if roofline:
        out["roofline"] = roofline
    if same_model_dense:
        out["dense_kernels_same_model"] = same_model_dense
        nl = same_model_dense.get("default_route_without_zero_block_lists")
        out["value_without_exact_zero_block_skipping"] = nl["rhs_evals_per_s"] if nl else same_model_dense["rhs_evals_per_s"]
        out["value_note"] = ("value = the product's default route: the reference's arithmetic (combine the operators with the "
                             "instance's coefficients, apply to the state) minus products with operator blocks that are "
                             "EXACTLY zero (parity sectors of the frame operator); value_without_exact_zero_block_skipping = "
                             "the same route over all blocks.  On both the static operator in the frame, U^+(G_d - F)U "
                             "with F = G_d, is exactly zero and inactive (the reference's U^+ G_d U - diag(d) leaves 1e-13 "
                             "rounding noise there).  mfma_gemm_route_same_model / dense_kernels_same_model: the k + 1 GEMM "
                             "formulation (default until round 3) on the same stack")
    if gemm_route:
        out["mfma_gemm_route_same_model"] = gemm_route
    if projected:
        out["projected_strong_scaling"] = projected
    if dense:
        out["dense_complex"] = dense
    if same_model_dense and dense:
        out["cfg3_three_numbers"] = {
            "structured_model_default_route": round(value, 1),
            "same_planes_no_zero_blocks": (same_model_dense.get("default_route_without_zero_block_lists") or same_model_dense)["rhs_evals_per_s"],
            "general_complex_operators": dense["rhs_evals_per_s"],
            "unit": "RHS evals/s on one GPU, 4096 instances",
            "read_as": "all three on the product's default route (combine + apply).  value (the first number) belongs to THIS "
                       "model: real Hamiltonians in their own eigenbasis with a parity symmetry.  A model with dense complex "
                       "frame-basis operators and a static operator (e.g. a random Hermitian frame) runs at the third number; "
                       "a model with real Hamiltonians but no conserved quantity at the second"}
    if measured_peaks:
        out["measured_peaks"] = measured_peaks

    # ---- cfg 2: single trajectory, HBM-bound streaming kernel (rank 0, N=1) -------------------------------------
    if rank == 0 and world == 1 and not args.no_single:
        s_total = min(len(sched.step_h), 1000)      # the whole configs[1] trajectory (one launch on the default route)
        rows1 = sched.step_rows[:s_total]
        nr1 = int(rows1.max()) + 1
        table1 = workloads.gaussian_coefficient_table(sched.times[:nr1], amps[:1], phs[:1], cfg["carrier"], T_FINAL)

        def one_trajectory(resident):
            """(wall seconds, stream ms, counters) of steps 8..s_total of one trajectory on the chosen route"""
            ctx.set_option("resident_rk4", 1 if resident else 0)
            try:
                p1 = qd.Rk4Plan(stack, sched.times[:nr1], table1, rows1, sched.step_h[:s_total], y0, 1, True)
                p1.run(0, 8)
                ctx.synchronize()
                t0_ = time.perf_counter()
                ctx.timer_start()
                p1.run(8, s_total)
                ev_ = ctx.timer_stop()
                el_ = time.perf_counter() - t0_
                cn_ = profile_pass(ctx, lambda: p1.run(8, s_total), ("rhs_stream", "rk4_resident"))
                yfin = p1.fetch()
                p1.close()
            finally:
                ctx.set_option("resident_rk4", 1)
            return el_, ev_, cn_, yfin

        el_res, ev_res, cn_res, y_res = one_trajectory(True)
        el1, ev1, cn1, y_stage = one_trajectory(False)
        # the floor of the default route: the same launch geometry with the arithmetic switched off -- every round still
        # publishes its row and polls its neighbours' rows (results are meaningless, the time is the store -> poll hop)
        os.environ["MIDYN_DEBUG_OPTIONS"] = "1"     # the library refuses this measurement switch otherwise
        ctx.set_option("resident_exchange_only", 1)
        try:
            el_floor, ev_floor, cn_floor, _ = one_trajectory(True)
        finally:
            ctx.set_option("resident_exchange_only", 0)
            os.environ.pop("MIDYN_DEBUG_OPTIONS", None)
        c1 = cn1["rhs_stream"]
        took_resident = cn_res["rk4_resident"]["launches"] > 0
        nseg = stack.n_segments
        bytes_alg = 16 * nseg * n * n + 32 * n                 # SURVEY 8(d) cfg 2: 151.03 MB
        avg_ms1 = ev1 / (4 * (s_total - 8))                    # back-to-back launches, same region as ms_per_step
        # single-plane stack (every operator purely real or purely imaginary) and plane skipping on:
        # the kernel streams only the non-zero planes, 8 B per operator element instead of 16 B
        planar = (not args.dense) and all(m in (1, 2, 3) for m in stack.segment_modes)
        n_act = sum(1 for m in stack.segment_modes if m != 3)
        streamed = stack.block_info()["streamed_fraction"] if planar else 1.0   # column hulls (symmetry sectors)
        executed_bytes = (8 * n_act * n * n * streamed + 32 * n) if planar else bytes_alg
        gbs_exec = executed_bytes / (avg_ms1 * 1e-3) / 1e9
        kname = "rhs_stream_plane_kernel<2, 3>" if planar else "rhs_stream_kernel<4, 3>"
        n_eval1 = 4 * (s_total - 8)
        out["single_trajectory"] = {
            "workload": "cfg2 (BASELINE configs[1]): same model, 1 trajectory, RK4",
            "rhs_evals_per_s": round(n_eval1 / el_res, 1), "ms_per_step": round(el_res / (s_total - 8) * 1e3, 5),
            "us_per_evaluation": round(el_res / n_eval1 * 1e6, 3),
            "route": "rk4_resident_kernel (one launch for the step range; operator planes in registers, stage input "
                     "through a polled ring in device memory)" if took_resident else "per-stage streaming kernel",
            "bound": "exchange latency: one store -> poll hop across the XCDs per stage (no operator traffic after "
                     "the launch has loaded its rows)" if took_resident else "hbm",
            "per_stage_route_rhs_evals_per_s": round(n_eval1 / el1, 1),
            "max_abs_difference_between_the_routes": float(np.max(np.abs(y_res - y_stage))),
            "steps_timed": s_total - 8}
        if took_resident and cn_floor["rk4_resident"]["launches"] > 0:
            us_round, us_floor = ev_res / n_eval1 * 1e3, ev_floor / n_eval1 * 1e3
            out["roofline_single_trajectory_default_route"] = {
                "kernel": "rk4_resident_kernel<8, 8, true> (one launch per step range; one exchange round per RHS evaluation)",
                "bound": "exchange latency (store -> poll hop through device memory across the XCDs)",
                "achieved": round(us_round, 3), "floor": round(us_floor, 3), "unit": "us per round (= per RHS evaluation)",
                "frac": round(us_floor / us_round, 4),
                "arithmetic_us_per_round": round(us_round - us_floor, 3),
                "workgroups": 128, "rounds_timed": n_eval1,
                "note": "floor = the same launch with option resident_exchange_only (every round publishes and polls, the row "
                        "product is skipped), measured in this run; frac = floor / achieved: the share of a round that is the "
                        "exchange.  No operator byte moves after the launch has loaded its rows, so neither the HBM nor the "
                        "MFMA roofline applies; MI355X_MICROARCH.md prices an idle 1-to-1 hand-off at 0.8-1.1 us and "
                        "2.3-2.9 us between loaded CUs"}
        out["roofline_single_trajectory"] = {
            "route": "per-stage streaming kernel (option resident_rk4=0): the kernel of evaluate_rhs and of single "
                     "trajectories whose operators do not fit the register files",
            "kernel": kname.split("<")[0], "bound": "hbm", "achieved": round(gbs_exec, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs_exec / HBM_PEAK_GBS, 4), "traffic": measured_traffic(kname)[0],
            "traffic_source": measured_traffic(kname)[1],
            "avg_launch_ms": round(avg_ms1, 5), "launches_timed": 4 * (s_total - 8),
            "avg_launch_ms_per_launch_events": round(c1["ms"] / max(c1["launches"], 1), 5),
            "executed_bytes_per_launch": executed_bytes, "streamed_fraction_of_the_planes": round(streamed, 4),
            "algorithmic_bytes_per_launch": bytes_alg,
            "algorithmic_gbs": round(bytes_alg / (avg_ms1 * 1e-3) / 1e9, 1),
            "note": "achieved = EXECUTED bytes / launch time; the operators of this model are purely imaginary, so the "
                    "kernel streams only their non-zero planes (75.5 MB; SURVEY 8(d) counts the full complex128 stack, "
                    "151 MB = algorithmic_bytes, exact same results); the planes fit the 256 MB Infinity Cache, so the "
                    "bytes come from MALL rather than HBM after the first launch"}

    # ---- cfg 4 / cfg 5 (rank 0, N=1): the other BASELINE configurations with their rooflines --------------------
    cfg5 = None
    stack5 = None
    want_cfg5 = not args.no_configs and not args.dense
    if rank == 0 and world == 1 and want_cfg5:
        try:
            out["cfg4"] = leg_cfg4(qd, ctx, workloads)
            out["cfg4"]["diag_frame_run"] = leg_cfg4_diag_frame(qd, ctx, workloads)
        except Exception as exc:  # pylint: disable=broad-except
            out["cfg4"] = {"error": repr(exc)}
        try:
            out["small_system_sweeps"] = leg_small_sweeps(qd, workloads)
        except Exception as exc:  # pylint: disable=broad-except
            out["small_system_sweeps"] = {"error": repr(exc)}
        try:
            out["diag_frame_rk4_sweep"] = leg_diag_frame_sweep(qd, ctx, workloads)
        except Exception as exc:  # pylint: disable=broad-except
            out["diag_frame_rk4_sweep"] = {"error": repr(exc)}
        # rows a11 / f2 / f3 / f4 of SURVEY section 8 on the driver's line: throughput, roofline of the dominant kernel from the
        # library's own counters, and the CPU algorithm on this host beside each
        for key_, leg_ in (("dense_expm", lambda: leg_dense_expm(qd, ctx)),
                           ("lindblad_rk4_unvectorized", lambda: leg_lindblad_rk4(qd, ctx, workloads)),
                           ("parallel_in_time", lambda: leg_parallel_in_time(qd, ctx, workloads)),
                           ("perturbative", lambda: leg_perturbative(qd, ctx))):
            if args.no_rows:
                break
            try:
                out[key_] = leg_()
            except Exception as exc:  # pylint: disable=broad-except
                out[key_] = {"error": repr(exc)}
    if want_cfg5:
        # second sharded leg: cfg 5, 1024 instances in total, 1024/N per GPU; the 2.4 GB stack is broadcast
        try:
            cfg5 = workloads.schrodinger_config(n_qubits=12, n_drives=8, t_final=5.0, max_dt=0.25) \
                if (rank == 0 or D.share) else dict(t_span=[0.0, 5.0], max_dt=0.25, t_final=5.0)
            if rank != 0 and not D.share:
                nu = 5.0 + 0.05 * np.arange(12)
                y5 = np.zeros(4096, dtype=complex)
                y5[0] = 1.0
                cfg5.update(carrier=nu[:8].copy(), y0=y5)
            stack5, _keep5, bcast5 = shared_stack(qd, ctx, D, lambda: build_diag_frame_stack(cfg5), 4096, 8,
                                                  "torch" if args.torch_broadcast else "abi")
            if rank == 0 and world == 1:
                shard = leg_cfg5(qd, ctx, workloads, stack5, cfg5, 0, 128)
                shard["workload"] = ("cfg5 shard: 12-qubit (n=4096) Schrodinger, k=8, diagonal frame, scipy_expm "
                                     "magnus_order=2, max_dt=0.25, 20 steps, 128 instances (1024-instance sweep / 8 GPUs)")
                out["cfg5"] = shard
            lo5, hi5 = shard_bounds(CFG5_SWEEP, rank, world)
            D.barrier()
            full = leg_cfg5(qd, ctx, workloads, stack5, cfg5, lo5, hi5 - lo5, with_profile=False)
            solve5 = D.max(full["solve_s"])
            if rank == 0:
                out["sharded_cfg5"] = {
                    "workload": "cfg5: 12-qubit (n=4096) Schrodinger sweep, 1024 instances in total, Magnus-2 expm, "
                                "20 steps, sharded 1024/N per GPU (strong scaling)",
                    "instances_total": CFG5_SWEEP, "instances_per_gpu": hi5 - lo5, "n_gpus": world,
                    "solve_s_max_over_ranks": round(solve5, 6),
                    "instance_steps_per_s": round(CFG5_SWEEP * full["steps"] / solve5, 1),
                    "max_norm_deviation_rank0": full["max_norm_deviation"], "stack_broadcast": bcast5}
                if world == 1 and "projected_strong_scaling" in out:
                    proj5 = {}
                    proj5p = {}
                    plan_full = full.get("plan", {}).get("solve_s")
                    for n_r in (2, 4, 8):
                        sh = leg_cfg5(qd, ctx, workloads, stack5, cfg5, 0, CFG5_SWEEP // n_r, with_profile=False)
                        proj5[str(n_r)] = {"instances_per_gpu": CFG5_SWEEP // n_r, "solve_s": sh["solve_s"],
                                           "efficiency": round(solve5 / (n_r * sh["solve_s"]), 4)}
                        plan_sh = sh.get("plan", {}).get("solve_s")
                        if plan_full and plan_sh:
                            proj5p[str(n_r)] = {"instances_per_gpu": CFG5_SWEEP // n_r, "solve_s": plan_sh,
                                                "efficiency": round(plan_full / (n_r * plan_sh), 4)}
                    out["projected_strong_scaling"]["cfg5"] = proj5
                    if proj5p:      # repeated solves through midyn_expm_plan_* (tables and buffers made once per model + time grid)
                        out["projected_strong_scaling"]["cfg5_plan"] = proj5p
        except Exception as exc:  # pylint: disable=broad-except
            if rank == 0:
                out["sharded_cfg5"] = {"error": repr(exc)}
        stack5 = None

    # ---- CPU baseline: the NumPy oracle on this host, bounded sample (rank 0, N=1 only) -------------------------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        ops, static, frame_im = host_arrays["v"][:3]
        out["cpu_baseline"] = leg_cpu_baseline(workloads, cfg, static, ops, frame_im, amps, phs)
        if not args.no_configs and not args.dense:
            try:       # one reference-algorithm step of cfg 4 / cfg 5 on this host, beside their device numbers
                cpu_cfg = leg_cpu_configs(workloads, out["cpu_baseline"]["cores_for_blas3"])
                for key in ("cfg4", "cfg5"):
                    if key in cpu_cfg and isinstance(out.get(key), dict):
                        out[key]["cpu_baseline"] = cpu_cfg[key]
            except Exception as exc:  # pylint: disable=broad-except
                out["cpu_baseline_configs_error"] = repr(exc)

    # ---- the complete cfg-3 solve through the public Solver API: model build, signal evaluation, PCIe
    #      and result unpacking included (rank 0, N=1; --full-solve adds the host-table variant) ------------------
    if rank == 0 and world == 1 and not args.no_end_to_end:
        t0f = time.perf_counter()
        solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"], rotating_frame=cfg["h_d"])
        t_model = time.perf_counter() - t0f
        sig_lists = []
        for b in range(b_loc):
            sig_lists.append([qd.Signal(lambda t, a=a: a * np.exp(-((t - T_FINAL / 2) ** 2) / (2 * 1.0**2)), nu, ph)
                              for a, nu, ph in zip(amps[b], cfg["carrier"], phs[b])])
        # pulses as DiscreteSignals (samples + carrier), the form pulse schedules arrive in: the coefficient
        # table is evaluated on the device
        disc_lists = [[qd.DiscreteSignal.from_Signal(sg, dt=0.05, n_samples=int(round(T_FINAL / 0.05))) for sg in sl]
                      for sl in sig_lists]
        t2f = time.perf_counter()
        res_d = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=disc_lists, method="RK4", max_dt=MAX_DT)
        t_solve_d = time.perf_counter() - t2f
        yd = np.array([r.y[-1] for r in res_d])
        out["end_to_end_solve"] = {
            "what": f"Solver.solve of {b_loc} instances x 1000 RK4 steps (list mode -> one batched device solve), "
                    "pulses as DiscreteSignal(dt=0.05) + carrier, coefficient table evaluated on the device; host "
                    "work, PCIe and result unpacking included",
            "model_build_s": round(t_model, 2), "solve_s": round(t_solve_d, 2),
            "rhs_evals_per_s_end_to_end": round(b_loc * 4000 / t_solve_d, 1),
            "max_norm_deviation": float(np.max(np.abs(np.linalg.norm(yd, axis=1) - 1.0)))}
        if args.full_solve:
            t1f = time.perf_counter()
            res = solver.solve(t_span=cfg["t_span"], y0=cfg["y0"], signals=sig_lists, method="RK4", max_dt=MAX_DT)
            t_solve = time.perf_counter() - t1f
            yf = np.array([r.y[-1] for r in res])
            out["end_to_end_solve_host_table"] = {
                "what": "same sweep with Python-callable Gaussian envelopes: coefficient table evaluated on the host",
                "solve_s": round(t_solve, 2), "rhs_evals_per_s_end_to_end": round(b_loc * 4000 / t_solve, 1),
                "max_norm_deviation": float(np.max(np.abs(np.linalg.norm(yf, axis=1) - 1.0)))}
    # the driver keeps 8 018 characters of stdout: the LAST line is the compact object (contract fields, roofline, cpu_baseline,
    # one {value, frac} per other configuration); everything above goes to bench_detail.json and stderr
    emit(out, json_out, ROOT, rank)
    D.close()


if __name__ == "__main__":
    main()
