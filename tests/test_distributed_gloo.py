"""World-size-2 CPU (gloo) test of the multi-GPU sweep plumbing: shard bounds, rendezvous from the
torchrun environment, and the result all-gather.  The RCCL broadcast of the device stack itself
needs GPUs and is exercised by bench.py --gpus N on the GPU box."""
import os
import socket
import subprocess
import sys
import textwrap

from conftest import ROOT

WORKER = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.environ["MIDYN_ROOT"])
    import numpy as np
    import torch.distributed as dist
    from qiskit_dynamics_amd.distributed import (init_process_group_from_env, shard_bounds,
                                                 gather_sweep_results)
    rank, world = init_process_group_from_env(backend="gloo")
    assert world == 2
    B = 7
    lo, hi = shard_bounds(B, rank, world)
    # each rank "solves" its shard: result for instance b is a deterministic function of b
    local = np.array([[b + 1j * (b * b), -b + 0.5j] for b in range(lo, hi)], dtype=np.complex128)
    full = gather_sweep_results(local, B)
    expect = np.array([[b + 1j * (b * b), -b + 0.5j] for b in range(B)], dtype=np.complex128)
    assert full.shape == expect.shape and np.array_equal(full, expect), (rank, full)
    # solve_sweep: shards of uneven size, shared and per-instance y0, fewer instances than ranks
    from scipy.integrate._ivp.ivp import OdeResult
    from qiskit_dynamics_amd.distributed import solve_sweep

    class FakeSolver:
        # Stands in for Solver.solve (which needs a GPU): the 'solution' encodes which signals and
        # which y0 it was given, and counts the calls (one batched solve per rank).
        calls = 0

        def solve(self, t_span, y0, signals, **kw):
            FakeSolver.calls += 1
            assert kw == {"method": "RK4", "max_dt": 0.1}
            out = []
            for i, sig in enumerate(signals):
                y = y0[i] if isinstance(y0, list) else y0
                out.append(OdeResult(t=np.array(t_span, dtype=float),
                                     y=np.array([y, y * sig[0] + 1j * sig[1]], dtype=complex)))
            return out

    for n_inst in (5, 2, 1):
        sigs = [[float(b + 1), float(10 * b)] for b in range(n_inst)]
        y_shared = np.array([1.0 + 0j, 2.0])
        FakeSolver.calls = 0
        res = solve_sweep(FakeSolver(), [0.0, 1.0], y_shared, sigs, method="RK4", max_dt=0.1)
        lo_, hi_ = shard_bounds(n_inst, rank, world)
        assert FakeSolver.calls == (1 if hi_ > lo_ else 0)
        assert len(res) == n_inst
        for b in range(n_inst):
            assert np.array_equal(res[b].t, [0.0, 1.0])
            assert np.array_equal(res[b].y[1], y_shared * (b + 1) + 10j * b), (rank, b, res[b].y)
        y_list = [np.array([b + 0.5j, -b]) for b in range(n_inst)]
        res = solve_sweep(FakeSolver(), [0.0, 1.0], y_list, sigs, method="RK4", max_dt=0.1)
        for b in range(n_inst):
            assert np.array_equal(res[b].y[0], y_list[b])
            assert np.array_equal(res[b].y[1], y_list[b] * (b + 1) + 10j * b)
        # gather="root": rank 1 holds the full list, rank 0 gets None; gather="none": the own shard and its offset
        res = solve_sweep(FakeSolver(), [0.0, 1.0], y_list, sigs, gather="root", root=1, method="RK4", max_dt=0.1)
        if rank == 1:
            assert len(res) == n_inst
            for b in range(n_inst):
                assert np.array_equal(res[b].y[1], y_list[b] * (b + 1) + 10j * b)
        else:
            assert res is None
        lo_n, own = solve_sweep(FakeSolver(), [0.0, 1.0], y_list, sigs, gather="none", method="RK4", max_dt=0.1)
        assert lo_n == lo_ and len(own) == hi_ - lo_
        for i, r_ in enumerate(own):
            assert np.array_equal(r_.y[1], y_list[lo_ + i] * (lo_ + i + 1) + 10j * (lo_ + i))
    # max-over-ranks timing reduction used by bench.py
    import torch
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert t.item() == 2.0
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_shard_and_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MIDYN_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank} failed:\n{out}"
        assert f"rank {rank} ok" in out


def test_bench_spawns_its_own_ranks_and_refuses_mismatched_worlds():
    """`python bench.py --gpus N` without a launcher spawns N ranks itself (one process per GPU); here with the
    CPU stand-in for the device solve (MIDYN_BENCH_STUB: gloo, no GPU) -- the launcher, the rendezvous, the
    strong / weak shard arithmetic, the barriers and the MAX reduction are the production code.  It must never
    print a line for fewer ranks than asked for."""
    import json

    bench = os.path.join(ROOT, "bench.py")
    env = dict(os.environ, MIDYN_BENCH_STUB="1")
    for k_ in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k_, None)
    p = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                       # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong"
    assert out["instances_total"] == 4096 and out["shard_rank0"] == [0, 2048]   # BASELINE configs[2]: 4096 in TOTAL
    p = subprocess.run([sys.executable, bench, "--gpus", "2", "--weak", "--batch", "100"], env=env, capture_output=True,
                       text=True, timeout=240)
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert out["scaling"] == "weak" and out["instances_total"] == 200 and out["shard_rank0"] == [0, 100]
    # N = 1 and N = 8 (the driver's scaling run launches 1 / 2 / 4 / 8): one line each, LAST on stdout, far below the 8 018
    # characters of stdout the driver keeps (the real line is covered in test_host_logic.py)
    for n_ranks in (1, 8):
        p = subprocess.run([sys.executable, bench, "--gpus", str(n_ranks)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout + p.stderr
        last = p.stdout.strip().splitlines()[-1]
        assert len(last) < 6144
        out = json.loads(last)
        assert out["n_gpus"] == n_ranks and out["instances_total"] == 4096 and out["shard_rank0"] == [0, 4096 // n_ranks]
    # a launcher-provided world that disagrees with --gpus is an error, not a smaller job
    env_bad = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, bench, "--gpus", "8"], env=env_bad, capture_output=True, text=True, timeout=60)
    assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    # without the stub and without GPUs, --gpus 8 refuses (no n_gpus = 1 line)
    import torch

    if not torch.cuda.is_available():
        env_real = {k_: v for k_, v in env.items() if k_ != "MIDYN_BENCH_STUB"}
        p = subprocess.run([sys.executable, bench, "--gpus", "8"], env=env_real, capture_output=True, text=True, timeout=300)
        assert p.returncode != 0 and not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_abi_broadcast_probe_child_is_contained(monkeypatch):
    """distributed.abi_broadcast_probe without a GPU: a child that hangs is killed by PID after the time limit, a
    child that cannot run the C-ABI calls (no HIP device here) exits non-zero -- both reported as (False, reason),
    which sends bench.py to the torch.distributed route."""
    from qiskit_dynamics_amd.distributed import abi_broadcast_probe

    monkeypatch.setenv("MIDYN_PROBE_HANG", "1")
    ok, msg = abi_broadcast_probe(0, 2, 0, b"\0" * 128, timeout_s=2)
    assert not ok and "killed" in msg
    monkeypatch.delenv("MIDYN_PROBE_HANG")
    ok, msg = abi_broadcast_probe(1, 2, 0, b"\0" * 128, timeout_s=120)
    assert not ok and "exited" in msg
