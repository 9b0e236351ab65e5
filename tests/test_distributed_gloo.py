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
