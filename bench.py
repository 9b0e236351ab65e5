#!/usr/bin/env python
"""bench.py -- RHS evals/s of the 10-qubit (dim 1024) Schrodinger sweep on N MI355X.

Workload (BASELINE.json metric / configs[2], SURVEY.md 8(d) cfg 3): chain Hamiltonian, n = 1024,
k = 8 drive operators + static operator, rotating frame = H_d (dense frame-basis operators), RK4 with
max_dt = 0.005 on t in [0, 5] (1000 steps, 4 RHS evaluations each), 4096 signal instances per GPU
(weak scaling: every rank integrates its own 4096-instance shard; the only collective is one RCCL
broadcast of the packed operator stack at setup).  A "step" is one RK4 step of the whole per-GPU
batch = 4 batched RHS evaluations = 4 * 4096 instance-evaluations.

Timed region: inputs (operator stack, coefficient table, states) are resident in HBM; K steps are
enqueued on the library's HIP stream and bracketed by barrier + device synchronisation; the max over
ranks is taken.  One JSON line is printed by rank 0.

Extra objects on the same line: `roofline` (dominant kernel = the fp64-MFMA batched RHS contraction;
average launch duration measured with HIP events on the stream the kernel runs on),
`roofline_single_trajectory` (cfg 2: the HBM-bound streaming kernel, measured the same way) and
`cpu_baseline` (the NumPy oracle on the host cores, rank 0, N=1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_QUBITS = 10
N_DRIVES = 8
T_FINAL = 5.0
MAX_DT = 0.005
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X vendor FP64 matrix peak (SURVEY.md 8(d) / BASELINE.md 3)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


def measured_traffic(kernel_prefix):
    """HBM bytes per dispatch of the newest committed rocprofv3 PMC summary (profiles/*.traffic.json),
    or None.  bench.py cannot collect PMC counters itself; the profile run is tools/profile_round.sh."""
    import glob

    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.traffic.json"))):
        try:
            data = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, t in data.get("kernels", {}).items():
            if name.startswith(kernel_prefix) and "fetch_bytes" in t:
                best = {"bytes": round(t["fetch_bytes"] + t.get("write_bytes", 0.0)), "kernel": name,
                        "source": os.path.basename(path)}
    return best


def build_frame_basis_stack(cfg):
    """Host model build (a3/a4): -iH, eigh of the frame, U^dagger . U.  Uses the product classes'
    own code path (RotatingFrame) so the stack is exactly what HamiltonianModel uploads."""
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame

    frame = RotatingFrame(cfg["h_d"])
    static = frame.operator_into_frame_basis(-1j * cfg["h_d"]) - np.diag(frame.frame_diag)
    ops = frame.operator_into_frame_basis(-1j * cfg["ops"])
    return ops, static, frame.frame_diag_imag


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=4096, help="sweep instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single", action="store_true", help="skip the cfg-2 single-trajectory leg")
    ap.add_argument("--dense", action="store_true",
                    help="disable exact-zero plane skipping (time the general dense-complex path)")
    ap.add_argument("--no-end-to-end", action="store_true",
                    help="skip the end-to-end Solver.solve of the whole sweep (about 12 s)")
    ap.add_argument("--full-solve", action="store_true",
                    help="also time the end-to-end solve with Python-callable envelopes (host-evaluated coefficient table)")
    ap.add_argument("--complex-3m", action="store_true", help="dense complex products with 3 real MFMAs (A/B testing)")
    ap.add_argument("--plane-kernel", action="store_true", help="A/B: planar two-tiles-per-barrier kernel (opt-in)")
    ap.add_argument("--ablate", type=int, default=0, help="profiling only: kernel ablation bits (results wrong)")
    ap.add_argument("--force-tile", type=int, default=0, help="0 auto | 64 | 128 | 12864 (kernel A/B testing)")
    args = ap.parse_args()

    import qiskit_dynamics_amd as qd
    from qiskit_dynamics_amd import workloads
    from qiskit_dynamics_amd.distributed import broadcast_stack, init_process_group_from_env
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    use_dist = world > 1 or bool(os.environ.get("MIDYN_BENCH_FORCE_DIST"))  # the latter: 1-GPU test of the RCCL path
    if use_dist:
        import torch  # BEFORE the first libmidyn call: the library then binds to torch's HIP runtime
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        init_process_group_from_env(backend="nccl")
    ctx = qd.default_context(local_rank)
    if args.dense:
        ctx.set_option("skip_zero_planes", 0)
    if args.force_tile:
        ctx.set_option("force_tile", args.force_tile)
    if args.ablate:
        ctx.set_option("ablate", args.ablate)
    if args.complex_3m:
        ctx.set_option("complex_3m", 1)
    if args.plane_kernel:
        ctx.set_option("plane_kernel", 1)

    cfg = workloads.schrodinger_config(N_QUBITS, N_DRIVES, T_FINAL, MAX_DT)
    n = 2**N_QUBITS
    k = N_DRIVES
    t_setup = time.time()
    if use_dist:
        ops = static = frame_im = None
        if rank == 0:
            ops, static, frame_im = build_frame_basis_stack(cfg)
        stack, _keep = broadcast_stack(ctx, ops, static, frame_im, n, k, src=0)
    else:
        ops, static, frame_im = build_frame_basis_stack(cfg)
        stack = qd.Stack(ctx, ops, static, frame_im)
    setup_s = time.time() - t_setup

    # schedule of the real 1000-step solve; the bench runs its first (warmup + steps) steps
    sched = FixedStepSchedule(cfg["t_span"], None, MAX_DT, _rk4_points)
    total = args.warmup + args.steps
    if total > len(sched.step_h):
        raise SystemExit(f"warmup+steps must be <= {len(sched.step_h)}")
    rows = sched.step_rows[:total]
    n_rows = int(rows.max()) + 1
    times = sched.times[:n_rows]
    b_loc = args.batch
    inst0 = rank * b_loc
    amps = np.empty((b_loc, k))
    phs = np.empty((b_loc, k))
    for b in range(b_loc):
        amps[b], phs[b] = workloads.sweep_parameters(inst0 + b, k)
    table = workloads.gaussian_coefficient_table(times, amps, phs, cfg["carrier"], T_FINAL)
    y0 = cfg["y0"].reshape(-1, 1)
    plan = qd.Rk4Plan(stack, times, table, rows, sched.step_h[:total], y0, b_loc, True)

    def sync_all():
        ctx.synchronize()
        if dist is not None:
            import torch

            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    plan.run(0, args.warmup)
    sync_all()
    t0 = time.perf_counter()
    plan.run(args.warmup, total)
    ctx.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        import torch

        torch.cuda.synchronize()
        dist.barrier()
        tt = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    final = plan.fetch()[:, :, 0]
    norm_dev = float(np.max(np.abs(np.linalg.norm(final, axis=1) - 1.0)))

    evals = world * b_loc * 4 * args.steps
    value = evals / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel: HIP-event timing of every launch on the ctx stream ----
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    prof_steps = min(args.steps, 10)
    plan.run(total - prof_steps, total)   # re-runs the last steps (state is re-phased automatically)
    ctx.synchronize()
    cnt = ctx.counters("rhs_gemm")
    ctx.set_option("profile", 0)
    roofline = None
    if cnt["launches"] > 0:
        avg_ms = cnt["ms"] / cnt["launches"]
        flops_per_launch = (4 * k + 10) * n * n * b_loc          # useful flops, SURVEY 8(d) cfg 3
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12
        n_act = stack.n_active_segments
        act_modes = [m for m in stack.segment_modes if m != 3]
        stack_um = act_modes[0] if act_modes and all(m == act_modes[0] for m in act_modes) else 3
        modes = stack.segment_modes if not args.dense else [0] * stack.n_segments
        executed = sum((6 if m == 0 else 4) for m in modes if m != 3) * n * n * b_loc
        roofline = {
            "kernel": "zgemm_seg_kernel (batched RHS, fp64 MFMA 16x16x4)", "bound": "mfma",
            "achieved": round(achieved, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / FP64_MFMA_PEAK_TFLOPS, 4),
            "traffic": measured_traffic("zgemm_seg_kernel<64, 64, 2, 2, 16, 4," if (args.dense or stack_um == 0)
                                        else "zgemm_seg_kernel<128, 128, 2, 4, 16, %d," % stack_um),
            "avg_launch_ms": round(avg_ms, 4), "launches_timed": int(cnt["launches"]),
            "algorithmic_flops_per_launch": flops_per_launch,
            "executed_mfma_flops_per_launch": executed,
            "executed_tflops": round(executed / (avg_ms * 1e-3) / 1e12, 3),
            "segment_plane_modes": modes,
            "active_segments": n_act, "zero_plane_skipping": not args.dense,
            "note": "achieved = useful flops (4k+10)n^2 per instance-eval (SURVEY 8(d)); the operators of this "
                    "model are purely imaginary in the frame basis, so exact-zero plane skipping executes 4 instead "
                    "of 8 real flops per complex MAC; see dense_complex for the general path",
        }
    # ---- the general dense-complex path on the same inputs (no exact-zero plane skipping) -------
    dense = None
    if not args.dense and roofline:
        ctx.set_option("skip_zero_planes", 0)
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        plan.run(total - min(args.steps, 5), total)
        ctx.synchronize()
        cd = ctx.counters("rhs_gemm")
        ctx.set_option("profile", 0)
        ctx.set_option("skip_zero_planes", 1)
        avg_d = cd["ms"] / max(cd["launches"], 1)
        ex_d = 6 * stack.n_segments * n * n * b_loc   # 3M: 3 real MFMA products per complex product
        dense = {"avg_launch_ms": round(avg_d, 4), "rhs_evals_per_s": round(b_loc / (avg_d * 1e-3), 1),
                 "useful_tflops": round(flops_per_launch / (avg_d * 1e-3) / 1e12, 3),
                 "executed_tflops": round(ex_d / (avg_d * 1e-3) / 1e12, 3),
                 "frac_of_peak_executed": round(ex_d / (avg_d * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS, 4),
                 "scheme": "3M complex multiplication (3 real fp64 MFMAs per complex product), 64x64 tiles"}
    plan.close()
    measured_peaks = None
    if rank == 0:
        measured_peaks = {"mfma_f64_tflops": round(ctx.microbench("mfma_f64"), 1),
                          "hbm_read_gbs": round(ctx.microbench("hbm_read"), 0),
                          "mall_read_gbs": round(ctx.microbench("mall_read"), 0)}

    out = {
        "metric": "RHS evals/sec, 10-qubit Schrodinger (dim 1024), 4096-param batch", "value": round(value, 1),
        "unit": "RHS evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64 (complex128)", "data": "synthetic",
        "config": {"workload": "cfg3: 10-qubit chain, n=1024, k=8 drives + static, rotating frame H_d, "
                               "RK4 max_dt=0.005, sweep of %d instances per GPU" % b_loc,
                   "instances_per_gpu": b_loc, "global_instances": b_loc * world,
                   "rhs_evals_per_step": 4 * b_loc * world, "parallelism": f"sweep-shard x{world}"},
        "solve_wall_clock_s": round(ms_per_step * len(sched.step_h) / 1e3, 3),
        "solve_wall_clock_note": "1000 RK4 steps = 1000 x ms_per_step (fixed step, identical work per step)",
        "setup_s": round(setup_s, 2), "max_norm_deviation": norm_dev,
    }
    if roofline:
        out["roofline"] = roofline
    if dense:
        out["dense_complex"] = dense
    if measured_peaks:
        out["measured_peaks"] = measured_peaks

    # ---- cfg 2: single trajectory, HBM-bound streaming kernel (rank 0 only) ---------------------
    if rank == 0 and not args.no_single:
        s_total = 64
        rows1 = sched.step_rows[:s_total]
        nr1 = int(rows1.max()) + 1
        table1 = workloads.gaussian_coefficient_table(sched.times[:nr1], amps[:1], phs[:1], cfg["carrier"], T_FINAL)
        p1 = qd.Rk4Plan(stack, sched.times[:nr1], table1, rows1, sched.step_h[:s_total], y0, 1, True)
        p1.run(0, 8)
        ctx.synchronize()
        t0 = time.perf_counter()
        p1.run(8, s_total)
        ctx.synchronize()
        el1 = time.perf_counter() - t0
        ctx.reset_counters()
        ctx.set_option("profile", 1)
        p1.run(8, s_total)
        ctx.synchronize()
        c1 = ctx.counters("rhs_stream")
        ctx.set_option("profile", 0)
        p1.close()
        nseg = stack.n_segments
        bytes_per_launch = 16 * nseg * n * n + 32 * n                 # SURVEY 8(d) cfg 2: 151.03 MB
        avg_ms1 = c1["ms"] / max(c1["launches"], 1)
        gbs = bytes_per_launch / (avg_ms1 * 1e-3) / 1e9
        # single-plane stack (every operator purely real or purely imaginary) and plane skipping on:
        # the kernel streams only the non-zero planes, 8 B per operator element instead of 16 B
        planar = (not args.dense) and all(m in (1, 2, 3) for m in stack.segment_modes)
        n_act = sum(1 for m in stack.segment_modes if m != 3)
        executed_bytes = (8 * n_act * n * n + 32 * n) if planar else bytes_per_launch
        gbs_exec = executed_bytes / (avg_ms1 * 1e-3) / 1e9
        kname = "rhs_stream_plane_kernel<2, 3>" if planar else "rhs_stream_kernel<4, 3>"
        out["single_trajectory"] = {
            "workload": "cfg2: same model, 1 trajectory, RK4", "rhs_evals_per_s": round(4 * (s_total - 8) / el1, 1),
            "ms_per_step": round(el1 / (s_total - 8) * 1e3, 4)}
        out["roofline_single_trajectory"] = {
            "kernel": kname.split("<")[0], "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": measured_traffic(kname),
            "avg_launch_ms": round(avg_ms1, 5), "launches_timed": int(c1["launches"]),
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "executed_bytes_per_launch": executed_bytes, "executed_gbs": round(gbs_exec, 1),
            "executed_frac": round(gbs_exec / HBM_PEAK_GBS, 4),
            "note": "achieved = SURVEY 8(d) algorithmic bytes (151 MB, complex128 stack) / launch time; the "
                    "operators of this model are purely imaginary, so the kernel streams only their non-zero "
                    "planes (executed_bytes, exact same results); the planes fit the 256 MB Infinity Cache"}

    # ---- CPU baseline: the NumPy oracle on this host, bounded sample (rank 0, N=1 only) ---------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import dynamics_oracle as orc

        from threadpoolctl import threadpool_info, threadpool_limits

        a_d, a = static, ops
        d = 1j * frame_im
        best = None
        for threads in sorted({8, 32, os.cpu_count() or 8}):      # short probe: which BLAS width is fastest here
            if threads > (os.cpu_count() or 8):
                continue
            with threadpool_limits(limits=threads):
                t0c = time.perf_counter()

                def rhs(t, y):
                    c = workloads.gaussian_coefficient_table(np.array([t]), amps[0], phs[0], cfg["carrier"], T_FINAL)[0]
                    return orc.generator_rhs(a_d, a, c, d, None, t, y)

                orc.rk4_solve(rhs, [0.0, 10 * MAX_DT], cfg["y0"], MAX_DT)
                rate = 40 / (time.perf_counter() - t0c)
            if best is None or rate > best[0]:
                best = (rate, threads)
        threads = best[1]
        n_inst = 4
        n_steps = int(min(200, max(20, best[0] * 15 / (4 * n_inst))))   # ~15 s of CPU work
        with threadpool_limits(limits=threads):
            t0c = time.perf_counter()
            for b in range(n_inst):
                def rhs(t, y, b=b):
                    c = workloads.gaussian_coefficient_table(np.array([t]), amps[b], phs[b], cfg["carrier"], T_FINAL)[0]
                    return orc.generator_rhs(a_d, a, c, d, None, t, y)

                orc.rk4_solve(rhs, [0.0, n_steps * MAX_DT], cfg["y0"], MAX_DT)
            cpu_s = time.perf_counter() - t0c
        best = (n_inst * n_steps * 4 / cpu_s, threads, cpu_s)
        out["cpu_baseline"] = {
            "value": round(best[0], 1), "unit": "RHS evals/s", "cores": best[1], "kind": "port",
            "sample": f"{n_inst} instances x {n_steps} RK4 steps ({n_inst * n_steps * 4} RHS evals) of the same "
                      f"model with the NumPy oracle (tensordot + matvec); best of BLAS thread counts 8/32/all on a "
                      f"{os.cpu_count()}-CPU host: {best[1]} threads, {best[2]:.1f} s",
            "host": {"cpu_count": os.cpu_count(), "numpy": np.__version__,
                     "blas": [f"{i.get('internal_api')} {i.get('version')} ({i.get('threading_layer') or i.get('user_api')})"
                              for i in threadpool_info()],
                     "OPENBLAS_NUM_THREADS": os.environ.get("OPENBLAS_NUM_THREADS")}}
    # ---- optional: the complete cfg-3 solve through the public Solver API (host work included) ----
    # ---- the complete cfg-3 solve through the public Solver API: model build, signal evaluation, PCIe
    #      and result unpacking included (rank 0, N=1; --full-solve adds the host-table variant) --------
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
    # ---- informational (NOT the BASELINE config): the same sweep set up in the diagonal frame diag(H_d) --
    #      same physics out of the frame, but the operators stay block sparse and the contraction runs on
    #      the work-list kernels.  Same table, same steps, timed like `value`.
    if rank == 0 and world == 1 and not args.no_end_to_end and not args.dense:
        try:
            from qiskit_dynamics_amd.rotating_frame import RotatingFrame

            fr = RotatingFrame(np.diag(cfg["h_d"]).real.copy())
            stack_d = qd.Stack(ctx, -1j * cfg["ops"], -1j * cfg["h_d"] - np.diag(fr.frame_diag), fr.frame_diag_imag)
            plan_d = qd.Rk4Plan(stack_d, times, table, rows, sched.step_h[:total], y0, b_loc, True)
            plan_d.run(0, args.warmup)
            ctx.synchronize()
            t0d = time.perf_counter()
            plan_d.run(args.warmup, total)
            ctx.synchronize()
            dtd = time.perf_counter() - t0d
            fin_d = plan_d.fetch()[:, :, 0]
            plan_d.close()
            out["diagonal_frame_variant"] = {
                "what": "same model, sweep and steps with rotating_frame=diag(H_d) (NOT the BASELINE config, which "
                        "rotates into the eigenbasis of H_d): block-sparse operators, work-list kernels",
                "rhs_evals_per_s": round(b_loc * 4 * args.steps / dtd, 1), "ms_per_step": round(dtd / args.steps * 1e3, 4),
                "max_norm_deviation": float(np.max(np.abs(np.linalg.norm(fin_d, axis=1) - 1.0)))}
            del stack_d
        except Exception as exc:  # pylint: disable=broad-except
            out["diagonal_frame_variant"] = {"error": repr(exc)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
