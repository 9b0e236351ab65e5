"""bench.py legs: BASELINE cfg 4 (6-qubit vectorised Lindblad, scipy_expm) in both frames."""
import os
import time

import numpy as np

from .common import (ALL_CLASSES, CFG5_SWEEP, FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, LDS_PEAK_GBS, MAX_DT, N_DRIVES, N_QUBITS, ROOT, SWEEP,  # noqa: F401
                     T_FINAL, ZGEMM_NOTE, _mfma_roofline, build_diag_frame_stack, build_frame_basis_stack, build_model_stack,
                     measured_traffic, profile_pass, sweep_table)


def leg_cfg4_diag_frame(qd, ctx, workloads):
    """cfg 4's "second run" of SURVEY 8(d): the same Lindbladian in the diagonal rotating frame diag(H_d), the same 100
    steps through the product's default route (pinned to the oracle at this shape by
    tests/test_gpu_production_shapes.py::test_cfg4_default_route_all_steps_vs_oracle[diag_frame])."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points

    cfg = workloads.lindblad_config()
    frame = np.diag(cfg["h_d"]).real.copy()
    t0 = time.perf_counter()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], rotating_frame=frame, vectorized=True)
    build_s = time.perf_counter() - t0
    stack = solver.model.stack
    sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(1))
    table, _, _ = sweep_table(workloads, sched.times, 0, 1, 6, cfg["carrier"], cfg["t_final"])
    y0 = cfg["rho0"].flatten(order="F").reshape(-1, 1)

    def run():
        return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 1,
                                y0, 1, True)

    run()
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.timer_start()
    ys = run()
    dev_ms = ctx.timer_stop()
    wall = time.perf_counter() - t0
    cs = profile_pass(ctx, run, ALL_CLASSES)
    n_steps = len(sched.step_h)
    rho = ys[0, -1, :, 0].reshape(64, 64, order="F")
    out = {"workload": "cfg4, second run of SURVEY 8(d): the same model in the diagonal rotating frame diag(H_d)",
           "steps": n_steps, "solve_s": round(wall, 4), "ms_per_step": round(wall / n_steps * 1e3, 4),
           "stream_ms_per_step": round(dev_ms / n_steps, 4), "model_build_s": round(build_s, 2),
           "route": "ell_resident_kernel<1>: the whole solve in ONE launch" if cs["rk4_resident"]["launches"] == 1
                    else "one launch per product",
           "launches": {c: int(v["launches"]) for c, v in cs.items() if v["launches"]},
           "trace_deviation": float(abs(np.trace(rho) - 1.0)), "hermiticity": float(np.linalg.norm(rho - rho.conj().T))}
    del solver
    return out


def leg_cfg4(qd, ctx, workloads):
    """cfg 4: 6-qubit vectorised Lindbladian (N = 4096 superoperators built on the device), 4 static dissipators,
    scipy_expm magnus_order 1, max_dt 0.05, T = 5 -> 100 steps, one trajectory ("replicas only")."""
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _magnus_points

    cfg = workloads.lindblad_config()
    t0 = time.perf_counter()
    solver = qd.Solver(static_hamiltonian=cfg["h_d"], hamiltonian_operators=cfg["ops"],
                       static_dissipators=cfg["static_dissipators"], vectorized=True)
    build_s = time.perf_counter() - t0
    stack = solver.model.stack
    n_big = stack.n
    sched = FixedStepSchedule(cfg["t_span"], None, cfg["max_dt"], _magnus_points(1))
    table, _, _ = sweep_table(workloads, sched.times, 0, 1, 6, cfg["carrier"], cfg["t_final"])
    y0 = cfg["rho0"].flatten(order="F").reshape(-1, 1)

    def run():
        return stack.expm_solve(sched.times, table, sched.step_rows, sched.step_h, sched.step_save, sched.n_save, 1,
                                y0, 1, True)

    def measure(resident):
        ctx.set_option("resident_rk4", 1 if resident else 0)
        try:
            run()                           # warm (lazy block lists, norms, allocations)
            ctx.synchronize()
            t0_ = time.perf_counter()
            ctx.timer_start()
            ys_ = run()
            dev_ = ctx.timer_stop()
            wall_ = time.perf_counter() - t0_
            cs_ = profile_pass(ctx, run, ALL_CLASSES)
        finally:
            ctx.set_option("resident_rk4", 1)
        return ys_, dev_, wall_, cs_

    ys_res, dev_res, wall_res, cs_res = measure(True)       # the product's default route
    ys, dev_ms, wall, cs = measure(False)                   # one launch per product (work-list streaming kernel)
    took_resident = cs_res["rk4_resident"]["launches"] > 0
    rho = ys_res[0, -1, :, 0].reshape(64, 64, order="F")
    n_steps = len(sched.step_h)
    blk = stack.block_info()
    dom = max(cs, key=lambda c: cs[c]["ms"])
    launches = cs[dom]["launches"]
    avg_ms = cs[dom]["ms"] / max(launches, 1)
    k_h, s_sq = 6, 0
    products = launches / n_steps if dom == "rhs_blocks" else None
    out = {"workload": "cfg4: 6-qubit vectorised LindbladModel (N=4096 superoperator), 4 static dissipators, no frame, "
                       "scipy_expm magnus_order=1, max_dt=0.05, 100 steps, 1 trajectory",
           "steps": n_steps, "solve_s": round(wall_res, 4), "ms_per_step": round(wall_res / n_steps * 1e3, 4),
           "stream_ms_per_step": round(dev_res / n_steps, 4), "model_build_s": round(build_s, 2),
           "route": ("ell_resident_kernel<1>: the whole solve in ONE launch, operator elements in registers (one lane "
                     "per row), one exchange round per series term") if took_resident else "one launch per product",
           "us_per_product": round(dev_res / n_steps / products * 1e3, 3) if products else None,
           "products_per_step": round(products, 2) if products else None,
           "bound": "exchange latency (store -> poll across XCDs per term)" if took_resident else "launch latency",
           "trace_deviation": float(abs(np.trace(rho) - 1.0)),
           "hermiticity": float(np.linalg.norm(rho - rho.conj().T)),
           "max_abs_difference_between_the_routes": float(np.max(np.abs(ys_res - ys))),
           "per_launch_route": {
               "solve_s": round(wall, 4), "ms_per_step": round(wall / n_steps * 1e3, 4),
               "stream_ms_per_step": round(dev_ms / n_steps, 4),
               "launches_per_step": {c: round(v["launches"] / n_steps, 2) for c, v in cs.items() if v["launches"]},
               "kernel_ms_per_step": {c: round(v["ms"] / n_steps, 4) for c, v in cs.items() if v["launches"]}}}
    if dom == "rhs_blocks" and blk["state"] == 1:
        # one product G.v on the work lists: every listed 16x16 block (4 KiB) is read once, plus state in / out
        bytes_launch = blk["nonzero_blocks"] * 16 * 16 * 16 + 2 * 16 * n_big
        gbs = bytes_launch / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {
            "route": "per-launch route (option resident_rk4=0)",
            "kernel": "rhs_blocks_kernel<1> (expm action: one product G.v per launch over the non-zero 16x16 blocks)",
            "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": measured_traffic("rhs_blocks_kernel<1>")[0],
            "traffic_source": measured_traffic("rhs_blocks_kernel<1>")[1], "avg_launch_ms": round(avg_ms, 5),
            "launches_timed": int(launches), "executed_bytes_per_launch": bytes_launch,
            "nonzero_blocks": blk["nonzero_blocks"], "block_density": round(blk["block_density"], 5),
            "products_per_step": round(launches / n_steps, 2),
            "note": "executed bytes of the block work lists; the kernel is a latency chain of a few dozen blocks per "
                    "row group, not a bandwidth problem (DESIGN 4.12)",
            "dense_form_price": {
                "labelled": "SURVEY 8(d) cfg 4 prices the reference's dense algorithm, which is NOT executed here",
                "assembly_bytes_per_step": 16 * (k_h + 1) * n_big * n_big,
                "expm_flops_per_step": 8.0 * n_big**3 * (7.33 + s_sq),
                "mfma_ceiling_ms_per_step": round(8.0 * n_big**3 * 7.33 / (FP64_MFMA_PEAK_TFLOPS * 1e12) * 1e3, 1),
                "measured_ms_per_step": round(wall_res / n_steps * 1e3, 4)}}
    else:
        out["roofline"] = {"kernel": dom, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": None, "traffic": None, "avg_launch_ms": round(avg_ms, 5)}
    del solver
    return out
