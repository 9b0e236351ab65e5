"""bench.py legs: the headline model in the diagonal frame (one-launch RK4 sweep) and the sweeps of small systems."""
import os
import time

import numpy as np

from .common import (ALL_CLASSES, CFG5_SWEEP, FP64_MFMA_PEAK_TFLOPS, HBM_PEAK_GBS, LDS_PEAK_GBS, MAX_DT, N_DRIVES, N_QUBITS, ROOT, SWEEP,  # noqa: F401
                     T_FINAL, ZGEMM_NOTE, _mfma_roofline, build_diag_frame_stack, build_frame_basis_stack, build_model_stack,
                     measured_traffic, profile_pass, sweep_table)


def leg_diag_frame_sweep(qd, ctx, workloads, instances=4096, steps=40):
    """The headline model set up in the DIAGONAL frame diag(H_d) instead of the full frame H_d (the same physics: results agree
    out of the frame; a choice the reference leaves to the user, models/rotating_frame.py): the operators stay in the computational
    basis -- ~20 non-zeros per row, every ELL slot one signed magnitude and one flip mask -- and the RK4 sweep is ONE launch of
    ell_sweep_rk4_kernel<1, 1024, 3> (no operator elements, csrc/midyn_flip.h / midyn_resident.h).  Kernel time from the library's
    HIP-event counters; the same kernel with 4-byte elements (option ell_sweep_flip = 0) and the work-list route beside it."""
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame
    from qiskit_dynamics_amd.solvers import FixedStepSchedule, _rk4_points

    cfg = workloads.schrodinger_config()
    frame = RotatingFrame(np.diag(cfg["h_d"]).real.copy())
    stack = qd.Stack(ctx, -1j * cfg["ops"], -1j * cfg["h_d"] - np.diag(frame.frame_diag), frame.frame_diag_imag)
    sched = FixedStepSchedule(cfg["t_span"], None, 0.005, _rk4_points)
    rows = sched.step_rows[:steps]
    nr = int(rows.max()) + 1
    k = cfg["ops"].shape[0]
    pars = [workloads.sweep_parameters(b, k) for b in range(instances)]
    table = workloads.gaussian_coefficient_table(sched.times[:nr], np.array([p[0] for p in pars]), np.array([p[1] for p in pars]),
                                                 cfg["carrier"], 5.0)
    y0 = cfg["y0"].reshape(-1, 1)
    save = np.full(steps, -1, dtype=np.int32)
    save[-1] = 1

    def run():
        return stack.rk4_solve(sched.times[:nr], table, rows, sched.step_h[:steps], save, 2, y0, instances, True)

    best, out_y = {}, {}
    for rnd in range(3):                       # interleaved, minimum per element form
        for flag in (1, 0):
            with ctx.options(ell_sweep_flip=flag, profile=1):
                ctx.reset_counters()
                ys = run()
                ms = ctx.counters("rk4_resident")["ms"]
                form = int(ctx.counters("sweep_split")["ms"])
            best[form] = min(best.get(form, 1e9), ms)
            out_y[form] = ys
    with ctx.options(ell_sweep=0):
        t0 = time.perf_counter()
        ref = run()
        wall_lists = time.perf_counter() - t0
    form = max(best)
    evals = 4.0 * instances * steps
    out = {"workload": "the headline model (10 qubits, n = 1024, 8 drives) in the diagonal frame diag(H_d), %d instances x %d RK4 "
                       "steps in ONE launch, inputs resident" % (instances, steps),
           "kernel": "ell_sweep_rk4_kernel<1, 1024, %d>" % form, "element_form": form,
           "rhs_evals_per_s_in_the_kernel": round(evals / (best[form] * 1e-3)),
           "kernel_ms_per_step": round(best[form] / steps, 4),
           "with_4_byte_elements_rhs_evals_per_s": round(evals / (best[min(best)] * 1e-3)) if len(best) > 1 else None,
           "work_list_route_rhs_evals_per_s_wall": round(evals / wall_lists),
           "max_abs_difference_to_the_work_list_route": float(np.max(np.abs(out_y[form] - ref))),
           "max_abs_difference_between_the_element_forms": float(np.max(np.abs(out_y[form] - out_y[min(best)]))),
           "max_norm_deviation": float(np.max(np.abs(np.linalg.norm(out_y[form][:, -1, :, 0], axis=1) - 1.0))),
           "note": "NOT the headline number: `value` is measured in the full frame H_d of BASELINE's configuration (dense frame-basis "
                   "operators, MFMA combine + apply).  This key shows what the same physics costs when the user keeps the operators "
                   "sparse; parity: tests/test_gpu_resident.py (element forms, random flip masks) and tools/fuzz_solver.py --pauli"}
    return out


def leg_small_sweeps(qd, workloads, instances=4096, steps=200):
    """Sweeps of SMALL systems through the product Solver (list mode): chains of three-level transmons in the frame of their
    static Hamiltonian, DiscreteSignal pulses with carriers -- the sizes pulse-level simulations have.  us per RK4 stage over
    the device part of the solve on the one-launch kernel (csrc/midyn_combine_sweep.h) and with a launch per stage."""
    out = {}
    dt = 0.005
    t_final = dt * steps
    rng = np.random.default_rng(7)
    for levels, sites in ((3, 3), (3, 4)):
        h_d, ops, freqs = workloads.transmon_chain(levels, sites)
        n = h_d.shape[0]
        solver = qd.Solver(static_hamiltonian=h_d, hamiltonian_operators=ops, rotating_frame=h_d)
        ctx = solver.model._ctx
        n_smp = max(4, int(round(t_final / 0.05)))
        lists = [[qd.DiscreteSignal(t_final / n_smp, rng.uniform(0.2, 1.0) * np.hanning(n_smp + 2)[1:-1], carrier_freq=f,
                                    phase=rng.uniform(0, 2 * np.pi)) for f in freqs] for _ in range(instances)]
        y0 = np.zeros(n, dtype=complex)
        y0[0] = 1.0

        def best(reps=3):
            devs, calls = [], []
            for _ in range(reps + 1):           # (the first one builds layouts / warms up)
                t0 = time.perf_counter()
                res = solver.solve(t_span=[0.0, t_final], y0=y0, signals=lists, method="RK4", max_dt=dt)
                calls.append(time.perf_counter() - t0)
                devs.append(res[0].wall_s)
            return min(devs[1:]), min(calls[1:]), res

        dev1, call1, res = best()
        ctx.set_option("combine_sweep", 0)
        try:
            dev0, call0, ref = best()
        finally:
            ctx.set_option("combine_sweep", 1)
        # the same sweep with scipy_expm (Magnus order 1, same steps): the expm action, one launch / a launch per product
        e_steps = steps
        expm = {}
        for key, opt in (("one_launch", 1), ("launch_per_product", 0)):
            ctx.set_option("combine_sweep", opt)
            try:
                devs = []
                for _ in range(3):
                    r_e = solver.solve(t_span=[0.0, t_final], y0=y0, signals=lists, method="scipy_expm", max_dt=dt)
                    devs.append(r_e[0].wall_s)
            finally:
                ctx.set_option("combine_sweep", 1)
            expm["us_per_step_" + key] = round(min(devs[1:]) / e_steps * 1e6, 2)
        expm["max_abs_difference_to_the_rk4_result_midpoint_magnus_vs_rk4"] = float(max(np.max(np.abs(a_.y[-1] - b_.y[-1])) for a_, b_ in zip(res[::257], r_e[::257])))
        evals = instances * 4 * steps
        out[f"{sites}_transmons_n{n}"] = {
            "scipy_expm_magnus1": expm,
            "instances": instances, "steps": steps, "operators": len(ops),
            "us_per_stage_one_launch": round(dev1 / (4 * steps) * 1e6, 2),
            "us_per_stage_launch_per_stage": round(dev0 / (4 * steps) * 1e6, 2),
            "rhs_evals_per_s_device": round(evals / dev1, 1), "rhs_evals_per_s_whole_call": round(evals / call1, 1),
            "max_abs_difference_between_the_routes": float(max(np.max(np.abs(a_.y[-1] - b_.y[-1])) for a_, b_ in
                                                               zip(res[::257], ref[::257]))),
            "max_norm_deviation": float(max(abs(np.linalg.norm(r.y[-1]) - 1.0) for r in res[::257]))}
    out["note"] = ("whole RK4 / scipy_expm solve of the sweep in ONE launch, 16 instances per workgroup, state in registers, stage "
                   "input in LDS (combine_sweep_kernel); device part = midyn_rk4_solve / midyn_expm_solve incl. PCIe, plan set-up "
                   "and (expm) the host-side choice of the series of every step; DESIGN 4.16")
    return out
