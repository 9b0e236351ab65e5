"""Shared pieces of bench.py's legs: constants of the BASELINE configurations and of the rooflines, stack builders, the sweep's
coefficient table, the profile pass (HIP-event counters of the library) and the PMC traffic look-up."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

N_QUBITS = 10
N_DRIVES = 8
T_FINAL = 5.0
MAX_DT = 0.005
SWEEP = 4096                   # BASELINE.json: 4096-parameter batch
CFG5_SWEEP = 1024              # BASELINE.json configs[4]: 1024-parameter sweep
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X vendor FP64 matrix peak (SURVEY.md 8(d) / BASELINE.md 3)
LDS_PEAK_GBS = 256.0 * 256 * 2.4      # ds_read_b64/b128: 256 B per clock and CU (MI355X_MICROARCH.md, LDS), 256 CUs, 2.4 GHz
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)


# -----------------------------------------------------------------------------------------------------------------
# helpers
# -----------------------------------------------------------------------------------------------------------------
def measured_traffic(kernel_prefix):
    """(HBM bytes per dispatch, source) of the newest committed rocprofv3 PMC summary (profiles/*.traffic.json:
    FETCH_SIZE x2 per the gfx950 note + WRITE_SIZE, separate --pmc passes), or (None, None).  bench.py cannot collect
    PMC counters itself; the profile run is tools/profile_round.sh."""
    import glob

    best = (None, None)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.traffic.json"))):
        try:
            data = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, t in data.get("kernels", {}).items():
            if name.startswith(kernel_prefix) and "fetch_bytes" in t:
                best = (round(t["fetch_bytes"] + t.get("write_bytes", 0.0)), f"{os.path.basename(path)}: {name}")
    return best


def build_frame_basis_stack(cfg):
    """Host model build (a3/a4) WITHOUT grouping symmetry sectors: -iH, eigh of the frame, U^dagger . U in the
    reference's ascending-eigenvalue order (the exact zeros of a symmetric model are scattered: dense kernels).
    Used by tools/ for dense-kernel measurements."""
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame

    frame = RotatingFrame(cfg["h_d"])
    static = frame.operator_into_frame_basis(-1j * cfg["h_d"]) - np.diag(frame.frame_diag)
    ops = frame.operator_into_frame_basis(-1j * cfg["ops"])
    return ops, static, frame.frame_diag_imag


def build_model_stack(cfg):
    """The stack exactly as HamiltonianModel uploads it (models.py): frame-basis vectors grouped by the symmetry
    sectors of the frame operator (rotating_frame._eigh_by_sectors), so that exactly-zero operator blocks are
    contiguous.  Returns (ops, static, frame_im, perm): internal position i holds the reference's index perm[i]."""
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame

    frame = RotatingFrame(cfg["h_d"])
    static = frame.generator_minus_frame_in_basis(-1j * cfg["h_d"])     # U^+ (G - F) U: exactly zero here (frame = H_d)
    ops = frame.operator_into_frame_basis(-1j * cfg["ops"])
    fim = frame.frame_diag_imag
    labels = frame.sector_labels
    if labels is None:
        return ops, static, fim, None
    perm = np.argsort(labels, kind="stable")
    take = lambda x: np.ascontiguousarray(np.take(np.take(x, perm, axis=-2), perm, axis=-1))  # noqa: E731
    return take(ops), take(static), np.ascontiguousarray(fim[perm]), perm


def build_diag_frame_stack(cfg):
    """cfg 5: diagonal rotating frame diag(H_d) (1-D frame, no eigh): operators stay in the computational basis."""
    from qiskit_dynamics_amd.rotating_frame import RotatingFrame

    fr = RotatingFrame(np.diag(cfg["h_d"]).real.copy())
    return -1j * cfg["ops"], -1j * cfg["h_d"] - np.diag(fr.frame_diag), fr.frame_diag_imag, None


def sweep_table(workloads, times, first, count, k, carrier, t_final):
    amps = np.empty((count, k))
    phs = np.empty((count, k))
    for b in range(count):
        amps[b], phs[b] = workloads.sweep_parameters(first + b, k)
    return workloads.gaussian_coefficient_table(times, amps, phs, carrier, t_final), amps, phs


# -----------------------------------------------------------------------------------------------------------------
# legs
# -----------------------------------------------------------------------------------------------------------------
def profile_pass(ctx, fn, classes):
    ctx.reset_counters()
    ctx.set_option("profile", 1)
    try:
        fn()
        ctx.synchronize()
        return {c: ctx.counters(c) for c in classes}
    finally:
        ctx.set_option("profile", 0)


ALL_CLASSES = ("rhs_stream", "rhs_gemm", "zgemm", "gen_eval", "elementwise", "rhs_blocks", "rhs_blocks_gemm", "rk4_resident")


# -----------------------------------------------------------------------------------------------------------------
def _mfma_roofline(kernel, flops, kernel_ms, note, **extra):
    tf = flops / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    return {"kernel": kernel, "bound": "mfma", "achieved": round(tf, 3), "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / FP64_MFMA_PEAK_TFLOPS, 4), "executed_flops": flops, "kernel_ms": round(kernel_ms, 4),
            "traffic": None, "note": note, **extra}


ZGEMM_NOTE = ("achieved = real flops the zgemm_seg_kernel launches EXECUTE (library counter flops:zgemm: M N K per launch x 6 "
              "with three real products per complex product (3M, the solver pipelines' mode) or 8 with four) / the HIP-event "
              "time of the same launches (counter class zgemm, profile on); elementwise passes (lincomb, norms) are listed "
              "beside it, not counted as flops")
