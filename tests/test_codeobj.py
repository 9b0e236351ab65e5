"""Register spills of the shipped kernels, read from the code-object metadata of libmidyn.so (no GPU needed).

A spilled kernel keeps part of its working set in scratch memory: round 2's ell_sweep_kernel<2,4,1024> ran its cfg 5 hot path
with 76 spilled registers (212 bytes of scratch per lane).  Every kernel the default routes can launch must have
`.vgpr_spill_count == 0` and no scratch; the opt-in A/B variants that are allowed to spill are listed by name.
"""
import os

import pytest

from conftest import ROOT

import codeobj

LIB = os.path.join(ROOT, "qiskit_dynamics_amd", "libmidyn.so")

# opt-in variants (never taken by a default route; ctx options in parentheses)
MAY_SPILL = (
    "ell_sweep_split_kernelILi2ELi2EE",      # two workgroups per instance at n = 4096 (ell_sweep_split >= 2)
)

# kernels of the default routes of the BASELINE configurations, by mangled-name fragment: they must exist (a rename must
# not silently empty this test) and must not spill
DEFAULT_ROUTE = (
    "zgemm_seg_kernelILi128ELi128ELi2ELi4ELi16ELi2ELi2ELb1EE",   # cfg 3 headline: sector work lists, imaginary-plane operators
    "zgemm_seg_kernelILi128ELi128ELi2ELi4ELi16ELi2ELi2ELb0EE",   # the same stack on the dense kernel
    "zgemm_seg_kernelILi64ELi64ELi2ELi2ELi16ELi4ELi2ELb0EE",     # dense complex operators (3M)
    "rk4_resident_kernelILi8ELi8ELb1EE",                         # cfg 2 single trajectory
    "rk4_resident_kernelILi2ELi8ELb0EE",
    "rk4_resident_kernelILi4ELi8ELb0EE",
    "ell_resident_kernelILi1ELi8EE",                             # cfg 4
    "ell_sweep_kernelILi2ELi4ELi1024ELi2EE",                     # cfg 5 (direct element form)
    "ell_sweep_kernelILi2ELi4ELi1024ELi1EE",
    "ell_sweep_kernelILi2ELi4ELi1024ELi0EE",
    "ell_sweep_kernelILi2ELi3ELi1024ELi0EE",
    "ell_sweep_kernelILi1ELi4ELi1024ELi2EE",
    "ell_sweep_rk4_kernelILi4ELi1024ELi2EE",
    "ell_sweep_rk4_kernelILi4ELi1024ELi0EE",
    "rhs_stream_plane_kernel",
    "rhs_blocks_kernel",
    "splitk_reduce_kernel",
)


@pytest.fixture(scope="module")
def kernels():
    if not os.path.exists(LIB):
        pytest.skip("libmidyn.so has not been built")
    return codeobj.library_kernels(LIB)


def test_code_object_is_gfx950_and_lists_the_kernels(kernels):
    assert len(kernels) > 200
    for frag in DEFAULT_ROUTE:
        assert any(frag in name for name in kernels), f"no kernel matches {frag}"


def test_no_kernel_of_a_default_route_spills_registers(kernels):
    spilled = {name: (k[".vgpr_spill_count"], k.get(".private_segment_fixed_size", 0)) for name, k in kernels.items()
               if k.get(".vgpr_spill_count", 0) or k.get(".private_segment_fixed_size", 0)}
    unexpected = {n: v for n, v in spilled.items() if not any(f in n for f in MAY_SPILL)}
    assert not unexpected, f"kernels with spilled registers / scratch (count, bytes per lane): {unexpected}"
    for frag in DEFAULT_ROUTE:
        for name, k in kernels.items():
            if frag in name:
                assert k.get(".vgpr_spill_count", 0) == 0 and k.get(".private_segment_fixed_size", 0) == 0, name


def test_register_budgets_of_the_one_workgroup_per_cu_kernels(kernels):
    """1024-thread workgroups get 128 registers per lane, 512-thread ones 256 (one workgroup per CU)."""
    for name, k in kernels.items():
        if "ell_sweep_kernelILi" in name and "ELi1024E" in name or "ell_sweep_rk4_kernel" in name and "ELi1024E" in name:
            assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 128, (name, k[".vgpr_count"])
        if "rk4_resident_kernelILi" in name and "ELi8EL" in name:
            assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 256, (name, k[".vgpr_count"])
