"""Register spills of the shipped kernels, read from the code-object metadata of libmidyn.so (no GPU needed).

A spilled kernel keeps part of its working set in scratch memory: round 2's ell_sweep_kernel<2,4,1024> ran its cfg 5 hot path
with 76 spilled registers (212 bytes of scratch per lane).  The rule: EVERY kernel of the library has `.vgpr_spill_count == 0`
and no scratch -- all 128 combine_sweep_kernel variants included since round 5 (round 4 let the two-tile variants with three
or four plane groups park 6-32 registers outside their loops; they were the 64-bit addresses of (row, array) pairs and the
per-lane plane columns, hoisted out of the stage loop) -- except the opt-in A/B variants listed by name in MAY_SPILL.
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
    "ell_flip_duo_kernelILi2ELi2ELi1024EE",                      # cfg 5 shard (round 5): two workgroups per instance, no operator elements
    "ell_flip_duo_kernelILi1ELi2ELi1024EE",
    "ell_flip_duo_kernelILi2ELi1ELi512EE",
    "ell_sweep_duo_kernelILi2ELi2ELi1024ELi2EE",                 # ... with 4-byte elements (several flip masks in a slot)
    "ell_sweep_duo_kernelILi2ELi2ELi1024ELi1EE",
    "ell_sweep_duo_kernelILi1ELi2ELi1024ELi2EE",
    "ell_sweep_kernelILi2ELi4ELi1024ELi3EE",                     # cfg 5, more than 128 instances per GPU (no operator elements)
    "ell_sweep_kernelILi2ELi4ELi1024ELi2EE",                     # ... direct element form
    "ell_sweep_kernelILi2ELi4ELi1024ELi1EE",
    "ell_sweep_kernelILi2ELi4ELi1024ELi0EE",
    "ell_sweep_kernelILi1ELi4ELi1024ELi2EE",
    "ell_sweep_rk4_kernelILi4ELi1024ELi2EE",
    "ell_sweep_rk4_kernelILi4ELi1024ELi0EE",
    "rhs_combine_kernelILi0ELi2ELi0EE",                          # cfg 3 headline (round 4): combine + apply, 8 imaginary planes
    "rhs_combine_kernelILi2ELi0ELi0EE",
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
        if ("ell_sweep_kernelILi" in name or "ell_sweep_rk4_kernel" in name or "ell_sweep_duo_kernel" in name or
                "ell_flip_duo_kernel" in name) and "ELi1024E" in name:
            assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 128, (name, k[".vgpr_count"])
        if "rk4_resident_kernelILi" in name and "ELi8EL" in name:
            assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 256, (name, k[".vgpr_count"])


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="llvm-objdump of the ROCm image not found")
def test_slot_loops_of_the_flip_kernel_wait_for_lds_only(tmp_path):
    """ell_flip_duo_kernel<2, 2, 1024> (cfg 5 shard): a slot loop is readlanes, four LDS gathers, eight multiply-adds -- no
    operator element is loaded, and NO `s_waitcnt vmcnt` sits between the gathers of a slot and the loop's back edge (hipcc put a
    vmcnt(0) in front of the loops that follow the loads of the crossing operands, and one inside every slot loop when a path
    with unconsumed loads reached the next pass: the exchange's latency, exposed; csrc/midyn_flip.h)."""
    text = codeobj.disassembly(LIB, tmp_path, "/opt/rocm/lib/llvm/bin/llvm-objdump")
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if "ell_flip_duo_kernelILi2ELi2ELi1024EE" in l and l.rstrip().endswith(">:"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].rstrip().endswith(">:"))
    ops = [l.split("//")[0].split() for l in lines[start + 1:end]]
    ops = [o for o in ops if o]
    assert not any(o[0].startswith("scratch_") for o in ops)
    # slot loops: a run of four ds_read_b128 followed (within 40 instructions) by a backward s_cbranch_scc1
    loops = 0
    i = 0
    while i < len(ops):
        if ops[i][0] == "ds_read_b128":
            reads = [j for j in range(i, min(i + 12, len(ops))) if ops[j][0] == "ds_read_b128"]
            if len(reads) >= 4:
                tail = next((j for j in range(reads[3], min(reads[3] + 40, len(ops)))
                             if ops[j][0] in ("s_cbranch_scc1", "s_cbranch_scc0") and int(ops[j][1]) > 60000), None)
                if tail is not None:
                    body = ops[reads[0]:tail]
                    assert not any(o[0] == "s_waitcnt" and "vmcnt" in " ".join(o) for o in body), body
                    assert not any(o[0].startswith(("global_load", "buffer_load")) for o in body), body
                    assert sum(o[0] in ("v_fma_f64", "v_fmac_f64_e32") for o in body) == 8, body
                    loops += 1
                    i = tail
            i = max(i + 1, reads[-1] + 1) if len(reads) >= 4 and tail is None else i + 1
        else:
            i += 1
    assert loops >= 8, loops          # two planes x four parts of the local slots, two passes per term (+ order-1 tail)


def test_flip_form_kernels_declare_no_static_lds(kernels):
    """The kernels of element form 3 address their gathers as (own LDS address) ^ flip without adding the base of the dynamic LDS:
    that base must be 0, i.e. the kernel may not declare static LDS (they trap otherwise; csrc/midyn_flip.h, sweep_pass)."""
    found = 0
    for name, k in kernels.items():
        if "ell_flip_duo_kernel" in name or (("ell_sweep_kernelILi" in name or "ell_sweep_rk4_kernelILi" in name) and "ELi3EEEv" in name):
            assert k.get(".group_segment_fixed_size", 0) == 0, (name, k.get(".group_segment_fixed_size"))
            found += 1
    assert found >= 8 + 10 + 5, found       # 8 flip duo variants, 10 expm sweep variants, 5 RK4 sweep variants of form 3


# ---- the tile loop of the MFMA contraction: every VALU instruction in it takes matrix-pipe cycles -------------------------
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _tile_loop(symbol_fragment, tmp_path):
    """Instructions between the first and the last MFMA of the kernel whose mangled name contains the fragment."""
    import subprocess

    text = codeobj.disassembly(LIB, tmp_path, OBJDUMP)
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if symbol_fragment in l and l.rstrip().endswith(">:"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].rstrip().endswith(">:"))
    ops = [l.split("//")[0].split() for l in lines[start + 1:end]]
    ops = [o for o in ops if o]
    mf = [i for i, o in enumerate(ops) if o[0].startswith("v_mfma")]
    return ops[mf[0]:mf[-1] + 1]


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image not found")
@pytest.mark.parametrize("fragment, mfmas", [
    ("zgemm_seg_kernelILi128ELi128ELi2ELi4ELi16ELi2ELi2ELb1EE", 64),   # headline: work lists, single plane
    ("zgemm_seg_kernelILi128ELi128ELi2ELi4ELi16ELi2ELi2ELb0EE", 64),   # dense, single plane
    ("zgemm_seg_kernelILi128ELi128ELi2ELi4ELi16ELi0ELi2ELb0EE", 128),  # dense complex, 4M
])
def test_tile_loop_of_the_contraction_stays_lean(fragment, mfmas, tmp_path):
    """Round 3 (tools/mfma_bank_probe.hip): a VALU instruction in the tile loop costs 3-7 cycles of the matrix
    pipe whichever wave issues it; the loop went from 46 to 30 of them per tile (16 are the coefficient scalings).  Guards
    what the compiler can silently undo: LDS-DMA in scalar-base form with SGPR destinations (no v_readfirstlane, no 64-bit
    VALU address per DMA), fragment reads with immediate k-step offsets, no spills."""
    if not os.path.exists(LIB):
        pytest.skip("libmidyn.so has not been built")
    loop = _tile_loop(fragment, tmp_path)
    names = [o[0] for o in loop]
    assert sum(n.startswith("v_mfma_f64_16x16x4") for n in names) == mfmas
    valu = [n for n in names if n.startswith("v_") and not n.startswith("v_mfma")]
    scalings = sum(n == "v_mul_f64" for n in valu)
    assert 12 <= scalings <= 16            # (the loop is rotated: the first k-step's four precede the first MFMA)
    assert len(valu) - scalings <= 30, valu          # static count incl. the once-per-64-entries list refill
    dma = [o for o in loop if o[0].startswith("global_load_lds")]
    assert dma and all(any(t.startswith("s[") for t in o) for o in dma), dma      # scalar base + 32-bit lane offset
    reads = [o for o in loop if o[0] in ("ds_read_b128", "ds_read_b64")]
    assert not any(n.startswith("ds_read2") for n in names)      # (paired 8-byte reads collide in the banks)
    if mfmas == 64:      # single plane: the operator fragments are 8-byte reads of the plane that feeds MFMAs
        assert sum(o[0] == "ds_read_b64" for o in reads) >= 12
    # (the six reads of the next tile's first k-step sit behind the barrier, outside this window; within a tile all but
    # the first fragment of each operand carry their k-step / row block as an immediate)
    assert len(reads) >= 18 and sum("offset:" in " ".join(o) for o in reads) >= len(reads) - 2
    assert not any(n.startswith("scratch_") or n.startswith("buffer_") for n in names)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image not found")
def test_combine_kernels_keep_scratch_out_of_their_loop(tmp_path):
    """rhs_combine_kernel (midyn_combine.h): whatever hipcc spills in the larger variants stays outside the kk loop -- no
    scratch instruction between the first and the last MFMA of any variant -- and the loads of a step are issued a step
    ahead of their use (the loop waits with vmcnt > 0: it never drains the loads it has just issued)."""
    import subprocess

    if not os.path.exists(LIB):
        pytest.skip("libmidyn.so has not been built")
    text = codeobj.disassembly(LIB, tmp_path, OBJDUMP)
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if "rhs_combine_kernel" in l and l.rstrip().endswith(">:")]
    assert len(starts) == 68
    for st in starts:
        end = next(i for i in range(st + 1, len(lines)) if lines[i].rstrip().endswith(">:") or i == len(lines) - 1)
        ops = [l.split("//")[0].split() for l in lines[st + 1:end]]
        ops = [o for o in ops if o]
        mf = [i for i, o in enumerate(ops) if o[0].startswith("v_mfma_f64_16x16x4")]
        assert mf, lines[st]
        loop = ops[mf[0]:mf[-1] + 1]
        assert not any(o[0].startswith("scratch_") for o in loop), lines[st]
        waits = [o for o in loop if o[0] == "s_waitcnt" and any(t.startswith("vmcnt") for t in o[1:])]
        assert waits and not any("vmcnt(0)" in " ".join(o) for o in waits), lines[st]


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image not found")
def test_sweep_kernels_keep_scratch_out_of_their_contraction_loops(tmp_path):
    """combine_sweep_kernel (midyn_combine_sweep.h), all 128 variants (RK4 and expm action): the innermost loops that hold MFMAs -- the kk loop
    all stages run through -- contain no scratch access and never wait for ALL outstanding loads (the operands of a kk step
    are fetched a step ahead).  Loops are found from the backward branches of the disassembly."""
    import re
    import subprocess

    if not os.path.exists(LIB):
        pytest.skip("libmidyn.so has not been built")
    text = codeobj.disassembly(LIB, tmp_path, OBJDUMP)
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if "combine_sweep_kernelILi" in l and l.rstrip().endswith(">:")]
    assert len(starts) == 128
    for st in starts:
        end = next(i for i in range(st + 1, len(lines)) if lines[i].rstrip().endswith(">:") or i == len(lines) - 1)
        ins = []          # (address, mnemonic, operand text)
        for l in lines[st + 1:end]:
            m = re.match(r"\s*(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", l)
            if m:
                ins.append((int(m.group(3), 16), m.group(1), m.group(2)))
        index = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (addr, op, args) in enumerate(ins):
            if op.startswith(("s_cbranch", "s_branch")) and args.split()[0].isdigit() and int(args.split()[0]) >= 32768:
                target = addr + 4 + (int(args.split()[0]) - 65536) * 4
                if target in index:
                    loops.append((index[target], i))
        inner = [(a, b) for a, b in loops if any(ins[j][1].startswith("v_mfma") for j in range(a, b))
                 and not any(a <= c and d < b for c, d in loops if (c, d) != (a, b))]
        assert len(inner) >= 1, (lines[st], len(loops))
        for a, b in inner:
            body = ins[a:b + 1]
            assert not any(op.startswith("scratch_") for _, op, _ in body), lines[st]
            assert not any(op == "s_waitcnt" and "vmcnt(0)" in args for _, op, args in body), lines[st]
