#!/usr/bin/env python
"""Generates tools/experiments/gemm_list_block.inc (experiment of round 3, see gemm_list_kernel.h): the hand-scheduled gfx950 instruction block that
zgemm_list_kernel (gemm_list_kernel.h) runs once per work-list entry -- all four k-steps of one 128 x 128 x 16
tile on a wave's 64 x 32 share: 24 ds_read_b128, 16 v_mul_f64 (coefficient scaling of the state fragments), 64
v_mfma_f64_16x16x4_f64 and the four LDS-DMA instructions that fetch the operator tile of the NEXT entry.

Why a generated block and not compiler-scheduled C++: hipcc places address arithmetic, DMA issue and list decode in
blocks between the k-steps, where both waves of a SIMD reach them together behind the tile barrier and the matrix pipe
idles (measured: 3.97 us per entry against 3.41 us of MFMA time); sched_group_barrier pipelines were not honoured for a
block of this size (and cost spills).  Here the order is explicit: every fragment read is issued 16+ MFMAs ahead of its
use, a fragment register is re-loaded right after the last MFMA that reads it has issued, the scalings of k-step s+1 sit
inside k-step s, and the DMA instructions fill the LDS latency of the very first reads.

Operands of the asm statement (see zgemm_list_body):
  %0..%15   accumulators, index ((mt * 2 + nt) * 2 + part), part 0 = real, 1 = imaginary   ("+v", 8 VGPRs each)
  %16       LDS byte address of this lane's A fragment, k-step 0 (k-step s is at address ^ 64 s)
  %17       LDS byte address of this lane's B fragment, k-step 0 (k-step s at + 8192 s, column block nt at + 256 nt)
  %18, %19  coefficients of the lane's two columns (doubles)
  %20       LDS-DMA lane offset (bytes) into the next operator tile
  %21..%24  64-bit scalar bases of the four 32-row chunks of the next operator tile
  %25       LDS byte address the first chunk lands at (the others at + 8192 each)
Fixed (clobbered) registers: v190, v191 addresses; v[192:215] / v[216:239] the two fragment sets (A: 4 x 4, B: 2 x 4
VGPRs); v[240:247] / v[248:255] the two sets of scaled B values.
"""
import os
import sys

FA = (192, 216)      # A fragments of set 0 / 1: mt -> +4 mt (x: +0, y: +2)
FB = (208, 232)      # B fragments: nt -> +4 nt
SB = (240, 248)      # scaled: br0 +0, bi0 +2, br1 +4, bi1 +6


def v(lo, n=2):
    return f"v[{lo}:{lo + n - 1}]"


def block(mode, dma=True, reads=True, tail=False):
    """mode 1: purely real operators (A.x), 2: purely imaginary (A.y).  dma / reads False: profiling variants.
    tail: only the 16 MFMAs of the carried k-step 3 (after the last entry)."""
    out = []
    outstanding = []          # tags of ds_reads in flight, oldest first

    def emit(s):
        if not dma and s.startswith("global_load_lds"):
            return
        if not reads and (s.startswith("ds_read") or s.startswith("s_waitcnt lgkmcnt")):
            return
        out.append(s)

    def read_a(st, mt, addr):
        emit(f"ds_read_b128 {v(FA[st] + 4 * mt, 4)}, {addr}" + (f" offset:{4096 * mt}" if mt else ""))
        outstanding.append(("a", st, mt))

    def read_b(st, nt, ks):
        off = 8192 * ks + 256 * nt
        emit(f"ds_read_b128 {v(FB[st] + 4 * nt, 4)}, %19" + (f" offset:{off}" if off else ""))
        outstanding.append(("b", st, nt))

    def wait_for(tags):
        last = max(i for i, t in enumerate(outstanding) if t in tags)
        n = len(outstanding) - 1 - last
        emit(f"s_waitcnt lgkmcnt({n})")
        del outstanding[:last + 1]

    def frags(st):
        return [("a", st, m) for m in range(4)] + [("b", st, n) for n in range(2)]

    def mul(st):
        # br = b.x * s ; bi = b.y * s (mode 1) or -(b.y * s) (mode 2: the product with i A_im)
        neg = "-" if mode == 2 else ""
        for nt, sc in ((0, "%20"), (1, "%21")):
            emit(f"v_mul_f64 {v(SB[st] + 4 * nt)}, {v(FB[st] + 4 * nt)}, {sc}")
            emit(f"v_mul_f64 {v(SB[st] + 4 * nt + 2)}, {neg}{v(FB[st] + 4 * nt + 2)}, {sc}")

    def mfma(st, mt, nt, part):
        acc = f"%{(mt * 2 + nt) * 2 + part}"
        a = v(FA[st] + 4 * mt + (0 if mode == 1 else 2))
        if mode == 1:
            b = v(SB[st] + 4 * nt + (0 if part == 0 else 2))      # re += Ar br ; im += Ar bi
        else:
            b = v(SB[st] + 4 * nt + (2 if part == 0 else 0))      # re += Ai (-bi) ; im += Ai br
        emit(f"v_mfma_f64_16x16x4_f64 {acc}, {a}, {b}, {acc}")

    def mfmas(st, mt):
        for nt in range(2):
            for part in range(2):
                mfma(st, mt, nt, part)

    if tail:
        for mt in range(4):
            mfmas(1, mt)
        return out

    # ---- behind the barrier: fragments of k-step 0 of this tile, the DMA of the next operator tile, and the MFMAs of
    # k-step 3 of the PREVIOUS tile (operands carried in set 1) to cover their latency
    emit("s_setprio 3")
    for mt in range(4):
        read_a(0, mt, "%18")
    for nt in range(2):
        read_b(0, nt, 0)
    for nt in range(2):
        read_b(1, nt, 1)          # the B fragments of set 1 were scaled before the barrier: free
    emit("v_xor_b32 v190, 64, %18")
    emit("s_mov_b32 m0, %27")
    for mt in range(4):
        mfmas(1, mt)
        read_a(1, mt, "v190")     # k-step 1 of this tile into the A fragment its last reader has just issued
        emit("s_nop 0")
        emit(f"global_load_lds_dwordx4 %22, %{23 + mt}")
        if mt < 3:
            emit("s_add_u32 m0, m0, 0x2000")
        if mt == 2:
            wait_for(frags(0))
            mul(0)
    for ks in range(3):
        st = ks & 1
        nxt2 = ks + 2 <= 3
        # the SIMD issues the MFMAs of its OLDER wave first: without this the younger wave runs its block almost alone
        # after the older one has reached the barrier, and every non-MFMA instruction of it idles the pipe (measured: the
        # barrier costs 5 % with, nothing without other instructions between the MFMAs).  Falling priority through the
        # block makes the wave that is behind win: the two waves of a SIMD alternate quarter by quarter.
        emit(f"s_setprio {2 - ks}")
        if nxt2:      # the B fragments of this set are free once scaled: fetch k-step ks+2 into them
            for nt in range(2):
                read_b(st, nt, ks + 2)
            emit(f"v_xor_b32 v{190 + st}, {64 * (ks + 2)}, %18")
        for mt in range(4):
            mfmas(st, mt)
            if nxt2:
                read_a(st, mt, f"v{190 + st}")
            if mt == 2:      # scale the fragments of the next k-step while this one still has MFMAs to issue
                o = st ^ 1
                wait_for(frags(o))
                mul(o)
    assert not outstanding, outstanding
    return out


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_list_block.inc")
    lines = ["// GENERATED by tools/experiments/gen_list_block.py -- do not edit; the schedule and the operand map are described there.", ""]
    for mode, name, kw in ((1, "MODE1", {}), (2, "MODE2", {}), (1, "TAIL_MODE1", {"tail": True}),
                           (2, "TAIL_MODE2", {"tail": True}), (2, "MODE2_NODMA", {"dma": False}),
                           (2, "MODE2_NOREADS", {"reads": False})):
        ins = block(mode, **kw)
        n_mfma = sum(1 for i in ins if i.startswith("v_mfma"))
        n_rd = sum(1 for i in ins if i.startswith("ds_read"))
        assert (n_mfma, n_rd) in ((64, 24), (64, 0), (16, 0)), (n_mfma, n_rd)
        if kw and "tail" not in kw:
            lines.append("#ifdef MIDYN_ABLATE   // profiling variants (results wrong)")
        lines.append(f"#define MIDYN_LIST_BLOCK_{name} \\")
        for i in ins:
            lines.append(f'    "{i}\\n" \\')
        lines[-1] = lines[-1][:-2]
        if kw and "tail" not in kw:
            lines.append("#endif")
        lines.append("")
    regs = ", ".join(f'"v{r}"' for r in list(range(190, 216)) + list(range(232, 248)))
    lines.append(f"#define MIDYN_LIST_BLOCK_CLOBBERS {regs}")
    lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))
    if "--print" in sys.argv:
        print("\n".join(block(2)))


if __name__ == "__main__":
    main()
