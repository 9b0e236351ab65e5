// midyn_flip.h -- ell_flip_duo_kernel: the two-workgroups-per-instance sweep kernel (cfg 5 shard, midyn_resident.h) for
// stacks whose operator slots are FLIP-STRUCTURED (round 5).
//
// Reference: solvers/fixed_step_solvers.py:345-363 (the Magnus propagator of one step), :62-73 step loop; the series of the
// action, the frame picture and the commutator-free Magnus-2 term are those of ell_sweep_kernel (midyn_resident.h).
//
// A stack built from Pauli strings without Z factors in the computational basis (drives X_q, couplings X_q X_q': cfg 5)
// has, after stack_ell_layout has ordered a row's entries by the bits in which column and row differ, slots that hold ONE
// signed magnitude AND one flip mask each: column = row ^ flip[e] in EVERY row.  Then there is nothing to read per element:
//   * the LDS address of an operand is the thread's own address XOR a per-slot constant: no element array, no L2 stream of
//     4 bytes per (slot, row) and pass (311 KB per pass and instance at cfg 5), no register ring that prefetches it -- and
//     the wave's vector-memory counter belongs to the exchange alone;
//   * so the exchange can be spread over the pass without draining anything: the rows are published before the first barrier;
//     the store acknowledgement and the flag store follow the first eighth of the local slots; the partner's flag is asked for,
//     without waiting, after 5/16 of them; the loads of the crossing operands leave after half of them (inside the slot loop:
//     see run_plane) and land behind the other half.  With element loads in the slot loop this split LOST
//     (profiles/r05_cfg5_duo.md, v3b: every part restarted and drained the element prefetch);
//   * the few slots that cross the halves read their operands STRAIGHT from the partner's payload (ONE set of 16-byte loads
//     into registers for all crossing slots whose flips share their thread bits): the partner's half is never staged in LDS --
//     one barrier and (RPT x ORDER) ds_write_b128 per thread and pass less, and the workgroup holds only its own half in LDS
//     (64 KB at n = 4096 -- twice: consecutive passes alternate between two operand buffers, see the kernel);
//   * the slot loops are bound by their VECTOR instructions, not by the LDS (ablations in profiles/r05_cfg5_duo.md), so a
//     slot carries as few as possible: its flip mask through ONE v_readlane_b32 from a lane-held copy (lane j: slot j; at
//     most 64 slots), its two coefficients through ONE 16-byte scalar load (constant address space) from a table a small
//     kernel fills per (instance, step, slot) before the launch, requested before the slot's gathers and waited for with them;
//     gathers addressed in the LDS address space (no base add); the scale of the second sum in the staged operand.
// Protocol, flags, payload slots, one-L2 detection, give-up behaviour: ell_sweep_duo_kernel's.  A wave that reads rows of
// another wave of its partner (a crossing flip with bits above the lane bits) waits for THAT wave's flag.
#pragma once
#include <type_traits>

#ifndef MIDYN_FLIP_ABLATE
#define MIDYN_FLIP_ABLATE 0
#endif
#ifndef MIDYN_FLIP_SMEM
#define MIDYN_FLIP_SMEM 1     // slot coefficients through the scalar cache (0: v_readlane from lane-held copies)
#endif
#define MIDYN_CONST_AS __attribute__((address_space(4)))

namespace midyn {

typedef double flip_d2 __attribute__((ext_vector_type(2)));

struct FlipDuoArgs {
    int n, n_pad, nsteps;
    int wsp;                    // operator slots per row, in the order they are applied:
    int n_loc;                  // [0, n_loc) keep every row's operand in the row's own half, [n_loc, wsp) cross the halves
    int n_re_loc;               // [0, n_re_loc) of the local slots hold real-plane values, [n_re_loc, n_loc) imaginary-plane values
    const int* meta;            // [wsp] flip mask | (imaginary-plane slot) << 31
    const flip_d2* cab;         // [B][nsteps][wsp] (c(t1), c(t2)) x signed magnitude of the slot (flip_cab_kernel)
    const double2* E;           // [R][n_pad] or nullptr
    const double2* Dt;          // order 2, framed: [nsteps][n_pad] E(t2) o conj(E(t1))
    const int* rows;            // [nsteps][3]
    const double* hs;           // [nsteps]
    const int* save;            // [nsteps] or nullptr
    const int* ser_K;           // per step: > 0 Chebyshev terms, < 0 -(Taylor degree)
    const int* ser_reps;
    const double* ser_par;
    const double* coef;         // [nsteps][stride]
    int stride;
    const double2* y0;          // [B | 1][n]
    int y0_shared;
    double2* out;               // [B][P][n]
    int P;
    double2* ring;              // [B][3][n_pad] payload slots
    int* flags;                 // [B][2][DUO_FLAG_WORDS]
    int* err;
    unsigned spin_limit;
    int protocol;               // ctx option exchange_protocol: 1 = payload written through (sc1) and agent-scope flags on one XCD too
    int part_major;
    int ablate;                 // profiling only (results wrong): 1 no exchange, 2 write-through stores on one XCD too, 4 no local slots,
                                // 8 no crossing slots (their waits and loads included)
};

// cab[b][st][j] = mag[j] * (S[b][rows[3 st]][seg_j], S[b][rows[3 st + 1]][seg_j]) -- the products ell_sweep_duo_kernel forms
// at the top of every step, for all steps, once (segment of slot j: tags[j] & 63; the static operator has coefficient 1)
MIDYN_GLOBAL __launch_bounds__(256) void flip_cab_kernel(const double* __restrict__ S, long long inst_stride, int k, int has_static,
                                                        const int* __restrict__ rows, int nsteps, const int* __restrict__ tags,
                                                        const double* __restrict__ mag, int wsp, int order, int B,
                                                        double2* __restrict__ cab) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * nsteps * wsp) return;
    const int j = (int)(idx % wsp);
    const int st = (int)((idx / wsp) % nsteps);
    const int b = (int)(idx / ((size_t)wsp * nsteps));
    const int seg = tags[j] & 63;
    const bool stat = has_static && seg == 0;
    const double* Sb = S + (size_t)b * inst_stride;
    const int r0 = rows[3 * st], r1 = rows[3 * st + 1];
    const double mg = mag[j];
    cab[idx] = make_double2(mg * (stat ? 1.0 : Sb[(size_t)r0 * k + seg - has_static]),
                            order == 2 ? mg * (stat ? 1.0 : Sb[(size_t)r1 * k + seg - has_static]) : 0.0);
}


template <int ORDER, int RPT, int TH>
__global__ __launch_bounds__(TH) void ell_flip_duo_kernel(const FlipDuoArgs a) {
    // own half: X1 [half] (| X2 32768 bytes behind, order 2), TWICE: the copies of consecutive passes alternate between two buffers
    // FLIP_BUF bytes apart, so a pass writes its operands while slower waves still gather those of the pass before -- ONE barrier
    // per pass (behind the writes) instead of two (round 5, last session: the barrier in front of the writes cost 0.45 us of a
    // 10.9 us term on the cfg 5 shard, tools/bench_cfg5_variants.py no_barrier1).  Safe with one barrier: a wave that writes
    // buffer p & 1 for pass p + 2 has passed the barrier of pass p + 1, which every wave reaches after its gathers of pass p.
    extern __shared__ __attribute__((aligned(16))) double2 flip_lds[];
    // (round 6: a software-pipelined slot loop was built, measured slower and removed -- profiles/r06_cfg5_pipeline.md)
    constexpr bool SMEM = MIDYN_FLIP_SMEM != 0;
    constexpr unsigned FLIP_BUF = ORDER == 2 ? 65536u : 32768u;
    const int tid = threadIdx.x, np = a.n_pad;
    const unsigned unp = (unsigned)np;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = gridDim.x >> 1;
    const int b = a.part_major ? (int)(blockIdx.x % nb) : (int)(blockIdx.x >> 1);
    const int part = a.part_major ? (int)(blockIdx.x / nb) : (int)(blockIdx.x & 1);
    const unsigned half = unp >> 1;                             // = TH * RPT rows per workgroup
    const unsigned row0 = part ? half : 0u;
    char* const lds = reinterpret_cast<char*>(flip_lds);
    auto rowof = [&](const int i_) {   // tid + TH i_, opaque to the optimiser: nothing derived from it is hoisted out of the loops
        unsigned r_ = (unsigned)(tid + TH * i_);
        asm volatile("" : "+v"(r_));
        return r_;
    };
#define AT16(base_, i_) (*reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base_) + (rowof(i_) << 4)))
    const int n_loc = a.n_loc, n_all = a.wsp;
    const int lane = tid & 63;
    const int meta_l = lane < n_all ? a.meta[lane] : 0;      // lane j: flip mask | plane of slot j
    const int xm_l = (meta_l & 0x7fffffff) << 4;             // lane j: flip mask of slot j as an LDS byte offset
    auto lane_i32 = [](const int v, const int j) { return __builtin_amdgcn_readlane(v, j); };
    auto lane_f64 = [](const double v, const int j) {
        const long long q = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)q, j), hi = (unsigned)__builtin_amdgcn_readlane((int)(q >> 32), j);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    // the payload slots of this instance through a buffer descriptor (16-byte loads / stores; aux 16 = sc1)
    const size_t ring_doubles2 = (size_t)3 * np;
    const __amdgpu_buffer_rsrc_t ring = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double2*>(a.ring) + (size_t)b * ring_doubles2, 0, (int)(ring_doubles2 * sizeof(double2)), 0x00020000);
    auto ring_off = [&](const int slot, const unsigned row) { return (unsigned)((slot * np + (int)row) << 4); };
    int* const my_flags = a.flags + ((size_t)b * 2 + part) * DUO_FLAG_WORDS;
    int* const partner_flags = a.flags + ((size_t)b * 2 + (1 - part)) * DUO_FLAG_WORDS;
    bool dead = false;
    auto wait_word = [&](int* word, const int want) {       // a bounded wait for a word of the partner to reach `want`
        unsigned spins = 0;
        for (;;) {
            if (dead) return 0;
            const int got = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got >= want) return got;
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) spins = a.spin_limit;
            if (spins >= a.spin_limit) {
                __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dead = true;
            }
        }
    };
    const double p2 = 0.14433756729740643;   // sqrt(3) / 12
    // a wave-uniform double into a pair of SCALAR registers (the per-step and per-term factors of the series are computed by
    // vector instructions and would otherwise each hold two vector registers through the passes -- round 6: with the pipeline's
    // operand sets in flight the compiler spilled them to scratch, a load + vmcnt(0) per term)
    auto uni = [](const double v) {
        const long long q = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)q), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(q >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    double2 pw[RPT], acc[RPT], cur[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const unsigned r = row0 + rowof(i);
        // (an unconditional load of a valid row and a select: a branch around the load kept its 64-bit address alive -- in scratch)
        const double2 yv = a.y0[(a.y0_shared ? 0 : (size_t)b * a.n) + (r < (unsigned)a.n ? r : 0u)];
        acc[i] = (r < (unsigned)a.n) ? yv : make_double2(0.0, 0.0);
        cur[i] = pw[i] = make_double2(0.0, 0.0);
    }
    const bool exch = !(a.ablate & 1) && n_all > n_loc;     // (no crossing slot: the halves are independent problems)
    bool one_l2 = false;
    if (exch) {      // do the partners share an XCD (one L2)?  Each publishes its XCC id + 1 once and reads the other's.
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        if (tid == 0) __hip_atomic_store(my_flags + 16, (int)xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int theirs = wait_word(partner_flags + 16, 1);
        one_l2 = !dead && theirs == (int)xcc + 1 && !(a.ablate & 2) && !a.protocol;
    }
    // the local slots in four parts when there is an exchange (see above): part p = slots [cut[p], cut[p + 1]).  The exchange needs
    // most of a pass from the stores to the last operand in a register (acknowledgement, flag, flag seen, 64 KB of loads per
    // workgroup): its steps sit early -- after 1/8, 5/16 and 1/2 of the local slots -- and the operands land behind the second half
    const bool any_cross = n_all > n_loc;
    const int cut[5] = {0, any_cross ? (n_loc + 4) / 8 : n_loc, any_cross ? (5 * n_loc + 8) / 16 : n_loc,
                        any_cross ? (n_loc + 1) / 2 : n_loc, n_loc};
    double cx_l = 0.0, cy_l = 0.0;                            // lane j: (c(t1), c(t2)) x signed magnitude of slot j, this step

    // one operator slot on this thread's rows: x1 / x2 are its operands in the two vectors
    auto slot_fma = [&](const bool im, const double ca, const double cb, const double2 (&x1)[RPT], const double2 (&x2)[RPT],
                        double2 (&o1)[RPT], double2 (&o2)[RPT]) {
        if (im) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                o1[i].x = fma(-ca, x1[i].y, o1[i].x);
                o1[i].y = fma(ca, x1[i].x, o1[i].y);
                if (ORDER == 2) {
                    o2[i].x = fma(-cb, x2[i].y, o2[i].x);
                    o2[i].y = fma(cb, x2[i].x, o2[i].y);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                o1[i].x = fma(ca, x1[i].x, o1[i].x);
                o1[i].y = fma(ca, x1[i].y, o1[i].y);
                if (ORDER == 2) {
                    o2[i].x = fma(cb, x2[i].x, o2[i].x);
                    o2[i].y = fma(cb, x2[i].y, o2[i].y);
                }
            }
        }
    };
    // the local slots [j_lo, j_hi): operands gathered from the own half in LDS at (own address) ^ (flip << 4).  The real-plane
    // slots come first ([0, n_re_loc)): two straight-line loops, no per-slot branch on the plane.  A step of the exchange that
    // issues loads (hook) runs INSIDE the loop, before slot j_ev: hipcc drains the vector-memory counter in front of a loop when
    // loads issued before it are still in flight (s_waitcnt vmcnt(0) in the preheader: the whole latency, exposed), but not
    // around the back edge of a loop that issues them itself.  (All three steps inside one loop over all local slots cost
    // 1 us per term in scalar compares: a slot is twenty instructions.)
    // The two coefficients of slot j as the current pass uses them (pass_swapped: the second pass of a term applies the
    // coefficient sets the other way round).  MIDYN_FLIP_SMEM: one 16-byte scalar load per slot from the table of this
    // (instance, step) -- requested before the slot's gathers and waited for with them (lgkmcnt), no VALU instruction; the
    // flip mask, which the gathers need FIRST, stays a v_readlane.  (Measured, cfg 5 shard, the slot loops alone: with the
    // gathers removed they still took 6.0 of their 6.7 us per term -- eight multiply-adds, five v_readlane and two XORs per slot
    // are the bound, not the LDS; the scale of the second sum went into the staged operand for the same reason.)
    // (Round 6: no "swapped" coefficient sets any more -- the second pass of a term hands its operands over in the order of the
    // coefficient sets instead, see the term loop: slot coefficient a always meets the operand in the X1 buffer.)
    const MIDYN_CONST_AS flip_d2* ccab = nullptr;     // the coefficient table of this (instance, step)
    auto coef_a = [&](const int j) {
        if constexpr (SMEM) return (double)ccab[j].x;
        else return lane_f64(cx_l, j);
    };
    auto coef_b = [&](const int j) {
        if constexpr (SMEM) return (double)ccab[j].y;
        else return lane_f64(cy_l, j);
    };
    // LDS byte address of this thread's rows as an INTEGER in the LDS address space: (la ^ flip) is then the operand's address with
    // no addition of the (link-time) base of the dynamic LDS per gather -- two of a slot's thirteen vector instructions.  XOR and
    // base commute only when the base has no bit below 64 KB: this kernel declares no static LDS (base 0); checked once.
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned lds_base = (unsigned)(size_t)(lds_char*)flip_lds;
    if (lds_base & 0x1ffffu) __builtin_trap();     // (bit 16 too: the buffer toggle)
    unsigned la[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) la[i] = (lds_base + ((unsigned)(tid + TH * i) << 4)) ^ FLIP_BUF;   // (the first pass toggles it back)
    unsigned boff = FLIP_BUF;       // byte offset of the buffer the current pass reads
    auto run_plane = [&](const int j_lo, const int j_hi, auto im_tag, const int j_ev, auto& hook, double2 (&o1)[RPT], double2 (&o2)[RPT]) {
        constexpr bool IM = decltype(im_tag)::value;
        for (int j = j_lo; j < j_hi; ++j) {
            if (j == j_ev) hook();
            const unsigned xm = (unsigned)lane_i32(xm_l, j);
            double2 x1[RPT], x2[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
#if MIDYN_FLIP_ABLATE == 2       // profiling only: no gathers
                x1[i] = make_double2(__hiloint2double((int)xm, (int)la[i]), 1.0);
                x2[i] = x1[i];
#else
                typedef __attribute__((address_space(3))) const flip_d2 lds_d2;
                lds_d2* q = (lds_d2*)(size_t)(la[i] ^ xm);
                const flip_d2 g1 = q[0], g2 = ORDER == 2 ? q[2048] : g1;       // (X2: 32768 bytes behind)
                x1[i] = make_double2(g1.x, g1.y);
                x2[i] = make_double2(g2.x, g2.y);
#endif
            }
#if MIDYN_FLIP_ABLATE == 1           // profiling only: one add per gathered operand instead of the multiply-adds
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                o1[i].x += x1[i].x + x1[i].y;
                o2[i].x += x2[i].x + x2[i].y;
            }
#else
            slot_fma(IM, coef_a(j), ORDER == 2 ? coef_b(j) : 0.0, x1, x2, o1, o2);
#endif
        }
    };
    auto run_local = [&](const int j_lo, const int j_hi, auto& hook, double2 (&o1)[RPT], double2 (&o2)[RPT]) {
        if (a.ablate & 4) return;
        const int nre = a.n_re_loc;
        run_plane(j_lo, j_hi < nre ? j_hi : nre, std::false_type(), j_lo, hook, o1, o2);
        run_plane(j_lo > nre ? j_lo : nre, j_hi, std::true_type(), j_lo, hook, o1, o2);
    };
    int rr = 0;       // exchange rounds so far
    // One exchange round and the pass it feeds.  in1 / in2: this thread's rows of the operand vectors as the operators see
    // them (X1 / X2 forms).  nv = 1: one vector travels (in1) and the X2 form of the partner's rows is dtab o in1 (dtab: the
    // frame phase between the Gauss points of this step; use_dp false: none); nv = 2: both forms travel.
    // o1 = sum over the slots of (coefficient a) x (in1 operand), o2 = of (coefficient b) x (in2 operand).
    // keep: o1 continues from its start value and its sum is scaled by scale1; else both sums start at zero.
    auto exchange_and_pass = [&](const double2 (&in1u)[RPT], const double2 (&in2)[RPT], const int nv, const double2* dtab,
                                 const bool use_dp, const bool keep, const double scale1,
                                 double2 (&o1)[RPT], double2 (&o2)[RPT]) {
        const int slot = ORDER == 2 ? (nv == 2 ? 1 : 0) : (rr & 1);      // payload slot(s) of this round
        // (the scale of the kept sum is applied to its OPERAND -- staged and published scaled -- not to every slot's coefficient)
        double2 in1[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) in1[i] = keep ? make_double2(scale1 * in1u[i].x, scale1 * in1u[i].y) : in1u[i];
        const double2 (&in2s)[RPT] = in2;
        if (exch) {                        // the rows leave first: nothing below needs them before the partner does
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const unsigned l = rowof(i);
#pragma unroll
                for (int v = 0; v < ORDER; ++v) {
                    if (v >= nv) continue;
                    const double2 z = v ? in2s[i] : in1[i];
                    sweep_u4 w;
                    const unsigned long long zx = (unsigned long long)__double_as_longlong(z.x), zy = (unsigned long long)__double_as_longlong(z.y);
                    w.x = (unsigned)zx; w.y = (unsigned)(zx >> 32); w.z = (unsigned)zy; w.w = (unsigned)(zy >> 32);
                    if (one_l2) __builtin_amdgcn_raw_buffer_store_b128(w, ring, (int)ring_off(slot + v, row0 + l), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(w, ring, (int)ring_off(slot + v, row0 + l), 0, 16);
                }
            }
        }
        boff ^= FLIP_BUF;                  // (no barrier here: the readers of the previous pass use the other buffer)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const unsigned l = rowof(i);
            la[i] ^= FLIP_BUF;
            *reinterpret_cast<double2*>(lds + boff + (l << 4)) = in1[i];
            if (ORDER == 2) *reinterpret_cast<double2*>(lds + boff + (l << 4) + 32768) = in2s[i];
            if (!keep) o1[i] = make_double2(0.0, 0.0);
            o2[i] = make_double2(0.0, 0.0);
        }
        __syncthreads();
        ++rr;
        // The crossing slots.  The operands of row r in slot j are the partner's row r ^ flip[j]: with the flip split into thread
        // bits ft, row-index bits fi (and the bit of the half), thread tid needs the RPT rows of the partner's thread tid ^ ft --
        // published by its wave  wave ^ (ft >> 6): wait for THAT wave's flag, once per round -- permuted by fi.  Crossing slots
        // with the same ft share ONE set of loads (cfg 5: the drive and the XX coupling of the top qubit, ft = 0); the loads of
        // the first set leave after half of the local slots and land behind the other half, further sets one by one.
        constexpr int LOG_TH = TH == 1024 ? 10 : (TH == 512 ? 9 : 8);
        sweep_u4 w[RPT][ORDER];
        sweep_u4 wd[RPT];       // (one vector travels, framed: the frame phases of the same rows for its second form)
        double2 xb1[RPT], xb2[RPT];
        unsigned seen = 0;      // partner waves whose flag of this round has been seen
        int ft_loaded = -1;     // thread bits of the set that is in w / xb (-1: none)
        bool unpacked = true;
        const bool with_dp = ORDER == 2 && nv == 1 && use_dp;
        auto cross_load = [&](const int ft) {
            const int pwv = wave ^ (ft >> 6);
            if (!((seen >> pwv) & 1u) && !(a.ablate & 32)) {
                (void)wait_word(partner_flags + pwv, rr);
                seen |= 1u << pwv;
            }
            asm volatile("" ::: "memory");          // (compiler: the payload is read after the flag, not before)
            const unsigned prow0 = part ? 0u : half;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const unsigned c = prow0 + ((rowof(i) & (unsigned)(TH - 1)) ^ (unsigned)ft) + (unsigned)(TH * i);
#pragma unroll
                for (int v = 0; v < ORDER; ++v)
                    if (v < nv) w[i][v] = __builtin_amdgcn_raw_buffer_load_b128(ring, (int)ring_off(slot + v, c), 0, 16);
                if (with_dp) wd[i] = *reinterpret_cast<const sweep_u4*>(reinterpret_cast<const char*>(dtab) + (c << 4));
            }
            ft_loaded = ft;
            unpacked = false;
        };
        auto cross_unpack = [&]() {
            auto unpack = [](const sweep_u4 q) {
                return make_double2(__longlong_as_double((long long)(((unsigned long long)q.y << 32) | q.x)),
                                    __longlong_as_double((long long)(((unsigned long long)q.w << 32) | q.z)));
            };
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                xb1[i] = unpack(w[i][0]);
                if (ORDER == 2) xb2[i] = nv == 2 ? unpack(w[i][ORDER - 1]) : (with_dp ? cmul(unpack(wd[i]), xb1[i]) : xb1[i]);
                else xb2[i] = xb1[i];
            }
            unpacked = true;
        };
        auto cross_apply = [&](const int j, const double ca, const double cb) {
            const int mt = lane_i32(meta_l, j);
            const unsigned m = (unsigned)(mt & 0x7fffffff);
            const int ft = (int)(m & (unsigned)(TH - 1)), fi = (int)((m >> LOG_TH) & (unsigned)(RPT - 1));
            if (ft != ft_loaded) cross_load(ft);
            if (!unpacked) cross_unpack();
            if (a.ablate & 512) {       // profiling only: the wait for the operands and one add, no multiply-adds
                o1[0].x += xb1[0].x + xb2[RPT - 1].y;
                return;
            }
            // Four straight-line variants (plane x row flip), every multiply-add IN PLACE (v_fmac_f64 through asm): written with
            // fma() the variants leave their sums in different registers and the merge costs eight 64-bit moves per variant --
            // the two crossing slots of cfg 5 took 1.0 us per term that way, as much as six local slots.
            auto fmac = [](double& acc, const double c, const double x) { asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "s"(c), "v"(x)); };
            const double na = -ca, nb = -cb;
            auto block = [&](auto im_tag, auto fi_tag) {
                constexpr bool IM = decltype(im_tag)::value;
                constexpr int FI = decltype(fi_tag)::value;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const double2 x1 = xb1[i ^ FI], x2 = xb2[i ^ FI];
                    if (IM) {
                        fmac(o1[i].x, na, x1.y);
                        fmac(o1[i].y, ca, x1.x);
                        if (ORDER == 2) {
                            fmac(o2[i].x, nb, x2.y);
                            fmac(o2[i].y, cb, x2.x);
                        }
                    } else {
                        fmac(o1[i].x, ca, x1.x);
                        fmac(o1[i].y, ca, x1.y);
                        if (ORDER == 2) {
                            fmac(o2[i].x, cb, x2.x);
                            fmac(o2[i].y, cb, x2.y);
                        }
                    }
                }
            };
            const bool im = mt < 0;
            if (RPT == 2 && fi) {       // (the slot flips the row index too: the same rows, swapped)
                if (im) block(std::true_type(), std::integral_constant<int, RPT == 2 ? 1 : 0>());
                else block(std::false_type(), std::integral_constant<int, RPT == 2 ? 1 : 0>());
            } else {
                if (im) block(std::true_type(), std::integral_constant<int, 0>());
                else block(std::false_type(), std::integral_constant<int, 0>());
            }
        };
        const bool xc = exch && !dead && !(a.ablate & 8);
        const int ft0 = (int)((unsigned)lane_i32(meta_l, n_loc < n_all ? n_loc : 0) & (unsigned)(TH - 1));
        const int pw0 = wave ^ (ft0 >> 6);
        // The steps of the exchange, before the local slots cut[1], cut[2], cut[3]:
        //   1. this wave's rows have reached memory (the L2 both partners share, or written through) by now: its flag says so;
        //   2. the flag of the wave the first set of crossing operands comes from is asked for EARLY, without waiting -- the
        //      answer is here at the next step, and when it says "published" nobody waits for a flag at all;
        //   3. the loads of that set leave (behind a wait for the flag if it was not up yet).
        int early = 0;
        bool flagged = !exch, asked = false;
        auto raise_flag = [&]() {
            if (!(a.ablate & 16)) __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
            if ((tid & 63) == 0) {
                // (one L2: a store that stays in it -- workgroup scope lowers to sc0, which keeps the line; agent scope writes through)
                if (one_l2) __hip_atomic_store(my_flags + wave, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(my_flags + wave, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            flagged = true;
        };
        auto nothing = [&]() {};
        auto ask = [&]() {
            if (xc && !(a.ablate & 32)) {
                early = __hip_atomic_load(partner_flags + pw0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asked = true;
            }
        };
        auto fetch = [&]() {
            if (xc && ft_loaded < 0) {
                if (asked && __builtin_amdgcn_readfirstlane(early) >= rr) seen |= 1u << pw0;
                cross_load(ft0);
            }
        };
        const int c3 = (a.ablate & 128) ? cut[2] + 1 : ((a.ablate & 256) ? (3 * n_loc) / 4 : cut[3]);
        run_local(cut[0], cut[1], nothing, o1, o2);
        if (!flagged) raise_flag();
        run_local(cut[1], cut[2], nothing, o1, o2);
        run_local(cut[2], c3, ask, o1, o2);
        run_local(c3, cut[4], fetch, o1, o2);
        if (xc && !(a.ablate & 64)) {
            // (the coefficients of the first two crossing slots are asked for together, before the wait for the operands: a scalar
            // load in front of every crossing slot was 0.4 us per term, all waves of the workgroup waiting for it at once)
            const int j0 = n_loc, j1 = n_loc + 1 < n_all ? n_loc + 1 : n_loc;
            const double ca0 = coef_a(j0), cb0 = ORDER == 2 ? coef_b(j0) : 0.0, ca1 = coef_a(j1), cb1 = ORDER == 2 ? coef_b(j1) : 0.0;
            cross_apply(j0, ca0, cb0);
            if (j1 > j0) cross_apply(j1, ca1, cb1);
            for (int jc = n_loc + 2; jc < n_all; ++jc) cross_apply(jc, coef_a(jc), ORDER == 2 ? coef_b(jc) : 0.0);
            // (from lane-held copies through v_readlane instead: the same 10.2 us per term)
        }
        // (nothing is in flight here -- said in a form the compiler's counter model reads: otherwise a path on which loaded
        // operands were never consumed reaches the next pass, and the slot loops that reuse their registers wait for vmcnt(0),
        // i.e. for the acknowledgement of the stores issued just before them)
        __builtin_amdgcn_s_waitcnt(0x0F70);
    };
    for (int st = 0; st < a.nsteps; ++st) {
        const int r0 = a.rows[3 * st];
        const double h = uni(a.hs[st]);
        ccab = (const MIDYN_CONST_AS flip_d2*)(a.cab + ((size_t)b * a.nsteps + st) * a.wsp);
        if (!SMEM && lane < n_all) {
            // (scalar base of this (instance, step) + a 32-bit lane offset the optimiser cannot hoist: otherwise the per-thread 64-bit
            // address a.cab + lane lives in a vector register pair through the whole kernel)
            unsigned lo = (unsigned)lane << 4;
            asm volatile("" : "+v"(lo));
            const char* const cbase = reinterpret_cast<const char*>(a.cab + ((size_t)b * a.nsteps + st) * a.wsp);
            const flip_d2 cc = *reinterpret_cast<const flip_d2*>(cbase + lo);
            cx_l = cc.x;
            cy_l = cc.y;
        }
        // (the two table rows of this step as SCALAR pointers the optimiser cannot look through: with the thread's offset added
        // outside the step loop their 64-bit per-thread addresses lived in vector registers for the whole kernel -- one in scratch)
        auto uni_ptr = [](const double2* q) {
            const unsigned long long v = (unsigned long long)q;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32));
            return (const double2*)(((unsigned long long)hi << 32) | lo);
        };
        const double2* const E0 = uni_ptr(a.E ? a.E + (size_t)r0 * np + row0 : nullptr);
        const bool framed2 = a.E && ORDER == 2;
        // the frame phase between the two Gauss points of this step for this thread's rows
        const double2* const dtab = framed2 ? a.Dt + (size_t)st * np : nullptr;
        const double2* const dtab_own = uni_ptr(framed2 ? dtab + row0 : nullptr);
        double2 dmr[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) dmr[i] = framed2 ? AT16(dtab_own, i) : make_double2(1.0, 0.0);
        const int Ks = a.ser_K[st], reps = a.ser_reps[st];
        const bool cheb = Ks > 0;
        const int K = cheb ? Ks : -Ks;
        double p2s = p2;                       // (sqrt(3) / 12 in a scalar pair, made here: as a vector-register constant hoisted
        asm volatile("" : "+s"(p2s));          // out of the step loop it was spilled to scratch)
        const double par = uni(a.ser_par[st]);
        // Chebyshev: the factor of a term is 1 / par (first term) or 2 / par -- ONE division per step (2 x (1 / par) == 2 / par bit
        // for bit), and the products with the step size that the terms need, all as scalars
        const double f_1 = uni(1.0 / par), f_2 = uni(2.0 * f_1);
        const double ca_1 = uni((ORDER == 2 ? 0.5 * h : h) * f_1), ca_2 = uni((ORDER == 2 ? 0.5 * h : h) * f_2);
        const double cb_1 = uni(p2s * h * h * f_1), cb_2 = uni(p2s * h * h * f_2);
        const double* coef = a.coef + (size_t)st * a.stride;
        const int slot = a.save ? a.save[st] : -1;
        for (int rep = 0; rep < reps; ++rep) {
            const double c0 = uni(cheb ? coef[0] : 1.0);
            // start of a series: phi_0 = the accumulated result (into the frame picture of the first Gauss point at the
            // first repetition: y~ = E(t1) o y)
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                double2 v = acc[i];
                if (rep == 0 && a.E) v = cmul(AT16(E0, i), v);
                cur[i] = v;
                acc[i] = make_double2(c0 * v.x, c0 * v.y);
                pw[i] = make_double2(0.0, 0.0);      // Chebyshev: phi_{j-2};  Taylor: nothing
            }
            for (int j = 1; j <= K; ++j) {
                // (Chebyshev: 1 / par or 2 / par -- ONE division per step, inv_par; 2 x (1 / par) == 2 / par bit for bit)
                double ca, cb;
                if (cheb) {
                    ca = j == 1 ? ca_1 : ca_2;
                    cb = j == 1 ? cb_1 : cb_2;
                } else {
                    const double f = 1.0 / (par * (double)j);
                    ca = uni((ORDER == 2 ? 0.5 * h : h) * f);
                    cb = uni(p2s * h * h * f);
                }
                double2 o1[RPT], o2[RPT], in2[RPT];
                if (ORDER == 2) {
#pragma unroll
                    for (int i = 0; i < RPT; ++i) in2[i] = framed2 ? cmul(dmr[i], cur[i]) : cur[i];
                }
                // o1 = C(t1) v~, o2 = C(t2) (D v~)
                exchange_and_pass(cur, ORDER == 2 ? in2 : cur, 1, dtab, framed2, false, 1.0, o1, o2);
                double2 w[RPT];
                if (ORDER == 2) {

                    double2 du1[RPT], u2[RPT], m0[RPT];
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const double2 u1 = o1[i];
                        u2[i] = o2[i];
                        du1[i] = u1;
                        if (framed2) {
                            u2[i] = cmul_conj_a(dmr[i], o2[i]);
                            du1[i] = cmul(dmr[i], u1);       // for g2~ = conj(D) C(t2) D
                        }
                        // the term so far, m = phi_{j-2} + ca (u1 + u2), rides through the second pass INSIDE its second sum
                        // (start value -m, the sum scaled by cb: cb v1 - o2 then IS m + cb (v1 - g1~ u2))
                        m0[i] = make_double2(-(pw[i].x + ca * (u1.x + u2[i].x)), -(pw[i].y + ca * (u1.y + u2[i].y)));
                    }
#pragma unroll
                    for (int i = 0; i < RPT; ++i) o2[i] = m0[i];
                    // (coefficient set a = C(t1) meets the first operand, b = C(t2) the second: u2 goes first, into the kept sum)
                    exchange_and_pass(u2, du1, 2, dtab, false, true, cb, o2, o1);           // o2 = -m + cb C(t1) u2, o1 = C(t2) (D u1)
#pragma unroll
                    for (int i = 0; i < RPT; ++i) {
                        const double2 v1 = framed2 ? cmul_conj_a(dmr[i], o1[i]) : o1[i];
                        w[i] = make_double2(cb * v1.x - o2[i].x, cb * v1.y - o2[i].y);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < RPT; ++i) w[i] = cfma_r(ca, o1[i], pw[i]);
                }
                // end of the term: w joins the result and is the next term's input; the old phi_{j-1} is the next phi_{j-2}
                const bool last = j == K;
                const double cj = uni(cheb ? 2.0 * coef[j] : 1.0);
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const unsigned r = row0 + rowof(i);
                    double2 ac = cfma_r(cj, w[i], acc[i]);
                    if (!last) {
                        pw[i] = cheb ? cur[i] : make_double2(0.0, 0.0);
                        cur[i] = w[i];
                    } else if (rep + 1 == reps) {    // out of the frame picture; saved states
                        if (a.E) ac = cmul_conj_a(AT16(E0, i), ac);
                        if (slot >= 0 && r < (unsigned)a.n) a.out[((size_t)b * a.P + slot) * a.n + r] = ac;
                    }
                    acc[i] = ac;
                }
            }
        }
    }
}
#undef AT16

// ---- the instantiations that exist: midyn_tu_flip.hip defines MIDYN_TU_FLIP -------------------------------------------
#ifdef MIDYN_TU_FLIP
#define MIDYN_FLIP_EXTERN
#else
#define MIDYN_FLIP_EXTERN extern
#endif
#define MIDYN_X(O_)                                                                                  \
    MIDYN_FLIP_EXTERN template __global__ void ell_flip_duo_kernel<O_, 2, 1024>(const FlipDuoArgs);  \
    MIDYN_FLIP_EXTERN template __global__ void ell_flip_duo_kernel<O_, 1, 1024>(const FlipDuoArgs);  \
    MIDYN_FLIP_EXTERN template __global__ void ell_flip_duo_kernel<O_, 1, 512>(const FlipDuoArgs);   \
    MIDYN_FLIP_EXTERN template __global__ void ell_flip_duo_kernel<O_, 1, 256>(const FlipDuoArgs);
MIDYN_X(1) MIDYN_X(2)
#undef MIDYN_X

}  // namespace midyn
