// midyn_combine_sweep.h -- fixed-step RK4 and scipy_expm (Magnus 1) sweeps of SMALL dense systems (n_pad <= 256) in ONE launch (round 4).
//
// The per-launch route runs a batched evaluation as one kernel: at n = 64 .. 256 and a few thousand instances such a
// launch holds microseconds of matrix-pipe work, so a solve is bound by launches, by the state going through memory four
// times per step and -- with 64 instances per wave -- by too few waves to fill the chip.  Trajectories of a sweep are
// independent (solvers/solver_classes.py:568-586 loops over them), so nothing has to cross workgroups: a workgroup owns 16
// INSTANCES through ALL steps (fixed_step_solvers.py:43-77, the RK4 step; :406-459, the step loop).
//
//   * the evaluation is COMBINE + APPLY of midyn_combine.h on the same re-packed operator planes and zero-block lists
//     (v_mfma_f64_16x16x4 over 4 planes x 16 rows x 16 instances, then 2 .. 4 vector FMAs per element), one wave per
//     RT 16-row tiles (RT = 1: n_pad <= 128, RT = 2 above), at n_pad = 64 two waves per tile that split the list, exchange
//     their partial sums through LDS and share the rows of the tile for the stage arithmetic;
//   * y and the RK4 accumulator of a wave's rows stay in its registers for the whole solve (RT = 1; at RT = 2 the registers
//     are taken by the contraction and a stage lasts tens of microseconds: they are read and written once per stage in
//     the workgroup's own columns of two [n_pad][ld] blocks); the stage input, already phased with the frame phases of the
//     stage time, is the only vector the waves exchange: two LDS copies [n_pad][16], written by the stage before, one
//     barrier per stage;
//   * coefficients c_j[instance] of a stage: one load per plane group and lane from the table S[B][R][k], a stage ahead.
//
// 256 workgroups of 16 instances fill the chip at 4096 instances; the time of a stage is the workgroup's matrix-pipe time
// (n_pad^2 x 16 x (plane slots + 2 .. 4) FMAs on one CU at 64 per clock) plus one to two microseconds of exchange, reduction
// and stage arithmetic.  Measured (tools/bench_small_sweeps.py, chains of three-level transmons / qubits in the frame of
// their static Hamiltonian, 4096 instances, us per RK4 stage, this kernel against the per-launch kernels of the same
// formulation): n = 27: 2.8 / 14.8; n = 64 (6 operators): 8.4 / 20.5; n = 81: 13.4 / 23.7; n = 128 (7): 27 / 33; n = 243: 60 / 84;
// n = 256 (8): 77 / 89.  With few workgroups the large sizes lose (n = 243, 256 instances: 55 against 19): the host takes this
// kernel above n_pad = 128 only when its workgroups fill the chip (midyn_rk4.inc: rk4_combine_sweep_one_launch).
// What did NOT matter, measured one by one: the look-ahead depth beyond a few steps, the coefficient / phase loads of a stage
// (cold, but a stage ahead), scalar loads in the loop.  What did: 64-bit vector address arithmetic in the kk step (vector
// integer instructions take issue cycles from the SIMD that feeds the matrix pipe).
#pragma once

namespace midyn {

struct CombineSweepArgs {
    const double* frags;      // the stack's COMBINE layout (midyn_combine.h): [entry][kk % 16][tile < 2][group][lane]
    const int* list_ptr;      // [n_pad / 32 + 1]
    const int* list_idx;
    const double2* stat;      // static operator [lda][lda] or nullptr
    int lda;
    int plane_col[16];          // (up to two groups per kind: combine_sweep_groups_ok)
    int n, n_pad, k, B;
    int splits;               // waves that share one row tile (1 or 2)
    const double* S;          // [B][R][k]
    long long inst_stride;    // R * k
    const double2* E;         // [R][n_pad] frame phases of the table times, or nullptr
    const int* rows;          // [nsteps][3] table rows of t, t + h / 2, t + h
    const double* hs;         // [nsteps]
    const int* save;          // [nsteps] slot of the saved states the step's result goes to (-1: none), or nullptr
    int nsteps;
    const double2* y0;        // [B | 1][n]
    int y0_shared;
    double2* out;             // [B][P][n]
    int P;
    double2* ybuf;            // RT == 2 (registers at their limit, stages of tens of microseconds): y and the RK4 accumulator of the
    double2* accbuf;          // workgroup's instances live in memory, [n_pad][ld], ybuf holding y0 at launch; RT == 1: unused
    double2* buf3;            // ... third state vector of the expm action
    int ld;
    // MODE 1, the expm ACTION of scipy_expm with magnus_order 1 (fixed_step_solvers.py:80-108,345-363; midyn_action.inc): the
    // solve as a flat list of series TERMS, one product each -- per term the table row of its step's generator, the scalars of
    //     w = pw + a G x;  acc += b w      (Taylor: a = h / (s j), b = 1;  Chebyshev: a = (1 | 2) h / rho, b = 2 J_j)
    // and flags: bit 0 the last term of a series (a repetition of the step), bit 1 ... of the step (frame phases, saved
    // state: slot + 1 in bits 8.., 0 = none), bit 2 Chebyshev recurrence (pw starts from phi_{j-2}); c = the factor of the
    // series that starts next (J_0 or 1).
    int nstage;
    const int* st_row;
    const int* st_flag;
    const double* st_a;
    const double* st_b;
    const double* st_c;
    double c_first;
};

// Operand look-ahead in kk steps: a step is RT (64 NQ + 42) pipe cycles of one wave against several hundred cycles from
// the L2 to a register.
constexpr int sweep_depth(int rt, int nq, int stat, int mode) {
    return rt == 2 ? (nq >= 2 || stat || mode ? 2 : 4) : (nq <= (mode ? 1 : 2) ? 8 : 4);   // (registers; two tiles: their steps are long anyway)
}

// MODE 0: fixed-step RK4 (four stages per step); MODE 1: expm action, Magnus order 1 (a stage = a series term).
template <int NRE4, int NIM4, int STAT, int RT, int MODE>
__global__ __launch_bounds__(512) void combine_sweep_kernel(const CombineSweepArgs a) {
    constexpr int NQ = NRE4 + NIM4, D = sweep_depth(RT, NQ, STAT, MODE);
    constexpr bool RE = NRE4 > 0 || (STAT & 1), IM = NIM4 > 0 || (STAT & 2);
    static_assert(combine_sweep_groups_ok(NRE4, NIM4) && (RT == 1 || RT == 2), "shape");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int np = a.n_pad;
    double2* const X0 = reinterpret_cast<double2*>(smem_raw);          // two copies of the phased stage input [n_pad][16]
    double* const red = reinterpret_cast<double*>(X0 + (size_t)2 * np * 16);   // splits == 2: partial sums of the second wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sp = wave % a.splits, wt = wave / a.splits;
    const int row0 = wt * 16 * RT;                       // this wave's rows: row0 + 16 t + lq + 4 r
    const int rg = row0 >> 5, t0 = (row0 >> 4) & 1;      // row group of the lists, first tile inside it
    const int lb = lane & 15, lq = lane >> 4;
    const int inst = blockIdx.x * 16 + lb;
    const bool live = inst < a.B;
    const int ic = live ? inst : a.B - 1;
    const double* __restrict__ Sb = a.S + (size_t)ic * a.inst_stride;
    // coefficient column of this lane's plane of group q.  Two tiles with three or four plane groups (256 registers, all in
    // use) read it from the kernel arguments where it is needed -- once per stage -- through an index the optimiser cannot see
    // through: held in registers across the stages it was the value hipcc parked in scratch.
    constexpr bool PC_REG = !(RT == 2 && NQ >= 3);
    int pc[PC_REG ? NQ : 1];
    if (PC_REG) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) pc[PC_REG ? q : 0] = a.plane_col[4 * q + lq];
    }
    auto pc_of = [&](const int q) {
        if (PC_REG) return pc[PC_REG ? q : 0];
        int idx = 4 * q + lq;
        asm volatile("" : "+v"(idx));
        return a.plane_col[idx];
    };

    int e0 = a.list_ptr[rg], e1 = a.list_ptr[rg + 1];
    if (a.splits > 1) {
        const int len = e1 - e0;
        e1 = e0 + (int)((long long)len * (sp + 1) / a.splits);
        e0 = e0 + (int)((long long)len * sp / a.splits);
    }
    const int steps = (e1 - e0) * 16;
    // (wave-uniform bases in scalar registers + one 32-bit lane offset: the loads take the scalar-base form and a step needs no
    // 64-bit vector address arithmetic -- vector integer work takes issue cycles from the matrix pipe of the SIMD)
    using gbytes = const __attribute__((address_space(1))) char*;      // (global, so that the loads are not flat ones)
    auto uniform_ptr = [](const void* p_) {
        const unsigned long long v = (unsigned long long)p_;
        return (gbytes)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                        (unsigned)__builtin_amdgcn_readfirstlane((int)v));
    };
    const gbytes frb = uniform_ptr(a.frags + (size_t)e0 * (16 * 2 * NQ * 64));
    // static rows of this lane by DPP (midyn_combine.h): lane lb fetches row lq + 4 (lb & 3) of tile (RT == 2 ? (lb >> 2) & 1 : t0),
    // plane lb >> 3
    const gbytes sbb = uniform_ptr(a.stat + (size_t)(rg * CMB_ROWS) * a.lda);
    const unsigned s_lane = ((unsigned)((lq + 4 * (lb & 3) + 16 * (RT == 2 ? ((lb >> 2) & 1) : t0)) * a.lda) * 2u + (unsigned)(lb >> 3)) * 8u;   // bytes

    // the state of this wave's rows.  MODE 0: y, acc of RK4; MODE 1: y = the accumulated result of the series, acc = phi_{j-1},
    // pw = phi_{j-2} (Chebyshev) / 0.  Two waves that split the list of a tile also split its ROWS for everything after the
    // contraction (register rows r < 2 / r >= 2): each sums the partner's partial results of its own rows and does their stage
    // arithmetic -- ~100 fp64 vector instructions per tile and stage that one wave did while its partner waited.
    const int own_lo = a.splits > 1 ? 2 * sp : 0, own_hi = a.splits > 1 ? 2 * sp + 2 : 4;
    constexpr int SR = RT == 1 ? 1 : 0;                  // RT == 2: the state lives in memory (ybuf / accbuf / buf3)
    const int first_row = MODE == 1 ? a.st_row[0] : a.rows[0];
    double2 y[RT][4], acc[SR ? RT : 1][4], pw[(SR && MODE == 1) ? RT : 1][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 16 * t + lq + 4 * r;
            y[t][r] = (row < a.n) ? a.y0[(a.y0_shared ? 0 : (size_t)ic * a.n) + row] : make_double2(0.0, 0.0);
            if constexpr (SR) acc[t][r] = y[t][r];
        }
    {
        const double2* Es = a.E ? a.E + (size_t)first_row * np : nullptr;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (RT == 1 && (r < own_lo || r >= own_hi)) continue;
                const int row = row0 + 16 * t + lq + 4 * r;
                const double2 v = Es ? cmul(Es[row], y[t][r]) : y[t][r];
                X0[(size_t)row * 16 + lb] = v;
                if (MODE == 1) {      // the first series starts from v: result c_first v, phi_0 = v
                    const double2 cv = make_double2(a.c_first * v.x, a.c_first * v.y);
                    if constexpr (SR) {
                        y[t][r] = cv;
                        acc[t][r] = v;
                        if constexpr (MODE == 1) pw[t][r] = make_double2(0.0, 0.0);
                    } else {
                        const size_t idx = (size_t)row * a.ld + blockIdx.x * 16 + lb;
                        a.ybuf[idx] = cv;
                        a.accbuf[idx] = v;
                        a.buf3[idx] = make_double2(0.0, 0.0);
                    }
                }
            }
    }
    __syncthreads();

    double afr[D][RT][NQ];
    double2 yv[D];
    double sv[D];
    d4 ore[RT], oim[RT];
    double cb[NQ], cbn[NQ];
    // What a stage needs from memory is requested ONE STAGE AHEAD (a stage of a 64-row system is 3 us): the coefficients and
    // frame phases of the next stage are loaded at the top of this one, the operator fragments of the first D - 1 kk steps by
    // the look-ahead of the last ones (the fragments are the same every stage).  No scalar loads inside the stage loop -- a
    // wait for an LDS read with a scalar load outstanding is a wait for everything (lgkmcnt counts both, scalar loads return
    // out of order): the per-stage scalars (table row, step size, save slot) of 64 stages sit in one register each, lane =
    // stage, refilled every 32 stages, and are read with v_readlane; the kk blocks of the wave's list likewise.
    const int nstage = MODE == 1 ? a.nstage : 4 * a.nsteps;
    int rowv = 0, savev = -1, kbv = 0;        // (savev: MODE 1 the flags word)
    double hv = 0.0, bv = 0.0, cv_ = 0.0;     // (hv: MODE 1 the scalar a)
    auto refill = [&](int base) {
        int i = base + lane;
        i = i < nstage ? i : nstage - 1;
        if (MODE == 1) {
            rowv = a.st_row[i];
            savev = a.st_flag[i];
            hv = a.st_a[i];
            bv = a.st_b[i];
            cv_ = a.st_c[i];
        } else {
            const int g = i & 3;
            rowv = a.rows[3 * (i >> 2) + (g == 0 ? 0 : (g == 3 ? 2 : 1))];
            hv = a.hs[i >> 2];
            savev = (g == 3 && a.save) ? a.save[i >> 2] : -1;
        }
    };
    auto lane_double = [&](double v, int l) {
        const long long b_ = __double_as_longlong(v);
        return __longlong_as_double((long long)(((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(b_ >> 32), l) << 32) |
                                                (unsigned)__builtin_amdgcn_readlane((int)b_, l)));
    };
    refill(0);
    {
        const int ne = e1 - e0;
        kbv = ne > 0 ? a.list_idx[e0 + (lane < ne ? lane : ne - 1)] : 0;      // (a share of a list has at most 16 entries)
    }
    auto kb_of = [&](int entry) { return __builtin_amdgcn_readlane(kbv, entry); };
    const int s_first = __builtin_amdgcn_readlane(rowv, 0);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int pcq = pc_of(q);
        cb[q] = pcq >= 0 ? Sb[(size_t)s_first * a.k + pcq] : 0.0;
    }
    double2 ec[RT][4], en[RT][4];
    if (RT == 1 && a.E) {
#pragma unroll
        for (int r = 0; r < 4; ++r) ec[0][r] = a.E[(size_t)s_first * np + row0 + lq + 4 * r];
    }
    const int kb_first = kb_of(0);
    // global operands of flat step `sf` (entry sf / 16 of this wave's share of the list) into slot b
    const unsigned f_lane = (unsigned)((RT == 2 ? 0 : t0) * NQ * 64 + lane) * 8u;      // bytes
    auto load_g = [&](int sf, int kk, int b) {
        const gbytes fn = frb + (unsigned)sf * (unsigned)(2 * NQ * 64 * 8);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < NQ; ++q) afr[b][t][q] = *(const __attribute__((address_space(1))) double*)(fn + f_lane + (unsigned)((t * NQ + q) * 64 * 8));
        if (STAT) sv[b] = *(const __attribute__((address_space(1))) double*)(sbb + (unsigned)kk * 16u + s_lane);
    };
    if (steps > 0) {
#pragma unroll
        for (int j = 0; j < D - 1; ++j) load_g(j, kb_first * 16 + j, j);
    }
    // all four stages of all steps run through ONE copy of the loops below (stage index 4 st + sg)
    for (int stage = 0; stage < nstage; ++stage) {
        const int sg = stage & 3;
        if ((stage & 31) == 0 && stage > 0) refill(stage);
        const int sl = stage & 31;
        const int srow = __builtin_amdgcn_readlane(rowv, sl), nrow = __builtin_amdgcn_readlane(rowv, sl + 1);
        const double h = lane_double(hv, sl);                  // MODE 1: a
        const int save_slot = __builtin_amdgcn_readlane(savev, sl);      // MODE 1: flags
        const double2* __restrict__ X = X0 + (size_t)(stage & 1) * np * 16;          // (RK4: a step starts on copy 0)
        double2* __restrict__ Xn = X0 + (size_t)((stage & 1) ^ 1) * np * 16;
        // (two tiles with three or four plane groups sit at the 256 registers of a 512-thread workgroup: they ask for the next
        // coefficients after the contraction, where they arrive behind the stage arithmetic, instead of holding them through it)
        constexpr bool CBN_LATE = RT == 2 && NQ >= 3;
        if (!CBN_LATE) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int pcq = pc_of(q);
                cbn[q] = pcq >= 0 ? Sb[(size_t)nrow * a.k + pcq] : 0.0;
            }
        }
        // frame phases of the next stage's rows (its input).  One tile per wave: loaded here, ahead of the contraction, and
        // kept for the result of the next stage; two tiles (n_pad > 128: stages of tens of microseconds, registers at their
        // limit): both sets after the contraction
        if (RT == 1 && a.E) {
#pragma unroll
            for (int r = 0; r < 4; ++r) en[0][r] = a.E[(size_t)nrow * np + row0 + lq + 4 * r];
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            ore[t] = d4{0.0, 0.0, 0.0, 0.0};
            oim[t] = d4{0.0, 0.0, 0.0, 0.0};
        }
        auto load_x = [&](int kk, int b) {
            yv[b] = X[(size_t)kk * 16 + lb];
            // (the loads stay HERE, D - 1 steps ahead of their use: left alone, the scheduler sinks them to their first use)
            __builtin_amdgcn_sched_barrier(0);
        };
        auto static_re = [&](int t, int b) {
            return (STAT & 1) ? (t == 0 ? d4{row_bcast<0>(sv[b]), row_bcast<1>(sv[b]), row_bcast<2>(sv[b]), row_bcast<3>(sv[b])}
                                        : d4{row_bcast<4>(sv[b]), row_bcast<5>(sv[b]), row_bcast<6>(sv[b]), row_bcast<7>(sv[b])})
                              : d4{0.0, 0.0, 0.0, 0.0};
        };
        auto static_im = [&](int t, int b) {
            return (STAT & 2) ? (t == 0 ? d4{row_bcast<8>(sv[b]), row_bcast<9>(sv[b]), row_bcast<10>(sv[b]), row_bcast<11>(sv[b])}
                                        : d4{row_bcast<12>(sv[b]), row_bcast<13>(sv[b]), row_bcast<14>(sv[b]), row_bcast<15>(sv[b])})
                              : d4{0.0, 0.0, 0.0, 0.0};
        };
        // COMBINE of one kk step (slot b) into g[..][p], APPLY of the step before from g[..][p ^ 1]: a wave with ONE tile has one
        // chain MFMA(s) -> vector FMAs per step, and in program order every step would wait for its own MFMA; software-pipelined
        // the FMAs of step s run behind the MFMAs of step s + 1 (PIPE; needs a look-ahead of more than one step)
        constexpr bool PIPE = D > 2;
        d4 gre[PIPE ? 2 : 1][RT], gim[PIPE ? 2 : 1][RT];
        auto combine = [&](int b, int p, int kinds) {      // kinds: bit 0 the real planes, bit 1 the imaginary ones
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                if (RE && (kinds & 1)) {
                    gre[p][t] = static_re(t, b);
#pragma unroll
                    for (int q = 0; q < NRE4; ++q) gre[p][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[b][t][q], cb[q], gre[p][t], 0, 0, 0);
                }
                if (IM && (kinds & 2)) {
                    gim[p][t] = static_im(t, b);
#pragma unroll
                    for (int q = 0; q < NIM4; ++q)
                        gim[p][t] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[b][t][NRE4 + q], cb[NRE4 + q], gim[p][t], 0, 0, 0);
                }
            }
        };
        auto apply = [&](int b, int p, int kinds) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RE && (kinds & 1)) {
                        ore[t][r] = fma(gre[p][t][r], yv[b].x, ore[t][r]);
                        oim[t][r] = fma(gre[p][t][r], yv[b].y, oim[t][r]);
                    }
                    if (IM && (kinds & 2)) {
                        ore[t][r] = fma(-gim[p][t][r], yv[b].y, ore[t][r]);
                        oim[t][r] = fma(gim[p][t][r], yv[b].x, oim[t][r]);
                    }
                }
        };
        if (steps > 0) {
            // the look-ahead of D - 1 steps wraps past the end of the list to its first steps: the operator fragments of the
            // NEXT stage (its state rows are re-read after the barrier)
#pragma unroll
            for (int j = 0; j < D - 1; ++j) load_x(kb_first * 16 + j, j);
            if (PIPE) combine(0, 0, 3);
            // (steps is a multiple of 16, D divides 16; the last D steps are a copy of the loop body of their own: only there can
            // the look-ahead pass the end, at positions known at compile time)
            auto body = [&](int s, bool last) {
#pragma unroll
                for (int j = 0; j < D; ++j) {
                    const int sfw = (last && j >= 1) ? j - 1 : s + j + D - 1;
                    const int kk = kb_of(sfw >> 4) * 16 + (sfw & 15);
                    load_g(sfw, kk, (j + D - 1) % D);
                    load_x(kk, (j + D - 1) % D);
                    if (PIPE) {
                        combine((j + 1) % D, (j + 1) & 1, 3);   // (after the last step: the first step of the next stage with this
                        apply(j, j & 1, 3);                     //  stage's coefficients -- discarded)
                    } else {                                    // two tiles, two or more plane groups: one plane kind at a time
                        combine(j, 0, 1);                       // (registers)
                        apply(j, 0, 1);
                        combine(j, 0, 2);
                        apply(j, 0, 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            for (int s = 0; s < steps - D; s += D) body(s, false);
            body(steps - D, true);
        }
        if (CBN_LATE) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int pcq = pc_of(q);
                cbn[q] = pcq >= 0 ? Sb[(size_t)nrow * a.k + pcq] : 0.0;
            }
        }
        if (a.splits > 1) {      // the partner's partial sums of this wave's rows (and this wave's of the partner's rows) through LDS
            double* slot = red + (size_t)wt * (RT * 8 * 64) + lane;
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RT == 2 || (r >= own_lo && r < own_hi)) continue;
                    slot[(t * 8 + r) * 64] = ore[t][r];
                    slot[(t * 8 + 4 + r) * 64] = oim[t][r];
                }
            __syncthreads();
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RT == 1 && (r < own_lo || r >= own_hi)) continue;
                    ore[t][r] += slot[(t * 8 + r) * 64];
                    oim[t][r] += slot[(t * 8 + 4 + r) * 64];
                }
        }
        {
            // MODE 0, RK4 stage arithmetic (fixed_step_solvers.py:43-77): acc' = (sg == 0 ? y : acc) + wa h k, input of the next
            // stage y + wc h k; the last stage: y' = acc + h k / 6, which is also the next input.
            // MODE 1, a series term (midyn_action.inc): w = pw + a G x joins the result with weight b and is the next x; at the end
            // of a series the result starts the next one -- at the end of a step through the frame phases of both steps.
            const double wa = (sg == 0 || sg == 3) ? h * (1.0 / 6) : h * (1.0 / 3);
            const double wc = sg == 2 ? h : 0.5 * h;
            const int flags = save_slot;
            const bool series_end = flags & 1, step_end = flags & 2, cheb = flags & 4;
            const int eslot = (int)((unsigned)flags >> 8) - 1;      // (bits 8..: save slot + 1, 0 = none)
            const double tb = MODE == 1 ? lane_double(bv, sl) : 0.0, tc = MODE == 1 ? lane_double(cv_, sl) : 0.0;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                double2 yy[4], aa[4], pp[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RT == 1 && (r < own_lo || r >= own_hi)) continue;
                    int row = row0 + 16 * t + lq + 4 * r;
                    if (RT == 2) asm volatile("" : "+v"(row));    // (no 64-bit address of a (row, array) pair is carried through the stage loop)
                    if (RT == 2 && a.E) {
                        ec[t][r] = a.E[(size_t)srow * np + row];
                        en[t][r] = a.E[(size_t)nrow * np + row];
                    }
                    if constexpr (SR) {
                        yy[r] = y[t][r];
                        aa[r] = acc[t][r];
                        if constexpr (MODE == 1) pp[r] = pw[t][r];
                    } else {
                        const size_t idx = (size_t)row * a.ld + blockIdx.x * 16 + lb;
                        yy[r] = a.ybuf[idx];
                        aa[r] = a.accbuf[idx];
                        if (MODE == 1) pp[r] = a.buf3[idx];
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (RT == 1 && (r < own_lo || r >= own_hi)) continue;
                    int row = row0 + 16 * t + lq + 4 * r;
                    if (RT == 2) asm volatile("" : "+v"(row));
                    const double2 o = make_double2(ore[t][r], oim[t][r]);
                    double2 cur;
                    if (MODE == 0) {
                        const double2 kv = a.E ? cmul_conj_a(ec[t][r], o) : o;
                        aa[r] = cfma_r(wa, kv, sg == 0 ? yy[r] : aa[r]);
                        cur = cfma_r(wc, kv, yy[r]);
                        if (sg == 3) {
                            yy[r] = aa[r];
                            cur = aa[r];
                        }
                        if (a.E) cur = cmul(en[t][r], cur);
                        if (save_slot >= 0 && live && row < a.n) a.out[((size_t)inst * a.P + save_slot) * a.n + row] = yy[r];
                    } else {
                        const double2 w = cfma_r(h, o, pp[r]);
                        double2 res = cfma_r(tb, w, yy[r]);
                        if (!series_end) {
                            cur = w;
                            pp[r] = cheb ? aa[r] : make_double2(0.0, 0.0);
                            aa[r] = w;
                            yy[r] = res;
                        } else {
                            if (step_end) {
                                if (a.E) res = cmul_conj_a(ec[t][r], res);
                                if (eslot >= 0 && live && row < a.n) a.out[((size_t)inst * a.P + eslot) * a.n + row] = res;
                                if (a.E) res = cmul(en[t][r], res);
                            }
                            cur = res;
                            aa[r] = res;
                            yy[r] = make_double2(tc * res.x, tc * res.y);
                            pp[r] = make_double2(0.0, 0.0);
                        }
                    }
                    Xn[(size_t)row * 16 + lb] = cur;
                    if constexpr (SR) {
                        y[t][r] = yy[r];
                        acc[t][r] = aa[r];
                        if constexpr (MODE == 1) pw[t][r] = pp[r];
                    } else {
                        const size_t idx = (size_t)row * a.ld + blockIdx.x * 16 + lb;
                        a.accbuf[idx] = aa[r];
                        if (MODE == 1 || sg == 3) a.ybuf[idx] = yy[r];
                        if (MODE == 1) a.buf3[idx] = pp[r];
                    }
                }
                // two tiles: one tile's state and phases at a time (left alone, the scheduler starts the loads of the second tile
                // before the arithmetic of the first and the larger variants park values in scratch)
                if (RT == 2) __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) cb[q] = cbn[q];
        if (RT == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) ec[0][r] = en[0][r];
        }
        __syncthreads();      // the next stage's input is complete (and nobody reads the copy it replaces any more)
    }
}

// ---- the instantiations that exist (see the end of midyn_combine.h): midyn_tu_combine_sweep{,_expm}.hip define them -------
#define MIDYN_SWEEP_FOR_RT(X, M_, ...) X(__VA_ARGS__, 1, M_) X(__VA_ARGS__, 2, M_)
#define MIDYN_X0(R_, I_, S_) MIDYN_SWEEP_FOR_RT(MIDYN_X, 0, R_, I_, S_)
#define MIDYN_X1(R_, I_, S_) MIDYN_SWEEP_FOR_RT(MIDYN_X, 1, R_, I_, S_)
#ifdef MIDYN_TU_COMBINE_SWEEP_RK4
#define MIDYN_X(R_, I_, S_, T_, M_) template __global__ void combine_sweep_kernel<R_, I_, S_, T_, M_>(const CombineSweepArgs);
#else
#define MIDYN_X(R_, I_, S_, T_, M_) extern template __global__ void combine_sweep_kernel<R_, I_, S_, T_, M_>(const CombineSweepArgs);
#endif
MIDYN_COMBINE_PAIRS_BOTH_KINDS(MIDYN_X0)
#undef MIDYN_X
#ifdef MIDYN_TU_COMBINE_SWEEP_EXPM
#define MIDYN_X(R_, I_, S_, T_, M_) template __global__ void combine_sweep_kernel<R_, I_, S_, T_, M_>(const CombineSweepArgs);
#else
#define MIDYN_X(R_, I_, S_, T_, M_) extern template __global__ void combine_sweep_kernel<R_, I_, S_, T_, M_>(const CombineSweepArgs);
#endif
MIDYN_COMBINE_PAIRS_BOTH_KINDS(MIDYN_X1)
#undef MIDYN_X
#undef MIDYN_X0
#undef MIDYN_X1

}  // namespace midyn
