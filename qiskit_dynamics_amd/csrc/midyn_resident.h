// midyn_resident.h -- rk4_resident_kernel: a single trajectory whose operators live in REGISTERS.
//
// One state column is launch-latency bound on the streaming route: a 10-qubit RHS evaluation reads 34 MB in 6.5 us
// and then pays a dependent kernel boundary, 8.4 us per stage, and reads the same operators again from HBM / MALL
// 4000 times per solve.  The active planes of such a stack fit the register files of the chip (256 CUs x 512 KB):
// here every wave owns ONE row of the generator and keeps that row of every active operator plane in registers for
// the whole solve (64 doubles per lane = a row of 8 single-plane operators over 512 non-zero columns).  A stage is
//   (1) the workgroup (4 rows) polls the pre-phased stage input y' of the columns its rows couple to straight out of
//       an exchange ring in device memory into LDS,
//   (2) every wave forms sum_e c_e(t) A_e[row, :] . y' from its registers, reduces across its lanes,
//   (3) lane 0 runs the fused RK4 stage of its row (frame phases, y / accumulator in registers) and publishes the
//       next stage input of its row.
// There is no grid barrier: the exchange is its own synchronisation.  Every 8-byte word of the ring starts as a
// sentinel (all ones: a NaN pattern arithmetic never produces) and a reader polls a word until it is not the
// sentinel -- one store and one load latency per stage instead of an atomic counter round trip on top.  Four ring
// buffers rotate; in round r a row's owner publishes into buffer r+1 and then re-arms its words of buffer r-1.  That
// is safe because the polled sets are SYMMETRIC by construction (whoever reads my rows is read by me, on 64 x 64
// granularity): having read round r from all my neighbours proves that every reader of my round r-1 words has
// finished with them.  The re-arming store is acknowledged before the owner publishes the NEXT round (a wait that
// costs nothing a round later), so a reader that has seen round r+2 can never see stale round r-1 data in the buffer
// it polls for round r+3.
//
// What this relies on beyond the HIP / LLVM memory model (relaxed agent-scope atomics order nothing by themselves), and
// where it is guarded: (1) a store is complete -- visible to every later agent-scope load on any XCD -- once `s_waitcnt
// vmcnt(0)` has let it through: agent-scope atomic stores are `sc1` write-through stores on gfx950 and the counter is
// decremented by their acknowledgement (MI355X_MICROARCH.md, inter-workgroup visibility); the wait is followed by a
// compiler fence so that hipcc cannot move a ring store above it.  The library is built for gfx950 only (no other code
// object exists in libmidyn.so), so no other memory system ever runs this.  (2) 8-byte words are never torn.  (3) A
// data word never equals the sentinel: the all-ones pattern is a NaN no arithmetic produces, but a y0 that CONTAINS it
// (or a wait that never ends for any other reason) makes a reader give up after `spin_limit` polls -- the host then
// re-runs the step range on the per-launch route, which propagates NaNs like NumPy does; nothing hangs, nothing fails.
//
// Precision: same products as the streaming kernel in a different summation order (lane-strided partial sums, then
// a butterfly), same fused stage arithmetic (apply_epilogue_t).
#pragma once

#ifndef MIDYN_RESIDENT_ABLATE
#define MIDYN_RESIDENT_ABLATE 0
#endif
#ifndef MIDYN_RESIDENT_PACK
#define MIDYN_RESIDENT_PACK 1     // publish a workgroup's rows with one store (0: lanes 0/1 of every wave; 2.58 vs 2.52 us)
#endif
#ifndef MIDYN_SWEEP_UNROLL
#define MIDYN_SWEEP_UNROLL 2     // slots per iteration of the pass loops of the sweep kernels
#endif
#ifndef MIDYN_SWEEP_PREFETCH
#define MIDYN_SWEEP_PREFETCH 2   // ell_sweep_kernel: register stages of operator elements fetched ahead of their slot (0: none)
#endif
#ifndef MIDYN_SWEEP_ABLATE
#define MIDYN_SWEEP_ABLATE 0     // 1: no operator pass; 2: element loads only (no gathers); 3: gathers only (no element loads)
#endif

namespace midyn {

constexpr unsigned long long RESIDENT_SENTINEL = 0xFFFFFFFFFFFFFFFFull;
constexpr int RESIDENT_DPL = 64;          // doubles of operator data per lane
constexpr int RESIDENT_MAX_POLL = 16;     // polled 64-column chunks per workgroup (8 words per thread)
constexpr unsigned RESIDENT_SPIN_LIMIT = 1u << 21;   // default of the polls a wait may take (ctx option resident_spin_limit)
#ifndef RESIDENT_MISS_STEP
#define RESIDENT_MISS_STEP 2       // pre-sleep units added after a round whose first poll missed ...
#endif
#ifndef RESIDENT_CLEAN_ROUNDS
#define RESIDENT_CLEAN_ROUNDS 8    // ... one unit removed after this many clean rounds in a row
#endif
constexpr int RESIDENT_INIT_PRESLEEP = 16, RESIDENT_MAX_PRESLEEP = 64;   // units of s_sleep(1) = 64 clocks

struct ResidentArgs {
    const double2* ops;       // [nseg][n_pad][n_pad]
    const int* pairs;         // [NE] entry e of every chunk: (segment << 1) | plane (0 real part, 1 imaginary part), -1 = unused
    int n, n_pad, has_static, k;
    const double* S;          // [R][k] coefficient table of the one instance
    const double2* E;         // [R][n_pad] frame phases or nullptr
    const int* rows;          // [nsteps][3]
    const double* hs;         // [nsteps]
    const int* save;          // [nsteps] output slot or -1, or nullptr
    int step_begin, step_end, nsteps;
    const int* chunk_ptr;     // [n_pad/16 + 1] operand chunks of every 16-row group ...
    const int* chunk_idx;     // ... as (slot in the poll list of the group's 64-row chunk) << 8 | chunk number
    const int* poll_ptr;      // [n_pad/64 + 1] chunks polled by the workgroups of a 64-row chunk ...
    const int* poll_idx;      // ... chunk numbers
    unsigned long long* ring; // [4][2 * n_pad] exchange ring (re, im interleaved), all sentinel at launch
    double2* y;               // [n_pad] state, updated in place
    double2* out;             // [P][n] saved states or nullptr
    int* err;                 // set when a wait gave up (results invalid: the host re-runs the range on the per-launch route)
    unsigned spin_limit;      // polls after which a wait gives up
    int protocol;             // ctx option exchange_protocol: 0 = the measured default, 1 = the conforming forms (see midyn_core.inc)
    int exchange_only;        // measurement only (ctx option resident_exchange_only): every round publishes and polls as
                              // usual but skips the row's arithmetic -- the store -> poll floor of this launch geometry
};

__device__ __forceinline__ double resident_lane_value(double v, int lane) {   // v of a compile-time lane as a wave-uniform value
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ double resident_dpp(double x) {   // x of the lane a DPP control selects (same row of 16)
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// sum over the 64 lanes, the same value (and summation order) in every lane: xor-butterfly inside the rows of 16
// on the DPP crossbar (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), then the four row sums
__device__ __forceinline__ double resident_wave_sum(double x) {
    x += resident_dpp<0xB1>(x);
    x += resident_dpp<0x4E>(x);
    x += resident_dpp<0x141>(x);
    x += resident_dpp<0x140>(x);
    return (resident_lane_value(x, 0) + resident_lane_value(x, 16)) + (resident_lane_value(x, 32) + resident_lane_value(x, 48));
}

template <int NE, int WAVES, bool HALFQ>
__global__ __launch_bounds__(64 * WAVES, 1) void rk4_resident_kernel(const ResidentArgs a) {
    constexpr int NQ = RESIDENT_DPL / NE;
    constexpr int NQ_USED = (HALFQ && NQ >= 2) ? NQ / 2 : NQ;   // chunk slots the arithmetic walks (host: nq <= NQ_USED)
    constexpr int THREADS = 64 * WAVES;                          // one row per wave: WAVES rows per workgroup
    constexpr int NPW = RESIDENT_MAX_POLL * 128 / THREADS;       // polled words per thread
    __shared__ __attribute__((aligned(16))) double ylds[2][RESIDENT_MAX_POLL * 128];
#if MIDYN_RESIDENT_PACK
    __shared__ __attribute__((aligned(16))) double2 pubs[2][WAVES];
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = blockIdx.x * WAVES + wave;           // this wave's row (wave-uniform: scalar loads and branches)
    const int grp = row >> 4, rchunk = row >> 6;
    const int n_pad = a.n_pad;
    unsigned long long* const ring = a.ring;

    // ---- operand chunks of this row group; the row of every (entry, chunk) into registers
    const int cbase = a.chunk_ptr[grp];
    const int nq = a.chunk_ptr[grp + 1] - cbase;
    double v[RESIDENT_DPL];
    // (LDS slot of every chunk, 4 bits each -- at most RESIDENT_MAX_POLL = 16 slots: 32 chunk slots in four words
    // instead of 32 registers, which is what made the NE = 2 full-walk instantiation spill)
    static_assert(RESIDENT_MAX_POLL <= 16, "chunk slots are packed in 4 bits");
    unsigned slot_pk[(NQ + 7) / 8];
#pragma unroll
    for (int w_ = 0; w_ < (NQ + 7) / 8; ++w_) slot_pk[w_] = 0u;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int packed = q < nq ? a.chunk_idx[cbase + q] : 0;
        slot_pk[q >> 3] |= (unsigned)(packed >> 8) << (4 * (q & 7));
        const int col = (packed & 0xff) * 64 + lane;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int pr = a.pairs[e];     // unused entries / chunks read element (0, row, lane) and drop it: no branch
            const double2 z = a.ops[((size_t)(pr >= 0 ? pr >> 1 : 0) * n_pad + row) * n_pad + col];
            v[q * NE + e] = (q < nq && pr >= 0) ? ((pr & 1) ? z.y : z.x) : 0.0;
        }
    }
    // entry e of the coefficient vectors lives in lane e
    const int my_pair = lane < NE ? a.pairs[lane] : -1;
    const int my_seg = my_pair >> 1;
    const bool my_static = a.has_static && my_seg == 0;
    const int my_cidx = my_seg - a.has_static;

    // ---- polled words of this thread: word i*256 + tid of the workgroup's poll list (128 words per chunk)
    const int pbase = a.poll_ptr[rchunk];
    const int npoll = a.poll_ptr[rchunk + 1] - pbase;
    const int npw = (npoll * 128 + THREADS - 1) / THREADS;
    int word_of[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int slot = (i * THREADS + tid) >> 7;
        word_of[i] = slot < npoll ? a.poll_idx[pbase + slot] * 128 + (tid & 127) : -1;
    }

    // unused chunk slots gather LDS slot 0: it must hold finite numbers even when this workgroup polls nothing
    for (int i = tid; i < 2 * RESIDENT_MAX_POLL * 128; i += THREADS) (&ylds[0][0])[i] = 0.0;
    __syncthreads();
    double2 yr = a.y[row];
    double2 acc_r = make_double2(0.0, 0.0);
    bool dead = false;
    if (lane == 0 && a.step_begin < a.step_end) {    // round 0 input: the state phased to the first stage time
        double2 y0 = yr;
        if (a.E) y0 = cmul(a.E[(size_t)a.rows[3 * a.step_begin] * n_pad + row], yr);
        __hip_atomic_store(ring + 2 * row, (unsigned long long)__double_as_longlong(y0.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ring + 2 * row + 1, (unsigned long long)__double_as_longlong(y0.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int presleep = RESIDENT_INIT_PRESLEEP, clean = 0;   // wave-uniform
    int rr = 0;     // round = stage counter of this launch; buffers rotate rr % 4
    int b_cur = 0;
    for (int st = a.step_begin; st < a.step_end; ++st) {
        const int r0 = a.rows[3 * st], r1 = a.rows[3 * st + 1], r2 = a.rows[3 * st + 2];
        const int rnext = (st + 1 < a.nsteps) ? a.rows[3 * (st + 1)] : r2;
        const double h = a.hs[st];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const int srow = sg == 0 ? r0 : (sg == 3 ? r2 : r1);
            const int nrow = sg == 0 ? r1 : (sg == 1 ? r1 : (sg == 2 ? r2 : rnext));
            // loads that do not depend on the exchange first: phases of this row, coefficients of this stage
            double2 e_cur = make_double2(1.0, 0.0), e_next = make_double2(1.0, 0.0);
            if (a.E) {
                e_cur = a.E[(size_t)srow * n_pad + row];
                e_next = a.E[(size_t)nrow * n_pad + row];
            }
            double cmine = 0.0;
            if (my_pair >= 0) cmine = my_static ? 1.0 : a.S[(size_t)srow * a.k + my_cidx];
            const double c_re = (my_pair >= 0 && !(my_pair & 1)) ? cmine : 0.0;
            const double c_im = (my_pair >= 0 && (my_pair & 1)) ? cmine : 0.0;
            // ---- (1) poll the stage input of the coupled columns into LDS
            const int b_nxt = (b_cur + 1) & 3, b_rearm = (b_cur + 3) & 3;
            {
                const unsigned long long* cur = ring + (size_t)b_cur * 2 * n_pad;
                unsigned long long w[NPW];
#pragma unroll
                for (int i = 0; i < NPW; ++i) w[i] = (i < npw && word_of[i] >= 0 && !dead) ? RESIDENT_SENTINEL : 0ull;
                unsigned spins = 0;
                // A poll that comes too early costs a whole load latency (and its traffic delays everybody's stores);
                // one that comes a little late costs that little.  So the wave sleeps before its first poll, for a
                // time it adapts: two units longer after a round whose first poll found a word missing, one unit
                // shorter after eight clean rounds in a row.
                for (int z = 0; z < presleep; ++z) __builtin_amdgcn_s_sleep(1);
                for (;;) {
                    // every outstanding word again, all loads in flight together (words that arrived keep their value)
                    unsigned long long f[NPW];
#pragma unroll
                    for (int i = 0; i < NPW; ++i)
                        if (i < npw) f[i] = a.protocol ? __hip_atomic_load(cur + (word_of[i] >= 0 ? word_of[i] : 0), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                                       : __hip_atomic_load(cur + (word_of[i] >= 0 ? word_of[i] : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool pending = false;
#pragma unroll
                    for (int i = 0; i < NPW; ++i)
                        if (i < npw) {
                            if (w[i] == RESIDENT_SENTINEL) w[i] = f[i];
                            pending |= (w[i] == RESIDENT_SENTINEL);
                        }
                    if (!pending) break;
                    __builtin_amdgcn_s_sleep(1);
                    ++spins;
                    if ((spins & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) spins = a.spin_limit;
                    if (spins >= a.spin_limit) {   // a neighbour never arrived: flag, stop waiting for good
                        __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        dead = true;
                        break;
                    }
                }
                if (__builtin_amdgcn_readfirstlane(__any(spins > 0) ? 1 : 0)) {
                    presleep = presleep + RESIDENT_MISS_STEP > RESIDENT_MAX_PRESLEEP ? RESIDENT_MAX_PRESLEEP : presleep + RESIDENT_MISS_STEP;
                    clean = 0;
                } else if (++clean == RESIDENT_CLEAN_ROUNDS) {
                    presleep = presleep > 0 ? presleep - 1 : 0;
                    clean = 0;
                }
#pragma unroll
                for (int i = 0; i < NPW; ++i)
                    if (i < npw) ylds[rr & 1][i * THREADS + tid] = __longlong_as_double((long long)w[i]);
            }
            __syncthreads();
            // ---- (2) this row of C(t) y': per chunk g = sum_e c_e A_e[row, col] (both parts), then g * y'[col]
            double cr[NE], ci[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                cr[e] = resident_lane_value(c_re, e);
                ci[e] = resident_lane_value(c_im, e);
            }
            double2 acc = make_double2(0.0, 0.0);
            // (the lane's offset passes through an empty asm per stage: otherwise the up to 32 LDS addresses of the chunk
            // slots are hoisted out of the step loop as loop invariants and spilled)
            int lane_o = 2 * lane;
            asm volatile("" : "+v"(lane_o));
            const double* yl = ylds[rr & 1] + lane_o;
#if MIDYN_RESIDENT_ABLATE == 1   // profiling only: the exchange without the arithmetic
            acc = *reinterpret_cast<const double2*>(yl);
#else
            // straight-line over the chunk slots (all NQ, or the first half -- HALFQ -- when at most half are in use): the registers
            // of unused chunks hold zeros and their LDS slot is 0 (a finite vector), so they contribute exact zeros;
            // two accumulators halve the dependent chain
            double2 acc_b = make_double2(0.0, 0.0);
#define MIDYN_RESIDENT_CHUNKS(LIMIT)                                                                     \
            _Pragma("unroll") for (int q = 0; q < (LIMIT); ++q) {                                        \
                double g_re = 0.0, g_im = 0.0;                                                           \
                _Pragma("unroll") for (int e = 0; e < NE; ++e) {                                         \
                    g_re = fma(cr[e], v[q * NE + e], g_re);                                              \
                    g_im = fma(ci[e], v[q * NE + e], g_im);                                              \
                }                                                                                        \
                const double2 yv = *reinterpret_cast<const double2*>(yl + ((slot_pk[q >> 3] >> (4 * (q & 7))) & 15u) * 128); \
                if (q & 1) {                                                                             \
                    acc_b.x = fma(g_re, yv.x, acc_b.x);                                                  \
                    acc_b.x = fma(-g_im, yv.y, acc_b.x);                                                 \
                    acc_b.y = fma(g_re, yv.y, acc_b.y);                                                  \
                    acc_b.y = fma(g_im, yv.x, acc_b.y);                                                  \
                } else {                                                                                 \
                    acc.x = fma(g_re, yv.x, acc.x);                                                      \
                    acc.x = fma(-g_im, yv.y, acc.x);                                                     \
                    acc.y = fma(g_re, yv.y, acc.y);                                                      \
                    acc.y = fma(g_im, yv.x, acc.y);                                                      \
                }                                                                                        \
            }
            if (a.exchange_only) {
                acc = *reinterpret_cast<const double2*>(yl);
            } else {
                MIDYN_RESIDENT_CHUNKS(NQ_USED)
            }
#undef MIDYN_RESIDENT_CHUNKS
            acc.x += acc_b.x;
            acc.y += acc_b.y;
#endif
            acc.x = resident_wave_sum(acc.x);
            acc.y = resident_wave_sum(acc.y);
            // ---- (3) the RK4 stage of this row (apply_epilogue_t's arithmetic), publish the next stage input
            const double2 kk = a.E ? cmul_conj_a(e_cur, acc) : acc;
            double2 pub;
            if (sg == 0) {
                acc_r = cfma_r(h * (1.0 / 6), kk, yr);
                pub = cfma_r(0.5 * h, kk, yr);
            } else if (sg == 1) {
                acc_r = cfma_r(h * (1.0 / 3), kk, acc_r);
                pub = cfma_r(0.5 * h, kk, yr);
            } else if (sg == 2) {
                acc_r = cfma_r(h * (1.0 / 3), kk, acc_r);
                pub = cfma_r(h, kk, yr);
            } else {
                yr = cfma_r(h * (1.0 / 6), kk, acc_r);
                pub = yr;
            }
            if (a.E) pub = cmul(e_next, pub);
#if MIDYN_RESIDENT_PACK
            // the workgroup's rows are consecutive: collect them in LDS and publish 16 * WAVES bytes with ONE store
            if (lane == 0) pubs[rr & 1][wave] = pub;
            __syncthreads();
            if (wave == 0 && lane < 2 * WAVES) {
                __builtin_amdgcn_s_waitcnt(0);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);   // (compiler: no ring store moves above the wait)
                const int row_wg = blockIdx.x * WAVES;
                unsigned long long* z = ring + (size_t)b_nxt * 2 * n_pad + 2 * row_wg + lane;
                // (exchange_protocol 1: the data word is a RELEASE store -- the re-arming store of the round before is ordered in front of
                // it by the memory model, not only by the wait above -- and the polls are ACQUIRE loads)
                if (a.protocol) __hip_atomic_store(z, (unsigned long long)__double_as_longlong(reinterpret_cast<const double*>(pubs[rr & 1])[lane]), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(z, (unsigned long long)__double_as_longlong(reinterpret_cast<const double*>(pubs[rr & 1])[lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long* zr = ring + (size_t)b_rearm * 2 * n_pad + 2 * row_wg + lane;
                __hip_atomic_store(zr, RESIDENT_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#else
            if (lane < 2) {   // every lane holds the same pub: lanes 0 / 1 store re / im in ONE 16-byte transaction
                // the re-arming stores of the PREVIOUS round are complete before this round's data leaves (they are a
                // round old: no stall); this round's re-arming follows the data
                __builtin_amdgcn_s_waitcnt(0);
                __atomic_signal_fence(__ATOMIC_SEQ_CST);   // (compiler: no ring store moves above the wait)
                unsigned long long* z = ring + (size_t)b_nxt * 2 * n_pad + 2 * row + lane;
                if (a.protocol) __hip_atomic_store(z, (unsigned long long)__double_as_longlong(lane ? pub.y : pub.x), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                else __hip_atomic_store(z, (unsigned long long)__double_as_longlong(lane ? pub.y : pub.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // re-arm this row's words of the buffer read LAST round: having read this round from all my neighbours
                // proves that every reader of those words has moved on (see the header)
                unsigned long long* zr = ring + (size_t)b_rearm * 2 * n_pad + 2 * row + lane;
                __hip_atomic_store(zr, RESIDENT_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#endif
            b_cur = b_nxt;
            ++rr;
        }
        if (a.save && a.out && lane == 0 && row < a.n) {
            const int slot = a.save[st];
            if (slot >= 0) a.out[(size_t)slot * a.n + row] = yr;
        }
    }
    if (lane == 0) a.y[row] = yr;
}


// ------------------------------------------------------------------------------------------------
// ell_resident_kernel: the same idea for VERY sparse operators (the vectorised Lindbladian of cfg 4 has at most 27
// non-zeros per row over 7 segments, N = 4096): one LANE per row.  A lane keeps the non-zero operator elements of
// its row -- (value of one plane, LDS index of the column, segment, plane) x at most ELL_W -- in registers; a
// workgroup is 64 consecutive rows (one 64-row chunk, whose symmetric poll list it uses): poll the coupled chunks
// into LDS, gather y'[col] per entry (no cross-lane reduction: a row is a lane), run the row's stage, publish 64
// rows with one coalesced store.  MODE 0: RK4 stages.  MODE 1: the expm
// action of Magnus order 1 (csrc/midyn_action.inc) -- per step the Chebyshev series, phi_{k+1} = 2 (h/rho) G phi_k +
// phi_{k-1}, or the scaled Taylor series, one round per term, the accumulator and the recurrence vectors in
// registers, every step of the solve in ONE launch.
// ------------------------------------------------------------------------------------------------
constexpr int ELL_W = 64;       // non-zero plane values per row at most
constexpr int ELL_WAVES = 4;    // waves per 64-row workgroup

struct EllArgs {
    const double* val;        // [wmax][n_pad] entry e of row r: value of one plane of one operator element
    const int* meta;          // [wmax][n_pad] LDS index of the column (bits 0-15) | segment << 16 | plane << 22 | valid << 23
    int wmax;
    int n, n_pad, has_static, k, nseg;
    const double* S;          // [R][k]
    const double2* E;         // [R][n_pad] or nullptr
    const double2* Dt;        // ell_sweep_kernel, order 2, framed: [nsteps][n_pad] E(t2) o conj(E(t1)) of every step
    double2* stash;           // ell_sweep_kernel, order 2: [B][2][n_pad] series vectors kept out of the registers
    const int* rows;          // [nsteps][3]
    const double* hs;         // [nsteps]
    const int* save;          // [nsteps] or nullptr
    int step_begin, step_end, nsteps;
    const int* poll_ptr;      // [n_pad/64 + 1]
    const int* poll_idx;
    unsigned long long* ring; // [4][2 * n_pad]
    double2* y;               // [n_pad]
    double2* out;             // [P][n] or nullptr
    int* err;
    unsigned spin_limit;      // polls after which a wait gives up
    int protocol;             // ctx option exchange_protocol: 0 = the measured default, 1 = the conforming forms (see midyn_core.inc)
    // MODE 1: per step K > 0 terms of the Chebyshev series (or -K = the Taylor degree), the repetitions, h / rho (or
    // h / scaling) and the Bessel coefficients J_0..J_K
    const int* cheb_K;
    const int* cheb_reps;
    const double* cheb_hrho;
    const double* cheb_coef;
    int cheb_stride;
};

// rows of the ELL arrays from the block lists: one wave per row walks the listed (segment, 16-column block) entries of
// its 16-row group four at a time; PASS 0 counts the non-zero plane values, PASS 1 writes them in that order
template <int PASS>
__global__ __launch_bounds__(256) void ell_build_kernel(const double2* __restrict__ ops, int n_pad, const int* __restrict__ blk_ptr,
                                                        const int* __restrict__ blk_idx, const int* __restrict__ slot_map,
                                                        int* __restrict__ counts, double* __restrict__ val,
                                                        int* __restrict__ meta, int* __restrict__ meta_col, int wmax) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = row >> 4, nc = n_pad >> 6;
    const int j0 = blk_ptr[g], j1 = blk_ptr[g + 1];
    int cnt = 0;
    for (int j = j0; j < j1; j += 4) {
        const int je = j + (lane >> 4);
        double2 z = make_double2(0.0, 0.0);
        int seg = 0, col = 0;
        if (je < j1) {
            const int entry = blk_idx[je];
            seg = entry >> 16;
            col = (entry & 0xffff) * 16 + (lane & 15);
            z = ops[((size_t)seg * n_pad + row) * n_pad + col];
        }
        const unsigned long long m_re = __ballot(z.x != 0.0), m_im = __ballot(z.y != 0.0);
        if (PASS == 1) {
            const unsigned long long below = (1ull << lane) - 1ull;
            const int slot = slot_map[(row >> 6) * nc + (col >> 6)];
            const int tag = (seg << 16) | (1 << 23);
            const int base = (slot * 64 + (col & 63)) | tag;     // column as an index into the polled chunks (resident kernel)
            const int gcol = col | tag;                          // column itself (source of the sweep kernel's arrays)
            if (z.x != 0.0) {
                const int pos = cnt + __popcll(m_re & below);
                if (pos < wmax) {
                    val[(size_t)pos * n_pad + row] = z.x;
                    meta[(size_t)pos * n_pad + row] = base;
                    meta_col[(size_t)pos * n_pad + row] = gcol;
                }
            }
            if (z.y != 0.0) {
                const int pos = cnt + __popcll(m_re) + __popcll(m_im & below);
                if (pos < wmax) {
                    val[(size_t)pos * n_pad + row] = z.y;
                    meta[(size_t)pos * n_pad + row] = base | (1 << 22);
                    meta_col[(size_t)pos * n_pad + row] = gcol | (1 << 22);
                }
            }
        }
        cnt += __popcll(m_re) + __popcll(m_im);
    }
    if (PASS == 0 && lane == 0) counts[row] = cnt;
}

template <int MODE, int EWU>
__global__ __launch_bounds__(64 * ELL_WAVES, 1) void ell_resident_kernel(const EllArgs a) {
    // One workgroup = 64 consecutive rows (lane = row) x ELL_WAVES waves.  A single wave polling 26 words per lane and
    // walking 27 entries is latency bound on its own instruction stream (measured 6.9 us per round at N = 4096);
    // so the waves share the round: each polls a quarter of the words, each owns every ELL_WAVES-th entry of every
    // row, the partial sums meet in LDS and wave 0 owns the rows' state, runs the stage and publishes.
    constexpr int THREADS = 64 * ELL_WAVES;
    constexpr int NPW = RESIDENT_MAX_POLL * 128 / THREADS;   // polled words per thread
    constexpr int EW = ELL_W / ELL_WAVES;                    // entries per lane at most (EWU of them are walked)
    static_assert(EWU <= ELL_W / ELL_WAVES, "EWU");
    __shared__ __attribute__((aligned(16))) double ylds[RESIDENT_MAX_POLL * 128];
    __shared__ __attribute__((aligned(16))) double2 part[ELL_WAVES][64];
    __shared__ double cl[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rchunk = blockIdx.x, row = rchunk * 64 + lane;
    const int n_pad = a.n_pad;
    unsigned long long* const ring = a.ring;
    double val[EW];
    int meta[EW];
#pragma unroll
    for (int j = 0; j < EW; ++j) {
        const int e = j * ELL_WAVES + wave;
        val[j] = e < a.wmax ? a.val[(size_t)e * n_pad + row] : 0.0;
        meta[j] = e < a.wmax ? a.meta[(size_t)e * n_pad + row] : 0;
        if (!(meta[j] & (1 << 23))) { val[j] = 0.0; meta[j] = 0; }
    }
    const int my_entries = (a.wmax - wave + ELL_WAVES - 1) / ELL_WAVES;   // entries e = j * ELL_WAVES + wave < wmax
    const int pbase = a.poll_ptr[rchunk];
    const int npoll = a.poll_ptr[rchunk + 1] - pbase;
    const int npw = (npoll * 128 + THREADS - 1) / THREADS;
    int word_of[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int slot = (i * THREADS + tid) >> 7;
        word_of[i] = slot < npoll ? a.poll_idx[pbase + slot] * 128 + (tid & 127) : -1;
    }
    const bool cstatic = a.has_static && tid == 0;
    const int cidx = tid - a.has_static;

    // unused entry slots gather LDS word 0: it must hold finite numbers even when this workgroup polls nothing
    for (int i = tid; i < RESIDENT_MAX_POLL * 128; i += THREADS) ylds[i] = 0.0;
    __syncthreads();
    double2 yr = a.y[row];     // (state only meaningful in wave 0)
    bool dead = false;
    int presleep = RESIDENT_INIT_PRESLEEP, clean = 0;
    int b_cur = 0;

    // one round: poll buffer b_cur into LDS, C = (sum_seg c_seg A_seg y')[row] (complete in wave 0)
    auto product = [&]() -> double2 {
        const unsigned long long* cur = ring + (size_t)b_cur * 2 * n_pad;
        for (int z = 0; z < presleep; ++z) __builtin_amdgcn_s_sleep(1);
        unsigned long long w[NPW];
#pragma unroll
        for (int i = 0; i < NPW; ++i) w[i] = (i < npw && word_of[i] >= 0 && !dead) ? RESIDENT_SENTINEL : 0ull;
        unsigned spins = 0;
        for (;;) {
            unsigned long long f[NPW];
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                if (i < npw) f[i] = a.protocol ? __hip_atomic_load(cur + (word_of[i] >= 0 ? word_of[i] : 0), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                               : __hip_atomic_load(cur + (word_of[i] >= 0 ? word_of[i] : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool pending = false;
#pragma unroll
            for (int i = 0; i < NPW; ++i)
                if (i < npw) {
                    if (w[i] == RESIDENT_SENTINEL) w[i] = f[i];
                    pending |= (w[i] == RESIDENT_SENTINEL);
                }
            if (!pending) break;
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 1023u) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) spins = a.spin_limit;
            if (spins >= a.spin_limit) {
                __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dead = true;
                break;
            }
        }
        if (__builtin_amdgcn_readfirstlane(__any(spins > 0) ? 1 : 0)) {
            presleep = presleep + RESIDENT_MISS_STEP > RESIDENT_MAX_PRESLEEP ? RESIDENT_MAX_PRESLEEP : presleep + RESIDENT_MISS_STEP;
            clean = 0;
        } else if (++clean == RESIDENT_CLEAN_ROUNDS) {
            presleep = presleep > 0 ? presleep - 1 : 0;
            clean = 0;
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i)
            if (i < npw) ylds[i * THREADS + tid] = __longlong_as_double((long long)w[i]);
        __syncthreads();
        // this wave's entries of every row; MODE 1 holds val[] already multiplied by its segment's coefficient
        // (constant within a step, see load_weighted)
        // straight-line over the first EWU entry slots (the host picks EWU >= the entries any lane holds; unused slots
        // hold zeros and gather LDS word 0), two accumulators
        double2 acc = make_double2(0.0, 0.0), acc_b = make_double2(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < EWU; ++j) {
            const int mt = meta[j];
            const double wgt = MODE == 1 ? val[j] : cl[(mt >> 16) & 63] * val[j];
            const double2 yv = *reinterpret_cast<const double2*>(ylds + 2 * (mt & 0xffff));
            const bool im = (mt >> 22) & 1;
            if (j & 1) {
                acc_b.x = fma(wgt, im ? -yv.y : yv.x, acc_b.x);
                acc_b.y = fma(wgt, im ? yv.x : yv.y, acc_b.y);
            } else {
                acc.x = fma(wgt, im ? -yv.y : yv.x, acc.x);
                acc.y = fma(wgt, im ? yv.x : yv.y, acc.y);
            }
        }
        acc.x += acc_b.x;
        acc.y += acc_b.y;
        part[wave][lane] = acc;
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w2 = 1; w2 < ELL_WAVES; ++w2) {
                const double2 p2 = part[w2][lane];
                acc.x += p2.x;
                acc.y += p2.y;
            }
        }
        return acc;
    };
    // wave 0: publish this row's next input into buffer b_cur + 1, re-arm its words of buffer b_cur - 1; all: advance
    auto publish = [&](double2 pub) {
        const int b_nxt = (b_cur + 1) & 3, b_rearm = (b_cur + 3) & 3;
        if (wave == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            __atomic_signal_fence(__ATOMIC_SEQ_CST);   // (compiler: no ring store moves above the wait)
            unsigned long long* z = ring + (size_t)b_nxt * 2 * n_pad + 2 * row;
            if (a.protocol) {       // (exchange_protocol 1: RELEASE stores / ACQUIRE polls, see rk4_resident_kernel)
                __hip_atomic_store(z, (unsigned long long)__double_as_longlong(pub.x), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(z + 1, (unsigned long long)__double_as_longlong(pub.y), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(z, (unsigned long long)__double_as_longlong(pub.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(z + 1, (unsigned long long)__double_as_longlong(pub.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            unsigned long long* zr = ring + (size_t)b_rearm * 2 * n_pad + 2 * row;
            __hip_atomic_store(zr, RESIDENT_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(zr + 1, RESIDENT_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        b_cur = b_nxt;
    };
    // coefficients of one table row into LDS; the reads of the previous round finished at its second barrier, the
    // barrier of the next poll publishes the new values
    auto set_coefficients = [&](int trow) {
        if (tid < a.nseg) cl[tid] = cstatic ? 1.0 : a.S[(size_t)trow * a.k + cidx];
    };
    // MODE 1: the coefficients are those of ONE time for every product of a step: fold them into the values
    auto load_weighted = [&](int trow) {
        set_coefficients(trow);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EW; ++j) {
            const int e = j * ELL_WAVES + wave;
            if (j < my_entries) val[j] = (meta[j] & (1 << 23)) ? a.val[(size_t)e * n_pad + row] * cl[(meta[j] >> 16) & 63] : 0.0;
        }
        __syncthreads();
    };

    if (wave == 0 && a.step_begin < a.step_end) {    // round 0 input: the state phased to the first product's time
        double2 y0 = yr;
        if (a.E) y0 = cmul(a.E[(size_t)a.rows[3 * a.step_begin] * n_pad + row], yr);
        __hip_atomic_store(ring + 2 * row, (unsigned long long)__double_as_longlong(y0.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ring + 2 * row + 1, (unsigned long long)__double_as_longlong(y0.y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (int st = a.step_begin; st < a.step_end; ++st) {
        const int r0 = a.rows[3 * st], r1 = a.rows[3 * st + 1], r2 = a.rows[3 * st + 2];
        const int rnext = (st + 1 < a.nsteps) ? a.rows[3 * (st + 1)] : (MODE == 0 ? r2 : r0);
        const double h = a.hs[st];
        if (MODE == 0) {
            double2 acc_r = make_double2(0.0, 0.0);
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const int srow = sg == 0 ? r0 : (sg == 3 ? r2 : r1);
                const int nrow = sg == 0 ? r1 : (sg == 1 ? r1 : (sg == 2 ? r2 : rnext));
                double2 e_cur = make_double2(1.0, 0.0), e_next = make_double2(1.0, 0.0);
                if (a.E && wave == 0) {
                    e_cur = a.E[(size_t)srow * n_pad + row];
                    e_next = a.E[(size_t)nrow * n_pad + row];
                }
                set_coefficients(srow);
                const double2 c = product();
                const double2 kk = a.E ? cmul_conj_a(e_cur, c) : c;
                double2 pub;
                if (sg == 0) {
                    acc_r = cfma_r(h * (1.0 / 6), kk, yr);
                    pub = cfma_r(0.5 * h, kk, yr);
                } else if (sg == 1) {
                    acc_r = cfma_r(h * (1.0 / 3), kk, acc_r);
                    pub = cfma_r(0.5 * h, kk, yr);
                } else if (sg == 2) {
                    acc_r = cfma_r(h * (1.0 / 3), kk, acc_r);
                    pub = cfma_r(h, kk, yr);
                } else {
                    yr = cfma_r(h * (1.0 / 6), kk, acc_r);
                    pub = yr;
                }
                if (a.E) pub = cmul(e_next, pub);
                publish(pub);
            }
        } else {
            // K > 0: Chebyshev series with K terms (hr = h / rho, coef = J_0..J_K); K < 0: Taylor series of degree -K
            // (hr = h / scaling); `reps` repetitions of the series per step
            const int Ks = a.cheb_K[st], reps = a.cheb_reps[st];
            const bool cheb = Ks > 0;
            const int K = cheb ? Ks : -Ks;
            const double hr = a.cheb_hrho[st];
            const double* coef = a.cheb_coef + (size_t)st * a.cheb_stride;
            double2 e_ph = make_double2(1.0, 0.0), e_nx = make_double2(1.0, 0.0);
            if (a.E && wave == 0) {
                e_ph = a.E[(size_t)r0 * n_pad + row];
                e_nx = a.E[(size_t)rnext * n_pad + row];
            }
            load_weighted(r0);
            for (int rep = 0; rep < reps; ++rep) {
                const double c0 = cheb ? coef[0] : 1.0;
                double2 accv = make_double2(c0 * yr.x, c0 * yr.y);
                double2 phi_prev = make_double2(0.0, 0.0), phi_cur = yr;
                for (int kt = 0; kt < K; ++kt) {
                    const double2 c = product();
                    const double2 kk = a.E ? cmul_conj_a(e_ph, c) : c;
                    double2 term;
                    if (cheb) {   // phi_{k+1} = (1 | 2) (h / rho) G phi_k + phi_{k-1};  acc += 2 J_{k+1} phi_{k+1}
                        const double alpha = (kt == 0 ? 1.0 : 2.0) * hr;
                        term = make_double2(alpha * kk.x + phi_prev.x, alpha * kk.y + phi_prev.y);
                        accv = cfma_r(2.0 * coef[kt + 1], term, accv);
                    } else {      // term_j = (h / (s j)) G term_{j-1};  acc += term_j   (EPI_TAYLOR's arithmetic)
                        const double hj = hr / (double)(kt + 1);
                        term = make_double2(hj * kk.x, hj * kk.y);
                        accv = make_double2(accv.x + term.x, accv.y + term.y);
                    }
                    phi_prev = phi_cur;
                    phi_cur = term;
                    double2 pub = term;
                    if (kt == K - 1) {
                        yr = accv;
                        pub = a.E ? cmul(rep + 1 < reps ? e_ph : e_nx, yr) : yr;
                    } else if (a.E) {
                        pub = cmul(e_ph, term);
                    }
                    publish(pub);
                }
            }
        }
        if (wave == 0 && a.save && a.out && row < a.n) {
            const int slot = a.save[st];
            if (slot >= 0) a.out[(size_t)slot * a.n + row] = yr;
        }
    }
    if (wave == 0) a.y[row] = yr;
}


// ------------------------------------------------------------------------------------------------
// ell_sweep_kernel<ORDER, RPT, TH, PACKED>: a SWEEP on a very sparse stack (cfg 5: n = 4096, at most 19 non-zeros per
// row), expm action of Magnus order 1 / 2.  Trajectories of a sweep are independent, so nothing has to cross workgroups
// at all: one workgroup (TH threads, RPT rows each, n_pad = TH RPT) integrates ONE instance through ALL steps.  The
// vector an operator is applied to is staged in LDS (two copies for order 2: the two Gauss points see different frame
// phases); every thread walks the operator elements of its rows (coalesced over the threads, served by L2: the arrays
// are shared by all instances) and gathers the operands from LDS.  Order 2, per term (commutator-free form,
// csrc/midyn_action.inc):
//     u1 = g1 v, u2 = g2 v (one pass: same v, two coefficient sets);  q = g2 u1 - g1 u2 (second pass);
//     w = a (u1 + u2) + b q  (+ phi_{j-2} for the Chebyshev recurrence).
// The MFMA work-list route multiplies 16 x 16 blocks that are 94 % zeros for such operators (17 tiles x 16 columns
// per row against 19 non-zeros).
//
// Round 3 re-formulated the kernel around what its passes wait for (tools/sweep_probe.hip: 39.0 -> 18.1 us per term on
// the cfg 5 shape, order 1: 14.1 -> 5.0).  A pass of the first form moved, per workgroup at n = 4096 with 19 slots,
// 934 KB of operator elements (4 B column + 8 B value) through the CU's 64 B/clk L1 fill path, 2.5 MB of LDS gathers,
// the frame phases of every row six times per term, and spilled 76 registers around its loops.
//   * PACKED 1: when every slot holds ONE magnitude (operators built from Pauli strings: the XX couplings and the drives
//     of cfg 5) an element is 4 bytes, column | sign << 31; the magnitude is folded into the slot's coefficient (one LDS
//     broadcast per slot) and the sign is an XOR into the gathered operand's sign bits.  Unused entries point at a zero
//     slot behind the LDS copies.  A third of the element bytes, no per-element v_mul_f64.
//     PACKED 2: every slot also has ONE sign and no unused entry: the element is the LDS byte address of its operand,
//     and the per-element work is the four fused multiply-adds and nothing else.  PACKED 0: any stack (12-byte elements).
//   * The whole step runs in the frame picture of its first Gauss point: y~ = E(t1) o y once per step, then
//     g1~ = C(t1) needs no phases at all and g2~ = conj(D) o C(t2) o D with D = E(t2) o conj(E(t1)) from a per-step table
//     (order 2 only; sweep_dtable_kernel) -- three loads and four complex multiplications per row and term instead of six
//     phase loads and six multiplications.
//   * The term vector is accumulated INTO the Chebyshev predecessor (w = phi_{j-2} + ...): one vector less; at order 2 the
//     vectors no pass touches (result, phi_{j-1}) live in a per-instance stash in device memory and the term between its
//     two passes rides through the second pass inside its second sum, so that a pass holds its two output vectors and its
//     gathers in flight and nothing else: no spills.
//   * All global addresses are a uniform base plus a 32-bit byte offset that passes through an empty asm at every use;
//     otherwise hipcc hoists the 64-bit address of every (array, row) pair out of the step loop and spills them.
// ------------------------------------------------------------------------------------------------
#ifndef MIDYN_CONST_AS
#define MIDYN_CONST_AS __attribute__((address_space(4)))
#endif
constexpr int SWEEP_THREADS = 1024;
constexpr int SWEEP_MAX_RPT = 4;      // rows per thread (template parameter RPT): n_pad = 1024 * RPT <= 4096
constexpr int SWEEP_MAX_SLOTS = 256;  // grouped slots per row at most

struct SweepArgs {
    // the operator elements grouped by (segment, plane): slot e of EVERY row belongs to the pair tags[e], so the
    // coefficient and the plane are wave-uniform per slot
    const double* val;        // [wsp][n_pad] (0 in unused slots)
    const int* col;           // [wsp][n_pad] column (0 in unused slots)
    const int* tags;          // [wsp] segment | plane << 8
    // packed form (every slot holds ONE magnitude: Pauli-built operators): 4 bytes per element instead of 12 --
    // pk[e][r] = column | sign << 31 (column n_pad = the zero slot of the LDS copies: unused entry), value = +-mag[e]
    const int* pk;            // [wsp][n_pad] or nullptr  (direct form: LDS byte address of the X1 operand, mag signed)
    const double* mag;        // [wsp]
    int wsp;
    int wre;                  // slots [0, wre) hold real-plane values, [wre, wsp) imaginary-plane values
    int wre_loc, wim_loc;     // two workgroups per instance (ell_sweep_duo_kernel): slots [0, wre_loc) and [wre, wim_loc) reference
                              // columns in the row's OWN half of the vector only (stack_ell_layout orders them first)
    int n, n_pad, has_static, k, nseg;
    const double* S;          // [B][R][k]
    long long inst_stride;    // R * k
    const double2* E;         // [R][n_pad] or nullptr
    const double2* Dt;        // ell_sweep_kernel, order 2, framed: [nsteps][n_pad] E(t2) o conj(E(t1)) of every step
    double2* stash;           // ell_sweep_kernel, order 2: [B][2][n_pad] series vectors kept out of the registers
    const int* rows;          // [nsteps][3]
    const double* hs;         // [nsteps]
    const int* save;          // [nsteps] or nullptr
    int nsteps;
    const int* ser_K;         // per step: > 0 Chebyshev terms, < 0 -(Taylor degree)
    const int* ser_reps;      // repetitions of the series (Taylor: the scaling s)
    const double* ser_par;    // Chebyshev: rho; Taylor: s
    const double* coef;       // [nsteps][stride] Bessel coefficients (Chebyshev steps)
    int stride;
    const double2* y0;        // [B | 1][n]
    int y0_shared;
    double2* out;             // [B][P][n] saved states
    int P;
};

// D[st][r] = E[rows[3 st + 1]][r] o conj(E[rows[3 st]][r]): the frame phase between the two Gauss points of every step
MIDYN_GLOBAL __launch_bounds__(256) void sweep_dtable_kernel(const double2* __restrict__ E, const int* __restrict__ rows, int nsteps, int np,
                                                           double2* __restrict__ Dt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nsteps * np) return;
    const int st = (int)(i / np), r = (int)(i % np);
    Dt[i] = cmul_conj_a(E[(size_t)rows[3 * st] * np + r], E[(size_t)rows[3 * st + 1] * np + r]);
}

// The 32-bit byte offset of row tid + TH i in an array of 2^shift-byte elements, passed through an empty asm at every
// use: otherwise the loop-invariant 64-bit sum base + offset of every (array, row) pair is hoisted out of the step loop
// -- 40 registers of addresses, most of them spilled -- and the loads lose the scalar-base addressing form.
template <int TH>
__device__ __forceinline__ unsigned sweep_boff(const int tid, const int i_, const int shift) {
    unsigned o = (unsigned)(tid + TH * i_) << shift;
    asm volatile("" : "+v"(o));
    return o;
}

// One pass over the operator elements of this thread's rows (rows tid + TH i):
//   o1 = (sum_e ca_e A_e) . X1,  o2 = (sum_e cb_e A_e) . X2   with (ca, cb) = cab[e], or swapped
// Two straight-line loops, no selects: the real-plane slots (A x = v x), then the imaginary-plane slots
// (A = i v: A x = v (-x.y, x.x)).  X1 / X2: the LDS copies of the operand vectors (sweep_lds: their base; PACKED 2
// elements are byte addresses relative to it, X2 operands 32768 bytes behind their X1 operands).
// PART (two workgroups per instance): 0 all slots; 1 the slots whose operands lie in the workgroup's own half (the sums
// start here); 2 the remaining slots (the sums continue).  row0: first row of the workgroup (rows row0 + tid + TH i).
template <int ORDER, int SWEEP_RPT, int TH, int PACKED, int PFD = MIDYN_SWEEP_PREFETCH, bool KEEP_O2 = false, int PART = 0>
__device__ __forceinline__ void sweep_pass(const SweepArgs& a, const double2* cab, const double2* sweep_lds, const double2* X1,
                                           const double2* X2, const int tid, const bool swapped, double2 (&o1)[SWEEP_RPT],
                                           double2 (&o2)[SWEEP_RPT], const double scale2 = 1.0,   // scale2: factor of the second sum
                                           const unsigned row0 = 0, const int re_lo = 0, const int re_hi = 0, const int im_lo = 0,
                                           const int im_hi = 0, const int flip_lane = 0) {      // PART 3: the real-plane slots [re_lo, re_hi) and the imaginary-
                                                                        // plane slots [im_lo, im_hi); the sums continue
    const int np = a.n_pad;
    const unsigned unp = (unsigned)np;
    auto boff = [&](const int i_, const int shift) { return sweep_boff<TH>(tid, i_, shift); };
#define ROW(i_) (row0 + (unsigned)(tid + TH * (i_)))

    if (PART != 2 && PART != 3) {
#pragma unroll
        for (int i = 0; i < SWEEP_RPT; ++i) {
            o1[i] = make_double2(0.0, 0.0);
            if (!KEEP_O2) o2[i] = make_double2(0.0, 0.0);     // KEEP_O2: the caller has put a start value into the second sum
        }
    }
#if MIDYN_SWEEP_ABLATE == 1   // profiling only: no operator pass at all
    for (int i = 0; i < SWEEP_RPT; ++i) { o1[i] = X1[tid + TH * i]; o2[i] = X2[tid + TH * i]; }
    return;
#endif
    // The element loads depend on nothing, but a wave that asks for them when it needs them waits out an L2 round
    // trip per slot.  A ring of PF register stages: stage s holds the elements of slot e0 + s and is refilled with
    // those of slot e0 + s + PF as soon as it has been copied out (the slot loop is unrolled by PF).
    // (measured, us per term, cfg 5 shape, none / 2 / 3 stages: direct form 19.9 / 18.1 / 19.3, packed 23.6 / 21.3 / 22.6; the
    // 12-byte elements of the general form do not have the registers: 31.7 / 35.2 / 40.3)
    constexpr bool PFON = (PACKED == 1 || PACKED == 2) && PFD > 0;     // (PACKED 3: there are no elements to fetch)
    // PACKED 3 (every slot has ONE flip mask, column = row ^ flip: csrc/midyn_flip.h): the LDS address of an operand is the
    // thread's own address XOR a per-slot constant -- lane e of flip_lane holds the constant of slot e (at most 64 slots; loaded
    // once per kernel), one v_readlane per slot: a scalar load there put its round trip in front of every slot's gathers.
    // Addresses are integers of the LDS address space: no addition of the vectors' base per gather; XOR and base commute
    // because the vectors start at a multiple of 64 KB (the kernel keeps its small tables BEHIND them; checked).
    typedef __attribute__((address_space(3))) char lds_char_t;
    typedef double lds_d2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const lds_d2v lds_d2_t;
    unsigned lax[PACKED == 3 ? SWEEP_RPT : 1];
    if (PACKED == 3) {
        const unsigned base_ = (unsigned)(size_t)(lds_char_t*)const_cast<double2*>(sweep_lds);
        if (base_ & 0xffffu) __builtin_trap();
#pragma unroll
        for (int i = 0; i < SWEEP_RPT; ++i) {
            const unsigned r_ = row0 + (unsigned)(tid + TH * i);
            lax[PACKED == 3 ? i : 0] = base_ + ((((r_ >> 11) << 12) | (r_ & 2047u)) << 4);
        }
    }
    constexpr int PF = PFON ? PFD : 1;
    int cn[PF][SWEEP_RPT];
    double vn[PF][SWEEP_RPT];
    auto fetch = [&](const int e_, int (&c_)[SWEEP_RPT], double (&v_)[SWEEP_RPT]) {
#pragma unroll
        for (int i = 0; i < SWEEP_RPT; ++i) {
            if (MIDYN_SWEEP_ABLATE == 3) {
                c_[i] = (int)((((unsigned)e_ * unp + ROW(i)) * 2654435761u) >> 8) & (np - 1);
                v_[i] = 1e-3;
            } else if (PACKED == 3) {
                c_[i] = 0;
            } else if (PACKED) {
                c_[i] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(a.pk + ((size_t)e_ * unp + row0)) + boff(i, 2));
            } else {
                c_[i] = *reinterpret_cast<const int*>(reinterpret_cast<const char*>(a.col + ((size_t)e_ * unp + row0)) + boff(i, 2));
                v_[i] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(a.val + ((size_t)e_ * unp + row0)) + boff(i, 3));
            }
        }
    };
#define MIDYN_SWEEP_K_SLOT(IM, S_)                                                                    \
    {                                                                                            \
        const double2 cc = cab[e];                                                               \
        const double ca = swapped ? cc.y : cc.x, cb = (swapped ? cc.x : cc.y) * (KEEP_O2 ? scale2 : 1.0); \
        int cl[SWEEP_RPT];                                                                       \
        double va[SWEEP_RPT];                                                                    \
        if (!PFON && PACKED != 3) fetch(e, cn[S_], vn[S_]);                                  \
        const unsigned xm_ = PACKED == 3 ? (unsigned)__builtin_amdgcn_readlane(flip_lane, e) : 0u; \
        _Pragma("unroll") for (int i = 0; i < SWEEP_RPT; ++i) {                                  \
            cl[i] = PACKED == 3 ? (int)(lax[PACKED == 3 ? i : 0] ^ xm_) : cn[S_][i];             \
            if (PACKED == 0) va[i] = vn[S_][i];                                                    \
        }                                                                                        \
        if (PFON && e + PF < hi) fetch(e + PF, cn[S_], vn[S_]);                              \
        double2 x1[SWEEP_RPT], x2[SWEEP_RPT];                                                    \
        _Pragma("unroll") for (int i = 0; i < SWEEP_RPT; ++i) {                                  \
            if (PACKED == 3) {   /* own address ^ flip, in the LDS address space */                   \
                lds_d2_t* q3 = (lds_d2_t*)(size_t)(unsigned)cl[i];                               \
                const lds_d2v g1 = q3[0], g2 = ORDER == 2 ? q3[2048] : g1;                       \
                x1[i] = make_double2(g1.x, g1.y);                                                \
                if (ORDER == 2) x2[i] = make_double2(g2.x, g2.y);                                \
            } else if (PACKED >= 2) {   /* the element IS the LDS byte address of its X1 operand */   \
                const char* q = reinterpret_cast<const char*>(sweep_lds) +                       \
                                (MIDYN_SWEEP_ABLATE == 2 ? (unsigned)((cl[i] & 48) + (tid << 4)) : (unsigned)cl[i]); \
                x1[i] = *reinterpret_cast<const double2*>(q);                                    \
                if (ORDER == 2) x2[i] = *reinterpret_cast<const double2*>(q + 32768);            \
            } else {                                                                             \
                int c = PACKED ? (cl[i] & 0x7fffffff) : cl[i];                                   \
                if (MIDYN_SWEEP_ABLATE == 2) c = (c & 3) + tid;                                  \
                x1[i] = X1[c];                                                                   \
                if (ORDER == 2) x2[i] = X2[c];                                                   \
            }                                                                                    \
        }                                                                                        \
        _Pragma("unroll") for (int i = 0; i < SWEEP_RPT; ++i) {                                  \
            double wa = ca, wb = cb;                                                             \
            if (PACKED >= 2) {                                                                   \
            } else if (PACKED == 1) {   /* the sign of the element goes into the gathered operand (in place) */ \
                const long long sgn = (long long)(((unsigned long long)(unsigned)cl[i] & 0x80000000ull) << 32); \
                x1[i].x = __longlong_as_double(__double_as_longlong(x1[i].x) ^ sgn);             \
                x1[i].y = __longlong_as_double(__double_as_longlong(x1[i].y) ^ sgn);             \
                if (ORDER == 2) {                                                                \
                    x2[i].x = __longlong_as_double(__double_as_longlong(x2[i].x) ^ sgn);         \
                    x2[i].y = __longlong_as_double(__double_as_longlong(x2[i].y) ^ sgn);         \
                }                                                                                \
            } else {                                                                             \
                wa = ca * va[i];                                                                 \
                wb = cb * va[i];                                                                 \
            }                                                                                    \
            if (MIDYN_SWEEP_ABLATE == 4) {   /* profiling: one add per gathered operand */        \
                o1[i].x += x1[i].x + x1[i].y;                                                    \
                if (ORDER == 2) o2[i].x += x2[i].x + x2[i].y;                                    \
                continue;                                                                        \
            }                                                                                    \
            if (IM) {                                                                            \
                o1[i].x = fma(-wa, x1[i].y, o1[i].x);                                            \
                o1[i].y = fma(wa, x1[i].x, o1[i].y);                                             \
            } else {                                                                             \
                o1[i].x = fma(wa, x1[i].x, o1[i].x);                                             \
                o1[i].y = fma(wa, x1[i].y, o1[i].y);                                             \
            }                                                                                    \
            if (ORDER == 2) {                                                                    \
                if (IM) {                                                                        \
                    o2[i].x = fma(-wb, x2[i].y, o2[i].x);                                        \
                    o2[i].y = fma(wb, x2[i].x, o2[i].y);                                         \
                } else {                                                                         \
                    o2[i].x = fma(wb, x2[i].x, o2[i].x);                                         \
                    o2[i].y = fma(wb, x2[i].y, o2[i].y);                                         \
                }                                                                                \
            }                                                                                    \
        }                                                                                        \
    }
#define MIDYN_SWEEP_K_RANGE(IM, LO_, HI_)                                                             \
    {                                                                                            \
        const int lo = (LO_), hi = (HI_);                                                        \
        if (PFON) {                                                                              \
            _Pragma("unroll") for (int s_ = 0; s_ < PF; ++s_)                                    \
                if (lo + s_ < hi) fetch(lo + s_, cn[s_], vn[s_]);                                \
        }                                                                                        \
        for (int e0 = lo; e0 < hi; e0 += PF) {                                                   \
            _Pragma("unroll") for (int s_ = 0; s_ < PF; ++s_) {                                  \
                const int e = e0 + s_;                                                           \
                if (e < hi) MIDYN_SWEEP_K_SLOT(IM, s_)                                            \
            }                                                                                    \
        }                                                                                        \
    }
    // (Tried: the gathers of slot e + 1 issued before the multiply-adds of slot e, two operand buffers -- their times ADD
    // in the loop below.  Order 2 has no registers for the second buffer (27 spills, 26.9 vs 19.3 us per term); order 1
    // has them and gains nothing (5.45 vs 5.33): the waves of a CU already overlap each other's phases as far as the LDS
    // pipeline lets them.)
    if (PART == 0) {
        MIDYN_SWEEP_K_RANGE(false, 0, a.wre)
        MIDYN_SWEEP_K_RANGE(true, a.wre, a.wsp)
    } else if (PART == 1) {
        MIDYN_SWEEP_K_RANGE(false, 0, a.wre_loc)
        MIDYN_SWEEP_K_RANGE(true, a.wre, a.wim_loc)
    } else if (PART == 2) {
        MIDYN_SWEEP_K_RANGE(false, a.wre_loc, a.wre)
        MIDYN_SWEEP_K_RANGE(true, a.wim_loc, a.wsp)
    } else {
        MIDYN_SWEEP_K_RANGE(false, re_lo, re_hi)
        MIDYN_SWEEP_K_RANGE(true, im_lo, im_hi)
    }
#undef MIDYN_SWEEP_K_RANGE
#undef MIDYN_SWEEP_K_SLOT
#undef ROW
}

template <int ORDER, int SWEEP_RPT, int TH, int PACKED>
__global__ __launch_bounds__(TH) void ell_sweep_kernel(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) double2 sweep_lds[];
    // per slot: (c1, c2) of its segment (x magnitude) of this step, and segment | plane << 8.  PACKED 3: BEHIND the vectors in the
    // dynamic LDS (no static LDS: the vectors start at LDS address 0, see sweep_pass); else static arrays
    __shared__ __attribute__((aligned(16))) double2 cab_static[PACKED == 3 ? 1 : SWEEP_MAX_SLOTS];
    __shared__ int stag_static[PACKED == 3 ? 1 : SWEEP_MAX_SLOTS];
    const size_t vec_bytes_ = (size_t)((a.n_pad + 2047) / 2048) * 65536;
    double2* const cab = PACKED == 3 ? reinterpret_cast<double2*>(reinterpret_cast<char*>(sweep_lds) + vec_bytes_) : cab_static;
    int* const stag = PACKED == 3 ? reinterpret_cast<int*>(reinterpret_cast<char*>(sweep_lds) + vec_bytes_ + SWEEP_MAX_SLOTS * sizeof(double2))
                                  : stag_static;
    const int flip_lane = (PACKED == 3 && (threadIdx.x & 63) < a.wsp) ? a.pk[threadIdx.x & 63] : 0;
    const int tid = threadIdx.x, b = blockIdx.x, np = a.n_pad;
    // LDS copies of the vectors the operators are applied to.  PACKED 0 / 1: X1[np + 1], X2[np + 1] ([np] = the zero
    // slot of unused packed elements).  PACKED 2: chunks of 2048 columns, [X1 chunk | X2 chunk] of 32 KB each, so that
    // the X2 operand of a column sits 32768 bytes behind its X1 operand -- an immediate offset of the same address.
    const int lstride = np + 1;
    double2* const X1 = sweep_lds;
    double2* const X2 = PACKED >= 2 ? sweep_lds + 2048 : sweep_lds + lstride;
    auto xrow = [&](const int r_) { return PACKED >= 2 ? ((r_ >> 11) << 12) | (r_ & 2047) : r_; };   // index of column r_ in X1 / X2
    auto rowof = [&](const int i_) {   // tid + TH i_, opaque to the optimiser (see boff below: nothing derived from it is hoisted)
        int r_ = tid + TH * i_;
        asm volatile("" : "+v"(r_));
        return r_;
    };
    // the two series vectors no pass touches live in a per-instance stash in device memory (L2): rows tid + TH i
    // (uniform base pointers + 32-bit BYTE offsets everywhere -- the scalar-base addressing form: one offset register per
    // row serves every array; with element indices hipcc builds a 64-bit address per row and array and spills them)
    double2* const sacc = a.stash + (size_t)b * 2 * np;   // the accumulated result
    double2* const scur = sacc + np;                       // phi_{j-1} (the next term's phi_{j-2})
    auto boff = [&](const int i_, const int shift) { return sweep_boff<TH>(tid, i_, shift); };   // (see sweep_boff)
#define ROW(i_) ((unsigned)(tid + TH * (i_)))
#define AT16(base_, i_) (*reinterpret_cast<double2*>(reinterpret_cast<char*>(const_cast<double2*>(base_)) + boff(i_, 4)))
    const double p2 = 0.14433756729740643;   // sqrt(3) / 12
    // order 1 has the registers for both vectors (one output vector per pass); order 2 does not
    constexpr bool STASH = ORDER == 2;
    double2 pw[SWEEP_RPT], racc[STASH ? 1 : SWEEP_RPT], rcur[STASH ? 1 : SWEEP_RPT];   // (pw: order 2 only outside the passes)
#define ACC(i_) (*(STASH ? &AT16(sacc, i_) : &racc[STASH ? 0 : (i_)]))
#define CUR(i_) (*(STASH ? &AT16(scur, i_) : &rcur[STASH ? 0 : (i_)]))
#pragma unroll
    for (int i = 0; i < SWEEP_RPT; ++i) {
        const int r = tid + TH * i;
        ACC(i) = (r < a.n) ? a.y0[(a.y0_shared ? 0 : (size_t)b * a.n) + r] : make_double2(0.0, 0.0);
    }
    for (int e = tid; e < a.wsp; e += TH) stag[e] = a.tags[e];
    if (tid == 0 && PACKED < 2) {
        X1[np] = make_double2(0.0, 0.0);
        if (ORDER == 2) X2[np] = make_double2(0.0, 0.0);
    }
    auto pass = [&](const bool swapped, double2 (&o1)[SWEEP_RPT], double2 (&o2)[SWEEP_RPT]) {
        sweep_pass<ORDER, SWEEP_RPT, TH, PACKED>(a, cab, sweep_lds, X1, X2, tid, swapped, o1, o2, 1.0, 0, 0, 0, 0, 0, flip_lane);
    };
    auto pass_keep = [&](const bool swapped, double2 (&o1)[SWEEP_RPT], double2 (&o2)[SWEEP_RPT], const double scale2) {
        // o2 continues from its start value, its sum scaled by scale2
        sweep_pass<ORDER, SWEEP_RPT, TH, PACKED, MIDYN_SWEEP_PREFETCH, true>(a, cab, sweep_lds, X1, X2, tid, swapped, o1, o2, scale2, 0, 0, 0, 0, 0, flip_lane);
    };
    for (int st = 0; st < a.nsteps; ++st) {
        const int r0 = a.rows[3 * st], r1 = a.rows[3 * st + 1];
        const double h = a.hs[st];
        __syncthreads();   // the previous step's readers of the coefficients are done (and stag is written)
        for (int e = tid; e < a.wsp; e += TH) {
            const int seg = stag[e] & 63;
            const bool stat = a.has_static && seg == 0;
            const double* Sb = a.S + (size_t)b * a.inst_stride;
            const double mg = PACKED ? a.mag[e] : 1.0;     // PACKED 2: signed
            cab[e] = make_double2(mg * (stat ? 1.0 : Sb[(size_t)r0 * a.k + seg - a.has_static]),
                                  (ORDER == 2) ? mg * (stat ? 1.0 : Sb[(size_t)r1 * a.k + seg - a.has_static]) : 0.0);
        }
        const double2* const E0 = a.E ? a.E + (size_t)r0 * np : nullptr;
        const double2* const D = (a.E && ORDER == 2) ? a.Dt + (size_t)st * np : nullptr;
        const int Ks = a.ser_K[st], reps = a.ser_reps[st];
        const bool cheb = Ks > 0;
        const int K = cheb ? Ks : -Ks;
        const double par = a.ser_par[st];
        const double* coef = a.coef + (size_t)st * a.stride;
        const int slot = a.save ? a.save[st] : -1;
        for (int rep = 0; rep < reps; ++rep) {
            const double c0 = cheb ? coef[0] : 1.0;
            // start of a series: phi_0 = the accumulated result (into the frame picture of the first Gauss point at the
            // first repetition: y~ = E(t1) o y), staged for the first term
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                const int r = rowof(i);
                double2 v = ACC(i);
                if (rep == 0 && a.E) v = cmul(AT16(E0, i), v);
                X1[xrow(r)] = v;
                if (ORDER == 2) X2[xrow(r)] = D ? cmul(AT16(D, i), v) : v;
                CUR(i) = v;
                ACC(i) = make_double2(c0 * v.x, c0 * v.y);
                pw[i] = make_double2(0.0, 0.0);      // Chebyshev: phi_{j-2};  Taylor: nothing
            }
            __syncthreads();
            for (int j = 1; j <= K; ++j) {
                const double f = cheb ? (j == 1 ? 1.0 : 2.0) / par : 1.0 / (par * (double)j);
                double hh = h;                       // (per-term scalars recomputed, not carried in registers)
                asm volatile("" : "+v"(hh));
                double2 o1[SWEEP_RPT], o2[SWEEP_RPT];
                pass(false, o1, o2);                 // o1 = C(t1) v~, o2 = C(t2) (D v~)
                if (ORDER == 2) {
                    const double ca = 0.5 * hh * f, cb = p2 * hh * hh * f;
                    __syncthreads();
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) {
                        const int r = rowof(i);
                        const double2 u1 = o1[i];
                        double2 u2 = o2[i], du1 = u1;
                        if (D) {
                            const double2 dd = AT16(D, i);
                            u2 = cmul_conj_a(dd, o2[i]);
                            du1 = cmul(dd, u1);
                        }
                        X1[xrow(r)] = du1;           // for g2~ = conj(D) C(t2) D
                        X2[xrow(r)] = u2;            // for g1~ = C(t1)
                        // the term so far, m = phi_{j-2} + ca (u1 + u2), rides through the second pass INSIDE its second sum
                        // (start value -m, the sum scaled by cb: cb v1 - o2 then IS m + cb (v1 - g1~ u2)) -- no vector to
                        // keep in registers or to stash across the pass
                        o2[i] = make_double2(-(pw[i].x + ca * (u1.x + u2.x)), -(pw[i].y + ca * (u1.y + u2.y)));
                    }
                    __syncthreads();
                    pass_keep(true, o1, o2, cb);     // o1 = C(t2) (D u1), o2 = -m + cb C(t1) u2
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) {
                        const double2 v1 = D ? cmul_conj_a(AT16(D, i), o1[i]) : o1[i];
                        pw[i] = make_double2(cb * v1.x - o2[i].x, cb * v1.y - o2[i].y);
                    }
                } else {
                    const double ca = hh * f;
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) pw[i] = cfma_r(ca, o1[i], pw[i]);
                }
                // end of the term: w = pw joins the result; unless it was the last, it is staged for the next term and
                // the old phi_{j-1} comes back from the stash as the next phi_{j-2}
                const bool last = j == K;
                const double cj = cheb ? 2.0 * coef[j] : 1.0;
                if (!last) __syncthreads();          // every reader of X1 / X2 of this term is done
#pragma unroll
                for (int i = 0; i < SWEEP_RPT; ++i) {
                    const int r = rowof(i);
                    const double2 w = pw[i];
                    double2 acc = cfma_r(cj, w, ACC(i));
                    if (!last) {
                        pw[i] = cheb ? CUR(i) : make_double2(0.0, 0.0);
                        CUR(i) = w;
                        X1[xrow(r)] = w;
                        if (ORDER == 2) X2[xrow(r)] = D ? cmul(AT16(D, i), w) : w;
                    } else if (rep + 1 == reps) {    // out of the frame picture; saved states
                        if (a.E) acc = cmul_conj_a(AT16(E0, i), acc);
                        if (slot >= 0 && r < a.n) AT16(a.out + ((size_t)b * a.P + (slot < 0 ? 0 : slot)) * a.n, i) = acc;
                    }
                    ACC(i) = acc;
                }
                if (!last) __syncthreads();
            }
        }
    }
}
#undef ROW
#undef AT16
#undef ACC
#undef CUR


// ------------------------------------------------------------------------------------------------
// ell_sweep_duo_kernel<ORDER, RPT, TH, PACKED>: TWO workgroups per instance (round 5) -- the 128-instance cfg 5 shard of an
// 8-GPU run on all 256 CUs.  What round 2's split kernel (below) paid for the second workgroup was the exchange: every
// operand vector all-gathered between the partners BEFORE the pass that needs it, through write-through stores of payload
// AND re-arming sentinels -- 74 MB per series term over the fabric for 256 workgroups, 7 us per all-gather at cfg 5's
// size: what the halved pass saves.  Here:
//   * each workgroup owns half of the rows (n_pad / 2 = TH RPT) and stages ITS half of an operand vector in LDS at once;
//   * stack_ell_layout orders the operator slots so that those whose operands lie in the row's own half come first
//     (a_.wre_loc / a_.wim_loc) -- operators built from Pauli strings couple the halves through the strings that flip the
//     top qubit only: 2 of cfg 5's 19 slots -- and the pass runs those slots first (sweep_pass<.., PART 1>) while the
//     partner's half is on its way; then the partner's rows are copied into the other half of the LDS vectors and the few
//     slots that reach across follow (PART 2);
//   * the hand-off is per WAVE: thread tid of one workgroup needs exactly the rows thread tid of the other has published
//     (the same tid + TH i of the other half), so wave w waits for wave w of its partner and for nobody else.  A wave
//     stores its rows into the round's payload slot, drains its stores (s_waitcnt vmcnt(0)) and then sets its flag word
//     to the round number; the reader polls that ONE word (sc1 load), then reads the payload with sc1 (L1-bypassing)
//     loads.  Rounds alternate between a one-vector and a two-vector slot (order 2; order 1: two one-vector slots), and
//     a slot is written again two rounds later, after the writer has seen the reader's flag of the round in between --
//     which the reader sets after it has finished reading.  Monotonic round numbers: nothing is re-armed;
//   * the partners of an instance compare their XCC ids once (agent-scope words).  On ONE XCD -- where the observed
//     b % 8 dispatch puts them (part_major), but nothing depends on it -- they share the L2, so payload and flag are PLAIN
//     stores that stay in that L2 and the sc1 loads of the reader are served by it: no fabric traffic at all.  On
//     different XCDs the stores are write-through (sc1), the same protocol (cdna_hip_programming.md G16, form R1).
// With half the rows per thread the series vectors of order 2 fit the registers again (no stash in device memory), and
// the frame picture, the packed element forms and the commutator-free Magnus-2 term are ell_sweep_kernel's.
// Both workgroups of every instance must be resident at once (one per CU: LDS): cooperative launch, bounded waits, and the
// host falls back to ell_sweep_kernel when a wait gives up (midyn_action.inc).
// ------------------------------------------------------------------------------------------------
struct SweepDuoArgs {
    SweepArgs a;
    double2* ring;              // [B][3 vectors][n_pad] payload slots
    int* flags;                 // [B][2 halves][DUO_FLAG_WORDS]: [0, 16) round number per wave, [16] XCC id + 1; all zero at launch
    int* err;
    unsigned spin_limit;        // polls after which a wait gives up
    int protocol;             // ctx option exchange_protocol: 0 = the measured default, 1 = the conforming forms (see midyn_core.inc)
    int part_major;             // 1: blocks [p B, (p + 1) B) hold half p of every instance (B a multiple of 8: the partners of an
                                // instance on one XCD under the observed b % 8 dispatch -- a speed matter only)
    int ablate;                 // profiling only (results wrong): 1 no exchange at all, 2 write-through stores on one XCD too,
                                // 4 no local slots, 8 no crossing slots
};
constexpr int DUO_FLAG_WORDS = 32;

typedef unsigned int sweep_u4 __attribute__((ext_vector_type(4)));

template <int ORDER, int SWEEP_RPT, int TH, int PACKED>
__global__ __launch_bounds__(TH) void ell_sweep_duo_kernel(const SweepDuoArgs da) {
    static_assert(PACKED == 1 || PACKED == 2, "packed element forms only");
    const SweepArgs& a = da.a;
    extern __shared__ __attribute__((aligned(16))) double2 sweep_lds[];
    __shared__ __attribute__((aligned(16))) double2 cab[SWEEP_MAX_SLOTS];
    __shared__ int stag[SWEEP_MAX_SLOTS];
    const int tid = threadIdx.x, np = a.n_pad;
    const unsigned unp = (unsigned)np;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = gridDim.x >> 1;
    const int b = da.part_major ? (int)(blockIdx.x % nb) : (int)(blockIdx.x >> 1);
    const int part = da.part_major ? (int)(blockIdx.x / nb) : (int)(blockIdx.x & 1);
    const unsigned half = unp >> 1;                             // = TH * SWEEP_RPT rows per workgroup
    const unsigned row0 = part ? half : 0u, prow0 = part ? 0u : half;
    const int lstride = np + 1;
    double2* const X1 = sweep_lds;
    double2* const X2 = PACKED == 2 ? sweep_lds + 2048 : sweep_lds + lstride;
    auto xrow = [&](const unsigned r_) { return PACKED == 2 ? ((r_ >> 11) << 12) | (r_ & 2047u) : r_; };   // index of column r_ in X1 / X2
    auto boff = [&](const int i_, const int shift) { return sweep_boff<TH>(tid, i_, shift); };   // (see sweep_boff)
    auto rowof = [&](const int i_) {   // tid + TH i_, opaque to the optimiser: no address derived from it is hoisted out of the loops
        unsigned r_ = (unsigned)(tid + TH * i_);
        asm volatile("" : "+v"(r_));
        return r_;
    };
#define AT16(base_, i_) (*reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base_) + boff(i_, 4)))
    // the payload slots of this instance through a buffer descriptor (16-byte loads / stores; aux 16 = sc1)
    const size_t ring_doubles2 = (size_t)3 * np;
    const __amdgpu_buffer_rsrc_t ring = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<double2*>(da.ring) + (size_t)b * ring_doubles2, 0, (int)(ring_doubles2 * sizeof(double2)), 0x00020000);
    auto ring_off = [&](const int slot, const unsigned row) { return (unsigned)((slot * np + (int)row) << 4); };
    int* const my_flags = da.flags + ((size_t)b * 2 + part) * DUO_FLAG_WORDS;
    int* const partner_flags = da.flags + ((size_t)b * 2 + (1 - part)) * DUO_FLAG_WORDS;
    bool dead = false;
    // a bounded wait for a word of the partner to reach `want` (every lane loads the same word)
    auto wait_word = [&](int* word, const int want) {
        unsigned spins = 0;
        for (;;) {
            if (dead) return 0;
            const int got = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (got >= want) return got;
            __builtin_amdgcn_s_sleep(1);
            ++spins;
            if ((spins & 1023u) == 0 && __hip_atomic_load(da.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) spins = da.spin_limit;
            if (spins >= da.spin_limit) {
                __hip_atomic_store(da.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dead = true;
            }
        }
    };
    // do the partners share an XCD (one L2)?  Each publishes its XCC id + 1 once and reads the other's.
    bool one_l2 = false;
    if (!(da.ablate & 1)) {
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        if (tid == 0) __hip_atomic_store(my_flags + 16, (int)xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int theirs = wait_word(partner_flags + 16, 1);
        one_l2 = !dead && theirs == (int)xcc + 1 && !(da.ablate & 2) && !da.protocol;    // (protocol 1: sc1 stores AND sc1 loads always)
    }
    const double p2 = 0.14433756729740643;   // sqrt(3) / 12
    double2 pw[SWEEP_RPT], acc[SWEEP_RPT], cur[SWEEP_RPT];
#pragma unroll
    for (int i = 0; i < SWEEP_RPT; ++i) {
        const unsigned r = row0 + rowof(i);
        acc[i] = (r < (unsigned)a.n) ? a.y0[(a.y0_shared ? 0 : (size_t)b * a.n) + r] : make_double2(0.0, 0.0);
        cur[i] = pw[i] = make_double2(0.0, 0.0);
    }
    for (int e = tid; e < a.wsp; e += TH) stag[e] = a.tags[e];
    if (tid == 0 && PACKED != 2) {
        X1[np] = make_double2(0.0, 0.0);
        if (ORDER == 2) X2[np] = make_double2(0.0, 0.0);
    }
    // Operator slots in the order they are applied: the local ones (real plane, then imaginary plane), then the crossing ones.
    const int n_loc = a.wre_loc + (a.wim_loc - a.wre), n_all = a.wsp;
    // the slots [j_lo, j_hi) of that order (sweep_pass with explicit ranges; the sums o1 / o2 continue)
    auto run_slots = [&](const int j_lo, const int j_hi, const bool swapped, const bool keep, const double scale2,
                         double2 (&o1)[SWEEP_RPT], double2 (&o2)[SWEEP_RPT]) {
        if (j_lo >= j_hi) return;
        int re_lo, re_hi, im_lo, im_hi;
        if (j_hi <= n_loc) {
            const int nre = a.wre_loc;
            re_lo = j_lo < nre ? j_lo : nre;
            re_hi = j_hi < nre ? j_hi : nre;
            im_lo = a.wre + (j_lo > nre ? j_lo - nre : 0);
            im_hi = a.wre + (j_hi > nre ? j_hi - nre : 0);
        } else {                           // (the crossing slots: all of them)
            re_lo = a.wre_loc;
            re_hi = a.wre;
            im_lo = a.wim_loc;
            im_hi = a.wsp;
        }
        if (keep) sweep_pass<ORDER, SWEEP_RPT, TH, PACKED, MIDYN_SWEEP_PREFETCH, true, 3>(a, cab, sweep_lds, X1, X2, tid, swapped, o1, o2, scale2, row0, re_lo, re_hi, im_lo, im_hi);
        else sweep_pass<ORDER, SWEEP_RPT, TH, PACKED, MIDYN_SWEEP_PREFETCH, false, 3>(a, cab, sweep_lds, X1, X2, tid, swapped, o1, o2, 1.0, row0, re_lo, re_hi, im_lo, im_hi);
    };
    int rr = 0;       // exchange rounds so far
    // One exchange round and the pass it feeds.  in1 / in2: this thread's rows of the operand vectors as the operators see
    // them (X1 / X2 forms).  nv = 1: one vector travels (in1) and the X2 form of the partner's rows is dtab o in1 (dtab: the
    // frame phase between the Gauss points of this step; use_dp false: none); nv = 2: both forms travel.
    // keep: o2 continues from its start value (its sum scaled by scale2); else both sums start at zero.
    auto exchange_and_pass = [&](const double2 (&in1)[SWEEP_RPT], const double2 (&in2)[SWEEP_RPT], const int nv,
                                 const double2* dtab, const bool use_dp, const bool swapped, const bool keep,
                                 const double scale2, double2 (&o1)[SWEEP_RPT], double2 (&o2)[SWEEP_RPT]) {
        const int slot = ORDER == 2 ? (nv == 2 ? 1 : 0) : (rr & 1);      // payload slot(s) of this round
        const bool exch = !(da.ablate & 1);
        __syncthreads();                   // every reader of the LDS copies of the previous pass is done
#pragma unroll
        for (int i = 0; i < SWEEP_RPT; ++i) {
            const unsigned r = row0 + rowof(i);
            if (exch) {
#pragma unroll
                for (int v = 0; v < ORDER; ++v) {
                    if (v >= nv) continue;
                    const double2 z = v ? in2[i] : in1[i];
                    sweep_u4 w;
                    const unsigned long long zx = (unsigned long long)__double_as_longlong(z.x), zy = (unsigned long long)__double_as_longlong(z.y);
                    w.x = (unsigned)zx; w.y = (unsigned)(zx >> 32); w.z = (unsigned)zy; w.w = (unsigned)(zy >> 32);
                    if (one_l2) __builtin_amdgcn_raw_buffer_store_b128(w, ring, (int)ring_off(slot + v, r), 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b128(w, ring, (int)ring_off(slot + v, r), 0, 16);
                }
            }
            X1[xrow(r)] = in1[i];
            if (ORDER == 2) X2[xrow(r)] = in2[i];
            o1[i] = make_double2(0.0, 0.0);
            if (!keep) o2[i] = make_double2(0.0, 0.0);
        }
        __syncthreads();
        ++rr;
        // This wave's rows have reached memory (the L2 both partners share, or written through): its flag says so.  (Measured:
        // splitting the local slots in three parts around the flag store and the partner's loads -- so that the store
        // acknowledgement and the load latency would hide behind slots -- is SLOWER, 17.2 against 15.4 us per term at 128
        // instances: every part restarts the element prefetch and drains it again; profiles/r05_cfg5_duo.md.)
        if (exch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if ((tid & 63) == 0) {
                // (one L2: a store that stays in it -- workgroup scope lowers to sc0, which keeps the line; agent scope writes through)
                if (one_l2) __hip_atomic_store(my_flags + wave, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_store(my_flags + wave, rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // the slots that stay inside this half, while the partner's half is on its way
        run_slots(0, (da.ablate & 4) ? 0 : n_loc, swapped, keep, scale2, o1, o2);
        // the partner's rows prow0 + tid + TH i: published by ITS thread tid, i.e. by its wave `wave`
        if (exch) {
            (void)wait_word(partner_flags + wave, rr);
            asm volatile("" ::: "memory");          // (compiler: the payload is read after the flag, not before)
            sweep_u4 w[SWEEP_RPT][ORDER];
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                const unsigned r = prow0 + rowof(i);
#pragma unroll
                for (int v = 0; v < ORDER; ++v)
                    if (v < nv) w[i][v] = __builtin_amdgcn_raw_buffer_load_b128(ring, (int)ring_off(slot + v, r), 0, 16);
            }
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                const unsigned r = prow0 + rowof(i);
                auto unpack = [](const sweep_u4 q) {
                    return make_double2(__longlong_as_double((long long)(((unsigned long long)q.y << 32) | q.x)),
                                        __longlong_as_double((long long)(((unsigned long long)q.w << 32) | q.z)));
                };
                const double2 xa = unpack(w[i][0]);
                X1[xrow(r)] = xa;
                if (ORDER == 2) X2[xrow(r)] = nv == 2 ? unpack(w[i][ORDER - 1]) : (use_dp ? cmul(AT16(dtab + prow0, i), xa) : xa);
            }
        }
        __syncthreads();
        if (!(da.ablate & 8)) run_slots(n_loc, n_all, swapped, keep, scale2, o1, o2);
    };
    for (int st = 0; st < a.nsteps; ++st) {
        const int r0 = a.rows[3 * st], r1 = a.rows[3 * st + 1];
        const double h = a.hs[st];
        __syncthreads();   // the previous step's readers of the coefficients are done (and stag is written)
        for (int e = tid; e < a.wsp; e += TH) {
            const int seg = stag[e] & 63;
            const bool stat = a.has_static && seg == 0;
            const double* Sb = a.S + (size_t)b * a.inst_stride;
            const double mg = a.mag[e];     // PACKED 2: signed
            cab[e] = make_double2(mg * (stat ? 1.0 : Sb[(size_t)r0 * a.k + seg - a.has_static]),
                                  (ORDER == 2) ? mg * (stat ? 1.0 : Sb[(size_t)r1 * a.k + seg - a.has_static]) : 0.0);
        }
        const double2* const E0 = a.E ? a.E + (size_t)r0 * np : nullptr;
        const bool framed2 = a.E && ORDER == 2;
        // the frame phase between the two Gauss points of this step for this thread's rows (the partner's rows: read with the payload)
        const double2* const dtab = framed2 ? a.Dt + (size_t)st * np : nullptr;
        constexpr bool DM_REG = true;
        double2 dmr[DM_REG ? SWEEP_RPT : 1];
        if (DM_REG) {
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) dmr[DM_REG ? i : 0] = framed2 ? AT16(dtab + row0, i) : make_double2(1.0, 0.0);
        }
        auto dm = [&](const int i_) { return DM_REG ? dmr[DM_REG ? i_ : 0] : AT16(dtab + row0, i_); };
        const int Ks = a.ser_K[st], reps = a.ser_reps[st];
        const bool cheb = Ks > 0;
        const int K = cheb ? Ks : -Ks;
        const double par = a.ser_par[st];
        const double* coef = a.coef + (size_t)st * a.stride;
        const int slot = a.save ? a.save[st] : -1;
        for (int rep = 0; rep < reps; ++rep) {
            const double c0 = cheb ? coef[0] : 1.0;
            // start of a series: phi_0 = the accumulated result (into the frame picture of the first Gauss point at the
            // first repetition: y~ = E(t1) o y)
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                double2 v = acc[i];
                if (rep == 0 && a.E) v = cmul(AT16(E0 + row0, i), v);
                cur[i] = v;
                acc[i] = make_double2(c0 * v.x, c0 * v.y);
                pw[i] = make_double2(0.0, 0.0);      // Chebyshev: phi_{j-2};  Taylor: nothing
            }
            for (int j = 1; j <= K; ++j) {
                const double f = cheb ? (j == 1 ? 1.0 : 2.0) / par : 1.0 / (par * (double)j);
                double2 o1[SWEEP_RPT], o2[SWEEP_RPT], in2[SWEEP_RPT];
                if (ORDER == 2) {
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) in2[i] = framed2 ? cmul(dm(i), cur[i]) : cur[i];
                }
                // o1 = C(t1) v~, o2 = C(t2) (D v~)
                exchange_and_pass(cur, ORDER == 2 ? in2 : cur, 1, dtab, framed2, false, false, 1.0, o1, o2);
                double2 w[SWEEP_RPT];
                if (ORDER == 2) {
                    const double ca = 0.5 * h * f, cb = p2 * h * h * f;
                    double2 du1[SWEEP_RPT], u2[SWEEP_RPT], m0[SWEEP_RPT];
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) {
                        const double2 u1 = o1[i];
                        u2[i] = o2[i];
                        du1[i] = u1;
                        if (framed2) {
                            u2[i] = cmul_conj_a(dm(i), o2[i]);
                            du1[i] = cmul(dm(i), u1);       // for g2~ = conj(D) C(t2) D
                        }
                        // the term so far, m = phi_{j-2} + ca (u1 + u2), rides through the second pass INSIDE its second sum
                        // (start value -m, the sum scaled by cb: cb v1 - o2 then IS m + cb (v1 - g1~ u2))
                        m0[i] = make_double2(-(pw[i].x + ca * (u1.x + u2[i].x)), -(pw[i].y + ca * (u1.y + u2[i].y)));
                    }
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) o2[i] = m0[i];
                    exchange_and_pass(du1, u2, 2, dtab, false, true, true, cb, o1, o2);     // o1 = C(t2) (D u1), o2 = -m + cb C(t1) u2
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) {
                        const double2 v1 = framed2 ? cmul_conj_a(dm(i), o1[i]) : o1[i];
                        w[i] = make_double2(cb * v1.x - o2[i].x, cb * v1.y - o2[i].y);
                    }
                } else {
                    const double ca = h * f;
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) w[i] = cfma_r(ca, o1[i], pw[i]);
                }
                // end of the term: w joins the result and is the next term's input; the old phi_{j-1} is the next phi_{j-2}
                const bool last = j == K;
                const double cj = cheb ? 2.0 * coef[j] : 1.0;
#pragma unroll
                for (int i = 0; i < SWEEP_RPT; ++i) {
                    const unsigned r = row0 + rowof(i);
                    double2 ac = cfma_r(cj, w[i], acc[i]);
                    if (!last) {
                        pw[i] = cheb ? cur[i] : make_double2(0.0, 0.0);
                        cur[i] = w[i];
                    } else if (rep + 1 == reps) {    // out of the frame picture; saved states
                        if (a.E) ac = cmul_conj_a(AT16(E0 + row0, i), ac);
                        if (slot >= 0 && r < (unsigned)a.n) a.out[((size_t)b * a.P + slot) * a.n + r] = ac;
                    }
                    acc[i] = ac;
                }
            }
        }
    }
}
#undef AT16


// ------------------------------------------------------------------------------------------------
// ell_sweep_split_kernel<ORDER, RPT>: the sweep kernel for SMALL shards (fewer instances than half the CUs: the
// 128-instance cfg 5 shard of an 8-GPU run leaves 128 of 256 CUs idle).  NSPLIT = 2 or 4 workgroups share one instance:
// each owns n_pad / NSPLIT rows (RPT per thread) -- half or a quarter of the operator elements per pass and of the
// series state per thread -- and every vector an operator is applied to is all-gathered among them through the
// sentinel-polled ring of the resident kernels (the partners of an instance read each other: symmetric by
// construction; four rotating buffers; the owner re-arms its words of the buffer read last round).  A workgroup
// writes its own rows straight into its LDS copy, publishes them unphased, and phases its partners' rows as it copies
// them in.  All NSPLIT * B workgroups must be resident at once (one per CU: LDS): cooperative launch.
// ------------------------------------------------------------------------------------------------
// ell_sweep_rk4_kernel<RPT, TH, PACKED>: the same one-workgroup-per-instance form for fixed-step RK4 sweeps (a9) on very
// sparse stacks: a stage is ONE pass over the operator elements (sweep_pass, order 1, the element forms of
// ell_sweep_kernel); y and the accumulator of the thread's rows stay in registers, the stage input is staged phased in
// LDS, the stage arithmetic is apply_epilogue_t's.
template <int SWEEP_RPT, int TH, int PACKED>
__global__ __launch_bounds__(TH) void ell_sweep_rk4_kernel(const SweepArgs a) {
    extern __shared__ __attribute__((aligned(16))) double2 sweep_lds[];
    // per slot: (coefficient of its segment at the stage time x magnitude, -), and segment | plane << 8.  PACKED 3: behind the
    // vector in the dynamic LDS (no static LDS: the vector starts at LDS address 0, see sweep_pass)
    __shared__ __attribute__((aligned(16))) double2 cab_static[PACKED == 3 ? 1 : SWEEP_MAX_SLOTS];
    __shared__ int stag_static[PACKED == 3 ? 1 : SWEEP_MAX_SLOTS];
    const int tid = threadIdx.x, b = blockIdx.x, np = a.n_pad;
    double2* const X1 = sweep_lds;            // layouts as in ell_sweep_kernel (PACKED 2 / 3: chunks of 2048 columns)
    const size_t vec_bytes_ = (size_t)((a.n_pad + 2047) / 2048) * 65536 - 32768;
    double2* const cab = PACKED == 3 ? reinterpret_cast<double2*>(reinterpret_cast<char*>(sweep_lds) + vec_bytes_) : cab_static;
    int* const stag = PACKED == 3 ? reinterpret_cast<int*>(reinterpret_cast<char*>(sweep_lds) + vec_bytes_ + SWEEP_MAX_SLOTS * sizeof(double2))
                                  : stag_static;
    const int flip_lane = (PACKED == 3 && (threadIdx.x & 63) < a.wsp) ? a.pk[threadIdx.x & 63] : 0;
    auto xrow = [&](const int r_) { return PACKED >= 2 ? ((r_ >> 11) << 12) | (r_ & 2047) : r_; };
    auto boff = [&](const int i_, const int shift) { return sweep_boff<TH>(tid, i_, shift); };
    double2 y[SWEEP_RPT], acc[SWEEP_RPT], cur[SWEEP_RPT];
#pragma unroll
    for (int i = 0; i < SWEEP_RPT; ++i) {
        const int r = tid + TH * i;
        y[i] = (r < a.n) ? a.y0[(a.y0_shared ? 0 : (size_t)b * a.n) + r] : make_double2(0.0, 0.0);
        acc[i] = cur[i] = y[i];
    }
    for (int e = tid; e < a.wsp; e += TH) stag[e] = a.tags[e];
    if (tid == 0 && PACKED < 2) X1[np] = make_double2(0.0, 0.0);
    const double* Sb = a.S + (size_t)b * a.inst_stride;
    for (int st = 0; st < a.nsteps; ++st) {
        const int r0 = a.rows[3 * st], r1 = a.rows[3 * st + 1], r2 = a.rows[3 * st + 2];
        const double h = a.hs[st];
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            const int srow = sg == 0 ? r0 : (sg == 3 ? r2 : r1);
            const double2* Es = a.E ? a.E + (size_t)srow * np : nullptr;
            __syncthreads();    // the previous stage's readers of X1 / cab are done (and stag is written)
            for (int e = tid; e < a.wsp; e += TH) {
                const int seg = stag[e] & 63;
                const double c = (a.has_static && seg == 0) ? 1.0 : Sb[(size_t)srow * a.k + seg - a.has_static];
                cab[e] = make_double2(PACKED ? a.mag[e] * c : c, 0.0);
            }
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                const int r = tid + TH * i;
                X1[xrow(r)] = Es ? cmul(*reinterpret_cast<const double2*>(reinterpret_cast<const char*>(Es) + boff(i, 4)), cur[i]) : cur[i];
            }
            __syncthreads();
            double2 o1[SWEEP_RPT], o2[SWEEP_RPT];
            sweep_pass<1, SWEEP_RPT, TH, PACKED, (SWEEP_RPT < 4 ? MIDYN_SWEEP_PREFETCH : 0)>(a, cab, sweep_lds, X1, X1, tid, false, o1, o2, 1.0, 0, 0, 0, 0, 0, flip_lane);   // (four rows per thread: no registers for the element ring)
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                const double2 kk = Es ? cmul_conj_a(*reinterpret_cast<const double2*>(reinterpret_cast<const char*>(Es) + boff(i, 4)), o1[i]) : o1[i];
                if (sg == 0) {
                    acc[i] = cfma_r(h * (1.0 / 6), kk, y[i]);
                    cur[i] = cfma_r(0.5 * h, kk, y[i]);
                } else if (sg == 1) {
                    acc[i] = cfma_r(h * (1.0 / 3), kk, acc[i]);
                    cur[i] = cfma_r(0.5 * h, kk, y[i]);
                } else if (sg == 2) {
                    acc[i] = cfma_r(h * (1.0 / 3), kk, acc[i]);
                    cur[i] = cfma_r(h, kk, y[i]);
                } else {
                    y[i] = cfma_r(h * (1.0 / 6), kk, acc[i]);
                    cur[i] = y[i];
                }
            }
        }
        if (a.save) {
            const int slot = a.save[st];
            if (slot >= 0) {
#pragma unroll
                for (int i = 0; i < SWEEP_RPT; ++i) {
                    const int r = tid + TH * i;
                    if (r < a.n) a.out[((size_t)b * a.P + slot) * a.n + r] = y[i];
                }
            }
        }
    }
}

struct SweepSplitArgs {
    SweepArgs a;
    int nsplit;
    unsigned long long* ring;   // [B][4][ORDER][2 * n_pad] words, all sentinel at launch
    int* err;
    unsigned spin_limit;        // polls after which a wait gives up
    int part_major;             // 1: blocks [p B, (p + 1) B) hold part p of every instance -- with B a multiple of 8 the
                                // partners of an instance sit on the same XCD under the observed b % 8 dispatch (a speed
                                // matter only: the exchange is agent-scope either way); 0: parts of an instance adjacent
};

template <int ORDER, int SWEEP_RPT>
__global__ __launch_bounds__(SWEEP_THREADS) void ell_sweep_split_kernel(const SweepSplitArgs sa) {
    const SweepArgs& a = sa.a;
    extern __shared__ __attribute__((aligned(16))) double2 sweep_lds[];
    __shared__ __attribute__((aligned(16))) double2 cab[SWEEP_MAX_SLOTS];
    __shared__ int stag[SWEEP_MAX_SLOTS];
    const int tid = threadIdx.x, np = a.n_pad, nsplit = sa.nsplit;
    const int nb = gridDim.x / nsplit;
    const int b = sa.part_major ? blockIdx.x % nb : blockIdx.x / nsplit, part = sa.part_major ? blockIdx.x / nb : blockIdx.x % nsplit;
    const int part_rows = np / nsplit, row0 = part * part_rows;     // this workgroup's rows: row0 + tid + 1024 i
    double2* const L1 = sweep_lds;
    double2* const L2 = sweep_lds + np;
    unsigned long long* const ring = sa.ring + (size_t)b * 4 * ORDER * 2 * np;
    const double p2 = 0.14433756729740643;   // sqrt(3) / 12
    double2 acc[SWEEP_RPT], cur[SWEEP_RPT], prev[SWEEP_RPT];
#pragma unroll
    for (int i = 0; i < SWEEP_RPT; ++i) {
        const int r = row0 + tid + SWEEP_THREADS * i;
        acc[i] = (r < a.n) ? a.y0[(a.y0_shared ? 0 : (size_t)b * a.n) + r] : make_double2(0.0, 0.0);
        cur[i] = prev[i] = make_double2(0.0, 0.0);
    }
    for (int e = tid; e < a.wsp; e += SWEEP_THREADS) stag[e] = a.tags[e];
    bool dead = false;
    int b_cur = 0;
    // All-gather of ORDER vectors: this thread's rows hold (va[i], vb[i]); LDS gets ea o va (and eb o vb) for ALL rows,
    // where ea / eb are the phase rows of two table times (nullptr: no frame).  `same`: vb IS va (the input of a term goes
    // to both Gauss points): one vector is published and read, both phased copies are made from it.  At order 2 the rounds
    // alternate same / two vectors, so the buffer re-armed in a round (published two rounds ago) holds what this round holds.
    auto all_gather = [&](const double2 (&va)[SWEEP_RPT], const double2 (&vb)[SWEEP_RPT], const double2* ea, const double2* eb,
                          const bool same) {
        const int b_nxt = (b_cur + 1) & 3, b_rearm = (b_cur + 3) & 3;
        const int nv = (ORDER == 2 && !same) ? 2 : 1;      // vectors through the ring this round
        unsigned long long* nxt = ring + (size_t)b_nxt * ORDER * 2 * np;
        __builtin_amdgcn_s_waitcnt(0);     // last round's re-arming stores are complete before this round's data leaves
        __atomic_signal_fence(__ATOMIC_SEQ_CST);   // (compiler: no ring store moves above the wait)
        __syncthreads();                   // every reader of the LDS copies of the previous pass is done
#pragma unroll
        for (int i = 0; i < SWEEP_RPT; ++i) {
            const int r = row0 + tid + SWEEP_THREADS * i;
            __hip_atomic_store(nxt + 2 * r, (unsigned long long)__double_as_longlong(va[i].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(nxt + 2 * r + 1, (unsigned long long)__double_as_longlong(va[i].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L1[r] = ea ? cmul(ea[r], va[i]) : va[i];
            if (ORDER == 2) {
                if (nv == 2) {
                    __hip_atomic_store(nxt + 2 * np + 2 * r, (unsigned long long)__double_as_longlong(vb[i].x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(nxt + 2 * np + 2 * r + 1, (unsigned long long)__double_as_longlong(vb[i].y), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                L2[r] = eb ? cmul(eb[r], vb[i]) : vb[i];
            }
        }
        // re-arm this thread's words of the buffer read last round (all partners have published this ... see below)
        {
            unsigned long long* old = ring + (size_t)b_rearm * ORDER * 2 * np;
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                const int r = row0 + tid + SWEEP_THREADS * i;
#pragma unroll
                for (int v = 0; v < ORDER; ++v) {
                    if (v >= nv) continue;
                    __hip_atomic_store(old + (size_t)v * 2 * np + 2 * r, RESIDENT_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(old + (size_t)v * 2 * np + 2 * r + 1, RESIDENT_SENTINEL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        // the partners' rows: part (part + d) % nsplit, d = 1 .. nsplit - 1, row tid + 1024 i of that part
        for (int d = 1; d < nsplit; ++d) {
            const int prow0 = ((part + d) % nsplit) * part_rows;
            unsigned long long w[SWEEP_RPT][ORDER][2];
            unsigned spins = 0;
            for (;;) {
                bool pending = false;
#pragma unroll
                for (int i = 0; i < SWEEP_RPT; ++i) {
                    const int r = prow0 + tid + SWEEP_THREADS * i;
#pragma unroll
                    for (int v = 0; v < ORDER; ++v) {
                        if (v >= nv) continue;
                        w[i][v][0] = dead ? 0ull : __hip_atomic_load(nxt + (size_t)v * 2 * np + 2 * r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        w[i][v][1] = dead ? 0ull : __hip_atomic_load(nxt + (size_t)v * 2 * np + 2 * r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
#pragma unroll
                for (int i = 0; i < SWEEP_RPT; ++i)
#pragma unroll
                    for (int v = 0; v < ORDER; ++v)
                        if (v < nv) pending |= (w[i][v][0] == RESIDENT_SENTINEL) | (w[i][v][1] == RESIDENT_SENTINEL);
                if (!pending) break;
                __builtin_amdgcn_s_sleep(1);
                ++spins;
                if ((spins & 1023u) == 0 && __hip_atomic_load(sa.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) spins = sa.spin_limit;
                if (spins >= sa.spin_limit) {
                    __hip_atomic_store(sa.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    dead = true;
                    break;
                }
            }
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                const int r = prow0 + tid + SWEEP_THREADS * i;
                const double2 xa = make_double2(__longlong_as_double((long long)w[i][0][0]), __longlong_as_double((long long)w[i][0][1]));
                L1[r] = ea ? cmul(ea[r], xa) : xa;
                if (ORDER == 2) {
                    const double2 xb = nv == 2 ? make_double2(__longlong_as_double((long long)w[i][ORDER - 1][0]),
                                                              __longlong_as_double((long long)w[i][ORDER - 1][1])) : xa;
                    L2[r] = eb ? cmul(eb[r], xb) : xb;
                }
            }
        }
        b_cur = b_nxt;
        __syncthreads();
    };
    // one pass over the operator elements of this thread's rows:
    //   o1 = (sum_e ca_e A_e) . X1,  o2 = (sum_e cb_e A_e) . X2   with (ca, cb) = (c1, c2), or (c2, c1) when swapped
    // Two straight-line loops, no selects: the real-plane slots (A = v: A x = v x), then the imaginary-plane slots
    // (A = i v: A x = v (-x.y, x.x)).
    auto pass = [&](const double2* X1, const double2* X2, bool swapped, double2 (&o1)[SWEEP_RPT], double2 (&o2)[SWEEP_RPT]) {
#pragma unroll
        for (int i = 0; i < SWEEP_RPT; ++i) o1[i] = o2[i] = make_double2(0.0, 0.0);
        const unsigned unp = (unsigned)np;
#if MIDYN_SWEEP_ABLATE == 1   // profiling only: no operator pass at all
        for (int i = 0; i < SWEEP_RPT; ++i) { o1[i] = X1[tid + SWEEP_THREADS * i]; o2[i] = X2[tid + SWEEP_THREADS * i]; }
        return;
#endif
#define MIDYN_SWEEP_SLOT(IM)                                                                         \
        {                                                                                            \
            const double2 cc = cab[e];                                                               \
            const double ca = swapped ? cc.y : cc.x, cb = swapped ? cc.x : cc.y;                     \
            int cl[SWEEP_RPT];                                                                       \
            double v[SWEEP_RPT];                                                                     \
            _Pragma("unroll") for (int i = 0; i < SWEEP_RPT; ++i) {                                  \
                const unsigned idx = (unsigned)e * unp + (unsigned)(row0 + tid + SWEEP_THREADS * i);                        \
                cl[i] = a.col[idx];                                                                  \
                v[i] = a.val[idx];                                                                   \
            }                                                                                        \
            _Pragma("unroll") for (int i = 0; i < SWEEP_RPT; ++i) {                                  \
                const double2 x1 = X1[cl[i]];                                                        \
                const double wa = ca * v[i];                                                         \
                if (IM) {                                                                            \
                    o1[i].x = fma(-wa, x1.y, o1[i].x);                                               \
                    o1[i].y = fma(wa, x1.x, o1[i].y);                                                \
                } else {                                                                             \
                    o1[i].x = fma(wa, x1.x, o1[i].x);                                                \
                    o1[i].y = fma(wa, x1.y, o1[i].y);                                                \
                }                                                                                    \
                if (ORDER == 2) {                                                                    \
                    const double2 x2 = X2[cl[i]];                                                    \
                    const double wb = cb * v[i];                                                     \
                    if (IM) {                                                                        \
                        o2[i].x = fma(-wb, x2.y, o2[i].x);                                           \
                        o2[i].y = fma(wb, x2.x, o2[i].y);                                            \
                    } else {                                                                         \
                        o2[i].x = fma(wb, x2.x, o2[i].x);                                            \
                        o2[i].y = fma(wb, x2.y, o2[i].y);                                            \
                    }                                                                                \
                }                                                                                    \
            }                                                                                        \
        }
#pragma unroll 4
        for (int e = 0; e < a.wre; ++e) MIDYN_SWEEP_SLOT(false)
#pragma unroll 4
        for (int e = a.wre; e < a.wsp; ++e) MIDYN_SWEEP_SLOT(true)
#undef MIDYN_SWEEP_SLOT
    };
    for (int st = 0; st < a.nsteps; ++st) {
        const int r0 = a.rows[3 * st], r1 = a.rows[3 * st + 1];
        const double h = a.hs[st];
        const double2* E0 = a.E ? a.E + (size_t)r0 * np : nullptr;
        const double2* E1 = a.E ? a.E + (size_t)r1 * np : nullptr;
        __syncthreads();
        for (int e = tid; e < a.wsp; e += SWEEP_THREADS) {
            const int seg = stag[e] & 63;
            const bool stat = a.has_static && seg == 0;
            const double* Sb = a.S + (size_t)b * a.inst_stride;
            cab[e] = make_double2(stat ? 1.0 : Sb[(size_t)r0 * a.k + seg - a.has_static],
                                  (ORDER == 2) ? (stat ? 1.0 : Sb[(size_t)r1 * a.k + seg - a.has_static]) : 0.0);
        }
        const int Ks = a.ser_K[st], reps = a.ser_reps[st];
        const bool cheb = Ks > 0;
        const int K = cheb ? Ks : -Ks;
        const double par = a.ser_par[st];
        const double* coef = a.coef + (size_t)st * a.stride;
        for (int rep = 0; rep < reps; ++rep) {
            const double c0 = cheb ? coef[0] : 1.0;
#pragma unroll
            for (int i = 0; i < SWEEP_RPT; ++i) {
                cur[i] = acc[i];
                acc[i] = make_double2(c0 * acc[i].x, c0 * acc[i].y);
                prev[i] = make_double2(0.0, 0.0);
            }
            for (int j = 1; j <= K; ++j) {
                const double f = cheb ? (j == 1 ? 1.0 : 2.0) / par : 1.0 / (par * (double)j);
                all_gather(cur, cur, E0, E1, true);           // L1 = E(t1) o v, L2 = E(t2) o v
                double2 o1[SWEEP_RPT], o2[SWEEP_RPT], w[SWEEP_RPT];
                pass(L1, L2, false, o1, o2);
                if (ORDER == 2) {
                    const double ca = 0.5 * h * f, cb = p2 * h * h * f;
                    double2 u1[SWEEP_RPT], u2[SWEEP_RPT];
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) {
                        const int r = row0 + tid + SWEEP_THREADS * i;
                        u1[i] = E0 ? cmul_conj_a(E0[r], o1[i]) : o1[i];
                        u2[i] = E1 ? cmul_conj_a(E1[r], o2[i]) : o2[i];
                        w[i] = make_double2(ca * (u1[i].x + u2[i].x), ca * (u1[i].y + u2[i].y));
                    }
                    all_gather(u1, u2, E1, E0, false);        // L1 = E(t2) o u1 (for g2), L2 = E(t1) o u2 (for g1)
                    pass(L1, L2, true, o1, o2);
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) {
                        const int r = row0 + tid + SWEEP_THREADS * i;
                        const double2 v1 = E1 ? cmul_conj_a(E1[r], o1[i]) : o1[i];
                        const double2 v2 = E0 ? cmul_conj_a(E0[r], o2[i]) : o2[i];
                        w[i].x += cb * (v1.x - v2.x);
                        w[i].y += cb * (v1.y - v2.y);
                    }
                } else {
                    const double ca = h * f;
#pragma unroll
                    for (int i = 0; i < SWEEP_RPT; ++i) {
                        const int r = row0 + tid + SWEEP_THREADS * i;
                        const double2 u1 = E0 ? cmul_conj_a(E0[r], o1[i]) : o1[i];
                        w[i] = make_double2(ca * u1.x, ca * u1.y);
                    }
                }
#pragma unroll
                for (int i = 0; i < SWEEP_RPT; ++i) {
                    if (cheb) {
                        w[i].x += prev[i].x;
                        w[i].y += prev[i].y;
                        acc[i] = cfma_r(2.0 * coef[j], w[i], acc[i]);
                        prev[i] = cur[i];
                    } else {
                        acc[i].x += w[i].x;
                        acc[i].y += w[i].y;
                    }
                    cur[i] = w[i];
                }
            }
        }
        if (a.save) {
            const int slot = a.save[st];
            if (slot >= 0) {
#pragma unroll
                for (int i = 0; i < SWEEP_RPT; ++i) {
                    const int r = row0 + tid + SWEEP_THREADS * i;
                    if (r < a.n) a.out[((size_t)b * a.P + slot) * a.n + r] = acc[i];
                }
            }
        }
    }
}

// ---- the instantiations that exist (see the end of midyn_kernels.h): midyn_tu_sweep.hip defines MIDYN_TU_SWEEP (the
// one-workgroup-per-instance sweep kernels), midyn_tu_resident.hip MIDYN_TU_RESIDENT (the single-trajectory kernels) -------
#ifdef MIDYN_TU_SWEEP
#define MIDYN_SWEEP_EXTERN
#else
#define MIDYN_SWEEP_EXTERN extern
#endif
#ifdef MIDYN_TU_RESIDENT
#define MIDYN_RESIDENT_EXTERN
#else
#define MIDYN_RESIDENT_EXTERN extern
#endif
#define MIDYN_FOR_ELEMENT_FORM(X, ...) X(__VA_ARGS__, 0) X(__VA_ARGS__, 1) X(__VA_ARGS__, 2)
#define MIDYN_SWEEP_SHAPES(X)   /* (rows per thread, threads) x element form */                                 \
    MIDYN_FOR_ELEMENT_FORM(X, 1, 256) MIDYN_FOR_ELEMENT_FORM(X, 1, 512) MIDYN_FOR_ELEMENT_FORM(X, 1, 1024) \
    MIDYN_FOR_ELEMENT_FORM(X, 2, 1024) MIDYN_FOR_ELEMENT_FORM(X, 4, 1024)      /* (three rows per thread -- n_pad = 3072 exactly -- is not built: census) */
#define MIDYN_X(R_, T_, P_)                                                                           \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_kernel<1, R_, T_, P_>(const SweepArgs);   \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_kernel<2, R_, T_, P_>(const SweepArgs);   \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_rk4_kernel<R_, T_, P_>(const SweepArgs);
MIDYN_SWEEP_SHAPES(MIDYN_X)
#undef MIDYN_X
#define MIDYN_X(R_, T_)   /* element form 3 (flip masks, no elements) */                                           \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_kernel<1, R_, T_, 3>(const SweepArgs);   \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_kernel<2, R_, T_, 3>(const SweepArgs);   \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_rk4_kernel<R_, T_, 3>(const SweepArgs);
MIDYN_X(1, 256) MIDYN_X(1, 512) MIDYN_X(1, 1024) MIDYN_X(2, 1024) MIDYN_X(4, 1024)
#undef MIDYN_X
#define MIDYN_X(O_, P_)                                                                                   \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_duo_kernel<O_, 2, 1024, P_>(const SweepDuoArgs);  \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_duo_kernel<O_, 1, 1024, P_>(const SweepDuoArgs);  \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_duo_kernel<O_, 1, 512, P_>(const SweepDuoArgs);   \
    MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_duo_kernel<O_, 1, 256, P_>(const SweepDuoArgs);
MIDYN_X(1, 1) MIDYN_X(1, 2) MIDYN_X(2, 1) MIDYN_X(2, 2)
#undef MIDYN_X
MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_split_kernel<1, 1>(const SweepSplitArgs);
MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_split_kernel<1, 2>(const SweepSplitArgs);
MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_split_kernel<2, 1>(const SweepSplitArgs);
MIDYN_SWEEP_EXTERN template __global__ void ell_sweep_split_kernel<2, 2>(const SweepSplitArgs);
#define MIDYN_X(NE_, W_) \
    MIDYN_RESIDENT_EXTERN template __global__ void rk4_resident_kernel<NE_, W_, false>(const ResidentArgs); \
    MIDYN_RESIDENT_EXTERN template __global__ void rk4_resident_kernel<NE_, W_, true>(const ResidentArgs);
MIDYN_X(2, 4) MIDYN_X(2, 8) MIDYN_X(4, 4) MIDYN_X(4, 8) MIDYN_X(8, 4) MIDYN_X(8, 8) MIDYN_X(16, 4) MIDYN_X(16, 8)
#undef MIDYN_X
#define MIDYN_X(M_) \
    MIDYN_RESIDENT_EXTERN template __global__ void ell_resident_kernel<M_, 4>(const EllArgs); \
    MIDYN_RESIDENT_EXTERN template __global__ void ell_resident_kernel<M_, 8>(const EllArgs); \
    MIDYN_RESIDENT_EXTERN template __global__ void ell_resident_kernel<M_, 16>(const EllArgs);
MIDYN_X(0) MIDYN_X(1)
#undef MIDYN_X

}  // namespace midyn
