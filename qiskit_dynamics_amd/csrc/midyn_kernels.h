// midyn_kernels.h -- gfx950 (CDNA4) device kernels of libmidyn.
//
// Data layout in HBM (everything complex128 interleaved, row major, leading dims padded):
//   operator stack   ops[seg][n_pad][n_pad]   seg 0 = static operator when present, then the k
//                                             time dependent operators; zero padded to n_pad
//   state block      Y[n_pad][ncol_pad]       one COLUMN per (instance, state column); the column
//                                             index is the fastest axis so that MFMA B-operand
//                                             tiles and C tiles are contiguous 256-B row pieces
//   phases           E[row][n_pad]            exp(d * t_row), one row per distinct time
//   coefficients     S[B][R][k] float64       host layout kept (read once per segment per wave)
//
// Kernels
//   rhs_stream_kernel   single column: HBM-bound fused  sum_j c_j A_j[row,:] . y'  (+ RK4 epilogue)
//   zgemm_seg_kernel    many columns: fp64 MFMA 16x16x4 complex GEMM over the K = nseg*n axis with
//                       the real signal coefficient applied to the B fragment (+ RK4 epilogue);
//                       also the plain zgemm of the expm pipeline
//                       SPARSE instantiation: the (K tile, segment) loop runs over per-row-panel work
//                       lists of the tiles that hold a non-zero (block-sparse stacks; tiles 16x64 .. 128x128)
//   rhs_stream_multi_*  2..8 columns with own coefficients, operator rows read once
//   rhs_blocks_kernel   1..8 columns of a block-sparse stack: only the listed 16x16 operator blocks are read
//   tiny_rk4/expm       small systems: the whole fixed-step solve in one persistent launch
//   gen_eval_kernel     G = scale * Delta(t) o (A_d + sum c_j A_j)
//   small elementwise kernels (lincomb, phase table, transposes, norms, block map, Krylov bookkeeping)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Kernels that are not templates exist once, in midyn.hip; the translation units that only instantiate a kernel family
// (midyn_tu_*.hip define MIDYN_FAMILY_TU; see the end of this file) see them as templates nobody instantiates.
#ifdef MIDYN_FAMILY_TU
#define MIDYN_GLOBAL template <int MIDYN_NOT_IN_THIS_UNIT = 0> __global__
#else
#define MIDYN_GLOBAL __global__
#endif

namespace midyn {

typedef double d4 __attribute__((ext_vector_type(4)));

// Kernel ablation switches (profiling builds only: -DMIDYN_ABLATE; results are wrong when used).
#ifdef MIDYN_ABLATE
#define MIDYN_ABL(g, bit) ((g).ablate & (bit))
#else
#define MIDYN_ABL(g, bit) false
#endif

__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ double2 cmul_conj_a(double2 a, double2 b) {  // conj(a) * b
    return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
// device-coherent load of one complex number (agent-scope atomic loads: served past non-coherent cache lines)
__device__ __forceinline__ double2 coherent_load2(const double2* p) {
    const double* q = reinterpret_cast<const double*>(p);
    return make_double2(__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                        __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ double2 cfma_r(double s, double2 a, double2 c) {  // c + s*a, s real
    return make_double2(fma(s, a.x, c.x), fma(s, a.y, c.y));
}

// ------------------------------------------------------------------------------------------------
// Epilogue shared by the stream and the MFMA kernels.
//   EPI_RHS    out = conj(Ecur[row]) * C                         (GeneratorModel.evaluate_rhs)
//              optionally also yin_next = Enext * out  (the result pre-phased for its consumer)
//   EPI_RK1..4 classic RK4 stages, fixed_step_solvers.py:62-73, fused:
//              k = conj(Ecur) * C
//              1: acc = y + h/6 k         ; yin_next = Enext * (y + h/2 k)
//              2: acc += h/3 k            ; yin_next = Enext * (y + h/2 k)
//              3: acc += h/3 k            ; yin_next = Enext * (y + h k)
//              4: y = acc + h/6 k         ; yin_next = Enext * y
//   EPI_PLAIN  out = alpha * C + beta * Z                          (expm pipeline zgemm)
//   EPI_TAYLOR one term of the Taylor series of the ACTION expm(Omega) y (Omega = h G(t)):
//              k = conj(Ecur) * C ; term = h * k ; acc += term ; yin_next = Enext * term
//   EPI_CHEB   one step of the Chebyshev recurrence of the same action (phi_{k+1} = 2 B phi_k + phi_{k-1}):
//              k = conj(Ecur) * C ; term = alpha * k + Z ; out = term ; acc += beta * term ; yin_next = Enext * term
// The stage input is kept PRE-PHASED (yin = exp(d t_stage) o y_stage) so the contraction kernels
// never touch the frame; yin ping-pongs between two buffers because other workgroups still read
// the current one while this one already writes the next.
// ------------------------------------------------------------------------------------------------
enum { EPI_RHS = 0, EPI_RK1 = 1, EPI_RK2 = 2, EPI_RK3 = 3, EPI_RK4 = 4, EPI_PLAIN = 5, EPI_TAYLOR = 6, EPI_CHEB = 7 };

struct Epilogue {
    int mode;
    int ld;                  // leading dimension (columns) of y/acc/yin_next/out/Z
    double h;                // step size (RK modes)
    double alpha, beta;      // EPI_PLAIN
    const double2* e_cur;    // [n_pad] or nullptr (no frame)
    const double2* e_next;   // [n_pad] or nullptr
    double2* y;              // RK: current state
    double2* acc;            // RK: accumulator
    double2* yin_next;       // RK: next pre-phased stage input
    double2* out;            // EPI_RHS / EPI_PLAIN destination
    const double2* z;        // EPI_PLAIN addend or nullptr
};

// Compile-time mode: every instantiation is straight-line code.  (With a run-time mode chain
// inlined into a store loop, hipcc 7.2 was seen to merge the branches' final stores through one
// pointer register and leave it unset on one path -- see splitk_reduce_kernel -- so all kernels
// dispatch on the mode ONCE, outside their store loops.)
// Split in two so that a tile can LOAD a batch of elements before it stores any (store_tile_t): the arithmetic lives in
// epilogue_finish, the loads in the callers.  v1 = y (RK1-3) or z (PLAIN / CHEB with an addend), v2 = acc.
template <int EMODE>
struct EpiLoads {
    static constexpr bool Y = EMODE == EPI_RK1 || EMODE == EPI_RK2 || EMODE == EPI_RK3;
    static constexpr bool Z = EMODE == EPI_PLAIN || EMODE == EPI_CHEB;
    static constexpr bool ACC = EMODE == EPI_RK2 || EMODE == EPI_RK3 || EMODE == EPI_RK4 || EMODE == EPI_TAYLOR || EMODE == EPI_CHEB;
};

template <int EMODE>
__device__ __forceinline__ void epilogue_finish(const Epilogue& e, size_t idx, double2 c, double2 ecur, double2 enext,
                                                double2 v1, double2 v2) {
    if (EMODE == EPI_PLAIN) {
        double2 r = make_double2(e.alpha * c.x, e.alpha * c.y);
        if (e.z) {
            r.x = fma(e.beta, v1.x, r.x);
            r.y = fma(e.beta, v1.y, r.y);
        }
        e.out[idx] = r;
        return;
    }
    double2 k = c;
    if (e.e_cur) k = cmul_conj_a(ecur, c);
    if (EMODE == EPI_RHS) {
        e.out[idx] = k;
        // optional second output: the same result already phased for the product that consumes it next
        // (saves a separate re-phasing pass over the state block)
        if (e.yin_next) e.yin_next[idx] = e.e_next ? cmul(enext, k) : k;
        return;
    }
    const double2 en = e.e_next ? enext : make_double2(1.0, 0.0);
    const double h = e.h;
    if (EMODE == EPI_TAYLOR) {
        const double2 term = make_double2(h * k.x, h * k.y);
        e.acc[idx] = make_double2(v2.x + term.x, v2.y + term.y);
        e.yin_next[idx] = cmul(en, term);
        return;
    }
    if (EMODE == EPI_CHEB) {
        double2 term = make_double2(e.alpha * k.x, e.alpha * k.y);
        if (e.z) {
            term.x += v1.x;
            term.y += v1.y;
        }
        e.out[idx] = term;
        e.acc[idx] = cfma_r(e.beta, term, v2);
        e.yin_next[idx] = cmul(en, term);
        return;
    }
    double2 next;
    if (EMODE == EPI_RK1) {
        e.acc[idx] = cfma_r(h * (1.0 / 6), k, v1);
        next = cmul(en, cfma_r(0.5 * h, k, v1));
    } else if (EMODE == EPI_RK2) {
        e.acc[idx] = cfma_r(h * (1.0 / 3), k, v2);
        next = cmul(en, cfma_r(0.5 * h, k, v1));
    } else if (EMODE == EPI_RK3) {
        e.acc[idx] = cfma_r(h * (1.0 / 3), k, v2);
        next = cmul(en, cfma_r(h, k, v1));
    } else {  // EPI_RK4
        const double2 yn = cfma_r(h * (1.0 / 6), k, v2);
        e.y[idx] = yn;
        next = cmul(en, yn);
    }
    e.yin_next[idx] = next;
}

template <int EMODE>
__device__ __forceinline__ void apply_epilogue_t(const Epilogue& e, int row, int col, double2 c) {
    const size_t idx = (size_t)row * e.ld + col;
    const double2 zero = make_double2(0.0, 0.0);
    const double2 ecur = (EMODE != EPI_PLAIN && e.e_cur) ? e.e_cur[row] : zero;
    const double2 enext = (EMODE != EPI_PLAIN && e.e_next) ? e.e_next[row] : zero;
    double2 v1 = zero, v2 = zero;
    if (EpiLoads<EMODE>::Y) v1 = e.y[idx];
    if (EpiLoads<EMODE>::Z && e.z) v1 = e.z[idx];
    if (EpiLoads<EMODE>::ACC) v2 = e.acc[idx];
    epilogue_finish<EMODE>(e, idx, c, ecur, enext, v1, v2);
}

// run-time dispatch for a single element (stream kernel: one output per workgroup)
__device__ __forceinline__ void apply_epilogue(const Epilogue& e, int row, int col, double2 c) {
    switch (e.mode) {
        case EPI_RHS: apply_epilogue_t<EPI_RHS>(e, row, col, c); break;
        case EPI_RK1: apply_epilogue_t<EPI_RK1>(e, row, col, c); break;
        case EPI_RK2: apply_epilogue_t<EPI_RK2>(e, row, col, c); break;
        case EPI_RK3: apply_epilogue_t<EPI_RK3>(e, row, col, c); break;
        case EPI_RK4: apply_epilogue_t<EPI_RK4>(e, row, col, c); break;
        case EPI_TAYLOR: apply_epilogue_t<EPI_TAYLOR>(e, row, col, c); break;
        case EPI_CHEB: apply_epilogue_t<EPI_CHEB>(e, row, col, c); break;
        default: apply_epilogue_t<EPI_PLAIN>(e, row, col, c); break;
    }
}

// store a wave's accumulator tile through the epilogue: D[row = (lane>>4) + 4*reg][col = lane & 15]
// One 16-row block of the wave tile at a time: ALL loads of the block first (4 NT elements: phases, y / z, acc), then the
// arithmetic and the stores.  Element by element -- load, wait, store, and the next load behind the store because the
// arrays may alias -- a tile of 64 elements per lane took 64 dependent trips to memory: 43 of the 57 us a launch of the
// cfg 3 contraction spent outside its tile loop (round 3; tools/experiments/fixed_cost.py, ISA).
template <int EMODE, int MT, int NT>
__device__ __forceinline__ void store_tile_t(const Epilogue& e, int row0, int col0, const d4 (&cre)[MT][NT],
                                             const d4 (&cim)[MT][NT]) {
    const double2 zero = make_double2(0.0, 0.0);
    constexpr int RB = MT * NT >= 8 ? 2 : 4;   // rows of a 16-row block per batch (registers: the accumulators are still live)
#pragma unroll
    for (int mb = 0; mb < MT * (4 / RB); ++mb) {
        const int mt = mb / (4 / RB), rb = (mb % (4 / RB)) * RB;
        double2 ecur[RB], enext[RB], v1[RB][NT], v2[RB][NT];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = row0 + mt * 16 + 4 * (rb + r);
            ecur[r] = (EMODE != EPI_PLAIN && e.e_cur) ? e.e_cur[row] : zero;
            enext[r] = (EMODE != EPI_PLAIN && e.e_next) ? e.e_next[row] : zero;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const size_t idx = (size_t)row * e.ld + col0 + nt * 16;
                v1[r][nt] = zero;
                v2[r][nt] = zero;
                if (EpiLoads<EMODE>::Y) v1[r][nt] = e.y[idx];
                if (EpiLoads<EMODE>::Z && e.z) v1[r][nt] = e.z[idx];
                if (EpiLoads<EMODE>::ACC) v2[r][nt] = e.acc[idx];
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = row0 + mt * 16 + 4 * (rb + r);
                const size_t idx = (size_t)row * e.ld + col0 + nt * 16;
                epilogue_finish<EMODE>(e, idx, make_double2(cre[mt][nt][rb + r], cim[mt][nt][rb + r]), ecur[r],
                                            enext[r], v1[r][nt], v2[r][nt]);
            }
    }
}

template <int MT, int NT>
__device__ __forceinline__ void store_tile(const Epilogue& e, int row0, int col0, const d4 (&cre)[MT][NT],
                                           const d4 (&cim)[MT][NT]) {
    switch (e.mode) {  // wave-uniform
        case EPI_RHS: store_tile_t<EPI_RHS, MT, NT>(e, row0, col0, cre, cim); break;
        case EPI_RK1: store_tile_t<EPI_RK1, MT, NT>(e, row0, col0, cre, cim); break;
        case EPI_RK2: store_tile_t<EPI_RK2, MT, NT>(e, row0, col0, cre, cim); break;
        case EPI_RK3: store_tile_t<EPI_RK3, MT, NT>(e, row0, col0, cre, cim); break;
        case EPI_RK4: store_tile_t<EPI_RK4, MT, NT>(e, row0, col0, cre, cim); break;
        case EPI_TAYLOR: store_tile_t<EPI_TAYLOR, MT, NT>(e, row0, col0, cre, cim); break;
        case EPI_CHEB: store_tile_t<EPI_CHEB, MT, NT>(e, row0, col0, cre, cim); break;
        default: store_tile_t<EPI_PLAIN, MT, NT>(e, row0, col0, cre, cim); break;
    }
}

// ------------------------------------------------------------------------------------------------
// rhs_stream_kernel: one state column.  One workgroup (256 threads = 4 waves) per output row; every
// lane streams 16-B complex elements of the (nseg) operator rows with fully coalesced 1-KiB wave
// loads, forms g = sum_seg c_seg A_seg[row][col] in registers (2 FMA per operator) and accumulates
// g * y'[col] (4 FMA); wave shuffle + LDS reduction; lane 0 runs the epilogue.
// Algorithmic traffic per launch: 16*nseg*n^2 + 32 n bytes; arithmetic intensity
// 0.29 F/B -> HBM bound.
// ------------------------------------------------------------------------------------------------
struct StreamArgs {
    const double2* ops;      // [nseg][n_pad][n_pad]
    const int* seg_list;     // active segment list: (seg<<2)|mode, n_act entries
    int n_act;
    int n_pad;
    int has_static;
    const double* coeff;     // [k] coefficients of this evaluation (device) or nullptr
    const double2* yin;      // [n_pad * ld] pre-phased input, column 0 used
    const double2* e_in;     // rhs_blocks_kernel only: phase row applied to yin on load (input NOT pre-phased), or nullptr
    const int2* hull;        // planar streaming kernel: [n_act][n_pad / 16] column ranges (lo, hi) in units of 16
                             // columns outside of which rows 16 rb .. 16 rb + 15 of the segment are exactly zero
                             // (symmetry sectors: contiguous ranges after the sector grouping), or nullptr
    Epilogue epi;
};

template <int UNROLL, int SEGU>
__global__ __launch_bounds__(256) void rhs_stream_kernel(StreamArgs a) {
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.n_pad;
    const size_t plane = (size_t)n * n;
    const int ld = a.epi.ld;
    double2 acc = make_double2(0.0, 0.0);
    // columns handled by this thread: tid, tid+256, ...
    for (int c0 = tid; c0 < n; c0 += 256 * UNROLL) {
        double2 g[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) g[u] = make_double2(0.0, 0.0);
        // SEGU operator rows are in flight together: SEGU*UNROLL independent 16-B loads per lane
        int s = 0;
        for (; s + SEGU <= a.n_act; s += SEGU) {
            double cf[SEGU];
            const double2* p[SEGU];
            double2 v[SEGU][UNROLL];
#pragma unroll
            for (int q = 0; q < SEGU; ++q) {
                const int seg = a.seg_list[s + q] >> 2;
                cf[q] = (a.has_static && seg == 0) ? 1.0 : a.coeff[seg - a.has_static];
                p[q] = a.ops + seg * plane + (size_t)row * n;
            }
#pragma unroll
            for (int q = 0; q < SEGU; ++q)
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int c = c0 + u * 256;
                    v[q][u] = c < n ? p[q][c] : make_double2(0.0, 0.0);
                }
#pragma unroll
            for (int q = 0; q < SEGU; ++q)
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    g[u].x = fma(cf[q], v[q][u].x, g[u].x);
                    g[u].y = fma(cf[q], v[q][u].y, g[u].y);
                }
        }
        for (; s < a.n_act; ++s) {
            const int seg = a.seg_list[s] >> 2;
            const double cf = (a.has_static && seg == 0) ? 1.0 : a.coeff[seg - a.has_static];
            const double2* p = a.ops + seg * plane + (size_t)row * n;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int c = c0 + u * 256;
                if (c < n) {
                    const double2 v = p[c];
                    g[u].x = fma(cf, v.x, g[u].x);
                    g[u].y = fma(cf, v.y, g[u].y);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int c = c0 + u * 256;
            if (c < n) {
                const double2 yv = a.yin[(size_t)c * ld];
                acc.x = fma(g[u].x, yv.x, acc.x);
                acc.x = fma(-g[u].y, yv.y, acc.x);
                acc.y = fma(g[u].x, yv.y, acc.y);
                acc.y = fma(g[u].y, yv.x, acc.y);
            }
        }
    }
    // wave reduction (64 lanes)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc.x += __shfl_down(acc.x, off, 64);
        acc.y += __shfl_down(acc.y, off, 64);
    }
    __shared__ double2 part[4];
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double2 c = part[0];
        c.x += part[1].x + part[2].x + part[3].x;
        c.y += part[1].y + part[2].y + part[3].y;
        apply_epilogue(a.epi, row, 0, c);
    }
}

// ------------------------------------------------------------------------------------------------
// rhs_stream_multi_kernel<C>: the streaming contraction for 2..C state columns (C = 2, 4, 8) that may
// belong to DIFFERENT instances (own coefficient vectors): the operator rows are read ONCE for all
// columns -- the MFMA path would pad to a 64-column tile and take ~58 us at n = 1024 where this takes
// about as long as the one-column kernel (HBM-bound: 2 nseg C + 4 C FMAs per 16 nseg bytes).
// Per lane: g[c] = sum_seg coeff[inst(c)][seg] A_seg[row][col] for each column c, then acc[c] += g[c] y[col][c].
// ------------------------------------------------------------------------------------------------
template <int C, int UNROLL>
__global__ __launch_bounds__(256) void rhs_stream_multi_kernel(StreamArgs a, int ncol, int m_cols, long long inst_stride) {
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.n_pad;
    const size_t plane = (size_t)n * n;
    const int ld = a.epi.ld;
    __shared__ double cf_s[64 * C];       // [seg][column] coefficients (<= 64 active segments)
    __shared__ double2 part[4][C];
    for (int i = tid; i < a.n_act * C; i += 256) {
        const int s = i / C, c = i - s * C;
        const int seg = a.seg_list[s] >> 2;
        double v = 0.0;
        if (c < ncol) v = (a.has_static && seg == 0) ? 1.0 : a.coeff[(size_t)(c / m_cols) * inst_stride + (seg - a.has_static)];
        cf_s[i] = v;
    }
    __syncthreads();
    double2 acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = make_double2(0.0, 0.0);
    for (int c0 = tid; c0 < n; c0 += 256 * UNROLL) {
        double2 g[UNROLL][C];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) g[u][c] = make_double2(0.0, 0.0);
        for (int s = 0; s < a.n_act; ++s) {
            const double2* p = a.ops + (size_t)(a.seg_list[s] >> 2) * plane + (size_t)row * n;
            double2 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int col = c0 + u * 256;
                v[u] = col < n ? p[col] : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const double cf = cf_s[s * C + c];
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    g[u][c].x = fma(cf, v[u].x, g[u][c].x);
                    g[u][c].y = fma(cf, v[u].y, g[u][c].y);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int col = c0 + u * 256;
            if (col < n) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double2 yv = a.yin[(size_t)col * ld + c];
                    acc[c].x = fma(g[u][c].x, yv.x, acc[c].x);
                    acc[c].x = fma(-g[u][c].y, yv.y, acc[c].x);
                    acc[c].y = fma(g[u][c].x, yv.y, acc[c].y);
                    acc[c].y = fma(g[u][c].y, yv.x, acc[c].y);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            acc[c].x += __shfl_down(acc[c].x, off, 64);
            acc[c].y += __shfl_down(acc[c].y, off, 64);
        }
        if ((tid & 63) == 0) part[tid >> 6][c] = acc[c];
    }
    __syncthreads();
    if (tid < C && tid < ncol) {
        double2 r = part[0][tid];
        r.x += part[1][tid].x + part[2][tid].x + part[3][tid].x;
        r.y += part[1][tid].y + part[2][tid].y + part[3][tid].y;
        apply_epilogue(a.epi, row, tid, r);
    }
}

// Planar variant of rhs_stream_multi_kernel for single-plane stacks (see rhs_stream_plane_kernel): only
// the non-zero plane of every operator is streamed (8 B per element); a real-only operator feeds the
// real part of g_c, an imaginary-only one its imaginary part.
template <int C, int UNROLL>
__global__ __launch_bounds__(256) void rhs_stream_multi_plane_kernel(StreamArgs a, const double* planes, int ncol,
                                                                     int m_cols, long long inst_stride) {
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.n_pad;
    const size_t plane = (size_t)n * n;
    const int ld = a.epi.ld;
    __shared__ double cr_s[64 * C], ci_s[64 * C];   // [seg][column]: coefficient on the real / imaginary part
    __shared__ double2 part[4][C];
    for (int i = tid; i < a.n_act * C; i += 256) {
        const int s = i / C, c = i - s * C;
        const int packed = a.seg_list[s];
        const int seg = packed >> 2;
        double v = 0.0;
        if (c < ncol) v = (a.has_static && seg == 0) ? 1.0 : a.coeff[(size_t)(c / m_cols) * inst_stride + (seg - a.has_static)];
        cr_s[i] = (packed & 3) == 2 ? 0.0 : v;
        ci_s[i] = (packed & 3) == 2 ? v : 0.0;
    }
    __syncthreads();
    double2 acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = make_double2(0.0, 0.0);
    for (int c0 = 2 * tid; c0 < n; c0 += 512 * UNROLL) {
        double2 g0[UNROLL][C], g1[UNROLL][C];     // complex g of column pair (col, col + 1)
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) g0[u][c] = g1[u][c] = make_double2(0.0, 0.0);
        for (int s = 0; s < a.n_act; ++s) {
            const double* p = planes + (size_t)s * plane + (size_t)row * n;
            int lo = 0, hi = n;
            if (a.hull) {   // column hull of this row group of the segment (symmetry sectors): exact zeros outside
                const int2 h = a.hull[(size_t)s * (n >> 4) + (row >> 4)];
                lo = h.x << 4;
                hi = h.y << 4;
            }
            double2 v[UNROLL];
            bool in[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int col = c0 + u * 512;
                in[u] = col >= lo && col < hi;
                v[u] = in[u] ? *reinterpret_cast<const double2*>(p + col) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (!in[u]) continue;      // (adding c * 0 would change nothing: skip the fp64 work as well)
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double cr = cr_s[s * C + c], ci = ci_s[s * C + c];
                    g0[u][c].x = fma(cr, v[u].x, g0[u][c].x);
                    g0[u][c].y = fma(ci, v[u].x, g0[u][c].y);
                    g1[u][c].x = fma(cr, v[u].y, g1[u][c].x);
                    g1[u][c].y = fma(ci, v[u].y, g1[u][c].y);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int col = c0 + u * 512;
            if (col < n) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const double2 y0 = a.yin[(size_t)col * ld + c], y1 = a.yin[(size_t)(col + 1) * ld + c];
                    acc[c].x = fma(g0[u][c].x, y0.x, acc[c].x);
                    acc[c].x = fma(-g0[u][c].y, y0.y, acc[c].x);
                    acc[c].y = fma(g0[u][c].x, y0.y, acc[c].y);
                    acc[c].y = fma(g0[u][c].y, y0.x, acc[c].y);
                    acc[c].x = fma(g1[u][c].x, y1.x, acc[c].x);
                    acc[c].x = fma(-g1[u][c].y, y1.y, acc[c].x);
                    acc[c].y = fma(g1[u][c].x, y1.y, acc[c].y);
                    acc[c].y = fma(g1[u][c].y, y1.x, acc[c].y);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            acc[c].x += __shfl_down(acc[c].x, off, 64);
            acc[c].y += __shfl_down(acc[c].y, off, 64);
        }
        if ((tid & 63) == 0) part[tid >> 6][c] = acc[c];
    }
    __syncthreads();
    if (tid < C && tid < ncol) {
        double2 r = part[0][tid];
        r.x += part[1][tid].x + part[2][tid].x + part[3][tid].x;
        r.y += part[1][tid].y + part[2][tid].y + part[3][tid].y;
        apply_epilogue(a.epi, row, tid, r);
    }
}

// ------------------------------------------------------------------------------------------------
// rhs_blocks_kernel<C>: the streaming contraction (1..C columns, own coefficients per column) for
// BLOCK-SPARSE stacks -- operators in a computational or diagonal-frame basis (Pauli strings, Kronecker
// superoperators) are dense arrays in the reference but almost all of their 16 x 16 blocks are exactly
// zero.  The stack keeps, per group of 16 rows, the list of (segment, column chunk) blocks that hold a
// non-zero (`idx[ptr[g] .. ptr[g+1])`, entry = (segment << 16) | chunk, built once per stack from
// block_map_kernel); only those blocks are read, straight from the dense arrays.  The skipped products
// are exact zeros, so the result equals the dense kernels' up to the summation order.
// One workgroup (4 waves) per row group; wave w takes entries w, w+4, ...; a lane owns row (lane >> 2)
// and the four columns 4 (lane & 3) .. +3 of a chunk (64 contiguous bytes), four blocks in flight.
// ------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void rhs_blocks_kernel(StreamArgs a, const int* __restrict__ ptr,
                                                         const int* __restrict__ idx, int ncol, int m_cols,
                                                         long long inst_stride) {
    constexpr int U = C <= 2 ? 8 : 4;  // blocks in flight per wave
    const int rg = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = a.n_pad;
    const size_t plane = (size_t)n * n;
    const int ld = a.epi.ld;
    __shared__ double cf_s[64 * C];  // [segment][column]   (a.n_act = number of SEGMENTS here, <= 64)
    __shared__ double2 part[4][16][C];
    const int r = lane >> 2, q = lane & 3;
    const size_t row_off = (size_t)(rg * 16 + r) * n;
    // The wave's entries (e0 + wave + 4 i) are fetched 64 at a time into one VGPR and handed out with
    // v_readlane: block addresses are scalars, and the first group of block loads is in flight while the
    // coefficients are staged (a row group holds a few dozen blocks: the kernel is a latency chain).
    const int e0 = ptr[rg], e1 = ptr[rg + 1];
    const int nw = (e1 - e0 - wave + 3) >> 2;  // entries of this wave
    double2 acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = make_double2(0.0, 0.0);
    bool staged = false;
    for (int base = 0; base < nw || !staged; base += 64) {
        const int mine = base + lane < nw ? idx[e0 + wave + 4 * (base + lane)] : 0;
        const int m = nw - base < 64 ? nw - base : 64;   // <= 0 for a wave without entries
        for (int i = 0; i < m || !staged; i += U) {
            int seg[U], col0[U];
            double2 v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ent = __builtin_amdgcn_readlane(mine, i + u < m ? i + u : 0);
                seg[u] = i + u < m ? ent >> 16 : -1;
                col0[u] = (ent & 0xffff) * 16 + 4 * q;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (seg[u] >= 0) {
                    const double2* p = a.ops + (size_t)seg[u] * plane + row_off + col0[u];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[u][j] = p[j];
                }
            }
            if (!staged) {  // once, behind the first block loads
                for (int t = tid; t < a.n_act * C; t += 256) {
                    const int sg = t / C, c = t - sg * C;
                    double cv = 0.0;
                    if (c < ncol)
                        cv = (a.has_static && sg == 0) ? 1.0
                                                       : a.coeff[(size_t)(c / m_cols) * inst_stride + (sg - a.has_static)];
                    cf_s[t] = cv;
                }
                __syncthreads();
                staged = true;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (seg[u] < 0) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const double cf = cf_s[seg[u] * C + c];
                        double2 yv = a.yin[(size_t)(col0[u] + j) * ld + c];
                        if (a.e_in) yv = cmul(a.e_in[col0[u] + j], yv);
                        const double gx = cf * v[u][j].x, gy = cf * v[u][j].y;
                        acc[c].x = fma(gx, yv.x, acc[c].x);
                        acc[c].x = fma(-gy, yv.y, acc[c].x);
                        acc[c].y = fma(gx, yv.y, acc[c].y);
                        acc[c].y = fma(gy, yv.x, acc[c].y);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        acc[c].x += __shfl_xor(acc[c].x, 1, 64);
        acc[c].y += __shfl_xor(acc[c].y, 1, 64);
        acc[c].x += __shfl_xor(acc[c].x, 2, 64);
        acc[c].y += __shfl_xor(acc[c].y, 2, 64);
        if (q == 0) part[wave][r][c] = acc[c];
    }
    __syncthreads();
    if (tid < 16 * C) {
        const int rr = tid / C, c = tid - rr * C;
        if (c < ncol) {
            double2 t = part[0][rr][c];
            t.x += part[1][rr][c].x + part[2][rr][c].x + part[3][rr][c].x;
            t.y += part[1][rr][c].y + part[2][rr][c].y + part[3][rr][c].y;
            apply_epilogue(a.epi, rg * 16 + rr, c, t);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// rhs_stream_plane_kernel: the same contraction for stacks whose active operators are ALL purely real
// or purely imaginary (-iH of a real-symmetric H is purely imaginary: cfg 2/3).  The exactly-zero
// plane of every operator is not stored at all (`planes[act][n][n]` doubles, built once by
// extract_planes_kernel), so an evaluation streams 8*n_act*n^2 bytes instead of 16*nseg*n^2: half
// the HBM traffic for bit-identical results (the skipped products are exact zeros).  Every lane
// loads 16 B = the entries of two neighbouring columns.
// ------------------------------------------------------------------------------------------------
template <int UNROLL, int SEGU>
__global__ __launch_bounds__(256) void rhs_stream_plane_kernel(StreamArgs a, const double* planes) {
    const int row = blockIdx.x;
    const int tid = threadIdx.x;
    const int n = a.n_pad;  // multiple of 64
    const size_t plane = (size_t)n * n;
    const int ld = a.epi.ld;
    double2 acc = make_double2(0.0, 0.0);
    for (int c0 = 2 * tid; c0 < n; c0 += 512 * UNROLL) {
        double2 gre[UNROLL], gim[UNROLL];  // (column c, column c+1) of Re g and Im g
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) gre[u] = gim[u] = make_double2(0.0, 0.0);
        int s = 0;
        for (; s + SEGU <= a.n_act; s += SEGU) {
            double cr[SEGU], ci[SEGU];
            const double* p[SEGU];
            double2 v[SEGU][UNROLL];
            int lo[SEGU], hi[SEGU];
#pragma unroll
            for (int q = 0; q < SEGU; ++q) {
                const int packed = a.seg_list[s + q];
                const int seg = packed >> 2;
                const double cf = (a.has_static && seg == 0) ? 1.0 : a.coeff[seg - a.has_static];
                cr[q] = (packed & 3) == 2 ? 0.0 : cf;
                ci[q] = (packed & 3) == 2 ? cf : 0.0;
                p[q] = planes + (size_t)(s + q) * plane + (size_t)row * n;
                lo[q] = 0;
                hi[q] = n;
                if (a.hull) {   // only the column range in which this row of the segment can be non-zero is streamed
                    const int2 h = a.hull[(size_t)(s + q) * (n >> 4) + (row >> 4)];
                    lo[q] = h.x << 4;
                    hi[q] = h.y << 4;
                }
            }
#pragma unroll
            for (int q = 0; q < SEGU; ++q)
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    const int c = c0 + u * 512;
                    v[q][u] = (c >= lo[q] && c < hi[q]) ? *reinterpret_cast<const double2*>(p[q] + c)
                                                        : make_double2(0.0, 0.0);
                }
#pragma unroll
            for (int q = 0; q < SEGU; ++q)
#pragma unroll
                for (int u = 0; u < UNROLL; ++u) {
                    gre[u].x = fma(cr[q], v[q][u].x, gre[u].x);
                    gre[u].y = fma(cr[q], v[q][u].y, gre[u].y);
                    gim[u].x = fma(ci[q], v[q][u].x, gim[u].x);
                    gim[u].y = fma(ci[q], v[q][u].y, gim[u].y);
                }
        }
        for (; s < a.n_act; ++s) {
            const int packed = a.seg_list[s];
            const int seg = packed >> 2;
            const double cf = (a.has_static && seg == 0) ? 1.0 : a.coeff[seg - a.has_static];
            const double cr = (packed & 3) == 2 ? 0.0 : cf, ci = (packed & 3) == 2 ? cf : 0.0;
            const double* p = planes + (size_t)s * plane + (size_t)row * n;
            int lo = 0, hi = n;
            if (a.hull) {
                const int2 h = a.hull[(size_t)s * (n >> 4) + (row >> 4)];
                lo = h.x << 4;
                hi = h.y << 4;
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int c = c0 + u * 512;
                if (c >= lo && c < hi) {
                    const double2 v = *reinterpret_cast<const double2*>(p + c);
                    gre[u].x = fma(cr, v.x, gre[u].x);
                    gre[u].y = fma(cr, v.y, gre[u].y);
                    gim[u].x = fma(ci, v.x, gim[u].x);
                    gim[u].y = fma(ci, v.y, gim[u].y);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int c = c0 + u * 512;
            if (c < n) {
                const double2 y0 = a.yin[(size_t)c * ld], y1 = a.yin[(size_t)(c + 1) * ld];
                acc.x = fma(gre[u].x, y0.x, acc.x);
                acc.x = fma(-gim[u].x, y0.y, acc.x);
                acc.y = fma(gre[u].x, y0.y, acc.y);
                acc.y = fma(gim[u].x, y0.x, acc.y);
                acc.x = fma(gre[u].y, y1.x, acc.x);
                acc.x = fma(-gim[u].y, y1.y, acc.x);
                acc.y = fma(gre[u].y, y1.y, acc.y);
                acc.y = fma(gim[u].y, y1.x, acc.y);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc.x += __shfl_down(acc.x, off, 64);
        acc.y += __shfl_down(acc.y, off, 64);
    }
    __shared__ double2 part[4];
    if ((tid & 63) == 0) part[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        double2 c = part[0];
        c.x += part[1].x + part[2].x + part[3].x;
        c.y += part[1].y + part[2].y + part[3].y;
        apply_epilogue(a.epi, row, 0, c);
    }
}

// ------------------------------------------------------------------------------------------------
// zgemm_seg_kernel:  C[M][N] = sum_{seg in active} A_seg[M][K] . ( B[K][N] o s_seg[col] )
// fp64 MFMA v_mfma_f64_16x16x4_f64; complex product = 4 real MFMAs on (re,im) of the fragments
// (2 when a plane of A_seg is exactly zero).  Fragment maps (one f64 per lane):
//     A[i = lane & 15][k = lane >> 4]     B[k = lane >> 4][j = lane & 15]
//     D[row = (lane >> 4) + 4 * reg][col = lane & 15]
// Both tiles reach LDS by LDS-DMA (global_load_lds, no VGPR staging).  The A tile is laid out per k-step,
// [k / 4][m][k % 4] complex, so that the 64 lanes of a fragment read (16 rows x 4 k) cover 1 KB contiguously: one
// conflict-free ds_read_b128 per lane whose k-step and row block are immediate offsets (read_frags); the B tile is a
// straight row copy [k][BN].  Double-buffered LDS, one barrier per K tile; one MFMA is 64 cycles on a SIMD so each
// 16-deep K tile is >= 8192 MFMA cycles per SIMD.  Every VALU instruction in the tile loop, from either wave of the
// SIMD, takes 3-7 of those cycles (tools/mfma_bank_probe.hip): the loop keeps addresses in SGPRs and immediates.
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    const double2* A;        // segment 0 base
    long long a_seg_stride;  // elements between segments
    int lda;
    const double2* B;
    int ldb;
    int M, N, K;             // multiples of BM / BN / BK
    const int* seg_list;     // (seg<<2)|mode ; mode 0 full, 1 re only, 2 im only
    int n_act;
    int has_static;
    const double* coeff;     // nullptr -> all ones;  else coeff[inst*inst_stride + (seg-has_static)]
    long long inst_stride;
    int m_cols;              // columns per instance
    int n_inst;              // number of instances (columns beyond are padding)
    int batch;               // > 1: blockIdx.y indexes independent problems (EPI_PLAIN only)
    long long batch_a, batch_b, batch_c;  // element strides of A, B and C/Z between problems
    const long long* batch_offs;          // optional [batch][3] element offsets of A, B, C/Z per problem
                                          // (replaces the strides: products of scattered matrices)
    int splits;              // split-K: gridDim = tiles * splits; split z handles K tiles [z*KT/splits, ...)
    double2* partial;        // [splits][M][N] raw partial sums when splits > 1 (epilogue runs in
                             // splitk_reduce_kernel, or in this launch when `sync` is set), nullptr otherwise
    int* sync;               // in-launch split-K reduction: per tile {arrived, departed} counters (all zero between
                             // launches); every workgroup of a tile waits for its `splits` siblings, then sums and
                             // finishes 1/splits of the tile.  Needs ALL workgroups of the launch co-resident.
    int* sync_err;           // host-mapped error word: incremented if the wait gave up (never in normal operation)
    // block-sparse stacks (SPARSE instantiation): the tiles of row panel bm that hold a non-zero, K tile
    // outer / segment inner like the dense loop: work_idx[work_ptr[bm] .. work_ptr[bm+1]) with
    // entry = (K tile << 8) | (seg << 2 | mode); split z of `splits` takes an equal share of the LIST
    const int* work_ptr;
    const int* work_idx;
    int ablate;              // profiling only (midyn_ctx_set_option "ablate"): 1 no barrier, 2 no DMA,
                             // 4 no epilogue, 8 no fragment reads -- results are WRONG when non-zero
    Epilogue epi;
};

constexpr int GEMM_BK = 16;  // K padding granularity (every tile depth divides it)

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// MFMAs of one k-step (4 k) of the wave's TM x TN complex tile.  MODE: 0 = A complex, 1 = A real
// only, 2 = A imaginary only (the other plane of the operator is exactly zero, so the two MFMAs that
// would multiply it are skipped), 3 = decided at run time per segment (mixed stacks).
//   4 = dense complex by the 3M (Karatsuba) scheme: T1 += Ar.Br, T2 += Ai.Bi, T3 += (Ar+Ai).(Br+Bi)
//       with C_re = T1 - T2, C_im = T3 - T1 - T2 formed in the epilogue: 3 real MFMAs per complex
//       product instead of 4 (cre = T1, cim = T3, c2 = T2).
template <int MODE, int MT, int NT>
__device__ __forceinline__ void mfma_kstep(const double2 (&a)[MT], const double2 (&b)[NT], int rt_mode,
                                           const double (&sc)[NT], d4 (&cre)[MT][NT], d4 (&cim)[MT][NT],
                                           d4 (&c2)[MT][NT]) {
    if (MODE == 4) {
        double br[NT], bi[NT], bs[NT], as[MT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            br[nt] = b[nt].x * sc[nt];
            bi[nt] = b[nt].y * sc[nt];
            bs[nt] = br[nt] + bi[nt];
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#ifdef MIDYN_3M_NOAS   // profiling only: what would a precomputed Ar + Ai plane save? (results wrong)
            as[mt] = a[mt].x;
#else
            as[mt] = a[mt].x + a[mt].y;
#endif
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                cre[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt].x, br[nt], cre[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                c2[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt].y, bi[nt], c2[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                cim[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[mt], bs[nt], cim[mt][nt], 0, 0, 0);
        return;
    }
    const bool do_re = MODE == 3 ? (rt_mode != 2) : (MODE != 2);
    const bool do_im = MODE == 3 ? (rt_mode != 1) : (MODE != 1);
    double br[NT], bi[NT], bin[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
#ifdef MIDYN_NOSCALE  // profiling only: how much do the fp64 VALU scalings cost? (results wrong)
        br[nt] = b[nt].x;
        bi[nt] = b[nt].y;
        bin[nt] = b[nt].y;
#else
        br[nt] = b[nt].x * sc[nt];
        bi[nt] = b[nt].y * sc[nt];
        bin[nt] = -bi[nt];
#endif
    }
    if (do_re) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                cre[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt].x, br[nt], cre[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                cim[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt].x, bi[nt], cim[mt][nt], 0, 0, 0);
    }
    if (do_im) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                cre[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt].y, bin[nt], cre[mt][nt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                cim[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt].y, br[nt], cim[mt][nt], 0, 0, 0);
    }
}

// Fragment read of k-step ks from the LDS tiles.
//   A tile in LDS: [k-step][m][4 k] complex -- element (m, k) at ((k / 4) * BM + m) * 4 + ((k % 4) ^ ((m / 4) & 2)).
//   The k-step and the 16-row block are IMMEDIATE offsets of the ds_read (one address register per tile instead of one
//   per k-step: every VALU instruction in the tile loop takes matrix-pipe cycles, tools/mfma_bank_probe.hip).  The XOR
//   is for the lane groups a ds_read_b128 is served in (MI355X_MICROARCH.md: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
//   and the same + 32): each holds 16 different rows, 8 with one k and 8 with the next, and must touch 16 distinct
//   16-byte slots modulo 256 B; unswizzled, rows m and m + 4 (k) / m + 8 (k + 1) share their banks (rocprofv3: bank
//   conflicts 40 % of the LDS cycles).
//   B tile in LDS: [k][BN] complex, straight; `Bb` already points at this lane's row (k % 4) and column.
template <int BM, int BN, int MT, int NT, int HALF = 0>
__device__ __forceinline__ void read_frags(const double2* __restrict__ Ab, const double2* __restrict__ Bb, int ks,
                                           double2 (&a)[MT], double2 (&b)[NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        if (HALF == 0) {
            a[mt] = Ab[ks * (BM * 4) + mt * 64];
        } else {
            // Only one plane (HALF 1: real, 2: imaginary) of A feeds MFMAs: an 8-byte read of that half.  volatile, in
            // the LDS address space: left to itself hipcc pairs such reads as ds_read2st64_b64, whose 32-dword bank
            // modulus makes the rows of a tile collide (round 1: 40 % of the LDS cycles; round 2 kept both halves live
            // to stay on ds_read_b128 instead, at twice the register-file traffic).
            typedef const volatile __attribute__((address_space(3))) double lds_vdouble_t;
            const unsigned o = (unsigned)(unsigned long long)Ab + (unsigned)(ks * (BM * 4) + mt * 64) * 16u + 8u * (HALF - 1);
            const double v = *(lds_vdouble_t*)(unsigned long long)o;
            if (HALF == 1) a[mt].x = v;
            else a[mt].y = v;
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = Bb[ks * (4 * BN) + nt * 16];
}

// second launch-bound argument = waves per SIMD the register allocation must allow: the 4-wave
// configurations are meant to run two workgroups per CU (2 waves per SIMD -> <= 256 registers).
template <int BM, int BN, int WM, int WN, int BK, int MODE, bool SPARSE>
__device__ __forceinline__ void zgemm_seg_body(const GemmArgs& g, const int batch_idx) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr int NWAVE = WM * WN;
    constexpr int TM = BM / WM;  // wave tile
    constexpr int TN = BN / WN;
    constexpr int MT = TM / 16;
    constexpr int NT = TN / 16;
    constexpr int A_CHUNKS = BM * BK / 64;       // 1-KiB DMA pieces per A tile
    constexpr int B_PER_ROW = BN / 64;           // 1-KiB DMA pieces per B tile row
    constexpr int B_CHUNKS = BK * B_PER_ROW;
    constexpr int A_PER_W = A_CHUNKS / NWAVE;
    constexpr int B_PER_W = B_CHUNKS / NWAVE;
    static_assert(A_PER_W * NWAVE == A_CHUNKS && B_PER_W * NWAVE == B_CHUNKS, "DMA split");
    static_assert(BK == 16 || BK == 8, "k-steps of 4");
    // single-plane stacks read only the plane of the operator fragments that feeds MFMAs (8-byte reads: half the VGPR
    // write traffic of the operator fragments; dense single-plane kernel +2.9 %, work lists +0.3 %, round 3)
    constexpr int A_HALF = MODE == 1 ? 1 : (MODE == 2 ? 2 : 0);
    static_assert(BM % 16 == 0, "16-row fragment blocks");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double2* As = reinterpret_cast<double2*>(smem_raw);             // [2][BM][16]
    double2* Bs = As + 2 * BK * BM;                                 // [2][BK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // (wave-uniform by construction; told to the compiler so that the LDS-DMA destinations live in SGPRs -- every VALU
    // instruction in the tile loop takes cycles from the matrix pipe, tools/mfma_bank_probe.hip)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;

    // M-fastest block order: blocks that share an A row panel land on the same XCD (id % 8).
    const int grid_m = g.M / BM;
    const int tiles = grid_m * (g.N / BN);
    const int tile_id = blockIdx.x % tiles;
    const int split = blockIdx.x / tiles;
    const int bm = tile_id % grid_m;
    const int bn = tile_id / grid_m;
    const int m0 = bm * BM;
    const int n0 = bn * BN;

    // split-K: this workgroup contracts K tiles [kt0, kt0 + KT) of every active segment
    const int KT_all = g.K / BK;
    const int KT = KT_all / g.splits;
    int kt0 = split * KT;
    int total = g.n_act * KT;
    // SPARSE: this workgroup's share [w0, w0 + total) of the row panel's tile list (K tiles absolute)
    int w0 = 0;
    if (SPARSE) {
        const int l0 = g.work_ptr[bm], len = g.work_ptr[bm + 1] - l0;
        const int a0 = (int)((long long)len * split / g.splits), a1 = (int)((long long)len * (split + 1) / g.splits);
        w0 = l0 + a0;
        total = a1 - a0;
        kt0 = 0;
    }

    // per-lane column bookkeeping for the coefficient scaling
    const int lcol = lane & 15;
    const int lk = lane >> 4;
    int inst[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        int col = n0 + wn * TN + nt * 16 + lcol;
        int in = col / g.m_cols;
        inst[nt] = in < g.n_inst ? in : g.n_inst - 1;
    }

    d4 cre[MT][NT], cim[MT][NT], c2[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            cre[i][j] = d4{0.0, 0.0, 0.0, 0.0};
            cim[i][j] = d4{0.0, 0.0, 0.0, 0.0};
            c2[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        }

    // LDS-DMA source offsets (per lane, fixed)
    // (byte offsets below 4 GB from the tile's scalar base: the DMA instructions take the scalar-base form, no 64-bit
    // VALU address per instruction; laundered at each use so that the zero-extension is not hoisted into a register pair)
    // A: the 1-KiB piece c of the LDS tile is k-step c / (BM/16), rows 16 (c % (BM/16)) .. +15: lane l fetches
    // row + l / 4 and one of the four k of the k-step (64 contiguous bytes of an operator row per four lanes)
    unsigned a_src[A_PER_W];
#pragma unroll
    for (int p = 0; p < A_PER_W; ++p) {
        const int c = wave + NWAVE * p;
        const int m = (c % (BM / 16)) * 16 + (lane >> 2);
        const int k = (c / (BM / 16)) * 4 + ((lane & 3) ^ ((lane >> 4) & 2));   // (the swizzle of read_frags)
        a_src[p] = (unsigned)(m * g.lda + k) * 16u;
    }
    unsigned b_src[B_PER_W];
#pragma unroll
    for (int p = 0; p < B_PER_W; ++p) {
        const int c = wave + NWAVE * p;
        b_src[p] = (unsigned)((c / B_PER_ROW) * g.ldb + (c % B_PER_ROW) * 64 + lane) * 16u;
    }
    long long off_a = (long long)batch_idx * g.batch_a, off_b = (long long)batch_idx * g.batch_b;
    long long off_c = (long long)batch_idx * g.batch_c;
    if (g.batch_offs) {
        off_a = g.batch_offs[3 * batch_idx];
        off_b = g.batch_offs[3 * batch_idx + 1];
        off_c = g.batch_offs[3 * batch_idx + 2];
    }
    const double2* Abase = g.A + off_a + (size_t)m0 * g.lda + (size_t)kt0 * BK;
    const double2* Bbase = g.B + off_b + n0 + (size_t)kt0 * BK * g.ldb;

    // Loop order: K tile outer, operator segment inner -- the B (state) tile is staged ONCE per K
    // tile and reused by all n_act operator tiles, so per launch the state block is read once per
    // M-block instead of n_act times (memory-side traffic / n_act).
    // The (<= 64 entry) segment table lives in one VGPR (lane s holds entry s) and is read with
    // v_readlane: no scalar-memory load sits between the barrier and the first ds_read of a tile.
    const int seg_vec = lane < g.n_act ? g.seg_list[lane] : 0;
    // global -> LDS direct (no VGPR staging, no ds_write).
    auto dma_a = [&](int kt_, int seg, int buf) {
        const char* Ab = reinterpret_cast<const char*>(Abase + seg * g.a_seg_stride + kt_ * BK);
        double2* Ad = As + buf * BK * BM;
#pragma unroll
        for (int p = 0; p < A_PER_W; ++p) {
            asm volatile("" : "+v"(a_src[p]));   // (in place: no copy)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ab + (unsigned long long)a_src[p]),
                                             (lds_void_t*)(Ad + (wave + NWAVE * p) * 64), 16, 0, 0);
        }
    };
    auto dma_b = [&](int kt_, int buf) {
        const char* Bb = reinterpret_cast<const char*>(Bbase + (size_t)(kt_ * BK) * g.ldb);
        double2* Bd = Bs + buf * BK * BN;
#pragma unroll
        for (int p = 0; p < B_PER_W; ++p) {
            asm volatile("" : "+v"(b_src[p]));
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(Bb + (unsigned long long)b_src[p]),
                                             (lds_void_t*)(Bd + (wave + NWAVE * p) * 64), 16, 0, 0);
        }
    };
    auto load_sc = [&](int seg, double (&scv)[NT]) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (g.coeff == nullptr || (g.has_static && seg == 0)) scv[nt] = 1.0;
            else scv[nt] = g.coeff[inst[nt] * g.inst_stride + (seg - g.has_static)];
        }
    };

    double sc[NT], sc_next[NT];
    int packed = __builtin_amdgcn_readlane(seg_vec, 0);
    // SPARSE: 64 list entries per VGPR (lane l holds entry 64 blk + l), read with v_readlane; the next
    // 64 are fetched one iteration into a block, so no memory load sits on the per-tile path
    int wvec = 0, wvec_next = 0, kt_first = 0;
    if (SPARSE) {
        wvec = lane < total ? g.work_idx[w0 + lane] : 0;
        const int ent = __builtin_amdgcn_readlane(wvec, 0);
        packed = ent & 255;
        kt_first = ent >> 8;
    }
    if (total > 0) {
        dma_a(kt_first, packed >> 2, 0);
        dma_b(kt_first, 0);
        load_sc(packed >> 2, sc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Software pipeline (per K tile, KS = BK/4 k-steps, fragments double-buffered in registers):
    //   k-step ks: read fragments of ks+1, then issue the MFMAs of ks;
    //   after k-step 0's MFMAs are queued: issue the LDS-DMA of the NEXT tile (+ coefficients);
    //   before the LAST k-step's MFMAs: wait for that DMA, barrier, read k-step 0 of the NEXT tile
    //   -- so the barrier release and the LDS latency behind it are covered by the last k-step's
    //   MFMAs instead of idling the matrix pipe.
    constexpr int KS = BK / 4;
    static_assert(KS % 2 == 0, "fragment ping-pong assumes an even number of k-steps");
    // single-plane tiles carry half the MFMAs per K tile: issue the next tile's LDS-DMA BEFORE the
    // first k-step so that it has three k-steps (not two) to land ahead of the barrier
    // issuing it before the first k-step: 1.6 % slower for single-plane tiles (16 MFMAs per k-step), 0.7 % faster for
    // complex ones (32 per k-step) -- tools/gemm_probe.hip, round 3
    constexpr bool DMA_EARLY = MODE == 0;
    double2 fa[MODE == 4 ? 1 : 2][MT], fb[MODE == 4 ? 1 : 2][NT];
    const int a_lane_off = (wm * TM + lcol) * 4 + (lk ^ ((lcol >> 2) & 2));
    const int b_lane_off = lk * BN + wn * TN + lcol;
    if (total > 0) read_frags<BM, BN, MT, NT, A_HALF>(As + a_lane_off, Bs + b_lane_off, 0, fa[0], fb[0]);

    int kt = SPARSE ? kt_first : 0, s = 0;
    int bb = 0;  // SPARSE: LDS buffer of the current B tile (toggles whenever the K tile changes)
    for (int it = 0; it < total; ++it) {
        int s_n = s + 1, kt_n = kt;
        int packed_n = 0, bb_n = bb, ent_n = 0;
        if (SPARSE) {
            const int j = it + 1;
            if ((j & 63) == 0) wvec = wvec_next;
            ent_n = j < total ? __builtin_amdgcn_readlane(wvec, j & 63) : ((kt << 8) | packed);
            if ((j & 63) == 1 && (j | 63) + 1 < total) {
                const int e = (j | 63) + 1 + lane;
                wvec_next = e < total ? g.work_idx[w0 + e] : 0;
            }
            kt_n = ent_n >> 8;
            // (integer arithmetic: a bool here is materialised in a VGPR and read back with v_readfirstlane)
            const int newk = (int)((unsigned)((kt_n - kt) | (kt - kt_n)) >> 31);
            s_n = newk ^ 1;  // 0 = a new B tile is needed
            bb_n = bb ^ newk;
        } else if (s_n == g.n_act) {
            s_n = 0;
            kt_n = kt + 1;
        }
        const double2* Ab = As + (it & 1) * BK * BM + a_lane_off;
        const double2* Bb = Bs + (SPARSE ? bb : (kt & 1)) * BK * BN + b_lane_off;
        const double2* Ab_n = As + ((it + 1) & 1) * BK * BM + a_lane_off;
        const double2* Bb_n = Bs + (SPARSE ? bb_n : (kt_n & 1)) * BK * BN + b_lane_off;
        const int mode = packed & 3;
        const int more = (int)((unsigned)(it + 1 - total) >> 31);   // it + 1 < total, as an integer (stays in an SGPR)
        auto issue_next_tile = [&]() {
            __builtin_amdgcn_sched_barrier(0);
            packed_n = SPARSE ? (ent_n & 255) : __builtin_amdgcn_readlane(seg_vec, s_n);
            if (more && !MIDYN_ABL(g, 2)) {
                dma_a(kt_n, packed_n >> 2, (it + 1) & 1);
                if (s_n == 0) dma_b(kt_n, SPARSE ? bb_n : (kt_n & 1));
            }
            load_sc(packed_n >> 2, sc_next);
            __builtin_amdgcn_sched_barrier(0);
        };
        if (MODE == 4) {
            // 3M keeps three accumulator sets (192 VGPRs): fragments are single-buffered here, the
            // other wave of the SIMD covers the LDS latency between k-steps.
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                mfma_kstep<MODE, MT, NT>(fa[0], fb[0], mode, sc, cre, cim, c2);
                if (ks == 0) issue_next_tile();
                if (ks + 1 < KS) {
                    read_frags<BM, BN, MT, NT, A_HALF>(Ab, Bb, ks + 1, fa[0], fb[0]);
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (more) read_frags<BM, BN, MT, NT, A_HALF>(Ab_n, Bb_n, 0, fa[0], fb[0]);
                }
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < KS) {
                    if (!MIDYN_ABL(g, 8)) read_frags<BM, BN, MT, NT, A_HALF>(Ab, Bb, ks + 1, fa[nxt], fb[nxt]);
                } else {
                    __builtin_amdgcn_sched_barrier(0);
                    if (!MIDYN_ABL(g, 1)) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __syncthreads();
                    }
                    if (more && !MIDYN_ABL(g, 8))
                        read_frags<BM, BN, MT, NT, A_HALF>(Ab_n, Bb_n, 0, fa[nxt], fb[nxt]);
                }
                if (ks == 0 && DMA_EARLY) issue_next_tile();
                mfma_kstep<MODE, MT, NT>(fa[cur], fb[cur], mode, sc, cre, cim, c2);
                if (MODE != 3) {
                    // Interleave the fragment reads of the NEXT k-step between the MFMAs of this one
                    // (2 MFMAs : 1 ds_read) instead of issuing them as a block ahead of the MFMAs:
                    // both waves of a SIMD run in lock-step behind the tile barrier, so a block of
                    // non-MFMA instructions idles the matrix pipe in both at once.
#pragma unroll
                    for (int i = 0; i < MT + NT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    // then exactly the MFMAs of THIS k-step that are left: a larger count (round 2: 64) also captures the
                    // MFMAs of the next k-step of the same scheduling region, whose fragment reads then trail them in a
                    // block and are waited for in front of the tile barrier (ISA of round 3; 1.092 -> 1.082 ms per launch)
                    constexpr int SGB_REST = (MODE == 0 ? 4 : 2) * MT * NT - 2 * (MT + NT);
                    if (SGB_REST > 0) __builtin_amdgcn_sched_group_barrier(0x008, SGB_REST, 0);
                }
                if (ks == 0 && !DMA_EARLY) issue_next_tile();
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) sc[nt] = sc_next[nt];
        s = s_n;
        kt = kt_n;
        bb = bb_n;
        packed = packed_n;
    }

    if (MODE == 4) {  // 3M recombination: C_re = T1 - T2, C_im = T3 - T1 - T2
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double t1 = cre[mt][nt][r], t2 = c2[mt][nt][r], t3 = cim[mt][nt][r];
                    cre[mt][nt][r] = t1 - t2;
                    cim[mt][nt][r] = (t3 - t1) - t2;
                }
    }
    // epilogue: D[row = (lane>>4) + 4*reg][col = lane & 15]
    if (g.partial) {   // raw partial sums (split-K, or one of several launches over chunks of the segment list)
        double2* P = g.partial + (size_t)split * g.M * g.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * TM + mt * 16 + lk + 4 * r;
                    const int col = n0 + wn * TN + nt * 16 + lcol;
                    if (g.sync) {
                        // device-coherent (write-through) stores: the sibling workgroups run on other XCDs with
                        // their own L2, and a full release fence (L2 write-back + invalidate by every workgroup)
                        // was measured at +120 us per launch
                        double* q = reinterpret_cast<double*>(P + (size_t)row * g.N + col);
                        __hip_atomic_store(q, cre[mt][nt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(q + 1, cim[mt][nt][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        P[(size_t)row * g.N + col] = make_double2(cre[mt][nt][r], cim[mt][nt][r]);
                    }
                }
        if (g.sync == nullptr) return;
        // ---- in-launch reduction: wait for the sibling splits of this tile, then sum (in split order: the result
        // is bit-identical to splitk_reduce_kernel's) and finish 1/splits of the tile -- no second launch ----------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's coherent stores have completed
        __syncthreads();
        int* cnt = g.sync + 2 * tile_id;
        if (tid == 0) {
            __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            long long spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.splits) {
                __builtin_amdgcn_s_sleep(4);
                if (++spins > (1ll << 26)) {   // seconds: a sibling never ran (the launch was not co-resident)
                    __hip_atomic_fetch_add(g.sync_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
        }
        __syncthreads();
        {
            constexpr int E = BM * BN;
            const int e0 = (int)((long long)E * split / g.splits), e1 = (int)((long long)E * (split + 1) / g.splits);
            const size_t plane_mn = (size_t)g.M * g.N;
            switch (g.epi.mode) {  // workgroup-uniform; one straight-line instantiation per mode (see store_tile)
#define MIDYN_SLICE(EM_)                                                                         \
    case EM_:                                                                                    \
        for (int e = e0 + tid; e < e1; e += THREADS) {                                           \
            const int row = m0 + e / BN, col = n0 + e % BN;                                      \
            const size_t idx = (size_t)row * g.N + col;                                          \
            double2 c = coherent_load2(g.partial + idx);                                         \
            for (int z = 1; z < g.splits; ++z) {                                                 \
                const double2 v = coherent_load2(g.partial + (size_t)z * plane_mn + idx);        \
                c.x += v.x;                                                                      \
                c.y += v.y;                                                                      \
            }                                                                                    \
            apply_epilogue_t<EM_>(g.epi, row, col, c);                                           \
        }                                                                                        \
        break;
                MIDYN_SLICE(EPI_RHS)
                MIDYN_SLICE(EPI_RK1)
                MIDYN_SLICE(EPI_RK2)
                MIDYN_SLICE(EPI_RK3)
                MIDYN_SLICE(EPI_RK4)
                MIDYN_SLICE(EPI_TAYLOR)
                MIDYN_SLICE(EPI_CHEB)
                default:
                    for (int e = e0 + tid; e < e1; e += THREADS) {
                        const int row = m0 + e / BN, col = n0 + e % BN;
                        const size_t idx = (size_t)row * g.N + col;
                        double2 c = coherent_load2(g.partial + idx);
                        for (int z = 1; z < g.splits; ++z) {
                            const double2 v = coherent_load2(g.partial + (size_t)z * plane_mn + idx);
                            c.x += v.x;
                            c.y += v.y;
                        }
                        apply_epilogue_t<EPI_PLAIN>(g.epi, row, col, c);
                    }
                    break;
#undef MIDYN_SLICE
            }
        }
        __syncthreads();
        if (tid == 0) {   // the last workgroup to leave the tile re-arms its counters for the next launch
            const int d = __hip_atomic_fetch_add(cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == g.splits - 1) {
                __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    Epilogue epi = g.epi;
    if (g.batch > 1 || g.batch_offs) {  // batched plain zgemm: every problem has its own output / addend block
        epi.out += off_c;
        if (epi.z) epi.z += off_c;
    }
    if (MIDYN_ABL(g, 4)) {
        double sdump = 0.0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sdump += cre[mt][nt][0] + cim[mt][nt][3];
        if (sdump == 1.2345e300) g.epi.out[0] = make_double2(sdump, 0.0);
        return;
    }
    store_tile<MT, NT>(epi, m0 + wm * TM + lk, n0 + wn * TN + lcol, cre, cim);
}

template <int BM, int BN, int WM, int WN, int BK, int MODE, int MINW = 2, bool SPARSE = false>
__global__ __launch_bounds__(64 * WM * WN, MINW) void zgemm_seg_kernel(GemmArgs g) {
    zgemm_seg_body<BM, BN, WM, WN, BK, MODE, SPARSE>(g, blockIdx.y);
}

// TWO independent contractions of the same shape in one launch (blockIdx.y selects the argument set): the
// independent products of a Magnus-2 term (g1.v and g2.v, then g2.u1 and g1.u2) on block-sparse stacks are short
// latency-bound launches (~28 us each with an 11 us fixed floor); side by side they share that floor
// and the second workgroup of a CU hides the first one's memory latency.  Results are bit-identical to two launches.
struct GemmPair {
    GemmArgs g[2];
};
template <int BM, int BN, int WM, int WN, int BK, int MODE, int MINW = 2, bool SPARSE = true>
__global__ __launch_bounds__(64 * WM * WN, MINW) void zgemm_seg_pair_kernel(GemmPair gp) {
    zgemm_seg_body<BM, BN, WM, WN, BK, MODE, SPARSE>(gp.g[blockIdx.y], 0);
}

#ifdef MIDYN_EXPERIMENT_LIST_KERNEL   // tools/gemm_probe.hip: the hand-scheduled work-list kernel of round 3 (not adopted)
#include MIDYN_EXPERIMENT_LIST_KERNEL
#endif

// ------------------------------------------------------------------------------------------------
// zgemm_plane_kernel: the batched RHS contraction for stacks whose operators are ALL single-plane
// (purely real or purely imaginary, e.g. -iH with real H in a real eigenbasis).  The non-zero plane of
// every segment is kept as a planar fp64 copy, so an operator tile is half the bytes and TWO
// operator tiles (segments s, s+1 of the same K tile) share one LDS stage and one barrier:
//     per barrier 2 x (4 k-steps x 16 MFMAs) per wave instead of 1 x -- the fixed per-tile cost
//     (barrier, DMA issue, LDS latency) is amortised over twice the matrix work.
//   A stage in LDS: [2 seg][128 m][16 k] f64, element (m,k) at 8-B slot k ^ ((m>>1 & 7) << 1): every
//     32-lane service group of ds_read_b64 (16 rows x 2 k) then touches 32 distinct 8-B bank pairs.
//     The XOR is an even number, so 16-B DMA granules (slot pairs) stay intact; it is applied on the
//     DMA source address.
//   B stage: [16 k][128 n] complex as in zgemm_seg_kernel.
//   real plane:  C_re += a * b_re, C_im += a * b_im;   imaginary plane: C_re += a * (-b_im), C_im += a * b_re.
// ------------------------------------------------------------------------------------------------
struct PlaneArgs {
    const double* planes;     // [n_act][M][lda] non-zero plane of the active segments (active order)
    long long seg_stride;     // elements between segments
    GemmArgs g;               // everything else (A / a_seg_stride unused)
};

template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, 2) void zgemm_plane_kernel(PlaneArgs pa) {
    const GemmArgs& g = pa.g;
    constexpr int BM = 128, BN = 128, BK = 16;
    constexpr int NWAVE = WM * WN;
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    constexpr int A_CHUNKS = BM / 8;             // 1-KiB DMA pieces per segment tile (8 rows x 128 B)
    constexpr int A_PER_W = A_CHUNKS / NWAVE;    // per segment
    constexpr int B_PER_ROW = BN / 64;
    constexpr int B_PER_W = BK * B_PER_ROW / NWAVE;
    static_assert(A_PER_W * NWAVE == A_CHUNKS, "DMA split");

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* As = reinterpret_cast<double*>(smem_raw);                        // [2 stage][2 seg][BM][BK]
    double2* Bs = reinterpret_cast<double2*>(As + 2 * 2 * BM * BK);          // [2][BK][BN]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int grid_m = g.M / BM;
    const int tiles = grid_m * (g.N / BN);
    const int tile_id = blockIdx.x % tiles;
    const int split = blockIdx.x / tiles;
    const int m0 = (tile_id % grid_m) * BM;
    const int n0 = (tile_id / grid_m) * BN;
    const int KT = (g.K / BK) / g.splits;
    const int kt0 = split * KT;
    const int npair = (g.n_act + 1) >> 1;
    const int total = npair * KT;

    const int lcol = lane & 15, lk = lane >> 4;
    int inst[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int in = (n0 + wn * TN + nt * 16 + lcol) / g.m_cols;
        inst[nt] = in < g.n_inst ? in : g.n_inst - 1;
    }
    d4 cre[MT][NT], cim[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            cre[i][j] = d4{0.0, 0.0, 0.0, 0.0};
            cim[i][j] = d4{0.0, 0.0, 0.0, 0.0};
        }

    // DMA source offsets: lane l of chunk c -> row m = 8c + l/8, 16-B granule p = l % 8 holding
    // k = 2 (p ^ ((m >> 1) & 7)) and k + 1
    int a_src[A_PER_W];
#pragma unroll
    for (int p = 0; p < A_PER_W; ++p) {
        const int m = (wave + NWAVE * p) * 8 + (lane >> 3);
        a_src[p] = m * g.lda + 2 * ((lane & 7) ^ ((m >> 1) & 7));
    }
    int b_src[B_PER_W];
#pragma unroll
    for (int p = 0; p < B_PER_W; ++p) {
        const int c = wave + NWAVE * p;
        b_src[p] = (c / B_PER_ROW) * g.ldb + (c % B_PER_ROW) * 64 + lane;
    }
    const double* Pbase = pa.planes + (size_t)m0 * g.lda + (size_t)kt0 * BK;
    const double2* Bbase = g.B + n0 + (size_t)kt0 * BK * g.ldb;
    const int seg_vec = lane < g.n_act ? g.seg_list[lane] : 0;   // (seg << 2) | mode, active order

    auto dma_a = [&](int kt_, int pair, int stage) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s_act = 2 * pair + q;
            if (s_act < g.n_act) {
                const double* Ab = Pbase + s_act * pa.seg_stride + kt_ * BK;
                double* Ad = As + (stage * 2 + q) * BM * BK;
#pragma unroll
                for (int p = 0; p < A_PER_W; ++p)
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ab + a_src[p]),
                                                     (lds_void_t*)(Ad + (wave + NWAVE * p) * 128), 16, 0, 0);
            }
        }
    };
    auto dma_b = [&](int kt_, int buf) {
        const double2* Bb = Bbase + (size_t)(kt_ * BK) * g.ldb;
        double2* Bd = Bs + buf * BK * BN;
#pragma unroll
        for (int p = 0; p < B_PER_W; ++p)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(Bb + b_src[p]),
                                             (lds_void_t*)(Bd + (wave + NWAVE * p) * 64), 16, 0, 0);
    };
    // per pair: coefficients of both segments and their plane kinds (bit0: seg0 imaginary, bit1: seg1)
    auto load_sc = [&](int pair, double (&scv)[2][NT], int& kinds) {
        kinds = 0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int s_act = 2 * pair + q;
            const int packed = __builtin_amdgcn_readlane(seg_vec, s_act < g.n_act ? s_act : 0);
            const int seg = packed >> 2;
            if ((packed & 3) == 2) kinds |= (1 << q);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (s_act >= g.n_act) scv[q][nt] = 0.0;
                else if (g.coeff == nullptr || (g.has_static && seg == 0)) scv[q][nt] = 1.0;
                else scv[q][nt] = g.coeff[inst[nt] * g.inst_stride + (seg - g.has_static)];
            }
        }
    };
    // fragments of k-step ks of BOTH segment tiles + the shared B fragment
    auto read_frags = [&](const double* Ab, const double2* Bb, int ks, double (&a)[2][MT], double2 (&b)[NT]) {
        const int k = ks * 4 + lk;
        const int slot = k ^ ((lcol >> 1) << 1);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[q][mt] = Ab[q * BM * BK + mt * 16 * BK + slot];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = Bb[k * BN + nt * 16];
    };
    auto mfma_kstep2 = [&](const double (&a)[2][MT], const double2 (&b)[NT], const double (&scv)[2][NT], int kinds,
                           bool second) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (q == 1 && !second) break;
            const bool imag = (kinds >> q) & 1;
            double o_re[NT], o_im[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const double br = b[nt].x * scv[q][nt], bi = b[nt].y * scv[q][nt];
                o_re[nt] = imag ? -bi : br;   // operand feeding C_re
                o_im[nt] = imag ? br : bi;    // operand feeding C_im
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cre[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][mt], o_re[nt], cre[mt][nt], 0, 0, 0);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    cim[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q][mt], o_im[nt], cim[mt][nt], 0, 0, 0);
        }
    };

    double sc[2][NT], sc_next[2][NT];
    int kinds = 0, kinds_next = 0;
    if (total > 0) {
        dma_a(0, 0, 0);
        dma_b(0, 0);
        load_sc(0, sc, kinds);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr int KS = BK / 4;
    double fa[2][2][MT];
    double2 fb[2][NT];
    const int a_lane_off = (wm * TM + lcol) * BK;
    const int b_lane_off = wn * TN + lcol;
    if (total > 0) read_frags(As + a_lane_off, Bs + b_lane_off, 0, fa[0], fb[0]);

    int kt = 0, pr = 0;
    for (int it = 0; it < total; ++it) {
        int pr_n = pr + 1, kt_n = kt;
        if (pr_n == npair) {
            pr_n = 0;
            kt_n = kt + 1;
        }
        const bool second = 2 * pr + 1 < g.n_act;
        const double* Ab = As + (it & 1) * 2 * BM * BK + a_lane_off;
        const double2* Bb = Bs + (kt & 1) * BK * BN + b_lane_off;
        const double* Ab_n = As + ((it + 1) & 1) * 2 * BM * BK + a_lane_off;
        const double2* Bb_n = Bs + (kt_n & 1) * BK * BN + b_lane_off;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < KS) {
                read_frags(Ab, Bb, ks + 1, fa[nxt], fb[nxt]);
            } else {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (it + 1 < total) read_frags(Ab_n, Bb_n, 0, fa[nxt], fb[nxt]);
            }
            mfma_kstep2(fa[cur], fb[cur], sc, kinds, second);
            if (ks == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (it + 1 < total) {
                    dma_a(kt_n, pr_n, (it + 1) & 1);
                    if (pr_n == 0) dma_b(kt_n, kt_n & 1);
                }
                load_sc(it + 1 < total ? pr_n : pr, sc_next, kinds_next);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) sc[q][nt] = sc_next[q][nt];
        kinds = kinds_next;
        pr = pr_n;
        kt = kt_n;
    }

    if (g.splits > 1) {
        double2* P = g.partial + (size_t)split * g.M * g.N;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * TM + mt * 16 + lk + 4 * r;
                    const int col = n0 + wn * TN + nt * 16 + lcol;
                    P[(size_t)row * g.N + col] = make_double2(cre[mt][nt][r], cim[mt][nt][r]);
                }
        return;
    }
    store_tile<MT, NT>(g.epi, m0 + wm * TM + lk, n0 + wn * TN + lcol, cre, cim);
}

// planes[act][i] = non-zero plane of active segment `act` (mode 1: real part, mode 2: imaginary part)
MIDYN_GLOBAL __launch_bounds__(256) void extract_planes_kernel(const double2* ops, const int* seg_act, int n_act,
                                                             size_t plane, double* planes) {
    const size_t total = plane * n_act;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int a = (int)(idx / plane);
        const size_t e = idx - (size_t)a * plane;
        const int packed = seg_act[a];
        const double2 v = ops[(size_t)(packed >> 2) * plane + e];
        planes[idx] = (packed & 3) == 2 ? v.y : v.x;
    }
}

// Sum the split-K partials and run the fused epilogue (one thread per element).
// The epilogue mode is a TEMPLATE parameter here: with the run-time mode chain of apply_epilogue
// inlined into this loop, hipcc (ROCm 7.2) merges the branches' final stores through one pointer
// register and leaves it unset on the EPI_RK4 path (store to a garbage address; found as a memory
// fault, verified in the ISA).  A compile-time mode gives every instantiation straight-line code.
// SPLITS > 0: the number of partials is a compile-time constant, so all their loads are in flight together (with a
// run-time count the loop issues load - wait - add per partial: 8 serial memory round trips per element).
template <int EMODE, int SPLITS>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const double2* partial, int splits, int M, int N,
                                                            Epilogue epi) {
    const size_t total = (size_t)M * N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        double2 c;
        if (SPLITS > 0) {
            double2 v[SPLITS > 0 ? SPLITS : 1];
#pragma unroll
            for (int z = 0; z < SPLITS; ++z) v[z] = partial[(size_t)z * total + idx];
            c = v[0];
#pragma unroll
            for (int z = 1; z < SPLITS; ++z) {   // same order as the generic loop: bit-identical sums
                c.x += v[z].x;
                c.y += v[z].y;
            }
        } else {
            c = partial[idx];
            for (int z = 1; z < splits; ++z) {
                const double2 v = partial[(size_t)z * total + idx];
                c.x += v.x;
                c.y += v.y;
            }
        }
        const int row = (int)(idx / N);
        apply_epilogue_t<EMODE>(epi, row, (int)(idx - (size_t)row * N), c);
    }
}

// ------------------------------------------------------------------------------------------------
// gen_eval_kernel:  G[r][c] = scale * conj(E[r]) * E[c] * (sum_seg cf_seg A_seg[r][c])
// (RotatingFrame._conjugate_and_add, rotating_frame.py:350-353, fused with the linear combination
// of operator_collections.py:113-114).  HBM bound: reads 16*nseg B, writes 16 B per element.
// ------------------------------------------------------------------------------------------------
struct GenArgs {
    const double2* ops;
    const int* seg_list;
    int n_act;
    int n_pad;
    int has_static;
    const double* coeff;
    const double2* e;  // [n_pad] phases or nullptr
    double scale;
    double2* out;      // [batch][n_pad][n_pad]
    int batch;         // instances evaluated at once (own coefficient row)
    long long coeff_stride;  // doubles between the coefficient rows of consecutive instances
    long long e_stride;      // 0: all instances share the phases e (same time); else elements between rows
    const double* scale_vec; // optional per-instance factor on top of `scale` (step sizes)
};

MIDYN_GLOBAL __launch_bounds__(256) void gen_eval_kernel(GenArgs a) {
    const int n = a.n_pad;
    const size_t plane = (size_t)n * n;
    const size_t total = plane * (a.batch > 0 ? a.batch : 1);
    for (size_t gidx = (size_t)blockIdx.x * 256 + threadIdx.x; gidx < total;
         gidx += (size_t)gridDim.x * 256) {
        const size_t inst = gidx / plane;
        const size_t idx = gidx - inst * plane;
        const double* coeff = a.coeff ? a.coeff + inst * a.coeff_stride : nullptr;
        double2 gsum = make_double2(0.0, 0.0);
        for (int s = 0; s < a.n_act; ++s) {
            const int seg = a.seg_list[s] >> 2;
            const double cf = (a.has_static && seg == 0) ? 1.0 : coeff[seg - a.has_static];
            const double2 v = a.ops[seg * plane + idx];
            gsum.x = fma(cf, v.x, gsum.x);
            gsum.y = fma(cf, v.y, gsum.y);
        }
        if (a.e) {
            const int r = (int)(idx / n);
            const int c = (int)(idx - (size_t)r * n);
            const double2* e = a.e + inst * a.e_stride;
            const double2 ph = cmul_conj_a(e[r], e[c]);
            gsum = cmul(ph, gsum);
        }
        const double sc = a.scale_vec ? a.scale * a.scale_vec[inst] : a.scale;
        a.out[gidx] = make_double2(sc * gsum.x, sc * gsum.y);
    }
}

// out = sum_i alpha_i X_i + gamma * I   (real alpha; up to 4 terms) on [n][n] with leading dim n
struct LinArgs {
    const double2* x[4];
    double alpha[4];
    int nterms;
    double gamma;
    int n;
    int batch;   // matrices laid out back to back; gamma * I is added to each
    double2* out;
};

MIDYN_GLOBAL __launch_bounds__(256) void lincomb_kernel(LinArgs a) {
    const size_t plane = (size_t)a.n * a.n;
    const size_t total = plane * (a.batch > 0 ? a.batch : 1);
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        double2 r = make_double2(0.0, 0.0);
        for (int i = 0; i < a.nterms; ++i) {
            const double2 v = a.x[i][idx];
            r.x = fma(a.alpha[i], v.x, r.x);
            r.y = fma(a.alpha[i], v.y, r.y);
        }
        if (a.gamma != 0.0) {
            const size_t e = idx % plane;
            const size_t rr = e / a.n;
            if (e - rr * a.n == rr) r.x += a.gamma;
        }
        a.out[idx] = r;
    }
}

// E[r][i] = exp(i * frame_im[i] * times[r])  (rotating_frame.py:255,350: exp(frame_diag * t))
MIDYN_GLOBAL __launch_bounds__(256) void phase_table_kernel(const double* frame_im, const double* times,
                                                          int n_pad, int rows, double2* E) {
    const size_t total = (size_t)rows * n_pad;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / n_pad);
        const int i = (int)(idx - (size_t)r * n_pad);
        double s, c;
        sincos(frame_im[i] * times[r], &s, &c);
        E[idx] = make_double2(c, s);
    }
}

// ------------------------------------------------------------------------------------------------
// signal_table_kernel: the coefficient table S[B][R][k] evaluated on the
// device from piecewise-constant samples + carriers,
//     S[b][r][j] = sum_{terms q of signal j of instance b} Re[ f_q(t_r) exp(i(2 pi nu_q t_r + phi_q)) ],
//     f_q(t) = samples_q[ clip( floor_divide(t - t0_q, dt_q), -1, len_q ) ]   (zero outside the window),
// i.e. SignalList.__call__ over SignalSums of DiscreteSignals (signals/signals.py:148-155,302-311,
// 574-577,801-803).  The arithmetic ORDER of the reference is kept: carrier argument (2 pi nu) * t + phi
// with separately rounded products, real part f.x cos - f.y sin, terms summed left to right; the
// sample index uses NumPy's floor_divide algorithm (fmod based) so that times that sit exactly on
// a sample edge pick the same sample as the reference.  dt_q == 0 marks a constant envelope.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double np_floor_divide(double a, double b) {
// plain operators under contract(off): hipcc's default -ffp-contract=fast would fuse a*b+c (and the
// __dmul_rn/__dadd_rn helpers are ordinary inline functions that carry the contract flag with them)
#pragma clang fp contract(off)
    // numpy/core/src/npymath: npy_divmod -> floor_divide for doubles
    double mod = fmod(a, b);
    double div = (a - mod) / b;
    if (mod != 0.0) {
        if ((b < 0) != (mod < 0)) div = div - 1.0;
    }
    if (div != 0.0) {
        double fl = floor(div);
        if (div - fl > 0.5) fl = fl + 1.0;
        return fl;
    }
    return copysign(0.0, a / b);
}

struct SigTableArgs {
    int B, R, k;
    const double* times;          // [R]
    const long long* term_ptr;    // [B*k + 1]
    const double* params;         // [n_terms][4] = dt, start_time, carrier_freq, phase
    const long long* sample_ptr;  // [n_terms][2] = offset, length (terms may share samples)
    const double2* samples;
    double* S;                    // [B][R][k]
};

MIDYN_GLOBAL __launch_bounds__(256) void signal_table_kernel(SigTableArgs a) {
// the reference rounds the carrier product, the phase addition and the two products of the real part
// separately: no FMA contraction in this kernel
#pragma clang fp contract(off)
    const size_t total = (size_t)a.B * a.R * a.k;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int j = (int)(idx % a.k);
        const size_t br = idx / a.k;
        const int r = (int)(br % a.R);
        const size_t b = br / a.R;
        const double t = a.times[r];
        const long long lo = a.term_ptr[b * a.k + j], hi = a.term_ptr[b * a.k + j + 1];
        double acc = 0.0;
        for (long long q = lo; q < hi; ++q) {
            const double dt = a.params[4 * q], t0 = a.params[4 * q + 1];
            const double freq = a.params[4 * q + 2], ph = a.params[4 * q + 3];
            const long long s0 = a.sample_ptr[2 * q], ns = a.sample_ptr[2 * q + 1];
            double2 f = make_double2(0.0, 0.0);
            if (dt == 0.0) {
                f = a.samples[s0];
            } else {
                const double fd = np_floor_divide(t - t0, dt);
                if (fd >= 0.0 && fd < (double)ns) f = a.samples[s0 + (long long)fd];
            }
            const double two_pi_nu = 6.283185307179586 * freq;
            const double arg = t * two_pi_nu + ph;
            double sn, cs;
            sincos(arg, &sn, &cs);
            const double re = f.x * cs - f.y * sn;
            acc = (q == lo) ? re : acc + re;
        }
        a.S[idx] = acc;
    }
}

// out[i][0..w) = src[rows[i]][0..w)  (rows of the coefficient table / of the phase table gathered
// into the order of a batch of time steps)
MIDYN_GLOBAL __launch_bounds__(256) void gather_rows_kernel(const double* src, const int* rows, int count,
                                                          int w, double* out) {
    const size_t total = (size_t)count * w;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t i = idx / w;
        out[idx] = src[(size_t)rows[i] * w + (idx - i * w)];
    }
}

// Host batch layout [B][n][m] (or shared [n][m]) -> state pool [n_pad][ldy] with instance b in the
// 64-aligned column block b*mpad .. b*mpad+m (every instance is its own GEMM operand), and back from
// the ping-pong half flags[b] of the pool.
MIDYN_GLOBAL __launch_bounds__(256) void scatter_padded_kernel(const double2* src, int shared, int B, int n, int m,
                                                             int mpad, int ldy, double2* y) {
    const size_t total = (size_t)B * n * m;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / ((size_t)n * m));
        const size_t rem = idx - (size_t)b * n * m;
        const int i = (int)(rem / m);
        const int j = (int)(rem - (size_t)i * m);
        y[(size_t)i * ldy + (size_t)b * mpad + j] = shared ? src[rem] : src[idx];
    }
}

MIDYN_GLOBAL __launch_bounds__(256) void gather_padded_kernel(const double2* y, const int* flags, size_t half, int B,
                                                            int n, int m, int mpad, int ldy, double2* dst) {
    const size_t total = (size_t)B * n * m;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / ((size_t)n * m));
        const size_t rem = idx - (size_t)b * n * m;
        const int i = (int)(rem / m);
        const int j = (int)(rem - (size_t)i * m);
        dst[idx] = y[(size_t)flags[b] * half + (size_t)i * ldy + (size_t)b * mpad + j];
    }
}

// A[t][j] = (mono[t][j], 0) for t < nb, j < M;  A[t][M] = (1, 0) when a constant term follows the
// M expansion terms; zero padding elsewhere (row f4: GEMM operand of the polynomial evaluation)
MIDYN_GLOBAL __launch_bounds__(256) void mono_operand_kernel(const double* mono, int nb, int M, int has_const,
                                                           int T, int K, double2* A) {
    const size_t total = (size_t)T * K;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int t = (int)(idx / K);
        const int j = (int)(idx - (size_t)t * K);
        double v = 0.0;
        if (t < nb) {
            if (j < M) v = mono[(size_t)t * M + j];
            else if (j == M && has_const) v = 1.0;
        }
        A[idx] = make_double2(v, 0.0);
    }
}

// The same operand for TWO steps per row (midyn_expansion: blockdiag packing, n <= 32): row t = (instance b, i < ns) holds the monomials
// of step i in columns [0, Mc) and those of step i + ns in [Mc, 2 Mc) (Mc = M + constant); mono is [instances][2 ns][M].
MIDYN_GLOBAL __launch_bounds__(256) void mono_operand2_kernel(const double* mono, int nb, int M, int has_const, int T, int K2, int ns,
                                                            double2* A) {
    const size_t total = (size_t)T * K2;
    const int Mc = M + has_const;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int t = (int)(idx / K2);
        const int j = (int)(idx - (size_t)t * K2);
        double v = 0.0;
        if (t < nb && j < 2 * Mc) {
            const int b = t / ns, i = t % ns, hi = j >= Mc ? 1 : 0, jj = j - hi * Mc;
            if (jj < M) v = mono[((size_t)b * 2 * ns + (size_t)hi * ns + i) * M + jj];
            else v = 1.0;
        }
        A[idx] = make_double2(v, 0.0);
    }
}

// The operand built from the Chebyshev COEFFICIENTS of the steps (midyn_expansion_solve_coeffs): c is [instances][n_vars][nsteps], labels
// [M][order] (-1 behind a label's last index); monomial I of a step = c[lab[0]] * (c[lab[1]] * (... c[lab[last]])) -- the association of
// perturbative.compute_monomials, so the table equals the host's bit for bit.  Row t of the chunk is table row row0 + t = (instance b,
// i < ns); pack: columns [0, Mc) belong to step i, [Mc, 2 Mc) to step i + ns (two steps per padded block), else columns [0, Mc) to step i.
MIDYN_GLOBAL __launch_bounds__(256) void mono_operand_labels_kernel(const double* __restrict__ c, const int* __restrict__ labels, int order,
                                                                  int n_vars, int nsteps, long long row0, int nb, int M, int has_const, int T,
                                                                  int K, int ns, int pack, double2* A) {
    const size_t total = (size_t)T * K;
    const int Mc = M + has_const;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int t = (int)(idx / K);
        const int j = (int)(idx - (size_t)t * K);
        double v = 0.0;
        if (t < nb && j < (pack ? 2 * Mc : Mc)) {
            const long long r = row0 + t;
            const long long b = r / ns;
            const int hi = pack && j >= Mc ? 1 : 0, jj = j - hi * Mc;
            if (jj < M) {
                const double* cb = c + (size_t)b * n_vars * nsteps + (size_t)(r - b * ns) + (size_t)hi * ns;
                const int* lab = labels + (size_t)jj * order;
                int k = order - 1;
                while (k > 0 && lab[k] < 0) --k;
                v = cb[(size_t)lab[k] * nsteps];
                for (--k; k >= 0; --k) v = cb[(size_t)lab[k] * nsteps] * v;
            } else {
                v = 1.0;
            }
        }
        A[idx] = make_double2(v, 0.0);
    }
}

// rows [0, blk) of every column -> rows [blk, 2 blk) (dir > 0) or back (dir < 0); the rows left behind are zeroed
MIDYN_GLOBAL __launch_bounds__(256) void expansion_shift_kernel(double2* Y, int ldy, int cols, int blk, int dir) {
    const size_t total = (size_t)blk * cols;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / cols), c = (int)(idx - (size_t)r * cols);
        double2* lo = Y + (size_t)r * ldy + c;
        double2* hi = Y + (size_t)(r + blk) * ldy + c;
        if (dir > 0) {
            *hi = *lo;
            *lo = make_double2(0.0, 0.0);
        } else {
            *lo = *hi;
            *hi = make_double2(0.0, 0.0);
        }
    }
}

// Host batch layout [B][n][m] (or shared [n][m]) -> device column block [n_pad][ld]; also writes the
// pre-phased copy yin = E o y.  Padding rows/cols are zeroed by the caller (memset).
MIDYN_GLOBAL __launch_bounds__(256) void scatter_state_kernel(const double2* src, int shared, int B, int n,
                                                            int m, int ld, const double2* e,
                                                            double2* y, double2* yin) {
    const size_t total = (size_t)B * n * m;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / ((size_t)n * m));
        const size_t rem = idx - (size_t)b * n * m;
        const int i = (int)(rem / m);
        const int j = (int)(rem - (size_t)i * m);
        const double2 v = shared ? src[rem] : src[idx];
        const size_t dst = (size_t)i * ld + (size_t)b * m + j;
        if (y) y[dst] = v;
        if (yin) yin[dst] = e ? cmul(e[i], v) : v;
    }
}

// device column block [n_pad][ld] -> out[b][slot][i][j] with out laid out [B][P][n][m]
MIDYN_GLOBAL __launch_bounds__(256) void gather_state_kernel(const double2* y, int B, int n, int m, int ld,
                                                           int P, int slot, double2* out) {
    const size_t total = (size_t)B * n * m;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / ((size_t)n * m));
        const size_t rem = idx - (size_t)b * n * m;
        const int i = (int)(rem / m);
        const int j = (int)(rem - (size_t)i * m);
        out[((size_t)b * P + slot) * n * m + rem] = y[(size_t)i * ld + (size_t)b * m + j];
    }
}

// ------------------------------------------------------------------------------------------------
// tiny_rk4_kernel: a WHOLE fixed-step RK4 solve in one launch for small systems (n <= 64 rows whose
// operator stack fits a 64 KB LDS slice): one wave per state column (instance b = column / m), lane r
// owns row r of the state; the operator stack sits in LDS, TRANSPOSED ([seg][col][row]) so that the
// lanes of a wave read consecutive 16-B slots while the input element u[col] is an LDS broadcast.
// The per-stage launches of the batched path cost ~10 us each whatever the size; here a stage of a
// 2..4-qubit model is a few hundred cycles, and thousands of sweep instances run concurrently
// (pulse-shape sweeps of small systems are the reference's everyday workload).
// Same arithmetic as the fused epilogue of the batched path (EPI_RK1..4): k = conj(E) o (C (E o v)).
// ------------------------------------------------------------------------------------------------
struct TinyArgs {
    const double2* ops;      // [nseg][n_pad][n_pad] device stack
    const int* seg_list;     // active segments, (seg << 2) | mode
    int n_act, n, n_pad, has_static, k;
    const double* S;         // [B][R][k] coefficient table (device)
    long long inst_stride;   // R * k
    const double2* E;        // [R][n_pad] phases or nullptr
    const int* rows;         // [nsteps][3] table rows of every step (device)
    const double* hs;        // [nsteps]
    const int* save;         // [nsteps] output slot or -1 (device) or nullptr
    int step_begin, step_end;
    int ncol, m, ld, P;
    double2* y;              // [n_pad][ld] state (column block), updated in place
    double2* out;            // [B][P][n][m] saved states or nullptr
};

// k = G(t) v for the lane's row r (all lanes of the wave call it together); e = E(t)[r], cf = the
// coefficient row of that time in LDS (one double per operator, static operator excluded)
__device__ __forceinline__ double2 tiny_rhs(bool has_e, int n_act, const int* cidx, const double2* At, double2* u,
                                            const double* cf_row, int n, int r, bool active, double2 e, double2 v) {
    if (active) u[r] = has_e ? cmul(e, v) : v;
    // LDS operations of one wave execute in order, so the reads below see every lane's write; only the
    // compiler must be kept from moving them (no fence: it would also wait for the global prefetches)
    __builtin_amdgcn_wave_barrier();
    double2 acc = make_double2(0.0, 0.0);
    for (int s = 0; s < n_act; ++s) {
        const int ci = cidx[s];                        // -1: static operator (coefficient 1)
        const double cf = ci < 0 ? 1.0 : cf_row[ci];
        const double2* As = At + (size_t)s * n * n + r;
        double2 p0 = make_double2(0.0, 0.0), p1 = make_double2(0.0, 0.0);
        int c = 0;
        // 8 columns per round: all 16 LDS reads are issued before the first FMA needs one (the loop is
        // otherwise bound by LDS latency, one dependent read per iteration), two accumulation chains
        for (; c + 8 <= n; c += 8) {
            double2 av[8], uv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                av[q] = As[(size_t)(c + q) * n];
                uv[q] = u[c + q];
            }
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                p0.x = fma(av[q].x, uv[q].x, p0.x);
                p0.y = fma(av[q].x, uv[q].y, p0.y);
                p1.x = fma(av[q + 1].x, uv[q + 1].x, p1.x);
                p1.y = fma(av[q + 1].x, uv[q + 1].y, p1.y);
                p0.x = fma(-av[q].y, uv[q].y, p0.x);
                p0.y = fma(av[q].y, uv[q].x, p0.y);
                p1.x = fma(-av[q + 1].y, uv[q + 1].y, p1.x);
                p1.y = fma(av[q + 1].y, uv[q + 1].x, p1.y);
            }
        }
        for (; c + 2 <= n; c += 2) {
            const double2 a0 = As[(size_t)c * n], a1 = As[(size_t)(c + 1) * n];
            const double2 u0 = u[c], u1 = u[c + 1];
            p0.x = fma(a0.x, u0.x, p0.x);
            p0.y = fma(a0.x, u0.y, p0.y);
            p1.x = fma(a1.x, u1.x, p1.x);
            p1.y = fma(a1.x, u1.y, p1.y);
            p0.x = fma(-a0.y, u0.y, p0.x);
            p0.y = fma(a0.y, u0.x, p0.y);
            p1.x = fma(-a1.y, u1.y, p1.x);
            p1.y = fma(a1.y, u1.x, p1.y);
        }
        if (c < n) {
            const double2 a0 = As[(size_t)c * n];
            const double2 u0 = u[c];
            p0.x = fma(a0.x, u0.x, p0.x);
            p0.y = fma(a0.x, u0.y, p0.y);
            p0.x = fma(-a0.y, u0.y, p0.x);
            p0.y = fma(a0.y, u0.x, p0.y);
        }
        acc.x = fma(cf, p0.x + p1.x, acc.x);
        acc.y = fma(cf, p0.y + p1.y, acc.y);
    }
    __builtin_amdgcn_wave_barrier();   // everyone has read u before the next product overwrites it
    return has_e ? cmul_conj_a(e, acc) : acc;
}

// common prologue: operator stack -> LDS (transposed), wave/column bookkeeping.  LDS per workgroup:
// At [n_act][n][n] | u [4 waves][n] | coefficient rows [4 waves][2 buffers][3 rows][k]
#define MIDYN_TINY_PROLOGUE                                                                       \
    extern __shared__ __attribute__((aligned(16))) char tiny_smem[];                              \
    double2* At = reinterpret_cast<double2*>(tiny_smem);                                          \
    const int n = a.n;                                                                            \
    double2* ubuf = At + (size_t)a.n_act * n * n;                                                 \
    double* cbuf_all = reinterpret_cast<double*>(ubuf + 4 * n);                                   \
    int* cidx = reinterpret_cast<int*>(cbuf_all + (size_t)4 * 2 * 3 * (a.k > 0 ? a.k : 1));        \
    const bool has_e = a.E != nullptr;                                                            \
    const int n_act = a.n_act;                                                                    \
    if (threadIdx.x < a.n_act) {                                                                  \
        const int seg_ = a.seg_list[threadIdx.x] >> 2;                                            \
        cidx[threadIdx.x] = (a.has_static && seg_ == 0) ? -1 : seg_ - a.has_static;               \
    }                                                                                             \
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                \
    for (int idx = tid; idx < a.n_act * n * n; idx += 256) {                                      \
        const int s_ = idx / (n * n);                                                             \
        const int rem_ = idx - s_ * n * n;                                                        \
        const int c_ = rem_ / n, r_ = rem_ - c_ * n;                                              \
        At[idx] = a.ops[((size_t)(a.seg_list[s_] >> 2) * a.n_pad + r_) * a.n_pad + c_];           \
    }                                                                                             \
    __syncthreads();                                                                              \
    const int col = blockIdx.x * 4 + wave;                                                        \
    if (col >= a.ncol) return; /* whole wave leaves together (no further block barriers) */       \
    const int inst = col / a.m;                                                                   \
    const bool active = lane < n;                                                                 \
    const int r = active ? lane : 0;                                                              \
    double2* u = ubuf + wave * n;                                                                 \
    const int kk_ = a.k > 0 ? a.k : 1;                                                            \
    double* cbuf = cbuf_all + (size_t)wave * 2 * 3 * kk_;                                         \
    const double* Sb = a.S ? a.S + (size_t)inst * a.inst_stride : nullptr;                        \
    double2 y = active ? a.y[(size_t)r * a.ld + col] : make_double2(0.0, 0.0);

// Software pipeline over the steps (a wave issues in order, so a load whose address depends on another
// load would stall the whole step): iteration st issues (1) the loads of the step TABLE entries (rows,
// h, save slot) of step st+2, (2) the phase / coefficient loads of step st+1 through the table entries
// fetched one iteration earlier, then computes step st from registers / LDS filled an iteration ago.
// Global-memory latency (~1 us, twice for the dependent pair) is thereby off the critical path.
#define MIDYN_TINY_FETCH(ra_, rb_, rc_, e0_, e1_, e2_, cval_)                                     \
    {                                                                                             \
        if (a.E) {                                                                                \
            e0_ = a.E[(size_t)(ra_) * a.n_pad + r];                                               \
            e1_ = a.E[(size_t)(rb_) * a.n_pad + r];                                               \
            e2_ = a.E[(size_t)(rc_) * a.n_pad + r];                                               \
        }                                                                                         \
        for (int q_ = 0; q_ < TINY_CQ; ++q_) {                                                    \
            const int i_ = lane + 64 * q_;                                                        \
            const int which_ = i_ / kk_;                                                          \
            const int row_ = which_ == 0 ? (ra_) : (which_ == 1 ? (rb_) : (rc_));                 \
            cval_[q_] = (Sb && i_ < 3 * a.k) ? Sb[(size_t)row_ * a.k + (i_ - which_ * kk_)] : 0.0; \
        }                                                                                         \
    }
struct TinyStep {
    int r0, r1, r2, save;
    double h;
};
__device__ __forceinline__ TinyStep tiny_step_entry(const TinyArgs& a, int st) {
    TinyStep t;
    t.r0 = a.rows[3 * st];
    t.r1 = a.rows[3 * st + 1];
    t.r2 = a.rows[3 * st + 2];
    t.h = a.hs[st];
    t.save = a.save ? a.save[st] : -1;
    return t;
}
#define MIDYN_TINY_STASH(buf_, cval_)                                                             \
    {                                                                                             \
        for (int q_ = 0; q_ < TINY_CQ; ++q_) {                                                    \
            const int i_ = lane + 64 * q_;                                                        \
            if (i_ < 3 * a.k) cbuf[(size_t)(buf_) * 3 * kk_ + i_] = cval_[q_];                    \
        }                                                                                         \
        __builtin_amdgcn_wave_barrier();                                                          \
    }

#define MIDYN_TINY_SAVE(slot_)                                                                    \
    if (a.out && active && (slot_) >= 0)                                                          \
        a.out[(((size_t)inst * a.P + (slot_)) * n + r) * a.m + (col - inst * a.m)] = y;

constexpr int TINY_CQ = 2;  // coefficient values fetched per lane: supports 3 k <= 128 (k <= 42 operators)

// pipeline prologue shared by the two kernels: cur = entry of the first step (its data fetched and
// stashed), nxt = entry of the second step
#define MIDYN_TINY_PIPE_PROLOGUE                                                                  \
    const double2 one = make_double2(1.0, 0.0);                                                   \
    double2 e0 = one, e1 = one, e2 = one, f0 = one, f1 = one, f2 = one;                           \
    double cv[TINY_CQ], cn[TINY_CQ];                                                              \
    TinyStep cur{0, 0, 0, -1, 0.0}, nxt{0, 0, 0, -1, 0.0}, nn{0, 0, 0, -1, 0.0};                  \
    if (a.step_begin < a.step_end) {                                                              \
        cur = tiny_step_entry(a, a.step_begin);                                                   \
        if (a.step_begin + 1 < a.step_end) nxt = tiny_step_entry(a, a.step_begin + 1);            \
        MIDYN_TINY_FETCH(cur.r0, cur.r1, cur.r2, e0, e1, e2, cv)                                  \
        MIDYN_TINY_STASH(0, cv)                                                                   \
    }                                                                                             \
    int cb = 0;
#define MIDYN_TINY_PIPE_ISSUE(st_)                                                                \
    if ((st_) + 2 < a.step_end) nn = tiny_step_entry(a, (st_) + 2);                               \
    if ((st_) + 1 < a.step_end) MIDYN_TINY_FETCH(nxt.r0, nxt.r1, nxt.r2, f0, f1, f2, cn)
#define MIDYN_TINY_PIPE_ROTATE(st_)                                                               \
    if ((st_) + 1 < a.step_end) {                                                                 \
        MIDYN_TINY_STASH(cb ^ 1, cn)                                                              \
        e0 = f0;                                                                                  \
        e1 = f1;                                                                                  \
        e2 = f2;                                                                                  \
        cb ^= 1;                                                                                  \
        cur = nxt;                                                                                \
        nxt = nn;                                                                                 \
    }

MIDYN_GLOBAL __launch_bounds__(256) void tiny_rk4_kernel(TinyArgs a) {
    MIDYN_TINY_PROLOGUE
    MIDYN_TINY_PIPE_PROLOGUE
    for (int st = a.step_begin; st < a.step_end; ++st) {
        MIDYN_TINY_PIPE_ISSUE(st)
        const double* c0 = cbuf + (size_t)cb * 3 * kk_;
        const double *c1 = c0 + kk_, *c2 = c0 + 2 * kk_;
        const double h = cur.h;
        double2 kk = tiny_rhs(has_e, n_act, cidx, At, u, c0, n, r, active, e0, y);
        double2 acc = cfma_r(h * (1.0 / 6), kk, y);
        double2 yt = cfma_r(0.5 * h, kk, y);
        kk = tiny_rhs(has_e, n_act, cidx, At, u, c1, n, r, active, e1, yt);
        acc = cfma_r(h * (1.0 / 3), kk, acc);
        yt = cfma_r(0.5 * h, kk, y);
        kk = tiny_rhs(has_e, n_act, cidx, At, u, c1, n, r, active, e1, yt);
        acc = cfma_r(h * (1.0 / 3), kk, acc);
        yt = cfma_r(h, kk, y);
        kk = tiny_rhs(has_e, n_act, cidx, At, u, c2, n, r, active, e2, yt);
        y = cfma_r(h * (1.0 / 6), kk, acc);
        MIDYN_TINY_SAVE(cur.save)
        MIDYN_TINY_PIPE_ROTATE(st)
    }
    if (active) a.y[(size_t)r * a.ld + col] = y;
}

// The expm ACTION of the Magnus-1/2 step (see expm_action_solve) for small systems, whole solve in
// one launch: per step the host-chosen Taylor degree and scaling (packed as deg | sc << 8 in the
// third row slot of the step table, which Magnus orders 1 and 2 do not use), products through tiny_rhs.
// order 1: Omega v = h G(t1) v;  order 2: h/2 (g1 v + g2 v) + sqrt(3)/12 h^2 (g2 g1 v - g1 g2 v).
// deg[st] < 0: Chebyshev series with K = -deg[st] terms in sc[st] repetitions, rho[st] the norm bound of Omega and
// cheb[st * cheb_stride + k] = J_k(rho / sc) (see the Chebyshev paragraph of expm_action_solve).
MIDYN_GLOBAL __launch_bounds__(256) void tiny_expm_kernel(TinyArgs a, int magnus_order, const int* deg, const int* sc,
                                                        const double* rho, const double* cheb, int cheb_stride) {
    MIDYN_TINY_PROLOGUE
    const double p2 = 0.14433756729740643;  // sqrt(3) / 12
    MIDYN_TINY_PIPE_PROLOGUE
    int p_cur = a.step_begin < a.step_end ? deg[a.step_begin] : 0, s_cur = a.step_begin < a.step_end ? sc[a.step_begin] : 0;
    int p_nxt = 0, s_nxt = 0;
    for (int st = a.step_begin; st < a.step_end; ++st) {
        if (st + 1 < a.step_end) {
            p_nxt = deg[st + 1];
            s_nxt = sc[st + 1];
        }
        MIDYN_TINY_PIPE_ISSUE(st)
        const double* c0 = cbuf + (size_t)cb * 3 * kk_;
        const double* c1 = c0 + kk_;
        const double h = cur.h;
        const int p = p_cur, s = s_cur;
        if (p < 0) {
            // Omega phi for the current step (Magnus 1: h G(t1); Magnus 2: commutator free), scaled by w
            auto omega = [&](double2 v, double w) -> double2 {
                if (magnus_order == 1) {
                    const double2 kk = tiny_rhs(has_e, n_act, cidx, At, u, c0, n, r, active, e0, v);
                    return make_double2(w * h * kk.x, w * h * kk.y);
                }
                const double2 u1 = tiny_rhs(has_e, n_act, cidx, At, u, c0, n, r, active, e0, v);
                const double2 u2 = tiny_rhs(has_e, n_act, cidx, At, u, c1, n, r, active, e1, v);
                const double2 v1 = tiny_rhs(has_e, n_act, cidx, At, u, c1, n, r, active, e1, u1);
                const double2 v2 = tiny_rhs(has_e, n_act, cidx, At, u, c0, n, r, active, e0, u2);
                const double ca = 0.5 * h * w, cb2 = p2 * h * h * w;
                return make_double2(ca * (u1.x + u2.x) + cb2 * (v1.x - v2.x), ca * (u1.y + u2.y) + cb2 * (v1.y - v2.y));
            };
            const int K = -p;
            const double* cj = cheb + (size_t)st * cheb_stride;
            const double w = 1.0 / rho[st];
            for (int rep = 0; rep < s; ++rep) {
                double2 prev = y;
                double2 phi = omega(y, w);                         // phi_1 = B y
                double2 acc = make_double2(cj[0] * y.x + 2.0 * cj[1] * phi.x, cj[0] * y.y + 2.0 * cj[1] * phi.y);
                for (int k = 1; k < K; ++k) {                      // phi_{k+1} = 2 B phi_k + phi_{k-1}
                    const double2 t = omega(phi, 2.0 * w);
                    const double2 nxt = make_double2(t.x + prev.x, t.y + prev.y);
                    acc.x += 2.0 * cj[k + 1] * nxt.x;
                    acc.y += 2.0 * cj[k + 1] * nxt.y;
                    prev = phi;
                    phi = nxt;
                }
                y = acc;
            }
        }
        for (int rep = 0; rep < s && p >= 0; ++rep) {
            double2 acc = y, term = y;
            for (int j = 1; j <= p; ++j) {
                const double f = 1.0 / ((double)s * j);
                if (magnus_order == 1) {
                    const double2 kk = tiny_rhs(has_e, n_act, cidx, At, u, c0, n, r, active, e0, term);
                    term = make_double2(h * f * kk.x, h * f * kk.y);
                } else {
                    const double2 u1 = tiny_rhs(has_e, n_act, cidx, At, u, c0, n, r, active, e0, term);
                    const double2 u2 = tiny_rhs(has_e, n_act, cidx, At, u, c1, n, r, active, e1, term);
                    const double2 v1 = tiny_rhs(has_e, n_act, cidx, At, u, c1, n, r, active, e1, u1);
                    const double2 v2 = tiny_rhs(has_e, n_act, cidx, At, u, c0, n, r, active, e0, u2);
                    const double ca = 0.5 * h * f, cb2 = p2 * h * h * f;
                    term = make_double2(ca * (u1.x + u2.x) + cb2 * (v1.x - v2.x), ca * (u1.y + u2.y) + cb2 * (v1.y - v2.y));
                }
                acc.x += term.x;
                acc.y += term.y;
            }
            y = acc;
        }
        MIDYN_TINY_SAVE(cur.save)
        MIDYN_TINY_PIPE_ROTATE(st)
        p_cur = p_nxt;
        s_cur = s_nxt;
    }
    if (active) a.y[(size_t)r * a.ld + col] = y;
}

// ------------------------------------------------------------------------------------------------
// Krylov (Arnoldi) pieces for the expm action of ONE large-norm generator on ONE vector (see
// krylov_expm_step in midyn.hip).  V holds the orthonormal basis, row i = vector v_i of length n
// (leading dimension ldv); Hm is the (m+1) x m Hessenberg matrix in a 64 x 64 row-major block.
// ------------------------------------------------------------------------------------------------
// hc[i] = v_i^H w for i < cnt (one workgroup per i); add != 0 accumulates into Hm[i][j] instead of setting it
MIDYN_GLOBAL __launch_bounds__(256) void krylov_dot_kernel(const double2* V, int ldv, const double2* w, int n, int j,
                                                         int add, double2* hc, double2* Hm) {
    const int i = blockIdx.x;
    const double2* v = V + (size_t)i * ldv;
    double2 acc = make_double2(0.0, 0.0);
    for (int r = threadIdx.x; r < n; r += 256) {
        const double2 a = v[r], b = w[r];
        acc.x = fma(a.x, b.x, acc.x);
        acc.x = fma(a.y, b.y, acc.x);
        acc.y = fma(a.x, b.y, acc.y);
        acc.y = fma(-a.y, b.x, acc.y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        acc.x += __shfl_down(acc.x, off, 64);
        acc.y += __shfl_down(acc.y, off, 64);
    }
    __shared__ double2 part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double2 t = part[0];
        t.x += part[1].x + part[2].x + part[3].x;
        t.y += part[1].y + part[2].y + part[3].y;
        hc[i] = t;
        double2* h = Hm + (size_t)i * 64 + j;
        if (add) *h = make_double2(h->x + t.x, h->y + t.y);
        else *h = t;
    }
}

// out[r] = base[r] + sign * sum_{i < cnt} coef[i * cstride] v_i[r]     (base may be out itself or nullptr)
MIDYN_GLOBAL __launch_bounds__(256) void krylov_axpy_kernel(const double2* V, int ldv, const double2* coef, int cstride,
                                                          int cnt, double sign, const double2* base, int n,
                                                          double2* out) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < n; r += gridDim.x * 256) {
        double2 acc = base ? base[r] : make_double2(0.0, 0.0);
        for (int i = 0; i < cnt; ++i) {
            const double2 c = coef[(size_t)i * cstride];
            const double2 v = V[(size_t)i * ldv + r];
            acc.x += sign * (c.x * v.x - c.y * v.y);
            acc.y += sign * (c.x * v.y + c.y * v.x);
        }
        out[r] = acc;
    }
}

// nrm = ||w||_2 (one workgroup); stores it to *nrm_out and, when Hm != nullptr, to Hm[j+1][j]; then
// vnext = w / nrm (nrm == 0: vnext = 0, the subspace is invariant)
MIDYN_GLOBAL __launch_bounds__(1024) void krylov_norm_scale_kernel(const double2* w, int n, int j, double2* Hm,
                                                                 double* nrm_out, double2* vnext) {
    double acc = 0.0;
    for (int r = threadIdx.x; r < n; r += 1024) {
        const double2 a = w[r];
        acc = fma(a.x, a.x, acc);
        acc = fma(a.y, a.y, acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[16];
    __shared__ double nrm_s;
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += part[i];
        nrm_s = sqrt(t);
        *nrm_out = nrm_s;
        if (Hm) Hm[(size_t)(j + 1) * 64 + j] = make_double2(nrm_s, 0.0);
    }
    __syncthreads();
    const double inv = nrm_s > 0.0 ? 1.0 / nrm_s : 0.0;
    for (int r = threadIdx.x; r < n; r += 1024) vnext[r] = make_double2(w[r].x * inv, w[r].y * inv);
}

// small = h * Hm[0..m)[0..m), zero elsewhere (64 x 64 block that dev_expm_inplace exponentiates)
MIDYN_GLOBAL __launch_bounds__(256) void krylov_small_kernel(const double2* Hm, int m, double h, double2* small) {
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < 64 * 64; idx += gridDim.x * 256) {
        const int i = idx >> 6, j = idx & 63;
        const double2 v = (i < m && j < m) ? Hm[idx] : make_double2(0.0, 0.0);
        small[idx] = make_double2(h * v.x, h * v.y);
    }
}

// Saad's a-posteriori estimate  beta * |h h_{m+1,m}| * |e_m^T expm(h H_m) e_1|  -> err[0]; err[1] = beta
MIDYN_GLOBAL void krylov_err_kernel(const double2* E, const double2* Hm, int m, double h, const double* beta, double* err) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double2 e = E[(size_t)(m - 1) * 64];
        const double hn = Hm[(size_t)m * 64 + (m - 1)].x;
        err[0] = beta[0] * fabs(h) * hn * hypot(e.x, e.y);
        err[1] = beta[0];
    }
}

// coef[i] = beta * E[i][0]
MIDYN_GLOBAL void krylov_coef_kernel(const double2* E, int m, const double* beta, double2* coef) {
    const int i = threadIdx.x;
    if (i < m) coef[i] = make_double2(beta[0] * E[(size_t)i * 64].x, beta[0] * E[(size_t)i * 64].y);
}

// yin = E o y  (re-phasing when a step starts from a time that is not the previous step's end)
MIDYN_GLOBAL __launch_bounds__(256) void rephase_kernel(const double2* y, const double2* e, int n_pad, int ld,
                                                      double2* yin) {
    const size_t total = (size_t)n_pad * ld;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int i = (int)(idx / ld);
        yin[idx] = e ? cmul(e[i], y[idx]) : y[idx];
    }
}

// dst = a * src (state blocks)
MIDYN_GLOBAL __launch_bounds__(256) void scale_copy_kernel(const double2* src, double a, size_t total, double2* dst) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const double2 v = src[idx];
        dst[idx] = make_double2(a * v.x, a * v.y);
    }
}

// One Taylor term of the commutator-free Magnus-2 action (fixed_step_solvers.py:348-363 applied to a
// vector):  w = a (u1 + u2) + b (v1 - v2),  u_i = g_i term, v1 = g2 u1, v2 = g1 u2;  acc += w.
// Optionally also writes the two phased copies of w the next term's products read (wp0 = e0 o w, wp1 = e1 o w).
// z / beta: the Chebyshev recurrence of the same action, w = (...) + z (phi_{k+1} = 2 B phi_k + phi_{k-1}; z may alias
// w), acc += beta w; the Taylor series passes z = nullptr, beta = 1.
MIDYN_GLOBAL __launch_bounds__(256) void magnus2_term_kernel(const double2* u1, const double2* u2, const double2* v1,
                                                           const double2* v2, double a, double b, size_t total,
                                                           double2* w, double2* acc, const double2* e0,
                                                           const double2* e1, int ld, double2* wp0, double2* wp1,
                                                           const double2* z, double beta) {
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const double2 p = u1[idx], q = u2[idx], r = v1[idx], t = v2[idx];
        double2 o = make_double2(a * (p.x + q.x) + b * (r.x - t.x), a * (p.y + q.y) + b * (r.y - t.y));
        if (z) {
            const double2 zz = z[idx];
            o.x += zz.x;
            o.y += zz.y;
        }
        w[idx] = o;
        const double2 c = acc[idx];
        acc[idx] = make_double2(c.x + beta * o.x, c.y + beta * o.y);
        if (wp0) {
            const size_t i = idx / ld;
            wp0[idx] = cmul(e0[i], o);
            wp1[idx] = cmul(e1[i], o);
        }
    }
}

// Column-stacking superoperators of the vectorised Lindblad model, written straight into the operator stack
// from the n x n operators (models/model_utils.py:31-118; N = n^2, row r = i n + k, column c = j n + m):
//   kind 0:  -i (I (x) A - A^T (x) I)                                    (vec_commutator)
//   kind 1:  conj(L) (x) L - (I (x) L^+L + (L^+L)^T (x) I) / 2           (vec_dissipator; ldl = L^+L)
// accumulate != 0 adds to what `out` holds.  Plain multiplies and adds in numpy's order (no contraction).
MIDYN_GLOBAL __launch_bounds__(256) void vec_lindblad_kernel(int n, int ld, const double2* a, int kind, const double2* ldl,
                                                           double2* out, int accumulate) {
#pragma clang fp contract(off)
    const size_t N = (size_t)n * n, total = N * N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / N), c = (int)(idx - (size_t)r * N);
        const int i = r / n, k = r - i * n, j = c / n, m = c - j * n;
        double2 val;
        if (kind == 0) {
            const double2 x = i == j ? a[k * n + m] : make_double2(0.0, 0.0);
            const double2 y = k == m ? a[j * n + i] : make_double2(0.0, 0.0);
            const double2 d = make_double2(x.x - y.x, x.y - y.y);
            val = make_double2(d.y, -d.x);  // -i d
        } else {
            const double2 lc = a[i * n + j], l2 = a[k * n + m];  // conj(lc) * l2
            const double2 outer = make_double2(lc.x * l2.x + lc.y * l2.y, lc.x * l2.y - lc.y * l2.x);
            const double2 x = i == j ? ldl[k * n + m] : make_double2(0.0, 0.0);
            const double2 y = k == m ? ldl[j * n + i] : make_double2(0.0, 0.0);
            val = make_double2(outer.x - 0.5 * (x.x + y.x), outer.y - 0.5 * (x.y + y.y));
        }
        double2* o = out + (size_t)r * ld + c;
        if (accumulate) val = make_double2(o->x + val.x, o->y + val.y);
        *o = val;
    }
}

// a += b over [rows][cols] blocks of leading dimension ld (static superoperator = commutator part + dissipator sum)
MIDYN_GLOBAL __launch_bounds__(256) void add_padded_kernel(double2* a, const double2* b, int rows, int cols, int ld) {
#pragma clang fp contract(off)
    const size_t total = (size_t)rows * cols;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const size_t r = idx / cols, off = r * ld + (idx - r * cols);
        a[off] = make_double2(a[off].x + b[off].x, a[off].y + b[off].y);
    }
}

// flags[2*seg + 0/1] = 1 if any real / imaginary part of segment seg is non zero
MIDYN_GLOBAL __launch_bounds__(256) void plane_flags_kernel(const double2* ops, size_t plane, int nseg,
                                                          int* flags) {
    const size_t total = plane * nseg;
    int fr = 0, fi = 0;
    int seg_of = -1;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int seg = (int)(idx / plane);
        if (seg != seg_of) {
            if (seg_of >= 0) {
                if (fr) flags[2 * seg_of] = 1;
                if (fi) flags[2 * seg_of + 1] = 1;
            }
            seg_of = seg;
            fr = fi = 0;
        }
        const double2 v = ops[idx];
        fr |= (v.x != 0.0);
        fi |= (v.y != 0.0);
    }
    if (seg_of >= 0) {
        if (fr) flags[2 * seg_of] = 1;
        if (fi) flags[2 * seg_of + 1] = 1;
    }
}

// map[(seg * nrb + rb) * ncb + cb] = 1 when the 16 x 16 block (rb, cb) of operator segment seg holds a
// non-zero entry (nrb = ncb = n_pad / 16; the map is zeroed by the caller; every writer stores 1).
// grid (nrb, nseg): one workgroup per 16-row strip.
MIDYN_GLOBAL __launch_bounds__(256) void block_map_kernel(const double2* ops, int n_pad, unsigned char* map) {
    const int rb = blockIdx.x, seg = blockIdx.y;
    const int nb = n_pad / 16;
    const double2* base = ops + (size_t)seg * n_pad * n_pad + (size_t)rb * 16 * n_pad;
    unsigned char* out = map + ((size_t)seg * nb + rb) * nb;
    for (int idx = threadIdx.x; idx < 16 * n_pad; idx += 256) {
        const double2 v = base[idx];
        if (v.x != 0.0 || v.y != 0.0) out[(idx % n_pad) / 16] = 1;
    }
}

// partial column abs sums of [n][n] matrices (ld n): sums[(mat * nchunk + z) * n + c] = sum over the
// rows of chunk z of |A[r][c]|; the host adds the chunks (fixed order) and takes max_c = the 1-norm.
// grid (ceil(n/256), batch, nchunk): enough workgroups to stream a large matrix at HBM rate.
MIDYN_GLOBAL __launch_bounds__(256) void colsum_kernel(const double2* A, int n, int nchunk, double* sums) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const double2* Ab = A + (size_t)blockIdx.y * n * n;   // blockIdx.y: matrix of a batch
    const int rows = (n + nchunk - 1) / nchunk;
    const int r0 = blockIdx.z * rows, r1 = min(n, r0 + rows);
    double s = 0.0;
    for (int r = r0; r < r1; ++r) {
        const double2 v = Ab[(size_t)r * n + c];
        s += hypot(v.x, v.y);
    }
    sums[((size_t)blockIdx.y * nchunk + blockIdx.z) * n + c] = s;
}

// The same partial column sums for |A^T| (mode 1: the infinity norm of A as the 1-norm of A^T) and for the
// Hermitian part |A + A^dagger| / 2 (mode 2); one-off per operator stack, so the strided transposed reads are fine.
MIDYN_GLOBAL __launch_bounds__(256) void colsum_mode_kernel(const double2* A, int n, int nchunk, int mode, double* sums) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= n) return;
    const double2* Ab = A + (size_t)blockIdx.y * n * n;
    const int rows = (n + nchunk - 1) / nchunk;
    const int r0 = blockIdx.z * rows, r1 = min(n, r0 + rows);
    double s = 0.0;
    for (int r = r0; r < r1; ++r) {
        const double2 t = Ab[(size_t)c * n + r];
        if (mode == 1) {
            s += hypot(t.x, t.y);
        } else {
            const double2 v = Ab[(size_t)r * n + c];
            s += 0.5 * hypot(v.x + t.x, v.y - t.y);
        }
    }
    sums[((size_t)blockIdx.y * nchunk + blockIdx.z) * n + c] = s;
}

// partial[(seg * nchunk + z)] = sum over the rows of chunk z of |A_seg[r][c] + conj(A_seg[c][r])|^2: the squared
// Frobenius norm of A + A^dagger in fixed-order pieces (the host adds them).  For the generators -iH of a
// Hamiltonian model this is || H - H^dagger ||_F^2, i.e. the reference's Hermiticity validation
// (models/hamiltonian_model.py:98-104) evaluated where the operators already are.
MIDYN_GLOBAL __launch_bounds__(256) void antiherm_defect_kernel(const double2* A, int n, int nchunk, double* partial) {
    const double2* Ab = A + (size_t)blockIdx.y * n * n;
    const int rows = (n + nchunk - 1) / nchunk;
    const int r0 = blockIdx.x * rows, r1 = min(n, r0 + rows);
    double acc = 0.0;
    for (size_t idx = (size_t)r0 * n + threadIdx.x; idx < (size_t)r1 * n; idx += 256) {
        const int r = (int)(idx / n), c = (int)(idx - (size_t)r * n);
        const double2 v = Ab[idx], t = Ab[(size_t)c * n + r];
        const double dx = v.x + t.x, dy = v.y - t.y;
        acc += dx * dx + dy * dy;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)blockIdx.y * nchunk + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}

// pad copy: src [rows][cols] (ld src_ld) -> dst (ld dst_ld), both complex
MIDYN_GLOBAL __launch_bounds__(256) void copy2d_kernel(const double2* src, int src_ld, double2* dst, int dst_ld,
                                                     int rows, int cols) {
    const size_t total = (size_t)rows * cols;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / cols);
        const int c = (int)(idx - (size_t)r * cols);
        dst[(size_t)r * dst_ld + c] = src[(size_t)r * src_ld + c];
    }
}

// X[a][b] *= e_a * conj(e_b)  (dir = +1: operator OUT of the frame, rotating_frame.py:397-436 with -t)
//           or conj(e_a) * e_b (dir = -1: operator INTO the frame, :372-395), e = exp(d t); in place or to dst
MIDYN_GLOBAL __launch_bounds__(256) void frame_mask_kernel(const double2* src, const double2* e, int n_pad, int dir,
                                                         double2* dst) {
    const size_t total = (size_t)n_pad * n_pad;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total;
         idx += (size_t)gridDim.x * 256) {
        const int a = (int)(idx / n_pad);
        const int b = (int)(idx - (size_t)a * n_pad);
        const double2 ea = e[a], eb = e[b];
        const double2 ph = dir > 0 ? cmul_conj_a(eb, ea) : cmul_conj_a(ea, eb);
        dst[idx] = cmul(ph, src[idx]);
    }
}

// Batched forms for a chunk of instances (matrices back to back, one phase row shared by the chunk):
// frame mask of every matrix, and T_b *= gamma_b (gamma_b = coeff[b * stride + j]: a dynamic dissipator's rate)
MIDYN_GLOBAL __launch_bounds__(256) void frame_mask_batch_kernel(const double2* src, const double2* e, int n_pad, int dir,
                                                               int batch, double2* dst) {
    const size_t plane = (size_t)n_pad * n_pad, total = plane * batch;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t idx = g % plane;
        const int a = (int)(idx / n_pad);
        const int b = (int)(idx - (size_t)a * n_pad);
        const double2 ea = e[a], eb = e[b];
        const double2 ph = dir > 0 ? cmul_conj_a(eb, ea) : cmul_conj_a(ea, eb);
        dst[g] = cmul(ph, src[g]);
    }
}

MIDYN_GLOBAL __launch_bounds__(256) void scale_batch_kernel(double2* x, size_t plane, int batch, const double* coeff,
                                                          long long stride) {
    const size_t total = plane * batch;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const double s = coeff[(size_t)(g / plane) * stride];
        x[g] = make_double2(s * x[g].x, s * x[g].y);
    }
}

// out[b][slot][i][j] = Y[b][i][j] (padded [np][np] -> [n][n]) for a chunk of instances
MIDYN_GLOBAL __launch_bounds__(256) void save_density_kernel(const double2* Y, int np, int n, int batch, int P, int slot,
                                                           double2* out) {
    const size_t nn = (size_t)n * n, total = nn * batch;
    for (size_t g = (size_t)blockIdx.x * 256 + threadIdx.x; g < total; g += (size_t)gridDim.x * 256) {
        const size_t b = g / nn, rem = g - b * nn;
        const int i = (int)(rem / n), j = (int)(rem - (size_t)i * n);
        out[(b * P + slot) * nn + rem] = Y[b * (size_t)np * np + (size_t)i * np + j];
    }
}

// ---- micro-benchmarks: the ceilings the roofline fractions are quoted against -------------------
// 8 independent fp64 MFMA accumulators per wave (all in VGPRs), `iters` rounds, 4 waves per SIMD:
// pure matrix-pipe throughput.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(double* sink, int iters) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a[NACC], b = 0.5 - threadIdx.x * 1e-9;
#pragma unroll
    for (int i = 0; i < NACC; ++i) a[i] = 1.0 + (threadIdx.x + 64 * i) * 1e-9;  // distinct: no CSE of chains
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b));
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(b), "v"(a[i]));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;
}

// The same pipe under the POWER of real data: the contraction kernels' cadence (512 threads = 2 waves per SIMD, 16
// accumulator quads per wave, one workgroup per CU), operands with random mantissas in [0.5, 1) and both signs.  The
// chip clocks to its power budget: a dense fp64 MFMA stream on such operands sustains a lower shader clock than the
// 2.4 GHz the 78.6 TFLOP/s peak is quoted at (operands that are all zero, or nearly constant as in mfma_peak_kernel,
// toggle few bits and keep the full clock).  sink[1], sink[2] = shader cycles and 100 MHz ticks of one workgroup.
MIDYN_GLOBAL __launch_bounds__(512, 2) void mfma_sustained_kernel(double* sink, int iters, int random_operands) {
    const long long c0 = clock64(), w0 = wall_clock64();
    d4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a[4], b[4];
    unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned lo0 = (h = h * 1664525u + 1013904223u), hi0 = (h = h * 1664525u + 1013904223u);
        const unsigned lo1 = (h = h * 1664525u + 1013904223u), hi1 = (h = h * 1664525u + 1013904223u);
        a[i] = random_operands ? __hiloint2double((int)(0x3FE00000u | (hi0 >> 12) | ((hi0 & 1u) << 31)), (int)lo0) : 0.0;
        b[i] = random_operands ? __hiloint2double((int)(0x3FE00000u | (hi1 >> 12) | ((hi1 & 1u) << 31)), (int)lo1) : 0.0;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[i >> 2]));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) sink[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        sink[1] = (double)(clock64() - c0);
        sink[2] = (double)(wall_clock64() - w0);
    }
}

// Do the fp64 matrix pipe and the fp64 vector ALUs run CONCURRENTLY?  Per round and wave: NMFMA independent
// v_mfma_f64_16x16x4 (2048 flops each) interleaved with NFMA independent v_fma_f64 (128 flops each).  MODE 0: both,
// 1: MFMAs only, 2: FMAs only.  If the pipes were independent, the combined round would take max(...) of the two.
template <int NMFMA, int NFMA, int MODE>
__global__ __launch_bounds__(256) void fp64_coissue_kernel(double* sink, int iters) {
    d4 acc[NMFMA];
    double f[NFMA];
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NFMA; ++i) f[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const double a = 1.0 + threadIdx.x * 1e-9, b = 0.5 - threadIdx.x * 1e-9, c = 1.0 - 1e-12 * threadIdx.x;
    constexpr int PER = NFMA / NMFMA;   // vector FMAs issued behind every MFMA
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
            if (MODE != 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (MODE != 1) {
#pragma unroll
                for (int j = 0; j < PER; ++j)
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[i * PER + j]) : "v"(c), "v"(b));
            }
        }
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < NFMA; ++i) s += f[i];
    if (s == 123.456) sink[0] = s;
}

// Norm bounds of the Magnus generator Omega of every step over the instances of a sweep (expm action, midyn_action.inc):
// out[3 st + {0, 1, 2}] = max_b of the 1-norm triangle bound, of the larger of the 1- and infinity-norm bounds, and of the
// bound of the Hermitian part, from the per-segment norms nrm[3][nseg] and the coefficient table S[B][R][k] ON THE DEVICE
// (a 4096-instance table came back to the host for this loop: 39 MB over PCIe + a strided pass, 7 ms of a 3 ms solve).
MIDYN_GLOBAL __launch_bounds__(256) void step_bounds_kernel(const double* __restrict__ S, int B, int R, int k, int nseg, int has_static,
                                                          const int* __restrict__ rows, const double* __restrict__ hs, int order,
                                                          const double* __restrict__ nrm, double* __restrict__ out) {
    const int st = blockIdx.x;
    const double ah = fabs(hs[st]);
    const double p2 = 0.14433756729740643;   // sqrt(3) / 12
    double best[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < B; b += 256) {
        double gn[2] = {0.0, 0.0}, G[2] = {0.0, 0.0}, Hm[2] = {0.0, 0.0};
        for (int i = 0; i < order; ++i) {
            const double* c = S + ((size_t)b * R + rows[3 * st + i]) * k;
            double ginf = 0.0;
            for (int seg = 0; seg < nseg; ++seg) {
                const double cf = (has_static && seg == 0) ? 1.0 : fabs(c[seg - has_static]);
                gn[i] += cf * nrm[seg];
                ginf += cf * nrm[nseg + seg];
                Hm[i] += cf * nrm[2 * nseg + seg];
            }
            G[i] = fmax(gn[i], ginf);
        }
        double v[3];
        if (order == 1) {
            v[0] = ah * gn[0];
            v[1] = ah * G[0];
            v[2] = ah * Hm[0];
        } else {
            v[0] = 0.5 * ah * (gn[0] + gn[1]) + 2 * p2 * ah * ah * gn[0] * gn[1];
            v[1] = 0.5 * ah * (G[0] + G[1]) + 2 * p2 * ah * ah * G[0] * G[1];
            v[2] = 0.5 * ah * (Hm[0] + Hm[1]) + 2 * p2 * ah * ah * (Hm[1] * G[0] + G[1] * Hm[0]);
        }
        for (int j = 0; j < 3; ++j) best[j] = (v[j] > best[j] || v[j] != v[j]) ? v[j] : best[j];      // (NaN propagates)
    }
    __shared__ double red[3][256];
    for (int j = 0; j < 3; ++j) red[j][threadIdx.x] = best[j];
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int j = 0; j < 3; ++j) {
                const double o = red[j][threadIdx.x + w], m_ = red[j][threadIdx.x];
                red[j][threadIdx.x] = (o > m_ || o != o) ? o : m_;
            }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        for (int j = 0; j < 3; ++j) out[3 * st + j] = red[j][0];
}

// streaming read of `n16` 16-byte elements (grid-stride, 4 independent loads per thread per round)
MIDYN_GLOBAL __launch_bounds__(256) void stream_read_kernel(const double2* src, size_t n16, double* sink) {
    double2 acc = make_double2(0.0, 0.0);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const double2 v0 = src[i], v1 = src[i + stride], v2 = src[i + 2 * stride], v3 = src[i + 3 * stride];
        acc.x += v0.x + v1.x + v2.x + v3.x;
        acc.y += v0.y + v1.y + v2.y + v3.y;
    }
    for (; i < n16; i += stride) {
        acc.x += src[i].x;
        acc.y += src[i].y;
    }
    if (acc.x == 123.456 && acc.y == 654.321) sink[0] = acc.x;
}

// ---- the instantiations of the MFMA contraction kernels that exist, and where -----------------------------------------------
// libmidyn.so is built from several translation units so that hipcc compiles the kernel families side by side (the device
// side of ONE unit is compiled serially: five minutes for everything in round 4).  The host side of every entry point stays
// in midyn.hip; a family's kernels are instantiated in its own unit (midyn_tu_gemm_dense.hip defines MIDYN_TU_GEMM_DENSE,
// midyn_tu_gemm_lists.hip MIDYN_TU_GEMM_LISTS, midyn_tu_gemm_pairs.hip MIDYN_TU_GEMM_PAIRS), every other unit sees `extern template` declarations of the same list and
// only takes the kernels' addresses.  A kernel the host code selects but the list does not name is an undefined symbol of
// the library: __graft_entry__.build() refuses it.
#define MIDYN_FOR_PLANE_MODE(X, ...) X(__VA_ARGS__, 0) X(__VA_ARGS__, 1) X(__VA_ARGS__, 2) X(__VA_ARGS__, 3)
#define MIDYN_GEMM_PAIR_TILES(X)                                                                                    \
    MIDYN_FOR_PLANE_MODE(X, 128, 128, 2, 4, 16) MIDYN_FOR_PLANE_MODE(X, 64, 64, 2, 2, 16) MIDYN_FOR_PLANE_MODE(X, 32, 128, 1, 4, 16) \
    MIDYN_FOR_PLANE_MODE(X, 32, 64, 1, 2, 16) MIDYN_FOR_PLANE_MODE(X, 16, 128, 1, 4, 16) MIDYN_FOR_PLANE_MODE(X, 16, 64, 1, 2, 16)
#define MIDYN_GEMM_TILES(X) MIDYN_GEMM_PAIR_TILES(X)     // (the 128 x 64 x 8 two-per-CU tile of round 2's A/B is no longer built: nothing selects it)
#define MIDYN_GEMM_DENSE_TILES(X) MIDYN_FOR_PLANE_MODE(X, 128, 128, 2, 4, 16) MIDYN_FOR_PLANE_MODE(X, 64, 64, 2, 2, 16)   // (dense: no panel tiles)
#ifdef MIDYN_TU_GEMM_DENSE
#define MIDYN_GEMM_DENSE_EXTERN
#else
#define MIDYN_GEMM_DENSE_EXTERN extern
#endif
#ifdef MIDYN_TU_GEMM_LISTS
#define MIDYN_GEMM_LISTS_EXTERN
#else
#define MIDYN_GEMM_LISTS_EXTERN extern
#endif
#ifdef MIDYN_TU_GEMM_PAIRS
#define MIDYN_GEMM_PAIRS_EXTERN
#else
#define MIDYN_GEMM_PAIRS_EXTERN extern
#endif
#define MIDYN_X(BM_, BN_, WM_, WN_, BK_, MODE_) \
    MIDYN_GEMM_DENSE_EXTERN template __global__ void zgemm_seg_kernel<BM_, BN_, WM_, WN_, BK_, MODE_, 2, false>(GemmArgs);
MIDYN_GEMM_DENSE_TILES(MIDYN_X)
MIDYN_X(64, 64, 2, 2, 16, 4)      // 3M: three accumulator sets fit the registers on the 64 x 64 tile only
#undef MIDYN_X
#define MIDYN_X(BM_, BN_, WM_, WN_, BK_, MODE_) \
    MIDYN_GEMM_LISTS_EXTERN template __global__ void zgemm_seg_kernel<BM_, BN_, WM_, WN_, BK_, MODE_, 2, true>(GemmArgs);
MIDYN_GEMM_TILES(MIDYN_X)
#undef MIDYN_X
#define MIDYN_X(BM_, BN_, WM_, WN_, BK_, MODE_) \
    MIDYN_GEMM_PAIRS_EXTERN template __global__ void zgemm_seg_pair_kernel<BM_, BN_, WM_, WN_, BK_, MODE_, 2, true>(GemmPair);
MIDYN_GEMM_PAIR_TILES(MIDYN_X)
#undef MIDYN_X
#define MIDYN_X(MODE_) \
    MIDYN_GEMM_DENSE_EXTERN template __global__ void splitk_reduce_kernel<MODE_, 0>(const double2*, int, int, int, Epilogue);  \
    MIDYN_GEMM_DENSE_EXTERN template __global__ void splitk_reduce_kernel<MODE_, 2>(const double2*, int, int, int, Epilogue);  \
    MIDYN_GEMM_DENSE_EXTERN template __global__ void splitk_reduce_kernel<MODE_, 4>(const double2*, int, int, int, Epilogue);  \
    MIDYN_GEMM_DENSE_EXTERN template __global__ void splitk_reduce_kernel<MODE_, 8>(const double2*, int, int, int, Epilogue);  \
    MIDYN_GEMM_DENSE_EXTERN template __global__ void splitk_reduce_kernel<MODE_, 16>(const double2*, int, int, int, Epilogue);
MIDYN_X(EPI_PLAIN) MIDYN_X(EPI_RHS) MIDYN_X(EPI_RK1) MIDYN_X(EPI_RK2) MIDYN_X(EPI_RK3) MIDYN_X(EPI_RK4) MIDYN_X(EPI_TAYLOR) MIDYN_X(EPI_CHEB)
#undef MIDYN_X

}  // namespace midyn
