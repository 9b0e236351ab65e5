// midyn_combine.h -- the sweep contraction as COMBINE + APPLY (round 4): for many instances with their own coefficients
//
//     out[r][b] = sum_kk ( G_d[r][kk] + sum_j c_j[b] G_j[r][kk] ) y[kk][b]
//
// is evaluated the way the reference evaluates it per instance (models/operator_collections.py:101-134: first the
// signal-weighted operator sum, then the product), instead of as k + 1 GEMMs whose results are scaled and added:
//
//   COMBINE  g[r][b] = sum_j a_j[r][kk] c_j[b]     v_mfma_f64_16x16x4: M = 16 rows, N = 16 instances, K = 4 operator PLANES
//                                                  per instruction (real and imaginary planes are separate "operators";
//                                                  planes that are exactly zero do not exist here); the static operator
//                                                  enters as the C input of the first MFMA (coefficient 1, no plane slot)
//   APPLY    out[r][b] += g[r][b] * y[kk][b]       2 (one plane kind) or 4 (both) v_fma_f64 in the lane that holds D[r][b]
//
// Per (row, kk, instance): P MFMA-FMAs for P time-dependent planes + 2..4 vector FMAs, against 2 P (single-plane
// operators) / 3..4 P (complex operators, 3M / 4M) MFMA-FMAs of the GEMM formulation -- both run on the same fp64 pipe
// of the SIMD.  BASELINE configs[2] (8 purely imaginary operators): 10 instead of 16 FMAs per element.
//
// No LDS and no barrier: the operator planes are re-packed once per stack in MFMA A-operand order (rows of a 16-row tile x
// 4 planes per 512-byte wave load, contiguous per wave through its list of kk blocks), the stage input row y[kk][.] and the
// static operator elements are read straight from memory one kk step ahead of their use (the state: 16 lanes per address;
// the static rows: one element per lane, broadcast within the 16-lane row by DPP).
// Exactly-zero 16-column blocks of a 32-row group are skipped through per-group lists (symmetry sectors).
#pragma once

namespace midyn {

constexpr int CMB_MAXQ = 3;     // plane groups (of 4) per kind when the stack has planes of both kinds: k <= 12 operators per kind (round 5: was 2)
constexpr int CMB_MAXQ1 = 4;    // ... of ONE kind only (real-symmetric Hamiltonians: purely imaginary generators): k <= 16
constexpr int CMB_SLOTS = 8 * CMB_MAXQ;   // plane slots of the widest layouts ((3, 3): 24; one kind alone: 16)
constexpr bool combine_groups_ok(int nre4, int nim4) {
    return nre4 + nim4 > 0 && ((nre4 <= CMB_MAXQ && nim4 <= CMB_MAXQ) || (nre4 == 0 && nim4 <= CMB_MAXQ1) || (nim4 == 0 && nre4 <= CMB_MAXQ1));
}
// the one-launch kernels of small systems (midyn_combine_sweep.h) exist for up to two groups per kind
constexpr bool combine_sweep_groups_ok(int nre4, int nim4) { return nre4 + nim4 > 0 && nre4 <= 2 && nim4 <= 2; }
constexpr int CMB_ROWS = 32;    // rows per row group (two 16-row MFMA tiles per wave)

struct CombineArgs {
    const double* frags;     // [list entry][kk % 16][tile < 2][group q < NRE4 + NIM4][lane] doubles (entries of a row group contiguous)
    const int* list_ptr;     // [n_row_groups + 1]
    const int* list_idx;     // kk block (16 columns of the operators) of every entry
    const double2* stat;     // static operator [lda][lda] (complex) or nullptr
    int lda;
    const double2* B;        // stage input [K][ldb], pre-phased
    int ldb;
    const double* coeff;     // [instance][inst_stride]: coefficient row of this evaluation
    long long inst_stride;
    int m_cols, n_inst;
    int plane_col[CMB_SLOTS];   // coefficient column of plane 4 q + i (first the real-plane groups, then the imaginary ones); -1: padding
    int n_row_groups;
    int wr;                  // waves of a workgroup along the rows (the others along the instances)
    int splits;              // > 1 (a power of two): that many waves share one (row group, instance block) and split its list (small
                             // sweeps: more waves than (row group, instance block) pairs are needed to fill the chip); summed through LDS
    Epilogue epi;
};

// Re-pack: one workgroup per list entry.  plane_seg[p] = segment of plane p (or -1), plane_im[p] = 1 for an imaginary plane.
struct CombinePackArgs {
    const double2* ops;
    long long seg_stride;
    int lda;
    const int* ent_rg;       // row group of every entry
    const int* list_idx;
    int nq;                  // NRE4 + NIM4
    int plane_seg[CMB_SLOTS];
    int plane_im[CMB_SLOTS];
    double* frags;
};

MIDYN_GLOBAL __launch_bounds__(256) void combine_pack_kernel(CombinePackArgs a) {
    const int e = blockIdx.x;
    const int rg = a.ent_rg[e], kb = a.list_idx[e];
    const int per_entry = 16 * 2 * a.nq * 64;
    double* out = a.frags + (size_t)e * per_entry;
    for (int i = threadIdx.x; i < per_entry; i += 256) {
        const int lane = i & 63;
        const int q = (i >> 6) % a.nq;
        const int t = ((i >> 6) / a.nq) & 1;
        const int k16 = (i >> 6) / (a.nq * 2);
        const int p = 4 * q + (lane >> 4);
        const int seg = a.plane_seg[p];
        double v = 0.0;
        if (seg >= 0) {
            const double2 z = a.ops[(size_t)seg * a.seg_stride + (size_t)(rg * CMB_ROWS + t * 16 + (lane & 15)) * a.lda + kb * 16 + k16];
            v = a.plane_im[p] ? z.y : z.x;
        }
        out[i] = v;
    }
}

// Value of lane N (0..15) of this lane's row of 16 lanes, in every lane of the row (DPP row_newbcast, gfx90a+).
template <int N>
__device__ __forceinline__ double row_bcast(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + N, 0xf, 0xf, true);    // (one v_mov_b64_dpp: row_newbcast is the one
                                                                              //  DPP control the 64-bit ALU ops accept)
}

// 16-instance groups per wave: 4 (64 instances) up to two plane groups, else 2 (registers: 2 x NG accumulator quad pairs, the
// combined elements of one plane kind at a time, two sets of prefetched operands; 256 per lane at two waves per SIMD).
constexpr int combine_ng(int nre4, int nim4, int stat) {
    return nre4 + nim4 <= 2 ? 4 : 2;
}

// STAT: bit 0 = the static operator has a real plane, bit 1 = an imaginary plane.  NG: 16-instance groups per wave.
template <int NRE4, int NIM4, int STAT, int NG>
__device__ __forceinline__ void combine_body(const CombineArgs& a) {
    constexpr int NQ = NRE4 + NIM4, RT = 2;
    constexpr bool RE = NRE4 > 0 || (STAT & 1), IM = NIM4 > 0 || (STAT & 2);
    static_assert(combine_groups_ok(NRE4, NIM4), "plane groups");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int sp = wave % a.splits, wv = wave / a.splits;
    const int waves = (blockDim.x >> 6) / a.splits;
    const int wr = wv % a.wr, wi = wv / a.wr, nwi = waves / a.wr;
    const int rgb = (a.n_row_groups + a.wr - 1) / a.wr;
    const int rg = (blockIdx.x % rgb) * a.wr + wr;
    // (rg < n_row_groups for every wave: launch_combine makes a.wr a divisor of the row-group count, so no wave of a
    //  workgroup leaves before the barriers of the list split below)
    const int col0 = ((blockIdx.x / rgb) * nwi + wi) * (NG * 16);
    const int lb = lane & 15, lq = lane >> 4;

    // B operands of the MFMAs: the coefficients c[plane 4 q + lq][column 16 g + lb] (constant over the launch)
    double cb[NG][NQ];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        int in = (col0 + 16 * g + lb) / a.m_cols;
        in = in < a.n_inst ? in : a.n_inst - 1;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int pc = a.plane_col[4 * q + lq];
            cb[g][q] = pc >= 0 ? a.coeff[(long long)in * a.inst_stride + pc] : 0.0;
        }
    }
    d4 ore[RT][NG], oim[RT][NG];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            ore[t][g] = d4{0.0, 0.0, 0.0, 0.0};
            oim[t][g] = d4{0.0, 0.0, 0.0, 0.0};
        }
    int e0 = a.list_ptr[rg], e1 = a.list_ptr[rg + 1];
    if (a.splits > 1) {      // this wave's share of the list
        const int len = e1 - e0;
        e1 = e0 + (int)((long long)len * (sp + 1) / a.splits);
        e0 = e0 + (int)((long long)len * sp / a.splits);
    }
    // (wave-uniform bases + one 32-bit lane offset each: the loads take the scalar-base form, no 64-bit address registers)
    const double* __restrict__ fr = a.frags + (size_t)e0 * (16 * RT * NQ * 64);
    const double2* __restrict__ yb = a.B + col0;
    // static rows lq + 4 r (+ 16 t) of this lane, as doubles (only the planes the static operator has are read)
    // The static operator (the C input of the combining MFMAs): the 16 lanes of a row (same lq) need the same 4 rows x 2
    // tiles x 2 planes = 16 doubles per kk.  Each lane fetches ONE of them -- lane lb: row lq + 4 (lb & 3) of tile
    // (lb >> 2) & 1, plane lb >> 3 -- and the row broadcasts them with DPP (16 lanes reading one address still cost the
    // texture path a full wave-load each: 8-16 of them per step were 27 % of the 8-plane kernel's time).
    const double* __restrict__ sb = reinterpret_cast<const double*>(a.stat + (size_t)(rg * CMB_ROWS) * a.lda);
    const unsigned s_lane = (unsigned)((lq + 4 * (lb & 3) + 16 * ((lb >> 2) & 1)) * a.lda) * 2u + (unsigned)(lb >> 3);

    double afr[2][RT][NQ];
    double2 yv[2][NG];
    double sv[2];
    // loads of flat step `sf` (entry sf / 16 of this row group, kk = 16 kb + sf % 16) into buffer b
    auto load = [&](int sf, int kb, int b) {
        const int kk = kb * 16 + (sf & 15);
        const double* __restrict__ fn = fr + (size_t)sf * (RT * NQ * 64);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < NQ; ++q) afr[b][t][q] = fn[(unsigned)((t * NQ + q) * 64 + lane)];
        const double2* __restrict__ yrow = yb + (size_t)kk * a.ldb;
#pragma unroll
        for (int g = 0; g < NG; ++g) yv[b][g] = yrow[(unsigned)(16 * g + lb)];
        if (STAT) sv[b] = (sb + 2 * (size_t)kk)[s_lane];
        // (the loads stay HERE, one step ahead of their use: left alone, the scheduler sinks them to their first use to
        // shorten live ranges, and every step then waits for its own loads)
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](int b) {
        // one plane kind at a time (the combined elements of the other kind are not live meanwhile): first the MFMAs of
        // all instance groups, then their vector FMAs -- the first results are ready when the last MFMA has issued
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if (RE) {
                d4 gre[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    gre[g] = (STAT & 1) ? (t == 0 ? d4{row_bcast<0>(sv[b]), row_bcast<1>(sv[b]), row_bcast<2>(sv[b]), row_bcast<3>(sv[b])}
                                                  : d4{row_bcast<4>(sv[b]), row_bcast<5>(sv[b]), row_bcast<6>(sv[b]), row_bcast<7>(sv[b])})
                                        : d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < NRE4; ++q)
                        gre[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[b][t][q], cb[g][q], gre[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {           // (g_re)(y_re + i y_im)
                        ore[t][g][r] = fma(gre[g][r], yv[b][g].x, ore[t][g][r]);
                        oim[t][g][r] = fma(gre[g][r], yv[b][g].y, oim[t][g][r]);
                    }
            }
            if (IM) {
                d4 gim[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    gim[g] = (STAT & 2) ? (t == 0 ? d4{row_bcast<8>(sv[b]), row_bcast<9>(sv[b]), row_bcast<10>(sv[b]), row_bcast<11>(sv[b])}
                                                  : d4{row_bcast<12>(sv[b]), row_bcast<13>(sv[b]), row_bcast<14>(sv[b]), row_bcast<15>(sv[b])})
                                        : d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < NIM4; ++q)
                        gim[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(afr[b][t][NRE4 + q], cb[g][NRE4 + q], gim[g], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {           // (i g_im)(y_re + i y_im) = -g_im y_im + i g_im y_re
                        ore[t][g][r] = fma(-gim[g][r], yv[b][g].y, ore[t][g][r]);
                        oim[t][g][r] = fma(gim[g][r], yv[b][g].x, oim[t][g][r]);
                    }
            }
        }
    };
    const int steps = (e1 - e0) * 16;
    if (steps > 0) {
        // kk blocks of the entry that holds step s and of the next one (scalar loads, an entry ahead of their first use)
        int kb_cur = a.list_idx[e0], kb_nxt = a.list_idx[e1 - e0 > 1 ? e0 + 1 : e0];
        load(0, kb_cur, 0);
        for (int s = 0; s < steps; s += 2) {      // (steps is a multiple of 16; the look-ahead past the end reads the padding entry)
            // the step after next may open a new entry (selects, no branch: the loop stays one block; the scalar load is
            // issued a whole step ahead of the select that consumes it)
            const int wrap = ((s + 2) & 15) == 0;
            const int en = e0 + ((s + 2) >> 4) + 1;
            const int kb_far = a.list_idx[en < e1 ? en : e1 - 1];
            load(s + 1, kb_cur, 1);
            compute(0);
            __builtin_amdgcn_sched_barrier(0);
            kb_cur = wrap ? kb_nxt : kb_cur;
            kb_nxt = wrap ? kb_far : kb_nxt;
            load(s + 2, kb_cur, 0);
            compute(1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // (the lane number again, from the execution mask count: the main loop uses all 256 registers, and a value of the prologue
    //  that is only needed down here would be parked in scratch around it -- tests/test_codeobj.py refuses that)
    int lane_t;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_t));
    const int lb_t = lane_t & 15, lq_t = lane_t >> 4;
    if (a.splits > 1) {
        // Partial sums through LDS by recursive HALVING (round 6; rounds 2-5: a tree that ended in ONE wave, which then ran the
        // fused epilogue of the whole 32 x (16 NG) tile alone while the other waves of the workgroup had left -- on the shards of
        // a strong-scaling run, where every workgroup splits one list, that tail was the epilogue's whole cost with nothing to
        // hide it behind).  The tile of a pair is RT x NG sub-tiles of 16 rows x 16 instances.  In every phase a wave and its
        // partner (sp ^ h, h = n / 2, n / 4, ..) keep one half of what they still hold each, hand the other half over and
        // add what they receive: first the halves in t (rows), then in g; after log2(n) phases every wave holds the complete sums
        // of 1 / n of the tile and runs the epilogue on that.  The pairs (w, w ^ 4), then (.., w ^ 2), then (.., w ^ 1) are
        // the tree's, and a + b = b + a bit for bit: the sums equal the tree's.  A wave writes what it hands over where it has
        // just READ (only it read there): ONE barrier per phase, (splits / 2) x NV x 64 doubles per pair as before.
        constexpr int NV = RT * NG * 8;     // doubles per lane
        double* const lds = reinterpret_cast<double*>(smem_raw) + (size_t)(wv * (a.splits >> 1)) * (NV * 64) + lane_t;
        int n = a.splits;                   // waves of the pair that still hold sums
        unsigned stride = (unsigned)(NV / 2) * 64;       // slot stride of the first halving phase
        if (RT * NG < n) {
            // (NG = 2, eight waves: four sub-tiles only -- waves 4..7 hand everything to 0..3 first and wait at the barriers below)
            if (sp >= 4) {
                double* mine = lds + (unsigned)(sp - 4) * (NV * 64);
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            mine[((t * NG + g) * 8 + r) * 64] = ore[t][g][r];
                            mine[((t * NG + g) * 8 + 4 + r) * 64] = oim[t][g][r];
                        }
            }
            __syncthreads();
            if (sp < 4) {
                const double* theirs = lds + (unsigned)sp * (NV * 64);
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ore[t][g][r] += theirs[((t * NG + g) * 8 + r) * 64];
                            oim[t][g][r] += theirs[((t * NG + g) * 8 + 4 + r) * 64];
                        }
            }
            n = 4;
            stride = (unsigned)NV * 64;
        }
        const bool holds = sp < n;          // wave-uniform
        // where this wave reads in the coming phase (= where its partner writes): phase 1: the partner's slot; later: see above
        int h = n >> 1;
        unsigned wr_at = (unsigned)sp * stride, rd_at = (unsigned)(sp ^ h) * stride;
        // ---- rows: keep t = (sp & h ? 1 : 0)
        const bool hi_t = (sp & h) != 0;
        if (holds) {
            if (hi_t) {
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const d4 xr = ore[0][g], xi = oim[0][g];
                    ore[0][g] = ore[1][g];
                    oim[0][g] = oim[1][g];
                    ore[1][g] = xr;
                    oim[1][g] = xi;
                }
            }
            double* mine = lds + wr_at;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    mine[(g * 8 + r) * 64] = ore[1][g][r];
                    mine[(g * 8 + 4 + r) * 64] = oim[1][g][r];
                }
        }
        __syncthreads();
        if (holds) {
            const double* theirs = lds + rd_at;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ore[0][g][r] += theirs[(g * 8 + r) * 64];
                    oim[0][g][r] += theirs[(g * 8 + 4 + r) * 64];
                }
        }
        const int row_t = rg * CMB_ROWS + lq_t + (hi_t ? 16 : 0);
        int col_g = col0 + lb_t, kept = NG;                // first column and number of the sub-tiles this wave still holds (ore[0][0 .. kept))
        if (n >= 4) {
            // ---- instance groups, upper / lower half (NG = 4: two phases; NG = 2: one)
            h >>= 1;
            wr_at = rd_at;                                     // (this wave's last read)
            rd_at = (unsigned)((sp ^ h) ^ (n >> 1)) * stride;  // (the partner's last read)
            const bool hi_g = (sp & h) != 0;
            constexpr int GH = NG / 2;
            if (holds) {
                if (hi_g) {
#pragma unroll
                    for (int g = 0; g < GH; ++g) {
                        const d4 xr = ore[0][g], xi = oim[0][g];
                        ore[0][g] = ore[0][GH + g];
                        oim[0][g] = oim[0][GH + g];
                        ore[0][GH + g] = xr;
                        oim[0][GH + g] = xi;
                    }
                }
                double* mine = lds + wr_at;
#pragma unroll
                for (int g = 0; g < GH; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mine[(g * 8 + r) * 64] = ore[0][GH + g][r];
                        mine[(g * 8 + 4 + r) * 64] = oim[0][GH + g][r];
                    }
            }
            __syncthreads();
            if (holds) {
                const double* theirs = lds + rd_at;
#pragma unroll
                for (int g = 0; g < GH; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ore[0][g][r] += theirs[(g * 8 + r) * 64];
                        oim[0][g][r] += theirs[(g * 8 + 4 + r) * 64];
                    }
            }
            col_g += hi_g ? 16 * GH : 0;
            kept = GH;
            if constexpr (NG == 4) {
                if (n >= 8) {
                    // ---- eight waves: the last halving, g = 0 / 1 of the kept pair
                    const int h1 = h >> 1;
                    wr_at = rd_at;
                    rd_at = (unsigned)(((sp ^ h1) ^ h) ^ (n >> 1)) * stride;
                    const bool hi_l = (sp & h1) != 0;
                    if (hi_l) {
                        const d4 xr = ore[0][0], xi = oim[0][0];
                        ore[0][0] = ore[0][1];
                        oim[0][0] = oim[0][1];
                        ore[0][1] = xr;
                        oim[0][1] = xi;
                    }
                    double* mine = lds + wr_at;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        mine[r * 64] = ore[0][1][r];
                        mine[(4 + r) * 64] = oim[0][1][r];
                    }
                    __syncthreads();
                    const double* theirs = lds + rd_at;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        ore[0][0][r] += theirs[r * 64];
                        oim[0][0][r] += theirs[(4 + r) * 64];
                    }
                    col_g += hi_l ? 16 : 0;
                    kept = 1;
                }
            }
        }
        if (!holds) return;
        // the fused epilogue on the kept sub-tiles, one 16 x 16 sub-tile per trip (ONE copy of the epilogue's code: the sub-tiles
        // move down through ore[0][0]; the whole tile took four trips on one wave before)
#pragma clang loop unroll(disable)
        for (int q = 0; q < kept; ++q) {
            const d4 fre[1][1] = {{ore[0][0]}}, fim[1][1] = {{oim[0][0]}};
            store_tile<1, 1>(a.epi, row_t, col_g + 16 * q, fre, fim);
#pragma unroll
            for (int g = 0; g + 1 < NG; ++g) {
                ore[0][g] = ore[0][g + 1];
                oim[0][g] = oim[0][g + 1];
            }
        }
        return;
    }
    store_tile<RT, NG>(a.epi, rg * CMB_ROWS + lq_t, col0 + lb_t, ore, oim);
}

template <int NRE4, int NIM4, int STAT>
__global__ __launch_bounds__(512, 2) void rhs_combine_kernel(const CombineArgs a) {
    combine_body<NRE4, NIM4, STAT, combine_ng(NRE4, NIM4, STAT)>(a);
}

// The same with 32 instances per wave for stacks of up to two plane groups: sweeps of so few instances that 64-instance
// waves, even eight to a pair, leave SIMDs without work (n = 1024: below 512 instances) get twice the pairs.
template <int NRE4, int NIM4, int STAT>
__global__ __launch_bounds__(512, 2) void rhs_combine_small_kernel(const CombineArgs a) {
    static_assert(NRE4 + NIM4 <= 2, "the wide variants already run 32 instances per wave");
    combine_body<NRE4, NIM4, STAT, 2>(a);
}

// (Measured in round 5 and not kept: 64 instances per wave for stacks with three or four plane groups of both kinds on ONE wave
// per SIMD -- 256-thread workgroups, 292 registers a lane, no spill, half the operand loads per MFMA.  General complex operators +
// static operator, n = 1024, 4096 instances: 3.27 ms per evaluation against 2.82 ms for two 32-instance waves per SIMD
// (profiles/r05_summary.md): without a partner wave every MFMA -> vector FMA dependency and every load that is late shows.)

// ---- the instantiations that exist, and where ------------------------------------------------------------------------------
// libmidyn.so is built from several translation units so that hipcc compiles the kernel families side by side (the device
// side of ONE unit is compiled serially: five minutes for everything).  The host side of every entry point stays in
// midyn.hip; a family's kernels are instantiated in its own unit (midyn_tu_combine.hip defines MIDYN_TU_COMBINE), every
// other unit sees `extern template` declarations of the same list and only takes the kernels' addresses.  A kernel the
// host code selects but the list does not name is an undefined symbol of the library: __graft_entry__.build() refuses it.
#define MIDYN_FOR_STAT(X, ...) X(__VA_ARGS__, 0) X(__VA_ARGS__, 1) X(__VA_ARGS__, 2) X(__VA_ARGS__, 3)
#define MIDYN_COMBINE_PAIRS_UP_TO_TWO_GROUPS(X) \
    MIDYN_FOR_STAT(X, 0, 1) MIDYN_FOR_STAT(X, 0, 2) MIDYN_FOR_STAT(X, 1, 0) MIDYN_FOR_STAT(X, 2, 0) MIDYN_FOR_STAT(X, 1, 1)
#define MIDYN_COMBINE_PAIRS_BOTH_KINDS(X) \
    MIDYN_COMBINE_PAIRS_UP_TO_TWO_GROUPS(X) MIDYN_FOR_STAT(X, 1, 2) MIDYN_FOR_STAT(X, 2, 1) MIDYN_FOR_STAT(X, 2, 2)
#ifdef MIDYN_TU_COMBINE
#define MIDYN_COMBINE_EXTERN
#else
#define MIDYN_COMBINE_EXTERN extern
#endif
#ifdef MIDYN_TU_COMBINE_WIDE     // midyn_tu_combine_wide.hip: three / four groups of one kind, and the 32-instance waves of small sweeps
#define MIDYN_COMBINE_WIDE_EXTERN
#else
#define MIDYN_COMBINE_WIDE_EXTERN extern
#endif
#define MIDYN_X(R_, I_, S_) MIDYN_COMBINE_EXTERN template __global__ void rhs_combine_kernel<R_, I_, S_>(const CombineArgs);
MIDYN_COMBINE_PAIRS_BOTH_KINDS(MIDYN_X)
#undef MIDYN_X
#define MIDYN_X(R_, I_, S_) MIDYN_COMBINE_WIDE_EXTERN template __global__ void rhs_combine_kernel<R_, I_, S_>(const CombineArgs);
MIDYN_FOR_STAT(MIDYN_X, 0, 3) MIDYN_FOR_STAT(MIDYN_X, 0, 4) MIDYN_FOR_STAT(MIDYN_X, 3, 0) MIDYN_FOR_STAT(MIDYN_X, 4, 0)
#undef MIDYN_X
#ifdef MIDYN_TU_COMBINE_MANY     // midyn_tu_combine_many.hip: 9 - 12 operators per kind with planes of both kinds (a third group)
#define MIDYN_COMBINE_MANY_EXTERN
#else
#define MIDYN_COMBINE_MANY_EXTERN extern
#endif
#define MIDYN_X(R_, I_, S_) MIDYN_COMBINE_MANY_EXTERN template __global__ void rhs_combine_kernel<R_, I_, S_>(const CombineArgs);
MIDYN_FOR_STAT(MIDYN_X, 3, 1) MIDYN_FOR_STAT(MIDYN_X, 3, 2) MIDYN_FOR_STAT(MIDYN_X, 3, 3) MIDYN_FOR_STAT(MIDYN_X, 1, 3) MIDYN_FOR_STAT(MIDYN_X, 2, 3)
#undef MIDYN_X
#define MIDYN_X(R_, I_, S_) MIDYN_COMBINE_WIDE_EXTERN template __global__ void rhs_combine_small_kernel<R_, I_, S_>(const CombineArgs);
MIDYN_COMBINE_PAIRS_UP_TO_TWO_GROUPS(MIDYN_X)
#undef MIDYN_X

}  // namespace midyn
