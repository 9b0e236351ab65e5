// probe: the batched RHS contraction zgemm_seg_kernel (csrc/midyn_kernels.h) on a synthetic cfg-3-shaped problem, timed
// standalone (compiles in seconds, no libmidyn): M = K = 1024, 8 purely imaginary operators that couple the two halves
// of the basis only (parity sectors), N state columns with their own coefficient rows, fused RK4 stage-2 epilogue.
//   variant 0: SPARSE 128x128 work-list kernel (the headline route)       variant 1: dense 128x128 kernel, same stack
//   variant 2: dense complex operators, 3M 64x64 kernel                  variant 3: dense complex, 4M 128x128
//   variant 4: SPARSE 64x128 tile (64-row panels)                        variant 5: SPARSE 64x64 tile
//   variant 9: SPARSE 128x128 on four waves with 64x64 wave tiles (see below)
//   variant 10 (with -DMIDYN_EXPERIMENT_LIST_KERNEL=...): hand-scheduled list kernel, tools/experiments/gemm_list_kernel.h
// Measured with this probe in round 3 and NOT adopted (kernels removed again; N = 4096, ms per launch, this probe's stack):
//   headline SPARSE kernel 1.091-1.103; the same with PLANAR operator tiles (8-byte elements, 16 KB per tile, ds_read_b64
//   fragments kept apart from ds_read2st64 pairing) 1.123; planar tiles + TWO list entries per barrier (128 MFMAs per
//   wave between barriers) 1.130, with 1:1 instead of 2:1 MFMA : ds_read interleave 1.133, reads first 1.176;
//   64 x 128 and 64 x 64 work-list tiles for the 512-column shard 168 / 172 us against 160 us (128 x 128, 8 splits).
//   variant 9 (still here): the same kernel source with FOUR waves and 64 x 64 wave tiles (32 MFMAs per k-step, one
//   workgroup of 256 threads per CU, 512 registers per lane): hipcc takes 256 VGPRs + 256 AGPRs and still spills 167
//   registers -- 3.20 ms.  A wave tile of that size needs hand-placed AGPR accumulators.
// Checks a few output rows against a host evaluation.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/probes/gp tools/gemm_probe.hip && build/probes/gp [N] [variant] [splits]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../include/midyn.h"
#include "../qiskit_dynamics_amd/csrc/midyn_kernels.h"
#ifdef GEMM_PROBE_EXTRA
#include GEMM_PROBE_EXTRA
#endif
using namespace midyn;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> T* upload(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
static double rnd() { return rand() / (double)RAND_MAX - 0.5; }

template <int BM, int BN, int WM, int WN, int BK, int MODE, bool SPARSE, int MINW = 2>
static int run(const GemmArgs& g, hipStream_t s, int reps, float* ms_out) {
    constexpr size_t SMEM = (size_t)2 * BK * (BM + BN) * sizeof(double2);
    auto kern = zgemm_seg_kernel<BM, BN, WM, WN, BK, MODE, MINW, SPARSE>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
    const int blocks = (g.M / BM) * (g.N / BN) * g.splits;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WM * WN), SMEM, s, g);
    CHECK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WM * WN), SMEM, s, g);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(ms_out, e0, e1));
    *ms_out /= reps;
    return 0;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 4096, variant = argc > 2 ? atoi(argv[2]) : 0, splits = argc > 3 ? atoi(argv[3]) : 1;
    const int n = 1024, k = 8, reps = 40;
    const bool cplx = variant == 2 || variant == 3 || variant == 13 || variant == 14;
    srand(7);
    // operators: purely imaginary, coupling rows of one half to columns of the other (variant 0/1); dense complex (2/3)
    std::vector<double2> A((size_t)k * n * n, make_double2(0.0, 0.0));
    for (int j = 0; j < k; ++j)
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c) {
                const bool coupled = (r < n / 2) != (c < n / 2);
                if (cplx) A[((size_t)j * n + r) * n + c] = make_double2(rnd(), rnd());
                else if (coupled) A[((size_t)j * n + r) * n + c] = make_double2(0.0, rnd());
            }
    std::vector<double2> Y((size_t)n * N), Ys((size_t)n * N), Acc((size_t)n * N);
    for (auto& v : Y) v = make_double2(rnd(), rnd());
    for (auto& v : Ys) v = make_double2(rnd(), rnd());
    for (auto& v : Acc) v = make_double2(rnd(), rnd());
    std::vector<double> S((size_t)N * k);
    for (auto& v : S) v = rnd();
    std::vector<double2> E(n), En(n);
    for (int r = 0; r < n; ++r) { E[r] = make_double2(cos(0.3 * r), sin(0.3 * r)); En[r] = make_double2(cos(0.7 * r), sin(0.7 * r)); }
    std::vector<int> seg_list(k);
    for (int j = 0; j < k; ++j) seg_list[j] = (j << 2) | (cplx ? 0 : 2);
    // work lists of PH-row panels (128; 64 for the 64-row tiles): K tile outer, segment inner, only the tiles of the other half
    const int PH = (variant == 4 || variant == 5) ? 64 : 128;
    std::vector<int> wptr(n / PH + 1, 0), widx;
    for (int bm = 0; bm < n / PH; ++bm) {
        for (int kt = 0; kt < n / 16; ++kt) {
            const bool coupled = (bm * PH < n / 2) != (kt * 16 < n / 2);
            if (!coupled) continue;
            for (int j = 0; j < k; ++j) widx.push_back((kt << 8) | seg_list[j]);
        }
        wptr[bm + 1] = (int)widx.size();
    }
    GemmArgs g{};
    double2* dA = upload(A);
    double2 *dYin = upload(Y), *dY = upload(Ys), *dAcc = upload(Acc), *dOut = nullptr, *dPart = nullptr;
    CHECK(hipMalloc(&dOut, (size_t)n * N * sizeof(double2)));
    g.A = dA; g.a_seg_stride = (long long)n * n; g.lda = n; g.B = dYin; g.ldb = N; g.M = n; g.N = N; g.K = n;
    g.seg_list = upload(seg_list); g.n_act = k; g.has_static = 0; g.coeff = upload(S); g.inst_stride = k; g.m_cols = 1; g.n_inst = N;
    g.ablate = argc > 4 ? atoi(argv[4]) : 0;
    g.batch = 1; g.splits = splits; g.work_ptr = upload(wptr); g.work_idx = upload(widx);
    if (splits > 1) { CHECK(hipMalloc(&dPart, (size_t)splits * n * N * sizeof(double2))); g.partial = dPart; }
    g.epi.mode = EPI_RK2; g.epi.ld = N; g.epi.h = 0.005; g.epi.e_cur = upload(E); g.epi.e_next = upload(En);
    g.epi.y = dY; g.epi.acc = dAcc; g.epi.yin_next = dOut;
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float ms = 0;
    double flops = 0;
    const double listed = (double)widx.size();
    int st = 0;
    if (variant == 0) { st = run<128, 128, 2, 4, 16, 2, true>(g, s, reps, &ms); flops = listed * 128 * 16 * (double)N * 4; }
    else if (variant == 1) { g.work_ptr = nullptr; g.work_idx = nullptr; st = run<128, 128, 2, 4, 16, 2, false>(g, s, reps, &ms); flops = (double)k * n * n * N * 4; }
    else if (variant == 2) { g.work_ptr = nullptr; g.work_idx = nullptr; st = run<64, 64, 2, 2, 16, 4, false>(g, s, reps, &ms); flops = (double)k * n * n * N * 6; }
    else if (variant == 4) { st = run<64, 128, 2, 4, 16, 2, true>(g, s, reps, &ms); flops = listed * 64 * 16 * (double)N * 4; }
    else if (variant == 5) { st = run<64, 64, 2, 2, 16, 2, true>(g, s, reps, &ms); flops = listed * 64 * 16 * (double)N * 4; }
    else if (variant == 11) { st = run<128, 128, 1, 8, 16, 2, true>(g, s, reps, &ms); flops = listed * 128 * 16 * (double)N * 4; }   // 8 waves, 128 x 16 wave tiles: half the scalings per MFMA
    else if (variant == 12) { g.work_ptr = nullptr; g.work_idx = nullptr; st = run<128, 128, 1, 8, 16, 2, false>(g, s, reps, &ms); flops = (double)k * n * n * N * 4; }
    else if (variant == 13) { g.work_ptr = nullptr; g.work_idx = nullptr; st = run<128, 64, 2, 4, 16, 4, false>(g, s, reps, &ms); flops = (double)k * n * n * N * 6; }   // 3M, 8 waves, 64 x 16 wave tiles
    else if (variant == 14) { g.work_ptr = nullptr; g.work_idx = nullptr; st = run<64, 128, 1, 8, 16, 4, false>(g, s, reps, &ms); flops = (double)k * n * n * N * 6; }
    else if (variant == 9) { st = run<128, 128, 2, 2, 16, 2, true, 1>(g, s, reps, &ms); flops = listed * 128 * 16 * (double)N * 4; }   // 4 waves, 64 x 64 wave tiles
#ifdef MIDYN_EXPERIMENT_LIST_KERNEL
    else if (variant == 10) {   // the hand-scheduled list kernel (tools/experiments/gemm_list_kernel.h; not adopted)
        constexpr size_t SMEM = (size_t)2 * 16 * 256 * sizeof(double2);
        auto kern = zgemm_list_kernel<2>;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        const int blocks = (g.M / 128) * (g.N / 128) * g.splits;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), SMEM, s, g);
        CHECK(hipEventRecord(e0, s));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), SMEM, s, g);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        ms /= reps;
        flops = listed * 128 * 16 * (double)N * 4;
#ifdef MIDYN_CLOCK_PROBE
        {
            long long* dclk = nullptr;
            CHECK(hipMalloc(&dclk, blocks * 2 * sizeof(long long)));
            GemmArgs gc = g;
            gc.sync = reinterpret_cast<int*>(dclk);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), SMEM, s, gc);
            CHECK(hipStreamSynchronize(s));
            std::vector<long long> hc(blocks * 2);
            CHECK(hipMemcpy(hc.data(), dclk, hc.size() * sizeof(long long), hipMemcpyDeviceToHost));
            double cs = 0, ws = 0, cmax = 0;
            for (int b = 0; b < blocks; ++b) { cs += hc[2 * b]; ws += hc[2 * b + 1]; cmax = std::max(cmax, (double)hc[2 * b]); }
            printf("clock probe: %.0f shader cycles, %.0f ticks of 100 MHz per workgroup -> %.3f GHz; workgroup lifetime %.1f us (max %.0f cycles)\n",
                   cs / blocks, ws / blocks, cs / ws * 0.1, ws / blocks * 0.01, cmax);
        }
#endif
    }
#endif
    else if (variant == 3) { g.work_ptr = nullptr; g.work_idx = nullptr; st = run<128, 128, 2, 4, 16, 0, false>(g, s, reps, &ms); flops = (double)k * n * n * N * 8; }
#ifdef GEMM_PROBE_EXTRA
    else st = probe_extra(variant, g, s, reps, &ms, &flops, A, n, k, N);
#endif
    if (st) return st;
    // check: acc' = acc + h/3 k, yin' = En o (y + h/2 k), k = conj(E) o C  (EPI_RK2; 40 launches accumulate into acc: use yin')
    std::vector<double2> out((size_t)n * N);
    if (splits == 1) {
        CHECK(hipMemcpy(out.data(), dOut, out.size() * sizeof(double2), hipMemcpyDeviceToHost));
        double worst = 0;
        for (int rr = 0; rr < 6; ++rr) {
            const int r = (rr * 397 + 5) % n;
            for (int cc = 0; cc < 8; ++cc) {
                const int c = (cc * 911 + 3) % N;
                double cr = 0, ci = 0;
                for (int j = 0; j < k; ++j) {
                    const double sj = S[(size_t)c * k + j];
                    for (int q = 0; q < n; ++q) {
                        const double2 a = A[((size_t)j * n + r) * n + q], b = Y[(size_t)q * N + c];
                        cr += sj * (a.x * b.x - a.y * b.y);
                        ci += sj * (a.x * b.y + a.y * b.x);
                    }
                }
                const double kr = E[r].x * cr + E[r].y * ci, ki = E[r].x * ci - E[r].y * cr;
                const double2 y = Ys[(size_t)r * N + c];
                const double zr = y.x + 0.5 * 0.005 * kr, zi = y.y + 0.5 * 0.005 * ki;
                const double er = En[r].x * zr - En[r].y * zi, ei = En[r].x * zi + En[r].y * zr;
                const double2 o = out[(size_t)r * N + c];
                worst = std::max(worst, std::max(fabs(o.x - er), fabs(o.y - ei)));
            }
        }
        printf("check max|d| = %.2e  ", worst);
    }
    printf("N %d variant %d splits %d: %.4f ms per launch, executed %.2f GFLOP = %.2f TFLOP/s = %.4f of 78.6\n", N, variant, splits, ms,
           flops / 1e9, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 78.6);
    return 0;
}
