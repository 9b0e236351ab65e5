// probe (round 4): the sweep contraction as COMBINE (fp64 MFMA, K = the operator planes) + APPLY (fp64 VALU):
//   g[r][b]  = sum_j a_j[r][kk] c_j[b]        one v_mfma_f64_16x16x4 per 4 planes: 16 rows x 16 instances, all FMAs useful
//   out[r][b] += (i g[r][b]) y[kk][b]          two v_fma_f64 per (row, instance) in the lane that holds D[r][b]
// against k + 1 MFMA GEMMs (2 MFMA-FMAs per plane, row and instance): per (row, kk, instance) 8 + 2 FMAs instead of 16.
// No LDS, no barrier: operator fragments are pre-arranged in MFMA operand order (coalesced 512-byte loads), the state
// row y[kk][.] is read straight from the stage input.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/probes/cbp tools/combine_probe.hip && build/probes/cbp [N] [sparse]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
static double rnd() { return rand() / (double)RAND_MAX - 0.5; }

// frags: [row group][list entry][kk % 16][t < RT][q < NQ][lane] doubles; lane l <-> (row l % 16 of tile t, plane 4 q + l / 16)
// list_ptr[row group .. +1], list_idx[entry] = kk block (16 kk); Y [kk][ldy] complex; coeff [instance][NQ * 4] doubles
template <int NQ, int RT, int NG, int WR, int WI>
__global__ __launch_bounds__(64 * WR * WI, 2) void combine_kernel(const double* __restrict__ frags, const int* __restrict__ list_ptr,
                                                                   const int* __restrict__ list_idx, const double2* __restrict__ Y,
                                                                   int ldy, const double* __restrict__ coeff, double2* __restrict__ Out,
                                                                   int n_row_groups) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave % WR, wi = wave / WR;
    const int rgb = n_row_groups / WR;
    const int rg = (blockIdx.x % rgb) * WR + wr;
    const int inst0 = ((blockIdx.x / rgb) * WI + wi) * (NG * 16);
    const int lb = lane & 15, lq = lane >> 4;
    // B operands of the MFMAs: c[plane 4 q + lq][instance 16 g + lb]
    double cb[NG][NQ];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int q = 0; q < NQ; ++q) cb[g][q] = coeff[(size_t)(inst0 + 16 * g + lb) * (NQ * 4) + 4 * q + lq];
    double ore[RT][NG][4], oim[RT][NG][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r) ore[t][g][r] = oim[t][g][r] = 0.0;
    const int e0 = list_ptr[rg], e1 = list_ptr[rg + 1];
    const double* __restrict__ fr = frags + (size_t)e0 * (16 * RT * NQ * 64) + lane;
    const double2* __restrict__ yb = Y + inst0 + lb;
    double a_buf[2][RT][NQ];
    double2 y_buf[2][NG];
    const int steps = (e1 - e0) * 16;
    auto load = [&](int s, int b) {
        const int sc = s < steps ? s : steps - 1;
        const int kk = list_idx[e0 + (sc >> 4)] * 16 + (sc & 15);
        const double* __restrict__ fn = fr + (size_t)sc * (RT * NQ * 64);
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int q = 0; q < NQ; ++q) a_buf[b][t][q] = fn[(t * NQ + q) * 64];
#pragma unroll
        for (int g = 0; g < NG; ++g) y_buf[b][g] = yb[(size_t)kk * ldy + 16 * g];
    };
    auto compute = [&](int b) {
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            d4 gi[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                gi[g] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int q = 0; q < NQ; ++q) gi[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_buf[b][t][q], cb[g][q], gi[g], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    ore[t][g][r] = fma(-gi[g][r], y_buf[b][g].y, ore[t][g][r]);
                    oim[t][g][r] = fma(gi[g][r], y_buf[b][g].x, oim[t][g][r]);
                }
        }
    };
    if (steps > 0) {
        load(0, 0);
        for (int s = 0; s < steps; s += 2) {      // (steps is a multiple of 16)
            load(s + 1, 1);
            compute(0);
            load(s + 2, 0);
            compute(1);
        }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                Out[(size_t)((rg * RT + t) * 16 + lq + 4 * r) * ldy + inst0 + 16 * g + lb] = make_double2(ore[t][g][r], oim[t][g][r]);
}

int main(int argc, char** argv) {
    const int n = 1024, N = argc > 1 ? atoi(argv[1]) : 4096, sparse = argc > 2 ? atoi(argv[2]) : 0, K = 8;
    constexpr int NQ = 2, RT = 2, NG = 4, WR = 4, WI = 2;
    srand(11);
    std::vector<double> A((size_t)K * n * n, 0.0);     // A[j][r][kk]: imaginary parts
    for (int j = 0; j < K; ++j)
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c)
                if (!sparse || ((r < n / 2) != (c < n / 2))) A[((size_t)j * n + r) * n + c] = rnd();
    std::vector<double2> Y((size_t)n * N);
    for (auto& v : Y) v = make_double2(rnd(), rnd());
    std::vector<double> C((size_t)N * K);
    for (auto& v : C) v = rnd();
    const int nrg = n / (16 * RT);
    std::vector<int> lptr(nrg + 1, 0), lidx;
    for (int rg = 0; rg < nrg; ++rg) {
        for (int kb = 0; kb < n / 16; ++kb)
            if (!sparse || ((rg * 16 * RT < n / 2) != (kb * 16 < n / 2))) lidx.push_back(kb);
        lptr[rg + 1] = (int)lidx.size();
    }
    std::vector<double> F((size_t)lidx.size() * 16 * RT * NQ * 64);
    for (int rg = 0; rg < nrg; ++rg)
        for (int e = lptr[rg]; e < lptr[rg + 1]; ++e)
            for (int k16 = 0; k16 < 16; ++k16)
                for (int t = 0; t < RT; ++t)
                    for (int q = 0; q < NQ; ++q)
                        for (int l = 0; l < 64; ++l)
                            F[((((size_t)e * 16 + k16) * RT + t) * NQ + q) * 64 + l] =
                                A[((size_t)(4 * q + l / 16) * n + (rg * RT + t) * 16 + l % 16) * n + lidx[e] * 16 + k16];
    double *dF, *dC;
    double2 *dY, *dOut;
    int *dP, *dI;
    CHECK(hipMalloc(&dF, F.size() * 8)); CHECK(hipMalloc(&dC, C.size() * 8)); CHECK(hipMalloc(&dY, Y.size() * 16)); CHECK(hipMalloc(&dOut, Y.size() * 16));
    CHECK(hipMalloc(&dP, lptr.size() * 4)); CHECK(hipMalloc(&dI, lidx.size() * 4));
    CHECK(hipMemcpy(dF, F.data(), F.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dY, Y.data(), Y.size() * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dP, lptr.data(), lptr.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dI, lidx.data(), lidx.size() * 4, hipMemcpyHostToDevice));
    auto kern = combine_kernel<NQ, RT, NG, WR, WI>;
    const int blocks = (nrg / WR) * (N / (NG * 16 * WI));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WR * WI), 0, 0, dF, dP, dI, dY, N, dC, dOut, nrg);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WR * WI), 0, 0, dF, dP, dI, dY, N, dC, dOut, nrg);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    CHECK(hipGetLastError());
    std::vector<double2> out((size_t)n * N);
    CHECK(hipMemcpy(out.data(), dOut, out.size() * 16, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int rr = 0; rr < 6; ++rr) {
        const int r = (rr * 397 + 5) % n;
        for (int cc = 0; cc < 8; ++cc) {
            const int b = (cc * 911 + 3) % N;
            double re = 0, im = 0;
            for (int kk = 0; kk < n; ++kk) {
                double g = 0;
                for (int j = 0; j < K; ++j) g += C[(size_t)b * K + j] * A[((size_t)j * n + r) * n + kk];
                re += -g * Y[(size_t)kk * N + b].y;
                im += g * Y[(size_t)kk * N + b].x;
            }
            worst = std::max(worst, std::max(fabs(out[(size_t)r * N + b].x - re), fabs(out[(size_t)r * N + b].y - im)));
        }
    }
    const double pairs = (double)lidx.size() * 16 * (16 * RT) * N;
    const double fl = pairs * (K + 2) * 2.0, gemm_fl = pairs * K * 4.0;
    printf("%d workgroups, check max|d| = %.2e  %.4f ms per launch: combine + apply %.2f GFLOP = %.2f TFLOP/s = %.4f of 78.6; the MFMA GEMM "
           "formulation executes %.2f GFLOP and would need %.4f of its peak for this time\n", blocks, worst, ms, fl / 1e9, fl / ms / 1e9,
           fl / ms / 1e9 / 78.6, gemm_fl / 1e9, gemm_fl / ms / 1e9 / 78.6);
    return 0;
}
