// probe (round 4): the sweep contraction as "combine first" on the fp64 VECTOR ALUs instead of k+1 MFMA GEMMs.
//   out[r][b] = sum_kk ( sum_j c_j[b] A_j[r][kk] ) y[kk][b]          lane = instance b, A_j[r][kk] wave-uniform (SGPR operand)
// Per instance and operator element: k real FMAs for the combination + 2 (single-plane operators) for the product, against
// 2 (k+1) MFMA-FMAs of the GEMM formulation -- fp64 vector and fp64 matrix peak are the same 78.6 TFLOP/s on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/probes/svp tools/sweep_valu_probe.hip && build/probes/svp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
static double rnd() { return rand() / (double)RAND_MAX - 0.5; }

// Ap: [row block][kk][r < R][j < K] doubles (the one non-zero plane of the operators; here: imaginary parts)
// Y:  [kk][N] complex, C: [N][K] doubles, Out: [row][N] complex
// ABL (profiling, results wrong): 1 = y is not re-loaded, 2 = every wave and iteration reads the same 512 B of A, 4 = both
template <int R, int I, int K, int WAVES, int ABL = 0>
__global__ __launch_bounds__(64 * WAVES, 2) void sweep_valu_kernel(const double* __restrict__ Ap, const double2* __restrict__ Y,
                                                                    const double* __restrict__ C, double2* __restrict__ Out, int n,
                                                                    int N) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row_groups = n / (R * WAVES);
    const int rg = blockIdx.x % row_groups, ig = blockIdx.x / row_groups;
    const int rb = rg * WAVES + wave;                 // this wave's row block
    const int inst0 = ig * (64 * I) + lane;           // lane's instances: inst0 + 64 i
    double c[I][K];
#pragma unroll
    for (int i = 0; i < I; ++i)
#pragma unroll
        for (int j = 0; j < K; ++j) c[i][j] = C[(size_t)(inst0 + 64 * i) * K + j];
    double ore[R][I], oim[R][I];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < I; ++i) ore[r][i] = oim[r][i] = 0.0;
    const double* __restrict__ a = Ap + (size_t)rb * n * (R * K);
    const double2* __restrict__ y = Y + inst0;
    double2 yc[I], yn[I];
#pragma unroll
    for (int i = 0; i < I; ++i) yc[i] = y[64 * i];
    for (int kk = 0; kk < n; ++kk) {
        const int kn = kk + 1 < n ? kk + 1 : kk;
#pragma unroll
        for (int i = 0; i < I; ++i) {
            if (ABL & 1) { yn[i].x = yc[i].y + 1e-9; yn[i].y = yc[i].x; }
            else yn[i] = y[(size_t)kn * N + 64 * i];
        }
        const double* __restrict__ ak = (ABL & 2) ? Ap + (kk & 1) * (R * K) : a + (size_t)kk * (R * K);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double av[K];
#pragma unroll
            for (int j = 0; j < K; ++j) av[j] = ak[r * K + j];      // wave-uniform: scalar loads
#pragma unroll
            for (int i = 0; i < I; ++i) {
                double g = c[i][0] * av[0];
#pragma unroll
                for (int j = 1; j < K; ++j) g = fma(c[i][j], av[j], g);
                // operators i*g (purely imaginary): (i g)(yr + i yi) = -g yi + i g yr
                ore[r][i] = fma(-g, yc[i].y, ore[r][i]);
                oim[r][i] = fma(g, yc[i].x, oim[r][i]);
            }
        }
#pragma unroll
        for (int i = 0; i < I; ++i) yc[i] = yn[i];
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int i = 0; i < I; ++i) Out[(size_t)(rb * R + r) * N + inst0 + 64 * i] = make_double2(ore[r][i], oim[r][i]);
}

template <int R, int I, int WAVES, int ABL = 0>
static int run(int n, int N, const double* dA, const double2* dY, const double* dC, double2* dOut, float* ms) {
    constexpr int K = 8;
    auto kern = sweep_valu_kernel<R, I, K, WAVES, ABL>;
    const int blocks = (n / (R * WAVES)) * (N / (64 * I));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), 0, 0, dA, dY, dC, dOut, n, N);
    CHECK(hipEventRecord(e0, 0));
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * WAVES), 0, 0, dA, dY, dC, dOut, n, N);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(ms, e0, e1));
    *ms /= reps;
    CHECK(hipGetLastError());
    printf("R %d I %d waves %d: %d workgroups, ", R, I, WAVES, blocks);
    return 0;
}

int main(int argc, char** argv) {
    const int n = 1024, N = argc > 1 ? atoi(argv[1]) : 4096, K = 8, variant = argc > 2 ? atoi(argv[2]) : 0;
    srand(11);
    std::vector<double> A((size_t)K * n * n);     // A[j][r][kk]: imaginary parts
    for (auto& v : A) v = rnd();
    std::vector<double2> Y((size_t)n * N);
    for (auto& v : Y) v = make_double2(rnd(), rnd());
    std::vector<double> C((size_t)N * K);
    for (auto& v : C) v = rnd();
    auto pack = [&](int R) {
        std::vector<double> P((size_t)K * n * n);
        for (int rb = 0; rb < n / R; ++rb)
            for (int kk = 0; kk < n; ++kk)
                for (int r = 0; r < R; ++r)
                    for (int j = 0; j < K; ++j)
                        P[(((size_t)rb * n + kk) * R + r) * K + j] = A[((size_t)j * n + rb * R + r) * n + kk];
        return P;
    };
    const int R = variant == 1 ? 4 : (variant == 2 ? 8 : 8);
    std::vector<double> P = pack(R);
    double *dA, *dC;
    double2 *dY, *dOut;
    CHECK(hipMalloc(&dA, P.size() * 8));
    CHECK(hipMalloc(&dC, C.size() * 8));
    CHECK(hipMalloc(&dY, Y.size() * 16));
    CHECK(hipMalloc(&dOut, Y.size() * 16));
    CHECK(hipMemcpy(dA, P.data(), P.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dY, Y.data(), Y.size() * 16, hipMemcpyHostToDevice));
    float ms = 0;
    int st = 0;
    if (variant == 0) st = run<8, 4, 8>(n, N, dA, dY, dC, dOut, &ms);
    else if (variant == 1) st = run<4, 4, 8>(n, N, dA, dY, dC, dOut, &ms);
    else if (variant == 2) st = run<8, 2, 8>(n, N, dA, dY, dC, dOut, &ms);
    else if (variant == 3) st = run<8, 4, 4>(n, N, dA, dY, dC, dOut, &ms);
    else if (variant == 10) st = run<8, 4, 8, 1>(n, N, dA, dY, dC, dOut, &ms);
    else if (variant == 11) st = run<8, 4, 8, 2>(n, N, dA, dY, dC, dOut, &ms);
    else if (variant == 12) st = run<8, 4, 8, 3>(n, N, dA, dY, dC, dOut, &ms);
    else if (variant == 13) st = run<8, 2, 8, 3>(n, N, dA, dY, dC, dOut, &ms);
    if (st) return st;
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
    const double fl = (double)n * n * N * (K + 2) * 2.0;
    const double gemm_fl = (double)n * n * N * K * 4.0;
    printf("check max|d| = %.2e  %.4f ms per launch: %.2f GFLOP executed = %.2f TFLOP/s = %.4f of 78.6 (vector fp64); the MFMA GEMM "
           "formulation of the same product executes %.2f GFLOP -> would need %.4f of its peak for this time\n",
           worst, ms, fl / 1e9, fl / ms / 1e9, fl / ms / 1e9 / 78.6, gemm_fl / 1e9, gemm_fl / ms / 1e9 / 78.6);
    return 0;
}
