// probe: ell_sweep_split_kernel<2, RPT> (csrc/midyn_resident.h) on the synthetic cfg-5-shaped problem of sweep_probe.hip
// (n = 4096, 19 slots per row, 20 steps x 9 Chebyshev terms, frame phases): 2 or 4 workgroups per instance, partners
// adjacent (part_major = 0) or on the same XCD (part_major = 1), checked against the one-workgroup kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ssp tools/sweep_split_probe.hip && /tmp/ssp [instances] [nsplit] [part_major]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/midyn.h"
#include "../qiskit_dynamics_amd/csrc/midyn_kernels.h"
#include "../qiskit_dynamics_amd/csrc/midyn_resident.h"
using namespace midyn;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> T* upload(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main(int argc, char** argv) {
    const int nsplit = argc > 2 ? atoi(argv[2]) : 2, part_major = argc > 3 ? atoi(argv[3]) : 1;
    const int B = argc > 1 ? atoi(argv[1]) : 128, n = 4096, wsp = 19, nseg = 9, k = 8, nsteps = 20, K = 9, P = 2;
    srand(1);
    std::vector<double> val((size_t)wsp * n);
    std::vector<int> col((size_t)wsp * n), tags(wsp);
    for (int e = 0; e < wsp; ++e) {
        tags[e] = (e < 11 ? 0 : e - 10) | (1 << 8);
        const int mask = e < 11 ? (3 << e) & (n - 1) : 1 << (e - 11);
        for (int r = 0; r < n; ++r) {
            col[(size_t)e * n + r] = r ^ (mask ? mask : 1);
            val[(size_t)e * n + r] = (rand() / (double)RAND_MAX - 0.5) * 0.02;
        }
    }
    const int R = 2 * nsteps;
    std::vector<double> S((size_t)B * R * k), hs(nsteps, 0.25), par(nsteps, 3.0), coef((size_t)nsteps * (K + 1));
    for (auto& x : S) x = rand() / (double)RAND_MAX;
    for (auto& x : coef) x = 0.1 * rand() / (double)RAND_MAX;
    std::vector<double2> E((size_t)R * n);
    for (int t = 0; t < R; ++t)
        for (int r = 0; r < n; ++r) E[(size_t)t * n + r] = make_double2(cos(0.001 * t * r), sin(0.001 * t * r));
    std::vector<int> rows(3 * nsteps), Kv(nsteps, K), reps(nsteps, 1), save(nsteps, -1);
    for (int st = 0; st < nsteps; ++st) { rows[3 * st] = 2 * st; rows[3 * st + 1] = 2 * st + 1; rows[3 * st + 2] = 2 * st + 1; }
    save[nsteps - 1] = 1;
    std::vector<double2> y(n);
    for (int r = 0; r < n; ++r) y[r] = make_double2(1.0 / sqrt((double)n), 0.0);
    SweepArgs a{};
    a.val = upload(val); a.col = upload(col); a.tags = upload(tags); a.wsp = wsp; a.wre = 0; a.n = n; a.n_pad = n; a.has_static = 1; a.k = k; a.nseg = nseg;
    a.S = upload(S); a.inst_stride = (long long)R * k; a.E = upload(E); a.rows = upload(rows); a.hs = upload(hs); a.save = upload(save);
    a.nsteps = nsteps; a.ser_K = upload(Kv); a.ser_reps = upload(reps); a.ser_par = upload(par); a.coef = upload(coef); a.stride = K + 1;
    a.y0 = upload(y); a.y0_shared = 1;
    double2 *out, *out_ref; CHECK(hipMalloc(&out, (size_t)B * P * n * sizeof(double2))); CHECK(hipMalloc(&out_ref, (size_t)B * P * n * sizeof(double2)));
    a.P = P;
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // reference: one workgroup per instance
    auto fn = ell_sweep_kernel<2, 4, 1024, 0>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    double2 *dt, *stash;
    CHECK(hipMalloc(&dt, (size_t)nsteps * n * sizeof(double2)));
    CHECK(hipMalloc(&stash, (size_t)B * 2 * n * sizeof(double2)));
    hipLaunchKernelGGL(sweep_dtable_kernel, dim3((nsteps * n + 255) / 256), dim3(256), 0, 0, a.E, a.rows, nsteps, n, dt);
    CHECK(hipDeviceSynchronize());
    a.Dt = dt; a.stash = stash;
    a.out = out_ref;
    float ms_ref = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(fn, dim3(B), dim3(SWEEP_THREADS), 2 * (n + 1) * sizeof(double2), s, a);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms_ref, e0, e1));
    }
    a.out = out;
    using SplitFn = void (*)(const SweepSplitArgs);
    const int rpt = n / (SWEEP_THREADS * nsplit);
    SplitFn sfn = rpt == 1 ? ell_sweep_split_kernel<2, 1> : ell_sweep_split_kernel<2, 2>;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sfn), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
    const size_t ring_bytes = (size_t)B * 4 * 2 * 2 * n * sizeof(unsigned long long);
    unsigned long long* ring; CHECK(hipMalloc(&ring, ring_bytes));
    int* err; CHECK(hipHostMalloc(&err, sizeof(int), hipHostMallocMapped)); *err = 0;
    int* derr; CHECK(hipHostGetDevicePointer((void**)&derr, err, 0));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemsetAsync(ring, 0xFF, ring_bytes, s));
        SweepSplitArgs sa{}; sa.a = a; sa.nsplit = nsplit; sa.ring = ring; sa.err = derr; sa.part_major = part_major;
        void* params[1] = {&sa};
        CHECK(hipEventRecord(e0, s));
        CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(sfn), dim3(B * nsplit), dim3(SWEEP_THREADS), params, (unsigned)(2 * n * sizeof(double2)), s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double2> o((size_t)B * P * n), oref((size_t)B * P * n);
        CHECK(hipMemcpy(o.data(), out, o.size() * sizeof(double2), hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(oref.data(), out_ref, o.size() * sizeof(double2), hipMemcpyDeviceToHost));
        double worst = 0;
        for (int bb = 0; bb < B; ++bb)
            for (int r = 0; r < n; ++r) {
                const size_t i = ((size_t)bb * P + 1) * n + r;
                worst = std::max(worst, std::max(fabs(o[i].x - oref[i].x), fabs(o[i].y - oref[i].y)));
            }
        printf("B %d nsplit %d part_major %d: %.3f ms per launch = %.2f us per term (one workgroup per instance: %.3f ms = %.2f us), max|d| = %.2e, err %d\n",
               B, nsplit, part_major, ms, ms * 1e3 / (nsteps * K), ms_ref, ms_ref * 1e3 / (nsteps * K), worst, *err);
    }
    return 0;
}
