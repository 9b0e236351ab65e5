// probe: ell_sweep_kernel (csrc/midyn_resident.h) in its three element forms -- general (12-byte elements), packed (column |
// sign), direct (LDS address; SP_DIRECT=1 in the environment gives every slot one sign and no unused entry) -- on a
// synthetic cfg-5-shaped problem (n = 4096, 19 slots per row with one magnitude per slot, 20 steps x 9 Chebyshev terms,
// frame phases): time per term of each form, difference of the results to the general form.
// -DMIDYN_SWEEP_ABLATE=1|2|3|4 ablates the operator pass / the gather pattern / the element loads / the multiply-adds,
// -DMIDYN_SWEEP_PREFETCH=n sets the element prefetch depth of the packed forms.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/probes/sp tools/sweep_probe.hip && build/probes/sp [instances] [order]
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
    const int order = argc > 2 ? atoi(argv[2]) : 2;
    const int B = argc > 1 ? atoi(argv[1]) : 128, n = 4096, wsp = 19, nseg = 9, k = 8, nsteps = 20, K = 9, P = 2;
    srand(1);
    std::vector<double> val((size_t)wsp * n);
    std::vector<int> col((size_t)wsp * n), tags(wsp), pk((size_t)wsp * n), pd((size_t)wsp * n);
    const bool direct = getenv("SP_DIRECT") != nullptr;   // one sign per slot, no unused entries
    std::vector<double> mag(wsp);
    for (int e = 0; e < wsp; ++e) {
        tags[e] = (e < 11 ? 0 : e - 10) | (1 << 8);
        const int mask = e < 11 ? (3 << e) & (n - 1) : 1 << (e - 11);
        mag[e] = (0.2 + rand() / (double)RAND_MAX) * 0.01;
        for (int r = 0; r < n; ++r) {
            const int c = r ^ (mask ? mask : 1);
            const bool neg = direct ? (e & 1) : (rand() & 1), unused = !direct && e == 18 && (r % 7 == 0);
            pd[(size_t)e * n + r] = (((c >> 11) << 12) | (c & 2047)) << 4;
            col[(size_t)e * n + r] = unused ? 0 : c;
            val[(size_t)e * n + r] = unused ? 0.0 : (neg ? -mag[e] : mag[e]);
            pk[(size_t)e * n + r] = unused ? n : (c | (neg ? (int)0x80000000 : 0));
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
    std::vector<double> smag(mag);
    for (int e = 0; e < wsp; ++e) if (e & 1) smag[e] = -smag[e];
    a.val = upload(val); a.col = upload(col); a.tags = upload(tags); a.pk = upload(pk); a.mag = upload(mag);
    const int* d_pk = a.pk; const double* d_mag = a.mag; const int* d_pd = upload(pd); const double* d_smag = upload(smag);
    a.wsp = wsp; a.wre = 0; a.n = n; a.n_pad = n; a.has_static = 1; a.k = k; a.nseg = nseg;
    a.S = upload(S); a.inst_stride = (long long)R * k; a.E = upload(E); a.rows = upload(rows); a.hs = upload(hs); a.save = upload(save);
    a.nsteps = nsteps; a.ser_K = upload(Kv); a.ser_reps = upload(reps); a.ser_par = upload(par); a.coef = upload(coef); a.stride = K + 1;
    a.y0 = upload(y); a.y0_shared = 1;
    double2* dt; CHECK(hipMalloc(&dt, (size_t)nsteps * n * sizeof(double2)));
    hipLaunchKernelGGL(sweep_dtable_kernel, dim3((nsteps * n + 255) / 256), dim3(256), 0, 0, a.E, a.rows, nsteps, n, dt);
    CHECK(hipDeviceSynchronize());
    a.Dt = dt;
    double2* stash; CHECK(hipMalloc(&stash, (size_t)B * 2 * n * sizeof(double2)));
    a.stash = stash;
    double2* outs[3];
    for (auto& o : outs) CHECK(hipMalloc(&o, (size_t)B * P * n * sizeof(double2)));
    a.P = P;
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    using Fn = void (*)(const SweepArgs);
    Fn fns[3] = {order == 2 ? (Fn)ell_sweep_kernel<2, 4, 1024, 0> : (Fn)ell_sweep_kernel<1, 4, 1024, 0>,
                 order == 2 ? (Fn)ell_sweep_kernel<2, 4, 1024, 1> : (Fn)ell_sweep_kernel<1, 4, 1024, 1>,
                 order == 2 ? (Fn)ell_sweep_kernel<2, 4, 1024, 2> : (Fn)ell_sweep_kernel<1, 4, 1024, 2>};
    const char* names[3] = {"ell_sweep_kernel<general>", "ell_sweep_kernel<packed>", "ell_sweep_kernel<direct>"};
    std::vector<std::vector<double2>> res(3, std::vector<double2>((size_t)B * n));
    for (int v = 0; v < (direct ? 3 : 2); ++v) {
        a.pk = v == 2 ? d_pd : d_pk; a.mag = v == 2 ? d_smag : d_mag;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fns[v]), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        a.out = outs[v];
        const size_t lds = v == 2 ? (size_t)(n / 2048) * 65536 : (size_t)order * (n + 1) * sizeof(double2);
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(fns[v], dim3(B), dim3(SWEEP_THREADS), lds, s, a);
            CHECK(hipEventRecord(e1, s));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms);
        }
        CHECK(hipGetLastError());
        std::vector<double2> o((size_t)B * P * n);
        CHECK(hipMemcpy(o.data(), outs[v], o.size() * sizeof(double2), hipMemcpyDeviceToHost));
        double nrm = 0, worst = 0;
        for (int bb = 0; bb < B; ++bb)
            for (int r = 0; r < n; ++r) {
                res[v][(size_t)bb * n + r] = o[((size_t)bb * P + 1) * n + r];
                const double2 x = res[v][(size_t)bb * n + r], x0 = res[0][(size_t)bb * n + r];
                if (bb == 0) nrm += x.x * x.x + x.y * x.y;
                worst = std::max(worst, std::max(fabs(x.x - x0.x), fabs(x.y - x0.y)));
            }
        printf("B %d order %d %-28s %.3f ms per launch = %.2f us per term (%d terms), |y|^2 = %.6e, max|d vs first| = %.2e\n", B, order, names[v], best,
               best * 1e3 / (nsteps * K), nsteps * K, nrm, worst);
    }
    return 0;
}
