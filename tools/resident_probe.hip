// probe: rk4_resident_kernel (csrc/midyn_resident.h) on a synthetic cfg-2-shaped problem (n = 1024, 8 single-plane
// operators, two symmetry sectors of 512), timed per RHS evaluation.  For kernel iteration without rebuilding the
// library:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/rp tools/resident_probe.hip && /tmp/rp
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
#ifndef PROBE_HALFQ
#define PROBE_HALFQ false
#endif
#ifndef PROBE_WAVES
#define PROBE_WAVES 4
#endif
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> T* upload(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 1024, nseg = 8, NE = 8, nsteps = 500;
    const int nc = n / 64, nb = n / 16, half = nc / 2;
    std::vector<double2> ops((size_t)nseg * n * n, make_double2(0.0, 0.0));
    srand(1);
    for (int s = 0; s < nseg; ++s)
        for (int r = 0; r < n; ++r)
            for (int c = 0; c < n; ++c)
                if ((r < n / 2) != (c < n / 2)) ops[((size_t)s * n + r) * n + c].y = (rand() / (double)RAND_MAX - 0.5) * 0.05;
    std::vector<int> pairs(NE);
    for (int e = 0; e < NE; ++e) pairs[e] = (e << 1) | 1;
    std::vector<int> chunk_ptr(nb + 1, 0), chunk_idx, poll_ptr(nc + 1, 0), poll_idx;
    for (int rc = 0; rc < nc; ++rc) {
        for (int c = 0; c < nc; ++c)
            if ((rc < half) != (c < half)) poll_idx.push_back(c);
        poll_ptr[rc + 1] = (int)poll_idx.size();
    }
    for (int rb = 0; rb < nb; ++rb) {
        int slot = 0;
        for (int c = 0; c < nc; ++c)
            if ((rb / 4 < half) != (c < half)) chunk_idx.push_back((slot++ << 8) | c);
        chunk_ptr[rb + 1] = (int)chunk_idx.size();
    }
    const int R = 2 * nsteps + 1;
    std::vector<double> S((size_t)R * nseg), hs(nsteps, 0.005);
    for (auto& x : S) x = rand() / (double)RAND_MAX;
    std::vector<double2> E((size_t)R * n);
    for (int t = 0; t < R; ++t)
        for (int r = 0; r < n; ++r) E[(size_t)t * n + r] = make_double2(cos(0.001 * t * r), sin(0.001 * t * r));
    std::vector<int> rows(3 * nsteps);
    for (int st = 0; st < nsteps; ++st) { rows[3 * st] = 2 * st; rows[3 * st + 1] = 2 * st + 1; rows[3 * st + 2] = 2 * st + 2; }
    std::vector<double2> y(n);
    for (int r = 0; r < n; ++r) y[r] = make_double2(1.0 / sqrt((double)n), 0.0);
    ResidentArgs a{};
    a.ops = upload(ops); a.pairs = upload(pairs); a.n = n; a.n_pad = n; a.has_static = 0; a.k = nseg;
    a.S = upload(S); a.E = upload(E); a.rows = upload(rows); a.hs = upload(hs); a.save = nullptr;
    a.nsteps = nsteps; a.chunk_ptr = upload(chunk_ptr); a.chunk_idx = upload(chunk_idx);
    a.poll_ptr = upload(poll_ptr); a.poll_idx = upload(poll_idx);
    unsigned long long* ring; CHECK(hipMalloc(&ring, 4 * 2 * n * 8)); a.ring = ring;
    a.y = upload(y); a.out = nullptr;
    int* err; CHECK(hipMalloc(&err, 4)); CHECK(hipMemset(err, 0, 4)); a.err = err;
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        a.step_begin = 0; a.step_end = nsteps;
        CHECK(hipMemcpy(a.y, y.data(), n * sizeof(double2), hipMemcpyHostToDevice));
        CHECK(hipMemsetAsync(ring, 0xFF, 4 * 2 * n * 8, s));
        void* params[1] = {&a};
        CHECK(hipEventRecord(e0, s));
        CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(rk4_resident_kernel<8, PROBE_WAVES, PROBE_HALFQ>), dim3(n / PROBE_WAVES), dim3(64 * PROBE_WAVES), params, 0, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double2> out(n);
        CHECK(hipMemcpy(out.data(), a.y, n * sizeof(double2), hipMemcpyDeviceToHost));
        int herr; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        double nrm = 0, cs = 0;
        for (int r = 0; r < n; ++r) { nrm += out[r].x * out[r].x + out[r].y * out[r].y; cs += out[r].x * (r + 1) + out[r].y; }
        printf("n %d: %.3f us per RHS evaluation (%d steps)  |y|^2 = %.12f  checksum %.12e  err %d\n", n, ms * 1e3 / (4 * nsteps), nsteps, nrm, cs, herr);
    }
    return 0;
}
