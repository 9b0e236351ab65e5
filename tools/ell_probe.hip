// probe: ell_resident_kernel<0> (csrc/midyn_resident.h) on a synthetic sparse stack (N rows, W entries per row spread
// over the 2 * HALF + 1 neighbouring 64-row chunks), timed per round.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ep tools/ell_probe.hip && /tmp/ep [N] [W] [HALF]
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
#ifndef PROBE_EWU
#define PROBE_EWU 8
#endif
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> T* upload(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
    hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, W = argc > 2 ? atoi(argv[2]) : 27, HALF = argc > 3 ? atoi(argv[3]) : 6;
    const int nsteps = 500, nc = n / 64, nseg = 7;
    srand(1);
    std::vector<int> poll_ptr(nc + 1, 0), poll_idx;
    std::vector<std::vector<int>> slot(nc, std::vector<int>(nc, -1));
    for (int rc = 0; rc < nc; ++rc) {
        for (int c = 0; c < nc; ++c) {
            int d = std::abs(c - rc);
            d = std::min(d, nc - d);
            if (d <= HALF) { slot[rc][c] = (int)poll_idx.size() - poll_ptr[rc]; poll_idx.push_back(c); }
        }
        poll_ptr[rc + 1] = (int)poll_idx.size();
    }
    std::vector<double> val((size_t)W * n);
    std::vector<int> meta((size_t)W * n);
    for (int r = 0; r < n; ++r)
        for (int e = 0; e < W; ++e) {
            const int rc = r / 64, np_ = poll_ptr[rc + 1] - poll_ptr[rc];
            const int c = poll_idx[poll_ptr[rc] + rand() % np_] * 64 + rand() % 64;
            val[(size_t)e * n + r] = (rand() / (double)RAND_MAX - 0.5) * 0.1;
            meta[(size_t)e * n + r] = (slot[rc][c / 64] * 64 + c % 64) | ((rand() % nseg) << 16) | ((rand() & 1) << 22) | (1 << 23);
        }
    const int R = 2 * nsteps + 1;
    std::vector<double> S((size_t)R * nseg), hs(nsteps, 0.005);
    for (auto& x : S) x = rand() / (double)RAND_MAX;
    std::vector<int> rows(3 * nsteps);
    for (int st = 0; st < nsteps; ++st) { rows[3 * st] = 2 * st; rows[3 * st + 1] = 2 * st + 1; rows[3 * st + 2] = 2 * st + 2; }
    std::vector<double2> y(n);
    for (int r = 0; r < n; ++r) y[r] = make_double2(1.0 / sqrt((double)n), 0.0);
    EllArgs a{};
    a.val = upload(val); a.meta = upload(meta); a.wmax = W; a.n = n; a.n_pad = n; a.has_static = 0; a.k = nseg; a.nseg = nseg;
    a.S = upload(S); a.E = nullptr; a.rows = upload(rows); a.hs = upload(hs); a.save = nullptr; a.nsteps = nsteps;
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
        CHECK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(ell_resident_kernel<0, PROBE_EWU>), dim3(n / 64), dim3(64 * ELL_WAVES), params, 0, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<double2> out(n);
        CHECK(hipMemcpy(out.data(), a.y, n * sizeof(double2), hipMemcpyDeviceToHost));
        int herr; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        double nrm = 0;
        for (int r = 0; r < n; ++r) nrm += out[r].x * out[r].x + out[r].y * out[r].y;
        printf("N %d W %d poll %d chunks: %.3f us per round  |y|^2 = %.12f err %d\n", n, W, 2 * HALF + 1, ms * 1e3 / (4 * nsteps), nrm, herr);
    }
    return 0;
}
