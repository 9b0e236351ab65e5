// probe: cost of one round of a grid-synchronised persistent kernel on MI355X -- every workgroup reads a vector that
// ALL workgroups wrote in the previous round (device-coherent loads), reduces it, writes its own 4 entries of the next
// vector and passes a grid barrier.  This is the skeleton of a "resident operator" solve (operators in registers /
// LDS, the state vector exchanged through memory once per product); the question is what a round costs next to the
// 8.4 us per product of the launch-per-stage path.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/grid_barrier_probe tools/grid_barrier_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int THREADS = 256;
constexpr unsigned SPIN_LIMIT = 1u << 24;

// MODE 0: flat counter, relaxed data atomics + release/acquire on the counter
// MODE 1: flat counter, everything relaxed + explicit s_waitcnt (data stores are write-through atomics)
// MODE 2: as 1, but the barrier counter is spread: 8 counters (blockIdx % 8), the last arriver of each bumps a root
template <int MODE>
__global__ __launch_bounds__(THREADS) void rounds_kernel(double* ya, double* yb, unsigned* cnt, int* err, int ny,
                                                         int rows_per_wg, int rounds, int barrier_only) {
    __shared__ double red[THREADS / 64];
    __shared__ double total;
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    double* cur = ya;
    double* nxt = yb;
    for (int r = 0; r < rounds; ++r) {
        double s = 0.0;
        if (!barrier_only) {
            if (ny == 8 * THREADS) {   // all loads in flight before the first use
                double v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = __hip_atomic_load(cur + tid + i * THREADS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
            } else
                for (int i = tid; i < ny; i += THREADS) s += __hip_atomic_load(cur + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if ((tid & 63) == 0) red[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) total = red[0] + red[1] + red[2] + red[3];
            __syncthreads();
            if (tid < rows_per_wg) {
                const int row = wg * rows_per_wg + tid;
                __hip_atomic_store(nxt + row, total / ny * 0.5 + row * 1e-3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- grid barrier
        if (MODE == 0) {
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(r + 1) * nwg;
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                }
            }
            __syncthreads();
        } else if (MODE == 1) {
            __builtin_amdgcn_s_waitcnt(0);   // my stores have left
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(r + 1) * nwg;
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                }
            }
            __syncthreads();
        } else if (MODE == 3) {   // one flag per workgroup (plain stores, no read-modify-write), everybody polls all flags
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) __hip_atomic_store(cnt + wg, (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            for (int w = tid; w < nwg; w += THREADS)
                while (__hip_atomic_load(cnt + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(r + 1)) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                }
            __syncthreads();
        } else {
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (tid == 0) {
                const int grp = wg & 7, per = (nwg + 7 - grp) / 8;   // workgroups with blockIdx % 8 == grp
                const unsigned a = __hip_atomic_fetch_add(cnt + 32 * (1 + grp), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a == (unsigned)(r + 1) * per - 1) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned target = (unsigned)(r + 1) * 8;
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { *err = 1; break; }
                }
            }
            __syncthreads();
        }
        double* t = cur; cur = nxt; nxt = t;
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs\n", prop.name, cus);
    double *ya, *yb; unsigned* cnt; int* err;
    const int max_ny = 4096;
    CHECK(hipMalloc(&ya, max_ny * 8)); CHECK(hipMalloc(&yb, max_ny * 8));
    CHECK(hipMalloc(&cnt, 4096)); CHECK(hipMalloc(&err, 4));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 4; ++mode)
        for (int nwg : {cus, cus / 2, cus / 4})
            for (int barrier_only : {1, 0}) {
                const int rows_per_wg = 8, ny = nwg * rows_per_wg, rounds = 2000;
                std::vector<double> h(ny);
                for (int i = 0; i < ny; ++i) h[i] = 1.0 + i * 1e-4;
                CHECK(hipMemcpy(ya, h.data(), ny * 8, hipMemcpyHostToDevice));
                CHECK(hipMemset(yb, 0, max_ny * 8)); CHECK(hipMemset(cnt, 0, 4096)); CHECK(hipMemset(err, 0, 4));
                void (*k)(double*, double*, unsigned*, int*, int, int, int, int) =
                    mode == 0 ? rounds_kernel<0> : (mode == 1 ? rounds_kernel<1> : (mode == 2 ? rounds_kernel<2> : rounds_kernel<3>));
                CHECK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k, dim3(nwg), dim3(THREADS), 0, s, ya, yb, cnt, err, ny, rows_per_wg, rounds, barrier_only);
                CHECK(hipEventRecord(e1, s));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                int herr; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                double maxdiff = 0;
                if (!barrier_only) {   // replay on the host
                    std::vector<double> a = h, b(ny);
                    for (int r = 0; r < rounds; ++r) {
                        // same summation order is not reproduced: compare loosely
                        double tot = 0; for (int i = 0; i < ny; ++i) tot += a[i];
                        for (int i = 0; i < ny; ++i) b[i] = tot / ny * 0.5 + i * 1e-3;
                        a.swap(b);
                    }
                    std::vector<double> d(ny);
                    CHECK(hipMemcpy(d.data(), (rounds % 2) ? yb : ya, ny * 8, hipMemcpyDeviceToHost));
                    for (int i = 0; i < ny; ++i) maxdiff = fmax(maxdiff, fabs(d[i] - a[i]));
                }
                printf("mode %d  %3d workgroups  %s: %.2f us per round  (spin-limit hit: %d, max |device - host| %.2e)\n", mode,
                       nwg, barrier_only ? "barrier only      " : "exchange + barrier", ms * 1e3 / rounds, herr, maxdiff);
            }
    return 0;
}
