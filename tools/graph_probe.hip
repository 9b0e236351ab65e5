// probe: launch cost of a chain of tiny kernels -- stream launches vs a captured hipGraph replay
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(double* x, const int* cursor, int work) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v = x[i];
    for (int k = 0; k < work; ++k) v = v * 1.0000001 + (double)cursor[0] * 1e-9;
    x[i] = v;
}
__global__ void bump(int* cursor) { cursor[0] += 1; }
int main() {
    double* x; int* cur;
    hipMalloc(&x, 256 * 256 * sizeof(double)); hipMemset(x, 0, 256 * 256 * sizeof(double));
    hipMalloc(&cur, 4); hipMemset(cur, 0, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int work : {1, 2000}) {
        for (int chain : {4, 16, 64}) {
            const int reps = 2000 / chain;
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, s, x, cur, work); hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, s, cur); }
            hipStreamSynchronize(s);
            double us_stream = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * (chain + 1));
            hipGraph_t g; hipGraphExec_t ge;
            hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
            for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, s, x, cur, work);
            hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, s, cur);
            hipStreamEndCapture(s, &g);
            auto ti = std::chrono::steady_clock::now();
            hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            double us_inst = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - ti).count();
            hipGraphLaunch(ge, s); hipStreamSynchronize(s);
            t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) hipGraphLaunch(ge, s);
            hipStreamSynchronize(s);
            double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (reps * (chain + 1));
            printf("work %d chain %d: stream %.2f us/kernel, graph replay %.2f us/kernel, instantiate %.0f us\n", work, chain, us_stream, us_graph, us_inst);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    }
    return 0;
}
