// probe: does the VGPR bank of the A / B operands of v_mfma_f64_16x16x4_f64 matter?  16 accumulators per wave (as in the
// contraction kernels), 8 waves per workgroup, one workgroup per CU; the A and B operands sit in fixed registers:
//   variant 0: A v[194:195], B v[242:243]  (same bank pair)      variant 1: A v[194:195], B v[240:241]  (different pairs)
//   variant 2: like 0 with 4 v_mul_f64 per 16 MFMAs               variant 3: like 1 with 4 v_mul_f64 per 16 MFMAs
//   variant 4: 64 MFMAs per wave then s_barrier; 5: the same without barrier; 6: 128 MFMAs per barrier
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/probes/mb tools/mfma_bank_probe.hip && build/probes/mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define M16(A_, B_)                                                                                                        \
    "v_mfma_f64_16x16x4_f64 %0, " A_ ", " B_ ", %0\n v_mfma_f64_16x16x4_f64 %1, " A_ ", " B_ ", %1\n"                        \
    "v_mfma_f64_16x16x4_f64 %2, " A_ ", " B_ ", %2\n v_mfma_f64_16x16x4_f64 %3, " A_ ", " B_ ", %3\n"                        \
    "v_mfma_f64_16x16x4_f64 %4, " A_ ", " B_ ", %4\n v_mfma_f64_16x16x4_f64 %5, " A_ ", " B_ ", %5\n"                        \
    "v_mfma_f64_16x16x4_f64 %6, " A_ ", " B_ ", %6\n v_mfma_f64_16x16x4_f64 %7, " A_ ", " B_ ", %7\n"                        \
    "v_mfma_f64_16x16x4_f64 %8, " A_ ", " B_ ", %8\n v_mfma_f64_16x16x4_f64 %9, " A_ ", " B_ ", %9\n"                        \
    "v_mfma_f64_16x16x4_f64 %10, " A_ ", " B_ ", %10\n v_mfma_f64_16x16x4_f64 %11, " A_ ", " B_ ", %11\n"                    \
    "v_mfma_f64_16x16x4_f64 %12, " A_ ", " B_ ", %12\n v_mfma_f64_16x16x4_f64 %13, " A_ ", " B_ ", %13\n"                    \
    "v_mfma_f64_16x16x4_f64 %14, " A_ ", " B_ ", %14\n v_mfma_f64_16x16x4_f64 %15, " A_ ", " B_ ", %15\n"
#define MUL4 "v_mul_f64 v[250:251], v[244:245], v[246:247]\n v_mul_f64 v[252:253], v[244:245], v[246:247]\n" \
             "v_mul_f64 v[250:251], v[244:245], v[246:247]\n v_mul_f64 v[252:253], v[244:245], v[246:247]\n"
template <int V>
__global__ __launch_bounds__(512, 2) void k(double* sink, int iters) {
    d4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = d4{0.0, 0.0, 0.0, 0.0};
    asm volatile("v_mov_b32 v194, 0\n v_mov_b32 v195, 0\n v_mov_b32 v240, 0\n v_mov_b32 v241, 0\n v_mov_b32 v242, 0\n v_mov_b32 v243, 0\n"
                 "v_mov_b32 v244, 0\n v_mov_b32 v245, 0\n v_mov_b32 v246, 0\n v_mov_b32 v247, 0\n" ::: "v194", "v195", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247");
    for (int it = 0; it < iters; ++it) {
#define OPS : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), \
              "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]) : : "v194", "v195", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v250", "v251", "v252", "v253"
        if (V == 0) asm volatile(M16("v[194:195]", "v[242:243]") OPS);
        if (V == 1) asm volatile(M16("v[194:195]", "v[240:241]") OPS);
        if (V == 2) asm volatile(MUL4 M16("v[194:195]", "v[242:243]") OPS);
        if (V == 3) asm volatile(MUL4 M16("v[194:195]", "v[240:241]") OPS);
        if (V == 4) {   // 64 MFMAs per wave, then a workgroup barrier (the tile loop's cadence)
            asm volatile(M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") "s_barrier\n" OPS);
        }
        if (V == 5) {   // the same without the barrier
            asm volatile(M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") OPS);
        }
        if (V >= 7 && V <= 12) {   // 4 x (16 MFMAs + 24 cheap VALU instructions in 4 clumps), then a barrier (7, 9, 11) or none (8, 10, 12)
#define GAP6 "v_mov_b32 v250, v244\n v_mov_b32 v251, v244\n v_mov_b32 v252, v244\n v_mov_b32 v253, v244\n v_mov_b32 v250, v245\n v_mov_b32 v251, v245\n"
#define Q16(P_) P_ GAP6 M16("v[194:195]", "v[240:241]") GAP6 GAP6 GAP6
            if (V == 7) asm volatile(Q16("") Q16("") Q16("") Q16("") "s_barrier\n" OPS);
            if (V == 8) asm volatile(Q16("") Q16("") Q16("") Q16("") OPS);
            if (V == 9) asm volatile(Q16("s_setprio 3\n") Q16("s_setprio 2\n") Q16("s_setprio 1\n") Q16("s_setprio 0\n") "s_barrier\n" OPS);
            if (V == 10) asm volatile(Q16("s_setprio 3\n") Q16("s_setprio 2\n") Q16("s_setprio 1\n") Q16("s_setprio 0\n") OPS);
        }
        if (V == 6) {   // 128 MFMAs per barrier
            asm volatile(M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") OPS);
            asm volatile(M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") M16("v[194:195]", "v[240:241]") "s_barrier\n" OPS);
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][3];
    if (s == 123.456) sink[0] = s;
}
template <int V> static void run(double* sink) {
    const int iters = 4096, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * iters * 16 * 2048.0 * (V == 6 ? 8 : V >= 4 ? 4 : 1);
    printf("variant %d: %.3f ms, %.2f TFLOP/s = %.4f of 78.6\n", V, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6);
}
int main() {
    double* sink; hipMalloc(&sink, 8);
    run<0>(sink); run<1>(sink); run<2>(sink); run<3>(sink); run<4>(sink); run<5>(sink); run<6>(sink); run<7>(sink); run<8>(sink); run<9>(sink); run<10>(sink);
    return 0;
}
