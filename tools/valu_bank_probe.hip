// probe (round 4): does the VGPR bank pattern of a v_fma_f64's operands change its rate?  (the apply phase of
// rhs_combine_kernel is 64 v_fma_f64 per kk step; measured rate of the instruction ~5.4 cycles per wave64 against a nominal 4)
//   variant 0: d = a*b + d with a, b, d in the same bank pair   1: all different bank pairs where possible
//   variant 2: one operand an SGPR pair                         3: v_fmac (VOP2) form
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/probes/vbp tools/valu_bank_probe.hip && build/probes/vbp
#include <hip/hip_runtime.h>
#include <cstdio>
#define R8(X) X X X X X X X X
template <int V>
__global__ __launch_bounds__(512, 2) void k(double* sink, int iters) {
    asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n"
                 "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n v_mov_b32 v128, 0\n v_mov_b32 v129, 0\n v_mov_b32 v132, 0\n v_mov_b32 v133, 0\n"
                 "v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v126, 0\n v_mov_b32 v127, 0\n v_mov_b32 v130, 0\n v_mov_b32 v131, 0\n v_mov_b32 v134, 0\n v_mov_b32 v135, 0\n"
                 "s_mov_b32 s20, 0\n s_mov_b32 s21, 0\n" ::: "v100", "v101", "v104", "v105", "v106", "v107", "v110", "v111", "v120", "v121", "v122", "v123", "v124", "v125",
                 "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "s20", "s21");
    for (int it = 0; it < iters; ++it) {
        // 8 independent accumulators per group, 8 groups = 64 instructions
        if (V == 0) asm volatile(R8("v_fma_f64 v[120:121], v[100:101], v[104:105], v[120:121]\n v_fma_f64 v[124:125], v[100:101], v[104:105], v[124:125]\n"
                                    "v_fma_f64 v[128:129], v[100:101], v[104:105], v[128:129]\n v_fma_f64 v[132:133], v[100:101], v[104:105], v[132:133]\n"
                                    "v_fma_f64 v[120:121], v[100:101], v[104:105], v[120:121]\n v_fma_f64 v[124:125], v[100:101], v[104:105], v[124:125]\n"
                                    "v_fma_f64 v[128:129], v[100:101], v[104:105], v[128:129]\n v_fma_f64 v[132:133], v[100:101], v[104:105], v[132:133]\n") ::: "v120", "v121", "v124", "v125", "v128", "v129", "v132", "v133");
        if (V == 1) asm volatile(R8("v_fma_f64 v[120:121], v[106:107], v[110:111], v[120:121]\n v_fma_f64 v[124:125], v[106:107], v[110:111], v[124:125]\n"
                                    "v_fma_f64 v[128:129], v[106:107], v[110:111], v[128:129]\n v_fma_f64 v[132:133], v[106:107], v[110:111], v[132:133]\n"
                                    "v_fma_f64 v[120:121], v[106:107], v[110:111], v[120:121]\n v_fma_f64 v[124:125], v[106:107], v[110:111], v[124:125]\n"
                                    "v_fma_f64 v[128:129], v[106:107], v[110:111], v[128:129]\n v_fma_f64 v[132:133], v[106:107], v[110:111], v[132:133]\n") ::: "v120", "v121", "v124", "v125", "v128", "v129", "v132", "v133");
        if (V == 2) asm volatile(R8("v_fma_f64 v[120:121], s[20:21], v[106:107], v[120:121]\n v_fma_f64 v[124:125], s[20:21], v[106:107], v[124:125]\n"
                                    "v_fma_f64 v[128:129], s[20:21], v[106:107], v[128:129]\n v_fma_f64 v[132:133], s[20:21], v[106:107], v[132:133]\n"
                                    "v_fma_f64 v[120:121], s[20:21], v[106:107], v[120:121]\n v_fma_f64 v[124:125], s[20:21], v[106:107], v[124:125]\n"
                                    "v_fma_f64 v[128:129], s[20:21], v[106:107], v[128:129]\n v_fma_f64 v[132:133], s[20:21], v[106:107], v[132:133]\n") ::: "v120", "v121", "v124", "v125", "v128", "v129", "v132", "v133");
        if (V == 3) asm volatile(R8("v_fmac_f64_e32 v[120:121], v[106:107], v[110:111]\n v_fmac_f64_e32 v[124:125], v[106:107], v[110:111]\n"
                                    "v_fmac_f64_e32 v[128:129], v[106:107], v[110:111]\n v_fmac_f64_e32 v[132:133], v[106:107], v[110:111]\n"
                                    "v_fmac_f64_e32 v[122:123], v[106:107], v[110:111]\n v_fmac_f64_e32 v[126:127], v[106:107], v[110:111]\n"
                                    "v_fmac_f64_e32 v[130:131], v[106:107], v[110:111]\n v_fmac_f64_e32 v[134:135], v[106:107], v[110:111]\n") ::: "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
        if (V == 4) asm volatile(R8("v_mul_f64 v[120:121], v[106:107], v[110:111]\n v_mul_f64 v[124:125], v[106:107], v[110:111]\n"
                                    "v_mul_f64 v[128:129], v[106:107], v[110:111]\n v_mul_f64 v[132:133], v[106:107], v[110:111]\n"
                                    "v_mul_f64 v[122:123], v[106:107], v[110:111]\n v_mul_f64 v[126:127], v[106:107], v[110:111]\n"
                                    "v_mul_f64 v[130:131], v[106:107], v[110:111]\n v_mul_f64 v[134:135], v[106:107], v[110:111]\n") ::: "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135");
    }
    if (threadIdx.x == 100000) sink[0] = 1.0;
}
template <int V> static void run(double* sink) {
    const int iters = 20000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<V>, dim3(blocks), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 2 waves x iters x 64 instructions
    const double instr = 2.0 * iters * 64;
    printf("variant %d: %.3f ms -> %.2f cycles per wave64 instruction at 2.4 GHz, %.2f TFLOP/s\n", V, ms, ms * 1e-3 * 2.4e9 / instr,
           (double)blocks * 8 * iters * 64 * 128.0 / ms / 1e9);
}
int main() {
    double* sink; hipMalloc(&sink, 64);
    run<0>(sink); run<1>(sink); run<2>(sink); run<3>(sink); run<4>(sink);
    return 0;
}
