// probe (round 4): do the accumulators of v_mfma_f64_16x16x4_f64 interfere less with the rest of the tile loop (VALU
// scalings, VGPR-writing LDS returns) when they live in AGPRs?  And what does ONE wave per SIMD with a 64 x 64 wave tile
// (32 accumulator quads = 256 AGPRs) sustain with the tile loop's other instructions between its MFMAs?
//   8 waves per workgroup (2 per SIMD), 16 accumulator quads per wave ("+v" or "+a" operands):
//     variant 0 bare MFMAs | 1: + 16 v_mul_f64 per 64 MFMAs | 2: + 96 v_mov_b32 per 64 | 3: + 24 ds_read_b128 per 64
//     | 4: all of 1 + 3 + s_barrier per 64 (the tile loop's mix)
//   4 waves per workgroup (1 per SIMD), 32 accumulator quads in FIXED AGPRs a[0:255]:
//     variant 10 bare | 11: + 32 v_mul_f64 per 128 MFMAs | 13: + 32 ds_read (16 b128 + 16 b64) per 128 | 14: 11 + 13 + s_barrier per 128
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/probes/map tools/mfma_acc_probe.hip && build/probes/map
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define A_ "v[100:101]"
#define B_ "v[102:103]"
#define M4(i0, i1, i2, i3)                                                                                         \
    "v_mfma_f64_16x16x4_f64 %" #i0 ", " A_ ", " B_ ", %" #i0 "\n v_mfma_f64_16x16x4_f64 %" #i1 ", " A_ ", " B_ ", %" #i1 "\n" \
    "v_mfma_f64_16x16x4_f64 %" #i2 ", " A_ ", " B_ ", %" #i2 "\n v_mfma_f64_16x16x4_f64 %" #i3 ", " A_ ", " B_ ", %" #i3 "\n"
#define MUL1 "v_mul_f64 v[110:111], v[104:105], v[106:107]\n"
#define MOV6 "v_mov_b32 v110, v104\n v_mov_b32 v111, v104\n v_mov_b32 v112, v104\n v_mov_b32 v113, v104\n v_mov_b32 v110, v105\n v_mov_b32 v111, v105\n"
#define RD1(o) "ds_read_b128 v[116:119], v108 offset:" #o "\n"
#define RD2(o) "ds_read_b128 v[120:123], v108 offset:" #o "\n"
// a quarter of a tile: 16 MFMAs with fillers F between the groups of four
#define Q(F0, F1, F2, F3) M4(0, 1, 2, 3) F0 M4(4, 5, 6, 7) F1 M4(8, 9, 10, 11) F2 M4(12, 13, 14, 15) F3

template <int V, bool AG>
__global__ __launch_bounds__(512, 2) void k8(double* sink, int iters) {
    __shared__ double lds[4096];
    const long long ck0 = clock64(), wk0 = wall_clock64();
    d4 c[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) c[i] = d4{0.0, 0.0, 0.0, 0.0};
    lds[threadIdx.x] = 0.0;
    __syncthreads();
#ifdef RANDOM_OPERANDS   // operands with random mantissas in [0.5, 1) / [-1, -0.5): the power (and so the clock) of real data
    {
        unsigned hsh = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
        unsigned lo0 = hsh * 1664525u + 1013904223u, lo1 = lo0 * 1664525u + 1013904223u;
        unsigned hi0 = 0x3FE00000u | (lo1 >> 12), hi1 = 0xBFE00000u | (lo0 >> 12);
        asm volatile("v_mov_b32 v100, %0\n v_mov_b32 v101, %1\n v_mov_b32 v102, %2\n v_mov_b32 v103, %3\n v_mov_b32 v104, %0\n v_mov_b32 v105, %1\n"
                     "v_mov_b32 v106, %2\n v_mov_b32 v107, %3\n v_lshlrev_b32 v108, 4, %4\n" ::"v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "v"(threadIdx.x & 63)
                     : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108");
    }
#else
    asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n"
                 "v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n v_lshlrev_b32 v108, 4, %0\n" ::"v"(threadIdx.x & 63)
                 : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108");
#endif
#define CL "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v110", "v111", "v112", "v113", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "memory"
#define OPSV : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]), "+v"(c[8]), "+v"(c[9]), "+v"(c[10]), "+v"(c[11]), "+v"(c[12]), "+v"(c[13]), "+v"(c[14]), "+v"(c[15]) : : CL
#define OPSA : "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3]), "+a"(c[4]), "+a"(c[5]), "+a"(c[6]), "+a"(c[7]), "+a"(c[8]), "+a"(c[9]), "+a"(c[10]), "+a"(c[11]), "+a"(c[12]), "+a"(c[13]), "+a"(c[14]), "+a"(c[15]) : : CL
#define RUN(TXT)                        \
    if (AG) asm volatile(TXT OPSA);     \
    else asm volatile(TXT OPSV);
    for (int it = 0; it < iters; ++it) {
        if (V == 0) { RUN(Q("", "", "", "") Q("", "", "", "") Q("", "", "", "") Q("", "", "", "")) }
        if (V == 1) { RUN(Q(MUL1, MUL1, MUL1, MUL1) Q(MUL1, MUL1, MUL1, MUL1) Q(MUL1, MUL1, MUL1, MUL1) Q(MUL1, MUL1, MUL1, MUL1)) }
        if (V == 2) { RUN(Q(MOV6, MOV6, MOV6, MOV6) Q(MOV6, MOV6, MOV6, MOV6) Q(MOV6, MOV6, MOV6, MOV6) Q(MOV6, MOV6, MOV6, MOV6)) }
        if (V == 3) {
            RUN(Q(RD1(0) RD2(1024), RD1(2048), RD2(3072) RD1(4096), RD2(5120)) "s_waitcnt lgkmcnt(0)\n" Q(RD1(0) RD2(1024), RD1(2048), RD2(3072) RD1(4096), RD2(5120))
                "s_waitcnt lgkmcnt(0)\n" Q(RD1(0) RD2(1024), RD1(2048), RD2(3072) RD1(4096), RD2(5120)) "s_waitcnt lgkmcnt(0)\n" Q(RD1(0) RD2(1024), RD1(2048), RD2(3072) RD1(4096), RD2(5120)) "s_waitcnt lgkmcnt(0)\n")
        }
        if (V == 4) {
            RUN(Q(RD1(0) RD2(1024), RD1(2048) MUL1, RD2(3072) RD1(4096) MUL1, RD2(5120) MUL1 MUL1) "s_waitcnt lgkmcnt(0)\n"
                Q(RD1(0) RD2(1024), RD1(2048) MUL1, RD2(3072) RD1(4096) MUL1, RD2(5120) MUL1 MUL1) "s_waitcnt lgkmcnt(0)\n"
                Q(RD1(0) RD2(1024), RD1(2048) MUL1, RD2(3072) RD1(4096) MUL1, RD2(5120) MUL1 MUL1) "s_waitcnt lgkmcnt(0)\n"
                Q(RD1(0) RD2(1024), RD1(2048) MUL1, RD2(3072) RD1(4096) MUL1, RD2(5120) MUL1 MUL1) "s_waitcnt lgkmcnt(0)\n s_barrier\n")
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += c[i][0] + c[i][3];
    if (s == 123.456) sink[0] = s + lds[5];
    if (threadIdx.x == 0 && blockIdx.x == 7) { sink[1] = (double)(clock64() - ck0); sink[2] = (double)(wall_clock64() - wk0); }
}

// one wave per SIMD: 32 accumulator quads a[8 i : 8 i + 7]
#define F4(i) "v_mfma_f64_16x16x4_f64 a[" #i ":" #i "+7], " A_ ", " B_ ", a[" #i ":" #i "+7]\n"
#define FM4(a, b, c, d) F4(a) F4(b) F4(c) F4(d)
#define RDH(o) "ds_read_b64 v[124:125], v108 offset:" #o "\n"
// a k-step of the 64 x 64 wave tile: 32 MFMAs, fillers between the groups of four
#define KS(F0, F1, F2, F3, F4_, F5, F6, F7) FM4(0, 8, 16, 24) F0 FM4(32, 40, 48, 56) F1 FM4(64, 72, 80, 88) F2 FM4(96, 104, 112, 120) F3 \
    FM4(128, 136, 144, 152) F4_ FM4(160, 168, 176, 184) F5 FM4(192, 200, 208, 216) F6 FM4(224, 232, 240, 248) F7
template <int V>
__global__ __launch_bounds__(256, 1) void k4(double* sink, int iters) {
    __shared__ double lds[4096];
    lds[threadIdx.x] = 0.0;
    __syncthreads();
#ifdef RANDOM_OPERANDS   // operands with random mantissas in [0.5, 1) / [-1, -0.5): the power (and so the clock) of real data
    {
        unsigned hsh = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
        unsigned lo0 = hsh * 1664525u + 1013904223u, lo1 = lo0 * 1664525u + 1013904223u;
        unsigned hi0 = 0x3FE00000u | (lo1 >> 12), hi1 = 0xBFE00000u | (lo0 >> 12);
        asm volatile("v_mov_b32 v100, %0\n v_mov_b32 v101, %1\n v_mov_b32 v102, %2\n v_mov_b32 v103, %3\n v_mov_b32 v104, %0\n v_mov_b32 v105, %1\n"
                     "v_mov_b32 v106, %2\n v_mov_b32 v107, %3\n v_lshlrev_b32 v108, 4, %4\n" ::"v"(lo0), "v"(hi0), "v"(lo1), "v"(hi1), "v"(threadIdx.x & 63)
                     : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108");
    }
#else
    asm volatile("v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n"
                 "v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n v_lshlrev_b32 v108, 4, %0\n" ::"v"(threadIdx.x & 63)
                 : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108");
#endif
    // zero the accumulators
    for (int z = 0; z < 1; ++z) {
#define Z8(i) "v_accvgpr_write_b32 a" #i ", 0\n"
        asm volatile(Z8(0) Z8(1) Z8(2) Z8(3) Z8(4) Z8(5) Z8(6) Z8(7) ::: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a255");
    }
    for (int it = 0; it < iters; ++it) {
        if (V == 10) asm volatile(KS("", "", "", "", "", "", "", "") KS("", "", "", "", "", "", "", "") KS("", "", "", "", "", "", "", "") KS("", "", "", "", "", "", "", "") ::: CL, "a0", "a255");
        if (V == 11) asm volatile(KS(MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1) KS(MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1)
                                  KS(MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1) KS(MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1, MUL1) ::: CL, "a0", "a255");
        if (V == 13) asm volatile(KS(RD1(0), RDH(1024), RD2(2048), RDH(3072), RD1(4096), RDH(5120), RD2(6144), RDH(7168)) "s_waitcnt lgkmcnt(0)\n"
                                  KS(RD1(0), RDH(1024), RD2(2048), RDH(3072), RD1(4096), RDH(5120), RD2(6144), RDH(7168)) "s_waitcnt lgkmcnt(0)\n"
                                  KS(RD1(0), RDH(1024), RD2(2048), RDH(3072), RD1(4096), RDH(5120), RD2(6144), RDH(7168)) "s_waitcnt lgkmcnt(0)\n"
                                  KS(RD1(0), RDH(1024), RD2(2048), RDH(3072), RD1(4096), RDH(5120), RD2(6144), RDH(7168)) "s_waitcnt lgkmcnt(0)\n" ::: CL, "v124", "v125", "a0", "a255");
        if (V == 14) asm volatile(KS(RD1(0) MUL1, RDH(1024) MUL1, RD2(2048) MUL1, RDH(3072) MUL1, RD1(4096) MUL1, RDH(5120) MUL1, RD2(6144) MUL1, RDH(7168) MUL1) "s_waitcnt lgkmcnt(0)\n"
                                  KS(RD1(0) MUL1, RDH(1024) MUL1, RD2(2048) MUL1, RDH(3072) MUL1, RD1(4096) MUL1, RDH(5120) MUL1, RD2(6144) MUL1, RDH(7168) MUL1) "s_waitcnt lgkmcnt(0)\n"
                                  KS(RD1(0) MUL1, RDH(1024) MUL1, RD2(2048) MUL1, RDH(3072) MUL1, RD1(4096) MUL1, RDH(5120) MUL1, RD2(6144) MUL1, RDH(7168) MUL1) "s_waitcnt lgkmcnt(0)\n"
                                  KS(RD1(0) MUL1, RDH(1024) MUL1, RD2(2048) MUL1, RDH(3072) MUL1, RD1(4096) MUL1, RDH(5120) MUL1, RD2(6144) MUL1, RDH(7168) MUL1) "s_waitcnt lgkmcnt(0)\n s_barrier\n" ::: CL, "v124", "v125", "a0", "a255");
    }
    float s = 0;
    asm volatile("s_nop 15\n s_nop 7\n v_accvgpr_read_b32 v110, a3\n v_accvgpr_read_b32 v111, a250\n v_add_f32 %0, v110, v111\n" : "=v"(s)::"v110", "v111");
    if (s == 123.456f) sink[0] = s + lds[5];
}

template <int V, bool AG> static void run8(double* sink) {
    const int iters = 2048, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k8<V, AG>), dim3(blocks), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k8<V, AG>), dim3(blocks), dim3(512), 0, 0, sink, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 8 * iters * 64 * 2048.0;
    double h[3]; hipMemcpy(h, sink, 24, hipMemcpyDeviceToHost);
    printf("8 waves, acc in %s, variant %d: %.3f ms, %.2f TFLOP/s = %.4f of 78.6  shader clock %.3f GHz (%s)\n", AG ? "AGPR" : "VGPR", V, ms, flops / ms / 1e9,
           flops / ms / 1e9 / 78.6, h[1] / h[2] * 0.1, hipGetErrorString(hipGetLastError()));
}
template <int V> static void run4(double* sink) {
    const int iters = 2048, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k4<V>), dim3(blocks), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k4<V>), dim3(blocks), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 128 * 2048.0;
    printf("4 waves, 32 quads in AGPR, variant %d: %.3f ms, %.2f TFLOP/s = %.4f of 78.6  (%s)\n", V, ms, flops / ms / 1e9, flops / ms / 1e9 / 78.6,
           hipGetErrorString(hipGetLastError()));
}
int main() {
    double* sink; hipMalloc(&sink, 64);
    run8<0, false>(sink); run8<0, true>(sink); run8<1, false>(sink); run8<1, true>(sink); run8<2, false>(sink); run8<2, true>(sink);
    run8<3, false>(sink); run8<3, true>(sink); run8<4, false>(sink); run8<4, true>(sink);
    run4<10>(sink); run4<11>(sink); run4<13>(sink); run4<14>(sink);
    return 0;
}
