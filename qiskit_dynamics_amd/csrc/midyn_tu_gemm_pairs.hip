// midyn_tu_gemm_pairs.hip -- translation unit of libmidyn.so that instantiates the work-list MFMA contraction kernels (zgemm_seg_pair_kernel)
// (list: the extern-template block at the end of the kernel header; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_GEMM_PAIRS 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
