// midyn_tu_gemm_dense.hip -- translation unit of libmidyn.so that instantiates the dense MFMA contraction kernels (zgemm_seg_kernel<.., SPARSE = false>, splitk_reduce_kernel)
// (list: the extern-template block at the end of the kernel header; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_GEMM_DENSE 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
