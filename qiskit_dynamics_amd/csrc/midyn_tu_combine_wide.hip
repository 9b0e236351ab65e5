// midyn_tu_combine_wide.hip -- translation unit of libmidyn.so that instantiates the COMBINE + APPLY kernels (rhs_combine_kernel with three or four plane groups of one kind, rhs_combine_small_kernel)
// (list: the extern-template block at the end of the kernel header; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_COMBINE_WIDE 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
#include "midyn_combine.h"
