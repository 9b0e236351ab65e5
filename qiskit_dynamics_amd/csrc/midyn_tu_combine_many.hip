// midyn_tu_combine_many.hip -- translation unit of libmidyn.so that instantiates the COMBINE + APPLY kernels (rhs_combine_kernel with a third plane group of a kind beside planes of the other: 9 - 12 operators per kind)
// (list: the extern-template block at the end of the kernel header; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_COMBINE_MANY 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
#include "midyn_combine.h"
