// midyn_tu_resident.hip -- translation unit of libmidyn.so that instantiates the register-resident single-trajectory kernels (rk4_resident_kernel, ell_resident_kernel)
// (list: the extern-template block at the end of the kernel header; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_RESIDENT 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
#include "midyn_resident.h"
