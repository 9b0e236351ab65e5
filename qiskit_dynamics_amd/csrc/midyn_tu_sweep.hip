// midyn_tu_sweep.hip -- translation unit of libmidyn.so that instantiates the one-workgroup-per-instance sweep kernels of very sparse stacks (ell_sweep_kernel, ell_sweep_rk4_kernel, ell_sweep_split_kernel)
// (list: the extern-template block at the end of the kernel header; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_SWEEP 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
#include "midyn_resident.h"
