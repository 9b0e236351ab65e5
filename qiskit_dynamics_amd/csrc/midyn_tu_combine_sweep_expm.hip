// midyn_tu_combine_sweep_expm.hip -- translation unit of libmidyn.so that instantiates the one-launch sweep kernels of small systems, expm action (combine_sweep_kernel<.., MODE 1>)
// (list: the extern-template block at the end of the kernel header; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_COMBINE_SWEEP_EXPM 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
#include "midyn_combine.h"
#include "midyn_combine_sweep.h"
