// midyn_tu_flip.hip -- translation unit of libmidyn.so that instantiates the two-workgroups-per-instance sweep kernels of flip-structured stacks (ell_flip_duo_kernel)
// (list: the extern-template block at the end of midyn_flip.h; host side: midyn.hip).
#define MIDYN_FAMILY_TU 1
#define MIDYN_TU_FLIP 1
#include <hip/hip_runtime.h>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
#include "midyn_resident.h"
#include "midyn_flip.h"
