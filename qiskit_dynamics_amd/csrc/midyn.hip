// midyn.hip -- host side of libmidyn.so: the C-ABI of include/midyn.h on top of the gfx950 kernels
// in midyn_kernels.h.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC midyn.hip
//
// No CPU fallback exists in this library: every entry point needs a HIP device and fails with a
// non-zero status (text via midyn_last_error) when there is none.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the entry points are resolved with dlsym (midyn_comm.inc)

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "../../include/midyn.h"
#include "midyn_kernels.h"
#include "midyn_resident.h"
#include "midyn_flip.h"
#include "midyn_combine.h"
#include "midyn_combine_sweep.h"

using namespace midyn;

// The implementation is one translation unit, split by concern (included in this order):
#include "midyn_core.inc"
#include "midyn_launch.inc"
#include "midyn_combine.inc"
#include "midyn_eval.inc"
#include "midyn_rk4.inc"
#include "midyn_expm.inc"
#include "midyn_sweep_plan.inc"
#include "midyn_action.inc"
#include "midyn_parallel.inc"
#include "midyn_expansion.inc"
#include "midyn_lindblad.inc"
#include "midyn_microbench.inc"
#include "midyn_comm.inc"
