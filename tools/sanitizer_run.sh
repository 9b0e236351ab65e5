#!/bin/bash
# Host side of libmidyn.so under AddressSanitizer + UndefinedBehaviorSanitizer (device code is not instrumented).
# Build (in the CPU container, ~2 min; one object per translation unit: midyn.hip + the kernel-family units midyn_tu_*.hip):
#   mkdir -p build/asan && cd build/asan
#   for u in ../../qiskit_dynamics_amd/csrc/midyn.hip ../../qiskit_dynamics_amd/csrc/midyn_tu_*.hip; do
#     hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -Wno-unused-value -Wno-unused-function -fsanitize=address,undefined \
#           -fno-sanitize=function -fno-gpu-sanitize -fno-omit-frame-pointer -c -o $(basename $u .hip)_asan.o $u &
#   done; wait
#   (-fno-sanitize=function: UBSan's function-type check puts a signature word in front of every function, and the HIP
#    runtime then no longer finds the kernels behind their host stubs -- every templated kernel launch silently does
#    nothing; found with tools/gemm_probe.hip: -O1, -O1 -g, -O1 + ASan are correct, + UBSan is not, + UBSan without
#    `function` is.  The C driver below is compiled WITH the check: it verifies that the C view of every entry point it
#    calls has the function type of the C++ definition -- midyn_complex is `double _Complex` in both for that reason)
#   hipcc -shared -fsanitize=address,undefined -o libmidyn_asan.so midyn*_asan.o -ldl && rm midyn*_asan.o
#   /opt/rocm/lib/llvm/bin/clang -std=c99 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -o abi_solve_asan \
#         ../../tests/abi_solve.c -ldl -lm -lstdc++   (libstdc++: ASan's __cxa_throw interceptor needs it when RCCL throws inside)
# Run (on the GPU box, through gpurun):  bash tools/sanitizer_run.sh  -> gpurun_out/sanitizer/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/sanitizer
mkdir -p $O
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
HIPRT=/opt/rocm/lib/libamdhip64.so
RCCL=/opt/rocm/lib/librccl.so.1
# 1. the C99 host program: context, stack, in-place + out-of-place RCCL broadcast, RHS, RK4, Magnus-2 through the C-ABI
$R/build/asan/abi_solve_asan $HIPRT $R/build/asan/libmidyn_asan.so $RCCL > $O/abi_solve.log 2>&1
echo "abi_solve_asan exit $?" >> $O/abi_solve.log
tail -3 $O/abi_solve.log
# 2. the Python binding's GPU tests on the sanitized library (ASan runtime preloaded into the interpreter)
cp $R/qiskit_dynamics_amd/libmidyn.so /tmp/libmidyn_plain.so
cp $R/build/asan/libmidyn_asan.so $R/qiskit_dynamics_amd/libmidyn.so
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
( cd $R && LD_PRELOAD=$RT MIDYN_HIP_RUNTIME=system timeout 1500 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_combine.py tests/test_gpu_resident.py \
    -m gpu -q -x -p no:cacheprovider > $O/pytest_asan.log 2>&1 ; echo "pytest exit $?" >> $O/pytest_asan.log )
# (round 6: the expansion entry points -- pinned staging block, kept offset tables, coefficient path -- and the expm plan object)
( cd $R && LD_PRELOAD=$RT MIDYN_HIP_RUNTIME=system timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider \
    -k "expansion or perturbative or dyson or magnus_solver" > $O/pytest_asan_expansion.log 2>&1 ; echo "pytest exit $?" >> $O/pytest_asan_expansion.log )
cp /tmp/libmidyn_plain.so $R/qiskit_dynamics_amd/libmidyn.so
tail -5 $O/pytest_asan.log
tail -5 $O/pytest_asan_expansion.log
grep -c "ERROR: AddressSanitizer\|runtime error:" $O/abi_solve.log $O/pytest_asan.log $O/pytest_asan_expansion.log
