/* C99 consumer of the C-ABI: proves include/midyn.h is plain C (no C++/torch types) and that every
 * declared entry point can be resolved from libmidyn.so with dlsym.  No GPU call is made. */
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include "../include/midyn.h"

#define CHECK(sym)                                          \
    do {                                                    \
        if (!dlsym(h, #sym)) {                              \
            fprintf(stderr, "missing symbol %s\n", #sym);   \
            return 2;                                       \
        }                                                   \
        /* the address-of below type-checks the prototype from the header */ \
        (void)&sym;                                         \
        ++n;                                                \
    } while (0)

int main(int argc, char** argv) {
    if (argc < 3) return 64;
    /* one HIP runtime must be global before libmidyn is opened */
    if (!dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL)) {
        fprintf(stderr, "hip runtime: %s\n", dlerror());
        return 3;
    }
    void* h = dlopen(argv[2], RTLD_NOW);
    if (!h) {
        fprintf(stderr, "libmidyn: %s\n", dlerror());
        return 4;
    }
    int n = 0;
    midyn_complex z = 1.0 + 2.0 * I; /* C99 complex is the ABI element type */
    (void)z;
    CHECK(midyn_ctx_create); CHECK(midyn_ctx_destroy); CHECK(midyn_ctx_synchronize); CHECK(midyn_last_error);
    CHECK(midyn_ctx_set_option); CHECK(midyn_ctx_get_option); CHECK(midyn_stack_packed_bytes); CHECK(midyn_stack_create);
    CHECK(midyn_stack_adopt); CHECK(midyn_stack_destroy); CHECK(midyn_stack_info);
    CHECK(midyn_stack_segment_modes); CHECK(midyn_eval_generator); CHECK(midyn_eval_rhs);
    CHECK(midyn_rk4_solve); CHECK(midyn_expm); CHECK(midyn_expm_solve); CHECK(midyn_zgemm);
    CHECK(midyn_rk4_plan_create); CHECK(midyn_rk4_plan_run); CHECK(midyn_rk4_plan_fetch);
    CHECK(midyn_rk4_plan_destroy); CHECK(midyn_expm_plan_create); CHECK(midyn_expm_plan_run); CHECK(midyn_expm_plan_fetch);
    CHECK(midyn_expm_plan_destroy); CHECK(midyn_get_counters); CHECK(midyn_reset_counters);
    CHECK(midyn_microbench); CHECK(midyn_lindblad_create); CHECK(midyn_lindblad_destroy);
    CHECK(midyn_lindblad_rhs); CHECK(midyn_lindblad_rk4_solve);
    CHECK(midyn_stack_create_lindblad); CHECK(midyn_stack_antiherm_defect); CHECK(midyn_sigtable_create);
    CHECK(midyn_sigtable_data); CHECK(midyn_sigtable_fetch); CHECK(midyn_sigtable_destroy);
    CHECK(midyn_parallel_solve); CHECK(midyn_expansion_create); CHECK(midyn_expansion_destroy);
    CHECK(midyn_expansion_solve); CHECK(midyn_expansion_set_monomials); CHECK(midyn_expansion_solve_coeffs); CHECK(midyn_host_alloc); CHECK(midyn_host_free);
    CHECK(midyn_ctx_timer); CHECK(midyn_stack_block_info);
    /* multi-GPU: the RCCL broadcast of the stack (librccl itself is only resolved at first use) */
    CHECK(midyn_comm_get_unique_id); CHECK(midyn_comm_init_rank); CHECK(midyn_comm_destroy); CHECK(midyn_comm_count);
    CHECK(midyn_stack_create_empty); CHECK(midyn_stack_broadcast); CHECK(midyn_stack_broadcast_from);
    /* a host-only entry point can be exercised without a GPU */
    int (*packed)(int, int, int, size_t*) = (int (*)(int, int, int, size_t*))dlsym(h, "midyn_stack_packed_bytes");
    size_t bytes = 0;
    if (packed(1024, 8, 1, &bytes) != 0 || bytes < (size_t)9 * 1024 * 1024 * 16) return 5;
    printf("ABI_OK %d symbols, packed stack %zu bytes\n", n, bytes);
    return 0;
}
