/*
 * midyn.h -- C-ABI of libmidyn.so, the MI355X (gfx950) implementation of the qiskit-dynamics
 * ODE-RHS hot path.
 *
 * The reference (qiskit-community/qiskit-dynamics, pure Python) has NO FFI for this path; its
 * seams are:
 *   - arraylias function registration          qiskit_dynamics/arraylias/alias.py:44-128
 *   - `array_library=` -> collection factory   qiskit_dynamics/models/generator_model.py:368-397
 *                                              qiskit_dynamics/models/lindblad_model.py:541-597
 *   - `method=` -> solver function             qiskit_dynamics/solvers/solver_functions.py:53-65,198-207,349-364
 *   - `Solver.solve` list mode                 qiskit_dynamics/solvers/solver_classes.py:384-676
 * Each entry point below names the reference function whose arithmetic it replaces.  The binding a
 * qiskit-dynamics maintainer would add is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - All arrays are caller-owned HOST buffers unless a parameter is called `dev_*`.
 *   - Complex numbers are C99 `double _Complex` (interleaved re,im), matrices C-order (row major).
 *   - Operator stack `ops` is [k][n][n]; static operator [n][n] or NULL; the rotating-frame
 *     diagonal d (purely imaginary) is passed as its imaginary part `frame_im[n]` or NULL.
 *   - A state is [n][m] with the m states as COLUMNS (generator_model tests :615-643); a batch of
 *     B instances is [B][n][m].  Vectorised density matrices are column stacked (caller's job).
 *   - Every function returns 0 on success, non-zero on failure; `midyn_last_error` gives the text.
 *   - A ctx owns one HIP stream and is single threaded.  One ctx per device / per process rank.
 *   - Handles are opaque; the library never keeps a host pointer after a call returns.  Objects
 *     created from a ctx (stacks, plans, tables, expansions, ...) must be destroyed BEFORE that ctx, and
 *     a plan / Lindblad handle before the stacks it was created from.
 */
#ifndef MIDYN_H
#define MIDYN_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#if defined(__clang__) || defined(__GNUC__)
typedef double _Complex midyn_complex;              /* the C type itself (GNU extension in C++): one function type for both views */
#else
typedef struct { double re, im; } midyn_complex;    /* layout-compatible stand-in */
#endif
#else
#include <complex.h>
typedef double _Complex midyn_complex;
#endif

typedef struct midyn_ctx midyn_ctx;
typedef struct midyn_stack midyn_stack;

/* ---- context ------------------------------------------------------------------------------ */
int midyn_ctx_create(int device, midyn_ctx** out);
int midyn_ctx_destroy(midyn_ctx* ctx);
int midyn_ctx_synchronize(midyn_ctx* ctx);
/* Text of the last error on this ctx (or of the last ctx-less failure when ctx == NULL). */
const char* midyn_last_error(midyn_ctx* ctx);
/* Options (defaults in brackets; all of them select between paths that give the same results):
 *   skip_zero_planes [1]  skip exact-zero real/imaginary planes of operators (MFMA contraction and, for
 *                         single-plane stacks, the planar streaming kernel `stream_planes` [1])
 *   skip_zero_blocks [1]  stacks whose 16 x 16 operator blocks are mostly exactly zero (operators in a
 *                         computational / diagonal-frame basis): contract only the occupied blocks
 *                         (work-list kernels; the skipped products are exact zeros)
 *   chebyshev [1]         expm action, Magnus order 1, nearly skew-Hermitian generator: Chebyshev series instead of
 *                         the scaled Taylor series when shorter (2: always, 0: never)
 *   expansion_pack [1]    midyn_expansion_solve with n <= 32: two step matrices share one padded 64 x 64 block (block diagonal), so the
 *                         batched expm / tree run on half as many padded matrices; 0: one step per block
 *   exchange_protocol [0] hand-offs between the workgroups of a one-launch kernel: 0 = the measured default, 1 = the conforming forms of
 *                         MI355X_MICROARCH.md (release / acquire, sc1 stores and loads everywhere); same results bit for bit
 *   expm_plan_cache [1]   one-shot midyn_expm_solve on the one-launch sweep route keeps its plan (frame phases, step tables, y0, exchange
 *                         slots, result block) in the stack: the next call with the same shapes, time grid, step tables and initial
 *                         states (compared byte for byte; per-instance y0 up to 1 MB) only uploads its coefficient table -- a scan
 *                         through the one-shot entry point costs what midyn_expm_plan_run + _fetch cost.  Any midyn_ctx_set_option
 *                         call, a different solve, a received stack or midyn_stack_destroy retires the plan.  0: made per call
 *   expm_direct_out [1]   midyn_expm_solve / midyn_expm_plan_run on the one-launch sweep route: saved states written by the kernel
 *                         straight into a device-writable (pinned) result block of the caller; 0: device block + copy
 *   cheb_tail [1]         where that Chebyshev series ends: 1 = where the dropped terms of a step sum to less than 2^-53 (the
 *                         unit-roundoff backward error the Taylor schemes and scipy.linalg.expm are built for), 0 = every
 *                         Bessel coefficient >= 1e-18 is kept (rounds 2-5: one or two terms more per step)
 *   sparse_bm [0]         row-panel height of the sparse MFMA route: 0 by list density, or 16 | 32 | 64 | 128
 *   krylov [1]            one column, Magnus order 1: Arnoldi instead of the scaled Taylor series when the
 *                         series is long enough to pay for it (2: always, 0: never)
 *   complex_3m [1]        dense complex products with 3 real MFMAs instead of 4 (normwise error bound): 0 never,
 *                         1 in solver loops / expm only (midyn_eval_rhs, midyn_zgemm stay 4M), 2 everywhere
 *   split_k [1], force_splits [0], force_tile [0 | 64 | 128 | 12864]   tile / split-K choice of the zgemm
 *   splitk_inlaunch [0]   sum the split-K partials inside the contraction launch (measured slower, opt-in)
 *   pair_launch [1]       sparse MFMA route: the two independent products of a Magnus-2 level share one launch
 *   combine [1]           sweeps (B > 1 instances with their own coefficients, >= combine_min_cols state columns, at most 8
 *                         operators with a real plane and 8 with an imaginary one): the contraction runs as COMBINE + APPLY
 *                         (csrc/midyn_combine.h: sum_j c_j[b] G_j per instance by fp64 MFMAs over the operator planes, then
 *                         the product with the state by vector FMAs -- the reference's order of operations,
 *                         models/operator_collections.py:101-134) instead of k + 1 GEMMs, where that is the faster
 *                         formulation (the planes of a kind fill groups of four MFMA slots: at least three quarters of
 *                         the slots must hold a plane); 2: wherever the kernels apply; 0: the MFMA GEMM routes
 *   combine_min_cols [256]   ... smallest padded column count of the state block that takes it
 *   combine_occupancy [2]  waves per SIMD the workgroup shape of a SMALL sweep aims at: up to 8 waves split the list of one (row
 *                         group, instance block) pair and add their accumulators through LDS as a tree; 1: one wave per SIMD,
 *                         at most 4 waves per pair (the rule until round 4; A/B)
 *   combine_sweep [1]     fixed-step RK4 sweeps (midyn_rk4_solve) and scipy_expm sweeps with magnus_order 1 (midyn_expm_solve;
 *                         B > 1 instances, one state column each) of stacks with that
 *                         layout and n_pad <= 256: the WHOLE solve in one launch, 16 instances per workgroup, state in
 *                         registers, stage input in LDS (csrc/midyn_combine_sweep.h); 1: always up to n_pad = 128, above only
 *                         when the workgroups fill the chip, 2: wherever it applies, 0: never (a launch per RK4 stage / series term)
 *   combine_first [1]     one instance, >= 8 columns: form C(t) once, then one n x n x m product per stage
 *   multi_stream [1]      2..8 state columns at n >= 256: multi-column streaming kernel
 *   tiny_rk4 [1]          small systems: whole fixed-step solve in one persistent launch
 *   resident_rk4 [1]      one RK4 trajectory whose active operator planes fit the register files (32 < n, at most 64
 *                         doubles per lane and row): whole step ranges in one launch, operators in registers, the
 *                         stage input exchanged through a polled ring in device memory (csrc/midyn_resident.h)
 *   ell_sweep [1]         sweeps (and single Magnus-2 trajectories) on very sparse stacks, 256 <= n_pad <= 4096, in
 *                         midyn_rk4_solve and the expm action of midyn_expm_solve: ONE launch, one workgroup per
 *                         instance through all steps, staged vectors in LDS, operator elements (ELL) from L2
 *                         (csrc/midyn_resident.h: ell_sweep_kernel, ell_sweep_rk4_kernel)
 *   ell_sweep_packed [1]  ... with the packed element form of the stack when it has one: 4-byte elements (column | sign,
 *                         or the LDS address of the operand) when every ELL slot holds one magnitude (operators built
 *                         from Pauli strings); 0: always the general 12-byte form.  Same results to rounding
 *   ell_sweep_split [1]   ... small shards of stacks WITHOUT a packed form: 4 workgroups per instance while 4 x instances
 *                         <= CUs (n_pad = 4096); the partners all-gather every operand vector through a sentinel-polled
 *                         ring.  2: also 2 workgroups per instance; 3: packed stacks too (measured slower than the
 *                         packed one-workgroup kernel at every shard size); 0: never
 *   ell_sweep_duo [1]     ... small shards (2 x instances <= CUs) of stacks WITH a packed form, n_pad a power of two in
 *                         [512, 4096]: TWO workgroups per instance, half of the rows each; the slots that stay inside a
 *                         half run while the partner's half arrives through the L2 the two share (ell_sweep_duo_kernel); 0: never
 *   ell_sweep_flip [1]    ... and when every slot of the stack also has ONE flip mask (column = row ^ flip in every row:
 *                         sums of Pauli strings without Z factors), the two-workgroup kernel that reads no operator
 *                         elements at all (csrc/midyn_flip.h: ell_flip_duo_kernel); 0: ell_sweep_duo_kernel
 *   resident_spin_limit [2^21]   polls a wait inside a one-launch kernel (rk4_resident, ell_resident, ell_sweep_split)
 *                         may take before it gives up; the solve then re-runs the step range on the launch-per-product
 *                         route by itself (counter "resident_fallbacks").  -1 restores the default; 0 = give up at the
 *                         first missing word (tests)
 *   resident_exchange_only [0]   measurement only: rk4_resident_kernel publishes and polls every round but skips the
 *                         row product (results wrong) -- the store -> poll floor bench.py reports; refused (error)
 *                         unless the process environment holds MIDYN_DEBUG_OPTIONS=1
 *   expm_action [1]       few columns, Magnus order <= 2: expm(Omega) y by matrix-vector products
 *   expm_degree [0]       0: Taylor degree of the dense expm chosen from the norm; else 2|4|6|9|12|16
 *   profile [0]           record HIP-event kernel times (midyn_get_counters)
 *   stream_variant, plane_kernel, prefer_duo, ablate          A/B and profiling switches */
int midyn_ctx_set_option(midyn_ctx* ctx, const char* name, long long value);
/* The present value of an option: what a caller that changes one for a while puts back afterwards (unknown name: error). */
int midyn_ctx_get_option(midyn_ctx* ctx, const char* name, long long* value);

/* ---- operator stack ----------------------------------------------------------------------
 * Replaces OperatorCollection.__init__ (models/operator_collections.py:54-81): the device-resident
 * (k,n,n) stack + static operator (+ frame diagonal of models/rotating_frame.py:59-112).
 * `midyn_stack_packed_bytes` gives the size of the single packed device buffer; when `dev_buffer`
 * is non-NULL the stack is built inside that caller-owned device allocation (e.g. a torch uint8
 * tensor, so that rank 0 can RCCL-broadcast it); `midyn_stack_adopt` wraps an already packed
 * (broadcast) buffer on the receiving ranks. */
int midyn_stack_packed_bytes(int n, int k, int has_static, size_t* bytes);
int midyn_stack_create(midyn_ctx* ctx, int n, int k, const midyn_complex* ops,
                       const midyn_complex* static_op, const double* frame_im, void* dev_buffer,
                       midyn_stack** out);
int midyn_stack_adopt(midyn_ctx* ctx, int n, int k, int has_static, int has_frame,
                      void* dev_buffer, midyn_stack** out);
/* Vectorised Lindblad model: the n^2 x n^2 column-stacking superoperators are built ON THE DEVICE from the
 * n x n operators (all in the frame basis), replacing the host Kronecker products of
 * models/operator_collections.py:851-1061 / models/model_utils.py:31-118:
 *   static    = vec_commutator(h_d) + sum_j vec_dissipator(n_static[j])        (h_d NULL / n_s = 0: part absent)
 *   operators = [vec_commutator(h_ops[j]), j < k_h ; vec_dissipator(l_ops[j]), j < k_l]
 * h_d (n,n), h_ops (k_h,n,n), n_static (n_s,n,n), l_ops (k_l,n,n) host arrays; frame_im (n^2) or NULL
 * (rotating_frame.py:510-582 vectorised frame diagonal).  The stack has dimension n^2 and k_h + k_l operators. */
int midyn_stack_create_lindblad(midyn_ctx* ctx, int n, const midyn_complex* h_d, int k_h,
                                const midyn_complex* h_ops, int n_s, const midyn_complex* n_static, int k_l,
                                const midyn_complex* l_ops, const double* frame_im, midyn_stack** out);
/* defect[seg] = || A_seg + A_seg^dagger ||_F for seg < n_segments: zero for anti-Hermitian generators.  For a
 * Hamiltonian model (A = -iH) this is || H - H^dagger ||_F, the quantity the reference's constructor validates
 * (models/hamiltonian_model.py:98-104, is_hermitian :196-222), evaluated on the device. */
int midyn_stack_antiherm_defect(midyn_stack* stack, double* defect);
int midyn_stack_destroy(midyn_stack* stack);
/* info[0..7] = n, n_pad, k, has_static, has_frame, n_segments, n_active_segments, packed_bytes>>20 */
int midyn_stack_info(midyn_stack* stack, long long* info);
/* modes[seg] for seg < n_segments: 0 dense complex, 1 real only, 2 imaginary only, 3 exactly zero
 * (exact-zero planes are detected once on the device; they let the MFMA contraction skip the
 * corresponding real products without changing any result bit). */
int midyn_stack_segment_modes(midyn_stack* stack, int* modes);

/* ---- single evaluations --------------------------------------------------------------------
 * midyn_eval_generator: GeneratorModel.evaluate in the frame basis
 *   (models/generator_model.py:256-279 -> OperatorCollection.evaluate operator_collections.py:101-122
 *    -> RotatingFrame._conjugate_and_add rotating_frame.py:286-370):
 *      G = Delta(t) o (G_d + sum_j c_j G_j),  Delta_ab = conj(e_a) e_b,  e = exp(d t).
 * midyn_eval_rhs: GeneratorModel.evaluate_rhs in the frame basis (generator_model.py:281-316):
 *      out = exp(-d t) o ( (G_d + sum_j c_j G_j) (exp(d t) o y) ),   y is [n][m]. */
int midyn_eval_generator(midyn_stack* stack, const double* coeffs, double t, midyn_complex* G_out);
int midyn_eval_rhs(midyn_stack* stack, const double* coeffs, double t, const midyn_complex* y,
                   int m, midyn_complex* out);

/* ---- coefficient table evaluated on the device ------------------------
 * SignalList.__call__ over SignalSums of DiscreteSignals / constants (signals/signals.py:148-155,
 * 302-311,574-577,801-803; the sweep precedent is _solve_schedule_list_jax, solvers/solver_classes.py:
 * 592-676, whose signals are all DiscreteSignals):
 *   S[b][r][j] = sum over the terms q of signal j of instance b of
 *                Re[ f_q(t_r) exp(i (2 pi nu_q t_r + phi_q)) ],
 *   f_q(t) = samples_q[ clip(floor_divide(t - t0_q, dt_q), -1, len_q) ]  (0 outside the window).
 * term_ptr[B*k+1] is a CSR index (terms of signal j of instance b are term_ptr[b*k+j] ..
 * term_ptr[b*k+j+1]-1); term_params[q] = (dt, start_time, carrier_freq, phase), dt == 0 marking a
 * constant envelope (= its one sample); sample_ptr[q] = (offset, length) into `samples`, so terms may
 * share sample arrays.  The table stays in HBM: pass the pointer from midyn_sigtable_data as `S`
 * to any solve entry point below (every `S` parameter accepts a host OR a device pointer). */
typedef struct midyn_sigtable midyn_sigtable;
int midyn_sigtable_create(midyn_ctx* ctx, int B, int k, int R, const double* times,
                          const long long* term_ptr, const double* term_params,
                          const long long* sample_ptr, const midyn_complex* samples,
                          midyn_sigtable** out);
/* dims (optional) = B, R, k */
int midyn_sigtable_data(midyn_sigtable* tab, const double** dev_S, long long* dims);
int midyn_sigtable_fetch(midyn_sigtable* tab, double* S_out);
int midyn_sigtable_destroy(midyn_sigtable* tab);

/* ---- fixed-step RK4 (solvers/fixed_step_solvers.py:43-77 inside the template :406-459) --------
 * B independent instances advance together (the loop of solvers/solver_classes.py:556-590 turned
 * into one batched contraction).  The caller evaluates the signals on the host into the table
 * S[B][R][k] (signals/signals.py:792-803) at the R distinct times `times[R]`; step s uses table
 * rows step_rows[s][0..2] = (t, t+h/2, t+h) and step size step_h[s]; after step s the state is
 * stored in output slot step_save[s] when that is >= 1 (negative: not saved).  Slot 0 always receives y0: step_save[s] == 0
 * is refused by the one-launch sweep route of midyn_expm_solve / midyn_expm_plan_* and should not be used on any route.
 * y0 is [B][n][m], or [n][m] when y0_shared != 0; Y_out is [B][P][n][m]. */
int midyn_rk4_solve(midyn_stack* stack, int B, int m, int R, const double* times, const double* S,
                    int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                    int P, const midyn_complex* y0, int y0_shared, midyn_complex* Y_out);

/* ---- matrix exponential (scipy.linalg.expm as called at solvers/fixed_step_solvers.py:22,104) --
 * E_out[b] = expm(A[b]) for `batch` n x n matrices.  Algorithm: scaling and squaring of a Taylor
 * polynomial (degree 2..16 chosen from the 1-norm) evaluated with Paterson-Stockmeyer (matrix
 * products only, all on the fp64 MFMA zgemm); see DESIGN.md for the parity statement. info (optional, [batch][2]):
 * squarings s and the 1-norm (as a truncated integer *1e6). */
int midyn_expm(midyn_ctx* ctx, int n, int batch, const midyn_complex* A, midyn_complex* E_out,
               long long* info);

/* ---- fixed-step Magnus/expm solver (solvers/fixed_step_solvers.py:80-108,321-403) ----------------
 * y <- expm(Omega_m) y per step; the generator evaluations use table rows step_rows[s][0..m-1]
 * (m = magnus_order Gauss points, in the order of fixed_step_solvers.py:345-377).
 * Instances advance together in chunks sized so that one batched launch fills the device (a
 * chunk is a single instance once one n x n product does; hundreds of instances for small n).
 * For states with few columns (16 m <= n_pad) and magnus_order <= 2 the exponential is not formed:
 * expm(Omega) y is evaluated as a scaled Taylor series of products Omega.v on the batched RHS
 * contraction (all instances at once; option "expm_action" = 0 forces the dense route). */
int midyn_expm_solve(midyn_stack* stack, int B, int m, int R, const double* times, const double* S,
                     int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                     int P, int magnus_order, const midyn_complex* y0, int y0_shared,
                     midyn_complex* Y_out);

/* ---- parallel-in-time propagation --------------------------------------
 * fixed_step_lmde_solver_parallel_template_jax (solvers/fixed_step_solvers.py:524-613) with the step
 * rule of jax_RK4_parallel_solver (:222-258; method 0) or jax_expm_parallel_solver (:289-316; method =
 * Magnus order 1..3): the propagators of all steps are formed by batched launches, the ones between
 * consecutive output times are multiplied by a binary tree (one batched zgemm per level), and the
 * interval propagators are applied to y0 in time order.  Arguments as midyn_rk4_solve /
 * midyn_expm_solve (step_rows[s] = the table rows of step s: (t, t+h/2, t+h) for method 0, the
 * Gauss points for Magnus).  Instances (B) are processed one after the other. */
int midyn_parallel_solve(midyn_stack* stack, int B, int m, int R, const double* times, const double* S,
                         int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                         int P, int method, const midyn_complex* y0, int y0_shared, midyn_complex* Y_out);

/* ---- perturbative Dyson / Magnus expansion step -------------------------
 * Run-time part of DysonSolver / MagnusSolver (solvers/perturbative_solvers/dyson_solver.py:187-207,
 * magnus_solver.py:107-129, perturbative_solver.py:172-219): per time step k the array polynomial
 * (perturbation/array_polynomial.py:524-544)   X_k = [constant_term +] sum_I mono[k][I] terms[I]
 * gives the step propagator P_k = X_k (use_expm == 0, Dyson; constant_term = Udt and the terms
 * already carry Udt) or P_k = post . expm(X_k) (use_expm != 0, Magnus; post = Udt), and
 * y <- P_{T-1} ... P_0 y.  terms is [M][n][n]; mono[B][nsteps][M] holds the monomials c^I of the
 * Chebyshev coefficients of every step (host or device pointer); y0 is [B or 1][n][m];
 * Y_out [B][n][m] receives the final states.  All steps of a chunk are evaluated by ONE GEMM and
 * multiplied by a binary tree (cf. midyn_parallel_solve). */
typedef struct midyn_expansion midyn_expansion;
int midyn_expansion_create(midyn_ctx* ctx, int n, int M, const midyn_complex* terms,
                           const midyn_complex* constant_term, const midyn_complex* post, int use_expm,
                           midyn_expansion** out);
int midyn_expansion_destroy(midyn_expansion* exp);
int midyn_expansion_solve(midyn_expansion* exp, int B, int nsteps, const double* mono, int m,
                          const midyn_complex* y0, int y0_shared, midyn_complex* Y_out);
/* The monomials on the device as well (the first half of ArrayPolynomial.__call__: perturbation/array_polynomial.py:524-528 with
 * :547-601): midyn_expansion_set_monomials hands over the multiset labels of the M terms once -- labels[M][order], row I = the indices
 * (< n_vars) of the coefficients whose product is monomial I, -1 behind the last one -- and midyn_expansion_solve_coeffs takes
 * coeffs[B][n_vars][nsteps], the Chebyshev coefficients of every step (expansion_model.py:410-551), instead of the (nsteps x M) table:
 * 32 KB instead of 0.3-0.5 MB per 1000-step solve cross the bus, nothing but the signal evaluation is left on the host.  The products are
 * associated as the host's table does (c[l0] * (c[l1] * ...)): both entry points give the same bits. */
int midyn_expansion_set_monomials(midyn_expansion* exp, int n_vars, int order, const int* labels);
int midyn_expansion_solve_coeffs(midyn_expansion* exp, int B, int nsteps, const double* coeffs, int m,
                                 const midyn_complex* y0, int y0_shared, midyn_complex* Y_out);

/* ---- non-vectorised Lindblad RHS -------------------------------------
 * LindbladCollection.evaluate_rhs (models/operator_collections.py:451-567) with n x n zgemms:
 *   rhs = (A+B) rho + rho (A-B) + sum_j N_j rho N_j^+ + sum_j gamma_j L_j rho L_j^+,
 *   B = -iH(t), A = -1/2 sum N^+N - 1/2 sum gamma_j L_j^+L_j,
 * evaluated in the rotating frame / frame basis as LindbladModel.evaluate_rhs does
 * (models/lindblad_model.py:477-538).  `left` / `right` are operator stacks for A+B and A-B that
 * share the coefficient vector c = (s_0..s_{k_h-1}, gamma_0..gamma_{n_dyn-1}) (the frame diagonal
 * is taken from `left`); `dissipators` = [n_static + n_dyn][n][n], static ones first.
 * rho is [batch][n][n]; the RK4 solve mirrors midyn_rk4_solve with rho0 [B or 1][n][n] and
 * out [B][P][n][n]. */
typedef struct midyn_lindblad midyn_lindblad;
int midyn_lindblad_create(midyn_stack* left, midyn_stack* right, int k_h, int n_static, int n_dyn,
                          const midyn_complex* dissipators, midyn_lindblad** out);
int midyn_lindblad_destroy(midyn_lindblad* lind);
int midyn_lindblad_rhs(midyn_lindblad* lind, const double* coeffs, double t, const midyn_complex* rho,
                       int batch, midyn_complex* out);
int midyn_lindblad_rk4_solve(midyn_lindblad* lind, int B, int R, const double* times, const double* S,
                             int nsteps, const int* step_rows, const double* step_h,
                             const int* step_save, int P, const midyn_complex* rho0, int rho0_shared,
                             midyn_complex* out);

/* ---- plain complex GEMM on the same MFMA kernel (yardstick + tests) --------------------------- */
int midyn_zgemm(midyn_ctx* ctx, int M, int N, int K, const midyn_complex* A, const midyn_complex* B,
                midyn_complex* C);

/* ---- bench hooks: work on DEVICE-RESIDENT data so the timed region excludes PCIe ---------------
 * midyn_rk4_plan_create uploads everything midyn_rk4_solve needs and returns a plan;
 * midyn_rk4_plan_run executes steps [step_begin, step_end) asynchronously on the ctx stream
 * (state continues from the previous call); midyn_rk4_plan_fetch copies the current state out as
 * [B][n][m]. */
typedef struct midyn_rk4_plan midyn_rk4_plan;
int midyn_rk4_plan_create(midyn_stack* stack, int B, int m, int R, const double* times,
                          const double* S, int nsteps, const int* step_rows, const double* step_h,
                          const midyn_complex* y0, int y0_shared, midyn_rk4_plan** out);
int midyn_rk4_plan_run(midyn_rk4_plan* plan, int step_begin, int step_end);
int midyn_rk4_plan_fetch(midyn_rk4_plan* plan, midyn_complex* Y_out);
int midyn_rk4_plan_destroy(midyn_rk4_plan* plan);

/* ---- result blocks the device writes directly -------------------------------------------------------------------------------------
 * midyn_host_alloc returns page-locked host memory that the GPU can write (hipHostMalloc); midyn_host_free gives it back.  The
 * one-launch sweep kernels store their saved states straight into a result block (Y_out of midyn_expm_solve, Y_direct of
 * midyn_expm_plan_run) when it is such memory -- no device copy, no download.  A block from midyn_host_alloc is recognised by its
 * address (a table inside the library); any other pointer is examined with hipPointerGetAttributes at every call (60-80 us), and
 * ordinary pageable memory simply takes the device block + copy.  The Python binding's result arrays (>= 1 MB) come from here. */
int midyn_host_alloc(size_t bytes, void** out);
int midyn_host_free(void* ptr);

/* ---- device-resident Magnus/expm solve of a parameter scan (solvers/fixed_step_solvers.py:80-108, 345-363, 406-459) ----
 * A scan re-solves the SAME model on the SAME time grid with new signal parameters (solvers/solver_classes.py:556-590).
 * midyn_expm_plan_create keeps what those solves share on the device: the frame-phase table of `times`, the step tables, y0, the
 * result block and the exchange slots of the one-launch sweep kernels (arguments as midyn_expm_solve, without S).
 * midyn_expm_plan_run(plan, S, Y_direct) uploads the coefficient table S[B][R][k] (host or device pointer), chooses the series of
 * every step from the norm bounds of THIS table, and launches; it returns when the launch is queued.  Y_direct (optional,
 * [B][P][n][m]): when it is device-writable host memory (hipHostMalloc / hipHostRegister) the kernel writes the saved states
 * straight into it (ctx option expm_direct_out [1]); otherwise they stay on the device until the fetch.  A block the plan has
 * accepted once must stay pinned for as long as it is handed to this plan (the query is made once per block).
 * midyn_expm_plan_fetch(plan, Y_out) waits for the launch and delivers [B][P][n][m] (slot 0 = y0); with Y_out == the Y_direct
 * of the run nothing is copied.  A plan can run any number of times.  Solves that the one-launch sweep kernels do not take
 * (dense stacks, matrix states, magnus_order 3) are run by midyn_expm_solve itself inside midyn_expm_plan_run: same results,
 * no saving.  midyn_expm_solve on the sweep route is create + run + fetch + destroy of this plan. */
typedef struct midyn_expm_plan midyn_expm_plan;
int midyn_expm_plan_create(midyn_stack* stack, int B, int m, int R, const double* times, int nsteps,
                           const int* step_rows, const double* step_h, const int* step_save, int P,
                           int magnus_order, const midyn_complex* y0, int y0_shared, midyn_expm_plan** out);
int midyn_expm_plan_run(midyn_expm_plan* plan, const double* S, midyn_complex* Y_direct);
int midyn_expm_plan_fetch(midyn_expm_plan* plan, midyn_complex* Y_out);
int midyn_expm_plan_destroy(midyn_expm_plan* plan);

/* ---- counters ----------------------------------------------------------------------------------
 * Kernel-time accounting measured with HIP events on the ctx stream.
 * names: "rhs_stream", "rhs_gemm", "zgemm", "gen_eval", "elementwise", and for the block-sparse routes
 * "rhs_blocks" (1..8 columns) and "rhs_blocks_gemm" (MFMA tiles over work lists); "rk4_resident" counts the
 * launches of the register-resident single-trajectory kernel (one launch = a whole step range).
 * out[0] = launches, out[1] = total ms (events are only recorded when profiling is enabled with
 * midyn_ctx_set_option(ctx, "profile", 1); it adds two event records per launch).
 * "sweep_series" describes the last ell_sweep_kernel launch: (series terms per instance summed over the steps, operator
 * slots per row); "sweep_split": (workgroups per instance of that launch, element form 0 general / 1 packed / 2 direct / 3 none: flip masks);
 * "resident_fallbacks": (step ranges a one-launch kernel gave up on and the launch-per-product route re-ran, 0).
 * Two more names describe the LAST launch of the sparse MFMA route: "sparse_tile" -> (BM, BN) of its tile,
 * "sparse_list" -> (listed (panel, K tile, operator) tiles of the stack for that panel height, split count),
 * "sparse_pair" -> (contractions in that launch: 2 when two independent products shared it, ctx option pair_launch).
 * "rhs_combine" counts the launches of the COMBINE + APPLY sweep kernel; its last launch is described by
 * "combine_info" -> (listed (32-row group, 16-column block) entries of the stack, 100 NRE4 + 10 NIM4 + STAT: groups of four
 * real / imaginary operator planes and the static operator's planes, bit 0 real, bit 1 imaginary) and "combine_shape" ->
 * ((row group, column block) pairs per workgroup, waves that split the list of one pair), "combine_wave" -> (state columns
 * per wave of that launch: 64, or 32 for stacks with more than two plane groups and for small sweeps, 0).  A one-launch RK4 sweep
 * (combine_sweep) counts as ONE "rhs_combine" launch; "combine_sweep" -> (workgroups of 16 instances, 10 x waves per
 * workgroup + 16-row tiles per wave) of the last one.
 * "flops:<class>" -> (real floating-point operations the dense MFMA contraction launches of that class executed while
 * `profile` was on -- per complex multiply-add 8 with four real products, 6 with three (3M), 4 for a one-plane operand --, 0):
 * what bench.py divides by the class's kernel time for the rooflines of the dense expm and the Lindblad products. */
int midyn_get_counters(midyn_ctx* ctx, const char* name, double* out);
int midyn_reset_counters(midyn_ctx* ctx);
/* Measured ceilings: "mfma_f64" -> out[0] = TFLOP/s of back-to-back v_mfma_f64_16x16x4_f64; "mfma_f64_sustained"
 * (operands with random mantissas) / "mfma_f64_sustained_zero" -> out[0] TFLOP/s, out[1] shader clock in GHz, out[2] ms of
 * a ~14 ms stream at the contraction kernels' cadence (2 waves per SIMD, 16 accumulator quads);
 * "hbm_read" -> GB/s streaming a 4 GiB buffer; "mall_read" -> GB/s re-reading 144 MiB (the size of
 * the cfg-2 operator stack, which fits the 256 MiB Infinity Cache);
 * "fp64_coissue" -> out[0..2] = ms of the same rounds with MFMAs + vector fp64 FMAs interleaved, MFMAs only, FMAs only,
 * out[3] = TFLOP/s of the combined run (do the fp64 matrix pipe and the fp64 vector ALUs run concurrently?).
 * `out` holds 4 doubles. */
int midyn_microbench(midyn_ctx* ctx, const char* name, double* out);
/* HIP-event stopwatch on the context's stream: midyn_ctx_timer(ctx, 0, NULL) records the start event,
 * midyn_ctx_timer(ctx, 1, &ms) records the stop event, waits for it and returns the elapsed milliseconds
 * (brackets a region of back-to-back launches without per-launch event records). */
int midyn_ctx_timer(midyn_ctx* ctx, int stop, double* ms);
/* Block occupancy of a stack (what the work-list kernels read and multiply): out[0] state (1 lists built, -1 not
 * applicable), out[1] fraction of non-zero 16x16 blocks of the active operators, out[2] their number, out[3] n_pad/16,
 * then for the MFMA tile lists of 64/128/32/16-row panels: out[4+2t] listed fraction, out[5+2t] listed tiles.
 * out[12] = fraction of the active planes the one-column streaming kernel reads (column hull of the non-zero blocks
 * per 16-row group; 1 when the hulls are not used).  `out` holds 13 doubles. */
int midyn_stack_block_info(midyn_stack* stack, double* out);

/* ---- multi-GPU: the one collective of the path --------------------------------------------------
 * Sweep instances are independent (the reference loops them sequentially, solvers/solver_classes.py:568-586);
 * the only exchange is ONE RCCL broadcast of the packed operator stack from the rank that built it.
 * One process (and one midyn_ctx) per GPU.  `nccl_comm` is an RCCL `ncclComm_t` passed as void*: either the
 * caller's own communicator or one made by midyn_comm_init_rank from a 128-byte ncclUniqueId that rank 0 obtains
 * with midyn_comm_get_unique_id and ships to the other ranks through any channel (file, socket, MPI, torch store).
 * librccl is resolved at first use (already loaded in the process, else MIDYN_RCCL_LIB / librccl.so.1).
 * Receiving ranks create an empty stack of the same shape and call midyn_stack_broadcast with the same root. */
#define MIDYN_COMM_ID_BYTES 128
int midyn_comm_get_unique_id(void* id128);
int midyn_comm_init_rank(midyn_ctx* ctx, int world, int rank, const void* id128, void** nccl_comm_out);
int midyn_comm_destroy(midyn_ctx* ctx, void* nccl_comm);
int midyn_comm_count(midyn_ctx* ctx, void* nccl_comm, int* ranks_out);   /* ncclCommCount: ranks of the communicator */
int midyn_stack_create_empty(midyn_ctx* ctx, int n, int k, int has_static, int has_frame, midyn_stack** out);
int midyn_stack_broadcast(midyn_stack* stack, void* nccl_comm, int root);
/* The same broadcast OUT OF PLACE: the root sends `src`'s packed buffer (read on the root only; NULL elsewhere), every
 * rank -- the root too -- receives into `dst` (same shape, normally from midyn_stack_create_empty) and derives its
 * host-side lists from the received content, exactly as a non-root rank of midyn_stack_broadcast does.  With a
 * one-rank communicator it is a copy through RCCL plus the receiving side, i.e. a single GPU can run what ranks
 * 1..N-1 run.  midyn_stack_broadcast(s, comm, root) == midyn_stack_broadcast_from(s, s, comm, root).
 * Errors and the other ranks: EVERY rank of the communicator must make the call (it is a collective).  A shape that
 * does not match the root's -- which only the rank that has it can see -- does not leave the others hanging: the ranks
 * first exchange 48 bytes (the root's shape to everybody, the minimum of everybody's verdict back), and either all of
 * them broadcast the buffer or none does and each returns non-zero.  What still hangs, as with any collective: a rank
 * that never calls, or that fails before the handshake (NULL arguments, no librccl, root out of range). */
int midyn_stack_broadcast_from(midyn_stack* dst, midyn_stack* src, void* nccl_comm, int root);

#ifdef __cplusplus
}
#endif
#endif /* MIDYN_H */
