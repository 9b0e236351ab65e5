// midyn.hip -- host side of libmidyn.so: the C-ABI of include/midyn.h on top of the gfx950 kernels
// in midyn_kernels.h.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC midyn.hip
//
// No CPU fallback exists in this library: every entry point needs a HIP device and fails with a
// non-zero status (text via midyn_last_error) when there is none.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/midyn.h"
#include "midyn_kernels.h"

using namespace midyn;

// -------------------------------------------------------------------------------------------------
// context, errors, profiling
// -------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

enum KClass { KC_STREAM = 0, KC_RHS_GEMM, KC_ZGEMM, KC_GEN, KC_ELEM, KC_BLOCKS, KC_BLOCKS_GEMM, KC_COUNT };
static const char* kclass_names[KC_COUNT] = {"rhs_stream", "rhs_gemm", "zgemm", "gen_eval", "elementwise",
                                             "rhs_blocks", "rhs_blocks_gemm"};  // the last two: block-sparse routes

struct EventPair {
    hipEvent_t a, b;
    int cls;
};

struct midyn_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool skip_zero_planes = true;
    int chebyshev = 1;             // expm action, Magnus order 1, nearly skew-Hermitian generator: Chebyshev series
                                   // instead of the scaled Taylor series (1: when shorter, 2: always, 0: never)
    int sparse_bm = 0;             // A/B: pin the row panels of the sparse MFMA route (16 | 32 | 64 | 128; 0 = by list density)
    bool skip_zero_blocks = true;  // block-sparse stacks: contract only the 16 x 16 operator blocks that hold a non-zero
    bool profile = false;
    int force_tile = 0;  // 0 auto, 64, 128, 12864
    bool prefer_duo = false;
    int ablate = 0;
    int stream_variant = 0;
    int expm_degree = 0;         // 0: Taylor degree chosen from the norm; else forced (2,4,6,9,12,16)
    int krylov = 1;              // one column, Magnus order 1: Arnoldi instead of the scaled Taylor series
                                 // (1: when the series is long enough to pay for it, 2: always, 0: never)
    bool expm_action = true;     // few state columns: y <- expm(Omega) y as a Taylor series of matrix-vector
                                 // products instead of forming expm(Omega) (Magnus orders 1 and 2)
    bool stream_planes = true;   // single-plane stacks: the one-column kernel streams only non-zero planes
    bool tiny_rk4 = true;        // small systems: whole RK4 solve in one persistent launch (tiny_rk4_kernel)
    bool multi_stream = true;    // 2..8 state columns at n >= 256: multi-column streaming kernel
    bool split_k = true;
    bool combine_first = true;
    bool plane_kernel = false;  // planar two-tiles-per-barrier variant: measured 4 % SLOWER (2.43 vs 2.33 ms), kept opt-in
    bool complex_3m = true;   // dense complex products by the 3M scheme (3 real MFMAs instead of 4)
    int force_splits = 0;
    void* splitk_ws = nullptr;
    size_t splitk_bytes = 0;
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> pool;
    double cls_ms[KC_COUNT] = {0};
    double cls_n[KC_COUNT] = {0};
    int* d_one_seg = nullptr;  // device int {0, 1}: single-segment lists for plain zgemm (dense A / real-only A)
    // device-memory pool: DevBuf blocks are recycled instead of hipMalloc/hipFree'd (a small solve is a
    // few hundred microseconds of kernels; a dozen allocations per call used to cost milliseconds)
    std::vector<std::pair<size_t, void*>> mem_pool;  // (capacity, block)
    size_t mem_pool_bytes = 0;
    static constexpr size_t POOL_MAX_BYTES = (size_t)2 << 30, POOL_MAX_BLOCK = (size_t)256 << 20;
    double* h_pinned = nullptr;                  // pinned host scratch for small device-to-host results (norms)
    static constexpr size_t PINNED_DOUBLES = 1 << 17;
    int num_cu = 256;
};

static int fail(midyn_ctx* ctx, const std::string& msg) {
    g_last_error = msg;
    if (ctx) ctx->err = msg;
    return 1;
}

#define HIPCHK(ctx, expr)                                                                   \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess)                                                               \
            return fail((ctx), std::string(#expr) + ": " + hipGetErrorString(_e) + " at " + \
                                   __FILE__ + ":" + std::to_string(__LINE__));              \
    } while (0)

#define CHK(expr)                  \
    do {                           \
        int _s = (expr);           \
        if (_s != 0) return _s;    \
    } while (0)

struct ProfScope {
    midyn_ctx* ctx;
    EventPair ep;
    bool on;
    ProfScope(midyn_ctx* c, int cls) : ctx(c), on(c->profile) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!ctx->pool.empty()) {
                e = ctx->pool.back();
                ctx->pool.pop_back();
            } else {
                hipEventCreate(&e);
            }
            return e;
        };
        ep.a = get();
        ep.b = get();
        ep.cls = cls;
        hipEventRecord(ep.a, ctx->stream);
    }
    ~ProfScope() {
        if (!on) return;
        hipEventRecord(ep.b, ctx->stream);
        ctx->pending.push_back(ep);
    }
};

static void drain_events(midyn_ctx* ctx) {
    if (ctx->pending.empty()) return;
    hipStreamSynchronize(ctx->stream);
    for (auto& ep : ctx->pending) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, ep.a, ep.b);
        ctx->cls_ms[ep.cls] += ms;
        ctx->cls_n[ep.cls] += 1;
        ctx->pool.push_back(ep.a);
        ctx->pool.push_back(ep.b);
    }
    ctx->pending.clear();
}

extern "C" const char* midyn_last_error(midyn_ctx* ctx) {
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

extern "C" int midyn_ctx_create(int device, midyn_ctx** out) {
    if (!out) return fail(nullptr, "midyn_ctx_create: out is NULL");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(nullptr, std::string("midyn_ctx_create: no HIP device available (") +
                                 hipGetErrorString(e) + "); libmidyn has no CPU fallback");
    if (device < 0 || device >= count)
        return fail(nullptr, "midyn_ctx_create: device index out of range");
    midyn_ctx* ctx = new midyn_ctx();
    ctx->device = device;
    HIPCHK(ctx, hipSetDevice(device));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, device));
    ctx->num_cu = prop.multiProcessorCount;
    if (const char* e = getenv("MIDYN_COMPLEX_3M")) ctx->complex_3m = atoi(e) != 0;
    HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_pinned, midyn_ctx::PINNED_DOUBLES * sizeof(double), hipHostMallocDefault));
    HIPCHK(ctx, hipMalloc(&ctx->d_one_seg, 2 * sizeof(int)));
    int one_seg[2] = {0, 1};  // [0]: segment 0, dense complex A;  [1]: segment 0, A real-only (mode 1)
    HIPCHK(ctx, hipMemcpy(ctx->d_one_seg, one_seg, sizeof(one_seg), hipMemcpyHostToDevice));
    *out = ctx;
    return 0;
}

extern "C" int midyn_ctx_destroy(midyn_ctx* ctx) {
    if (!ctx) return 0;
    hipSetDevice(ctx->device);
    drain_events(ctx);
    for (auto e : ctx->pool) hipEventDestroy(e);
    if (ctx->d_one_seg) hipFree(ctx->d_one_seg);
    if (ctx->h_pinned) hipHostFree(ctx->h_pinned);
    for (auto& blk : ctx->mem_pool) hipFree(blk.second);
    ctx->mem_pool.clear();
    if (ctx->splitk_ws) hipFree(ctx->splitk_ws);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return 0;
}

extern "C" int midyn_ctx_synchronize(midyn_ctx* ctx) {
    if (!ctx) return fail(nullptr, "NULL ctx");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int midyn_ctx_set_option(midyn_ctx* ctx, const char* name, long long value) {
    if (!ctx || !name) return fail(ctx, "midyn_ctx_set_option: NULL argument");
    std::string n(name);
    if (n == "skip_zero_planes") ctx->skip_zero_planes = value != 0;
    else if (n == "skip_zero_blocks") ctx->skip_zero_blocks = value != 0;
    else if (n == "sparse_bm") ctx->sparse_bm = (int)value;
    else if (n == "chebyshev") ctx->chebyshev = (int)value;
    else if (n == "profile") {
        if (!value) drain_events(ctx);
        ctx->profile = value != 0;
    } else if (n == "force_tile") ctx->force_tile = (int)value;
    else if (n == "prefer_duo") ctx->prefer_duo = value != 0;
    else if (n == "ablate") ctx->ablate = (int)value;
    else if (n == "stream_variant") ctx->stream_variant = (int)value;
    else if (n == "stream_planes") ctx->stream_planes = value != 0;
    else if (n == "tiny_rk4") ctx->tiny_rk4 = value != 0;
    else if (n == "multi_stream") ctx->multi_stream = value != 0;
    else if (n == "expm_degree") ctx->expm_degree = (int)value;
    else if (n == "expm_action") ctx->expm_action = value != 0;
    else if (n == "krylov") ctx->krylov = (int)value;
    else if (n == "split_k") ctx->split_k = value != 0;
    else if (n == "combine_first") ctx->combine_first = value != 0;
    else if (n == "complex_3m") ctx->complex_3m = value != 0;
    else if (n == "plane_kernel") ctx->plane_kernel = value != 0;
    else if (n == "force_splits") ctx->force_splits = (int)value;
    else return fail(ctx, "midyn_ctx_set_option: unknown option " + n);
    return 0;
}

extern "C" int midyn_get_counters(midyn_ctx* ctx, const char* name, double* out) {
    if (!ctx || !name || !out) return fail(ctx, "midyn_get_counters: NULL argument");
    drain_events(ctx);
    for (int i = 0; i < KC_COUNT; ++i)
        if (std::string(name) == kclass_names[i]) {
            out[0] = ctx->cls_n[i];
            out[1] = ctx->cls_ms[i];
            return 0;
        }
    return fail(ctx, std::string("midyn_get_counters: unknown counter ") + name);
}

extern "C" int midyn_reset_counters(midyn_ctx* ctx) {
    if (!ctx) return fail(nullptr, "NULL ctx");
    drain_events(ctx);
    for (int i = 0; i < KC_COUNT; ++i) ctx->cls_ms[i] = ctx->cls_n[i] = 0;
    return 0;
}

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
static inline int grid_for(size_t total, int cap = 4096) {
    size_t b = (total + 255) / 256;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

// -------------------------------------------------------------------------------------------------
// operator stack
// -------------------------------------------------------------------------------------------------
struct midyn_stack {
    midyn_ctx* ctx = nullptr;
    int n = 0, n_pad = 0, k = 0, has_static = 0, has_frame = 0, nseg = 0;
    char* buf = nullptr;  // packed device buffer
    bool owns = false;
    size_t bytes = 0;
    double2* ops = nullptr;     // [nseg][n_pad][n_pad]
    double* frame_im = nullptr; // [n_pad]
    int* flags = nullptr;       // [2*nseg] plane non-zero flags (device)
    int* seg_all = nullptr;     // [nseg] every segment, mode 0 (device)
    int* seg_act = nullptr;     // [nseg] active list with plane modes (device)
    bool all_single_plane = false;  // every active segment is purely real or purely imaginary
    double* planes = nullptr;       // [n_act][n_pad][n_pad] planar copy of the non-zero planes (lazy)
    struct midyn_rk4_plan* eval_plan = nullptr;  // cached buffers of midyn_eval_rhs (keyed by m)
    int eval_m = 0;
    int n_act = 0;
    int uniform_mode = 3;       // plane mode shared by all active segments, or 3 (mixed)
    std::vector<int> h_flags;
    std::vector<int> h_modes;   // per segment: 0 full, 1 real only, 2 imaginary only, 3 zero
    std::vector<double> seg_norm1;  // ||A_seg||_1 per segment (lazy; norm bounds of the expm action)
    std::vector<double> seg_norminf, seg_herm1;  // ||A_seg||_inf and ||(A_seg + A_seg^dagger)/2||_1 (lazy; Chebyshev action)
    // block occupancy (lazy, stack_block_lists): which 16 x 16 blocks of the active segments hold a non-zero
    int blk_state = 0;              // 0 not examined, 1 lists built, -1 not applicable
    double blk_density = 1.0;       // non-zero 16 x 16 blocks / all blocks of the active segments
    int* blk_ptr = nullptr;         // streaming lists per group of 16 rows: [n_pad/16 + 1]
    int* blk_idx = nullptr;         // entry = (segment << 16) | column chunk
    int* gw_ptr[4] = {nullptr, nullptr, nullptr, nullptr};  // MFMA tile lists per row panel of 64 / 128 / 32 / 16 rows: [M/BM + 1]
    int* gw_idx[4] = {nullptr, nullptr, nullptr, nullptr};  // entry = (K tile << 8) | (seg << 2 | mode)
    double gw_density[4] = {1.0, 1.0, 1.0, 1.0};            // listed tiles / all (panel, K tile, active segment) tiles
    double gw_avg[4] = {0.0, 0.0, 0.0, 0.0};                // average list length per row panel
};

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct PackLayout {
    size_t off_ops, off_frame, off_flags, off_all, off_act, total;
};
static PackLayout pack_layout(int n_pad, int nseg) {
    PackLayout L;
    size_t o = 0;
    L.off_ops = o;
    o = align256(o + (size_t)nseg * n_pad * n_pad * sizeof(double2));
    L.off_frame = o;
    o = align256(o + (size_t)n_pad * sizeof(double));
    L.off_flags = o;
    o = align256(o + (size_t)2 * std::max(nseg, 1) * sizeof(int));
    L.off_all = o;
    o = align256(o + (size_t)std::max(nseg, 1) * sizeof(int));
    L.off_act = o;
    o = align256(o + (size_t)std::max(nseg, 1) * sizeof(int));
    L.total = o;
    return L;
}

extern "C" int midyn_stack_packed_bytes(int n, int k, int has_static, size_t* bytes) {
    if (!bytes || n <= 0 || k < 0) return fail(nullptr, "midyn_stack_packed_bytes: bad argument");
    *bytes = pack_layout(round_up(n, 64), k + (has_static ? 1 : 0)).total;
    return 0;
}

static void stack_bind(midyn_stack* s) {
    PackLayout L = pack_layout(s->n_pad, s->nseg);
    s->bytes = L.total;
    s->ops = reinterpret_cast<double2*>(s->buf + L.off_ops);
    s->frame_im = reinterpret_cast<double*>(s->buf + L.off_frame);
    s->flags = reinterpret_cast<int*>(s->buf + L.off_flags);
    s->seg_all = reinterpret_cast<int*>(s->buf + L.off_all);
    s->seg_act = reinterpret_cast<int*>(s->buf + L.off_act);
}

// derive the active segment list from the plane flags (host copy) and upload both lists
static int stack_finish_lists(midyn_stack* s) {
    midyn_ctx* ctx = s->ctx;
    s->h_flags.assign(2 * std::max(s->nseg, 1), 0);
    if (s->nseg > 0)
        HIPCHK(ctx, hipMemcpy(s->h_flags.data(), s->flags, 2 * s->nseg * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> all(std::max(s->nseg, 1), 0), act(std::max(s->nseg, 1), 0);
    s->n_act = 0;
    s->h_modes.assign(std::max(s->nseg, 1), 3);
    int um = -1;
    for (int seg = 0; seg < s->nseg; ++seg) {
        all[seg] = seg << 2;
        const int fr = s->h_flags[2 * seg], fi = s->h_flags[2 * seg + 1];
        if (!fr && !fi) continue;            // exactly zero operator: contributes nothing
        int mode = 0;
        if (fr && !fi) mode = 1;             // real only
        if (!fr && fi) mode = 2;             // imaginary only
        s->h_modes[seg] = mode;
        um = (um == -1 || um == mode) ? mode : 3;
        act[s->n_act++] = (seg << 2) | mode;
    }
    s->uniform_mode = um < 0 ? 0 : um;
    s->all_single_plane = s->n_act > 0;
    for (int seg = 0; seg < s->nseg; ++seg)
        if (s->h_modes[seg] == 0) s->all_single_plane = false;
    HIPCHK(ctx, hipMemcpy(s->seg_all, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(s->seg_act, act.data(), act.size() * sizeof(int), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int midyn_stack_create(midyn_ctx* ctx, int n, int k, const midyn_complex* ops,
                                  const midyn_complex* static_op, const double* frame_im,
                                  void* dev_buffer, midyn_stack** out) {
    if (!ctx || !out) return fail(ctx, "midyn_stack_create: NULL ctx/out");
    if (n <= 0 || k < 0) return fail(ctx, "midyn_stack_create: n must be > 0 and k >= 0");
    if (k > 0 && !ops) return fail(ctx, "midyn_stack_create: k > 0 but ops is NULL");
    if (k == 0 && !static_op)
        return fail(ctx, "midyn_stack_create: neither static operator nor operators given");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    midyn_stack* s = new midyn_stack();
    s->ctx = ctx;
    s->n = n;
    s->n_pad = round_up(n, 64);
    s->k = k;
    s->has_static = static_op ? 1 : 0;
    s->has_frame = frame_im ? 1 : 0;
    s->nseg = k + s->has_static;
    PackLayout L = pack_layout(s->n_pad, s->nseg);
    if (dev_buffer) {
        s->buf = static_cast<char*>(dev_buffer);
        s->owns = false;
    } else {
        HIPCHK(ctx, hipMalloc(&s->buf, L.total));
        s->owns = true;
    }
    stack_bind(s);
    HIPCHK(ctx, hipMemsetAsync(s->buf, 0, L.total, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t plane = (size_t)s->n_pad * s->n_pad;
    int seg = 0;
    if (static_op) {
        HIPCHK(ctx, hipMemcpy2D(s->ops, (size_t)s->n_pad * sizeof(double2), static_op,
                                (size_t)n * sizeof(double2), (size_t)n * sizeof(double2), n,
                                hipMemcpyHostToDevice));
        seg = 1;
    }
    for (int j = 0; j < k; ++j, ++seg) {
        const char* src = reinterpret_cast<const char*>(ops) + (size_t)j * n * n * sizeof(double2);
        HIPCHK(ctx, hipMemcpy2D(s->ops + seg * plane, (size_t)s->n_pad * sizeof(double2), src,
                                (size_t)n * sizeof(double2), (size_t)n * sizeof(double2), n,
                                hipMemcpyHostToDevice));
    }
    if (frame_im)
        HIPCHK(ctx, hipMemcpy(s->frame_im, frame_im, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    if (s->nseg > 0) {
        hipLaunchKernelGGL(plane_flags_kernel, dim3(grid_for(plane * s->nseg)), dim3(256), 0, ctx->stream,
                           s->ops, plane, s->nseg, s->flags);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    CHK(stack_finish_lists(s));
    *out = s;
    return 0;
}

extern "C" int midyn_stack_adopt(midyn_ctx* ctx, int n, int k, int has_static, int has_frame,
                                 void* dev_buffer, midyn_stack** out) {
    if (!ctx || !out || !dev_buffer) return fail(ctx, "midyn_stack_adopt: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    midyn_stack* s = new midyn_stack();
    s->ctx = ctx;
    s->n = n;
    s->n_pad = round_up(n, 64);
    s->k = k;
    s->has_static = has_static ? 1 : 0;
    s->has_frame = has_frame ? 1 : 0;
    s->nseg = k + s->has_static;
    s->buf = static_cast<char*>(dev_buffer);
    s->owns = false;
    stack_bind(s);
    CHK(stack_finish_lists(s));
    *out = s;
    return 0;
}

extern "C" int midyn_rk4_plan_destroy(struct midyn_rk4_plan* p);

extern "C" int midyn_stack_destroy(midyn_stack* s) {
    if (!s) return 0;
    hipSetDevice(s->ctx->device);
    hipStreamSynchronize(s->ctx->stream);
    if (s->eval_plan) midyn_rk4_plan_destroy(s->eval_plan);
    s->eval_plan = nullptr;
    if (s->planes) hipFree(s->planes);
    s->planes = nullptr;
    for (int* q : {s->blk_ptr, s->blk_idx, s->gw_ptr[0], s->gw_idx[0], s->gw_ptr[1], s->gw_idx[1], s->gw_ptr[2], s->gw_idx[2],
                   s->gw_ptr[3], s->gw_idx[3]})
        if (q) hipFree(q);
    if (s->owns && s->buf) hipFree(s->buf);
    delete s;
    return 0;
}

extern "C" int midyn_stack_info(midyn_stack* s, long long* info) {
    if (!s || !info) return fail(nullptr, "midyn_stack_info: NULL argument");
    info[0] = s->n;
    info[1] = s->n_pad;
    info[2] = s->k;
    info[3] = s->has_static;
    info[4] = s->has_frame;
    info[5] = s->nseg;
    info[6] = s->n_act;
    info[7] = (long long)(s->bytes >> 20);
    return 0;
}

extern "C" int midyn_stack_segment_modes(midyn_stack* s, int* modes) {
    if (!s || !modes) return fail(nullptr, "midyn_stack_segment_modes: NULL argument");
    for (int i = 0; i < s->nseg; ++i) modes[i] = s->h_modes[i];
    return 0;
}

// -------------------------------------------------------------------------------------------------
// kernel launch helpers
// -------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int BK, int MODE, int MINW = 2, bool SPARSE = false>
static int launch_gemm_mode(midyn_ctx* ctx, const GemmArgs& g) {
    constexpr int THREADS = 64 * WM * WN;
    constexpr size_t SMEM = (size_t)2 * BK * (BM + BN) * sizeof(double2);
    static bool attr_set[16] = {false};
    auto kern = zgemm_seg_kernel<BM, BN, WM, WN, BK, MODE, MINW, SPARSE>;
    if (!attr_set[ctx->device & 15]) {
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        attr_set[ctx->device & 15] = true;
    }
    const int blocks = (g.M / BM) * (g.N / BN) * g.splits;
    hipLaunchKernelGGL(kern, dim3(blocks, g.batch > 1 ? g.batch : 1), dim3(THREADS), SMEM, ctx->stream, g);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// uniform_mode: 0/1/2 when every active segment has that plane mode (straight-line specialised
// kernel), 3 when the stack is mixed (per-segment run-time flags)
template <int BM, int BN, int WM, int WN, int BK>
static int launch_gemm_cfg(midyn_ctx* ctx, const GemmArgs& g, int uniform_mode) {
    if (g.work_ptr) {  // block-sparse stack: tile lists instead of the full (K tile, segment) loop
        switch (uniform_mode) {
            case 1: return launch_gemm_mode<BM, BN, WM, WN, BK, 1, 2, true>(ctx, g);
            case 2: return launch_gemm_mode<BM, BN, WM, WN, BK, 2, 2, true>(ctx, g);
            case 0: return launch_gemm_mode<BM, BN, WM, WN, BK, 0, 2, true>(ctx, g);
            default: return launch_gemm_mode<BM, BN, WM, WN, BK, 3, 2, true>(ctx, g);
        }
    }
    switch (uniform_mode) {
        case 0: return launch_gemm_mode<BM, BN, WM, WN, BK, 0>(ctx, g);
        case 1: return launch_gemm_mode<BM, BN, WM, WN, BK, 1>(ctx, g);
        case 2: return launch_gemm_mode<BM, BN, WM, WN, BK, 2>(ctx, g);
        case 4: return launch_gemm_mode<BM, BN, WM, WN, BK, 4>(ctx, g);
        default: return launch_gemm_mode<BM, BN, WM, WN, BK, 3>(ctx, g);
    }
}

// split-K workspace + bookkeeping (g.splits / g.partial)
static int setup_splits(midyn_ctx* ctx, GemmArgs& g, int splits) {
    g.splits = 1;
    g.partial = nullptr;
    if (splits <= 1) return 0;
    const size_t need = (size_t)splits * g.M * g.N * sizeof(double2);
    if (ctx->splitk_bytes < need) {
        if (ctx->splitk_ws) hipFree(ctx->splitk_ws);
        ctx->splitk_ws = nullptr;
        ctx->splitk_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->splitk_ws, need));
        ctx->splitk_bytes = need;
    }
    g.splits = splits;
    g.partial = static_cast<double2*>(ctx->splitk_ws);
    return 0;
}

// sum the split-K partials and run the epilogue (no-op when the launch was not split)
static int launch_reduce(midyn_ctx* ctx, const GemmArgs& g) {
    if (g.splits <= 1) return 0;
    const int splits = g.splits;
    const dim3 rgrid(grid_for((size_t)g.M * g.N, 2048)), rblock(256);
#define MIDYN_REDUCE(MODE_) \
    hipLaunchKernelGGL(splitk_reduce_kernel<MODE_>, rgrid, rblock, 0, ctx->stream, g.partial, splits, g.M, g.N, g.epi)
    switch (g.epi.mode) {
        case EPI_RHS: MIDYN_REDUCE(EPI_RHS); break;
        case EPI_RK1: MIDYN_REDUCE(EPI_RK1); break;
        case EPI_RK2: MIDYN_REDUCE(EPI_RK2); break;
        case EPI_RK3: MIDYN_REDUCE(EPI_RK3); break;
        case EPI_RK4: MIDYN_REDUCE(EPI_RK4); break;
        case EPI_TAYLOR: MIDYN_REDUCE(EPI_TAYLOR); break;
        case EPI_CHEB: MIDYN_REDUCE(EPI_CHEB); break;
        default: MIDYN_REDUCE(EPI_PLAIN); break;
    }
#undef MIDYN_REDUCE
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// Block-sparse stacks: which tile lists to run on -- 0: 64-row panels (64 x 64 tiles), 1: 128-row panels
// (128 x 128), 2: 32-row panels (32 x 128), 3: 16-row panels (16 x 128).  A tall panel lists every K tile that
// ANY of its 16-row groups touches, so for scattered patterns the short panels execute far fewer zero tiles:
// cfg 5 (n = 4096, 9 operators), 128 instances -- listed tiles 3584 / 3840 / 4096 / 4352 for 128 / 64 / 32 / 16
// rows, i.e. the listed WORK halves with the panel height; microseconds per contraction for 128 / 512
// instances: 86.5 / 256.7, 55.6 / 152.6, 38.1 / 94.1, 28.2 / 77.0.  Time per unit of listed density relative to
// the 128 x 128 tile (from those runs): 64 rows 1.2, 32 rows 1.55, 16 rows 2.15 -- the weights below, so that
// block-dense patterns keep the big tile.
static int sparse_tile(const midyn_ctx* ctx, const midyn_stack* s, int M, int N) {
    // (the 32- and 16-row panels run 128 columns wide, or 64 wide when the state block is not a multiple of 128)
    const bool ok[4] = {true, M % 128 == 0 && N % 128 == 0 && s->gw_ptr[1] != nullptr,
                        M % 32 == 0 && s->gw_ptr[2] != nullptr, s->gw_ptr[3] != nullptr};
    if (ctx->sparse_bm == 64 || ctx->force_tile == 64) return 0;
    if (ctx->sparse_bm == 128 && ok[1]) return 1;
    if (ctx->sparse_bm == 32 && ok[2]) return 2;
    if (ctx->sparse_bm == 16 && ok[3]) return 3;
    static const double weight[4] = {1.2, 1.0, 1.55, 2.15};
    int best = 0;
    for (int t = 1; t < 4; ++t)
        if (ok[t] && s->gw_density[t] * weight[t] < s->gw_density[best] * weight[best]) best = t;
    return best;
}

// tile choice: 128x128 (8 waves) when that still gives >= 1 block per CU, else 64x64 (4 waves)
static int launch_gemm(midyn_ctx* ctx, const GemmArgs& g_in, int cls, int uniform_mode = 0,
                       const midyn_stack* sparse = nullptr) {
    const GemmArgs& g0 = g_in;
    if (g0.M % 64 || g0.N % 64 || g0.K % GEMM_BK)
        return fail(ctx, "launch_gemm: dimensions must be padded to 64/64/16");
    if (g0.n_act > 64)
        return fail(ctx, "more than 64 non-zero operator segments are not supported by the MFMA contraction yet");
    GemmArgs g = g_in;
    if (uniform_mode == 0 && ctx->complex_3m && !sparse) uniform_mode = 4;  // dense complex: 3 real MFMAs per product
    g.work_ptr = g.work_idx = nullptr;
    g.ablate = ctx->ablate;
    g.splits = 1;
    g.partial = nullptr;
    ProfScope ps(ctx, cls);
    // ---- tile / split choice -----------------------------------------------------------------
    //  * 128x128x16 (8 waves, 1 workgroup per CU) whenever M and N are multiples of 128, else 64x64x16;
    //  * fewer tiles than CUs: split the K loop over `splits` workgroups per tile (partials in a
    //    workspace, summed + epilogue in splitk_reduce_kernel) so that the whole chip contracts.
    //  force_tile (64 | 128 | 12864) pins the tile for A/B runs; 12864 = 128x64x8 two-per-CU variant.
    const int KT = g.K / GEMM_BK;
    // `fill`: workgroups per CU the split may create (the 64-tile keeps two workgroups per CU busy)
    auto best_splits = [&](long long tiles, int fill) {
        int sp = 1;
        if (g.batch > 1 || g.batch_offs) return 1;  // the batch dimension already fills the chip
        const long long cap = (long long)fill * ctx->num_cu;
        if (ctx->split_k && tiles < cap)
            while ((long long)sp * 2 * tiles <= cap && KT % (sp * 2) == 0 && KT / (sp * 2) >= 2) sp *= 2;
        if (ctx->force_splits > 0 && KT % ctx->force_splits == 0) sp = ctx->force_splits;
        return sp;
    };
    const bool can128 = (g.M % 128 == 0) && (g.N % 128 == 0) && ctx->force_tile != 64;
    const long long tiles128 = can128 ? (long long)(g.M / 128) * (g.N / 128) : 0;
    const long long tiles64 = (long long)(g.M / 64) * (g.N / 64);
    if (ctx->force_tile == 12864 && g.M % 128 == 0) return launch_gemm_cfg<128, 64, 2, 2, 8>(ctx, g, uniform_mode);
    bool t128;
    if (ctx->force_tile == 128 && can128) t128 = true;
    else if (ctx->force_tile == 64) t128 = false;
    else if (uniform_mode == 4) t128 = false;  // 3M: three accumulator sets only fit the 32x32 wave tile
    else t128 = can128;  // measured: 128-tile + split-K beats 64-tile without split (n=1024: 49.9 vs 46.4 TF)
    // (tried: 64-tiles with two workgroups per CU for narrow state blocks -- n = 1024: 128 columns 92 vs 99 us,
    //  but n = 4096, 128 columns 41.1 vs 37.1 ms: not a rule)
    int splits = best_splits(t128 ? tiles128 : tiles64, 1);
    if (sparse) {
        // the list of a row panel is shared out by COUNT: one workgroup per CU at most, a share keeps >= 4 tiles
        // (measured on 128-row panels, n = 4096, 112 tiles per panel: 32 panels x 8 splits 87 us, x 16 107 us, x 4 133 us)
        const int t = sparse_tile(ctx, sparse, g.M, g.N);
        const int wide = g.N % 128 == 0 ? 128 : 64;
        const int bm_of[4] = {64, 128, 32, 16}, bn_of[4] = {64, 128, wide, wide};
        g.work_ptr = sparse->gw_ptr[t];
        g.work_idx = sparse->gw_idx[t];
        const long long tiles = (long long)(g.M / bm_of[t]) * (g.N / bn_of[t]);
        splits = 1;
        if (ctx->split_k)
            while ((long long)splits * 2 * tiles <= ctx->num_cu && sparse->gw_avg[t] / (splits * 2) >= 4.0) splits *= 2;
        if (ctx->force_splits > 0) splits = ctx->force_splits;
        CHK(setup_splits(ctx, g, splits));
        int sts;
        if (t == 1) sts = launch_gemm_cfg<128, 128, 2, 4, 16>(ctx, g, uniform_mode);
        else if (t == 2 && wide == 128) sts = launch_gemm_cfg<32, 128, 1, 4, 16>(ctx, g, uniform_mode);
        else if (t == 2) sts = launch_gemm_cfg<32, 64, 1, 2, 16>(ctx, g, uniform_mode);
        else if (t == 3 && wide == 128) sts = launch_gemm_cfg<16, 128, 1, 4, 16>(ctx, g, uniform_mode);
        else if (t == 3) sts = launch_gemm_cfg<16, 64, 1, 2, 16>(ctx, g, uniform_mode);
        else sts = launch_gemm_cfg<64, 64, 2, 2, 16>(ctx, g, uniform_mode);
        if (sts) return sts;
        return launch_reduce(ctx, g);
    }
    CHK(setup_splits(ctx, g, splits));
    int st = t128 ? launch_gemm_cfg<128, 128, 2, 4, 16>(ctx, g, uniform_mode)
                  : launch_gemm_cfg<64, 64, 2, 2, 16>(ctx, g, uniform_mode);
    if (st) return st;
    return launch_reduce(ctx, g);
}

// The batched RHS contraction on the planar single-plane kernel (two operator tiles per barrier).
static int launch_gemm_plane(midyn_ctx* ctx, const GemmArgs& g_in, const double* planes, long long seg_stride) {
    GemmArgs g = g_in;
    g.ablate = 0;
    g.splits = 1;
    g.partial = nullptr;
    if (g.M % 128 || g.N % 128 || g.K % GEMM_BK || g.n_act > 64 || g.n_act < 1)
        return fail(ctx, "launch_gemm_plane: unsupported shape");
    ProfScope ps(ctx, KC_RHS_GEMM);
    const long long tiles = (long long)(g.M / 128) * (g.N / 128);
    const int KT = g.K / GEMM_BK;
    int splits = 1;
    if (ctx->split_k && tiles < ctx->num_cu)
        while ((long long)splits * 2 * tiles <= ctx->num_cu && KT % (splits * 2) == 0 && KT / (splits * 2) >= 2) splits *= 2;
    if (ctx->force_splits > 0 && KT % ctx->force_splits == 0) splits = ctx->force_splits;
    CHK(setup_splits(ctx, g, splits));
    constexpr size_t SMEM = (size_t)2 * 2 * 128 * 16 * sizeof(double) + (size_t)2 * 16 * 128 * sizeof(double2);
    static bool attr_set[16] = {false};
    auto kern = zgemm_plane_kernel<2, 4>;
    if (!attr_set[ctx->device & 15]) {
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM));
        attr_set[ctx->device & 15] = true;
    }
    PlaneArgs pa{};
    pa.planes = planes;
    pa.seg_stride = seg_stride;
    pa.g = g;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * splits)), dim3(512), SMEM, ctx->stream, pa);
    HIPCHK(ctx, hipGetLastError());
    return launch_reduce(ctx, g);
}

static int launch_stream(midyn_ctx* ctx, const StreamArgs& a, const double* planes = nullptr) {
    ProfScope ps(ctx, KC_STREAM);
    if (planes) {  // single-plane stack: stream only the non-zero planes (half the bytes)
        if (a.n_pad >= 1024) {
            switch (ctx->stream_variant) {
                case 1: hipLaunchKernelGGL((rhs_stream_plane_kernel<2, 1>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a, planes); break;
                case 2: hipLaunchKernelGGL((rhs_stream_plane_kernel<1, 3>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a, planes); break;
                case 3: hipLaunchKernelGGL((rhs_stream_plane_kernel<2, 9>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a, planes); break;
                case 4: hipLaunchKernelGGL((rhs_stream_plane_kernel<1, 9>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a, planes); break;
                default: hipLaunchKernelGGL((rhs_stream_plane_kernel<2, 3>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a, planes); break;
            }
        } else {
            hipLaunchKernelGGL((rhs_stream_plane_kernel<1, 1>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a, planes);
        }
        HIPCHK(ctx, hipGetLastError());
        return 0;
    }
    if (a.n_pad >= 1024) {
        switch (ctx->stream_variant) {
            case 1: hipLaunchKernelGGL((rhs_stream_kernel<4, 1>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a); break;
            case 2: hipLaunchKernelGGL((rhs_stream_kernel<2, 3>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a); break;
            case 3: hipLaunchKernelGGL((rhs_stream_kernel<4, 9>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a); break;
            case 4: hipLaunchKernelGGL((rhs_stream_kernel<2, 9>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a); break;
            default: hipLaunchKernelGGL((rhs_stream_kernel<4, 3>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a); break;
        }
    } else {
        hipLaunchKernelGGL((rhs_stream_kernel<1, 1>), dim3(a.n_pad), dim3(256), 0, ctx->stream, a);
    }
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

static int launch_stream_multi(midyn_ctx* ctx, const StreamArgs& a, int ncol, int m_cols, long long inst_stride,
                               const double* planes = nullptr) {
    ProfScope ps(ctx, KC_STREAM);
    const dim3 grid(a.n_pad), block(256);
    if (planes) {  // single-plane stack: stream only the non-zero planes
#define MIDYN_MULTI_P(C_)                                                                                       \
    if (a.n_pad >= 1024)                                                                                        \
        hipLaunchKernelGGL((rhs_stream_multi_plane_kernel<C_, 2>), grid, block, 0, ctx->stream, a, planes, ncol, m_cols, \
                           inst_stride);                                                                        \
    else                                                                                                        \
        hipLaunchKernelGGL((rhs_stream_multi_plane_kernel<C_, 1>), grid, block, 0, ctx->stream, a, planes, ncol, m_cols, \
                           inst_stride)
        if (ncol <= 2) { MIDYN_MULTI_P(2); }
        else if (ncol <= 4) { MIDYN_MULTI_P(4); }
        else { MIDYN_MULTI_P(8); }
#undef MIDYN_MULTI_P
        HIPCHK(ctx, hipGetLastError());
        return 0;
    }
#define MIDYN_MULTI(C_)                                                                                        \
    if (a.n_pad >= 512)                                                                                        \
        hipLaunchKernelGGL((rhs_stream_multi_kernel<C_, 2>), grid, block, 0, ctx->stream, a, ncol, m_cols, inst_stride); \
    else                                                                                                       \
        hipLaunchKernelGGL((rhs_stream_multi_kernel<C_, 1>), grid, block, 0, ctx->stream, a, ncol, m_cols, inst_stride)
    if (ncol <= 2) { MIDYN_MULTI(2); }
    else if (ncol <= 4) { MIDYN_MULTI(4); }
    else { MIDYN_MULTI(8); }
#undef MIDYN_MULTI
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// block-sparse stack, 1..8 columns: only the listed 16 x 16 operator blocks are read
static int launch_blocks(midyn_ctx* ctx, const StreamArgs& a, const midyn_stack* s, int ncol, int m_cols,
                         long long inst_stride) {
    ProfScope ps(ctx, KC_BLOCKS);
    const dim3 grid(a.n_pad / 16), block(256);
#define MIDYN_BLOCKS(C_) \
    hipLaunchKernelGGL((rhs_blocks_kernel<C_>), grid, block, 0, ctx->stream, a, s->blk_ptr, s->blk_idx, ncol, m_cols, inst_stride)
    if (ncol <= 1) { MIDYN_BLOCKS(1); }
    else if (ncol <= 2) { MIDYN_BLOCKS(2); }
    else if (ncol <= 4) { MIDYN_BLOCKS(4); }
    else { MIDYN_BLOCKS(8); }
#undef MIDYN_BLOCKS
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// plain zgemm on device buffers: C = alpha * A.B + beta * Z; `batch` independent problems whose
// operands are `sa`, `sb`, `sc` elements apart (C and Z share the stride)
static int dev_zgemm_batched(midyn_ctx* ctx, int batch, int M, int N, int K, const double2* A, int lda, long long sa,
                             const double2* B, int ldb, long long sb, double2* C, int ldc, long long sc, double alpha,
                             double beta, const double2* Z, const long long* d_offs = nullptr,
                             bool a_real_only = false) {
    GemmArgs g{};
    g.batch_offs = d_offs;
    g.A = A;
    g.a_seg_stride = 0;
    g.lda = lda;
    g.B = B;
    g.ldb = ldb;
    g.M = M;
    g.N = N;
    g.K = K;
    g.seg_list = ctx->d_one_seg;
    g.n_act = 1;
    g.has_static = 0;
    g.coeff = nullptr;
    g.inst_stride = 0;
    g.m_cols = 1;
    g.n_inst = N;
    g.batch = batch;
    g.batch_a = sa;
    g.batch_b = sb;
    g.batch_c = sc;
    g.epi.mode = EPI_PLAIN;
    g.epi.ld = ldc;
    g.epi.alpha = alpha;
    g.epi.beta = beta;
    g.epi.out = C;
    g.epi.z = Z;
    if (a_real_only) {  // Im A == 0 exactly: the two real MFMAs that would multiply it are skipped
        g.seg_list = ctx->d_one_seg + 1;
        return launch_gemm(ctx, g, KC_ZGEMM, 1);
    }
    return launch_gemm(ctx, g, KC_ZGEMM);
}

static int dev_zgemm(midyn_ctx* ctx, int M, int N, int K, const double2* A, int lda, const double2* B,
                     int ldb, double2* C, int ldc, double alpha, double beta, const double2* Z) {
    return dev_zgemm_batched(ctx, 1, M, N, K, A, lda, 0, B, ldb, 0, C, ldc, 0, alpha, beta, Z);
}

// square [np][np] matrices laid out back to back
static int dev_sqgemm(midyn_ctx* ctx, int batch, int np, const double2* A, const double2* B, double2* C, double alpha,
                      double beta, const double2* Z) {
    const long long st = (long long)np * np;
    return dev_zgemm_batched(ctx, batch, np, np, np, A, np, st, B, np, st, C, np, st, alpha, beta, Z);
}

static int dev_lincomb(midyn_ctx* ctx, int n, double2* out, int nterms, const double2* const* xs,
                       const double* alphas, double gamma, int batch = 1) {
    LinArgs a{};
    a.nterms = nterms;
    for (int i = 0; i < nterms; ++i) {
        a.x[i] = xs[i];
        a.alpha[i] = alphas[i];
    }
    a.gamma = gamma;
    a.n = n;
    a.batch = batch;
    a.out = out;
    ProfScope ps(ctx, KC_ELEM);
    hipLaunchKernelGGL(lincomb_kernel, dim3(grid_for((size_t)n * n * batch)), dim3(256), 0, ctx->stream, a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

// simple device buffer holder
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;      // requested size
    size_t cap = 0;        // capacity of the block (>= bytes)
    midyn_ctx* owner = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (!p) return;
        if (owner && cap <= midyn_ctx::POOL_MAX_BLOCK && owner->mem_pool_bytes + cap <= midyn_ctx::POOL_MAX_BYTES) {
            owner->mem_pool.emplace_back(cap, p);
            owner->mem_pool_bytes += cap;
        } else {
            hipFree(p);
        }
        p = nullptr;
        bytes = cap = 0;
    }
    int alloc(midyn_ctx* ctx, size_t b) {
        release();
        bytes = b;
        owner = ctx;
        if (b == 0) return 0;
        const size_t want = (b + 4095) / 4096 * 4096;
        // best fit among cached blocks of capacity in [want, 2 want]
        int best = -1;
        for (int i = 0; i < (int)ctx->mem_pool.size(); ++i) {
            const size_t c = ctx->mem_pool[i].first;
            if (c >= want && c <= 2 * want && (best < 0 || c < ctx->mem_pool[best].first)) best = i;
        }
        if (best >= 0) {
            // a recycled block may still be read by kernels queued before it was released, and the
            // caller may fill it with a host-synchronous (null-stream) copy: drain the stream first
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            cap = ctx->mem_pool[best].first;
            p = ctx->mem_pool[best].second;
            ctx->mem_pool_bytes -= cap;
            ctx->mem_pool.erase(ctx->mem_pool.begin() + best);
            return 0;
        }
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess && !ctx->mem_pool.empty()) {  // out of memory: give the cache back and retry
            (void)hipGetLastError();
            for (auto& blk : ctx->mem_pool) hipFree(blk.second);
            ctx->mem_pool.clear();
            ctx->mem_pool_bytes = 0;
            e = hipMalloc(&p, want);
        }
        if (e != hipSuccess) {
            p = nullptr;
            bytes = 0;
            return fail(ctx, std::string("hipMalloc: ") + hipGetErrorString(e));
        }
        cap = want;
        return 0;
    }
    template <class T>
    T* as() { return static_cast<T*>(p); }
};

// Copy `bytes` from a host OR device pointer into device memory, ordered on the ctx stream and
// complete on return.  (A plain hipMemcpy from a DEVICE source may return before the copy has
// run, and it runs on the null stream, which the non-blocking ctx stream does not wait for: kernels
// launched next would read a half-filled table.)
static hipError_t copy_to_device_any(midyn_ctx* ctx, void* dst, const void* src, size_t bytes) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    return e;
}

// 1-norms of `batch` [np][np] matrices stored back to back (one small D2H copy + stream sync)
// mode 0: 1-norms; 1: infinity norms; 2: 1-norms of the Hermitian parts (A + A^dagger) / 2
static int dev_norm1(midyn_ctx* ctx, const double2* A, int np, int batch, DevBuf& scratch, std::vector<double>& norms,
                     int mode = 0) {
    const int nchunk = std::max(1, std::min(32, np / 128));
    const size_t cnt = (size_t)batch * nchunk * np;
    if (scratch.bytes < cnt * sizeof(double)) CHK(scratch.alloc(ctx, cnt * sizeof(double)));
    if (mode == 0)
        hipLaunchKernelGGL(colsum_kernel, dim3((np + 255) / 256, batch, nchunk), dim3(256), 0, ctx->stream, A, np, nchunk,
                           scratch.as<double>());
    else
        hipLaunchKernelGGL(colsum_mode_kernel, dim3((np + 255) / 256, batch, nchunk), dim3(256), 0, ctx->stream, A, np,
                           nchunk, mode, scratch.as<double>());
    HIPCHK(ctx, hipGetLastError());
    std::vector<double> h_pageable;
    double* h = ctx->h_pinned;
    if (cnt > midyn_ctx::PINNED_DOUBLES) {
        h_pageable.resize(cnt);
        h = h_pageable.data();
    }
    HIPCHK(ctx, hipMemcpyAsync(h, scratch.p, cnt * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    norms.assign(batch, 0.0);
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < np; ++c) {
            double sum = 0.0;
            for (int z = 0; z < nchunk; ++z) sum += h[((size_t)b * nchunk + z) * np + c];
            norms[b] = std::max(norms[b], sum);
        }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// single evaluations
// -------------------------------------------------------------------------------------------------
static const int* stack_seg_list(midyn_stack* s, int* n_act) {
    if (s->ctx->skip_zero_planes) {
        *n_act = s->n_act;
        return s->seg_act;
    }
    *n_act = s->nseg;
    return s->seg_all;
}

static int launch_gen_eval(midyn_stack* s, const double* d_coeff, const double2* d_e, double scale,
                           double2* d_out, int batch = 1, long long coeff_stride = 0, long long e_stride = 0,
                           const double* scale_vec = nullptr) {
    midyn_ctx* ctx = s->ctx;
    GenArgs a{};
    a.batch = batch;
    a.coeff_stride = coeff_stride;
    a.e_stride = e_stride;
    a.scale_vec = scale_vec;
    a.ops = s->ops;
    a.seg_list = stack_seg_list(s, &a.n_act);
    a.n_pad = s->n_pad;
    a.has_static = s->has_static;
    a.coeff = d_coeff;
    a.e = d_e;
    a.scale = scale;
    a.out = d_out;
    ProfScope ps(ctx, KC_GEN);
    hipLaunchKernelGGL(gen_eval_kernel, dim3(grid_for((size_t)s->n_pad * s->n_pad * batch, 8192)), dim3(256), 0,
                       ctx->stream, a);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

static int make_phase_rows(midyn_stack* s, const double* h_times, int rows, DevBuf& d_times, DevBuf& d_E) {
    midyn_ctx* ctx = s->ctx;
    if (!s->has_frame) return 0;
    CHK(d_times.alloc(ctx, (size_t)rows * sizeof(double)));
    CHK(d_E.alloc(ctx, (size_t)rows * s->n_pad * sizeof(double2)));
    HIPCHK(ctx, hipMemcpyAsync(d_times.p, h_times, (size_t)rows * sizeof(double), hipMemcpyHostToDevice,
                               ctx->stream));
    hipLaunchKernelGGL(phase_table_kernel, dim3(grid_for((size_t)rows * s->n_pad)), dim3(256), 0,
                       ctx->stream, s->frame_im, d_times.as<double>(), s->n_pad, rows, d_E.as<double2>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // h_times may be a stack temporary
    return 0;
}

// -------------------------------------------------------------------------------------------------
// row f1: coefficient table evaluated on the device
// -------------------------------------------------------------------------------------------------
struct midyn_sigtable {
    midyn_ctx* ctx = nullptr;
    int B = 0, R = 0, k = 0;
    DevBuf d_S;
};

extern "C" int midyn_sigtable_create(midyn_ctx* ctx, int B, int k, int R, const double* times,
                                     const long long* term_ptr, const double* term_params,
                                     const long long* sample_ptr, const midyn_complex* samples,
                                     midyn_sigtable** out) {
    if (!ctx || !out) return fail(ctx, "midyn_sigtable_create: NULL argument");
    if (B <= 0 || k <= 0 || R <= 0) return fail(ctx, "midyn_sigtable_create: bad sizes");
    if (!times || !term_ptr || !term_params || !sample_ptr || !samples)
        return fail(ctx, "midyn_sigtable_create: NULL argument");
    const size_t nsig = (size_t)B * k;
    const long long n_terms = term_ptr[nsig];
    if (term_ptr[0] != 0 || n_terms < 0) return fail(ctx, "midyn_sigtable_create: term_ptr must start at 0");
    for (size_t i = 0; i < nsig; ++i)
        if (term_ptr[i + 1] < term_ptr[i]) return fail(ctx, "midyn_sigtable_create: term_ptr must be non-decreasing");
    long long n_samples = 0;
    for (long long q = 0; q < n_terms; ++q) {
        // terms may SHARE sample ranges: sample_ptr is [n_terms][2] = (offset, length)
        const long long off = sample_ptr[2 * q], len = sample_ptr[2 * q + 1];
        if (off < 0 || len < 0) return fail(ctx, "midyn_sigtable_create: negative sample range");
        if (term_params[4 * q] == 0.0 && len < 1)
            return fail(ctx, "midyn_sigtable_create: a constant term needs one sample");
        n_samples = std::max(n_samples, off + len);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    midyn_sigtable* tab = new midyn_sigtable();
    tab->ctx = ctx;
    tab->B = B;
    tab->R = R;
    tab->k = k;
    DevBuf d_times, d_ptr, d_par, d_sp, d_smp;
    std::vector<long long> sp((size_t)n_terms + 1, 0);
    int st = 0;
    auto guard = [&](int r) { if (r && !st) st = r; };
    guard(tab->d_S.alloc(ctx, (size_t)B * R * k * sizeof(double)));
    guard(d_times.alloc(ctx, (size_t)R * sizeof(double)));
    guard(d_ptr.alloc(ctx, (nsig + 1) * sizeof(long long)));
    guard(d_par.alloc(ctx, std::max<size_t>(1, (size_t)n_terms) * 4 * sizeof(double)));
    guard(d_sp.alloc(ctx, std::max<size_t>(1, (size_t)n_terms) * 2 * sizeof(long long)));
    guard(d_smp.alloc(ctx, std::max<size_t>(1, (size_t)n_samples) * sizeof(double2)));
    auto cp = [&](void* d, const void* h, size_t bytes) {
        if (st || bytes == 0) return;
        hipError_t e = hipMemcpy(d, h, bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) st = fail(ctx, std::string("midyn_sigtable_create upload: ") + hipGetErrorString(e));
    };
    cp(d_times.p, times, (size_t)R * sizeof(double));
    cp(d_ptr.p, term_ptr, (nsig + 1) * sizeof(long long));
    cp(d_par.p, term_params, (size_t)n_terms * 4 * sizeof(double));
    cp(d_sp.p, sample_ptr, (size_t)n_terms * 2 * sizeof(long long));
    cp(d_smp.p, samples, (size_t)n_samples * sizeof(double2));
    if (!st) {
        SigTableArgs a{};
        a.B = B;
        a.R = R;
        a.k = k;
        a.times = d_times.as<double>();
        a.term_ptr = d_ptr.as<long long>();
        a.params = d_par.as<double>();
        a.sample_ptr = d_sp.as<long long>();
        a.samples = d_smp.as<double2>();
        a.S = tab->d_S.as<double>();
        {
            ProfScope ps(ctx, KC_ELEM);
            hipLaunchKernelGGL(signal_table_kernel, dim3(grid_for((size_t)B * R * k, 16384)), dim3(256), 0, ctx->stream,
                               a);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) st = fail(ctx, std::string("midyn_sigtable_create: ") + hipGetErrorString(e));
    }
    if (st) {
        delete tab;
        return st;
    }
    *out = tab;
    return 0;
}

extern "C" int midyn_sigtable_data(midyn_sigtable* tab, const double** dev_S, long long* dims) {
    if (!tab || !dev_S) return fail(tab ? tab->ctx : nullptr, "midyn_sigtable_data: NULL argument");
    *dev_S = tab->d_S.as<double>();
    if (dims) {
        dims[0] = tab->B;
        dims[1] = tab->R;
        dims[2] = tab->k;
    }
    return 0;
}

extern "C" int midyn_sigtable_fetch(midyn_sigtable* tab, double* S_out) {
    if (!tab || !S_out) return fail(tab ? tab->ctx : nullptr, "midyn_sigtable_fetch: NULL argument");
    HIPCHK(tab->ctx, hipSetDevice(tab->ctx->device));
    HIPCHK(tab->ctx, hipMemcpy(S_out, tab->d_S.p, tab->d_S.bytes, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int midyn_sigtable_destroy(midyn_sigtable* tab) {
    if (!tab) return 0;
    hipSetDevice(tab->ctx->device);
    delete tab;
    return 0;
}

extern "C" int midyn_eval_generator(midyn_stack* s, const double* coeffs, double t, midyn_complex* G_out) {
    if (!s || !G_out) return fail(s ? s->ctx : nullptr, "midyn_eval_generator: NULL argument");
    midyn_ctx* ctx = s->ctx;
    if (s->k > 0 && !coeffs) return fail(ctx, "midyn_eval_generator: coeffs is NULL but the stack has operators");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf d_coeff, d_times, d_E, d_G;
    CHK(d_coeff.alloc(ctx, std::max(1, s->k) * sizeof(double)));
    if (s->k > 0)
        HIPCHK(ctx, hipMemcpy(d_coeff.p, coeffs, s->k * sizeof(double), hipMemcpyHostToDevice));
    CHK(make_phase_rows(s, &t, 1, d_times, d_E));
    CHK(d_G.alloc(ctx, (size_t)s->n_pad * s->n_pad * sizeof(double2)));
    CHK(launch_gen_eval(s, d_coeff.as<double>(), s->has_frame ? d_E.as<double2>() : nullptr, 1.0,
                        d_G.as<double2>()));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy2D(G_out, (size_t)s->n * sizeof(double2), d_G.p, (size_t)s->n_pad * sizeof(double2),
                            (size_t)s->n * sizeof(double2), s->n, hipMemcpyDeviceToHost));
    return 0;
}

// -------------------------------------------------------------------------------------------------
// RK4 plan: all device state of a batched fixed-step RK4 solve
// -------------------------------------------------------------------------------------------------
struct midyn_rk4_plan {
    midyn_stack* stack = nullptr;
    int B = 0, m = 0, ncol = 0, ld = 0, R = 0, nsteps = 0, P = 0;
    bool stream_path = false;
    DevBuf d_S, d_times, d_E, d_y, d_acc, d_yin[2], d_out, d_tmp, d_G, d_eval_out, d_eval_tmp;
    bool combine_first = false;  // one instance, many columns: form C(t) once, then ONE n^3 zgemm
    bool multi_stream = false;   // 2..8 columns, large n: multi-column streaming kernel instead of a padded MFMA tile
    bool blocks = false;         // block-sparse stack: work-list kernels (rhs_blocks_kernel / SPARSE zgemm_seg_kernel)
    std::vector<int> rows;     // [nsteps][3]
    std::vector<double> hs;    // [nsteps]
    std::vector<int> save;     // [nsteps] or empty
    int cur_yin = 0;           // which yin buffer holds the input of the next stage-1
    int next_step = 0;         // next step expected (state continuity)
    bool tiny = false;         // small system: the whole step loop runs inside tiny_rk4_kernel
    size_t tiny_smem = 0;
    DevBuf d_rows, d_hs, d_save;
};

// Block occupancy of the stack (once per stack): the 16 x 16 map of every segment comes from
// block_map_kernel; the host turns it into the work lists of rhs_blocks_kernel (per 16 rows) and of the
// SPARSE zgemm_seg_kernel (per row panel of 64 / 128 rows, K tile outer / segment inner).
static int stack_block_lists(midyn_stack* s) {
    if (s->blk_state) return 0;
    midyn_ctx* ctx = s->ctx;
    s->blk_state = -1;
    const int np = s->n_pad, nb = np / 16;
    if (s->n_act < 1 || s->nseg > 64 || np < 256 || nb > 0xffff) return 0;
    const size_t map_bytes = (size_t)s->nseg * nb * nb;
    DevBuf d_map;
    CHK(d_map.alloc(ctx, map_bytes));
    HIPCHK(ctx, hipMemsetAsync(d_map.p, 0, map_bytes, ctx->stream));
    hipLaunchKernelGGL(block_map_kernel, dim3(nb, s->nseg), dim3(256), 0, ctx->stream, s->ops, np,
                       d_map.as<unsigned char>());
    HIPCHK(ctx, hipGetLastError());
    std::vector<unsigned char> map(map_bytes);
    HIPCHK(ctx, hipMemcpyAsync(map.data(), d_map.p, map_bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<int> act;  // packed (seg << 2 | mode), the order of seg_act
    for (int seg = 0; seg < s->nseg; ++seg)
        if (s->h_modes[seg] != 3) act.push_back((seg << 2) | s->h_modes[seg]);
    size_t nz = 0;
    for (int a : act) {
        const unsigned char* m = map.data() + (size_t)(a >> 2) * nb * nb;
        for (size_t i = 0; i < (size_t)nb * nb; ++i) nz += m[i];
    }
    s->blk_density = (double)nz / ((double)act.size() * nb * nb);
    if (s->blk_density > 0.5) return 0;  // dense enough: the dense kernels are the right ones
    auto upload = [&](const std::vector<int>& h, int** d) -> int {
        HIPCHK(ctx, hipMalloc(d, std::max<size_t>(h.size(), 1) * sizeof(int)));
        if (!h.empty()) HIPCHK(ctx, hipMemcpy(*d, h.data(), h.size() * sizeof(int), hipMemcpyHostToDevice));
        return 0;
    };
    {   // streaming lists: per 16-row group, segment outer / chunk inner
        std::vector<int> ptr(nb + 1, 0), idx;
        idx.reserve(nz);
        for (int rb = 0; rb < nb; ++rb) {
            for (size_t ai = 0; ai < act.size(); ++ai) {
                const unsigned char* m = map.data() + ((size_t)(act[ai] >> 2) * nb + rb) * nb;
                for (int cb = 0; cb < nb; ++cb)
                    if (m[cb]) idx.push_back(((act[ai] >> 2) << 16) | cb);
            }
            ptr[rb + 1] = (int)idx.size();
        }
        CHK(upload(ptr, &s->blk_ptr));
        CHK(upload(idx, &s->blk_idx));
    }
    for (int t = 0; t < 4; ++t) {  // MFMA tile lists: BM = 64, 128, 32, 16 (K tile = GEMM_BK = 16 columns = one chunk)
        const int BM = t == 0 ? 64 : (t == 1 ? 128 : (t == 2 ? 32 : 16));
        if (np % BM) continue;
        const int panels = np / BM, rpb = BM / 16;
        std::vector<int> ptr(panels + 1, 0), idx;
        for (int pm = 0; pm < panels; ++pm) {
            for (int kt = 0; kt < nb; ++kt)
                for (int a : act) {
                    const unsigned char* m = map.data() + ((size_t)(a >> 2) * nb + (size_t)pm * rpb) * nb + kt;
                    bool any = false;
                    for (int r = 0; r < rpb && !any; ++r) any = m[(size_t)r * nb] != 0;
                    if (any) idx.push_back((kt << 8) | a);
                }
            ptr[pm + 1] = (int)idx.size();
        }
        s->gw_density[t] = (double)idx.size() / ((double)panels * nb * act.size());
        s->gw_avg[t] = (double)idx.size() / panels;
        CHK(upload(ptr, &s->gw_ptr[t]));
        CHK(upload(idx, &s->gw_idx[t]));
    }
    s->blk_state = 1;
    return 0;
}

// planar copy of a single-plane stack: planes[act] = the non-zero plane of active segment `act`
static int stack_planes(midyn_stack* s) {
    if (s->planes) return 0;
    midyn_ctx* ctx = s->ctx;
    const size_t plane = (size_t)s->n_pad * s->n_pad;
    HIPCHK(ctx, hipMalloc(&s->planes, plane * s->n_act * sizeof(double)));
    hipLaunchKernelGGL(extract_planes_kernel, dim3(grid_for(plane * s->n_act)), dim3(256), 0, ctx->stream, s->ops,
                       s->seg_act, s->n_act, plane, s->planes);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}

static const double2* plan_E(midyn_rk4_plan* p, int row) {
    if (!p->stack->has_frame) return nullptr;
    return p->d_E.as<double2>() + (size_t)row * p->stack->n_pad;
}

// e_in (block-sparse streaming route only): `yin` is NOT pre-phased, the kernel applies this phase row on load
static int plan_rhs_launch(midyn_rk4_plan* p, int row, const Epilogue& epi, const double2* yin,
                           const double2* e_in = nullptr) {
    midyn_stack* s = p->stack;
    midyn_ctx* ctx = s->ctx;
    if (e_in && !(p->blocks && p->ncol <= 8)) return fail(ctx, "plan_rhs_launch: fused input phase needs the block route");
    if (p->blocks && p->ncol <= 8) {
        StreamArgs a{};
        a.ops = s->ops;
        a.seg_list = nullptr;
        a.n_act = s->nseg;  // the block lists carry segment numbers: coefficients are staged for every segment
        a.n_pad = s->n_pad;
        a.has_static = s->has_static;
        a.coeff = s->k > 0 ? p->d_S.as<double>() + (size_t)row * s->k : nullptr;
        a.yin = yin;
        a.e_in = e_in;
        a.epi = epi;
        return launch_blocks(ctx, a, s, p->ncol, p->m, (long long)p->R * s->k);
    }
    if (p->stream_path) {
        StreamArgs a{};
        a.ops = s->ops;
        a.seg_list = stack_seg_list(s, &a.n_act);
        a.n_pad = s->n_pad;
        a.has_static = s->has_static;
        a.coeff = s->k > 0 ? p->d_S.as<double>() + (size_t)row * s->k : nullptr;
        a.yin = yin;
        a.epi = epi;
        const double* planes = nullptr;
        if (ctx->skip_zero_planes && ctx->stream_planes && s->all_single_plane) {
            CHK(stack_planes(s));
            planes = s->planes;
        }
        return launch_stream(ctx, a, planes);
    }
    if (p->multi_stream) {
        // 2..8 columns at a size where the padded MFMA tile would mostly multiply zeros
        StreamArgs a{};
        a.ops = s->ops;
        a.seg_list = stack_seg_list(s, &a.n_act);
        a.n_pad = s->n_pad;
        a.has_static = s->has_static;
        a.coeff = s->k > 0 ? p->d_S.as<double>() + (size_t)row * s->k : nullptr;
        a.yin = yin;
        a.epi = epi;
        const double* planes = nullptr;
        if (ctx->skip_zero_planes && ctx->stream_planes && s->all_single_plane) {
            CHK(stack_planes(s));
            planes = s->planes;
        }
        return launch_stream_multi(ctx, a, p->ncol, p->m, (long long)p->R * s->k, planes);
    }
    if (p->combine_first) {
        // All columns share the coefficients (B == 1): C(t) = sum_seg c_seg A_seg costs nseg*n^2
        // element operations, after which the contraction is a single n x n x (m) zgemm instead of
        // nseg of them (unitary / propagator simulations with m ~ n: nseg-fold fewer flops).
        const double* cf = s->k > 0 ? p->d_S.as<double>() + (size_t)row * s->k : nullptr;
        CHK(launch_gen_eval(s, cf, nullptr, 1.0, p->d_G.as<double2>()));
        GemmArgs g{};
        g.A = p->d_G.as<double2>();
        g.a_seg_stride = 0;
        g.lda = s->n_pad;
        g.B = yin;
        g.ldb = p->ld;
        g.M = s->n_pad;
        g.N = p->ld;
        g.K = s->n_pad;
        g.seg_list = ctx->d_one_seg;
        g.n_act = 1;
        g.has_static = 0;
        g.coeff = nullptr;
        g.inst_stride = 0;
        g.m_cols = p->m;
        g.n_inst = 1;
        g.epi = epi;
        const int um = (ctx->skip_zero_planes && (s->uniform_mode == 1 || s->uniform_mode == 2)) ? s->uniform_mode : 0;
        return launch_gemm(ctx, g, KC_RHS_GEMM, um);
    }
    GemmArgs g{};
    g.A = s->ops;
    g.a_seg_stride = (long long)s->n_pad * s->n_pad;
    g.lda = s->n_pad;
    g.B = yin;
    g.ldb = p->ld;
    g.M = s->n_pad;
    g.N = p->ld;
    g.K = s->n_pad;
    g.seg_list = stack_seg_list(s, &g.n_act);
    g.has_static = s->has_static;
    g.coeff = s->k > 0 ? p->d_S.as<double>() + (size_t)row * s->k : nullptr;
    g.inst_stride = (long long)p->R * s->k;
    g.m_cols = p->m;
    g.n_inst = p->B;
    g.epi = epi;
    if (ctx->skip_zero_planes && ctx->plane_kernel && s->all_single_plane && g.M % 128 == 0 && g.N % 128 == 0 &&
        ctx->force_tile == 0) {
        CHK(stack_planes(s));
        return launch_gemm_plane(ctx, g, s->planes, (long long)s->n_pad * s->n_pad);
    }
    return launch_gemm(ctx, g, p->blocks ? KC_BLOCKS_GEMM : KC_RHS_GEMM, ctx->skip_zero_planes ? s->uniform_mode : 0,
                       p->blocks ? s : nullptr);
}

extern "C" int midyn_rk4_plan_destroy(midyn_rk4_plan* p) {
    if (!p) return 0;
    hipSetDevice(p->stack->ctx->device);
    hipStreamSynchronize(p->stack->ctx->stream);
    delete p;
    return 0;
}

static int plan_create_impl(midyn_stack* s, int B, int m, int R, const double* times, const double* S,
                            int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                            int P, const midyn_complex* y0, int y0_shared, midyn_rk4_plan** out) {
    midyn_ctx* ctx = s->ctx;
    if (B <= 0 || m <= 0 || R <= 0 || nsteps < 0) return fail(ctx, "rk4 plan: bad sizes");
    if (!times || !step_rows || !step_h || !y0 || (s->k > 0 && !S)) return fail(ctx, "rk4 plan: NULL argument");
    for (int i = 0; i < 3 * nsteps; ++i)
        if (step_rows[i] < 0 || step_rows[i] >= R) return fail(ctx, "rk4 plan: step_rows out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    midyn_rk4_plan* p = new midyn_rk4_plan();
    p->stack = s;
    p->B = B;
    p->m = m;
    p->ncol = B * m;
    p->stream_path = (p->ncol == 1);
    // column padding: 64, or a multiple of 128 beyond that so that the 128x128 tile (+ split-K) applies
    // (300 instances: 320 columns on 64-tiles 350 us per evaluation, 384 columns on 128-tiles faster)
    p->ld = p->stream_path ? 1 : (p->ncol > 64 ? round_up(p->ncol, 128) : 64);
    p->R = R;
    p->nsteps = nsteps;
    p->P = P;
    p->rows.assign(step_rows, step_rows + 3 * nsteps);
    p->hs.assign(step_h, step_h + nsteps);
    if (step_save) p->save.assign(step_save, step_save + nsteps);
    const size_t state_bytes = (size_t)s->n_pad * p->ld * sizeof(double2);
    int st = 0;
    auto guard = [&](int r) { if (r && !st) st = r; };
    guard(p->d_y.alloc(ctx, state_bytes));
    guard(p->d_acc.alloc(ctx, state_bytes));
    guard(p->d_yin[0].alloc(ctx, state_bytes));
    guard(p->d_yin[1].alloc(ctx, state_bytes));
    if (s->k > 0) guard(p->d_S.alloc(ctx, (size_t)B * R * s->k * sizeof(double)));
    const size_t y0_elems = (size_t)(y0_shared ? 1 : B) * s->n * m;
    guard(p->d_tmp.alloc(ctx, y0_elems * sizeof(double2)));
    if (P > 0) guard(p->d_out.alloc(ctx, (size_t)B * P * s->n * m * sizeof(double2)));
    p->multi_stream = (!p->stream_path && p->ncol <= 8 && s->n_pad >= 256 && s->nseg <= 64 && ctx->multi_stream);
    p->combine_first = (B == 1 && m >= 8 && s->nseg > 1 && ctx->combine_first && !p->multi_stream);
    if (ctx->skip_zero_blocks && ctx->skip_zero_planes && s->n_pad >= 256 && !st) {
        guard(stack_block_lists(s));
        if (s->blk_state == 1) {
            const int t = sparse_tile(ctx, s, s->n_pad, p->ld);
            if (p->ncol <= 8) p->blocks = s->blk_density <= 0.25;
            else p->blocks = s->gw_ptr[t] && s->gw_density[t] <= 0.5 && (ctx->force_tile == 0 || ctx->force_tile == 64);
            // one instance, many columns: forming C(t) first contracts ONE dense operator; the per-segment tile
            // lists only win when they hold less than that in total
            if (p->blocks && p->combine_first && s->gw_density[t] * s->n_act >= 0.8) p->blocks = false;
        }
        if (p->blocks) p->combine_first = false;
    }
    if (p->combine_first) guard(p->d_G.alloc(ctx, (size_t)s->n_pad * s->n_pad * sizeof(double2)));
    if (st) {
        delete p;
        return st;
    }
    auto bail = [&](int r) {
        delete p;
        return r;
    };
    if (s->k > 0) {
        // S may live on the host or on the device (midyn_sigtable_data): hipMemcpyDefault resolves it
        hipError_t e = copy_to_device_any(ctx, p->d_S.p, S, (size_t)B * R * s->k * sizeof(double));
        if (e != hipSuccess) return bail(fail(ctx, std::string("upload S: ") + hipGetErrorString(e)));
    }
    if (int r = make_phase_rows(s, times, R, p->d_times, p->d_E)) return bail(r);
    hipMemsetAsync(p->d_y.p, 0, state_bytes, ctx->stream);
    hipMemsetAsync(p->d_acc.p, 0, state_bytes, ctx->stream);
    hipMemsetAsync(p->d_yin[0].p, 0, state_bytes, ctx->stream);
    hipMemsetAsync(p->d_yin[1].p, 0, state_bytes, ctx->stream);
    {
        hipError_t e = hipMemcpyAsync(p->d_tmp.p, y0, y0_elems * sizeof(double2), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return bail(fail(ctx, std::string("upload y0: ") + hipGetErrorString(e)));
    }
    const int row0 = nsteps > 0 ? p->rows[0] : 0;
    hipLaunchKernelGGL(scatter_state_kernel, dim3(grid_for((size_t)B * s->n * m)), dim3(256), 0, ctx->stream,
                       p->d_tmp.as<double2>(), y0_shared ? 1 : 0, B, s->n, m, p->ld, plan_E(p, row0),
                       p->d_y.as<double2>(), p->d_yin[0].as<double2>());
    if (P > 0)
        hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for((size_t)B * s->n * m)), dim3(256), 0, ctx->stream,
                           p->d_y.as<double2>(), B, s->n, m, p->ld, P, 0, p->d_out.as<double2>());
    {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) return bail(fail(ctx, std::string("rk4 plan init: ") + hipGetErrorString(e)));
    }
    p->cur_yin = 0;
    p->next_step = 0;
    // Small systems (rows fit one wave, active operators fit a 64 KB LDS slice): the step loop of
    // midyn_rk4_plan_run runs inside ONE persistent kernel instead of 4 launches per step.
    {
        int n_act = 0;
        (void)stack_seg_list(s, &n_act);
        const size_t smem = ((size_t)n_act * s->n * s->n + 4 * (size_t)s->n) * sizeof(double2) +
                            (size_t)4 * 2 * 3 * std::max(1, s->k) * sizeof(double) + (size_t)n_act * sizeof(int);
        // n <= 16: always (measured 2-2.5x over the batched stages for 2048-4096 instances; at n = 32 the MFMA
        // path has caught up for large sweeps); up to 32 rows when there are few columns, where the
        // batched path would be ~10 us of launch per stage for almost no work (at 64 rows the one-wave
        // product is LDS-bound and loses: 131 vs 40 ms for 8 instances x 2000 steps).
        // LDS: 64 KB slices (two or more workgroups per CU) for big sweeps, up to 152 KB of the 160 KB when
        // there are at most 1024 columns (<= 256 workgroups: one per CU anyway)
        const size_t smem_max = p->ncol <= 1024 ? (size_t)152 * 1024 : (size_t)64 * 1024;
        if (ctx->tiny_rk4 && s->n <= 32 && (s->n <= 16 || p->ncol <= 64) && n_act >= 1 && smem <= smem_max &&
            s->k <= 42 && nsteps > 0) {
            p->tiny = true;
            p->tiny_smem = smem;
            int st2 = p->d_rows.alloc(ctx, (size_t)3 * nsteps * sizeof(int));
            if (!st2) st2 = p->d_hs.alloc(ctx, (size_t)nsteps * sizeof(double));
            if (!st2 && step_save) st2 = p->d_save.alloc(ctx, (size_t)nsteps * sizeof(int));
            if (st2) return bail(st2);
            hipError_t e = hipMemcpy(p->d_rows.p, step_rows, (size_t)3 * nsteps * sizeof(int), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(p->d_hs.p, step_h, (size_t)nsteps * sizeof(double), hipMemcpyHostToDevice);
            if (e == hipSuccess && step_save)
                e = hipMemcpy(p->d_save.p, step_save, (size_t)nsteps * sizeof(int), hipMemcpyHostToDevice);
            if (e != hipSuccess) return bail(fail(ctx, std::string("rk4 plan (tiny) upload: ") + hipGetErrorString(e)));
        }
    }
    *out = p;
    return 0;
}

static TinyArgs tiny_args(midyn_rk4_plan* p, int step_begin, int step_end) {
    midyn_stack* s = p->stack;
    TinyArgs a{};
    a.ops = s->ops;
    a.seg_list = stack_seg_list(s, &a.n_act);
    a.n = s->n;
    a.n_pad = s->n_pad;
    a.has_static = s->has_static;
    a.k = s->k;
    a.S = s->k > 0 ? p->d_S.as<double>() : nullptr;
    a.inst_stride = (long long)p->R * s->k;
    a.E = s->has_frame ? p->d_E.as<double2>() : nullptr;
    a.rows = p->d_rows.as<int>();
    a.hs = p->d_hs.as<double>();
    a.save = (!p->save.empty() && p->P > 0) ? p->d_save.as<int>() : nullptr;
    a.step_begin = step_begin;
    a.step_end = step_end;
    a.ncol = p->ncol;
    a.m = p->m;
    a.ld = p->ld;
    a.P = p->P;
    a.y = p->d_y.as<double2>();
    a.out = p->P > 0 ? p->d_out.as<double2>() : nullptr;
    return a;
}

extern "C" int midyn_rk4_plan_create(midyn_stack* s, int B, int m, int R, const double* times,
                                     const double* S, int nsteps, const int* step_rows, const double* step_h,
                                     const midyn_complex* y0, int y0_shared, midyn_rk4_plan** out) {
    if (!s || !out) return fail(s ? s->ctx : nullptr, "midyn_rk4_plan_create: NULL argument");
    return plan_create_impl(s, B, m, R, times, S, nsteps, step_rows, step_h, nullptr, 0, y0, y0_shared, out);
}

extern "C" int midyn_rk4_plan_run(midyn_rk4_plan* p, int step_begin, int step_end) {
    if (!p) return fail(nullptr, "midyn_rk4_plan_run: NULL plan");
    midyn_stack* s = p->stack;
    midyn_ctx* ctx = s->ctx;
    if (step_begin < 0 || step_end > p->nsteps || step_begin > step_end)
        return fail(ctx, "midyn_rk4_plan_run: step range out of bounds");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (p->tiny) {
        if (step_begin == step_end) return 0;
        if (!p->save.empty())
            for (int st = step_begin; st < step_end; ++st)
                if (p->save[st] >= p->P && p->P > 0) return fail(ctx, "midyn_rk4_plan_run: save slot out of range");
        TinyArgs a = tiny_args(p, step_begin, step_end);
        static bool attr_set[16] = {false};
        if (!attr_set[ctx->device & 15]) {
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tiny_rk4_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            attr_set[ctx->device & 15] = true;
        }
        {
            ProfScope ps(ctx, KC_STREAM);
            hipLaunchKernelGGL(tiny_rk4_kernel, dim3((p->ncol + 3) / 4), dim3(256), p->tiny_smem, ctx->stream, a);
        }
        HIPCHK(ctx, hipGetLastError());
        p->next_step = step_end;
        return 0;
    }
    for (int st = step_begin; st < step_end; ++st) {
        const int r0 = p->rows[3 * st], r1 = p->rows[3 * st + 1], r2 = p->rows[3 * st + 2];
        if (st != p->next_step) {
            // non-contiguous call: rebuild the pre-phased input from y at this step's start time
            hipLaunchKernelGGL(rephase_kernel, dim3(grid_for((size_t)s->n_pad * p->ld)), dim3(256), 0,
                               ctx->stream, p->d_y.as<double2>(), plan_E(p, r0), s->n_pad, p->ld,
                               p->d_yin[p->cur_yin].as<double2>());
            HIPCHK(ctx, hipGetLastError());
        }
        // the time at which the NEXT stage-1 input must be phased
        const int rnext = (st + 1 < p->nsteps) ? p->rows[3 * (st + 1)] : r2;
        Epilogue e{};
        e.ld = p->ld;
        e.h = p->hs[st];
        e.y = p->d_y.as<double2>();
        e.acc = p->d_acc.as<double2>();
        const int stage_row[4] = {r0, r1, r1, r2};
        const int next_row[4] = {r1, r1, r2, rnext};
        for (int sg = 0; sg < 4; ++sg) {
            e.mode = EPI_RK1 + sg;
            e.e_cur = plan_E(p, stage_row[sg]);
            e.e_next = plan_E(p, next_row[sg]);
            const double2* yin = p->d_yin[p->cur_yin].as<double2>();
            e.yin_next = p->d_yin[p->cur_yin ^ 1].as<double2>();
            CHK(plan_rhs_launch(p, stage_row[sg], e, yin));
            p->cur_yin ^= 1;
        }
        p->next_step = st + 1;
        if (!p->save.empty() && p->save[st] >= 0 && p->P > 0) {
            if (p->save[st] >= p->P) return fail(ctx, "midyn_rk4_plan_run: save slot out of range");
            hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for((size_t)p->B * s->n * p->m)), dim3(256), 0,
                               ctx->stream, p->d_y.as<double2>(), p->B, s->n, p->m, p->ld, p->P, p->save[st],
                               p->d_out.as<double2>());
            HIPCHK(ctx, hipGetLastError());
        }
    }
    return 0;
}

extern "C" int midyn_rk4_plan_fetch(midyn_rk4_plan* p, midyn_complex* Y_out) {
    if (!p || !Y_out) return fail(nullptr, "midyn_rk4_plan_fetch: NULL argument");
    midyn_stack* s = p->stack;
    midyn_ctx* ctx = s->ctx;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    DevBuf tmp;
    const size_t elems = (size_t)p->B * s->n * p->m;
    CHK(tmp.alloc(ctx, elems * sizeof(double2)));
    hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for(elems)), dim3(256), 0, ctx->stream,
                       p->d_y.as<double2>(), p->B, s->n, p->m, p->ld, 1, 0, tmp.as<double2>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(Y_out, tmp.p, elems * sizeof(double2), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int midyn_rk4_solve(midyn_stack* s, int B, int m, int R, const double* times, const double* S,
                               int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                               int P, const midyn_complex* y0, int y0_shared, midyn_complex* Y_out) {
    if (!s || !Y_out) return fail(s ? s->ctx : nullptr, "midyn_rk4_solve: NULL argument");
    if (P < 1) return fail(s->ctx, "midyn_rk4_solve: P must be >= 1 (slot 0 holds y0)");
    midyn_rk4_plan* p = nullptr;
    CHK(plan_create_impl(s, B, m, R, times, S, nsteps, step_rows, step_h, step_save, P, y0, y0_shared, &p));
    int st = midyn_rk4_plan_run(p, 0, nsteps);
    if (!st) {
        hipError_t e = hipStreamSynchronize(s->ctx->stream);
        if (e == hipSuccess)
            e = hipMemcpy(Y_out, p->d_out.p, (size_t)B * P * s->n * m * sizeof(double2), hipMemcpyDeviceToHost);
        if (e != hipSuccess) st = fail(s->ctx, std::string("midyn_rk4_solve: ") + hipGetErrorString(e));
    }
    midyn_rk4_plan_destroy(p);
    return st;
}

extern "C" int midyn_eval_rhs(midyn_stack* s, const double* coeffs, double t, const midyn_complex* y, int m,
                              midyn_complex* out) {
    if (!s || !y || !out) return fail(s ? s->ctx : nullptr, "midyn_eval_rhs: NULL argument");
    midyn_ctx* ctx = s->ctx;
    if (m <= 0) return fail(ctx, "midyn_eval_rhs: m must be positive");
    if (s->k > 0 && !coeffs) return fail(ctx, "midyn_eval_rhs: coeffs is NULL but the stack has operators");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // One instance, m columns, one table row.  The device buffers are cached in the stack (keyed by
    // m) so that a caller evaluating the RHS in a loop (e.g. an adaptive host integrator using the
    // model as a callback) pays kernel launches and two small copies per call, not allocations.
    const size_t elems = (size_t)s->n * m;
    midyn_rk4_plan* p = s->eval_plan;
    if (!p || s->eval_m != m) {
        if (p) midyn_rk4_plan_destroy(p);
        s->eval_plan = nullptr;
        int rows3[3] = {0, 0, 0};
        double h0 = 0.0;
        CHK(plan_create_impl(s, 1, m, 1, &t, coeffs, 1, rows3, &h0, nullptr, 0, y, 1, &p));
        int st = p->d_eval_out.alloc(ctx, (size_t)s->n_pad * p->ld * sizeof(double2));
        if (!st) st = p->d_eval_tmp.alloc(ctx, elems * sizeof(double2));
        if (st) {
            midyn_rk4_plan_destroy(p);
            return st;
        }
        s->eval_plan = p;
        s->eval_m = m;
    } else {
        if (s->k > 0)
            HIPCHK(ctx, hipMemcpyAsync(p->d_S.p, coeffs, s->k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        if (s->has_frame) {
            HIPCHK(ctx, hipMemcpyAsync(p->d_times.p, &t, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(phase_table_kernel, dim3(grid_for(s->n_pad)), dim3(256), 0, ctx->stream, s->frame_im,
                               p->d_times.as<double>(), s->n_pad, 1, p->d_E.as<double2>());
        }
        HIPCHK(ctx, hipMemcpyAsync(p->d_tmp.p, y, elems * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(scatter_state_kernel, dim3(grid_for(elems)), dim3(256), 0, ctx->stream,
                           p->d_tmp.as<double2>(), 1, 1, s->n, m, p->ld, plan_E(p, 0), (double2*)nullptr,
                           p->d_yin[0].as<double2>());
        HIPCHK(ctx, hipGetLastError());
    }
    Epilogue e{};
    e.mode = EPI_RHS;
    e.ld = p->ld;
    e.e_cur = plan_E(p, 0);
    e.out = p->d_eval_out.as<double2>();
    CHK(plan_rhs_launch(p, 0, e, p->d_yin[0].as<double2>()));
    hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for(elems)), dim3(256), 0, ctx->stream,
                       p->d_eval_out.as<double2>(), 1, s->n, m, p->ld, 1, 0, p->d_eval_tmp.as<double2>());
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, p->d_eval_tmp.p, elems * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// -------------------------------------------------------------------------------------------------
// zgemm (host buffers) and expm
// -------------------------------------------------------------------------------------------------
static int upload_padded(midyn_ctx* ctx, const midyn_complex* h, int rows, int cols, double2* d, int ld) {
    HIPCHK(ctx, hipMemcpy2D(d, (size_t)ld * sizeof(double2), h, (size_t)cols * sizeof(double2),
                            (size_t)cols * sizeof(double2), rows, hipMemcpyHostToDevice));
    return 0;
}

extern "C" int midyn_zgemm(midyn_ctx* ctx, int M, int N, int K, const midyn_complex* A, const midyn_complex* B,
                           midyn_complex* C) {
    if (!ctx || !A || !B || !C || M <= 0 || N <= 0 || K <= 0) return fail(ctx, "midyn_zgemm: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int Mp = round_up(M, 64), Np = round_up(N, 64), Kp = round_up(K, 64);
    DevBuf dA, dB, dC;
    CHK(dA.alloc(ctx, (size_t)Mp * Kp * sizeof(double2)));
    CHK(dB.alloc(ctx, (size_t)Kp * Np * sizeof(double2)));
    CHK(dC.alloc(ctx, (size_t)Mp * Np * sizeof(double2)));
    HIPCHK(ctx, hipMemset(dA.p, 0, dA.bytes));
    HIPCHK(ctx, hipMemset(dB.p, 0, dB.bytes));
    CHK(upload_padded(ctx, A, M, K, dA.as<double2>(), Kp));
    CHK(upload_padded(ctx, B, K, N, dB.as<double2>(), Np));
    CHK(dev_zgemm(ctx, Mp, Np, Kp, dA.as<double2>(), Kp, dB.as<double2>(), Np, dC.as<double2>(), Np, 1.0, 0.0, nullptr));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy2D(C, (size_t)N * sizeof(double2), dC.p, (size_t)Np * sizeof(double2),
                            (size_t)N * sizeof(double2), M, hipMemcpyDeviceToHost));
    return 0;
}

// expm workspace: powers and temporaries for `batch` matrices of [np][np]
struct ExpmWork {
    int np = 0, batch = 0;
    DevBuf A2, A3, A4, T0, T1, colsum;
    int ensure(midyn_ctx* ctx, int n_pad, int nb) {
        if (np == n_pad && batch >= nb) return 0;
        const size_t b = (size_t)nb * n_pad * n_pad * sizeof(double2);
        CHK(A2.alloc(ctx, b));
        CHK(A3.alloc(ctx, b));
        CHK(A4.alloc(ctx, b));
        CHK(T0.alloc(ctx, b));
        CHK(T1.alloc(ctx, b));
        CHK(colsum.alloc(ctx, (size_t)nb * n_pad * sizeof(double)));
        np = n_pad;
        batch = nb;
        return 0;
    }
};

// Taylor polynomial of degree m evaluated with Paterson-Stockmeyer in blocks of A^q:
//   T_m(A) = sum_{j=0}^{r} (A^q)^j B_j,  B_j = sum_{i<q} c[qj+i] A^i  (+ c[m] A^q in the top block),
// m = q (r + 1): q - 1 products for the powers + r Horner products.  theta = largest ||A||_1 for which
// the truncation error theta^(m+1)/(m+1)! stays below the fp64 unit round-off (with a safety margin);
// larger norms are scaled by 2^-s and squared s times.  The (degree, s) pair of least cost is used,
// so that the small generators of rotating-frame / Magnus steps do not pay for degree 16
// (scipy's expm picks its Pade degree from the norm in the same spirit, a11).
struct TaylorScheme {
    int degree, q, r;
    double theta;
};
static const TaylorScheme EXPM_SCHEMES[] = {
    {2, 2, 0, 8.0e-6}, {4, 2, 1, 1.5e-3}, {6, 3, 1, 1.6e-2}, {9, 3, 2, 0.1}, {12, 4, 2, 0.3}, {16, 4, 3, 0.75},
};
static const int EXPM_N_SCHEMES = 6;

static void expm_choose(double norm1, int force_degree, int* scheme_out, int* s_out) {
    int best = EXPM_N_SCHEMES - 1, best_s = 0, best_cost = 1 << 30;
    for (int i = 0; i < EXPM_N_SCHEMES; ++i) {
        const TaylorScheme& sc = EXPM_SCHEMES[i];
        if (force_degree > 0 && sc.degree != force_degree) continue;
        int s = 0;
        if (norm1 > sc.theta) s = std::max(0, (int)std::ceil(std::log2(norm1 / sc.theta)));
        const int cost = (sc.q - 1) + sc.r + s;
        if (cost <= best_cost) {  // ties: the higher degree (fewer squarings)
            best_cost = cost;
            best = i;
            best_s = s;
        }
    }
    *scheme_out = best;
    *s_out = best_s;
}

// In place: X[b] <- expm(X[b]) for `batch` matrices [np][np] stored back to back on the device
// (padding rows/cols zero; the padded block of the result becomes the identity, which is harmless).
// Scaling and squaring of a Taylor polynomial (matrix products only, all on the fp64 MFMA zgemm);
// a batch shares the scheme and s of its largest matrix (over-scaling a matrix is harmless) so that
// every step is ONE batched launch.
static int dev_expm_inplace(midyn_ctx* ctx, ExpmWork& w, double2* X, int np, int* s_out, double* norm_out,
                            int batch = 1) {
    CHK(w.ensure(ctx, np, batch));
    std::vector<double> norms;
    CHK(dev_norm1(ctx, X, np, batch, w.colsum, norms));
    double norm1 = 0.0;
    for (double v : norms) norm1 = std::max(norm1, v);
    if (!std::isfinite(norm1)) return fail(ctx, "midyn_expm: matrix has non-finite entries");
    int scheme = 0, s = 0;
    expm_choose(norm1, ctx->expm_degree, &scheme, &s);
    const TaylorScheme& sc = EXPM_SCHEMES[scheme];
    if (s_out) *s_out = s;
    if (norm_out) *norm_out = norm1;
    const double scale = std::ldexp(1.0, -s);
    double c[17];
    c[0] = 1.0;
    for (int i = 1; i <= 16; ++i) c[i] = c[i - 1] / i;
    double2* A = X;
    double2* pw[5] = {nullptr, A, w.A2.as<double2>(), w.A3.as<double2>(), w.A4.as<double2>()};
    double2* T0 = w.T0.as<double2>();
    double2* T1 = w.T1.as<double2>();
    if (s > 0) {
        const double2* xs[1] = {A};
        double al[1] = {scale};
        CHK(dev_lincomb(ctx, np, A, 1, xs, al, 0.0, batch));
    }
    const int q = sc.q, r = sc.r;
    CHK(dev_sqgemm(ctx, batch, np, A, A, pw[2], 1.0, 0.0, nullptr));
    if (q >= 3) CHK(dev_sqgemm(ctx, batch, np, pw[2], A, pw[3], 1.0, 0.0, nullptr));
    if (q >= 4) CHK(dev_sqgemm(ctx, batch, np, pw[2], pw[2], pw[4], 1.0, 0.0, nullptr));
    // block j: c[qj] I + c[qj+1] A + ... + c[qj+q-1] A^(q-1)   (+ c[q(r+1)] A^q for the top block j = r)
    auto block = [&](int j, bool top, double2* out) {
        const double2* xs[4];
        double al[4];
        int nt = 0;
        for (int i = 1; i < q; ++i) {
            xs[nt] = pw[i];
            al[nt++] = c[q * j + i];
        }
        if (top) {
            xs[nt] = pw[q];
            al[nt++] = c[q * (r + 1)];
        }
        return dev_lincomb(ctx, np, out, nt, xs, al, c[q * j], batch);
    };
    if (r == 0) {
        CHK(block(0, true, X));  // elementwise, in place on A
    } else {
        CHK(block(r, true, T0));
        double2* P = T0;
        double2* Q = T1;
        for (int j = r - 1; j >= 0; --j) {
            // Q = B_j + A^q . P ; the last product goes back into X (= A), B_0 being built first
            CHK(block(j, false, Q));
            CHK(dev_sqgemm(ctx, batch, np, pw[q], P, j > 0 ? Q : X, 1.0, 1.0, Q));
            std::swap(P, Q);
        }
    }
    // squarings: X <- X.X, ping-pong through T0
    double2* cur = X;
    double2* oth = T0;
    for (int i = 0; i < s; ++i) {
        CHK(dev_sqgemm(ctx, batch, np, cur, cur, oth, 1.0, 0.0, nullptr));
        std::swap(cur, oth);
    }
    if (cur != X)
        HIPCHK(ctx, hipMemcpyAsync(X, cur, (size_t)batch * np * np * sizeof(double2), hipMemcpyDeviceToDevice,
                                   ctx->stream));
    return 0;
}

// how many [np][np] problems are advanced together: enough to fill the chip, bounded by ~3 GB of
// workspace (12 matrices per problem)
static int expm_chunk(midyn_ctx* ctx, int np, int total) {
    const size_t per = (size_t)np * np * sizeof(double2) * 12;
    long long cap = (long long)(((size_t)3 << 30) / per);
    if (np >= 1024) cap = 1;          // one such expm already fills the device
    cap = std::max(1LL, std::min<long long>(cap, 4096));
    return (int)std::min<long long>(cap, total);
}

extern "C" int midyn_expm(midyn_ctx* ctx, int n, int batch, const midyn_complex* A, midyn_complex* E_out,
                          long long* info) {
    if (!ctx || !A || !E_out || n <= 0 || batch <= 0) return fail(ctx, "midyn_expm: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int np = round_up(n, 64);
    const int chunk = expm_chunk(ctx, np, batch);
    const size_t mat = (size_t)np * np;
    ExpmWork w;
    DevBuf X;
    CHK(X.alloc(ctx, (size_t)chunk * mat * sizeof(double2)));
    // many small matrices: pad / unpad on the host and move each chunk with ONE copy per direction
    const bool pack = batch > 1 && np <= 512;
    std::vector<double2> stage;
    if (pack) stage.resize((size_t)chunk * mat);
    for (int b0 = 0; b0 < batch; b0 += chunk) {
        const int nb = std::min(chunk, batch - b0);
        if (pack) {
            std::fill(stage.begin(), stage.begin() + (size_t)nb * mat, make_double2(0.0, 0.0));
            for (int b = 0; b < nb; ++b)
                for (int r = 0; r < n; ++r)
                    memcpy(&stage[(size_t)b * mat + (size_t)r * np], A + ((size_t)(b0 + b) * n + r) * n,
                           (size_t)n * sizeof(double2));
            HIPCHK(ctx, hipMemcpy(X.p, stage.data(), (size_t)nb * mat * sizeof(double2), hipMemcpyHostToDevice));
        } else {
            HIPCHK(ctx, hipMemset(X.p, 0, (size_t)nb * mat * sizeof(double2)));
            for (int b = 0; b < nb; ++b)
                CHK(upload_padded(ctx, A + (size_t)(b0 + b) * n * n, n, n, X.as<double2>() + b * mat, np));
        }
        int s = 0;
        double nrm = 0;
        CHK(dev_expm_inplace(ctx, w, X.as<double2>(), np, &s, &nrm, nb));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (pack) HIPCHK(ctx, hipMemcpy(stage.data(), X.p, (size_t)nb * mat * sizeof(double2), hipMemcpyDeviceToHost));
        for (int b = 0; b < nb; ++b) {
            if (pack) {
                for (int r = 0; r < n; ++r)
                    memcpy(E_out + ((size_t)(b0 + b) * n + r) * n, &stage[(size_t)b * mat + (size_t)r * np],
                           (size_t)n * sizeof(double2));
            } else {
                HIPCHK(ctx, hipMemcpy2D(E_out + (size_t)(b0 + b) * n * n, (size_t)n * sizeof(double2),
                                        X.as<double2>() + b * mat, (size_t)np * sizeof(double2),
                                        (size_t)n * sizeof(double2), n, hipMemcpyDeviceToHost));
            }
            if (info) {
                info[2 * (b0 + b)] = s;
                info[2 * (b0 + b) + 1] = (long long)(nrm * 1e6);
            }
        }
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Magnus / expm fixed-step solver
// -------------------------------------------------------------------------------------------------
static int commutator(midyn_ctx* ctx, int np, const double2* a, const double2* b, double2* out, double2* tmp,
                      int batch = 1) {
    // out = a.b - b.a
    CHK(dev_sqgemm(ctx, batch, np, b, a, tmp, 1.0, 0.0, nullptr));
    CHK(dev_sqgemm(ctx, batch, np, a, b, out, 1.0, -1.0, tmp));
    return 0;
}

// Omega_m of the Magnus step (solvers/fixed_step_solvers.py:345-392) for `nb` problems at once.
// gen(i, scale, out) must write scale * G(t_i) for the i-th Gauss point of every problem; with
// generators that already carry their step size (gen ignores nothing, h == 1) the same code serves the
// parallel-in-time solver, because Omega_m is homogeneous in h G.
template <class Gen>
static int magnus_omega(midyn_ctx* ctx, int np, int nb, int magnus_order, double h, Gen&& gen, DevBuf* G, DevBuf* W,
                        double2* Omega) {
    if (magnus_order == 1) {
        CHK(gen(0, h, Omega));
    } else if (magnus_order == 2) {
        // fixed_step_solvers.py:348-363
        CHK(gen(0, 1.0, G[0].as<double2>()));
        CHK(gen(1, 1.0, G[1].as<double2>()));
        CHK(commutator(ctx, np, G[1].as<double2>(), G[0].as<double2>(), W[0].as<double2>(), W[1].as<double2>(), nb));
        const double p2 = std::sqrt(3.0) / 12;
        const double2* xs[3] = {G[0].as<double2>(), G[1].as<double2>(), W[0].as<double2>()};
        double al[3] = {h / 2, h / 2, p2 * (h * h)};
        CHK(dev_lincomb(ctx, np, Omega, 3, xs, al, 0.0, nb));
    } else {
        // fixed_step_solvers.py:365-392
        const double c0 = std::sqrt(15.0) / 3, c1 = 10.0 / 3;
        CHK(gen(0, 1.0, G[0].as<double2>()));
        CHK(gen(1, 1.0, G[1].as<double2>()));
        CHK(gen(2, 1.0, G[2].as<double2>()));
        double2 *g1 = G[0].as<double2>(), *g2 = G[1].as<double2>(), *g3 = G[2].as<double2>();
        double2 *w0 = W[0].as<double2>(), *w1 = W[1].as<double2>(), *w2 = W[2].as<double2>(),
                *w3 = W[3].as<double2>();
        // a1 -> g2 (in place), a2 -> w0, a3 -> w1
        {
            const double2* xs[2] = {g3, g1};
            double al[2] = {c0 * h, -c0 * h};
            CHK(dev_lincomb(ctx, np, w0, 2, xs, al, 0.0, nb));
        }
        {
            const double2* xs[3] = {g3, g2, g1};
            double al[3] = {c1 * h, -2 * c1 * h, c1 * h};
            CHK(dev_lincomb(ctx, np, w1, 3, xs, al, 0.0, nb));
        }
        {
            const double2* xs[1] = {g2};
            double al[1] = {h};
            CHK(dev_lincomb(ctx, np, g2, 1, xs, al, 0.0, nb));
        }
        double2 *a1 = g2, *a2 = w0, *a3 = w1;
        // comm1 = [a1, a2] -> w2 (tmp g1)
        CHK(commutator(ctx, np, a1, a2, w2, g1, nb));
        double2* comm1 = w2;
        // X = 2 a3 + comm1 -> g3 ; comm2 = [X, a1]/60 -> w3 (tmp g1)
        {
            const double2* xs[2] = {a3, comm1};
            double al[2] = {2.0, 1.0};
            CHK(dev_lincomb(ctx, np, g3, 2, xs, al, 0.0, nb));
        }
        CHK(commutator(ctx, np, g3, a1, w3, g1, nb));
        // Y2 = a2 + comm2/60 -> g3 ; Y1 = -20 a1 - a3 + comm1 -> g1
        {
            const double2* xs[2] = {a2, w3};
            double al[2] = {1.0, 1.0 / 60};
            CHK(dev_lincomb(ctx, np, g3, 2, xs, al, 0.0, nb));
        }
        {
            const double2* xs[3] = {a1, a3, comm1};
            double al[3] = {-20.0, -1.0, 1.0};
            CHK(dev_lincomb(ctx, np, g1, 3, xs, al, 0.0, nb));
        }
        // comm3 = [Y1, Y2] -> w3 (tmp w2: comm1 no longer needed)
        CHK(commutator(ctx, np, g1, g3, w3, w2, nb));
        {
            const double2* xs[3] = {a1, a3, w3};
            double al[3] = {1.0, 1.0 / 12, 1.0 / 240};
            CHK(dev_lincomb(ctx, np, Omega, 3, xs, al, 0.0, nb));
        }
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// expm ACTION: y <- expm(Omega_m) y without forming the exponential (a10/a11 for states with few
// columns).  The reference computes scipy.linalg.expm(Omega) (n^3 work, ~(7+s) zgemm) and multiplies
// it into y even when y is a single vector; the result only needs  expm(Omega) y = sum_j Omega^j y / j!,
// i.e. products Omega.v, which for
//     order 1:  Omega v = h G(t1) v
//     order 2:  Omega v = h/2 (g1 v + g2 v) + sqrt(3)/12 h^2 (g2 (g1 v) - g1 (g2 v))     (commutator free)
// are exactly the batched RHS contraction of row a2/a7 (all instances of a sweep in ONE MFMA GEMM over
// the operator stack, per-instance coefficients; the streaming kernel for one column).  Scaling:
// y <- (T_p(Omega / s))^s y with (p, s) of least p*s such that bound/s <= theta_p, where
// bound >= ||Omega||_1 follows from the per-segment norms: ||G(t)||_1 <= sum_seg |c_seg| ||A_seg||_1
// (the frame phases have modulus 1).  One instance: G(t_i) is formed once per step (gen_eval) and the
// products run on that single matrix.
// -------------------------------------------------------------------------------------------------
struct ActionScheme {
    int p;
    double theta;
};
static const ActionScheme ACTION_SCHEMES[] = {{2, 8.0e-6}, {3, 2.0e-4}, {4, 1.5e-3}, {5, 6.0e-3}, {6, 1.6e-2},
                                              {8, 6.5e-2}, {10, 0.16}, {12, 0.3},   {15, 0.62},  {20, 1.35}};

static void action_choose(double bound, int* p_out, int* s_out) {
    long long best_cost = -1;
    for (const ActionScheme& sc : ACTION_SCHEMES) {
        const int s = bound > sc.theta ? (int)std::ceil(bound / sc.theta) : 1;
        const long long cost = (long long)sc.p * s;
        if (best_cost < 0 || cost <= best_cost) {
            best_cost = cost;
            *p_out = sc.p;
            *s_out = s;
        }
    }
}

static int stack_seg_norms(midyn_stack* s) {
    if (!s->seg_norm1.empty()) return 0;
    midyn_ctx* ctx = s->ctx;
    DevBuf cs;
    CHK(dev_norm1(ctx, s->ops, s->n_pad, s->nseg, cs, s->seg_norm1));
    return 0;
}

static int stack_seg_aux_norms(midyn_stack* s) {
    if (!s->seg_herm1.empty()) return 0;
    midyn_ctx* ctx = s->ctx;
    DevBuf cs;
    CHK(dev_norm1(ctx, s->ops, s->n_pad, s->nseg, cs, s->seg_norminf, 1));
    CHK(dev_norm1(ctx, s->ops, s->n_pad, s->nseg, cs, s->seg_herm1, 2));
    return 0;
}

// Bessel functions J_0..J_K of the first kind at x > 0 by Miller's backward recurrence (normalised with
// J_0 + 2 sum J_2k = 1); K is chosen by the caller, the recurrence starts far enough above it.
static std::vector<double> bessel_j(double x, int K) {
    const int M = 2 * ((std::max(K, (int)std::ceil(x)) + 40) / 2 + 8);
    std::vector<double> j(M + 2, 0.0);
    j[M] = 1e-280;
    for (int k = M; k >= 1; --k) {
        j[k - 1] = (2.0 * k / x) * j[k] - j[k + 1];
        if (std::fabs(j[k - 1]) > 1e250)
            for (int q = k - 1; q <= M; ++q) j[q] *= 1e-250;
    }
    double norm = j[0];
    for (int k = 2; k <= M; k += 2) norm += 2.0 * j[k];
    j.resize(K + 1);
    for (double& v : j) v /= norm;
    return j;
}

static int expm_action_solve(midyn_stack* s, int B, int m, int R, const double* times, const double* S_host,
                             const double* S_any, int nsteps, const int* step_rows, const double* step_h,
                             const int* step_save, int P, int magnus_order, const midyn_complex* y0, int y0_shared,
                             midyn_complex* Y_out) {
    midyn_ctx* ctx = s->ctx;
    CHK(stack_seg_norms(s));
    midyn_rk4_plan* p = nullptr;
    CHK(plan_create_impl(s, B, m, R, times, S_any, nsteps, step_rows, step_h, step_save, P, y0, y0_shared, &p));
    struct Guard {
        midyn_rk4_plan* p;
        ~Guard() { midyn_rk4_plan_destroy(p); }
    } guard{p};
    const int np = s->n_pad, ld = p->ld;
    const size_t stv = (size_t)np * ld, state_bytes = stv * sizeof(double2);
    // one instance: explicit G(t_i), products on a single matrix -- unless the stack is block sparse, where
    // the per-segment work lists touch far fewer bytes than one dense n x n matrix
    const bool one = (B == 1) && !p->blocks;
    const int npts = magnus_order;
    DevBuf Gx[2], U[2], V[2], W, d_cs;
    std::vector<double> h_cs;
    if (one)
        for (int i = 0; i < npts; ++i) CHK(Gx[i].alloc(ctx, (size_t)np * np * sizeof(double2)));
    // Magnus 2 with a frame, products through the plan and no fused input phase (sweeps, MFMA routes): the
    // producers write the phased copies of their results (see the term loop)
    const bool chain_phases = magnus_order == 2 && !one && s->has_frame && !(p->blocks && p->ncol <= 8);
    DevBuf TP[4];
    if (magnus_order == 2) {
        for (int i = 0; i < 2; ++i) {
            CHK(U[i].alloc(ctx, state_bytes));
            CHK(V[i].alloc(ctx, state_bytes));
        }
        CHK(W.alloc(ctx, state_bytes));
        HIPCHK(ctx, hipMemsetAsync(W.p, 0, state_bytes, ctx->stream));
        if (chain_phases)
            for (int i = 0; i < 4; ++i) {
                CHK(TP[i].alloc(ctx, state_bytes));
                HIPCHK(ctx, hipMemsetAsync(TP[i].p, 0, state_bytes, ctx->stream));
            }
    }
    double2* y = p->d_y.as<double2>();
    double2* acc = p->d_acc.as<double2>();
    double2* yin[2] = {p->d_yin[0].as<double2>(), p->d_yin[1].as<double2>()};
    // out = G(point i) . w   (EPI_RHS)  or the fused Taylor update (EPI_TAYLOR) with the given epilogue
    auto product = [&](int i, int row, const double2* w_plain, const double2* w_phased, Epilogue epi) -> int {
        if (one) {
            epi.e_cur = nullptr;   // gen_eval already applied the frame: G = Delta(t) o C(t)
            epi.e_next = nullptr;
            if (p->stream_path) {
                StreamArgs a{};
                a.ops = Gx[i].as<double2>();
                a.seg_list = ctx->d_one_seg;
                a.n_act = 1;
                a.n_pad = np;
                a.has_static = 1;
                a.coeff = nullptr;
                a.yin = w_plain;
                a.epi = epi;
                return launch_stream(ctx, a);
            }
            GemmArgs g{};
            g.A = Gx[i].as<double2>();
            g.lda = np;
            g.B = w_plain;
            g.ldb = ld;
            g.M = np;
            g.N = ld;
            g.K = np;
            g.seg_list = ctx->d_one_seg;
            g.n_act = 1;
            g.m_cols = m;
            g.n_inst = 1;
            g.epi = epi;
            return launch_gemm(ctx, g, KC_RHS_GEMM);
        }
        (void)row;
        return plan_rhs_launch(p, row, epi, w_phased);
    };
    auto rephase = [&](const double2* src, int row, double2* dst) -> int {
        hipLaunchKernelGGL(rephase_kernel, dim3(grid_for(stv)), dim3(256), 0, ctx->stream, src, plan_E(p, row), np, ld,
                           dst);
        HIPCHK(ctx, hipGetLastError());
        return 0;
    };
    // out = conj-phase(G(row) . w) for an UN-phased w: explicit G (one), the block kernels' fused input phase, or
    // a rephase pass into `scratch` followed by the plan's contraction
    const bool fuse_phase = !one && p->blocks && p->ncol <= 8;
    auto product_plain = [&](int i, int row, const double2* w, double2* scratch, Epilogue epi) -> int {
        if (one) return product(i, row, w, w, epi);
        epi.e_cur = plan_E(p, row);
        if (fuse_phase || !plan_E(p, row)) return plan_rhs_launch(p, row, epi, w, fuse_phase ? plan_E(p, row) : nullptr);
        CHK(rephase(w, row, scratch));
        return plan_rhs_launch(p, row, epi, scratch);
    };
    const double p2 = std::sqrt(3.0) / 12;
    // norm bound of Omega over the instances for one step (triangle inequality over the segments)
    auto step_bound = [&](int st) {
        const double h = step_h[st];
        const int* rr = step_rows + 3 * st;
        double bound = 0.0;
        for (int b = 0; b < B; ++b) {
            double gn[2] = {0.0, 0.0};
            for (int i = 0; i < npts; ++i) {
                const double* c = s->k > 0 ? S_host + ((size_t)b * R + rr[i]) * s->k : nullptr;
                for (int seg = 0; seg < s->nseg; ++seg) {
                    const double cf = (s->has_static && seg == 0) ? 1.0 : std::fabs(c[seg - s->has_static]);
                    gn[i] += cf * s->seg_norm1[seg];
                }
            }
            const double ah = std::fabs(h);
            const double bb = magnus_order == 1 ? ah * gn[0] : 0.5 * ah * (gn[0] + gn[1]) + 2 * p2 * ah * ah * gn[0] * gn[1];
            bound = std::max(bound, bb);
        }
        return bound;
    };
    // the same triangle bound with another per-segment norm table (infinity norms, Hermitian parts), order 1
    auto step_bound_with = [&](int st, const std::vector<double>& seg_norm) {
        const double ah = std::fabs(step_h[st]);
        const int row = step_rows[3 * st];
        double bound = 0.0;
        for (int b = 0; b < B; ++b) {
            const double* c = s->k > 0 ? S_host + ((size_t)b * R + row) * s->k : nullptr;
            double g = 0.0;
            for (int seg = 0; seg < s->nseg; ++seg)
                g += ((s->has_static && seg == 0) ? 1.0 : std::fabs(c[seg - s->has_static])) * seg_norm[seg];
            bound = std::max(bound, ah * g);
        }
        return bound;
    };
    if (p->tiny) {
        // small system: the whole solve in one persistent launch (tiny_expm_kernel), the Taylor degree and
        // scaling of every step chosen here from the same bound
        std::vector<int> deg(nsteps), scv(nsteps);
        for (int st = 0; st < nsteps; ++st) {
            const double bound = step_bound(st);
            if (!std::isfinite(bound)) return fail(ctx, "midyn_expm_solve: non-finite generator norm");
            action_choose(bound, &deg[st], &scv[st]);
        }
        DevBuf d_deg, d_sc;
        CHK(d_deg.alloc(ctx, (size_t)nsteps * sizeof(int)));
        CHK(d_sc.alloc(ctx, (size_t)nsteps * sizeof(int)));
        HIPCHK(ctx, hipMemcpy(d_deg.p, deg.data(), (size_t)nsteps * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(d_sc.p, scv.data(), (size_t)nsteps * sizeof(int), hipMemcpyHostToDevice));
        for (int st = 0; st < nsteps; ++st)
            if (step_save && step_save[st] >= P) return fail(ctx, "midyn_expm_solve: save slot out of range");
        static bool attr_set[16] = {false};
        if (!attr_set[ctx->device & 15]) {
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(tiny_expm_kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024));
            attr_set[ctx->device & 15] = true;
        }
        TinyArgs a = tiny_args(p, 0, nsteps);
        {
            ProfScope ps(ctx, KC_STREAM);
            hipLaunchKernelGGL(tiny_expm_kernel, dim3((p->ncol + 3) / 4), dim3(256), p->tiny_smem, ctx->stream, a,
                               magnus_order, d_deg.as<int>(), d_sc.as<int>());
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(Y_out, p->d_out.p, (size_t)B * P * s->n * m * sizeof(double2), hipMemcpyDeviceToHost));
        return 0;
    }
    // ---- Krylov alternative for ONE column and a large norm (Magnus order 1): Arnoldi on G(t1) with
    // classical Gram-Schmidt + one re-orthogonalisation, all on the device (no host round trip per
    // iteration); every few iterations expm(h H_m) of the small Hessenberg matrix (64 x 64 block, the
    // batched-expm machinery) and Saad's a-posteriori estimate beta h_{m+1,m} |e_m^T expm(h H_m) e_1|;
    // y <- beta V_m expm(h H_m) e_1.  The scaled Taylor series needs ~15 products per unit of ||h G||_1,
    // Arnoldi about 1.5 x the spectral radius + 20 in total (cfg 4, no frame: 160 -> ~40 products).
    DevBuf kV, kH, kE, kw, khc, kbeta, kerr, kcoef, cheb_buf[2];
    ExpmWork kwork;
    const int KM = 60;
    auto krylov_step = [&](double h, int row, double bound, const double2* ycur, double2* ynew, bool* converged) -> int {
        *converged = false;
        if (!kV.p) {
            CHK(kV.alloc(ctx, (size_t)(KM + 1) * np * sizeof(double2)));
            CHK(kH.alloc(ctx, 64 * 64 * sizeof(double2)));
            CHK(kE.alloc(ctx, 64 * 64 * sizeof(double2)));
            CHK(kw.alloc(ctx, (size_t)np * sizeof(double2)));
            CHK(khc.alloc(ctx, 64 * sizeof(double2)));
            CHK(kbeta.alloc(ctx, 2 * sizeof(double)));
            CHK(kerr.alloc(ctx, 2 * sizeof(double)));
            CHK(kcoef.alloc(ctx, 64 * sizeof(double2)));
        }
        double2* Vb = kV.as<double2>();
        double2* Hm = kH.as<double2>();
        double2* Es = kE.as<double2>();
        double2* wv = kw.as<double2>();
        HIPCHK(ctx, hipMemsetAsync(kH.p, 0, kH.bytes, ctx->stream));
        hipLaunchKernelGGL(krylov_norm_scale_kernel, dim3(1), dim3(1024), 0, ctx->stream, ycur, np, 0, (double2*)nullptr,
                           kbeta.as<double>(), Vb);
        int next_check = std::min(KM, std::max(8, (int)(1.5 * bound) + 14));
        int m = 0;
        bool done = false;
        for (int j = 0; j < KM && !done; ++j) {
            Epilogue e{};
            e.mode = EPI_RHS;
            e.ld = ld;
            e.out = wv;
            CHK(product_plain(0, row, Vb + (size_t)j * np, yin[0], e));   // w = G v_j
            for (int pass = 0; pass < 2; ++pass) {                                 // CGS + re-orthogonalisation
                hipLaunchKernelGGL(krylov_dot_kernel, dim3(j + 1), dim3(256), 0, ctx->stream, Vb, np, wv, np, j, pass,
                                   khc.as<double2>(), Hm);
                hipLaunchKernelGGL(krylov_axpy_kernel, dim3(grid_for(np, 64)), dim3(256), 0, ctx->stream, Vb, np,
                                   khc.as<double2>(), 1, j + 1, -1.0, wv, np, wv);
            }
            hipLaunchKernelGGL(krylov_norm_scale_kernel, dim3(1), dim3(1024), 0, ctx->stream, wv, np, j, Hm,
                               kbeta.as<double>() + 1, Vb + (size_t)(j + 1) * np);
            HIPCHK(ctx, hipGetLastError());
            m = j + 1;
            if (m == next_check || m == KM) {
                hipLaunchKernelGGL(krylov_small_kernel, dim3(16), dim3(256), 0, ctx->stream, Hm, m, h, Es);
                CHK(dev_expm_inplace(ctx, kwork, Es, 64, nullptr, nullptr, 1));
                hipLaunchKernelGGL(krylov_err_kernel, dim3(1), dim3(64), 0, ctx->stream, Es, Hm, m, h, kbeta.as<double>(),
                                   kerr.as<double>());
                HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, kerr.p, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                const double err = ctx->h_pinned[0], beta = ctx->h_pinned[1];
                if (!std::isfinite(err)) return fail(ctx, "midyn_expm_solve: non-finite Krylov estimate");
                if (err <= 1e-15 * beta) done = true;
                else next_check = std::min(KM, m + 6);
            }
        }
        if (!done) return 0;   // not converged within KM vectors: the caller falls back to the Taylor series
        hipLaunchKernelGGL(krylov_coef_kernel, dim3(1), dim3(64), 0, ctx->stream, Es, m, kbeta.as<double>(),
                           kcoef.as<double2>());
        hipLaunchKernelGGL(krylov_axpy_kernel, dim3(grid_for(np, 64)), dim3(256), 0, ctx->stream, Vb, np,
                           kcoef.as<double2>(), 1, m, 1.0, (const double2*)nullptr, np, ynew);
        HIPCHK(ctx, hipGetLastError());
        *converged = true;
        return 0;
    };
    for (int st = 0; st < nsteps; ++st) {
        const double h = step_h[st];
        const int* rr = step_rows + 3 * st;
        // ---- norm bound over the instances -> (degree, scaling)
        double bound = step_bound(st);
        if (one) {
            for (int i = 0; i < npts; ++i)
                CHK(launch_gen_eval(s, s->k > 0 ? p->d_S.as<double>() + (size_t)rr[i] * s->k : nullptr, plan_E(p, rr[i]),
                                    1.0, Gx[i].as<double2>()));
            // The generators exist explicitly: their exact 1-norms can replace the triangle bound.  That
            // costs a stream synchronisation, so only when the products it may save are worth > 1 ms.
            int deg0 = 2, sc0 = 1;
            action_choose(bound, &deg0, &sc0);
            const double product_s = std::max(5e-6, (double)np * np * 16.0 / 5e12);
            if ((double)deg0 * sc0 * (magnus_order == 1 ? 1 : 4) * product_s > 1e-3) {
                double gn[2] = {0.0, 0.0};
                for (int i = 0; i < npts; ++i) {
                    CHK(dev_norm1(ctx, Gx[i].as<double2>(), np, 1, d_cs, h_cs));
                    gn[i] = h_cs[0];
                }
                const double ah = std::fabs(h);
                bound = std::min(bound, magnus_order == 1 ? ah * gn[0]
                                                          : 0.5 * ah * (gn[0] + gn[1]) + 2 * p2 * ah * ah * gn[0] * gn[1]);
            }
        }
        if (!std::isfinite(bound)) return fail(ctx, "midyn_expm_solve: non-finite generator norm");
        int deg = 2, sc = 1;
        action_choose(bound, &deg, &sc);
        bool stepped = false;
        // ---- Chebyshev series for a nearly skew-Hermitian Omega = h G (Hamiltonian models exactly, Lindbladians
        // with weak dissipation): with B = Omega / rho, rho >= the numerical radius,
        //     expm(Omega) y = J_0(rho) phi_0 + 2 sum_k J_k(rho) phi_k,   phi_0 = y, phi_1 = B y,
        //     phi_{k+1} = 2 B phi_k + phi_{k-1}          (phi_k = i^k T_k(B / i) y, Bessel J_k),
        // which needs about rho + 10 rho^(1/3) + 10 products where the scaled Taylor series needs ~15 rho
        // (cfg 4 without a frame, rho = 10.8: 39 products instead of 160), no orthogonalisation, one launch per
        // term (EPI_CHEB epilogue), sweeps included.  The series is only used when the Hermitian part is small
        // enough that the polynomials cannot grow:  K sqrt(2 ||herm(Omega)|| / rho) <= 0.7.
        if (magnus_order == 1 && ctx->chebyshev && ctx->krylov < 2) {
            CHK(stack_seg_aux_norms(s));
            double rho = std::max(bound, step_bound_with(st, s->seg_norminf));
            if (one) rho = std::max(step_bound(st), rho);  // `bound` may have been tightened to the exact 1-norm
            const double herm = step_bound_with(st, s->seg_herm1);
            const int reps = std::max(1, (int)std::ceil(rho / 128.0));  // Bessel table accurate to a few 1e-15 up to here
            const double rr_ = rho / reps;
            int K = 0;
            std::vector<double> coef;
            if (rr_ > 0.0 && std::isfinite(rr_)) {
                const int kmax = (int)(rr_ + 10.0 * std::cbrt(rr_) + 40.0);
                coef = bessel_j(rr_, kmax);
                K = kmax;
                while (K > 1 && std::fabs(coef[K]) < 1e-18) --K;
            }
            const bool stable = K > 0 && (double)K * std::sqrt(2.0 * herm / std::max(rho, 1e-300)) <= 0.7;
            const bool shorter = (long long)reps * (K + 1) * 10 < (long long)deg * sc * 8;
            if (K > 0 && K < (int)coef.size() - 1 && stable && (shorter || ctx->chebyshev >= 2)) {
                if (!cheb_buf[0].p)
                    for (int i = 0; i < 2; ++i) CHK(cheb_buf[i].alloc(ctx, state_bytes));
                double2* P2[2] = {cheb_buf[0].as<double2>(), cheb_buf[1].as<double2>()};
                for (int rep = 0; rep < reps; ++rep) {
                    hipLaunchKernelGGL(scale_copy_kernel, dim3(grid_for(stv)), dim3(256), 0, ctx->stream, y, coef[0], stv, acc);
                    HIPCHK(ctx, hipGetLastError());
                    int cur = 0;
                    if (one) HIPCHK(ctx, hipMemcpyAsync(yin[0], y, state_bytes, hipMemcpyDeviceToDevice, ctx->stream));
                    else CHK(rephase(y, rr[0], yin[0]));
                    for (int k = 0; k < K; ++k) {
                        Epilogue e{};
                        e.mode = EPI_CHEB;
                        e.ld = ld;
                        e.alpha = (k == 0 ? 1.0 : 2.0) * h / (rr_ * reps);
                        e.beta = 2.0 * coef[k + 1];
                        e.z = k == 0 ? nullptr : (k == 1 ? y : P2[k & 1]);
                        e.out = P2[k & 1];
                        e.e_cur = plan_E(p, rr[0]);
                        e.e_next = plan_E(p, rr[0]);
                        e.acc = acc;
                        e.yin_next = yin[cur ^ 1];
                        CHK(product(0, rr[0], yin[cur], yin[cur], e));
                        cur ^= 1;
                    }
                    std::swap(y, acc);
                }
                stepped = true;
            }
        }
        // Arnoldi pays ~6 launches per vector (product, two Gram-Schmidt passes, normalisation) against one per
        // Taylor term: with the microsecond products of a block-sparse stack both are launch bound and the
        // series wins unless it is several times longer (cfg 4, 100 steps: Taylor 160 terms 0.100 s, Arnoldi 28
        // vectors 0.113 s); with dense streamed products (tens of microseconds each) Arnoldi wins from 64 terms
        const long long krylov_min = ctx->krylov >= 2 ? 0 : (p->blocks ? (long long)(6.0 * (1.5 * bound + 20.0)) : 64);
        if (!stepped && (one || p->blocks) && p->stream_path && magnus_order == 1 && ctx->krylov &&
            (long long)deg * sc >= krylov_min) {
            bool conv = false;
            CHK(krylov_step(h, rr[0], bound, y, acc, &conv));
            if (conv) {
                std::swap(y, acc);
                stepped = true;
            }
        }
        for (int rep = 0; rep < sc && !stepped; ++rep) {
            HIPCHK(ctx, hipMemcpyAsync(acc, y, state_bytes, hipMemcpyDeviceToDevice, ctx->stream));
            if (magnus_order == 1) {
                int cur = 0;
                if (one) HIPCHK(ctx, hipMemcpyAsync(yin[0], y, state_bytes, hipMemcpyDeviceToDevice, ctx->stream));
                else CHK(rephase(y, rr[0], yin[0]));
                for (int j = 1; j <= deg; ++j) {
                    Epilogue e{};
                    e.mode = EPI_TAYLOR;
                    e.ld = ld;
                    e.h = h / ((double)sc * j);
                    e.e_cur = plan_E(p, rr[0]);
                    e.e_next = plan_E(p, rr[0]);
                    e.acc = acc;
                    e.yin_next = yin[cur ^ 1];
                    CHK(product(0, rr[0], yin[cur], yin[cur], e));
                    cur ^= 1;
                }
            } else {
                const double2* term = y;
                double2 *u1 = U[0].as<double2>(), *u2 = U[1].as<double2>(), *v1 = V[0].as<double2>(),
                        *v2 = V[1].as<double2>(), *w = W.as<double2>();
                if (chain_phases) {
                    // Frames, products through the plan, no fused input phase: every product input must be
                    // pre-phased.  The producers write the phased copies themselves -- the EPI_RHS epilogue's
                    // second output for u1, u2, the combination kernel for the next term -- so a term is
                    // 4 products + 1 combination instead of 4 re-phasing passes on top.
                    double2 *tp0 = TP[0].as<double2>(), *tp1 = TP[1].as<double2>(), *u1p = TP[2].as<double2>(),
                            *u2p = TP[3].as<double2>();
                    const double2 *E0 = plan_E(p, rr[0]), *E1 = plan_E(p, rr[1]);
                    CHK(rephase(y, rr[0], tp0));
                    CHK(rephase(y, rr[1], tp1));
                    for (int j = 1; j <= deg; ++j) {
                        Epilogue e{};
                        e.mode = EPI_RHS;
                        e.ld = ld;
                        e.e_cur = E0;  // u1 = g1 term, and E1 o u1 for v1
                        e.out = u1;
                        e.e_next = E1;
                        e.yin_next = u1p;
                        CHK(plan_rhs_launch(p, rr[0], e, tp0));
                        e.e_cur = E1;  // u2 = g2 term, and E0 o u2 for v2
                        e.out = u2;
                        e.e_next = E0;
                        e.yin_next = u2p;
                        CHK(plan_rhs_launch(p, rr[1], e, tp1));
                        e.e_next = nullptr;
                        e.yin_next = nullptr;
                        e.e_cur = E1;  // v1 = g2 u1
                        e.out = v1;
                        CHK(plan_rhs_launch(p, rr[1], e, u1p));
                        e.e_cur = E0;  // v2 = g1 u2
                        e.out = v2;
                        CHK(plan_rhs_launch(p, rr[0], e, u2p));
                        const double f = 1.0 / ((double)sc * j);
                        hipLaunchKernelGGL(magnus2_term_kernel, dim3(grid_for(stv)), dim3(256), 0, ctx->stream, u1, u2, v1,
                                           v2, 0.5 * h * f, p2 * h * h * f, stv, w, acc, E0, E1, ld, tp0, tp1);
                        HIPCHK(ctx, hipGetLastError());
                    }
                } else {
                    for (int j = 1; j <= deg; ++j) {
                        Epilogue e{};
                        e.mode = EPI_RHS;
                        e.ld = ld;
                        // u1 = g1 term, u2 = g2 term
                        e.out = u1;
                        CHK(product_plain(0, rr[0], term, yin[0], e));
                        e.out = u2;
                        CHK(product_plain(1, rr[1], term, yin[1], e));
                        // v1 = g2 u1, v2 = g1 u2
                        e.out = v1;
                        CHK(product_plain(1, rr[1], u1, yin[0], e));
                        e.out = v2;
                        CHK(product_plain(0, rr[0], u2, yin[1], e));
                        const double f = 1.0 / ((double)sc * j);
                        hipLaunchKernelGGL(magnus2_term_kernel, dim3(grid_for(stv)), dim3(256), 0, ctx->stream, u1, u2, v1,
                                           v2, 0.5 * h * f, p2 * h * h * f, stv, w, acc, (const double2*)nullptr,
                                           (const double2*)nullptr, ld, (double2*)nullptr, (double2*)nullptr);
                        HIPCHK(ctx, hipGetLastError());
                        term = w;
                    }
                }
            }
            std::swap(y, acc);
        }
        if (step_save && step_save[st] >= 0) {
            if (step_save[st] >= P) return fail(ctx, "midyn_expm_solve: save slot out of range");
            hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for((size_t)B * s->n * m)), dim3(256), 0, ctx->stream, y,
                               B, s->n, m, ld, P, step_save[st], p->d_out.as<double2>());
            HIPCHK(ctx, hipGetLastError());
        }
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(Y_out, p->d_out.p, (size_t)B * P * s->n * m * sizeof(double2), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int midyn_expm_solve(midyn_stack* s, int B, int m, int R, const double* times, const double* S,
                                int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                                int P, int magnus_order, const midyn_complex* y0, int y0_shared,
                                midyn_complex* Y_out) {
    if (!s || !Y_out || !y0 || !times || !step_rows || !step_h)
        return fail(s ? s->ctx : nullptr, "midyn_expm_solve: NULL argument");
    midyn_ctx* ctx = s->ctx;
    if (magnus_order < 1 || magnus_order > 3) return fail(ctx, "Only magnus_order 1, 2, and 3 are supported.");
    if (B <= 0 || m <= 0 || R <= 0 || P < 1) return fail(ctx, "midyn_expm_solve: bad sizes");
    if (s->k > 0 && !S) return fail(ctx, "midyn_expm_solve: S is NULL but the stack has operators");
    for (int i = 0; i < 3 * nsteps; ++i)
        if (step_rows[i] < 0 || step_rows[i] >= R) return fail(ctx, "midyn_expm_solve: step_rows out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int np = s->n_pad;
    if (ctx->expm_action && magnus_order <= 2 && (long long)m * 16 <= np) {
        // few columns per instance: expm(Omega) y by matrix-vector products (see expm_action_solve)
        std::vector<double> s_host;
        const double* S_h = S;
        if (s->k > 0) {
            hipPointerAttribute_t at{};
            if (hipPointerGetAttributes(&at, S) == hipSuccess && at.type == hipMemoryTypeDevice) {
                s_host.resize((size_t)B * R * s->k);
                HIPCHK(ctx, hipMemcpy(s_host.data(), S, s_host.size() * sizeof(double), hipMemcpyDeviceToHost));
                S_h = s_host.data();
            } else {
                (void)hipGetLastError();
            }
        }
        return expm_action_solve(s, B, m, R, times, S_h, S, nsteps, step_rows, step_h, step_save, P, magnus_order, y0,
                                 y0_shared, Y_out);
    }
    const int ld = round_up(m, 64);
    // Instances advance together in chunks: every generator evaluation, Magnus combination, expm
    // product and propagation is ONE batched launch over the chunk (a single instance per chunk
    // when one n x n expm already fills the device).
    const int chunk = expm_chunk(ctx, np, B);
    const size_t mat = (size_t)np * np;          // elements per matrix
    const size_t stv = (size_t)np * ld;          // elements per state block
    DevBuf d_S, d_times, d_E, d_y[2], d_tmp, d_out, G[3], W[4], Om;
    ExpmWork w;
    if (s->k > 0) {
        CHK(d_S.alloc(ctx, (size_t)B * R * s->k * sizeof(double)));
        HIPCHK(ctx, copy_to_device_any(ctx, d_S.p, S, d_S.bytes));  // host or device table
    }
    CHK(make_phase_rows(s, times, R, d_times, d_E));
    CHK(d_y[0].alloc(ctx, chunk * stv * sizeof(double2)));
    CHK(d_y[1].alloc(ctx, chunk * stv * sizeof(double2)));
    const size_t inst_elems = (size_t)s->n * m;
    CHK(d_tmp.alloc(ctx, inst_elems * sizeof(double2)));
    CHK(d_out.alloc(ctx, (size_t)chunk * P * inst_elems * sizeof(double2)));
    CHK(Om.alloc(ctx, chunk * mat * sizeof(double2)));
    for (int i = 0; i < magnus_order; ++i) CHK(G[i].alloc(ctx, chunk * mat * sizeof(double2)));
    if (magnus_order >= 2)
        for (int i = 0; i < (magnus_order == 2 ? 2 : 4); ++i) CHK(W[i].alloc(ctx, chunk * mat * sizeof(double2)));
    auto Erow = [&](int row) -> const double2* {
        return s->has_frame ? d_E.as<double2>() + (size_t)row * np : nullptr;
    };
    const long long cstride = (long long)R * s->k;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        HIPCHK(ctx, hipMemsetAsync(d_y[0].p, 0, nb * stv * sizeof(double2), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(d_y[1].p, 0, nb * stv * sizeof(double2), ctx->stream));
        for (int b = 0; b < nb; ++b) {
            const midyn_complex* y0b = y0 + (y0_shared ? 0 : (size_t)(b0 + b) * inst_elems);
            if (b == 0 || !y0_shared) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // d_tmp is reused
                HIPCHK(ctx, hipMemcpyAsync(d_tmp.p, y0b, inst_elems * sizeof(double2), hipMemcpyHostToDevice,
                                           ctx->stream));
            }
            hipLaunchKernelGGL(scatter_state_kernel, dim3(grid_for(inst_elems)), dim3(256), 0, ctx->stream,
                               d_tmp.as<double2>(), 1, 1, s->n, m, ld, (const double2*)nullptr,
                               d_y[0].as<double2>() + b * stv, (double2*)nullptr);
            hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for(inst_elems)), dim3(256), 0, ctx->stream,
                               d_y[0].as<double2>() + b * stv, 1, s->n, m, ld, P, 0,
                               d_out.as<double2>() + (size_t)b * P * inst_elems);
        }
        HIPCHK(ctx, hipGetLastError());
        int cur = 0;
        const double* coeff_b = s->k > 0 ? d_S.as<double>() + (size_t)b0 * R * s->k : nullptr;
        for (int st = 0; st < nsteps; ++st) {
            const double h = step_h[st];
            const int* rr = step_rows + 3 * st;
            auto cf = [&](int row) { return coeff_b ? coeff_b + (size_t)row * s->k : nullptr; };
            auto gen = [&](int gi, double scale, double2* out) {
                return launch_gen_eval(s, cf(rr[gi]), Erow(rr[gi]), scale, out, nb, cstride);
            };
            double2* Omega = Om.as<double2>();
            CHK(magnus_omega(ctx, np, nb, magnus_order, h, gen, G, W, Omega));
            CHK(dev_expm_inplace(ctx, w, Omega, np, nullptr, nullptr, nb));
            // y <- expm(Omega) y  for every instance of the chunk
            CHK(dev_zgemm_batched(ctx, nb, np, ld, np, Omega, np, (long long)mat, d_y[cur].as<double2>(), ld,
                                  (long long)stv, d_y[cur ^ 1].as<double2>(), ld, (long long)stv, 1.0, 0.0, nullptr));
            cur ^= 1;
            if (step_save && step_save[st] >= 0) {
                if (step_save[st] >= P) return fail(ctx, "midyn_expm_solve: save slot out of range");
                for (int b = 0; b < nb; ++b)
                    hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for(inst_elems)), dim3(256), 0, ctx->stream,
                                       d_y[cur].as<double2>() + b * stv, 1, s->n, m, ld, P, step_save[st],
                                       d_out.as<double2>() + (size_t)b * P * inst_elems);
                HIPCHK(ctx, hipGetLastError());
            }
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(Y_out + (size_t)b0 * P * inst_elems, d_out.p, (size_t)nb * P * inst_elems * sizeof(double2),
                              hipMemcpyDeviceToHost));
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Parallel-in-time propagation (fixed_step_lmde_solver_parallel_template_jax,
// solvers/fixed_step_solvers.py:524-613, with the step rules of jax_RK4_parallel_solver :222-258 and
// jax_expm_parallel_solver :289-316):
//   1. the propagators of ALL time steps of a chunk are formed by batched launches
//        RK4:   P_i = I + (k1 + 2 k2 + 2 k3 + k4)/6,  k1 = hG(t), k2 = hG(t+h/2)(I + k1/2), ...
//        expm:  P_i = expm(Omega_m(t_i, h_i))
//   2. the propagators between consecutive output times are multiplied by a binary tree, every tree
//      level being ONE batched zgemm over all pairs of all intervals (work T-1 products, depth log2),
//   3. the (few) interval propagators are applied to the state in time order.
// The reference scans all prefix products (associative_scan) and keeps the ones at t_list; only those
// are formed here.  Products are re-associated, so results agree with the sequential methods to
// rounding, not bit for bit (as in the reference, whose tests compare the two the same way).
// -------------------------------------------------------------------------------------------------
extern "C" int midyn_parallel_solve(midyn_stack* s, int B, int m, int R, const double* times, const double* S,
                                    int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                                    int P, int method, const midyn_complex* y0, int y0_shared,
                                    midyn_complex* Y_out) {
    if (!s || !Y_out || !y0 || !times || !step_rows || !step_h || !step_save)
        return fail(s ? s->ctx : nullptr, "midyn_parallel_solve: NULL argument");
    midyn_ctx* ctx = s->ctx;
    if (method < 0 || method > 3) return fail(ctx, "midyn_parallel_solve: method must be 0 (RK4) or a Magnus order 1..3");
    if (B <= 0 || m <= 0 || R <= 0 || P < 1 || nsteps < 0) return fail(ctx, "midyn_parallel_solve: bad sizes");
    if (s->k > 0 && !S) return fail(ctx, "midyn_parallel_solve: S is NULL but the stack has operators");
    for (int i = 0; i < 3 * nsteps; ++i)
        if (step_rows[i] < 0 || step_rows[i] >= R) return fail(ctx, "midyn_parallel_solve: step_rows out of range");
    for (int i = 0; i < nsteps; ++i)
        if (step_save[i] >= P) return fail(ctx, "midyn_parallel_solve: save slot out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int np = s->n_pad;
    const int ld = round_up(m, 64);
    const size_t mat = (size_t)np * np, stv = (size_t)np * ld;
    const int npts = method == 0 ? 3 : method;  // generator evaluations per step
    const int cap = std::max(1, expm_chunk(ctx, np, std::max(1, nsteps)));
    const int k = s->k;
    DevBuf d_S, d_times, d_E, d_rows, d_h, d_C[3], d_Eg[3], X, G[3], W[4], d_offs, d_y[2], d_tmp, d_out;
    ExpmWork w;
    if (k > 0) CHK(d_S.alloc(ctx, (size_t)R * k * sizeof(double)));
    CHK(make_phase_rows(s, times, R, d_times, d_E));
    // step tables, transposed to [point][step] so that a chunk of steps is contiguous
    std::vector<int> rows_t((size_t)3 * std::max(1, nsteps));
    for (int i = 0; i < nsteps; ++i)
        for (int gi = 0; gi < 3; ++gi) rows_t[(size_t)gi * nsteps + i] = step_rows[3 * i + gi];
    CHK(d_rows.alloc(ctx, rows_t.size() * sizeof(int)));
    CHK(d_h.alloc(ctx, (size_t)std::max(1, nsteps) * sizeof(double)));
    if (nsteps > 0) {
        HIPCHK(ctx, hipMemcpy(d_rows.p, rows_t.data(), rows_t.size() * sizeof(int), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(d_h.p, step_h, (size_t)nsteps * sizeof(double), hipMemcpyHostToDevice));
    }
    for (int gi = 0; gi < npts; ++gi) {
        if (k > 0) CHK(d_C[gi].alloc(ctx, (size_t)cap * k * sizeof(double)));
        if (s->has_frame) CHK(d_Eg[gi].alloc(ctx, (size_t)cap * np * sizeof(double2)));
    }
    CHK(X.alloc(ctx, 2 * (size_t)cap * mat * sizeof(double2)));
    const int n_g = method == 0 ? 3 : (method == 1 ? 0 : method);
    const int n_w = method == 0 ? 3 : (method == 2 ? 2 : (method == 3 ? 4 : 0));
    for (int i = 0; i < n_g; ++i) CHK(G[i].alloc(ctx, (size_t)cap * mat * sizeof(double2)));
    for (int i = 0; i < n_w; ++i) CHK(W[i].alloc(ctx, (size_t)cap * mat * sizeof(double2)));
    CHK(d_offs.alloc(ctx, (size_t)3 * cap * sizeof(long long)));
    CHK(d_y[0].alloc(ctx, stv * sizeof(double2)));
    CHK(d_y[1].alloc(ctx, stv * sizeof(double2)));
    const size_t inst_elems = (size_t)s->n * m;
    CHK(d_tmp.alloc(ctx, inst_elems * sizeof(double2)));
    CHK(d_out.alloc(ctx, (size_t)P * inst_elems * sizeof(double2)));
    double2* Xb = X.as<double2>();
    std::vector<long long> offs;
    std::vector<int> loc(cap), round_start;
    for (int b = 0; b < B; ++b) {
        if (k > 0)  // host or device table (midyn_sigtable_data)
            HIPCHK(ctx, copy_to_device_any(ctx, d_S.p, S + (size_t)b * R * k, (size_t)R * k * sizeof(double)));
        HIPCHK(ctx, hipMemsetAsync(d_y[0].p, 0, stv * sizeof(double2), ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(d_y[1].p, 0, stv * sizeof(double2), ctx->stream));
        if (b == 0 || !y0_shared)
            HIPCHK(ctx, hipMemcpy(d_tmp.p, y0 + (y0_shared ? 0 : (size_t)b * inst_elems), inst_elems * sizeof(double2),
                                  hipMemcpyHostToDevice));
        hipLaunchKernelGGL(scatter_state_kernel, dim3(grid_for(inst_elems)), dim3(256), 0, ctx->stream,
                           d_tmp.as<double2>(), 1, 1, s->n, m, ld, (const double2*)nullptr, d_y[0].as<double2>(),
                           (double2*)nullptr);
        hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for(inst_elems)), dim3(256), 0, ctx->stream,
                           d_y[0].as<double2>(), 1, s->n, m, ld, P, 0, d_out.as<double2>());
        HIPCHK(ctx, hipGetLastError());
        int cur = 0;
        for (int c0 = 0; c0 < nsteps; c0 += cap) {
            const int nb = std::min(cap, nsteps - c0);
            // -- 1. coefficient / phase rows of the chunk's steps, then all step propagators -> X[0..nb)
            for (int gi = 0; gi < npts; ++gi) {
                const int* rows_g = d_rows.as<int>() + (size_t)gi * nsteps + c0;
                if (k > 0)
                    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)nb * k)), dim3(256), 0, ctx->stream,
                                       d_S.as<double>(), rows_g, nb, k, d_C[gi].as<double>());
                if (s->has_frame)
                    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((size_t)nb * np * 2)), dim3(256), 0,
                                       ctx->stream, d_E.as<double>(), rows_g, nb, 2 * np, d_Eg[gi].as<double>());
            }
            HIPCHK(ctx, hipGetLastError());
            const double* hvec = d_h.as<double>() + c0;
            auto gen = [&](int gi, double scale, double2* out) {
                return launch_gen_eval(s, k > 0 ? d_C[gi].as<double>() : nullptr,
                                       s->has_frame ? d_Eg[gi].as<double2>() : nullptr, scale, out, nb, k, np, hvec);
            };
            if (method == 0) {
                double2 *k1 = G[0].as<double2>(), *gh = G[1].as<double2>(), *g1 = G[2].as<double2>();
                double2 *k2 = W[0].as<double2>(), *k3 = W[1].as<double2>(), *k4 = W[2].as<double2>();
                CHK(gen(0, 1.0, k1));
                CHK(gen(1, 1.0, gh));
                CHK(gen(2, 1.0, g1));
                CHK(dev_sqgemm(ctx, nb, np, gh, k1, k2, 0.5, 1.0, gh));  // k2 = hG(t+h/2) (I + k1/2)
                CHK(dev_sqgemm(ctx, nb, np, gh, k2, k3, 0.5, 1.0, gh));  // k3 = hG(t+h/2) (I + k2/2)
                CHK(dev_sqgemm(ctx, nb, np, g1, k3, k4, 1.0, 1.0, g1));  // k4 = hG(t+h)   (I + k3)
                const double2* xs[4] = {k1, k2, k3, k4};
                double al[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
                CHK(dev_lincomb(ctx, np, Xb, 4, xs, al, 1.0, nb));
            } else {
                CHK(magnus_omega(ctx, np, nb, method, 1.0, gen, G, W, Xb));  // generators carry h already
                CHK(dev_expm_inplace(ctx, w, Xb, np, nullptr, nullptr, nb));
            }
            // -- 2. binary-tree products inside every interval [a, e) between output times
            std::vector<std::pair<int, int>> segs;
            for (int a = 0; a < nb;) {
                int e = a;
                while (e < nb && step_save[c0 + e] < 0) ++e;
                e = std::min(nb, e + 1);
                segs.emplace_back(a, e);
                a = e;
            }
            std::fill(loc.begin(), loc.begin() + nb, 0);
            auto slot = [&](int which, int i) { return (long long)((size_t)which * cap + i) * (long long)mat; };
            int longest = 0;
            for (auto& sg : segs) longest = std::max(longest, sg.second - sg.first);
            // all levels' offset tables go to the device in ONE copy (a level has at most half the entries of
            // the one before: fewer than nb products in total)
            offs.clear();
            round_start.clear();
            for (int st = 1; st < longest; st *= 2) {
                round_start.push_back((int)(offs.size() / 3));
                for (auto& sg : segs)
                    for (int i = sg.first; i + st < sg.second; i += 2 * st) {
                        offs.push_back(slot(loc[i + st], i + st));  // later steps multiply from the left
                        offs.push_back(slot(loc[i], i));
                        offs.push_back(slot(loc[i] ^ 1, i));
                        loc[i] ^= 1;
                    }
            }
            round_start.push_back((int)(offs.size() / 3));
            if (!offs.empty()) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the previous chunk may still read d_offs
                HIPCHK(ctx, hipMemcpy(d_offs.p, offs.data(), offs.size() * sizeof(long long), hipMemcpyHostToDevice));
                for (size_t lv = 0; lv + 1 < round_start.size(); ++lv) {
                    const int cnt = round_start[lv + 1] - round_start[lv];
                    if (cnt > 0)
                        CHK(dev_zgemm_batched(ctx, cnt, np, np, np, Xb, np, 0, Xb, np, 0, Xb, np, 0, 1.0, 0.0, nullptr,
                                              d_offs.as<long long>() + (size_t)3 * round_start[lv]));
                }
            }
            // -- 3. apply the interval propagators in time order, store the states at the output times
            for (auto& sg : segs) {
                const double2* Q = Xb + slot(loc[sg.first], sg.first);
                CHK(dev_zgemm(ctx, np, ld, np, Q, np, d_y[cur].as<double2>(), ld, d_y[cur ^ 1].as<double2>(), ld, 1.0,
                              0.0, nullptr));
                cur ^= 1;
                const int sv = step_save[c0 + sg.second - 1];
                if (sv >= 0) {
                    hipLaunchKernelGGL(gather_state_kernel, dim3(grid_for(inst_elems)), dim3(256), 0, ctx->stream,
                                       d_y[cur].as<double2>(), 1, s->n, m, ld, P, sv, d_out.as<double2>());
                    HIPCHK(ctx, hipGetLastError());
                }
            }
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(Y_out + (size_t)b * P * inst_elems, d_out.p, (size_t)P * inst_elems * sizeof(double2),
                              hipMemcpyDeviceToHost));
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Perturbative (Dyson / Magnus expansion) step, parallel in time:
//   per step k:  X_k = [constant +] sum_I mono[k][I] A_I      (ArrayPolynomial.__call__,
//                                                              perturbation/array_polynomial.py:524-544)
//   Dyson:       P_k = X_k                                      (dyson_solver.py:204-207)
//   Magnus:      P_k = post . expm(X_k)                         (magnus_solver.py:122-125)
//   y <- P_{T-1} ... P_1 P_0 y                                  (perturbative_solver.py:172-192, and
//                                                              the associative scan of :195-219)
// The polynomial of ALL steps of a chunk is ONE real-by-complex GEMM  X[T][n_pad^2] = mono[T][M] .
// terms[M][n_pad^2] on the MFMA kernel (the row of step k IS its padded n_pad x n_pad matrix), then
// batched expm / post-multiplication, then the tree product of midyn_parallel_solve.
// -------------------------------------------------------------------------------------------------
struct midyn_expansion {
    midyn_ctx* ctx = nullptr;
    int n = 0, np = 0, M = 0, K = 0;   // K = padded number of GEMM rows of `terms` (M + constant)
    bool has_const = false, has_post = false, use_expm = false;
    DevBuf d_terms, d_post;
    // work buffers kept between solves (a solve of ~1000 small steps is a few hundred microseconds of
    // kernels; allocating ~0.5 GB of scratch per call would dominate it)
    int w_cap = 0;
    DevBuf X, d_mono, d_A, d_offs, d_y[2], d_tmp, d_res;   // d_y[0]: state pool, d_y[1]: per-instance half flags
    ExpmWork work;
};

extern "C" int midyn_expansion_create(midyn_ctx* ctx, int n, int M, const midyn_complex* terms,
                                      const midyn_complex* constant_term, const midyn_complex* post, int use_expm,
                                      midyn_expansion** out) {
    if (!ctx || !out || !terms || n <= 0 || M <= 0) return fail(ctx, "midyn_expansion_create: bad argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    midyn_expansion* e = new midyn_expansion();
    e->ctx = ctx;
    e->n = n;
    e->np = round_up(n, 64);
    e->M = M;
    e->has_const = constant_term != nullptr;
    e->has_post = post != nullptr;
    e->use_expm = use_expm != 0;
    e->K = round_up(M + (e->has_const ? 1 : 0), GEMM_BK);
    const size_t mat = (size_t)e->np * e->np;
    std::vector<double2> host((size_t)e->K * mat, make_double2(0.0, 0.0));
    auto put = [&](size_t slot, const midyn_complex* src) {
        for (int r = 0; r < n; ++r)
            memcpy(&host[slot * mat + (size_t)r * e->np], src + (size_t)r * n, (size_t)n * sizeof(double2));
    };
    for (int i = 0; i < M; ++i) put(i, terms + (size_t)i * n * n);
    if (e->has_const) put(M, constant_term);
    int st = e->d_terms.alloc(ctx, host.size() * sizeof(double2));
    if (!st && hipMemcpy(e->d_terms.p, host.data(), host.size() * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess)
        st = fail(ctx, "midyn_expansion_create: upload failed");
    if (!st && e->has_post) {
        st = e->d_post.alloc(ctx, mat * sizeof(double2));
        if (!st) st = hipMemset(e->d_post.p, 0, mat * sizeof(double2)) == hipSuccess ? 0 : fail(ctx, "memset");
        if (!st) st = upload_padded(ctx, post, n, n, e->d_post.as<double2>(), e->np);
    }
    if (st) {
        delete e;
        return st;
    }
    *out = e;
    return 0;
}

extern "C" int midyn_expansion_destroy(midyn_expansion* e) {
    if (!e) return 0;
    hipSetDevice(e->ctx->device);
    hipStreamSynchronize(e->ctx->stream);
    delete e;
    return 0;
}

extern "C" int midyn_expansion_solve(midyn_expansion* e, int B, int nsteps, const double* mono, int m,
                                     const midyn_complex* y0, int y0_shared, midyn_complex* Y_out) {
    if (!e || !mono || !y0 || !Y_out) return fail(e ? e->ctx : nullptr, "midyn_expansion_solve: NULL argument");
    midyn_ctx* ctx = e->ctx;
    if (B <= 0 || nsteps < 0 || m <= 0) return fail(ctx, "midyn_expansion_solve: bad sizes");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int np = e->np, n = e->n, M = e->M, K = e->K;
    const int ldm = round_up(m, 64);                 // columns per instance on the device
    const size_t mat = (size_t)np * np;
    const size_t inst_elems = (size_t)n * m;
    // Instances are processed in groups whose states fit ~2 GB; inside a group ALL (instance, step)
    // pairs are rows of one long table that is cut into chunks of `cap` rows, whatever instance
    // they belong to: a sweep of many short solves fills the device like one long solve does.
    const int Bg = (int)std::max<long long>(1, std::min<long long>(B, ((long long)2 << 30) / (long long)(np * (size_t)ldm * 32)));
    int cap = round_up(expm_chunk(ctx, np, std::max(1, std::min(B, Bg) * std::max(1, nsteps))), 64);
    DevBuf &X = e->X, &d_mono = e->d_mono, &d_A = e->d_A, &d_offs = e->d_offs, &d_tmp = e->d_tmp, &d_res = e->d_res;
    DevBuf& Ypool = e->d_y[0];
    DevBuf& d_flags = e->d_y[1];
    ExpmWork& w = e->work;
    if (cap > e->w_cap) {
        CHK(X.alloc(ctx, 2 * (size_t)cap * mat * sizeof(double2)));
        CHK(d_mono.alloc(ctx, (size_t)cap * M * sizeof(double)));
        CHK(d_A.alloc(ctx, (size_t)cap * K * sizeof(double2)));
        CHK(d_offs.alloc(ctx, (size_t)3 * 3 * cap * sizeof(long long)));  // all tree levels (< 2 cap) + applications
        e->w_cap = cap;
    } else {
        cap = e->w_cap;  // slot layout of X follows the allocated capacity
    }
    const int ldy = Bg * ldm;                       // leading dimension of the state pool
    const size_t half = (size_t)np * ldy;           // elements of one ping-pong half
    if (Ypool.bytes < 2 * half * sizeof(double2)) CHK(Ypool.alloc(ctx, 2 * half * sizeof(double2)));
    if (d_flags.bytes < (size_t)Bg * sizeof(int)) CHK(d_flags.alloc(ctx, (size_t)Bg * sizeof(int)));
    if (d_tmp.bytes < (size_t)Bg * inst_elems * sizeof(double2)) CHK(d_tmp.alloc(ctx, (size_t)Bg * inst_elems * sizeof(double2)));
    if (d_res.bytes < (size_t)Bg * inst_elems * sizeof(double2)) CHK(d_res.alloc(ctx, (size_t)Bg * inst_elems * sizeof(double2)));
    double2* Xb = X.as<double2>();
    double2* Yb = Ypool.as<double2>();
    std::vector<long long> offs;
    std::vector<int> loc(cap), ycur(Bg), level_start, level_cnt;
    auto slot = [&](int which, int i) { return (long long)((size_t)which * cap + i) * (long long)mat; };
    auto ystate = [&](int which, int b) { return (long long)((size_t)which * half + (size_t)b * ldm); };
    for (int g0 = 0; g0 < B; g0 += Bg) {
        const int gb = std::min(Bg, B - g0);
        HIPCHK(ctx, hipMemsetAsync(Ypool.p, 0, 2 * half * sizeof(double2), ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(d_tmp.p, y0 + (y0_shared ? 0 : (size_t)g0 * inst_elems),
                                   (y0_shared ? 1 : gb) * inst_elems * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(scatter_padded_kernel, dim3(grid_for((size_t)gb * inst_elems)), dim3(256), 0, ctx->stream,
                           d_tmp.as<double2>(), y0_shared ? 1 : 0, gb, n, m, ldm, ldy, Yb);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // d_tmp / host y0 may be reused
        std::fill(ycur.begin(), ycur.begin() + gb, 0);
        const long long rows = (long long)gb * nsteps;
        for (long long r0 = 0; r0 < rows; r0 += cap) {
            const int nb = (int)std::min<long long>(cap, rows - r0);
            const int T = round_up(nb, 64);
            // -- 1. monomial rows of the chunk -> complex GEMM operand (imaginary part exactly zero)
            HIPCHK(ctx, copy_to_device_any(ctx, d_mono.p, mono + ((size_t)g0 * nsteps + (size_t)r0) * M,
                                           (size_t)nb * M * sizeof(double)));
            hipLaunchKernelGGL(mono_operand_kernel, dim3(grid_for((size_t)T * K)), dim3(256), 0, ctx->stream,
                               d_mono.as<double>(), nb, M, e->has_const ? 1 : 0, T, K, d_A.as<double2>());
            HIPCHK(ctx, hipGetLastError());
            // -- 2. all step matrices in one GEMM: row k of the product is the padded matrix of step k
            CHK(dev_zgemm_batched(ctx, 1, T, (int)mat, K, d_A.as<double2>(), K, 0, e->d_terms.as<double2>(), (int)mat, 0,
                                  Xb, (int)mat, 0, 1.0, 0.0, nullptr, nullptr, true));
            std::fill(loc.begin(), loc.begin() + nb, 0);
            if (e->use_expm) {
                CHK(dev_expm_inplace(ctx, w, Xb, np, nullptr, nullptr, nb));
                if (e->has_post) {  // P_k = post . expm(X_k) -> second slot set
                    CHK(dev_zgemm_batched(ctx, nb, np, np, np, e->d_post.as<double2>(), np, 0, Xb, np, (long long)mat,
                                          Xb + slot(1, 0), np, (long long)mat, 1.0, 0.0, nullptr));
                    std::fill(loc.begin(), loc.begin() + nb, 1);
                }
            }
            // -- 3. segments = runs of rows of one instance; tree product inside every segment, then ONE
            //       batched application of the segment products to their instances' states.  All
            //       offset tables of the chunk go to the device in one copy.
            std::vector<std::pair<int, int>> segs;
            for (int a = 0; a < nb;) {
                const long long inst = (r0 + a) / nsteps;
                const int eidx = (int)std::min<long long>(nb, (inst + 1) * nsteps - r0);
                segs.emplace_back(a, eidx);
                a = eidx;
            }
            int longest = 0;
            for (auto& sg : segs) longest = std::max(longest, sg.second - sg.first);
            offs.clear();
            level_start.clear();
            level_cnt.clear();
            for (int st = 1; st < longest; st *= 2) {
                const size_t before = offs.size();
                for (auto& sg : segs)
                    for (int i = sg.first; i + st < sg.second; i += 2 * st) {
                        offs.push_back(slot(loc[i + st], i + st));  // later steps multiply from the left
                        offs.push_back(slot(loc[i], i));
                        offs.push_back(slot(loc[i] ^ 1, i));
                        loc[i] ^= 1;
                    }
                level_start.push_back((int)(before / 3));
                level_cnt.push_back((int)((offs.size() - before) / 3));
            }
            const int apply_start = (int)(offs.size() / 3);
            for (auto& sg : segs) {
                const int bl = (int)((r0 + sg.first) / nsteps);   // instance within the group
                offs.push_back(slot(loc[sg.first], sg.first));
                offs.push_back(ystate(ycur[bl], bl));
                offs.push_back(ystate(ycur[bl] ^ 1, bl));
                ycur[bl] ^= 1;
            }
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the previous chunk may still read d_offs
            HIPCHK(ctx, hipMemcpy(d_offs.p, offs.data(), offs.size() * sizeof(long long), hipMemcpyHostToDevice));
            const long long* dof = d_offs.as<long long>();
            for (size_t lv = 0; lv < level_cnt.size(); ++lv)
                if (level_cnt[lv] > 0)
                    CHK(dev_zgemm_batched(ctx, level_cnt[lv], np, np, np, Xb, np, 0, Xb, np, 0, Xb, np, 0, 1.0, 0.0, nullptr,
                                          dof + (size_t)3 * level_start[lv]));
            CHK(dev_zgemm_batched(ctx, (int)segs.size(), np, ldm, np, Xb, np, 0, Yb, ldy, 0, Yb, ldy, 0, 1.0, 0.0, nullptr,
                                  dof + (size_t)3 * apply_start));
        }
        HIPCHK(ctx, hipMemcpyAsync(d_flags.p, ycur.data(), (size_t)gb * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(gather_padded_kernel, dim3(grid_for((size_t)gb * inst_elems)), dim3(256), 0, ctx->stream, Yb,
                           d_flags.as<int>(), half, gb, n, m, ldm, ldy, d_res.as<double2>());
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(Y_out + (size_t)g0 * inst_elems, d_res.p, (size_t)gb * inst_elems * sizeof(double2),
                              hipMemcpyDeviceToHost));
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// Non-vectorised Lindblad RHS with n x n zgemms (SURVEY section 8 row f2;
// LindbladCollection.evaluate_rhs, models/operator_collections.py:451-567):
//     rhs = (A + B) rho + rho (A - B) + sum_j N_j rho N_j^+ + sum_j gamma_j(t) L_j rho L_j^+
//     B = -i H(t),  A = -1/2 sum N^+N - 1/2 sum gamma_j L_j^+ L_j
// The caller passes two ordinary operator stacks that share one coefficient vector c = (s, gamma):
//     left  = A + B = [ -iH_d - 1/2 sum N^+N ;  -iH_j ;  -1/2 L_j^+L_j ]
//     right = A - B = [ +iH_d - 1/2 sum N^+N ;  +iH_j ;  -1/2 L_j^+L_j ]
// and the dissipators N_j (coefficient 1) followed by L_j.  In a rotating frame (frame basis)
//     rhs = conj(e_a) e_b o R( e_a conj(e_b) o rho ),  e = exp(d t)   (lindblad_model.py:477-538).
// Per evaluation: 2 gen_eval passes (HBM bound) + (2 + 2 n_diss) zgemm of n^3 (MFMA bound).
// -------------------------------------------------------------------------------------------------
struct midyn_lindblad {
    midyn_ctx* ctx = nullptr;
    midyn_stack* left = nullptr;
    midyn_stack* right = nullptr;
    int n = 0, np = 0, k = 0, k_h = 0, n_static = 0, n_dyn = 0;
    DevBuf diss, diss_adj;           // [n_static + n_dyn][np][np]
    DevBuf ML, MR, Xp, T, R, Y, Yt, K[4], coeff, E;
};

extern "C" int midyn_lindblad_create(midyn_stack* left, midyn_stack* right, int k_h, int n_static, int n_dyn,
                                     const midyn_complex* dissipators, midyn_lindblad** out) {
    if (!left || !right || !out) return fail(nullptr, "midyn_lindblad_create: NULL argument");
    midyn_ctx* ctx = left->ctx;
    if (right->ctx != ctx || right->n != left->n || right->k != left->k)
        return fail(ctx, "midyn_lindblad_create: left/right stacks do not match");
    if (k_h < 0 || n_static < 0 || n_dyn < 0 || k_h + n_dyn != left->k)
        return fail(ctx, "midyn_lindblad_create: k_h + n_dyn must equal the number of stack operators");
    if (n_static + n_dyn > 0 && !dissipators) return fail(ctx, "midyn_lindblad_create: dissipators is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    midyn_lindblad* L = new midyn_lindblad();
    L->ctx = ctx;
    L->left = left;
    L->right = right;
    L->n = left->n;
    L->np = left->n_pad;
    L->k = left->k;
    L->k_h = k_h;
    L->n_static = n_static;
    L->n_dyn = n_dyn;
    const int nd = n_static + n_dyn;
    const size_t mat = (size_t)L->np * L->np * sizeof(double2);
    int st = 0;
    auto guard = [&](int r) { if (r && !st) st = r; };
    if (nd > 0) {
        guard(L->diss.alloc(ctx, mat * nd));
        guard(L->diss_adj.alloc(ctx, mat * nd));
    }
    for (DevBuf* b : {&L->ML, &L->MR, &L->Xp, &L->T, &L->R, &L->Y, &L->Yt, &L->K[0], &L->K[1], &L->K[2], &L->K[3]})
        guard(b->alloc(ctx, mat));
    guard(L->coeff.alloc(ctx, std::max(1, L->k) * sizeof(double)));
    guard(L->E.alloc(ctx, (size_t)L->np * sizeof(double2)));
    if (st) {
        delete L;
        return st;
    }
    if (nd > 0) {
        hipMemset(L->diss.p, 0, mat * nd);
        hipMemset(L->diss_adj.p, 0, mat * nd);
        std::vector<midyn_complex> adj((size_t)L->n * L->n);
        for (int j = 0; j < nd; ++j) {
            const midyn_complex* src = dissipators + (size_t)j * L->n * L->n;
            for (int a = 0; a < L->n; ++a)
                for (int b = 0; b < L->n; ++b) {
                    adj[(size_t)b * L->n + a].re = src[(size_t)a * L->n + b].re;
                    adj[(size_t)b * L->n + a].im = -src[(size_t)a * L->n + b].im;
                }
            if (upload_padded(ctx, src, L->n, L->n, L->diss.as<double2>() + (size_t)j * L->np * L->np, L->np) ||
                upload_padded(ctx, adj.data(), L->n, L->n, L->diss_adj.as<double2>() + (size_t)j * L->np * L->np, L->np)) {
                delete L;
                return 1;
            }
        }
    }
    *out = L;
    return 0;
}

extern "C" int midyn_lindblad_destroy(midyn_lindblad* L) {
    if (!L) return 0;
    hipSetDevice(L->ctx->device);
    hipStreamSynchronize(L->ctx->stream);
    delete L;
    return 0;
}

// out = rhs(t, X) on device buffers (np x np, ld np); coefficients c_host (k) are also copied to the device
static int lindblad_rhs_dev(midyn_lindblad* L, const double* c_host, double t, const double2* X, double2* out) {
    midyn_ctx* ctx = L->ctx;
    const int np = L->np;
    const bool framed = L->left->has_frame;
    if (L->k > 0)
        HIPCHK(ctx, hipMemcpyAsync(L->coeff.p, c_host, L->k * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    const double2* Xin = X;
    if (framed) {
        // phases for this time: a one-row phase table written by phase_table_kernel
        double* d_t = reinterpret_cast<double*>(L->T.p);  // scratch: T is free here
        HIPCHK(ctx, hipMemcpyAsync(d_t, &t, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(phase_table_kernel, dim3(grid_for(np)), dim3(256), 0, ctx->stream, L->left->frame_im, d_t, np,
                           1, L->E.as<double2>());
        hipLaunchKernelGGL(frame_mask_kernel, dim3(grid_for((size_t)np * np)), dim3(256), 0, ctx->stream, X,
                           L->E.as<double2>(), np, +1, L->Xp.as<double2>());
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // &t / c_host are caller stack memory
        Xin = L->Xp.as<double2>();
    } else if (L->k > 0) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    const double* dc = L->k > 0 ? L->coeff.as<double>() : nullptr;
    CHK(launch_gen_eval(L->left, dc, nullptr, 1.0, L->ML.as<double2>()));
    CHK(launch_gen_eval(L->right, dc, nullptr, 1.0, L->MR.as<double2>()));
    double2* R = L->R.as<double2>();
    CHK(dev_zgemm(ctx, np, np, np, L->ML.as<double2>(), np, Xin, np, R, np, 1.0, 0.0, nullptr));
    CHK(dev_zgemm(ctx, np, np, np, Xin, np, L->MR.as<double2>(), np, R, np, 1.0, 1.0, R));
    const size_t mat = (size_t)np * np;
    for (int j = 0; j < L->n_static + L->n_dyn; ++j) {
        const double gam = j < L->n_static ? 1.0 : c_host[L->k_h + (j - L->n_static)];
        if (gam == 0.0) continue;
        CHK(dev_zgemm(ctx, np, np, np, L->diss.as<double2>() + j * mat, np, Xin, np, L->T.as<double2>(), np, 1.0, 0.0,
                      nullptr));
        CHK(dev_zgemm(ctx, np, np, np, L->T.as<double2>(), np, L->diss_adj.as<double2>() + j * mat, np, R, np, gam, 1.0, R));
    }
    if (framed) {
        hipLaunchKernelGGL(frame_mask_kernel, dim3(grid_for((size_t)np * np)), dim3(256), 0, ctx->stream, R,
                           L->E.as<double2>(), np, -1, out);
        HIPCHK(ctx, hipGetLastError());
    } else {
        HIPCHK(ctx, hipMemcpyAsync(out, R, mat * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
    }
    return 0;
}

extern "C" int midyn_lindblad_rhs(midyn_lindblad* L, const double* coeffs, double t, const midyn_complex* rho, int batch,
                                  midyn_complex* out) {
    if (!L || !rho || !out || batch <= 0) return fail(L ? L->ctx : nullptr, "midyn_lindblad_rhs: bad argument");
    midyn_ctx* ctx = L->ctx;
    if (L->k > 0 && !coeffs) return fail(ctx, "midyn_lindblad_rhs: coeffs is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t nn = (size_t)L->n * L->n;
    for (int b = 0; b < batch; ++b) {
        HIPCHK(ctx, hipMemsetAsync(L->Y.p, 0, L->Y.bytes, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        CHK(upload_padded(ctx, rho + b * nn, L->n, L->n, L->Y.as<double2>(), L->np));
        CHK(lindblad_rhs_dev(L, coeffs, t, L->Y.as<double2>(), L->K[0].as<double2>()));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy2D(out + b * nn, (size_t)L->n * sizeof(double2), L->K[0].p, (size_t)L->np * sizeof(double2),
                                (size_t)L->n * sizeof(double2), L->n, hipMemcpyDeviceToHost));
    }
    return 0;
}

// RK4 for a sweep of density matrices: the instances of a chunk advance TOGETHER -- per RHS evaluation
// two batched generator evaluations (own coefficient rows), 2 + 2 n_diss batched zgemms, two batched
// frame masks, whatever the number of instances; coefficient table and phase rows are device resident
// (no host round trip per evaluation).
extern "C" int midyn_lindblad_rk4_solve(midyn_lindblad* L, int B, int R, const double* times, const double* S,
                                        int nsteps, const int* step_rows, const double* step_h, const int* step_save,
                                        int P, const midyn_complex* rho0, int rho0_shared, midyn_complex* out) {
    if (!L || !times || !step_rows || !step_h || !rho0 || !out)
        return fail(L ? L->ctx : nullptr, "midyn_lindblad_rk4_solve: NULL argument");
    midyn_ctx* ctx = L->ctx;
    if (B <= 0 || R <= 0 || P < 1) return fail(ctx, "midyn_lindblad_rk4_solve: bad sizes");
    if (L->k > 0 && !S) return fail(ctx, "midyn_lindblad_rk4_solve: S is NULL");
    for (int i = 0; i < 3 * nsteps; ++i)
        if (step_rows[i] < 0 || step_rows[i] >= R) return fail(ctx, "midyn_lindblad_rk4_solve: step_rows out of range");
    for (int i = 0; i < nsteps; ++i)
        if (step_save && step_save[i] >= P) return fail(ctx, "midyn_lindblad_rk4_solve: save slot out of range");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int np = L->np, n = L->n, k = L->k;
    const size_t nn = (size_t)n * n, mat = (size_t)np * np;
    const bool framed = L->left->has_frame;
    const int nd = L->n_static + L->n_dyn;
    // instances per chunk: 11 matrices each, ~2 GB of scratch, at most 4096
    const int chunk = (int)std::max<long long>(1, std::min<long long>(std::min(B, 4096), ((long long)2 << 30) / (long long)(mat * 16 * 11)));
    DevBuf d_S, d_times, d_E, ML, MR, Xp, T, Rb, Y, Yt, K[4], d_in, d_out;
    if (k > 0) {
        CHK(d_S.alloc(ctx, (size_t)B * R * k * sizeof(double)));
        HIPCHK(ctx, copy_to_device_any(ctx, d_S.p, S, d_S.bytes));
    }
    CHK(make_phase_rows(L->left, times, R, d_times, d_E));
    for (DevBuf* bp : {&ML, &MR, &Xp, &T, &Rb, &Y, &Yt, &K[0], &K[1], &K[2], &K[3]}) CHK(bp->alloc(ctx, (size_t)chunk * mat * sizeof(double2)));
    CHK(d_in.alloc(ctx, (size_t)chunk * nn * sizeof(double2)));
    CHK(d_out.alloc(ctx, (size_t)chunk * P * nn * sizeof(double2)));
    const long long cstride = (long long)R * k;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = std::min(chunk, B - b0);
        // initial states -> padded device matrices (one strided copy per instance block via a staging buffer)
        HIPCHK(ctx, hipMemsetAsync(Y.p, 0, (size_t)nb * mat * sizeof(double2), ctx->stream));
        if (rho0_shared) {
            for (int b = 0; b < nb; ++b) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                CHK(upload_padded(ctx, rho0, n, n, Y.as<double2>() + (size_t)b * mat, np));
            }
        } else {
            HIPCHK(ctx, hipMemcpyAsync(d_in.p, rho0 + (size_t)b0 * nn, (size_t)nb * nn * sizeof(double2), hipMemcpyHostToDevice,
                                       ctx->stream));
            for (int b = 0; b < nb; ++b)
                hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(nn)), dim3(256), 0, ctx->stream, d_in.as<double2>() + (size_t)b * nn,
                                   n, Y.as<double2>() + (size_t)b * mat, np, n, n);
            HIPCHK(ctx, hipGetLastError());
        }
        hipLaunchKernelGGL(save_density_kernel, dim3(grid_for((size_t)nb * nn)), dim3(256), 0, ctx->stream, Y.as<double2>(), np,
                           n, nb, P, 0, d_out.as<double2>());
        const double* Sc = k > 0 ? d_S.as<double>() + (size_t)b0 * cstride : nullptr;
        // rhs of the whole chunk at table row `row`
        auto rhs = [&](int row, const double2* X, double2* dst) -> int {
            const double2* Erow = framed ? d_E.as<double2>() + (size_t)row * np : nullptr;
            const double2* Xin = X;
            if (framed) {
                hipLaunchKernelGGL(frame_mask_batch_kernel, dim3(grid_for((size_t)nb * mat)), dim3(256), 0, ctx->stream, X, Erow,
                                   np, +1, nb, Xp.as<double2>());
                Xin = Xp.as<double2>();
            }
            const double* cf = Sc ? Sc + (size_t)row * k : nullptr;
            CHK(launch_gen_eval(L->left, cf, nullptr, 1.0, ML.as<double2>(), nb, cstride));
            CHK(launch_gen_eval(L->right, cf, nullptr, 1.0, MR.as<double2>(), nb, cstride));
            double2* Rr = Rb.as<double2>();
            CHK(dev_sqgemm(ctx, nb, np, ML.as<double2>(), Xin, Rr, 1.0, 0.0, nullptr));
            CHK(dev_sqgemm(ctx, nb, np, Xin, MR.as<double2>(), Rr, 1.0, 1.0, Rr));
            for (int j = 0; j < nd; ++j) {
                // T_b = N_j X_b (shared left operand), [T_b *= gamma_b,j], R_b += T_b N_j^+ (shared right operand)
                CHK(dev_zgemm_batched(ctx, nb, np, np, np, L->diss.as<double2>() + (size_t)j * mat, np, 0, Xin, np, (long long)mat,
                                      T.as<double2>(), np, (long long)mat, 1.0, 0.0, nullptr));
                if (j >= L->n_static) {
                    hipLaunchKernelGGL(scale_batch_kernel, dim3(grid_for((size_t)nb * mat)), dim3(256), 0, ctx->stream,
                                       T.as<double2>(), mat, nb, cf + L->k_h + (j - L->n_static), cstride);
                }
                CHK(dev_zgemm_batched(ctx, nb, np, np, np, T.as<double2>(), np, (long long)mat,
                                      L->diss_adj.as<double2>() + (size_t)j * mat, np, 0, Rr, np, (long long)mat, 1.0, 1.0, Rr));
            }
            if (framed)
                hipLaunchKernelGGL(frame_mask_batch_kernel, dim3(grid_for((size_t)nb * mat)), dim3(256), 0, ctx->stream, Rr, Erow,
                                   np, -1, nb, dst);
            else
                HIPCHK(ctx, hipMemcpyAsync(dst, Rr, (size_t)nb * mat * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
            HIPCHK(ctx, hipGetLastError());
            return 0;
        };
        double2 *y = Y.as<double2>(), *yt = Yt.as<double2>();
        double2* kk[4] = {K[0].as<double2>(), K[1].as<double2>(), K[2].as<double2>(), K[3].as<double2>()};
        for (int st = 0; st < nsteps; ++st) {
            const double h = step_h[st];
            const int* rr = step_rows + 3 * st;
            // fixed_step_solvers.py:62-73
            CHK(rhs(rr[0], y, kk[0]));
            {
                const double2* xs[2] = {y, kk[0]};
                double al[2] = {1.0, 0.5 * h};
                CHK(dev_lincomb(ctx, np, yt, 2, xs, al, 0.0, nb));
            }
            CHK(rhs(rr[1], yt, kk[1]));
            {
                const double2* xs[2] = {y, kk[1]};
                double al[2] = {1.0, 0.5 * h};
                CHK(dev_lincomb(ctx, np, yt, 2, xs, al, 0.0, nb));
            }
            CHK(rhs(rr[1], yt, kk[2]));
            {
                const double2* xs[2] = {y, kk[2]};
                double al[2] = {1.0, h};
                CHK(dev_lincomb(ctx, np, yt, 2, xs, al, 0.0, nb));
            }
            CHK(rhs(rr[2], yt, kk[3]));
            {
                const double2* xs[4] = {kk[0], kk[1], kk[2], kk[3]};
                double al[4] = {1.0, 2.0, 2.0, 1.0};
                CHK(dev_lincomb(ctx, np, yt, 4, xs, al, 0.0, nb));
                const double2* ys[2] = {y, yt};
                double bl[2] = {1.0, (1.0 / 6) * h};
                CHK(dev_lincomb(ctx, np, y, 2, ys, bl, 0.0, nb));
            }
            if (step_save && step_save[st] >= 0) {
                hipLaunchKernelGGL(save_density_kernel, dim3(grid_for((size_t)nb * nn)), dim3(256), 0, ctx->stream, y, np, n, nb,
                                   P, step_save[st], d_out.as<double2>());
                HIPCHK(ctx, hipGetLastError());
            }
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipMemcpy(out + (size_t)b0 * P * nn, d_out.p, (size_t)nb * P * nn * sizeof(double2), hipMemcpyDeviceToHost));
    }
    return 0;
}

// -------------------------------------------------------------------------------------------------
// micro-benchmarks (measured ceilings printed next to the vendor peaks)
// -------------------------------------------------------------------------------------------------
extern "C" int midyn_microbench(midyn_ctx* ctx, const char* name, double* out) {
    if (!ctx || !name || !out) return fail(ctx, "midyn_microbench: NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::string n(name);
    hipEvent_t e0, e1;
    HIPCHK(ctx, hipEventCreate(&e0));
    HIPCHK(ctx, hipEventCreate(&e1));
    DevBuf sink;
    CHK(sink.alloc(ctx, 64));
    float ms = 0.f;
    if (n == "mfma_f64" || n == "mfma_f64_w1" || n == "mfma_f64_w2" || n == "mfma_f64_w2a16") {
        // 256-thread blocks = one wave per SIMD each; `wps` of them per CU -> waves per SIMD
        const int wps = n == "mfma_f64" ? 4 : (n == "mfma_f64_w1" ? 1 : 2);
        const int blocks = ctx->num_cu * wps, iters = 4000;
        const bool a16 = n == "mfma_f64_w2a16";
        auto launch = [&](int it) {
            if (a16) hipLaunchKernelGGL(mfma_peak_kernel<16>, dim3(blocks), dim3(256), 0, ctx->stream, sink.as<double>(), it);
            else hipLaunchKernelGGL(mfma_peak_kernel<8>, dim3(blocks), dim3(256), 0, ctx->stream, sink.as<double>(), it);
        };
        launch(100);
        HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
        launch(iters);
        HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(e1));
        HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)blocks * 4 /*waves*/ * iters * (a16 ? 32 : 16) * 2048.0;
        out[0] = flops / (ms * 1e-3) / 1e12;  // TFLOP/s
    } else if (n == "hbm_read" || n == "mall_read") {
        const size_t bytes = (n == "hbm_read") ? ((size_t)4 << 30) : ((size_t)144 << 20);
        DevBuf buf;
        CHK(buf.alloc(ctx, bytes));
        HIPCHK(ctx, hipMemsetAsync(buf.p, 1, bytes, ctx->stream));
        const int reps = (n == "hbm_read") ? 4 : 40;
        hipLaunchKernelGGL(stream_read_kernel, dim3(ctx->num_cu * 8), dim3(256), 0, ctx->stream,
                           buf.as<double2>(), bytes / 16, sink.as<double>());
        HIPCHK(ctx, hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < reps; ++r)
            hipLaunchKernelGGL(stream_read_kernel, dim3(ctx->num_cu * 8), dim3(256), 0, ctx->stream,
                               buf.as<double2>(), bytes / 16, sink.as<double>());
        HIPCHK(ctx, hipEventRecord(e1, ctx->stream));
        HIPCHK(ctx, hipEventSynchronize(e1));
        HIPCHK(ctx, hipEventElapsedTime(&ms, e0, e1));
        out[0] = (double)bytes * reps / (ms * 1e-3) / 1e9;  // GB/s
    } else {
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        return fail(ctx, "midyn_microbench: unknown benchmark " + n);
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    HIPCHK(ctx, hipGetLastError());
    return 0;
}
