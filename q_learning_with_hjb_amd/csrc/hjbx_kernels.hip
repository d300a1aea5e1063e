// hjbx_kernels.hip -- gfx950 kernels + C ABI (include/hjbx.h) for batched control-affine rollouts
// and HJB residuals.  MI355X only: wave64, one lane per environment, state vectors in VGPRs,
// system/task constants in SGPRs (kernarg), row-vector global accesses, wave-shuffle reductions.
//
// All of these kernels are HBM-bandwidth bound (a few dozen flops + one sincos per 36-128 bytes);
// algorithmic byte counts per environment are tabulated in DESIGN.md.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"
#include "hjbx_host.hpp"
#include "hjbx_stream_kernels.hpp"

using namespace hjbx;
static_assert(kRolloutTerminate == HJBX_ROLLOUT_TERMINATE && kRolloutStopAtTarget == HJBX_ROLLOUT_STOP_AT_TARGET, "rollout flags");

// ----------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int hjbx_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// process-wide knobs (include/hjbx.h: hjbx_option)
static std::atomic<int> g_options[5] = {{0}, {0}, {0}, {0}, {0}};   // (HJBX_OPT_MLP_ARITHMETIC defaults to 0 = float32 MFMA, the reference's arithmetic)
int hjbx_option_value(int option) { return (option >= 0 && option < 5) ? g_options[option].load(std::memory_order_relaxed) : 0; }

#define HJBX_REQUIRE(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) return hjbx_set_error(HJBX_EINVAL, __VA_ARGS__); \
    } while (0)

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "%s: %s", what, hipGetErrorString(e));
    return HJBX_OK;
}

// The kernels themselves are the device function templates of hjbx_stream_kernels.hpp (shared with the run-time compiled user systems of
// hjbx_user.hip); here: their __global__ wrappers for the built-in systems, and the host side of the C ABI.
static inline dim3 grid_for(int64_t B) { return dim3((unsigned)((B + kBlock - 1) / kBlock)); }
static inline dim3 grid_rows(int64_t B, int R) { return dim3((unsigned)((B + (int64_t)kBlock * R - 1) / ((int64_t)kBlock * R))); }
// rows per thread of the streaming kernels: enough loads in flight to cover the HBM latency once the batch fills the chip; small
// batches keep one row per thread (more workgroups); float64 keeps one row (its row state alone is 2x the registers)
template <typename T> static inline int rows_per_thread(int64_t B, int n, bool reducing = false) {
    if (sizeof(T) != 4) return 1;
    const int forced = hjbx_option_value(HJBX_OPT_STREAM_ROWS);
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // measured with buffers rotated through 640 MB (tools/kernel_bench.py --rows 1|2|4, B = 2^20): one row per thread is fastest for
    // simulate / vhjb_step (more rows cost resident waves: near-hover vhjb_step 32 -> 46 -> 64 us), two rows help the residual
    // kernel, whose grid is capped for the in-kernel reduction (cartpole 22 -> 19 us)
    (void)n;
    return reducing && B >= (1 << 18) ? 2 : 1;
}


template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_affine(S sys, const T* __restrict__ x, T* __restrict__ f1,
                                                   T* __restrict__ f2, int64_t B) {
    k_affine_body<S, T>(sys, x, f1, f2, B);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_wrap(S sys, const T* x, T* out, int64_t B) {
    k_wrap_body<S, T>(sys, x, out, B);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_xdot(S sys, const T* __restrict__ x, const T* __restrict__ u,
                                                 T* __restrict__ xd, int64_t B) {
    k_xdot_body<S, T>(sys, x, u, xd, B);
}

template <int INTEG, int R, typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_simulate(S sys, Limits<T, S::M> lim, const T* x, const T* __restrict__ u,
                                                     T* xn, int64_t B) {
    k_simulate_body<INTEG, R, S, T>(sys, lim, x, u, xn, B);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_initial_state(S sys, X0P<S, T> p, const T* __restrict__ u01,
                                                          T* __restrict__ x0, int64_t B) {
    k_initial_state_body<S, T>(sys, p, u01, x0, B);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_running_cost(S sys, TaskP<T, S::N, S::M> tk, const T* __restrict__ x,
                                                         const T* __restrict__ u, T* __restrict__ cost, int64_t B) {
    k_running_cost_body<S, T>(sys, tk, x, u, cost, B);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_termination_cost(S sys, TaskP<T, S::N, S::M> tk, const T* __restrict__ x,
                                                             T* __restrict__ cost, int64_t B) {
    k_termination_cost_body<S, T>(sys, tk, x, cost, B);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_control_from_grad(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim,
                                                              const T* __restrict__ x, const T* __restrict__ g,
                                                              T* __restrict__ u, int64_t B) {
    k_control_from_grad_body<S, T>(sys, tk, lim, x, g, u, B);
}

template <int CK, typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_controller(S sys, CtrlP<T, S::N, S::M> c, Limits<T, S::M> lim,
                                                       const T* __restrict__ x, T* __restrict__ u, int64_t B) {
    k_controller_body<CK, S, T>(sys, c, lim, x, u, B);
}

template <int MODE, int R, typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_hjb_residual(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim,
                                                         const T* __restrict__ x, const T* __restrict__ g,
                                                         const T* __restrict__ done, T* __restrict__ loss_i,
                                                         T* __restrict__ dl_dg, unsigned char* ws, T* __restrict__ sums, int64_t B) {
    k_hjb_residual_body<MODE, R, S, T>(sys, tk, lim, x, g, done, loss_i, dl_dg, ws, sums, B);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_termination_residual(T eps, const T* __restrict__ V, const T* __restrict__ cost,
                                                                 const T* __restrict__ done, T* __restrict__ loss_i,
                                                                 T* __restrict__ dl_dV, unsigned char* ws, T* __restrict__ sums, int64_t B) {
    k_termination_residual_body<T>(eps, V, cost, done, loss_i, dl_dV, ws, sums, B);
}

template <int INTEG, int R, typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_vhjb_step(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim, int t, int T_max,
                                                      const T* x, const T* __restrict__ g, T* xn, T* __restrict__ u_out,
                                                      T* __restrict__ cost_t, T* __restrict__ done_t,
                                                      int32_t* __restrict__ done_step, T* __restrict__ resid_t, int64_t B) {
    k_vhjb_step_body<INTEG, R, S, T>(sys, tk, lim, t, T_max, x, g, xn, u_out, cost_t, done_t, done_step, resid_t, B);
}

template <int INTEG, int CK, typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_rollout_feedback(S sys, TaskP<T, S::N, S::M> tk, CtrlP<T, S::N, S::M> c,
                                                             Limits<T, S::M> lim, uint32_t flags, int has_task, int T_steps,
                                                             const T* __restrict__ x0, T* __restrict__ traj,
                                                             T* __restrict__ u_log, T* __restrict__ cost,
                                                             int32_t* __restrict__ done_step, T* __restrict__ total_cost,
                                                             T* __restrict__ x_final, int64_t B) {
    k_rollout_feedback_body<INTEG, CK, S, T>(sys, tk, c, lim, flags, has_task, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B);
}

// ----------------------------------------------------------------------------------------------
// user-defined systems (HJBX_SYS_USER, hjbx_system_create_from_source): the same kernel bodies, compiled at run time for the user's struct
// (hjbx_user.hip / hjbx_user_kernels.hpp).  Here: the typed kernel arguments for the handle's (n, m) and the launch by name.
// ----------------------------------------------------------------------------------------------
template <typename T> struct UserBlob { T p[HJBX_USER_MAX_PARAMS]; };   // the kernel's first argument is `struct { T p[n_params]; }`
template <typename T> static UserBlob<T> user_blob(const hjbx_system* s) {
    UserBlob<T> b;
    for (int i = 0; i < HJBX_USER_MAX_PARAMS; ++i) b.p[i] = i < s->n_params ? (T)s->p[i] : T(0);
    return b;
}
template <int N_> struct DimsOnly { static constexpr int N = N_; };      // X0P<S, T> depends on S::N only
template <typename T> struct UName {
    char buf[64];
    explicit UName(const char* base) { snprintf(buf, sizeof buf, "hjbx_u_%s_%s", base, sizeof(T) == 4 ? "f32" : "f64"); }
    operator const char*() const { return buf; }
};
// calls f(integral_constant<int, n>, integral_constant<int, m>) for the handle's dimensions
template <typename F> static int with_user_dims(const hjbx_system* s, F&& f) {
#define HJBX_UD(NN)                                                              \
    case NN:                                                                     \
        if (s->m == 1) return f(std::integral_constant<int, NN>{}, std::integral_constant<int, 1>{}); \
        if (s->m == 2) return f(std::integral_constant<int, NN>{}, std::integral_constant<int, 2>{}); \
        if (s->m == 3) return f(std::integral_constant<int, NN>{}, std::integral_constant<int, 3>{}); \
        break;
    switch (s->n) { HJBX_UD(1) HJBX_UD(2) HJBX_UD(3) HJBX_UD(4) HJBX_UD(5) HJBX_UD(6) HJBX_UD(7) HJBX_UD(8) HJBX_UD(9) HJBX_UD(10) }
#undef HJBX_UD
    return hjbx_set_error(HJBX_EUNSUPPORTED, "user system with n=%d m=%d", s->n, s->m);
}
#define HJBX_USER(sys, ...) \
    if ((sys)->kind == HJBX_SYS_USER) return with_user_dims(sys, [&](auto Nc, auto Mc) -> int { \
        constexpr int N = decltype(Nc)::value, M = decltype(Mc)::value; (void)N; (void)M;      \
        auto blob = user_blob<T>(sys);                                                         \
        __VA_ARGS__                                                                            \
    })
static unsigned ugrid(int64_t B) { return (unsigned)((B + kBlock - 1) / kBlock); }

// host side: descriptor conversion and dispatch live in hjbx_host.hpp (shared with hjbx_mlp.hip)

static int unsupported(const hjbx_system* s) {
    return hjbx_set_error(HJBX_EUNSUPPORTED, "no kernel for system kind %d with n=%d m=%d", s->kind, s->n, s->m);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// a (B, cols) row-major buffer is accessed with RowIO's vector width: 16, 8 or 4 bytes
static bool aligned_rows(const void* p, size_t row_bytes) {
    const uintptr_t a = (row_bytes % 16 == 0) ? 15u : (row_bytes % 8 == 0) ? 7u : 3u;
    return (reinterpret_cast<uintptr_t>(p) & a) == 0;
}

#define HJBX_CHECK_COMMON(sys, B)                                                   \
    HJBX_REQUIRE((sys) != nullptr, "system handle is NULL");                        \
    HJBX_REQUIRE((B) >= 0, "negative batch size %lld", (long long)(B));             \
    if ((B) == 0) return HJBX_OK;

// ROWS(p, cols): non-NULL (B, cols) buffer of T aligned for its row vector width; OPT: may be NULL
#define HJBX_CHECK_ROWS(p, cols) \
    HJBX_REQUIRE((p) != nullptr && aligned_rows(p, (size_t)(cols) * sizeof(T)), #p " must be a non-NULL device pointer aligned to its row vector width")
#define HJBX_CHECK_OPT(p, cols) \
    HJBX_REQUIRE((p) == nullptr || aligned_rows(p, (size_t)(cols) * sizeof(T)), #p " must be aligned to its row vector width")

// ---- typed implementations ---------------------------------------------------------------------
template <typename T> static int affine_impl(const hjbx_system* sys, const T* x, T* f1, T* f2, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(f1, sys->n); HJBX_CHECK_ROWS(f2, sys->n * sys->m);
    HJBX_USER(sys, void* a[] = {&blob, (void*)&x, (void*)&f1, (void*)&f2, (void*)&B}; return hjbx_user_launch(sys, UName<T>("affine"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            hipLaunchKernelGGL((k_affine<decltype(S), T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, x, f1, f2, B);
        })) return unsupported(sys);
    return check_launch("hjbx_affine");
}

template <typename T> static int wrap_impl(const hjbx_system* sys, const T* x, T* out, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(out, sys->n);
    HJBX_USER(sys, void* a[] = {&blob, (void*)&x, (void*)&out, (void*)&B}; return hjbx_user_launch(sys, UName<T>("wrap"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            hipLaunchKernelGGL((k_wrap<decltype(S), T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, x, out, B);
        })) return unsupported(sys);
    return check_launch("hjbx_wrap");
}

template <typename T> static int xdot_impl(const hjbx_system* sys, const T* x, const T* u, T* xd, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m); HJBX_CHECK_ROWS(xd, sys->n);
    HJBX_USER(sys, void* a[] = {&blob, (void*)&x, (void*)&u, (void*)&xd, (void*)&B}; return hjbx_user_launch(sys, UName<T>("xdot"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            hipLaunchKernelGGL((k_xdot<decltype(S), T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, x, u, xd, B);
        })) return unsupported(sys);
    return check_launch("hjbx_dynamics_step");
}

template <typename T>
static int simulate_impl(const hjbx_system* sys, int integ, const T* x, const T* u, T* xn, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m); HJBX_CHECK_ROWS(xn, sys->n);
    if (int rc = check_integrator(sys, integ, "hjbx_simulate")) return rc;
    HJBX_USER(sys, auto lim = make_limits<T, M>(sys); void* a[] = {&blob, &lim, (void*)&x, (void*)&u, (void*)&xn, (void*)&B};
              return hjbx_user_launch(sys, UName<T>(integ == HJBX_RK4 ? "simulate_i1" : "simulate_i0"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto lim = make_limits<T, SS::M>(sys);
            auto go = [&](auto integc, auto rc) {
                constexpr int I = decltype(integc)::value, R = decltype(rc)::value;
                hipLaunchKernelGGL((k_simulate<I, R, SS, T>), grid_rows(B, R), dim3(kBlock), 0, (hipStream_t)st, S, lim, x, u, xn, B);
            };
            auto with_r = [&](auto integc) {
                const int R = rows_per_thread<T>(B, SS::N);
                if (R == 4) go(integc, std::integral_constant<int, 4>{});
                else if (R == 2) go(integc, std::integral_constant<int, 2>{});
                else go(integc, std::integral_constant<int, 1>{});
            };
            if (integ == HJBX_EULER) with_r(std::integral_constant<int, 0>{});
            else if (integ == HJBX_RK4) with_r(std::integral_constant<int, 1>{});
            else if constexpr (is_linear<SS>::value) with_r(std::integral_constant<int, 2>{});
        })) return unsupported(sys);
    return check_launch("hjbx_simulate");
}

template <typename T>
static int initial_state_impl(const hjbx_system* sys, const double* mean, const double* sd, const T* u01, T* x0, int64_t B,
                              void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(u01, sys->n); HJBX_CHECK_ROWS(x0, sys->n);
    HJBX_REQUIRE(mean && sd, "x0_mean / x0_std are NULL");
    HJBX_USER(sys, X0P<DimsOnly<N>, T> p; for (int i = 0; i < N; ++i) { p.mean[i] = (T)mean[i]; p.std[i] = (T)sd[i]; }
              void* a[] = {&blob, &p, (void*)&u01, (void*)&x0, (void*)&B}; return hjbx_user_launch(sys, UName<T>("initial_state"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            X0P<SS, T> p;
            for (int i = 0; i < SS::N; ++i) { p.mean[i] = (T)mean[i]; p.std[i] = (T)sd[i]; }
            hipLaunchKernelGGL((k_initial_state<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, p, u01, x0, B);
        })) return unsupported(sys);
    return check_launch("hjbx_initial_state");
}

template <typename T>
static int running_cost_impl(const hjbx_system* sys, const hjbx_task* task, const T* x, const T* u, T* cost, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc; HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m); HJBX_CHECK_ROWS(cost, 1);
    HJBX_USER(sys, auto tk = make_task<T, N, M>(task); void* a[] = {&blob, &tk, (void*)&x, (void*)&u, (void*)&cost, (void*)&B};
              return hjbx_user_launch(sys, UName<T>("running_cost"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            hipLaunchKernelGGL((k_running_cost<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S,
                               make_task<T, SS::N, SS::M>(task), x, u, cost, B);
        })) return unsupported(sys);
    return check_launch("hjbx_running_cost");
}

template <typename T>
static int termination_cost_impl(const hjbx_system* sys, const hjbx_task* task, const T* x, T* cost, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc; HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(cost, 1);
    HJBX_USER(sys, auto tk = make_task<T, N, M>(task); void* a[] = {&blob, &tk, (void*)&x, (void*)&cost, (void*)&B};
              return hjbx_user_launch(sys, UName<T>("termination_cost"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            hipLaunchKernelGGL((k_termination_cost<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S,
                               make_task<T, SS::N, SS::M>(task), x, cost, B);
        })) return unsupported(sys);
    return check_launch("hjbx_termination_cost");
}

template <typename T>
static int control_from_grad_impl(const hjbx_system* sys, const hjbx_task* task, const T* x, const T* g, T* u, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc; HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(g, sys->n); HJBX_CHECK_ROWS(u, sys->m);
    HJBX_USER(sys, auto tk = make_task<T, N, M>(task); auto lim = make_limits<T, M>(sys);
              void* a[] = {&blob, &tk, &lim, (void*)&x, (void*)&g, (void*)&u, (void*)&B}; return hjbx_user_launch(sys, UName<T>("control_from_grad"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            hipLaunchKernelGGL((k_control_from_grad<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S,
                               make_task<T, SS::N, SS::M>(task), make_limits<T, SS::M>(sys), x, g, u, B);
        })) return unsupported(sys);
    return check_launch("hjbx_control_from_grad");
}

static inline int reduce_grid(int64_t B) {
    int64_t g = (B + kBlock - 1) / kBlock;
    return (int)(g < kReduceBlocks ? g : kReduceBlocks);
}

template <typename T>
static int hjb_residual_impl(const hjbx_system* sys, const hjbx_task* task, int mode, const T* x, const T* g, const T* done,
                             T* loss_i, T* dl_dg, T* sums, void* workspace, int64_t B, void* st) {
    HJBX_REQUIRE(sys != nullptr, "system handle is NULL");
    HJBX_REQUIRE(B >= 0, "negative batch size");
    if (int rc = check_task(task)) return rc;
    HJBX_REQUIRE(mode == HJBX_RESIDUAL_NORMALISED || mode == HJBX_RESIDUAL_RAW, "unknown residual mode %d", mode);
    HJBX_REQUIRE(!sums || (workspace && aligned16(workspace)), "sums requested but workspace is NULL/unaligned");
    if (B == 0) {
        if (sums) {
            hipError_t e = hipMemsetAsync(sums, 0, 3 * sizeof(T), (hipStream_t)st);
            if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return HJBX_OK;
    }
    HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(g, sys->n); HJBX_CHECK_ROWS(done, 1); HJBX_CHECK_OPT(loss_i, 1); HJBX_CHECK_OPT(dl_dg, sys->n);
    const int grid = reduce_grid(B);
    unsigned char* ws = sums ? (unsigned char*)workspace : nullptr;
    HJBX_USER(sys, auto tk = make_task<T, N, M>(task); auto lim = make_limits<T, M>(sys);
              void* a[] = {&blob, &tk, &lim, (void*)&x, (void*)&g, (void*)&done, (void*)&loss_i, (void*)&dl_dg, (void*)&ws, (void*)&sums, (void*)&B};
              return hjbx_user_launch(sys, UName<T>(mode == HJBX_RESIDUAL_RAW ? "hjb_residual_m1" : "hjb_residual_m0"), (unsigned)grid, a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto tk = make_task<T, SS::N, SS::M>(task);
            auto lim = make_limits<T, SS::M>(sys);
            auto go = [&](auto modec, auto rc) {
                constexpr int MD = decltype(modec)::value, R = decltype(rc)::value;
                hipLaunchKernelGGL((k_hjb_residual<MD, R, SS, T>), dim3(grid), dim3(kBlock), 0, (hipStream_t)st, S, tk, lim, x, g, done, loss_i, dl_dg,
                                   ws, sums, B);
            };
            auto with_r = [&](auto modec) {
                const int R = rows_per_thread<T>(B, SS::N, true);
                if (R == 4) go(modec, std::integral_constant<int, 4>{});
                else if (R == 2) go(modec, std::integral_constant<int, 2>{});
                else go(modec, std::integral_constant<int, 1>{});
            };
            if (mode == HJBX_RESIDUAL_NORMALISED) with_r(std::integral_constant<int, 0>{});
            else with_r(std::integral_constant<int, 1>{});
        })) return unsupported(sys);
    return check_launch("hjbx_hjb_residual");
}

template <typename T>
static int termination_residual_impl(double eps, const T* V, const T* cost, const T* done, T* loss_i, T* dl_dV, T* sums,
                                     void* workspace, int64_t B, void* st) {
    HJBX_REQUIRE(B >= 0, "negative batch size");
    HJBX_REQUIRE(!sums || (workspace && aligned16(workspace)), "sums requested but workspace is NULL/unaligned");
    if (B == 0) {
        if (sums) {
            hipError_t e = hipMemsetAsync(sums, 0, 3 * sizeof(T), (hipStream_t)st);
            if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return HJBX_OK;
    }
    HJBX_REQUIRE(V && cost && done, "V/cost/done must be non-NULL");
    const int grid = reduce_grid(B);
    unsigned char* ws = sums ? (unsigned char*)workspace : nullptr;
    hipLaunchKernelGGL((k_termination_residual<T>), dim3(grid), dim3(kBlock), 0, (hipStream_t)st, (T)eps, V, cost, done,
                       loss_i, dl_dV, ws, sums, B);
    return check_launch("hjbx_termination_residual");
}

template <typename T>
static int vhjb_step_impl(const hjbx_system* sys, const hjbx_task* task, int integ, int t, int T_max, const T* x, const T* g,
                          T* xn, T* u_out, T* cost_t, T* done_t, int32_t* done_step, T* resid_t, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc;
    if (int rc = check_integrator(sys, integ, "hjbx_vhjb_step")) return rc;
    HJBX_REQUIRE(t >= 0 && T_max >= 0, "negative step index");
    HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(g, sys->n); HJBX_CHECK_ROWS(xn, sys->n); HJBX_CHECK_OPT(u_out, sys->m);
    HJBX_REQUIRE(cost_t && done_t && done_step, "cost_t/done_t/done_step must be non-NULL");
    HJBX_USER(sys, auto tk = make_task<T, N, M>(task); auto lim = make_limits<T, M>(sys);
              void* a[] = {&blob, &tk, &lim, (void*)&t, (void*)&T_max, (void*)&x, (void*)&g, (void*)&xn, (void*)&u_out, (void*)&cost_t, (void*)&done_t,
                           (void*)&done_step, (void*)&resid_t, (void*)&B};
              return hjbx_user_launch(sys, UName<T>(integ == HJBX_RK4 ? "vhjb_step_i1" : "vhjb_step_i0"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto tk = make_task<T, SS::N, SS::M>(task);
            auto lim = make_limits<T, SS::M>(sys);
            auto go = [&](auto integc, auto rc) {
                constexpr int I = decltype(integc)::value, R = decltype(rc)::value;
                hipLaunchKernelGGL((k_vhjb_step<I, R, SS, T>), grid_rows(B, R), dim3(kBlock), 0, (hipStream_t)st, S, tk, lim, t, T_max, x,
                                   g, xn, u_out, cost_t, done_t, done_step, resid_t, B);
            };
            auto with_r = [&](auto integc) {
                const int R = rows_per_thread<T>(B, SS::N);
                if (R == 4) go(integc, std::integral_constant<int, 4>{});
                else if (R == 2) go(integc, std::integral_constant<int, 2>{});
                else go(integc, std::integral_constant<int, 1>{});
            };
            if (integ == HJBX_EULER) with_r(std::integral_constant<int, 0>{});
            else if (integ == HJBX_RK4) with_r(std::integral_constant<int, 1>{});
            else if constexpr (is_linear<SS>::value) with_r(std::integral_constant<int, 2>{});
        })) return unsupported(sys);
    return check_launch("hjbx_vhjb_step");
}

template <typename S> struct is_cartpole { static constexpr bool value = false; };
template <typename T> struct is_cartpole<Cartpole<T>> { static constexpr bool value = true; };
template <typename S> struct is_acrobot { static constexpr bool value = false; };
template <typename T> struct is_acrobot<Acrobot<T>> { static constexpr bool value = true; };

static int check_ctrl(const hjbx_system* sys, const hjbx_controller* c) {
    HJBX_REQUIRE(c, "controller is NULL");
    HJBX_REQUIRE(c->kind >= HJBX_CTRL_LINEAR_FEEDBACK && c->kind <= HJBX_CTRL_DI_TIME_OPTIMAL, "unknown controller kind %d", c->kind);
    if (c->kind == HJBX_CTRL_DI_TIME_OPTIMAL && !(sys->kind == HJBX_SYS_LINEAR && sys->n == 2 && sys->m == 1))
        return hjbx_set_error(HJBX_EINVAL, "the time-optimal bang-bang controller needs the double integrator (LINEAR, n=2, m=1)");
    if (sys->kind == HJBX_SYS_USER && c->kind != HJBX_CTRL_LINEAR_FEEDBACK)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "user-defined systems take the linear feedback controller only");
    if (c->kind == HJBX_CTRL_CARTPOLE_ENERGY && sys->kind != HJBX_SYS_CARTPOLE)
        return hjbx_set_error(HJBX_EINVAL, "cartpole energy-shaping controller needs a cartpole system");
    if (c->kind == HJBX_CTRL_ACROBOT_ENERGY && sys->kind != HJBX_SYS_ACROBOT)
        return hjbx_set_error(HJBX_EINVAL, "acrobot energy-shaping controller needs an acrobot system");
    return HJBX_OK;
}

template <typename T>
static int controller_impl(const hjbx_system* sys, const hjbx_controller* c, const T* x, T* u, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B);
    if (int rc = check_ctrl(sys, c)) return rc;
    HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m);
    HJBX_USER(sys, auto cp = make_ctrl<T, N, M>(c); auto lim = make_limits<T, M>(sys);
              void* a[] = {&blob, &cp, &lim, (void*)&x, (void*)&u, (void*)&B}; return hjbx_user_launch(sys, UName<T>("controller"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto cp = make_ctrl<T, SS::N, SS::M>(c);
            auto lim = make_limits<T, SS::M>(sys);
            if constexpr (is_cartpole<SS>::value) {
                if (c->kind == HJBX_CTRL_CARTPOLE_ENERGY) {
                    hipLaunchKernelGGL((k_controller<1, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
                    return;
                }
            }
            if constexpr (is_di<SS>::value) {
                if (c->kind == HJBX_CTRL_DI_TIME_OPTIMAL) {
                    hipLaunchKernelGGL((k_controller<3, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
                    return;
                }
            }
            if constexpr (is_acrobot<SS>::value) {
                if (c->kind == HJBX_CTRL_ACROBOT_ENERGY) {
                    hipLaunchKernelGGL((k_controller<2, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
                    return;
                }
            }
            hipLaunchKernelGGL((k_controller<0, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
        })) return unsupported(sys);
    return check_launch("hjbx_controller");
}

template <int INTEG, typename SS, typename T>
static void launch_rollout(const hjbx_system* sys, SS S, const hjbx_task* task, const hjbx_controller* c, uint32_t flags,
                           int T_steps, const T* x0, T* traj, T* u_log, T* cost, int32_t* done_step, T* total_cost, T* x_final,
                           int64_t B, void* st) {
    auto tk = make_task<T, SS::N, SS::M>(task);
    auto cp = make_ctrl<T, SS::N, SS::M>(c);
    auto lim = make_limits<T, SS::M>(sys);
    const int has_task = task != nullptr;
#define HJBX_LAUNCH_RO(CK)                                                                                                    \
    hipLaunchKernelGGL((k_rollout_feedback<INTEG, CK, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, tk, cp, lim, \
                       flags, has_task, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B)
    if constexpr (is_cartpole<SS>::value) {
        if (c->kind == HJBX_CTRL_CARTPOLE_ENERGY) { HJBX_LAUNCH_RO(1); return; }
    }
    if constexpr (is_di<SS>::value) {
        if (c->kind == HJBX_CTRL_DI_TIME_OPTIMAL) { HJBX_LAUNCH_RO(3); return; }
    }
    if constexpr (is_acrobot<SS>::value) {
        if (c->kind == HJBX_CTRL_ACROBOT_ENERGY) { HJBX_LAUNCH_RO(2); return; }
    }
    HJBX_LAUNCH_RO(0);
#undef HJBX_LAUNCH_RO
}

template <typename T>
static int rollout_feedback_impl(const hjbx_system* sys, const hjbx_task* task, const hjbx_controller* c, int integ, uint32_t flags,
                                 int T_steps, const T* x0, T* traj, T* u_log, T* cost, int32_t* done_step, T* total_cost,
                                 T* x_final, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B);
    if (int rc = check_ctrl(sys, c)) return rc;
    if (int rc = check_integrator(sys, integ, "hjbx_rollout_feedback")) return rc;
    HJBX_REQUIRE(T_steps >= 0, "negative horizon");
    HJBX_REQUIRE((flags & ~(HJBX_ROLLOUT_TERMINATE | HJBX_ROLLOUT_STOP_AT_TARGET)) == 0, "unknown rollout flags 0x%x", flags);
    HJBX_REQUIRE(task || !(flags & HJBX_ROLLOUT_TERMINATE), "HJBX_ROLLOUT_TERMINATE needs a task");
    if (task) { if (int rc = check_task(task)) return rc; }
    HJBX_REQUIRE(task || (!cost && !total_cost), "cost outputs need a task");
    HJBX_CHECK_ROWS(x0, sys->n); HJBX_CHECK_OPT(traj, sys->n); HJBX_CHECK_OPT(u_log, sys->m); HJBX_CHECK_OPT(x_final, sys->n);
    HJBX_USER(sys, auto tk = make_task<T, N, M>(task); auto cp = make_ctrl<T, N, M>(c); auto lim = make_limits<T, M>(sys);
              int has_task = task != nullptr;
              void* a[] = {&blob, &tk, &cp, &lim, (void*)&flags, &has_task, (void*)&T_steps, (void*)&x0, (void*)&traj, (void*)&u_log, (void*)&cost,
                           (void*)&done_step, (void*)&total_cost, (void*)&x_final, (void*)&B};
              return hjbx_user_launch(sys, UName<T>(integ == HJBX_RK4 ? "rollout_feedback_i1" : "rollout_feedback_i0"), ugrid(B), a, st););
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            if (integ == HJBX_EULER)
                launch_rollout<0, SS, T>(sys, S, task, c, flags, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B, st);
            else if (integ == HJBX_RK4)
                launch_rollout<1, SS, T>(sys, S, task, c, flags, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B, st);
            else if constexpr (is_linear<SS>::value)
                launch_rollout<2, SS, T>(sys, S, task, c, flags, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B, st);
        })) return unsupported(sys);
    return check_launch("hjbx_rollout_feedback");
}

// ----------------------------------------------------------------------------------------------
// extern "C" surface
// ----------------------------------------------------------------------------------------------
extern "C" {

int hjbx_version(void) { return HJBX_VERSION; }

size_t hjbx_last_error(char* buf, size_t buflen) {
    const size_t len = strlen(g_err);
    if (buf && buflen) {
        const size_t ncopy = len < buflen - 1 ? len : buflen - 1;
        memcpy(buf, g_err, ncopy);
        buf[ncopy] = '\0';
    }
    return len;
}

int hjbx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int hjbx_set_option(int option, int value) {
    HJBX_REQUIRE(option == HJBX_OPT_ROLLOUT_SCHEDULE || option == HJBX_OPT_ROLLOUT_EXTRA_WORKGROUPS || option == HJBX_OPT_STREAM_ROWS || option == HJBX_OPT_MLP_ARITHMETIC || option == HJBX_OPT_TRAIN_KERNEL, "unknown option %d", option);
    HJBX_REQUIRE(option != HJBX_OPT_TRAIN_KERNEL || value <= 1, "train kernel must be 0 (cooperative single kernel) or 1 (the round-2 pair), got %d", value);
    HJBX_REQUIRE(option != HJBX_OPT_MLP_ARITHMETIC || value <= 2, "mlp arithmetic must be 0 (f32 MFMA), 1 (bf16x3-split MFMA) or 2 (f16x2-split MFMA), got %d", value);
    HJBX_REQUIRE(option != HJBX_OPT_ROLLOUT_SCHEDULE || value <= 1, "rollout schedule must be 0 or 1, got %d", value);
    HJBX_REQUIRE(value <= 64, "option value %d out of range", value);
    return value < 0 ? g_options[option].load() : g_options[option].exchange(value);
}

size_t hjbx_reduce_workspace_bytes(void) { return kCounterBytes + (size_t)kReduceBlocks * 3 * sizeof(double); }

int hjbx_system_create(int kind, int n, int m, double dt, const double* umin, const double* umax, const double* params,
                       int n_params, hjbx_system** out) {
    HJBX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    HJBX_REQUIRE(umin && umax && params, "umin/umax/params must be non-NULL");
    HJBX_REQUIRE(dt > 0 && std::isfinite(dt), "dt must be positive and finite");
    int en = 0, em = 0, ep = 0;
    switch (kind) {
    case HJBX_SYS_LINEAR:
        HJBX_REQUIRE(n >= 1 && n <= HJBX_MAX_N && m >= 1 && m <= HJBX_MAX_M, "linear system needs 1<=n<=%d, 1<=m<=%d", HJBX_MAX_N, HJBX_MAX_M);
        en = n; em = m; ep = n * n + n * m; break;
    case HJBX_SYS_CARTPOLE: en = 4; em = 1; ep = 4; break;
    case HJBX_SYS_ACROBOT: en = 4; em = 1; ep = 7; break;
    case HJBX_SYS_QUAD2D: en = 6; em = 2; ep = 4; break;
    case HJBX_SYS_NEARHOVER: en = 10; em = 3; ep = 4; break;
    default: return hjbx_set_error(HJBX_EINVAL, "unknown system kind %d", kind);
    }
    HJBX_REQUIRE(n == en && m == em, "system kind %d has n=%d m=%d, got n=%d m=%d", kind, en, em, n, m);
    HJBX_REQUIRE(n_params == ep || (kind == HJBX_SYS_LINEAR && n_params == 2 * ep),
                 "system kind %d takes %d parameters%s, got %d", kind, ep, kind == HJBX_SYS_LINEAR ? " (or twice that with Ad, Bd)" : "", n_params);
    for (int j = 0; j < m; ++j) HJBX_REQUIRE(umin[j] <= umax[j], "umin[%d] > umax[%d]", j, j);
    hjbx_system* s = new (std::nothrow) hjbx_system();
    if (!s) return hjbx_set_error(HJBX_EINVAL, "out of host memory");
    memset(s, 0, sizeof(*s));
    s->kind = kind; s->n = n; s->m = m; s->dt = dt; s->n_params = n_params;
    for (int j = 0; j < m; ++j) { s->umin[j] = umin[j]; s->umax[j] = umax[j]; }
    for (int i = 0; i < n_params; ++i) s->p[i] = params[i];
    *out = s;
    return HJBX_OK;
}

void hjbx_system_destroy(hjbx_system* sys) {
    if (sys && sys->user) hjbx_user_release(sys->user);
    delete sys;
}

int hjbx_dims(const hjbx_system* sys, int* n, int* m) {
    HJBX_REQUIRE(sys && n && m, "NULL argument");
    *n = sys->n; *m = sys->m;
    return HJBX_OK;
}

#define HJBX_DEFINE(T, SFX)                                                                                                   \
    int hjbx_affine_##SFX(const hjbx_system* s, const T* x, T* f1, T* f2, int64_t B, void* st) { return affine_impl<T>(s, x, f1, f2, B, st); } \
    int hjbx_wrap_##SFX(const hjbx_system* s, const T* x, T* o, int64_t B, void* st) { return wrap_impl<T>(s, x, o, B, st); }  \
    int hjbx_dynamics_step_##SFX(const hjbx_system* s, const T* x, const T* u, T* xd, int64_t B, void* st) { return xdot_impl<T>(s, x, u, xd, B, st); } \
    int hjbx_simulate_##SFX(const hjbx_system* s, int integ, const T* x, const T* u, T* xn, int64_t B, void* st) { return simulate_impl<T>(s, integ, x, u, xn, B, st); } \
    int hjbx_initial_state_##SFX(const hjbx_system* s, const double* mean, const double* sd, const T* u01, T* x0, int64_t B, void* st) { return initial_state_impl<T>(s, mean, sd, u01, x0, B, st); } \
    int hjbx_running_cost_##SFX(const hjbx_system* s, const hjbx_task* t, const T* x, const T* u, T* c, int64_t B, void* st) { return running_cost_impl<T>(s, t, x, u, c, B, st); } \
    int hjbx_termination_cost_##SFX(const hjbx_system* s, const hjbx_task* t, const T* x, T* c, int64_t B, void* st) { return termination_cost_impl<T>(s, t, x, c, B, st); } \
    int hjbx_control_from_grad_##SFX(const hjbx_system* s, const hjbx_task* t, const T* x, const T* g, T* u, int64_t B, void* st) { return control_from_grad_impl<T>(s, t, x, g, u, B, st); } \
    int hjbx_hjb_residual_##SFX(const hjbx_system* s, const hjbx_task* t, int mode, const T* x, const T* g, const T* done, T* li, T* dg, T* sums, void* ws, int64_t B, void* st) { return hjb_residual_impl<T>(s, t, mode, x, g, done, li, dg, sums, ws, B, st); } \
    int hjbx_termination_residual_##SFX(double eps, const T* V, const T* cost, const T* done, T* li, T* dV, T* sums, void* ws, int64_t B, void* st) { return termination_residual_impl<T>(eps, V, cost, done, li, dV, sums, ws, B, st); } \
    int hjbx_vhjb_step_##SFX(const hjbx_system* s, const hjbx_task* t, int integ, int step, int T_max, const T* x, const T* g, T* xn, T* uo, T* c, T* d, int32_t* ds, T* rs, int64_t B, void* st) { return vhjb_step_impl<T>(s, t, integ, step, T_max, x, g, xn, uo, c, d, ds, rs, B, st); } \
    int hjbx_controller_##SFX(const hjbx_system* s, const hjbx_controller* c, const T* x, T* u, int64_t B, void* st) { return controller_impl<T>(s, c, x, u, B, st); } \
    int hjbx_rollout_feedback_##SFX(const hjbx_system* s, const hjbx_task* t, const hjbx_controller* c, int integ, uint32_t flags, int T_steps, const T* x0, T* traj, T* ul, T* cost, int32_t* ds, T* tc, T* xf, int64_t B, void* st) { return rollout_feedback_impl<T>(s, t, c, integ, flags, T_steps, x0, traj, ul, cost, ds, tc, xf, B, st); }

HJBX_DEFINE(float, f32)
HJBX_DEFINE(double, f64)

}  // extern "C"
