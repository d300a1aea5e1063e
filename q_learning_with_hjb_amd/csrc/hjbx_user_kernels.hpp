// hjbx_user_kernels.hpp -- the translation unit hiprtc compiles for a USER-DEFINED system (hjbx_system_create_from_source, include/hjbx.h):
// the reference's open plugin surface.  Any subclass of the reference's `Dynamics` may define get_M / get_C / get_G / get_B and inherit
// get_control_affine_matrix (dynamics/dynamics_basic.py:64-94), or override get_control_affine_matrix itself (linear.py:20-22,
// quadrotors.py:17-46); here the subclass hands over the same methods as a device-code snippet and gets the library's streaming kernels
// (hjbx_stream_kernels.hpp -- the very code the five built-in systems run) instantiated for it at run time.
//
// Defined by the host before this file is compiled:  HJBX_USER_N, HJBX_USER_M (state / control dimension), HJBX_USER_NP (number of
// parameters, p[0..NP-1]), HJBX_USER_KIND (0 = affine, 1 = manipulator), and the in-memory header "hjbx_user_snippet.hpp" = the user's
// member functions, written for a scalar type `T` (float and double are both instantiated):
//   both kinds     HJBX_DEV void wrap(T* x) const                                       Dynamics.states_wrap for ONE state (in place)
//   kind 0         HJBX_DEV void affine(const T* x, T* f1, T* f2) const                 f1 (N), f2 (N x M row-major)
//   kind 1         HJBX_DEV void get_M(const T* x, T* Mq) const                         (D x D row-major, D = N / 2, symmetric positive definite)
//                  HJBX_DEV void get_C(const T* x, T* Cq) const                         (D x D)
//                  HJBX_DEV void get_G(const T* x, T* Gq) const                         (D)
//                  HJBX_DEV void get_B(T* Bq) const                                     (D x M)
// The snippet may use p[i], T, N, M, the helpers of hjbx_systems.hpp (sincos_t, tan_t, wrap_angle, sqrt_t, ...) and plain C++.
#pragma once
#include "hjbx_stream_kernels.hpp"

namespace hjbx {

template <typename T> struct UserSystem {
    static constexpr int N = HJBX_USER_N, M = HJBX_USER_M;
    static constexpr bool kHasZoh = false;
    T p[HJBX_USER_NP];   // the ONLY data member: the host builds this struct as a plain array of HJBX_USER_NP values

#include "hjbx_user_snippet.hpp"

#if HJBX_USER_KIND == 1
    // dynamics_basic.py:78-92: q, dq = x[:D], x[D:];  f1 = [dq; -inv(M) (C dq + G)],  f2 = [0; inv(M) B].  M is a mass matrix (symmetric
    // positive definite): the two solves share one Gauss-Jordan elimination without pivoting instead of forming inv(M).
    HJBX_DEV void affine(const T* x, T* f1, T* f2) const {
        constexpr int D = N / 2;
        static_assert(N % 2 == 0, "a manipulator state is (q, dq)");
        T Mq[D * D], Cq[D * D], Gq[D], Bq[D * M], rhs[D * (M + 1)];
        get_M(x, Mq);
        get_C(x, Cq);
        get_G(x, Gq);
        get_B(Bq);
#pragma unroll
        for (int i = 0; i < D; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < D; ++j) acc += Cq[i * D + j] * x[D + j];
            rhs[i * (M + 1)] = acc + Gq[i];
#pragma unroll
            for (int j = 0; j < M; ++j) rhs[i * (M + 1) + 1 + j] = Bq[i * M + j];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const T ip = T(1) / Mq[k * D + k];
#pragma unroll
            for (int j = 0; j < D; ++j) Mq[k * D + j] *= ip;
#pragma unroll
            for (int j = 0; j < M + 1; ++j) rhs[k * (M + 1) + j] *= ip;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                if (i == k) continue;
                const T f = Mq[i * D + k];
#pragma unroll
                for (int j = 0; j < D; ++j) Mq[i * D + j] -= f * Mq[k * D + j];
#pragma unroll
                for (int j = 0; j < M + 1; ++j) rhs[i * (M + 1) + j] -= f * rhs[k * (M + 1) + j];
            }
        }
#pragma unroll
        for (int i = 0; i < D; ++i) {
            f1[i] = x[D + i];
            f1[D + i] = -rhs[i * (M + 1)];
#pragma unroll
            for (int j = 0; j < M; ++j) {
                f2[i * M + j] = T(0);
                f2[(D + i) * M + j] = rhs[i * (M + 1) + 1 + j];
            }
        }
    }
#endif

    // Dynamics.dynamics_step (dynamics_basic.py:101-103): f1 + f2 @ u
    HJBX_DEV void xdot(const T* x, const T* u, T* xd) const {
        T f1[N], f2[N * M];
        affine(x, f1, f2);
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < M; ++j) acc += f2[i * M + j] * u[j];
            xd[i] = f1[i] + acc;
        }
    }
};

}  // namespace hjbx

using namespace hjbx;

// extern "C" kernels (names are looked up by hjbx_user.hip): one row per thread, the integrator / residual mode in the name
#define HJBX_U_KERNELS(T, SFX)                                                                                                              \
    using U_##SFX = UserSystem<T>;                                                                                                          \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_affine_##SFX(U_##SFX s, const T* x, T* f1, T* f2, int64_t B) {              \
        k_affine_body<U_##SFX, T>(s, x, f1, f2, B);                                                                                         \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_wrap_##SFX(U_##SFX s, const T* x, T* out, int64_t B) {                      \
        k_wrap_body<U_##SFX, T>(s, x, out, B);                                                                                              \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_xdot_##SFX(U_##SFX s, const T* x, const T* u, T* xd, int64_t B) {           \
        k_xdot_body<U_##SFX, T>(s, x, u, xd, B);                                                                                            \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_simulate_i0_##SFX(U_##SFX s, Limits<T, U_##SFX::M> lim, const T* x,         \
                                                                                 const T* u, T* xn, int64_t B) {                           \
        k_simulate_body<0, 1, U_##SFX, T>(s, lim, x, u, xn, B);                                                                             \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_simulate_i1_##SFX(U_##SFX s, Limits<T, U_##SFX::M> lim, const T* x,         \
                                                                                 const T* u, T* xn, int64_t B) {                           \
        k_simulate_body<1, 1, U_##SFX, T>(s, lim, x, u, xn, B);                                                                             \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_initial_state_##SFX(U_##SFX s, X0P<U_##SFX, T> p, const T* u01, T* x0,      \
                                                                                   int64_t B) {                                            \
        k_initial_state_body<U_##SFX, T>(s, p, u01, x0, B);                                                                                 \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_running_cost_##SFX(U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk,          \
                                                                                  const T* x, const T* u, T* cost, int64_t B) {            \
        k_running_cost_body<U_##SFX, T>(s, tk, x, u, cost, B);                                                                              \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_termination_cost_##SFX(U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk,      \
                                                                                      const T* x, T* cost, int64_t B) {                    \
        k_termination_cost_body<U_##SFX, T>(s, tk, x, cost, B);                                                                             \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_control_from_grad_##SFX(U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk,     \
                                                                                       Limits<T, U_##SFX::M> lim, const T* x, const T* g,  \
                                                                                       T* u, int64_t B) {                                  \
        k_control_from_grad_body<U_##SFX, T>(s, tk, lim, x, g, u, B);                                                                       \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_controller_##SFX(U_##SFX s, CtrlP<T, U_##SFX::N, U_##SFX::M> c,             \
                                                                                Limits<T, U_##SFX::M> lim, const T* x, T* u, int64_t B) {  \
        k_controller_body<0, U_##SFX, T>(s, c, lim, x, u, B);                                                                               \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_hjb_residual_m0_##SFX(                                                      \
        U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk, Limits<T, U_##SFX::M> lim, const T* x, const T* g, const T* done, T* loss_i,        \
        T* dl_dg, unsigned char* ws, T* sums, int64_t B) {                                                                                  \
        k_hjb_residual_body<0, 1, U_##SFX, T>(s, tk, lim, x, g, done, loss_i, dl_dg, ws, sums, B);                                          \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_hjb_residual_m1_##SFX(                                                      \
        U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk, Limits<T, U_##SFX::M> lim, const T* x, const T* g, const T* done, T* loss_i,        \
        T* dl_dg, unsigned char* ws, T* sums, int64_t B) {                                                                                  \
        k_hjb_residual_body<1, 1, U_##SFX, T>(s, tk, lim, x, g, done, loss_i, dl_dg, ws, sums, B);                                          \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_vhjb_step_i0_##SFX(                                                         \
        U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk, Limits<T, U_##SFX::M> lim, int t, int T_max, const T* x, const T* g, T* xn,         \
        T* u_out, T* cost_t, T* done_t, int32_t* done_step, T* resid_t, int64_t B) {                                                        \
        k_vhjb_step_body<0, 1, U_##SFX, T>(s, tk, lim, t, T_max, x, g, xn, u_out, cost_t, done_t, done_step, resid_t, B);                   \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_vhjb_step_i1_##SFX(                                                         \
        U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk, Limits<T, U_##SFX::M> lim, int t, int T_max, const T* x, const T* g, T* xn,         \
        T* u_out, T* cost_t, T* done_t, int32_t* done_step, T* resid_t, int64_t B) {                                                        \
        k_vhjb_step_body<1, 1, U_##SFX, T>(s, tk, lim, t, T_max, x, g, xn, u_out, cost_t, done_t, done_step, resid_t, B);                   \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_rollout_feedback_i0_##SFX(                                                  \
        U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk, CtrlP<T, U_##SFX::N, U_##SFX::M> c, Limits<T, U_##SFX::M> lim, uint32_t flags,      \
        int has_task, int T_steps, const T* x0, T* traj, T* u_log, T* cost, int32_t* done_step, T* total_cost, T* x_final, int64_t B) {     \
        k_rollout_feedback_body<0, 0, U_##SFX, T>(s, tk, c, lim, flags, has_task, T_steps, x0, traj, u_log, cost, done_step, total_cost,    \
                                                  x_final, B);                                                                             \
    }                                                                                                                                       \
    extern "C" __global__ __launch_bounds__(kBlock) void hjbx_u_rollout_feedback_i1_##SFX(                                                  \
        U_##SFX s, TaskP<T, U_##SFX::N, U_##SFX::M> tk, CtrlP<T, U_##SFX::N, U_##SFX::M> c, Limits<T, U_##SFX::M> lim, uint32_t flags,      \
        int has_task, int T_steps, const T* x0, T* traj, T* u_log, T* cost, int32_t* done_step, T* total_cost, T* x_final, int64_t B) {     \
        k_rollout_feedback_body<1, 0, U_##SFX, T>(s, tk, c, lim, flags, has_task, T_steps, x0, traj, u_log, cost, done_step, total_cost,    \
                                                  x_final, B);                                                                             \
    }

HJBX_U_KERNELS(float, f32)
HJBX_U_KERNELS(double, f64)
