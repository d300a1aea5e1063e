// hjbx_host.hpp -- host-side conversion of the ABI descriptors (doubles) into the by-value kernel argument PODs and
// the handle -> concrete device system dispatch.  Shared by hjbx_kernels.hip and hjbx_mlp.hip; not part of the ABI.
#pragma once
#include <atomic>
#include <cstring>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"

using namespace hjbx;

// compile-time facts about a device system type, used to instantiate system-specific kernels only where they exist
template <typename S> struct is_linear { static constexpr bool value = S::kHasZoh; };                            // HJBX_ZOH stepping
template <typename S> struct is_di { static constexpr bool value = S::kHasZoh && S::N == 2 && S::M == 1; };  // double integrator

template <typename T, int M> inline Limits<T, M> make_limits(const hjbx_system* s) {
    Limits<T, M> l;
    for (int j = 0; j < M; ++j) { l.umin[j] = (T)s->umin[j]; l.umax[j] = (T)s->umax[j]; }
    l.dt = (T)s->dt;
    return l;
}

template <typename T, int N, int M> inline TaskP<T, N, M> make_task(const hjbx_task* t) {
    TaskP<T, N, M> k;
    memset(&k, 0, sizeof(k));
    if (!t) return k;
    for (int i = 0; i < N * N; ++i) { k.Q[i] = (T)t->Q[i]; k.P[i] = (T)t->P[i]; }
    for (int i = 0; i < M * M; ++i) { k.R[i] = (T)t->R[i]; k.Rinv[i] = (T)t->Rinv[i]; }
    for (int i = 0; i < N; ++i) { k.xf[i] = (T)t->xf[i]; k.omin[i] = (T)t->obs_min[i]; k.omax[i] = (T)t->obs_max[i]; }
    for (int j = 0; j < M; ++j) k.uf[j] = (T)t->uf[j];
    k.eps = (T)t->eps;
    k.target_r2 = (T)t->target_r2;
    k.law = t->law;
    return k;
}

inline int check_task(const hjbx_task* t) {
    if (!t) return hjbx_set_error(HJBX_EINVAL, "task is NULL");
    if (t->law != HJBX_LAW_QUADRATIC && t->law != HJBX_LAW_BANGBANG) return hjbx_set_error(HJBX_EINVAL, "unknown control law %d", t->law);
    if (t->law == HJBX_LAW_BANGBANG && !(t->target_r2 >= 0.0))
        return hjbx_set_error(HJBX_EINVAL, "bang-bang law: target_r2 must be >= 0, got %g", t->target_r2);
    return HJBX_OK;
}

template <typename T, int N, int M> inline CtrlP<T, N, M> make_ctrl(const hjbx_controller* c) {
    CtrlP<T, N, M> k;
    memset(&k, 0, sizeof(k));
    k.wrap_error = c->wrap_error;
    for (int i = 0; i < M * N; ++i) k.K[i] = (T)c->K[i];
    for (int i = 0; i < N; ++i) k.xf[i] = (T)c->xf[i];
    for (int j = 0; j < M; ++j) k.uf[j] = (T)c->uf[j];
    for (int i = 0; i < N * N; ++i) k.P[i] = (T)c->P[i];
    for (int i = 0; i < 3; ++i) k.Kes[i] = (T)c->Kes[i];
    k.eps_energy = (T)c->eps_energy;
    k.eps_state = (T)c->eps_state;
    k.eps_region = (T)c->eps_region;
    return k;
}

template <typename T, int N, int M> inline Linear<T, N, M> make_linear(const hjbx_system* s) {
    Linear<T, N, M> l;
    for (int i = 0; i < N * N; ++i) l.A[i] = (T)s->p[i];
    for (int i = 0; i < N * M; ++i) l.Bm[i] = (T)s->p[N * N + i];
    const bool zoh = s->n_params == 2 * (N * N + N * M);
    for (int i = 0; i < N * N; ++i) l.Ad[i] = zoh ? (T)s->p[N * N + N * M + i] : T(0);
    for (int i = 0; i < N * M; ++i) l.Bd[i] = zoh ? (T)s->p[2 * N * N + N * M + i] : T(0);
    return l;
}

// Calls f(system_pod) with the concrete device system type for this handle; false if unsupported.
template <typename T, typename F> inline bool with_system(const hjbx_system* s, F&& f) {
    switch (s->kind) {
    case HJBX_SYS_LINEAR:
        if (s->n == 2 && s->m == 1) { f(make_linear<T, 2, 1>(s)); return true; }
        if (s->n == 2 && s->m == 2) { f(make_linear<T, 2, 2>(s)); return true; }
        if (s->n == 4 && s->m == 1) { f(make_linear<T, 4, 1>(s)); return true; }
        if (s->n == 4 && s->m == 2) { f(make_linear<T, 4, 2>(s)); return true; }
        if (s->n == 6 && s->m == 2) { f(make_linear<T, 6, 2>(s)); return true; }
        return false;
    case HJBX_SYS_CARTPOLE: { Cartpole<T> c{(T)s->p[0], (T)s->p[1], (T)s->p[2], (T)s->p[3]}; f(c); return true; }
    case HJBX_SYS_ACROBOT: {
        Acrobot<T> a{(T)s->p[0], (T)s->p[1], (T)s->p[2], (T)s->p[3], (T)s->p[4], (T)s->p[5], (T)s->p[6]};
        f(a); return true;
    }
    case HJBX_SYS_QUAD2D: { Quad2D<T> q{(T)s->p[0], (T)s->p[1], (T)s->p[2], (T)s->p[3]}; f(q); return true; }
    case HJBX_SYS_NEARHOVER: { NearHover<T> q{(T)s->p[0], (T)s->p[1], (T)s->p[2], (T)s->p[3]}; f(q); return true; }
    }
    return false;
}


// Compute units of the CURRENT device (sizes the persistent grids and their workspaces); cached per device ordinal, not per process:
// a process that moves to a second GPU must not size its launches with the first one's count.  0 = no device.
static constexpr int kMaxDevices = 64;
inline int hjbx_current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices ? dev : -1;
}
inline int hjbx_device_cus() {
    static std::atomic<int> n_cu[kMaxDevices];
    const int dev = hjbx_current_device();
    if (dev < 0) return 0;
    int n = n_cu[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        n = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        n_cu[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

// integrator argument check shared by every stepping entry point: HJBX_ZOH needs a LINEAR handle created with Ad, Bd
inline int check_integrator(const hjbx_system* s, int integ, const char* who) {
    if (integ == HJBX_EULER || integ == HJBX_RK4) return HJBX_OK;
    if (integ == HJBX_ZOH) {
        if (s->kind != HJBX_SYS_LINEAR) return hjbx_set_error(HJBX_EUNSUPPORTED, "%s: HJBX_ZOH exists for LINEAR systems only", who);
        if (s->n_params != 2 * (s->n * s->n + s->n * s->m))
            return hjbx_set_error(HJBX_EINVAL, "%s: HJBX_ZOH needs a system created with Ad, Bd (2(n*n+n*m) parameters)", who);
        return HJBX_OK;
    }
    return hjbx_set_error(HJBX_EINVAL, "%s: unknown integrator %d", who, integ);
}
