// hjbx_systems.hpp -- per-environment device math for the five control-affine systems (gfx950).
//
// Every system is a POD passed BY VALUE as a kernel argument (kernarg segment -> scalar loads into
// SGPRs, uniform across the wave), with compile-time state/control dimensions so a state vector
// lives in VGPRs and all loops unroll.  One lane = one environment.
//
// The manipulator systems use algebraically reduced closed forms (no 2x2 inverse); the CPU oracle
// keeps the reference's M/C/G + inverse structure, so agreement between the two is a real check.
// Reference statements restated here are cited per function (paths under the reference repo).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace hjbx {

#define HJBX_DEV __device__ __forceinline__

template <typename T> struct Const;
template <> struct Const<float> {
    static constexpr float pi = 3.14159274101257324219f;      // float32(np.pi)
    static constexpr float two_pi = 6.28318548202514648438f;  // float32(2*np.pi)
};
template <> struct Const<double> {
    static constexpr double pi = 3.14159265358979311600;
    static constexpr double two_pi = 6.28318530717958623200;
};

HJBX_DEV void sincos_t(float a, float* s, float* c) { sincosf(a, s, c); }   // precise (ocml), not v_sin_f32
HJBX_DEV void sincos_t(double a, double* s, double* c) { sincos(a, s, c); }
HJBX_DEV float tan_t(float a) { return tanf(a); }
HJBX_DEV double tan_t(double a) { return tan(a); }
HJBX_DEV float fmod_t(float a, float b) { return fmodf(a, b); }
HJBX_DEV double fmod_t(double a, double b) { return fmod(a, b); }
HJBX_DEV float sqrt_t(float a) { return sqrtf(a); }
HJBX_DEV double sqrt_t(double a) { return sqrt(a); }
HJBX_DEV float abs_t(float a) { return fabsf(a); }
HJBX_DEV double abs_t(double a) { return fabs(a); }

// np.remainder(th + pi, 2 pi) - pi  (cartpole.py:61-63, quadrotors.py:67-69,167-169, acrobot.py:78-79).
// NumPy's remainder is fmod followed by a shift into [0, b); fmod is exact, and for |a| < 2b it is
// a, a-b (exact by Sterbenz) or a (then +b), so the three fast branches are bit-identical to it.
template <typename T> HJBX_DEV T wrap_angle(T th) {
    constexpr T pi = Const<T>::pi, b = Const<T>::two_pi;
    const T a = th + pi;
    T r;
    if (a >= T(0) && a < b) r = a;
    else if (a >= b && a < T(2) * b) r = a - b;
    else if (a < T(0) && a >= -b) r = a + b;
    else {
        r = fmod_t(a, b);
        if (r != T(0)) { if (r < T(0)) r += b; } else r = T(0);
    }
    return r - pi;
}

template <typename T> HJBX_DEV T clamp_t(T v, T lo, T hi) {  // np.clip: min(max(v, lo), hi)
    v = v < lo ? lo : v;
    v = v > hi ? hi : v;
    return v;
}

// ---- Linear: dynamics/linear.py:7-22 --------------------------------------------------------------
template <typename T, int N_, int M_> struct Linear {
    static constexpr int N = N_, M = M_;
    static constexpr bool kHasZoh = true;
    T A[N * N], Bm[N * M];
    T Ad[N * N], Bd[N * M];  // exact zero-order-hold discretisation over dt (HJBX_ZOH); zero when the handle has none
    HJBX_DEV void wrap(T*) const {}
    // x' = Ad x + Bd u  (scipy.signal.cont2discrete, examples/double_integrator_optimal_time.ipynb cell 4)
    HJBX_DEV void zoh_step(const T* x, const T* u, T* xn) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += Ad[i * N + j] * x[j];
            T bu = T(0);
#pragma unroll
            for (int j = 0; j < M; ++j) bu += Bd[i * M + j] * u[j];
            xn[i] = acc + bu;
        }
    }
    HJBX_DEV void affine(const T* x, T* f1, T* f2) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += A[i * N + j] * x[j];
            f1[i] = acc;
#pragma unroll
            for (int j = 0; j < M; ++j) f2[i * M + j] = Bm[i * M + j];
        }
    }
    HJBX_DEV void xdot(const T* x, const T* u, T* xd) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            T acc = T(0);
#pragma unroll
            for (int j = 0; j < N; ++j) acc += A[i * N + j] * x[j];
            T bu = T(0);
#pragma unroll
            for (int j = 0; j < M; ++j) bu += Bm[i * M + j] * u[j];
            xd[i] = acc + bu;
        }
    }
};

// ---- Cartpole: dynamics/cartpole.py:19-64 through dynamics_basic.py:64-94 -------------------------
// D = mc + mp s^2;  f1 = [xd, thd, (mp l thd^2 s + mp g s c)/D, -(mp l thd^2 s c + (mc+mp) g s)/(l D)]
// f2 = [0, 0, 1/D, -c/(l D)]      (theta = pi is upright)
template <typename T> struct Cartpole {
    static constexpr bool kHasZoh = false;
    static constexpr int N = 4, M = 1;
    T mc, mp, l, g;
    T il = T(1) / l;  // float kernels multiply by this (an IEEE division is ~10 VALU instructions); double keeps the reference's "/ l"
    HJBX_DEV T over_l(T v) const {
        if constexpr (sizeof(T) == 4) return v * il;
        else return v / l;
    }
    HJBX_DEV void wrap(T* x) const { x[1] = wrap_angle(x[1]); }
    HJBX_DEV void affine(const T* x, T* f1, T* f2) const {
        T s, c;
        sincos_t(x[1], &s, &c);
        const T invD = T(1) / (mc + mp * s * s);
        const T w = mp * l * x[3] * x[3] * s;
        f1[0] = x[2];
        f1[1] = x[3];
        f1[2] = (w + mp * g * s * c) * invD;
        f1[3] = over_l(-(w * c + (mc + mp) * g * s) * invD);
        f2[0] = T(0);
        f2[1] = T(0);
        f2[2] = invD;
        f2[3] = over_l(-c * invD);
    }
    HJBX_DEV void xdot(const T* x, const T* u, T* xd) const {
        T f1[4], f2[4];
        affine(x, f1, f2);
#pragma unroll
        for (int i = 0; i < 4; ++i) xd[i] = f1[i] + f2[i] * u[0];
    }
};

// ---- Acrobot: dynamics/acrobot.py:39-81 (constructor is stale upstream; math only) -----------------
template <typename T> struct Acrobot {
    static constexpr bool kHasZoh = false;
    static constexpr int N = 4, M = 1;
    T m1, m2, l1, l2, I1, I2, g;
    HJBX_DEV void wrap(T* x) const { x[0] = wrap_angle(x[0]); x[1] = wrap_angle(x[1]); }
    // M = [[a, b],[b, d]], h = C qd + G
    HJBX_DEV void mch(const T* x, T& a, T& b, T& d, T& h0, T& h1) const {
        T s1, c1, s2, c2, s12, c12;
        sincos_t(x[0], &s1, &c1);
        sincos_t(x[1], &s2, &c2);
        sincos_t(x[0] + x[1], &s12, &c12);
        const T k = m2 * l1 * l2 / T(2);
        a = I1 + I2 + m2 * l1 * l1 + T(2) * k * c2;
        b = I2 + k * c2;
        d = I2;
        const T G0 = (m1 * l1 / T(2) + m2 * l1) * g * s1 + m2 * g * l2 / T(2) * s12;
        const T G1 = m2 * g * l2 / T(2) * s12;
        h0 = (-T(2) * k * s2 * x[3]) * x[2] + (-k * s2 * x[3]) * x[3] + G0;
        h1 = (k * s2 * x[2]) * x[2] + G1;
    }
    HJBX_DEV T energy(const T* x) const {  // acrobot.py:61-70
        T s1, c1, s2, c2, s12, c12;
        sincos_t(x[0], &s1, &c1);
        sincos_t(x[1], &s2, &c2);
        sincos_t(x[0] + x[1], &s12, &c12);
        const T k = m2 * l1 * l2 / T(2);
        const T T1 = T(0.5) * I1 * x[2] * x[2];
        const T T2 = T(0.5) * (m2 * l1 * l1 + I2 + T(2) * k * c2) * x[2] * x[2] + T(0.5) * I2 * x[3] * x[3] +
                     (I2 + k * c2) * x[2] * x[3];
        const T U = -m1 * g * l1 / T(2) * c1 - m2 * g * (l1 * c1 + l2 / T(2) * c12);
        return T1 + T2 + U;
    }
    HJBX_DEV void affine(const T* x, T* f1, T* f2) const {
        T a, b, d, h0, h1;
        mch(x, a, b, d, h0, h1);
        const T idet = T(1) / (a * d - b * b);
        f1[0] = x[2];
        f1[1] = x[3];
        f1[2] = -(d * h0 - b * h1) * idet;
        f1[3] = -(-b * h0 + a * h1) * idet;
        f2[0] = T(0);
        f2[1] = T(0);
        f2[2] = -b * idet;  // Minv @ [0,1]
        f2[3] = a * idet;
    }
    HJBX_DEV void xdot(const T* x, const T* u, T* xd) const {
        T f1[4], f2[4];
        affine(x, f1, f2);
#pragma unroll
        for (int i = 0; i < 4; ++i) xd[i] = f1[i] + f2[i] * u[0];
    }
};

// ---- Quadrotors2D: dynamics/quadrotors.py:17-70 ---------------------------------------------------
template <typename T> struct Quad2D {
    static constexpr bool kHasZoh = false;
    static constexpr int N = 6, M = 2;
    T m, r, I, g;
    T im = T(1) / m, rI = r / I;  // float kernels multiply by these; double keeps the reference's divisions (quadrotors.py:35-44)
    HJBX_DEV T over_m(T v) const {
        if constexpr (sizeof(T) == 4) return v * im;
        else return v / m;
    }
    HJBX_DEV T r_over_I() const {
        if constexpr (sizeof(T) == 4) return rI;
        else return r / I;
    }
    HJBX_DEV void wrap(T* x) const { x[2] = wrap_angle(x[2]); }
    HJBX_DEV void affine(const T* x, T* f1, T* f2) const {
        T s, c;
        sincos_t(x[2], &s, &c);
        f1[0] = x[3]; f1[1] = x[4]; f1[2] = x[5]; f1[3] = T(0); f1[4] = -g; f1[5] = T(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) f2[i] = T(0);
        f2[6] = over_m(-s); f2[7] = over_m(-s);
        f2[8] = over_m(c);  f2[9] = over_m(c);
        f2[10] = r_over_I(); f2[11] = -r_over_I();
    }
    HJBX_DEV void xdot(const T* x, const T* u, T* xd) const {
        T s, c;
        sincos_t(x[2], &s, &c);
        // f1 + f2 @ u with f2's rows written out (same operation order as the generic affine form)
        xd[0] = x[3]; xd[1] = x[4]; xd[2] = x[5];
        xd[3] = T(0) + (over_m(-s) * u[0] + over_m(-s) * u[1]);
        xd[4] = -g + (over_m(c) * u[0] + over_m(c) * u[1]);
        xd[5] = T(0) + (r_over_I() * u[0] + (-r_over_I()) * u[1]);
    }
};

// ---- NearHoverQuadcopter: dynamics/quadrotors.py:118-170 ------------------------------------------
template <typename T> struct NearHover {
    static constexpr bool kHasZoh = false;
    static constexpr int N = 10, M = 3;
    T g, m, kT, n0;
    T kTm_f = kT / m;  // float kernels use this; double keeps the reference's division in place (quadrotors.py:143)
    HJBX_DEV T kT_over_m() const {
        if constexpr (sizeof(T) == 4) return kTm_f;
        else return kT / m;
    }
    HJBX_DEV void wrap(T* x) const { x[3] = wrap_angle(x[3]); x[4] = wrap_angle(x[4]); }
    HJBX_DEV void affine(const T* x, T* f1, T* f2) const {
#pragma unroll
        for (int i = 0; i < 5; ++i) f1[i] = x[5 + i];
        f1[5] = g * tan_t(x[3]); f1[6] = g * tan_t(x[4]); f1[7] = -g; f1[8] = T(0); f1[9] = T(0);
#pragma unroll
        for (int i = 0; i < 30; ++i) f2[i] = T(0);
        f2[21] = kT_over_m(); f2[25] = n0; f2[29] = n0;
    }
    HJBX_DEV void xdot(const T* x, const T* u, T* xd) const {
#pragma unroll
        for (int i = 0; i < 5; ++i) xd[i] = x[5 + i];
        xd[5] = g * tan_t(x[3]);
        xd[6] = g * tan_t(x[4]);
        xd[7] = -g + kT_over_m() * u[0];
        xd[8] = n0 * u[1];
        xd[9] = n0 * u[2];
    }
};

// ---- shared per-environment pieces ---------------------------------------------------------------
template <typename T, int M> struct Limits { T umin[M], umax[M], dt; };

template <typename T, int N, int M> struct TaskP {
    T Q[N * N], R[M * M], Rinv[M * M], P[N * N], xf[N], uf[M], omin[N], omax[N], eps;
    T target_r2;  // HJBX_LAW_BANGBANG: squared radius of the target ball
    int law;      // hjbx_control_law
};

template <typename T, int N, int M> struct CtrlP {
    int wrap_error;
    T K[M * N], xf[N], uf[M], P[N * N], Kes[3], eps_energy, eps_state, eps_region;
};

template <typename T, int M> HJBX_DEV void clip_u(const Limits<T, M>& lim, const T* u, T* uc) {
#pragma unroll
    for (int j = 0; j < M; ++j) uc[j] = clamp_t(u[j], lim.umin[j], lim.umax[j]);
}

// Dynamics.simulate body after the clip (dynamics_basic.py:120); RK4 is this library's extension.
template <int INTEG, typename S, typename T> HJBX_DEV void integrate(const S& sys, T dt, const T* x, const T* u, T* xn) {
    constexpr int N = S::N;
    if constexpr (INTEG == 2) {  // HJBX_ZOH: linear systems only (the host refuses it elsewhere)
        static_assert(S::kHasZoh, "zero-order-hold stepping exists for LinearDynamics only");
        sys.zoh_step(x, u, xn);
        sys.wrap(xn);
        return;
    }
    T k1[N];
    sys.xdot(x, u, k1);
    if constexpr (INTEG == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) xn[i] = x[i] + k1[i] * dt;
    } else {
        T k2[N], k3[N], k4[N], xt[N];
#pragma unroll
        for (int i = 0; i < N; ++i) xt[i] = x[i] + (dt / T(2)) * k1[i];
        sys.xdot(xt, u, k2);
#pragma unroll
        for (int i = 0; i < N; ++i) xt[i] = x[i] + (dt / T(2)) * k2[i];
        sys.xdot(xt, u, k3);
#pragma unroll
        for (int i = 0; i < N; ++i) xt[i] = x[i] + dt * k3[i];
        sys.xdot(xt, u, k4);
#pragma unroll
        for (int i = 0; i < N; ++i) xn[i] = x[i] + (dt / T(6)) * (k1[i] + T(2) * k2[i] + T(2) * k3[i] + k4[i]);
    }
    sys.wrap(xn);
}

// e = states_wrap(x - xf)  (vhjb.py:163, 168, 176)
template <typename S, typename T> HJBX_DEV void error_coords(const S& sys, const T* xf, const T* x, T* e) {
#pragma unroll
    for (int i = 0; i < S::N; ++i) e[i] = x[i] - xf[i];
    sys.wrap(e);
}

template <int N, typename T> HJBX_DEV T quad_form(const T* A, const T* v) {
    T acc = T(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        T row = T(0);
#pragma unroll
        for (int j = 0; j < N; ++j) row += A[i * N + j] * v[j];
        acc += v[i] * row;
    }
    return acc;
}

// vhjb.py:220: u_raw = -Rinv f2' g / 2 + uf ; u = clip(u_raw).  HJBX_LAW_BANGBANG (time-optimal notebook, cells 9 and 11:
// u = -sign(gradV @ B)): u_j = umax_j / umin_j / 0 by the sign of (f2' g)_j; u_raw = u (nothing is "open": du/dg = 0).
template <typename S, typename T>
HJBX_DEV void control_from_grad(const TaskP<T, S::N, S::M>& tk, const Limits<T, S::M>& lim, const T* f2, const T* g,
                                T* u_raw, T* u) {
    constexpr int N = S::N, M = S::M;
    T f2tg[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < N; ++i) acc += f2[i * M + j] * g[i];
        f2tg[j] = acc;
    }
    const bool bang = tk.law != 0;  // selected per element, not by an early return: keeps u / u_raw in registers (no scratch)
#pragma unroll
    for (int j = 0; j < M; ++j) {
        T acc = T(0);
#pragma unroll
        for (int k = 0; k < M; ++k) acc += tk.Rinv[j * M + k] * f2tg[k];
        const T quad = -acc / T(2) + tk.uf[j];
        const T bb = (f2tg[j] < T(0)) ? lim.umax[j] : ((f2tg[j] > T(0)) ? lim.umin[j] : T(0));
        u_raw[j] = bang ? bb : quad;
        u[j] = bang ? bb : clamp_t(quad, lim.umin[j], lim.umax[j]);
    }
}

// running cost given the error coordinates: e'Qe + (u-uf)'R(u-uf)   (vhjb.py:162-165); HJBX_LAW_BANGBANG: 1 outside the
// target ball, 0 inside (np.where(norms > metric, 1, 0), time-optimal notebook cell 7)
template <typename S, typename T> HJBX_DEV T norm2_e(const T* e) {
    T s = T(0);
#pragma unroll
    for (int i = 0; i < S::N; ++i) s += e[i] * e[i];
    return s;
}
template <typename S, typename T>
HJBX_DEV T running_cost_e(const TaskP<T, S::N, S::M>& tk, const T* e, const T* u) {
    if (tk.law != 0) return (norm2_e<S, T>(e) > tk.target_r2) ? T(1) : T(0);
    T du[S::M];
#pragma unroll
    for (int j = 0; j < S::M; ++j) du[j] = u[j] - tk.uf[j];
    return quad_form<S::N>(tk.Q, e) + quad_form<S::M>(tk.R, du);
}

template <typename S, typename T> HJBX_DEV bool out_of_box(const TaskP<T, S::N, S::M>& tk, const T* e) {
    // vhjb.py:176-177: any(e > obs_max) or any(e < obs_min), strict.  Written as negated non-strict compares so a
    // NaN error coordinate (a diverged environment, e.g. tan near pi/2 in the near-hover model) also terminates
    // instead of integrating NaNs for the rest of the horizon; for finite e the two forms are identical.
    bool out = false;
#pragma unroll
    for (int i = 0; i < S::N; ++i) out = out || !(e[i] <= tk.omax[i]) || !(e[i] >= tk.omin[i]);
    return out;
}

// One iteration of rollout_trajectory's loop (reference vhjb.py:175-191) for ONE environment, given dV/dx at its
// state: shared by the per-step kernel (hjbx_vhjb_step) and the fused whole-rollout kernel (hjbx_vhjb_rollout), so
// the two produce identical bits.  ds < 0 = live.  A live env outside the observation box (or t >= T_max) emits the
// terminal tuple (c = e'Pe, d = 1, ds = t) and holds; otherwise u from gradV, c = l dt, d = 0, xo = simulate(x, u),
// res = gradV.xdot/(l+eps) + 1 (vhjb.py:233, signed) when want_res.  Dead envs emit zeros and hold.
template <int INTEG, typename S, typename T>
HJBX_DEV void vhjb_step_env(const S& sys, const TaskP<T, S::N, S::M>& tk, const Limits<T, S::M>& lim, int t, int T_max, bool want_res,
                            const T* xs, const T* gs, int32_t& ds, T* xo, T* u, T& c, T& d, T& res) {
    constexpr int N = S::N, M = S::M;
#pragma unroll
    for (int k = 0; k < N; ++k) xo[k] = xs[k];
#pragma unroll
    for (int j = 0; j < M; ++j) u[j] = T(0);
    c = T(0); d = T(0); res = T(0);
    if (ds < 0) {
        T e[N];
        error_coords(sys, tk.xf, xs, e);
        const bool reached = tk.law != 0 && norm2_e<S, T>(e) <= tk.target_r2;  // time-optimal notebook cell 9: x'x <= metric
        if (t >= T_max || reached || out_of_box<S, T>(tk, e)) {  // vhjb.py:176-181 and 188-191
            c = quad_form<N>(tk.P, e);
            d = T(1);
            ds = t;
        } else {  // vhjb.py:183-186
            T f1[N], f2[N * M], ur[M];
            sys.affine(xs, f1, f2);
            control_from_grad<S, T>(tk, lim, f2, gs, ur, u);
            const T l = running_cost_e<S, T>(tk, e, u);
            c = l * lim.dt;
            if (want_res) {  // vhjb.py:231-233: gradV . (f1 + f2 u) / (l + eps) + 1
                T vdot = T(0);
#pragma unroll
                for (int r = 0; r < N; ++r) {
                    T a = T(0);
#pragma unroll
                    for (int j = 0; j < M; ++j) a += f2[r * M + j] * u[j];
                    vdot += gs[r] * (f1[r] + a);
                }
                res = vdot / (l + tk.eps) + T(1);
            }
            integrate<INTEG>(sys, lim.dt, xs, u, xo);
        }
    }
}

// hjb_loss body for ONE sample (reference vhjb.py:227-241) with the analytic derivative of the loss w.r.t. gradV: shared
// by hjbx_hjb_residual and the fused value-loss-gradient kernel (hjbx_train.hip), so the two produce identical bits.
//   u_raw = -Rinv f2' g / 2 + uf, u = clip(u_raw); xdot = f1 + f2 u; l = e'Qe + (u-uf)'R(u-uf)
//   r = g.xdot / (l + eps) + 1 (MODE 0, vhjb.py:233) | g.xdot + l (MODE 1: cartpole notebook cell 11); loss = |r| (1 - done)
//   du/dg = -1/2 D Rinv f2' (D = 1 where the clip is inactive); dV./dg = xdot + (du/dg)' f2' g; dl/dg = (du/dg)' (R+R')(u-uf)
template <int MODE, typename S, typename T>
HJBX_DEV void hjb_residual_env(const S& sys, const TaskP<T, S::N, S::M>& tk, const Limits<T, S::M>& lim, const T* xs, const T* gs, T dn,
                               bool want_grad, T& li, T* out) {
    constexpr int N = S::N, M = S::M;
    T f1[N], f2[N * M], ur[M], u[M], e[N], xd[N];
    sys.affine(xs, f1, f2);
    control_from_grad<S, T>(tk, lim, f2, gs, ur, u);
    T vdot = T(0);
#pragma unroll
    for (int r = 0; r < N; ++r) {
        T a = T(0);
#pragma unroll
        for (int j = 0; j < M; ++j) a += f2[r * M + j] * u[j];
        xd[r] = f1[r] + a;
        vdot += gs[r] * xd[r];
    }
    error_coords(sys, tk.xf, xs, e);
    const T l = running_cost_e<S, T>(tk, e, u);
    const T den = l + tk.eps;
    // The residual itself divides like the reference in both precisions (vhjb.py:233: one rounding; with a reciprocal the float32 loss
    // was 2.3x further from the float64 oracle than the oracle's own float build, round 3).  float: one reciprocal per sample for the
    // GRADIENT below (it would otherwise need 2 divisions per state dimension); double keeps every division in place.
    const T iden = T(1) / den;
    T r;
    if constexpr (MODE != 0) r = vdot + l;
    else r = vdot / den + T(1);
    const T w = T(1) - dn;
    li = abs_t(r) * w;
    if (!want_grad) return;
    T f2tg[M], rdu[M];
    bool open[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        T a = T(0);
#pragma unroll
        for (int k = 0; k < N; ++k) a += f2[k * M + j] * gs[k];
        f2tg[j] = a;
        T b = T(0);
#pragma unroll
        for (int k = 0; k < M; ++k) b += (tk.R[j * M + k] + tk.R[k * M + j]) * (u[k] - tk.uf[k]);
        rdu[j] = b;
        open[j] = tk.law == 0 && (ur[j] > lim.umin[j]) && (ur[j] < lim.umax[j]);
    }
    const T sg = (r > T(0)) ? T(1) : ((r < T(0)) ? T(-1) : T(0));
#pragma unroll
    for (int k = 0; k < N; ++k) {
        T dv = xd[k], dl = T(0);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            T a = T(0);
#pragma unroll
            for (int q = 0; q < M; ++q) a += tk.Rinv[j * M + q] * f2[k * M + q];
            const T dudg = open[j] ? -a / T(2) : T(0);
            dv += dudg * f2tg[j];
            dl += dudg * rdu[j];
        }
        T dr;
        if constexpr (MODE != 0) dr = dv + dl;
        else if constexpr (sizeof(T) == 4) dr = dv * iden - (vdot * iden * iden) * dl;
        else dr = dv / den - vdot * dl / (den * den);
        out[k] = sg * w * dr;
    }
}

// termination_loss body for ONE sample (vhjb.py:243-253): loss = |V / (cost + eps) - 1| done; dloss/dV = sign(.) done / (cost + eps)
template <typename T> HJBX_DEV void termination_residual_env(T eps, T V, T cost, T dn, T& li, T& dl_dV) {
    const T den = cost + eps;
    const T r = V / den - T(1);
    li = abs_t(r) * dn;
    dl_dV = ((r > T(0)) ? T(1) : ((r < T(0)) ? T(-1) : T(0))) * dn / den;
}

// ---- closed-form controllers -----------------------------------------------------------
// CK = 0 linear feedback (lqr.py:25-26; quadrotors_model_based_controller.py:36-38, 73-75)
// CK = 1 cartpole energy shaping (cartpole_energy_shaping.py:65-110), CK = 2 acrobot (acrobot_energy_shaping.py:74-121)
template <int CK, typename S, typename T>
HJBX_DEV void controller_eval(const S& sys, const CtrlP<T, S::N, S::M>& c, const Limits<T, S::M>& lim, const T* x, T* u) {
    constexpr int N = S::N, M = S::M;
    T ur[M];
    if constexpr (CK == 0) {
        T e[N];
#pragma unroll
        for (int i = 0; i < N; ++i) e[i] = x[i] - c.xf[i];
        if (c.wrap_error) sys.wrap(e);
#pragma unroll
        for (int j = 0; j < M; ++j) {
            T acc = T(0);
#pragma unroll
            for (int i = 0; i < N; ++i) acc += c.K[j * N + i] * e[i];
            ur[j] = -acc + c.uf[j];
        }
    } else if constexpr (CK == 1) {
        T dx[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) dx[i] = x[i] - c.xf[i];
        sys.wrap(dx);
        T sth, cth, sf, cf;
        sincos_t(x[1], &sth, &cth);
        sincos_t(c.xf[1], &sf, &cf);
        const T de = (T(0.5) * x[3] * x[3] - cth) - (T(0.5) * c.xf[3] * c.xf[3] - cf);
        const T nrm = sqrt_t(dx[1] * dx[1] + dx[3] * dx[3]);
        if (abs_t(de) < c.eps_energy && nrm < c.eps_state) {
            T acc = T(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += c.K[i] * dx[i];
            ur[0] = -acc;
        } else {
            const T u_bar = de * x[3] * cth;
            const T ddq1 = c.Kes[0] * (-x[0]) + c.Kes[1] * (-x[2]) + c.Kes[2] * u_bar;
            const T ddq2 = -cth / sys.l * ddq1 - sys.g * sth / sys.l;
            ur[0] = (sys.mc + sys.mp) * ddq1 + sys.mp * sys.l * cth * ddq2 - sys.mp * sys.l * sth * x[3] * x[3];
        }
    } else if constexpr (CK == 3) {
        // double integrator, analytic minimum-time law (examples/double_integrator_optimal_time.ipynb cell 18):
        // 0 inside the target ball, else +a above / -a below the switching curve p = -v|v|/(2a), a = umax
        const T p0 = x[0] - c.xf[0], v0 = x[1] - c.xf[1];
        const T a = lim.umax[0];
        if (p0 * p0 + v0 * v0 <= c.eps_region) ur[0] = T(0);
        else if ((v0 < T(0) && p0 <= T(0.5) * v0 * v0 / a) || (v0 >= T(0) && p0 < -T(0.5) * v0 * v0 / a)) ur[0] = a;
        else ur[0] = -a;
    } else {
        T dx[4];
        dx[0] = wrap_angle(x[0] - c.xf[0]);
        dx[1] = wrap_angle(x[1] - c.xf[1]);
        dx[2] = x[2] - c.xf[2];
        dx[3] = x[3] - c.xf[3];
        if (quad_form<4>(c.P, dx) < c.eps_region) {
            T acc = T(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc += c.K[i] * dx[i];
            ur[0] = -acc;
        } else {
            T a, b, d, h0, h1;
            sys.mch(x, a, b, d, h0, h1);
            const T ubar = (sys.energy(x) - sys.energy(c.xf)) * x[2];
            const T ddq2 = c.Kes[0] * (-wrap_angle(x[1])) + c.Kes[1] * (-x[3]) + c.Kes[2] * ubar;
            ur[0] = (d - b * b / a) * ddq2 + h1 - b / a * h0;
        }
    }
    clip_u<T, M>(lim, ur, u);
}

}  // namespace hjbx
