/*
 * oracle_impl.h -- CPU restatement of the reference's hot-path arithmetic, generic over REAL.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under q_learning_with_hjb_amd/ may include, link, load or call
 * this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, as the checker.
 *
 * Included twice by hjbx_oracle.c (REAL=double, SFX=f64 and REAL=float, SFX=f32).  The f64 build
 * mirrors the reference's CPU rollout precision; it is pinned against golden vectors
 * produced by the reference's own NumPy branch (tests/golden, tools/gen_golden.py).  Every function
 * cites the reference statements (path:line under /root/reference) it restates.  The manipulator
 * systems deliberately keep the reference's M/C/G/B + inverse structure (the HIP kernels use
 * algebraically reduced closed forms, so the two are independent derivations).
 *
 * Parts of controller/vhjb.py need JAX/Flax/optax and could not be executed here; the functions
 * restating them (value_grad, hjb_residual, termination_residual, vhjb_step, vhjb_rollout) are
 * marked PARITY UNPINNED and are checked by known-answer identities in tests/ instead.
 */

#define FN_(a, b) a##_##b
#define FN__(a, b) FN_(a, b)
#define FN(name) FN__(name, SFX)

/* np.remainder(a, b) for b > 0 (numpy npy_divmod semantics: fmod, then shift into [0,b)) */
static inline REAL FN(np_remainder)(REAL a, REAL b) {
    REAL mod = (REAL)fmod((double)a, (double)b); /* fmod is exact in either precision */
    if (mod != 0) {
        if (mod < 0) mod += b;
    } else {
        mod = 0;
    }
    return mod;
}

/* cartpole.py:61-63 / quadrotors.py:67-69,167-169 / acrobot.py:78-79:
 * remainder(theta + pi, 2 pi) - pi  (pi, 2*pi rounded to REAL like np.pi / jnp.pi) */
static inline REAL FN(wrap_angle)(REAL th) {
    const REAL pi = (REAL)M_PI, two_pi = (REAL)(2.0 * M_PI);
    return FN(np_remainder)(th + pi, two_pi) - pi;
}

/* Dynamics.states_wrap, in place */
static void FN(wrap1)(const orc_system* s, REAL* x) {
    switch (s->kind) {
    case HJBX_SYS_LINEAR: break;                                   /* linear.py:17-18 identity */
    case HJBX_SYS_CARTPOLE: x[1] = FN(wrap_angle)(x[1]); break;    /* cartpole.py:52-64 */
    case HJBX_SYS_ACROBOT: x[0] = FN(wrap_angle)(x[0]); x[1] = FN(wrap_angle)(x[1]); break; /* acrobot.py:72-81 */
    case HJBX_SYS_QUAD2D: x[2] = FN(wrap_angle)(x[2]); break;      /* quadrotors.py:48-70 */
    case HJBX_SYS_NEARHOVER: x[3] = FN(wrap_angle)(x[3]); x[4] = FN(wrap_angle)(x[4]); break; /* :151-170 */
    }
}

/* get_M / get_C / get_G / get_B of the two manipulator systems; 2x2 row-major */
static void FN(manip)(const orc_system* s, const REAL* x, REAL* M, REAL* C, REAL* G, REAL* Bv) {
    if (s->kind == HJBX_SYS_CARTPOLE) { /* cartpole.py:19-50 */
        const REAL mc = (REAL)s->p[0], mp = (REAL)s->p[1], l = (REAL)s->p[2], g = (REAL)s->p[3];
        const REAL c = (REAL)cos((double)x[1]), sn = (REAL)sin((double)x[1]);
        M[0] = mc + mp; M[1] = mp * l * c; M[2] = mp * l * c; M[3] = mp * l * l;
        C[0] = 0; C[1] = -mp * l * x[3] * sn; C[2] = 0; C[3] = 0;
        G[0] = 0; G[1] = mp * g * l * sn;
        Bv[0] = 1; Bv[1] = 0;
    } else { /* acrobot.py:39-59 */
        const REAL m1 = (REAL)s->p[0], m2 = (REAL)s->p[1], l1 = (REAL)s->p[2], l2 = (REAL)s->p[3];
        const REAL I1 = (REAL)s->p[4], I2 = (REAL)s->p[5], g = (REAL)s->p[6];
        const REAL c2 = (REAL)cos((double)x[1]), s2 = (REAL)sin((double)x[1]);
        const REAL s1 = (REAL)sin((double)x[0]), s12 = (REAL)sin((double)(x[0] + x[1]));
        M[0] = I1 + I2 + m2 * l1 * l1 + 2 * m2 * l1 * l2 / 2 * c2;
        M[1] = I2 + m2 * l1 * l2 / 2 * c2;
        M[2] = M[1];
        M[3] = I2;
        C[0] = -2 * m2 * l1 * l2 / 2 * s2 * x[3]; C[1] = -m2 * l1 * l2 / 2 * s2 * x[3];
        C[2] = m2 * l1 * l2 / 2 * s2 * x[2];      C[3] = 0;
        G[0] = (m1 * l1 / 2 + m2 * l1) * g * s1 + m2 * g * l2 / 2 * s12;
        G[1] = m2 * g * l2 / 2 * s12;
        Bv[0] = 0; Bv[1] = 1;
    }
}

/* Acrobot.energy, acrobot.py:61-70 */
static REAL FN(acrobot_energy)(const orc_system* s, const REAL* x) {
    const REAL m1 = (REAL)s->p[0], m2 = (REAL)s->p[1], l1 = (REAL)s->p[2], l2 = (REAL)s->p[3];
    const REAL I1 = (REAL)s->p[4], I2 = (REAL)s->p[5], g = (REAL)s->p[6];
    const REAL c1 = (REAL)cos((double)x[0]), c2 = (REAL)cos((double)x[1]);
    const REAL T1 = (REAL)0.5 * I1 * x[2] * x[2];
    const REAL T2 = (REAL)0.5 * (m2 * l1 * l1 + I2 + 2 * m2 * l1 * l2 / 2 * c2) * x[2] * x[2] +
                    (REAL)0.5 * I2 * x[3] * x[3] + (I2 + m2 * l1 * l2 / 2 * c2) * x[2] * x[3];
    const REAL U = -m1 * g * l1 / 2 * c1 - m2 * g * (l1 * c1 + l2 / 2 * (REAL)cos((double)(x[0] + x[1])));
    return T1 + T2 + U;
}

/* Dynamics.get_control_affine_matrix for one state: f1 (n), f2 (n*m row-major) */
static void FN(affine1)(const orc_system* s, const REAL* x, REAL* f1, REAL* f2) {
    const int n = s->n, m = s->m;
    for (int i = 0; i < n * m; ++i) f2[i] = 0;
    switch (s->kind) {
    case HJBX_SYS_LINEAR: { /* linear.py:20-22: A @ x, B */
        const double* A = s->p; const double* Bm = s->p + n * n;
        for (int i = 0; i < n; ++i) {
            REAL acc = 0;
            for (int j = 0; j < n; ++j) acc += (REAL)A[i * n + j] * x[j];
            f1[i] = acc;
            for (int j = 0; j < m; ++j) f2[i * m + j] = (REAL)Bm[i * m + j];
        }
    } break;
    case HJBX_SYS_CARTPOLE:
    case HJBX_SYS_ACROBOT: { /* dynamics_basic.py:64-94 (generic manipulator form) */
        REAL M[4], C[4], G[2], Bv[2];
        FN(manip)(s, x, M, C, G, Bv);
        const REAL det = M[0] * M[3] - M[1] * M[2];
        const REAL Mi[4] = {M[3] / det, -M[1] / det, -M[2] / det, M[0] / det};
        const REAL v0 = C[0] * x[2] + C[1] * x[3] + G[0];
        const REAL v1 = C[2] * x[2] + C[3] * x[3] + G[1];
        f1[0] = x[2]; f1[1] = x[3];
        f1[2] = -(Mi[0] * v0 + Mi[1] * v1);
        f1[3] = -(Mi[2] * v0 + Mi[3] * v1);
        f2[2] = Mi[0] * Bv[0] + Mi[1] * Bv[1];
        f2[3] = Mi[2] * Bv[0] + Mi[3] * Bv[1];
    } break;
    case HJBX_SYS_QUAD2D: { /* quadrotors.py:17-46 */
        const REAL mq = (REAL)s->p[0], r = (REAL)s->p[1], I = (REAL)s->p[2], g = (REAL)s->p[3];
        f1[0] = x[3]; f1[1] = x[4]; f1[2] = x[5]; f1[3] = 0; f1[4] = -g; f1[5] = 0;
        const REAL sn = (REAL)sin((double)x[2]), c = (REAL)cos((double)x[2]);
        f2[3 * 2 + 0] = -sn / mq; f2[3 * 2 + 1] = -sn / mq;
        f2[4 * 2 + 0] = c / mq;   f2[4 * 2 + 1] = c / mq;
        f2[5 * 2 + 0] = r / I;    f2[5 * 2 + 1] = -r / I;
    } break;
    case HJBX_SYS_NEARHOVER: { /* quadrotors.py:118-149 */
        const REAL g = (REAL)s->p[0], mq = (REAL)s->p[1], kT = (REAL)s->p[2], n0 = (REAL)s->p[3];
        for (int i = 0; i < 5; ++i) f1[i] = x[5 + i];
        f1[5] = g * (REAL)tan((double)x[3]); f1[6] = g * (REAL)tan((double)x[4]); f1[7] = -g; f1[8] = 0; f1[9] = 0;
        f2[7 * 3 + 0] = kT / mq; f2[8 * 3 + 1] = n0; f2[9 * 3 + 2] = n0;
    } break;
    }
}

/* Dynamics.dynamics_step, dynamics_basic.py:96-105: f1 + f2 @ u */
static void FN(xdot1)(const orc_system* s, const REAL* x, const REAL* u, REAL* xd) {
    REAL f1[HJBX_MAX_N], f2[HJBX_MAX_N * HJBX_MAX_M];
    FN(affine1)(s, x, f1, f2);
    for (int i = 0; i < s->n; ++i) {
        REAL acc = 0;
        for (int j = 0; j < s->m; ++j) acc += f2[i * s->m + j] * u[j];
        xd[i] = f1[i] + acc;
    }
}

static void FN(clip_u)(const orc_system* s, const REAL* u, REAL* uc) { /* np.clip(u, umin, umax) */
    for (int j = 0; j < s->m; ++j) {
        REAL v = u[j];
        if (v < (REAL)s->umin[j]) v = (REAL)s->umin[j];
        if (v > (REAL)s->umax[j]) v = (REAL)s->umax[j];
        uc[j] = v;
    }
}

/* Dynamics.simulate, dynamics_basic.py:107-122 (EULER); RK4 is this build's own extension */
static void FN(simulate1)(const orc_system* s, int integrator, const REAL* x, const REAL* u, REAL* xn) {
    const int n = s->n;
    const REAL dt = (REAL)s->dt;
    REAL uc[HJBX_MAX_M], k1[HJBX_MAX_N];
    FN(clip_u)(s, u, uc);
    if (integrator == HJBX_ZOH) { /* x' = Ad x + Bd u: examples/double_integrator_optimal_time.ipynb cell 4 (LINEAR only) */
        const int m = s->m;
        const double* Ad = s->p + n * n + n * m; const double* Bd = Ad + n * n;
        for (int i = 0; i < n; ++i) {
            REAL acc = 0, bu = 0;
            for (int j = 0; j < n; ++j) acc += (REAL)Ad[i * n + j] * x[j];
            for (int j = 0; j < m; ++j) bu += (REAL)Bd[i * m + j] * uc[j];
            xn[i] = acc + bu;
        }
        FN(wrap1)(s, xn);
        return;
    }
    FN(xdot1)(s, x, uc, k1);
    if (integrator == HJBX_EULER) {
        for (int i = 0; i < n; ++i) xn[i] = x[i] + k1[i] * dt;
    } else {
        REAL k2[HJBX_MAX_N], k3[HJBX_MAX_N], k4[HJBX_MAX_N], xt[HJBX_MAX_N];
        for (int i = 0; i < n; ++i) xt[i] = x[i] + (dt / 2) * k1[i];
        FN(xdot1)(s, xt, uc, k2);
        for (int i = 0; i < n; ++i) xt[i] = x[i] + (dt / 2) * k2[i];
        FN(xdot1)(s, xt, uc, k3);
        for (int i = 0; i < n; ++i) xt[i] = x[i] + dt * k3[i];
        FN(xdot1)(s, xt, uc, k4);
        for (int i = 0; i < n; ++i) xn[i] = x[i] + (dt / 6) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
    }
    FN(wrap1)(s, xn);
}

/* error coordinates e = states_wrap(x - xf), vhjb.py:163,168 */
static void FN(err1)(const orc_system* s, const double* xf, const REAL* x, REAL* e) {
    for (int i = 0; i < s->n; ++i) e[i] = x[i] - (REAL)xf[i];
    FN(wrap1)(s, e);
}

static REAL FN(quad_form)(int n, const double* A, const REAL* v) { /* v' A v */
    REAL acc = 0;
    for (int i = 0; i < n; ++i) {
        REAL row = 0;
        for (int j = 0; j < n; ++j) row += (REAL)A[i * n + j] * v[j];
        acc += v[i] * row;
    }
    return acc;
}

/* e'e <= target_r2: the "x.T @ x <= metric" test of the time-optimal notebook (cells 7, 9) */
static int FN(in_target)(const orc_system* s, const hjbx_task* t, const REAL* x) {
    REAL e[HJBX_MAX_N], n2 = 0;
    FN(err1)(s, t->xf, x, e);
    for (int i = 0; i < s->n; ++i) n2 += e[i] * e[i];
    return n2 <= (REAL)t->target_r2;
}

/* VHJBController.running_cost, vhjb.py:162-165; HJBX_LAW_BANGBANG: np.where(norms > metric, 1, 0) (notebook cell 7) */
static REAL FN(running_cost1)(const orc_system* s, const hjbx_task* t, const REAL* x, const REAL* u) {
    REAL e[HJBX_MAX_N], du[HJBX_MAX_M];
    if (t->law == HJBX_LAW_BANGBANG) return FN(in_target)(s, t, x) ? (REAL)0 : (REAL)1;
    FN(err1)(s, t->xf, x, e);
    for (int j = 0; j < s->m; ++j) du[j] = u[j] - (REAL)t->uf[j];
    return FN(quad_form)(s->n, t->Q, e) + FN(quad_form)(s->m, t->R, du);
}

/* VHJBController.termination_cost, vhjb.py:167-169 */
static REAL FN(termination_cost1)(const orc_system* s, const hjbx_task* t, const REAL* x) {
    REAL e[HJBX_MAX_N];
    FN(err1)(s, t->xf, x, e);
    return FN(quad_form)(s->n, t->P, e);
}

/* control law of vhjb.py:218-220 (PARITY UNPINNED: vhjb.py not executable here):
 * u_raw = -Rinv f2' g / 2 + uf ; u = clip(u_raw) */
static void FN(control_from_grad1)(const orc_system* s, const hjbx_task* t, const REAL* f2, const REAL* g,
                                   REAL* u_raw, REAL* u) {
    const int n = s->n, m = s->m;
    REAL f2tg[HJBX_MAX_M];
    for (int j = 0; j < m; ++j) {
        REAL acc = 0;
        for (int i = 0; i < n; ++i) acc += f2[i * m + j] * g[i];
        f2tg[j] = acc;
    }
    if (t->law == HJBX_LAW_BANGBANG) {
        /* examples/double_integrator_optimal_time.ipynb cells 9, 11: u = -sign(gradV @ B) with |u| <= 1; for a general
         * box the minimiser of gradV.f2 u: umax where (f2'g)_j < 0, umin where > 0, and sign(0) = 0 */
        for (int j = 0; j < m; ++j)
            u_raw[j] = u[j] = (f2tg[j] < 0) ? (REAL)s->umax[j] : ((f2tg[j] > 0) ? (REAL)s->umin[j] : (REAL)0);
        return;
    }
    for (int j = 0; j < m; ++j) {
        REAL acc = 0;
        for (int k = 0; k < m; ++k) acc += (REAL)t->Rinv[j * m + k] * f2tg[k];
        u_raw[j] = -acc / 2 + (REAL)t->uf[j];
    }
    FN(clip_u)(s, u_raw, u);
}

/* hjb_loss body, vhjb.py:228-234, plus analytic d loss/d gradV. PARITY UNPINNED. */
static REAL FN(hjb_residual1)(const orc_system* s, const hjbx_task* t, int mode, const REAL* x, const REAL* g,
                              REAL done, REAL* dl_dg /* n or NULL */) {
    const int n = s->n, m = s->m;
    REAL f1[HJBX_MAX_N], f2[HJBX_MAX_N * HJBX_MAX_M], u_raw[HJBX_MAX_M], u[HJBX_MAX_M], xd[HJBX_MAX_N];
    FN(affine1)(s, x, f1, f2);
    FN(control_from_grad1)(s, t, f2, g, u_raw, u);
    REAL vdot = 0;
    for (int i = 0; i < n; ++i) {
        REAL acc = 0;
        for (int j = 0; j < m; ++j) acc += f2[i * m + j] * u[j];
        xd[i] = f1[i] + acc;
        vdot += g[i] * xd[i];
    }
    const REAL l = FN(running_cost1)(s, t, x, u);
    const REAL eps = (REAL)t->eps;
    const REAL r = (mode == HJBX_RESIDUAL_NORMALISED) ? vdot / (l + eps) + 1 : vdot + l;
    const REAL w = 1 - done;
    if (dl_dg) {
        /* du/dg = -1/2 D Rinv f2'  (m x n), D = 1 on unclipped controls */
        REAL dudg[HJBX_MAX_M * HJBX_MAX_N];
        for (int j = 0; j < m; ++j) {
            const int open = t->law == HJBX_LAW_QUADRATIC && (u_raw[j] > (REAL)s->umin[j]) && (u_raw[j] < (REAL)s->umax[j]);
            for (int i = 0; i < n; ++i) {
                REAL acc = 0;
                for (int k = 0; k < m; ++k) acc += (REAL)t->Rinv[j * m + k] * f2[i * m + k];
                dudg[j * n + i] = open ? -acc / 2 : 0;
            }
        }
        REAL f2tg[HJBX_MAX_M], Rdu[HJBX_MAX_M];
        for (int j = 0; j < m; ++j) {
            REAL a = 0, b = 0;
            for (int i = 0; i < n; ++i) a += f2[i * m + j] * g[i];
            for (int k = 0; k < m; ++k) b += (REAL)t->R[j * m + k] * (u[k] - (REAL)t->uf[k]);
            f2tg[j] = a;
            /* d/du of (u-uf)'R(u-uf) = (R + R')(u-uf); the reference's R is symmetric */
            REAL bt = 0;
            for (int k = 0; k < m; ++k) bt += (REAL)t->R[k * m + j] * (u[k] - (REAL)t->uf[k]);
            Rdu[j] = b + bt;
        }
        const REAL sg = (r > 0) - (r < 0);
        for (int i = 0; i < n; ++i) {
            REAL dv = xd[i], dl = 0;
            for (int j = 0; j < m; ++j) {
                dv += dudg[j * n + i] * f2tg[j];
                dl += dudg[j * n + i] * Rdu[j];
            }
            REAL dr = (mode == HJBX_RESIDUAL_NORMALISED) ? dv / (l + eps) - vdot * dl / ((l + eps) * (l + eps))
                                                         : dv + dl;
            dl_dg[i] = sg * w * dr;
        }
    }
    return (REAL)fabs((double)r) * w;
}

/* ValueFunctionApproximator.__call__ (vhjb.py:29-60, batch norm off) and its input gradient
 * (get_v_gradient, vhjb.py:201-202) by hand-written reverse mode. PARITY UNPINNED.
 * W1 (n,h1), W2 (h1,h2), W3 (h2,h3) row-major = Flax Dense kernels, y = x @ W. */
static REAL FN(value_grad1)(const orc_system* s, const orc_mlp* p, const REAL* W1, const REAL* W2, const REAL* W3,
                            const REAL* x, REAL* g /* n or NULL */, REAL* scratch /* 3*(h1+h2)+h3 */) {
    const int n = s->n, h1 = p->h1, h2 = p->h2, h3 = p->h3;
    REAL e[HJBX_MAX_N], z[HJBX_MAX_N];
    REAL *a1 = scratch, *a2 = a1 + h1, *y = a2 + h2, *d1 = y + h3, *d2 = d1 + h1;
    FN(err1)(s, p->xf, x, e);
    REAL ee = 0;
    for (int i = 0; i < n; ++i) { ee += e[i] * e[i]; z[i] = (e[i] - (REAL)p->mean[i]) / (REAL)p->std[i]; }
    for (int j = 0; j < h1; ++j) a1[j] = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < h1; ++j) a1[j] += z[i] * W1[i * h1 + j];
    /* activation: relu (vhjb.py:52,56) or the notebooks' tanh / sin; a1, a2 hold the activations, s1, s2 their derivatives */
    const int act = p->activation;
    REAL *s1 = d2 + h2, *s2 = s1 + h1;
#define ORC_ACT(v, dv)                                                                                             \
    do {                                                                                                           \
        const REAL zz = (v);                                                                                       \
        if (act == HJBX_ACT_TANH) { const REAL t = (REAL)tanh((double)zz); (v) = t; (dv) = 1 - t * t; }           \
        else if (act == HJBX_ACT_SIN) { (v) = (REAL)sin((double)zz); (dv) = (REAL)cos((double)zz); }               \
        else { (v) = zz > 0 ? zz : 0; (dv) = zz > 0 ? (REAL)1 : (REAL)0; }                                         \
    } while (0)
    for (int j = 0; j < h1; ++j) ORC_ACT(a1[j], s1[j]);
    for (int j = 0; j < h2; ++j) a2[j] = 0;
    for (int i = 0; i < h1; ++i) { const REAL a = a1[i]; if (a != 0) for (int j = 0; j < h2; ++j) a2[j] += a * W2[i * h2 + j]; }
    for (int j = 0; j < h2; ++j) ORC_ACT(a2[j], s2[j]);
#undef ORC_ACT
    for (int j = 0; j < h3; ++j) y[j] = 0;
    for (int i = 0; i < h2; ++i) { const REAL a = a2[i]; if (a != 0) for (int j = 0; j < h3; ++j) y[j] += a * W3[i * h3 + j]; }
    REAL V = 0;
    for (int j = 0; j < h3; ++j) V += y[j] * y[j];
    V += (REAL)p->eps_scalar * ee;
    if (g) {
        for (int i = 0; i < h2; ++i) {
            REAL acc = 0;
            if (s2[i] != 0) for (int j = 0; j < h3; ++j) acc += W3[i * h3 + j] * (2 * y[j]);
            d2[i] = acc * s2[i];
        }
        for (int i = 0; i < h1; ++i) {
            REAL acc = 0;
            if (s1[i] != 0) for (int j = 0; j < h2; ++j) acc += W2[i * h2 + j] * d2[j];
            d1[i] = acc * s1[i];
        }
        for (int i = 0; i < n; ++i) {
            REAL acc = 0;
            for (int j = 0; j < h1; ++j) acc += W1[i * h1 + j] * d1[j];
            g[i] = acc / (REAL)p->std[i] + 2 * (REAL)p->eps_scalar * e[i];
        }
    }
    return V;
}

/* closed-form controllers for one state */
static void FN(controller1)(const orc_system* s, const hjbx_controller* c, const REAL* x, REAL* u) {
    const int n = s->n, m = s->m;
    REAL ur[HJBX_MAX_M];
    if (c->kind == HJBX_CTRL_LINEAR_FEEDBACK) {
        /* lqr.py:25-26 (xf=0, uf=0, no wrap); quadrotors_model_based_controller.py:36-38, 73-75 */
        REAL e[HJBX_MAX_N];
        for (int i = 0; i < n; ++i) e[i] = x[i] - (REAL)c->xf[i];
        if (c->wrap_error) FN(wrap1)(s, e);
        for (int j = 0; j < m; ++j) {
            REAL acc = 0;
            for (int i = 0; i < n; ++i) acc += (REAL)c->K[j * n + i] * e[i];
            ur[j] = -acc + (REAL)c->uf[j];
        }
    } else if (c->kind == HJBX_CTRL_CARTPOLE_ENERGY) { /* cartpole_energy_shaping.py:65-110 */
        const REAL mc = (REAL)s->p[0], mp = (REAL)s->p[1], l = (REAL)s->p[2], g = (REAL)s->p[3];
        REAL dx[4];
        for (int i = 0; i < 4; ++i) dx[i] = x[i] - (REAL)c->xf[i];
        FN(wrap1)(s, dx);
        const REAL cth = (REAL)cos((double)x[1]), sth = (REAL)sin((double)x[1]);
        const REAL Exf = -(REAL)cos(c->xf[1]);                      /* energy(xf): xf[3] = 0 */
        const REAL de = ((REAL)0.5 * x[3] * x[3] - cth) - ((REAL)0.5 * (REAL)c->xf[3] * (REAL)c->xf[3] + Exf);
        const REAL nrm = (REAL)sqrt((double)(dx[1] * dx[1] + dx[3] * dx[3]));
        if ((REAL)fabs((double)de) < (REAL)c->eps_energy && nrm < (REAL)c->eps_state) {
            REAL acc = 0;
            for (int i = 0; i < 4; ++i) acc += (REAL)c->K[i] * dx[i];
            ur[0] = -acc;
        } else {
            const REAL u_bar = de * x[3] * cth;
            const REAL ddq1 = (REAL)c->Kes[0] * (-x[0]) + (REAL)c->Kes[1] * (-x[2]) + (REAL)c->Kes[2] * u_bar;
            const REAL ddq2 = -cth / l * ddq1 - g * sth / l;
            ur[0] = (mc + mp) * ddq1 + mp * l * cth * ddq2 - mp * l * sth * x[3] * x[3];
        }
    } else if (c->kind == HJBX_CTRL_DI_TIME_OPTIMAL) { /* get_analytical_control, double_integrator_optimal_time.ipynb cell 18 */
        const REAL p0 = x[0] - (REAL)c->xf[0], v0 = x[1] - (REAL)c->xf[1], a = (REAL)s->umax[0];
        if (p0 * p0 + v0 * v0 <= (REAL)c->eps_region) ur[0] = 0;
        else if ((v0 < 0 && p0 <= (REAL)0.5 * v0 * v0 / a) || (v0 >= 0 && p0 < -(REAL)0.5 * v0 * v0 / a)) ur[0] = a;
        else ur[0] = -a;
    } else { /* acrobot_energy_shaping.py:74-121 */
        REAL dx[4];
        dx[0] = FN(wrap_angle)(x[0] - (REAL)c->xf[0]);
        dx[1] = FN(wrap_angle)(x[1] - (REAL)c->xf[1]);
        dx[2] = x[2] - (REAL)c->xf[2];
        dx[3] = x[3] - (REAL)c->xf[3];
        if (FN(quad_form)(4, c->P, dx) < (REAL)c->eps_region) {
            REAL acc = 0;
            for (int i = 0; i < 4; ++i) acc += (REAL)c->K[i] * dx[i];
            ur[0] = -acc;
        } else {
            REAL M[4], C[4], G[2], Bv[2], xfr[4];
            FN(manip)(s, x, M, C, G, Bv);
            for (int i = 0; i < 4; ++i) xfr[i] = (REAL)c->xf[i];
            const REAL ubar = (FN(acrobot_energy)(s, x) - FN(acrobot_energy)(s, xfr)) * x[2];
            const REAL ddq2 = (REAL)c->Kes[0] * (-FN(wrap_angle)(x[1])) + (REAL)c->Kes[1] * (-x[3]) + (REAL)c->Kes[2] * ubar;
            const REAL h0 = G[0] + C[0] * x[2] + C[1] * x[3];
            const REAL h1 = G[1] + C[2] * x[2] + C[3] * x[3];
            ur[0] = (M[3] - M[1] * M[1] / M[0]) * ddq2 + h1 - M[2] / M[0] * h0;
        }
    }
    FN(clip_u)(s, ur, u);
}

static int FN(out_of_box)(const orc_system* s, const hjbx_task* t, const REAL* x) { /* vhjb.py:176-177, strict */
    REAL e[HJBX_MAX_N];
    FN(err1)(s, t->xf, x, e);
    for (int i = 0; i < s->n; ++i) /* negated non-strict form: identical for finite e, NaN (diverged env) terminates too */
        if (!(e[i] <= (REAL)t->obs_max[i]) || !(e[i] >= (REAL)t->obs_min[i])) return 1;
    return 0;
}

/* ============================ exported batch entry points ================================== */

void FN(orc_affine)(const orc_system* s, const REAL* x, REAL* f1, REAL* f2, int64_t B) {
    for (int64_t b = 0; b < B; ++b) FN(affine1)(s, x + b * s->n, f1 + b * s->n, f2 + b * s->n * s->m);
}
void FN(orc_wrap)(const orc_system* s, const REAL* x, REAL* out, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        REAL t[HJBX_MAX_N];
        for (int i = 0; i < s->n; ++i) t[i] = x[b * s->n + i];
        FN(wrap1)(s, t);
        for (int i = 0; i < s->n; ++i) out[b * s->n + i] = t[i];
    }
}
void FN(orc_manip)(const orc_system* s, const REAL* x, REAL* M, REAL* C, REAL* G, REAL* E, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        REAL Bv[2];
        FN(manip)(s, x + b * 4, M + b * 4, C + b * 4, G + b * 2, Bv);
        if (E && s->kind == HJBX_SYS_ACROBOT) E[b] = FN(acrobot_energy)(s, x + b * 4);
    }
}
void FN(orc_dynamics_step)(const orc_system* s, const REAL* x, const REAL* u, REAL* xd, int64_t B) {
    for (int64_t b = 0; b < B; ++b) FN(xdot1)(s, x + b * s->n, u + b * s->m, xd + b * s->n);
}
void FN(orc_simulate)(const orc_system* s, int integrator, const REAL* x, const REAL* u, REAL* xn, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        REAL t[HJBX_MAX_N];
        FN(simulate1)(s, integrator, x + b * s->n, u + b * s->m, t);
        for (int i = 0; i < s->n; ++i) xn[b * s->n + i] = t[i];
    }
}
/* Dynamics.get_initial_state, dynamics_basic.py:28-29: np.random.uniform(low=-std, high=std) is
 * low + (high-low)*u01 */
void FN(orc_initial_state)(const orc_system* s, const double* mean, const double* std, const REAL* u01, REAL* x0, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        REAL t[HJBX_MAX_N];
        for (int i = 0; i < s->n; ++i) {
            const REAL lo = -(REAL)std[i], hi = (REAL)std[i];
            t[i] = (lo + (hi - lo) * u01[b * s->n + i]) + (REAL)mean[i];
        }
        FN(wrap1)(s, t);
        for (int i = 0; i < s->n; ++i) x0[b * s->n + i] = t[i];
    }
}
void FN(orc_running_cost)(const orc_system* s, const hjbx_task* t, const REAL* x, const REAL* u, REAL* c, int64_t B) {
    for (int64_t b = 0; b < B; ++b) c[b] = FN(running_cost1)(s, t, x + b * s->n, u + b * s->m);
}
void FN(orc_termination_cost)(const orc_system* s, const hjbx_task* t, const REAL* x, REAL* c, int64_t B) {
    for (int64_t b = 0; b < B; ++b) c[b] = FN(termination_cost1)(s, t, x + b * s->n);
}
void FN(orc_control_from_grad)(const orc_system* s, const hjbx_task* t, const REAL* x, const REAL* g, REAL* u, int64_t B) {
    for (int64_t b = 0; b < B; ++b) {
        REAL f1[HJBX_MAX_N], f2[HJBX_MAX_N * HJBX_MAX_M], ur[HJBX_MAX_M];
        FN(affine1)(s, x + b * s->n, f1, f2);
        FN(control_from_grad1)(s, t, f2, g + b * s->n, ur, u + b * s->m);
    }
}
/* sums = {sum loss_i, sum (1-done), sum done} accumulated in double in index order */
void FN(orc_hjb_residual)(const orc_system* s, const hjbx_task* t, int mode, const REAL* x, const REAL* g,
                          const REAL* done, REAL* loss_i, REAL* dl_dg, double* sums, int64_t B) {
    double a = 0, nb = 0, nd = 0;
    for (int64_t b = 0; b < B; ++b) {
        const REAL l = FN(hjb_residual1)(s, t, mode, x + b * s->n, g + b * s->n, done[b], dl_dg ? dl_dg + b * s->n : NULL);
        if (loss_i) loss_i[b] = l;
        a += (double)l; nb += 1.0 - (double)done[b]; nd += (double)done[b];
    }
    if (sums) { sums[0] = a; sums[1] = nb; sums[2] = nd; }
}
/* termination_loss body, vhjb.py:244-247. PARITY UNPINNED. */
void FN(orc_termination_residual)(double eps, const REAL* V, const REAL* cost, const REAL* done, REAL* loss_i,
                                  REAL* dl_dV, double* sums, int64_t B) {
    double a = 0, nb = 0, nd = 0;
    for (int64_t b = 0; b < B; ++b) {
        const REAL den = cost[b] + (REAL)eps;
        const REAL r = V[b] / den - 1;
        const REAL l = (REAL)fabs((double)r) * done[b];
        if (loss_i) loss_i[b] = l;
        if (dl_dV) dl_dV[b] = (REAL)((r > 0) - (r < 0)) * done[b] / den;
        a += (double)l; nb += 1.0 - (double)done[b]; nd += (double)done[b];
    }
    if (sums) { sums[0] = a; sums[1] = nb; sums[2] = nd; }
}
void FN(orc_controller)(const orc_system* s, const hjbx_controller* c, const REAL* x, REAL* u, int64_t B) {
    for (int64_t b = 0; b < B; ++b) FN(controller1)(s, c, x + b * s->n, u + b * s->m);
}
void FN(orc_value_grad)(const orc_system* s, const orc_mlp* p, const REAL* W1, const REAL* W2, const REAL* W3,
                        const REAL* x, REAL* V, REAL* g, int64_t B) {
#pragma omp parallel
    {
        REAL* scratch = (REAL*)malloc(sizeof(REAL) * (size_t)(3 * (p->h1 + p->h2) + p->h3));
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            const REAL v = FN(value_grad1)(s, p, W1, W2, W3, x + b * s->n, g ? g + b * s->n : NULL, scratch);
            if (V) V[b] = v;
        }
        free(scratch);
    }
}

/* One iteration of rollout_trajectory (vhjb.py:175-191) for every env, given gradV of the current
 * states: the batch twin of hjbx_vhjb_step (semantics in include/hjbx.h). PARITY UNPINNED. */
void FN(orc_vhjb_step)(const orc_system* s, const hjbx_task* t, int integrator, int step, int T_max, const REAL* x,
                       const REAL* g, REAL* xn, REAL* u_out, REAL* cost_t, REAL* done_t, int32_t* done_step, REAL* resid_t,
                       int64_t B) {
    const int n = s->n, m = s->m;
    for (int64_t b = 0; b < B; ++b) {
        const REAL* xb = x + b * n;
        REAL xo[HJBX_MAX_N], u[HJBX_MAX_M];
        for (int i = 0; i < n; ++i) xo[i] = xb[i];
        for (int j = 0; j < m; ++j) u[j] = 0;
        REAL c = 0, d = 0, res = 0;
        if (done_step[b] < 0) {
            const int reached = t->law == HJBX_LAW_BANGBANG && FN(in_target)(s, t, xb);
            if (step >= T_max || reached || FN(out_of_box)(s, t, xb)) {
                c = FN(termination_cost1)(s, t, xb); d = 1; done_step[b] = step;
            } else {
                REAL f1[HJBX_MAX_N], f2[HJBX_MAX_N * HJBX_MAX_M], ur[HJBX_MAX_M];
                FN(affine1)(s, xb, f1, f2);
                FN(control_from_grad1)(s, t, f2, g + b * n, ur, u);
                const REAL l = FN(running_cost1)(s, t, xb, u);
                c = l * (REAL)s->dt;
                REAL vdot = 0;   /* vhjb.py:231-233, signed residual before the abs */
                for (int i = 0; i < n; ++i) {
                    REAL acc = 0;
                    for (int j = 0; j < m; ++j) acc += f2[i * m + j] * u[j];
                    vdot += g[b * n + i] * (f1[i] + acc);
                }
                res = vdot / (l + (REAL)t->eps) + 1;
                FN(simulate1)(s, integrator, xb, u, xo);
            }
        }
        for (int i = 0; i < n; ++i) xn[b * n + i] = xo[i];
        if (u_out) for (int j = 0; j < m; ++j) u_out[b * m + j] = u[j];
        cost_t[b] = c; done_t[b] = d;
        if (resid_t) resid_t[b] = res;
    }
}

/* rollout_trajectory (vhjb.py:171-193) env by env -- the reference's execution model (batch-1
 * serial loop, value gradient evaluated per step), parallel over envs with OpenMP when built with it.
 * Outputs time-major like the GPU path: traj (T+1,B,n), cost (T+1,B), done_step (B). Slots after
 * an env's terminal tuple hold the held state / 0. Returns the number of live env-steps executed.
 * PARITY UNPINNED. */
int64_t FN(orc_vhjb_rollout)(const orc_system* s, const hjbx_task* t, const orc_mlp* p, const REAL* W1, const REAL* W2,
                             const REAL* W3, int integrator, int T_max, const REAL* x0, REAL* traj, REAL* cost,
                             int32_t* done_step, int64_t B) {
    const int n = s->n;
    int64_t live_steps = 0;
#pragma omp parallel reduction(+ : live_steps)
    {
        REAL* scratch = (REAL*)malloc(sizeof(REAL) * (size_t)(3 * (p->h1 + p->h2) + p->h3));
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            REAL x[HJBX_MAX_N], g[HJBX_MAX_N], xn[HJBX_MAX_N], u[HJBX_MAX_M], ur[HJBX_MAX_M];
            REAL f1[HJBX_MAX_N], f2[HJBX_MAX_N * HJBX_MAX_M];
            for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
            int ds = -1;
            for (int step = 0; step <= T_max; ++step) {
                if (traj) for (int i = 0; i < n; ++i) traj[((int64_t)step * B + b) * n + i] = x[i];
                REAL c = 0;
                if (ds < 0) {
                    const int reached = t->law == HJBX_LAW_BANGBANG && FN(in_target)(s, t, x);
                    if (step == T_max || reached || FN(out_of_box)(s, t, x)) {
                        c = FN(termination_cost1)(s, t, x); ds = step;
                    } else {
                        FN(value_grad1)(s, p, W1, W2, W3, x, g, scratch);
                        FN(affine1)(s, x, f1, f2);
                        FN(control_from_grad1)(s, t, f2, g, ur, u);
                        c = FN(running_cost1)(s, t, x, u) * (REAL)s->dt;
                        FN(simulate1)(s, integrator, x, u, xn);
                        for (int i = 0; i < n; ++i) x[i] = xn[i];
                        live_steps += 1;
                    }
                }
                if (cost) cost[(int64_t)step * B + b] = c;
            }
            if (done_step) done_step[b] = ds;
        }
        free(scratch);
    }
    return live_steps;
}

/* closed loop under a closed-form controller; batch twin of hjbx_rollout_feedback */
int64_t FN(orc_rollout_feedback)(const orc_system* s, const hjbx_task* t, const hjbx_controller* c, int integrator,
                                 uint32_t flags, int T_steps, const REAL* x0, REAL* traj, REAL* u_log, REAL* cost,
                                 int32_t* done_step, REAL* total_cost, REAL* x_final, int64_t B) {
    const int n = s->n, m = s->m;
    int64_t live_steps = 0;
#pragma omp parallel for schedule(static) reduction(+ : live_steps)
    for (int64_t b = 0; b < B; ++b) {
        REAL x[HJBX_MAX_N], xn[HJBX_MAX_N], u[HJBX_MAX_M];
        for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
        int ds = -1;
        REAL tot = 0;
        for (int step = 0; step <= T_steps; ++step) {
            if (traj) for (int i = 0; i < n; ++i) traj[((int64_t)step * B + b) * n + i] = x[i];
            REAL cst = 0;
            for (int j = 0; j < m; ++j) u[j] = 0;
            if (ds < 0) {
                const int term = (flags & HJBX_ROLLOUT_TERMINATE) != 0;
                int reached = 0;
                if (flags & HJBX_ROLLOUT_STOP_AT_TARGET) { /* notebook cell 9: if x.T @ x <= metric: record t; break */
                    REAL d2 = 0;
                    for (int i = 0; i < n; ++i) d2 += (x[i] - (REAL)c->xf[i]) * (x[i] - (REAL)c->xf[i]);
                    reached = d2 <= (REAL)c->eps_region;
                }
                if (step == T_steps || reached || (term && FN(out_of_box)(s, t, x))) {
                    if (t && !reached) cst = FN(termination_cost1)(s, t, x);
                    ds = step;
                } else {
                    FN(controller1)(s, c, x, u);
                    if (t) cst = FN(running_cost1)(s, t, x, u) * (REAL)s->dt;
                    FN(simulate1)(s, integrator, x, u, xn);
                    for (int i = 0; i < n; ++i) x[i] = xn[i];
                    live_steps += 1;
                }
            }
            /* without TERMINATE the last slot carries no terminal cost (plain closed loop) */
            if (!(flags & HJBX_ROLLOUT_TERMINATE) && step == T_steps) cst = 0;
            tot += cst;
            if (cost) cost[(int64_t)step * B + b] = cst;
            if (u_log && step < T_steps) for (int j = 0; j < m; ++j) u_log[((int64_t)step * B + b) * m + j] = u[j];
        }
        if (done_step) done_step[b] = ds;
        if (total_cost) total_cost[b] = tot;
        if (x_final) for (int i = 0; i < n; ++i) x_final[b * n + i] = x[i];
    }
    return live_steps;
}

#undef FN
#undef FN_
#undef FN__
