/*
 * asan_check.c -- AddressSanitizer / UBSan exercise of the CPU oracle.  TEST INFRASTRUCTURE (like everything under oracle/).
 * `make -C oracle asan` compiles hjbx_oracle.c into this driver with -fsanitize=address,undefined and runs it: every batch
 * entry point, both precisions, all five systems, ragged batch sizes, exactly-sized heap buffers (an out-of-bounds access of the
 * restatement -- the code the GPU kernels are checked against -- would abort here).  The GPU itself cannot run sanitizers on this pool.
 */
#include <stdio.h>
#include "hjbx_oracle.c"

static double frand(unsigned* st) { *st = *st * 1664525u + 1013904223u; return (double)(*st >> 8) / 16777216.0; }

#define CHECK_ALL(REAL, SFX)                                                                                                        \
    static int run_##SFX(const orc_system* s, unsigned seed) {                                                                       \
        const int n = s->n, m = s->m;                                                                                                \
        const int64_t B = 37;                                                                                                        \
        unsigned st = seed;                                                                                                          \
        REAL* x = malloc(sizeof(REAL) * B * n); REAL* u = malloc(sizeof(REAL) * B * m); REAL* g = malloc(sizeof(REAL) * B * n);      \
        REAL* f1 = malloc(sizeof(REAL) * B * n); REAL* f2 = malloc(sizeof(REAL) * B * n * m); REAL* xn = malloc(sizeof(REAL) * B * n); \
        REAL* c = malloc(sizeof(REAL) * B); REAL* d = malloc(sizeof(REAL) * B); REAL* li = malloc(sizeof(REAL) * B);                  \
        REAL* dg = malloc(sizeof(REAL) * B * n); REAL* V = malloc(sizeof(REAL) * B); REAL* uo = malloc(sizeof(REAL) * B * m);         \
        int32_t* ds = malloc(sizeof(int32_t) * B); REAL* rs = malloc(sizeof(REAL) * B);                                               \
        for (int64_t i = 0; i < B * n; ++i) { x[i] = (REAL)(2 * frand(&st) - 1); g[i] = (REAL)(6 * frand(&st) - 3); }                 \
        for (int64_t i = 0; i < B * m; ++i) u[i] = (REAL)(4 * frand(&st) - 2);                                                        \
        for (int64_t i = 0; i < B; ++i) { d[i] = (REAL)(frand(&st) < 0.3); ds[i] = -1; }                                              \
        hjbx_task t; memset(&t, 0, sizeof(t));                                                                                        \
        for (int i = 0; i < n; ++i) { t.Q[i * n + i] = 1; t.P[i * n + i] = 2; t.obs_min[i] = -0.9; t.obs_max[i] = 0.9; }              \
        for (int j = 0; j < m; ++j) { t.R[j * m + j] = 1; t.Rinv[j * m + j] = 1; }                                                    \
        t.eps = 1e-10;                                                                                                                \
        orc_affine_##SFX(s, x, f1, f2, B); orc_wrap_##SFX(s, x, xn, B); orc_dynamics_step_##SFX(s, x, u, xn, B);                      \
        orc_simulate_##SFX(s, 0, x, u, xn, B); orc_simulate_##SFX(s, 1, x, u, xn, B);                                                 \
        orc_running_cost_##SFX(s, &t, x, u, c, B); orc_termination_cost_##SFX(s, &t, x, c, B);                                        \
        orc_control_from_grad_##SFX(s, &t, x, g, uo, B);                                                                              \
        double sums[3];                                                                                                               \
        orc_hjb_residual_##SFX(s, &t, 0, x, g, d, li, dg, sums, B); orc_hjb_residual_##SFX(s, &t, 1, x, g, d, li, dg, sums, B);       \
        for (int64_t i = 0; i < B; ++i) V[i] = (REAL)frand(&st);                                                                      \
        orc_termination_residual_##SFX(1e-10, V, c, d, li, rs, sums, B);                                                              \
        orc_vhjb_step_##SFX(s, &t, 0, 3, 9, x, g, xn, uo, c, li, ds, rs, B);                                                          \
        orc_mlp p; memset(&p, 0, sizeof(p)); p.h1 = 128; p.h2 = 128; p.h3 = 64; p.eps_scalar = 1e-3;                                   \
        for (int i = 0; i < n; ++i) p.std[i] = 1;                                                                                     \
        REAL* W1 = malloc(sizeof(REAL) * n * 128); REAL* W2 = malloc(sizeof(REAL) * 128 * 128); REAL* W3 = malloc(sizeof(REAL) * 128 * 64); \
        for (int i = 0; i < n * 128; ++i) W1[i] = (REAL)(frand(&st) - 0.5);                                                            \
        for (int i = 0; i < 128 * 128; ++i) W2[i] = (REAL)(0.2 * (frand(&st) - 0.5));                                                  \
        for (int i = 0; i < 128 * 64; ++i) W3[i] = (REAL)(0.2 * (frand(&st) - 0.5));                                                   \
        orc_value_grad_##SFX(s, &p, W1, W2, W3, x, V, dg, B);                                                                          \
        const int T = 7;                                                                                                              \
        REAL* traj = malloc(sizeof(REAL) * (T + 1) * B * n); REAL* cost = malloc(sizeof(REAL) * (T + 1) * B);                          \
        int64_t live = orc_vhjb_rollout_##SFX(s, &t, &p, W1, W2, W3, 0, T, x, traj, cost, ds, B);                                      \
        hjbx_controller k; memset(&k, 0, sizeof(k)); k.kind = HJBX_CTRL_LINEAR_FEEDBACK; k.wrap_error = 1;                             \
        for (int j = 0; j < m; ++j) for (int i = 0; i < n; ++i) k.K[j * n + i] = 0.3 * (i + 1) / (j + 1);                               \
        REAL* ulog = malloc(sizeof(REAL) * T * B * m); REAL* tot = malloc(sizeof(REAL) * B);                                           \
        live += orc_rollout_feedback_##SFX(s, &t, &k, 1, HJBX_ROLLOUT_TERMINATE, T, x, traj, ulog, cost, ds, tot, xn, B);              \
        free(x); free(u); free(g); free(f1); free(f2); free(xn); free(c); free(d); free(li); free(dg); free(V); free(uo); free(ds);    \
        free(rs); free(W1); free(W2); free(W3); free(traj); free(cost); free(ulog); free(tot);                                         \
        return live >= 0 ? 0 : 1;                                                                                                      \
    }
CHECK_ALL(double, f64)
CHECK_ALL(float, f32)

int main(void) {
    orc_system s[5];
    memset(s, 0, sizeof(s));
    const int kind[5] = {HJBX_SYS_LINEAR, HJBX_SYS_CARTPOLE, HJBX_SYS_ACROBOT, HJBX_SYS_QUAD2D, HJBX_SYS_NEARHOVER};
    const int n[5] = {2, 4, 4, 6, 10}, m[5] = {1, 1, 1, 2, 3};
    const double par[5][8] = {{0, 1, 0, 0, 0, 1}, {1, 0.1, 1, 9.81}, {8, 8, 0.5, 1, 2, 8, 10}, {1, 0.25, 0.0625, 9.81}, {9.81, 1, 0.91, 10}};
    int rc = 0;
    for (int k = 0; k < 5; ++k) {
        s[k].kind = kind[k]; s[k].n = n[k]; s[k].m = m[k]; s[k].dt = 0.02;
        for (int j = 0; j < m[k]; ++j) { s[k].umin[j] = -3; s[k].umax[j] = 3; }
        for (int i = 0; i < 8; ++i) s[k].p[i] = par[k][i];
        rc |= run_f64(&s[k], 17u + k);
        rc |= run_f32(&s[k], 91u + k);
    }
    printf("oracle sanitizer run: %s\n", rc ? "FAILED" : "ok");
    return rc;
}
