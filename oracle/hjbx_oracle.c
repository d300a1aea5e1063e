/*
 * hjbx_oracle.c -- CPU oracle for the hjbx hot path.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement (gcc, optional OpenMP) of the reference arithmetic on the path
 * Dynamics.simulate / get_control_affine_matrix + VHJBController control law, HJB residual and
 * rollout loop.  See oracle_impl.h for the per-function reference citations, oracle/README.md for
 * how it is pinned.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the resulting liborc.so; the product library (libhjbx.so) never does and has no CPU path.
 *
 * It shares include/hjbx.h with the product for the enum values and the POD descriptor layouts
 * only (hjbx_task, hjbx_controller); that header contains no arithmetic.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/hjbx.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* host-side description of a system: what Dynamics.__init__ stores (dynamics_basic.py:17-26) */
typedef struct orc_system {
    int32_t kind, n, m, _pad;
    double dt;
    double umin[HJBX_MAX_M], umax[HJBX_MAX_M];
    double p[2 * (HJBX_MAX_N * HJBX_MAX_N + HJBX_MAX_N * HJBX_MAX_M)]; /* same packing as hjbx_system_create (A, B [, Ad, Bd]) */
} orc_system;

/* value-network hyper-parameters (weights are passed as separate host arrays) */
typedef struct orc_mlp {
    int32_t h1, h2, h3, activation; /* hjbx_activation */
    double mean[HJBX_MAX_N], std[HJBX_MAX_N], xf[HJBX_MAX_N];
    double eps_scalar;
} orc_mlp;

size_t orc_sizeof_system(void) { return sizeof(orc_system); }
size_t orc_sizeof_mlp(void) { return sizeof(orc_mlp); }
size_t orc_sizeof_task(void) { return sizeof(hjbx_task); }
size_t orc_sizeof_controller(void) { return sizeof(hjbx_controller); }
#ifdef _OPENMP
#include <omp.h>
#endif
/* number of OpenMP threads the batch loops will use; n > 0 sets it first */
int orc_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}
int orc_has_openmp(void) {
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

#define REAL double
#define SFX f64
#include "oracle_impl.h"
#undef REAL
#undef SFX

#define REAL float
#define SFX f32
#include "oracle_impl.h"
#undef REAL
#undef SFX
