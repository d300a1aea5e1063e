/*
 * hjbx.h -- C ABI of libhjbx.so: MI355X (gfx950) batched control-affine rollouts + HJB residuals.
 *
 * This is the drop-in boundary for the hot path of HaoxiangYou/Q_Learning_with_HJB (SURVEY.md
 * section 8b).  The reference has no FFI: its "operator API" is two Python base classes
 * (dynamics/dynamics_basic.py:7-122 `Dynamics`, controller/controller_basic.py:1-5 `Controller`)
 * plus the math inside controller/vhjb.py.  Each entry point below names the reference statement(s)
 * it replaces; q_learning_with_hjb_amd/_abi.py is the ctypes binding a maintainer would add
 * (INTEGRATION.md shows the reference-side patch).
 *
 * Conventions
 *  - Every array argument is a raw DEVICE pointer (hipMalloc / torch.Tensor.data_ptr()) to a
 *    contiguous row-major buffer; `hjbx_task` / `hjbx_controller` / `hjbx_mlp` descriptors are
 *    HOST structs read during the call (copied into kernel arguments).  No torch types cross.
 *  - State batches are (B, n) row-major ("batch_size, state_dim": dynamics_basic.py:58-60),
 *    controls (B, m), f2 (B, n, m).  Trajectory slabs are time-major: (T+1, B, n).
 *  - `_f32` entry points take float buffers, `_f64` double buffers.  Descriptor fields are double
 *    and are rounded to float once per call for `_f32`.
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Kernels are enqueued and
 *    the call returns; nothing here synchronises, allocates device memory or copies to the host.
 *  - Return value: HJBX_OK (0) or a negative hjbx_status.  hjbx_last_error() gives the message of
 *    the calling thread's last failure.  No C++ exception crosses this boundary.
 *  - Handles are immutable after creation and may be shared between host threads.
 */
#ifndef HJBX_H
#define HJBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HJBX_VERSION 111 /* major*100 + minor */
#define HJBX_MAX_N 10    /* largest state dimension (NearHoverQuadcopter) */
#define HJBX_MAX_M 3     /* largest control dimension */

typedef enum hjbx_status {
    HJBX_OK = 0,
    HJBX_EINVAL = -1,       /* bad argument (NULL pointer, bad size, bad enum) */
    HJBX_EUNSUPPORTED = -2, /* valid request this build has no kernel for */
    HJBX_EHIP = -3,         /* HIP runtime error (message carries hipGetErrorString) */
    HJBX_ENODEVICE = -4     /* no usable gfx950 device */
} hjbx_status;

/* Which dynamics class of the reference the handle stands for. */
typedef enum hjbx_system_kind {
    HJBX_SYS_LINEAR = 0,    /* dynamics/linear.py:7-22      params = A (n*n row-major), B (n*m) [, Ad (n*n), Bd (n*m) for HJBX_ZOH] */
    HJBX_SYS_CARTPOLE = 1,  /* dynamics/cartpole.py:10-64   params = mc, mp, l, g                    */
    HJBX_SYS_ACROBOT = 2,   /* dynamics/acrobot.py:19-81    params = m1, m2, l1, l2, I1, I2, g       */
    HJBX_SYS_QUAD2D = 3,    /* dynamics/quadrotors.py:9-70  params = m, r, I, g                      */
    HJBX_SYS_NEARHOVER = 4, /* dynamics/quadrotors.py:102-170 params = g, m, kT, n0                  */
    HJBX_SYS_USER = 5       /* a user-defined Dynamics subclass: created by hjbx_system_create_from_source only */
} hjbx_system_kind;

/* What a user-defined system supplies (hjbx_system_create_from_source). */
typedef enum hjbx_user_kind {
    HJBX_USER_AFFINE = 0,      /* the subclass overrides get_control_affine_matrix (like dynamics/linear.py:20-22, quadrotors.py:17-46):
                                  the source defines wrap(x) and affine(x, f1, f2) */
    HJBX_USER_MANIPULATOR = 1  /* the subclass defines get_M / get_C / get_G / get_B and inherits the generic manipulator form
                                  f1 = [dq; -inv(M)(C dq + G)], f2 = [0; inv(M) B] (dynamics/dynamics_basic.py:64-94): the source
                                  defines wrap(x), get_M, get_C, get_G, get_B; n even, q = x[:n/2] */
} hjbx_user_kind;
#define HJBX_USER_MAX_PARAMS 16

typedef enum hjbx_integrator {
    HJBX_EULER = 0, /* x' = wrap(x + dt*xdot): the reference's integrator, dynamics_basic.py:120 (parity mode) */
    HJBX_RK4 = 1,   /* classic RK4 with zero-order-hold control, wrap applied once at the end (new mode)         */
    HJBX_ZOH = 2    /* LINEAR systems only: exact zero-order-hold step x' = Ad x + Bd u, the discretisation the
                       reference's LQR / time-optimal notebooks use (scipy.signal.cont2discrete:
                       examples/double_integrator_optimal_time.ipynb cell 4); needs Ad, Bd in the handle          */
} hjbx_integrator;

typedef enum hjbx_residual_mode {
    HJBX_RESIDUAL_NORMALISED = 0, /* |gradV.xdot/(l+eps) + 1| (1-done): controller/vhjb.py:233       */
    HJBX_RESIDUAL_RAW = 1         /* |gradV.xdot + l| (1-done): examples/cartpole_balancing.ipynb cell 11; with
                                     HJBX_LAW_BANGBANG and done = 0 this is pd_hjb_loss of the time-optimal notebook (cell 11) */
} hjbx_residual_mode;

/* How u is obtained from gradV, and which running cost goes with it (hjbx_task.law). */
typedef enum hjbx_control_law {
    HJBX_LAW_QUADRATIC = 0, /* u = clip(-Rinv f2' gradV / 2 + uf), l = e'Qe + (u-uf)'R(u-uf): controller/vhjb.py:162-165, 218-220 */
    HJBX_LAW_BANGBANG = 1   /* time-optimal learning (examples/double_integrator_optimal_time.ipynb cells 7, 9, 11):
                               u_j = umax_j if (f2' gradV)_j < 0, umin_j if > 0, 0 if == 0  (= -sign(gradV @ B) for +-1 limits);
                               l = 1 outside the target ball e'e > target_r2, 0 inside; rollouts end inside the ball
                               (terminal tuple, cost e'Pe) as well as outside the observation box; d u / d gradV = 0 */
} hjbx_control_law;

typedef enum hjbx_controller_kind {
    HJBX_CTRL_LINEAR_FEEDBACK = 0, /* u = clip(-K e + uf): controller/lqr.py:25-26, quadrotors_model_based_controller.py:36-38, 73-75 */
    HJBX_CTRL_CARTPOLE_ENERGY = 1, /* controller/cartpole_energy_shaping.py:65-110 */
    HJBX_CTRL_ACROBOT_ENERGY = 2,  /* controller/acrobot_energy_shaping.py:74-121  */
    HJBX_CTRL_DI_TIME_OPTIMAL = 3  /* double integrator (LINEAR n=2, m=1), analytic bang-bang law with its switching curve:
                                      get_analytical_control, examples/double_integrator_optimal_time.ipynb cell 18;
                                      u = 0 inside the target ball e'e <= eps_region, else +-umax */
} hjbx_controller_kind;

/* rollout flags */
#define HJBX_ROLLOUT_TERMINATE 1u /* stop an environment when wrap(x-xf) leaves [obs_min, obs_max] (vhjb.py:176-181) */
#define HJBX_ROLLOUT_STOP_AT_TARGET 2u /* stop an environment once e'e <= ctrl.eps_region, e = x - ctrl.xf, checked before
                                          the control of each step (time-to-origin loops of the time-optimal notebook,
                                          cell 9); done_step = index of that step */

/* Process-wide tuning / test knobs (hjbx_set_option; a negative value queries).  Options 0-2 do not change results. */
typedef enum hjbx_option {
    HJBX_OPT_ROLLOUT_SCHEDULE = 0,         /* work distribution of hjbx_vhjb_rollout_f32: 0 (default) = equal static shares per workgroup, with
                                              the shares of workgroups that have not started taken over by the waves that finish first;
                                              1 = device-wide tile queue (one atomic per 32-environment tile) */
    HJBX_OPT_ROLLOUT_EXTRA_WORKGROUPS = 1, /* TEST HOOK: launch this many workgroups more than there are CUs (they cannot be resident until
                                              others finish -- the situation the take-over above exists for); default 0 */
    HJBX_OPT_STREAM_ROWS = 2,              /* TUNING: rows per thread of the float32 streaming kernels (simulate, vhjb_step, hjb_residual):
                                              1, 2 or 4; 0 (default) = the library's choice */
    HJBX_OPT_MLP_ARITHMETIC = 3            /* arithmetic of the ReLU value network inside hjbx_value_grad_f32 / hjbx_vhjb_rollout_f32 and of the eight
                                              128 / 64-wide products of hjbx_value_loss_grad_f32 (mode 1 runs those on the f32 MFMA) (THE ONE KNOB
                                              THAT CHANGES RESULTS, within float32 rounding; inputs, outputs, accumulation, layer 1 and everything
                                              outside the network are float32 in every mode; tanh and sin networks always run mode 0):
                                              0 (DEFAULT) = float32 MFMA, bitwise an fmaf chain: the arithmetic of the reference's float32 network
                                                  (controller/vhjb.py:17-60);
                                              1 = OPT-IN bf16x3: every float32 operand split EXACTLY into three bfloat16 pieces, the six largest piece
                                                  products on the bf16 matrix cores (dropped products <= 2^-23 of each term);
                                              2 = OPT-IN f16x2: every operand scaled by a power of two (ONE exponent per weight matrix; one per
                                                  environment and product for the activations) and rounded to two float16 pieces, the three
                                                  largest piece products on the f16 matrix cores.  An operand within 2^16 of its scaling maximum
                                                  keeps 22 significant bits (perturbation <= 3 x 2^-22 = 7e-7 of its term); a smaller one loses
                                                  low bits of its lo piece (float16 subnormals), i.e. the error bound is ABSOLUTE: 2^-39 of the
                                                  matrix's largest |weight| (of the environment's largest activation) per operand, so one outlier
                                                  weight 2^17 above the rest degrades every other entry of its matrix.
                                              Modes 1 and 2 are faster and narrower than the reference's arithmetic: never the default.
                                              tests/test_gpu_f32_parity.py runs every test in every mode against the same bounds, and
                                              test_fused_value_grad_split_arithmetics_stay_within_their_stated_bound checks modes 1 / 2 against
                                              mode 0 on the device. */,
    HJBX_OPT_TRAIN_KERNEL = 4              /* implementation of hjbx_value_loss_grad_f32 in the float32 arithmetic (results agree to float32 summation
                                              order): 0 (default) = the cooperative single kernel (no scratch in HBM, a tile's chains split over the four
                                              waves of a workgroup; ReLU, tanh, sin); 1 = the round-2 pair of kernels (chains + outer products through a
                                              5-KB-per-sample scratch; ReLU only) -- kept for A/B measurements and for the f16x2 arithmetic */
} hjbx_option;

typedef struct hjbx_system hjbx_system; /* opaque */

/* Task = the cost / target / observation-box part of VHJBControllerConfig
 * (configs/controller/vhjb_controller_config.py:47-55) plus P from vhjb.py:156-160. Row-major,
 * only the leading n*n / m*m / n / m entries are read. */
typedef struct hjbx_task {
    double Q[HJBX_MAX_N * HJBX_MAX_N];
    double R[HJBX_MAX_M * HJBX_MAX_M];
    double Rinv[HJBX_MAX_M * HJBX_MAX_M];
    double P[HJBX_MAX_N * HJBX_MAX_N]; /* terminal cost e'Pe, vhjb.py:167-169 */
    double xf[HJBX_MAX_N];
    double uf[HJBX_MAX_M];
    double obs_min[HJBX_MAX_N]; /* bounds on the error coordinates wrap(x-xf), strict compares */
    double obs_max[HJBX_MAX_N];
    double eps;                 /* VHJBControllerConfig.epsilon */
    int32_t law;                /* hjbx_control_law; 0 = the reference's VHJBController */
    int32_t _pad;
    double target_r2;           /* HJBX_LAW_BANGBANG: squared radius of the target ball ("metric", notebook cell 7) */
} hjbx_task;

/* Closed-form feedback laws. */
typedef struct hjbx_controller {
    int32_t kind;       /* hjbx_controller_kind */
    int32_t wrap_error; /* LINEAR_FEEDBACK: e = wrap_error ? wrap(x-xf) : x-xf */
    double K[HJBX_MAX_M * HJBX_MAX_N]; /* (m, n) row-major state-feedback gain */
    double xf[HJBX_MAX_N];
    double uf[HJBX_MAX_M];
    double P[HJBX_MAX_N * HJBX_MAX_N]; /* ACROBOT_ENERGY: LQR region is e'Pe < eps_region */
    double Kes[3];      /* energy-shaping gains (cartpole [4,4,10], acrobot [1,2,1]) */
    double eps_energy;  /* CARTPOLE_ENERGY: |E-E(xf)| < eps_energy ... */
    double eps_state;   /* ... and ||(dtheta_err, dtheta_dot)|| < eps_state selects the LQR branch */
    double eps_region;  /* ACROBOT_ENERGY: LQR region; DI_TIME_OPTIMAL / STOP_AT_TARGET: squared radius of the target ball */
} hjbx_controller;

typedef enum hjbx_activation {
    HJBX_ACT_RELU = 0, /* controller/vhjb.py:52,56 and examples/drone_hovering.ipynb, 10D_quadcopte.ipynb */
    HJBX_ACT_TANH = 1, /* examples/cartpole_balancing.ipynb cell 6 */
    HJBX_ACT_SIN = 2   /* examples/double_integrator_optimal_time.ipynb cell 5 (sin and cos evaluated to ~1e-7 absolute, branch-free) */
} hjbx_activation;

/* Value network of controller/vhjb.py:17-60 (no bias, BatchNorm off): device weight pointers,
 * Flax Dense layout (in, out) row-major, y = x @ W. */
typedef struct hjbx_mlp {
    const void* W1; /* (n,  h1) */
    const void* W2; /* (h1, h2) */
    const void* W3; /* (h2, h3) */
    int32_t h1, h2, h3;
    int32_t activation;      /* hjbx_activation between the Dense layers */
    double mean[HJBX_MAX_N]; /* normalization_mean */
    double std[HJBX_MAX_N];  /* normalization_std  */
    double xf[HJBX_MAX_N];
    double eps_scalar;       /* epsilon_scalar */
} hjbx_mlp;

/* ---- library / handles --------------------------------------------------------------------- */
int hjbx_version(void);
/* Copies the calling thread's last error message (NUL terminated) into buf; returns its length. */
size_t hjbx_last_error(char* buf, size_t buflen);
/* Number of visible HIP devices whose arch is gfx950 (0 when there is none / no driver). */
int hjbx_device_count(void);
/* Sets a hjbx_option and returns its previous value (value < 0: query only); HJBX_EINVAL for an unknown option. */
int hjbx_set_option(int option, int value);

/* Replaces Dynamics.__init__ + subclass __init__ (dynamics_basic.py:17-26 etc.). */
int hjbx_system_create(int kind, int n, int m, double dt, const double* umin, const double* umax,
                       const double* params, int n_params, hjbx_system** out);
void hjbx_system_destroy(hjbx_system* sys);
/* The OPEN half of the reference's plugin surface: "any subclass of Dynamics" (dynamics/dynamics_basic.py:7-122).  `device_source` is the text
 * of the subclass's per-state methods as device code (member functions of a struct template on the scalar type T with the parameters in
 * p[0..n_params-1]; the exact contract is at the top of csrc/hjbx_user_kernels.hpp).  It is compiled at run time (hiprtc, gfx950, the
 * library's own build flags) INTO THE LIBRARY'S OWN STREAMING KERNELS, float32 and float64: the returned handle works with every
 * hjbx_*_f32 / _f64 entry point of the HJBX_DECLARE block below (wrap, affine, dynamics_step, simulate with HJBX_EULER / HJBX_RK4,
 * initial_state, costs, control_from_grad, hjb_residual, vhjb_step, controller and rollout_feedback with HJBX_CTRL_LINEAR_FEEDBACK).
 * The matrix-core entry points (hjbx_value_grad_f32, hjbx_vhjb_rollout_f32, hjbx_value_loss_grad_f32) exist for the built-in systems
 * only and return HJBX_EUNSUPPORTED for such a handle.  Compilation needs no GPU.  HJBX_EINVAL when the source does not compile
 * (hjbx_last_compile_log returns the compiler's messages for the calling thread's last call), HJBX_EUNSUPPORTED when libhiprtc.so
 * is not available. */
int hjbx_system_create_from_source(int user_kind, const char* device_source, int n, int m, double dt, const double* umin,
                                   const double* umax, const double* params, int n_params, hjbx_system** out);
size_t hjbx_last_compile_log(char* buf, size_t buflen);
/* Dynamics.get_dimension, dynamics_basic.py:31-36 */
int hjbx_dims(const hjbx_system* sys, int* n, int* m);
/* Bytes of device scratch the reducing entry points need (hjb_residual, termination_residual).  The workspace holds the
 * per-workgroup partial sums and the arrival counters of the in-kernel final reduction: it must be 16-byte aligned and ZERO-FILLED
 * ONCE after allocation (hipMemset); every call leaves the counters at zero again, so no per-call memset is needed.  One
 * workspace serves one stream at a time (calls on the same stream may share it; concurrent streams need one each). */
size_t hjbx_reduce_workspace_bytes(void);

#define HJBX_DECLARE(T, SFX)                                                                          \
    /* Dynamics.get_control_affine_matrix (dynamics_basic.py:64-94, linear.py:20-22,                 \
       quadrotors.py:17-46, 118-149): f1 (B,n), f2 (B,n,m). */                                       \
    int hjbx_affine_##SFX(const hjbx_system* sys, const T* x, T* f1, T* f2, int64_t B, void* stream); \
    /* Dynamics.states_wrap (cartpole.py:52-64, acrobot.py:72-81, quadrotors.py:48-70, 151-170);     \
       out may alias x (the NumPy branch of the reference wraps in place). */                        \
    int hjbx_wrap_##SFX(const hjbx_system* sys, const T* x, T* out, int64_t B, void* stream);         \
    /* Dynamics.dynamics_step (dynamics_basic.py:96-105): xdot = f1 + f2 u, u NOT clipped. */         \
    int hjbx_dynamics_step_##SFX(const hjbx_system* sys, const T* x, const T* u, T* xdot, int64_t B,  \
                                 void* stream);                                                       \
    /* Dynamics.simulate (dynamics_basic.py:107-122): u clipped to [umin,umax], one integrator step, \
       wrap.  x_next may alias x. */                                                                  \
    int hjbx_simulate_##SFX(const hjbx_system* sys, int integrator, const T* x, const T* u,           \
                            T* x_next, int64_t B, void* stream);                                      \
    /* Dynamics.get_initial_state (dynamics_basic.py:28-29) for B environments:                      \
       x0 = wrap(-x0_std + 2 x0_std * u01 + x0_mean); u01 (B,n) are caller-supplied uniforms so the   \
       RNG stays with the caller (the reference uses NumPy's global MT19937). */                      \
    int hjbx_initial_state_##SFX(const hjbx_system* sys, const double* x0_mean, const double* x0_std, \
                                 const T* u01, T* x0, int64_t B, void* stream);                       \
    /* VHJBController.running_cost (vhjb.py:162-165): cost (B,) = e'Qe + (u-uf)'R(u-uf). */           \
    int hjbx_running_cost_##SFX(const hjbx_system* sys, const hjbx_task* task, const T* x,            \
                                const T* u, T* cost, int64_t B, void* stream);                        \
    /* VHJBController.termination_cost (vhjb.py:167-169): cost (B,) = e'Pe. */                        \
    int hjbx_termination_cost_##SFX(const hjbx_system* sys, const hjbx_task* task, const T* x,        \
                                    T* cost, int64_t B, void* stream);                                \
    /* Control law of get_control_efforts_with_additional_term (vhjb.py:218-220):                    \
       u = clip(-Rinv f2' gradV / 2 + uf, umin, umax). */                                             \
    int hjbx_control_from_grad_##SFX(const hjbx_system* sys, const hjbx_task* task, const T* x,       \
                                     const T* gradV, T* u, int64_t B, void* stream);                  \
    /* hjb_loss body (vhjb.py:227-241) forward + analytic d(loss_i)/d(gradV).           \
       done (B,) is the 0/1 mask as T.  loss_i (B,) and dloss_dgrad (B,n) may be NULL.                \
       sums[0..2] = {sum loss_i, sum (1-done), sum done}, reduced deterministically (fixed order, no  \
       float atomics) inside the same launch through the caller's workspace (see                      \
       hjbx_reduce_workspace_bytes()); sums may be NULL. */                    \
    int hjbx_hjb_residual_##SFX(const hjbx_system* sys, const hjbx_task* task, int mode, const T* x,  \
                                const T* gradV, const T* done, T* loss_i, T* dloss_dgrad, T* sums,    \
                                void* workspace, int64_t B, void* stream);                            \
    /* termination_loss body (vhjb.py:243-253): loss_i = |V/(cost+eps) - 1| done,                    \
       dloss_dV = sign(.) done/(cost+eps); sums as above. */                                          \
    int hjbx_termination_residual_##SFX(double eps, const T* V, const T* cost, const T* done,         \
                                        T* loss_i, T* dloss_dV, T* sums, void* workspace, int64_t B,  \
                                        void* stream);                                                \
    /* One iteration of rollout_trajectory's loop (vhjb.py:175-191) for B environments, given the    \
       value gradient of the current states.  step t in [0,T]: environments with done_step<0 are     \
       live.  A live env outside the observation box (or t==T) emits (cost=e'Pe, done=1), latches    \
       done_step=t and holds its state; otherwise u from gradV, cost=l(x,u)*dt, done=0, x_next=       \
       simulate(x,u).  Dead envs emit cost=0, done=0 and hold.  done_step (B,) int32 must be          \
       initialised to -1 before t=0.  u_out may be NULL.  resid_t (B,) may be NULL; when given it     \
       receives the normalised HJB residual gradV.xdot/(l+eps) + 1 (vhjb.py:233, signed, before the   \
       abs) of every live, non-terminating environment at its current state and 0 elsewhere -- the    \
       residual comes for free here because u, xdot and l are already in registers. */                \
    int hjbx_vhjb_step_##SFX(const hjbx_system* sys, const hjbx_task* task, int integrator, int t,    \
                             int T_max, const T* x, const T* gradV, T* x_next, T* u_out, T* cost_t,   \
                             T* done_t, int32_t* done_step, T* resid_t, int64_t B, void* stream);     \
    /* Controller.get_control_efforts for the closed-form controllers, u (B,m). */      \
    int hjbx_controller_##SFX(const hjbx_system* sys, const hjbx_controller* ctrl, const T* x, T* u,  \
                              int64_t B, void* stream);                                               \
    /* Whole closed loop in one kernel: for t in 0..T-1: u=ctrl(x); log; x=simulate(x,u)              \
       (scripts/test_vhjb_policy.py:143-151, cartpole_energy_shaping.py:123-125).  With              \
       HJBX_ROLLOUT_TERMINATE it follows rollout_trajectory (vhjb.py:171-193) with `task`:            \
       per-step cost l*dt, terminal e'Pe, done_step.  Outputs (any may be NULL):                      \
       traj (T+1,B,n) time-major states, u_log (T,B,m), cost (T+1,B), done_step (B,) int32,           \
       total_cost (B,) sum of the emitted costs (get_trajectory_cost, vhjb.py:195-199).               \
       task may be NULL when neither cost nor TERMINATE is requested. */                              \
    int hjbx_rollout_feedback_##SFX(const hjbx_system* sys, const hjbx_task* task,                    \
                                    const hjbx_controller* ctrl, int integrator, uint32_t flags,      \
                                    int T_steps, const T* x0, T* traj, T* u_log, T* cost,             \
                                    int32_t* done_step, T* total_cost, T* x_final, int64_t B,         \
                                    void* stream);

HJBX_DECLARE(float, f32)
HJBX_DECLARE(double, f64)
#undef HJBX_DECLARE

/* ValueFunctionApproximator.__call__ + get_v_gradient (vhjb.py:17-60, 201-202) fused on the matrix
 * cores: V (B,) and gradV (B,n) = dV/dx in one pass, weights staged in LDS.  f32 only (the JAX side
 * of the reference is float32).  V or gradV may be NULL. */
int hjbx_value_grad_f32(const hjbx_system* sys, const hjbx_mlp* mlp, const float* x, float* V,
                        float* gradV, int64_t B, void* stream);

/* rollout_trajectory (vhjb.py:171-193) for B environments, `n_steps` consecutive iterations t = t_first ..
 * t_first+n_steps-1 of its loop in ONE launch: per step the value gradient (as hjbx_value_grad_f32, weights staged
 * in LDS once per launch) followed by exactly hjbx_vhjb_step_f32's per-environment code; the state stays in
 * registers between steps.  Bit-identical to calling those two entry points n_steps times.
 *   x (B,n): states at step t_first.            done_step (B,) int32 in/out (-1 = live), as hjbx_vhjb_step.
 *   traj (n_steps+1,B,n) or NULL: slab k = state at step t_first+k (slab 0 = x).     x_out (B,n) or NULL: final state.
 *   cost, done (n_steps,B): the tuples emitted at each step.   resid (n_steps,B) or NULL.   u_log (n_steps,B,m) or NULL.
 * A whole reference rollout is t_first = 0, n_steps = T_max + 1 (the last iteration emits the forced terminal tuple).
 *   env_order (B,) int32 or NULL: a permutation of 0..B-1 = the order in which environments are packed into the 32-wide
 *   tiles of the kernel (all arrays stay indexed by environment).  A tile whose environments have all finished skips the
 *   value network and only writes its log rows, so listing the live environments first (a stable sort by done_step >= 0,
 *   refreshed between launches) makes a batch in which most environments have terminated cost what its live part
 *   costs.  Results do not depend on the order.
 *   x may be the same buffer as slab 0 of traj (each row is written back with the value just read).
 *   workspace: hjbx_rollout_workspace_bytes() bytes of 16-byte aligned device memory, ZERO-FILLED ONCE after allocation; the
 *   kernel keeps its work-distribution words there and leaves them zeroed.  One workspace per stream in flight. */
size_t hjbx_rollout_workspace_bytes(void);
int hjbx_vhjb_rollout_f32(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int integrator, int t_first,
                          int n_steps, int T_max, const float* x, float* traj, float* u_log, float* cost, float* done,
                          float* resid, int32_t* done_step, float* x_out, const int32_t* env_order, int64_t B, void* workspace,
                          void* stream);

/* The parameter gradient of one value-learning step (vhjb.py:227-253 and the jax.grad calls of :282-284) for a minibatch of B samples
 * (x (B,n), cost (B,), done (B,) as 0/1 floats), fused on the matrix cores:
 *   flat = [ d(sum_b hjb_loss_b)/dW1 (n,h1) | /dW2 (h1,h2) | /dW3 (h2,h3) | d(sum_b termination_loss_b)/dW1 | /dW2 | /dW3 |
 *            sum_b hjb_loss_b, sum_b termination_loss_b, sum_b (1-done_b), sum_b done_b ]                  (2P + 4 floats)
 * i.e. the gradients of the loss SUMS (the caller divides by the counts, vhjb.py:241, 253, and mixes with the regularisation weight,
 * :284) -- the buffer a data-parallel step all-reduces once.  hjb_loss is a function of dV/dx, so its gradient is a second-order
 * reverse sweep; both are evaluated in closed form (no autograd graph).  `mode` = hjbx_residual_mode.  Deterministic: no float atomics,
 * fixed summation order (the order depends on B and the device's CU count only).  ReLU, tanh and (state dimension <= 4) sin networks with features [128,128,64]
 * (HJBX_EUNSUPPORTED otherwise: the PyTorch autograd path remains); W1, W2, W3 16-byte aligned.  workspace: hjbx_value_loss_grad_workspace_bytes(B) bytes (it depends on
 * HJBX_OPT_MLP_ARITHMETIC / HJBX_OPT_TRAIN_KERNEL: ask again after changing them), 256-byte aligned, need not be initialised. */
size_t hjbx_value_loss_grad_workspace_bytes(int64_t B);
int hjbx_value_loss_grad_f32(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x,
                             const float* cost, const float* done, float* flat, void* workspace, int64_t B, void* stream);

/* The step between that buffer (after the all-reduce, if any) and Adam: mixed[k] = flat[k] / (#interior + eps) + reg * flat[P + k] / (#done + eps)
 * for the P = n_params parameter entries (vhjb.py:241, 253, 284) and losses[0..2] = {hjb + reg termination, hjb, termination} (vhjb.py:285-288;
 * losses may be NULL).  reg is read from reg_dev[0] when reg_dev is non-NULL (a device scalar survives hipGraph replay), else from `reg`.
 * loss_accum (3 floats, may be NULL): the three losses are ADDED to it -- `total_losses += ...` of train (vhjb.py:320-322) on the device;
 * step_counter (one int32, may be NULL): incremented by one -- `update_counter += 1` (vhjb.py:323). */
int hjbx_mix_gradients_f32(const float* flat, int64_t n_params, const float* reg_dev, double reg, double eps, float* mixed, float* losses,
                           float* loss_accum, int32_t* step_counter, void* stream);

/* The same step with optax.adam (vhjb.py:120, 262-263: b1, b2, eps as given, eps_root 0) applied in the same launch, the mixed gradient never
 * materialised:  g as above;  m <- m + (1 - b1)(g - m);  v <- b2 v + (1 - b2) g^2;  t <- t + 1;
 *                w <- w - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps).
 * param / exp_avg / exp_avg_sq: the three weight matrices W1, W2, W3 (row-major, numel[i] entries, in the order of `flat`) and Adam's moments,
 * all updated in place; step[i]: device scalar holding t as float32, one per tensor (the layout torch.optim.Adam(fused / capturable) keeps, so
 * its state tensors can be handed over as they are; t is read from step[0], t + 1 written to all three, which may alias); ticket: one zero-initialised device word for the kernel's use, left zero. */
typedef struct hjbx_adam_state {
    float* param[3];
    float* exp_avg[3];
    float* exp_avg_sq[3];
    int64_t numel[3];
    float* step[3];
    unsigned int* ticket;
    double lr, beta1, beta2, eps;
} hjbx_adam_state;
int hjbx_mix_adam_f32(const float* flat, const float* reg_dev, double reg, double eps, const hjbx_adam_state* adam, float* losses, float* loss_accum,
                      int32_t* step_counter, void* stream);

/* params_update (vhjb.py:255-288) in ONE call for a single process: hjbx_value_loss_grad_f32 followed by hjbx_mix_adam_f32, with the flat
 * buffer never materialised on the default (cooperative) path -- the reduction of the per-workgroup partial sums, the division by the counts,
 * the mix, the losses and Adam's update run in one epilogue kernel, i.e. two launches per update.  Arguments as in those two entry points;
 * adam->param[] must be the network's W1, W2, W3.  workspace: hjbx_value_loss_adam_workspace_bytes(B) bytes, 256-byte aligned, need not be
 * initialised.  (A data-parallel step needs the flat buffer for its all-reduce: hjbx_value_loss_grad_f32, all-reduce, hjbx_mix_adam_f32.) */
size_t hjbx_value_loss_adam_workspace_bytes(int64_t B);
/* optional last duty of that call: assemble the NEXT update's minibatch (what hjbx_replay_gather_f32 would do for index step_counter + 1, same
 * arguments and bounds behaviour) inside the epilogue kernel, so that a captured fit-phase update is two launches.  reg_out may be the buffer
 * reg_dev points to (it is written after every read of this update). */
typedef struct hjbx_next_minibatch {
    const float* buf_x; const float* buf_cost; const float* buf_done;
    int64_t capacity; int n;
    const int32_t* perm; int64_t perm_len;
    const float* reg_table; int64_t table_len;
    int64_t batch;
    float* xs; float* costs; float* dones; float* reg_out;
} hjbx_next_minibatch;
int hjbx_value_loss_adam_f32(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost,
                             const float* done, const float* reg_dev, double reg, double eps, const hjbx_adam_state* adam, float* losses,
                             float* loss_accum, int32_t* step_counter, const hjbx_next_minibatch* next, void* workspace, int64_t B, void* stream);

/* The minibatch of one update, assembled on the device: DataLoader(batch_size, shuffle=True, drop_last=True) + np_collate of the reference
 * (vhjb.py:151-154, 314; utils/utils.py:7-14) for a device-resident replay buffer (buf_x (capacity, n), buf_cost, buf_done (capacity,)):
 *   xs[s] = buf_x[perm[k * batch + s]] (likewise costs, dones), s = 0..batch-1, with k = step_dev[0] read ON THE DEVICE (NULL: k = 0), and
 *   reg_out[0] = reg_table[k] when reg_out is non-NULL (the regularisation weight of update k of the epoch, vhjb.py:323-324).
 * perm: perm_len int32 row indices < capacity (one random permutation of the buffer per epoch); reg_table: table_len floats.  A counter
 * beyond the permutation or the table, or an index outside the buffer, gathers nothing (no out-of-bounds access) and sets reg_out to NaN.
 * With step_dev = the counter hjbx_mix_gradients_f32 / hjbx_mix_adam_f32 increment, a captured hipGraph of gather ->
 * hjbx_value_loss_grad_f32 -> mix + Adam replays with no host-side work between two updates. */
int hjbx_replay_gather_f32(const float* buf_x, const float* buf_cost, const float* buf_done, int64_t capacity, int n, const int32_t* perm,
                           int64_t perm_len, const int32_t* step_dev, const float* reg_table, int64_t table_len, int64_t batch, float* xs,
                           float* costs, float* dones, float* reg_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HJBX_H */
