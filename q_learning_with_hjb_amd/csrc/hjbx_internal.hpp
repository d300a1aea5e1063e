// hjbx_internal.hpp -- definitions shared by the translation units of libhjbx.so (not part of the ABI)
#pragma once
#include "../../include/hjbx.h"

// the opaque handle of include/hjbx.h: what Dynamics.__init__ stores (dynamics_basic.py:17-26)
struct hjbx_system {
    int kind, n, m;
    double dt;
    double umin[HJBX_MAX_M], umax[HJBX_MAX_M];
    double p[2 * (HJBX_MAX_N * HJBX_MAX_N + HJBX_MAX_N * HJBX_MAX_M)];  // packing documented at hjbx_system_kind
    int n_params;
    void* user;   // HJBX_SYS_USER: the run-time compiled program (hjbx_user.hip); NULL otherwise
};

// HJBX_SYS_USER (hjbx_user.hip): launch an extern "C" kernel of the handle's code object (args[0] points at the system blob), and free it
int hjbx_user_launch(const hjbx_system* s, const char* kernel, unsigned grid, void** args, void* stream);
void hjbx_user_release(void* user_program);

// records the calling thread's error message and returns `code`
int hjbx_set_error(int code, const char* fmt, ...);

// current value of a hjbx_option (hjbx_set_option), 0 for an unknown one
int hjbx_option_value(int option);

// the cooperative single-kernel parameter gradient (hjbx_train_coop.hip), called by hjbx_value_loss_grad_f32 / hjbx_value_loss_adam_f32 after
// argument validation.  fuse == NULL: the partial sums are reduced into `flat`; fuse != NULL (hjbx_adam.hpp): reduced, mixed and applied to the
// weights by Adam in the same epilogue kernel, `flat` unused
struct FuseArgs;
int hjbx_train_coop(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost, const float* done,
                    float* flat, void* workspace, int64_t B, void* stream, const FuseArgs* fuse);
size_t hjbx_train_coop_workspace_bytes(int64_t B, int n);
