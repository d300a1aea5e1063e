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
