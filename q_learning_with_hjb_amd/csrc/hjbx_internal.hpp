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
};

// records the calling thread's error message and returns `code`
int hjbx_set_error(int code, const char* fmt, ...);

// current value of a hjbx_option (hjbx_set_option), 0 for an unknown one
int hjbx_option_value(int option);
