// placeholder until the MFMA kernel lands (next commit)
#include "../../include/hjbx.h"
int hjbx_set_error(int code, const char* fmt, ...);
extern "C" int hjbx_value_grad_f32(const hjbx_system*, const hjbx_mlp*, const float*, float*, float*, int64_t, void*) {
    return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_grad_f32: not built yet");
}
