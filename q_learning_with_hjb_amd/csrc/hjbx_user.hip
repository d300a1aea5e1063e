// hjbx_user.hip -- user-defined systems: the open half of the reference's plugin surface (dynamics/dynamics_basic.py:64-94: any subclass may
// define get_M / get_C / get_G / get_B, or get_control_affine_matrix itself).  hjbx_system_create_from_source compiles the subclass's
// device-code snippet with hiprtc into the library's own streaming kernels (hjbx_user_kernels.hpp + hjbx_stream_kernels.hpp, embedded
// below as text), keeps the code object in the handle, and the entry points of hjbx_kernels.hip launch it through the module API.  The
// MFMA entry points (value network, fused rollout, parameter gradient) exist for the built-in systems only.
//
// hiprtc is opened with dlopen at first use: libhjbx.so has no link-time dependency on it, and a process that never creates a user system
// never loads it.  Compilation needs no GPU (the CPU test compiles a snippet); modules are loaded per device at the first launch.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "hjbx_internal.hpp"
#include "hjbx_host.hpp"

// ---- the header texts handed to hiprtc, embedded at build time (host pass only) ------------------------------------------------------
#if !defined(__HIP_DEVICE_COMPILE__)
#ifndef HJBX_CSRC_DIR
#error "compile hjbx_user.hip with -DHJBX_CSRC_DIR=\"<absolute path of csrc>\""
#endif
#define HJBX_EMBED(sym, file)                                                                                                     \
    asm(".pushsection .rodata\n.global " #sym "\n.type " #sym ", @object\n" #sym ":\n.incbin \"" HJBX_CSRC_DIR "/" file "\"\n.byte 0\n" \
        ".popsection\n")
HJBX_EMBED(hjbx_src_systems, "hjbx_systems.hpp");
HJBX_EMBED(hjbx_src_stream, "hjbx_stream_kernels.hpp");
HJBX_EMBED(hjbx_src_user, "hjbx_user_kernels.hpp");
#endif
extern "C" const char hjbx_src_systems[];
extern "C" const char hjbx_src_stream[];
extern "C" const char hjbx_src_user[];

// what hiprtc's built-in runtime header does not bring: the two system headers the library's own headers include
static const char kStubRuntime[] = "// hip/hip_runtime.h: provided by hiprtc itself\n";
static const char kStubStdint[] =
    "#pragma once\n"
    "typedef signed char int8_t; typedef unsigned char uint8_t; typedef short int16_t; typedef unsigned short uint16_t;\n"
    "typedef int int32_t; typedef unsigned int uint32_t; typedef long int64_t; typedef unsigned long uint64_t; typedef unsigned long uintptr_t;\n";

// ---- hiprtc through dlopen -----------------------------------------------------------------------------------------------------
namespace {
struct Rtc {
    decltype(&hiprtcCreateProgram) create = nullptr;
    decltype(&hiprtcCompileProgram) compile = nullptr;
    decltype(&hiprtcGetProgramLogSize) log_size = nullptr;
    decltype(&hiprtcGetProgramLog) log = nullptr;
    decltype(&hiprtcGetCodeSize) code_size = nullptr;
    decltype(&hiprtcGetCode) code = nullptr;
    decltype(&hiprtcDestroyProgram) destroy = nullptr;
    bool ok = false;
};

const Rtc& rtc() {
    static Rtc r;
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        for (const char* name : {"libhiprtc.so.7", "libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
#define HJBX_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, #sym))
        HJBX_SYM(create, hiprtcCreateProgram); HJBX_SYM(compile, hiprtcCompileProgram); HJBX_SYM(log_size, hiprtcGetProgramLogSize);
        HJBX_SYM(log, hiprtcGetProgramLog); HJBX_SYM(code_size, hiprtcGetCodeSize); HJBX_SYM(code, hiprtcGetCode);
        HJBX_SYM(destroy, hiprtcDestroyProgram);
#undef HJBX_SYM
        r.ok = r.create && r.compile && r.log_size && r.log && r.code_size && r.code && r.destroy;
    });
    return r;
}

thread_local std::string g_compile_log;

struct UserProgram {
    std::vector<char> code;                          // the gfx950 code object
    std::mutex mu;
    hipModule_t mod[kMaxDevices] = {};
    std::map<std::string, hipFunction_t> fn[kMaxDevices];
};
}  // namespace

extern "C" size_t hjbx_last_compile_log(char* buf, size_t buflen) {
    const size_t len = g_compile_log.size();
    if (buf && buflen) {
        const size_t n = len < buflen - 1 ? len : buflen - 1;
        memcpy(buf, g_compile_log.data(), n);
        buf[n] = '\0';
    }
    return len;
}

void hjbx_user_release(void* up) {
    UserProgram* u = static_cast<UserProgram*>(up);
    if (!u) return;
    for (int d = 0; d < kMaxDevices; ++d)
        if (u->mod[d]) (void)hipModuleUnload(u->mod[d]);
    delete u;
}

extern "C" int hjbx_system_create_from_source(int user_kind, const char* device_source, int n, int m, double dt, const double* umin,
                                              const double* umax, const double* params, int n_params, hjbx_system** out) {
    if (!out) return hjbx_set_error(HJBX_EINVAL, "out is NULL");
    *out = nullptr;
    g_compile_log.clear();
    if (user_kind != HJBX_USER_AFFINE && user_kind != HJBX_USER_MANIPULATOR) return hjbx_set_error(HJBX_EINVAL, "unknown user system kind %d", user_kind);
    if (!device_source || !umin || !umax) return hjbx_set_error(HJBX_EINVAL, "device_source / umin / umax must be non-NULL");
    if (n < 1 || n > HJBX_MAX_N || m < 1 || m > HJBX_MAX_M) return hjbx_set_error(HJBX_EINVAL, "user system needs 1<=n<=%d, 1<=m<=%d", HJBX_MAX_N, HJBX_MAX_M);
    if (user_kind == HJBX_USER_MANIPULATOR && n % 2) return hjbx_set_error(HJBX_EINVAL, "a manipulator state is (q, dq): n must be even, got %d", n);
    if (n_params < 0 || n_params > HJBX_USER_MAX_PARAMS || (n_params > 0 && !params))
        return hjbx_set_error(HJBX_EINVAL, "user system takes 0..%d parameters", HJBX_USER_MAX_PARAMS);
    if (!(dt > 0) || !(dt < 1e300)) return hjbx_set_error(HJBX_EINVAL, "dt must be positive and finite");
    for (int j = 0; j < m; ++j)
        if (!(umin[j] <= umax[j])) return hjbx_set_error(HJBX_EINVAL, "umin[%d] > umax[%d]", j, j);
    const Rtc& R = rtc();
    if (!R.ok) {
        const char* why = dlerror();       // (one call: dlerror clears the message it returns)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_system_create_from_source: libhiprtc.so could not be loaded (%s)", why ? why : "symbols missing");
    }

    const char* headers[] = {hjbx_src_systems, hjbx_src_stream, hjbx_src_user, device_source, kStubRuntime, kStubStdint};
    const char* names[] = {"hjbx_systems.hpp", "hjbx_stream_kernels.hpp", "hjbx_user_kernels.hpp", "hjbx_user_snippet.hpp", "hip/hip_runtime.h", "stdint.h"};
    hiprtcProgram prog = nullptr;
    if (R.create(&prog, "#include \"hjbx_user_kernels.hpp\"\n", "hjbx_user_system.hip", 6, headers, names) != HIPRTC_SUCCESS)
        return hjbx_set_error(HJBX_EHIP, "hiprtcCreateProgram failed");
    char dn[32], dm[32], dp[32], dk[32];
    snprintf(dn, sizeof dn, "-DHJBX_USER_N=%d", n);
    snprintf(dm, sizeof dm, "-DHJBX_USER_M=%d", m);
    snprintf(dp, sizeof dp, "-DHJBX_USER_NP=%d", n_params > 0 ? n_params : 1);
    snprintf(dk, sizeof dk, "-DHJBX_USER_KIND=%d", user_kind);
    // the flags of the library's own build: -ffp-contract=on keeps a user system's fused rollout bit-identical to its step kernels
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", dn, dm, dp, dk};
    const hiprtcResult rc = R.compile(prog, 8, opts);
    size_t ls = 0;
    if (R.log_size(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        g_compile_log.resize(ls);
        if (R.log(prog, &g_compile_log[0]) != HIPRTC_SUCCESS) g_compile_log.clear();
    }
    if (rc != HIPRTC_SUCCESS) {
        R.destroy(&prog);
        return hjbx_set_error(HJBX_EINVAL, "hjbx_system_create_from_source: the device source does not compile (hjbx_last_compile_log has the "
                                           "compiler's messages): %.300s", g_compile_log.c_str());
    }
    size_t cs = 0;
    UserProgram* u = new (std::nothrow) UserProgram();
    if (!u || R.code_size(prog, &cs) != HIPRTC_SUCCESS || cs == 0) {
        R.destroy(&prog);
        delete u;
        return hjbx_set_error(HJBX_EHIP, "hjbx_system_create_from_source: no code object");
    }
    u->code.resize(cs);
    const hiprtcResult rg = R.code(prog, u->code.data());
    R.destroy(&prog);
    if (rg != HIPRTC_SUCCESS) { delete u; return hjbx_set_error(HJBX_EHIP, "hiprtcGetCode failed"); }

    hjbx_system* s = new (std::nothrow) hjbx_system();
    if (!s) { delete u; return hjbx_set_error(HJBX_EINVAL, "out of host memory"); }
    memset(s, 0, sizeof(*s));
    s->kind = HJBX_SYS_USER; s->n = n; s->m = m; s->dt = dt; s->n_params = n_params;
    for (int j = 0; j < m; ++j) { s->umin[j] = umin[j]; s->umax[j] = umax[j]; }
    for (int i = 0; i < n_params; ++i) s->p[i] = params[i];
    s->user = u;
    *out = s;
    return HJBX_OK;
}

// Launch `kernel` (an extern "C" name of hjbx_user_kernels.hpp) of this handle's code object: `grid` workgroups of 256 threads on `stream`,
// args = pointers to the kernel's arguments in order (the first one a UserBlob).  The module is loaded on the current device on first use.
int hjbx_user_launch(const hjbx_system* s, const char* kernel, unsigned grid, void** args, void* stream) {
    UserProgram* u = static_cast<UserProgram*>(s->user);
    if (!u) return hjbx_set_error(HJBX_EINVAL, "system handle has no user program");
    const int dev = hjbx_current_device();
    if (dev < 0) return hjbx_set_error(HJBX_ENODEVICE, "%s: no HIP device", kernel);
    hipFunction_t f = nullptr;
    {
        std::lock_guard<std::mutex> lock(u->mu);
        if (!u->mod[dev]) {
            const hipError_t e = hipModuleLoadData(&u->mod[dev], u->code.data());
            if (e != hipSuccess) { u->mod[dev] = nullptr; return hjbx_set_error(HJBX_EHIP, "hipModuleLoadData: %s", hipGetErrorString(e)); }
        }
        auto it = u->fn[dev].find(kernel);
        if (it == u->fn[dev].end()) {
            const hipError_t e = hipModuleGetFunction(&f, u->mod[dev], kernel);
            if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hipModuleGetFunction(%s): %s", kernel, hipGetErrorString(e));
            u->fn[dev][kernel] = f;
        } else {
            f = it->second;
        }
    }
    const hipError_t e = hipModuleLaunchKernel(f, grid, 1, 1, 256, 1, 1, 0, (hipStream_t)stream, args, nullptr);
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "%s: %s", kernel, hipGetErrorString(e));
    return HJBX_OK;
}
