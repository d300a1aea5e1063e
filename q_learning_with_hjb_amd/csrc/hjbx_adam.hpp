// hjbx_adam.hpp -- optax.adam (reference controller/vhjb.py:120, 262-263) as device code shared by hjbx_mix_adam_f32 (hjbx_fit.hip) and the
// fused reduce + mix + Adam epilogue of the cooperative parameter-gradient kernel (hjbx_train_coop.hip).  Not part of the ABI.
//   m <- m + (1 - b1)(g - m);  v <- b2 v + (1 - b2) g^2;  t <- t + 1;  w <- w - lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#pragma once
#include <hip/hip_runtime.h>

#include "hjbx_internal.hpp"

struct AdamArgs {
    float* w[3]; float* m[3]; float* v[3];
    float* step[3]; unsigned int* ticket;
    int64_t end0, end1, P;             // parameter k lives in tensor 0 if k < end0, 1 if k < end1, else 2
    double lr, b1, b2; float eps;
};

// what follows the gradient in params_update: the mix (vhjb.py:284), the losses (:285-288) and train's device-side bookkeeping (:320-323)
struct MixArgs {
    const float* reg_dev; float reg_host, eps;
    float* losses; float* loss_accum; int32_t* step_counter;
};

// The minibatch of an update, assembled on the device (hjbx_replay_gather_f32; DataLoader(shuffle, drop_last) + np_collate, vhjb.py:151-154, 314):
// rows perm[k batch .. (k + 1) batch) of the replay buffer.  enabled == 0: nothing to gather.
struct GatherArgs {
    const float* bx; const float* bc; const float* bd; int64_t capacity; int n;
    const int32_t* perm; int64_t perm_len; const float* reg_table; int64_t table_len; int64_t batch;
    float* ox; float* oc; float* od; float* oreg;
    int enabled;
};
// A counter that has run past the epoch (or a permutation entry outside the buffer) must not become an out-of-bounds access -- a GPU fault takes
// the whole node down: such a launch gathers nothing and leaves the regularisation weight at NaN, so the step it feeds fails visibly.
__device__ __forceinline__ bool gather_in_range(const GatherArgs& g, int64_t k) {
    return k >= 0 && (k + 1) * g.batch <= g.perm_len && (!g.reg_table || k < g.table_len);
}
__device__ __forceinline__ void gather_elements(const GatherArgs& g, int64_t k, int64_t first, int64_t stride) {   // element t = (sample, column)
    if (!gather_in_range(g, k)) return;
    for (int64_t t = first; t < g.batch * g.n; t += stride) {
        const int64_t smp = t / g.n;
        const int c = (int)(t - smp * g.n);
        const int64_t row = g.perm[k * g.batch + smp];
        if (row < 0 || row >= g.capacity) continue;
        g.ox[t] = g.bx[row * g.n + c];
        if (c == 0) { g.oc[smp] = g.bc[row]; g.od[smp] = g.bd[row]; }
    }
}
__device__ __forceinline__ void gather_reg(const GatherArgs& g, int64_t k) {
    if (g.oreg && g.reg_table) g.oreg[0] = gather_in_range(g, k) ? g.reg_table[k] : __builtin_nanf("");
}

// the epilogue of the fused update (hjbx_value_loss_adam_f32): reduction of the per-workgroup partial sums, mix and Adam in one kernel -- and,
// optionally, the NEXT update's minibatch (index step_counter + 1), so that a fit-phase graph is two kernels per update
struct FuseArgs { AdamArgs a; MixArgs mx; GatherArgs next; };

// host: validate an hjbx_adam_state and turn it into kernel arguments (`who` prefixes the error message)
inline int adam_args_from(const hjbx_adam_state* adam, const char* who, AdamArgs& a) {
    if (!adam) return hjbx_set_error(HJBX_EINVAL, "%s: NULL Adam state", who);
    int64_t P = 0;
    for (int i = 0; i < 3; ++i) {
        if (!adam->param[i] || !adam->exp_avg[i] || !adam->exp_avg_sq[i] || adam->numel[i] <= 0)
            return hjbx_set_error(HJBX_EINVAL, "%s: parameter tensor %d: NULL pointer or non-positive size", who, i);
        if (!adam->step[i]) return hjbx_set_error(HJBX_EINVAL, "%s: NULL step count of tensor %d", who, i);
        a.w[i] = adam->param[i]; a.m[i] = adam->exp_avg[i]; a.v[i] = adam->exp_avg_sq[i]; a.step[i] = adam->step[i];
        P += adam->numel[i];
    }
    if (!adam->ticket) return hjbx_set_error(HJBX_EINVAL, "%s: NULL ticket", who);
    if (!(adam->lr > 0) || !(adam->beta1 >= 0 && adam->beta1 < 1) || !(adam->beta2 >= 0 && adam->beta2 < 1) || !(adam->eps >= 0))
        return hjbx_set_error(HJBX_EINVAL, "%s: bad hyper-parameters", who);
    a.ticket = adam->ticket;
    a.end0 = adam->numel[0]; a.end1 = adam->numel[0] + adam->numel[1]; a.P = P;
    a.lr = adam->lr; a.b1 = adam->beta1; a.b2 = adam->beta2; a.eps = (float)adam->eps;
    return HJBX_OK;
}

struct AdamCoef { float t, step_size, c2s, w1, b2, w2; };

// Every workgroup reads the step count BEFORE the last one to finish (adam_finish's ticket) writes the incremented value.  sc: 2 floats of LDS.
// Two halves so that a kernel can put independent work between them: adam_coef_begin (thread 0 evaluates the two bias corrections in double:
// a microsecond of latency), then -- after a __syncthreads of the caller's -- adam_coef_end.
__device__ __forceinline__ float adam_coef_begin(const AdamArgs& a, float* sc) {
    const float t = a.step[0][0] + 1.0f;
    if (threadIdx.x == 0) {
        sc[0] = (float)(a.lr / (1.0 - pow(a.b1, (double)t)));
        sc[1] = (float)sqrt(1.0 - pow(a.b2, (double)t));
    }
    return t;
}
__device__ __forceinline__ AdamCoef adam_coef_end(const AdamArgs& a, float t, const float* sc) {
    AdamCoef c;
    c.t = t;
    c.step_size = sc[0]; c.c2s = sc[1];
    c.w1 = (float)(1.0 - a.b1); c.b2 = (float)a.b2; c.w2 = (float)(1.0 - a.b2);
    return c;
}
__device__ __forceinline__ AdamCoef adam_coef(const AdamArgs& a, float* sc) {   // both halves (contains a __syncthreads)
    const float t = adam_coef_begin(a, sc);
    __syncthreads();
    return adam_coef_end(a, t, sc);
}

__device__ __forceinline__ void adam_element(const AdamArgs& a, const AdamCoef& c, int which, int64_t j, float g) {
    float m = a.m[which][j], v = a.v[which][j];
    m = m + c.w1 * (g - m);
    v = c.b2 * v + c.w2 * g * g;
    a.m[which][j] = m;
    a.v[which][j] = v;
    a.w[which][j] -= c.step_size * m / (sqrtf(v) / c.c2s + a.eps);
}

// After a workgroup's elements (contains a __syncthreads): the last workgroup to arrive writes the new step count, the losses and the counters.
// h, tl: hjb and termination loss (sums already divided by their counts); reg: the regularisation weight of this update.
// -> true for thread 0 of that last workgroup (for work that must follow every other workgroup's reads)
__device__ __forceinline__ bool adam_finish(const AdamArgs& a, const AdamCoef& c, const MixArgs& mx, float h, float tl, float reg) {
    __syncthreads();
    bool last = false;
    if (threadIdx.x == 0) {
        const unsigned int old = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == gridDim.x - 1) {   // all workgroups have read the old step count
            a.step[0][0] = c.t; a.step[1][0] = c.t; a.step[2][0] = c.t;
            __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (mx.losses) { mx.losses[0] = h + reg * tl; mx.losses[1] = h; mx.losses[2] = tl; }
            if (mx.loss_accum) { mx.loss_accum[0] += h + reg * tl; mx.loss_accum[1] += h; mx.loss_accum[2] += tl; }   // total_losses += ... (vhjb.py:320-322)
            if (mx.step_counter) mx.step_counter[0] += 1;                                                              // update_counter += 1 (vhjb.py:323)
            last = true;
        }
    }
    return last;
}
