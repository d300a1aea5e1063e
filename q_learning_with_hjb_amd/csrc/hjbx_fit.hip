// The small kernels of the fit phase around the parameter gradient (hjbx_train*.hip): the minibatch of one update gathered on the device,
// the division by the counts + the mix of the two gradients, and the same with the Adam step applied in the same launch.
//
// reference: controller/vhjb.py:151-154, 314 (DataLoader, shuffle, drop_last), :241, 253, 284-288 (means, mix, losses), :120 and :262-263
// (optax.adam(lr), optax.apply_updates), :320-324 (running loss sums, update counter, regularisation schedule).
#include <hip/hip_runtime.h>

#include "hjbx_internal.hpp"
#include "hjbx_adam.hpp"

// ---- divide by the counts and mix (vhjb.py:241, 253, 284) --------------------------------------------------------------------------
// mixed = g_h / (#interior + eps) + reg g_t / (#done + eps); losses = {hjb + reg termination, hjb, termination}.  One launch instead of the
// dozen element-wise launches the same arithmetic costs in torch (at a minibatch of 256 the whole step is launch bound).
__global__ __launch_bounds__(256) void k_mix_gradients(const float* __restrict__ flat, int64_t P, const float* __restrict__ reg_dev, float reg_host,
                                                      float eps, float* __restrict__ mixed, float* __restrict__ losses, float* __restrict__ loss_accum,
                                                      int32_t* __restrict__ step_counter) {
    const float reg = reg_dev ? reg_dev[0] : reg_host;
    const float ih = 1.0f / (flat[2 * P + 2] + eps), it = 1.0f / (flat[2 * P + 3] + eps);
    const float wt = reg * it;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < P; k += (int64_t)gridDim.x * 256) mixed[k] = flat[k] * ih + flat[P + k] * wt;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const float h = flat[2 * P] * ih, t = flat[2 * P + 1] * it;
        if (losses) { losses[0] = h + reg * t; losses[1] = h; losses[2] = t; }
        if (loss_accum) { loss_accum[0] += h + reg * t; loss_accum[1] += h; loss_accum[2] += t; }   // total_losses += ... of train (vhjb.py:320-322)
        if (step_counter) step_counter[0] += 1;                                                     // update_counter += 1 (vhjb.py:323)
    }
}

extern "C" int hjbx_mix_gradients_f32(const float* flat, int64_t n_params, const float* reg_dev, double reg, double eps, float* mixed, float* losses,
                                      float* loss_accum, int32_t* step_counter, void* stream) {
    if (!flat || !mixed || n_params <= 0) return hjbx_set_error(HJBX_EINVAL, "hjbx_mix_gradients_f32: NULL buffer or non-positive parameter count");
    const int grid = (int)((n_params + 255) / 256 < 512 ? (n_params + 255) / 256 : 512);
    hipLaunchKernelGGL(k_mix_gradients, dim3(grid), dim3(256), 0, (hipStream_t)stream, flat, n_params, reg_dev, (float)reg, (float)eps, mixed, losses, loss_accum,
                       step_counter);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_mix_gradients_f32: %s", hipGetErrorString(e));
    return HJBX_OK;
}

// ---- the minibatch of one update, assembled on the device ---------------------------------------------------------------------------------
// DataLoader(batch_size, shuffle=True, drop_last=True) + np_collate of the reference (vhjb.py:151-154, 314; utils/utils.py:7-14) for the
// device-resident replay buffer: minibatch k of an epoch is rows perm[k batch .. (k + 1) batch) of the buffer.  k is read from DEVICE memory
// (the counter hjbx_mix_gradients_f32 increments) and so is the regularisation weight of that update (a per-epoch table of the schedule,
// vhjb.py:323-324): a captured hipGraph of gather -> gradient -> mix -> Adam replays with NO host-side work between two updates.
__global__ __launch_bounds__(256) void k_replay_gather(GatherArgs g, const int32_t* __restrict__ step) {
    const int64_t k = step ? (int64_t)step[0] : 0;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t == 0) gather_reg(g, k);
    gather_elements(g, k, t, (int64_t)gridDim.x * 256);
}

extern "C" int hjbx_replay_gather_f32(const float* buf_x, const float* buf_cost, const float* buf_done, int64_t capacity, int n, const int32_t* perm,
                                      int64_t perm_len, const int32_t* step_dev, const float* reg_table, int64_t table_len, int64_t batch, float* xs,
                                      float* costs, float* dones, float* reg_out, void* stream) {
    if (!buf_x || !buf_cost || !buf_done || !perm || !xs || !costs || !dones)
        return hjbx_set_error(HJBX_EINVAL, "hjbx_replay_gather_f32: NULL buffer");
    if (n < 1 || n > HJBX_MAX_N || batch < 0 || capacity < 1 || perm_len < 0 || table_len < 0)
        return hjbx_set_error(HJBX_EINVAL, "hjbx_replay_gather_f32: bad n, batch, capacity or length");
    if (reg_out && !reg_table) return hjbx_set_error(HJBX_EINVAL, "hjbx_replay_gather_f32: reg_out needs reg_table");
    if (batch == 0) return HJBX_OK;
    if (batch > perm_len) return hjbx_set_error(HJBX_EINVAL, "hjbx_replay_gather_f32: a minibatch of %lld rows from a permutation of %lld", (long long)batch, (long long)perm_len);
    const int64_t nthreads = batch * n;
    const GatherArgs g{buf_x, buf_cost, buf_done, capacity, n, perm, perm_len, reg_table, table_len, batch, xs, costs, dones, reg_out, 1};
    hipLaunchKernelGGL(k_replay_gather, dim3((unsigned)((nthreads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g, step_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_replay_gather_f32: %s", hipGetErrorString(e));
    return HJBX_OK;
}

// ---- mix + Adam in one launch -------------------------------------------------------------------------------------------------------------
// optax.adam(lr) of the reference (vhjb.py:120: b1 0.9, b2 0.999, eps 1e-8, eps_root 0) applied to the mixed gradient without materialising it:
//   g = flat[k] / (#interior + eps) + reg flat[P + k] / (#done + eps), then the update of hjbx_adam.hpp
// (the form torch's fused Adam evaluates, so the state tensors of a torch.optim.Adam can be handed over as they are).  At the reference's
// minibatch the update is launch bound: this replaces the mix kernel and Adam's two launches.
__global__ __launch_bounds__(256) void k_mix_adam(const float* __restrict__ flat, AdamArgs a, MixArgs mx) {
    __shared__ float sc[2];
    const int64_t P = a.P;
    const AdamCoef c = adam_coef(a, sc);
    const float reg = mx.reg_dev ? mx.reg_dev[0] : mx.reg_host;
    const float ih = 1.0f / (flat[2 * P + 2] + mx.eps), it = 1.0f / (flat[2 * P + 3] + mx.eps);
    const float wt = reg * it;
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < P; k += (int64_t)gridDim.x * 256) {
        const float g = flat[k] * ih + flat[P + k] * wt;
        const int which = k < a.end0 ? 0 : (k < a.end1 ? 1 : 2);
        const int64_t j = k - (which == 0 ? 0 : (which == 1 ? a.end0 : a.end1));
        adam_element(a, c, which, j, g);
    }
    adam_finish(a, c, mx, flat[2 * P] * ih, flat[2 * P + 1] * it, reg);
}

extern "C" int hjbx_mix_adam_f32(const float* flat, const float* reg_dev, double reg, double eps, const hjbx_adam_state* adam, float* losses,
                                 float* loss_accum, int32_t* step_counter, void* stream) {
    if (!flat) return hjbx_set_error(HJBX_EINVAL, "hjbx_mix_adam_f32: NULL buffer");
    AdamArgs a{};
    if (int rc = adam_args_from(adam, "hjbx_mix_adam_f32", a)) return rc;
    const MixArgs mx{reg_dev, (float)reg, (float)eps, losses, loss_accum, step_counter};
    const int grid = (int)((a.P + 255) / 256 < 512 ? (a.P + 255) / 256 : 512);
    hipLaunchKernelGGL(k_mix_adam, dim3(grid), dim3(256), 0, (hipStream_t)stream, flat, a, mx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_mix_adam_f32: %s", hipGetErrorString(e));
    return HJBX_OK;
}
