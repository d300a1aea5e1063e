// hjbx_mlp.hip -- ValueFunctionApproximator forward + input gradient (reference controller/vhjb.py:17-60
// and get_v_gradient :201-202) fused into one gfx950 kernel on the matrix cores: this file holds the two kernels and the float32-MFMA
// arithmetic described below; the same kernels instantiate the 16-bit split-operand arithmetics of hjbx_mlp_x3.hpp (bf16 x 3) and
// hjbx_mlp_h2.hpp (f16 x 2; both opt-in, HJBX_OPT_MLP_ARITHMETIC) through the AR template parameter.
//
//   e = wrap(x - xf); z = (e - mean)/std; h1 = act(z W1); h2 = act(h1 W2); y = h2 W3        (act = relu | tanh | sin)
//   V = |y|^2 + eps_s |e|^2
//   dV/dx = ((((2y) W3') . act'(h2)) W2' . act'(h1)) W1' / std + 2 eps_s e
//
// Design (CDNA4):
//  * v_mfma_f32_32x32x2_f32 (exact f32, 157 TFLOP/s dense peak) -- the network is float32 in the reference.
//  * Everything is computed TRANSPOSED (features x environments): a wave owns TL tiles of 32 environments
//    (the MFMA column index = lane & 31) and the accumulator registers of one layer ARE the B operands of
//    the next one, forward and backward, with no cross-lane movement and no LDS round trip: accumulator
//    register s of lane-half h holds feature perm(s) + 4h, and the weight (A) operand for k-step s is
//    simply fetched for that same feature.
//  * All three weight matrices live in LDS once per workgroup (106 KB, one copy serves W and W'): rows
//    padded to an ODD stride (129 / 65 floats) so that both the row walk of the forward pass and the
//    column walk of the backward pass hit 32 distinct banks per ds_read_b32 lane group.
//  * A VALU-category instruction issued between two MFMAs costs matrix-pipe time (~4 cycles each, tools/ubench/mfma_mix.hip;
//    LDS reads and s_waitcnt issue beside the MFMAs for free), so a chain is ONLY ds_read / s_waitcnt / MFMA: weight
//    operands are prefetched by inline-asm ds_reads two k-steps ahead and retired by counted s_waitcnt; the element-wise
//    work (activation, its derivative, |y|^2, 2y) runs in place on the accumulators in VALU-only passes between the
//    chains, where it overlaps the SIMD partner's MFMAs; activation derivatives are taken from the activations (nothing
//    extra is stored; layer 1 is recomputed for the last one); the last backward product (only n useful rows) runs on the
//    VALU.  The activation is a template parameter: ReLU (one integer max) or tanh (exp2 + rcp).
//  * Persistent grid, one workgroup per CU (2 waves per SIMD); waves pull tile groups from an LDS counter so SIMD
//    partners finish together.  In the rollout kernel the per-environment constants (system, task, limits) are staged
//    in LDS too: as kernel-argument SGPRs they spilled into v_readlane / v_writelane inside the chains.
// Per environment: 4(128 n + 128*128 + 128*64) flop; algorithmic HBM traffic 4(2n+1) bytes -> MFMA bound.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cstddef>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"
#include "hjbx_host.hpp"
#include "hjbx_mlp_core.hpp"
#include "hjbx_mlp_x3.hpp"
#include "hjbx_mlp_h2.hpp"

using namespace hjbx;

// launch shape: TL tiles of 32 environments per wave, WAVES waves per workgroup (one workgroup per CU).
//   (TL, WAVES) = (1, 8): two waves per SIMD, 256 VGPRs each.   (2, 4): one wave per SIMD, 512 VGPRs.
// This file is compiled once per variant (-DHJBX_MLP_ACT=0 relu, =1 tanh, =2 relu with the bf16x3-split arithmetic of hjbx_mlp_x3.hpp,
// =3 relu with the f16x2-split arithmetic of hjbx_mlp_h2.hpp, =4 sin: 30 kernel instantiations each, side by side); the relu object also
// carries the two C entry points, which validate and hand over to the object of the requested variant.
#ifndef HJBX_MLP_ACT
#error "compile hjbx_mlp.hip with -DHJBX_MLP_ACT=0 (relu + the C entry points), =1 (tanh), =2 (relu, bf16x3-split MFMA), =3 (relu, f16x2-split MFMA) and =4 (sin)"
#endif
static constexpr int kArith = HJBX_MLP_ACT == 2 ? 1 : HJBX_MLP_ACT == 3 ? 2 : 0;  // 0 = f32 MFMA, 1 = bf16x3, 2 = f16x2 (HJBX_OPT_MLP_ARITHMETIC)
static constexpr int kAct = kArith ? HJBX_ACT_RELU : HJBX_MLP_ACT == 4 ? HJBX_ACT_SIN : HJBX_MLP_ACT;
static_assert(kAct == HJBX_ACT_RELU || kAct == HJBX_ACT_TANH || kAct == HJBX_ACT_SIN, "fused kernels exist for relu, tanh and sin");
template <int N, int AR> using MlpLdsT = std::conditional_t<AR == 1, MlpLdsX3<N>, std::conditional_t<AR == 2, MlpLdsH2<N>, MlpLds<N>>>;
#define HJBX_MLP_CAT2(a, b) a##b
#define HJBX_MLP_CAT(a, b) HJBX_MLP_CAT2(a, b)
#define HJBX_MLP_SYM(name) HJBX_MLP_CAT(name, HJBX_MLP_ACT)
#define HJBX_HIDDEN __attribute__((visibility("hidden")))
// per-activation dispatchers (system kind -> kernel instantiation), one pair per object file
HJBX_HIDDEN int hjbx_mlp_value_grad_act0(const hjbx_system*, const hjbx_mlp*, const float*, float*, float*, int64_t, void*);
HJBX_HIDDEN int hjbx_mlp_value_grad_act1(const hjbx_system*, const hjbx_mlp*, const float*, float*, float*, int64_t, void*);
HJBX_HIDDEN int hjbx_mlp_value_grad_act2(const hjbx_system*, const hjbx_mlp*, const float*, float*, float*, int64_t, void*);
HJBX_HIDDEN int hjbx_mlp_value_grad_act3(const hjbx_system*, const hjbx_mlp*, const float*, float*, float*, int64_t, void*);
HJBX_HIDDEN int hjbx_mlp_value_grad_act4(const hjbx_system*, const hjbx_mlp*, const float*, float*, float*, int64_t, void*);
HJBX_HIDDEN int hjbx_mlp_rollout_act3(const hjbx_system*, const hjbx_task*, const hjbx_mlp*, int, int, int, int, const float*, float*, float*, float*,
                                      float*, float*, int32_t*, float*, const int32_t*, int64_t, void*, void*);
HJBX_HIDDEN int hjbx_mlp_rollout_act2(const hjbx_system*, const hjbx_task*, const hjbx_mlp*, int, int, int, int, const float*, float*, float*, float*,
                                      float*, float*, int32_t*, float*, const int32_t*, int64_t, void*, void*);
HJBX_HIDDEN int hjbx_mlp_rollout_act0(const hjbx_system*, const hjbx_task*, const hjbx_mlp*, int, int, int, int, const float*, float*, float*, float*,
                                      float*, float*, int32_t*, float*, const int32_t*, int64_t, void*, void*);
HJBX_HIDDEN int hjbx_mlp_rollout_act1(const hjbx_system*, const hjbx_task*, const hjbx_mlp*, int, int, int, int, const float*, float*, float*, float*,
                                      float*, float*, int32_t*, float*, const int32_t*, int64_t, void*, void*);
HJBX_HIDDEN int hjbx_mlp_rollout_act4(const hjbx_system*, const hjbx_task*, const hjbx_mlp*, int, int, int, int, const float*, float*, float*, float*,
                                      float*, float*, int32_t*, float*, const int32_t*, int64_t, void*, void*);

#ifndef HJBX_MLP_TL
#define HJBX_MLP_TL 1
#endif
#ifndef HJBX_MLP_WAVES
#define HJBX_MLP_WAVES 8
#endif

// ---- kernel 1: V and dV/dx for a batch of states (hjbx_value_grad_f32) ---------------------------------------
template <typename S, int TL, int WAVES, int ACT, int AR>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_value_grad_mfma(S sys, MlpP<S::N> p, const float* __restrict__ W1g,
                                                                         const float* __restrict__ W2g, const float* __restrict__ W3g,
                                                                         const float* __restrict__ x, float* __restrict__ Vout,
                                                                         float* __restrict__ gout, int64_t B, int64_t ngroups) {
    constexpr int N = S::N;
    static_assert(N % 2 == 0, "state dimension must be even (k-steps of 2)");
    static_assert(AR == 0 || TL == 1, "the split-operand chains hold one tile per wave");
    __shared__ __attribute__((aligned(256))) MlpLdsT<N, AR> L;
    const int tid = threadIdx.x;
    if (tid == 0) L.next = WAVES;  // groups 0..WAVES-1 of the range are taken statically
#ifdef HJBX_DIAG_CLOCK
    const unsigned long long tentry = __builtin_amdgcn_s_memrealtime();
#endif
    if constexpr (AR == 1) mlp_fill_lds_x3<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    else if constexpr (AR == 2) mlp_fill_lds_h2<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    else mlp_fill_lds<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const auto c = [&] {
        if constexpr (AR == 1) return mlp_ctx_x3<N>(L, lane);
        else if constexpr (AR == 2) return mlp_ctx_h2<N>(L, lane);
        else return mlp_ctx<N>(L, lane);
    }();
    const int i = c.i, h = c.h;

    // Work distribution: the workgroup owns a contiguous range of tile groups and its waves pull the next one
    // from an LDS counter.  (With a static stride the older wave of each SIMD pair wins the matrix-pipe
    // arbitration, finishes its share ~25 % early and leaves its partner running alone.)
    const int64_t groups_per_wg = (ngroups + gridDim.x - 1) / gridDim.x;
    const int64_t g_begin = (int64_t)blockIdx.x * groups_per_wg;
    const int64_t g_end = (g_begin + groups_per_wg < ngroups) ? g_begin + groups_per_wg : ngroups;

    // one row of x per lane and tile; both lane halves read the same row (the second read hits the same lines)
    auto load_rows = [&](int64_t grp, float (&dst)[TL][N]) {
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            const int64_t en = (grp * TL + t) * 32 + i;
            if (grp < g_end && en < B) {
                load_row<N>(x, en, dst[t]);
            } else {
#pragma unroll
                for (int k = 0; k < N; ++k) dst[t][k] = p.xf[k];
            }
        }
    };

    int64_t grp = g_begin + wave;
#ifdef HJBX_DIAG_CLOCK
    // DIAGNOSTIC BUILD ONLY (tools/diag_clock.py): shader-clock and 100 MHz wall stamps around the tile loop
    const unsigned long long t0c = __builtin_amdgcn_s_memtime(), t0r = __builtin_amdgcn_s_memrealtime();
#endif
    float xs[TL][N], xn[TL][N];
    load_rows(grp, xs);
    for (; grp < g_end;) {
        // the weights are loop invariant: without this barrier LICM hoists LDS reads out of the tile loop
        asm volatile("" ::: "memory");
        // claim the next group now and fetch its rows: the HBM latency hides behind this group's MFMAs
        int nxt = 0;
        if (lane == 0) nxt = atomicAdd(&L.next, 1);
        const int64_t grp_next = g_begin + __builtin_amdgcn_readfirstlane(nxt);
        load_rows(grp_next, xn);

        float V[TL], g[TL][N];
        if constexpr (AR == 1) mlp_value_grad_x3<S>(sys, p, c, xs, gout != nullptr, V, g);
        else if constexpr (AR == 2) mlp_value_grad_h2<S>(sys, p, c, xs, gout != nullptr, V, g);
        else mlp_value_grad<S, TL, ACT>(sys, p, c, xs, gout != nullptr, V, g);
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            const int64_t env = (grp * TL + t) * 32 + i;
            if (env < B && h == 0) {
#ifndef HJBX_DIAG_CLOCK
                if (Vout) Vout[env] = V[t];
#endif
                if (gout) store_row<N>(gout, env, g[t]);
            }
#pragma unroll
            for (int k = 0; k < N; ++k) xs[t][k] = xn[t][k];
        }
        grp = grp_next;
    }
#ifdef HJBX_DIAG_CLOCK
    {
        const unsigned long long t1c = __builtin_amdgcn_s_memtime(), t1r = __builtin_amdgcn_s_memrealtime();
        // the caller of the diagnostic build passes a scratch "V" buffer of >= 4*gridDim.x*WAVES floats and gradV != NULL
        if (Vout && gout && lane == 0) {
            float* d = Vout + 4 * ((int64_t)blockIdx.x * WAVES + wave);
            d[0] = (float)(t1c - t0c); d[1] = (float)(t1r - t0r);
            d[2] = (float)(t0r - tentry); d[3] = (float)(tentry & 0xFFFFFFull);
        }
    }
#endif
}

// ---- kernel 2: the whole VHJB closed loop for n_steps steps in one launch (hjbx_vhjb_rollout_f32) ----------------
// Per environment tile: state in registers; per step: value gradient on the matrix cores (above), then exactly the
// per-environment code of hjbx_vhjb_step (vhjb_step_env: bounds / termination, control from gradV, cost, HJB
// residual, Euler or RK4 step), then the time-major log slabs.  Tiles are independent, so a wave runs all steps of
// one tile group before pulling the next; the weights are staged into LDS once per launch instead of once per step.
template <int N, int M> struct RolloutOut {
    float* traj;   // (n_steps+1, B, N) or NULL: slab k = state at step t_first + k
    float* u_log;  // (n_steps, B, M) or NULL
    float* cost;   // (n_steps, B)
    float* done;   // (n_steps, B)
    float* resid;  // (n_steps, B) or NULL
    int32_t* done_step;  // (B) in/out
    float* x_out;  // (B, N) or NULL
};

// Work distribution of the persistent rollout kernel.  The caller's workspace (hjbx_rollout_workspace_bytes(), zero-filled once,
// left zeroed by every launch) holds, as 32-bit words:
static constexpr int kWsQueue = 0;                      // schedule 1: head of the device-wide tile queue
static constexpr int kWsStarted = 32;                   // workgroups that have started
static constexpr int kWsExited = 64;                    // workgroups whose waves have all finished (the last one zeroes the workspace)
static constexpr int kWsFlags = 96;                     // [kMaxGrid] 0 = not started, 1 = running its own range, 2 = range open to every wave
static constexpr int kMaxGrid = 1024;
static constexpr int kWsOpen = kWsFlags + kMaxGrid;     // [kMaxGrid] next unclaimed pick of an open range
static constexpr int kWsWords = kWsOpen + kMaxGrid;
#define HJBX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

template <int INTEG, typename S, int WAVES, int ACT, int AR>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_vhjb_rollout_mfma(S sys_k, MlpP<S::N> p_k, TaskP<float, S::N, S::M> tk_k,
                                                                           Limits<float, S::M> lim_k, const float* __restrict__ W1g,
                                                                           const float* __restrict__ W2g, const float* __restrict__ W3g,
                                                                           int t_first, int n_steps, int T_max, const float* x /* may alias traj slab 0 */,
                                                                           const int32_t* __restrict__ order, RolloutOut<S::N, S::M> o, int64_t B,
                                                                           int64_t ngroups, unsigned* ws, int sched) {
    constexpr int N = S::N, M = S::M;
    static_assert(N % 2 == 0, "state dimension must be even (k-steps of 2)");
    __shared__ __attribute__((aligned(256))) MlpLdsT<N, AR> L;
    // System, task, limits and normalisation constants are staged in LDS: as kernel arguments they are ~100-250 wave-uniform
    // scalars that do not fit the SGPR file next to the address arithmetic, and hipcc spilled them to VGPR lanes
    // (hundreds of v_readlane / v_writelane per step, some inside the MFMA chains).  LDS broadcast reads cost no SGPRs.
    __shared__ __attribute__((aligned(16))) unsigned char sys_raw[sizeof(S)];  // S has default member initialisers: no __shared__ S
    S& sys_s = *reinterpret_cast<S*>(sys_raw);
    __shared__ MlpP<N> p_s;
    __shared__ TaskP<float, N, M> tk_s;
    __shared__ Limits<float, M> lim_s;
    const int tid = threadIdx.x;
    // One queue per SIMD (the waves w and w + 4 of a workgroup share SIMD w & 3): SIMD q works through the picks q, q + 4, q + 8, ...
    // of this workgroup.  A tile here is a whole n_steps-step rollout, so with a single queue per workgroup the four SIMDs of a
    // CU could end up with 9 / 7 tiles instead of 8 / 8 at B = 2^18 and the CU waited for the unlucky one (+-12 % from build to build).
    __shared__ int q_next[4];
    __shared__ int waves_done;
    __shared__ unsigned my_flag;
    if (tid < 4) q_next[tid] = WAVES / 4;
    if (tid == 0) {
        waves_done = 0;
        // announce this workgroup: 0 -> 1; a 2 coming back means the others have already opened (and taken) its range
        unsigned seen = 0u;
        __hip_atomic_compare_exchange_strong(ws + kWsFlags + blockIdx.x, &seen, 1u, __ATOMIC_RELAXED, HJBX_RLX_AGENT);
        my_flag = seen;
        __hip_atomic_fetch_add(ws + kWsStarted, 1u, HJBX_RLX_AGENT);
        sys_s = sys_k; p_s = p_k; tk_s = tk_k; lim_s = lim_k;
    }
    if constexpr (AR == 1) mlp_fill_lds_x3<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    else if constexpr (AR == 2) mlp_fill_lds_h2<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    else mlp_fill_lds<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    __syncthreads();
    const S& sys = sys_s;
    const MlpP<N>& p = p_s;
    const TaskP<float, N, M>& tk = tk_s;
    const Limits<float, M>& lim = lim_s;
    // wave index and this workgroup's flag are wave uniform IN FACT; through readfirstlane they are uniform TO THE COMPILER too, so the
    // tile group a wave works on (`grp`) lives in SGPRs and `if (grp < 0) grp = next_group()` is a scalar branch.  With `grp` in VGPRs
    // that branch was an EXEC-predicated region, and a register-allocator spill store placed inside it ran with EXEC = 0 for waves
    // that already had a group: the reload after the join returned garbage (round 2: n = 10 / m = 2 kernels with 14 spilled VGPRs
    // produced wrong trajectories).  A CPU test also keeps every instantiation at zero scratch.
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned my_flag_u = (unsigned)__builtin_amdgcn_readfirstlane((int)my_flag);
    const auto c = [&] {
        if constexpr (AR == 1) return mlp_ctx_x3<N>(L, lane);
        else if constexpr (AR == 2) return mlp_ctx_h2<N>(L, lane);
        else return mlp_ctx<N>(L, lane);
    }();
    const int i = c.i, h = c.h;
    const int simd = wave & 3;
    const int G = (int)gridDim.x;
    // schedule 0 (default): every workgroup owns an equal range of tile groups -- contiguous in natural order (measured 12 % faster
    // at B = 2^18 than dealing single groups round-robin); with `order` the live environments come first, so the groups are dealt
    // round-robin (group = workgroup + pick * gridDim) to spread the live tiles over all CUs -- and its SIMDs work through it via the
    // LDS queues above.  A workgroup that finds no free CU when the launch starts (one workgroup fills a CU) would only run after
    // another one has finished, doubling the launch: so a wave that has finished its own share looks for workgroups that have NOT
    // STARTED, opens their ranges (flag 0 -> 2) and every finishing wave takes tiles from the open ranges, one returning atomic per
    // tile; the late workgroup then finds its range taken and only helps.  With every workgroup resident this costs one atomic load.
    // schedule 1: a device-wide queue, one returning atomic per tile (the first tile of every wave is static).
    const int64_t picks_per_wg = (ngroups + G - 1) / G;
    auto group_of = [&](int w, int64_t pick) -> int64_t {
        if (pick >= picks_per_wg) return -1;
        const int64_t g = order ? (int64_t)w + pick * G : (int64_t)w * picks_per_wg + pick;
        return g < ngroups ? g : -1;
    };
    int victim = -1;                                    // schedule 0: < 0 = own range, else the workgroup whose open range is being drained
    auto next_group = [&]() -> int64_t {
        if (sched == 1) {
            unsigned t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(ws + kWsQueue, 1u, HJBX_RLX_AGENT);
            const int64_t g = (int64_t)G * WAVES + (unsigned)__builtin_amdgcn_readfirstlane((int)t);
            return g < ngroups ? g : -1;
        }
        if (victim < 0) {
            int nxt = 0;
            if (lane == 0) nxt = atomicAdd(&q_next[simd], 1);
            const int64_t g = my_flag_u == 0u ? group_of(blockIdx.x, simd + 4 * (int64_t)__builtin_amdgcn_readfirstlane(nxt)) : -1;
            if (g >= 0) return g;
            victim = 0;
            unsigned started = 0;
            if (lane == 0) started = __hip_atomic_load(ws + kWsStarted, HJBX_RLX_AGENT);
            if (__builtin_amdgcn_readfirstlane((int)started) >= G) victim = G;     // every workgroup is resident: nothing to take over
        }
        while (victim < G) {
            // flags of workgroups victim .. victim + 63; an unstarted one is opened here and now
            const int w = victim + lane;
            unsigned f = 1u;
            if (w < G) {
                f = __hip_atomic_load(ws + kWsFlags + w, HJBX_RLX_AGENT);
                if (f == 0u && __hip_atomic_compare_exchange_strong(ws + kWsFlags + w, &f, 2u, __ATOMIC_RELAXED, HJBX_RLX_AGENT)) f = 2u;
            }
            unsigned long long open = __builtin_amdgcn_ballot_w64(f == 2u);
            while (open) {
                const int b = __builtin_ctzll(open);
                unsigned t = 0;
                if (lane == 0) t = __hip_atomic_fetch_add(ws + kWsOpen + victim + b, 1u, HJBX_RLX_AGENT);
                const int64_t g = group_of(victim + b, (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)t));
                if (g >= 0) { victim += b; return g; }      // (the scan resumes at this workgroup next time)
                open &= open - 1;
            }
            victim = (victim + 64 < G) ? victim + 64 : G;
        }
        return -1;
    };
    int64_t grp;
    if (sched == 1) grp = (int64_t)blockIdx.x * WAVES + wave < ngroups ? (int64_t)blockIdx.x * WAVES + wave : -1;
    else grp = my_flag_u == 0u ? group_of(blockIdx.x, wave) : -1;   // wave = simd + 4 * (wave >> 2): the first WAVES / 4 picks of each SIMD are static
    if (grp < 0) grp = next_group();
    while (grp >= 0) {
        const int64_t slot = grp * 32 + i;
        const bool valid = slot < B;
        int64_t env = valid ? (order ? (int64_t)order[slot] : slot) : 0;
        env = env < 0 ? 0 : (env >= B ? B - 1 : env);  // a corrupt `order` entry must not become an out-of-bounds access
        const bool writer = valid && h == 0;  // both lane halves carry the same environment; half 0 stores
        float xs[1][N];
        int32_t ds = 0;                        // padding lanes are "done": they hold xf and emit nothing
        if (valid) {
            load_row<N>(x, env, xs[0]);
            ds = o.done_step[env];
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) xs[0][k] = p.xf[k];
        }
        if (o.traj && writer) store_row<N>(o.traj, env, xs[0]);
        for (int k = 0; k < n_steps; ++k) {
            asm volatile("" ::: "memory");
            float xo[N], u[M], cst, dn, res;
            if (__builtin_amdgcn_ballot_w64(ds < 0) == 0) {
                // every environment of this tile has finished: vhjb_step_env would emit zeros and hold the state whatever the
                // value gradient is, so the network is skipped (a finished tile costs its log writes only).  Kept as a separate
                // arm: sharing vhjb_step_env behind a conditional network call made hipcc spill the prefetched weights.
#pragma unroll
                for (int q = 0; q < N; ++q) xo[q] = xs[0][q];
#pragma unroll
                for (int j = 0; j < M; ++j) u[j] = 0.0f;
                cst = dn = res = 0.0f;
            } else {
                float V[1], g[1][N];
                if constexpr (AR == 1) mlp_value_grad_x3<S>(sys, p, c, xs, true, V, g);
                else if constexpr (AR == 2) mlp_value_grad_h2<S>(sys, p, c, xs, true, V, g);
                else mlp_value_grad<S, 1, ACT>(sys, p, c, xs, true, V, g);
                vhjb_step_env<INTEG>(sys, tk, lim, t_first + k, T_max, o.resid != nullptr, xs[0], g[0], ds, xo, u, cst, dn, res);
            }
            if (writer) {
                const int64_t row = (int64_t)k * B + env;
                o.cost[row] = cst;
                o.done[row] = dn;
                if (o.resid) o.resid[row] = res;
                if (o.u_log) store_row<M>(o.u_log + (int64_t)k * B * M, env, u);
                if (o.traj) store_row<N>(o.traj + (int64_t)(k + 1) * B * N, env, xo);
            }
#pragma unroll
            for (int q = 0; q < N; ++q) xs[0][q] = xo[q];
        }
        if (writer) {
            o.done_step[env] = ds;
            if (o.x_out) store_row<N>(o.x_out, env, xs[0]);
        }
        grp = next_group();
    }
    // the last wave of the last workgroup leaves the workspace zeroed for the next launch
    int lastw = 0;
    if (lane == 0 && atomicAdd(&waves_done, 1) == WAVES - 1) lastw = __hip_atomic_fetch_add(ws + kWsExited, 1u, HJBX_RLX_AGENT) == (unsigned)(G - 1);
    if (__builtin_amdgcn_readfirstlane(lastw)) {
        for (int w = lane; w < kWsWords; w += 64)
            if (w < kWsFlags || (w - kWsFlags) % kMaxGrid < G) __hip_atomic_store(ws + w, 0u, HJBX_RLX_AGENT);
    }
}

#if HJBX_MLP_ACT == 0
extern "C" size_t hjbx_rollout_workspace_bytes(void) { return (size_t)kWsWords * sizeof(unsigned); }
static int check_activation(const hjbx_mlp* mlp, const char* who) {
    if (mlp->activation == HJBX_ACT_RELU || mlp->activation == HJBX_ACT_TANH || mlp->activation == HJBX_ACT_SIN) return HJBX_OK;
    return hjbx_set_error(HJBX_EINVAL, "%s: unknown activation %d", who, mlp->activation);
}
#endif

template <typename S> static int launch_value_grad(S sys, const hjbx_mlp* mlp, const float* x, float* V, float* g, int64_t B, void* st) {
    constexpr int N = S::N;
    constexpr int TL = HJBX_MLP_TL, WAVES = HJBX_MLP_WAVES;
    MlpP<N> p;
    for (int k = 0; k < N; ++k) { p.mean[k] = (float)mlp->mean[k]; p.istd[k] = (float)(1.0 / mlp->std[k]); p.xf[k] = (float)mlp->xf[k]; }
    p.eps_s = (float)mlp->eps_scalar;
    const int64_t ngroups = (B + 32 * TL - 1) / (32 * TL);
    const int n_cu = hjbx_device_cus();
    if (n_cu <= 0) return hjbx_set_error(HJBX_ENODEVICE, "hjbx_value_grad_f32: no HIP device");
    // one resident workgroup per CU (106 KB of LDS each); small batches are spread one tile group per CU
    // rather than packed eight to a workgroup, so up to n_cu matrix pipes work on them
    int64_t grid = ngroups < n_cu ? ngroups : n_cu;
    hipLaunchKernelGGL((k_value_grad_mfma<S, TL, WAVES, kAct, kArith>), dim3((unsigned)grid), dim3(WAVES * 64), 0, (hipStream_t)st, sys, p,
                       (const float*)mlp->W1, (const float*)mlp->W2, (const float*)mlp->W3, x, V, g, B, ngroups);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_value_grad_f32: %s", hipGetErrorString(e));
    return HJBX_OK;
}

// system kind -> instantiation of this object's activation (arguments already validated by the C entry point)
int HJBX_MLP_SYM(hjbx_mlp_value_grad_act)(const hjbx_system* sys, const hjbx_mlp* mlp, const float* x, float* V, float* g, int64_t B, void* stream) {
#ifdef HJBX_MLP_DEV  // development builds: cartpole only (30 instantiations take a minute per variant)
    if (sys->kind == HJBX_SYS_CARTPOLE) { Cartpole<float> c{}; return launch_value_grad(c, mlp, x, V, g, B, stream); }
#ifdef HJBX_MLP_DEV_QUAD2D
    if (sys->kind == HJBX_SYS_QUAD2D) { Quad2D<float> q{}; return launch_value_grad(q, mlp, x, V, g, B, stream); }
#endif
    return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_grad_f32: development build (cartpole only)");
#else
    switch (sys->kind) {
    case HJBX_SYS_LINEAR:
        if (sys->n == 2) {
            Linear<float, 2, 1> l{};  // wrap is the identity; A, B unused here
            return launch_value_grad(l, mlp, x, V, g, B, stream);
        }
        if (sys->n == 4) { Linear<float, 4, 1> l{}; return launch_value_grad(l, mlp, x, V, g, B, stream); }
        if (sys->n == 6) { Linear<float, 6, 2> l{}; return launch_value_grad(l, mlp, x, V, g, B, stream); }
        break;
    case HJBX_SYS_CARTPOLE: { Cartpole<float> c{}; return launch_value_grad(c, mlp, x, V, g, B, stream); }
    case HJBX_SYS_ACROBOT: { Acrobot<float> a{}; return launch_value_grad(a, mlp, x, V, g, B, stream); }
    case HJBX_SYS_QUAD2D: { Quad2D<float> q{}; return launch_value_grad(q, mlp, x, V, g, B, stream); }
    case HJBX_SYS_NEARHOVER: { NearHover<float> q{}; return launch_value_grad(q, mlp, x, V, g, B, stream); }
    }
    return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_grad_f32: no kernel for system kind %d with n=%d", sys->kind, sys->n);
#endif
}


#if HJBX_MLP_ACT == 0
extern "C" int hjbx_value_grad_f32(const hjbx_system* sys, const hjbx_mlp* mlp, const float* x, float* V, float* g, int64_t B,
                                   void* stream) {
    if (!sys || !mlp) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: NULL system or mlp descriptor");
    if (B < 0) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: negative batch size");
    if (B == 0 || (!V && !g)) return HJBX_OK;
    if (!x || !mlp->W1 || !mlp->W2 || !mlp->W3) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: NULL x or weight pointer");
    if (mlp->h1 != kH1 || mlp->h2 != kH2 || mlp->h3 != kH3)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_grad_f32: features must be [128,128,64], got [%d,%d,%d]", mlp->h1, mlp->h2,
                              mlp->h3);
    if (int rc = check_activation(mlp, "hjbx_value_grad_f32")) return rc;
    const size_t row = (size_t)sys->n * sizeof(float);
    const uintptr_t am = (row % 16 == 0) ? 15u : 7u;
    if ((reinterpret_cast<uintptr_t>(x) & am) || (g && (reinterpret_cast<uintptr_t>(g) & am)))
        return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: x / gradV must be aligned to their row vector width");
    for (int k = 0; k < sys->n; ++k)
        if (!(mlp->std[k] != 0.0)) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: normalization_std[%d] is zero", k);
    if (mlp->activation == HJBX_ACT_TANH) return hjbx_mlp_value_grad_act1(sys, mlp, x, V, g, B, stream);
    if (mlp->activation == HJBX_ACT_SIN) return hjbx_mlp_value_grad_act4(sys, mlp, x, V, g, B, stream);
    const int arith = hjbx_option_value(HJBX_OPT_MLP_ARITHMETIC);
    return arith == 1 ? hjbx_mlp_value_grad_act2(sys, mlp, x, V, g, B, stream)
         : arith == 2 ? hjbx_mlp_value_grad_act3(sys, mlp, x, V, g, B, stream)
                      : hjbx_mlp_value_grad_act0(sys, mlp, x, V, g, B, stream);
}
#endif

template <typename S>
static int launch_vhjb_rollout(const hjbx_system* sysh, S sys, const hjbx_task* task, const hjbx_mlp* mlp, int integrator, int t_first,
                               int n_steps, int T_max, const float* x, float* traj, float* u_log, float* cost, float* done, float* resid,
                               int32_t* done_step, float* x_out, const int32_t* order, int64_t B, void* workspace, void* st) {
    constexpr int N = S::N, M = S::M;
    constexpr int WAVES = HJBX_MLP_WAVES;
    MlpP<N> p;
    for (int k = 0; k < N; ++k) { p.mean[k] = (float)mlp->mean[k]; p.istd[k] = (float)(1.0 / mlp->std[k]); p.xf[k] = (float)mlp->xf[k]; }
    p.eps_s = (float)mlp->eps_scalar;
    const auto tk = make_task<float, N, M>(task);
    const auto lim = make_limits<float, M>(sysh);
    RolloutOut<N, M> o{traj, u_log, cost, done, resid, done_step, x_out};
    const int64_t ngroups = (B + 31) / 32;
    const int n_cu = hjbx_device_cus();
    if (n_cu <= 0) return hjbx_set_error(HJBX_ENODEVICE, "hjbx_vhjb_rollout_f32: no HIP device");
    int64_t grid = ngroups < n_cu ? ngroups : n_cu;  // as in launch_value_grad
    const int sched = hjbx_option_value(HJBX_OPT_ROLLOUT_SCHEDULE);
    grid += hjbx_option_value(HJBX_OPT_ROLLOUT_EXTRA_WORKGROUPS);   // test hook: workgroups that cannot be resident before others finish
    if (grid > kMaxGrid) grid = kMaxGrid;
    const float *W1 = (const float*)mlp->W1, *W2 = (const float*)mlp->W2, *W3 = (const float*)mlp->W3;
    auto launch = [&](auto integ) {
        hipLaunchKernelGGL((k_vhjb_rollout_mfma<decltype(integ)::value, S, WAVES, kAct, kArith>), dim3((unsigned)grid), dim3(WAVES * 64), 0,
                           (hipStream_t)st, sys, p, tk, lim, W1, W2, W3, t_first, n_steps, T_max, x, order, o, B, ngroups, (unsigned*)workspace, sched);
    };
    if (integrator == HJBX_EULER) launch(std::integral_constant<int, 0>{});
    else if (integrator == HJBX_RK4) launch(std::integral_constant<int, 1>{});
    else if constexpr (S::kHasZoh) launch(std::integral_constant<int, 2>{});
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_vhjb_rollout_f32: %s", hipGetErrorString(e));
    return HJBX_OK;
}

int HJBX_MLP_SYM(hjbx_mlp_rollout_act)(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int integrator, int t_first, int n_steps,
                                       int T_max, const float* x, float* traj, float* u_log, float* cost, float* done, float* resid,
                                       int32_t* done_step, float* x_out, const int32_t* env_order, int64_t B, void* workspace, void* stream) {
    int rc = HJBX_EUNSUPPORTED;
#ifdef HJBX_MLP_DEV
    if (sys->kind == HJBX_SYS_CARTPOLE && integrator == HJBX_EULER) {
        Cartpole<float> S{(float)sys->p[0], (float)sys->p[1], (float)sys->p[2], (float)sys->p[3]};
        constexpr int WAVES = HJBX_MLP_WAVES;
        MlpP<4> p;
        for (int k = 0; k < 4; ++k) { p.mean[k] = (float)mlp->mean[k]; p.istd[k] = (float)(1.0 / mlp->std[k]); p.xf[k] = (float)mlp->xf[k]; }
        p.eps_s = (float)mlp->eps_scalar;
        RolloutOut<4, 1> o{traj, u_log, cost, done, resid, done_step, x_out};
        const int64_t ngroups = (B + 31) / 32;
        int64_t grid = ngroups < 256 ? ngroups : 256;
        hipLaunchKernelGGL((k_vhjb_rollout_mfma<0, Cartpole<float>, WAVES, kAct, kArith>), dim3((unsigned)grid), dim3(WAVES * 64), 0, (hipStream_t)stream, S, p,
                           make_task<float, 4, 1>(task), make_limits<float, 1>(sys), (const float*)mlp->W1, (const float*)mlp->W2, (const float*)mlp->W3, t_first,
                           n_steps, T_max, x, env_order, o, B, ngroups, (unsigned*)workspace, hjbx_option_value(HJBX_OPT_ROLLOUT_SCHEDULE));
        return hipGetLastError() == hipSuccess ? HJBX_OK : hjbx_set_error(HJBX_EHIP, "launch");
    }
    return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_vhjb_rollout_f32: development build (cartpole, Euler only)");
#else
    const bool ok = with_system<float>(sys, [&](auto S) {
        using SS = decltype(S);
        if constexpr (SS::N % 2 == 0)
            rc = launch_vhjb_rollout<SS>(sys, S, task, mlp, integrator, t_first, n_steps, T_max, x, traj, u_log, cost, done, resid, done_step,
                                         x_out, env_order, B, workspace, stream);
    });
    if (!ok || rc == HJBX_EUNSUPPORTED)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_vhjb_rollout_f32: no kernel for system kind %d with n=%d m=%d", sys->kind, sys->n, sys->m);
    return rc;
#endif
}

#if HJBX_MLP_ACT == 0
extern "C" int hjbx_vhjb_rollout_f32(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int integrator, int t_first,
                                     int n_steps, int T_max, const float* x, float* traj, float* u_log, float* cost, float* done,
                                     float* resid, int32_t* done_step, float* x_out, const int32_t* env_order, int64_t B, void* workspace,
                                     void* stream) {
    if (!sys || !task || !mlp) return hjbx_set_error(HJBX_EINVAL, "hjbx_vhjb_rollout_f32: NULL system, task or mlp descriptor");
    if (int rc = check_task(task)) return rc;
    if (B < 0 || n_steps < 0 || t_first < 0 || T_max < 0) return hjbx_set_error(HJBX_EINVAL, "hjbx_vhjb_rollout_f32: negative size or step index");
    if (int rc = check_integrator(sys, integrator, "hjbx_vhjb_rollout_f32")) return rc;
    if (B == 0) return HJBX_OK;
    if (!x || !cost || !done || !done_step || !mlp->W1 || !mlp->W2 || !mlp->W3)
        return hjbx_set_error(HJBX_EINVAL, "hjbx_vhjb_rollout_f32: x, cost, done, done_step and the weights must be non-NULL");
    if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15u))
        return hjbx_set_error(HJBX_EINVAL, "hjbx_vhjb_rollout_f32: workspace must be a 16-byte aligned device buffer of hjbx_rollout_workspace_bytes() zero-filled bytes");
    if (mlp->h1 != kH1 || mlp->h2 != kH2 || mlp->h3 != kH3)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_vhjb_rollout_f32: features must be [128,128,64], got [%d,%d,%d]", mlp->h1, mlp->h2,
                              mlp->h3);
    if (int rc = check_activation(mlp, "hjbx_vhjb_rollout_f32")) return rc;
    const size_t row = (size_t)sys->n * sizeof(float);
    const uintptr_t am = (row % 16 == 0) ? 15u : 7u;
    const size_t urow = (size_t)sys->m * sizeof(float);
    const uintptr_t um = (urow % 16 == 0) ? 15u : (urow % 8 == 0) ? 7u : 3u;
    if ((reinterpret_cast<uintptr_t>(x) & am) || (traj && (reinterpret_cast<uintptr_t>(traj) & am)) ||
        (x_out && (reinterpret_cast<uintptr_t>(x_out) & am)) || (u_log && (reinterpret_cast<uintptr_t>(u_log) & um)))
        return hjbx_set_error(HJBX_EINVAL, "hjbx_vhjb_rollout_f32: x / traj / x_out / u_log must be aligned to their row vector width");
    for (int k = 0; k < sys->n; ++k)
        if (!(mlp->std[k] != 0.0)) return hjbx_set_error(HJBX_EINVAL, "hjbx_vhjb_rollout_f32: normalization_std[%d] is zero", k);
    if (mlp->activation == HJBX_ACT_TANH)
        return hjbx_mlp_rollout_act1(sys, task, mlp, integrator, t_first, n_steps, T_max, x, traj, u_log, cost, done, resid, done_step, x_out, env_order, B, workspace, stream);
    if (mlp->activation == HJBX_ACT_SIN)
        return hjbx_mlp_rollout_act4(sys, task, mlp, integrator, t_first, n_steps, T_max, x, traj, u_log, cost, done, resid, done_step, x_out, env_order, B, workspace, stream);
    const int arith = hjbx_option_value(HJBX_OPT_MLP_ARITHMETIC);
    return (arith == 1 ? hjbx_mlp_rollout_act2 : arith == 2 ? hjbx_mlp_rollout_act3 : hjbx_mlp_rollout_act0)(
        sys, task, mlp, integrator, t_first, n_steps, T_max, x, traj, u_log, cost, done, resid, done_step, x_out, env_order, B, workspace, stream);
}
#endif
