// hjbx_stream_kernels.hpp -- the HBM-bound per-environment kernels of libhjbx.so as DEVICE FUNCTION TEMPLATES (`k_*_body`), one lane per
// environment.  Two users:
//   * hjbx_kernels.hip wraps each body in a `__global__` template and instantiates it for the five built-in systems (hipcc, build time);
//   * hjbx_user.hip hands this very text to hiprtc together with a USER-DEFINED system struct (hjbx_system_create_from_source: the open
//     plugin surface of the reference's dynamics_basic.py:64-94) and wraps the bodies in `extern "C"` kernels for that struct (run time).
// So a user system runs exactly the code the built-in systems run.  No host code in this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hjbx_systems.hpp"

namespace hjbx {

// ----------------------------------------------------------------------------------------------
// row-vector global memory access: a (B, N) row-major row is moved with the widest naturally
// aligned vector the row size allows (16 B for n=4 f32: one global_load_dwordx4 per lane).
// ----------------------------------------------------------------------------------------------
template <int BYTES> struct VecOf;
template <> struct VecOf<16> { using type = uint4; };
template <> struct VecOf<8> { using type = uint2; };
template <> struct VecOf<4> { using type = uint32_t; };

template <typename T, int N> struct RowIO {
    static constexpr int BYTES = N * (int)sizeof(T);
    static constexpr int W = (BYTES % 16 == 0) ? 16 : (BYTES % 8 == 0) ? 8 : 4;
    static constexpr int CNT = BYTES / W;
    using V = typename VecOf<W>::type;
    static HJBX_DEV void load(const T* base, int64_t row, T* out) {
        const V* p = reinterpret_cast<const V*>(base + row * N);
        union { V v[CNT]; T t[N]; } u;
#pragma unroll
        for (int k = 0; k < CNT; ++k) u.v[k] = p[k];
#pragma unroll
        for (int i = 0; i < N; ++i) out[i] = u.t[i];
    }
    static HJBX_DEV void store(T* base, int64_t row, const T* in) {
        V* p = reinterpret_cast<V*>(base + row * N);
        union { V v[CNT]; T t[N]; } u;
#pragma unroll
        for (int i = 0; i < N; ++i) u.t[i] = in[i];
#pragma unroll
        for (int k = 0; k < CNT; ++k) p[k] = u.v[k];
    }
};

static constexpr int kBlock = 256;        // 4 waves per workgroup
static constexpr uint32_t kRolloutTerminate = 1u, kRolloutStopAtTarget = 2u;   // HJBX_ROLLOUT_* of include/hjbx.h (this header must compile without it)
#ifndef HJBX_REDUCE_BLOCKS
#define HJBX_REDUCE_BLOCKS 1024
#endif
static constexpr int kReduceBlocks = HJBX_REDUCE_BLOCKS;  // grid cap of the reducing kernels (4 per CU).  Measured again in round 3 with the parallel final
                                                          // reduction (hjb_residual, B = 2^20, cartpole / near-hover): 1024: 16.8 / 33.4 us, 2048: 17.5 / 35.2, 4096: 23.5 / 38.3

// ----------------------------------------------------------------------------------------------
// pointwise kernels
// ----------------------------------------------------------------------------------------------
template <typename S, typename T>
HJBX_DEV void k_affine_body(S sys, const T* __restrict__ x, T* __restrict__ f1,
                                                   T* __restrict__ f2, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N], a[S::N], b[S::N * S::M];
    RowIO<T, S::N>::load(x, i, xs);
    sys.affine(xs, a, b);
    RowIO<T, S::N>::store(f1, i, a);
    RowIO<T, S::N * S::M>::store(f2, i, b);
}

template <typename S, typename T>
HJBX_DEV void k_wrap_body(S sys, const T* x, T* out, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N];
    RowIO<T, S::N>::load(x, i, xs);
    sys.wrap(xs);
    RowIO<T, S::N>::store(out, i, xs);
}

template <typename S, typename T>
HJBX_DEV void k_xdot_body(S sys, const T* __restrict__ x, const T* __restrict__ u,
                                                 T* __restrict__ xd, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N], us[S::M], d[S::N];
    RowIO<T, S::N>::load(x, i, xs);
    RowIO<T, S::M>::load(u, i, us);
    // f1 + f2 @ u, evaluated like the reference (dynamics_basic.py:101-103)
    T f1[S::N], f2[S::N * S::M];
    sys.affine(xs, f1, f2);
#pragma unroll
    for (int r = 0; r < S::N; ++r) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < S::M; ++j) acc += f2[r * S::M + j] * us[j];
        d[r] = f1[r] + acc;
    }
    RowIO<T, S::N>::store(xd, i, d);
}

// R rows per thread, all loads issued before the first use: a 36 MB kernel at 6 TB/s lasts 6 us, and with one row per thread the
// 16 workgroups a CU receives run as two resident rounds of (HBM latency + compute + store) -- latency bound, 4.1 TB/s measured
// with buffers that miss the Infinity Cache (profiles/r02_kernel_bench.json).  Row r of a thread is block_base + r kBlock + tid,
// so every load instruction of a wave stays one coalesced segment.
template <int INTEG, int R, typename S, typename T>
HJBX_DEV void k_simulate_body(S sys, Limits<T, S::M> lim, const T* x, const T* __restrict__ u,
                                                     T* xn, int64_t B) {
    const int64_t base = (int64_t)blockIdx.x * (kBlock * R) + threadIdx.x;
    T xs[R][S::N], us[R][S::M];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = base + r * kBlock;
        if (i < B) {
            RowIO<T, S::N>::load(x, i, xs[r]);
            RowIO<T, S::M>::load(u, i, us[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = base + r * kBlock;
        if (i < B) {
            T uc[S::M], o[S::N];
            clip_u<T, S::M>(lim, us[r], uc);
            integrate<INTEG>(sys, lim.dt, xs[r], uc, o);
            RowIO<T, S::N>::store(xn, i, o);
        }
    }
}

template <typename S, typename T> struct X0P { T mean[S::N], std[S::N]; };

template <typename S, typename T>
HJBX_DEV void k_initial_state_body(S sys, X0P<S, T> p, const T* __restrict__ u01,
                                                          T* __restrict__ x0, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T r[S::N], o[S::N];
    RowIO<T, S::N>::load(u01, i, r);
#pragma unroll
    for (int k = 0; k < S::N; ++k) {
        const T lo = -p.std[k], hi = p.std[k];  // np.random.uniform(low, high): low + (high-low)*u
        o[k] = (lo + (hi - lo) * r[k]) + p.mean[k];
    }
    sys.wrap(o);
    RowIO<T, S::N>::store(x0, i, o);
}

template <typename S, typename T>
HJBX_DEV void k_running_cost_body(S sys, TaskP<T, S::N, S::M> tk, const T* __restrict__ x,
                                                         const T* __restrict__ u, T* __restrict__ cost, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N], us[S::M], e[S::N];
    RowIO<T, S::N>::load(x, i, xs);
    RowIO<T, S::M>::load(u, i, us);
    error_coords(sys, tk.xf, xs, e);
    cost[i] = running_cost_e<S, T>(tk, e, us);
}

template <typename S, typename T>
HJBX_DEV void k_termination_cost_body(S sys, TaskP<T, S::N, S::M> tk, const T* __restrict__ x,
                                                             T* __restrict__ cost, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N], e[S::N];
    RowIO<T, S::N>::load(x, i, xs);
    error_coords(sys, tk.xf, xs, e);
    cost[i] = quad_form<S::N>(tk.P, e);
}

template <typename S, typename T>
HJBX_DEV void k_control_from_grad_body(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim,
                                                              const T* __restrict__ x, const T* __restrict__ g,
                                                              T* __restrict__ u, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N], gs[S::N], f1[S::N], f2[S::N * S::M], ur[S::M], uo[S::M];
    RowIO<T, S::N>::load(x, i, xs);
    RowIO<T, S::N>::load(g, i, gs);
    sys.affine(xs, f1, f2);
    control_from_grad<S, T>(tk, lim, f2, gs, ur, uo);
    RowIO<T, S::M>::store(u, i, uo);
}

template <int CK, typename S, typename T>
HJBX_DEV void k_controller_body(S sys, CtrlP<T, S::N, S::M> c, Limits<T, S::M> lim,
                                                       const T* __restrict__ x, T* __restrict__ u, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N], uo[S::M];
    RowIO<T, S::N>::load(x, i, xs);
    controller_eval<CK>(sys, c, lim, xs, uo);
    RowIO<T, S::M>::store(u, i, uo);
}

// ----------------------------------------------------------------------------------------------
// deterministic 3-way sum inside ONE launch: lane partials (double) -> wave64 shuffle tree -> LDS across the 4 waves
// -> one (3 x double) record per workgroup in the caller's workspace -> the workgroup that arrives LAST sums the
// records in index order and writes `sums`.  No float atomics and a fixed summation order: results are bitwise
// reproducible run to run.  (Round 1 did the last stage in a second single-wave launch: 5.7 us + a kernel boundary.)
//
// Cross-workgroup hand-off (guide 6 G16, R1 form): the record is stored write-through (8-byte agent-scope atomic stores =
// global_store sc1), the storing wave drains them (s_waitcnt vmcnt(0)), then ONE lane takes a ticket with a returning
// agent-scope atomic add.  Tickets are sharded over kShards counters (each on a 128-byte line of its own; the last arriver
// of a shard takes a ticket of the top counter): 1024 workgroups finishing together would otherwise serialise on one word
// (~11 ns per atomic).  The last arriver issues one agent-scope acquire and reads the records with agent-scope loads.
// The counters are left at zero by the workgroups that saw the last tickets: the workspace must be zero-filled once after
// allocation and is zero again after every call.
// ----------------------------------------------------------------------------------------------
static constexpr int kShards = 32;
static constexpr int kShardStrideWords = 32;                                           // 128 bytes per counter
static constexpr size_t kCounterBytes = (size_t)(kShards + 1) * kShardStrideWords * 4;  // shard counters + the top counter
#define HJBX_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

HJBX_DEV double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <typename T> HJBX_DEV void block_sum3(double a, double b, double c, unsigned char* ws, T* __restrict__ sums) {
    __shared__ double lds[3][kBlock / 64];
    __shared__ unsigned last_s;
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { lds[0][wave] = a; lds[1][wave] = b; lds[2][wave] = c; }
    __syncthreads();
    unsigned* cnt = reinterpret_cast<unsigned*>(ws);
    double* rec = reinterpret_cast<double*>(ws + kCounterBytes);
    if (threadIdx.x == 0) {
        unsigned last = 0;
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) { s0 += lds[0][w]; s1 += lds[1][w]; s2 += lds[2][w]; }
        double* r = rec + 3 * (size_t)blockIdx.x;
        __hip_atomic_store(r + 0, s0, HJBX_RLX_AGENT);
        __hip_atomic_store(r + 1, s1, HJBX_RLX_AGENT);
        __hip_atomic_store(r + 2, s2, HJBX_RLX_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the record (and this wave's other stores) have left before the ticket is taken
        const unsigned shard = blockIdx.x % kShards;
        const unsigned in_shard = (gridDim.x - shard + kShards - 1) / kShards;          // workgroups with blockIdx % kShards == shard
        unsigned* sc = cnt + shard * kShardStrideWords;
        // The tickets stay RELAXED agent-scope atomics behind the explicit drain above (guide 6 G16, R1 form: write-through record
        // stores + s_waitcnt vmcnt(0) + returning atomic).  Round 3 tried __ATOMIC_RELEASE tickets + an acquire fence per shard:
        // a release at agent scope is a buffer_wbl2 of the L2, which here holds the 16-52 MB of loss_i / dloss_dgrad the kernel has just
        // written -- the cartpole entry point went 18.1 -> 40.5 us at B = 2^20 (gpurun_out/r03_bench1.json).  Reverted.
        if (__hip_atomic_fetch_add(sc, 1u, HJBX_RLX_AGENT) == in_shard - 1) {
            __hip_atomic_store(sc, 0u, HJBX_RLX_AGENT);                                 // every workgroup of this shard has arrived
            unsigned* top = cnt + kShards * kShardStrideWords;
            const unsigned nshards = gridDim.x < (unsigned)kShards ? gridDim.x : (unsigned)kShards;
            if (__hip_atomic_fetch_add(top, 1u, HJBX_RLX_AGENT) == nshards - 1) {
                __hip_atomic_store(top, 0u, HJBX_RLX_AGENT);
                last = 1;
            }
        }
        last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    // The last workgroup to arrive adds the records -- with ALL its threads and every load issued before the first add (one memory latency
    // instead of gridDim / 64 of them one after the other on one wave: that serial loop was ~4 us of the 18 us the cartpole entry point takes at
    // B = 2^20), in a fixed order: thread t takes records t, t + 256, ..., then the fixed shuffle tree, then the waves in index order.
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr int PER = (kReduceBlocks + kBlock - 1) / kBlock;
    double va[PER], vb[PER], vc[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const unsigned r = threadIdx.x + q * kBlock;
        const bool in = r < gridDim.x;
        va[q] = in ? __hip_atomic_load(rec + 3 * (size_t)r + 0, HJBX_RLX_AGENT) : 0.0;
        vb[q] = in ? __hip_atomic_load(rec + 3 * (size_t)r + 1, HJBX_RLX_AGENT) : 0.0;
        vc[q] = in ? __hip_atomic_load(rec + 3 * (size_t)r + 2, HJBX_RLX_AGENT) : 0.0;
    }
    double fa = 0, fb = 0, fc = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) { fa += va[q]; fb += vb[q]; fc += vc[q]; }
    for (unsigned r = threadIdx.x + PER * kBlock; r < gridDim.x; r += kBlock) {      // (grids beyond kReduceBlocks: not launched by this library)
        fa += __hip_atomic_load(rec + 3 * (size_t)r + 0, HJBX_RLX_AGENT);
        fb += __hip_atomic_load(rec + 3 * (size_t)r + 1, HJBX_RLX_AGENT);
        fc += __hip_atomic_load(rec + 3 * (size_t)r + 2, HJBX_RLX_AGENT);
    }
    fa = wave_sum(fa); fb = wave_sum(fb); fc = wave_sum(fc);
    __syncthreads();                                  // (lds is reused: every thread has passed the first use)
    if (lane == 0) { lds[0][wave] = fa; lds[1][wave] = fb; lds[2][wave] = fc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) { s0 += lds[0][w]; s1 += lds[1][w]; s2 += lds[2][w]; }
        sums[0] = (T)s0; sums[1] = (T)s1; sums[2] = (T)s2;
    }
}

// hjb_loss body (vhjb.py:227-241) + analytic d loss_i / d gradV
template <int MODE, int R, typename S, typename T>
HJBX_DEV void k_hjb_residual_body(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim,
                                                         const T* __restrict__ x, const T* __restrict__ g,
                                                         const T* __restrict__ done, T* __restrict__ loss_i,
                                                         T* __restrict__ dl_dg, unsigned char* ws, T* __restrict__ sums, int64_t B) {
    constexpr int N = S::N;
    // R rows in flight per thread (see k_simulate); the grid is capped at kReduceBlocks workgroups
    double acc_l = 0, acc_nb = 0, acc_nd = 0;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; i0 < B; i0 += R * stride) {
        T xs[R][N], gs[R][N], dnv[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = i0 + r * stride;
            if (i < B) {
                RowIO<T, N>::load(x, i, xs[r]);
                RowIO<T, N>::load(g, i, gs[r]);
                dnv[r] = done[i];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t i = i0 + r * stride;
            if (i < B) {
                T out[N], li;
                const T dn = dnv[r];
                hjb_residual_env<MODE>(sys, tk, lim, xs[r], gs[r], dn, dl_dg != nullptr, li, out);
                if (loss_i) loss_i[i] = li;
                if (dl_dg) RowIO<T, N>::store(dl_dg, i, out);
                acc_l += (double)li;
                acc_nb += (double)(T(1) - dn);
                acc_nd += (double)dn;
            }
        }
    }
    if (ws) block_sum3<T>(acc_l, acc_nb, acc_nd, ws, sums);
}

// termination_loss body (vhjb.py:243-253)
template <typename T>
HJBX_DEV void k_termination_residual_body(T eps, const T* __restrict__ V, const T* __restrict__ cost,
                                                                 const T* __restrict__ done, T* __restrict__ loss_i,
                                                                 T* __restrict__ dl_dV, unsigned char* ws, T* __restrict__ sums, int64_t B) {
    double acc_l = 0, acc_nb = 0, acc_nd = 0;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < B; i += (int64_t)gridDim.x * kBlock) {
        const T dn = done[i];
        T li, dv;
        termination_residual_env<T>(eps, V[i], cost[i], dn, li, dv);
        if (loss_i) loss_i[i] = li;
        if (dl_dV) dl_dV[i] = dv;
        acc_l += (double)li;
        acc_nb += 1.0 - (double)dn;
        acc_nd += (double)dn;
    }
    if (ws) block_sum3<T>(acc_l, acc_nb, acc_nd, ws, sums);
}

// ----------------------------------------------------------------------------------------------
// closed loop: one VHJB step given gradV, and whole rollouts under closed-form controllers
// ----------------------------------------------------------------------------------------------
template <int INTEG, int R, typename S, typename T>
HJBX_DEV void k_vhjb_step_body(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim, int t, int T_max,
                                                      const T* x, const T* __restrict__ g, T* xn, T* __restrict__ u_out,
                                                      T* __restrict__ cost_t, T* __restrict__ done_t,
                                                      int32_t* __restrict__ done_step, T* __restrict__ resid_t, int64_t B) {
    constexpr int N = S::N, M = S::M;
    const int64_t base = (int64_t)blockIdx.x * (kBlock * R) + threadIdx.x;
    T xs[R][N], gs[R][N];
    int32_t dsv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {   // (see k_simulate: all loads first)
        const int64_t i = base + r * kBlock;
        if (i < B) {
            RowIO<T, N>::load(x, i, xs[r]);
            RowIO<T, N>::load(g, i, gs[r]);
            dsv[r] = done_step[i];
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = base + r * kBlock;
        if (i < B) {
            T xo[N], u[M];
            int32_t ds = dsv[r];
            T c, d, res;
            vhjb_step_env<INTEG>(sys, tk, lim, t, T_max, resid_t != nullptr, xs[r], gs[r], ds, xo, u, c, d, res);
            if (ds != dsv[r]) done_step[i] = ds;
            RowIO<T, N>::store(xn, i, xo);
            if (u_out) RowIO<T, M>::store(u_out, i, u);
            cost_t[i] = c;
            done_t[i] = d;
            if (resid_t) resid_t[i] = res;
        }
    }
}

template <int INTEG, int CK, typename S, typename T>
HJBX_DEV void k_rollout_feedback_body(S sys, TaskP<T, S::N, S::M> tk, CtrlP<T, S::N, S::M> c,
                                                             Limits<T, S::M> lim, uint32_t flags, int has_task, int T_steps,
                                                             const T* __restrict__ x0, T* __restrict__ traj,
                                                             T* __restrict__ u_log, T* __restrict__ cost,
                                                             int32_t* __restrict__ done_step, T* __restrict__ total_cost,
                                                             T* __restrict__ x_final, int64_t B) {
    constexpr int N = S::N, M = S::M;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T x[N], xn[N], u[M];
    RowIO<T, N>::load(x0, i, x);
    const bool term = (flags & kRolloutTerminate) != 0;
    const bool stop_at_target = (flags & kRolloutStopAtTarget) != 0;
    int ds = -1;
    T tot = T(0);
    for (int t = 0; t <= T_steps; ++t) {
        if (traj) RowIO<T, N>::store(traj + (int64_t)t * B * N, i, x);
        T cst = T(0);
#pragma unroll
        for (int j = 0; j < M; ++j) u[j] = T(0);
        if (ds < 0) {
            T e[N];
            bool oob = false;
            if (has_task) {
                error_coords(sys, tk.xf, x, e);
                oob = term && out_of_box<S, T>(tk, e);
            }
            bool reached = false;
            if (stop_at_target) {  // cell 9 of the time-optimal notebook: `if x.T @ x <= metric: record t; break`
                T d2 = T(0);
#pragma unroll
                for (int k = 0; k < N; ++k) d2 += (x[k] - c.xf[k]) * (x[k] - c.xf[k]);
                reached = d2 <= c.eps_region;
            }
            if (t == T_steps || oob || reached) {
                if (has_task && term && !reached) cst = quad_form<N>(tk.P, e);
                ds = t;
            } else {
                controller_eval<CK>(sys, c, lim, x, u);
                if (has_task) cst = running_cost_e<S, T>(tk, e, u) * lim.dt;
                integrate<INTEG>(sys, lim.dt, x, u, xn);
#pragma unroll
                for (int k = 0; k < N; ++k) x[k] = xn[k];
            }
        }
        tot += cst;
        if (cost) cost[(int64_t)t * B + i] = cst;
        if (u_log && t < T_steps) RowIO<T, M>::store(u_log + (int64_t)t * B * M, i, u);
    }
    if (done_step) done_step[i] = ds;
    if (total_cost) total_cost[i] = tot;
    if (x_final) RowIO<T, N>::store(x_final, i, x);
}

}  // namespace hjbx
