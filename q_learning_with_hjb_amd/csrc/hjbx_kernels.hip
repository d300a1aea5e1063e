// hjbx_kernels.hip -- gfx950 kernels + C ABI (include/hjbx.h) for batched control-affine rollouts
// and HJB residuals.  MI355X only: wave64, one lane per environment, state vectors in VGPRs,
// system/task constants in SGPRs (kernarg), row-vector global accesses, wave-shuffle reductions.
//
// All of these kernels are HBM-bandwidth bound (a few dozen flops + one sincos per 36-128 bytes);
// algorithmic byte counts per environment are tabulated in DESIGN.md.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <type_traits>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"
#include "hjbx_host.hpp"

using namespace hjbx;

// ----------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int hjbx_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// process-wide knobs (include/hjbx.h: hjbx_option)
static std::atomic<int> g_options[4] = {{0}, {0}, {0}, {0}};   // (HJBX_OPT_MLP_ARITHMETIC defaults to 0 = float32 MFMA, the reference's arithmetic)
int hjbx_option_value(int option) { return (option >= 0 && option < 4) ? g_options[option].load(std::memory_order_relaxed) : 0; }

#define HJBX_REQUIRE(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) return hjbx_set_error(HJBX_EINVAL, __VA_ARGS__); \
    } while (0)

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "%s: %s", what, hipGetErrorString(e));
    return HJBX_OK;
}

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
static constexpr int kReduceBlocks = 1024;  // grid cap of the reducing kernels (4 per CU); 4096 measured no better

static inline dim3 grid_for(int64_t B) { return dim3((unsigned)((B + kBlock - 1) / kBlock)); }
static inline dim3 grid_rows(int64_t B, int R) { return dim3((unsigned)((B + (int64_t)kBlock * R - 1) / ((int64_t)kBlock * R))); }
// rows per thread of the streaming kernels: enough loads in flight to cover the HBM latency once the batch fills the chip; small
// batches keep one row per thread (more workgroups); float64 keeps one row (its row state alone is 2x the registers)
template <typename T> static inline int rows_per_thread(int64_t B, int n, bool reducing = false) {
    if (sizeof(T) != 4) return 1;
    const int forced = hjbx_option_value(HJBX_OPT_STREAM_ROWS);
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    // measured with buffers rotated through 640 MB (tools/kernel_bench.py --rows 1|2|4, B = 2^20): one row per thread is fastest for
    // simulate / vhjb_step (more rows cost resident waves: near-hover vhjb_step 32 -> 46 -> 64 us), two rows help the residual
    // kernel, whose grid is capped for the in-kernel reduction (cartpole 22 -> 19 us)
    (void)n;
    return reducing && B >= (1 << 18) ? 2 : 1;
}

// ----------------------------------------------------------------------------------------------
// pointwise kernels
// ----------------------------------------------------------------------------------------------
template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_affine(S sys, const T* __restrict__ x, T* __restrict__ f1,
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
__global__ __launch_bounds__(kBlock) void k_wrap(S sys, const T* x, T* out, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N];
    RowIO<T, S::N>::load(x, i, xs);
    sys.wrap(xs);
    RowIO<T, S::N>::store(out, i, xs);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_xdot(S sys, const T* __restrict__ x, const T* __restrict__ u,
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
__global__ __launch_bounds__(kBlock) void k_simulate(S sys, Limits<T, S::M> lim, const T* x, const T* __restrict__ u,
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
__global__ __launch_bounds__(kBlock) void k_initial_state(S sys, X0P<S, T> p, const T* __restrict__ u01,
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
__global__ __launch_bounds__(kBlock) void k_running_cost(S sys, TaskP<T, S::N, S::M> tk, const T* __restrict__ x,
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
__global__ __launch_bounds__(kBlock) void k_termination_cost(S sys, TaskP<T, S::N, S::M> tk, const T* __restrict__ x,
                                                             T* __restrict__ cost, int64_t B) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= B) return;
    T xs[S::N], e[S::N];
    RowIO<T, S::N>::load(x, i, xs);
    error_coords(sys, tk.xf, xs, e);
    cost[i] = quad_form<S::N>(tk.P, e);
}

template <typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_control_from_grad(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim,
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
__global__ __launch_bounds__(kBlock) void k_controller(S sys, CtrlP<T, S::N, S::M> c, Limits<T, S::M> lim,
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
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { lds[0][wave] = a; lds[1][wave] = b; lds[2][wave] = c; }
    __syncthreads();
    if (wave != 0) return;
    unsigned* cnt = reinterpret_cast<unsigned*>(ws);
    double* rec = reinterpret_cast<double*>(ws + kCounterBytes);
    unsigned last = 0;
    if (lane == 0) {
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
        // the tickets are agent-scope RELEASE operations (paired with the acquire fence of the last arriver below): the ordering of
        // record before ticket then holds by the memory model, not only by the explicit drain above
        if (__hip_atomic_fetch_add(sc, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == in_shard - 1) {
            __hip_atomic_store(sc, 0u, HJBX_RLX_AGENT);                                 // every workgroup of this shard has arrived
            unsigned* top = cnt + kShards * kShardStrideWords;
            const unsigned nshards = gridDim.x < (unsigned)kShards ? gridDim.x : (unsigned)kShards;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                          // (the other workgroups' records of this shard)
            if (__hip_atomic_fetch_add(top, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == nshards - 1) {
                __hip_atomic_store(top, 0u, HJBX_RLX_AGENT);
                last = 1;
            }
        }
    }
    if (!__builtin_amdgcn_readfirstlane((int)last)) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double fa = 0, fb = 0, fc = 0;
    for (unsigned r = lane; r < gridDim.x; r += 64) {      // records in index order per lane, then the fixed shuffle tree
        fa += __hip_atomic_load(rec + 3 * (size_t)r + 0, HJBX_RLX_AGENT);
        fb += __hip_atomic_load(rec + 3 * (size_t)r + 1, HJBX_RLX_AGENT);
        fc += __hip_atomic_load(rec + 3 * (size_t)r + 2, HJBX_RLX_AGENT);
    }
    fa = wave_sum(fa); fb = wave_sum(fb); fc = wave_sum(fc);
    if (lane == 0) { sums[0] = (T)fa; sums[1] = (T)fb; sums[2] = (T)fc; }
}

// hjb_loss body (vhjb.py:227-241) + analytic d loss_i / d gradV
template <int MODE, int R, typename S, typename T>
__global__ __launch_bounds__(kBlock) void k_hjb_residual(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim,
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
__global__ __launch_bounds__(kBlock) void k_termination_residual(T eps, const T* __restrict__ V, const T* __restrict__ cost,
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
__global__ __launch_bounds__(kBlock) void k_vhjb_step(S sys, TaskP<T, S::N, S::M> tk, Limits<T, S::M> lim, int t, int T_max,
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
__global__ __launch_bounds__(kBlock) void k_rollout_feedback(S sys, TaskP<T, S::N, S::M> tk, CtrlP<T, S::N, S::M> c,
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
    const bool term = (flags & HJBX_ROLLOUT_TERMINATE) != 0;
    const bool stop_at_target = (flags & HJBX_ROLLOUT_STOP_AT_TARGET) != 0;
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

// host side: descriptor conversion and dispatch live in hjbx_host.hpp (shared with hjbx_mlp.hip)

static int unsupported(const hjbx_system* s) {
    return hjbx_set_error(HJBX_EUNSUPPORTED, "no kernel for system kind %d with n=%d m=%d", s->kind, s->n, s->m);
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// a (B, cols) row-major buffer is accessed with RowIO's vector width: 16, 8 or 4 bytes
static bool aligned_rows(const void* p, size_t row_bytes) {
    const uintptr_t a = (row_bytes % 16 == 0) ? 15u : (row_bytes % 8 == 0) ? 7u : 3u;
    return (reinterpret_cast<uintptr_t>(p) & a) == 0;
}

#define HJBX_CHECK_COMMON(sys, B)                                                   \
    HJBX_REQUIRE((sys) != nullptr, "system handle is NULL");                        \
    HJBX_REQUIRE((B) >= 0, "negative batch size %lld", (long long)(B));             \
    if ((B) == 0) return HJBX_OK;

// ROWS(p, cols): non-NULL (B, cols) buffer of T aligned for its row vector width; OPT: may be NULL
#define HJBX_CHECK_ROWS(p, cols) \
    HJBX_REQUIRE((p) != nullptr && aligned_rows(p, (size_t)(cols) * sizeof(T)), #p " must be a non-NULL device pointer aligned to its row vector width")
#define HJBX_CHECK_OPT(p, cols) \
    HJBX_REQUIRE((p) == nullptr || aligned_rows(p, (size_t)(cols) * sizeof(T)), #p " must be aligned to its row vector width")

// ---- typed implementations ---------------------------------------------------------------------
template <typename T> static int affine_impl(const hjbx_system* sys, const T* x, T* f1, T* f2, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(f1, sys->n); HJBX_CHECK_ROWS(f2, sys->n * sys->m);
    if (!with_system<T>(sys, [&](auto S) {
            hipLaunchKernelGGL((k_affine<decltype(S), T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, x, f1, f2, B);
        })) return unsupported(sys);
    return check_launch("hjbx_affine");
}

template <typename T> static int wrap_impl(const hjbx_system* sys, const T* x, T* out, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(out, sys->n);
    if (!with_system<T>(sys, [&](auto S) {
            hipLaunchKernelGGL((k_wrap<decltype(S), T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, x, out, B);
        })) return unsupported(sys);
    return check_launch("hjbx_wrap");
}

template <typename T> static int xdot_impl(const hjbx_system* sys, const T* x, const T* u, T* xd, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m); HJBX_CHECK_ROWS(xd, sys->n);
    if (!with_system<T>(sys, [&](auto S) {
            hipLaunchKernelGGL((k_xdot<decltype(S), T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, x, u, xd, B);
        })) return unsupported(sys);
    return check_launch("hjbx_dynamics_step");
}

template <typename T>
static int simulate_impl(const hjbx_system* sys, int integ, const T* x, const T* u, T* xn, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m); HJBX_CHECK_ROWS(xn, sys->n);
    if (int rc = check_integrator(sys, integ, "hjbx_simulate")) return rc;
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto lim = make_limits<T, SS::M>(sys);
            auto go = [&](auto integc, auto rc) {
                constexpr int I = decltype(integc)::value, R = decltype(rc)::value;
                hipLaunchKernelGGL((k_simulate<I, R, SS, T>), grid_rows(B, R), dim3(kBlock), 0, (hipStream_t)st, S, lim, x, u, xn, B);
            };
            auto with_r = [&](auto integc) {
                const int R = rows_per_thread<T>(B, SS::N);
                if (R == 4) go(integc, std::integral_constant<int, 4>{});
                else if (R == 2) go(integc, std::integral_constant<int, 2>{});
                else go(integc, std::integral_constant<int, 1>{});
            };
            if (integ == HJBX_EULER) with_r(std::integral_constant<int, 0>{});
            else if (integ == HJBX_RK4) with_r(std::integral_constant<int, 1>{});
            else if constexpr (is_linear<SS>::value) with_r(std::integral_constant<int, 2>{});
        })) return unsupported(sys);
    return check_launch("hjbx_simulate");
}

template <typename T>
static int initial_state_impl(const hjbx_system* sys, const double* mean, const double* sd, const T* u01, T* x0, int64_t B,
                              void* st) {
    HJBX_CHECK_COMMON(sys, B); HJBX_CHECK_ROWS(u01, sys->n); HJBX_CHECK_ROWS(x0, sys->n);
    HJBX_REQUIRE(mean && sd, "x0_mean / x0_std are NULL");
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            X0P<SS, T> p;
            for (int i = 0; i < SS::N; ++i) { p.mean[i] = (T)mean[i]; p.std[i] = (T)sd[i]; }
            hipLaunchKernelGGL((k_initial_state<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, p, u01, x0, B);
        })) return unsupported(sys);
    return check_launch("hjbx_initial_state");
}

template <typename T>
static int running_cost_impl(const hjbx_system* sys, const hjbx_task* task, const T* x, const T* u, T* cost, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc; HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m); HJBX_CHECK_ROWS(cost, 1);
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            hipLaunchKernelGGL((k_running_cost<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S,
                               make_task<T, SS::N, SS::M>(task), x, u, cost, B);
        })) return unsupported(sys);
    return check_launch("hjbx_running_cost");
}

template <typename T>
static int termination_cost_impl(const hjbx_system* sys, const hjbx_task* task, const T* x, T* cost, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc; HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(cost, 1);
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            hipLaunchKernelGGL((k_termination_cost<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S,
                               make_task<T, SS::N, SS::M>(task), x, cost, B);
        })) return unsupported(sys);
    return check_launch("hjbx_termination_cost");
}

template <typename T>
static int control_from_grad_impl(const hjbx_system* sys, const hjbx_task* task, const T* x, const T* g, T* u, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc; HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(g, sys->n); HJBX_CHECK_ROWS(u, sys->m);
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            hipLaunchKernelGGL((k_control_from_grad<SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S,
                               make_task<T, SS::N, SS::M>(task), make_limits<T, SS::M>(sys), x, g, u, B);
        })) return unsupported(sys);
    return check_launch("hjbx_control_from_grad");
}

static inline int reduce_grid(int64_t B) {
    int64_t g = (B + kBlock - 1) / kBlock;
    return (int)(g < kReduceBlocks ? g : kReduceBlocks);
}

template <typename T>
static int hjb_residual_impl(const hjbx_system* sys, const hjbx_task* task, int mode, const T* x, const T* g, const T* done,
                             T* loss_i, T* dl_dg, T* sums, void* workspace, int64_t B, void* st) {
    HJBX_REQUIRE(sys != nullptr, "system handle is NULL");
    HJBX_REQUIRE(B >= 0, "negative batch size");
    if (int rc = check_task(task)) return rc;
    HJBX_REQUIRE(mode == HJBX_RESIDUAL_NORMALISED || mode == HJBX_RESIDUAL_RAW, "unknown residual mode %d", mode);
    HJBX_REQUIRE(!sums || (workspace && aligned16(workspace)), "sums requested but workspace is NULL/unaligned");
    if (B == 0) {
        if (sums) {
            hipError_t e = hipMemsetAsync(sums, 0, 3 * sizeof(T), (hipStream_t)st);
            if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return HJBX_OK;
    }
    HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(g, sys->n); HJBX_CHECK_ROWS(done, 1); HJBX_CHECK_OPT(loss_i, 1); HJBX_CHECK_OPT(dl_dg, sys->n);
    const int grid = reduce_grid(B);
    unsigned char* ws = sums ? (unsigned char*)workspace : nullptr;
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto tk = make_task<T, SS::N, SS::M>(task);
            auto lim = make_limits<T, SS::M>(sys);
            auto go = [&](auto modec, auto rc) {
                constexpr int MD = decltype(modec)::value, R = decltype(rc)::value;
                hipLaunchKernelGGL((k_hjb_residual<MD, R, SS, T>), dim3(grid), dim3(kBlock), 0, (hipStream_t)st, S, tk, lim, x, g, done, loss_i, dl_dg,
                                   ws, sums, B);
            };
            auto with_r = [&](auto modec) {
                const int R = rows_per_thread<T>(B, SS::N, true);
                if (R == 4) go(modec, std::integral_constant<int, 4>{});
                else if (R == 2) go(modec, std::integral_constant<int, 2>{});
                else go(modec, std::integral_constant<int, 1>{});
            };
            if (mode == HJBX_RESIDUAL_NORMALISED) with_r(std::integral_constant<int, 0>{});
            else with_r(std::integral_constant<int, 1>{});
        })) return unsupported(sys);
    return check_launch("hjbx_hjb_residual");
}

template <typename T>
static int termination_residual_impl(double eps, const T* V, const T* cost, const T* done, T* loss_i, T* dl_dV, T* sums,
                                     void* workspace, int64_t B, void* st) {
    HJBX_REQUIRE(B >= 0, "negative batch size");
    HJBX_REQUIRE(!sums || (workspace && aligned16(workspace)), "sums requested but workspace is NULL/unaligned");
    if (B == 0) {
        if (sums) {
            hipError_t e = hipMemsetAsync(sums, 0, 3 * sizeof(T), (hipStream_t)st);
            if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return HJBX_OK;
    }
    HJBX_REQUIRE(V && cost && done, "V/cost/done must be non-NULL");
    const int grid = reduce_grid(B);
    unsigned char* ws = sums ? (unsigned char*)workspace : nullptr;
    hipLaunchKernelGGL((k_termination_residual<T>), dim3(grid), dim3(kBlock), 0, (hipStream_t)st, (T)eps, V, cost, done,
                       loss_i, dl_dV, ws, sums, B);
    return check_launch("hjbx_termination_residual");
}

template <typename T>
static int vhjb_step_impl(const hjbx_system* sys, const hjbx_task* task, int integ, int t, int T_max, const T* x, const T* g,
                          T* xn, T* u_out, T* cost_t, T* done_t, int32_t* done_step, T* resid_t, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B); if (int rc = check_task(task)) return rc;
    if (int rc = check_integrator(sys, integ, "hjbx_vhjb_step")) return rc;
    HJBX_REQUIRE(t >= 0 && T_max >= 0, "negative step index");
    HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(g, sys->n); HJBX_CHECK_ROWS(xn, sys->n); HJBX_CHECK_OPT(u_out, sys->m);
    HJBX_REQUIRE(cost_t && done_t && done_step, "cost_t/done_t/done_step must be non-NULL");
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto tk = make_task<T, SS::N, SS::M>(task);
            auto lim = make_limits<T, SS::M>(sys);
            auto go = [&](auto integc, auto rc) {
                constexpr int I = decltype(integc)::value, R = decltype(rc)::value;
                hipLaunchKernelGGL((k_vhjb_step<I, R, SS, T>), grid_rows(B, R), dim3(kBlock), 0, (hipStream_t)st, S, tk, lim, t, T_max, x,
                                   g, xn, u_out, cost_t, done_t, done_step, resid_t, B);
            };
            auto with_r = [&](auto integc) {
                const int R = rows_per_thread<T>(B, SS::N);
                if (R == 4) go(integc, std::integral_constant<int, 4>{});
                else if (R == 2) go(integc, std::integral_constant<int, 2>{});
                else go(integc, std::integral_constant<int, 1>{});
            };
            if (integ == HJBX_EULER) with_r(std::integral_constant<int, 0>{});
            else if (integ == HJBX_RK4) with_r(std::integral_constant<int, 1>{});
            else if constexpr (is_linear<SS>::value) with_r(std::integral_constant<int, 2>{});
        })) return unsupported(sys);
    return check_launch("hjbx_vhjb_step");
}

template <typename S> struct is_cartpole { static constexpr bool value = false; };
template <typename T> struct is_cartpole<Cartpole<T>> { static constexpr bool value = true; };
template <typename S> struct is_acrobot { static constexpr bool value = false; };
template <typename T> struct is_acrobot<Acrobot<T>> { static constexpr bool value = true; };

static int check_ctrl(const hjbx_system* sys, const hjbx_controller* c) {
    HJBX_REQUIRE(c, "controller is NULL");
    HJBX_REQUIRE(c->kind >= HJBX_CTRL_LINEAR_FEEDBACK && c->kind <= HJBX_CTRL_DI_TIME_OPTIMAL, "unknown controller kind %d", c->kind);
    if (c->kind == HJBX_CTRL_DI_TIME_OPTIMAL && !(sys->kind == HJBX_SYS_LINEAR && sys->n == 2 && sys->m == 1))
        return hjbx_set_error(HJBX_EINVAL, "the time-optimal bang-bang controller needs the double integrator (LINEAR, n=2, m=1)");
    if (c->kind == HJBX_CTRL_CARTPOLE_ENERGY && sys->kind != HJBX_SYS_CARTPOLE)
        return hjbx_set_error(HJBX_EINVAL, "cartpole energy-shaping controller needs a cartpole system");
    if (c->kind == HJBX_CTRL_ACROBOT_ENERGY && sys->kind != HJBX_SYS_ACROBOT)
        return hjbx_set_error(HJBX_EINVAL, "acrobot energy-shaping controller needs an acrobot system");
    return HJBX_OK;
}

template <typename T>
static int controller_impl(const hjbx_system* sys, const hjbx_controller* c, const T* x, T* u, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B);
    if (int rc = check_ctrl(sys, c)) return rc;
    HJBX_CHECK_ROWS(x, sys->n); HJBX_CHECK_ROWS(u, sys->m);
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            auto cp = make_ctrl<T, SS::N, SS::M>(c);
            auto lim = make_limits<T, SS::M>(sys);
            if constexpr (is_cartpole<SS>::value) {
                if (c->kind == HJBX_CTRL_CARTPOLE_ENERGY) {
                    hipLaunchKernelGGL((k_controller<1, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
                    return;
                }
            }
            if constexpr (is_di<SS>::value) {
                if (c->kind == HJBX_CTRL_DI_TIME_OPTIMAL) {
                    hipLaunchKernelGGL((k_controller<3, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
                    return;
                }
            }
            if constexpr (is_acrobot<SS>::value) {
                if (c->kind == HJBX_CTRL_ACROBOT_ENERGY) {
                    hipLaunchKernelGGL((k_controller<2, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
                    return;
                }
            }
            hipLaunchKernelGGL((k_controller<0, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, cp, lim, x, u, B);
        })) return unsupported(sys);
    return check_launch("hjbx_controller");
}

template <int INTEG, typename SS, typename T>
static void launch_rollout(const hjbx_system* sys, SS S, const hjbx_task* task, const hjbx_controller* c, uint32_t flags,
                           int T_steps, const T* x0, T* traj, T* u_log, T* cost, int32_t* done_step, T* total_cost, T* x_final,
                           int64_t B, void* st) {
    auto tk = make_task<T, SS::N, SS::M>(task);
    auto cp = make_ctrl<T, SS::N, SS::M>(c);
    auto lim = make_limits<T, SS::M>(sys);
    const int has_task = task != nullptr;
#define HJBX_LAUNCH_RO(CK)                                                                                                    \
    hipLaunchKernelGGL((k_rollout_feedback<INTEG, CK, SS, T>), grid_for(B), dim3(kBlock), 0, (hipStream_t)st, S, tk, cp, lim, \
                       flags, has_task, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B)
    if constexpr (is_cartpole<SS>::value) {
        if (c->kind == HJBX_CTRL_CARTPOLE_ENERGY) { HJBX_LAUNCH_RO(1); return; }
    }
    if constexpr (is_di<SS>::value) {
        if (c->kind == HJBX_CTRL_DI_TIME_OPTIMAL) { HJBX_LAUNCH_RO(3); return; }
    }
    if constexpr (is_acrobot<SS>::value) {
        if (c->kind == HJBX_CTRL_ACROBOT_ENERGY) { HJBX_LAUNCH_RO(2); return; }
    }
    HJBX_LAUNCH_RO(0);
#undef HJBX_LAUNCH_RO
}

template <typename T>
static int rollout_feedback_impl(const hjbx_system* sys, const hjbx_task* task, const hjbx_controller* c, int integ, uint32_t flags,
                                 int T_steps, const T* x0, T* traj, T* u_log, T* cost, int32_t* done_step, T* total_cost,
                                 T* x_final, int64_t B, void* st) {
    HJBX_CHECK_COMMON(sys, B);
    if (int rc = check_ctrl(sys, c)) return rc;
    if (int rc = check_integrator(sys, integ, "hjbx_rollout_feedback")) return rc;
    HJBX_REQUIRE(T_steps >= 0, "negative horizon");
    HJBX_REQUIRE((flags & ~(HJBX_ROLLOUT_TERMINATE | HJBX_ROLLOUT_STOP_AT_TARGET)) == 0, "unknown rollout flags 0x%x", flags);
    HJBX_REQUIRE(task || !(flags & HJBX_ROLLOUT_TERMINATE), "HJBX_ROLLOUT_TERMINATE needs a task");
    if (task) { if (int rc = check_task(task)) return rc; }
    HJBX_REQUIRE(task || (!cost && !total_cost), "cost outputs need a task");
    HJBX_CHECK_ROWS(x0, sys->n); HJBX_CHECK_OPT(traj, sys->n); HJBX_CHECK_OPT(u_log, sys->m); HJBX_CHECK_OPT(x_final, sys->n);
    if (!with_system<T>(sys, [&](auto S) {
            using SS = decltype(S);
            if (integ == HJBX_EULER)
                launch_rollout<0, SS, T>(sys, S, task, c, flags, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B, st);
            else if (integ == HJBX_RK4)
                launch_rollout<1, SS, T>(sys, S, task, c, flags, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B, st);
            else if constexpr (is_linear<SS>::value)
                launch_rollout<2, SS, T>(sys, S, task, c, flags, T_steps, x0, traj, u_log, cost, done_step, total_cost, x_final, B, st);
        })) return unsupported(sys);
    return check_launch("hjbx_rollout_feedback");
}

// ----------------------------------------------------------------------------------------------
// extern "C" surface
// ----------------------------------------------------------------------------------------------
extern "C" {

int hjbx_version(void) { return HJBX_VERSION; }

size_t hjbx_last_error(char* buf, size_t buflen) {
    const size_t len = strlen(g_err);
    if (buf && buflen) {
        const size_t ncopy = len < buflen - 1 ? len : buflen - 1;
        memcpy(buf, g_err, ncopy);
        buf[ncopy] = '\0';
    }
    return len;
}

int hjbx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; ++d) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
}

int hjbx_set_option(int option, int value) {
    HJBX_REQUIRE(option == HJBX_OPT_ROLLOUT_SCHEDULE || option == HJBX_OPT_ROLLOUT_EXTRA_WORKGROUPS || option == HJBX_OPT_STREAM_ROWS || option == HJBX_OPT_MLP_ARITHMETIC, "unknown option %d", option);
    HJBX_REQUIRE(option != HJBX_OPT_MLP_ARITHMETIC || value <= 2, "mlp arithmetic must be 0 (f32 MFMA), 1 (bf16x3-split MFMA) or 2 (f16x2-split MFMA), got %d", value);
    HJBX_REQUIRE(option != HJBX_OPT_ROLLOUT_SCHEDULE || value <= 1, "rollout schedule must be 0 or 1, got %d", value);
    HJBX_REQUIRE(value <= 64, "option value %d out of range", value);
    return value < 0 ? g_options[option].load() : g_options[option].exchange(value);
}

size_t hjbx_reduce_workspace_bytes(void) { return kCounterBytes + (size_t)kReduceBlocks * 3 * sizeof(double); }

int hjbx_system_create(int kind, int n, int m, double dt, const double* umin, const double* umax, const double* params,
                       int n_params, hjbx_system** out) {
    HJBX_REQUIRE(out, "out is NULL");
    *out = nullptr;
    HJBX_REQUIRE(umin && umax && params, "umin/umax/params must be non-NULL");
    HJBX_REQUIRE(dt > 0 && std::isfinite(dt), "dt must be positive and finite");
    int en = 0, em = 0, ep = 0;
    switch (kind) {
    case HJBX_SYS_LINEAR:
        HJBX_REQUIRE(n >= 1 && n <= HJBX_MAX_N && m >= 1 && m <= HJBX_MAX_M, "linear system needs 1<=n<=%d, 1<=m<=%d", HJBX_MAX_N, HJBX_MAX_M);
        en = n; em = m; ep = n * n + n * m; break;
    case HJBX_SYS_CARTPOLE: en = 4; em = 1; ep = 4; break;
    case HJBX_SYS_ACROBOT: en = 4; em = 1; ep = 7; break;
    case HJBX_SYS_QUAD2D: en = 6; em = 2; ep = 4; break;
    case HJBX_SYS_NEARHOVER: en = 10; em = 3; ep = 4; break;
    default: return hjbx_set_error(HJBX_EINVAL, "unknown system kind %d", kind);
    }
    HJBX_REQUIRE(n == en && m == em, "system kind %d has n=%d m=%d, got n=%d m=%d", kind, en, em, n, m);
    HJBX_REQUIRE(n_params == ep || (kind == HJBX_SYS_LINEAR && n_params == 2 * ep),
                 "system kind %d takes %d parameters%s, got %d", kind, ep, kind == HJBX_SYS_LINEAR ? " (or twice that with Ad, Bd)" : "", n_params);
    for (int j = 0; j < m; ++j) HJBX_REQUIRE(umin[j] <= umax[j], "umin[%d] > umax[%d]", j, j);
    hjbx_system* s = new (std::nothrow) hjbx_system();
    if (!s) return hjbx_set_error(HJBX_EINVAL, "out of host memory");
    memset(s, 0, sizeof(*s));
    s->kind = kind; s->n = n; s->m = m; s->dt = dt; s->n_params = n_params;
    for (int j = 0; j < m; ++j) { s->umin[j] = umin[j]; s->umax[j] = umax[j]; }
    for (int i = 0; i < n_params; ++i) s->p[i] = params[i];
    *out = s;
    return HJBX_OK;
}

void hjbx_system_destroy(hjbx_system* sys) { delete sys; }

int hjbx_dims(const hjbx_system* sys, int* n, int* m) {
    HJBX_REQUIRE(sys && n && m, "NULL argument");
    *n = sys->n; *m = sys->m;
    return HJBX_OK;
}

#define HJBX_DEFINE(T, SFX)                                                                                                   \
    int hjbx_affine_##SFX(const hjbx_system* s, const T* x, T* f1, T* f2, int64_t B, void* st) { return affine_impl<T>(s, x, f1, f2, B, st); } \
    int hjbx_wrap_##SFX(const hjbx_system* s, const T* x, T* o, int64_t B, void* st) { return wrap_impl<T>(s, x, o, B, st); }  \
    int hjbx_dynamics_step_##SFX(const hjbx_system* s, const T* x, const T* u, T* xd, int64_t B, void* st) { return xdot_impl<T>(s, x, u, xd, B, st); } \
    int hjbx_simulate_##SFX(const hjbx_system* s, int integ, const T* x, const T* u, T* xn, int64_t B, void* st) { return simulate_impl<T>(s, integ, x, u, xn, B, st); } \
    int hjbx_initial_state_##SFX(const hjbx_system* s, const double* mean, const double* sd, const T* u01, T* x0, int64_t B, void* st) { return initial_state_impl<T>(s, mean, sd, u01, x0, B, st); } \
    int hjbx_running_cost_##SFX(const hjbx_system* s, const hjbx_task* t, const T* x, const T* u, T* c, int64_t B, void* st) { return running_cost_impl<T>(s, t, x, u, c, B, st); } \
    int hjbx_termination_cost_##SFX(const hjbx_system* s, const hjbx_task* t, const T* x, T* c, int64_t B, void* st) { return termination_cost_impl<T>(s, t, x, c, B, st); } \
    int hjbx_control_from_grad_##SFX(const hjbx_system* s, const hjbx_task* t, const T* x, const T* g, T* u, int64_t B, void* st) { return control_from_grad_impl<T>(s, t, x, g, u, B, st); } \
    int hjbx_hjb_residual_##SFX(const hjbx_system* s, const hjbx_task* t, int mode, const T* x, const T* g, const T* done, T* li, T* dg, T* sums, void* ws, int64_t B, void* st) { return hjb_residual_impl<T>(s, t, mode, x, g, done, li, dg, sums, ws, B, st); } \
    int hjbx_termination_residual_##SFX(double eps, const T* V, const T* cost, const T* done, T* li, T* dV, T* sums, void* ws, int64_t B, void* st) { return termination_residual_impl<T>(eps, V, cost, done, li, dV, sums, ws, B, st); } \
    int hjbx_vhjb_step_##SFX(const hjbx_system* s, const hjbx_task* t, int integ, int step, int T_max, const T* x, const T* g, T* xn, T* uo, T* c, T* d, int32_t* ds, T* rs, int64_t B, void* st) { return vhjb_step_impl<T>(s, t, integ, step, T_max, x, g, xn, uo, c, d, ds, rs, B, st); } \
    int hjbx_controller_##SFX(const hjbx_system* s, const hjbx_controller* c, const T* x, T* u, int64_t B, void* st) { return controller_impl<T>(s, c, x, u, B, st); } \
    int hjbx_rollout_feedback_##SFX(const hjbx_system* s, const hjbx_task* t, const hjbx_controller* c, int integ, uint32_t flags, int T_steps, const T* x0, T* traj, T* ul, T* cost, int32_t* ds, T* tc, T* xf, int64_t B, void* st) { return rollout_feedback_impl<T>(s, t, c, integ, flags, T_steps, x0, traj, ul, cost, ds, tc, xf, B, st); }

HJBX_DEFINE(float, f32)
HJBX_DEFINE(double, f64)

}  // extern "C"
