// hjbx_mlp.hip -- ValueFunctionApproximator forward + input gradient (reference controller/vhjb.py:17-60
// and get_v_gradient :201-202) fused into one gfx950 kernel on the f32 matrix cores.
//
//   e = wrap(x - xf); z = (e - mean)/std; h1 = relu(z W1); h2 = relu(h1 W2); y = h2 W3
//   V = |y|^2 + eps_s |e|^2
//   dV/dx = ((((2y) W3') . [h2>0]) W2' . [h1>0]) W1' / std + 2 eps_s e
//
// Design (CDNA4):
//  * v_mfma_f32_32x32x2_f32 (exact f32, 155 TFLOP/s dense) -- the network is float32 in the reference.
//  * Everything is computed TRANSPOSED (features x environments): a wave owns a tile of 32
//    environments (the MFMA column index = lane & 31) and the accumulator registers of one layer ARE
//    the B operands of the next one, forward and backward, with no cross-lane movement and no LDS
//    round trip: accumulator register s of lane-half h holds feature perm(s) + 4h, and the weight (A)
//    operand for k-step s is simply fetched for that same feature.
//  * All three weight matrices live in LDS once per workgroup (104 KB, one copy serves W and W'):
//    rows padded to an ODD stride (129 / 65 floats) so that both the row-walk of the forward pass
//    and the column-walk of the backward pass hit 32 distinct banks per ds_read_b32 lane group.
//  * Persistent grid: one 512-thread workgroup per CU (two waves per SIMD share the matrix pipe, one
//    computes while the other waits on LDS); every wave strides over environment tiles.
// Per environment: 4(128 n + 128*128 + 128*64) flop; algorithmic HBM traffic 4(2n+1) bytes -> MFMA bound.
#include <hip/hip_runtime.h>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"

using namespace hjbx;

using f32x16 = __attribute__((ext_vector_type(16))) float;

static constexpr int kH1 = 128, kH2 = 128, kH3 = 64;
static constexpr int kLD1 = 129, kLD2 = 129, kLD3 = 65;  // odd LDS row strides (floats)
static constexpr int kWaves = 8;                          // 512 threads: 2 waves per SIMD
static constexpr int kThreads = kWaves * 64;

template <int N> struct MlpP { float mean[N], std[N], xf[N], eps_s; };

// accumulator register s of lane-half h holds row perm(s) + 4h of its 32-row block
__device__ __forceinline__ constexpr int perm(int s) { return (s & 3) + 8 * (s >> 2); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)


// One GEMM of the chain: acc[o] += A_o(step) x b(step) for step = 0..NSTEPS-1, o = 0..NOUT-1.
// The A operands (one LDS dword per MFMA) are fetched DEPTH steps ahead of the MFMAs that consume
// them, so a wave keeps the matrix pipe busy on its own instead of exposing the ds_read latency
// before every group (the naive loop waits lgkmcnt(0) in front of each group: 51 % of peak).
template <int NSTEPS, int NOUT, int DEPTH, typename LoadA, typename GetB>
__device__ __forceinline__ void mfma_chain(f32x16 (&acc)[NOUT], LoadA loadA, GetB getB) {
    float ring[DEPTH + 1][NOUT];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int o = 0; o < NOUT; ++o) ring[d][o] = (d < NSTEPS) ? loadA(d, o) : 0.f;
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
        if (st + DEPTH < NSTEPS) {
#pragma unroll
            for (int o = 0; o < NOUT; ++o) ring[(st + DEPTH) % (DEPTH + 1)][o] = loadA(st + DEPTH, o);
        }
        // keep the prefetch in front of this step's MFMAs (the machine scheduler otherwise sinks the
        // ds_read behind them and re-uses the operand registers, serialising read -> wait -> MFMA)
        __builtin_amdgcn_sched_barrier(0);
        const float b = getB(st);
#pragma unroll
        for (int o = 0; o < NOUT; ++o) acc[o] = MFMA(ring[st % (DEPTH + 1)][o], b, acc[o]);
    }
}

template <typename S>
__global__ __launch_bounds__(kThreads, 2) void k_value_grad_mfma(S sys, MlpP<S::N> p, const float* __restrict__ W1g,
                                                                const float* __restrict__ W2g, const float* __restrict__ W3g,
                                                                const float* __restrict__ x, float* __restrict__ Vout,
                                                                float* __restrict__ gout, int64_t B, int64_t ntiles) {
    constexpr int N = S::N;
    static_assert(N % 2 == 0, "state dimension must be even (k-steps of 2)");
    __shared__ float sW1[N * kLD1];
    __shared__ float sW2[kH1 * kLD2];
    __shared__ float sW3[kH2 * kLD3];

    const int tid = threadIdx.x;
    for (int idx = tid; idx < N * kH1; idx += kThreads) sW1[(idx / kH1) * kLD1 + (idx % kH1)] = W1g[idx];
    for (int idx = tid; idx < kH1 * kH2; idx += kThreads) sW2[(idx / kH2) * kLD2 + (idx % kH2)] = W2g[idx];
    for (int idx = tid; idx < kH2 * kH3; idx += kThreads) sW3[(idx / kH3) * kLD3 + (idx % kH3)] = W3g[idx];
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31;  // A-operand row / environment column
    const int h = lane >> 5;  // k parity / accumulator row-half

    // lane-dependent LDS bases; everything else is a compile-time offset
    const float* w1f = sW1 + h * kLD1 + i;          // forward:  W1[2s + h][32 fb + i]
    const float* w2f = sW2 + 4 * h * kLD2 + i;      //           W2[32 kb + perm(s) + 4h][32 fb + i]
    const float* w3f = sW3 + 4 * h * kLD3 + i;      //           W3[32 kb + perm(s) + 4h][32 ob + i]
    const float* w3b = sW3 + i * kLD3 + 4 * h;      // backward: W3[32 fb + i][32 kb + perm(s) + 4h]
    const float* w2b = sW2 + i * kLD2 + 4 * h;      //           W2[32 fb + i][32 kb + perm(s) + 4h]
    const float* w1b = sW1 + (i < N ? i : 0) * kLD1 + 4 * h;  //  W1[i][32 kb + perm(s) + 4h], rows >= N are zero
    const bool w1row = i < N;

    for (int64_t tile = (int64_t)blockIdx.x * kWaves + wave; tile < ntiles; tile += (int64_t)gridDim.x * kWaves) {
        // the weights are loop invariant: without this barrier LICM hoists hundreds of LDS reads out of the
        // tile loop into registers and spills them to scratch
        asm volatile("" ::: "memory");
        const int64_t env = tile * 32 + i;
        const bool valid = env < B;
        float xs[N], e[N], z[N];
        if (valid) {
            // both lane halves read the same row (second read hits the same lines)
            const float4* rp = reinterpret_cast<const float4*>(x + env * N);
            if constexpr ((N * 4) % 16 == 0) {
#pragma unroll
                for (int q = 0; q < N / 4; ++q) {
                    const float4 v = rp[q];
                    xs[4 * q] = v.x; xs[4 * q + 1] = v.y; xs[4 * q + 2] = v.z; xs[4 * q + 3] = v.w;
                }
            } else {
                const float2* rp2 = reinterpret_cast<const float2*>(x + env * N);
#pragma unroll
                for (int q = 0; q < N / 2; ++q) {
                    const float2 v = rp2[q];
                    xs[2 * q] = v.x; xs[2 * q + 1] = v.y;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < N; ++k) xs[k] = p.xf[k];
        }
        float ee = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) e[k] = xs[k] - p.xf[k];
        sys.wrap(e);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            ee += e[k] * e[k];
            z[k] = (e[k] - p.mean[k]) / p.std[k];
        }

        // ---- layer 1: H1' (128 x 32) = W1' (128 x N) . Z' (N x 32) --------------------------------------
        f32x16 a1[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[fb][r] = 0.f;
        mfma_chain<N / 2, 4, 1>(
            a1, [&](int st, int fb) { return w1f[2 * st * kLD1 + 32 * fb]; }, [&](int st) { return h ? z[2 * st + 1] : z[2 * st]; });
        uint32_t m1[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                m |= (a1[fb][r] > 0.f ? 1u : 0u) << r;
                a1[fb][r] = fmaxf(a1[fb][r], 0.f);
            }
            m1[fb] = m;
        }

        // ---- layer 2: H2' (128 x 32) = W2' . H1' ----------------------------------------------------------
        f32x16 a2[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[fb][r] = 0.f;
        mfma_chain<64, 4, 1>(
            a2, [&](int st, int fb) { return w2f[(32 * (st >> 4) + perm(st & 15)) * kLD2 + 32 * fb]; },
            [&](int st) { return a1[st >> 4][st & 15]; });
        uint32_t m2[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                m |= (a2[fb][r] > 0.f ? 1u : 0u) << r;
                a2[fb][r] = fmaxf(a2[fb][r], 0.f);
            }
            m2[fb] = m;
        }

        // ---- layer 3: Y' (64 x 32) = W3' . H2' ------------------------------------------------------------
        f32x16 y[2];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[ob][r] = 0.f;
        mfma_chain<64, 2, 2>(
            y, [&](int st, int ob) { return w3f[(32 * (st >> 4) + perm(st & 15)) * kLD3 + 32 * ob]; },
            [&](int st) { return a2[st >> 4][st & 15]; });

        float vpart = 0.f;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vpart += y[ob][r] * y[ob][r];
                y[ob][r] = 2.f * y[ob][r];  // dV/dy
            }
        const float vsum = vpart + __shfl_xor(vpart, 32, 64);
        if (Vout && valid && h == 0) Vout[env] = vsum + p.eps_s * ee;
        if (!gout) continue;

        // ---- backward 3: dH2' (128 x 32) = W3 (128 x 64) . dY' (64 x 32), masked by h2 > 0 ------------------
        f32x16 d2[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[fb][r] = 0.f;
        mfma_chain<32, 4, 1>(
            d2, [&](int st, int fb) { return w3b[32 * fb * kLD3 + 32 * (st >> 4) + perm(st & 15)]; },
            [&](int st) { return y[st >> 4][st & 15]; });
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[fb][r] = ((m2[fb] >> r) & 1u) ? d2[fb][r] : 0.f;

        // ---- backward 2: dH1' (128 x 32) = W2 . dH2', masked by h1 > 0 -------------------------------------
        f32x16 d1[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d1[fb][r] = 0.f;
        mfma_chain<64, 4, 1>(
            d1, [&](int st, int fb) { return w2b[32 * fb * kLD2 + 32 * (st >> 4) + perm(st & 15)]; },
            [&](int st) { return d2[st >> 4][st & 15]; });
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d1[fb][r] = ((m1[fb] >> r) & 1u) ? d1[fb][r] : 0.f;

        // ---- backward 1: dZ' (N x 32, padded to 32 rows) = W1 (N x 128) . dH1' ------------------------------
        f32x16 dzv[1];
#pragma unroll
        for (int r = 0; r < 16; ++r) dzv[0][r] = 0.f;
        mfma_chain<64, 1, 4>(
            dzv, [&](int st, int) { const float wv = w1b[32 * (st >> 4) + perm(st & 15)]; return w1row ? wv : 0.f; },
            [&](int st) { return d1[st >> 4][st & 15]; });
        const f32x16 dz = dzv[0];

        // row k of dZ' sits in register (k&3) + 4(k>>3) of lane-half (k>>2)&1; gather the N rows on half 0
        float g[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            constexpr int dummy = 0;
            (void)dummy;
            const int rk = (k & 3) + 4 * (k >> 3);
            const float own = dz[rk];
            const float other = __shfl_xor(own, 32, 64);
            const float v = (((k >> 2) & 1) == 0) ? own : other;
            g[k] = v / p.std[k] + 2.f * p.eps_s * e[k];
        }
        if (valid && h == 0) {
            if constexpr ((N * 4) % 16 == 0) {
                float4* op = reinterpret_cast<float4*>(gout + env * N);
#pragma unroll
                for (int q = 0; q < N / 4; ++q) op[q] = make_float4(g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]);
            } else {
                float2* op = reinterpret_cast<float2*>(gout + env * N);
#pragma unroll
                for (int q = 0; q < N / 2; ++q) op[q] = make_float2(g[2 * q], g[2 * q + 1]);
            }
        }
    }
}

template <typename S> static int launch_value_grad(S sys, const hjbx_mlp* mlp, const float* x, float* V, float* g, int64_t B, void* st) {
    constexpr int N = S::N;
    MlpP<N> p;
    for (int k = 0; k < N; ++k) { p.mean[k] = (float)mlp->mean[k]; p.std[k] = (float)mlp->std[k]; p.xf[k] = (float)mlp->xf[k]; }
    p.eps_s = (float)mlp->eps_scalar;
    const int64_t ntiles = (B + 31) / 32;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
            return hjbx_set_error(HJBX_ENODEVICE, "hjbx_value_grad_f32: no HIP device");
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    int64_t grid = (ntiles + kWaves - 1) / kWaves;
    if (grid > n_cu) grid = n_cu;  // one resident workgroup per CU (104 KB of LDS each), waves stride over tiles
    hipLaunchKernelGGL((k_value_grad_mfma<S>), dim3((unsigned)grid), dim3(kThreads), 0, (hipStream_t)st, sys, p,
                       (const float*)mlp->W1, (const float*)mlp->W2, (const float*)mlp->W3, x, V, g, B, ntiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_value_grad_f32: %s", hipGetErrorString(e));
    return HJBX_OK;
}

extern "C" int hjbx_value_grad_f32(const hjbx_system* sys, const hjbx_mlp* mlp, const float* x, float* V, float* g, int64_t B,
                                   void* stream) {
    if (!sys || !mlp) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: NULL system or mlp descriptor");
    if (B < 0) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: negative batch size");
    if (B == 0 || (!V && !g)) return HJBX_OK;
    if (!x || !mlp->W1 || !mlp->W2 || !mlp->W3) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: NULL x or weight pointer");
    if (mlp->h1 != kH1 || mlp->h2 != kH2 || mlp->h3 != kH3)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_grad_f32: features must be [128,128,64], got [%d,%d,%d]", mlp->h1, mlp->h2,
                              mlp->h3);
    const size_t row = (size_t)sys->n * sizeof(float);
    const uintptr_t am = (row % 16 == 0) ? 15u : 7u;
    if ((reinterpret_cast<uintptr_t>(x) & am) || (g && (reinterpret_cast<uintptr_t>(g) & am)))
        return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: x / gradV must be aligned to their row vector width");
    for (int k = 0; k < sys->n; ++k)
        if (!(mlp->std[k] != 0.0)) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_grad_f32: normalization_std[%d] is zero", k);
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
}
