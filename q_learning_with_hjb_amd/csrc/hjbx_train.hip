// hjbx_train.hip -- the parameter gradient of the value-learning step (reference controller/vhjb.py:227-253, 282-284) on the
// matrix cores (chains: f32 MFMA, or f16x2 split-operand products with HJBX_OPT_MLP_ARITHMETIC = 2; outer products: f32 MFMA; since round 3 this two-kernel form is opt-in, hjbx_train_coop.hip is the default): d(sum_b hjb_loss_b)/dW and d(sum_b termination_loss_b)/dW for W1, W2, W3 of the 3-layer value network, plus the loss
// sums and the two counts, for a batch of B samples (x, cost, done).  hjb_loss depends on the weights through dV/dx, so its
// parameter gradient is a second-order reverse sweep ("double back-prop": jax.grad of a function of jax.grad in the reference).
//
// Per sample (notation of hjbx_mlp.hip; s1, s2 = act'(a1), act'(a2); ReLU, so act'' = 0):
//   forward            h1 = act(W1'z)   h2 = act(W2'h1)   y = W3'h2   V = |y|^2 + eps_s |e|^2
//   input gradient     dy = 2y   d2 = (W3 dy).s2   d1 = (W2 d2).s1   g = (W1 d1)/std + 2 eps_s e
//   losses             l_h(g; x) with q = dl_h/dg (hjb_residual_env)      l_t(V) with r = dl_t/dV (termination_residual_env)
//   reverse sweep (q)  gzb = q/std   dh1b = (W1'gzb).s1   dh2b = (W2'dh1b).s2   yb = 2 W3'dh2b   a2b = (W3 yb).s2   a1b = (W2 a2b).s1
//   hjb gradient       dW1 = gzb (x) d1 + z (x) a1b      dW2 = dh1b (x) d2 + h1 (x) a2b      dW3 = dh2b (x) dy + h2 (x) yb
//   termination grad.  dW1 = z (x) (r d1)                dW2 = h1 (x) (r d2)                 dW3 = h2 (x) (r dy)
// ((x) = outer product summed over the batch; the termination gradient is ordinary back-prop of r V, whose intermediates are r times
// the input-gradient intermediates.)
//
// Two kernels, because the two halves want opposite register layouts:
//  * k_train_chains -- every matrix-vector product above as an MFMA chain, one 32-sample tile per wave, exactly like the inference
//    kernel (features x samples: a layer's accumulators are the next chain's B operands, weights in LDS).  It writes the twelve
//    operand arrays of the outer products to a scratch buffer, 5 KB per sample, laid out as the LDS images of the second kernel.
//  * k_train_outer -- the outer products are GEMMs whose contraction index is the SAMPLE, so both operands need the feature on the
//    lane.  One workgroup per CU streams the scratch through LDS by LDS-DMA (two 80-KiB half-tile images) and its 8 waves own the 56
//    32x32 output blocks (hjb + termination sets of W1, W2, W3) as MFMA accumulators for the whole launch; per-workgroup partial
//    sums go to the workspace and k_train_reduce adds them in workgroup order (deterministic, no float atomics) into the flat
//    buffer [dW1_h | dW2_h | dW3_h | dW1_t | dW2_t | dW3_t | sum l_h, sum l_t, #interior, #done] -- the layout of the single
//    all-reduce of the data-parallel step (controller/vhjb.py: pack_flat).
// Scratch traffic (written once, read once) is 10 KB per sample against 0.36 Mflop: both kernels stay MFMA bound.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"
#include "hjbx_host.hpp"
#include "hjbx_mlp_core.hpp"
#include "hjbx_mlp_h2.hpp"
#include "hjbx_adam.hpp"

using namespace hjbx;

static constexpr int kTrWaves = 8;

// ---- scratch layout ----------------------------------------------------------------------------------------------------------
// Per 32-sample tile TWO images, one per half tile (16 samples); an image is byte for byte the LDS image k_train_outer consumes, so it
// is copied by LDS-DMA (global_load_lds_dwordx4: 80 wave-wide 1-KiB pieces, no registers, no ds_write).  A 128-feature array is 32
// groups of [16 sample slots][4 features]: group (fb, q, hh) holds features 32 fb + 8 q + 4 hh + 0..3 = the accumulator registers
// 4q..4q+3 of block fb in lane half hh, one float4 per sample.  Sample e sits in slot (e & 15) ^ (2 q + hh): with that XOR the operand
// read of k_train_outer -- lanes = 32 consecutive features (8 groups x 4), one sample -- hits 32 distinct LDS banks without padding
// (an LDS-DMA image must be lane-linear, so padding is not available: guide 5, Caveat; rule 21: swizzle on both sides).
enum { A_H1 = 0, A_DH1B, A_D2, A_A2B, A_H2, A_DH2B, A_D1, A_A1B, A_DY, A_YB, A_COUNT };
static constexpr int kGroups128 = 32, kGroups64 = 16;
static constexpr int kTileGroups = 8 * kGroups128 + 2 * kGroups64;      // 288 groups per image
static constexpr int kImgGroup = 16 * 4;                                // floats per group
static constexpr int kImgZ = kTileGroups * kImgGroup;                   // z block   [16 samples][32] (columns >= n are zero)
static constexpr int kImgG = kImgZ + 16 * 32;                           // gzb block [16 samples][32] (columns >= n are zero)
static constexpr int kImgR = kImgG + 16 * 32;                           // r [16 samples]
static constexpr int kImgDiag = kImgR + 16;                             // dV/dx [16][HJBX_MAX_N] as this kernel computed it (diagnostic only)
static constexpr int kImgFloats = 20480;                                // 81,920 bytes = 80 LDS-DMA pieces
static_assert(kImgDiag + 16 * HJBX_MAX_N <= kImgFloats, "image layout");
static constexpr int kTileFloats = 2 * kImgFloats;                      // 163,840 bytes per tile
__host__ __device__ constexpr int group0(int a) { return a < 8 ? a * kGroups128 : 8 * kGroups128 + (a - 8) * kGroups64; }

// byte offsets of this lane's four float4 slots (q = 0..3) inside a 128-byte-aligned group block: computed once per kernel, so a store
// is wave-uniform base (SGPRs) + constant + one of four 32-bit lane offsets -- no 64-bit per-lane address arithmetic per store
struct LaneSlots { uint32_t off[4]; };
__device__ __forceinline__ LaneSlots lane_slots(int e, int hh) {
    LaneSlots l;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        l.off[q] = (uint32_t)(((e >> 4) * kImgFloats + (q * 2 + hh) * kImgGroup + (((e & 15) ^ (2 * q + hh)) * 4)) * sizeof(float));
    return l;
}
// `sc`: what the values are multiplied with on their way out (a power of two: the f16x2 chains keep their accumulators scaled, 1 otherwise)
template <int NB> __device__ __forceinline__ void store_tile_array(float* __restrict__ tile, int a, const f32x16 (&v)[NB], const LaneSlots& ls, float sc = 1.0f) {
#pragma unroll
    for (int fb = 0; fb < NB; ++fb) {
        uint32_t boff = (uint32_t)((group0(a) + fb * 8) * kImgGroup * sizeof(float));
        asm volatile("" : "+s"(boff));   // keep base + offset next to its stores: formed ahead for all 36 blocks it took 80 SGPRs (spilled)
        char* blk = reinterpret_cast<char*>(tile) + boff;                               // wave uniform
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(blk + ls.off[q]) = make_float4(v[fb][4 * q] * sc, v[fb][4 * q + 1] * sc, v[fb][4 * q + 2] * sc, v[fb][4 * q + 3] * sc);
    }
}

// ReLU derivative masks: the 64 features a lane half holds per 128-feature array -> 2 words (bit 16 (fb & 1) + r of word fb >> 1),
// taken from the activations (h >= +0, so h > 0 <=> its bits are non-zero).  Two registers per layer instead of keeping the 64
// activation registers alive through the whole reverse sweep (the first version of this kernel did, and spilled 120 registers).
// Both helpers are two-instruction inline asm on purpose.  Written in C++, hipcc recognised the AND with a sign-extended bit as a
// select, turned every mask bit into a v_cmp whose 64-bit lane mask it kept in an SGPR pair for all three uses, and spilled the
// resulting 256 SGPRs to VGPR lanes: ~700 v_readlane / v_writelane / v_cndmask per tile (profiles: chains at 69 % of their MFMA time).
__device__ __forceinline__ void relu_mask_build(const f32x16 (&hv)[4], uint32_t (&m)[2]) {
    mfma_results_barrier<16>();   // hv comes straight out of an MFMA chain and is read by asm below (hjbx_mlp_core.hpp)
    m[0] = m[1] = 0u;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float hf = hv[fb][r];   // (never bit_cast a vector ELEMENT expression directly: clang 19 miscompiles it)
            uint32_t t;
            asm("v_min_u32_e32 %0, 1, %2\n\tv_lshl_or_b32 %1, %0, %3, %1" : "=&v"(t), "+v"(m[fb >> 1]) : "v"(hf), "n"(16 * (fb & 1) + r));
        }
}
__device__ __forceinline__ void relu_mask_apply(f32x16 (&v)[4], const uint32_t (&m)[2]) {
    mfma_results_barrier<16>();   // (as above)
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = v[fb][r];
            float y;   // v_bfe_i32: bit -> 0 / -1; v_and_b32 in place: no temporaries, no condition codes
            asm("v_bfe_i32 %0, %1, %2, 1\n\tv_and_b32_e32 %0, %0, %3" : "=&v"(y) : "v"(m[fb >> 1]), "n"(16 * (fb & 1) + r), "v"(x));
            v[fb][r] = y;
        }
}

// Sum over the wave, result in lane 0 (a fixed shuffle tree: deterministic).  `self` is the lane index, rebuilt by the caller at the END of
// the kernel behind an opaque asm: __shfl_down's own __lane_id() is CSE'd with the one from the top of the kernel, and keeping that value (or
// its `<< 2` / `<< 3` address forms) alive through the whole tile loop was the one register hipcc spilled in the f16x2 instantiations.
__device__ __forceinline__ double wave_sum_d(double v, int self) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int idx = ((self + off) & 63) << 2;   // lanes >= 64 - off read wrapped lanes: never on lane 0's path
        const uint64_t b = __builtin_bit_cast(uint64_t, v);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)(uint32_t)b), hi = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)(uint32_t)(b >> 32));
        v += __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    }
    return v;
}

// ---- kernel 1: all matrix-vector chains of one tile per wave --------------------------------------------------------------
// H2: the eight 128 / 64-wide products as f16x2 split-operand chains (hjbx_mlp_h2.hpp; HJBX_OPT_MLP_ARITHMETIC = 2) instead of f32 MFMA chains.
// Element-wise passes and masks work on the scaled accumulators (ReLU and its mask do not care about a positive factor); every array is
// multiplied by 2^-E of its lane (environment) when it is written to the scratch, so k_train_outer sees true values.
template <int MODE, typename S, int WAVES, bool H2>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k_train_chains(S sys_k, MlpP<S::N> p_k, TaskP<float, S::N, S::M> tk_k,
                                                                      Limits<float, S::M> lim_k, const float* __restrict__ W1g,
                                                                      const float* __restrict__ W2g, const float* __restrict__ W3g,
                                                                      const float* __restrict__ x, const float* __restrict__ cost,
                                                                      const float* __restrict__ done, float eps_term, float* __restrict__ scratch,
                                                                      double* __restrict__ sums_rec, int64_t B, int64_t ntiles) {
    constexpr int N = S::N, M = S::M, ACT = HJBX_ACT_RELU;
    constexpr int NP = MlpLds<N>::NP;
    static_assert(N % 2 == 0 && N <= 32, "state dimension");
    __shared__ __attribute__((aligned(256))) std::conditional_t<H2, MlpLdsH2<N>, MlpLds<N>> L;
    __shared__ __attribute__((aligned(16))) unsigned char sys_raw[sizeof(S)];
    S& sys_s = *reinterpret_cast<S*>(sys_raw);
    __shared__ MlpP<N> p_s;
    __shared__ TaskP<float, N, M> tk_s;
    __shared__ Limits<float, M> lim_s;
    __shared__ double red[4][WAVES];
    const int tid = threadIdx.x;
    if (tid == 0) { sys_s = sys_k; p_s = p_k; tk_s = tk_k; lim_s = lim_k; }
    if constexpr (H2) mlp_fill_lds_h2<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    else mlp_fill_lds<N, WAVES * 64>(L, W1g, W2g, W3g, tid);
    __syncthreads();
    const S& sys = sys_s;
    const MlpP<N>& p = p_s;
    const TaskP<float, N, M>& tk = tk_s;
    const Limits<float, M>& lim = lim_s;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // SGPR: tile pointers stay wave uniform (saddr stores)
    const auto c = [&] {
        if constexpr (H2) return mlp_ctx_h2<N>(L, lane);
        else return mlp_ctx<N>(L, lane);
    }();
    const int i = c.i, h = c.h;
    double acc_h = 0, acc_t = 0, acc_ni = 0, acc_nd = 0;
    // the four big products; each returns the exponent its result carries on top of its input's (0 for the f32 chains).  LDS reads run two
    // units ahead of their MFMAs (one for n = 6, whose instantiations were a register over: a spill is not an option here)
    constexpr int kDepth = N == 6 ? 1 : 2;
    uint32_t m0[2] = {0u, 0u};
    auto prod_w2f = [&](f32x16 (&out)[4], const f32x16 (&in)[4], float (&ring)[3][4]) -> int {
        if constexpr (H2) return h2_chain<H2Fwd, 4, 4, kPreNone, kDepth>(out, in, c.f2, m0) + c.kw2;
        else { f32x16 (&o1)[1][4] = reinterpret_cast<f32x16 (&)[1][4]>(out); mfma_chain<OffW2F, 64, 4, 2, 1>(o1, ring, c.w2f, [&](int st, int) { return in[st >> 4][st & 15]; }); return 0; }
    };
    auto prod_w3f = [&](f32x16 (&out)[2], const f32x16 (&in)[4], float (&ring)[3][2]) -> int {
        if constexpr (H2) return h2_chain<H2Fwd, 2, 4, kPreNone, kDepth>(out, in, c.f3, m0) + c.kw3;
        else { f32x16 (&o1)[1][2] = reinterpret_cast<f32x16 (&)[1][2]>(out); mfma_chain<OffW3F, 64, 2, 2, 1>(o1, ring, c.w3f, [&](int st, int) { return in[st >> 4][st & 15]; }); return 0; }
    };
    auto prod_w3b = [&](f32x16 (&out)[4], const f32x16 (&in)[2], float (&ring)[3][4]) -> int {
        if constexpr (H2) return h2_chain<H2Bwd, 4, 2, kPreNone, kDepth>(out, in, c.t3, m0) + c.kw3;
        else { f32x16 (&o1)[1][4] = reinterpret_cast<f32x16 (&)[1][4]>(out); mfma_chain<OffW3B, 32, 4, 2, 1>(o1, ring, c.w3b, [&](int st, int) { return in[st >> 4][st & 15]; }); return 0; }
    };
    auto prod_w2b = [&](f32x16 (&out)[4], const f32x16 (&in)[4], float (&ring)[3][4]) -> int {
        if constexpr (H2) return h2_chain<H2Bwd, 4, 4, kPreNone, kDepth>(out, in, c.t2, m0) + c.kw2;
        else { f32x16 (&o1)[1][4] = reinterpret_cast<f32x16 (&)[1][4]>(out); mfma_chain<OffW2B, 64, 4, 2, 1>(o1, ring, c.w2b, [&](int st, int) { return in[st >> 4][st & 15]; }); return 0; }
    };
    auto unscale = [](int E) { return __builtin_amdgcn_ldexpf(1.0f, -E); };
    const LaneSlots ls = lane_slots(i, h);
    // tiles are dealt wave-major over the workgroups (tile = (wave + WAVES k) gridDim + block): a small batch spreads one wave per CU
    // inputs of a tile are fetched while the previous tile is in flight (after its residual): vmcnt retires in order, so a load issued
    // behind a tile's ~160 stores could only be waited for by draining them all
    auto fetch = [&](int64_t tile, float (&xv)[N], float& dnv, float& cstv) __attribute__((always_inline)) {
        const int64_t env = tile * 32 + i;
        const bool ok = tile < ntiles && env < B;
        if (ok) load_row<N>(x, env, xv);
        else {
#pragma unroll
            for (int k = 0; k < N; ++k) xv[k] = p.xf[k];
        }
        dnv = ok ? done[env] : 0.f;
        cstv = ok ? cost[env] : 1.f;
    };
    float xs_n[N], dn_n, cst_n;
    fetch((int64_t)wave * gridDim.x + blockIdx.x, xs_n, dn_n, cst_n);
    for (int64_t tile = (int64_t)wave * gridDim.x + blockIdx.x; tile < ntiles; tile += (int64_t)WAVES * gridDim.x) {
        asm volatile("" ::: "memory");   // the weights are loop invariant: keep their LDS reads inside the loop (see hjbx_mlp.hip)
        const bool valid = tile * 32 + i < B;
        float xs[N];
#pragma unroll
        for (int k = 0; k < N; ++k) xs[k] = xs_n[k];
        float dn = dn_n, cst = cst_n;
        asm volatile("" : "+v"(dn), "+v"(cst));   // wait for them HERE (only the previous tile's oldest stores are drained), not at their first
                                                       // use after ~70 more stores of this tile: vmcnt retires in order
        float* tb = scratch + tile * (int64_t)kTileFloats;
        float e[N], z[N], ee = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) e[k] = xs[k] - p.xf[k];
        sys.wrap(e);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            ee += e[k] * e[k];
            z[k] = (e[k] - p.mean[k]) * p.istd[k];
        }
        float ring4[3][4], ring2[3][2];
        auto zB = [&](int st, int) { return h ? z[2 * st + 1] : z[2 * st]; };

        // ---- forward -------------------------------------------------------------------------------------------------
        uint32_t m1[2], m2[2];
        f32x16 a1[1][4];
        zero_acc(a1);
        mfma_chain<OffW1F, N / 2, 4, 2, 1>(a1, ring4, c.w1f, zB);
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) a1[0][fb][r] = act1<ACT>(a1[0][fb][r]);
        relu_mask_build(a1[0], m1);
        store_tile_array<4>(tb, A_H1, a1[0], ls);
        f32x16 a2[1][4];
        zero_acc(a2);
        int E = prod_w2f(a2[0], a1[0], ring4);     // (E: the exponent the current accumulators carry, per environment)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) a2[0][fb][r] = act1<ACT>(a2[0][fb][r]);
        relu_mask_build(a2[0], m2);
        store_tile_array<4>(tb, A_H2, a2[0], ls, unscale(E));
        f32x16 y[1][2];
        zero_acc(y);
        E += prod_w3f(y[0], a2[0], ring2);
        float vpart = 0.f;
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vpart += y[0][ob][r] * y[0][ob][r];
                y[0][ob][r] += y[0][ob][r];   // dy = 2y
            }
        const float V = __builtin_amdgcn_ldexpf(vpart + __shfl_xor(vpart, 32, 64), -2 * E) + p.eps_s * ee;
        store_tile_array<2>(tb, A_DY, y[0], ls, unscale(E));

        // ---- input gradient ------------------------------------------------------------------------------------------
        f32x16 d2[1][4];
        zero_acc(d2);
        E += prod_w3b(d2[0], y[0], ring4);
        relu_mask_apply(d2[0], m2);
        store_tile_array<4>(tb, A_D2, d2[0], ls, unscale(E));
        f32x16 d1[1][4];
        zero_acc(d1);
        E += prod_w2b(d1[0], d2[0], ring4);
        relu_mask_apply(d1[0], m1);
        const float un1 = unscale(E);
        store_tile_array<4>(tb, A_D1, d1[0], ls, un1);
        float g[N];
        {
            f32x2 part[NP / 2];
#pragma unroll
            for (int k = 0; k < NP / 2; ++k) part[k] = f32x2{0.f, 0.f};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    const float dv = d1[0][kb][s];
                    const f32x2 dv2{dv, dv};
#pragma unroll
                    for (int q = 0; q < NP / 4; ++q) {
                        const float4 w = c.w1t[(32 * kb + perm(s)) * (NP / 4) + q];
                        part[2 * q + 0] = __builtin_elementwise_fma(f32x2{w.x, w.y}, dv2, part[2 * q + 0]);
                        part[2 * q + 1] = __builtin_elementwise_fma(f32x2{w.z, w.w}, dv2, part[2 * q + 1]);
                    }
                }
#pragma unroll
            for (int k = 0; k < N; ++k) {
                const float pk = part[k >> 1][k & 1];
                g[k] = (pk + __shfl_xor(pk, 32, 64)) * un1 * p.istd[k] + 2.f * p.eps_s * e[k];
            }
        }

        // ---- the two losses and their derivatives w.r.t. dV/dx and V (the same device functions as hjbx_hjb_residual /
        //      hjbx_termination_residual) ---------------------------------------------------------------------------------
        float li, q[N], lt, r;
        hjb_residual_env<MODE>(sys, tk, lim, xs, g, dn, true, li, q);
        termination_residual_env<float>(eps_term, V, cst, dn, lt, r);
        if (!valid) {   // padding lanes of the last tile contribute nothing
            li = lt = r = 0.f;
#pragma unroll
            for (int k = 0; k < N; ++k) q[k] = 0.f;
        }
        if (h == 0 && valid) { acc_h += (double)li; acc_t += (double)lt; acc_ni += 1.0 - (double)dn; acc_nd += (double)dn; }
        fetch(tile + (int64_t)WAVES * gridDim.x, xs_n, dn_n, cst_n);
        float gzb[N];
#pragma unroll
        for (int k = 0; k < N; ++k) gzb[k] = q[k] * p.istd[k];
        if (h == 0) {   // per-sample operands of dW1 (zero-padded to 32 columns: unconditional operand reads in k_train_outer) and r
            float* img = tb + (i >> 4) * kImgFloats;
            const int el = i & 15;
            float4* zp = reinterpret_cast<float4*>(img + kImgZ + el * 32);
            float4* gp = reinterpret_cast<float4*>(img + kImgG + el * 32);
            auto at = [&](const float (&v)[N], int k) { return k < N ? v[k < N ? k : 0] : 0.f; };
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                zp[k] = make_float4(at(z, 4 * k), at(z, 4 * k + 1), at(z, 4 * k + 2), at(z, 4 * k + 3));
                gp[k] = make_float4(at(gzb, 4 * k), at(gzb, 4 * k + 1), at(gzb, 4 * k + 2), at(gzb, 4 * k + 3));
            }
            img[kImgR + el] = r;
#pragma unroll
            for (int k = 0; k < N; ++k) img[kImgDiag + el * HJBX_MAX_N + k] = g[k];
        }

        // ---- reverse sweep of the input gradient with the cotangent q --------------------------------------------------------
        f32x16 t1[1][4];
        zero_acc(t1);
        mfma_chain<OffW1F, N / 2, 4, 2, 1>(t1, ring4, c.w1f, [&](int st, int) { return h ? gzb[2 * st + 1] : gzb[2 * st]; });
        relu_mask_apply(t1[0], m1);
        store_tile_array<4>(tb, A_DH1B, t1[0], ls);
        f32x16 t2[1][4];
        zero_acc(t2);
        E = prod_w2f(t2[0], t1[0], ring4);
        relu_mask_apply(t2[0], m2);
        store_tile_array<4>(tb, A_DH2B, t2[0], ls, unscale(E));
        f32x16 t3[1][2];
        zero_acc(t3);
        E += prod_w3f(t3[0], t2[0], ring2);
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) t3[0][ob][rr] += t3[0][ob][rr];   // yb = 2 W3' dh2b
        store_tile_array<2>(tb, A_YB, t3[0], ls, unscale(E));
        f32x16 t4[1][4];
        zero_acc(t4);
        E += prod_w3b(t4[0], t3[0], ring4);
        relu_mask_apply(t4[0], m2);
        store_tile_array<4>(tb, A_A2B, t4[0], ls, unscale(E));
        f32x16 t5[1][4];
        zero_acc(t5);
        E += prod_w2b(t5[0], t4[0], ring4);
        relu_mask_apply(t5[0], m1);
        store_tile_array<4>(tb, A_A1B, t5[0], ls, unscale(E));
    }
    // loss sums and counts of this workgroup: lanes -> wave (shuffle tree) -> LDS -> one record (fixed order: deterministic)
    int l4;   // the lane index, rebuilt here (see wave_sum_d)
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l4));
    acc_h = wave_sum_d(acc_h, l4); acc_t = wave_sum_d(acc_t, l4); acc_ni = wave_sum_d(acc_ni, l4); acc_nd = wave_sum_d(acc_nd, l4);
    if (l4 == 0) { red[0][wave] = acc_h; red[1][wave] = acc_t; red[2][wave] = acc_ni; red[3][wave] = acc_nd; }
    __syncthreads();
    if (wave == 0 && l4 < 4) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) s += red[l4][w];
        sums_rec[4 * (int64_t)blockIdx.x + l4] = s;
    }
}

// ---- kernel 2: the outer products, summed over the samples ------------------------------------------------------------------
static constexpr int kBlocksPerSet = 4 + 16 + 8;                       // W1: 4 column blocks; W2: 4 x 4; W3: 4 x 2
static constexpr int kBlocks = 2 * kBlocksPerSet;

template <int N>
__global__ __launch_bounds__(512, 2) void k_train_outer(const float* __restrict__ scratch, float* __restrict__ partial, int64_t ntiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // two images (the ONLY LDS object of this kernel: guide 5, trap 4a)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    // LDS-DMA of one image: piece p (1 KiB) by wave p % 8; the destination is wave-uniform base + lane x 16 bytes
    auto dma = [&](int64_t tile, int hf, float* buf) __attribute__((always_inline)) {
        const float* src = scratch + tile * (int64_t)kTileFloats + hf * kImgFloats;
#pragma unroll
        for (int k = 0; k < kImgFloats / 256 / 8; ++k) {
            const int p = wave + 8 * k;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + p * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(buf + p * 256), 16, 0, 0);
        }
    };
    // operand of lane (i, kh) at k-step s: group x = i >> 2 of its 8-group block, feature c = i & 3, sample slot (2 s + kh) ^ x.
    // lo[s] is that lane offset; per half tile it is rebased once onto the image (B operands), onto this wave's row block (A operands)
    // and onto the part of the image beyond the 64-KiB reach of a ds_read offset field, so that every operand read below is
    // `ds_read_b32 v, vaddr offset:constant` with no address arithmetic of its own.
    int lo[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) lo[s] = (i >> 2) * kImgGroup + (((2 * s + kh) ^ (i >> 2)) * 4) + (i & 3);
    constexpr int kFar = group0(A_DY) * kImgGroup;                       // first float beyond 64 KiB: the dy / yb arrays and the small blocks
    static_assert(kFar * sizeof(float) == 65536 && (kImgFloats - kFar) * sizeof(float) < 65536, "ds_read offset reach");
    const int rowblk = (wave & 3) * 8 * kImgGroup;                       // this wave's row block ib = wave & 3 of dW2 / dW3

    f32x16 acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const int64_t nmine = blockIdx.x < ntiles ? (ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    const int64_t nhalf = 2 * nmine;
    if (nhalf > 0) dma(blockIdx.x, 0, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t ht = 0; ht < nhalf; ++ht) {
        const float* buf = lds + (ht & 1) * kImgFloats;
        // the next image streams into the other buffer while this one is consumed (every wave passed the barrier that ended that buffer's use)
        if (ht + 1 < nhalf) dma(blockIdx.x + ((ht + 1) >> 1) * gridDim.x, (int)((ht + 1) & 1), lds + ((ht + 1) & 1) * kImgFloats);
        if (wave < 4) {
            // waves 0-3: row block ib = wave of dW2 -- acc[0..3] hjb set, acc[4..7] termination set.  Operands of k-step s + 1 are read
            // while the 12 MFMAs of k-step s issue (two register sets, pinned by sched_barrier).
            struct Ops { float aD, aH, rr, bD[4], bA[4]; };
            auto load = [&](int s) __attribute__((always_inline)) {
                Ops o;
                const float* pa = buf + rowblk + lo[s];
                const float* pb = buf + lo[s];
                o.aD = pa[group0(A_DH1B) * kImgGroup];
                o.aH = pa[group0(A_H1) * kImgGroup];
                o.rr = (buf + kFar)[kImgR - kFar + 2 * s + kh];
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    o.bD[jb] = pb[(group0(A_D2) + jb * 8) * kImgGroup];
                    o.bA[jb] = pb[(group0(A_A2B) + jb * 8) * kImgGroup];
                }
                return o;
            };
            Ops cur = load(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                Ops nxt = cur;
                if (s + 1 < 8) nxt = load(s + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    acc[jb] = MFMA(cur.aD, cur.bD[jb], acc[jb]);
                    acc[jb] = MFMA(cur.aH, cur.bA[jb], acc[jb]);
                    acc[4 + jb] = MFMA(cur.aH, cur.rr * cur.bD[jb], acc[4 + jb]);
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        } else {
            // waves 4-7: row block ib = wave - 4 of dW3 (acc[0..1] hjb, acc[2..3] termination) and column block jb = wave - 4 of dW1
            // (acc[4] hjb, acc[5] termination; rows >= N of those blocks multiply zero operands)
            struct Ops { float aD, aH, rr, az, ag, bD[2], bY[2], b1, bA1; };
            auto load = [&](int s) __attribute__((always_inline)) {
                Ops o;
                const float* pa = buf + rowblk + lo[s];
                const float* pb = buf + lo[s];
                const float* far = buf + kFar;
                o.aD = pa[group0(A_DH2B) * kImgGroup];
                o.aH = pa[group0(A_H2) * kImgGroup];
                o.rr = far[kImgR - kFar + 2 * s + kh];
                o.az = far[kImgZ - kFar + (2 * s + kh) * 32 + i];
                o.ag = far[kImgG - kFar + (2 * s + kh) * 32 + i];
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    o.bD[jb] = (far + lo[s])[(group0(A_DY) + jb * 8) * kImgGroup - kFar];
                    o.bY[jb] = (far + lo[s])[(group0(A_YB) + jb * 8) * kImgGroup - kFar];
                }
                o.b1 = (pb + rowblk)[group0(A_D1) * kImgGroup];          // column block jb = wave - 4: the same 8-group offset as the row block
                o.bA1 = (pb + rowblk)[group0(A_A1B) * kImgGroup];
                return o;
            };
            Ops cur = load(0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                Ops nxt = cur;
                if (s + 1 < 8) nxt = load(s + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    acc[jb] = MFMA(cur.aD, cur.bD[jb], acc[jb]);
                    acc[jb] = MFMA(cur.aH, cur.bY[jb], acc[jb]);
                    acc[2 + jb] = MFMA(cur.aH, cur.rr * cur.bD[jb], acc[2 + jb]);
                }
                acc[4] = MFMA(cur.ag, cur.b1, acc[4]);
                acc[4] = MFMA(cur.az, cur.bA1, acc[4]);
                acc[5] = MFMA(cur.az, cur.rr * cur.b1, acc[5]);
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next image have landed
        __syncthreads();
    }
    // partial sums of this workgroup, raw accumulator layout [block][register][lane]
    float* out = partial + (int64_t)blockIdx.x * kBlocks * 1024;
    auto put = [&](int blk, const f32x16& a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[blk * 1024 + r * 64 + lane] = a[r];
    };
    if (wave < 4) {
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) { put(4 + wave * 4 + jb, acc[jb]); put(kBlocksPerSet + 4 + wave * 4 + jb, acc[4 + jb]); }
    } else {
        const int ib = wave - 4;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) { put(20 + ib * 2 + jb, acc[jb]); put(kBlocksPerSet + 20 + ib * 2 + jb, acc[2 + jb]); }
        put(ib, acc[4]);
        put(kBlocksPerSet + ib, acc[5]);
    }
}

// ---- kernel 3: partial sums -> flat gradient buffer, in workgroup order ------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_train_reduce(const float* __restrict__ partial, int nparts, const double* __restrict__ sums_rec, int nrec,
                                                     float* __restrict__ flat) {
    constexpr int P = N * kH1 + kH1 * kH2 + kH2 * kH3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < kBlocks * 1024) {
        // fixed order: four interleaved running sums over the workgroups (so four loads are in flight), then ((s0 + s1) + (s2 + s3))
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        int gI = 0;
        for (; gI + 4 <= nparts; gI += 4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s4[k] += partial[(int64_t)(gI + k) * kBlocks * 1024 + t];
        }
        for (int k = 0; gI < nparts; ++gI, ++k) s4[k] += partial[(int64_t)gI * kBlocks * 1024 + t];
        const float s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
        const int blk = t >> 10, reg = (t >> 6) & 15, lane = t & 63;
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
        const int set = blk / kBlocksPerSet, b = blk % kBlocksPerSet;
        float* o = flat + set * P;
        if (b < 4) {
            if (row < N) o[row * kH1 + 32 * b + col] = s;
        } else if (b < 20) {
            o[N * kH1 + (32 * ((b - 4) >> 2) + row) * kH2 + 32 * ((b - 4) & 3) + col] = s;
        } else {
            o[N * kH1 + kH1 * kH2 + (32 * ((b - 20) >> 1) + row) * kH3 + 32 * ((b - 20) & 1) + col] = s;
        }
    } else if (t < kBlocks * 1024 + 4) {
        const int k = t - kBlocks * 1024;
        double s = 0;
        for (int w = 0; w < nrec; ++w) s += sums_rec[4 * w + k];
        flat[2 * P + k] = (float)s;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------
static int device_cus() { return hjbx_device_cus(); }   // per device ordinal (hjbx_host.hpp)

struct TrainWs { size_t scratch, partial, sums, total; int64_t ntiles; int n_cu; };
static TrainWs train_ws(int64_t B) {
    TrainWs w{};
    w.n_cu = device_cus();
    if (w.n_cu <= 0) w.n_cu = 256;
    w.ntiles = (B + 31) / 32;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    w.scratch = up((size_t)w.ntiles * kTileFloats * sizeof(float));
    w.partial = up((size_t)w.n_cu * kBlocks * 1024 * sizeof(float));
    w.sums = up((size_t)w.n_cu * 4 * sizeof(double));
    w.total = w.scratch + w.partial + w.sums;
    return w;
}

template <typename S>
static int launch_train(const hjbx_system* sysh, S sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost,
                        const float* done, float* flat, void* workspace, int64_t B, void* st) {
    constexpr int N = S::N, M = S::M;
    if constexpr (N % 2 != 0 || N > 32) {
        return HJBX_EUNSUPPORTED;
    } else {
        MlpP<N> p;
        for (int k = 0; k < N; ++k) { p.mean[k] = (float)mlp->mean[k]; p.istd[k] = (float)(1.0 / mlp->std[k]); p.xf[k] = (float)mlp->xf[k]; }
        p.eps_s = (float)mlp->eps_scalar;
        const auto tk = make_task<float, N, M>(task);
        const auto lim = make_limits<float, M>(sysh);
        const TrainWs w = train_ws(B);
        if (device_cus() <= 0) return hjbx_set_error(HJBX_ENODEVICE, "hjbx_value_loss_grad_f32: no HIP device");
        float* scratch = (float*)workspace;
        float* partial = (float*)((char*)workspace + w.scratch);
        double* sums = (double*)((char*)workspace + w.scratch + w.partial);
        const int gridA = (int)(w.ntiles < w.n_cu ? w.ntiles : w.n_cu);
        const int gridB = gridA;
        const float *W1 = (const float*)mlp->W1, *W2 = (const float*)mlp->W2, *W3 = (const float*)mlp->W3;
        hipStream_t s = (hipStream_t)st;
        // the chains in the library's value-network arithmetic (HJBX_OPT_MLP_ARITHMETIC): f16x2 split-operand products, or the f32 MFMA for
        // modes 0 (f32) and 1 (bf16x3 has no training instantiation)
        auto chains = [&](auto mode_c, auto h2_c) {
            hipLaunchKernelGGL((k_train_chains<decltype(mode_c)::value, S, kTrWaves, decltype(h2_c)::value>), dim3(gridA), dim3(kTrWaves * 64), 0, s, sys, p, tk, lim,
                               W1, W2, W3, x, cost, done, (float)task->eps, scratch, sums, B, w.ntiles);
        };
        const bool h2 = hjbx_option_value(HJBX_OPT_MLP_ARITHMETIC) == 2;
        if (mode == HJBX_RESIDUAL_NORMALISED) {
            if (h2) chains(std::integral_constant<int, 0>{}, std::true_type{});
            else chains(std::integral_constant<int, 0>{}, std::false_type{});
        } else {
            if (h2) chains(std::integral_constant<int, 1>{}, std::true_type{});
            else chains(std::integral_constant<int, 1>{}, std::false_type{});
        }
        static std::atomic<bool> attr_set[kMaxDevices];   // 160 KiB of dynamic LDS need the opt-in once per kernel AND DEVICE
        const size_t lds_bytes = 2 * (size_t)kImgFloats * sizeof(float);
        const int dev = hjbx_current_device();
        if (dev < 0) return hjbx_set_error(HJBX_ENODEVICE, "hjbx_value_loss_grad_f32: no HIP device");
        if (!attr_set[dev].load(std::memory_order_relaxed)) {
            if (hipFuncSetAttribute((const void*)k_train_outer<N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess)
                return hjbx_set_error(HJBX_EHIP, "hjbx_value_loss_grad_f32: cannot reserve %zu bytes of LDS", lds_bytes);
            attr_set[dev].store(true, std::memory_order_relaxed);
        }
        hipLaunchKernelGGL((k_train_outer<N>), dim3(gridB), dim3(512), lds_bytes, s, scratch, partial, w.ntiles);
        const int nthreads = kBlocks * 1024 + 4;
        hipLaunchKernelGGL((k_train_reduce<N>), dim3((nthreads + 255) / 256), dim3(256), 0, s, partial, gridB, sums, gridA, flat);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_value_loss_grad_f32: %s", hipGetErrorString(e));
        return HJBX_OK;
    }
}

// which implementation a call takes: the cooperative single kernel (hjbx_train_coop.hip: float32 MFMA, ReLU and tanh, no scratch) unless the
// f16x2 value-network arithmetic is selected (its split-operand chains live in the round-2 pair of kernels below) or HJBX_OPT_TRAIN_KERNEL = 1
// asks for that pair (A/B measurements); tanh networks exist in the cooperative kernel only
static bool use_two_kernels(int activation) {
    if (activation == HJBX_ACT_TANH || activation == HJBX_ACT_SIN) return false;
    return hjbx_option_value(HJBX_OPT_MLP_ARITHMETIC) == 2 || hjbx_option_value(HJBX_OPT_TRAIN_KERNEL) == 1;
}

extern "C" size_t hjbx_value_loss_grad_workspace_bytes(int64_t B) {
    if (B <= 0) return 0;
    const size_t coop = hjbx_train_coop_workspace_bytes(B, HJBX_MAX_N);
    const size_t pair = use_two_kernels(HJBX_ACT_RELU) ? train_ws(B).total : 0;
    return coop > pair ? coop : pair;
}

// argument checks shared by hjbx_value_loss_grad_f32 and hjbx_value_loss_adam_f32 (B > 0)
static int check_vlg(const char* who, const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost,
                     const float* done, const void* workspace, int64_t B) {
    if (!sys || !task || !mlp) return hjbx_set_error(HJBX_EINVAL, "%s: NULL system, task or mlp descriptor", who);
    if (int rc = check_task(task)) return rc;
    if (mode != HJBX_RESIDUAL_NORMALISED && mode != HJBX_RESIDUAL_RAW) return hjbx_set_error(HJBX_EINVAL, "%s: unknown residual mode %d", who, mode);
    if (B < 0) return hjbx_set_error(HJBX_EINVAL, "%s: negative batch size", who);
    if (B == 0) return HJBX_OK;
    if (!x || !cost || !done || !workspace || !mlp->W1 || !mlp->W2 || !mlp->W3)
        return hjbx_set_error(HJBX_EINVAL, "%s: x, cost, done, workspace and the weights must be non-NULL", who);
    if (mlp->h1 != kH1 || mlp->h2 != kH2 || mlp->h3 != kH3)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "%s: features must be [128,128,64], got [%d,%d,%d]", who, mlp->h1, mlp->h2, mlp->h3);
    if (mlp->activation != HJBX_ACT_RELU && mlp->activation != HJBX_ACT_TANH && mlp->activation != HJBX_ACT_SIN)
        return hjbx_set_error(HJBX_EINVAL, "%s: unknown activation %d", who, mlp->activation);
    const size_t row = (size_t)sys->n * sizeof(float);
    const uintptr_t am = (row % 16 == 0) ? 15u : 7u;
    if ((reinterpret_cast<uintptr_t>(x) & am) || (reinterpret_cast<uintptr_t>(workspace) & 255u))
        return hjbx_set_error(HJBX_EINVAL, "%s: x must be aligned to its row vector width and workspace to 256 bytes", who);
    if ((reinterpret_cast<uintptr_t>(mlp->W1) | reinterpret_cast<uintptr_t>(mlp->W2) | reinterpret_cast<uintptr_t>(mlp->W3)) & 15u)
        return hjbx_set_error(HJBX_EINVAL, "%s: the weight matrices must be 16-byte aligned", who);
    for (int k = 0; k < sys->n; ++k)
        if (!(mlp->std[k] != 0.0)) return hjbx_set_error(HJBX_EINVAL, "%s: normalization_std[%d] is zero", who, k);
    return HJBX_OK;
}

// the round-2 pair of kernels -> flat (arguments already validated)
static int run_pair(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost, const float* done,
                    float* flat, void* workspace, int64_t B, void* stream) {
    int rc = HJBX_EUNSUPPORTED;
#ifdef HJBX_TRAIN_DEV   // development builds: cartpole only (the full set of instantiations takes minutes to compile)
    bool ok = false;
    if (sys->kind == HJBX_SYS_CARTPOLE) {
        Cartpole<float> cp{(float)sys->p[0], (float)sys->p[1], (float)sys->p[2], (float)sys->p[3]};
        rc = launch_train<Cartpole<float>>(sys, cp, task, mlp, mode, x, cost, done, flat, workspace, B, stream);
        ok = true;
    }
#else
    const bool ok = with_system<float>(sys, [&](auto S) { rc = launch_train<decltype(S)>(sys, S, task, mlp, mode, x, cost, done, flat, workspace, B, stream); });
#endif
    if (!ok || rc == HJBX_EUNSUPPORTED)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_loss_grad_f32: no kernel for system kind %d with n=%d m=%d", sys->kind, sys->n, sys->m);
    return rc;
}

extern "C" int hjbx_value_loss_grad_f32(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x,
                                        const float* cost, const float* done, float* flat, void* workspace, int64_t B, void* stream) {
    if (int rc = check_vlg("hjbx_value_loss_grad_f32", sys, task, mlp, mode, x, cost, done, workspace, B)) return rc;
    if (!flat) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_grad_f32: flat output buffer is NULL");
    const size_t P = (size_t)sys->n * kH1 + (size_t)kH1 * kH2 + (size_t)kH2 * kH3;
    if (B == 0) {
        hipError_t e = hipMemsetAsync(flat, 0, (2 * P + 4) * sizeof(float), (hipStream_t)stream);
        if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hipMemsetAsync: %s", hipGetErrorString(e));
        return HJBX_OK;
    }
    if (!use_two_kernels(mlp->activation)) return hjbx_train_coop(sys, task, mlp, mode, x, cost, done, flat, workspace, B, stream, nullptr);
    return run_pair(sys, task, mlp, mode, x, cost, done, flat, workspace, B, stream);
}

// ---- params_update in one call (vhjb.py:255-288): gradient + counts + mix + losses + Adam -------------------------------------------------
// Cooperative path: two launches (k_train_coop, then the reduction of its partial sums, the mix and the Adam step in ONE epilogue kernel; the
// flat buffer is never written).  Pair path (f16x2 chains / HJBX_OPT_TRAIN_KERNEL = 1): the flat buffer goes to the tail of the workspace and
// hjbx_mix_adam_f32's kernel follows.  Single process only: a data-parallel step needs the flat buffer for its all-reduce.
extern "C" size_t hjbx_value_loss_adam_workspace_bytes(int64_t B) {
    if (B <= 0) return 0;
    const size_t P = (size_t)HJBX_MAX_N * kH1 + (size_t)kH1 * kH2 + (size_t)kH2 * kH3;
    return ((hjbx_value_loss_grad_workspace_bytes(B) + 255) & ~(size_t)255) + (((2 * P + 4) * sizeof(float) + 255) & ~(size_t)255);
}

extern "C" int hjbx_mix_adam_f32(const float* flat, const float* reg_dev, double reg, double eps, const hjbx_adam_state* adam, float* losses,
                                 float* loss_accum, int32_t* step_counter, void* stream);
extern "C" int hjbx_replay_gather_f32(const float* buf_x, const float* buf_cost, const float* buf_done, int64_t capacity, int n, const int32_t* perm,
                                      int64_t perm_len, const int32_t* step_dev, const float* reg_table, int64_t table_len, int64_t batch, float* xs,
                                      float* costs, float* dones, float* reg_out, void* stream);

extern "C" int hjbx_value_loss_adam_f32(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost,
                                        const float* done, const float* reg_dev, double reg, double eps, const hjbx_adam_state* adam, float* losses,
                                        float* loss_accum, int32_t* step_counter, const hjbx_next_minibatch* next, void* workspace, int64_t B,
                                        void* stream) {
    if (int rc = check_vlg("hjbx_value_loss_adam_f32", sys, task, mlp, mode, x, cost, done, workspace, B)) return rc;
    if (B == 0) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_adam_f32: empty minibatch");
    FuseArgs f{};
    if (int rc = adam_args_from(adam, "hjbx_value_loss_adam_f32", f.a)) return rc;
    if (adam->param[0] != mlp->W1 || adam->param[1] != mlp->W2 || adam->param[2] != mlp->W3)
        return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_adam_f32: the Adam state's parameters must be the network's W1, W2, W3");
    const int64_t P = (int64_t)sys->n * kH1 + (int64_t)kH1 * kH2 + (int64_t)kH2 * kH3;
    if (f.a.P != P) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_adam_f32: the Adam state holds %lld parameters, the network %lld", (long long)f.a.P, (long long)P);
    f.mx = MixArgs{reg_dev, (float)reg, (float)eps, losses, loss_accum, step_counter};
    if (next) {
        if (!step_counter) return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_adam_f32: the next minibatch is numbered by step_counter, which is NULL");
        if (!next->buf_x || !next->buf_cost || !next->buf_done || !next->perm || !next->xs || !next->costs || !next->dones)
            return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_adam_f32: next minibatch: NULL buffer");
        if (next->n != sys->n || next->batch < 1 || next->capacity < 1 || next->perm_len < next->batch || next->table_len < 0 || (next->reg_out && !next->reg_table))
            return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_adam_f32: next minibatch: bad n, batch, capacity, lengths, or reg_out without reg_table");
        f.next = GatherArgs{next->buf_x, next->buf_cost, next->buf_done, next->capacity, next->n, next->perm, next->perm_len, next->reg_table, next->table_len,
                            next->batch, next->xs, next->costs, next->dones, next->reg_out, 1};
    }
    if (!use_two_kernels(mlp->activation)) return hjbx_train_coop(sys, task, mlp, mode, x, cost, done, nullptr, workspace, B, stream, &f);
    float* flat = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((hjbx_value_loss_grad_workspace_bytes(B) + 255) & ~(size_t)255));
    if (int rc = run_pair(sys, task, mlp, mode, x, cost, done, flat, workspace, B, stream)) return rc;
    if (int rc = hjbx_mix_adam_f32(flat, reg_dev, reg, eps, adam, losses, loss_accum, step_counter, stream)) return rc;
    if (!next) return HJBX_OK;     // (the counter has been incremented by the launch above: the gather below reads the next index from it)
    return hjbx_replay_gather_f32(next->buf_x, next->buf_cost, next->buf_done, next->capacity, next->n, next->perm, next->perm_len, step_counter, next->reg_table,
                                  next->table_len, next->batch, next->xs, next->costs, next->dones, next->reg_out, stream);
}
