// hjbx_mlp_core.hpp -- device code shared by the MFMA kernels (hjbx_mlp.hip: inference; hjbx_train.hip: value-loss gradient):
// the software-pipelined f32 MFMA chain, the LDS image of the value network's weights and the fused forward + input-gradient
// of one 32-environment tile.  See the top of hjbx_mlp.hip for the design.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"

using namespace hjbx;

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

static constexpr int kH1 = 128, kH2 = 128, kH3 = 64;
static constexpr int kLD1 = 129, kLD2 = 129, kLD3 = 65;  // odd LDS row strides (floats)

// istd = 1 / normalization_std, rounded once on the host: the kernels multiply (an IEEE division is ~10 VALU instructions, 2N of
// them per tile and step otherwise)
template <int N> struct MlpP { float mean[N], istd[N], xf[N], eps_s; };

// accumulator register s of lane-half h holds row perm(s) + 4h of its 32-row block
__device__ __forceinline__ constexpr int perm(int s) { return (s & 3) + 8 * (s >> 2); }

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// relu as ONE instruction: fmaxf / fmed3f compile to a canonicalising v_max plus the v_max.  max on the raw
// bits as a signed integer is the same function (negative floats and -0.0 have the sign bit set -> 0; positive
// floats and +NaN pass through) and stays visible to the compiler's MFMA hazard padding, which an inline-asm
// v_max_f32 would not.
__device__ __forceinline__ float relu1(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

// tanh for the notebooks' networks (hjbx_mlp.activation = HJBX_ACT_TANH): 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp
// units (5 VALU ops, two of them quarter rate); absolute error ~1e-7 over the whole range (exp2 overflow -> +1, underflow -> -1),
// which is what matters for V = |y|^2 and its gradient.  The derivative comes from the value: 1 - tanh^2.
__device__ __forceinline__ float tanh1(float v) {
    const float ex = __builtin_amdgcn_exp2f(v * 2.8853900817779268f);  // exp(2x) = 2^(2x log2 e)
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(ex + 1.0f);
}
// sin / cos for the time-optimal notebook's network (hjbx_mlp.activation = HJBX_ACT_SIN), branch-free and without the stack frame of the
// library's large-argument path (the matrix-core kernels must stay free of scratch): k = rint(x 2/pi), r = x - k pi/2 by a two-constant
// Cody-Waite reduction under fma (exact for |k| < 2^12 or so: pre-activations are O(1) to O(100)), the Cephes single-precision minimax
// polynomials on [-pi/4, pi/4] (about 1 ulp), quadrant by bit operations.  Absolute error ~1e-7 for |x| < 1e3.
__device__ __forceinline__ void sincos1(float x, float& sn, float& cs) {
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.5707963705062866f, x);
    r = fmaf(k, 4.371139000186243e-08f, r);
    const float r2 = r * r;
    const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
    const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f), r2 * r2, fmaf(-0.5f, r2, 1.0f));
    // quadrant without compares (a compare per element becomes an SGPR lane mask; 64 elements x 3 of them spill): odd k swaps the two
    // polynomials through a bit-field insert under an all-ones / all-zeros mask, bits 1 of k and k + 1 go straight into the sign bits
    const uint32_t q = (uint32_t)(int)k;
    const uint32_t m = 0u - (q & 1u);
    const uint32_t us = __builtin_bit_cast(uint32_t, ps), uc = __builtin_bit_cast(uint32_t, pc);
    const uint32_t a = (uc & m) | (us & ~m), b = (us & m) | (uc & ~m);
    sn = __builtin_bit_cast(float, a ^ ((q & 2u) << 30));
    cs = __builtin_bit_cast(float, b ^ (((q + 1u) & 2u) << 30));
}
__device__ __forceinline__ float sin1(float x) { float s, c; sincos1(x, s, c); return s; }
__device__ __forceinline__ float cos1(float x) { float s, c; sincos1(x, s, c); return c; }

// ACT = hjbx_activation: the activation applied in place to a pre-activation, and the back-propagation factor d * act'(z)
// written in terms of the ACTIVATION h = act(z) (relu: [h > 0]; tanh: 1 - h^2), so no pre-activation has to be kept.
// sin (examples/double_integrator_optimal_time.ipynb cell 5): act' = cos(z) cannot be had from sin(z) (the sign is lost), so the kernels
// keep cos(z) where relu / tanh keep the activation: `h` of dact1 is then that cosine (sincos1 above).
template <int ACT> __device__ __forceinline__ float act1(float v) {
    if constexpr (ACT == HJBX_ACT_TANH) return tanh1(v);
    else if constexpr (ACT == HJBX_ACT_SIN) return sin1(v);
    else return relu1(v);
}
template <int ACT> __device__ __forceinline__ float dact1(float h, float d) {
    if constexpr (ACT == HJBX_ACT_TANH) return d - d * h * h;
    else if constexpr (ACT == HJBX_ACT_SIN) return d * h;          // h = cos(z)
    // relu: d * [h > 0] as two multiplies, the first with the clamp output modifier (v_mul_f32 ... clamp gives exactly 1 for
    // every normal h > 0, and 0 for h <= 0 or NaN).  Same op count as v_cmp + v_cndmask, but no VCC in between: that pair
    // costs an s_nop per element (VALU write of VCC -> VALU read), 192 of them per tile and step.
    else return d * fminf(fmaxf(h * 3.0e38f, 0.f), 1.f);
}

// ---- software-pipelined MFMA chain ------------------------------------------------------------------------
// One GEMM of the chain: acc[t][o] += A_o(step) x b_t(step), step = 0..NSTEPS-1, o = 0..NOUT-1 output blocks,
// t = 0..TL-1 environment tiles; every A operand is one LDS dword per lane.  hipcc sinks compiler-visible
// ds_reads down to their MFMAs and re-uses two operand registers (read -> lgkmcnt(0) -> 2 MFMAs), whatever the
// source order or sched_barrier placement.  So the reads are issued from inline asm DEPTH steps ahead and
// retired with counted s_waitcnt lgkmcnt(N) statements fenced by sched_barrier (guide 5.7, form iii; the ISA is
// audited by tools/audit_asm_loads.py): LDS returns in order, so "at most N newer operations outstanding" means this step's operands have
// landed.  Any LDS / SMEM operation the compiler adds in between only makes the count conservative.
template <int BYTE_OFF> __device__ __forceinline__ float lds_read_b32(uint32_t addr) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 65536, "ds_read_b32 offset field is 16 bits");
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(BYTE_OFF));
    return v;
}
// Counted wait + scheduling fence (guide 5.7, form iii).  The wait is the compiler-visible builtin rather than an asm
// statement: hipcc pads an s_nop between ANY inline-asm statement and a following MFMA (it must assume the statement
// wrote the MFMA's operands with a VALU op), which would cost one more issue slot per MFMA group.
// s_waitcnt simm16 on gfx9: vmcnt = bits 3:0 + 15:14, expcnt = 6:4, lgkmcnt = 11:8 -> 0xC07F leaves vmcnt/expcnt unwaited.
template <int CNT> __device__ __forceinline__ void lds_wait() {
    static_assert(CNT >= 0 && CNT <= 15, "");
    __builtin_amdgcn_s_waitcnt(0xC07F | (CNT << 8));
    __builtin_amdgcn_sched_barrier(0);
}

template <typename Off, int ST, int NOUT, int O = 0> __device__ __forceinline__ void chain_issue(float (&slot)[NOUT], uint32_t base) {
    if constexpr (O < NOUT) {
        slot[O] = lds_read_b32<Off::at(ST, O)>(base);
        chain_issue<Off, ST, NOUT, O + 1>(slot, base);
    }
}

template <typename Off, int NSTEPS, int NOUT, int DEPTH, int TL, int ST, typename GetB>
__device__ __forceinline__ void mfma_chain_step(f32x16 (&acc)[TL][NOUT], float (&ring)[DEPTH + 1][NOUT], float (&b)[2][TL], uint32_t base,
                                                GetB getB) {
    if constexpr (ST < NSTEPS) {
        // order inside a step: asm reads for step ST+DEPTH | B operands of step ST+1 (VALU) | counted wait | MFMAs of
        // step ST.  Two compiler-visible instructions sit between the last asm statement and the first MFMA, and the
        // MFMAs' B operands were written a whole step earlier, so hipcc needs no s_nop pad in front of the group.
        if constexpr (ST + DEPTH < NSTEPS) chain_issue<Off, ST + DEPTH, NOUT>(ring[(ST + DEPTH) % (DEPTH + 1)], base);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ST + 1 < NSTEPS) {
#pragma unroll
            for (int t = 0; t < TL; ++t) b[(ST + 1) & 1][t] = getB(ST + 1, t);
        }
        constexpr int ahead = (NSTEPS - 1 - ST < DEPTH ? NSTEPS - 1 - ST : DEPTH) * NOUT;  // reads issued after this step's
        float(&cur)[NOUT] = ring[ST % (DEPTH + 1)];
        lds_wait<ahead>();
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int t = 0; t < TL; ++t) acc[t][o] = MFMA(cur[o], b[ST & 1][t], acc[t][o]);
        __builtin_amdgcn_sched_barrier(0);  // keep this step's MFMAs in front of the next step's reads and wait
        mfma_chain_step<Off, NSTEPS, NOUT, DEPTH, TL, ST + 1>(acc, ring, b, base, getB);
    }
}

template <typename Off, int NSTEPS, int NOUT, int DEPTH, int TL, typename GetB>
__device__ __forceinline__ void mfma_chain(f32x16 (&acc)[TL][NOUT], float (&ring)[DEPTH + 1][NOUT], uint32_t base, GetB getB) {
    static_assert(NOUT == 1 || NOUT == 2 || NOUT == 4, "");
    static_assert(NOUT * DEPTH <= 15, "lgkmcnt is a 4-bit field");
    static_assert(DEPTH <= 3 && NSTEPS >= 1, "");
    // prologue: the first DEPTH steps' operands and the first step's B operands
    if constexpr (0 < DEPTH && 0 < NSTEPS) chain_issue<Off, 0, NOUT>(ring[0], base);
    if constexpr (1 < DEPTH && 1 < NSTEPS) chain_issue<Off, 1, NOUT>(ring[1], base);
    if constexpr (2 < DEPTH && 2 < NSTEPS) chain_issue<Off, 2, NOUT>(ring[2], base);
    float b[2][TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) b[0][t] = getB(0, t);
    mfma_chain_step<Off, NSTEPS, NOUT, DEPTH, TL, 0>(acc, ring, b, base, getB);
}

// hipcc's hazard recogniser does not look inside inline asm: an asm VALU statement (relu_mask, mask_apply, v_fma_mix) that READS an
// accumulator while the MFMA that writes it is still in flight gets no wait states and sees stale registers -- and such statements are
// pure, so the scheduler may place them directly behind the last MFMA of a chain (seen on one instantiation only, the
// planar quadrotor's f16x2 kernel: its last 32 layer-1 units were masked with half-written values).  Every chain whose results are
// consumed by asm statements is therefore followed by this barrier: enough idle issue
// slots for the last MFMA to retire (8-pass MFMA: 11 wait states, 16-pass: 19), fenced so that nothing moves across it.
template <int PASSES> __device__ __forceinline__ void mfma_results_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PASSES > 8) asm volatile("s_nop 15\n\ts_nop 7");
    else asm volatile("s_nop 15");
    __builtin_amdgcn_sched_barrier(0);
}

// byte offsets (from the lane-dependent base) of the A operand of (step, output block) for each product
struct OffW1F { static constexpr int at(int st, int fb) { return (2 * st * kLD1 + 32 * fb) * 4; } };
struct OffW2F { static constexpr int at(int st, int fb) { return ((32 * (st >> 4) + perm(st & 15)) * kLD2 + 32 * fb) * 4; } };
struct OffW3F { static constexpr int at(int st, int ob) { return ((32 * (st >> 4) + perm(st & 15)) * kLD3 + 32 * ob) * 4; } };
struct OffW3B { static constexpr int at(int st, int fb) { return (32 * fb * kLD3 + 32 * (st >> 4) + perm(st & 15)) * 4; } };
struct OffW2B { static constexpr int at(int st, int fb) { return (32 * fb * kLD2 + 32 * (st >> 4) + perm(st & 15)) * 4; } };

template <int N> struct MlpLds {
    static constexpr int NP = (N + 3) & ~3;  // W1' rows padded to whole float4s
    float W1T[kH1 * NP];                     // W1 transposed [feature][k] (16-byte aligned: first member)
    float W1[N * kLD1];
    float W2[kH1 * kLD2];
    float W3[kH2 * kLD3];
    int next;                                // next unclaimed tile group of this workgroup's range
};

template <int TL, int NOUT> __device__ __forceinline__ void zero_acc(f32x16 (&a)[TL][NOUT]) {
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
        for (int o = 0; o < NOUT; ++o)
#pragma unroll
            for (int r = 0; r < 16; ++r) a[t][o][r] = 0.f;
}

// lane-dependent LDS operand bases of one wave; everything else is a compile-time offset
struct MlpCtx {
    uint32_t w1f, w2f, w3f, w3b, w2b;
    const float4* w1t;
    int i, h;  // i = lane & 31: A-operand row / environment column; h = lane >> 5: k parity / accumulator row-half
};

template <int N, int THREADS>
__device__ __forceinline__ void mlp_fill_lds(MlpLds<N>& L, const float* __restrict__ W1g, const float* __restrict__ W2g,
                                             const float* __restrict__ W3g, int tid) {
    constexpr int NP = MlpLds<N>::NP;
    for (int idx = tid; idx < N * kH1; idx += THREADS) L.W1[(idx / kH1) * kLD1 + (idx % kH1)] = W1g[idx];
    for (int idx = tid; idx < kH1 * NP; idx += THREADS) {
        const int f = idx / NP, k = idx % NP;
        L.W1T[idx] = k < N ? W1g[k * kH1 + f] : 0.f;
    }
    for (int idx = tid; idx < kH1 * kH2; idx += THREADS) L.W2[(idx / kH2) * kLD2 + (idx % kH2)] = W2g[idx];
    for (int idx = tid; idx < kH2 * kH3; idx += THREADS) L.W3[(idx / kH3) * kLD3 + (idx % kH3)] = W3g[idx];
}

template <int N> __device__ __forceinline__ MlpCtx mlp_ctx(MlpLds<N>& L, int lane) {
    constexpr int NP = MlpLds<N>::NP;
    MlpCtx c;
    c.i = lane & 31;
    c.h = lane >> 5;
    // (the low 32 bits of a flat pointer into the LDS aperture are the LDS byte address)
    const uint32_t lds0 = (uint32_t)(uintptr_t)&L;
    c.w1f = lds0 + (uint32_t)offsetof(MlpLds<N>, W1) + 4u * (c.h * kLD1 + c.i);      // forward:  W1[2s + h][32 fb + i]
    c.w2f = lds0 + (uint32_t)offsetof(MlpLds<N>, W2) + 4u * (4 * c.h * kLD2 + c.i);  //           W2[32 kb + perm(s) + 4h][32 fb + i]
    c.w3f = lds0 + (uint32_t)offsetof(MlpLds<N>, W3) + 4u * (4 * c.h * kLD3 + c.i);  //           W3[32 kb + perm(s) + 4h][32 ob + i]
    c.w3b = lds0 + (uint32_t)offsetof(MlpLds<N>, W3) + 4u * (c.i * kLD3 + 4 * c.h);  // backward: W3[32 fb + i][32 kb + perm(s) + 4h]
    c.w2b = lds0 + (uint32_t)offsetof(MlpLds<N>, W2) + 4u * (c.i * kLD2 + 4 * c.h);  //           W2[32 fb + i][32 kb + perm(s) + 4h]
    c.w1t = reinterpret_cast<const float4*>(L.W1T + 4 * c.h * NP);                   // W1'[32 kb + perm(s) + 4h][0..NP)
    return c;
}

// V and dV/dx of the TL tiles whose state rows are in xs (one environment per lane, identical in both lane
// halves).  On return every lane holds its environment's V and (if want_grad) gradient.
template <typename S, int TL, int ACT = HJBX_ACT_RELU>
__device__ __forceinline__ void mlp_value_grad(const S& sys, const MlpP<S::N>& p, const MlpCtx& c, const float (&xs)[TL][S::N],
                                               bool want_grad, float (&V)[TL], float (&g)[TL][S::N]) {
    constexpr int N = S::N;
    constexpr int NP = MlpLds<N>::NP;
    const int h = c.h;
    float e[TL][N], z[TL][N], ee[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) {
        ee[t] = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) e[t][k] = xs[t][k] - p.xf[k];
        sys.wrap(e[t]);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            ee[t] += e[t][k] * e[t][k];
            z[t][k] = (e[t][k] - p.mean[k]) * p.istd[k];
        }
    }
    float ring4[3][4], ring2[3][2];  // operand rings of the chains (DEPTH = 2)

    // ---- layer 1: H1' (128 x 32) = W1' (128 x N) . Z' (N x 32) --------------------------------------
    f32x16 a1[TL][4];
    zero_acc(a1);
    mfma_chain<OffW1F, N / 2, 4, 2, TL>(a1, ring4, c.w1f, [&](int st, int t) { return h ? z[t][2 * st + 1] : z[t][2 * st]; });

    // Element-wise work between the products (ReLU, mask, 2y, |y|^2) is done in place on the accumulators in short
    // VALU-only passes BEFORE each chain: those overlap the SIMD partner's MFMAs, whereas every instruction issued
    // inside a chain delays this wave's next MFMA (~4 cycles each, tools/ubench/mfma_mix.hip).  Inside a chain a step
    // is then 4 ds_read_b32 + 1 counted wait + 4 MFMAs.

    // ---- layer 2: H2' (128 x 32) = W2' . relu(H1') ---------------------------------------------------
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                a1[t][fb][r] = act1<ACT>(a1[t][fb][r]);
                // sin: 64 independent polynomial evaluations -- left alone the scheduler interleaves them all and their temporaries spill
                if constexpr (ACT == HJBX_ACT_SIN) {
                    asm volatile("" : "+v"(a1[t][fb][r]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
    f32x16 a2[TL][4];
    zero_acc(a2);
    mfma_chain<OffW2F, 64, 4, 2, TL>(a2, ring4, c.w2f, [&](int st, int t) { return a1[t][st >> 4][st & 15]; });

    // ---- layer 3: Y' (64 x 32) = W3' . relu(H2') ------------------------------------------------------
    f32x16 c2[ACT == HJBX_ACT_SIN ? TL : 1][ACT == HJBX_ACT_SIN ? 4 : 1];   // sin only: cos of layer 2's pre-activations
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if constexpr (ACT == HJBX_ACT_SIN) {   // sin for layer 3; cos kept for backward 2 (a1 is dead here: layer 1 is recomputed for backward 1)
                    float sn, cs;
                    sincos1(a2[t][fb][r], sn, cs);
                    asm volatile("" : "+v"(sn), "+v"(cs));   // evaluated HERE: pure code is otherwise sunk to its uses (chain 3, backward 2) with the pre-activations kept alive
                    a2[t][fb][r] = sn;
                    c2[t][fb][r] = cs;
                    __builtin_amdgcn_sched_barrier(0);   // (see layer 2)
                } else {
                    a2[t][fb][r] = act1<ACT>(a2[t][fb][r]);  // the activation also carries act' for backward 2 (dact1)
                }
            }
    f32x16 y[TL][2];
    zero_acc(y);
    mfma_chain<OffW3F, 64, 2, 2, TL>(y, ring2, c.w3f, [&](int st, int t) { return a2[t][st >> 4][st & 15]; });

    float vpart[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) {
        f32x2 acc2{0.f, 0.f};  // packed: one v_pk_fma_f32 and one v_pk_add_f32 per two outputs
#pragma unroll
        for (int ob = 0; ob < 2; ++ob)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 yy{y[t][ob][r], y[t][ob][r + 1]};
                acc2 = __builtin_elementwise_fma(yy, yy, acc2);
                const f32x2 y2 = yy + yy;  // dV/dy
                y[t][ob][r] = y2[0];
                y[t][ob][r + 1] = y2[1];
            }
        vpart[t] = acc2[0] + acc2[1];
        V[t] = vpart[t] + __shfl_xor(vpart[t], 32, 64) + p.eps_s * ee[t];
    }
    if (!want_grad) return;

    // ---- backward 3: dH2' (128 x 32) = W3 (128 x 64) . (2 Y') -------------------------------------------
    f32x16 d2[TL][4];
    zero_acc(d2);
    mfma_chain<OffW3B, 32, 4, 2, TL>(d2, ring4, c.w3b, [&](int st, int t) { return y[t][st >> 4][st & 15]; });

    // ---- backward 2: dH1' (128 x 32) = W2 . (dH2' . [h2 > 0]) -------------------------------------------
#pragma unroll
    for (int t = 0; t < TL; ++t)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[t][fb][r] = dact1<ACT>(ACT == HJBX_ACT_SIN ? c2[ACT == HJBX_ACT_SIN ? t : 0][ACT == HJBX_ACT_SIN ? fb : 0][r] : a2[t][fb][r], d2[t][fb][r]);
    f32x16 d1[TL][4];
    zero_acc(d1);
    mfma_chain<OffW2B, 64, 4, 2, TL>(d1, ring4, c.w2b, [&](int st, int t) { return d2[t][st >> 4][st & 15]; });

    // ---- backward 1: dZ' (N x 32) = W1 (N x 128) . (dH1' . [h1 > 0]) on the VALU ---------------------------
    // Only N of an MFMA tile's 32 rows would be useful here (7.6 % of all MFMA time for n = 4); instead each
    // lane dots its 64 resident features with W1' rows (wave-uniform float4 LDS broadcasts) and the two lane
    // halves are added with one cross-half shuffle per row.  [h1 > 0] is re-derived by recomputing layer 1
    // (N/2 x 4 MFMAs, 1 % of the tile): cheaper in issue slots than carrying 128 mask bits per lane.
    zero_acc(a1);
    mfma_chain<OffW1F, N / 2, 4, 2, TL>(a1, ring4, c.w1f, [&](int st, int t) { return h ? z[t][2 * st + 1] : z[t][2 * st]; });
#pragma unroll
    for (int t = 0; t < TL; ++t) {
        f32x2 part[NP / 2];  // packed pairs: one v_pk_fma_f32 per two rows of W1
#pragma unroll
        for (int k = 0; k < NP / 2; ++k) part[k] = f32x2{0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float dv = dact1<ACT>(ACT == HJBX_ACT_RELU ? a1[t][kb][s] : ACT == HJBX_ACT_SIN ? cos1(a1[t][kb][s]) : act1<ACT>(a1[t][kb][s]), d1[t][kb][s]);
                if constexpr (ACT == HJBX_ACT_SIN) {
                    asm volatile("" : "+v"(dv));
                    __builtin_amdgcn_sched_barrier(0);
                }
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
            const float v = pk + __shfl_xor(pk, 32, 64);
            g[t][k] = v * p.istd[k] + 2.f * p.eps_s * e[t][k];
        }
    }
}

template <int N> __device__ __forceinline__ void load_row(const float* __restrict__ x, int64_t env, float (&dst)[N]) {
    if constexpr ((N * 4) % 16 == 0) {
        const float4* rp = reinterpret_cast<const float4*>(x + env * N);
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 v = rp[q];
            dst[4 * q] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
        }
    } else {
        const float2* rp2 = reinterpret_cast<const float2*>(x + env * N);
#pragma unroll
        for (int q = 0; q < N / 2; ++q) {
            const float2 v = rp2[q];
            dst[2 * q] = v.x; dst[2 * q + 1] = v.y;
        }
    }
}

template <int N> __device__ __forceinline__ void store_row(float* __restrict__ out, int64_t env, const float (&src)[N]) {
    if constexpr ((N * 4) % 16 == 0) {
        float4* op = reinterpret_cast<float4*>(out + env * N);
#pragma unroll
        for (int q = 0; q < N / 4; ++q) op[q] = make_float4(src[4 * q], src[4 * q + 1], src[4 * q + 2], src[4 * q + 3]);
    } else if constexpr ((N * 4) % 8 == 0) {
        float2* op = reinterpret_cast<float2*>(out + env * N);
#pragma unroll
        for (int q = 0; q < N / 2; ++q) op[q] = make_float2(src[2 * q], src[2 * q + 1]);
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) out[env * N + q] = src[q];
    }
}

