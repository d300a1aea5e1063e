// hjbx_mlp_h2.hpp -- the value network's forward + input gradient with every float32 operand represented by TWO float16 pieces
// (x 2^k = hi + lo + r: hi = f16(x 2^k), lo = f16(x 2^k - hi), round to nearest, |r| <= 2^-22 |x| 2^k) and the three largest piece
// products on the f16 matrix cores (v_mfma_f32_32x32x16_f16, float32 accumulation):
//
//   x w  =  hi hi + hi lo + lo hi  +  [lo lo, r w, x r': dropped, together <= 3 x 2^-22 |x w|, typically ~2^-23]
//
// Half the matrix-pipe work of the bf16x3 scheme (hjbx_mlp_x3.hpp) and two thirds of its LDS image, at ~4x the rounding of a float32
// product in the worst case (measured: indistinguishable) -- 14x inside the 1e-5 tolerance of the path.  OPT-IN (HJBX_OPT_MLP_ARITHMETIC = 2;
// the library default is the f32 MFMA = the reference's arithmetic; this mode was the default for the second half of round 2); every
// per-element parity test runs in it, in bf16x3 and in the f32 MFMA mode against the same bounds, and a device property test holds it to
// |delta| <= c 2^-22 (sum of the element's |terms|) + 2^-39 (scaling maximum) against mode 0 on the same inputs.
//
// float16 has a 5-bit exponent, so every operand is scaled by a power of two first (exact):
//  * weights: one exponent per matrix, chosen when the LDS image is built, so that max |w| 2^kw is in [2^12, 2^13);
//  * activations / back-propagated values: one exponent PER ENVIRONMENT and product, from the largest magnitude among the environment's
//    inputs of that product (a v_max3 pass over the accumulators + one cross-half exchange), so that it lands in [2^13, 2^14).  A column of
//    the B operand is one environment, so its scale factors out of the product: the accumulators simply carry a per-lane exponent E
//    (relu and its mask do not care), which is taken out once at the end (V: 2^-2E, dV/dx: 2^-E).
//  Values more than 2^17 below the largest one of their environment lose low bits of their lo piece (float16 subnormals): an ABSOLUTE
//  error of 2^-39 of that largest value.
// Layout, chain structure, masks: as hjbx_mlp_x3.hpp (one swizzled [output][input] image per piece, row reads forward, transposed reads
// backward, element-wise work inside the consuming chain).
#pragma once
#include "hjbx_mlp_x3.hpp"

using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;

#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)

static constexpr int kH2Half = 2 * kImgPiece;  // [piece][64 rows]
static constexpr int kH2W2Img = 2 * kH2Half;   // 64 KiB: every offset fits the 16-bit field of one lane base
static constexpr int kH2W3Img = kH2Half;

__host__ __device__ constexpr int h2_img_off(int piece, int row, int c4) {
    return (row >> 6) * kH2Half + piece * kImgPiece + (row & 63) * kImgRow + 16 * ((c4 >> 1) ^ img_sw(row)) + 8 * ((c4 & 1) ^ ((row >> 4) & 1));
}

template <int N> struct MlpLdsH2 {
    static constexpr int NP = (N + 3) & ~3;
    unsigned char W2i[kH2W2Img];  // first member, 256-byte aligned
    unsigned char W3i[kH2W3Img];
    float W1T[kH1 * NP];
    float W1[N * kLD1];
    int next;
    unsigned wmax[2];             // bit patterns of max |W2|, max |W3| (fill), then
    int kw[2];                    // the weight exponents of W2, W3
};

// {f16(a), f16(b)} round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, f16x2));
}
// pieces of the pair (x0 s, x1 s): packed {x0 piece, x1 piece}.  The products with the power of two s and the residuals are exact.
// Six instructions (2 v_mul, v_cvt_pk_f16_f32, 2 v_fma_mix_f32, v_cvt_pk_f16_f32).  Measured alternatives, cartpole B = 2^20: the same with
// the v_fma_mix as inline asm 0.2662 ms / step (this: 0.2625); v_cvt_f32_f16 + v_fma_f32 instead of each v_fma_mix 0.2676; four
// instructions with v_fma_mixlo_f16 / v_fma_mixhi_f16 (fma and round to float16 in one) 0.2839 -- fewer but slower instructions.
__device__ __forceinline__ void h2_split_pair(float x0, float x1, float s, uint32_t& hi, uint32_t& lo) {
    hi = cvt_pk_f16(x0 * s, x1 * s);
    const f16x2 h = __builtin_bit_cast(f16x2, hi);
    lo = cvt_pk_f16(__builtin_fmaf(x0, s, -(float)h[0]), __builtin_fmaf(x1, s, -(float)h[1]));   // hipcc: one v_fma_mix_f32 each
}

__device__ __forceinline__ float pow2f(int k) { return __builtin_bit_cast(float, (uint32_t)(127 + k) << 23); }  // -126 <= k <= 127
// exponent that brings a largest magnitude mx into [2^(T-1), 2^T)
template <int T> __device__ __forceinline__ int scale_exponent(float mx) {
    int k = T - __builtin_amdgcn_frexp_expf(mx);  // mx = f 2^e, f in [0.5, 1); frexp_exp(0) = 0
    return k > 100 ? 100 : (k < -100 ? -100 : k);
}

template <int NO, int THREADS> __device__ __forceinline__ unsigned weight_absmax_bits(const float* __restrict__ Wg, int tid) {
    float m = 0.f;
    for (int idx = tid; idx < NO * 128; idx += THREADS) m = fmaxf(m, fabsf(Wg[idx]));
    return __builtin_bit_cast(unsigned, m);  // non-negative floats order like their bit patterns
}

template <int NO, int THREADS> __device__ __forceinline__ void h2_fill_image(unsigned char* img, const float* __restrict__ Wg, float s, int tid) {
    for (int idx = tid; idx < NO * 32; idx += THREADS) {
        const int fo = idx % NO, c4 = idx / NO;
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = Wg[(4 * c4 + e) * NO + fo];
        uint32_t h0, l0, h1, l1;
        h2_split_pair(w[0], w[1], s, h0, l0);
        h2_split_pair(w[2], w[3], s, h1, l1);
        *reinterpret_cast<uint2*>(img + h2_img_off(0, fo, c4)) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(img + h2_img_off(1, fo, c4)) = make_uint2(l0, l1);
    }
}

// NOTE: contains two __syncthreads(): every thread of the workgroup must call it
template <int N, int THREADS>
__device__ __forceinline__ void mlp_fill_lds_h2(MlpLdsH2<N>& L, const float* __restrict__ W1g, const float* __restrict__ W2g,
                                                const float* __restrict__ W3g, int tid) {
    constexpr int NP = MlpLdsH2<N>::NP;
    if (tid < 2) L.wmax[tid] = 0u;
    __syncthreads();
    for (int idx = tid; idx < N * kH1; idx += THREADS) L.W1[(idx / kH1) * kLD1 + (idx % kH1)] = W1g[idx];
    for (int idx = tid; idx < kH1 * NP; idx += THREADS) {
        const int f = idx / NP, k = idx % NP;
        L.W1T[idx] = k < N ? W1g[k * kH1 + f] : 0.f;
    }
    atomicMax(&L.wmax[0], weight_absmax_bits<kH2, THREADS>(W2g, tid));
    atomicMax(&L.wmax[1], weight_absmax_bits<kH3, THREADS>(W3g, tid));
    __syncthreads();
    const int k2 = scale_exponent<13>(__builtin_bit_cast(float, L.wmax[0])), k3 = scale_exponent<13>(__builtin_bit_cast(float, L.wmax[1]));
    h2_fill_image<kH2, THREADS>(L.W2i, W2g, pow2f(k2), tid);
    h2_fill_image<kH3, THREADS>(L.W3i, W3g, pow2f(k3), tid);
    if (tid == 0) { L.kw[0] = k2; L.kw[1] = k3; }
}

struct MlpCtxH2 {
    uint32_t w1f;
    const float4* w1t;
    int i, h;
    uint32_t f2, f3, t2, t3;  // lane bases: row reads / transposed reads of the W2 and W3 images
    int kw2, kw3;
};

// (call after the barrier that follows mlp_fill_lds_h2)
template <int N> __device__ __forceinline__ MlpCtxH2 mlp_ctx_h2(MlpLdsH2<N>& L, int lane) {
    constexpr int NP = MlpLdsH2<N>::NP;
    MlpCtxH2 c;
    c.i = lane & 31;
    c.h = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&L;
    c.w1f = lds0 + (uint32_t)offsetof(MlpLdsH2<N>, W1) + 4u * (c.h * kLD1 + c.i);
    c.w1t = reinterpret_cast<const float4*>(L.W1T + 4 * c.h * NP);
    const uint32_t r = c.i, h = c.h;
    const uint32_t fwd = r * kImgRow + 16u * (uint32_t)img_sw((int)r) + 8u * (h ^ (r >> 4));
    const uint32_t q = (lane >> 2) & 3, pp = lane & 3, G1 = (lane >> 4) & 1;
    const uint32_t trd = (4u * h + q) * kImgRow + 16u * ((q << 2) | ((2u * G1 + (pp >> 1)) ^ h)) + 8u * (pp & 1);
    const uint32_t w2 = lds0 + (uint32_t)offsetof(MlpLdsH2<N>, W2i), w3 = lds0 + (uint32_t)offsetof(MlpLdsH2<N>, W3i);
    c.f2 = w2 + fwd; c.f3 = w3 + fwd;
    c.t2 = w2 + trd; c.t3 = w3 + trd;
    c.kw2 = L.kw[0]; c.kw3 = L.kw[1];
    return c;
}

struct H2Fwd {
    static constexpr bool kTr = false;
    static constexpr int xorc(int K, int O, int b) { return (2 * K + b) << 4; }
    static constexpr int imm(int piece, int K, int O, int b) { return (O >> 1) * kH2Half + piece * kImgPiece + (O & 1) * 32 * kImgRow; }
};
struct H2Bwd {
    static constexpr bool kTr = true;
    static constexpr int xorc(int K, int O, int b) { return (O << 6) | (b << 5) | ((K & 1) << 3); }
    static constexpr int imm(int piece, int K, int O, int b) {
        return (K >> 2) * kH2Half + piece * kImgPiece + (((K >> 1) & 1) * 32 + (K & 1) * 16 + b * 8) * kImgRow;
    }
};

struct H2Ops { u32x2 r[2][2]; };  // [piece][b]

template <typename Dir, int K, int O> __device__ __forceinline__ void h2_issue(H2Ops& s, uint32_t base, uint32_t& root) {
    if constexpr (O == 0) {  // (see x3_issue: keeps hipcc from hoisting ~50 loop-invariant addresses into scratch)
        root = base;
        asm volatile("" : "+v"(root));
    }
    const uint32_t a0 = root ^ (uint32_t)Dir::xorc(K, O, 0), a1 = root ^ (uint32_t)Dir::xorc(K, O, 1);
    if constexpr (Dir::kTr) {
        s.r[0][0] = lds_read_tr16_b64<Dir::imm(0, K, O, 0)>(a0); s.r[0][1] = lds_read_tr16_b64<Dir::imm(0, K, O, 1)>(a1);
        s.r[1][0] = lds_read_tr16_b64<Dir::imm(1, K, O, 0)>(a0); s.r[1][1] = lds_read_tr16_b64<Dir::imm(1, K, O, 1)>(a1);
    } else {
        s.r[0][0] = lds_read_b64<Dir::imm(0, K, O, 0)>(a0); s.r[0][1] = lds_read_b64<Dir::imm(0, K, O, 1)>(a1);
        s.r[1][0] = lds_read_b64<Dir::imm(1, K, O, 0)>(a0); s.r[1][1] = lds_read_b64<Dir::imm(1, K, O, 1)>(a1);
    }
}

template <int NIN, int K, int JJ, int PRE>
__device__ __forceinline__ void h2_split(const f32x16 (&in)[NIN], float s, uint32_t (&Bp)[2][4], uint32_t (&m)[2]) {
    constexpr int kb = K >> 1, r0 = 8 * (K & 1) + 2 * JJ;
    float x0 = in[kb][r0], x1 = in[kb][r0 + 1];
    if constexpr (PRE == kPreReluMask) {
        relu_mask<16 * (kb & 1) + r0>(x0, m[kb >> 1]);
        relu_mask<16 * (kb & 1) + r0 + 1>(x1, m[kb >> 1]);
    } else if constexpr (PRE == kPreMaskApply) {
        x0 = mask_apply<16 * (kb & 1) + r0>(x0, m[kb >> 1]);
        x1 = mask_apply<16 * (kb & 1) + r0 + 1>(x1, m[kb >> 1]);
    }
    h2_split_pair(x0, x1, s, Bp[0][JJ], Bp[1][JJ]);
}

// The LDS reads of a unit are issued kH2Depth units (3 x 96 matrix-pipe cycles) ahead of its MFMAs: with eight waves reading 85 B / clk
// from the CU's LDS a read took longer than the two units of the first version to come back (SQ_WAIT_INST_ANY: 48 % of the wave cycles).
static constexpr int kH2Depth = 3;   // x 4 reads per unit <= 15 (lgkmcnt is a 4-bit counter)

template <typename Dir, int NK, int NOUT, int NIN, int PRE, int DEPTH, int U>
__device__ __forceinline__ void h2_unit(f32x16 (&out)[NOUT], const f32x16 (&in)[NIN], uint32_t base, float s, H2Ops (&ring)[DEPTH + 1],
                                        uint32_t (&Bp)[2][2][4], uint32_t& root, uint32_t (&m)[2]) {
    constexpr int NU = NK * NOUT;
    if constexpr (U < NU) {
        constexpr int K = U / NOUT, O = U % NOUT;
        if constexpr (U + DEPTH < NU)
            h2_issue<Dir, (U + DEPTH) / NOUT, (U + DEPTH) % NOUT>(ring[(U + DEPTH) % (DEPTH + 1)], base, root);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (K + 1 < NK) {
            constexpr int PP = 4 / NOUT;
            h2_split<NIN, K + 1, O * PP, PRE>(in, s, Bp[(K + 1) & 1], m);
            if constexpr (PP == 2) h2_split<NIN, K + 1, O * PP + 1, PRE>(in, s, Bp[(K + 1) & 1], m);
        }
        constexpr int ahead = (NU - 1 - U < DEPTH ? NU - 1 - U : DEPTH) * 4;
        lds_wait<ahead>();
        const H2Ops& o = ring[U % (DEPTH + 1)];
        const u32x4 ah{o.r[0][0][0], o.r[0][0][1], o.r[0][1][0], o.r[0][1][1]};
        const u32x4 al{o.r[1][0][0], o.r[1][0][1], o.r[1][1][0], o.r[1][1][1]};
        const uint32_t(&B)[2][4] = Bp[K & 1];
        const u32x4 bh{B[0][0], B[0][1], B[0][2], B[0][3]}, bl{B[1][0], B[1][1], B[1][2], B[1][3]};
        f32x16 acc = out[O];
        acc = MFMA_F16(al, bh, acc);
        acc = MFMA_F16(ah, bl, acc);
        acc = MFMA_F16(ah, bh, acc);
        out[O] = acc;
        __builtin_amdgcn_sched_barrier(0);
        h2_unit<Dir, NK, NOUT, NIN, PRE, DEPTH, U + 1>(out, in, base, s, ring, Bp, root, m);
    }
}

// Largest magnitude among the inputs of a product, over both lane halves of the environment (RELU: only positive values survive PRE)
template <int NIN, bool RELU> __device__ __forceinline__ float h2_absmax(const f32x16 (&in)[NIN]) {
    float mx = 0.f;
#pragma unroll
    for (int b = 0; b < NIN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = RELU ? fmaxf(mx, in[b][r]) : fmaxf(mx, fabsf(in[b][r]));
    return fmaxf(mx, __shfl_xor(mx, 32, 64));
}

// out[o] = 2^(k + kw) x (W-image product of PRE(in)); returns k (the caller adds k + kw to the lane's exponent).  DEPTH: how many units
// ahead the LDS reads run (8 registers each; the training kernel, tighter on registers, uses 2)
template <typename Dir, int NOUT, int NIN, int PRE, int DEPTH = kH2Depth>
__device__ __forceinline__ int h2_chain(f32x16 (&out)[NOUT], const f32x16 (&in)[NIN], uint32_t base, uint32_t (&m)[2]) {
    constexpr int NK = 2 * NIN;
    static_assert(NOUT == 2 || NOUT == 4, "");
    const int k = scale_exponent<14>(h2_absmax<NIN, PRE == kPreReluMask>(in));
    const float s = pow2f(k);
    static_assert(DEPTH >= 1 && DEPTH <= 3, "");
    H2Ops ring[DEPTH + 1];
    uint32_t Bp[2][2][4];
    uint32_t root;
    h2_issue<Dir, 0, 0>(ring[0], base, root);
    if constexpr (DEPTH > 1) h2_issue<Dir, 1 / NOUT, 1 % NOUT>(ring[1], base, root);
    if constexpr (DEPTH > 2) h2_issue<Dir, 2 / NOUT, 2 % NOUT>(ring[2], base, root);
    __builtin_amdgcn_sched_barrier(0);
    h2_split<NIN, 0, 0, PRE>(in, s, Bp[0], m);
    h2_split<NIN, 0, 1, PRE>(in, s, Bp[0], m);
    h2_split<NIN, 0, 2, PRE>(in, s, Bp[0], m);
    h2_split<NIN, 0, 3, PRE>(in, s, Bp[0], m);
    h2_unit<Dir, NK, NOUT, NIN, PRE, DEPTH, 0>(out, in, base, s, ring, Bp, root, m);
    mfma_results_barrier<8>();
    return k;
}

template <int NB> __device__ __forceinline__ void zero_blocks(f32x16 (&a)[NB]) {
#pragma unroll
    for (int o = 0; o < NB; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) a[o][r] = 0.f;
}

// V and dV/dx of one tile of 32 environments (ReLU network); same contract as mlp_value_grad with TL = 1
template <typename S>
__device__ __forceinline__ void mlp_value_grad_h2(const S& sys, const MlpP<S::N>& p, const MlpCtxH2& c, const float (&xs)[1][S::N], bool want_grad,
                                                  float (&V)[1], float (&g)[1][S::N]) {
    constexpr int N = S::N;
    constexpr int NP = MlpLdsH2<N>::NP;
    const int h = c.h;
    float e[N], z[1][N], ee = 0.f;
#pragma unroll
    for (int k = 0; k < N; ++k) e[k] = xs[0][k] - p.xf[k];
    sys.wrap(e);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        ee += e[k] * e[k];
        z[0][k] = (e[k] - p.mean[k]) * p.istd[k];
    }
    float ring4[3][4];
    f32x16 a1[1][4];
    zero_acc(a1);
    mfma_chain<OffW1F, N / 2, 4, 2, 1>(a1, ring4, c.w1f, [&](int st, int t) { return h ? z[t][2 * st + 1] : z[t][2 * st]; });
    mfma_results_barrier<16>();
    uint32_t m1[2] = {0u, 0u}, m2[2] = {0u, 0u}, m0[2] = {0u, 0u};
    int E = 0;  // the accumulators below hold 2^E x their true values (per environment)
    f32x16 a2[4];
    zero_blocks(a2);
    E += h2_chain<H2Fwd, 4, 4, kPreReluMask>(a2, a1[0], c.f2, m1) + c.kw2;
    f32x16 y[2];
    zero_blocks(y);
    E += h2_chain<H2Fwd, 2, 4, kPreReluMask>(y, a2, c.f3, m2) + c.kw3;
    float vpart = 0.f;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) vpart = fmaf(y[ob][r], y[ob][r], vpart);
    V[0] = __builtin_amdgcn_ldexpf(vpart + __shfl_xor(vpart, 32, 64), -2 * E) + p.eps_s * ee;
    if (!want_grad) return;
    f32x16 d2[4];
    zero_blocks(d2);
    E += h2_chain<H2Bwd, 4, 2, kPreNone>(d2, y, c.t3, m0) + c.kw3;
    f32x16 d1[4];
    zero_blocks(d1);
    E += h2_chain<H2Bwd, 4, 4, kPreMaskApply>(d1, d2, c.t2, m2) + c.kw2;
    mask_apply_block<0, 0>(d1, m1);
    // backward 1 on the VALU: each lane dots its 64 resident features with W1' rows (LDS broadcasts).  The rows are fetched GRP features at a
    // time (one wait per group): read-wait-use per feature left ~100 cycles of LDS latency exposed 64 times a tile.  (Fetching a group AHEAD of
    // the FMAs as well cost 64 - 96 more registers and spilled in the n = 10 and m = 2 rollout kernels.)
    f32x2 part[NP / 2];
#pragma unroll
    for (int k = 0; k < NP / 2; ++k) part[k] = f32x2{0.f, 0.f};
    constexpr int GRP = NP <= 4 ? 8 : 4, NG = 64 / GRP;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        float4 wbuf[GRP][NP / 4];
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
            const int f = gi * GRP + j;
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) wbuf[j][q] = c.w1t[(32 * (f >> 4) + perm(f & 15)) * (NP / 4) + q];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < GRP; ++j) {
            const int f = gi * GRP + j;
            const float dv = d1[f >> 4][f & 15];
            const f32x2 dv2{dv, dv};
#pragma unroll
            for (int q = 0; q < NP / 4; ++q) {
                const float4 w = wbuf[j][q];
                part[2 * q + 0] = __builtin_elementwise_fma(f32x2{w.x, w.y}, dv2, part[2 * q + 0]);
                part[2 * q + 1] = __builtin_elementwise_fma(f32x2{w.z, w.w}, dv2, part[2 * q + 1]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const float pk = part[k >> 1][k & 1];
        const float v = __builtin_amdgcn_ldexpf(pk + __shfl_xor(pk, 32, 64), 1 - E);  // dV/dy = 2 y: the products ran on y
        g[0][k] = v * p.istd[k] + 2.f * p.eps_s * e[k];
    }
}
