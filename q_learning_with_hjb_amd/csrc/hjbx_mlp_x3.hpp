// hjbx_mlp_x3.hpp -- the value network's forward + input gradient with every float32 operand split EXACTLY into three bfloat16 pieces
// (x = hi + mid + lo, hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid; round to nearest: |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|, every
// residual exact in float32) and the six largest of the nine piece products accumulated on the bf16 matrix cores
// (v_mfma_f32_32x32x16_bf16, float32 accumulation):
//
//   x w  =  hi hi + (hi mid + mid hi) + (hi lo + mid mid + lo hi)  +  [mid lo + lo mid + lo lo: dropped, <= 2^-23 |x w|]
//
// Every kept product of two 8-bit significands is exact in float32, so the only differences from the float32 FMA chain of hjbx_mlp.hip are the
// three dropped products (relative size <= 2^-23 per term, next to the 2^-24 rounding of each float32 FMA) and the accumulation order.  Six
// bf16 MFMAs do 16 k-steps in 6 x 32 cycles where the f32 MFMA needs 8 x 64: 2.7x less matrix-pipe time per product.  This is an
// OPT-IN arithmetic (HJBX_OPT_MLP_ARITHMETIC = 1) with its own per-element parity tests (tests/test_gpu_f32_parity.py).
//
// Layout (see hjbx_mlp.hip for the transposed features x environments formulation, which is kept):
//  * The accumulator registers of one layer are the B operands of the next: lane half h, registers 8s .. 8s+7 of a 32-feature block hold
//    features 16 s + 8 (j >> 2) + 4 h + (j & 3), j = 0..7, which is one lane's k-slice of a 32x32x16 MFMA (guide 3, "accumulator tile as
//    the next MFMA's operand").  The registers are split into pieces on the fly, one k-step ahead of their MFMAs (5.5 VALU ops / element).
//  * ONE bf16 image per weight matrix and piece in LDS, [output feature][input feature], serves both directions: the forward product
//    (sum over input features) reads a lane's row with two ds_read_b64 (columns 4 h .. 4 h + 3 and 8 + 4 h .. of the k-step), the
//    backward product (sum over output features) reads the same image COLUMN-wise with ds_read_b64_tr_b16 (guide T10).  Rows are 256 B;
//    the 16-byte chunks of a row are XOR-swizzled with ((row & 3) << 2 | (row >> 2) & 3) and the 8-byte halves with (row >> 4) & 1, which
//    makes both kinds of read bank-conflict free (32 lanes x 8 B hit 32 distinct 8-byte slots of the 256-byte bank line).
//  * 3 pieces x (128 x 128 + 64 x 128) x 2 B = 144 KiB of the CU's 160 KiB; layer 1 (k = n <= 10) and the last backward product stay on the
//    f32 paths of hjbx_mlp_core.hpp (their weights are 2 - 11 KB as float32).
//  * ReLU derivatives are kept as bit masks (64 bits per lane and layer) instead of being re-derived from the activations: the
//    activations' registers are needed for the operand rings.
#pragma once
#include "hjbx_mlp_core.hpp"

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) uint32_t;
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

static constexpr int kImgRow = 256;               // bytes per image row: 128 input features x 2 B
static constexpr int kImgPiece = 64 * kImgRow;    // one piece of 64 rows
static constexpr int kImgHalf = 3 * kImgPiece;    // [piece][64 rows]: everything a lane base + 16-bit offset field has to reach
static constexpr int kW2Img = 2 * kImgHalf;       // 128 output features
static constexpr int kW3Img = kImgHalf;           // 64 output features

__host__ __device__ constexpr int img_sw(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// byte offset of the 8-byte slot that holds input features 4 c4 .. 4 c4 + 3 of output feature `row`, piece `piece`
__host__ __device__ constexpr int img_off(int piece, int row, int c4) {
    return (row >> 6) * kImgHalf + piece * kImgPiece + (row & 63) * kImgRow + 16 * ((c4 >> 1) ^ img_sw(row)) + 8 * ((c4 & 1) ^ ((row >> 4) & 1));
}

template <int N> struct MlpLdsX3 {
    static constexpr int NP = (N + 3) & ~3;
    unsigned char W2i[kW2Img];  // first member: the struct is declared with 256-byte alignment (the swizzle XORs address bits 3..7)
    unsigned char W3i[kW3Img];
    float W1T[kH1 * NP];
    float W1[N * kLD1];
    int next;
};

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
// {bf16(a) in bits 15:0, bf16(b) in bits 31:16}, round to nearest even: one v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// pieces of a pair of float32 values: x = hi + mid + lo EXACTLY (hi = bf16(x), mid = bf16(x - hi), lo = x - hi - mid: the residuals are
// exact in float32 and the last one has at most 8 significant bits), |mid| <= 2^-8 |x|, |lo| <= 2^-16 |x|; packed {x0 piece, x1 piece}
__device__ __forceinline__ void x3_split_pair(float x0, float x1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    hi = cvt_pk_bf16(x0, x1);
    const float r0 = x0 - __builtin_bit_cast(float, hi << 16), r1 = x1 - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = cvt_pk_bf16(r0, r1);
    const float s0 = r0 - __builtin_bit_cast(float, mid << 16), s1 = r1 - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = cvt_pk_bf16(s0, s1);
}

// W (global, [input feature][output feature], row length NO) -> three swizzled bf16 images [output][input]
template <int NO, int THREADS> __device__ __forceinline__ void fill_image(unsigned char* img, const float* __restrict__ Wg, int tid) {
    for (int idx = tid; idx < NO * 32; idx += THREADS) {
        const int fo = idx % NO, c4 = idx / NO;  // consecutive threads read consecutive output features: coalesced
        float w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = Wg[(4 * c4 + e) * NO + fo];
        uint32_t h0, m0, l0, h1, m1, l1;
        x3_split_pair(w[0], w[1], h0, m0, l0);
        x3_split_pair(w[2], w[3], h1, m1, l1);
        *reinterpret_cast<uint2*>(img + img_off(0, fo, c4)) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(img + img_off(1, fo, c4)) = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(img + img_off(2, fo, c4)) = make_uint2(l0, l1);
    }
}

template <int N, int THREADS>
__device__ __forceinline__ void mlp_fill_lds_x3(MlpLdsX3<N>& L, const float* __restrict__ W1g, const float* __restrict__ W2g,
                                                const float* __restrict__ W3g, int tid) {
    constexpr int NP = MlpLdsX3<N>::NP;
    for (int idx = tid; idx < N * kH1; idx += THREADS) L.W1[(idx / kH1) * kLD1 + (idx % kH1)] = W1g[idx];
    for (int idx = tid; idx < kH1 * NP; idx += THREADS) {
        const int f = idx / NP, k = idx % NP;
        L.W1T[idx] = k < N ? W1g[k * kH1 + f] : 0.f;
    }
    fill_image<kH2, THREADS>(L.W2i, W2g, tid);
    fill_image<kH3, THREADS>(L.W3i, W3g, tid);
}

struct MlpCtxX3 {
    uint32_t w1f;
    const float4* w1t;
    int i, h;
    uint32_t f2[2], f3[2];  // forward (row-read) lane bases: W2 image halves 0 / 1; W3 image (one half, twice)
    uint32_t t2[2], t3[2];  // transposed-read lane bases
};

template <int N> __device__ __forceinline__ MlpCtxX3 mlp_ctx_x3(MlpLdsX3<N>& L, int lane) {
    constexpr int NP = MlpLdsX3<N>::NP;
    MlpCtxX3 c;
    c.i = lane & 31;
    c.h = lane >> 5;
    const uint32_t lds0 = (uint32_t)(uintptr_t)&L;
    c.w1f = lds0 + (uint32_t)offsetof(MlpLdsX3<N>, W1) + 4u * (c.h * kLD1 + c.i);
    c.w1t = reinterpret_cast<const float4*>(L.W1T + 4 * c.h * NP);
    // forward: lane (r, h) reads row r (+ 32 per output block: immediate), 8-byte half h of chunk (2 kstep + b) ^ sw(r)
    const uint32_t r = c.i, h = c.h;
    const uint32_t fwd = r * kImgRow + 16u * (uint32_t)img_sw((int)r) + 8u * (h ^ (r >> 4));
    // transposed: lane 4 q + p of 16-lane group G supplies row 4 h + q (+ 32 kb + 16 s + 8 b: immediate) and columns 16 (G & 1) + 4 p .. + 3 (+ 32 fb: XOR)
    const uint32_t q = (lane >> 2) & 3, pp = lane & 3, G1 = (lane >> 4) & 1;
    const uint32_t trd = (4u * h + q) * kImgRow + 16u * ((q << 2) | ((2u * G1 + (pp >> 1)) ^ h)) + 8u * (pp & 1);
    const uint32_t w2 = lds0 + (uint32_t)offsetof(MlpLdsX3<N>, W2i), w3 = lds0 + (uint32_t)offsetof(MlpLdsX3<N>, W3i);
    c.f2[0] = w2 + fwd; c.f2[1] = w2 + kImgHalf + fwd;
    c.f3[0] = c.f3[1] = w3 + fwd;
    c.t2[0] = w2 + trd; c.t2[1] = w2 + kImgHalf + trd;
    c.t3[0] = c.t3[1] = w3 + trd;
    return c;
}

template <int BYTE_OFF> __device__ __forceinline__ u32x2 lds_read_b64(uint32_t addr) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 65536 && BYTE_OFF % 8 == 0, "");
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(BYTE_OFF));
    return v;
}
// EXEC must be all ones (the gather crosses lanes): the chains run with every lane active, padding lanes included
template <int BYTE_OFF> __device__ __forceinline__ u32x2 lds_read_tr16_b64(uint32_t addr) {
    static_assert(BYTE_OFF >= 0 && BYTE_OFF < 65536 && BYTE_OFF % 8 == 0, "");
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(BYTE_OFF));
    return v;
}

// One unit of a chain = (k-step K of 16 input features, 32-row output block O): 6 LDS reads (3 pieces x 2) and 6 MFMAs.
struct DirFwd {  // sum over the image's columns: rows = this product's output features
    static constexpr bool kTr = false;
    static constexpr int half(int K, int O) { return O >> 1; }
    static constexpr int xorc(int K, int O, int b) { return (2 * K + b) << 4; }
    static constexpr int imm(int piece, int K, int O, int b) { return piece * kImgPiece + (O & 1) * 32 * kImgRow; }
};
struct DirBwd {  // sum over the image's rows: columns = this product's output features
    static constexpr bool kTr = true;
    static constexpr int half(int K, int O) { return K >> 2; }
    static constexpr int xorc(int K, int O, int b) { return (O << 6) | (b << 5) | ((K & 1) << 3); }
    static constexpr int imm(int piece, int K, int O, int b) { return piece * kImgPiece + (((K >> 1) & 1) * 32 + (K & 1) * 16 + b * 8) * kImgRow; }
};

struct X3Ops { u32x2 r[3][2]; };  // [piece][b]: fragment elements 4 b .. 4 b + 3

// `root` is the lane base of this k-step, refreshed (and made opaque) once per k-step: the addresses are base ^ constant, loop invariant
// over tiles and repeated every other k-step, so hipcc otherwise computes all of them once per kernel and keeps ~50 of them in scratch.
template <typename Dir, int K, int O, int NOUT> __device__ __forceinline__ void x3_issue(X3Ops& s, const uint32_t (&base)[2], uint32_t& root) {
    if constexpr (O == 0 || (NOUT == 4 && O == 2 && !Dir::kTr)) {  // (forward: output blocks 2, 3 live in the other half of the image)
        root = base[Dir::half(K, O)];
        asm volatile("" : "+v"(root));
    }
    const uint32_t a0 = root ^ (uint32_t)Dir::xorc(K, O, 0), a1 = root ^ (uint32_t)Dir::xorc(K, O, 1);
    if constexpr (Dir::kTr) {
        s.r[0][0] = lds_read_tr16_b64<Dir::imm(0, K, O, 0)>(a0); s.r[0][1] = lds_read_tr16_b64<Dir::imm(0, K, O, 1)>(a1);
        s.r[1][0] = lds_read_tr16_b64<Dir::imm(1, K, O, 0)>(a0); s.r[1][1] = lds_read_tr16_b64<Dir::imm(1, K, O, 1)>(a1);
        s.r[2][0] = lds_read_tr16_b64<Dir::imm(2, K, O, 0)>(a0); s.r[2][1] = lds_read_tr16_b64<Dir::imm(2, K, O, 1)>(a1);
    } else {
        s.r[0][0] = lds_read_b64<Dir::imm(0, K, O, 0)>(a0); s.r[0][1] = lds_read_b64<Dir::imm(0, K, O, 1)>(a1);
        s.r[1][0] = lds_read_b64<Dir::imm(1, K, O, 0)>(a0); s.r[1][1] = lds_read_b64<Dir::imm(1, K, O, 1)>(a1);
        s.r[2][0] = lds_read_b64<Dir::imm(2, K, O, 0)>(a0); s.r[2][1] = lds_read_b64<Dir::imm(2, K, O, 1)>(a1);
    }
}

// pieces of one pair of B-operand elements (2 jj, 2 jj + 1) -> dword jj of the three fragments
// v = relu(v) and bit BIT of m = [v > 0]
template <int BIT> __device__ __forceinline__ void relu_mask(float& v, uint32_t& m) {
    v = relu1(v);
    uint32_t t;
    asm("v_min_u32_e32 %0, 1, %2\n\tv_lshl_or_b32 %1, %0, %3, %1" : "=&v"(t), "+v"(m) : "v"(v), "n"(BIT));
}
// x * [bit BIT of m]
template <int BIT> __device__ __forceinline__ float mask_apply(float x, uint32_t m) {
    float y;
    asm("v_bfe_i32 %0, %1, %2, 1\n\tv_and_b32_e32 %0, %0, %3" : "=&v"(y) : "v"(m), "n"(BIT), "v"(x));
    return y;
}

// What happens to the chain's input registers on their way into the B fragments (PRE): the element-wise work between two products runs
// inside the consuming chain, two elements per unit next to their split, instead of in a VALU-only pass in front of it (where this
// wave issues no MFMA for ~500 instructions and relies on its SIMD partner being inside a chain at that moment).
enum { kPreNone = 0, kPreReluMask = 1, kPreMaskApply = 2 };  // relu + record [v > 0] in the mask | multiply by the recorded mask bit

template <int NIN, int K, int JJ, int PRE> __device__ __forceinline__ void x3_split(const f32x16 (&in)[NIN], uint32_t (&Bp)[3][4], uint32_t (&m)[2]) {
    constexpr int kb = K >> 1, r0 = 8 * (K & 1) + 2 * JJ;  // registers r0, r0 + 1 of block kb: mask bits 16 (kb & 1) + r of m[kb >> 1]
    float x0 = in[kb][r0], x1 = in[kb][r0 + 1];
    if constexpr (PRE == kPreReluMask) {
        relu_mask<16 * (kb & 1) + r0>(x0, m[kb >> 1]);
        relu_mask<16 * (kb & 1) + r0 + 1>(x1, m[kb >> 1]);
    } else if constexpr (PRE == kPreMaskApply) {
        x0 = mask_apply<16 * (kb & 1) + r0>(x0, m[kb >> 1]);
        x1 = mask_apply<16 * (kb & 1) + r0 + 1>(x1, m[kb >> 1]);
    }
    x3_split_pair(x0, x1, Bp[0][JJ], Bp[1][JJ], Bp[2][JJ]);
}

template <typename Dir, int NK, int NOUT, int NIN, int PRE, int U>
__device__ __forceinline__ void x3_unit(f32x16 (&out)[NOUT], const f32x16 (&in)[NIN], const uint32_t (&base)[2], X3Ops (&ring)[3],
                                        uint32_t (&Bp)[2][3][4], uint32_t& root, uint32_t (&m)[2]) {
    constexpr int NU = NK * NOUT;
    if constexpr (U < NU) {
        constexpr int K = U / NOUT, O = U % NOUT;
        if constexpr (U + 2 < NU) x3_issue<Dir, (U + 2) / NOUT, (U + 2) % NOUT, NOUT>(ring[(U + 2) % 3], base, root);
        __builtin_amdgcn_sched_barrier(0);
        // this unit's share of the NEXT k-step's B pieces (4 element pairs per k-step, spread over its NOUT units)
        if constexpr (K + 1 < NK) {
            constexpr int PP = 4 / NOUT;
            if constexpr (PP >= 1) {
                x3_split<NIN, K + 1, O * PP, PRE>(in, Bp[(K + 1) & 1], m);
                if constexpr (PP == 2) x3_split<NIN, K + 1, O * PP + 1, PRE>(in, Bp[(K + 1) & 1], m);
            }
        }
        constexpr int ahead = (NU - 1 - U < 2 ? NU - 1 - U : 2) * 6;
        lds_wait<ahead>();
        const X3Ops& s = ring[U % 3];
        const u32x4 ah{s.r[0][0][0], s.r[0][0][1], s.r[0][1][0], s.r[0][1][1]};
        const u32x4 am{s.r[1][0][0], s.r[1][0][1], s.r[1][1][0], s.r[1][1][1]};
        const u32x4 al{s.r[2][0][0], s.r[2][0][1], s.r[2][1][0], s.r[2][1][1]};
        const uint32_t(&B)[3][4] = Bp[K & 1];
        const u32x4 bh{B[0][0], B[0][1], B[0][2], B[0][3]}, bm{B[1][0], B[1][1], B[1][2], B[1][3]}, bl{B[2][0], B[2][1], B[2][2], B[2][3]};
        f32x16 acc = out[O];
        acc = MFMA_BF16(al, bh, acc);  // smallest terms first
        acc = MFMA_BF16(ah, bl, acc);
        acc = MFMA_BF16(am, bm, acc);
        acc = MFMA_BF16(am, bh, acc);
        acc = MFMA_BF16(ah, bm, acc);
        acc = MFMA_BF16(ah, bh, acc);
        out[O] = acc;
        __builtin_amdgcn_sched_barrier(0);
        x3_unit<Dir, NK, NOUT, NIN, PRE, U + 1>(out, in, base, ring, Bp, root, m);
    }
}

// out[o] (+)= W-image product of PRE(`in`) (NIN 32-feature blocks); NK = 2 NIN k-steps.  `m`: the ReLU mask PRE records into / applies
// (128 bits per lane: bit 16 (fb & 1) + r of m[fb >> 1] belongs to register r of block fb).
template <typename Dir, int NOUT, int NIN, int PRE>
__device__ __forceinline__ void x3_chain(f32x16 (&out)[NOUT], const f32x16 (&in)[NIN], const uint32_t (&base)[2], uint32_t (&m)[2]) {
    constexpr int NK = 2 * NIN;
    static_assert(NOUT == 2 || NOUT == 4, "");
    X3Ops ring[3];
    uint32_t Bp[2][3][4];
    uint32_t root;
    x3_issue<Dir, 0, 0, NOUT>(ring[0], base, root);
    x3_issue<Dir, 1 / NOUT, 1 % NOUT, NOUT>(ring[1], base, root);
    __builtin_amdgcn_sched_barrier(0);
    x3_split<NIN, 0, 0, PRE>(in, Bp[0], m);
    x3_split<NIN, 0, 1, PRE>(in, Bp[0], m);
    x3_split<NIN, 0, 2, PRE>(in, Bp[0], m);
    x3_split<NIN, 0, 3, PRE>(in, Bp[0], m);
    x3_unit<Dir, NK, NOUT, NIN, PRE, 0>(out, in, base, ring, Bp, root, m);
    mfma_results_barrier<8>();
}

template <int FB, int R> __device__ __forceinline__ void mask_apply_block(f32x16 (&a)[4], const uint32_t (&m)[2]) {
    if constexpr (FB < 4) {
        const float v = a[FB][R];
        a[FB][R] = mask_apply<16 * (FB & 1) + R>(v, m[FB >> 1]);
        if constexpr (R + 1 < 16) mask_apply_block<FB, R + 1>(a, m);
        else mask_apply_block<FB + 1, 0>(a, m);
    }
}

// V and dV/dx of one tile of 32 environments (ReLU network); same contract as mlp_value_grad with TL = 1
template <typename S>
__device__ __forceinline__ void mlp_value_grad_x3(const S& sys, const MlpP<S::N>& p, const MlpCtxX3& c, const float (&xs)[1][S::N], bool want_grad,
                                                  float (&V)[1], float (&g)[1][S::N]) {
    constexpr int N = S::N;
    constexpr int NP = MlpLdsX3<N>::NP;
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
    // ---- layer 1 on the f32 matrix cores (k = n) ------------------------------------------------------------
    float ring4[3][4];
    f32x16 a1[1][4];
    zero_acc(a1);
    mfma_chain<OffW1F, N / 2, 4, 2, 1>(a1, ring4, c.w1f, [&](int st, int t) { return h ? z[t][2 * st + 1] : z[t][2 * st]; });
    mfma_results_barrier<16>();
    uint32_t m1[2] = {0u, 0u}, m2[2] = {0u, 0u}, m0[2] = {0u, 0u};
    // ---- layer 2 ------------------------------------------------------------------------------------------
    f32x16 a2[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[o][r] = 0.f;
    x3_chain<DirFwd, 4, 4, kPreReluMask>(a2, a1[0], c.f2, m1);  // relu(a1) and its mask m1 on the way in
    // ---- layer 3 ------------------------------------------------------------------------------------------
    f32x16 y[2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) y[o][r] = 0.f;
    x3_chain<DirFwd, 2, 4, kPreReluMask>(y, a2, c.f3, m2);
    float vpart = 0.f;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float yy = y[ob][r];
            vpart = fmaf(yy, yy, vpart);
        }
    V[0] = vpart + __shfl_xor(vpart, 32, 64) + p.eps_s * ee;
    if (!want_grad) return;
    // ---- backward 3, 2 (on y instead of dV/dy = 2 y: see the last line) ------------------------------------------
    f32x16 d2[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) d2[o][r] = 0.f;
    x3_chain<DirBwd, 4, 2, kPreNone>(d2, y, c.t3, m0);
    f32x16 d1[4];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[o][r] = 0.f;
    x3_chain<DirBwd, 4, 4, kPreMaskApply>(d1, d2, c.t2, m2);
    mask_apply_block<0, 0>(d1, m1);
    // ---- backward 1 on the VALU (as in mlp_value_grad) ---------------------------------------------------------
    f32x2 part[NP / 2];
#pragma unroll
    for (int k = 0; k < NP / 2; ++k) part[k] = f32x2{0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float dv = d1[kb][s];
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
        g[0][k] = (v + v) * p.istd[k] + 2.f * p.eps_s * e[k];  // dV/dy = 2 y: the backward products ran on y, the factor 2 (exact) comes in here
    }
}
