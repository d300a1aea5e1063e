// hjbx_train_coop.hip -- the parameter gradient of the value-learning step (reference controller/vhjb.py:227-253, 282-284) in ONE kernel with
// NO scratch in HBM (round 3; hjbx_train.hip holds the round-2 pair of kernels, which moved 10 KB per sample through HBM and walked a tile's
// 1,552 MFMAs on one wave).  ReLU (controller/vhjb.py), tanh (examples/cartpole_balancing.ipynb cell 6) and sin
// (examples/double_integrator_optimal_time.ipynb cell 5) networks, float32 MFMA.
//
// Math per sample (s = act'(a); ReLU: act'' = 0, tanh: act'' = -2 h s, sin: act'' = -h), q = d loss_hjb / d gradV, r = d loss_term / d V:
//   forward            h1 = act(W1'z)   h2 = act(W2'h1)   y = W3'h2   V = |y|^2 + eps_s |e|^2
//   input gradient     dy = 2y   d2 = (W3 dy).s2   d1 = (W2 d2).s1   g = (W1 d1)/std + 2 eps_s e
//   reverse sweep (q)  gzb = q/std   t1 = W1'gzb   dh1b = t1.s1   t2 = W2'dh1b   dh2b = t2.s2   yb = 2 W3'dh2b
//                      a2b = (W3 yb).s2 - 2 h2.d2.t2 [tanh] - h2.(W3 dy).t2 [sin]      a1b = (W2 a2b).s1 - 2 h1.d1.t1 [tanh] - h1.(W2 d2).t1 [sin]
//   hjb gradient       dW1 = gzb (x) d1 + z (x) a1b      dW2 = dh1b (x) d2 + h1 (x) a2b      dW3 = dh2b (x) dy + h2 (x) yb
//   termination grad.  dW1 = z (x) (r d1)                dW2 = h1 (x) (r d2)                 dW3 = h2 (x) (r dy)
// (the second-order terms of tanh: the adjoint of s = 1 - h^2 is (adjoint of d).(W d_next) and d s / d a = -2 h s, so a_bar gains
//  -2 h . d . t with t the pre-mask value of the reverse sweep.)
//
// Design.  A workgroup = 4 waves = one wave per SIMD with 512 registers; it works on ONE 32-sample tile at a time, COOPERATIVELY:
//  * every 128-wide array (h1, h2, d2, d1, ...) is split by 32-feature block over the four waves, so a product W'X is 64 k-steps of ONE
//    MFMA per wave instead of 256 MFMAs on one wave (latency of a tile / 4: what the reference's minibatch of 256 = 8 tiles needs), and
//    each wave keeps its blocks of h1, h2, dy, d1 (and the tanh corrections) in registers across the whole tile;
//  * a product needs all 128 input features as B operands, so each result block goes through a [feature][sample] image in LDS (stride 33:
//    conflict-free for the column-wise write, the chain's B read and the outer products' A / B reads alike).  Those images ARE the
//    transposition the outer products need (their contraction index is the sample): no scratch, no second kernel.  Three 16.5-KiB
//    images suffice; h1, h2, dy are written again from registers when their partner of an outer product arrives;
//  * the 48 32x32 output blocks of dW2 / dW3 (hjb + termination sets) are MFMA accumulators for the whole launch, 12 per wave (row block
//    w of dW2 and of dW3: their A operands are shared); dW1 (n x 128: 3 % of the flops) is accumulated on the VALU, one feature per thread;
//  * y = W3'h2 has only two 32-row blocks: its contraction is split in halves over wave pairs and summed through LDS, so all four matrix
//    pipes work in every phase.  g = W1 d1 is a 16-k-step MFMA per wave over its own block (B operands straight from the accumulators) +
//    an LDS sum; every wave then evaluates the two residuals for its sample redundantly (same bits) instead of waiting for one.
// LDS: weights 102-104 KB (f32, odd strides, one copy for W and W') + 3 x 16.5 KB images + 5 KB small = 157 KB (n = 10).
// Per tile and wave: 394 chain MFMAs + 336 outer-product MFMAs; HBM traffic = the inputs (4(n+2) B per sample) + the partial sums.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "hjbx_internal.hpp"
#include "hjbx_systems.hpp"
#include "hjbx_host.hpp"
#include "hjbx_mlp_core.hpp"
#include "hjbx_adam.hpp"

using namespace hjbx;

static constexpr int kExLd = 33;                      // row stride of an exchange image (floats)
static constexpr int kExFloats = 128 * kExLd;
static constexpr int kCoopBlocks = 48;                // per set 24: dW2 (ib, jb) -> ib * 4 + jb; dW3 (ib, jb) -> 16 + ib * 2 + jb
static constexpr int kCoopSet = 24;
static constexpr int kCoopMaxGrid = 512;             // workgroups of a launch (one per CU at most) the fused epilogue keeps records for

template <int N> struct CoopLds {
    static constexpr int NP = (N + 3) & ~3;
    float W1[N * kLD1];
    float W2[kH1 * kLD2];
    float W3[kH2 * kLD3];
    float E[3][kExFloats];
    __attribute__((aligned(16))) float zs[32 * NP];    // [sample][k]: z, zero padded to NP
    __attribute__((aligned(16))) float gzbs[32 * NP];  // [sample][k]: q / std
    float rs[32];                                      // r = d loss_term / d V per sample
    float vp[2][32];                                   // |y|^2 partial sums of the two 32-row blocks of y
    float zeros[32];                                   // A operand of the lanes that stand for rows >= n of W1 (g product)
    double sums[4][32];                                // running loss sums / counts per sample slot (wave 0 adds to them once per tile: 8 registers less
                                                       // through the whole tile loop than four double accumulators per lane)
};

// ---- a chain whose A (weights) AND B (an exchange image) operands both come from LDS ------------------------------------------------
// Same discipline as mfma_chain (hjbx_mlp_core.hpp): inline-asm ds_reads DEPTH steps ahead, retired by counted s_waitcnt lgkmcnt, so
// that a step is 2 reads + 1 wait + 1 MFMA and hipcc cannot sink the reads to their use.  AOFF / BOFF: byte offset of step st from the
// lane-dependent bases (compile-time constants).
template <int AOFF, int BOFF, int ST> __device__ __forceinline__ void coop_issue(float& a, float& b, uint32_t abase, uint32_t bbase) {
    a = lds_read_b32<AOFF * ST>(abase);
    b = lds_read_b32<BOFF * ST>(bbase);
}
// `filler(integral_constant<int, st>)` runs after the MFMA of step st: independent VALU / LDS work placed there issues while the matrix pipe
// executes that MFMA (64 cycles), i.e. for free -- with ONE wave per SIMD nothing else can hide it.  (LDS reads the compiler adds between
// the asm reads only make the counted waits conservative: LDS returns in order.)
struct NoFiller { template <typename I> __device__ __forceinline__ void operator()(I) const {} };
template <int AOFF, int BOFF, int NSTEPS, int DEPTH, int ST, typename Filler>
__device__ __forceinline__ void coop_chain_step(f32x16& acc, float (&ra)[DEPTH + 1], float (&rb)[DEPTH + 1], uint32_t abase, uint32_t bbase, const Filler& filler) {
    if constexpr (ST < NSTEPS) {
        if constexpr (ST + DEPTH < NSTEPS) coop_issue<AOFF, BOFF, ST + DEPTH>(ra[(ST + DEPTH) % (DEPTH + 1)], rb[(ST + DEPTH) % (DEPTH + 1)], abase, bbase);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int ahead = (NSTEPS - 1 - ST < DEPTH ? NSTEPS - 1 - ST : DEPTH) * 2;
        lds_wait<ahead>();
        acc = MFMA(ra[ST % (DEPTH + 1)], rb[ST % (DEPTH + 1)], acc);
        __builtin_amdgcn_sched_barrier(0);
        filler(std::integral_constant<int, ST>{});
        coop_chain_step<AOFF, BOFF, NSTEPS, DEPTH, ST + 1>(acc, ra, rb, abase, bbase, filler);
    }
}
template <int AOFF, int BOFF, int NSTEPS, typename Filler = NoFiller>
__device__ __forceinline__ void coop_chain(f32x16& acc, uint32_t abase, uint32_t bbase, const Filler& filler = Filler()) {
    constexpr int DEPTH = 3;
    static_assert(AOFF * (NSTEPS - 1) < 65536 && BOFF * (NSTEPS - 1) < 65536, "ds_read_b32 offset field is 16 bits");
    float ra[DEPTH + 1], rb[DEPTH + 1];
    coop_issue<AOFF, BOFF, 0>(ra[0], rb[0], abase, bbase);
    if constexpr (NSTEPS > 1) coop_issue<AOFF, BOFF, 1>(ra[1], rb[1], abase, bbase);
    if constexpr (NSTEPS > 2) coop_issue<AOFF, BOFF, 2>(ra[2], rb[2], abase, bbase);
    coop_chain_step<AOFF, BOFF, NSTEPS, DEPTH, 0>(acc, ra, rb, abase, bbase, filler);
}

// A operands of the two products whose B operands are registers (mfma_chain of hjbx_mlp_core.hpp, one output block)
struct OffW1Fc { static constexpr int at(int st, int) { return 2 * st * kLD1 * 4; } };   // W1[2 st + h][32 w + i]
struct OffW1Gc { static constexpr int at(int st, int) { return perm(st) * 4; } };        // W1[i][32 w + perm(st) + 4 h]

// 32-bit pointers into LDS: address arithmetic on them stays `ds_read_b32 v, vaddr offset:constant`.  Inside the tile loop every lane base
// is re-derived from an OPAQUE copy of the image pointers (opaque3): left alone, hipcc hoists the ~200 loop-invariant `base + constant`
// addresses of the outer products out of the tile loop, keeps them in registers for the whole kernel and spills them (78 dwords of scratch
// in the first build of this kernel; the same trap as the XOR-swizzled bases of hjbx_mlp_h2.hpp).
using LP = __attribute__((address_space(3))) float*;
using LPc = const __attribute__((address_space(3))) float*;
__device__ __forceinline__ uint32_t lds_addr(LPc q) { return (uint32_t)(uintptr_t)q; }

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// this lane's 16 accumulator values (rows perm(r) + 4 h of its wave's 32-row block, column = sample i) into / out of a [feature][sample] image
__device__ __forceinline__ void ex_write(LP blk /* &E[(32 w + 4 h) * kExLd + i] */, const f32x16& v) {
#pragma unroll
    for (int r = 0; r < 16; ++r) blk[perm(r) * kExLd] = v[r];
}
__device__ __forceinline__ void ex_add(f32x16& v, LPc blk) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] += blk[perm(r) * kExLd];
}

// Outer products of one 32-sample tile: acch[j] += A_h (x) B_h,j and acct[j] += A_t (x) (r B_t,j) over the 16 k-steps (2 samples each).
// A = rows 32 w + i of an image (exA = (32 w + i) kExLd + h), B_j = rows 32 j + i of another (exBj = i kExLd + h), rs = r per sample.
template <int NB, bool HJB, bool TERM, typename Filler = NoFiller>
__device__ __forceinline__ void coop_outer(f32x16 (&acch)[NB], LPc Ah, LPc Bh, f32x16 (&acct)[NB], LPc At, LPc Bt, LPc rs, int exA, int exBj, int h,
                                           const Filler& filler = Filler()) {
    LPc ah_p = Ah + exA, bh_p = Bh + exBj, at_p = At + exA, bt_p = Bt + exBj, r_p = rs + h;   // lane bases; everything below is base + constant
    // software pipeline, pinned by sched_barrier: the operands of k-step s + 1 are read while the MFMAs of k-step s issue; left to itself the
    // scheduler reads many k-steps ahead and the registers of those loads push long-lived values into scratch
    struct Ops { float ah, at, rr, bh[NB], bt[NB]; };
    auto load = [&](int s2) __attribute__((always_inline)) {
        Ops o;
        o.ah = o.at = o.rr = 0.f;
        if constexpr (HJB) o.ah = ah_p[2 * s2];
        if constexpr (TERM) { o.at = at_p[2 * s2]; o.rr = r_p[2 * s2]; }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            o.bh[j] = o.bt[j] = 0.f;
            if constexpr (HJB) o.bh[j] = bh_p[32 * j * kExLd + 2 * s2];
            if constexpr (TERM) o.bt[j] = bt_p[32 * j * kExLd + 2 * s2];
        }
        return o;
    };
    Ops cur = load(0);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        Ops nxt = cur;
        if (s + 1 < 16) nxt = load(s + 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if constexpr (HJB) acch[j] = MFMA(cur.ah, cur.bh[j], acch[j]);
            if constexpr (TERM) acct[j] = MFMA(cur.at, cur.rr * cur.bt[j], acct[j]);
        }
        filler(s);                               // (independent VALU work in the shadow of this k-step's MFMAs)
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    }
}

// DEVELOPMENT timing switches (tools/dev/coop_variants.sh builds variants with them; results are garbage, only the time means something)
#ifdef HJBX_COOP_NO_OUTER
#define COOP_OUTER(...)
#else
#define COOP_OUTER(...) __VA_ARGS__
#endif
#ifdef HJBX_COOP_NO_CHAINS
#define COOP_CHAIN(...)
#else
#define COOP_CHAIN(...) __VA_ARGS__
#endif
#ifdef HJBX_COOP_NO_BARRIER
#define COOP_SYNC() __builtin_amdgcn_sched_barrier(0)
#elif defined(HJBX_COOP_STAMPS)   // development: wall-clock stamps (100 MHz) of one non-owner workgroup at every barrier, into its unused dW1 record
#define COOP_STAMP() do { if (stamp_on) stamp_buf[stamp_idx++] = wall_clock64(); } while (0)
#define COOP_SYNC() do { __syncthreads(); COOP_STAMP(); } while (0)
#else
#define COOP_SYNC() __syncthreads()
#endif
#ifndef COOP_STAMP
#define COOP_STAMP() do { } while (0)
#endif

template <int MODE, int ACT, int PS, typename S>
__global__ __launch_bounds__(256, 1) void k_train_coop(S sys_k, MlpP<S::N> p_k, TaskP<float, S::N, S::M> tk_k, Limits<float, S::M> lim_k,
                                                       const float* __restrict__ W1g, const float* __restrict__ W2g, const float* __restrict__ W3g,
                                                       const float* __restrict__ x, const float* __restrict__ cost, const float* __restrict__ done,
                                                       float eps_term, float* __restrict__ partial, float* __restrict__ partial_w1,
                                                       double* __restrict__ sums_rec, int64_t B, int64_t ntiles) {
    constexpr int N = S::N, M = S::M;
    constexpr int NP = CoopLds<N>::NP;
    static_assert(N % 2 == 0 && N <= HJBX_MAX_N, "state dimension");
    __shared__ __attribute__((aligned(256))) CoopLds<N> L;
    __shared__ __attribute__((aligned(16))) unsigned char sys_raw[sizeof(S)];
    S& sys_s = *reinterpret_cast<S*>(sys_raw);
    __shared__ MlpP<N> p_s;
    __shared__ TaskP<float, N, M> tk_s;
    __shared__ Limits<float, M> lim_s;
    const int tid = threadIdx.x;
#ifdef HJBX_COOP_STAMPS
    const bool stamp_on = PS == 4 && blockIdx.x == 1 && tid == 0;
    unsigned long long* stamp_buf = reinterpret_cast<unsigned long long*>(partial_w1 + (int64_t)blockIdx.x * 2 * (2 * S::N * 128));
    int stamp_idx = 0;
    COOP_STAMP();                                                                       // 0: kernel entry
#endif
    if (tid == 0) { sys_s = sys_k; p_s = p_k; tk_s = tk_k; lim_s = lim_k; }
#ifndef HJBX_COOP_NO_FILL   // (development timing switch, see COOP_OUTER)
    {   // weights -> LDS (odd row strides), 16 bytes per global load: at the reference's minibatch (8 tiles) this fill is on the latency path
        static_assert(kH1 % 4 == 0 && kH2 % 4 == 0 && kH3 % 4 == 0, "");
        const float4* W1v = reinterpret_cast<const float4*>(W1g);
        const float4* W2v = reinterpret_cast<const float4*>(W2g);
        const float4* W3v = reinterpret_cast<const float4*>(W3g);
        auto put4 = [](float* dst, const float4& v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; };
        // every load of a thread is issued before its first LDS write (16 + 8 + 1..2 float4 per thread: ~100 registers that nothing else needs
        // yet): with the loads issued a few at a time the fill took 2.9 us of the 24.6 us a workgroup spends on an 8-tile minibatch (wall-clock
        // stamps, tools/dev/coop_stamps.py)
        constexpr int Q1 = (N * kH1 / 4 + 255) / 256, Q2 = kH1 * kH2 / 4 / 256, Q3 = kH2 * kH3 / 4 / 256;
        static_assert(kH1 * kH2 / 4 % 256 == 0 && kH2 * kH3 / 4 % 256 == 0, "");
        float4 v1[Q1], v2[Q2], v3[Q3];
#pragma unroll
        for (int q = 0; q < Q2; ++q) v2[q] = W2v[tid + 256 * q];
#pragma unroll
        for (int q = 0; q < Q3; ++q) v3[q] = W3v[tid + 256 * q];
#pragma unroll
        for (int q = 0; q < Q1; ++q) v1[q] = tid + 256 * q < N * kH1 / 4 ? W1v[tid + 256 * q] : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            const int idx = tid + 256 * q;
            if (idx < N * kH1 / 4) put4(&L.W1[(idx / (kH1 / 4)) * kLD1 + 4 * (idx % (kH1 / 4))], v1[q]);
        }
#pragma unroll
        for (int q = 0; q < Q2; ++q) { const int idx = tid + 256 * q; put4(&L.W2[(idx / (kH2 / 4)) * kLD2 + 4 * (idx % (kH2 / 4))], v2[q]); }
#pragma unroll
        for (int q = 0; q < Q3; ++q) { const int idx = tid + 256 * q; put4(&L.W3[(idx / (kH3 / 4)) * kLD3 + 4 * (idx % (kH3 / 4))], v3[q]); }
    }
#endif
    if (tid < 32) L.zeros[tid] = 0.f;
    if (tid < 128) L.sums[tid >> 5][tid & 31] = 0.0;
    for (int idx = tid; idx < kExFloats; idx += 256) L.E[2][idx] = 0.f;   // (the first tile's chain 2 reads "the previous tile's a1b" from here)
    for (int idx = tid; idx < 32 * NP; idx += 256) { L.zs[idx] = 0.f; L.gzbs[idx] = 0.f; }
    __syncthreads();
    COOP_STAMP();                                                                       // 1: LDS filled
    const S& sys = sys_s;
    const MlpP<N>& p = p_s;
    const TaskP<float, N, M>& tk = tk_s;
    const Limits<float, M>& lim = lim_s;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int ob = w & 1, kh = w >> 1;                 // y = W3'h2: output block and contraction half of this wave
    // Small batches (PS = 4: at most a quarter as many tiles as CUs): FOUR workgroups work on the same tile -- each runs the tile's chains
    // (redundantly: that costs no time) but accumulates only column block `part` of the outer products (dW2: one of four; dW3: parts 0, 1
    // one of two each), dW1 and the loss sums going to part 0: a tile's 288 outer-product MFMAs per wave shrink to 48-96 on the latency
    // path of the reference's minibatch of 256 (8 tiles -> 32 CUs).  PS is a template parameter: a run-time choice of the owned blocks
    // inside the MFMA loops cost 60-110 spilled registers.
    static_assert(PS == 1 || PS == 4, "");
    constexpr int psplit = PS;
    constexpr int NB2 = PS == 4 ? 1 : 4, NB3 = PS == 4 ? 1 : 2;
    const int part = PS == 4 ? (int)(blockIdx.x & 3u) : 0;
    const int col2 = PS == 4 ? 32 * part * kExLd : 0;          // offset of this part's column block inside a 128-row image (dW2)
    const int col3 = PS == 4 ? 32 * (part & 1) * kExLd : 0;    //                                            64-row image (dW3)
    const bool do3 = PS == 1 || part < 2;
    const bool own1 = part == 0;                               // dW1 and the loss sums
    const float m1 = own1 ? 1.0f : 0.0f;                       // (dW1 of the other parts accumulates zeros: a factor, not a branch inside the MFMA loops)
    // LDS byte addresses (the low 32 bits of a flat pointer into the LDS aperture are the LDS byte address)
    auto lds = [](const void* q) { return (uint32_t)(uintptr_t)q; };
    auto lds3 = [](LPc q) { return lds_addr(q); };
    const uint32_t aW1f = lds(&L.W1[h * kLD1 + 32 * w + i]);                       // W1[2 st + h][32 w + i]
    const uint32_t aW1g = i < N ? lds(&L.W1[i * kLD1 + 32 * w + 4 * h]) : lds(&L.zeros[0]);   // W1[i][32 w + perm(st) + 4 h], rows >= n read zeros
    const uint32_t aW2f = lds(&L.W2[h * kLD2 + 32 * w + i]);                       // W2[2 st + h][32 w + i]
    const uint32_t aW2b = lds(&L.W2[(32 * w + i) * kLD2 + h]);                     // W2[32 w + i][2 st + h]
    const uint32_t aW3f = lds(&L.W3[(64 * kh + h) * kLD3 + 32 * ob + i]);          // W3[64 kh + 2 st + h][32 ob + i]
    const uint32_t aW3b = lds(&L.W3[(32 * w + i) * kLD3 + h]);                     // W3[32 w + i][2 st + h]
    const LP E0g = (LP)&L.E[0][0], E1g = (LP)&L.E[1][0], E2g = (LP)&L.E[2][0];
    const LP rsg = (LP)&L.rs[0], zsg = (LP)&L.zs[0], gzbsg = (LP)&L.gzbs[0];
    const int exB = h * kExLd + i;                     // B operand of step st: image[(2 st + h)][i]
    const int exW = (32 * w + 4 * h) * kExLd + i;      // this lane's writes of its wave's 128-wide block
    const int exWy = (32 * ob + 4 * h) * kExLd + i;    // ... of its 64-wide block (y, dy, yb)
    const int exA = (32 * w + i) * kExLd + h;          // outer products: A operand of k-step s = image[32 w + i][2 s + h]
    const int exO = i * kExLd + h;                     //                 B operand of column block j = image[32 j + i][2 s + h]
    constexpr int AO1 = 2 * kLD2 * 4, AO3 = 2 * kLD3 * 4, BOX = 2 * kExLd * 4;

    f32x16 acc2h[NB2], acc2t[NB2], acc3h[NB3], acc3t[NB3];   // dW2 row block w (hjb, termination), dW3 row block w: accumulators of the whole launch
#pragma unroll
    for (int j = 0; j < NB2; ++j) { zero16(acc2h[j]); zero16(acc2t[j]); }
#pragma unroll
    for (int j = 0; j < NB3; ++j) { zero16(acc3h[j]); zero16(acc3t[j]); }
    f32x2 w1h[N / 2], w1t[N / 2];                      // dW1[k][f], f = tid & 127, over the samples 16 (tid >> 7) .. + 15 of every tile
#pragma unroll                                         // (pairs of k: one v_pk_fma_f32 per two entries; N is even)
    for (int k = 0; k < N / 2; ++k) w1h[k] = w1t[k] = f32x2{0.f, 0.f};
    const int fW1 = tid & 127, sW1 = 16 * (tid >> 7);

    // outer products of one 32-sample tile (coop_outer below): acc[j] += A (x) B_j over the 16 k-steps (2 samples each)
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
    const int64_t tile_stride = gridDim.x / (unsigned)psplit;
    fetch(blockIdx.x / (unsigned)psplit, xs_n, dn_n, cst_n);
    for (int64_t tile = blockIdx.x / (unsigned)psplit; tile < ntiles; tile += tile_stride) {
        asm volatile("" ::: "memory");   // the weights are loop invariant: keep their LDS reads inside the loop (see hjbx_mlp.hip)
        LP E0 = E0g, E1 = E1g, E2 = E2g, rsp = rsg, zsp = zsg, gzbsp = gzbsg;
        asm volatile("" : "+v"(E0), "+v"(E1), "+v"(E2), "+v"(rsp), "+v"(zsp), "+v"(gzbsp));   // (see LP above)
        const bool valid = tile * 32 + i < B;
        float xs[N];
#pragma unroll
        for (int k = 0; k < N; ++k) xs[k] = xs_n[k];
        const float dn = dn_n, cst = cst_n;
        float e[N], z[N], ee = 0.f;
#pragma unroll
        for (int k = 0; k < N; ++k) e[k] = xs[k] - p.xf[k];
        sys.wrap(e);
#pragma unroll
        for (int k = 0; k < N; ++k) {
            ee += e[k] * e[k];
            z[k] = (e[k] - p.mean[k]) * p.istd[k];
        }
        float ring1[3][1];
        f32x16 t[1][1];

        // ---- 1. h1 = act(W1'z): B operands are this lane's own z ------------------------------------------------------------------------
        zero_acc(t);
        mfma_chain<OffW1Fc, N / 2, 1, 2, 1>(t, ring1, aW1f, [&](int st, int) { return h ? z[2 * st + 1] : z[2 * st]; });
        constexpr bool SIN = ACT == HJBX_ACT_SIN;   // act' = cos(a) is kept beside the activation (s1r, s2r); relu / tanh derive it from the activation
        f32x16 h1r, s1r;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (SIN) {
                float sn, cs;
                sincos1(t[0][0][r], sn, cs);
                asm volatile("" : "+v"(sn), "+v"(cs));   // evaluated here, not sunk to the uses (see mlp_value_grad)
                h1r[r] = sn;
                s1r[r] = cs;
            } else {
                h1r[r] = act1<ACT>(t[0][0][r]);
            }
        }
        auto dmul1 = [&](int r, float v) __attribute__((always_inline)) { if constexpr (SIN) return v * s1r[r]; else return dact1<ACT>(h1r[r], v); };
        ex_write(E0 + exW, h1r);
        COOP_SYNC();                                                                    // (A) E0 = h1
        // ---- 2. h2 = act(W2'h1) -------------------------------------------------------------------------------------------------------
        f32x16 acc;
        zero16(acc);
        {   // in the shadow of this chain's MFMAs: the PREVIOUS tile's z (x) a1b (a1b in E2, its z still in zs; both zero before the first tile)
            using f32x4 = __attribute__((ext_vector_type(4))) float;
            using LP4 = const __attribute__((address_space(3))) f32x4*;
            LPc a1p = E2 + fW1 * kExLd + sW1;
            const LP4 zz4 = (LP4)(zsp + sW1 * NP);
            auto w1_part2 = [&](auto st_c) __attribute__((always_inline)) {
                constexpr int st = decltype(st_c)::value;
#if !defined(HJBX_COOP_NO_W1) && !defined(HJBX_COOP_NO_FILL2)
                if constexpr (st % 4 == 0) {
                    constexpr int s2 = st / 4;
                    const float a = a1p[s2] * m1;
                    const f32x2 a2v{a, a};
#pragma unroll
                    for (int k4 = 0; k4 < NP / 4; ++k4) {
                        const f32x4 zz = zz4[s2 * (NP / 4) + k4];
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            if (4 * k4 + 2 * c < N) w1h[2 * k4 + c] = __builtin_elementwise_fma(f32x2{zz[2 * c], zz[2 * c + 1]}, a2v, w1h[2 * k4 + c]);
                    }
#pragma unroll
                    for (int k = 0; k < N / 2; ++k) asm volatile("" : "+v"(w1h[k]));   // (pinned: see w1_part1)
                }
#endif
            };
#ifdef HJBX_COOP_NO_CHAINS
            (void)w1_part2;
#endif
            COOP_CHAIN(coop_chain<AO1, BOX, 64>(acc, aW2f, lds3(E0 + exB), w1_part2);)
        }
        f32x16 h2r, s2r;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (SIN) {
                float sn, cs;
                sincos1(acc[r], sn, cs);
                asm volatile("" : "+v"(sn), "+v"(cs));
                h2r[r] = sn;
                s2r[r] = cs;
            } else {
                h2r[r] = act1<ACT>(acc[r]);
            }
        }
        auto dmul2 = [&](int r, float v) __attribute__((always_inline)) { if constexpr (SIN) return v * s2r[r]; else return dact1<ACT>(h2r[r], v); };
        ex_write(E1 + exW, h2r);
        COOP_SYNC();                                                                    // (B) E1 = h2
        // ---- 3. y = W3'h2: block ob, contraction half kh; halves summed through E2; V, r --------------------------------------------------
        zero16(acc);
        COOP_CHAIN(coop_chain<AO3, BOX, 32>(acc, aW3f, lds3(E1 + 64 * kh * kExLd + exB));)
        if (kh == 1) ex_write(E2 + exWy, acc);
        COOP_SYNC();                                                                    // (C) E2[0:64] = the upper half's partial y
        f32x16 dyr;                                                                         // (waves 0, 1: block ob of dy = 2 y)
        zero16(dyr);
        if (kh == 0) {
            ex_add(acc, E2 + exWy);
            float vpart = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                vpart += acc[r] * acc[r];
                dyr[r] = acc[r] + acc[r];
            }
            vpart += __shfl_xor(vpart, 32, 64);
            if (h == 0) L.vp[ob][i] = vpart;
            ex_write(E0 + exWy, dyr);
        }
        COOP_SYNC();                                                                    // (D) E0[0:64] = dy, vp
        const float V = (L.vp[0][i] + L.vp[1][i]) + p.eps_s * ee;
        float lt, rterm;
        termination_residual_env<float>(eps_term, V, cst, dn, lt, rterm);
        if (!valid) lt = rterm = 0.f;
        if (w == 0 && h == 0) {
            L.rs[i] = rterm;
            if (valid && own1) { L.sums[1][i] += (double)lt; L.sums[2][i] += 1.0 - (double)dn; L.sums[3][i] += (double)dn; }
        }
        // ---- 4. d2 = (W3 dy).s2 -------------------------------------------------------------------------------------------------------
        zero16(acc);
        COOP_CHAIN(coop_chain<2 * 4, BOX, 32>(acc, aW3b, lds3(E0 + exB));)
        f32x16 d2r;
        f32x16 c2r;                                                                         // second-order term of a2b (tanh, sin), completed at step 7
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d2r[r] = dmul2(r, acc[r]);
            if constexpr (SIN) c2r[r] = -h2r[r] * acc[r];                                   // act'' (W3 dy) = -sin(a2) . (pre-mask value)
        }
        ex_write(E1 + exW, d2r);                                                            // (h2's readers finished before (C))
        COOP_SYNC();                                                                    // (E) E1 = d2, rs
        // ---- 5. d1 = (W2 d2).s1; g = W1 d1 / std + 2 eps_s e; the hjb residual ------------------------------------------------------------
        zero16(acc);
        COOP_CHAIN(coop_chain<2 * 4, BOX, 64>(acc, aW2b, lds3(E1 + exB));)
        f32x16 d1r;
        f32x16 c1r;                                                                         // second-order term of a1b (tanh, sin), completed at step 6
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            d1r[r] = dmul1(r, acc[r]);
            if constexpr (SIN) c1r[r] = -h1r[r] * acc[r];
        }
        ex_write(E2 + exW, d1r);                                                            // (the partial y's readers finished before (D))
        zero_acc(t);
        mfma_chain<OffW1Gc, 16, 1, 2, 1>(t, ring1, aW1g, [&](int st, int) { return d1r[st]; });
        {   // partial g of this wave's 32 features: rows k < n of the result, into E0 (dy's readers finished before (E)) as [w][k][sample]
            LP gp = E0 + w * (N * 32) + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k0 = perm(r);                  // row of lane half 0; lane half 1 holds row k0 + 4
                if (k0 + 4 * h < N && (k0 < N)) gp[(k0 + 4 * h) * 32] = t[0][0][r];
            }
        }
        COOP_SYNC();                                                                    // (F) E2 = d1, E0 = partial g
        float g[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
            LPc gp = E0 + k * 32 + i;
            g[k] = ((gp[0] + gp[N * 32]) + (gp[2 * N * 32] + gp[3 * N * 32])) * p.istd[k] + 2.f * p.eps_s * e[k];
        }
        float li, q[N];
#ifdef HJBX_COOP_NO_RESID
        li = g[0];
#pragma unroll
        for (int k = 0; k < N; ++k) q[k] = g[k];
#else
        hjb_residual_env<MODE>(sys, tk, lim, xs, g, dn, true, li, q);
#endif
        if (!valid) {   // padding lanes of the last tile contribute nothing
            li = 0.f;
#pragma unroll
            for (int k = 0; k < N; ++k) q[k] = 0.f;
        }
        float gzb[N];
#pragma unroll
        for (int k = 0; k < N; ++k) gzb[k] = q[k] * p.istd[k];
        if (w == 0 && h == 0) {
            if (valid && own1) L.sums[0][i] += (double)li;
#pragma unroll
            for (int k = 0; k < N; ++k) { L.zs[i * NP + k] = z[k]; L.gzbs[i * NP + k] = gzb[k]; }
        }
        // ---- 6. t1 = W1'gzb, dh1b = t1.s1 (B operands: this lane's own gzb) ---------------------------------------------------------------
        zero_acc(t);
        mfma_chain<OffW1Fc, N / 2, 1, 2, 1>(t, ring1, aW1f, [&](int st, int) { return h ? gzb[2 * st + 1] : gzb[2 * st]; });
        f32x16 dh1b;                                                                        // tanh: c1 = -2 h1 . d1 . t1; sin: c1 = -h1 . (W2 d2) . t1
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (ACT == HJBX_ACT_TANH) c1r[r] = -2.f * h1r[r] * d1r[r] * t[0][0][r];
            if constexpr (SIN) c1r[r] *= t[0][0][r];
            dh1b[r] = dmul1(r, t[0][0][r]);
        }
        COOP_SYNC();                                                                    // (G) zs, gzbs visible; partial g read
        ex_write(E0 + exW, dh1b);
        COOP_SYNC();                                                                    // (I) E0 = dh1b, E1 = d2, E2 = d1
        // dW2 += dh1b (x) d2, and in the shadow of its MFMAs the first part of dW1 on the VALU: gzb (x) d1 and z (x) (r d1), d1 from E2
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        using LP4 = const __attribute__((address_space(3))) f32x4*;
        LPc d1p = E2 + fW1 * kExLd + sW1, rp1 = rsp + sW1;
        const LP4 gz4 = (LP4)(gzbsp + sW1 * NP), zz4 = (LP4)(zsp + sW1 * NP);
        auto w1_part1 = [&](int s2) __attribute__((always_inline)) {
#if !defined(HJBX_COOP_NO_W1) && !defined(HJBX_COOP_NO_FILL1)
            const float d = d1p[s2] * m1;
            const float rd = rp1[s2] * d;
            const f32x2 d2v{d, d}, rd2v{rd, rd};
#pragma unroll
            for (int k4 = 0; k4 < NP / 4; ++k4) {
                const f32x4 gz = gz4[s2 * (NP / 4) + k4];
                const f32x4 zz = zz4[s2 * (NP / 4) + k4];
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (4 * k4 + 2 * c < N) {
                        w1h[2 * k4 + c] = __builtin_elementwise_fma(f32x2{gz[2 * c], gz[2 * c + 1]}, d2v, w1h[2 * k4 + c]);
                        w1t[2 * k4 + c] = __builtin_elementwise_fma(f32x2{zz[2 * c], zz[2 * c + 1]}, rd2v, w1t[2 * k4 + c]);
                    }
            }
            // pin the slice HERE: fma is a pure operation, and instruction selection sinks pure operations towards their use -- the store at
            // the end of the kernel -- so all sixteen slices' loaded operands (26 registers each) stayed live and 330 registers spilled
#pragma unroll
            for (int k = 0; k < N / 2; ++k) asm volatile("" : "+v"(w1h[k]), "+v"(w1t[k]));
#endif
        };
#ifdef HJBX_COOP_NO_OUTER
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) w1_part1(s2);
#endif
        COOP_OUTER(coop_outer<NB2, true, false>(acc2h, E0, E1 + col2, acc2t, E0, E1 + col2, rsp, exA, exO, h, w1_part1);)
        // ---- 7. t2 = W2'dh1b, dh2b = t2.s2 -------------------------------------------------------------------------------------------------
        zero16(acc);
        COOP_CHAIN(coop_chain<AO1, BOX, 64>(acc, aW2f, lds3(E0 + exB));)
        f32x16 dh2b;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if constexpr (ACT == HJBX_ACT_TANH) c2r[r] = -2.f * h2r[r] * d2r[r] * acc[r];
            if constexpr (SIN) c2r[r] *= acc[r];
            dh2b[r] = dmul2(r, acc[r]);
        }
        COOP_SYNC();                                                                    // (J) the outer products above have read E1; d1 (E2) is used up
        ex_write(E1 + exW, dh2b);
        if (kh == 0) ex_write(E2 + exWy, dyr);
        COOP_SYNC();                                                                    // (K) E1 = dh2b, E2[0:64] = dy
        COOP_OUTER(coop_outer<NB3, true, false>(acc3h, E1, E2 + col3, acc3t, E1, E2 + col3, rsp, exA, exO, h);)   // (PS = 4: parts 2, 3 compute it too and discard it: no run-time branch here)    // dW3 += dh2b (x) dy
        // the next tile's inputs: issued here, not at the top of the tile -- their N + 2 registers would be live through the phases with the
        // highest register pressure (steps 4-7), and three phases (~3 us) still cover the HBM latency
        fetch(tile + tile_stride, xs_n, dn_n, cst_n);
        // ---- 8. yb = 2 W3'dh2b (halves summed through E0) ------------------------------------------------------------------------------------
        zero16(acc);
        COOP_CHAIN(coop_chain<AO3, BOX, 32>(acc, aW3f, lds3(E1 + 64 * kh * kExLd + exB));)
        if (kh == 1) ex_write(E0 + exWy, acc);                                              // (dh1b's readers, chain 7, finished before (J))
        COOP_SYNC();                                                                    // (L) E0[0:64] = the upper half's partial
        if (kh == 0) {
            ex_add(acc, E0 + exWy);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += acc[r];
            ex_write(E0 + exWy, acc);                                                       // (each wave rewrites exactly the rows it has just read)
        }
        ex_write(E1 + exW, h2r);                                                            // (dh2b's readers, the outer products and chain 8, are past (L))
        COOP_SYNC();                                                                    // (M) E0[0:64] = yb, E1 = h2, E2[0:64] = dy
        COOP_OUTER(coop_outer<NB3, true, true>(acc3h, E1, E0 + col3, acc3t, E1, E2 + col3, rsp, exA, exO, h);)              // dW3 += h2 (x) yb;  dW3_t += h2 (x) (r dy)
        // ---- 9. a2b = (W3 yb).s2 [+ c2] ------------------------------------------------------------------------------------------------------
        zero16(acc);
        COOP_CHAIN(coop_chain<2 * 4, BOX, 32>(acc, aW3b, lds3(E0 + exB));)
        f32x16 a2b;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a2b[r] = dmul2(r, acc[r]);
            if constexpr (ACT != HJBX_ACT_RELU) a2b[r] += c2r[r];
        }
        COOP_SYNC();                                                                    // (N) the outer products above have read E1, E2
        ex_write(E2 + exW, a2b);
        ex_write(E1 + exW, h1r);
        ex_write(E0 + exW, d2r);                                                            // (yb's readers, chain 9 and the outer products, finished before (N))
        COOP_SYNC();                                                                    // (O) E2 = a2b, E1 = h1, E0 = d2
        COOP_OUTER(coop_outer<NB2, true, true>(acc2h, E1, E2 + col2, acc2t, E1, E0 + col2, rsp, exA, exO, h);)     // dW2 += h1 (x) a2b;  dW2_t += h1 (x) (r d2)
        // ---- 10. a1b = (W2 a2b).s1 [+ c1]; dW1 second part ----------------------------------------------------------------------------------
        zero16(acc);
        COOP_CHAIN(coop_chain<2 * 4, BOX, 64>(acc, aW2b, lds3(E2 + exB));)
        f32x16 a1b;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            a1b[r] = dmul1(r, acc[r]);
            if constexpr (ACT != HJBX_ACT_RELU) a1b[r] += c1r[r];
        }
        COOP_SYNC();                                                                    // (P) chain 10 and the outer products above have read E0, E1, E2
        ex_write(E2 + exW, a1b);   // read -- as the second part of dW1, z (x) a1b -- in the shadow of the NEXT tile's chain 2 (visible after its (A)), or below
    }
    COOP_SYNC();
    {   // the last tile's z (x) a1b
        using f32x4 = __attribute__((ext_vector_type(4))) float;
        using LP4 = const __attribute__((address_space(3))) f32x4*;
        LPc a1p = E2g + fW1 * kExLd + sW1;
        const LP4 zz4 = (LP4)(zsg + sW1 * NP);
#ifndef HJBX_COOP_NO_W1
#pragma unroll 4
        for (int s2 = 0; s2 < 16; ++s2) {
            const float a = a1p[s2] * m1;
            const f32x2 a2v{a, a};
#pragma unroll
            for (int k4 = 0; k4 < NP / 4; ++k4) {
                const f32x4 zz = zz4[s2 * (NP / 4) + k4];
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    if (4 * k4 + 2 * c < N) w1h[2 * k4 + c] = __builtin_elementwise_fma(f32x2{zz[2 * c], zz[2 * c + 1]}, a2v, w1h[2 * k4 + c]);
            }
        }
#endif
    }
    COOP_STAMP();                                                                       // tile loop and dW1 tail done
    // ---- partial sums of this workgroup (added in workgroup order by k_train_coop_reduce: deterministic, no float atomics) ------------------
    float* out = partial + (int64_t)blockIdx.x * kCoopBlocks * 1024;
    auto put = [&](int blk, const f32x16& a) {
#pragma unroll
        for (int r = 0; r < 16; ++r) out[blk * 1024 + r * 64 + lane] = a[r];
    };
    if constexpr (PS == 4) {   // only the blocks this part owns are written -- and only those are read: the epilogue kernels know the ownership
        put(w * 4 + part, acc2h[0]);
        put(kCoopSet + w * 4 + part, acc2t[0]);
        if (do3) {
            put(16 + w * 2 + (part & 1), acc3h[0]);
            put(kCoopSet + 16 + w * 2 + (part & 1), acc3t[0]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NB2; ++j) { put(w * 4 + j, acc2h[j]); put(kCoopSet + w * 4 + j, acc2t[j]); }
#pragma unroll
        for (int j = 0; j < NB3; ++j) { put(16 + w * 2 + j, acc3h[j]); put(kCoopSet + 16 + w * 2 + j, acc3t[j]); }
    }
    if (own1) {     // (PS = 4: dW1 and the loss sums belong to part 0)
        float* o1 = partial_w1 + ((int64_t)blockIdx.x * 2 + (tid >> 7)) * (2 * N * 128);
#pragma unroll
        for (int k = 0; k < N; ++k) { o1[k * 128 + fW1] = w1h[k >> 1][k & 1]; o1[(N + k) * 128 + fW1] = w1t[k >> 1][k & 1]; }
    }
    if (w == 0 && own1) {   // loss sums and counts: sample slots -> wave (fixed shuffle tree) -> one record
        double acc_h = h == 0 ? L.sums[0][i] : 0.0, acc_t = h == 0 ? L.sums[1][i] : 0.0;
        double acc_ni = h == 0 ? L.sums[2][i] : 0.0, acc_nd = h == 0 ? L.sums[3][i] : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            acc_h += __shfl_down(acc_h, off, 64); acc_t += __shfl_down(acc_t, off, 64);
            acc_ni += __shfl_down(acc_ni, off, 64); acc_nd += __shfl_down(acc_nd, off, 64);
        }
        if (lane == 0) {
            double* rec = sums_rec + 4 * (int64_t)blockIdx.x;
            rec[0] = acc_h; rec[1] = acc_t; rec[2] = acc_ni; rec[3] = acc_nd;
        }
    }
    COOP_STAMP();                                                                       // partial sums stored
#ifdef HJBX_COOP_STAMPS
    if (stamp_on) stamp_buf[63] = (unsigned long long)stamp_idx;
#endif
}

// Sum of `count` records base[g stride] in a FIXED order: eight interleaved running sums (eight loads in flight: these reductions are latency
// bound at the reference's minibatch), then ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7)).  Two of them at once for the fused epilogue.
__device__ __forceinline__ float coop_sum8(const float* __restrict__ base, int64_t stride, int count) {
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int gI = 0;
    for (; gI + 8 <= count; gI += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s8[k] += base[(int64_t)(gI + k) * stride];
    }
    for (int k = 0; gI < count; ++gI, ++k) s8[k] += base[(int64_t)gI * stride];
    return ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
}
__device__ __forceinline__ void coop_sum8x2(const float* __restrict__ a, const float* __restrict__ b, int64_t stride, int count, float& sa, float& sb) {
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int gI = 0;
    for (; gI + 8 <= count; gI += 8) {
        float va[8], vb[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { va[k] = a[(int64_t)(gI + k) * stride]; vb[k] = b[(int64_t)(gI + k) * stride]; }   // sixteen loads in flight
#pragma unroll
        for (int k = 0; k < 8; ++k) { p[k] += va[k]; q[k] += vb[k]; }
    }
    for (int k = 0; gI < count; ++gI, ++k) { p[k] += a[(int64_t)gI * stride]; q[k] += b[(int64_t)gI * stride]; }
    sa = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    sb = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
}

// dW1: two records (sample halves) per owning workgroup.  psplit = 1: the 2 nparts records in order (one sequence); psplit = 4: the owners are the
// workgroups 0, 4, 8, ..., i.e. the records 8 ts + half -- the two halves as two sequences, added at the end
template <int N> __device__ __forceinline__ float coop_sum_w1(const float* __restrict__ base /* partial_w1 + (set, k, f) */, int nparts, int psplit) {
    if (psplit != 4) return coop_sum8(base, 2 * N * 128, 2 * nparts);
    float s0, s1;
    coop_sum8x2(base, base + 2 * N * 128, (int64_t)8 * (2 * N * 128), nparts / 4, s0, s1);
    return s0 + s1;
}

// Which workgroups hold a block of the partial sums.  psplit = 1: every workgroup, all of it.  psplit = 4 (k_train_coop<.., PS = 4, ..>: four
// workgroups per tile): output block b of dW2 / dW3 lives only in the workgroups whose part is b's column block, dW1 and the loss sums only in
// part 0 -- a quarter of the records to add at the reference's minibatch.  -> first workgroup, step between workgroups, number of them
struct CoopOwners { int first, step, count; };
__device__ __forceinline__ CoopOwners coop_owners(int b /* block within its set, or -1: dW1 / loss sums */, int nparts, int psplit) {
    if (psplit != 4) return CoopOwners{0, 1, nparts};
    const int jb = b < 0 ? 0 : (b < 16 ? (b & 3) : ((b - 16) & 1));
    return CoopOwners{jb, 4, nparts / 4};
}

// ---- partial sums -> flat gradient buffer, in workgroup order ---------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(256) void k_train_coop_reduce(const float* __restrict__ partial, const float* __restrict__ partial_w1, int nparts, int psplit,
                                                          const double* __restrict__ sums_rec, float* __restrict__ flat) {
    constexpr int P = N * kH1 + kH1 * kH2 + kH2 * kH3;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t < kCoopBlocks * 1024) {
        const CoopOwners ow = coop_owners((t >> 10) % kCoopSet, nparts, psplit);
        const float s = coop_sum8(partial + (int64_t)ow.first * kCoopBlocks * 1024 + t, (int64_t)ow.step * kCoopBlocks * 1024, ow.count);
        const int blk = t >> 10, reg = (t >> 6) & 15, lane = t & 63;
        const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5), col = lane & 31;
        const int set = blk / kCoopSet, b = blk % kCoopSet;
        float* o = flat + set * P;
        if (b < 16) o[N * kH1 + (32 * (b >> 2) + row) * kH2 + 32 * (b & 3) + col] = s;
        else o[N * kH1 + kH1 * kH2 + (32 * ((b - 16) >> 1) + row) * kH3 + 32 * ((b - 16) & 1) + col] = s;
    } else if (t < kCoopBlocks * 1024 + 2 * N * 128) {
        const int u = t - kCoopBlocks * 1024;          // (set, k, f)
        const int set = u / (N * 128), kf = u % (N * 128);
        flat[set * P + kf] = coop_sum_w1<N>(partial_w1 + u, nparts, psplit);
    } else if (t < kCoopBlocks * 1024 + 2 * N * 128 + 4) {
        const int k = t - (kCoopBlocks * 1024 + 2 * N * 128);
        const CoopOwners ow = coop_owners(-1, nparts, psplit);
        double s = 0;
        for (int g = 0; g < ow.count; ++g) s += sums_rec[4 * (ow.first + g * ow.step) + k];
        flat[2 * P + k] = (float)s;
    }
}

// ---- partial sums -> mixed gradient -> Adam, in workgroup order, one launch (hjbx_value_loss_adam_f32) ------------------------------------
// The same sums in the same order as k_train_coop_reduce, but each thread takes ONE parameter of BOTH sets (hjb, termination), divides by the
// counts, mixes (vhjb.py:241, 253, 284) and applies optax.adam's update (hjbx_adam.hpp) to it: the flat buffer is never written.  At the
// reference's minibatch this removes one launch-bound kernel from the update (gather -> gradient -> this).
template <int N>
__global__ __launch_bounds__(256) void k_train_coop_update(const float* __restrict__ partial, const float* __restrict__ partial_w1, int nparts, int psplit,
                                                          const double* __restrict__ sums_rec, AdamArgs a, MixArgs mx, GatherArgs next) {
    __shared__ float sc[2];
    __shared__ double tot[4];
    __shared__ double rec[4 * kCoopMaxGrid];          // the loss-sum records of the workgroups (at most one workgroup per CU)
    // Order of work: everything that needs no other thread first -- thread 0's bias corrections (double pow), the records into LDS, and each
    // thread's own long sums over the workgroups -- then ONE barrier, the four scalar totals, a second barrier, mix + Adam.  (Written in
    // program order -- coefficients, totals, sums -- every thread waited through two serial latencies before it started its loads: 54 us per
    // 256-sample update instead of 47.)
    const float tstep = adam_coef_begin(a, sc);
    for (int idx = threadIdx.x; idx < 4 * nparts; idx += 256) rec[idx] = sums_rec[idx];
    const int t = blockIdx.x * 256 + threadIdx.x;
    float sh = 0.f, st = 0.f;
    int which = -1;
    int64_t j = 0;
    if (t < kCoopSet * 1024) {
        const int b = t >> 10, reg16 = (t >> 6) & 15, lane = t & 63;
        const CoopOwners ow = coop_owners(b, nparts, psplit);
        const float* src = partial + (int64_t)ow.first * kCoopBlocks * 1024 + t;
        coop_sum8x2(src, src + kCoopSet * 1024, (int64_t)ow.step * kCoopBlocks * 1024, ow.count, sh, st);
        const int row = (reg16 & 3) + 8 * (reg16 >> 2) + 4 * (lane >> 5), col = lane & 31;
        if (b < 16) { which = 1; j = (int64_t)(32 * (b >> 2) + row) * kH2 + 32 * (b & 3) + col; }
        else { which = 2; j = (int64_t)(32 * ((b - 16) >> 1) + row) * kH3 + 32 * ((b - 16) & 1) + col; }
    } else if (t < kCoopSet * 1024 + N * 128) {
        const int kf = t - kCoopSet * 1024;            // (k, f) of W1
        sh = coop_sum_w1<N>(partial_w1 + kf, nparts, psplit);
        st = coop_sum_w1<N>(partial_w1 + N * 128 + kf, nparts, psplit);
        which = 0; j = kf;
    }
    __syncthreads();
    if (threadIdx.x < 4) {   // loss sums and counts: the records in workgroup order, like k_train_coop_reduce (every workgroup computes them)
        const CoopOwners ow = coop_owners(-1, nparts, psplit);
        double s = 0;
        for (int g = 0; g < ow.count; ++g) s += rec[4 * (ow.first + g * ow.step) + threadIdx.x];
        tot[threadIdx.x] = (double)(float)s;          // (the flat buffer holds them as float32)
    }
    __syncthreads();
    const AdamCoef c = adam_coef_end(a, tstep, sc);
    const float reg = mx.reg_dev ? mx.reg_dev[0] : mx.reg_host;
    const float ih = 1.0f / ((float)tot[2] + mx.eps), it = 1.0f / ((float)tot[3] + mx.eps);
    const float wt = reg * it;
    if (which >= 0) adam_element(a, c, which, j, sh * ih + st * wt);
    // the NEXT update's minibatch (index = this update's + 1; every workgroup reads the counter before the last one increments it below): its
    // rows by all threads, its regularisation weight by the last workgroup -- the buffer may be the one this launch read `reg` from
    const int64_t k_next = next.enabled && mx.step_counter ? (int64_t)mx.step_counter[0] + 1 : 0;
    if (next.enabled) gather_elements(next, k_next, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
    if (adam_finish(a, c, mx, (float)tot[0] * ih, (float)tot[1] * it, reg) && next.enabled) gather_reg(next, k_next);
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
struct CoopWs { size_t partial, partial_w1, sums, total; int grid, psplit; };
static CoopWs coop_ws(int64_t B, int n) {
    CoopWs w{};
    int n_cu = hjbx_device_cus();
    if (n_cu <= 0) n_cu = 256;
    const int64_t ntiles = (B + 31) / 32;
    w.psplit = 4 * ntiles <= n_cu ? 4 : 1;                             // workgroups per tile (small batches: see k_train_coop)
    w.grid = (int)(ntiles * w.psplit < n_cu ? ntiles * w.psplit : n_cu);
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    w.partial = up((size_t)w.grid * kCoopBlocks * 1024 * sizeof(float));
    w.partial_w1 = up((size_t)w.grid * 2 * 2 * n * 128 * sizeof(float));
    w.sums = up((size_t)w.grid * 4 * sizeof(double));
    w.total = w.partial + w.partial_w1 + w.sums;
    return w;
}

size_t hjbx_train_coop_workspace_bytes(int64_t B, int n) { return B > 0 ? coop_ws(B, n).total : 0; }

template <typename S>
static int launch_coop(const hjbx_system* sysh, S sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost,
                       const float* done, float* flat, void* workspace, int64_t B, void* st, const FuseArgs* fuse) {
    constexpr int N = S::N, M = S::M;
    if constexpr (N % 2 != 0 || N > HJBX_MAX_N) {
        return HJBX_EUNSUPPORTED;
    } else {
        if (hjbx_device_cus() <= 0) return hjbx_set_error(HJBX_ENODEVICE, "hjbx_value_loss_grad_f32: no HIP device");
        MlpP<N> p;
        for (int k = 0; k < N; ++k) { p.mean[k] = (float)mlp->mean[k]; p.istd[k] = (float)(1.0 / mlp->std[k]); p.xf[k] = (float)mlp->xf[k]; }
        p.eps_s = (float)mlp->eps_scalar;
        const auto tk = make_task<float, N, M>(task);
        const auto lim = make_limits<float, M>(sysh);
        const CoopWs w = coop_ws(B, N);
        float* partial = (float*)workspace;
        float* partial_w1 = (float*)((char*)workspace + w.partial);
        double* sums = (double*)((char*)workspace + w.partial + w.partial_w1);
        const int64_t ntiles = (B + 31) / 32;
        const float *W1 = (const float*)mlp->W1, *W2 = (const float*)mlp->W2, *W3 = (const float*)mlp->W3;
        hipStream_t s = (hipStream_t)st;
        auto go = [&](auto mode_c, auto act_c) {
            if (w.psplit == 4)
                hipLaunchKernelGGL((k_train_coop<decltype(mode_c)::value, decltype(act_c)::value, 4, S>), dim3(w.grid), dim3(256), 0, s, sys, p, tk, lim, W1, W2, W3,
                                   x, cost, done, (float)task->eps, partial, partial_w1, sums, B, ntiles);
            else
                hipLaunchKernelGGL((k_train_coop<decltype(mode_c)::value, decltype(act_c)::value, 1, S>), dim3(w.grid), dim3(256), 0, s, sys, p, tk, lim, W1, W2, W3,
                                   x, cost, done, (float)task->eps, partial, partial_w1, sums, B, ntiles);
        };
        auto with_act = [&](auto mode_c) {
            if (mlp->activation == HJBX_ACT_TANH) go(mode_c, std::integral_constant<int, HJBX_ACT_TANH>{});
            else if (mlp->activation == HJBX_ACT_SIN) {
                // sin keeps act' = cos(a) beside the activations (32 registers) and its second-order factors from steps 4 / 5 on: with the
                // state-sized registers of a 6-D or 10-D residual that is 9-12 registers over the 512 of a wave -- n <= 4 only (the network
                // belongs to the 2-D double integrator of the time-optimal notebook); larger systems keep the autograd path
                if constexpr (N <= 4) go(mode_c, std::integral_constant<int, HJBX_ACT_SIN>{});
            }
            else go(mode_c, std::integral_constant<int, HJBX_ACT_RELU>{});
        };
        if (mode == HJBX_RESIDUAL_NORMALISED) with_act(std::integral_constant<int, 0>{});
        else with_act(std::integral_constant<int, 1>{});
        if (fuse) {
            if (fuse->a.end0 != N * kH1 || fuse->a.end1 - fuse->a.end0 != kH1 * kH2 || fuse->a.P - fuse->a.end1 != kH2 * kH3)
                return hjbx_set_error(HJBX_EINVAL, "hjbx_value_loss_adam_f32: the Adam state's tensors must be W1 (%d x 128), W2 (128 x 128), W3 (128 x 64)", N);
            if (w.grid > kCoopMaxGrid) return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_loss_adam_f32: %d workgroups (more than %d CUs?)", w.grid, kCoopMaxGrid);
            const int nthreads = kCoopSet * 1024 + N * 128;
            hipLaunchKernelGGL((k_train_coop_update<N>), dim3((nthreads + 255) / 256), dim3(256), 0, s, partial, partial_w1, w.grid, w.psplit, sums, fuse->a, fuse->mx, fuse->next);
        } else {
            const int nthreads = kCoopBlocks * 1024 + 2 * N * 128 + 4;
            hipLaunchKernelGGL((k_train_coop_reduce<N>), dim3((nthreads + 255) / 256), dim3(256), 0, s, partial, partial_w1, w.grid, w.psplit, sums, flat);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hjbx_set_error(HJBX_EHIP, "hjbx_value_loss_grad_f32: %s", hipGetErrorString(e));
        return HJBX_OK;
    }
}

// called by hjbx_value_loss_grad_f32 (hjbx_train.hip) after it has validated its arguments
int hjbx_train_coop(const hjbx_system* sys, const hjbx_task* task, const hjbx_mlp* mlp, int mode, const float* x, const float* cost, const float* done,
                    float* flat, void* workspace, int64_t B, void* stream, const FuseArgs* fuse) {
    if (mlp->activation == HJBX_ACT_SIN && sys->n > 4)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_loss_grad_f32: the sin network's fused parameter gradient exists for n <= 4 (n = %d)", sys->n);
    int rc = HJBX_EUNSUPPORTED;
#ifdef HJBX_TRAIN_DEV   // development builds: cartpole and the 10-D quadcopter only
    bool ok = false;
    if (sys->kind == HJBX_SYS_CARTPOLE) {
        Cartpole<float> cp{(float)sys->p[0], (float)sys->p[1], (float)sys->p[2], (float)sys->p[3]};
        rc = launch_coop<Cartpole<float>>(sys, cp, task, mlp, mode, x, cost, done, flat, workspace, B, stream, fuse);
        ok = true;
    } else if (sys->kind == HJBX_SYS_NEARHOVER) {
        NearHover<float> q{(float)sys->p[0], (float)sys->p[1], (float)sys->p[2], (float)sys->p[3]};
        rc = launch_coop<NearHover<float>>(sys, q, task, mlp, mode, x, cost, done, flat, workspace, B, stream, fuse);
        ok = true;
    }
#else
    const bool ok = with_system<float>(sys, [&](auto S) { rc = launch_coop<decltype(S)>(sys, S, task, mlp, mode, x, cost, done, flat, workspace, B, stream, fuse); });
#endif
    if (!ok || rc == HJBX_EUNSUPPORTED)
        return hjbx_set_error(HJBX_EUNSUPPORTED, "hjbx_value_loss_grad_f32: no kernel for system kind %d with n=%d m=%d", sys->kind, sys->n, sys->m);
    return rc;
}
