// Diagnostic micro-benchmark (not part of the product): do the VALU-only phases between MFMA chains of one wave overlap the
// MFMA chains of its SIMD partner?  Each wave loops { chain of CH MFMAs (4 accumulators) ; VN VALU ops on the accumulators }.
// Reported: matrix-pipe cycles per MFMA per SIMD (64 = the pipe never idles).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_phase tools/ubench/mfma_phase.hip && ./mfma_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int CH, int VN, int MODE> __global__ __launch_bounds__(1024) void k(float* out, unsigned long long* stamps, int iters, const float* src) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = src[(threadIdx.x + r + 16 * a) & 4095];
    float a0 = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    const int wave = threadIdx.x >> 6;
    if (MODE == 1 && wave >= 4) {  // stagger: the second wave of each SIMD starts half a chain + half a VALU phase late
        for (int u = 0; u < CH / 8; ++u) {
            acc[0] = MFMA(a0, b, acc[0]); acc[1] = MFMA(a0, b, acc[1]); acc[2] = MFMA(a0, b, acc[2]); acc[3] = MFMA(a0, b, acc[3]);
        }
    }
    if (MODE == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);  // static priority for the second wave
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // one 100 MHz counter for the whole chip
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) __builtin_amdgcn_s_setprio(0);
        if (MODE == 4) __builtin_amdgcn_s_setprio(3);  // chains run at raised priority: a ready MFMA wins the issue port
#pragma unroll
        for (int u = 0; u < CH / 4; ++u) {
            acc[0] = MFMA(a0, b, acc[0]); acc[1] = MFMA(a0, b, acc[1]); acc[2] = MFMA(a0, b, acc[2]); acc[3] = MFMA(a0, b, acc[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 3) __builtin_amdgcn_s_setprio(2);  // VALU phases run at raised priority
        if (MODE == 4) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int v = 0; v < VN; ++v) {
            acc[(v >> 4) & 3][v & 15] = __builtin_fmaxf(acc[(v >> 4) & 3][v & 15] * 0.999f, -1.0f);  // 2 VALU ops per element
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        stamps[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = t0;
        stamps[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = t1;
    }
}

static double g_ref = 0;  // s_memtime ticks per MFMA of the pure stream (== 64 core cycles)
template <int CH, int VN, int MODE> void run(int threads, const char* tag) {
    const int blocks = 256, iters = 400000 / CH;
    float *out, *src;
    unsigned long long* st;
    hipMalloc(&out, blocks * threads * 4);
    hipMalloc(&st, blocks * (threads / 64) * 16);
    hipMalloc(&src, 4096 * 4);
    std::vector<float> hs(4096);
    for (int i = 0; i < 4096; ++i) hs[i] = (i % 7) - 3.0f;
    hipMemcpy(src, hs.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k<CH, VN, MODE>), dim3(blocks), dim3(threads), 0, 0, out, st, iters, src);
    hipDeviceSynchronize();
    const int wpb = threads / 64;
    std::vector<unsigned long long> h(2 * blocks * wpb);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    // per workgroup (= CU): span from the first wave's start to the last wave's end, against the MFMAs each SIMD issued
    double sum = 0;
    for (int b = 0; b < blocks; ++b) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int w = 0; w < wpb; ++w) {
            lo = std::min(lo, h[2 * (b * wpb + w)]);
            hi = std::max(hi, h[2 * (b * wpb + w) + 1]);
        }
        sum += (double)(hi - lo);
    }
    const double ticks = sum / blocks / ((double)iters * CH * (threads / 256));   // 100 MHz ticks per MFMA per SIMD
    if (g_ref == 0) g_ref = ticks;
    printf("%-34s CH=%4d VALU=%4d  %d waves/SIMD: %6.1f cycles per MFMA per SIMD  (pipe busy %.1f %%)\n", tag, CH, 2 * VN, threads / 256,
           64.0 * ticks / g_ref, 100.0 * g_ref / ticks);
    hipFree(out); hipFree(st); hipFree(src);
}

int main() {
    run<256, 0, 0>(256, "pure MFMA (reference)");
    run<256, 0, 0>(512, "pure MFMA");
    for (int t : {256, 512}) {
        run<256, 64, 0>(t, "chain + VALU");
        run<256, 256, 0>(t, "chain + VALU");
        run<128, 256, 0>(t, "chain + VALU");
    }
    for (int t : {768, 1024}) {   // three / four waves per SIMD: does a third wave hide the VALU phases better?
        run<256, 64, 0>(t, "chain + VALU");
        run<256, 256, 0>(t, "chain + VALU");
        run<128, 256, 0>(t, "chain + VALU");
    }
    run<256, 64, 1>(512, "staggered partner");
    run<256, 256, 1>(512, "staggered partner");
    run<128, 256, 1>(512, "staggered partner");
    run<256, 64, 2>(512, "partner at s_setprio 1");
    run<256, 256, 2>(512, "partner at s_setprio 1");
    run<128, 256, 2>(512, "partner at s_setprio 1");
    run<256, 64, 3>(512, "VALU phases at s_setprio 2");
    run<256, 256, 3>(512, "VALU phases at s_setprio 2");
    run<128, 256, 3>(512, "VALU phases at s_setprio 2");
    run<256, 64, 4>(512, "MFMA chains at s_setprio 3");
    run<256, 256, 4>(512, "MFMA chains at s_setprio 3");
    run<128, 256, 4>(512, "MFMA chains at s_setprio 3");
    return 0;
}
