// Diagnostic micro-benchmark (not part of the product): issue interval of v_mfma_f32_32x32x16_f16 (8 passes) when consecutive MFMAs
// accumulate into the SAME registers (the f16x2 kernel's unit: three piece products on one accumulator block) against round-robin over 2 / 4
// accumulator blocks, one and two waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_f16_dep.bin mfma_f16_dep.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
template <int VAR> __global__ void k(float* out, unsigned long long* stamps, int iters, const float* src) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    f16x8 A[3], B[2];
    for (int j = 0; j < 8; ++j) {
        A[0][j] = (_Float16)src[(threadIdx.x + j) & 4095]; A[1][j] = (_Float16)src[(threadIdx.x + j + 8) & 4095]; A[2][j] = (_Float16)src[(threadIdx.x + j + 16) & 4095];
        B[0][j] = (_Float16)src[(threadIdx.x + j + 24) & 4095]; B[1][j] = (_Float16)src[(threadIdx.x + j + 32) & 4095];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (VAR == 0) {          // 12 MFMAs on ONE accumulator block
#pragma unroll
                for (int q = 0; q < 12; ++q) acc[0] = MFMA(A[q % 3], B[q & 1], acc[0]);
            } else if constexpr (VAR == 1) {   // the kernel's order: 3 dependent MFMAs per block, 4 blocks
#pragma unroll
                for (int o = 0; o < 4; ++o) { acc[o] = MFMA(A[1], B[0], acc[o]); acc[o] = MFMA(A[0], B[1], acc[o]); acc[o] = MFMA(A[0], B[0], acc[o]); }
            } else if constexpr (VAR == 2) {   // pairs of blocks alternating
#pragma unroll
                for (int o = 0; o < 4; o += 2) {
                    acc[o] = MFMA(A[1], B[0], acc[o]); acc[o + 1] = MFMA(A[2], B[0], acc[o + 1]);
                    acc[o] = MFMA(A[0], B[1], acc[o]); acc[o + 1] = MFMA(A[1], B[1], acc[o + 1]);
                    acc[o] = MFMA(A[0], B[0], acc[o]); acc[o + 1] = MFMA(A[2], B[1], acc[o + 1]);
                }
            } else {                           // round-robin over the 4 blocks
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int o = 0; o < 4; ++o) acc[o] = MFMA(A[p], B[p & 1], acc[o]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int VAR> void run(int threads, const char* tag) {
    const int blocks = 256, iters = 400;
    float *out, *src; unsigned long long* st;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&st, blocks * (threads / 64) * 8); hipMalloc(&src, 4096 * 4);
    std::vector<float> hs(4096); for (int i = 0; i < 4096; ++i) hs[i] = ((i * 37) % 113) / 113.0f - 0.5f;
    hipMemcpy(src, hs.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 0, 0, out, st, iters, src);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 0, 0, out, st, iters, src);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double ticks = sum / h.size(), n_mfma = (double)iters * 8 * 12;
    // s_memtime ticks at a fixed rate; the wall time of the launch gives the MFMA rate per SIMD directly
    printf("%-52s %d wave(s)/SIMD: %.2f memtime ticks per MFMA per SIMD, %.1f ns per MFMA per SIMD (wall)\n", tag, threads / 256, ticks / n_mfma / (threads / 256),
           ms * 1e6 / (n_mfma * (threads / 256)));
    hipFree(out); hipFree(st); hipFree(src);
}
int main() {
    for (int threads : {256, 512}) {
        run<0>(threads, "12 MFMAs on one accumulator block");
        run<1>(threads, "3 dependent MFMAs per block, 4 blocks (the kernel)");
        run<2>(threads, "two blocks alternating");
        run<3>(threads, "round-robin over 4 blocks");
    }
    return 0;
}
