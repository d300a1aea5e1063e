// Diagnostic micro-benchmark (not part of the product): what the instructions between groups of 4 f32 MFMAs cost
// one wave per SIMD (256 threads) or two (512).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
template <int VAR> __global__ void k(float* out, unsigned long long* stamps, int iters, const float* src) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = src[i];
    __syncthreads();
    f32x16 acc[4], prev[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) { acc[a][r] = 0.f; prev[a][r] = src[(threadIdx.x + r + 16 * a) & 4095]; }
    float a0 = 1.0f + threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 0.5f;
    float sink = 0.f;
    const uint32_t base = (uint32_t)(uintptr_t)lds + 4 * (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if constexpr (VAR == 1) {  // 4 independent VALU
                sink = fmaxf(sink, a0); sink = fmaxf(sink, a1); sink = fmaxf(sink, a2); sink = fmaxf(sink, a3);
            }
            if constexpr (VAR == 2 || VAR == 3 || VAR == 4) b = fmaxf(prev[u >> 2][(u * 4) & 15], 0.f);  // VALU writes next b
            if constexpr (VAR == 3 || VAR == 4) {
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a0) : "v"(base), "i"(0 + 1024 * (0)));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a1) : "v"(base), "i"(256));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a2) : "v"(base), "i"(512));
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a3) : "v"(base), "i"(768));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            }
            if constexpr (VAR == 4) {  // + mask building like the kernel
                sink += (prev[u >> 2][(u * 4 + 1) & 15] > 0.f ? 1.f : 0.f);
            }
            acc[0] = MFMA(a0, b, acc[0]); acc[1] = MFMA(a1, b, acc[1]); acc[2] = MFMA(a2, b, acc[2]); acc[3] = MFMA(a3, b, acc[3]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sink;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int VAR> void run(int threads, const char* tag) {
    const int blocks = 256, iters = 500;
    float *out, *src; unsigned long long* st;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&st, blocks * (threads / 64) * 8); hipMalloc(&src, 4096 * 4);
    std::vector<float> hs(4096); for (int i = 0; i < 4096; ++i) hs[i] = (i % 7) - 3.0f;
    hipMemcpy(src, hs.data(), 4096 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(threads), 0, 0, out, st, iters, src);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double cyc = sum / h.size(), n_mfma = (double)iters * 16 * 4;
    printf("%-44s %d waves/SIMD: %.1f cycles per MFMA per SIMD\n", tag, threads / 256, cyc / n_mfma / (threads / 256));
    hipFree(out); hipFree(st); hipFree(src);
}
int main() {
    for (int t : {256, 512}) {
        run<0>(t, "pure MFMA x4");
        run<1>(t, "+ 4 independent VALU per group");
        run<2>(t, "+ VALU writing the next B operand");
        run<3>(t, "+ that + 4 asm ds_read_b32 + lgkmcnt(0)");
        run<4>(t, "+ that + cmp/cndmask/add");
    }
    return 0;
}
