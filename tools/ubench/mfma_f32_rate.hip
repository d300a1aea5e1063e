// Diagnostic micro-benchmark (not part of the product): issue interval of v_mfma_f32_32x32x2_f32 on gfx950 with
// 1 or 2 waves per SIMD, 4 independent accumulators per wave, operands in registers.  Prints cycles per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
template <int NACC> __global__ void k(float* out, unsigned long long* stamps, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float av = 1.0f + threadIdx.x * 1e-3f, bv = 0.5f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int NACC> void run(int threads, const char* tag) {
    const int blocks = 256, iters = 2000;
    float* out; unsigned long long* st;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&st, blocks * (threads / 64) * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, out, st, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * (threads / 64));
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0; for (auto v : h) sum += (double)v;
    const double cyc = sum / h.size();
    const int waves_per_simd = threads / 256;
    const double n_mfma = (double)iters * 8 * NACC;
    printf("%s: %d waves/SIMD, %d accumulators: %.1f cycles per MFMA per wave, %.1f per SIMD\n", tag, waves_per_simd, NACC, cyc / n_mfma,
           cyc / n_mfma / waves_per_simd);
    hipFree(out); hipFree(st);
}
int main() {
    run<4>(256, "f32 32x32x2"); run<4>(512, "f32 32x32x2"); run<2>(512, "f32 32x32x2"); run<1>(512, "f32 32x32x2"); run<1>(256, "f32 32x32x2");
    return 0;
}
