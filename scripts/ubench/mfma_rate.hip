// Microbenchmark: issue rate of f32-input MFMAs on gfx950 (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    float s = 0; for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int e = 0; e < 16; ++e) acc[t][e] = 0;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    unsigned long long t1 = clock64();
    float s = 0; for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
template <int NACC, int K32>
__global__ __launch_bounds__(512) void kh(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
    f16x4 a4, b4; f16x8 a8, b8;
    for (int e = 0; e < 4; ++e) { a4[e] = (_Float16)(threadIdx.x * 1e-3f); b4[e] = (_Float16)(1.0f + e); }
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(threadIdx.x * 1e-3f); b8[e] = (_Float16)(1.0f + e); }
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if constexpr (K32) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[t], 0, 0, 0);
            }
    }
    unsigned long long t1 = clock64();
    float s = 0; for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <class K>
void run(const char* name, K kern, int threads, int nacc, int flop_per_mfma) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<256, threads>>>(out, iters, cyc);
    hipEventRecord(e0); kern<<<256, threads>>>(out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < 256; ++i) c += h[i]; c /= 256;
    const double n_mfma_wave = (double)iters * 4 * nacc;
    const double waves_per_simd = threads / 64 / 4.0;
    printf("%-28s threads %3d: %.1f clk per MFMA per wave, %.1f clk per MFMA per SIMD, %.1f TFLOP/s, clock %.2f GHz\n", name, threads,
           c / n_mfma_wave, c / (n_mfma_wave * waves_per_simd), 256.0 * (threads / 64) * n_mfma_wave * flop_per_mfma / (ms * 1e-3) / 1e12,
           c / (ms * 1e-3) / 1e9);
    hipFree(out); hipFree(cyc);
}
int main() {
    run("16x16x4 f32, 5 acc", k16<5>, 256, 5, 2048); run("16x16x4 f32, 5 acc", k16<5>, 512, 5, 2048);
    run("16x16x4 f32, 1 acc", k16<1>, 256, 1, 2048); run("16x16x4 f32, 2 acc", k16<2>, 256, 2, 2048);
    run("32x32x2 f32, 2 acc", k32<2>, 256, 2, 4096); run("32x32x2 f32, 2 acc", k32<2>, 512, 2, 4096);
    run("32x32x2 f32, 1 acc", k32<1>, 256, 1, 4096);
    run("16x16x16 f16, 5 acc", kh<5, 0>, 256, 5, 8192); run("16x16x16 f16, 1 acc (dependent)", kh<1, 0>, 256, 1, 8192);
    run("16x16x32 f16, 5 acc", kh<5, 1>, 256, 5, 16384); run("16x16x32 f16, 1 acc (dependent)", kh<1, 1>, 256, 1, 16384);
    run("16x16x32 f16, 2 acc", kh<2, 1>, 256, 2, 16384); run("16x16x16 f16, 2 acc", kh<2, 0>, 256, 2, 8192);
    return 0;
}
