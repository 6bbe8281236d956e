// Two waves per SIMD, each alternating a burst of 20 independent-enough MFMAs (5 accumulators x 4 steps) with a gap of
// scalar / vector work: how much of the MFMA pipe is lost, with and without s_setprio (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int GAP_S, int GAP_V, int PRIO, int GAP_L = 0>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned long long* cyc) {
    __shared__ f32x4 lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = f32x4{1, 2, 3, 4};
    __syncthreads();
    f32x4 ld[8];
    for (int i = 0; i < 8; ++i) ld[i] = f32x4{0, 0, 0, 0};
    f32x4 acc[5];
    for (int t = 0; t < 5; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, v = 1.0f;
    if (PRIO && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
#pragma unroll
        for (int gsi = 0; gsi < GAP_S; ++gsi) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc");
#pragma unroll
        for (int gvi = 0; gvi < GAP_V; ++gvi) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
#pragma unroll
        for (int gl = 0; gl < GAP_L; ++gl) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ld[gl]) : "v"((threadIdx.x & 63) * 16), "n"(gl * 1024));
        if (GAP_L) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = clock64();
    float s = v; for (int t = 0; t < 5; ++t) s += acc[t][0] + acc[t][3];
    for (int i = 0; i < 8; ++i) s += ld[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}
template <class K>
void run(const char* name, K kern, int threads) {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 500;
    kern<<<256, threads>>>(out, iters, cyc); (void)hipDeviceSynchronize();
    kern<<<256, threads>>>(out, iters, cyc); (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int waves = threads / 64;
    double cmax = 0; for (int i = 0; i < waves; ++i) cmax = h[i] > cmax ? h[i] : cmax;
    const double blocks_per_simd = (double)iters * waves / 4.0;
    printf("%-46s %d waves/SIMD: %.0f clk per 20-MFMA block per SIMD (ideal 640); wave ends", name, waves / 4, cmax / blocks_per_simd);
    for (int i = 0; i < waves; ++i) printf(" %llu", h[i] / 1000);
    printf(" k\n");
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run("no gap", k<0, 0, 0>, 256); run("no gap", k<0, 0, 0>, 512);
    run("gap 12 SALU", k<12, 0, 0>, 256); run("gap 12 SALU", k<12, 0, 0>, 512); run("gap 12 SALU, prio", k<12, 0, 1>, 512);
    run("gap 12 SALU + 4 VALU", k<12, 4, 0>, 512); run("gap 12 SALU + 4 VALU, prio", k<12, 4, 1>, 512);
    run("gap 24 SALU + 8 VALU", k<24, 8, 0>, 256); run("gap 24 SALU + 8 VALU", k<24, 8, 0>, 512); run("gap 24 SALU + 8 VALU, prio", k<24, 8, 1>, 512);
    run("gap 5 ds_read_b128 (+wait)", k<0, 0, 0, 5>, 256); run("gap 5 ds_read_b128 (+wait)", k<0, 0, 0, 5>, 512);
    run("gap 12 SALU + 2 VALU + 5 ds_read_b128", k<12, 2, 0, 5>, 512);
    run("gap 20 VALU", k<0, 20, 0>, 512); run("gap 20 VALU, prio", k<0, 20, 1>, 512);
    return 0;
}
