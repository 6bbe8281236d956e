// float64 VALU / DPP / MFMA issue-rate and latency calibration for one wave on gfx950 (development aid for the
// factorisation's diagonal pass).   hipcc -O3 --offload-arch=gfx950 f64_rate.hip -o f64_rate && ./f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
using f64x4 = __attribute__((ext_vector_type(4))) double;

#define BENCH(name, body)                                                                              \
    __global__ void name(double* out, long long* clk, int n) {                                        \
        double a = threadIdx.x * 1.0 + 1.0, b = 1.0001, c = 0.5, d = 0.25, e = 0.125, f = 3.0;         \
        float fa = 1.0f + threadIdx.x;                                                                  \
        f64x4 acc = {0.0, 0.0, 0.0, 0.0};                                                              \
        const long long t0 = clock64();                                                                \
        for (int i = 0; i < n; ++i) { body }                                                           \
        const long long t1 = clock64();                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + fa + acc[0] + acc[1] + acc[2] + acc[3]; \
        if ((threadIdx.x & 63) == 0) clk[threadIdx.x / 64] = t1 - t0;                                  \
    }

BENCH(k_f32_dep, REP64(asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(fa));))
BENCH(k_fma_dep, REP64(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
BENCH(k_fma_indep, REP16(asm volatile("v_fmac_f64 %0, %4, %5\n v_fmac_f64 %1, %4, %5\n v_fmac_f64 %2, %4, %5\n v_fmac_f64 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));))
BENCH(k_fmac_dpp, REP16(asm volatile("v_fmac_f64_dpp %0, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %2, -%4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, -%4, %5 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));))
BENCH(k_movdpp_2fmac, REP16(asm volatile("v_mov_b64_dpp v[100:101], %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64 %0, v[100:101], %5\n v_fmac_f64 %1, v[100:101], %4\n v_mov_b64_dpp v[102:103], %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64 %2, v[102:103], %5\n v_fmac_f64 %3, v[102:103], %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "v100", "v101", "v102", "v103");))
BENCH(k_readlane_fmac, REP16(asm volatile("v_readlane_b32 s20, %2, 3\n v_readlane_b32 s21, %3, 3\n s_nop 0\n v_fmac_f64 %0, s[20:21], %4\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s23, %3, 5\n s_nop 0\n v_fmac_f64 %1, s[22:23], %4" : "+v"(a), "+v"(b) : "v"(__double2loint(e)), "v"(__double2hiint(e)), "v"(f) : "s20", "s21", "s22", "s23");))
BENCH(k_mul_dep, REP64(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));))
BENCH(k_rsq_dep, REP64(asm volatile("v_rsq_f64 %0, %0\n s_nop 0" : "+v"(a));))
BENCH(k_rsq_f32_dep, REP64(asm volatile("v_rsq_f32 %0, %0\n s_nop 0" : "+v"(fa));))
BENCH(k_movdpp_dep, REP64(asm volatile("s_nop 1\n v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a));))
BENCH(k_mfma_dep, REP16(acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);))
BENCH(k_cvt_rint, REP16(asm volatile("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));))

template <class K>
static void run(const char* name, K kern, int per_iter, double* out, long long* clk) {
    for (int threads : {64, 256, 1024}) {
        std::vector<long long> h(16);
        const int n = 64;
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, clk, n); (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, clk, n); (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), clk, 128, hipMemcpyDeviceToHost);
        printf("%-40s %4d threads: %7.2f ticks per group (wave 0)", name, threads, (double)h[0] / (n * per_iter));
        if (threads == 1024) printf("   waves 4 / 8 / 12 (its SIMD): %.2f %.2f %.2f", (double)h[4] / (n * per_iter), (double)h[8] / (n * per_iter), (double)h[12] / (n * per_iter));
        printf("\n");
    }
}

int main() {
    double* out; long long* clk;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&clk, 4096);
    run("v_fma_f32 dependent (calibration)", k_f32_dep, 64, out, clk);
    run("v_fma_f64 dependent", k_fma_dep, 64, out, clk);
    run("v_fmac_f64 x4 independent (per 4)", k_fma_indep, 16, out, clk);
    run("v_fmac_f64_dpp x4 independent (per 4)", k_fmac_dpp, 16, out, clk);
    run("2 x (mov_b64_dpp + 2 fmac) (per 6)", k_movdpp_2fmac, 16, out, clk);
    run("2 x (2 readlane + fmac sgpr) (per 6)", k_readlane_fmac, 16, out, clk);
    run("v_mul_f64 dependent", k_mul_dep, 64, out, clk);
    run("v_rsq_f64 dependent", k_rsq_dep, 64, out, clk);
    run("v_rsq_f32 dependent", k_rsq_f32_dep, 64, out, clk);
    run("v_mov_b64_dpp dependent (+s_nop 1)", k_movdpp_dep, 64, out, clk);
    run("v_mfma_f64_16x16x4 dependent", k_mfma_dep, 16, out, clk);
    run("v_rndne_f64 x4 independent (per 4)", k_cvt_rint, 16, out, clk);
    return 0;
}
