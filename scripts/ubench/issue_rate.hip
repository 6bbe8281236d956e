// VALU / LDS / scalar issue-rate calibration for one wave on gfx950 (development aid).
//   hipcc -O3 --offload-arch=gfx950 issue_rate.hip -o issue_rate && ./issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

#define BENCH(name, body)                                                                              \
    __global__ void name(float* out, long long* clk, int n) {                                         \
        __shared__ float lds[4096];                                                                    \
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f + i;                        \
        __syncthreads();                                                                               \
        float a = threadIdx.x * 1.0f, b = 1.0001f, c = 0.5f, d = 0.25f; int ia = threadIdx.x, ib = 3;  \
        unsigned la = (threadIdx.x & 63) * 4;                                                          \
        const long long t0 = clock64();                                                                \
        for (int i = 0; i < n; ++i) { body }                                                           \
        const long long t1 = clock64();                                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + ia + ib + la;                     \
        if ((threadIdx.x & 63) == 0) clk[threadIdx.x / 64] = t1 - t0;                                  \
    }

BENCH(k_fma_dep, REP64(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));))
BENCH(k_fma_indep, REP16(asm volatile("v_fma_f32 %0, %4, %5, %0\n v_fma_f32 %1, %4, %5, %1\n v_fma_f32 %2, %4, %5, %2\n v_fma_f32 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(1.0f), "v"(0.5f));))
BENCH(k_mul_lo, REP64(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia) : "v"(ib));))
BENCH(k_cmp_cnd, REP64(asm volatile("v_cmp_gt_i32_e64 s[20:21], %1, %0\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(ia) : "v"(ib) : "s20", "s21");))
BENCH(k_add_dep, REP64(asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(ia) : "v"(ib));))
BENCH(k_salu, REP64(asm volatile("s_add_i32 s20, s20, 1" ::: "s20", "scc");))
BENCH(k_branchy, REP64(asm volatile("s_cmp_eq_u32 s20, 77\n s_cbranch_scc1 1\n s_nop 0" ::: "s20", "scc");))
BENCH(k_lds_dep, REP64(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xffc, %0" : "+v"(la));))
BENCH(k_lds_8, REP16(asm volatile("ds_read_b32 v100, %0\n ds_read_b32 v101, %0 offset:256\n ds_read_b32 v102, %0 offset:512\n ds_read_b32 v103, %0 offset:768\n ds_read_b32 v104, %0 offset:1024\n ds_read_b32 v105, %0 offset:1280\n ds_read_b32 v106, %0 offset:1536\n ds_read_b32 v107, %0 offset:1792\n s_waitcnt lgkmcnt(0)" :: "v"(la) : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");))
BENCH(k_readlane, REP64(asm volatile("v_readfirstlane_b32 s20, %0\n v_add_u32_e32 %0, s20, %0" : "+v"(ia) :: "s20");))

template <class K>
static void run(const char* name, K kern, int per_iter, float* out, long long* clk) {
    for (int threads : {64, 512}) {
        std::vector<long long> h(8);
        const int n = 64;
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, clk, n); (void)hipDeviceSynchronize();
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, clk, n); (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), clk, 64, hipMemcpyDeviceToHost);
        printf("%-34s %3d threads: %6.1f clk per instruction group\n", name, threads, (double)h[0] / (n * per_iter));
    }
}

int main() {
    float* out; long long* clk;
    (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&clk, 4096);
    run("v_fma_f32 dependent", k_fma_dep, 64, out, clk);
    run("v_fma_f32 x4 independent (per 4)", k_fma_indep, 16, out, clk);
    run("v_mul_lo_u32 dependent", k_mul_lo, 64, out, clk);
    run("v_cmp_e64 + v_cndmask_e64 (pair)", k_cmp_cnd, 64, out, clk);
    run("v_add_u32 dependent", k_add_dep, 64, out, clk);
    run("s_add_i32 dependent", k_salu, 64, out, clk);
    run("s_cmp + s_cbranch (not taken) + nop", k_branchy, 64, out, clk);
    run("ds_read_b32 dependent (+and)", k_lds_dep, 64, out, clk);
    run("8 x ds_read_b32 imm offsets + wait", k_lds_8, 16, out, clk);
    run("v_readfirstlane + v_add (s operand)", k_readlane, 64, out, clk);
    return 0;
}
