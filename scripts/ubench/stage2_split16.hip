// Microbenchmark: the stage-2 inner loop with split-f16 operands (x = h1 + h2, three v_mfma_f32_16x16x32_f16 per 16x32 slab and
// sub-tile) against the fp32 loop of stage2_loop.hip (eight v_mfma_f32_16x16x4_f32 per the same slab).  Development aid.
//   hipcc -O3 --offload-arch=gfx950 stage2_split16.hip -o stage2_split16 && ./stage2_split16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
typedef const __attribute__((address_space(1))) f32x4* gptr4;
constexpr int NS = 5, NSAMP = 80, NKC = 4;          // M = 128: four 32-deep k-chunks

// A stream: per slab 2 planes x 64 lanes x 16 B (plane-major: [slab][plane][lane]); B tile in LDS: [plane][(kc*4 + g) * NSAMP + sample] 16-B vectors
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k16(const f32x4* __restrict__ A, float* out, int nslabs, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* at = reinterpret_cast<f32x4*>(smem);                       // 2 planes x NKC*4*NSAMP vectors
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4, jq = lane & 15;
    for (int i = tid; i < 2 * NKC * 4 * NSAMP; i += THREADS) at[i] = f32x4{0.001f * i, 0.5f, -0.25f, 1.f + 1e-3f * i};
    __syncthreads();
    gptr4 P = (gptr4)A + (size_t)wave * nslabs * 128 + lane;
    f32x4 acc[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) acc[t] = f32x4{0, 0, 0, 0};
    f32x4 r1[4], r2[4];
#pragma unroll
    for (int u = 0; u < 3; ++u) { r1[u] = P[(size_t)u * 128]; r2[u] = P[(size_t)u * 128 + 64]; }
    const f32x4* B1 = at + gq * NSAMP + jq;
    const f32x4* B2 = B1 + NKC * 4 * NSAMP;
    int c = 0;
    const unsigned long long t0 = clock64();
    for (int q0 = 0; q0 < nslabs; q0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u;
            const size_t nx = (size_t)(q + 3 < nslabs ? q + 3 : nslabs - 1) * 128;
            r1[(u + 3) & 3] = P[nx]; r2[(u + 3) & 3] = P[nx + 64];
            const f16x8 a1 = __builtin_bit_cast(f16x8, r1[u]), a2 = __builtin_bit_cast(f16x8, r2[u]);
            f32x4 b1[NS], b2[NS];
#pragma unroll
            for (int t = 0; t < NS; ++t) { b1[t] = B1[c * 4 * NSAMP + 16 * t]; b2[t] = B2[c * 4 * NSAMP + 16 * t]; }
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, __builtin_bit_cast(f16x8, b2[t]), acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, __builtin_bit_cast(f16x8, b1[t]), acc[t], 0, 0, 0);
            if (++c == NKC) c = 0;
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0;
    for (int t = 0; t < NS; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[(size_t)blockIdx.x * THREADS + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    const int nslabs = 100 / 8 * 8 + 4;             // ~ R * 20 slabs of a layer split over 8 waves -> 13 per wave; use 100 per wave for timing resolution
    const int per_wave = 100;
    f32x4* A; hipMalloc(&A, (size_t)8 * per_wave * 128 * 16);
    std::vector<_Float16> h((size_t)8 * per_wave * 128 * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (_Float16)(0.01f * (float)((i * 37) % 101) - 0.5f);
    hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const size_t lds = 2 * NKC * 4 * NSAMP * 16;
    hipFuncSetAttribute((const void*)k16<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int grid : {1, 256}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k16<512><<<grid, 512, lds>>>(A, out, per_wave, cyc);
        hipEventRecord(e0); k16<512><<<grid, 512, lds>>>(A, out, per_wave, cyc); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> c(256 * 8); hipMemcpy(c.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
        double mx = 0;
        for (int b = 0; b < grid; ++b) { double m = 0; for (int w = 0; w < 8; ++w) m = std::max(m, (double)c[b * 8 + w]); mx += m; }
        mx /= grid;
        // per SIMD: two waves share it: 2 * per_wave slabs
        printf("split-f16, %3d workgroups: slowest wave %7.0f clk = %5.0f clk per slab per SIMD (MFMA floor 3*NS*16 = 240; fp32: 1280); kernel %.1f us; "
               "A stream %.0f KB per workgroup -> %.2f TB/s from L2\n", grid, mx, mx / (2.0 * per_wave), ms * 1e3,
               8.0 * per_wave * 2, grid * 8.0 * per_wave * 2048 / (ms * 1e-3) / 1e12);
    }
    (void)nslabs;
    return 0;
}
