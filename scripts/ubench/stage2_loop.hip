// Microbenchmark of the stage-2 inner loop of k_dgp_forward (development aid): a wave streams packed 1-KiB A
// blocks from L2 and multiplies each with NS B tiles read from LDS (4*NS MFMAs 16x16x4 f32 per block).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef const __attribute__((address_space(1))) f32x4* gptr4;
constexpr int NS = 5, NSAMP = 80, NBK = 8;

template <int VAR, int THREADS>
__global__ __launch_bounds__(THREADS) void k(const f32x4* __restrict__ A, float* out, int nblocks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* at = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4, jq = lane & 15;
    for (int i = tid; i < NBK * 4 * NSAMP; i += THREADS) at[i] = f32x4{0.001f * i, 0.5f, -0.25f, 1.f + 1e-3f * i};
    __syncthreads();
    gptr4 P = (gptr4)A + (size_t)wave * nblocks * 64 + lane;      // every workgroup streams the same blocks (L2-resident)
    f32x4 acc[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) acc[t] = f32x4{0, 0, 0, 0};
    f32x4 ring[4];
    ring[0] = P[0]; ring[1] = P[64]; ring[2] = P[128];
    const f32x4* Bp = at + gq * NSAMP + jq;
    f32x4 bA[NS], bB[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) bA[t] = Bp[16 * t];
    int c = NBK;
    const unsigned long long t0 = clock64();
    for (int q0 = 0; q0 < nblocks; q0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u;
            if (VAR != 4) ring[(u + 3) & 3] = P[(size_t)(q + 3 < nblocks ? q + 3 : nblocks - 1) * 64];
            const f32x4 a_cur = ring[VAR == 4 ? 0 : u];
            const f32x4* Bn = (c > 1) ? Bp + 4 * NSAMP : at + gq * NSAMP + jq;
            if ((u & 1) == 0) {
                if (VAR != 3) {
#pragma unroll
                    for (int t = 0; t < NS; ++t) bB[t] = Bn[16 * t];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], bA[t][s], acc[t], 0, 0, 0);
                        if (VAR == 1 && s == 0) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    }
            } else {
                if (VAR != 3) {
#pragma unroll
                    for (int t = 0; t < NS; ++t) bA[t] = Bn[16 * t];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[s], bB[t][s], acc[t], 0, 0, 0);
                        if (VAR == 1 && s == 0) { __builtin_amdgcn_sched_group_barrier(0x8, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                    }
            }
            Bp = Bn;
            if (--c == 0) c = NBK;
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0;
    for (int t = 0; t < NS; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[(size_t)blockIdx.x * THREADS + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int VAR, int THREADS>
void run(const char* name, const f32x4* A, int nblocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * THREADS * 4); hipMalloc(&cyc, 256 * 8 * 8); hipMemset(cyc, 0, 256 * 8 * 8);
    const size_t lds = NBK * 4 * NSAMP * 16;
    hipFuncSetAttribute((const void*)k<VAR, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<VAR, THREADS><<<256, THREADS, lds>>>(A, out, nblocks, cyc);
    hipEventRecord(e0); k<VAR, THREADS><<<256, THREADS, lds>>>(A, out, nblocks, cyc); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    const int nw = THREADS / 64;
    double mx = 0, first = 0, last = 0;
    for (int b = 0; b < 256; ++b) { double m = 0; for (int w = 0; w < nw; ++w) m = std::max(m, (double)h[b * 8 + w]); mx += m; first += h[b * 8]; last += h[b * 8 + nw - 1]; }
    mx /= 256; first /= 256; last /= 256;
    const double blocks_per_simd = (double)nblocks * nw / 4;
    printf("%-34s %3d thr: slowest wave %7.0f clk = %5.0f clk per block per SIMD (ideal 640); wave0 %6.0f, last wave %6.0f; kernel %.1f us\n",
           name, THREADS, mx, mx / blocks_per_simd, first, last, ms * 1e3);
    hipFree(out); hipFree(cyc);
}

// paired variant: one wave streams TWO A blocks per step (two row-blocks with the same k-range) against the same B tiles
template <int THREADS>
__global__ __launch_bounds__(THREADS) void kpair(const f32x4* __restrict__ A, float* out, int nblocks, unsigned long long* cyc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    f32x4* at = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gq = lane >> 4, jq = lane & 15;
    for (int i = tid; i < NBK * 4 * NSAMP; i += THREADS) at[i] = f32x4{0.001f * i, 0.5f, -0.25f, 1.f + 1e-3f * i};
    __syncthreads();
    gptr4 P = (gptr4)A + (size_t)wave * 2 * nblocks * 64 + lane;
    gptr4 Q = P + (size_t)nblocks * 64;
    f32x4 acc[NS], acd[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) { acc[t] = f32x4{0, 0, 0, 0}; acd[t] = f32x4{0, 0, 0, 0}; }
    f32x4 rp[4], rq[4];
    rp[0] = P[0]; rp[1] = P[64]; rp[2] = P[128]; rq[0] = Q[0]; rq[1] = Q[64]; rq[2] = Q[128];
    const f32x4* Bp = at + gq * NSAMP + jq;
    f32x4 bA[NS], bB[NS];
#pragma unroll
    for (int t = 0; t < NS; ++t) bA[t] = Bp[16 * t];
    int c = NBK;
    const unsigned long long t0 = clock64();
    for (int q0 = 0; q0 < nblocks; q0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = q0 + u;
            const size_t nx = (size_t)(q + 3 < nblocks ? q + 3 : nblocks - 1) * 64;
            rp[(u + 3) & 3] = P[nx]; rq[(u + 3) & 3] = Q[nx];
            const f32x4 a1 = rp[u], a2 = rq[u];
            const f32x4* Bn = (c > 1) ? Bp + 4 * NSAMP : at + gq * NSAMP + jq;
            if ((u & 1) == 0) {
#pragma unroll
                for (int t = 0; t < NS; ++t) bB[t] = Bn[16 * t];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bA[t][s], acc[t], 0, 0, 0);
                        acd[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], bA[t][s], acd[t], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int t = 0; t < NS; ++t) bA[t] = Bn[16 * t];
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int t = 0; t < NS; ++t) {
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[s], bB[t][s], acc[t], 0, 0, 0);
                        acd[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[s], bB[t][s], acd[t], 0, 0, 0);
                    }
            }
            Bp = Bn;
            if (--c == 0) c = NBK;
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0;
    for (int t = 0; t < NS; ++t) s += acc[t][0] + acc[t][1] + acd[t][2] + acd[t][3];
    out[(size_t)blockIdx.x * THREADS + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
template <int THREADS>
void runpair(const char* name, const f32x4* A, int nblocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * THREADS * 4); hipMalloc(&cyc, 256 * 8 * 8); hipMemset(cyc, 0, 256 * 8 * 8);
    const size_t lds = NBK * 4 * NSAMP * 16;
    hipFuncSetAttribute((const void*)kpair<THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    kpair<THREADS><<<256, THREADS, lds>>>(A, out, nblocks, cyc);
    kpair<THREADS><<<256, THREADS, lds>>>(A, out, nblocks, cyc); hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8); hipMemcpy(h.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
    const int nw = THREADS / 64;
    double mx = 0;
    for (int b = 0; b < 256; ++b) { double m = 0; for (int w = 0; w < nw; ++w) m = std::max(m, (double)h[b * 8 + w]); mx += m; }
    mx /= 256;
    const double blocks_per_simd = (double)nblocks * 2 * nw / 4;
    printf("%-34s %3d thr: slowest wave %7.0f clk = %5.0f clk per block per SIMD (ideal 640)\n", name, THREADS, mx, mx / blocks_per_simd);
    hipFree(out); hipFree(cyc);
}
int main() {
    const int nblocks = 24;
    f32x4* A; hipMalloc(&A, (size_t)8 * 64 * 64 * 16 * 2);
    std::vector<float> h((size_t)8 * 64 * 64 * 4 * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 1e-3f - 0.5f;
    hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 512>("ring + B double buffer", A, nblocks);
    run<1, 512>("  + sched_group_barrier", A, nblocks);
    run<3, 512>("  no LDS reads", A, nblocks);
    run<4, 512>("  no global loads", A, nblocks);
    run<0, 256>("ring + B double buffer", A, 2 * nblocks);
    run<1, 256>("  + sched_group_barrier", A, 2 * nblocks);
    run<3, 256>("  no LDS reads", A, 2 * nblocks);
    run<4, 256>("  no global loads", A, 2 * nblocks);
    runpair<512>("paired (2 A streams, shared B)", A, nblocks / 2);
    runpair<256>("paired (2 A streams, shared B)", A, nblocks);
    return 0;
}
