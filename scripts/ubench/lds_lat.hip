// LDS latency / issue-rate calibration on gfx950 (development aid).
//   hipcc -O3 --offload-arch=gfx950 lds_lat.hip -o lds_lat && ./lds_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_chain(int* out, long long* clk, int iters) {
    __shared__ int buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = (i * 17 + 5) & 4095;
    __syncthreads();
    int idx = threadIdx.x & 63;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) idx = buf[idx];                  // dependent chain: pure latency
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int N>
__global__ void k_indep(int* out, long long* clk, int iters, int stride) {
    __shared__ float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = (float)i;
    __syncthreads();
    float acc = 0.f;
    const int base = (threadIdx.x & 63) * stride;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        float v[N];
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = buf[(base + e * 67 + i) & 8191];     // N independent reads in flight
#pragma unroll
        for (int e = 0; e < N; ++e) acc += v[e];
    }
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (int)acc;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void k_uniform_read(int* out, long long* clk, int iters) {
    __shared__ int buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = (i * 17 + 5) & 4095;
    __syncthreads();
    int idx = 3;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) idx = __builtin_amdgcn_readfirstlane(buf[idx]);   // ds_read + readfirstlane chain
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = idx;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

__global__ void k_barrier(int* out, long long* clk, int iters) {
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) __syncthreads();
    const long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = 0;
    if ((threadIdx.x & 63) == 0) clk[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
    int* out; long long* clk;
    hipMalloc(&out, 1 << 20); hipMalloc(&clk, 4096);
    std::vector<long long> h(16);
    const int iters = 256;
    for (int threads : {64, 128, 512}) {
        hipLaunchKernelGGL(k_chain, dim3(1), dim3(threads), 0, 0, out, clk, iters); hipDeviceSynchronize();
        hipMemcpy(h.data(), clk, 64, hipMemcpyDeviceToHost);
        printf("dependent ds_read_b32 chain, %3d threads: %.1f clk / read\n", threads, (double)h[0] / iters);
        hipLaunchKernelGGL(k_uniform_read, dim3(1), dim3(threads), 0, 0, out, clk, iters); hipDeviceSynchronize();
        hipMemcpy(h.data(), clk, 64, hipMemcpyDeviceToHost);
        printf("ds_read + readfirstlane chain, %3d threads: %.1f clk / step\n", threads, (double)h[0] / iters);
        for (int stride : {1, 13}) {
            hipLaunchKernelGGL(k_indep<1>, dim3(1), dim3(threads), 0, 0, out, clk, iters, stride); hipDeviceSynchronize();
            hipMemcpy(h.data(), clk, 64, hipMemcpyDeviceToHost);
            const double c1 = (double)h[0] / iters;
            hipLaunchKernelGGL(k_indep<8>, dim3(1), dim3(threads), 0, 0, out, clk, iters, stride); hipDeviceSynchronize();
            hipMemcpy(h.data(), clk, 64, hipMemcpyDeviceToHost);
            const double c8 = (double)h[0] / iters;
            hipLaunchKernelGGL(k_indep<16>, dim3(1), dim3(threads), 0, 0, out, clk, iters, stride); hipDeviceSynchronize();
            hipMemcpy(h.data(), clk, 64, hipMemcpyDeviceToHost);
            const double c16 = (double)h[0] / iters;
            printf("batched ds_read_b32 (lane stride %2d), %3d threads: 1 read %.0f clk, 8 reads %.0f clk, 16 reads %.0f clk per batch\n",
                   stride, threads, c1, c8, c16);
        }
        hipLaunchKernelGGL(k_barrier, dim3(1), dim3(threads), 0, 0, out, clk, iters); hipDeviceSynchronize();
        hipMemcpy(h.data(), clk, 64, hipMemcpyDeviceToHost);
        printf("__syncthreads, %3d threads: %.1f clk\n", threads, (double)h[0] / iters);
    }
    return 0;
}
