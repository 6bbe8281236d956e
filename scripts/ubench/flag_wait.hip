// One launch, producer workgroups first: does a consumer workgroup that spins on a device flag ever starve the producers, what does the
// hand-off cost, and where do the workgroups beyond one-per-CU start?  (development aid for the merged precompute + forward launch)
//   hipcc -O3 --offload-arch=gfx950 flag_wait.hip -o flag_wait && ./flag_wait
// Model: NPRE producer workgroups (lowest block ids) are busy for pre_us, publish 64 KiB each and release a flag; 256 consumers are busy
// for front_us, acquire-spin on every flag (bounded), verify the data, are busy for post_us.  150 KiB of LDS per workgroup: one per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ void busy_us(double us) {
    const unsigned long long t0 = wall_clock64();                       // 100 MHz
    while ((double)(wall_clock64() - t0) < us * 100.0) __builtin_amdgcn_s_sleep(8);
}

extern __shared__ unsigned char smem[];

__global__ __launch_bounds__(512) void k_flag(int mode, int npre, double pre_us, double front_us, double post_us, unsigned* flags, float* data,
                                              unsigned long long* stamps, int* bad) {
    const int tid = threadIdx.x, b = blockIdx.x;
    volatile float* lds = reinterpret_cast<volatile float*>(smem);
    lds[tid] = (float)b;
    if (tid == 0) stamps[b * 4 + 0] = wall_clock64();
    if (b < npre) {
        if (tid == 0) busy_us(pre_us);
        __syncthreads();
        float* d = data + (size_t)b * 16384;
        for (int i = tid; i < 16384; i += 512) d[i] = (float)(i + b);          // plain stores
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            stamps[b * 4 + 1] = wall_clock64();
            if (mode == 0) __hip_atomic_store(&flags[b], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(&flags[32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);   // one counter for all producers
        }
    } else {
        if (tid == 0) busy_us(front_us);
        __syncthreads();
        if (tid == 0) {
            stamps[b * 4 + 1] = wall_clock64();
            if (mode == 0) {
                for (int p = 0; p < npre; ++p) {
                    int spins = 0;
                    while (__hip_atomic_load(&flags[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                        __builtin_amdgcn_s_sleep(16);
                        if (++spins > (1 << 20)) { atomicAdd(bad, 1 << 16); break; }
                    }
                }
            } else {       // relaxed polls of ONE counter (no cache maintenance per poll), one acquire fence at the end
                int spins = 0;
                while (__hip_atomic_load(&flags[32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)npre) {
                    if (mode == 1) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1 << 22)) { atomicAdd(bad, 1 << 16); break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            stamps[b * 4 + 2] = wall_clock64();
        }
        __syncthreads();
        int wrong = 0;
        for (int p = 0; p < npre; ++p) {
            const float* d = data + (size_t)p * 16384;
            for (int i = tid; i < 16384; i += 512) wrong += (d[i] != (float)(i + p));
        }
        if (wrong) atomicAdd(bad, wrong);
        if (tid == 0) busy_us(post_us);
        __syncthreads();
    }
    if (tid == 0) stamps[b * 4 + 3] = wall_clock64();
}

__global__ void k_reset(unsigned* flags, float* data, int npre) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 64) flags[i] = 0;
    if (i < npre * 16384) data[i] = -1.f;
}

int main() {
    const int lds = 150 * 1024;
    (void)hipFuncSetAttribute((const void*)k_flag, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned* flags; float* data; unsigned long long* stamps; int* bad;
    (void)hipMalloc(&flags, 64 * 4); (void)hipMalloc(&data, 16 * 65536); (void)hipMalloc(&stamps, 512 * 4 * 8); (void)hipMalloc(&bad, 4);
    (void)hipMemset(bad, 0, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode : {0, 1, 2}) for (int npre : {2, 8}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int grid = npre + 256;
            hipLaunchKernelGGL(k_reset, dim3((npre * 16384 + 255) / 256), dim3(256), 0, 0, flags, data, npre);
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_flag, dim3(grid), dim3(512), lds, 0, mode, npre, 24.0, 9.0, 22.0, flags, data, stamps, bad);
            (void)hipEventRecord(e1, 0);
            (void)hipDeviceSynchronize();
            float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(grid * 4); int hb = 0;
            (void)hipMemcpy(h.data(), stamps, grid * 32, hipMemcpyDeviceToHost); (void)hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
            unsigned long long t0 = ~0ull, tend = 0, trel = 0;
            for (int b = 0; b < grid; ++b) { t0 = std::min(t0, h[b * 4]); tend = std::max(tend, h[b * 4 + 3]); }
            for (int b = 0; b < npre; ++b) trel = std::max(trel, h[b * 4 + 1]);
            std::vector<double> start, seen;
            for (int b = npre; b < grid; ++b) { start.push_back((h[b * 4] - t0) / 100.0); seen.push_back(((double)h[b * 4 + 2] - (double)trel) / 100.0); }
            std::sort(start.begin(), start.end()); std::sort(seen.begin(), seen.end());
            printf("mode %d npre %d: event %.1f us, in-kernel span %.1f us, producers released at %.1f us; consumer start: median %.1f, 3 latest %.1f %.1f %.1f us; "
                   "flag seen after release: min %.2f median %.2f max %.2f us; bad %d\n", mode, npre, ms * 1e3, (tend - t0) / 100.0, (trel - t0) / 100.0,
                   start[start.size() / 2], start[start.size() - 3], start[start.size() - 2], start.back(), seen[0], seen[seen.size() / 2], seen.back(), hb);
        }
    }
    return 0;
}
