// Micro-benchmark: random 16-byte loads (one table slot) as a function of the region they land
// in and of the loads kept in flight per thread - the ceiling for cov_kernel's table probes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
template <int ILP>
__global__ __launch_bounds__(256) void k(const uint4* t, uint64_t mask, uint64_t n_per_thread, uint64_t salt, uint32_t* sink) {
    uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (uint64_t i = 0; i < n_per_thread; i += ILP) {
        uint4 v[ILP];
#pragma unroll
        for (int u = 0; u < ILP; u++) v[u] = t[mix64(id * n_per_thread + i + u + salt) & mask];
#pragma unroll
        for (int u = 0; u < ILP; u++) acc += v[u].x ^ v[u].z;
    }
    if (acc == 0x12345) sink[0] = acc;
}
int main() {
    const uint64_t max_slots = 1ull << 31;  // 32 GB
    uint4* t; if (hipMalloc(&t, max_slots * 16) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 0x5A, max_slots * 16);
    uint32_t* sink; hipMalloc(&sink, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint64_t per = 256;
    for (int ilp : {1, 4, 16}) {
        for (int blocks : {2048, 8192}) {
            for (int lg = 18; lg <= 31; lg += (lg < 28 ? 4 : 1)) {
                uint64_t mask = (1ull << lg) - 1;
                const uint64_t threads = (uint64_t)blocks * 256;
                hipDeviceSynchronize();
                hipEventRecord(a);
                if (ilp == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, t, mask, per, 7 + lg, sink);
                if (ilp == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, t, mask, per, 7 + lg, sink);
                if (ilp == 16) hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), 0, 0, t, mask, per, 7 + lg, sink);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                printf("ilp %2d blocks %5d region %9.1f MB : %.2f G loads/s\n", ilp, blocks, (double)(mask + 1) * 16 / 1e6, threads * per / ms / 1e6);
            }
        }
    }
    return 0;
}
