// Micro-benchmark: returning 64-bit atomicAdd on ONE address (a cursor) from one lane per workgroup - the rate at which
// a kernel's workgroups can reserve output space through a single global counter, and the latency one of them sees.
// Also: S cursors (workgroup b uses cursor b % S), to see how far sharding the cursor helps.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(unsigned long long *cur, uint32_t n_cur, uint32_t per, uint32_t spin, unsigned long long *sink,
                                         unsigned long long *lat) {
    unsigned long long acc = 0, t_lat = 0;
    if (threadIdx.x == 0) {
        unsigned long long *c = cur + (size_t)(blockIdx.x % n_cur) * 32;  // cursors 256 bytes apart
        for (uint32_t i = 0; i < per; i++) {
            const unsigned long long t0 = __builtin_readcyclecounter();
            acc += atomicAdd(c, 3000ull);
            __builtin_amdgcn_s_waitcnt(0);
            t_lat += __builtin_readcyclecounter() - t0;
            for (uint32_t s = 0; s < spin; s++) __builtin_amdgcn_s_sleep(8);  // ~ the work between two reservations
        }
        sink[blockIdx.x] = acc;
        atomicAdd(lat, t_lat);
    }
}
int main() {
    unsigned long long *cur, *sink, *lat;
    hipMalloc(&cur, 1 << 20); hipMalloc(&sink, 65536 * 8); hipMalloc(&lat, 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint32_t grids[] = {256, 512, 2048, 16384};
    const uint32_t curs[] = {1, 8, 64, 512};
    const uint32_t spins[] = {0, 16, 128};
    for (uint32_t g : grids) for (uint32_t nc : curs) for (uint32_t sp : spins) {
        const uint32_t per = (1u << 20) / g * 4;  // 4 M reservations in all
        hipMemset(cur, 0, 1 << 20); hipMemset(lat, 0, 8);
        hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, cur, nc, 4u, 0u, sink, lat);
        hipDeviceSynchronize(); hipMemset(lat, 0, 8);
        hipEventRecord(a);
        hipLaunchKernelGGL(k, dim3(g), dim3(256), 0, 0, cur, nc, per, sp, sink, lat);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long l; hipMemcpy(&l, lat, 8, hipMemcpyDeviceToHost);
        const double n = (double)per * g;
        printf("workgroups %5u cursors %3u spin %3u: %7.1f M reservations/s  (%.2f ms for %.1f M), mean latency %.0f cycles\n", g, nc, sp,
               n / ms / 1e3, ms, n / 1e6, (double)l / n);
    }
    return 0;
}
