// Micro-benchmark: random 64-bit CAS + 32-bit atomic add throughput as a function of the
// size of the region they land in (is a MALL/L2-resident sub-table worth partitioning for?).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31);
}
struct Slot { uint64_t key; uint32_t count; uint32_t pad; };
template <int MODE>
__global__ __launch_bounds__(256) void k(Slot* t, uint64_t mask, uint64_t n_per_thread, uint64_t salt) {
    uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (uint64_t i = 0; i < n_per_thread; i++) {
        uint64_t h = mix64(id * n_per_thread + i + salt);
        uint64_t s = h & mask;
        if (MODE == 0) atomicAdd(&t[s].count, 1u);
        else if (MODE == 1) { // load + CAS (insert-like)
            uint64_t cur = __hip_atomic_load(&t[s].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == ~0ull) atomicCAS((unsigned long long*)&t[s].key, ~0ull, (unsigned long long)(h >> 2));
        } else if (MODE == 2) { // workgroup-scope atomic add (L2-local?)
            __hip_atomic_fetch_add(&t[s].count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else { // plain load + store (non-atomic RMW)
            uint32_t c = t[s].count; t[s].count = c + 1;
        }
    }
}
int main() {
    const uint64_t max_slots = 1ull << 32;  // 64 GB
    Slot* t; if (hipMalloc(&t, max_slots * sizeof(Slot)) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(t, 0xFF, max_slots * sizeof(Slot));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const uint64_t threads = 2048ull * 256, per = 256;
    for (int mode = 0; mode < 4; mode++) {
        for (int lg = 18; lg <= 32; lg += 2) {
            uint64_t mask = (1ull << lg) - 1;
            hipLaunchKernelGGL(k<0>, dim3(64), dim3(256), 0, 0, t, mask, 4, 1);  // warm
            hipDeviceSynchronize();
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(2048), dim3(256), 0, 0, t, mask, per, 7 + lg);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(2048), dim3(256), 0, 0, t, mask, per, 7 + lg);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(2048), dim3(256), 0, 0, t, mask, per, 7 + lg);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(2048), dim3(256), 0, 0, t, mask, per, 7 + lg);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("mode %d region %8.1f MB : %.2f G ops/s\n", mode, (double)(mask + 1) * 16 / 1e6, threads * per / ms / 1e6);
        }
    }
    return 0;
}
