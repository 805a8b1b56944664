// Micro-benchmark: LDS operation rates per CU for the range build's dedupe (kt_bulk.hip): random-address
// 64-bit CAS, 32-bit CAS, 32-bit add (returning / not), plain 64-bit write / read, from 2 x 512 threads per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
constexpr int T = 512, SLOTS = 5120, ITER = 4096;
template <int MODE>
__global__ __launch_bounds__(T) void k(uint64_t *out, uint32_t seed) {
    __shared__ unsigned long long a64[SLOTS];
    __shared__ unsigned int a32[SLOTS * 2];
    for (int i = threadIdx.x; i < SLOTS; i += T) { a64[i] = ~0ull; a32[i] = 0; a32[i + SLOTS] = 0; }
    __syncthreads();
    uint32_t x = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    uint64_t acc = 0;
    for (int i = 0; i < ITER; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t s = (x >> 8) % SLOTS;
        if (MODE == 0) acc += atomicCAS(&a64[s], ~0ull, (unsigned long long)x);
        if (MODE == 1) acc += atomicCAS(&a32[s], 0u, x);
        if (MODE == 2) acc += atomicAdd(&a32[s], 1u);
        if (MODE == 3) atomicAdd(&a32[s], 1u);
        if (MODE == 4) a64[s] = x;
        if (MODE == 5) acc += a64[s];
        if (MODE == 6) { a32[s] = x; }
        if (MODE == 7) acc += a32[s];
        if (MODE == 8) acc += atomicMax(&a32[s], x);
        if (MODE == 9) acc += atomicExch(&a32[s], x);
    }
    if (acc == 0x1234567) out[0] = acc;
}
int main() {
    uint64_t *d; hipMalloc(&d, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char *names[] = {"cas64 rtn", "cas32 rtn", "add32 rtn", "add32 noret", "write64", "read64", "write32", "read32", "max32 rtn", "xchg32 rtn"};
    for (int mode = 0; mode < 10; mode++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(a);
#define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(512), dim3(T), 0, 0, d, 7u + rep)
            L(0); L(1); L(2); L(3); L(4); L(5); L(6); L(7); L(8); L(9);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("%-12s %7.3f ms  %6.2f lane-ops/clk/CU (2.4 GHz, 256 CUs)\n", names[mode], ms,
                            512.0 * T * ITER / (ms * 1e-3) / 256 / 2.4e9);
        }
    }
    return 0;
}
