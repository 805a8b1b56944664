// Micro-benchmark: can a compute unit hide a store burst of one workgroup behind the arithmetic of another?
// Every workgroup alternates a compute phase (a dependent VALU chain of `work` trips per thread, no memory) and a store
// phase (its share of a round of level 1's appends: runs of 8-byte keys to 1024 buckets through per-XCD cursors, as
// tools/ubench/xcd_append.hip mode 2).  Measured for one 1024-thread workgroup per CU and for two of 512, with the
// compute phase sized to 0, 1x, 2x the store phase: if the two workgroups of a CU overlap, (2 x 512) runs at
// max(compute, store); if not, at their sum - like one workgroup.
//   hipcc -O3 --offload-arch=gfx950 store_overlap.hip -o store_overlap ; ./store_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr uint32_t S = 1024;
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}
template <int T>
__global__ __launch_bounds__(T) void k(uint64_t *out, unsigned long long *cur, uint32_t rounds, uint64_t room, uint32_t work,
                                       int do_store, int atomics) {
    __shared__ uint64_t base[S];
    const uint32_t tid = threadIdx.x, set = xcc_id();
    constexpr uint32_t KEYS = T * 16, RUN = KEYS / S;  // keys per round, per bucket
    unsigned long long *const mycur = cur + (uint64_t)set * S;
    uint64_t *const myout = out + (uint64_t)set * S * room;
    float x = (float)tid;
    uint64_t acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t w = 0; w < work; w++) x = x * 1.0001f + 0.5f;  // compute phase: a dependent chain
        for (uint32_t d = tid; d < S; d += T) {
            unsigned long long old = atomics ? __hip_atomic_fetch_add(&mycur[d], (unsigned long long)RUN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                             : (unsigned long long)r * RUN * 64;
            base[d] = old;
        }
        __syncthreads();
        if (do_store) {
#pragma unroll
            for (uint32_t u = 0; u < 16; u++) {
                const uint32_t i = tid + u * T, s = i / RUN, kk = i % RUN;
                const uint64_t pos = (uint64_t)s * room + base[s] + kk;
                myout[pos] = pos + r;
            }
        } else {
            acc += base[(tid * 7u) & (S - 1)];
        }
        __syncthreads();
    }
    if (x == 1.2345f || acc == 0x1234567) out[0] = (uint64_t)x + acc;
}
int main() {
    const uint32_t CU = 256;
    const uint64_t total_keys = 1ull << 29;  // 4 GB per measurement
    const uint64_t room = total_keys / S / 2 + (1 << 16);
    uint64_t *out;
    unsigned long long *cur;
    CHECK(hipMalloc(&out, 8ull * S * room * 8));
    CHECK(hipMalloc(&cur, 8 * S * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    printf("%6s %6s %7s %7s | %9s\n", "T", "work", "store", "atomic", "ms");
    for (int T : {1024, 512})
        for (uint32_t work : {0u, 1500u, 3000u, 6000u})
            for (int st = 0; st < 2; st++)
                for (int at = 1; at < 2; at++) {
                    const uint32_t wgs = CU * (1024 / T);
                    const uint32_t rounds = (uint32_t)(total_keys / wgs / (T * 16));
                    float best = 1e9;
                    for (int rep = 0; rep < 3; rep++) {
                        CHECK(hipMemset(cur, 0, 8 * S * 8));
                        CHECK(hipEventRecord(a));
                        if (T == 1024) hipLaunchKernelGGL(k<1024>, dim3(wgs), dim3(T), 0, 0, out, cur, rounds, room, work, st, at);
                        else hipLaunchKernelGGL(k<512>, dim3(wgs), dim3(T), 0, 0, out, cur, rounds, room, work, st, at);
                        CHECK(hipEventRecord(b));
                        CHECK(hipEventSynchronize(b));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, a, b));
                        if (ms < best) best = ms;
                    }
                    printf("%6d %6u %7d %7d | %9.3f\n", T, work, st, at, best);
                }
    return 0;
}
