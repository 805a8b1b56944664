// Micro-benchmark: level 1's append pattern with cursors that are SHARED by the workgroups of one XCD.
//
// scatter1w appends a ~128-byte run per (workgroup, bucket, round) through workgroup-private pages: a bucket's open line
// is completed by the same workgroup a round (~18 us) later, 32 workgroups x 1024 open lines x 128 B = the whole 4 MB L2
// of an XCD - the partial lines are evicted half written (31 GB of write requests for 24 GB of keys).  If
// the 32 workgroups of an XCD append to ONE cursor per bucket, an XCD has 1024 open lines (128 KB) and a line is completed
// within a microsecond: the L2 can merge the runs and evict whole lines.  Price: one returning atomic per (workgroup,
// bucket, round).  Modes:
//   0  workgroup-private streams, plain adds (the pattern of scatter_runs.hip: the baseline)
//   1  one cursor set per XCD (the hardware's XCC_ID), atomics at WORKGROUP scope (they execute in the XCD's L2)
//   2  one cursor set per XCD, atomics at AGENT scope (memory side)
//   3  one cursor set for the device, AGENT scope
// Every round: thread d takes `run` slots of stream d from the cursor (the answer goes to LDS), barrier, the 1024 threads
// write the round's 1024 x run keys (run consecutive lanes = one run).  Cursors start at random offsets (unaligned runs).
// Checked afterwards: every cursor advanced by exactly what was added to it (a lost update = atomics not coherent).
//   hipcc -O3 --offload-arch=gfx950 xcd_append.hip -o xcd_append ; ./xcd_append
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr uint32_t S = 1024;  // streams (level-1 buckets)

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7u;
}

template <int MODE>
__global__ __launch_bounds__(1024) void append_kernel(uint64_t *out, unsigned long long *cur, uint32_t run, uint32_t rounds,
                                                      uint64_t room, unsigned long long *added, int only_atomics) {
    __shared__ uint64_t base[S];
    const uint32_t tid = threadIdx.x;
    const uint32_t set = MODE == 0 ? blockIdx.x : MODE == 3 ? 0u : xcc_id();
    unsigned long long *const mycur = cur + (uint64_t)set * S;
    uint64_t *const myout = out + (uint64_t)set * S * room;
    const uint32_t per_pass = 1024 / run, passes = S / per_pass;
    const uint32_t k = tid % run, sl = tid / run;
    uint64_t mine = 0, acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        {
            unsigned long long old;
            if (MODE == 0) {
                old = mycur[tid];
                mycur[tid] = old + run;
            } else if (MODE == 1) {
                old = __hip_atomic_fetch_add(&mycur[tid], (unsigned long long)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                old = __hip_atomic_fetch_add(&mycur[tid], (unsigned long long)run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            base[tid] = old;
            mine += run;
        }
        __syncthreads();
        if (!only_atomics) {
            for (uint32_t p = 0; p < passes; p++) {
                const uint32_t s = p * per_pass + sl;
                const uint64_t pos = (uint64_t)s * room + base[s] + k;
                myout[pos] = pos + r;
                acc += pos;
            }
        } else {
            acc += base[(tid * 7u) & (S - 1)];
        }
        __syncthreads();
    }
    if (acc == 0x1234567) out[0] = acc;
    // what this workgroup added to its set (one figure per set: the check below)
    atomicAdd(&added[set], (unsigned long long)mine);
}

int main(int argc, char **argv) {
    const uint32_t WG = argc > 1 ? atoi(argv[1]) : 256;
    const uint64_t total_keys = 1ull << 30;  // 8 GB written per measurement
    const uint64_t room = total_keys / S + (1 << 16);  // (mode 3: one set takes everything)
    uint64_t *out;
    unsigned long long *cur, *added;
    CHECK(hipMalloc(&out, ((uint64_t)S * room + (uint64_t)WG * S * (total_keys / WG / S + 4096)) * 8));
    CHECK(hipMalloc(&cur, (uint64_t)WG * S * 8));
    CHECK(hipMalloc(&added, WG * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<unsigned long long> init((size_t)WG * S), fin((size_t)WG * S), add(WG);
    for (size_t i = 0; i < init.size(); i++) init[i] = (i * 2654435761ull >> 7) % 61;  // unaligned starts
    printf("WG=%u streams=%u, 8 GB of 8-byte keys per measurement\n", WG, S);
    printf("%5s %8s %6s | %9s %9s %s\n", "mode", "run[B]", "atomic", "ms", "GB/s", "check");
    for (int only = 0; only < 2; only++)
        for (uint32_t run : {8u, 16u, 32u})
            for (int mode = 0; mode < 4; mode++) {
                const uint32_t rounds = (uint32_t)(total_keys / WG / S / run);
                // room per (set, stream): mode 0 = per workgroup, 1/2 = per XCD (if XCC_ID spreads unevenly: room for all), 3 = all
                const uint64_t rm = mode == 0 ? total_keys / WG / S + 4096 : room;
                float best = 1e9;
                bool ok = true;
                for (int rep = 0; rep < 3; rep++) {
                    CHECK(hipMemcpy(cur, init.data(), init.size() * 8, hipMemcpyHostToDevice));
                    CHECK(hipMemset(added, 0, WG * 8));
                    CHECK(hipEventRecord(a));
                    const uint64_t rmm = mode == 1 || mode == 2 ? room / 8 * 2 : rm;  // (XCD sets: a quarter of everything each)
                    if (mode == 0) hipLaunchKernelGGL(append_kernel<0>, dim3(WG), dim3(1024), 0, 0, out, cur, run, rounds, rmm, added, only);
                    if (mode == 1) hipLaunchKernelGGL(append_kernel<1>, dim3(WG), dim3(1024), 0, 0, out, cur, run, rounds, rmm, added, only);
                    if (mode == 2) hipLaunchKernelGGL(append_kernel<2>, dim3(WG), dim3(1024), 0, 0, out, cur, run, rounds, rmm, added, only);
                    if (mode == 3) hipLaunchKernelGGL(append_kernel<3>, dim3(WG), dim3(1024), 0, 0, out, cur, run, rounds, rmm, added, only);
                    CHECK(hipEventRecord(b));
                    CHECK(hipEventSynchronize(b));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                    CHECK(hipMemcpy(fin.data(), cur, fin.size() * 8, hipMemcpyDeviceToHost));
                    CHECK(hipMemcpy(add.data(), added, WG * 8, hipMemcpyDeviceToHost));
                    const uint32_t nset = mode == 0 ? WG : mode == 3 ? 1 : 8;
                    for (uint32_t s = 0; s < nset; s++) {
                        // every thread of every workgroup of the set added rounds * run to ITS stream: added[set] = sum over the
                        // set's workgroups of 1024 * rounds * run; each stream's cursor moved by added[set] / 1024
                        const unsigned long long per_stream = add[s] / 1024;
                        for (uint32_t d = 0; d < S; d++)
                            if (fin[(size_t)s * S + d] - init[(size_t)s * S + d] != per_stream) ok = false;
                    }
                }
                const double bytes = (double)rounds * WG * S * run * 8;
                printf("%5d %8u %6s | %9.3f %9.1f %s\n", mode, run * 8, only ? "only" : "+data", best, only ? 0.0 : bytes / best / 1e6,
                       ok ? "cursors exact" : "LOST UPDATES");
            }
    return 0;
}
