// Micro-benchmark: what the memory system makes of a radix partition's write pattern - every workgroup appends short
// runs of 8-byte keys to S output streams (fixed regions), round after round - as a function of the run length and of
// whether the runs are aligned to cache lines.  Optional streaming read beside it (the partition's input).
//   hipcc -O3 --offload-arch=gfx950 scatter_runs.hip -o scatter_runs ; ./scatter_runs
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// one workgroup = 1024 threads; stream s of workgroup w owns region (w * S + s) * room bytes.  A round appends `run`
// bytes to every stream: run / 8 consecutive lanes write one run; lanes of a wave cover 512 / run streams.
template <bool READ>
__global__ __launch_bounds__(1024) void scatter_kernel(uint64_t *out, const uint64_t *in, uint32_t S, uint32_t run_keys,
                                                       uint32_t rounds, uint64_t room_keys, uint32_t skew_keys, uint32_t jitter) {
    const uint32_t tid = threadIdx.x;
    const uint32_t per_pass = 1024 / run_keys;          // streams covered by one pass of the workgroup
    const uint32_t passes = S / per_pass;
    const uint32_t k = tid % run_keys, sl = tid / run_keys;
    uint64_t acc = 0;
    const uint64_t wbase = (uint64_t)blockIdx.x * S * room_keys;
    const uint64_t *rin = in + (uint64_t)blockIdx.x * rounds * passes * 1024;
    for (uint32_t r = 0; r < rounds; r++) {
        for (uint32_t p = 0; p < passes; p++) {
            const uint32_t s = p * per_pass + sl;
            // variable run lengths: the cursor of stream s after r rounds = r * run + a stream-dependent wobble
            const uint32_t wob = jitter ? ((s * 2654435761u + r * 40503u) >> 28) % jitter : 0u;  // (just shifts the run)
            const uint64_t pos = wbase + (uint64_t)s * room_keys + skew_keys + (uint64_t)r * run_keys + wob + k;
            uint64_t v = pos;
            if (READ) v += rin[((uint64_t)r * passes + p) * 1024 + tid];
            out[pos] = v;
            acc += v;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}

int main(int argc, char **argv) {
    const uint32_t WG = argc > 1 ? atoi(argv[1]) : 256, S = 1024;
    const uint64_t total_keys = 1ull << 30;  // 8 GB written per measurement
    uint64_t *out, *in;
    const uint64_t room_keys = total_keys / WG / S + 4096;
    CHECK(hipMalloc(&out, (uint64_t)WG * S * room_keys * 8));
    CHECK(hipMalloc(&in, total_keys * 8 + (1 << 20)));
    CHECK(hipMemset(in, 1, total_keys * 8));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    printf("WG=%u streams/WG=%u  8 GB of 8-byte keys per measurement; GB/s of written bytes\n", WG, S);
    printf("%8s %6s %6s %6s | %9s %9s\n", "run[B]", "skew", "jitter", "read", "ms", "GB/s");
    const bool fine = argc > 2;  // second argument: sweep the misalignment in 8-byte steps instead
    for (int read = 0; read < 2; read++)
        for (uint32_t run_keys : {8u, 16u, 32u, 64u, 128u})
            for (uint32_t skew : {0u, 1u, 2u, 3u, 4u, 5u, 8u, 12u})
                for (uint32_t jitter : {0u, 7u}) {
                    if (!fine && skew != 0 && skew != 5) continue;
                    if (fine && (jitter || (run_keys != 16 && run_keys != 32))) continue;
                    if (jitter && !skew) continue;
                    const uint32_t rounds = (uint32_t)(total_keys / WG / S / run_keys);
                    float best = 1e9;
                    for (int rep = 0; rep < 3; rep++) {
                        CHECK(hipEventRecord(a));
                        if (read) hipLaunchKernelGGL(scatter_kernel<true>, dim3(WG), dim3(1024), 0, 0, out, in, S, run_keys, rounds, room_keys, skew, jitter);
                        else hipLaunchKernelGGL(scatter_kernel<false>, dim3(WG), dim3(1024), 0, 0, out, in, S, run_keys, rounds, room_keys, skew, jitter);
                        CHECK(hipEventRecord(b));
                        CHECK(hipEventSynchronize(b));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, a, b));
                        if (ms < best) best = ms;
                    }
                    const double bytes = (double)rounds * WG * S * run_keys * 8;
                    printf("%8u %6u %6u %6d | %9.3f %9.1f\n", run_keys * 8, skew * 8, jitter, read, best, bytes / best / 1e6);
                }
    return 0;
}
