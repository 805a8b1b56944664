// kt_synth.hip - deterministic synthetic reads generated directly in HBM,
// so benchmark inputs never cross PCIe.  Counter-based: base (read, pos) depends only on
// (seed, read, pos); the parity tests keep a bit-exact CPU mirror (kto_synth_reads, under oracle/) used
// to regenerate any slice.
#include "kt_device.hpp"
#include "kt_internal.hpp"

namespace {

__global__ __launch_bounds__(256) void synth_kernel(uint64_t seed, uint64_t first_read, uint64_t n_reads,
                                                    uint32_t read_len, int noise, uint64_t genome_len,
                                                    uint8_t *__restrict__ bases, uint64_t *__restrict__ offsets) {
    const uint64_t total = n_reads * read_len;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t ri = i / read_len;
        const uint32_t p = (uint32_t)(i - ri * read_len);
        const uint64_t rid = first_read + ri;
        const uint64_t g = rid * read_len + p;
        const uint64_t h = ktd::mix64(seed + g);
        uint32_t code;
        if (genome_len) {
            const uint64_t hr = ktd::mix64(seed ^ 0x5eed5eed5eedull ^ ktd::mix64(rid));
            const uint64_t start = (hr >> 1) % (genome_len - read_len + 1);
            const uint32_t strand = (uint32_t)(hr & 1);
            const uint64_t j = strand ? start + (read_len - 1 - p) : start + p;
            code = (uint32_t)(ktd::mix64((seed ^ 0x67656e6f6d65ull) + j) & 3);
            if (strand) code = 3 - code;
            if (((h >> 40) % 100) == 0) code = (code + 1 + (uint32_t)((h >> 50) % 3)) & 3;
        } else {
            code = (uint32_t)(h & 3);
        }
        uint8_t c = (uint8_t)((0x54474341u >> (8 * code)) & 0xFF);  // "ACGT"
        if (noise) {
            if (((h >> 8) & 0xFFFFF) < 1049) c = 'N';
            else if (((h >> 28) & 0xFFF) < 41) c = (uint8_t)(c | 0x20);
        }
        bases[i] = c;
    }
    if (offsets) {
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i <= n_reads; i += stride)
            offsets[i] = i * read_len;
    }
}

}  // namespace

extern "C" int kt_synth_reads(kt_ctx *ctx, uint64_t seed, uint64_t first_read, uint64_t n_reads,
                              uint32_t read_len, int noise, uint64_t genome_len, uint8_t *bases_dev,
                              uint64_t *offsets_dev) {
    if (!ctx || !bases_dev) return kt::fail(KT_ERR_ARG, "kt_synth_reads: null");
    if (read_len == 0) return kt::fail(KT_ERR_ARG, "kt_synth_reads: read_len must be > 0");
    if (genome_len && genome_len < read_len) return kt::fail(KT_ERR_ARG, "kt_synth_reads: genome shorter than a read");
    if (int rc = ctx->use()) return rc;
    if (n_reads == 0) return KT_OK;
    const uint64_t total = n_reads * read_len;
    uint64_t blocks = (total + 255) / 256;
    const uint64_t cap = (uint64_t)ctx->n_cu * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, ctx->stream, seed, first_read, n_reads,
                       read_len, noise, genome_len, bases_dev, offsets_dev);
    KT_HIP(hipGetLastError());
    return KT_OK;
}
