// Host side of the C ABI: error reporting, ctx lifetime, and the integer helpers
// (canonical-rank map, reverse complement, numeric<->ACGT, CGR coordinates).
// Behaviour follows the reference functions cited in include/kmertools_hip.h; the
// implementations are this library's own (enumeration in ascending order instead
// of HashSet+sort, bit tricks instead of per-base loops).
#include <string.h>

#include <new>
#include <vector>

#include "kt_device.hpp"
#include "kt_internal.hpp"

namespace kt {

static thread_local std::string g_err;

void set_error(const std::string &msg) { g_err = msg; }
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

int Scratch::reserve(size_t bytes) {
    if (bytes <= cap) return KT_OK;
    if (p) {
        (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
        p = nullptr;
        return fail(KT_ERR_NOMEM, std::string("hipMalloc scratch: ") + hipGetErrorString(e));
    }
    cap = want;
    return KT_OK;
}
void Scratch::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}

uint64_t rev_comp_bits(uint64_t kmer, int k) { return ktd::rev_comp(kmer, k); }

uint32_t build_canon_lut(int k, uint16_t *lut_full) {
    const uint64_t n = 1ull << (2 * k);
    // ascending enumeration visits canonical k-mers (km <= rc(km)) in sorted order,
    // so their rank is a running counter; non-canonical k-mers take their partner's rank.
    uint32_t rank = 0;
    for (uint64_t km = 0; km < n; km++) {
        uint64_t rc = ktd::rev_comp(km, k);
        if (km <= rc) lut_full[km] = (uint16_t)rank++;
    }
    for (uint64_t km = 0; km < n; km++) {
        uint64_t rc = ktd::rev_comp(km, k);
        if (km > rc) lut_full[km] = lut_full[rc];
    }
    return rank;
}

}  // namespace kt

int kt_ctx::use() {
    KT_HIP(hipSetDevice(device));
    return KT_OK;
}

int kt_ctx::canon_lut(int k, const uint16_t **out) {
    if (k < 1 || k > kt::KT_MAX_OLIGO_K) return kt::fail(KT_ERR_ARG, "canon_lut: k out of range");
    if (!lut_dev[k]) {
        const size_t n = (size_t)1 << (2 * k);
        std::vector<uint16_t> h(n);
        kt::build_canon_lut(k, h.data());
        for (auto &x : h) x = (uint16_t)(x * 4u);  // the device copy holds byte offsets of the u32 counters (k <= 7: < 2^15)
        uint16_t *d = nullptr;
        KT_HIP(hipMalloc((void **)&d, n * sizeof(uint16_t)));
        hipError_t e = hipMemcpy(d, h.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d);
            return kt::fail(KT_ERR_HIP, std::string("hipMemcpy lut: ") + hipGetErrorString(e));
        }
        lut_dev[k] = d;
    }
    *out = lut_dev[k];
    return KT_OK;
}

extern "C" {

int kt_version(void) { return 100; }

const char *kt_last_error(void) { return kt::g_err.c_str(); }

int kt_device_count(int *count) {
    if (!count) return kt::fail(KT_ERR_ARG, "kt_device_count: null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return kt::fail(KT_ERR_NODEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    }
    *count = n;
    return KT_OK;
}

int kt_ctx_create(int device, void *stream, int own_stream, kt_ctx **out) {
    if (!out) return kt::fail(KT_ERR_ARG, "kt_ctx_create: null out");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return kt::fail(KT_ERR_NODEVICE, std::string("no HIP device: ") +
                                             (e != hipSuccess ? hipGetErrorString(e) : "count = 0"));
    if (device < 0 || device >= n) return kt::fail(KT_ERR_ARG, "kt_ctx_create: device index out of range");
    kt_ctx *c = new (std::nothrow) kt_ctx();
    if (!c) return kt::fail(KT_ERR_NOMEM, "kt_ctx_create: out of host memory");
    c->device = device;
    e = hipSetDevice(device);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (e == hipSuccess) {
        if (!own_stream) {
            c->stream = (hipStream_t)stream;
        } else {
            e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
            c->own_stream = (e == hipSuccess);
        }
    }
    if (e != hipSuccess) {
        delete c;
        return kt::fail(KT_ERR_HIP, std::string("kt_ctx_create: ") + hipGetErrorString(e));
    }
    *out = c;
    return KT_OK;
}

int kt_ctx_destroy(kt_ctx *ctx) {
    if (!ctx) return KT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->lut_dev)
        if (p) (void)hipFree(p);
    for (auto &p : ctx->lut32_dev)
        if (p) (void)hipFree(p);
    ctx->oligo_tune.release();
    ctx->s_bases.release();
    ctx->s_offsets.release();
    ctx->s_out.release();
    ctx->s_aux0.release();
    ctx->s_aux1.release();
    ctx->s_aux2.release();
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return KT_OK;
}

int kt_ctx_sync(kt_ctx *ctx) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_ctx_sync: null ctx");
    if (int rc = ctx->use()) return rc;
    KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

int kt_device_memory(kt_ctx *ctx, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (!ctx || !free_bytes || !total_bytes) return kt::fail(KT_ERR_ARG, "kt_device_memory: null");
    if (int rc = ctx->use()) return rc;
    size_t f = 0, t = 0;
    KT_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = f;
    *total_bytes = t;
    return KT_OK;
}

// Where a large device array lies in the HBM decides how fast the store-bound kernels can write it (DESIGN.md 4.1,
// profiles/r3_oligo_placement.txt: the same oligo launch ran at 1.90-2.40 ms on 23 equal allocations of one process,
// reproducibly per allocation).  A caller that keeps an output array for many launches can ask for a good one.
int kt_device_alloc_placed(kt_ctx *ctx, uint64_t bytes, int candidates, int launches, kt_probe_fn probe, void *user,
                           void **out, double *ms, int *n_tried, int *picked) {
    if (!ctx || !out || !bytes) return kt::fail(KT_ERR_ARG, "kt_device_alloc_placed: null");
    if (int rc = ctx->use()) return rc;
    *out = nullptr;
    if (candidates < 1 || !probe) candidates = 1;
    if (candidates > 16) candidates = 16;
    if (launches < 2) launches = 2;
    // all candidates alive together (so that they are different memory) - as long as they take at most 60 % of what is free
    size_t free_b = 0, total_b = 0;
    KT_HIP(hipMemGetInfo(&free_b, &total_b));
    const uint64_t fit = (uint64_t)((double)free_b * 0.6) / bytes;
    if ((uint64_t)candidates > fit) candidates = fit < 1 ? 1 : (int)fit;
    void *arr[16] = {};
    int n = 0;
    for (; n < candidates; n++) {
        if (hipMalloc(&arr[n], bytes) != hipSuccess) {
            (void)hipGetLastError();  // (somebody else's memory: fewer candidates)
            arr[n] = nullptr;
            break;
        }
    }
    if (n == 0) return kt::fail(KT_ERR_NOMEM, "kt_device_alloc_placed: out of device memory");
    int best = 0;
    int rc = KT_OK;
    if (n > 1) {
        hipEvent_t a = nullptr, b = nullptr;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) rc = kt::fail(KT_ERR_HIP, "kt_device_alloc_placed: events");
        double best_ms = 0;
        for (int w = 0; rc == KT_OK && w < 20; w++) rc = probe(user, arr[0]);  // (nothing is measured cold)
        for (int i = 0; rc == KT_OK && i < n; i++) {
            rc = probe(user, arr[i]);  // (the first call into an array is not counted)
            if (rc == KT_OK && hipEventRecord(a, ctx->stream) != hipSuccess) rc = kt::fail(KT_ERR_HIP, "kt_device_alloc_placed: event");
            for (int l = 1; rc == KT_OK && l < launches; l++) rc = probe(user, arr[i]);
            if (rc == KT_OK && (hipEventRecord(b, ctx->stream) != hipSuccess || hipEventSynchronize(b) != hipSuccess))
                rc = kt::fail(KT_ERR_HIP, "kt_device_alloc_placed: event");
            float t = 0;
            if (rc == KT_OK && hipEventElapsedTime(&t, a, b) != hipSuccess) rc = kt::fail(KT_ERR_HIP, "kt_device_alloc_placed: event");
            const double per = (double)t / (launches - 1);
            if (ms) ms[i] = per;
            if (i == 0 || per < best_ms) {
                best_ms = per;
                best = i;
            }
        }
        if (a) (void)hipEventDestroy(a);
        if (b) (void)hipEventDestroy(b);
        if (rc != KT_OK && rc > 0 && rc != KT_ERR_HIP) kt::set_error("kt_device_alloc_placed: the probe failed");
    } else if (ms) {
        ms[0] = 0;
    }
    for (int i = 0; i < n; i++)
        if (i != best || rc != KT_OK) (void)hipFree(arr[i]);
    if (rc != KT_OK) return rc;
    *out = arr[best];
    if (n_tried) *n_tried = n;
    if (picked) *picked = best;
    return KT_OK;
}

int kt_device_free(kt_ctx *ctx, void *ptr) {
    if (!ctx) return kt::fail(KT_ERR_ARG, "kt_device_free: null ctx");
    if (!ptr) return KT_OK;
    if (int rc = ctx->use()) return rc;
    KT_HIP(hipFree(ptr));
    return KT_OK;
}

int kt_host_register(kt_ctx *ctx, void *ptr, size_t bytes) {
    if (!ctx || !ptr || !bytes) return kt::fail(KT_ERR_ARG, "kt_host_register: null");
    if (int rc = ctx->use()) return rc;
    KT_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return KT_OK;
}

int kt_host_unregister(kt_ctx *ctx, void *ptr) {
    if (!ctx || !ptr) return kt::fail(KT_ERR_ARG, "kt_host_unregister: null");
    if (int rc = ctx->use()) return rc;
    KT_HIP(hipHostUnregister(ptr));
    return KT_OK;
}

int kt_bins(int k, int count_min, uint64_t *bins) {
    if (!bins) return kt::fail(KT_ERR_ARG, "kt_bins: null");
    if (k < 1 || k > 15) return kt::fail(KT_ERR_ARG, "kt_bins: k must be in 1..15");
    const uint64_t n = 1ull << (2 * k);
    // palindromes exist only for even k: (4^k + 4^(k/2)) / 2, else 4^k / 2  (kmer.rs:55)
    *bins = count_min ? ((k & 1) ? n / 2 : (n + (1ull << k)) / 2) : n;
    return KT_OK;
}

int kt_pos_map(int k, uint32_t *min_mer_pos_map, uint64_t *pos_min_mer, uint32_t *kcount) {
    if (k < 1 || k > 12) return kt::fail(KT_ERR_ARG, "kt_pos_map: k must be in 1..12");
    const uint64_t n = 1ull << (2 * k);
    uint32_t rank = 0;
    for (uint64_t km = 0; km < n; km++) {
        const uint64_t rc = ktd::rev_comp(km, k);
        if (km <= rc) {
            if (min_mer_pos_map) min_mer_pos_map[km] = rank;
            if (pos_min_mer) pos_min_mer[rank] = km;
            rank++;
        } else if (min_mer_pos_map) {
            min_mer_pos_map[km] = 0;  // the reference leaves non-canonical slots at 0
        }
    }
    if (kcount) *kcount = rank;
    return KT_OK;
}

uint64_t kt_rev_comp(uint64_t kmer, int k) {
    if (k < 1 || k > 32) return 0;
    return ktd::rev_comp(kmer, k);
}

int kt_numeric_to_kmer(uint64_t kmer, int k, char *out) {
    if (!out || k < 0 || k > 32) return kt::fail(KT_ERR_ARG, "kt_numeric_to_kmer: bad argument");
    for (int i = 0; i < k; i++) out[i] = "ACGT"[(kmer >> (2 * (k - 1 - i))) & 3];
    out[k] = 0;
    return KT_OK;
}

int kt_kmer_to_numeric(const char *kmer, uint64_t len, uint64_t *fwd, uint64_t *rev) {
    if (!kmer || !fwd || !rev) return kt::fail(KT_ERR_ARG, "kt_kmer_to_numeric: null");
    if (len > 32)  // pybindings/src/kmer.rs:58-63
        return kt::fail(KT_ERR_ARG, "Invalid k-mer length: " + std::to_string(len) + ", must be <= 32");
    if (len == 0) return kt::fail(KT_ERR_ARG, "kt_kmer_to_numeric: empty k-mer");
    // like the reference, no validity check: an invalid letter contributes code 4
    // (lib.rs:42-47), i.e. it ORs into the next base's low bit.
    const uint64_t mask = len >= 32 ? ~0ull : ((1ull << (2 * len)) - 1);
    const uint64_t shift = 2 * (len - 1);
    uint64_t f = 0, r = 0;
    for (uint64_t i = 0; i < len; i++) {
        const unsigned char c = (unsigned char)kmer[i];
        uint64_t v;
        switch (c) {
            case 0: case 'A': case 'a': v = 0; break;
            case 1: case 'C': case 'c': v = 1; break;
            case 2: case 'G': case 'g': v = 2; break;
            case 3: case 'T': case 't': case 'U': case 'u': v = 3; break;
            default: v = 4; break;
        }
        f = ((f << 2) | v) & mask;
        r = (r >> 2) | ((v ^ 3) << shift);
    }
    *fwd = f;
    *rev = r;
    return KT_OK;
}

int kt_cgr_coords(int k, double vecsize, double *xy) {
    if (!xy) return kt::fail(KT_ERR_ARG, "kt_cgr_coords: null");
    if (k < 1 || k > 12) return kt::fail(KT_ERR_ARG, "kt_cgr_coords: k must be in 1..12");
    // corners (oligocgr.rs:166-170): A(0,0) C(0,vs) G(vs,vs) T(vs,0); start at the centre,
    // one midpoint step per base, first (most significant) base first.
    const double cx[4] = {0.0, 0.0, vecsize, vecsize};
    const double cy[4] = {0.0, vecsize, vecsize, 0.0};
    const uint64_t n = 1ull << (2 * k);
    uint64_t i = 0;
    for (uint64_t km = 0; km < n; km++) {
        if (km > ktd::rev_comp(km, k)) continue;
        double x = vecsize / 2.0, y = vecsize / 2.0;
        for (int j = k - 1; j >= 0; j--) {
            const int c = (int)((km >> (2 * j)) & 3);
            x = (cx[c] + x) / 2.0;
            y = (cy[c] + y) / 2.0;
        }
        xy[2 * i] = x;
        xy[2 * i + 1] = y;
        i++;
    }
    return KT_OK;
}

uint32_t kt_owner_of(uint64_t kmer, uint32_t n_owners) {
    return n_owners ? ktd::owner_of(kmer, n_owners) : 0;
}

}  // extern "C"
