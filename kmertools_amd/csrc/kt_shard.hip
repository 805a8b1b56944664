// kt_shard.hip - one k-mer table sharded over the GPUs of a node by hash prefix, behind the C ABI.
//
// Replaces the reference's `min_mer % n_parts` partitioning and per-partition merge (counter/src/lib.rs:100,127,
// 188-231): the partitions are GPUs, rank o owns every canonical k-mer with kt_owner_of(kmer, n_ranks) == o, and the
// merge is one exchange of raw 8-byte k-mers ("route, then count": nearly every 31-mer of a read set is unique, so
// counting before the exchange would move 12 bytes per k-mer instead of 8).
//
// One rank = one process (or thread) = one kt_ctx; every rank makes the same sequence of collective calls
// (add_reads, finalize).  A batch is cut into KT_SHARD_SLICES slices of whole 8192-base segments and pipelined over two
// streams:
//     main stream   route(i+1): one front-end pass writes the slice's canonical k-mers into per-owner regions of
//                   fixed capacity (count in the region's header: no sizes are exchanged, no host round trip)
//     comm stream   exchange(i): grouped ncclSend / ncclRecv of the fixed-size regions with every peer - all seven
//                   xGMI links of a GPU busy at once - through librccl (loaded with dlopen; the torch extension's
//                   copy when there is one), or a caller-supplied host all-to-all (tests: gloo; MPI would fit too)
//     main stream   level 1 of the bulk build over what slice i-1 brought (kt_bulk_add_keys with the device-side
//                   counts), appended to the same partition buffers; after the last slice: level 2 + range builds.
// A region that overflows (a batch dominated by few k-mers sends most of its keys to one owner) parks the excess in
// a local pending list, which finalize delivers in fixed-size rounds through the probing path.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <string>
#include <vector>

#include "kt_internal.hpp"
#include "kt_launch.hpp"
#include "kt_segment.hpp"
#include "kt_table.hpp"

namespace {

using ktseg::SegArgs;
using ktseg::SegShared;
constexpr int BLOCK = ktseg::BLOCK;
constexpr int MAX_RANKS = 64;
constexpr uint64_t HDR_U64 = 8;  // 64-byte region header: [0] keys in the region (may exceed the capacity: clamp),
                                 // [1] keys still pending at the sender (finalize rounds)
constexpr uint64_t FIN_CAP = 1u << 17;  // keys per peer and finalize round (1 MiB messages)

// ---- librccl, resolved at run time -----------------------------------------------------------------------------
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

int load_rccl(Rccl **out) {
    static Rccl r;
    static int state = 0;  // 0 = not tried, 1 = ok, 2 = failed
    if (state == 0) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);  // by soname: the copy a torch extension has loaded, if any
            if (r.h) break;
        }
        state = 2;
        if (r.h) {
#define KT_SYM(field, name) *(void **)(&r.field) = dlsym(r.h, name)
            KT_SYM(GetUniqueId, "ncclGetUniqueId");
            KT_SYM(CommInitRank, "ncclCommInitRank");
            KT_SYM(CommDestroy, "ncclCommDestroy");
            KT_SYM(GroupStart, "ncclGroupStart");
            KT_SYM(GroupEnd, "ncclGroupEnd");
            KT_SYM(Send, "ncclSend");
            KT_SYM(Recv, "ncclRecv");
            KT_SYM(GetErrorString, "ncclGetErrorString");
#undef KT_SYM
            if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv &&
                r.GetErrorString)
                state = 1;
        }
    }
    if (state != 1) return kt::fail(KT_ERR_HIP, "librccl.so.1 could not be loaded (dlopen / missing symbols)");
    *out = &r;
    return KT_OK;
}

#define KT_NCCL(rc_, expr)                                                                            \
    do {                                                                                              \
        ncclResult_t _r = (expr);                                                                     \
        if (_r != ncclSuccess) return kt::fail(KT_ERR_HIP, std::string(#expr) + ": " + (rc_)->GetErrorString(_r)); \
    } while (0)

// ---- route: the slice's canonical k-mers into per-owner regions --------------------------------------------------
// One front-end pass.  Per 8192-base segment: the thread's 32 canonical k-mers stay in registers, the workgroup
// counts them per owner (LDS), reserves one contiguous run per owner in the region (one global atomic per owner and
// segment: ~1000 keys = 8 KB runs for 8 owners) and every lane writes its keys into the runs.  What does not fit the
// region goes to the pending list.
__global__ __launch_bounds__(BLOCK) void route_regions_kernel(SegArgs a, uint64_t seg_lo, uint64_t seg_hi,
                                                              uint32_t n_owners, uint64_t cap_keys,
                                                              uint64_t msg_stride, uint64_t *__restrict__ msgs,
                                                              uint64_t *__restrict__ pend_keys, uint64_t pend_cap,
                                                              uint64_t *__restrict__ pend_n,
                                                              uint32_t *__restrict__ flags) {
    __shared__ SegShared sm;
    __shared__ uint32_t cnt[MAX_RANKS], fill[MAX_RANKS];
    __shared__ uint64_t base[MAX_RANKS];
    if (threadIdx.x < MAX_RANKS) cnt[threadIdx.x] = 0;
    ktd::lds_barrier();
    for (uint64_t g = seg_lo + blockIdx.x; g < seg_hi; g += gridDim.x) {
        uint64_t keys[ktseg::PER_THREAD];
        uint32_t ok;
        ktseg::collect_kmers(a, g, sm, keys, ok);
#pragma unroll
        for (uint32_t j = 0; j < ktseg::PER_THREAD; j++)
            if ((ok >> j) & 1u) atomicAdd(&cnt[ktd::owner_of(keys[j], n_owners)], 1u);
        ktd::lds_barrier();
        if (threadIdx.x < n_owners) {
            const uint32_t c = cnt[threadIdx.x];
            base[threadIdx.x] = c ? atomicAdd(reinterpret_cast<unsigned long long *>(msgs + threadIdx.x * msg_stride),
                                              (unsigned long long)c)
                                  : 0;
            fill[threadIdx.x] = 0;
            cnt[threadIdx.x] = 0;
        }
        ktd::lds_barrier();
#pragma unroll
        for (uint32_t j = 0; j < ktseg::PER_THREAD; j++) {
            if (!((ok >> j) & 1u)) continue;
            const uint32_t o = ktd::owner_of(keys[j], n_owners);
            const uint64_t pos = base[o] + atomicAdd(&fill[o], 1u);
            if (pos < cap_keys) {
                msgs[o * msg_stride + HDR_U64 + pos] = keys[j];
            } else {
                const uint64_t at = atomicAdd(reinterpret_cast<unsigned long long *>(pend_n), 1ull);
                if (at < pend_cap) pend_keys[at] = keys[j];
                else atomicOr(flags, 1u);
            }
        }
        ktd::lds_barrier();
    }
}

// The same routing with the shape of kt_bulk.hip's scatter1w: a workgroup is four 256-thread groups, each staging its
// own segment; the thread's walk over its 32 window starts is cut in halves, so a round handles 16 K k-mers with 32 key
// registers and 16 waves fit a CU.  The round's keys are counting-sorted by owner in LDS (ranks from LDS atomics on
// per-wave counters: a wave's lanes only contend with each other), one global atomic per owner reserves the run in
// the owner's region, and the runs - ~2000 keys per owner with 8 owners - are copied out with consecutive lanes on
// consecutive keys.  (route_regions_kernel has every lane store its own 8 bytes somewhere in an 8 KB run: 25 ms for
// the 25 M-read batch against 18 ms for level 1 itself.)  Up to RW_MAXO owners; more take the kernel above.
constexpr int RW_T = 1024, RW_PER = 16, RW_MAXO = 16;
struct RouteWShared {
    union {  // (the staged segments are dead once every thread has loaded its window words)
        SegShared seg[RW_T / BLOCK];
        uint64_t sorted[RW_T * RW_PER];
    };
    uint32_t wcnt[RW_T / 64][RW_MAXO];  // keys of (wave, owner) this round; then where the wave's keys of the owner start
    uint32_t tot[RW_MAXO];              // keys of the owner this round
    uint32_t start[RW_MAXO + 1];        // the owners' runs in sorted[]
    uint64_t gbase[RW_MAXO];            // where the owner's run goes in its region
};
static_assert(sizeof(RouteWShared) <= 160 * 1024, "LDS of a CU");

__global__ __launch_bounds__(RW_T) void route_wide_kernel(SegArgs a, uint64_t seg_lo, uint64_t seg_hi, uint32_t n_owners,
                                                          uint64_t cap_keys, uint64_t msg_stride,
                                                          uint64_t *__restrict__ msgs, uint64_t *__restrict__ pend_keys,
                                                          uint64_t pend_cap, uint64_t *__restrict__ pend_n,
                                                          uint32_t *__restrict__ flags) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    RouteWShared &sm = *reinterpret_cast<RouteWShared *>(smem_raw);
    constexpr int NQ = ktseg::PER_THREAD / RW_PER;
    constexpr uint32_t GROUPS = RW_T / BLOCK;
    const uint32_t tid = threadIdx.x, grp = tid / BLOCK, t = tid % BLOCK, wave = tid >> 6;
    for (uint64_t g0 = seg_lo + (uint64_t)blockIdx.x * GROUPS; g0 < seg_hi; g0 += (uint64_t)gridDim.x * GROUPS) {
        const bool valid = g0 + grp < seg_hi;  // (a group past the end walks the last segment and keeps nothing)
        ktseg::stage_segment(a, valid ? g0 + grp : seg_hi - 1, sm.seg[grp], t);
        ktseg::Window w(sm.seg[grp], t, a.k);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            uint64_t keys[RW_PER];
            uint32_t ok = 0;
#pragma unroll
            for (int j = 0; j < RW_PER; j++) {
                keys[j] = w.f < w.r ? w.f : w.r;
                ok |= (valid && w.ok((uint32_t)(q * RW_PER + j)) ? 1u : 0u) << j;
                w.step();
            }
            if (tid < (RW_T / 64) * RW_MAXO) (&sm.wcnt[0][0])[tid] = 0;
            ktd::lds_barrier();
            uint32_t rk[RW_PER / 2];  // rank of the key among its wave's keys of the same owner, two per register
            uint64_t own = 0;         // owners, four bits each
#pragma unroll
            for (int j = 0; j < RW_PER; j++) {
                uint32_t r = 0;
                if ((ok >> j) & 1u) {
                    const uint32_t o = ktd::owner_of(keys[j], n_owners);
                    own |= (uint64_t)o << (4 * j);
                    r = atomicAdd(&sm.wcnt[wave][o], 1u);
                }
                rk[j / 2] = (j & 1) ? rk[j / 2] | (r << 16) : r;
            }
            ktd::lds_barrier();
            if (tid < n_owners) {  // the owner's keys of the sixteen waves, in wave order; its run in the region
                uint32_t run = 0;
                for (uint32_t v = 0; v < RW_T / 64; v++) {
                    const uint32_t c = sm.wcnt[v][tid];
                    sm.wcnt[v][tid] = run;
                    run += c;
                }
                sm.tot[tid] = run;
                sm.gbase[tid] = run ? atomicAdd(reinterpret_cast<unsigned long long *>(msgs + tid * msg_stride),
                                                (unsigned long long)run)
                                    : 0;
            }
            ktd::lds_barrier();
            if (tid <= n_owners) {
                uint32_t before = 0;
                for (uint32_t o = 0; o < tid; o++) before += sm.tot[o];
                sm.start[tid] = before;
            }
            ktd::lds_barrier();
#pragma unroll
            for (int j = 0; j < RW_PER; j++) {
                if ((ok >> j) & 1u) {
                    const uint32_t o = (uint32_t)(own >> (4 * j)) & 15u;
                    const uint32_t r = (j & 1) ? rk[j / 2] >> 16 : rk[j / 2] & 0xFFFFu;
                    sm.sorted[sm.start[o] + sm.wcnt[wave][o] + r] = keys[j];
                }
            }
            ktd::lds_barrier();
            for (uint32_t o = 0; o < n_owners; o++) {
                const uint32_t s0 = sm.start[o], c = sm.start[o + 1] - s0;
                const uint64_t b = sm.gbase[o];
                for (uint32_t i = tid; i < c; i += RW_T) {
                    const uint64_t key = sm.sorted[s0 + i], pos = b + i;
                    if (pos < cap_keys) {
                        msgs[o * msg_stride + HDR_U64 + pos] = key;
                    } else {
                        const uint64_t at = atomicAdd(reinterpret_cast<unsigned long long *>(pend_n), 1ull);
                        if (at < pend_cap) pend_keys[at] = key;
                        else atomicOr(flags, 1u);
                    }
                }
            }
            ktd::lds_barrier();
        }
    }
}

// finalize round: up to FIN_CAP pending keys per owner into the round's messages; the rest stays (compacted by
// leaving KT_EMPTY_KEY holes: the list is rescanned next round).  left[0] = keys still pending after this round.
__global__ __launch_bounds__(BLOCK) void pack_pending_kernel(uint64_t *__restrict__ pend_keys, uint64_t n,
                                                             uint32_t n_owners, uint64_t msg_stride,
                                                             uint64_t *__restrict__ msgs, uint64_t *__restrict__ left) {
    uint64_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t key = pend_keys[i];
        if (key == KT_EMPTY_KEY) continue;
        const uint32_t o = ktd::owner_of(key, n_owners);
        const uint64_t pos = atomicAdd(reinterpret_cast<unsigned long long *>(msgs + o * msg_stride), 1ull);
        if (pos < FIN_CAP) {
            msgs[o * msg_stride + HDR_U64 + pos] = key;
            pend_keys[i] = KT_EMPTY_KEY;
        } else {
            mine++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(reinterpret_cast<unsigned long long *>(left), (unsigned long long)mine);
}

__global__ void stamp_left_kernel(uint64_t *msgs, uint32_t n_owners, uint64_t msg_stride, const uint64_t *left) {
    if (threadIdx.x < n_owners) msgs[threadIdx.x * msg_stride + 1] = *left;
}

}  // namespace

struct kt_sharded {
    kt_ctx *ctx = nullptr;
    kt_ctr *table = nullptr;
    int k = 0, n_ranks = 1, rank = 0, n_slices = 4;
    bool routed = false;  // false: a single rank, everything goes straight to the table
    uint64_t max_batch_bases = 0;
    uint64_t cap_keys = 0, msg_u64 = 0;  // per (slice, owner) region: keys of room; message size in 8-byte words
    uint64_t *send[2] = {nullptr, nullptr};  // n_ranks messages each
    uint64_t *recv = nullptr;                // n_slices * n_ranks messages: kept until the build has finished
    uint64_t *fin_send = nullptr, *fin_recv = nullptr;  // n_ranks messages of FIN_CAP keys each
    uint64_t *pend_keys = nullptr, *pend_n = nullptr, *fin_left = nullptr;
    uint64_t pend_cap = 0;
    uint32_t *flags = nullptr;
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_routed[2] = {nullptr, nullptr}, ev_sent[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> ev_recv;
    // transport
    Rccl *rccl = nullptr;
    ncclComm_t comm = nullptr;
    kt_alltoall_fn fn = nullptr;
    void *fn_user = nullptr;
    void *h_send = nullptr, *h_recv = nullptr;  // pinned staging for the host transport
    size_t h_bytes = 0;
    uint64_t exchanged_bytes = 0;  // sent to other ranks so far (statistics)
};

namespace {

// moves n_ranks fixed-size messages: message p of `src` goes to rank p, message p of `dst` comes from rank p.
// Enqueued on the comm stream (RCCL) or carried out before returning (host transport).
int exchange(kt_sharded *s, const uint64_t *src, uint64_t *dst, uint64_t words) {
    const size_t bytes = words * 8;
    if (s->fn) {
        const size_t all = bytes * s->n_ranks;
        if (s->h_bytes < all) {
            if (s->h_send) (void)hipHostFree(s->h_send);
            if (s->h_recv) (void)hipHostFree(s->h_recv);
            s->h_send = s->h_recv = nullptr;
            s->h_bytes = 0;
            KT_HIP(hipHostMalloc(&s->h_send, all, hipHostMallocDefault));
            KT_HIP(hipHostMalloc(&s->h_recv, all, hipHostMallocDefault));
            s->h_bytes = all;
        }
        KT_HIP(hipMemcpyAsync(s->h_send, src, all, hipMemcpyDeviceToHost, s->comm_stream));
        KT_HIP(hipStreamSynchronize(s->comm_stream));
        if (s->fn(s->fn_user, s->h_send, s->h_recv, (uint64_t)bytes) != 0)
            return kt::fail(KT_ERR_HIP, "sharded counter: the caller's all-to-all failed");
        KT_HIP(hipMemcpyAsync(dst, s->h_recv, all, hipMemcpyHostToDevice, s->comm_stream));
        KT_HIP(hipStreamSynchronize(s->comm_stream));  // the staging buffers are reused by the next exchange
    } else {
        KT_NCCL(s->rccl, s->rccl->GroupStart());
        for (int p = 0; p < s->n_ranks; p++) {
            if (p == s->rank) continue;
            KT_NCCL(s->rccl, s->rccl->Send(src + (uint64_t)p * words, bytes, ncclUint8, p, s->comm, s->comm_stream));
            KT_NCCL(s->rccl, s->rccl->Recv(dst + (uint64_t)p * words, bytes, ncclUint8, p, s->comm, s->comm_stream));
        }
        KT_NCCL(s->rccl, s->rccl->GroupEnd());
        KT_HIP(hipMemcpyAsync(dst + (uint64_t)s->rank * words, src + (uint64_t)s->rank * words, bytes,
                              hipMemcpyDeviceToDevice, s->comm_stream));
    }
    s->exchanged_bytes += bytes * (uint64_t)(s->n_ranks - 1);
    return KT_OK;
}

int sharded_alloc(kt_sharded *s) {
    kt_ctx *ctx = s->ctx;
    if (int rc = ctx->use()) return rc;
    const uint64_t per_region = s->max_batch_bases / (uint64_t)s->n_slices / (uint64_t)s->n_ranks;
    s->cap_keys = (per_region + per_region / 8 + 4096 + 7) & ~7ull;
    s->msg_u64 = HDR_U64 + s->cap_keys;
    const size_t msg = s->msg_u64 * 8;
    const uint64_t fin_u64 = HDR_U64 + FIN_CAP;
    s->pend_cap = s->max_batch_bases / 4 + (1u << 16);  // (a batch that sends more than a quarter of its k-mers past the
                                                         // regions' room - one k-mer making up most of it - fails loudly)
    hipError_t e = hipSuccess;
    for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipMalloc((void **)&s->send[b], msg * s->n_ranks);
    if (e == hipSuccess) e = hipMalloc((void **)&s->recv, msg * s->n_ranks * s->n_slices);
    if (e == hipSuccess) e = hipMalloc((void **)&s->fin_send, fin_u64 * 8 * s->n_ranks);
    if (e == hipSuccess) e = hipMalloc((void **)&s->fin_recv, fin_u64 * 8 * s->n_ranks);
    if (e == hipSuccess) e = hipMalloc((void **)&s->pend_keys, s->pend_cap * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->pend_n, 256);
    if (e == hipSuccess) e = hipMemset(s->pend_n, 0, 256);
    if (e != hipSuccess) return kt::fail(KT_ERR_NOMEM, std::string("sharded counter: hipMalloc: ") + hipGetErrorString(e));
    s->fin_left = s->pend_n + 8;
    s->flags = reinterpret_cast<uint32_t *>(s->pend_n + 16);
    {   // the exchange's kernels should start the moment their slice is routed, whatever the main stream is running
        int lo = 0, hi = 0;
        KT_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        KT_HIP(hipStreamCreateWithPriority(&s->comm_stream, hipStreamNonBlocking, hi));
    }
    for (int b = 0; b < 2; b++) {
        KT_HIP(hipEventCreateWithFlags(&s->ev_routed[b], hipEventDisableTiming));
        KT_HIP(hipEventCreateWithFlags(&s->ev_sent[b], hipEventDisableTiming));
    }
    s->ev_recv.resize(s->n_slices);
    for (auto &ev : s->ev_recv) KT_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return KT_OK;
}

int sharded_new(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                kt_sharded **out) {
    if (!ctx || !out) return kt::fail(KT_ERR_ARG, "kt_sharded_create: null");
    *out = nullptr;
    if (n_ranks < 1 || n_ranks > MAX_RANKS || rank < 0 || rank >= n_ranks)
        return kt::fail(KT_ERR_ARG, "kt_sharded_create: need 1 <= n_ranks <= 64 and 0 <= rank < n_ranks");
    if (max_batch_bases == 0) return kt::fail(KT_ERR_ARG, "kt_sharded_create: max_batch_bases must be > 0");
    kt_sharded *s = new (std::nothrow) kt_sharded();
    if (!s) return kt::fail(KT_ERR_NOMEM, "kt_sharded_create: host alloc");
    s->ctx = ctx;
    s->k = k;
    s->n_ranks = n_ranks;
    s->rank = rank;
    s->max_batch_bases = max_batch_bases;
    const char *env = getenv("KT_SHARD_SLICES");
    s->n_slices = env && atoi(env) > 0 ? atoi(env) : 4;
    if (s->n_slices > 64) s->n_slices = 64;
    const char *force = getenv("KT_SHARD_FORCE");  // tests: run the routed path with a single rank too
    s->routed = n_ranks > 1 || (force && atoi(force) > 0);
    int rc = kt_ctr_create(ctx, k, capacity_slots, &s->table);
    if (rc == KT_OK && s->routed) rc = sharded_alloc(s);
    if (rc != KT_OK) {
        kt_sharded_destroy(s);
        return rc;
    }
    *out = s;
    return KT_OK;
}

}  // namespace

extern "C" {

int kt_rccl_unique_id(uint8_t *id128) {
    if (!id128) return kt::fail(KT_ERR_ARG, "kt_rccl_unique_id: null");
    Rccl *r = nullptr;
    if (int rc = load_rccl(&r)) return rc;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    KT_NCCL(r, r->GetUniqueId(&id));
    memcpy(id128, &id, 128);
    return KT_OK;
}

int kt_sharded_create_rccl(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                           const uint8_t *id128, kt_sharded **out) {
    if (n_ranks > 1 && !id128) return kt::fail(KT_ERR_ARG, "kt_sharded_create_rccl: null id");
    kt_sharded *s = nullptr;
    if (int rc = sharded_new(ctx, k, capacity_slots, max_batch_bases, n_ranks, rank, &s)) return rc;
    if (s->routed) {
        int rc = load_rccl(&s->rccl);
        if (rc == KT_OK) {
            ncclUniqueId id;
            ncclResult_t r = ncclSuccess;
            if (id128) memcpy(&id, id128, 128);
            else r = s->rccl->GetUniqueId(&id);  // (a single rank made to take the routed path: tests)
            if (r == ncclSuccess) r = s->rccl->CommInitRank(&s->comm, n_ranks, id, rank);
            if (r != ncclSuccess) rc = kt::fail(KT_ERR_HIP, std::string("ncclCommInitRank: ") + s->rccl->GetErrorString(r));
        }
        if (rc != KT_OK) {
            kt_sharded_destroy(s);
            return rc;
        }
    }
    *out = s;
    return KT_OK;
}

int kt_sharded_create_host(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                           kt_alltoall_fn fn, void *user, kt_sharded **out) {
    if (!fn) return kt::fail(KT_ERR_ARG, "kt_sharded_create_host: null all-to-all function");
    kt_sharded *s = nullptr;
    if (int rc = sharded_new(ctx, k, capacity_slots, max_batch_bases, n_ranks, rank, &s)) return rc;
    s->fn = fn;
    s->fn_user = user;
    *out = s;
    return KT_OK;
}

int kt_sharded_destroy(kt_sharded *s) {
    if (!s) return KT_OK;
    if (s->ctx) {
        (void)hipSetDevice(s->ctx->device);
        if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
        (void)hipStreamSynchronize(s->ctx->stream);
    }
    if (s->comm && s->rccl) (void)s->rccl->CommDestroy(s->comm);
    for (int b = 0; b < 2; b++) {
        if (s->send[b]) (void)hipFree(s->send[b]);
        if (s->ev_routed[b]) (void)hipEventDestroy(s->ev_routed[b]);
        if (s->ev_sent[b]) (void)hipEventDestroy(s->ev_sent[b]);
    }
    for (auto ev : s->ev_recv)
        if (ev) (void)hipEventDestroy(ev);
    if (s->recv) (void)hipFree(s->recv);
    if (s->fin_send) (void)hipFree(s->fin_send);
    if (s->fin_recv) (void)hipFree(s->fin_recv);
    if (s->pend_keys) (void)hipFree(s->pend_keys);
    if (s->pend_n) (void)hipFree(s->pend_n);
    if (s->h_send) (void)hipHostFree(s->h_send);
    if (s->h_recv) (void)hipHostFree(s->h_recv);
    if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
    if (s->table) kt_ctr_destroy(s->table);
    delete s;
    return KT_OK;
}

int kt_sharded_table(kt_sharded *s, kt_ctr **table) {
    if (!s || !table) return kt::fail(KT_ERR_ARG, "kt_sharded_table: null");
    *table = s->table;
    return KT_OK;
}

int kt_sharded_clear(kt_sharded *s) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_clear: null");
    if (s->routed) {
        if (int rc = s->ctx->use()) return rc;
        KT_HIP(hipMemsetAsync(s->pend_n, 0, 256, s->ctx->stream));
    }
    return kt_ctr_clear(s->table);
}

int kt_sharded_exchanged_bytes(kt_sharded *s, uint64_t *bytes) {
    if (!s || !bytes) return kt::fail(KT_ERR_ARG, "kt_sharded_exchanged_bytes: null");
    *bytes = s->exchanged_bytes;
    return KT_OK;
}

uint64_t kt_sharded_message_bytes(uint64_t max_batch_bases, int n_ranks, int n_slices) {
    if (n_ranks < 1 || n_slices < 1) return 0;
    const uint64_t per_region = max_batch_bases / (uint64_t)n_slices / (uint64_t)n_ranks;
    return (HDR_U64 + ((per_region + per_region / 8 + 4096 + 7) & ~7ull)) * 8;
}

int kt_sharded_add_reads(kt_sharded *s, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int mem) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: null");
    if (!s->routed) return kt_ctr_add_reads(s->table, bases, offsets, n_reads, mem);
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: bad mem flag");
    kt_ctx *ctx = s->ctx;
    if (int rc = ctx->use()) return rc;
    // a rank without reads still takes part in the exchange
    uint64_t total = 0;
    if (n_reads) {
        if (!offsets) return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: null offsets");
        if (int rc = ktl::total_bases_of(ctx, offsets, n_reads, mem, &total)) return rc;
    }
    if (total > s->max_batch_bases)
        return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: batch larger than max_batch_bases (split it)");
    if (total && !bases) return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: null bases");
    const uint8_t *d_bases = bases;
    const uint64_t *d_offsets = offsets;
    if (mem == KT_MEM_HOST && total) {
        if (int rc = ktl::stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) return rc;
    }
    SegArgs a{};
    if (total) {
        if (int rc = ktl::make_seg_args(ctx, d_bases, d_offsets, n_reads, total, s->k, &a)) return rc;
    }
    const int P = s->n_slices;
    const uint64_t words = s->msg_u64, all_words = words * (uint64_t)s->n_ranks;
    // the receiving side: one bulk job over everything the slices bring (level 1 per slice, the rest at the end);
    // when the table or the batch does not suit it the received k-mers go through the probing path
    int bulk = 0;
    if (int rc = kt_bulk_begin(s->table, (uint64_t)P * s->n_ranks * s->cap_keys, &bulk)) return rc;

    auto route = [&](int i) -> int {
        const int b = i & 1;
        KT_HIP(hipStreamWaitEvent(ctx->stream, s->ev_sent[b], 0));  // the slice that used this buffer last has left it
        for (int p = 0; p < s->n_ranks; p++)
            KT_HIP(hipMemsetAsync(s->send[b] + (uint64_t)p * words, 0, HDR_U64 * 8, ctx->stream));
        const uint64_t lo = a.n_seg * (uint64_t)i / (uint64_t)P, hi = a.n_seg * (uint64_t)(i + 1) / (uint64_t)P;
        const char *rw = getenv("KT_ROUTE_WIDE");  // 0: the one-lane-one-key kernel (comparison)
        if (hi > lo && s->n_ranks <= RW_MAXO && !(rw && atoi(rw) == 0)) {
            const uint64_t quads = (hi - lo + RW_T / BLOCK - 1) / (RW_T / BLOCK);
            const uint32_t wgs = (uint32_t)(quads < (uint64_t)ctx->n_cu ? quads : (uint64_t)ctx->n_cu);
            KT_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(route_wide_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(RouteWShared)));
            hipLaunchKernelGGL(route_wide_kernel, dim3(wgs), dim3(RW_T), sizeof(RouteWShared), ctx->stream, a, lo, hi,
                               (uint32_t)s->n_ranks, s->cap_keys, words, s->send[b], s->pend_keys, s->pend_cap, s->pend_n,
                               s->flags);
            KT_HIP(hipGetLastError());
        } else if (hi > lo) {
            hipLaunchKernelGGL(route_regions_kernel, dim3(ktl::grid_for(ctx, hi - lo, 4)), dim3(BLOCK), 0, ctx->stream, a,
                               lo, hi, (uint32_t)s->n_ranks, s->cap_keys, words, s->send[b], s->pend_keys, s->pend_cap,
                               s->pend_n, s->flags);
            KT_HIP(hipGetLastError());
        }
        KT_HIP(hipEventRecord(s->ev_routed[b], ctx->stream));
        return KT_OK;
    };
    auto swap = [&](int i) -> int {
        const int b = i & 1;
        KT_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_routed[b], 0));
        if (int rc = exchange(s, s->send[b], s->recv + (uint64_t)i * all_words, words)) return rc;
        KT_HIP(hipEventRecord(s->ev_sent[b], s->comm_stream));
        KT_HIP(hipEventRecord(s->ev_recv[i], s->comm_stream));
        return KT_OK;
    };
    auto count = [&](int i) -> int {
        KT_HIP(hipStreamWaitEvent(ctx->stream, s->ev_recv[i], 0));
        for (int p = 0; p < s->n_ranks; p++) {
            const uint64_t *msg = s->recv + (uint64_t)i * all_words + (uint64_t)p * words;
            if (bulk) {
                if (int rc = kt_bulk_add_keys(s->table, msg + HDR_U64, s->cap_keys, msg)) return rc;
            } else {
                if (int rc = kt_ctr_add_keys_counted(s->table, msg + HDR_U64, s->cap_keys, msg)) return rc;
            }
        }
        return KT_OK;
    };
    // schedule: route(i+1) and level 1 of slice i-1 run on the main stream while slice i is on the wires
    if (int rc = route(0)) return rc;
    for (int i = 0; i < P; i++) {
        if (i + 1 < P)
            if (int rc = route(i + 1)) return rc;
        if (int rc = swap(i)) return rc;
        if (i >= 1)
            if (int rc = count(i - 1)) return rc;
    }
    if (int rc = count(P - 1)) return rc;
    if (bulk) {
        if (int rc = kt_bulk_finish(s->table)) return rc;
    }
    if (mem == KT_MEM_HOST) KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

int kt_sharded_finalize(kt_sharded *s) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_finalize: null");
    if (!s->routed) return KT_OK;
    kt_ctx *ctx = s->ctx;
    if (int rc = ctx->use()) return rc;
    const uint64_t words = HDR_U64 + FIN_CAP, all_words = words * (uint64_t)s->n_ranks;
    std::vector<uint64_t> hdr(all_words ? (size_t)s->n_ranks * HDR_U64 : 0);
    for (int round = 0; round < 1 << 20; round++) {
        uint64_t h[24] = {};
        KT_HIP(hipMemcpyAsync(h, s->pend_n, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        KT_HIP(hipStreamSynchronize(ctx->stream));
        if (reinterpret_cast<uint32_t *>(h + 16)[0])
            return kt::fail(KT_ERR_FULL, "sharded counter: the pending list overflowed (batch far too skewed for its owners)");
        const uint64_t n_pend = h[0] < s->pend_cap ? h[0] : s->pend_cap;
        for (int p = 0; p < s->n_ranks; p++)
            KT_HIP(hipMemsetAsync(s->fin_send + (uint64_t)p * words, 0, HDR_U64 * 8, ctx->stream));
        KT_HIP(hipMemsetAsync(s->fin_left, 0, 8, ctx->stream));
        if (n_pend) {
            hipLaunchKernelGGL(pack_pending_kernel, dim3(ktl::grid_for(ctx, (n_pend + BLOCK - 1) / BLOCK, 4)), dim3(BLOCK), 0,
                               ctx->stream, s->pend_keys, n_pend, (uint32_t)s->n_ranks, words, s->fin_send, s->fin_left);
        }
        hipLaunchKernelGGL(stamp_left_kernel, dim3(1), dim3(64), 0, ctx->stream, s->fin_send, (uint32_t)s->n_ranks, words,
                           s->fin_left);
        KT_HIP(hipGetLastError());
        KT_HIP(hipEventRecord(s->ev_routed[0], ctx->stream));
        KT_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_routed[0], 0));
        if (int rc = exchange(s, s->fin_send, s->fin_recv, words)) return rc;
        // the headers decide whether another round is needed: every rank sees every rank's remainder
        for (int p = 0; p < s->n_ranks; p++)
            KT_HIP(hipMemcpyAsync(hdr.data() + (size_t)p * HDR_U64, s->fin_recv + (uint64_t)p * words, HDR_U64 * 8,
                                  hipMemcpyDeviceToHost, s->comm_stream));
        KT_HIP(hipStreamSynchronize(s->comm_stream));
        bool more = false;
        for (int p = 0; p < s->n_ranks; p++) {
            const uint64_t n = hdr[(size_t)p * HDR_U64] < FIN_CAP ? hdr[(size_t)p * HDR_U64] : FIN_CAP;
            more |= hdr[(size_t)p * HDR_U64 + 1] != 0;
            if (n)
                if (int rc = kt_ctr_add_pairs(s->table, s->fin_recv + (uint64_t)p * words + HDR_U64, nullptr, n, KT_MEM_DEVICE))
                    return rc;
        }
        if (!more) {
            KT_HIP(hipMemsetAsync(s->pend_n, 0, 8, ctx->stream));  // everything pending has been delivered
            KT_HIP(hipStreamSynchronize(ctx->stream));             // (fin_recv is reused by the next finalize)
            return KT_OK;
        }
        KT_HIP(hipStreamSynchronize(ctx->stream));
    }
    return kt::fail(KT_ERR_HIP, "sharded counter: finalize did not converge");
}

}  // extern "C"
