// kt_shard.hip - canonical k-mer counting sharded over the GPUs of a node, behind the C ABI.
//
// Replaces the reference's `min_mer % n_parts` partitioning and per-partition merge (counter/src/lib.rs:100,127,
// 188-231): the partitions are GPUs.  The owner of a k-mer is a function of its MINIMISER (kt_superkmer.hpp: the
// canonical m-mer of the k-mer with the smallest hash; semantics of a minimiser: kmer/src/minimiser.rs:61-175), so the
// consecutive k-mers of a read fall into a few runs with one owner each and what crosses the links is the runs' BASES at
// 2 bits - records of at most 8 k-mers in 80 bits, ~1.7 bytes per k-mer at k = 31 where the k-mers themselves are 8 -
// and every rank's table is a whole table of its own:
//
//     main stream   route_kernel over the rank's own reads: per window start the owner (rolling canonical m-mer hashes,
//                   sliding minimum), the runs, the records - appended to one region of the send buffer per owner
//     host          the regions' fills come back (one read-back); the ranks tell each other, in one small exchange,
//                   whether they can go on and how many records each will get from each (so every message has its exact
//                   size and both ends know it)
//     comm stream   the regions leave in KT_SHARD_SLICES pieces: grouped ncclSend / ncclRecv with every peer - all seven
//                   xGMI links of a GPU busy at once - through librccl (loaded with dlopen; the copy a torch extension has
//                   loaded when there is one), or a caller-supplied host all-to-all (tests: gloo; MPI would fit too)
//     main stream   the ORDINARY single-GPU pipeline (kt_bulk.hip) with records as its source: level 1 over the rank's
//                   own region while the first pieces travel, then over every piece as it has arrived; level 2 and the
//                   range builds behind the last one.  No pre-split, no sliced level-1 outputs, no received key blocks.
//
// A region that overflows (a batch dominated by few k-mers sends most of its records to one owner) counts the k-mers
// of the records that do not fit in a small table of the sender's own (k-mer -> count: such batches are many copies
// of few k-mers), whose pairs kt_sharded_finalize delivers to their owners in fixed-size rounds through the probing path.
// Errors are agreed on: every fallible local step of a call comes before the small exchange, which carries each rank's
// status - a rank that cannot take part (a batch larger than agreed, a full pending table, a failed copy) still enters
// it, and then either all ranks move data or ALL return an error; nobody is left waiting for a peer that went home.
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
#include "kt_superkmer.hpp"
#include "kt_table.hpp"

namespace {

using ktseg::SegArgs;
using ktseg::SegShared;
using kttab::Slot;
using kttab::TableRef;
constexpr int BLOCK = ktseg::BLOCK;
constexpr int MAX_RANKS = (int)ktsk::MAX_OWNERS;
constexpr uint64_t HDR_U64 = 8;  // 64-byte header of a finalize message: [0] keys in it (may exceed the capacity: clamp),
                                 // [1] keys still pending at the sender, [2] the sender's status (0 = fine)
constexpr uint64_t FIN_CAP = 1u << 17;  // keys per peer and finalize round (1 MiB messages)
constexpr uint32_t GO_WORDS = 8;        // the words the ranks exchange before data moves (kt_sharded_add_reads)

// ---- librccl, resolved at run time -----------------------------------------------------------------------------
struct Rccl {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;  // (optional: statistics only)
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
            KT_SYM(CommCount, "ncclCommCount");
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

// ---- the route pass -----------------------------------------------------------------------------------------------
// the sender's small table for what does not fit a region (k-mer -> copies)
struct PendRef {
    Slot *slots;
    kttab::Geom g;
    uint32_t *flags;     // bit 0 = it is full
    uint64_t *distinct;
};

struct RouteArgs {
    SegArgs a;
    uint64_t seg_lo, seg_hi;
    uint32_t m, w, n_owners;
    uint64_t *regions;              // the send buffer: region o = regions + o * region_words
    uint64_t region_words;          // room / 1024 blocks of 1280 words
    uint64_t room;                  // records a region takes (a multiple of 1024)
    unsigned long long *cursors;    // [n_owners] records appended to every owner's stream so far (may run past the room)
    unsigned long long *kmers;      // [n_owners] k-mers in them
    PendRef pend;
};

struct RouteShared {
    SegShared seg2[2];         // (two: a slow wave still cuts records out of a segment's codes while the next is staged)
    uint32_t own4[8][BLOCK];   // the owners of the thread's 32 window starts, four to a word
    uint32_t ext[BLOCK];       // k-mers of the run that reaches the thread's last window start, inside the thread; bit 31: all 32
    uint32_t cnt[ktsk::MAX_OWNERS], km[ktsk::MAX_OWNERS];  // records / k-mers of the segment per owner
    uint32_t cur[ktsk::MAX_OWNERS];   // ... placed so far
    uint32_t base[ktsk::MAX_OWNERS];  // where they begin in the owner's stream (saturated: >= room = none fit)
    uint64_t *region[ktsk::MAX_OWNERS];           // the owners' regions
    uint16_t list[BLOCK / 64][1024];  // the wave's records (finder lane, window start, length), dealt out evenly to its lanes
};

__device__ __forceinline__ uint32_t low_bits(uint32_t n) { return n >= 32u ? 0xFFFFFFFFu : (1u << n) - 1u; }

// One workgroup per 8192-base segment of the rank's reads (grid-stride), a thread per 32 window starts - the front end of
// every other k-mer kernel (kt_segment.hpp), its global reads requested a segment ahead, the staged codes in one of two
// buffers - and then, every WAVE on its own (its 2048 window starts begin a run of their own: one cut per 2048 bases):
//   hashes   the canonical m-mers that start at the thread's 32 + w - 1 bases, from the two staged code words (32-bit
//            words, one funnel shift each), hashed; the minimum over every window of w of them in log2(w) in-place passes
//   owners   of the 32 window starts; a window start CONTINUES the run of the one before it when both are k-mers of one
//            owner (the lane before's last one by DPP)
//   records  start where a run starts and every 8 k-mers from there (the run's start may lie in an earlier lane: the
//            lanes' run lengths are looked back through LDS - one step unless a run spans whole lanes); a record's
//            length is what follows of its run, 8 at most, read off the lane's and the next lane's continuation bits
//   append   counts per owner (LDS); ONE workgroup barrier; one returning atomic per owner and SEGMENT on the owner's
//            cursor (per wave - four times the atomics on the same handful of addresses - the kernel ran three times as
//            long), its answer looked at behind the making of the wave's record list; one more barrier; then the wave's
//            records, listed in LDS by a scan of the lanes' counts, are dealt out evenly to its lanes (a lane finds 0 .. 32
//            records, 4 on average), cut out of the staged codes (a 192-bit funnel shift) and stored: a[pos], b[pos].
// A record beyond the region's room is not written: its k-mers are counted in the pending table.
#ifndef KT_ROUTE_ABL
#define KT_ROUTE_ABL 0  // (timing builds only, tools/build_variant_tu.sh: 1 = no records cut out or written, 2 = no m-mer hashes,
#endif                  //  4 = no staging of the bases: what each phase of route_kernel costs.  The shipped library: 0.)
#ifndef KT_ROUTE_WAVES
#define KT_ROUTE_WAVES 4  // waves per SIMD the kernel is compiled for (4: <= 128 registers)
#endif
__global__ __launch_bounds__(BLOCK, KT_ROUTE_WAVES) void route_kernel(RouteArgs ra) {
    __shared__ RouteShared sm;
    const uint32_t tid = threadIdx.x;
    const uint32_t k = ra.a.k, m = ra.m, w = ra.w, N = ra.n_owners;
    const uint32_t lane = tid & 63u, wv = tid >> 6;
    if (tid < ktsk::MAX_OWNERS) {
        sm.cnt[tid] = 0;
        sm.km[tid] = 0;
    }
    if (tid < N) sm.region[tid] = ra.regions + (uint64_t)tid * ra.region_words;
    const uint32_t room32 = (uint32_t)ra.room;  // (< 2^32 - 2^20: sharded_alloc)
    // the global reads of a segment - the thread's 32 bases, its first read start, the halo item and the segment's place in
    // the offsets - are requested a segment ahead (plain loads: nothing here needs them before they have long landed)
    const uint64_t g_first = ra.seg_lo + blockIdx.x;
    ktseg::SegAhead ah{};
    if (g_first < ra.seg_hi)
        ah = ktseg::request_ahead(ra.a, g_first, ktd::load_uniform(ra.a.seg_first + ktd::uniform64(g_first)), g_first + gridDim.x, tid);
    uint32_t par = 0;
    for (uint64_t g = g_first; g < ra.seg_hi; g += gridDim.x, par ^= 1u) {
        SegShared &seg = sm.seg2[par];
#if KT_ROUTE_ABL & 4
        seg.codes[tid] = g * 0x9E3779B97F4A7C15ull + tid;
        seg.inv[tid] = 0;
        seg.bnd[tid] = tid & 1 ? 0x10000u : 0u;
        if (tid == 0) { seg.codes[BLOCK] = g; seg.inv[BLOCK] = 0; seg.bnd[BLOCK] = 0; }
        ktd::lds_barrier();
#else
        {
            const uint64_t first_next = ah.first_next;
            ktseg::stage_ahead(ra.a, g, seg, tid, ah);  // (ends with a barrier)
            if (g + gridDim.x < ra.seg_hi) ah = ktseg::request_ahead(ra.a, g + gridDim.x, first_next, g + 2ull * gridDim.x, tid);
        }
#endif
        const ktseg::Window win(seg, tid, k);
        const uint32_t okm = win.okm;
        const uint64_t c0 = seg.codes[tid], c1 = seg.codes[tid + 1];
        uint32_t h[47];
        {
            const uint32_t x[5] = {(uint32_t)(c0 >> 32), (uint32_t)c0, (uint32_t)(c1 >> 32), (uint32_t)c1, 0u};
            const uint32_t msh = 32u - 2u * m, rsh = 2u * (m - 1u);
            uint32_t r = 0;
#pragma unroll
            for (int i = 0; i < 47; i++) {
                const int j = i >> 4, s = 2 * (i & 15);
                const uint32_t top = s ? __builtin_amdgcn_alignbit(x[j], x[j + 1], 32 - s) : x[j];  // bases i .. i + 15
                const uint32_t f = top >> msh;
#if KT_ROUTE_ABL & 2
                h[i] = f + r;
#else
                r = i ? (r >> 2) | ((3u - (f & 3u)) << rsh) : ktsk::rev_comp32(f, m);
                h[i] = ktsk::mhash(f < r ? f : r);
#endif
            }
        }
#pragma unroll
        for (int st = 1; st < 16; st <<= 1) {
            if (w > (uint32_t)st) {
#pragma unroll
                for (int i = 0; i + st < 47; i++) h[i] = h[i] < h[i + st] ? h[i] : h[i + st];
            }
        }
        // owners; eqm bit i: window starts i - 1 and i have the same one
        uint32_t eqm = 0, o_first = 0, o_last = 0;
        {
            uint32_t prev_o = 0, word = 0;
#pragma unroll
            for (int i = 0; i < 32; i++) {
                const uint32_t o = ktsk::owner_of_min(h[i], N);
                if (i == 0) o_first = o;
                else eqm |= (o == prev_o ? 1u : 0u) << i;
                prev_o = o;
                word |= o << (8 * (i & 3));
                if ((i & 3) == 3) {
                    sm.own4[i >> 2][tid] = word;
                    word = 0;
                }
            }
            o_last = prev_o;
        }
        // (from here on a wave works alone: the neighbours it talks to are its own lanes - a wave's 2048 window starts begin a
        // run of their own, one cut more per 2048 bases - and what goes through LDS is ordered by the wave itself)
        {
            const uint32_t prev = ktd::wave_shr1((okm >> 31) ? o_last : 0xFFu, 0xFFu);  // the owner of the lane before's last start
            eqm |= prev == o_first ? 1u : 0u;  // (0xFF is no owner: the start before was no k-mer, or this is the wave's first)
        }
        const uint32_t cont = okm & ((okm << 1) | 1u) & eqm;
        const uint32_t rs = okm & ~cont;  // run starts
        sm.ext[tid] = cont == 0xFFFFFFFFu ? (0x80000000u | 32u) : (uint32_t)__builtin_clz(~cont) + 1u;
        // (the lane behind's word, 0 for the wave's last lane - DPP wave_shl:1; not a shuffle under `lane == 63 ? 0 : ...`: only
        // the chosen arm of ?: is evaluated, and a lane that sits the shuffle out is not read by its neighbour)
        const uint32_t cont_next = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cont, 0x130, 0xf, 0xf, false);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        // which window starts begin a record - and, run by run, what the thread adds to its owners' streams
        auto owner_at = [&](uint32_t i) { return (sm.own4[i >> 2][tid] >> (8u * (i & 3u))) & 0xFFu; };
        uint32_t recmask = 0;
        if (cont & 1u) {  // the thread's first window starts carry on a run of the lanes before (lane 0's never do)
            uint32_t carry = 0;
            for (uint32_t j = tid - 1;; j--) {
                const uint32_t e = sm.ext[j];
                carry += e & 0xFFFFu;
                if (!(e >> 31) || (j & 63u) == 0) break;
            }
            const uint32_t l0 = cont == 0xFFFFFFFFu ? 32u : (uint32_t)__builtin_ctz(~cont);
            const uint32_t first = (8u - (carry & 7u)) & 7u;
            recmask = (0x01010101u << first) & low_bits(l0);
            if (recmask) atomicAdd(&sm.cnt[o_first], (uint32_t)__builtin_popcount(recmask));
            atomicAdd(&sm.km[o_first], l0);
        }
        for (uint32_t rem = rs; rem; rem &= rem - 1u) {
            const uint32_t s = (uint32_t)__builtin_ctz(rem);
            const uint32_t x = s < 31u ? cont >> (s + 1u) : 0u;
            const uint32_t e = 1u + (uint32_t)__builtin_ctz(~x);  // the run's window starts inside this thread
            const uint32_t rm = (0x01010101u << s) & (low_bits(e) << s);
            recmask |= rm;
            const uint32_t o = owner_at(s);
            atomicAdd(&sm.cnt[o], (uint32_t)__builtin_popcount(rm));
            atomicAdd(&sm.km[o], e);
        }
        const uint64_t cont64 = (uint64_t)cont | ((uint64_t)cont_next << 32);
        auto len_at = [&](uint32_t i) {
            const uint64_t follow = ~(cont64 >> (i + 1u));  // (bits 63 - i .. 63 of the shifted word are 0: a stop bit)
            const uint32_t more = (uint32_t)__builtin_ctzll(follow);
            return 1u + (more < 7u ? more : 7u);
        };
        ktd::lds_barrier();
        // one returning atomic per owner and segment - the owners' cursors are a handful of addresses, and an atomic on one
        // address takes its turn behind every other: per wave (four times as many, a quarter of the waves waiting for each)
        // the kernel ran three times as long.  Its answer is looked at behind the list's making.
        unsigned long long got = 0;
        if (tid < N) {
            const uint32_t c = sm.cnt[tid], n = sm.km[tid];
            got = c ? atomicAdd(&ra.cursors[tid], (unsigned long long)c) : 0ull;
            if (n) atomicAdd(&ra.kmers[tid], (unsigned long long)n);
            sm.cnt[tid] = 0;
            sm.km[tid] = 0;
            sm.cur[tid] = 0;  // (every wave is past the segment before: the barrier above)
        }
        // The records, dealt out evenly: a thread holds 0 .. 32 of them (4 on average, the fullest lane of a wave 9), so the
        // lanes that cut them out are not the lanes that found them - every wave lists the records of its window starts
        // 0 .. 15, then 16 .. 31 (at most 1024 each) in LDS, places by a scan of the lanes' counts, and lane l takes entries
        // l, l + 64, ...: the owner and the length ride in the entry, the bases come from the staged codes of the finder.
        uint16_t *const list = sm.list[wv];
        // (mask: which of the thread's record starts go into this list - all of them, unless the wave has more than 1024)
        auto make_list = [&](uint32_t mask) -> uint32_t {
            uint32_t mh = recmask & mask;
            const uint32_t n_t = (uint32_t)__builtin_popcount(mh);
            const uint32_t inc = ktd::wave_incl_scan(n_t);
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);  // (the same in every lane)
            for (uint32_t at = inc - n_t; mh; mh &= mh - 1u, at++) {
                const uint32_t i = (uint32_t)__builtin_ctz(mh);
                list[at] = (uint16_t)((lane << 9) | (i << 4) | len_at(i));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
            return total;
        };
        auto cut_out = [&](uint32_t total) {
            for (uint32_t r = lane; r < total; r += 64u) {
                const uint32_t d = list[r];
                const uint32_t len = d & 15u, i = (d >> 4) & 31u, src = (wv << 6) | (d >> 9);
                const uint32_t o = (sm.own4[i >> 2][src] >> (8u * (i & 3u))) & 0xFFu;
                const uint64_t s0 = seg.codes[src], s1 = seg.codes[src + 1];
                const uint64_t s2 = seg.codes[src + 2];  // (index <= 257: allocated; only records that reach it read into it)
                uint64_t A = s0, B = s1;
                if (i) {
                    A = (s0 << (2u * i)) | (s1 >> (64u - 2u * i));
                    B = (s1 << (2u * i)) | (s2 >> (64u - 2u * i));
                }
                const uint32_t nb = 2u * (len + k - 1u);  // <= 76 bits of bases
                if (nb <= 64u) {
                    A &= nb == 64u ? ~0ull : ~0ull << (64u - nb);
                    B = 0;
                } else {
                    B &= ~0ull << (128u - nb);
                }
                const uint32_t pos = sm.base[o] + atomicAdd(&sm.cur[o], 1u);  // (a segment has < 2^14 records: no wrap)
                if (pos < room32) {
                    const uint32_t idx = pos & (ktsk::BLOCK_RECS - 1u);
                    uint64_t *const blk = sm.region[o] + ktd::mad64(pos >> 10, ktsk::BLOCK_WORDS, 0u);
                    blk[idx] = A;
                    reinterpret_cast<uint16_t *>(blk + ktsk::BLOCK_RECS)[idx] = (uint16_t)(((uint32_t)(B >> 52) << 4) | len);
                } else {
                    // no room in the owner's region (a batch dominated by few k-mers): counted aside, delivered by finalize
                    for (uint32_t j = 0; j < len; j++) {
                        const uint64_t top = j ? (A << (2u * j)) | (B >> (64u - 2u * j)) : A;
                        const uint64_t f = top >> (64u - 2u * k), rc = ktd::rev_comp(f, (int)k);
                        const uint32_t st = kttab::table_add(TableRef{ra.pend.slots, ra.pend.g, ra.pend.flags}, f < rc ? f : rc, 1u);
                        if (st == 0u) atomicOr(ra.pend.flags, 1u);
                        else if (st == 2u) atomicAdd(reinterpret_cast<unsigned long long *>(ra.pend.distinct), 1ull);
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();  // (the list is rewritten by the next half)
        };
#if KT_ROUTE_ABL & 1
        if (tid < N) sm.base[tid] = (uint32_t)got;
        ktd::lds_barrier();
        if (recmask == 0x12345u && cont64 == 77u) list[0] = (uint16_t)(len_at(3) + owner_at(2));
#else
        // (a wave has ~280 records; more than the list holds - every window start a run of its own: k <= 8 - go in two halves)
        const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)ktd::wave_incl_scan((uint32_t)__builtin_popcount(recmask)), 63);
        const bool halves = all > 1024u;
        const uint32_t total0 = make_list(halves ? 0x0000FFFFu : 0xFFFFFFFFu);
        if (tid < N) sm.base[tid] = got < (unsigned long long)room32 ? (uint32_t)got : room32;
        ktd::lds_barrier();
        cut_out(total0);
        if (halves) cut_out(make_list(0xFFFF0000u));
#endif
        // (no barrier here: the next segment is staged into the other code buffer, its counts go to cnt[] / km[], and what the
        // cutting-out reads - base[], cur[], region[], the wave's list - is next written behind two of its barriers)
    }
}

// finalize round: up to FIN_CAP pending (k-mer, count) pairs per owner into the round's messages - a message is its
// header, FIN_CAP keys, FIN_CAP counts; the rest stays (sent ones become KT_EMPTY_KEY: the arrays are rescanned next
// round).  left[0] = pairs still pending after this round.
__global__ __launch_bounds__(BLOCK) void pack_pending_kernel(uint64_t *__restrict__ pend_keys,
                                                             const uint32_t *__restrict__ pend_counts, uint64_t n,
                                                             uint32_t n_owners, uint32_t k, uint64_t msg_stride,
                                                             uint64_t *__restrict__ msgs, uint64_t *__restrict__ left) {
    uint64_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t key = pend_keys[i];
        if (key == KT_EMPTY_KEY) continue;
        const uint32_t o = ktsk::owner_of_kmer(key, k, n_owners);
        const uint64_t pos = atomicAdd(reinterpret_cast<unsigned long long *>(msgs + o * msg_stride), 1ull);
        if (pos < FIN_CAP) {
            msgs[o * msg_stride + HDR_U64 + pos] = key;
            reinterpret_cast<uint32_t *>(msgs + o * msg_stride + HDR_U64 + FIN_CAP)[pos] = pend_counts[i];
            pend_keys[i] = KT_EMPTY_KEY;
        } else {
            mine++;
        }
    }
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(reinterpret_cast<unsigned long long *>(left), (unsigned long long)mine);
}

__global__ void stamp_left_kernel(uint64_t *msgs, uint32_t n_owners, uint64_t msg_stride, const uint64_t *left,
                                  uint64_t status) {
    if (threadIdx.x < n_owners) {
        msgs[threadIdx.x * msg_stride + 1] = *left;
        msgs[threadIdx.x * msg_stride + 2] = status;  // this rank cannot go on (finalize: every rank learns it)
    }
}

}  // namespace

struct kt_sharded {
    kt_ctx *ctx = nullptr;
    kt_ctr *table = nullptr;
    int k = 0, n_ranks = 1, rank = 0, n_slices = 4;
    // the owners the route pass sorts into: the ranks - or, a single rank made to take the routed path (KT_SHARD_FORCE =
    // n: tests, and the one-GPU measurement of what a rank of n does), n regions that all stay here
    int n_owners = 1;
    bool routed = false;  // false: a single rank, everything goes straight to the table
    uint32_t m = 0, w = 0;
    uint64_t max_batch_bases = 0;
    uint64_t room = 0;          // records per region (a multiple of 1024)
    uint64_t region_words = 0;  // u64 words per region
    uint64_t *send = nullptr;   // n_owners regions: what the route pass writes; region `rank` is read where it lies
    uint64_t *recv = nullptr;   // n_ranks regions: what rank p sent lies in region p (region `rank` unused)
    unsigned long long *cursors = nullptr;  // device: [0, 64) records per owner, [64, 128) k-mers per owner
    uint64_t *go_send = nullptr, *go_recv = nullptr;  // n_ranks x GO_WORDS: the words exchanged before data moves
    uint64_t *go_cannot = nullptr;  // GO_WORDS device words written at creation: "cannot go on" (a rank whose copies fail)
    uint64_t *fin_send = nullptr, *fin_recv = nullptr;  // n_ranks messages of FIN_CAP keys each
    kt_ctr *pend = nullptr;  // what did not fit the regions, counted: k-mer -> copies (delivered by finalize)
    uint64_t *pend_keys = nullptr, *fin_left = nullptr;  // finalize: the pending table's pairs, exported
    uint32_t *pend_counts = nullptr;
    uint64_t pend_cap = 0;
    bool roomy = false;  // jobs are planned with a quarter more room than the announced k-mers (see kt_sharded_add_reads)
    bool pend_touched = false;  // a route pass has run since the pending table was last known to be empty
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_main = nullptr;
    std::vector<hipEvent_t> ev_recv;
    // transport
    Rccl *rccl = nullptr;
    ncclComm_t comm = nullptr;
    kt_alltoall_fn fn = nullptr;
    void *fn_user = nullptr;
    void *h_send = nullptr, *h_recv = nullptr;  // pinned staging for the host transport
    size_t h_bytes = 0;
    uint64_t exchanged_bytes = 0;  // sent to other ranks so far (statistics; a forced single rank: what would have left)
    uint64_t add_calls = 0;        // kt_sharded_add_reads calls so far (tests: KT_SHARD_FAIL_LOCAL)
    std::vector<uint64_t> last_records, last_kmers;  // what the last batch's route pass put into every owner's region
    uint64_t *region(uint64_t *buf, int o) const { return buf + (uint64_t)o * region_words; }
};

namespace {

// One piece per peer each way: piece p of the send list goes to rank p, piece p of the receive list comes from rank p
// (this rank's own entries are ignored: what a rank keeps for itself it reads where it lies).  Pieces may differ in
// size per peer - both sides know them.  Enqueued on the comm stream (RCCL) or carried out before returning (host
// transport, which moves equal blocks: the largest piece, the shorter ones padded).
struct Piece {
    const void *src;
    size_t src_bytes;
    void *dst;
    size_t dst_bytes;
};

int host_staging(kt_sharded *s, size_t all) {
    if (s->h_bytes >= all) return KT_OK;
    if (s->h_send) (void)hipHostFree(s->h_send);
    if (s->h_recv) (void)hipHostFree(s->h_recv);
    s->h_send = s->h_recv = nullptr;
    s->h_bytes = 0;
    KT_HIP(hipHostMalloc(&s->h_send, all, hipHostMallocDefault));
    KT_HIP(hipHostMalloc(&s->h_recv, all, hipHostMallocDefault));
    s->h_bytes = all;
    return KT_OK;
}

// block_all: the size of the host transport's equal blocks when the pieces are not the same on every rank (the regions:
// every pair of ranks has its own count) - a figure every rank computes alike; 0: the largest piece of this rank, which
// is every rank's when the pieces are symmetric
int exchange_v(kt_sharded *s, const std::vector<Piece> &pc, size_t block_all = 0) {
    if (!s->fn && !s->comm) {
        if (s->n_ranks == 1) return KT_OK;  // (a single rank that was never connected: nothing to move)
        return kt::fail(KT_ERR_ARG, "sharded counter: no transport (kt_sharded_connect_rccl / kt_sharded_connect_host first)");
    }
    if (s->fn) {
        size_t block = block_all;
        for (int p = 0; p < s->n_ranks; p++) {
            if (p == s->rank) continue;
            if (pc[p].src_bytes > block) block = pc[p].src_bytes;
            if (pc[p].dst_bytes > block) block = pc[p].dst_bytes;
        }
        block = (block + 63) & ~(size_t)63;
        if (block == 0) return KT_OK;
        const size_t all = block * (size_t)s->n_ranks;
        if (int rc = host_staging(s, all)) return rc;
        for (int p = 0; p < s->n_ranks; p++)
            if (p != s->rank && pc[p].src_bytes)
                KT_HIP(hipMemcpyAsync((char *)s->h_send + (size_t)p * block, pc[p].src, pc[p].src_bytes, hipMemcpyDeviceToHost,
                                      s->comm_stream));
        KT_HIP(hipStreamSynchronize(s->comm_stream));
        // (a transport that fails must not leave the previous exchange's words to be taken for this one's)
        for (int p = 0; p < s->n_ranks; p++) memset((char *)s->h_recv + (size_t)p * block, 0xFF, block < 64 ? block : 64);
        if (s->fn(s->fn_user, s->h_send, s->h_recv, (uint64_t)block) != 0)
            return kt::fail(KT_ERR_HIP, "sharded counter: the caller's all-to-all failed");
        for (int p = 0; p < s->n_ranks; p++)
            if (p != s->rank && pc[p].dst_bytes)
                KT_HIP(hipMemcpyAsync(pc[p].dst, (char *)s->h_recv + (size_t)p * block, pc[p].dst_bytes, hipMemcpyHostToDevice,
                                      s->comm_stream));
        KT_HIP(hipStreamSynchronize(s->comm_stream));  // the staging buffers are reused by the next exchange
    } else {
        KT_NCCL(s->rccl, s->rccl->GroupStart());
        // (a call that fails inside the group must not leave it open: the group is closed, then the first failure is reported)
        ncclResult_t first = ncclSuccess;
        for (int p = 0; p < s->n_ranks && first == ncclSuccess; p++) {
            if (p == s->rank) continue;
            if (pc[p].src_bytes) first = s->rccl->Send(pc[p].src, pc[p].src_bytes, ncclUint8, p, s->comm, s->comm_stream);
            if (first == ncclSuccess && pc[p].dst_bytes)
                first = s->rccl->Recv(pc[p].dst, pc[p].dst_bytes, ncclUint8, p, s->comm, s->comm_stream);
        }
        const ncclResult_t closed = s->rccl->GroupEnd();
        if (first != ncclSuccess) KT_NCCL(s->rccl, first);
        KT_NCCL(s->rccl, closed);
    }
    for (int p = 0; p < s->n_ranks; p++)
        if (p != s->rank) s->exchanged_bytes += pc[p].src_bytes;
    return KT_OK;
}

// n_ranks equal messages, rank's own included (finalize rounds): message p of `src` to rank p, message p of `dst` from rank p
int exchange(kt_sharded *s, const uint64_t *src, uint64_t *dst, uint64_t words) {
    std::vector<Piece> pc((size_t)s->n_ranks);
    for (int p = 0; p < s->n_ranks; p++)
        pc[p] = Piece{src + (uint64_t)p * words, words * 8, dst + (uint64_t)p * words, words * 8};
    if (int rc = exchange_v(s, pc)) return rc;
    KT_HIP(hipMemcpyAsync(dst + (uint64_t)s->rank * words, src + (uint64_t)s->rank * words, words * 8, hipMemcpyDeviceToDevice,
                          s->comm_stream));
    return KT_OK;
}

// records of room per owner's region for batches of at most max_batch_bases: twice what uniform reads put there - a run
// of one owner is (w + 1) / 2 * n / (n - 1) k-mers on average and makes one record per 8 of them, rounded up - and never
// less than 64 blocks; a multiple of 1024
uint64_t region_room(uint64_t max_batch_bases, uint32_t w, int n_owners) {
    const double per_kmer = 2.0 / (double)(w + 1) * (n_owners > 1 ? (double)(n_owners - 1) / n_owners : 0.0) + 1.0 / 8.0 + 1.0 / 64.0;
    double recs = 2.0 * per_kmer * (double)max_batch_bases / (double)n_owners;
    if (recs > (double)max_batch_bases) recs = (double)max_batch_bases;  // (a record holds a k-mer at least)
    uint64_t r = (uint64_t)recs + 64 * ktsk::BLOCK_RECS;
    return (r + ktsk::BLOCK_RECS - 1) / ktsk::BLOCK_RECS * ktsk::BLOCK_RECS;
}

int sharded_alloc(kt_sharded *s) {
    kt_ctx *ctx = s->ctx;
    if (int rc = ctx->use()) return rc;
    const uint64_t fin_u64 = HDR_U64 + FIN_CAP + FIN_CAP / 2;
    // the pending table: DISTINCT k-mers that overflow their regions (a batch may be one k-mer a billion times over - one
    // entry); a batch whose overflow has more distinct k-mers than this is not skewed, it is larger than agreed
    uint64_t pend_slots = s->max_batch_bases / 64;
    if (pend_slots < (1u << 20)) pend_slots = 1u << 20;
    if (int rc = kt_ctr_create(ctx, s->k, pend_slots, &s->pend)) return rc;
    if (int rc = kt_ctr_capacity(s->pend, &s->pend_cap)) return rc;
    s->room = region_room(s->max_batch_bases, s->w, s->n_owners);
    if (const char *e = getenv("KT_SHARD_ROOM_BLOCKS"))  // tests: regions of that many blocks (a flood that overflows them)
        if (atoll(e) > 0) s->room = (uint64_t)atoll(e) * ktsk::BLOCK_RECS;
    if (s->room >= 0xFFF00000ull)  // (the route pass keeps a record's place in its owner's stream in 32 bits)
        return kt::fail(KT_ERR_ARG, "kt_sharded_create: max_batch_bases too large for the exchange regions");
    s->region_words = s->room / ktsk::BLOCK_RECS * ktsk::BLOCK_WORDS;
    hipError_t e = hipMalloc((void **)&s->send, s->region_words * 8 * (size_t)s->n_owners);
    if (e == hipSuccess && s->n_ranks > 1) e = hipMalloc((void **)&s->recv, s->region_words * 8 * (size_t)s->n_ranks);
    if (e == hipSuccess) e = hipMalloc((void **)&s->cursors, 2 * ktsk::MAX_OWNERS * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->go_send, (size_t)s->n_ranks * GO_WORDS * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->go_recv, (size_t)s->n_ranks * GO_WORDS * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->go_cannot, GO_WORDS * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->fin_send, fin_u64 * 8 * s->n_ranks);
    if (e == hipSuccess) e = hipMalloc((void **)&s->fin_recv, fin_u64 * 8 * s->n_ranks);
    if (e == hipSuccess) e = hipMalloc((void **)&s->pend_keys, s->pend_cap * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->pend_counts, s->pend_cap * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&s->fin_left, 256);
    if (e == hipSuccess) {
        uint64_t wd[GO_WORDS] = {};
        wd[0] = 1;  // status: cannot go on
        e = hipMemcpy(s->go_cannot, wd, sizeof wd, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) return kt::fail(KT_ERR_NOMEM, std::string("sharded counter: hipMalloc: ") + hipGetErrorString(e));
    {   // the exchange should start the moment its piece is ready, whatever the main stream is running
        int lo = 0, hi = 0;
        KT_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        KT_HIP(hipStreamCreateWithPriority(&s->comm_stream, hipStreamNonBlocking, hi));
    }
    KT_HIP(hipEventCreateWithFlags(&s->ev_main, hipEventDisableTiming));
    s->ev_recv.resize((size_t)s->n_slices);
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
    if (k < 1 || k > 31) return kt::fail(KT_ERR_ARG, "kt_sharded_create: k must be in 1..31");
    kt_sharded *s = new (std::nothrow) kt_sharded();
    if (!s) return kt::fail(KT_ERR_NOMEM, "kt_sharded_create: host alloc");
    s->ctx = ctx;
    s->k = k;
    s->w = ktsk::window_of((uint32_t)k);
    s->m = ktsk::mmer_of((uint32_t)k);
    s->n_ranks = n_ranks;
    s->rank = rank;
    s->max_batch_bases = max_batch_bases;
    const char *env = getenv("KT_SHARD_SLICES");
    s->n_slices = env && atoi(env) > 0 ? atoi(env) : 4;
    if (s->n_slices > 64) s->n_slices = 64;
    const char *force = getenv("KT_SHARD_FORCE");  // tests: the routed path with a single rank too, into n owners' regions
    const int forced = force ? atoi(force) : 0;
    s->routed = n_ranks > 1 || forced > 0;
    s->n_owners = n_ranks > 1 ? n_ranks : forced > 1 ? (forced > MAX_RANKS ? MAX_RANKS : forced) : 1;
    int rc = kt_ctr_create(ctx, k, capacity_slots, &s->table);
    if (rc == KT_OK) {
        s->table->n_owners = (uint32_t)n_ranks;  // (kt_cov_batch_part: the k-mers of the other ranks are not absent, they are elsewhere)
        s->table->owner = (uint32_t)rank;
    }
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

int kt_sharded_create_local(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                            kt_sharded **out) {
    return sharded_new(ctx, k, capacity_slots, max_batch_bases, n_ranks, rank, out);
}

int kt_sharded_connect_rccl(kt_sharded *s, const uint8_t *id128) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_connect_rccl: null");
    if (s->n_ranks > 1 && !id128) return kt::fail(KT_ERR_ARG, "kt_sharded_connect_rccl: null id");
    if (!s->routed || s->comm || s->fn) return KT_OK;
    if (int rc = s->ctx->use()) return rc;
    if (int rc = load_rccl(&s->rccl)) return rc;
    ncclUniqueId id;
    ncclResult_t r = ncclSuccess;
    if (id128) memcpy(&id, id128, 128);
    else r = s->rccl->GetUniqueId(&id);  // (a single rank made to take the routed path: tests)
    if (r == ncclSuccess) r = s->rccl->CommInitRank(&s->comm, s->n_ranks, id, s->rank);
    if (r != ncclSuccess) return kt::fail(KT_ERR_HIP, std::string("ncclCommInitRank: ") + s->rccl->GetErrorString(r));
    if (s->n_ranks > 1) {
        // The pieces travel (RCCL's send / receive kernels, comm stream) while level 1 counts the pieces before them - and a
        // level-1 workgroup holds a whole CU's LDS for the length of its launch: on a chip that level 1 fills, the exchange's
        // kernels would wait for a launch to end.  Level 1 leaves a few CUs free (KT_SHARD_COMM_CUS, default 16 of 256: +6 %
        // of level 1's 11.5 ms, against an exchange that would otherwise not overlap at all).  NOT measured: no multi-GPU box.
        const char *e = getenv("KT_SHARD_COMM_CUS");
        s->table->l1_spare_cus = e ? (uint32_t)atoi(e) : 16u;
    }
    return KT_OK;
}

int kt_sharded_connect_host(kt_sharded *s, kt_alltoall_fn fn, void *user) {
    if (!s || !fn) return kt::fail(KT_ERR_ARG, "kt_sharded_connect_host: null");
    s->fn = fn;
    s->fn_user = user;
    return KT_OK;
}

int kt_sharded_create_rccl(kt_ctx *ctx, int k, uint64_t capacity_slots, uint64_t max_batch_bases, int n_ranks, int rank,
                           const uint8_t *id128, kt_sharded **out) {
    if (n_ranks > 1 && !id128) return kt::fail(KT_ERR_ARG, "kt_sharded_create_rccl: null id");
    kt_sharded *s = nullptr;
    if (int rc = sharded_new(ctx, k, capacity_slots, max_batch_bases, n_ranks, rank, &s)) return rc;
    if (int rc = kt_sharded_connect_rccl(s, id128)) {
        kt_sharded_destroy(s);
        return rc;
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
    if (s->ev_main) (void)hipEventDestroy(s->ev_main);
    for (auto ev : s->ev_recv)
        if (ev) (void)hipEventDestroy(ev);
    if (s->send) (void)hipFree(s->send);
    if (s->recv) (void)hipFree(s->recv);
    if (s->cursors) (void)hipFree(s->cursors);
    if (s->go_send) (void)hipFree(s->go_send);
    if (s->go_recv) (void)hipFree(s->go_recv);
    if (s->go_cannot) (void)hipFree(s->go_cannot);
    if (s->fin_send) (void)hipFree(s->fin_send);
    if (s->fin_recv) (void)hipFree(s->fin_recv);
    if (s->pend_keys) (void)hipFree(s->pend_keys);
    if (s->pend_counts) (void)hipFree(s->pend_counts);
    if (s->fin_left) (void)hipFree(s->fin_left);
    if (s->pend) kt_ctr_destroy(s->pend);
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
    if (s->routed && s->pend_touched) {
        // (the pending table is GBs of slots that nearly always hold nothing: it is cleared - 2 ms at BASELINE sizes - only when
        // a route pass really put something there; finding out is one 8-byte read)
        uint64_t n = 0;
        const int rc = kt_ctr_size(s->pend, &n);
        if (rc != KT_OK || n) {
            kt::set_error("");
            if (int rc2 = kt_ctr_clear(s->pend)) return rc2;
        }
        s->pend_touched = false;
    }
    return kt_ctr_clear(s->table);
}

int kt_sharded_exchanged_bytes(kt_sharded *s, uint64_t *bytes) {
    if (!s || !bytes) return kt::fail(KT_ERR_ARG, "kt_sharded_exchanged_bytes: null");
    *bytes = s->exchanged_bytes;
    return KT_OK;
}

int kt_sharded_comm_info(kt_sharded *s, int *n_ranks, int *rccl_ranks, int *transport) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_comm_info: null");
    if (n_ranks) *n_ranks = s->n_ranks;
    if (transport) *transport = s->fn ? 2 : s->comm ? 1 : 0;
    if (rccl_ranks) {
        *rccl_ranks = 0;
        if (s->comm && s->rccl && s->rccl->CommCount) {
            int n = 0;
            KT_NCCL(s->rccl, s->rccl->CommCount(s->comm, &n));
            *rccl_ranks = n;
        }
    }
    return KT_OK;
}

int kt_sharded_route_stats(kt_sharded *s, uint32_t *n_owners, uint64_t *records, uint64_t *kmers) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_route_stats: null");
    if (n_owners) *n_owners = (uint32_t)s->n_owners;
    for (size_t o = 0; o < (size_t)s->n_owners; o++) {
        if (records) records[o] = o < s->last_records.size() ? s->last_records[o] : 0;
        if (kmers) kmers[o] = o < s->last_kmers.size() ? s->last_kmers[o] : 0;
    }
    return KT_OK;
}

int kt_shard_minimiser(int k, uint32_t *m, uint32_t *w) {
    if (k < 1 || k > 31) return kt::fail(KT_ERR_ARG, "kt_shard_minimiser: k must be in 1..31");
    if (m) *m = ktsk::mmer_of((uint32_t)k);
    if (w) *w = ktsk::window_of((uint32_t)k);
    return KT_OK;
}

uint32_t kt_shard_owner_of(uint64_t kmer, int k, uint32_t n_ranks) {
    if (k < 1 || k > 31 || n_ranks <= 1) return 0u;
    return ktsk::owner_of_kmer(kmer, (uint32_t)k, n_ranks > (uint32_t)MAX_RANKS ? (uint32_t)MAX_RANKS : n_ranks);
}

int kt_sharded_owner_of(kt_sharded *s, uint64_t kmer, uint32_t *owner) {
    if (!s || !owner) return kt::fail(KT_ERR_ARG, "kt_sharded_owner_of: null");
    *owner = s->n_ranks > 1 ? ktsk::owner_of_kmer(kmer, (uint32_t)s->k, (uint32_t)s->n_ranks) : 0u;
    return KT_OK;
}

int kt_sharded_add_reads(kt_sharded *s, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int mem) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: null");
    if (!s->routed) return kt_ctr_add_reads(s->table, bases, offsets, n_reads, mem);
    kt_ctx *ctx = s->ctx;
    if (int rc = ctx->use()) return rc;
    // What this rank finds wrong - with its arguments, or while it sets the batch up: a pending table that filled up in an
    // earlier batch, a failed copy - is not returned at once: a rank that left now would leave its peers waiting in the
    // exchange.  Every fallible local step comes first (the route pass among them); then the ranks tell each other in
    // one small exchange whether they can go on, and either all of them move data or none does.
    std::string my_error;
    int my_code = KT_OK;
    uint64_t total = 0;
    auto local_fail = [&](int rc) {
        if (my_code == KT_OK) {
            my_code = rc;
            my_error = kt_last_error();
        }
        total = 0;
    };
    auto arg_fail = [&](const char *what) {
        if (my_code == KT_OK) {
            my_code = KT_ERR_ARG;
            my_error = what;
        }
        total = 0;
    };
    auto hip_fail = [&](const char *what, hipError_t e) {
        kt::set_error(std::string("kt_sharded_add_reads: ") + what + ": " + hipGetErrorString(e));
        local_fail(KT_ERR_HIP);
    };
    if (mem != KT_MEM_HOST && mem != KT_MEM_DEVICE) arg_fail("kt_sharded_add_reads: bad mem flag");
    if (my_code == KT_OK && n_reads) {
        if (!offsets) arg_fail("kt_sharded_add_reads: null offsets");
        else if (int rc = ktl::total_bases_of(ctx, offsets, n_reads, mem, &total)) local_fail(rc);
    }
    if (my_code == KT_OK && total > s->max_batch_bases) arg_fail("kt_sharded_add_reads: batch larger than max_batch_bases (split it)");
    if (my_code == KT_OK && total && !bases) arg_fail("kt_sharded_add_reads: null bases");
    {
        // tests: KT_SHARD_FAIL_LOCAL=<rank>:<call> - that rank's call number <call> (0-based, per counter) fails locally here
        const char *inj = getenv("KT_SHARD_FAIL_LOCAL");
        int r = -1, c = -1;
        if (inj && sscanf(inj, "%d:%d", &r, &c) == 2 && r == s->rank && (uint64_t)c == s->add_calls) {
            kt::set_error("kt_sharded_add_reads: injected local failure (KT_SHARD_FAIL_LOCAL)");
            local_fail(KT_ERR_FULL);
        }
        s->add_calls++;
    }
    const uint8_t *d_bases = bases;
    const uint64_t *d_offsets = offsets;
    if (my_code == KT_OK && mem == KT_MEM_HOST && total) {
        if (int rc = ktl::stage_batch(ctx, bases, offsets, n_reads, &d_bases, &d_offsets)) local_fail(rc);
    }
    SegArgs a{};
    if (my_code == KT_OK && total) {
        if (int rc = ktl::make_seg_args(ctx, d_bases, d_offsets, n_reads, total, s->k, &a)) local_fail(rc);
    }
    const int P = s->n_slices, N = s->n_ranks, V = s->n_owners, me = s->rank;
    // ---- the route pass over this rank's reads: one region of records per owner
    std::vector<unsigned long long> h_cur(2 * ktsk::MAX_OWNERS, 0ull);
    if (my_code == KT_OK && total) {
        if (int rc = ktl::table_ready(s->pend)) local_fail(rc);  // (a deferred clear happens now; a full table is reported)
    }
    if (my_code == KT_OK && total) {
        s->pend->empty = false;
        s->pend_touched = true;
        RouteArgs ra{};
        ra.a = a;
        ra.seg_lo = 0;
        ra.seg_hi = a.n_seg;
        ra.m = s->m;
        ra.w = s->w;
        ra.n_owners = (uint32_t)V;
        ra.regions = s->send;
        ra.region_words = s->region_words;
        ra.room = s->room;
        ra.cursors = s->cursors;
        ra.kmers = s->cursors + ktsk::MAX_OWNERS;
        ra.pend = PendRef{(Slot *)s->pend->slots, ktl::geom_of(s->pend), s->pend->flags, s->pend->distinct};
        hipError_t e = hipMemsetAsync(s->cursors, 0, 2 * ktsk::MAX_OWNERS * 8, ctx->stream);
        if (e == hipSuccess) {
            const char *ge = getenv("KT_ROUTE_GRID");  // (experiments: workgroups per CU)
            const uint32_t per_cu = ge && atoi(ge) > 0 ? (uint32_t)atoi(ge) : 16u;
            hipLaunchKernelGGL(route_kernel, dim3(ktl::grid_for(ctx, a.n_seg, per_cu)), dim3(BLOCK), 0, ctx->stream, ra);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h_cur.data(), s->cursors, 2 * ktsk::MAX_OWNERS * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) hip_fail("the route pass", e);
    }
    std::vector<uint64_t> fill((size_t)V, 0), kmers((size_t)V, 0);
    if (my_code == KT_OK)
        for (int o = 0; o < V; o++) {
            fill[(size_t)o] = h_cur[(size_t)o] < s->room ? h_cur[(size_t)o] : s->room;
            kmers[(size_t)o] = h_cur[ktsk::MAX_OWNERS + (size_t)o];
        }
    s->last_records.assign(h_cur.begin(), h_cur.begin() + V);  // (statistics: kt_sharded_route_stats)
    s->last_kmers = kmers;
    // ---- can everybody go on, and how much will each send each?  go word p -> rank p: [0] status (0 = fine), [1] records for
    // rank p, [2] their k-mers at most, [3] the room of a region here (all ranks must have made their counters alike),
    // [4] the most records this rank sends any peer (the host transport's equal blocks: every rank takes the largest)
    std::vector<uint64_t> from_rec((size_t)N, 0), from_km((size_t)N, 0);
    uint64_t max_piece = 0;
    if (N > 1) {
        std::vector<uint64_t> h((size_t)N * GO_WORDS, 0);
        uint64_t my_max = 0;
        for (int p = 0; p < N; p++)
            if (p != me && fill[(size_t)p] > my_max) my_max = fill[(size_t)p];
        for (int p = 0; p < N; p++) {
            uint64_t *wd = &h[(size_t)p * GO_WORDS];
            wd[0] = my_code != KT_OK ? 1u : 0u;
            wd[1] = fill[(size_t)p];
            wd[2] = kmers[(size_t)p];
            wd[3] = s->room;
            wd[4] = my_max;
        }
        // (a rank that cannot even put its words on the device still enters the exchange - leaving here would leave the peers
        // waiting in it - and sends "cannot" from go_cannot: device words written at creation, which no copy of this call has
        // to reach)
        bool words_ok = hipMemcpyAsync(s->go_send, h.data(), h.size() * 8, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
        if (words_ok) words_ok = hipStreamSynchronize(ctx->stream) == hipSuccess;  // (h lives on this frame; the comm stream must see the words)
        if (!words_ok) {
            (void)hipGetLastError();
            kt::set_error("kt_sharded_add_reads: the status words could not be copied to the device");
            local_fail(KT_ERR_HIP);
        }
        std::vector<Piece> pc((size_t)N);
        for (int p = 0; p < N; p++)
            if (p != me)
                pc[p] = Piece{words_ok ? s->go_send + (size_t)p * GO_WORDS : s->go_cannot, GO_WORDS * 8, s->go_recv + (size_t)p * GO_WORDS, GO_WORDS * 8};
        if (int rc = exchange_v(s, pc)) return rc;  // (a transport that fails fails for everybody)
        KT_HIP(hipStreamSynchronize(s->comm_stream));
        std::vector<uint64_t> got((size_t)N * GO_WORDS, 0);
        KT_HIP(hipMemcpy(got.data(), s->go_recv, got.size() * 8, hipMemcpyDeviceToHost));
        bool peer = false, unlike = false;
        max_piece = my_max;
        for (int p = 0; p < N; p++) {
            if (p == me) continue;
            const uint64_t *wd = &got[(size_t)p * GO_WORDS];
            peer |= wd[0] != 0;
            if (wd[0] != 0) continue;
            // (every rank compares every peer's room with its own: a mismatch is seen by both ends, in this same round)
            unlike |= wd[3] != s->room || wd[1] > s->room;
            from_rec[(size_t)p] = wd[1];
            from_km[(size_t)p] = wd[2];
            if (wd[4] > max_piece) max_piece = wd[4];
        }
        if (my_code != KT_OK) return kt::fail(my_code, my_error);
        if (peer)
            return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: another rank could not take part (its batch was refused or its "
                                        "set-up failed); nothing was counted");
        if (unlike)
            return kt::fail(KT_ERR_ARG, "kt_sharded_add_reads: the ranks' counters differ (max_batch_bases / k must be the same on "
                                        "every rank); nothing was counted");
    } else if (my_code != KT_OK) {
        return kt::fail(my_code, my_error);
    }
    // ---- what this rank will count: its own region (read where it lies) and what the peers send - or, a single rank made
    // to take this path, all of its own regions
    uint64_t all_km = 0, all_rec = 0;
    if (N > 1) {
        all_km = kmers[(size_t)me];
        all_rec = fill[(size_t)me];
        for (int p = 0; p < N; p++)
            if (p != me) {
                all_km += from_km[(size_t)p];
                all_rec += from_rec[(size_t)p];
            }
    } else {
        for (int o = 0; o < V; o++) {
            all_km += kmers[(size_t)o];
            all_rec += fill[(size_t)o];
        }
    }
    // blocks [lo, hi) of a region of n_rec records that piece i of P holds
    auto piece_of = [&](uint64_t n_rec, int i, uint64_t *lo, uint64_t *hi) {
        const uint64_t nb = (n_rec + ktsk::BLOCK_RECS - 1) / ktsk::BLOCK_RECS;
        *lo = nb * (uint64_t)i / (uint64_t)P;
        *hi = nb * (uint64_t)(i + 1) / (uint64_t)P;
    };
    auto run_of = [&](const uint64_t *region, uint64_t n_rec, uint64_t lo, uint64_t hi) {
        uint64_t n = hi * ktsk::BLOCK_RECS < n_rec ? (hi - lo) * ktsk::BLOCK_RECS : n_rec - lo * ktsk::BLOCK_RECS;
        if (hi <= lo) n = 0;
        return ktsk::RecRun{region + lo * ktsk::BLOCK_WORDS, n};
    };
    int eligible = 0;
    if (all_rec) {
        // The job is planned for the k-mers that will arrive (the ranks announced them): level 1 runs 0.8 ms faster into
        // regions of that size than into the quarter more a job over reads gets by being planned for one k-mer per BASE.  But
        // the k-mers of a genome's repeats do not fit fine regions that tight - a fifth of the buckets went through the exact
        // pass, +2.8 ms per step on genome-sampled reads: once a job has had more than 2 % of its buckets redone the counter
        // plans the quarter more from then on.
        if (s->table->l2_buckets && s->table->l2_redone * 50u > s->table->l2_buckets) s->roomy = true;
        if (int rc = kt_bulk_begin(s->table, all_km + (s->roomy ? all_km / 4 : 0) + 1, &eligible)) return rc;
    }
    std::vector<ktsk::RecRun> late;  // (the probing path counts everything behind the last piece)
    auto count_runs = [&](const std::vector<ktsk::RecRun> &runs) -> int {
        if (runs.empty()) return KT_OK;
        if (eligible) return kt_bulk_add_records(s->table, runs.data(), (uint32_t)runs.size(), 0);
        late.insert(late.end(), runs.begin(), runs.end());
        return KT_OK;
    };
    if (N > 1) {
        // this rank's own region at once, under the first pieces' flight
        KT_HIP(hipEventRecord(s->ev_main, ctx->stream));
        KT_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_main, 0));  // (the route pass has long finished: the host read its cursors)
        if (fill[(size_t)me]) {
            std::vector<ktsk::RecRun> own{ktsk::RecRun{s->region(s->send, me), fill[(size_t)me]}};
            if (int rc = count_runs(own)) return rc;
        }
        uint64_t mlo = 0, mhi = 0, piece_blocks = 0;
        for (int i = 0; i < P; i++) {  // (the host transport's equal blocks: the largest piece any rank sends any other)
            piece_of(max_piece, i, &mlo, &mhi);
            if (mhi - mlo > piece_blocks) piece_blocks = mhi - mlo;
        }
        for (int i = 0; i < P; i++) {
            std::vector<Piece> pc((size_t)N);
            std::vector<ktsk::RecRun> runs;
            for (int p = 0; p < N; p++) {
                if (p == me) continue;
                uint64_t slo, shi, rlo, rhi;
                piece_of(fill[(size_t)p], i, &slo, &shi);
                piece_of(from_rec[(size_t)p], i, &rlo, &rhi);
                pc[p] = Piece{s->region(s->send, p) + slo * ktsk::BLOCK_WORDS, (size_t)(shi - slo) * ktsk::BLOCK_BYTES,
                              s->region(s->recv, p) + rlo * ktsk::BLOCK_WORDS, (size_t)(rhi - rlo) * ktsk::BLOCK_BYTES};
                const ktsk::RecRun r = run_of(s->region(s->recv, p), from_rec[(size_t)p], rlo, rhi);
                if (r.n_rec) runs.push_back(r);
            }
            if (int rc = exchange_v(s, pc, (size_t)piece_blocks * ktsk::BLOCK_BYTES)) return rc;
            KT_HIP(hipEventRecord(s->ev_recv[(size_t)i], s->comm_stream));
            KT_HIP(hipStreamWaitEvent(ctx->stream, s->ev_recv[(size_t)i], 0));
            if (int rc = count_runs(runs)) return rc;
        }
    } else {
        // a single rank: the same launches over its own regions - region 0 whole, the others piece by piece
        for (int i = -1; i < P; i++) {
            std::vector<ktsk::RecRun> runs;
            for (int o = (i < 0 ? 0 : 1); o < (i < 0 ? 1 : V); o++) {
                uint64_t lo = 0, hi = (fill[(size_t)o] + ktsk::BLOCK_RECS - 1) / ktsk::BLOCK_RECS;
                if (i >= 0) piece_of(fill[(size_t)o], i, &lo, &hi);
                const ktsk::RecRun r = run_of(s->region(s->send, o), fill[(size_t)o], lo, hi);
                if (r.n_rec) runs.push_back(r);
                if (i >= 0) s->exchanged_bytes += (hi - lo) * ktsk::BLOCK_BYTES;  // (what a rank of V would have sent)
            }
            if (int rc = count_runs(runs)) return rc;
        }
    }
    if (eligible) {
        if (int rc = kt_bulk_finish(s->table)) return rc;
    } else if (!late.empty()) {
        if (int rc = kt_ctr_count_records(s->table, late.data(), (uint32_t)late.size())) return rc;
    }
    if (mem == KT_MEM_HOST) KT_HIP(hipStreamSynchronize(ctx->stream));
    return KT_OK;
}

int kt_sharded_finalize(kt_sharded *s) {
    if (!s) return kt::fail(KT_ERR_ARG, "kt_sharded_finalize: null");
    if (!s->routed) return KT_OK;
    kt_ctx *ctx = s->ctx;
    if (int rc = ctx->use()) return rc;
    const uint64_t words = HDR_U64 + FIN_CAP + FIN_CAP / 2, all_words = words * (uint64_t)s->n_ranks;
    std::vector<uint64_t> hdr(all_words ? (size_t)s->n_ranks * HDR_U64 : 0);
    // the pending table's pairs, once: the rounds below send them FIN_CAP per owner at a time.  A table that filled up
    // is this rank's failure - told to every rank in the round's headers, so that all of them return it after the
    // exchange (a rank that left here would leave the others waiting).
    uint64_t n_pend = 0;
    bool overflowed = false;
    {
        const int rc = kt_ctr_size(s->pend, &n_pend);
        if (rc == KT_ERR_FULL) overflowed = true;
        else if (rc != KT_OK) return rc;
        if (!overflowed && n_pend) {
            uint64_t got = 0;
            if (int rc2 = kt_ctr_export(s->pend, s->pend_keys, s->pend_counts, s->pend_cap, &got, KT_MEM_DEVICE)) return rc2;
            n_pend = got;
        } else {
            n_pend = 0;
        }
        kt::set_error("");
    }
    for (int round = 0; round < 1 << 20; round++) {
        for (int p = 0; p < s->n_ranks; p++)
            KT_HIP(hipMemsetAsync(s->fin_send + (uint64_t)p * words, 0, HDR_U64 * 8, ctx->stream));
        KT_HIP(hipMemsetAsync(s->fin_left, 0, 8, ctx->stream));
        if (n_pend) {
            hipLaunchKernelGGL(pack_pending_kernel, dim3(ktl::grid_for(ctx, (n_pend + BLOCK - 1) / BLOCK, 4)), dim3(BLOCK), 0,
                               ctx->stream, s->pend_keys, (const uint32_t *)s->pend_counts, n_pend, (uint32_t)s->n_ranks,
                               (uint32_t)s->k, words, s->fin_send, s->fin_left);
        }
        hipLaunchKernelGGL(stamp_left_kernel, dim3(1), dim3(64), 0, ctx->stream, s->fin_send, (uint32_t)s->n_ranks, words,
                           s->fin_left, (uint64_t)(overflowed ? 1 : 0));
        KT_HIP(hipGetLastError());
        KT_HIP(hipEventRecord(s->ev_main, ctx->stream));
        KT_HIP(hipStreamWaitEvent(s->comm_stream, s->ev_main, 0));
        if (int rc = exchange(s, s->fin_send, s->fin_recv, words)) return rc;
        // the headers decide whether another round is needed: every rank sees every rank's remainder and status
        for (int p = 0; p < s->n_ranks; p++)
            KT_HIP(hipMemcpyAsync(hdr.data() + (size_t)p * HDR_U64, s->fin_recv + (uint64_t)p * words, HDR_U64 * 8,
                                  hipMemcpyDeviceToHost, s->comm_stream));
        KT_HIP(hipStreamSynchronize(s->comm_stream));
        bool more = false, failed = false;
        for (int p = 0; p < s->n_ranks; p++) {
            more |= hdr[(size_t)p * HDR_U64 + 1] != 0;
            failed |= hdr[(size_t)p * HDR_U64 + 2] != 0;
        }
        if (failed)
            return kt::fail(KT_ERR_FULL, overflowed ? "sharded counter: too many distinct k-mers did not fit their exchange regions (batch larger than max_batch_bases allows?)"
                                                    : "sharded counter: another rank's pending table overflowed");
        for (int p = 0; p < s->n_ranks; p++) {
            const uint64_t n = hdr[(size_t)p * HDR_U64] < FIN_CAP ? hdr[(size_t)p * HDR_U64] : FIN_CAP;
            if (n)
                if (int rc = kt_ctr_add_pairs(s->table, s->fin_recv + (uint64_t)p * words + HDR_U64,
                                              reinterpret_cast<const uint32_t *>(s->fin_recv + (uint64_t)p * words + HDR_U64 + FIN_CAP),
                                              n, KT_MEM_DEVICE))
                    return rc;
        }
        if (!more) {
            KT_HIP(hipStreamSynchronize(ctx->stream));  // (fin_recv is reused by the next finalize)
            if (!n_pend && !overflowed) {               // nothing was pending: the table is as clean as it was
                s->pend_touched = false;
                return KT_OK;
            }
            s->pend_touched = false;
            return kt_ctr_clear(s->pend);               // everything pending has been delivered
        }
        KT_HIP(hipStreamSynchronize(ctx->stream));
    }
    return kt::fail(KT_ERR_HIP, "sharded counter: finalize did not converge");
}

}  // extern "C"
