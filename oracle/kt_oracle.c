/*
 * kt_oracle.c - CPU restatement of kmertools' k-mer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product path (kmertools_amd/, libkmertools_hip.so) never links or calls it.
 *
 * Parity pinning: the reference is pure Rust and cannot be built in this image
 * (no cargo/rustc), so this restatement is pinned by the reference's committed
 * fixtures and in-file known-answer tests (tests/golden/, tests/test_oracle_golden.py):
 * expected_fa.kmers, expected_fa_batch_unnorm.kmers, expected_fa_header.kmers,
 * expected_reads.k4.cgr, expected_counts.part_0_chunk_0, the merge fixtures,
 * and the 70-entry k=31 list of kmer/src/kmer_minimisers.rs:216-288.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose algorithm it follows.  The structure is deliberately the reference's
 * (a serial rolling generator), NOT the GPU's position-parallel formulation,
 * so that the two are independent derivations of the same contract.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* kmer/src/kmer.rs:6-15 (duplicated at kmer/src/lib.rs:7-16): byte -> code.
 * A/a=0 C/c=1 G/g=2 T/t/U/u=3, raw bytes 0..3 map to themselves, rest = 4. */
static uint8_t NT4[256];
static pthread_once_t nt4_once = PTHREAD_ONCE_INIT;
static void nt4_init(void) {
    memset(NT4, 4, sizeof NT4);
    NT4[0] = 0; NT4[1] = 1; NT4[2] = 2; NT4[3] = 3;
    NT4['A'] = NT4['a'] = 0;
    NT4['C'] = NT4['c'] = 1;
    NT4['G'] = NT4['g'] = 2;
    NT4['T'] = NT4['t'] = 3;
    NT4['U'] = NT4['u'] = 3;
}
uint8_t kto_nt4(uint8_t c) {
    pthread_once(&nt4_once, nt4_init);
    return NT4[c];
}

/* ------------------------------------------------------------------------ */
/* kmer/src/kmer.rs:18-41 KmerGenerator state + new() */
typedef struct {
    const uint8_t *seq;
    uint64_t len_seq;
    uint64_t fval, rval;
    uint64_t len, pos, ksize;
    uint64_t mask, shift;
} kto_gen;

void kto_gen_init(kto_gen *g, const uint8_t *seq, uint64_t n, uint64_t k) {
    pthread_once(&nt4_once, nt4_init);
    g->seq = seq; g->len_seq = n;
    g->fval = g->rval = 0; g->len = 0; g->pos = 0; g->ksize = k;
    g->mask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1); /* kmer.rs:38 (k<=31 in practice) */
    g->shift = 2 * (k - 1);                                 /* kmer.rs:39 */
}

/* kmer/src/kmer.rs:80-106 Iterator::next.  Returns 1 and fills (f,r), or 0 at end. */
int kto_gen_next(kto_gen *g, uint64_t *f, uint64_t *r) {
    for (;;) {
        if (g->pos == g->len_seq) return 0;                 /* :82-84 */
        uint64_t fv = NT4[g->seq[g->pos]];                   /* :85-86 */
        uint64_t rv = fv ^ 3;                                /* :87 */
        g->pos += 1;                                         /* :88 */
        if (fv < 4) {                                        /* :90-94 */
            g->fval = ((g->fval << 2) | fv) & g->mask;
            g->rval = (g->rval >> 2) | (rv << g->shift);
            g->len += 1;
        } else {
            g->len = 0;                                      /* :95-98 */
        }
        if (g->len == g->ksize) {                            /* :100-103 */
            g->len -= 1;
            *f = g->fval; *r = g->rval;
            return 1;
        }
    }
}

/* All (fwd,rev) pairs of one sequence in positional order; returns the count.
 * `end_pos` (optional) receives the index of each k-mer's last base. */
uint64_t kto_kmers(const uint8_t *seq, uint64_t n, uint64_t k,
                   uint64_t *fwd, uint64_t *rev, uint64_t *end_pos) {
    kto_gen g; kto_gen_init(&g, seq, n, k);
    uint64_t f, r, c = 0;
    while (kto_gen_next(&g, &f, &r)) {
        if (fwd) fwd[c] = f;
        if (rev) rev[c] = r;
        if (end_pos) end_pos[c] = g.pos - 1;
        c++;
    }
    return c;
}

/* kmer/src/kmer.rs:43-52 rev_comp */
uint64_t kto_rev_comp(uint64_t kmer, uint64_t k) {
    uint64_t rk = 0;
    for (uint64_t i = 0; i < k; i++) {
        rk <<= 2;
        rk |= (kmer & 3) ^ 3;
        kmer >>= 2;
    }
    return rk;
}

/* kmer/src/kmer.rs:54-73 kmer_pos_maps.
 * min_mer_pos_map[4^k] (0 for non-canonical slots), pos_min_mer[count] (the
 * HashMap pos->kmer as a dense array), returns count.  The reference collects
 * min(kmer,rc) into a HashSet, sorts ascending, and numbers them. */
static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}
uint64_t kto_pos_maps(uint64_t k, uint64_t *min_mer_pos_map, uint64_t *pos_min_mer) {
    uint64_t n = 1ULL << (2 * k);
    uint64_t *set = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t cnt = 0;
    for (uint64_t km = 0; km < n; km++) {
        uint64_t rc = kto_rev_comp(km, k);
        uint64_t mn = km < rc ? km : rc;
        if (mn == km) set[cnt++] = km;   /* set membership: each canonical value once */
        if (min_mer_pos_map) min_mer_pos_map[km] = 0;
    }
    qsort(set, cnt, sizeof(uint64_t), cmp_u64);
    for (uint64_t pos = 0; pos < cnt; pos++) {
        if (min_mer_pos_map) min_mer_pos_map[set[pos]] = pos;
        if (pos_min_mer) pos_min_mer[pos] = set[pos];
    }
    free(set);
    return cnt;
}

/* kmer/src/lib.rs:19-34 numeric_to_kmer (out must hold k+1 bytes) */
void kto_numeric_to_kmer(uint64_t kmer, uint64_t k, char *out) {
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t i = 0; i < k; i++) {
        out[k - 1 - i] = L[kmer & 3];
        kmer >>= 2;
    }
    out[k] = 0;
}

/* kmer/src/lib.rs:36-50 kmer_to_numeric (no validity check: 'N' contributes 4) */
void kto_kmer_to_numeric(const char *s, uint64_t n, uint64_t *fwd, uint64_t *rev) {
    pthread_once(&nt4_once, nt4_init);
    uint64_t f = 0, r = 0;
    uint64_t shift = 2 * (n - 1);
    uint64_t mask = (n >= 32) ? ~0ULL : ((1ULL << (2 * n)) - 1);
    for (uint64_t i = 0; i < n; i++) {
        uint64_t fv = NT4[(uint8_t)s[i]];
        uint64_t rv = fv ^ 3;
        f = ((f << 2) | fv) & mask;
        r = (r >> 2) | (rv << shift);
    }
    *fwd = f; *rev = r;
}

/* ------------------------------------------------------------------------ */
/* composition/src/oligo.rs:231-259 vectorise_one (CLI crate, total_step=1)
 * pybindings/src/oligo.rs:39-69   vectorise_one (python; raw mode total += 2)
 * composition/src/oligocgr.rs:145-163 seq_to_kmer (== count_min=1)
 * out has kcount (count_min) or 4^k (raw) doubles. */
void kto_oligo_one(const uint8_t *seq, uint64_t n, uint64_t k, int count_min, int norm,
                   double total_step, const uint64_t *pos_map, uint64_t bins, double *out) {
    for (uint64_t i = 0; i < bins; i++) out[i] = 0.0;
    double total = 0.0;
    kto_gen g; kto_gen_init(&g, seq, n, k);
    uint64_t f, r;
    while (kto_gen_next(&g, &f, &r)) {
        if (count_min) {
            uint64_t mn = f < r ? f : r;           /* oligo.rs:243 */
            out[pos_map[mn]] += 1.0;               /* :246-247 */
            total += 1.0;                          /* :248 */
        } else {
            out[f] += 1.0;                         /* :250 */
            total += total_step;                   /* :251 (1.0) / pybindings oligo.rs:61 (2.0) */
        }
    }
    if (norm) {
        double d = total > 1.0 ? total : 1.0;      /* f64::max(1, total) :255-257 */
        for (uint64_t i = 0; i < bins; i++) out[i] /= d;
    }
}

typedef struct {
    const uint8_t *bases; const uint64_t *offsets; uint64_t n_reads; uint64_t k;
    int count_min, norm; double total_step; const uint64_t *pos_map; uint64_t bins;
    double *out; _Atomic uint64_t *next;
} oligo_job;

static void *oligo_worker(void *p) {
    oligo_job *j = (oligo_job *)p;
    for (;;) {
        /* analogue of rayon par_iter over in-memory reads, pybindings/src/oligo.rs:77-81 */
        uint64_t i0 = atomic_fetch_add(j->next, 256);
        if (i0 >= j->n_reads) break;
        uint64_t i1 = i0 + 256 < j->n_reads ? i0 + 256 : j->n_reads;
        for (uint64_t i = i0; i < i1; i++)
            kto_oligo_one(j->bases + j->offsets[i], j->offsets[i + 1] - j->offsets[i], j->k,
                          j->count_min, j->norm, j->total_step, j->pos_map, j->bins,
                          j->out + i * j->bins);
    }
    return NULL;
}

/* Batch over CSR reads into a preallocated n_reads x bins f64 matrix, `threads` workers. */
int kto_oligo_batch(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, uint64_t k,
                    int count_min, int norm, double total_step, double *out, int threads) {
    uint64_t n4k = 1ULL << (2 * k);
    uint64_t *pos_map = (uint64_t *)malloc(n4k * sizeof(uint64_t));
    uint64_t kcount = kto_pos_maps(k, pos_map, NULL);
    uint64_t bins = count_min ? kcount : n4k;
    _Atomic uint64_t next = 0;
    oligo_job job = {bases, offsets, n_reads, k, count_min, norm, total_step, pos_map, bins, out, &next};
    if (threads <= 1) {
        oligo_worker(&job);
    } else {
        pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * threads);
        for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, oligo_worker, &job);
        for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
        free(t);
    }
    free(pos_map);
    return 0;
}

/* composition/src/oligocgr.rs:165-189 cgr_maps + :123-143 vectorise_one coordinate part.
 * xy[2*i],xy[2*i+1] for canonical k-mer i (ascending numeric order); the coordinates
 * are read-independent: start at centre, per base m = (corner + m)/2. */
void kto_cgr_coords(uint64_t k, double vecsize, double *xy) {
    uint64_t n4k = 1ULL << (2 * k);
    uint64_t *pos_kmer = (uint64_t *)malloc(n4k * sizeof(uint64_t));
    uint64_t kcount = kto_pos_maps(k, NULL, pos_kmer);
    /* corners: A(0,0) T(vs,0) G(vs,vs) C(0,vs)  oligocgr.rs:166-170 ; codes A0 C1 G2 T3 */
    const double cx[4] = {0.0, 0.0, vecsize, vecsize};
    const double cy[4] = {0.0, vecsize, vecsize, 0.0};
    char s[40];
    for (uint64_t i = 0; i < kcount; i++) {
        kto_numeric_to_kmer(pos_kmer[i], k, s);     /* oligocgr.rs:36-38 */
        double mx = vecsize / 2.0, my = vecsize / 2.0;
        for (uint64_t j = 0; j < k; j++) {          /* :129-134, string order = MSB base first */
            int c = s[j] == 'A' ? 0 : s[j] == 'C' ? 1 : s[j] == 'G' ? 2 : 3;
            mx = (cx[c] + mx) / 2.0;
            my = (cy[c] + my) / 2.0;
        }
        xy[2 * i] = mx; xy[2 * i + 1] = my;
    }
    free(pos_kmer);
}

/* ------------------------------------------------------------------------ */
/* counter/src/lib.rs:92-170 count_chunk + :172-234 merge, in memory.
 * The reference keeps `n_parts` concurrent maps (scc::HashMap), owner = min_mer % n_parts
 * (:100,:127), `entry(m).and_modify(+1).or_insert(1)` (:126-130), u32 counts.  An scc map locks
 * one BUCKET (a cache-line sized cell of 32 entries) per operation and grows by rehashing in
 * steps, so threads that hit different buckets of one map do not wait for each other.
 * Here: every map is an array of GROUPS of 8 slots - 128 bytes, two adjacent cache lines: lock,
 * counts, keys - each with its own spin lock; a key hashes to a group and probes inside it only.  A group that is full hands the key to a small
 * overflow table behind a mutex (so an add never fails and never needs the map to grow while
 * other threads are in it); the map itself is resized between the threaded phases
 * (map_reserve: before a batch is added it gets room for the batch's k-mers - what scc does
 * step by step while adding).  Worker-pull over reads = records.lock().next() (:119). */
#define KTO_EMPTY 0xFFFFFFFFFFFFFFFFULL   /* canonical k-mers are < 2^62 (kmer.rs:38) */
#define KTO_G 8                           /* slots per group */

typedef struct {
    _Atomic uint32_t lock; uint32_t n;     /* n = keys in the group */
    uint32_t vals[KTO_G];
    uint64_t keys[KTO_G];
    char pad[128 - 8 - 4 * KTO_G - 8 * KTO_G];
} kto_group;                               /* 128 bytes: an aligned pair of cache lines per group */

typedef struct {                           /* open addressing, grows by doubling; callers serialise */
    uint64_t *keys; uint32_t *vals; uint64_t cap, used;
} kto_flat;

typedef struct {
    kto_group *groups; uint64_t n_groups;  /* power of two */
    kto_flat ov; pthread_mutex_t ov_mu;    /* keys whose group is full */
    char pad[64];                          /* maps of one counter sit in an array: no shared lines */
} kto_map;

static inline uint64_t mix64(uint64_t z) {
    z += 0x9e3779b97f4a7c15ULL;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static void flat_init(kto_flat *m, uint64_t cap) {
    m->cap = cap; m->used = 0;
    m->keys = (uint64_t *)malloc(cap * sizeof(uint64_t));
    m->vals = (uint32_t *)calloc(cap, sizeof(uint32_t));
    memset(m->keys, 0xFF, cap * sizeof(uint64_t));
}
static void flat_add(kto_flat *m, uint64_t key, uint32_t add);
static void flat_grow(kto_flat *m) {
    kto_flat n; flat_init(&n, m->cap * 2);
    for (uint64_t i = 0; i < m->cap; i++)
        if (m->keys[i] != KTO_EMPTY) flat_add(&n, m->keys[i], m->vals[i]);
    free(m->keys); free(m->vals);
    *m = n;
}
static void flat_add(kto_flat *m, uint64_t key, uint32_t add) {
    if ((m->used + 1) * 10 > m->cap * 7) flat_grow(m);
    uint64_t h = mix64(key) & (m->cap - 1);
    for (;;) {
        if (m->keys[h] == key) { m->vals[h] += add; return; }        /* and_modify / += (u32 wrap) */
        if (m->keys[h] == KTO_EMPTY) { m->keys[h] = key; m->vals[h] = add; m->used++; return; } /* or_insert */
        h = (h + 1) & (m->cap - 1);
    }
}
static int flat_get(const kto_flat *m, uint64_t key, uint32_t *val) {
    uint64_t h = mix64(key) & (m->cap - 1);
    for (;;) {
        if (m->keys[h] == key) { *val = m->vals[h]; return 1; }
        if (m->keys[h] == KTO_EMPTY) return 0;
        h = (h + 1) & (m->cap - 1);
    }
}

static kto_group *groups_new(uint64_t n_groups) {
    /* large tables: 2 MB alignment + MADV_HUGEPAGE - every add is a random access, and with 4 KB pages most of
     * its cost would be the page walk */
    const size_t bytes = n_groups * sizeof(kto_group);
    kto_group *g = NULL;
    if (bytes >= (4u << 20)) {
        if (posix_memalign((void **)&g, 2u << 20, (bytes + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1)) != 0) g = NULL;
#ifdef MADV_HUGEPAGE
        if (g) (void)madvise(g, bytes, MADV_HUGEPAGE);
#endif
    }
    if (!g) g = (kto_group *)aligned_alloc(128, bytes);
    for (uint64_t i = 0; i < n_groups; i++) {
        atomic_init(&g[i].lock, 0); g[i].n = 0;
        memset(g[i].keys, 0xFF, sizeof g[i].keys);
        memset(g[i].vals, 0, sizeof g[i].vals);
    }
    return g;
}
static void map_init(kto_map *m, uint64_t slots) {
    uint64_t ng = 1;
    while (ng * KTO_G < slots) ng <<= 1;
    m->groups = groups_new(ng); m->n_groups = ng;
    flat_init(&m->ov, 64);
    pthread_mutex_init(&m->ov_mu, NULL);
}
static void map_free(kto_map *m) {
    free(m->groups); free(m->ov.keys); free(m->ov.vals); pthread_mutex_destroy(&m->ov_mu);
}

/* table[key] += add; safe to call from many threads */
static void map_add(kto_map *m, uint64_t key, uint32_t add) {
    const uint64_t h = mix64(key);
    kto_group *g = &m->groups[(h >> 8) & (m->n_groups - 1)];
    uint32_t s = (uint32_t)h & (KTO_G - 1);
    while (atomic_exchange_explicit(&g->lock, 1u, memory_order_acquire)) {   /* the bucket's lock */
        uint32_t spins = 0;
        while (atomic_load_explicit(&g->lock, memory_order_relaxed)) {
            if (++spins < 64) __builtin_ia32_pause();
            else { sched_yield(); spins = 0; }                               /* the holder may have been descheduled */
        }
    }
    for (uint32_t i = 0; i < KTO_G; i++, s = (s + 1) & (KTO_G - 1)) {
        if (g->keys[s] == key) { g->vals[s] += add; goto done; }             /* and_modify / += (u32 wrap) */
        if (g->keys[s] == KTO_EMPTY) { g->keys[s] = key; g->vals[s] = add; g->n++; goto done; } /* or_insert */
    }
    atomic_store_explicit(&g->lock, 0u, memory_order_release);
    pthread_mutex_lock(&m->ov_mu);                                           /* a full group: overflow table */
    flat_add(&m->ov, key, add);
    pthread_mutex_unlock(&m->ov_mu);
    return;
done:
    atomic_store_explicit(&g->lock, 0u, memory_order_release);
}

/* no writers around (lookups of cov, exports) */
static uint32_t map_get(const kto_map *m, uint64_t key) {
    const uint64_t h = mix64(key);
    const kto_group *g = &m->groups[(h >> 8) & (m->n_groups - 1)];
    uint32_t s = (uint32_t)h & (KTO_G - 1);
    for (uint32_t i = 0; i < KTO_G; i++, s = (s + 1) & (KTO_G - 1)) {
        if (g->keys[s] == key) return g->vals[s];
        if (g->keys[s] == KTO_EMPTY) return 0;
    }
    uint32_t v = 0;
    return flat_get(&m->ov, key, &v) ? v : 0;
}

static uint64_t map_used(const kto_map *m) {
    uint64_t n = m->ov.used;
    for (uint64_t i = 0; i < m->n_groups; i++) n += m->groups[i].n;
    return n;
}

/* visits every (key, value); no writers around */
#define KTO_MAP_FOREACH(m, K, V, BODY)                                                   \
    do {                                                                                 \
        for (uint64_t gi_ = 0; gi_ < (m)->n_groups; gi_++)                               \
            for (uint32_t si_ = 0; si_ < KTO_G; si_++) {                                 \
                if ((m)->groups[gi_].keys[si_] == KTO_EMPTY) continue;                   \
                const uint64_t K = (m)->groups[gi_].keys[si_];                           \
                const uint32_t V = (m)->groups[gi_].vals[si_];                           \
                BODY                                                                     \
            }                                                                            \
        for (uint64_t oi_ = 0; oi_ < (m)->ov.cap; oi_++) {                               \
            if ((m)->ov.keys[oi_] == KTO_EMPTY) continue;                                \
            const uint64_t K = (m)->ov.keys[oi_];                                        \
            const uint32_t V = (m)->ov.vals[oi_];                                        \
            BODY                                                                         \
        }                                                                                \
    } while (0)

/* room for `keys` keys in all at load <= 0.5: ~2 % of the groups fill up and use the overflow table
 * (single-threaded: between the threaded phases) */
static void map_reserve(kto_map *m, uint64_t keys) {
    uint64_t want = 1;
    while (want * KTO_G * 5 < keys * 10) want <<= 1;
    if (want <= m->n_groups) return;
    kto_map n;
    n.groups = groups_new(want); n.n_groups = want;
    flat_init(&n.ov, 64);
    pthread_mutex_init(&n.ov_mu, NULL);
    KTO_MAP_FOREACH(m, key, val, { map_add(&n, key, val); });
    pthread_mutex_destroy(&n.ov_mu);
    free(m->groups); free(m->ov.keys); free(m->ov.vals);
    m->groups = n.groups; m->n_groups = n.n_groups; m->ov = n.ov;
}

typedef struct {
    kto_map *parts; uint64_t n_parts;
} kto_counter;

kto_counter *kto_counter_new(uint64_t n_parts) {
    kto_counter *c = (kto_counter *)malloc(sizeof(kto_counter));
    c->n_parts = n_parts ? n_parts : 1;
    c->parts = (kto_map *)malloc(sizeof(kto_map) * c->n_parts);
    for (uint64_t i = 0; i < c->n_parts; i++) map_init(&c->parts[i], 1024);
    return c;
}
void kto_counter_free(kto_counter *c) {
    for (uint64_t i = 0; i < c->n_parts; i++) map_free(&c->parts[i]);
    free(c->parts); free(c);
}

typedef struct {
    kto_counter *c; const uint8_t *bases; const uint64_t *offsets; uint64_t n_reads, k;
    _Atomic uint64_t *next;
} count_job;

static void *count_worker(void *p) {
    count_job *j = (count_job *)p;
    kto_counter *c = j->c;
    for (;;) {
        uint64_t i0 = atomic_fetch_add(j->next, 64);          /* records.lock().next() lib.rs:119 */
        if (i0 >= j->n_reads) break;
        uint64_t i1 = i0 + 64 < j->n_reads ? i0 + 64 : j->n_reads;
        for (uint64_t i = i0; i < i1; i++) {
            kto_gen g;
            kto_gen_init(&g, j->bases + j->offsets[i], j->offsets[i + 1] - j->offsets[i], j->k);
            uint64_t f, r;
            while (kto_gen_next(&g, &f, &r)) {                /* lib.rs:123 */
                uint64_t mn = f < r ? f : r;                   /* :124 */
                map_add(&c->parts[mn % c->n_parts], mn, 1);    /* :127-130 */
            }
        }
    }
    return NULL;
}

/* count_chunk's in-memory part: add every canonical k-mer of the reads. */
int kto_counter_add_reads(kto_counter *c, const uint8_t *bases, const uint64_t *offsets,
                          uint64_t n_reads, uint64_t k, int threads) {
    pthread_once(&nt4_once, nt4_init);
    /* room before the threads start: an empty map gets what this batch can add at most (one k-mer per base,
     * spread over the maps by `% n_parts`); a map that holds data doubles once it is half full - within a batch
     * the overflow tables take what the groups cannot */
    {
        const uint64_t bound = offsets[n_reads] / c->n_parts + offsets[n_reads] / c->n_parts / 8 + 1024;
        for (uint64_t i = 0; i < c->n_parts; i++) {
            const uint64_t used = map_used(&c->parts[i]);
            map_reserve(&c->parts[i], used ? 2 * used : bound);
        }
    }
    _Atomic uint64_t next = 0;
    count_job job = {c, bases, offsets, n_reads, k, &next};
    if (threads <= 1) {
        count_worker(&job);
    } else {
        pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * threads);
        for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, count_worker, &job);
        for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
        free(t);
    }
    return 0;
}

/* room for `keys` distinct keys in all (what scc reaches by growing while it adds) */
void kto_counter_reserve(kto_counter *c, uint64_t keys) {
    for (uint64_t i = 0; i < c->n_parts; i++) map_reserve(&c->parts[i], keys / c->n_parts + keys / c->n_parts / 8 + 1024);
}

/* merge's arithmetic (lib.rs:201-210): `*map.entry(kmer).or_insert(0) += count`. */
int kto_counter_add_pairs(kto_counter *c, const uint64_t *keys, const uint32_t *counts, uint64_t n) {
    for (uint64_t i = 0; i < c->n_parts; i++) map_reserve(&c->parts[i], map_used(&c->parts[i]) + n / c->n_parts + 1024);
    for (uint64_t i = 0; i < n; i++)
        map_add(&c->parts[keys[i] % c->n_parts], keys[i], counts[i]);
    return 0;
}

uint64_t kto_counter_size(const kto_counter *c) {
    uint64_t n = 0;
    for (uint64_t i = 0; i < c->n_parts; i++) n += map_used(&c->parts[i]);
    return n;
}

typedef struct { uint64_t k; uint32_t v; } kv_t;
static int cmp_kv(const void *a, const void *b) {
    uint64_t x = ((const kv_t *)a)->k, y = ((const kv_t *)b)->k;
    return (x > y) - (x < y);
}

/* map.scan (lib.rs:162-165, :220-230).  The reference's line order is unspecified
 * (its own tests sort); `sorted` != 0 returns keys ascending for comparisons. */
uint64_t kto_counter_export(const kto_counter *c, uint64_t *keys, uint32_t *counts, int sorted) {
    uint64_t n = kto_counter_size(c), o = 0;
    kv_t *kv = (kv_t *)malloc((n ? n : 1) * sizeof(kv_t));
    for (uint64_t p = 0; p < c->n_parts; p++) {
        const kto_map *m = &c->parts[p];
        KTO_MAP_FOREACH(m, key, val, { kv[o].k = key; kv[o].v = val; o++; });
    }
    if (sorted) qsort(kv, n, sizeof(kv_t), cmp_kv);
    for (uint64_t i = 0; i < n; i++) { keys[i] = kv[i].k; counts[i] = kv[i].v; }
    free(kv);
    return n;
}

/* ------------------------------------------------------------------------ */
/* The on-disk half of the reference's counter, restated so that the CPU baseline can pay
 * what the real CLI pays: count_chunk's spill (counter/src/lib.rs:151-167: one text file
 * "temp_kmers.part_P_chunk_C" per partition, lines "{kmer}\t{count}\n", partitions written
 * in parallel) and merge (:188-231: partitions one after the other; the chunk files of a
 * partition parsed by parallel tasks into one shared map with `*entry(kmer).or_insert(0) +=
 * count`; the map scanned into "kmers.counts", numeric or ACGT keys; temp files deleted). */
typedef struct { const kto_counter *c; const char *dir; uint64_t chunk; _Atomic uint64_t *next; int err; } spill_job;

static void *spill_worker(void *p) {
    spill_job *j = (spill_job *)p;
    char path[4096];
    for (;;) {
        uint64_t part = atomic_fetch_add(j->next, 1);            /* counts_table.par_iter().enumerate() :152-155 */
        if (part >= j->c->n_parts) break;
        snprintf(path, sizeof path, "%s/temp_kmers.part_%llu_chunk_%llu", j->dir,
                 (unsigned long long)part, (unsigned long long)j->chunk);   /* :156-159 */
        FILE *f = fopen(path, "w");
        if (!f) { j->err = 1; continue; }
        const kto_map *m = &j->c->parts[part];
        KTO_MAP_FOREACH(m, key, val, { fprintf(f, "%llu\t%u\n", (unsigned long long)key, val); }); /* map.scan :162-165 */
        fclose(f);
    }
    return NULL;
}

/* writes the counter's partitions as chunk `chunk` of `dir`; 0 on success */
int kto_counter_spill(const kto_counter *c, const char *dir, uint64_t chunk, int threads) {
    _Atomic uint64_t next = 0;
    spill_job job = {c, dir, chunk, &next, 0};
    if (threads <= 1) {
        spill_worker(&job);
    } else {
        pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * threads);
        for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, spill_worker, &job);
        for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
        free(t);
    }
    return job.err;
}

typedef struct { kto_map *map; const char *dir; uint64_t part, chunks; _Atomic uint64_t *next; int del, err; } merge_job;

static void *merge_worker(void *p) {
    merge_job *j = (merge_job *)p;
    char path[4096], line[128];
    for (;;) {
        uint64_t chunk = atomic_fetch_add(j->next, 1);           /* for chunk in 0..chunks: scope.spawn :194-199 */
        if (chunk >= j->chunks) break;
        snprintf(path, sizeof path, "%s/temp_kmers.part_%llu_chunk_%llu", j->dir,
                 (unsigned long long)j->part, (unsigned long long)chunk);
        FILE *f = fopen(path, "r");
        if (!f) { j->err = 1; continue; }
        while (fgets(line, sizeof line, f)) {                     /* buff.lines() :203 */
            char *tab = NULL;
            unsigned long long kmer = strtoull(line, &tab, 10);   /* parts.next().parse::<Kmer>() :205 */
            if (tab == line || *tab != '\t') continue;
            uint32_t count = (uint32_t)strtoul(tab + 1, NULL, 10);/* :206 */
            map_add(j->map, kmer, count);                         /* *entry(kmer).or_insert(0) += count :207 */
        }
        fclose(f);
        if (j->del) remove(path);                                 /* :208-210 */
    }
    return NULL;
}

/* merge(delete): `dir`/kmers.counts from the chunk files of n_parts partitions; returns the
 * number of lines written, or (uint64_t)-1 on an I/O error.  k is only used with acgt != 0. */
uint64_t kto_merge(const char *dir, uint64_t n_parts, uint64_t chunks, int threads, int acgt, uint64_t k, int del) {
    char path[4096], mer[40];
    snprintf(path, sizeof path, "%s/kmers.counts", dir);          /* :178 */
    FILE *out = fopen(path, "w");
    if (!out) return (uint64_t)-1;
    uint64_t lines = 0;
    int err = 0;
    for (uint64_t part = 0; part < n_parts; part++) {             /* for part in 0..n_parts :188 (serial) */
        kto_map map; map_init(&map, 1024);                        /* SccMap::new() :190 */
        {   /* room for the partition's lines (scc grows while the tasks add): a line is at least 4 bytes */
            uint64_t bytes = 0;
            for (uint64_t ch = 0; ch < chunks; ch++) {
                snprintf(path, sizeof path, "%s/temp_kmers.part_%llu_chunk_%llu", dir, (unsigned long long)part,
                         (unsigned long long)ch);
                FILE *pf = fopen(path, "r");
                if (pf) { fseek(pf, 0, SEEK_END); bytes += (uint64_t)ftell(pf); fclose(pf); }
            }
            map_reserve(&map, bytes / 4 + 1024);
        }
        _Atomic uint64_t next = 0;
        merge_job job = {&map, dir, part, chunks, &next, del, 0};
        int T = threads < 1 ? 1 : threads;
        if ((uint64_t)T > chunks) T = (int)chunks;
        if (T <= 1) {
            merge_worker(&job);
        } else {
            pthread_t *t = (pthread_t *)malloc(sizeof(pthread_t) * T);
            for (int i = 0; i < T; i++) pthread_create(&t[i], NULL, merge_worker, &job);
            for (int i = 0; i < T; i++) pthread_join(t[i], NULL);
            free(t);
        }
        err |= job.err;
        KTO_MAP_FOREACH(&map, key, val, {                          /* map_arc.scan :220-230 */
            if (acgt) {
                kto_numeric_to_kmer(key, k, mer);
                fprintf(out, "%s\t%u\n", mer, val);
            } else {
                fprintf(out, "%llu\t%u\n", (unsigned long long)key, val);
            }
            lines++;
        });
        map_free(&map);
    }
    fclose(out);
    return err ? (uint64_t)-1 : lines;
}

/* ------------------------------------------------------------------------ */
/* composition/src/cgr.rs:12-36 cgr_maps + :127-144 CgrComputer::vectorise_one (same loop in
 * pybindings/src/cgr.rs:38-54): whole-sequence chaos game walk.  Corners A=(0,0) T/U=(V,0)
 * G=(V,V) C=(0,V), either case; marker starts at (V/2, V/2); every base moves the marker half
 * way to its corner and emits it.  Any other byte is an error (returns the index of the first
 * bad byte + 1, 0 on success).  out = 2 doubles per base. */
uint64_t kto_cgr_one(const uint8_t *seq, uint64_t n, double vecsize, double *out) {
    double mx = vecsize / 2.0, my = vecsize / 2.0;
    for (uint64_t i = 0; i < n; i++) {
        double cx, cy;
        switch (seq[i]) {
            case 'A': case 'a': cx = 0.0; cy = 0.0; break;
            case 'T': case 't': case 'U': case 'u': cx = vecsize; cy = 0.0; break;
            case 'G': case 'g': cx = vecsize; cy = vecsize; break;
            case 'C': case 'c': cx = 0.0; cy = vecsize; break;
            default: return i + 1;                 /* cgr.rs:139-141 */
        }
        mx = (cx + mx) / 2.0;                      /* :134-137 */
        my = (cy + my) / 2.0;
        out[2 * i] = mx;
        out[2 * i + 1] = my;
    }
    return 0;
}

/* the same walk over a CSR batch (the par_iter of cgr.rs:92-106, serial here); out = 2 doubles per base */
uint64_t kto_cgr_batch(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, double vecsize, double *out) {
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t bad = kto_cgr_one(bases + offsets[i], offsets[i + 1] - offsets[i], vecsize, out + 2 * offsets[i]);
        if (bad) return offsets[i] + bad;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* kmer/src/minimiser.rs:21-175 MinimiserGenerator: state + Iterator::next, statement for statement.
 * The VecDeque (capacity wsize - msize + 1) is a ring buffer here.  Emits (minimiser, window start,
 * window end) every time the active minimiser of the sliding window changes, at an ambiguous base
 * that ends a full window, and once more at the last base.  wsize is used as given (the callers pass
 * seq.len() for "0", misc/src/minimisers.rs:44-48). */
typedef struct {
    const uint8_t *seq;
    uint64_t len_seq, pos, wsize, msize, m_mask, m_window_start, m_window_end;
    uint64_t m_val_f, m_val_r, m_val_l, m_active, m_shift;
    uint64_t *buff;
    uint64_t cap, head, blen, buff_pos;
} kto_mingen;

static uint64_t mg_get(const kto_mingen *g, uint64_t j) { return g->buff[(g->head + j) % g->cap]; }

static int kto_mingen_next(kto_mingen *g, uint64_t *kmer, uint64_t *ws, uint64_t *we) {
    const uint64_t full = g->wsize - g->msize + 1;   /* wraps like usize if wsize < msize - 1: callers exclude it */
    for (;;) {
        if (g->pos == g->len_seq) return 0;                                   /* :68-70 */
        uint64_t fv = NT4[g->seq[g->pos]];
        uint64_t rv = fv ^ 3;
        if (fv < 4) {                                                         /* :75-80 */
            g->m_val_f = ((g->m_val_f << 2) | fv) & g->m_mask;
            g->m_val_r = (g->m_val_r >> 2) | (rv << g->m_shift);
            g->m_val_l += 1;
        } else {                                                              /* :81-101 */
            int should_return = g->blen == full;
            uint64_t pm = g->m_active, pws = g->m_window_start, pwe = g->pos;
            g->buff_pos = 0;
            g->m_active = ~0ULL;
            g->m_val_f = g->m_val_r = g->m_val_l = 0;
            g->m_window_end = 0;
            g->m_window_start = g->pos + 1;
            g->blen = 0; g->head = 0;
            g->pos += 1;
            if (should_return) { *kmer = pm; *ws = pws; *we = pwe; return 1; }
            continue;
        }
        if (g->m_val_l < g->msize) { g->pos += 1; continue; }                 /* :103-106 */
        g->m_val_l -= 1;
        uint64_t mv = g->m_val_f < g->m_val_r ? g->m_val_f : g->m_val_r;      /* :110 */
        if (g->blen == full) {                                                /* :113 */
            g->head = (g->head + 1) % g->cap;                                 /* pop_front */
            g->buff[(g->head + g->blen - 1) % g->cap] = mv;                   /* push_back */
            if (g->buff_pos == 0) {                                           /* :119-138 */
                uint64_t new_min = ~0ULL;
                for (uint64_t j = 0; j < g->blen; j++)
                    if (mg_get(g, j) < new_min) { g->buff_pos = j; new_min = mg_get(g, j); }
                if (new_min != g->m_active) {
                    g->m_window_end = g->pos;
                    *kmer = g->m_active; *ws = g->m_window_start; *we = g->m_window_end;
                    g->m_active = new_min;
                    g->m_window_start = g->pos - g->wsize + 1;
                    g->pos += 1;
                    return 1;
                }
            } else if (mv < g->m_active) {                                    /* :139-149 */
                g->m_window_end = g->pos;
                *kmer = g->m_active; *ws = g->m_window_start; *we = g->m_window_end;
                g->m_active = mv;
                g->buff_pos = g->blen - 1;
                g->m_window_start = g->pos - g->wsize + 1;
                g->pos += 1;
                return 1;
            } else {
                g->buff_pos -= 1;                                             /* :150-152 */
            }
        } else {
            g->buff[(g->head + g->blen) % g->cap] = mv;                       /* :153-156 */
            g->blen += 1;
        }
        if (g->m_active == ~0ULL && g->blen == full) {                        /* :158-166 */
            for (uint64_t j = 0; j < g->blen; j++)
                if (mg_get(g, j) < g->m_active) { g->buff_pos = j; g->m_active = mg_get(g, j); }
        }
        if (g->pos == g->len_seq - 1) {                                       /* :168-171 */
            g->pos += 1;
            *kmer = g->m_active; *ws = g->m_window_start; *we = g->len_seq;
            return 1;
        }
        g->pos += 1;
    }
}

/* all (minimiser, start, end) triples of one sequence; returns the count (<= n + 1) */
uint64_t kto_minimisers(const uint8_t *seq, uint64_t n, uint64_t wsize, uint64_t msize,
                        uint64_t *kmers, uint64_t *starts, uint64_t *ends) {
    pthread_once(&nt4_once, nt4_init);
    kto_mingen g;
    memset(&g, 0, sizeof g);
    g.seq = seq; g.len_seq = n; g.wsize = wsize; g.msize = msize;
    g.m_active = ~0ULL;
    g.m_mask = (1ULL << (2 * msize)) - 1;
    g.m_shift = 2 * (msize - 1);
    g.cap = wsize - msize + 2;
    g.buff = (uint64_t *)malloc(g.cap * sizeof(uint64_t));
    uint64_t c = 0, k, s, e;
    while (kto_mingen_next(&g, &k, &s, &e)) { kmers[c] = k; starts[c] = s; ends[c] = e; c++; }
    free(g.buff);
    return c;
}

/* the per-record loop of misc/src/minimisers.rs:43-54 over a CSR batch (serial): returns the number of triples;
 * wsize 0 = each read's own length; reads shorter than msize are skipped in that mode */
uint64_t kto_minimisers_batch(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, uint64_t wsize,
                              uint64_t msize, uint64_t *kmers, uint64_t *starts, uint64_t *ends, uint64_t cap) {
    uint64_t total = 0;
    uint64_t tk[4096], ts[4096], te[4096];
    for (uint64_t i = 0; i < n_reads; i++) {
        const uint64_t n = offsets[i + 1] - offsets[i];
        const uint64_t w = wsize ? wsize : n;
        if (w < msize || n + 2 > 4096) continue;
        const uint64_t c = kto_minimisers(bases + offsets[i], n, w, msize, tk, ts, te);
        for (uint64_t j = 0; j < c && total + j < cap; j++) {
            kmers[total + j] = tk[j]; starts[total + j] = ts[j]; ends[total + j] = te[j];
        }
        total += c;
    }
    return total;
}

/* ------------------------------------------------------------------------ */
/* coverage/src/lib.rs:165-184 CovComputer::vectorise_one over a CSR batch:
 * per k-mer: count = table[min(f,r)] or 0 (:171), bin = min(floor(count / bin_size),
 * bin_count - 1) (:172-173), vec[bin] += 1, total += 1; norm: /= max(1, total) (:180-182). */
static uint32_t counter_get(const kto_counter *c, uint64_t key) {
    return map_get(&c->parts[key % c->n_parts], key);
}

int kto_cov_batch(const kto_counter *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                  uint64_t k, uint64_t bin_size, uint64_t bin_count, int norm, double *out) {
    pthread_once(&nt4_once, nt4_init);
    for (uint64_t i = 0; i < n_reads; i++) {
        double *vec = out + i * bin_count;
        for (uint64_t b = 0; b < bin_count; b++) vec[b] = 0.0;
        double total = 0.0;
        kto_gen g;
        kto_gen_init(&g, bases + offsets[i], offsets[i + 1] - offsets[i], k);
        uint64_t f, r;
        while (kto_gen_next(&g, &f, &r)) {
            uint64_t mn = f < r ? f : r;
            uint32_t count = counter_get(c, mn);
            uint64_t kmer_bin = (uint64_t)((double)count / (double)bin_size); /* floor of a non-negative f64 */
            uint64_t vb = kmer_bin < bin_count - 1 ? kmer_bin : bin_count - 1;
            vec[vb] += 1.0;
            total += 1.0;
        }
        if (norm) {
            double d = total > 1.0 ? total : 1.0;
            for (uint64_t b = 0; b < bin_count; b++) vec[b] /= d;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Synthetic reads: NOT a reference algorithm - the build's own
 * deterministic generator, mirrored bit-for-bit by the device generator
 * (kmertools_amd/csrc/kt_synth.hip) so the oracle can regenerate any slice.
 *   uniform   : base = mix64(seed + g) & 3, g = read*read_len + pos
 *   genome    : reads sampled (both strands) from a genome_len-bp random genome
 *               with 1 % substitution errors
 *   noise     : base -> 'N' with p ~ 0.001, lower-cased with p ~ 0.01 */
void kto_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint64_t read_len,
                     int noise, uint64_t genome_len, uint8_t *bases) {
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t rid = first_read + i;
        uint64_t start = 0; int strand = 0;
        if (genome_len) {
            uint64_t hr = mix64(seed ^ 0x5eed5eed5eedULL ^ mix64(rid));
            start = (hr >> 1) % (genome_len - read_len + 1);
            strand = (int)(hr & 1);
        }
        for (uint64_t p = 0; p < read_len; p++) {
            uint64_t g = rid * read_len + p;
            uint64_t h = mix64(seed + g);
            unsigned code;
            if (genome_len) {
                uint64_t j = strand ? start + (read_len - 1 - p) : start + p;
                code = (unsigned)(mix64((seed ^ 0x67656e6f6d65ULL) + j) & 3);
                if (strand) code = 3 - code;
                if (((h >> 40) % 100) == 0) code = (code + 1 + (unsigned)((h >> 50) % 3)) & 3;
            } else {
                code = (unsigned)(h & 3);
            }
            uint8_t c = (uint8_t)L[code];
            if (noise) {
                if (((h >> 8) & 0xFFFFF) < 1049) c = 'N';
                else if (((h >> 28) & 0xFFF) < 41) c = (uint8_t)(c | 0x20);
            }
            bases[i * read_len + p] = c;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* The reference's text formats for whole matrices, written to a file - so that the CLI's
 * multi-batch outputs (gigabytes of text) can be compared byte for byte without formatting
 * 10^8 numbers in Python.
 *   fixed6  : Rust `format!("{:.6}", x)` (composition/src/oligo.rs:132-134) = printf("%.6f")
 *   display : Rust `Display` for f64 (oligo.rs:136, oligocgr.rs:95): the shortest digit string
 *             that round-trips, positional notation only, integral values without ".0"
 * Rows are formatted by `threads` workers (contiguous shares), written in row order. */
static int fmt_display_c(double x, char *out) {
    if (x != x) return sprintf(out, "NaN");
    if (x == 0.0) return sprintf(out, (1.0 / x < 0) ? "-0" : "0");
    if (x - x != 0.0) return sprintf(out, x > 0 ? "inf" : "-inf");
    if (x == (double)(long long)x && x > -1e15 && x < 1e15) return sprintf(out, "%lld", (long long)x);
    char buf[40];
    int p;
    for (p = 1; p <= 17; p++) {                       /* shortest precision that round-trips */
        snprintf(buf, sizeof buf, "%.*e", p - 1, x);
        if (strtod(buf, NULL) == x) break;
    }
    /* buf = [-]d.ddddde[+-]XX  ->  positional */
    char digits[24]; int nd = 0, neg = 0;
    const char *s = buf;
    if (*s == '-') { neg = 1; s++; }
    for (; *s && *s != 'e'; s++) if (*s != '.') digits[nd++] = *s;
    int ex = atoi(s + 1);
    while (nd > 1 && digits[nd - 1] == '0') nd--;     /* trailing zeros of the mantissa */
    int point = 1 + ex;                               /* digits before the decimal point */
    char *o = out;
    if (neg) *o++ = '-';
    if (point <= 0) {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -point; i++) *o++ = '0';
        for (int i = 0; i < nd; i++) *o++ = digits[i];
    } else if (point >= nd) {
        for (int i = 0; i < nd; i++) *o++ = digits[i];
        for (int i = nd; i < point; i++) *o++ = '0';
    } else {
        for (int i = 0; i < point; i++) *o++ = digits[i];
        *o++ = '.';
        for (int i = point; i < nd; i++) *o++ = digits[i];
    }
    *o = 0;
    return (int)(o - out);
}

int kto_fmt_display(double x, char *out) { return fmt_display_c(x, out); }

typedef struct {
    const double *mat; uint64_t r0, r1, cols; int mode; char delim; const double *xy;
    char *buf; uint64_t len, cap;
} text_job;

static void text_put(text_job *j, const char *s, int n) {
    if (j->len + (uint64_t)n + 1 > j->cap) {
        j->cap = (j->cap + (uint64_t)n) * 2 + 4096;
        j->buf = (char *)realloc(j->buf, j->cap);
    }
    memcpy(j->buf + j->len, s, (size_t)n);
    j->len += (uint64_t)n;
}

/* mode 0 = display, 1 = fixed6, 2 = "(x,y,v)" triples of `comp cgr -k` (oligocgr.rs:93-97, xy = cgr_coords) */
static void *text_worker(void *p) {
    text_job *j = (text_job *)p;
    char t[400];   /* (a denormal in positional notation has ~330 characters) */
    /* display-mode values repeat (counts / read lengths): a small memo keyed by the bit pattern */
    enum { MEMO = 1 << 14, MEMO_LEN = 48 };
    uint64_t *mk = NULL; char (*mv)[MEMO_LEN] = NULL;
    if (j->mode != 1) { mk = (uint64_t *)malloc(MEMO * 8); mv = malloc(MEMO * MEMO_LEN); memset(mk, 0xFF, MEMO * 8); }
    for (uint64_t r = j->r0; r < j->r1; r++) {
        const double *row = j->mat + r * j->cols;
        for (uint64_t c = 0; c < j->cols; c++) {
            if (c) text_put(j, &j->delim, 1);
            if (j->mode == 1) {
                text_put(j, t, snprintf(t, sizeof t, "%.6f", row[c]));
                continue;
            }
            if (j->mode == 2) {
                text_put(j, "(", 1);
                text_put(j, t, fmt_display_c(j->xy[2 * c], t));
                text_put(j, ",", 1);
                text_put(j, t, fmt_display_c(j->xy[2 * c + 1], t));
                text_put(j, ",", 1);
            }
            uint64_t bits; memcpy(&bits, &row[c], 8);
            uint64_t h = (bits * 0x9e3779b97f4a7c15ULL) >> 50;
            if (mk[h] != bits || bits == ~0ULL) {
                const int n = fmt_display_c(row[c], t);
                if (n >= MEMO_LEN) { text_put(j, t, n); if (j->mode == 2) text_put(j, ")", 1); continue; }
                mk[h] = bits; memcpy(mv[h], t, (size_t)n + 1);
            }
            text_put(j, mv[h], (int)strlen(mv[h]));
            if (j->mode == 2) text_put(j, ")", 1);
        }
        text_put(j, "\n", 1);
    }
    free(mk); free(mv);
    return NULL;
}

/* appends (append != 0) or writes the text of `rows` x `cols` values to `path`; 0 on success */
int kto_matrix_text_file(const double *mat, uint64_t rows, uint64_t cols, int mode, char delim, const double *xy,
                         const char *path, int append, int threads) {
    if (threads < 1) threads = 1;
    if ((uint64_t)threads > rows) threads = rows ? (int)rows : 1;
    text_job *jobs = (text_job *)calloc((size_t)threads, sizeof(text_job));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    for (int i = 0; i < threads; i++) {
        jobs[i].mat = mat; jobs[i].cols = cols; jobs[i].mode = mode; jobs[i].delim = delim; jobs[i].xy = xy;
        jobs[i].r0 = rows * (uint64_t)i / (uint64_t)threads;
        jobs[i].r1 = rows * (uint64_t)(i + 1) / (uint64_t)threads;
        pthread_create(&th[i], NULL, text_worker, &jobs[i]);
    }
    FILE *f = fopen(path, append ? "ab" : "wb");
    int err = f == NULL;
    for (int i = 0; i < threads; i++) {
        pthread_join(th[i], NULL);
        if (f && jobs[i].len && fwrite(jobs[i].buf, 1, jobs[i].len, f) != jobs[i].len) err = 1;
        free(jobs[i].buf);
    }
    if (f && fclose(f) != 0) err = 1;
    free(jobs); free(th);
    return err;
}
