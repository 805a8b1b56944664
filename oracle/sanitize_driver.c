/* sanitize_driver.c - runs the threaded parts of the CPU oracle under AddressSanitizer / UBSan / ThreadSanitizer
 * (oracle/Makefile: kt_oracle_asan, kt_oracle_tsan; tests/test_oracle_sanitize.py).  Test infrastructure only.
 * Exit code 0 = the threaded results equal the single-threaded ones (the sanitizers abort on their own findings). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct kto_counter kto_counter;
kto_counter *kto_counter_new(uint64_t n_parts);
void kto_counter_free(kto_counter *c);
void kto_counter_reserve(kto_counter *c, uint64_t keys);
int kto_counter_add_reads(kto_counter *c, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, uint64_t k, int threads);
uint64_t kto_counter_size(const kto_counter *c);
uint64_t kto_counter_export(const kto_counter *c, uint64_t *keys, uint32_t *counts, int sorted);
int kto_counter_spill(const kto_counter *c, const char *dir, uint64_t chunk, int threads);
uint64_t kto_merge(const char *dir, uint64_t n_parts, uint64_t chunks, int threads, int acgt, uint64_t k, int del);
int kto_oligo_batch(const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, uint64_t k, int count_min, int norm,
                    double total_step, double *out, int threads);
int kto_matrix_text_file(const double *mat, uint64_t rows, uint64_t cols, int mode, char delim, const double *xy,
                         const char *path, int append, int threads);
void kto_synth_reads(uint64_t seed, uint64_t first_read, uint64_t n_reads, uint64_t read_len, int noise, uint64_t genome_len,
                     uint8_t *bases);

static int fail(const char *what) { fprintf(stderr, "sanitize_driver: %s\n", what); return 1; }

int main(int argc, char **argv) {
    const char *dir = argc > 1 ? argv[1] : "/tmp";
    const uint64_t n = argc > 2 ? strtoull(argv[2], NULL, 10) : 20000, L = 150;
    const int T = argc > 3 ? atoi(argv[3]) : 8;
    uint8_t *bases = (uint8_t *)malloc(n * L);
    uint64_t *offsets = (uint64_t *)malloc((n + 1) * 8);
    for (uint64_t i = 0; i <= n; i++) offsets[i] = i * L;
    kto_synth_reads(7, 0, n, L, 1, 200000, bases);   /* genome-sampled with noise: repeated k-mers, N, lower case */

    /* counter: T threads into T maps (twice: second batch into maps that hold data) against one thread into one map */
    kto_counter *a = kto_counter_new((uint64_t)T), *b = kto_counter_new(1);
    for (int rep = 0; rep < 2; rep++) {
        kto_counter_add_reads(a, bases, offsets, n, 21, T);
        kto_counter_add_reads(b, bases, offsets, n, 21, 1);
    }
    const uint64_t na = kto_counter_size(a), nb = kto_counter_size(b);
    if (na != nb) return fail("threaded counter: different number of distinct k-mers");
    uint64_t *ka = (uint64_t *)malloc((na + 1) * 8), *kb = (uint64_t *)malloc((nb + 1) * 8);
    uint32_t *ca = (uint32_t *)malloc((na + 1) * 4), *cb = (uint32_t *)malloc((nb + 1) * 4);
    kto_counter_export(a, ka, ca, 1);
    kto_counter_export(b, kb, cb, 1);
    if (memcmp(ka, kb, na * 8) || memcmp(ca, cb, na * 4)) return fail("threaded counter: different counts");

    /* spill + merge with T threads: as many lines as distinct k-mers */
    if (kto_counter_spill(a, dir, 0, T)) return fail("spill");
    if (kto_counter_spill(a, dir, 1, T)) return fail("spill");
    if (kto_merge(dir, (uint64_t)T, 2, T, 0, 21, 1) != na) return fail("merge: line count");

    /* per-read histograms and the text writer, T threads against one */
    const uint64_t bins = 136;
    double *m1 = (double *)malloc(n * bins * 8), *mt = (double *)malloc(n * bins * 8);
    kto_oligo_batch(bases, offsets, n, 4, 1, 1, 1.0, m1, 1);
    kto_oligo_batch(bases, offsets, n, 4, 1, 1, 1.0, mt, T);
    if (memcmp(m1, mt, n * bins * 8)) return fail("threaded oligo rows differ");
    char p1[4096], p2[4096];
    snprintf(p1, sizeof p1, "%s/rows_1.txt", dir);
    snprintf(p2, sizeof p2, "%s/rows_t.txt", dir);
    if (kto_matrix_text_file(m1, n, bins, 1, ' ', NULL, p1, 0, 1)) return fail("text 1");
    if (kto_matrix_text_file(mt, n / 2, bins, 1, ' ', NULL, p2, 0, T)) return fail("text t");
    if (kto_matrix_text_file(mt + (n / 2) * bins, n - n / 2, bins, 1, ' ', NULL, p2, 1, T)) return fail("text t append");
    FILE *f1 = fopen(p1, "rb"), *f2 = fopen(p2, "rb");
    if (!f1 || !f2) return fail("reopen");
    int c1, c2;
    do { c1 = fgetc(f1); c2 = fgetc(f2); } while (c1 == c2 && c1 != EOF);
    fclose(f1); fclose(f2);
    remove(p1); remove(p2);
    if (c1 != c2) return fail("threaded text differs");

    kto_counter_free(a); kto_counter_free(b);
    free(ka); free(kb); free(ca); free(cb); free(m1); free(mt); free(bases); free(offsets);
    printf("sanitize_driver ok: %llu reads, %d threads, %llu distinct 21-mers\n", (unsigned long long)n, T,
           (unsigned long long)na);
    return 0;
}
