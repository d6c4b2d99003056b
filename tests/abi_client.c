/* abi_client.c -- a plain C caller of the C-ABI (include/isccsearch.h), the way a non-Python host (cgo / JNI / N-API) would
 * bind it: no ctypes, no numpy, no torch.  tests/test_gpu_abi_client.py compiles it with gcc against libisccsearch_hip.so,
 * runs it on the GPU and compares what it prints with the oracle's answer for the same rows and queries.
 *
 *   abi_client <rows> <queries> <k>
 * Rows: code(i) = splitmix64(seed + 4 i), key(i) = 1000 + i, added from host arrays through isccsearch_add, after the same
 * number of rows generated on the device by isccsearch_add_synthetic with another seed.  Queries: query j is row (7919 j mod rows)
 * with its lowest j mod 5 bits flipped.  Prints, per query: "q <j> <count> : <key>:<hamming> ..." and, for the range-limited
 * search with max_hamming = 2, "w <j> <count> : ...".
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "isccsearch.h"

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

#define CHECK(call)                                                                              \
    do {                                                                                         \
        int rc_ = (call);                                                                        \
        if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, isccsearch_last_error()); return 2; } \
    } while (0)

int main(int argc, char** argv) {
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 100000;
    const uint32_t nq = argc > 2 ? (uint32_t)atoi(argv[2]) : 8;
    const uint32_t k = argc > 3 ? (uint32_t)atoi(argv[3]) : 5;
    const uint64_t seed_a = 0x1511CC00ull, seed_b = 0x0BADC0DEull;

    isccsearch_handle* h = NULL;
    CHECK(isccsearch_create(0, &h));
    uint32_t table = 0;
    CHECK(isccsearch_table_open(h, ISCCSEARCH_METRIC_HAMMING, 1, 8, &table));

    uint64_t* keys = malloc(n * sizeof *keys);
    uint64_t* words = malloc(n * sizeof *words);
    for (uint64_t i = 0; i < n; ++i) { keys[i] = 1000 + i; words[i] = splitmix64(seed_a + 4 * i); }
    /* device-generated rows first (they bypass the host key index), then the checked add, which builds that index over both */
    CHECK(isccsearch_add_synthetic(h, table, 8, n, seed_b, 0, 1000 + n));      /* keys 1000 + n + i, codes splitmix64(seed_b + 4 i) */
    CHECK(isccsearch_add(h, table, n, keys, words, NULL, 0));
    if (isccsearch_size(h, table) != 2 * n) { fprintf(stderr, "size %llu\n", (unsigned long long)isccsearch_size(h, table)); return 3; }

    uint64_t* q = malloc(nq * sizeof *q);
    for (uint32_t j = 0; j < nq; ++j) q[j] = words[(7919ull * j) % n] ^ ((1ull << (j % 5)) - 1);

    uint64_t* out_keys = malloc((size_t)nq * k * sizeof *out_keys);
    uint32_t* out_h = malloc((size_t)nq * k * sizeof *out_h);
    uint16_t* out_p = malloc((size_t)nq * k * sizeof *out_p);
    uint32_t* out_c = malloc(nq * sizeof *out_c);
    CHECK(isccsearch_search(h, table, nq, q, NULL, k, out_keys, out_h, out_p, out_c));
    for (uint32_t j = 0; j < nq; ++j) {
        printf("q %u %u :", j, out_c[j]);
        for (uint32_t i = 0; i < out_c[j]; ++i) printf(" %llu:%u", (unsigned long long)out_keys[(size_t)j * k + i], out_h[(size_t)j * k + i]);
        printf("\n");
    }
    CHECK(isccsearch_search_within(h, table, nq, q, NULL, k, 2, out_keys, out_h, out_p, out_c));
    for (uint32_t j = 0; j < nq; ++j) {
        printf("w %u %u :", j, out_c[j]);
        for (uint32_t i = 0; i < out_c[j]; ++i) printf(" %llu:%u", (unsigned long long)out_keys[(size_t)j * k + i], out_h[(size_t)j * k + i]);
        printf("\n");
    }
    /* remove the first host-added row and look it up again: its exact match must be gone */
    uint64_t removed = 0;
    CHECK(isccsearch_remove(h, table, 1, keys, &removed));
    uint8_t found = 1;
    CHECK(isccsearch_contains(h, table, 1, keys, &found));
    printf("removed %llu found %u size %llu\n", (unsigned long long)removed, found, (unsigned long long)isccsearch_size(h, table));

    isccsearch_stats st;
    CHECK(isccsearch_stats_get(h, &st, 0));
    printf("searches %llu queries %llu\n", (unsigned long long)st.searches, (unsigned long long)st.queries);
    CHECK(isccsearch_table_drop(h, table));
    CHECK(isccsearch_destroy(h));
    free(keys); free(words); free(q); free(out_keys); free(out_h); free(out_p); free(out_c);
    return 0;
}
