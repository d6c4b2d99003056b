/* abi_client.c -- a plain C caller of the C-ABI (include/isccsearch.h), the way a non-Python host (cgo / JNI / N-API) would
 * bind it: no ctypes, no numpy, no torch.  tests/test_gpu_abi_client.py compiles it with gcc against libisccsearch_hip.so,
 * runs it on the GPU and compares what it prints with the oracle's answer for the same rows and queries.
 *
 *   abi_client <rows> <queries> <k>
 * Rows: code(i) = splitmix64(seed + 4 i), key(i) = 1000 + i, added from host arrays through isccsearch_add, after the same
 * number of rows generated on the device by isccsearch_add_synthetic with another seed.  Queries: query j is row (7919 j mod rows)
 * with its lowest j mod 5 bits flipped.  Prints, per query: "q <j> <count> : <key>:<hamming> ..." and, for the range-limited
 * search with max_hamming = 2, "w <j> <count> : ...".
 *
 *   abi_client simprint <assets> <chunks> <queries> <limit>
 * A table with 128-bit keys (asset a = 1 .. assets, chunk c: key = (a, (10 c) << 32 | 10 + c)) of 64-bit simprints drawn from a
 * pool of 50 -- simprint(a, c) = pool(h mod 50) with (h >> 32) mod 4 bits flipped from bit (h >> 40) mod 60 on, h =
 * splitmix64(77 + 131 a + c), pool(i) = splitmix64(0x51 + i) -- and one isccsearch_simprint_score call for the query simprints
 * pool(j mod 50) with their lowest j mod 3 bits flipped (count = 4 limit, threshold 0.8, total_assets = assets, dup_limit 1000).
 * Prints per result "r <asset> <score %.17g> <matches> : <query>:<key_lo>:<hamming>:<freq>:<stored word> ...".
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

static int simprint_mode(int argc, char** argv) {
    const uint64_t assets = argc > 2 ? strtoull(argv[2], NULL, 10) : 300;
    const uint64_t chunks = argc > 3 ? strtoull(argv[3], NULL, 10) : 6;
    const uint32_t nq = argc > 4 ? (uint32_t)atoi(argv[4]) : 20;
    const uint32_t limit = argc > 5 ? (uint32_t)atoi(argv[5]) : 10;
    const uint64_t n = assets * chunks;

    isccsearch_handle* h = NULL;
    CHECK(isccsearch_create(0, &h));
    uint32_t table = 0;
    CHECK(isccsearch_table_open(h, ISCCSEARCH_METRIC_HAMMING, 2, 8, &table));
    uint64_t* keys = malloc(2 * n * sizeof *keys);
    uint64_t* words = malloc(n * sizeof *words);
    for (uint64_t a = 1, i = 0; a <= assets; ++a)
        for (uint64_t c = 0; c < chunks; ++c, ++i) {
            const uint64_t hh = splitmix64(77 + 131 * a + c);
            keys[2 * i] = a;
            keys[2 * i + 1] = ((10 * c) << 32) | (10 + c);
            words[i] = splitmix64(0x51 + hh % 50) ^ ((((uint64_t)1 << ((hh >> 32) % 4)) - 1) << ((hh >> 40) % 60));
        }
    CHECK(isccsearch_add(h, table, n, keys, words, NULL, 0));

    uint64_t* q = malloc(nq * sizeof *q);
    for (uint32_t j = 0; j < nq; ++j) q[j] = splitmix64(0x51 + j % 50) ^ (((uint64_t)1 << (j % 3)) - 1);
    isccsearch_simprint_result* res = malloc(limit * sizeof *res);
    isccsearch_simprint_chunk* ch = malloc((size_t)limit * nq * sizeof *ch);
    uint64_t* cw = malloc((size_t)limit * nq * sizeof *cw);
    uint32_t info[4] = {0, 0, 0, 0};
    CHECK(isccsearch_simprint_score(h, table, nq, q, 4 * limit, -1, 0.8, limit, (int64_t)assets, 1000, res, ch, cw, info));
    printf("info %u %u %u %u\n", info[0], info[1], info[2], info[3]);
    for (uint32_t r = 0; r < info[0]; ++r) {
        printf("r %llu %.17g %u :", (unsigned long long)res[r].asset, res[r].score, res[r].matches);
        for (uint32_t i = res[r].first_chunk; i < res[r].first_chunk + res[r].matches; ++i)
            printf(" %u:%llu:%u:%u:%llu", ch[i].query, (unsigned long long)ch[i].key_lo, ch[i].hamming, ch[i].freq, (unsigned long long)cw[i]);
        printf("\n");
    }
    CHECK(isccsearch_table_drop(h, table));
    CHECK(isccsearch_destroy(h));
    free(keys); free(words); free(q); free(res); free(ch); free(cw);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "simprint") == 0) return simprint_mode(argc, argv);
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
