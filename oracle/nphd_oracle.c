/*
 * nphd_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the exact k-nearest-neighbour search that the reference
 * performs (approximately, through HNSW) at
 *   iscc_search/indexes/usearch/index.py:2036-2037      (ShardedNphdIndex.search, NPHD metric)
 *   iscc_search/indexes/simprint/usearch_core.py:161-165 (ShardedIndex128.search, Hamming metric)
 * The arithmetic itself lives in third-party wheels that are absent from
 * /root/reference (iscc-usearch 0.8.1 -> usearch-iscc 2.24.6 -> simsimd 6.5.16,
 * uv.lock:765-766, :2491-2492, :2284-2285).  The published definition restated here is
 *   NPHD(a, b) = hamming(a[:p], b[:p]) / p,  p = min(len(a), len(b)) in bits
 *   (docs/explanation/similarity-search.md:24-29), score = 1 - NPHD (:31-32),
 * and for the fixed-length simprint tables the raw Hamming bit count
 * (tests/test_usearch_search.py:122-167 pins distances 0, 1, 16 as raw counts).
 *
 * Pinning: tests/golden/kat_hamming.json holds every literal distance known-answer
 * the reference's tests carry for this boundary; the oracle is
 * checked against all of them in tests/test_oracle.py.  Mixed-length NPHD is pinned
 * only by the prose formula ("NPHD mixed-length parity: unpinned by reference tests").
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Order of results (the reference leaves ties unspecified, usearch/index.py:836):
 * ascending (distance, key) where distance is compared as the exact rational h/p
 * (cross-multiplication, no floats) and key is the unsigned 64- or 128-bit integer.
 *
 * Data layout handed in by the caller (identical to the product's C-ABI, include/isccsearch.h):
 *   code_words[n * max_words]  -- each code's bytes packed big-endian into 64-bit words
 *                                 (byte 0 is the most significant byte of word 0), zero padded
 *   nbytes[n]                  -- code length in bytes (1..32); NULL => every row is max_words*8
 *                                 bytes, or `fixed_nbytes` when that is non-zero
 *   keys[n * key_words]        -- key_words = 1 (u64) or 2 (hi, lo of a 128-bit big-endian key)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    uint32_t h;      /* hamming distance over the common prefix            */
    uint32_t p;      /* prefix length in bits                              */
    uint64_t khi;    /* key, high word (0 for 64-bit keys)                 */
    uint64_t klo;    /* key, low word                                      */
} hit_t;

/* exact order on (h/p, key): returns <0, 0, >0 */
static inline int hit_cmp(const hit_t* a, const hit_t* b) {
    uint64_t l = (uint64_t)a->h * b->p, r = (uint64_t)b->h * a->p;
    if (l != r) return l < r ? -1 : 1;
    if (a->khi != b->khi) return a->khi < b->khi ? -1 : 1;
    if (a->klo != b->klo) return a->klo < b->klo ? -1 : 1;
    return 0;
}

/* hamming distance over the first `pbytes` bytes of two big-endian packed codes */
static inline uint32_t prefix_hamming(const uint64_t* a, const uint64_t* b, uint32_t pbytes) {
    uint32_t full = pbytes >> 3, rem = pbytes & 7, h = 0;
    for (uint32_t i = 0; i < full; ++i) h += (uint32_t)__builtin_popcountll(a[i] ^ b[i]);
    if (rem) {
        uint64_t mask = ~0ULL << (8 * (8 - rem));
        h += (uint32_t)__builtin_popcountll((a[full] ^ b[full]) & mask);
    }
    return h;
}

/* max-heap of the k best hits seen so far (root = worst kept) */
static void heap_sift_down(hit_t* hp, size_t n, size_t i) {
    for (;;) {
        size_t l = 2 * i + 1, r = l + 1, m = i;
        if (l < n && hit_cmp(&hp[l], &hp[m]) > 0) m = l;
        if (r < n && hit_cmp(&hp[r], &hp[m]) > 0) m = r;
        if (m == i) return;
        hit_t t = hp[i]; hp[i] = hp[m]; hp[m] = t;
        i = m;
    }
}
static void heap_sift_up(hit_t* hp, size_t i) {
    while (i) {
        size_t parent = (i - 1) / 2;
        if (hit_cmp(&hp[i], &hp[parent]) <= 0) return;
        hit_t t = hp[i]; hp[i] = hp[parent]; hp[parent] = t;
        i = parent;
    }
}
static int hit_qsort_cmp(const void* a, const void* b) { return hit_cmp((const hit_t*)a, (const hit_t*)b); }

/* distances of one tile of fixed-length rows (whole words) to one query; the compiler vectorises this loop */
#define TILE_ROWS 2048
#define TILE_BODY                                                                                        \
    if (W == 1) {                                                                                        \
        const uint64_t q0 = q[0];                                                                        \
        for (uint32_t i = 0; i < n; ++i) out[i] = (uint32_t)__builtin_popcountll(rows[i] ^ q0);          \
    } else if (W == 2) {                                                                                 \
        const uint64_t q0 = q[0], q1 = q[1];                                                             \
        for (uint32_t i = 0; i < n; ++i)                                                                 \
            out[i] = (uint32_t)(__builtin_popcountll(rows[2 * i] ^ q0) + __builtin_popcountll(rows[2 * i + 1] ^ q1)); \
    } else {                                                                                             \
        for (uint32_t i = 0; i < n; ++i) {                                                               \
            uint32_t h = 0;                                                                              \
            for (int w = 0; w < W; ++w) h += (uint32_t)__builtin_popcountll(rows[(uint64_t)i * W + w] ^ q[w]); \
            out[i] = h;                                                                                  \
        }                                                                                                \
    }
static void tile_hamming_base(const uint64_t* rows, int W, const uint64_t* q, uint32_t n, uint32_t* out) { TILE_BODY }
#if defined(__x86_64__) && defined(__GNUC__)
__attribute__((target("avx512f,avx512vl,avx512bw,avx512vpopcntdq")))
static void tile_hamming_vpopcnt(const uint64_t* rows, int W, const uint64_t* q, uint32_t n, uint32_t* out) { TILE_BODY }
static int have_vpopcnt(void) {
    static int cached = -1;
    if (cached < 0) cached = __builtin_cpu_supports("avx512vpopcntdq") && __builtin_cpu_supports("avx512vl") ? 1 : 0;
    return cached;
}
#else
static int have_vpopcnt(void) { return 0; }
#define tile_hamming_vpopcnt tile_hamming_base
#endif
/* run-time dispatch: AVX-512 VPOPCNTQ where the host has it, plain popcnt elsewhere */
static inline void tile_hamming(const uint64_t* rows, int W, const uint64_t* q, uint32_t n, uint32_t* out) {
    if (have_vpopcnt()) tile_hamming_vpopcnt(rows, W, q, n, out);
    else tile_hamming_base(rows, W, q, n, out);
}
int oracle_uses_vpopcnt(void) { return have_vpopcnt(); }

static inline void heap_offer(hit_t* heap, uint32_t* cnt, uint32_t k, const hit_t* c) {
    if (*cnt < k) {
        heap[*cnt] = *c;
        heap_sift_up(heap, *cnt);
        ++*cnt;
    } else if (hit_cmp(c, &heap[0]) < 0) {
        heap[0] = *c;
        heap_sift_down(heap, *cnt, 0);
    }
}

/*
 * Exact top-k for nq queries.  metric: 0 = fixed-length Hamming, 1 = NPHD.
 * Outputs (caller allocated): out_keys[nq*k*key_words], out_hamming[nq*k],
 * out_prefix_bits[nq*k] (the p the distance is normalised by; for metric 0 the table's
 * bit length), out_count[nq] = min(k, n).  Returns 0, or -1 on bad arguments.
 * threads <= 0 uses every core OpenMP sees.
 *
 * Parallelisation: the rows are split across threads; every thread walks its rows in cache-sized
 * tiles and scores each tile against ALL queries (a tile is read from DRAM once), keeping one
 * k-heap per query; the per-thread heaps are merged at the end.
 */
int oracle_topk(int metric, int key_words, int max_words, uint32_t fixed_nbytes,
                uint64_t n, const uint64_t* keys, const uint64_t* code_words, const uint8_t* nbytes,
                uint32_t nq, const uint64_t* q_words, const uint8_t* q_nbytes, uint32_t k,
                uint64_t* out_keys, uint32_t* out_hamming, uint16_t* out_prefix_bits,
                uint32_t* out_count, int threads) {
    if (k < 1 || max_words < 1 || max_words > 4 || (key_words != 1 && key_words != 2)) return -1;
    int nt = 1;
#ifdef _OPENMP
    nt = threads > 0 ? threads : omp_get_max_threads();
#else
    (void)threads;
#endif
    if (nq == 0) return 0;
    /* keep the per-thread heaps within ~256 MB */
    while (nt > 1 && (uint64_t)nt * nq * k * sizeof(hit_t) > (256ull << 20)) nt /= 2;
    if ((uint64_t)nt > n / TILE_ROWS + 1) nt = (int)(n / TILE_ROWS + 1);
    hit_t* heaps = (hit_t*)malloc(sizeof(hit_t) * (size_t)k * nq * (size_t)nt);
    uint32_t* cnts = (uint32_t*)calloc((size_t)nq * (size_t)nt, sizeof(uint32_t));
    if (!heaps || !cnts) { free(heaps); free(cnts); return -1; }
    const uint32_t table_bytes = fixed_nbytes ? fixed_nbytes : (uint32_t)max_words * 8;
    /* fast path: fixed-length rows made of whole words */
    const int whole_words = (!nbytes) && (table_bytes % 8 == 0) && (!metric || !q_nbytes);
    const int W = (int)(table_bytes / 8);

#pragma omp parallel num_threads(nt)
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const uint64_t lo = n * (uint64_t)tid / (uint64_t)nt, hi = n * (uint64_t)(tid + 1) / (uint64_t)nt;
        hit_t* my = heaps + (size_t)tid * nq * k;
        uint32_t* mycnt = cnts + (size_t)tid * nq;
        uint32_t dist[TILE_ROWS];
        for (uint64_t t0 = lo; t0 < hi; t0 += TILE_ROWS) {
            const uint32_t m = (uint32_t)((hi - t0) < TILE_ROWS ? (hi - t0) : TILE_ROWS);
            for (uint32_t qi = 0; qi < nq; ++qi) {
                hit_t* heap = my + (size_t)qi * k;
                uint32_t* cnt = &mycnt[qi];
                const uint64_t* q = q_words + (uint64_t)qi * max_words;
                if (whole_words && W == max_words) {
                    tile_hamming(code_words + t0 * (uint64_t)max_words, W, q, m, dist);
                    for (uint32_t i = 0; i < m; ++i) {
                        /* cheap reject: strictly worse than the worst kept hit */
                        if (*cnt == k && dist[i] > heap[0].h) continue;
                        hit_t c;
                        c.h = dist[i];
                        c.p = metric ? table_bytes * 8 : 1;
                        c.khi = key_words == 2 ? keys[2 * (t0 + i)] : 0;
                        c.klo = key_words == 2 ? keys[2 * (t0 + i) + 1] : keys[t0 + i];
                        heap_offer(heap, cnt, k, &c);
                    }
                } else {
                    const uint32_t qb = q_nbytes ? q_nbytes[qi] : table_bytes;
                    for (uint32_t i = 0; i < m; ++i) {
                        const uint64_t r = t0 + i;
                        const uint32_t rb = nbytes ? nbytes[r] : table_bytes;
                        const uint32_t pb = metric ? (rb < qb ? rb : qb) : rb;
                        hit_t c;
                        c.h = prefix_hamming(code_words + r * (uint64_t)max_words, q, pb);
                        c.p = metric ? pb * 8 : 1;
                        c.khi = key_words == 2 ? keys[2 * r] : 0;
                        c.klo = key_words == 2 ? keys[2 * r + 1] : keys[r];
                        heap_offer(heap, cnt, k, &c);
                    }
                }
            }
        }
    }

    /* merge the per-thread heaps */
    int failed = 0;
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt)
    for (int64_t qi = 0; qi < (int64_t)nq; ++qi) {
        size_t total = 0;
        for (int t = 0; t < nt; ++t) total += cnts[(size_t)t * nq + qi];
        hit_t* all = (hit_t*)malloc(sizeof(hit_t) * (total ? total : 1));
        if (!all) { failed = 1; continue; }
        size_t m = 0;
        for (int t = 0; t < nt; ++t) {
            const uint32_t c = cnts[(size_t)t * nq + qi];
            memcpy(all + m, heaps + ((size_t)t * nq + qi) * k, sizeof(hit_t) * c);
            m += c;
        }
        qsort(all, m, sizeof(hit_t), hit_qsort_cmp);
        const uint32_t cnt = m < k ? (uint32_t)m : k;
        for (uint32_t i = 0; i < cnt; ++i) {
            const uint64_t o = (uint64_t)qi * k + i;
            if (key_words == 2) { out_keys[2 * o] = all[i].khi; out_keys[2 * o + 1] = all[i].klo; }
            else out_keys[o] = all[i].klo;
            out_hamming[o] = all[i].h;
            out_prefix_bits[o] = (uint16_t)(metric ? all[i].p : table_bytes * 8);
        }
        out_count[qi] = cnt;
        free(all);
    }
    free(heaps);
    free(cnts);
    return failed ? -1 : 0;
}

/* splitmix64: the synthetic-data generator of SURVEY.md section 8d, restated for host-side checks */
uint64_t oracle_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* fill out[i] = splitmix64(seed + stride*(first+i) + lane) for i in [0, n) */
void oracle_fill_splitmix64(uint64_t* out, uint64_t n, uint64_t seed, uint64_t first, uint64_t stride, uint64_t lane) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) out[i] = oracle_splitmix64(seed + stride * (first + (uint64_t)i) + lane);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
