/*
 * isccsearch.h -- C-ABI of libisccsearch_hip.so, the MI355X (gfx950) brute-force
 * Hamming / NPHD k-nearest-neighbour engine behind the `hip:///` ISCC index backend.
 *
 * Drop-in boundary.  The reference (iscc/iscc-search, 100 % Python) reaches its hot path through
 * two third-party objects; every entry point below replaces one of their methods at the call
 * sites listed (paths relative to the reference tree):
 *
 *   iscc_usearch.ShardedNphdIndex   (metric = ISCCSEARCH_METRIC_NPHD, 64-bit keys)
 *     ctor    iscc_search/indexes/usearch/index.py:1617-1625   -> isccsearch_table_open
 *     .add    iscc_search/indexes/usearch/index.py:440         -> isccsearch_add
 *     .remove iscc_search/indexes/usearch/index.py:436         -> isccsearch_remove
 *     .search iscc_search/indexes/usearch/index.py:2037        -> isccsearch_search
 *     `in`    iscc_search/indexes/usearch/index.py:560         -> isccsearch_contains
 *     .size   iscc_search/indexes/usearch/index.py:444,1631    -> isccsearch_size
 *     .reset/.close  index.py:1699, :936                       -> isccsearch_table_drop / isccsearch_destroy
 *   iscc_usearch.ShardedIndex128    (metric = ISCCSEARCH_METRIC_HAMMING, 128-bit keys)
 *     ctor    iscc_search/indexes/simprint/usearch_core.py:73-83 -> isccsearch_table_open
 *     .add    iscc_search/indexes/simprint/usearch_core.py:108   -> isccsearch_add
 *     .remove iscc_search/indexes/simprint/usearch_core.py:119   -> isccsearch_remove
 *     .search iscc_search/indexes/simprint/usearch_core.py:165   -> isccsearch_search
 *     .get    iscc_search/indexes/simprint/usearch_core.py:221   -> isccsearch_get
 *     `in`    iscc_search/indexes/simprint/usearch_core.py:135   -> isccsearch_contains
 *   LMDB dupsort simprint table (hard-boundary matching and document frequency; the reference keeps it beside
 *   the ShardedIndex128, here the same device table answers both)
 *     search_simprints_exact  iscc_search/indexes/simprint/lmdb_ops.py:169-249 -> isccsearch_search_within (max_hamming 0)
 *     count_doc_freq          iscc_search/indexes/simprint/lmdb_ops.py:139-166 -> isccsearch_doc_freq (by code),
 *                                                                                 isccsearch_get_freq (by key)
 *   LMDB dupsort INSTANCE table
 *     _search_instance_unit   iscc_search/indexes/usearch/index.py:1957-2022   -> isccsearch_search_within (max_hamming 0)
 *
 * Unlike the reference's HNSW, every search here is EXACT: the k rows with the smallest
 * (distance, key), ascending, where distance is the rational hamming/prefix_bits for NPHD tables
 * and the raw bit count for Hamming tables (the reference leaves tie order unspecified,
 * iscc_search/indexes/usearch/index.py:836; this library defines it).
 *
 * Conventions
 *   - plain pointers and sizes only; the caller allocates every output; the library owns device memory
 *   - return 0 on success, a negative errno-style code otherwise; isccsearch_last_error() returns
 *     a thread-local human-readable message for the last failure on the calling thread
 *   - every entry point takes the handle's mutex: calls from many threads are serialised
 *   - codes are handed over as 64-bit words: the code's bytes packed BIG-ENDIAN (byte 0 is the
 *     most significant byte of word 0), zero padded to max_words = ceil(max_bytes / 8) words
 *   - 128-bit keys are two words (hi, lo) of the big-endian 16-byte key
 *     (iscc_search/indexes/simprint/lmdb_ops.py:30-49: iscc_id_body(8) | offset(4) | size(4))
 */
#ifndef ISCCSEARCH_H
#define ISCCSEARCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISCCSEARCH_METRIC_HAMMING 0 /* fixed length, distance = differing bits                    */
#define ISCCSEARCH_METRIC_NPHD    1 /* variable length, distance = hamming(prefix) / prefix bits  */

#define ISCCSEARCH_MAX_BYTES 32     /* longest code: 256 bits (index.py:135 max_dim)              */
#define ISCCSEARCH_MAX_K     4096   /* largest k per search (reference default limit 100; simprint
                                       oversampling 40 x limit = 4000, usearch_core.py:164)        */

#define ISCCSEARCH_ADD_TRUSTED_UNIQUE 1u /* caller guarantees the keys are new (usearch_core.py:89-92
                                            "skips the expensive contains-check")                   */

typedef struct isccsearch_handle isccsearch_handle;

/* One search result as it lives in device memory (multi-GPU exchange format, 24 bytes). */
typedef struct isccsearch_record {
    uint64_t key_hi;      /* 0 for 64-bit keys */
    uint64_t key_lo;
    uint32_t dist_rank;   /* order-preserving integer for the exact distance (library internal) */
    uint16_t hamming;     /* differing bits over the compared prefix */
    uint16_t prefix_bits; /* bits compared (NPHD denominator; table bit length for Hamming) */
} isccsearch_record;

typedef struct isccsearch_stats {
    uint64_t searches;        /* isccsearch_search* calls served                              */
    uint64_t queries;         /* queries answered                                             */
    uint64_t scan_launches;   /* launches of the collect scan kernel                          */
    uint64_t scan_passes;     /* query groups streamed over a segment by those launches       */
    uint64_t scan_bytes;      /* ALGORITHMIC bytes of those passes: rows * 8 * words compared */
    double   scan_ms;         /* summed device time of those launches (only while profiling)  */
    uint64_t sample_bytes;    /* ALGORITHMIC bytes of the threshold levels (first stretch, read once) */
    uint64_t fallback_queries;/* queries that took the exact full-histogram fallback          */
    uint32_t queries_per_pass;/* T_q: queries held in SGPRs per streaming pass                */
    uint32_t compute_units;   /* CUs of the device                                            */
    uint64_t freq_builds;     /* document-frequency columns (re)built by isccsearch_get_freq  */
    uint64_t mfma_launches;   /* scan launches (single pass, levels, collect) that ran on the matrix cores (FP4) */
    uint64_t mfma_pair_words; /* (row, query, 64-bit word) triples those launches scored: 128 operations each   */
    uint64_t scan_pair_words; /* (row, REAL query, word) triples of the collect launches counted in scan_launches / scan_ms */
    uint64_t scan_mfma_launches; /* how many of scan_launches ran on the matrix cores                          */
    /* the threshold levels (the first stretch of every segment, same scan kernels in MODE_BOTH), timed like the collect launches */
    uint64_t level_launches;
    uint64_t level_pair_words;
    uint64_t level_mfma_launches;
    double   level_ms;
    /* batches whose single self-tightening pass (k <= 512 on the matrix cores) overflowed a candidate list and were answered
       again with threshold levels, which prune after every level, before any query took the per-query exact fallback */
    uint64_t self_retries;
    uint64_t mfma_pack_launches; /* how many of mfma_launches ran the packed form (64-bit codes: two row tiles per accumulator) */
    /* small batches answered by ONE speculative range-limited pass under the previous search's k-th distance / sent on to the
       ordinary path because a query found fewer than k rows within it (or a list overflowed) */
    uint64_t spec_hits, spec_misses;
    /* with option "count_candidates" = 1: candidate-list entries the scan launches appended (summed over queries; the lists of the LAST pipeline run of
       every batch, read back after its synchronisation), and the batches they were counted over */
    uint64_t candidates, candidate_batches;
} isccsearch_stats;

/* Engine lifetime.  One handle drives one GPU (one process per GPU). */
int isccsearch_create(int device_id, isccsearch_handle** out);
int isccsearch_destroy(isccsearch_handle* h);
const char* isccsearch_last_error(void);

/* Options: "mfma" (0|1, default 1: batches of >= "mfma_min_queries" (17) queries over >= "mfma_min_rows" (65 536) rows are
 * scanned on the matrix cores -- bits as FP4 0/+-1, exact f32 sums, csrc/mfma_scan.hip -- instead of XOR + popcount; 64-bit codes,
 * whose packed form of that kernel is cheaper, from "mfma_pack_min_queries" (9) queries -- as do longer codes over segments of at
 * most "mfma_few_rows" (12 Mi) rows (at 25 M rows one pass of 16 on the XOR + popcount kernel is ahead again);
 * "self_tighten" (0|1, default 1): for k <= "self_max_k" (4 096: every k) that scan is ONE pass whose thresholds tighten themselves,
 * bootstrapped from max("self_boot_rows" (65 536), "self_boot_per_k" (1 024) x k) rows and looking at the live thresholds every "self_refresh_steps" (1) steps,
 * instead of threshold levels growing by "mfma_level_growth" (4)); "mfma_pack" (0|1, default 1: 64-bit codes run the packed
 * form of that kernel -- two row tiles per accumulator, v_pk_minimum3_f16 fold -- unless the batch holds an all-zero query);
 * "speculate" (0|1, default 1: a batch of up to "spec_max_queries" (128) queries over a one-segment table is first tried as ONE
 * range-limited pass under the k-th distance the previous such search ended at + 2, and verified: exact either way; larger
 * batches START their single pass under that hint instead of a bootstrap sample's threshold, "self_hint" (0|1, default 1), verified
 * the same way; "device_search_hint" (-1 | 0..256, one-shot): the next isccsearch_search_device_async starts under THIS distance and
 * its lists hold the table's nearest rows within it -- fewer than k if it was too tight, which the caller must check (sharded.py));
 * "candidate_cap" (16 384: floor of the per-query candidate
 * buffer, in entries); "queries_per_pass" (8|16, XOR + popcount kernel), "profile"
 * (0|1: time every collect-scan launch with HIP events, read back through isccsearch_stats_get), "stretch_mb" (XOR + popcount
 * kernel: rows per collect launch, in MB of codes, when several query groups share a launch: they then read the stretch from the
 * caches instead of HBM; default 128, 0 = one streaming pass per group; matrix-core launches whose query chunks share the rows take
 * "mfma_stretch_factor" (3) times that), "fold_tau" (whole 64-bit codes: query groups whose
 * thresholds are all <= this take the folded 3.5-op fast path; default 11, 0 = never); tuning: "blocks_per_cu", "boot_rows",
 * "level_growth", "repick"; "count_candidates" (0|1: after every batch read back how many candidates its scan appended --
 * statistics `candidates` / `candidate_batches`; one more copy and synchronisation per batch, for accounting runs);
 * "tiny_rows" (16 384: a segment of at most this many rows is answered by ONE launch, one block per query -- distances of every row,
 * candidate list, select; 0: never.  Tables the caller sends to the matrix cores by lowering "mfma_min_rows" keep that path);
 * "select_wide_from" (2 048: selects whose LDS sort buffer has at least this many slots -- k > 256 -- run 1 024-thread blocks
 * instead of 256-thread ones when the batch has fewer queries than half the CUs, and whatever the batch from twice that many
 * slots; a larger value than any buffer turns that off).
 * "nontemporal" accepts only 1 (the only variant built). */
int isccsearch_set_option(isccsearch_handle* h, const char* name, int64_t value);
int isccsearch_stats_get(isccsearch_handle* h, isccsearch_stats* out, int reset);

/* Tables.  max_bytes = longest code in bytes (1..32); for Hamming tables every code has exactly
 * max_bytes bytes.  key_words = 1 (u64 keys) or 2 (128-bit keys). */
int isccsearch_table_open(isccsearch_handle* h, int metric, int key_words, int max_bytes, uint32_t* table_id);
int isccsearch_table_drop(isccsearch_handle* h, uint32_t table);
int isccsearch_reserve(isccsearch_handle* h, uint32_t table, int nbytes, uint64_t rows);
uint64_t isccsearch_size(isccsearch_handle* h, uint32_t table);

/* Rows.  keys[n*key_words], code_words[n*max_words], nbytes[n] (NULL for Hamming tables).
 * A key that is already present (or repeated in the batch) fails the whole call with -EEXIST
 * unless ISCCSEARCH_ADD_TRUSTED_UNIQUE is set. */
int isccsearch_add(isccsearch_handle* h, uint32_t table, uint64_t n, const uint64_t* keys,
                   const uint64_t* code_words, const uint8_t* nbytes, uint32_t flags);
int isccsearch_remove(isccsearch_handle* h, uint32_t table, uint64_t n, const uint64_t* keys, uint64_t* n_removed);
int isccsearch_contains(isccsearch_handle* h, uint32_t table, uint64_t n, const uint64_t* keys, uint8_t* out_found);
/* out_nbytes[i] = 0 when keys[i] is absent; out_words[n*max_words] zero padded */
int isccsearch_get(isccsearch_handle* h, uint32_t table, uint64_t n, const uint64_t* keys,
                   uint64_t* out_words, uint8_t* out_nbytes);

/* Snapshot support (the reference's flush/close/rebuild of HNSW shard files has no equivalent: codes live
 * in HBM; iscc_search/indexes/usearch/index.py:883-967).  Segments are addressed by code length in bytes.
 *   segments:    out_rows[ISCCSEARCH_MAX_BYTES + 1], out_rows[b] = rows whose codes are b bytes long
 *   export:      rows [first_row, first_row+n) of one segment: out_keys[n*key_words] and out_cols laid out
 *                COLUMN-major [W][n] (W = ceil(nbytes/8)), i.e. exactly the device layout
 *   add_columns: append n rows of one length from column-major words (no host transposition) */
int isccsearch_segments(isccsearch_handle* h, uint32_t table, uint64_t* out_rows);
int isccsearch_export(isccsearch_handle* h, uint32_t table, int nbytes, uint64_t first_row, uint64_t n,
                      uint64_t* out_keys, uint64_t* out_cols);
int isccsearch_add_columns(isccsearch_handle* h, uint32_t table, int nbytes, uint64_t n, const uint64_t* keys,
                           const uint64_t* cols, uint32_t flags);

/* Bench / test helper: append n rows generated ON THE DEVICE, nbytes long each:
 *   word w of row i = splitmix64(seed + 4*(first_row + i) + w),  key = key_base + first_row + i
 * (SURVEY.md section 8d synthetic generator).  Rows are not entered into the host key index. */
int isccsearch_add_synthetic(isccsearch_handle* h, uint32_t table, int nbytes, uint64_t n,
                             uint64_t seed, uint64_t first_row, uint64_t key_base);

/* Exact k-NN.  q_words[nq*max_words], q_nbytes[nq] (NULL for Hamming tables), 1 <= k <= ISCCSEARCH_MAX_K.
 * Outputs: out_keys[nq*k*key_words], out_hamming[nq*k], out_prefix_bits[nq*k], out_count[nq]
 * (= min(k, rows)); entries past out_count are zero. */
int isccsearch_search(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                      const uint8_t* q_nbytes, uint32_t k,
                      uint64_t* out_keys, uint32_t* out_hamming, uint16_t* out_prefix_bits, uint32_t* out_count);

/* Range-limited exact k-NN: as isccsearch_search, but only rows whose Hamming distance over the compared
 * prefix is <= max_hamming (0..256) are reported -- nearest first, ties by ascending key, at most k per query;
 * out_count[q] may be 0.  One streaming pass at a fixed threshold (no sampling passes).
 * max_hamming = 0 is the hard-boundary collision lookup the reference serves from its LMDB dupsort table
 * (search_simprints_exact, iscc_search/indexes/simprint/lmdb_ops.py:169-249; called with exact=True from
 * iscc_search/indexes/usearch/index.py:1261-1304): every row equal to the query, in ascending key order, which
 * is the byte order LMDB iterates duplicate chunk pointers in; k plays the part of its dup_limit (:197-203).
 * On an NPHD table the same call is the INSTANCE prefix match (usearch/index.py:1957-2022). */
int isccsearch_search_within(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                             const uint8_t* q_nbytes, uint32_t k, uint32_t max_hamming,
                             uint64_t* out_keys, uint32_t* out_hamming, uint16_t* out_prefix_bits, uint32_t* out_count);

/* Several searches in one call, one device synchronisation: what UsearchIndex.search_assets issues as a sequence of
 * per-unit searches (similarity units iscc_search/indexes/usearch/index.py:786-806 -> :2037, INSTANCE unit :791-797
 * -> :1957-2022) for ONE request.  Request i is isccsearch_search (max_hamming < 0) or isccsearch_search_within
 * (max_hamming >= 0) on its own table with its own k and outputs; reqs[i].status receives its result code.  Returns 0
 * or the first request's error code (the others still ran). */
typedef struct isccsearch_request {
    uint32_t table, nq, k;
    int32_t max_hamming;
    const uint64_t* q_words;      /* [nq*max_words] */
    const uint8_t* q_nbytes;      /* [nq], NULL for Hamming tables */
    uint64_t* out_keys;           /* [nq*k*key_words] */
    uint32_t* out_hamming;        /* [nq*k] */
    uint16_t* out_prefix_bits;    /* [nq*k] */
    uint32_t* out_count;          /* [nq] */
    int32_t status;               /* out */
} isccsearch_request;
int isccsearch_search_many(isccsearch_handle* h, uint32_t n, isccsearch_request* reqs);

/* Document frequency of nq codes: out_freq[q] = number of DISTINCT assets among the first dup_limit rows
 * (ascending key) that equal code q.  The asset is the first key word of a 2-word key (the ISCC-ID body of a
 * chunk pointer, lmdb_ops.py:30-49); with 1-word keys every row is its own asset.
 * Replaces count_doc_freq (lmdb_ops.py:139-166, called per matched simprint from usearch/index.py:1395-1403). */
int isccsearch_doc_freq(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                        const uint8_t* q_nbytes, uint32_t dup_limit, uint32_t* out_freq);
/* The same, and out_collisions[q] = how many rows equal to code q were looked at (<= dup_limit).  What a table SHARDED by
 * asset needs (sharded_engine.py: an asset's chunks share a rank): when the collisions of all shards together stay within
 * dup_limit -- the normal case -- the shards' distinct-asset counts simply add; only otherwise is the merged list needed.
 * Same reference call site: count_doc_freq, lmdb_ops.py:139-166. */
int isccsearch_doc_freq_counted(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                                const uint8_t* q_nbytes, uint32_t dup_limit, uint32_t* out_freq, uint32_t* out_collisions);

/* Document frequency of STORED codes, by key: out_freq[i] = frequency (as isccsearch_doc_freq defines it) of
 * the code stored under keys[i], 0 when the key is absent.  Served from a per-segment frequency column that is
 * built on the device by sorting the rows by (code, asset) and is rebuilt lazily after rows were added or
 * removed -- so the per-match lookups of the approximate simprint path (doc_freq_fn called once per matched
 * chunk, usearch/index.py:1395-1403 -> usearch_core.py:201-236) cost one gather instead of one scan each.
 * Hamming (fixed-length) tables only. */
int isccsearch_get_freq(isccsearch_handle* h, uint32_t table, uint64_t n, const uint64_t* keys,
                        uint32_t dup_limit, uint32_t* out_freq);

/* Simprint search WITH its asset scoring, on the device: what UsearchSimprintIndex.search_raw does in one call
 * (iscc_search/indexes/simprint/usearch_core.py:137-269) -- the oversampled batched neighbour search (:161-165, `count` =
 * limit x oversampling neighbours per query simprint; max_hamming >= 0 lists the rows within that radius instead, see
 * isccsearch_search_within), then on the neighbour lists as they lie in device memory: the match threshold on
 * score = 1 - distance / ndim (:182-184), the best chunk per (asset, query simprint) in the reference's visiting order
 * (:175-196; the asset is the first word of the 128-bit chunk-pointer key), the IDF-weighted mean per asset in the
 * reference's ORDER of float64 additions (:215-236, idf = log(1 + total_assets / (1 + freq)), lmdb_ops.py:67-81, with log()
 * of the host libm), order (-score, asset) and the cut to `limit` (:268-269).  Only the <= limit winners (and, when
 * out_chunks is given, their matched chunks: `detailed`, :238-255) leave the device.
 *   dup_limit > 0   document frequencies are the device's: of a matched STORED simprint from the table's frequency column
 *                   (isccsearch_get_freq), of an unmatched QUERY simprint the distinct assets among its first dup_limit
 *                   collisions (isccsearch_doc_freq) -- what the reference's doc_freq_fn computes with an LMDB cursor walk
 *                   (usearch/index.py:1395-1403, lmdb_ops.py:139-166)
 *   dup_limit == 0  every frequency is 1 (the reference's doc_freq_fn = None, :204-211)
 * 128-bit-key Hamming tables only, nq <= ISCCSEARCH_MAX_SCORED_SIMPRINTS.  out_results[limit]; out_chunks[limit * nq] and out_chunk_words[limit * nq * max_words]
 * (both or neither; chunks of result r are out_chunks[first_chunk .. first_chunk + matches), ascending query index, words =
 * the STORED simprint); out_info[4] = {results written, assets matched, longest neighbour list (what a caller that asked for
 * a radius compares with `count` to see a list that filled the cap), chunks written}. */
#define ISCCSEARCH_MAX_SCORED_SIMPRINTS 8192
typedef struct isccsearch_simprint_result {
    uint64_t asset;        /* first key word: ISCC-ID body */
    double   score;
    uint32_t matches;      /* query simprints matched by this asset */
    uint32_t first_chunk;  /* index of its first entry in out_chunks */
} isccsearch_simprint_result;
typedef struct isccsearch_simprint_chunk {
    uint64_t key_lo;       /* second key word: offset (high 32 bits) | size (low 32 bits) */
    uint32_t query;        /* index of the query simprint */
    uint32_t hamming;      /* differing bits: score = 1 - hamming / ndim */
    uint32_t freq;         /* document frequency of the stored simprint (1 when dup_limit == 0) */
    uint32_t reserved;
} isccsearch_simprint_chunk;
int isccsearch_simprint_score(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                              uint32_t count, int32_t max_hamming, double threshold, uint32_t limit,
                              int64_t total_assets, uint32_t dup_limit,
                              isccsearch_simprint_result* out_results, isccsearch_simprint_chunk* out_chunks,
                              uint64_t* out_chunk_words, uint32_t* out_info);

/* Hard-boundary simprint search WITH its scoring, on the device: search_simprints_exact (iscc_search/indexes/simprint/lmdb_ops.py:169-301,
 * called with exact=True from iscc_search/indexes/usearch/index.py:1261-1304).  q_words[n_distinct * max_words] are the DISTINCT query
 * simprints; given[n_given] names, for every query simprint of valid length AS GIVEN (repeats included: a repeated simprint is matched
 * again, :197), its index among the distinct ones; `queried` = the number of query simprints the caller was given (the denominator
 * of the coverage, :286).  Per distinct simprint every stored row EQUAL to it is listed in ascending key order, at most dup_limit
 * (:199-210; capped at ISCCSEARCH_MAX_K); on those lists, in device memory: document frequencies (distinct assets per list, :213-215),
 * the matches of every asset in visiting order, coverage x quality (:252-301) with the reference's float64 operations in the
 * reference's order, the threshold (:222-223), order (-score, asset) and the cut to `limit` (:247-248).
 * out_results[limit]; out_chunks (nullable) [sum of list lengths over the given simprints]: chunks of result r are out_chunks[first_chunk
 * .. first_chunk + matches), in visiting order -- `query` = position in given[], hamming 0, freq = the simprint's document frequency;
 * out_info[4] = {results written, assets kept, longest collision list, chunks written}.  128-bit-key Hamming tables only. */
int isccsearch_simprint_exact(isccsearch_handle* h, uint32_t table, uint32_t n_distinct, const uint64_t* q_words,
                              uint32_t n_given, const uint32_t* given, uint32_t queried, uint32_t dup_limit, double threshold, uint32_t limit,
                              isccsearch_simprint_result* out_results, isccsearch_simprint_chunk* out_chunks, uint32_t* out_info);

/* Multi-GPU building blocks (row-range shards, one process per GPU).
 * search_device: same search, results left in caller-provided DEVICE memory
 *   d_records[nq*k] (isccsearch_record), d_counts[nq]; queries must share one byte length.
 *   The call returns after the library's stream has drained, so the buffers can go straight
 *   into an RCCL all-gather on any stream.
 * merge_device: k-way merge of n_lists result sets held in DEVICE memory into host outputs shaped
 *   as for isccsearch_search.  List l has its records [nq][k] at d_records + l*list_stride and its
 *   counts [nq] at d_counts + l*count_stride (strides in bytes), so one all-gathered buffer of
 *   per-rank blocks {records | counts} can be merged in place. */
int isccsearch_search_device(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                             const uint8_t* q_nbytes, uint32_t k, void* d_records, uint32_t* d_counts);
/* search_within_device: the range-limited search (isccsearch_search_within) with device-resident results. */
int isccsearch_search_within_device(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                                    const uint8_t* q_nbytes, uint32_t k, uint32_t max_hamming,
                                    void* d_records, uint32_t* d_counts);
int isccsearch_merge_device(isccsearch_handle* h, uint32_t n_lists, uint32_t nq, uint32_t k, int key_words,
                            const void* d_records, const void* d_counts, uint64_t list_stride, uint64_t count_stride,
                            uint64_t* out_keys, uint32_t* out_hamming, uint16_t* out_prefix_bits, uint32_t* out_count);

/* The same two steps WITHOUT host round-trips in between (one synchronisation per multi-GPU search step instead of four):
 *   search_device_async   enqueues the search on the library's stream and makes `consumer_stream` (a hipStream_t, e.g. the
 *                         stream the all-gather is issued on) wait for it; it does not wait itself.  max_hamming < 0 = plain
 *                         top-k.  A query whose candidate list overflowed cannot take the exact fallback without the host,
 *                         so its count is written as ISCCSEARCH_COUNT_OVERFLOW; merge_kernel propagates the marker and the
 *                         caller re-runs the step through the synchronous entry points (rare; every rank sees the same
 *                         merged counts, so all ranks take the same decision).  Tables with several segments and batches
 *                         beyond 1 024 queries run the synchronous path inside the call.
 *   merge_device_after    as merge_device, ordered after everything queued on `producer_stream` (the stream the gathered
 *                         blocks were produced on); ONE device->host copy and ONE synchronisation.  out_count[q] ==
 *                         ISCCSEARCH_COUNT_OVERFLOW reports the marker. */
#define ISCCSEARCH_COUNT_OVERFLOW 0xFFFFFFFFu
/* The library's stream (a hipStream_t, owned by the handle).  A host that issues the exchange between the two calls ON this
 * stream (torch: torch.cuda.ExternalStream) and passes it as consumer_stream / producer_stream needs no event between the
 * streams: search, exchange and merge are one in-order queue (two cross-queue hand-overs, ~12 and ~21 us per step, dropped). */
void* isccsearch_stream(isccsearch_handle* h);
int isccsearch_search_device_async(isccsearch_handle* h, uint32_t table, uint32_t nq, const uint64_t* q_words,
                                   const uint8_t* q_nbytes, uint32_t k, int32_t max_hamming,
                                   void* d_records, uint32_t* d_counts, void* consumer_stream);
int isccsearch_merge_device_after(isccsearch_handle* h, uint32_t n_lists, uint32_t nq, uint32_t k, int key_words,
                                  const void* d_records, const void* d_counts, uint64_t list_stride, uint64_t count_stride,
                                  void* producer_stream,
                                  uint64_t* out_keys, uint32_t* out_hamming, uint16_t* out_prefix_bits, uint32_t* out_count);

/* Several merge_device_after calls behind ONE synchronisation (the per-unit searches of one request on a sharded index share one
 * all-gather: their merges are queued back to back and read after a single synchronisation).  Every request is shaped as the
 * arguments of isccsearch_merge_device_after; all results together must fit the directly written result block (1 MB), else -E2BIG. */
typedef struct isccsearch_merge_request {
    uint32_t n_lists, nq, k;
    int32_t key_words;
    const void* d_records;
    const void* d_counts;
    uint64_t list_stride, count_stride;
    uint64_t* out_keys;
    uint32_t* out_hamming;
    uint16_t* out_prefix_bits;
    uint32_t* out_count;
} isccsearch_merge_request;
int isccsearch_merge_many_after(isccsearch_handle* h, uint32_t n, isccsearch_merge_request* reqs, void* producer_stream);

#ifdef __cplusplus
}
#endif
#endif /* ISCCSEARCH_H */
