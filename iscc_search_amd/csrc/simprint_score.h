// Simprint asset scoring on the device (internal interface between isccsearch.hip and simprint_score.hip).
//
// What UsearchSimprintIndex.search_raw does AFTER its batched neighbour search
// (iscc_search/indexes/simprint/usearch_core.py:171-269): threshold on the distance, best chunk per (asset, query
// simprint) in (query, rank) visiting order, IDF-weighted mean per asset in the reference's order of float64
// additions, order (-score, asset), first `limit`.  Input: the neighbour lists as select_kernel left them in
// device memory -- records [nq][k] ascending (hamming, key), the segment row of every record, counts [nq].
//
//   per batch of <= 1 024 query simprints (queued behind its select_kernel, before the batch's one synchronisation):
//     mark     one block per query: matches = the prefix with hamming <= h_max; first occurrence of every asset in it
//              (LDS hash: asset -> smallest rank) = the best chunk of (asset, query); the query's own document
//              frequency from its hamming-0 prefix
//     offsets  exclusive scan of the per-query best counts (one block)
//     compact  the best entries of the batch, in (query, rank) order, appended to (asset[], entry[])
//   once per request:
//     stable radix sort by asset (rocPRIM) -> every asset's entries adjacent, ascending query
//     score    one thread per asset run, sequential float64 sums in the reference's order (no contraction into fused
//              multiply-adds: -ffp-contract=off; IEEE rounding); IDF and similarity values come from host-computed tables
//              (log() of the host libm = CPython's math.log)
//     stable radix sort by score, descending (assets are ascending already: ties keep ascending asset order)
//     emit     the first `limit` assets (+ their matched chunks) written straight into pinned host memory
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/isccsearch.h"

namespace isksp {

constexpr uint32_t MAX_QUERY_SIMPRINTS = 8192;   // score_kernel keeps one bit per query simprint and thread in LDS (64 threads x 1 KB)
constexpr uint32_t INFO_WORDS = 4;   // per batch, behind counts / flags / k-th distances: {entries appended so far, some query's
                                     // document frequency could not be read off its list, -, -}

// device buffers of one request (owned by the handle, grown on demand)
struct Buffers {
    const isccsearch_record* rec;   // [nq][k]
    const uint32_t* rows;           // [nq][k] segment row of every record
    uint8_t* best;                  // [nq][k] 1 = best chunk of its (asset, query)
    uint32_t* nbest;                // [nq]
    uint32_t* offs;                 // [nq] position of a query's first best entry in the compacted arrays
    uint32_t* freq_q;               // [nq] document frequency of the query simprint itself
    uint32_t* unknown;              // [nq] 1 = freq_q could not be decided from the list (k equal rows, k < dup_limit)
    uint64_t* c_asset[2];           // [nq*k] compacted asset ids (double buffer of the sort)
    uint32_t* c_entry[2];           // [nq*k] entry index q*k + rank
    double* score[2];               // [entries] per sorted entry: the asset's score at its run head, -1 elsewhere
    uint32_t* order[2];             // [entries] sort payload: index of the run head
    uint32_t* matches;              // [entries] matched query simprints of the run starting here
    double* ws;                     // [entries] IDF weight x similarity of every sorted entry
    double* idf_q;                  // [nq] IDF of every query simprint's own document frequency
    uint32_t* n_assets;             // [1]
    void* temp;                     // rocPRIM scratch
    size_t temp_bytes;
};

size_t sort_temp_bytes(size_t entries);

struct BatchArgs {
    uint32_t pos, m, k;             // queries [pos, pos + m) of the request, k neighbours each
    const uint32_t* cnt;            // [m] valid records per query (this batch)
    int h_max;                      // matches have hamming <= h_max (-1: none)
    uint32_t dup_limit;             // 0: frequencies are not wanted (every one is 1)
    uint32_t base;                  // entries appended by the batches before this one
    uint32_t* info;                 // [INFO_WORDS] device, travels to the host with the batch's counts
};
// mark + offsets + compact of one batch, queued on `stream`
hipError_t queue_batch(const Buffers& b, const BatchArgs& a, hipStream_t stream);

struct ScoreArgs {
    uint32_t nq, k, entries, limit;
    const double* sim_tab;          // [bits + 1] device: 1 - h / bits
    const double* idf_tab;          // [dup_limit + 1] device: idf of a document frequency (dup_limit == 0: one entry, idf of 1)
    uint32_t dup_limit;
    const uint32_t* freq_col;       // segment's document-frequency column (dup_limit > 0)
    const uint64_t* col[4];         // segment's code columns (chunk detail)
    uint32_t W;
    // pinned outputs
    isccsearch_simprint_result* out_results;   // [limit]
    isccsearch_simprint_chunk* out_chunks;     // nullable [limit * nq]
    uint64_t* out_chunk_words;                 // nullable [limit * nq * W]
    uint32_t* out_info;                        // [4] {results, assets matched, -, chunks written}
};
// sort by asset -> score -> sort by score -> emit
hipError_t queue_score(Buffers& b, const ScoreArgs& a, hipStream_t stream);

// Hard-boundary requests (search_simprints_exact, lmdb_ops.py:169-301).  Buffers: rec = the collision lists [nd][k] of the DISTINCT
// query simprints, freq_q [nd] their document frequencies, nbest / unknown / offs sized for the GIVEN simprints [ng].
hipError_t exact_prepare(const Buffers& b, const uint32_t* cnt, const uint32_t* d_of_g, uint32_t nd, uint32_t ng, uint32_t k, uint32_t* info, hipStream_t stream);
struct ExactArgs {
    uint32_t nd, ng, k, entries, limit, queried;
    const uint32_t* d_of_g;         // [ng] device: distinct lookup of every given simprint
    double threshold;
    isccsearch_simprint_result* out_results;   // pinned [limit]
    isccsearch_simprint_chunk* out_chunks;     // pinned, nullable [entries]
    uint32_t* out_info;                        // pinned [4]
};
hipError_t queue_exact(Buffers& b, const ExactArgs& a, hipStream_t stream);

}  // namespace isksp
