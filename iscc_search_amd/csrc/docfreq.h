// Document-frequency column of one segment (internal interface between isccsearch.hip and docfreq.hip).
//
// freq[row] = number of DISTINCT assets among the first `dup_limit` rows (ascending key) that hold the same
// code as `row` -- what the reference computes per matched simprint with an LMDB cursor walk
// (count_doc_freq, iscc_search/indexes/simprint/lmdb_ops.py:139-166).  The asset is the first word of a
// 2-word key (ISCC-ID body of a chunk pointer); with 1-word keys every row is its own asset.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

namespace iskdf {

// All pointers are device memory.  col[w][row], keys[row*KW + i].  n < 2^32.  Temporary buffers
// (~44 bytes per row) are allocated for the call and released before it returns; the stream is drained.
// Returns 0, or a negative errno-style code with *err describing it.
int build_freq_column(const uint64_t* const* col, int W, const uint64_t* keys, int KW, uint64_t n,
                      uint32_t dup_limit, uint32_t* freq_out, hipStream_t stream, std::string* err);

}  // namespace iskdf
