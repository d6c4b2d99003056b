// Simprint asset scoring on the device: see simprint_score.h for the pipeline and the reference lines it restates
// (iscc_search/indexes/simprint/usearch_core.py:171-269).  Built for gfx950 only.
#include <cstring>   // rocPRIM's texture iterator calls memset without including it

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "simprint_score.h"

namespace isksp {
namespace {

constexpr int BLOCK = 256;
constexpr uint64_t EMPTY = ~0ULL;
constexpr uint32_t COUNT_OVERFLOW = ISCCSEARCH_COUNT_OVERFLOW;

// block-wide exclusive prefix of one value per thread (4 waves); `total` = sum over the block
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* wtot /*[4] shared*/, uint32_t& total) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = v;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = __shfl_up(incl, off);
        if (lane >= (uint32_t)off) incl += o;
    }
    __syncthreads();                 // (wtot may still be read from the previous use)
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
    total = 0;
    for (uint32_t w = 0; w < 4; ++w) { if (w < wave) before += wtot[w]; total += wtot[w]; }
    return before + incl - v;
}

// ---------------------------------------------------------------------------------------------
// mark_kernel: one block per query simprint of the batch.
//   matches   = the prefix of the list with hamming <= h_max (the list ascends by (hamming, key)): the reference's threshold
//               on score = 1 - d / ndim (usearch_core.py:182-184), restated on the integer by the host
//   best      = first occurrence of an asset (key_hi) among the matches = the chunk the reference keeps for (asset, query)
//               (:191-196: the first one seen wins, a later one only with a strictly better score -- never in an ascending list)
//   freq_q    = document frequency of the query simprint itself: distinct assets among the first dup_limit rows EQUAL to it
//               = among the hamming-0 prefix of its list (ascending key, so an asset's rows are adjacent).  Undecidable only when
//               the whole list of k rows is at distance 0 and k < dup_limit: flagged, the host asks isccsearch_doc_freq.
// dynamic LDS: keys[S] u64 | idx[S] u32, S = 2^log2s >= 2 k
// ---------------------------------------------------------------------------------------------
struct MarkParams {
    const isccsearch_record* rec;   // batch base
    const uint32_t* cnt;
    uint8_t* best;                  // batch base
    uint32_t* nbest;                // batch base
    uint32_t* freq_q;               // batch base
    uint32_t* unknown;              // batch base
    uint32_t* info;
    uint32_t k;
    int h_max;
    uint32_t dup_limit, log2s;
};
__device__ __forceinline__ uint32_t slot_of(uint64_t a, uint32_t log2s) { return (uint32_t)((a * 0x9E3779B97F4A7C15ULL) >> (64 - log2s)); }

__global__ __launch_bounds__(BLOCK) void mark_kernel(const MarkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_nm, s_n0, s_ones, s_best, s_dist;
    const uint32_t S = 1u << p.log2s, mask = S - 1;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    uint32_t* idx = reinterpret_cast<uint32_t*>(smem + (size_t)S * 8);
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const uint32_t c = p.cnt[q];
    const uint32_t n = c == COUNT_OVERFLOW ? 0 : (c < p.k ? c : p.k);
    const isccsearch_record* r = p.rec + (uint64_t)q * p.k;
    uint8_t* best = p.best + (uint64_t)q * p.k;
    for (uint32_t i = tid; i < S; i += BLOCK) { keys[i] = EMPTY; idx[i] = 0xFFFFFFFFu; }
    if (tid == 0) { s_nm = 0; s_n0 = 0; s_ones = 0xFFFFFFFFu; s_best = 0; s_dist = 0; }
    __syncthreads();
    for (uint32_t i = tid; i < n; i += BLOCK) {
        const int h = r[i].hamming;
        const int hn = i + 1 < n ? (int)r[i + 1].hamming : 0x10000;
        if (h <= p.h_max && hn > p.h_max) s_nm = i + 1;
        if (h == 0 && hn != 0) s_n0 = i + 1;
    }
    __syncthreads();
    const uint32_t nm = s_nm, n0 = s_n0;
    for (uint32_t i = tid; i < nm; i += BLOCK) {
        const uint64_t a = r[i].key_hi;
        if (a == EMPTY) { atomicMin(&s_ones, i); continue; }
        uint32_t s = slot_of(a, p.log2s);
        for (;;) {
            const unsigned long long old = atomicCAS(&keys[s], (unsigned long long)EMPTY, (unsigned long long)a);
            if (old == EMPTY || old == a) { atomicMin(&idx[s], i); break; }
            s = (s + 1) & mask;
        }
    }
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t i = tid; i < p.k; i += BLOCK) {
        uint8_t b = 0;
        if (i < nm) {
            const uint64_t a = r[i].key_hi;
            if (a == EMPTY) b = s_ones == i;
            else {
                uint32_t s = slot_of(a, p.log2s);
                while (keys[s] != a) s = (s + 1) & mask;
                b = idx[s] == i;
            }
        }
        best[i] = b;
        mine += b;
    }
    // the query's own document frequency
    const uint32_t m0 = p.dup_limit ? (n0 < p.dup_limit ? n0 : p.dup_limit) : 0;
    uint32_t dist = 0;
    for (uint32_t i = tid; i < m0; i += BLOCK) dist += (i == 0 || r[i].key_hi != r[i - 1].key_hi) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) { mine += __shfl_down(mine, off); dist += __shfl_down(dist, off); }
    if ((tid & 63) == 0) { if (mine) atomicAdd(&s_best, mine); if (dist) atomicAdd(&s_dist, dist); }
    __syncthreads();
    if (tid == 0) {
        p.nbest[q] = s_best;
        p.freq_q[q] = p.dup_limit ? s_dist : 1u;
        const uint32_t unk = (p.dup_limit && n0 == n && n == p.k && p.k < p.dup_limit) ? 1u : 0u;
        p.unknown[q] = unk;
    }
}

// offs[q] = base + exclusive prefix of nbest over the batch (m <= 1 024: four queries per thread); info = {base + total, some
// query's frequency is unknown, 0, 0}
__global__ __launch_bounds__(BLOCK) void offsets_kernel(const uint32_t* nbest, const uint32_t* unknown, uint32_t* offs, uint32_t m, uint32_t base, uint32_t* info) {
    __shared__ uint32_t wtot[4], wunk[4];
    const uint32_t tid = threadIdx.x;
    uint32_t running = base, unk = 0;
    for (uint32_t c0 = 0; c0 < m; c0 += 4 * BLOCK) {
        uint32_t v[4], sum = 0;
        for (uint32_t j = 0; j < 4; ++j) { const uint32_t q = c0 + 4 * tid + j; v[j] = q < m ? nbest[q] : 0; sum += v[j]; unk |= q < m ? unknown[q] : 0; }
        uint32_t total;
        uint32_t at = running + block_exclusive(sum, wtot, total);
        for (uint32_t j = 0; j < 4; ++j) { const uint32_t q = c0 + 4 * tid + j; if (q < m) offs[q] = at; at += v[j]; }
        running += total;
    }
    const unsigned long long any = __ballot(unk != 0);
    if ((tid & 63) == 0) wunk[tid >> 6] = any != 0;
    __syncthreads();
    if (tid == 0) { info[0] = running; info[1] = wunk[0] | wunk[1] | wunk[2] | wunk[3]; info[2] = 0; info[3] = 0; }
}

// the best entries of query q, in rank order, to positions offs[q] ...
struct CompactParams {
    const isccsearch_record* rec;   // batch base
    const uint8_t* best;            // batch base
    const uint32_t* offs;           // batch base
    uint64_t* c_asset;
    uint32_t* c_entry;
    uint32_t k, pos;
};
__global__ __launch_bounds__(BLOCK) void compact_kernel(const CompactParams p) {
    __shared__ uint32_t wtot[4];
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    const isccsearch_record* r = p.rec + (uint64_t)q * p.k;
    const uint8_t* best = p.best + (uint64_t)q * p.k;
    uint32_t running = p.offs[q];
    for (uint32_t c0 = 0; c0 < p.k; c0 += BLOCK) {
        const uint32_t i = c0 + tid;
        const uint32_t f = i < p.k ? best[i] : 0;
        uint32_t total;
        const uint32_t at = running + block_exclusive(f, wtot, total);
        if (f) { p.c_asset[at] = r[i].key_hi; p.c_entry[at] = (p.pos + q) * p.k + i; }
        running += total;
    }
}

// ---------------------------------------------------------------------------------------------
// weights_kernel: per sorted entry, its IDF weight and weight x similarity (two roundings, as `idf * sim` then `+=` in
// usearch_core.py:224-226), and the query index; per query simprint, the IDF of its own document frequency (:232-234).
// Everything score_kernel's sequential sums need lies in contiguous arrays afterwards: no pointer chasing in the chain.
// ---------------------------------------------------------------------------------------------
struct WeightParams {
    const uint32_t* entry;          // sorted by asset
    const isccsearch_record* rec;
    const uint32_t* rows;
    const uint32_t* freq_col;
    const uint32_t* freq_q;
    const double* sim_tab;
    const double* idf_tab;
    double* w;                      // [entries]
    double* ws;                     // [entries]
    uint32_t* q;                    // [entries]
    double* idf_q;                  // [nq]
    uint32_t* n_assets;             // zeroed here for score_kernel
    uint32_t entries, nq, k, dup_limit;
};
__global__ __launch_bounds__(BLOCK) void weights_kernel(const WeightParams p) {
#pragma clang fp contract(off)
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t == 0) *p.n_assets = 0;
    if (t < p.entries) {
        const uint32_t ent = p.entry[t];
        uint32_t f = 0;
        if (p.dup_limit) { f = p.freq_col[p.rows[ent]]; f = f < p.dup_limit ? f : p.dup_limit; }
        const double idf = p.idf_tab[f];
        p.w[t] = idf;
        p.ws[t] = idf * p.sim_tab[p.rec[ent].hamming];
        p.q[t] = ent / p.k;
    }
    if (t < p.nq) {
        uint32_t f = 0;
        if (p.dup_limit) { f = p.freq_q[t]; f = f < p.dup_limit ? f : p.dup_limit; }
        p.idf_q[t] = p.idf_tab[f];
    }
}

// ---------------------------------------------------------------------------------------------
// score_kernel: one thread per sorted entry; the thread at the head of an asset's run scores the asset.
//   usearch_core.py:215-236 -- total_idf and weighted_sim over the matched query simprints in ascending query order (the
//   insertion order of best_per_query), then total_idf over every unmatched query simprint in ascending order; score =
//   weighted / total when total > 0.  Sequential float64 operations with IEEE rounding, none contracted.
//   The run is walked four entries at a time (its loads do not depend on the sums), and leaves a bit per matched query in the
//   thread's LDS row, so that the second sum runs over the query simprints without touching memory it has to wait for.
// dynamic LDS: mask[words][threads] u64, words = ceil(nq / 64)
// ---------------------------------------------------------------------------------------------
struct ScoreParams {
    const uint64_t* asset;          // sorted
    const double* w;
    const double* ws;
    const uint32_t* q;
    const double* idf_q;
    double* score;
    uint32_t* order;
    uint32_t* matches;
    uint32_t* n_assets;
    uint32_t entries, nq, words;
};
__global__ void score_kernel(const ScoreParams p) {
    // HIP's __dadd_rn / __dmul_rn are plain `+` / `*`, and device code is compiled with -ffp-contract=fast: without this pragma
    // (and the file's -ffp-contract=off) `weighted + idf * sim` becomes ONE fused multiply-add -- a single rounding where the
    // reference's Python does two (found by tests/test_gpu_simprint_score.py: scores one ulp off)
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(smem);
    const uint32_t T = blockDim.x, tid = threadIdx.x, lane = tid & 63;
    const uint32_t e = blockIdx.x * T + tid;
    const uint64_t a = e < p.entries ? p.asset[e] : 0;
    const bool head = e < p.entries && (e == 0 || p.asset[e - 1] != a);
    double total = 0.0, weighted = 0.0;
    uint32_t end = e;
    if (e < p.entries) p.order[e] = e;
    if (head) {
        // 1. the matched query simprints, ascending (the run is sorted by asset, stably: by query inside an asset)
        for (uint32_t w = 0; w < p.words; ++w) mask[w * T + tid] = 0;
        constexpr int U = 8;
        uint32_t j = e;
        for (bool more = true; more;) {
            uint64_t aa[U]; double ww[U], ss[U]; uint32_t qq[U];
            for (int u = 0; u < U; ++u) {
                const uint32_t jj = j + u < p.entries ? j + u : p.entries - 1;
                aa[u] = p.asset[jj]; ww[u] = p.w[jj]; ss[u] = p.ws[jj]; qq[u] = p.q[jj];
            }
            int u = 0;
            for (; u < U; ++u) {
                if (j + u >= p.entries || aa[u] != a) { more = false; break; }
                total = total + ww[u];
                weighted = weighted + ss[u];
                mask[(qq[u] >> 6) * T + tid] |= 1ull << (qq[u] & 63);
            }
            j += u;
        }
        end = j;
    }
    // 2. every unmatched query simprint, ascending.  The whole wave walks the queries together -- lane l holds the IDF of query
    //    64 w + l, each step broadcasts one of them from a register (v_readlane) -- so nothing in a head's chain of additions waits
    //    for memory; lanes that head no run tag along.
    if (__ballot(head)) {
        for (uint32_t w = 0; w < p.words; ++w) {
            const uint32_t q0 = 64 * w, n = p.nq - q0 < 64 ? p.nq - q0 : 64;
            // (bits of queries beyond nq are set: skipped like matched ones, so that the loop below has a fixed trip count)
            const unsigned long long m = (head ? mask[w * T + tid] : ~0ULL) | (n < 64 ? ~0ULL << n : 0ULL);
            const double mine = lane < n ? p.idf_q[q0 + lane] : 0.0;
            const int lo = __double2loint(mine), hi = __double2hiint(mine);
#pragma unroll
            for (uint32_t i = 0; i < 64; ++i) {
                const double x = __hiloint2double(__builtin_amdgcn_readlane(hi, i), __builtin_amdgcn_readlane(lo, i));
                total = (m >> i) & 1 ? total : total + x;
            }
        }
    }
    if (e < p.entries) {
        p.score[e] = !head ? -1.0 : (total > 0.0 ? weighted / total : 0.0);       // (f64 division: correctly rounded by default)
        p.matches[e] = end - e;
    }
    const unsigned long long heads = __ballot(head);
    if (lane == 0 && heads) atomicAdd(p.n_assets, (uint32_t)__popcll(heads));
}

// ---------------------------------------------------------------------------------------------
// emit_kernel: block r writes result r of the first min(limit, assets) runs in (-score, asset) order -- and its matched
// chunks -- into pinned host memory; its chunks start behind those of the results before it.
// ---------------------------------------------------------------------------------------------
struct EmitParams {
    const double* score;            // sorted descending
    const uint32_t* order;          // run head of every sorted position
    const uint64_t* asset;
    const uint32_t* entry;
    const uint32_t* matches;
    const uint32_t* n_assets;
    const isccsearch_record* rec;
    const uint32_t* rows;
    const uint32_t* freq_col;
    const uint64_t* col[4];
    isccsearch_simprint_result* out_results;
    isccsearch_simprint_chunk* out_chunks;
    uint64_t* out_chunk_words;
    uint32_t* out_info;
    uint32_t limit, k, W, dup_limit;
    // hard-boundary requests (queue_exact): an entry is (given simprint g) * k + rank; its record belongs to distinct lookup d_of_g[g]
    const uint32_t* d_of_g;         // nullptr: the approximate path
    const uint32_t* freq_d;         // document frequency per distinct lookup
};
__global__ __launch_bounds__(BLOCK) void emit_kernel(const EmitParams p) {
    __shared__ uint32_t s_first;
    const uint32_t tid = threadIdx.x, r = blockIdx.x;
    const uint32_t assets = *p.n_assets;
    const uint32_t n = assets < p.limit ? assets : p.limit;
    if (r == 0 && tid == 0) { p.out_info[0] = n; p.out_info[1] = assets; p.out_info[2] = 0; if (n == 0) p.out_info[3] = 0; }
    if (r >= n) return;
    if (tid == 0) s_first = 0;
    __syncthreads();
    // chunks of the results before this one (block n - 1 also adds its own: the total)
    uint32_t before = 0;
    for (uint32_t i = tid; i < r; i += BLOCK) before += p.matches[p.order[i]];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
    if ((tid & 63) == 0 && before) atomicAdd(&s_first, before);
    __syncthreads();
    const uint32_t at = s_first, e = p.order[r], m = p.matches[e];
    if (tid == 0) {
        isccsearch_simprint_result res;
        res.asset = p.asset[e];
        res.score = p.score[r];
        res.matches = m;
        res.first_chunk = at;
        p.out_results[r] = res;
        if (r == n - 1) p.out_info[3] = p.out_chunks ? at + m : 0;
    }
    if (!p.out_chunks) return;
    for (uint32_t j = tid; j < m; j += BLOCK) {
        const uint32_t ent = p.entry[e + j];
        isccsearch_simprint_chunk c;
        c.query = ent / p.k;
        c.reserved = 0;
        if (p.d_of_g) {                       // a collision: stored simprint == query simprint, score 1.0 (lmdb_ops.py:228-235)
            const uint32_t d = p.d_of_g[c.query];
            c.key_lo = p.rec[(uint64_t)d * p.k + ent % p.k].key_lo;
            c.hamming = 0;
            c.freq = p.freq_d[d];
            p.out_chunks[at + j] = c;
            continue;
        }
        const isccsearch_record& rec = p.rec[ent];
        const uint32_t row = p.rows[ent];
        c.key_lo = rec.key_lo;
        c.hamming = rec.hamming;
        c.freq = p.dup_limit ? p.freq_col[row] : 1u;
        p.out_chunks[at + j] = c;
        for (uint32_t w = 0; w < p.W; ++w) p.out_chunk_words[(uint64_t)(at + j) * p.W + w] = p.col[w][row];
    }
}

// =====================================================================================================================
// Hard-boundary requests: search_simprints_exact (iscc_search/indexes/simprint/lmdb_ops.py:169-301) on the device.
// Input: the collision lists of the DISTINCT query simprints (records [nd][k] of a range-limited search at distance 0: every row
// equal to the query, ascending key = the order LMDB iterates the duplicates of a simprint key, k = dup_limit) and, for every query
// simprint AS GIVEN (repeats included, :197), the distinct lookup it maps to.
//   exact_freq_kernel     document frequency of every looked-up simprint: distinct assets in its list (:213-215)
//   exact_count_kernel    hits of every given simprint = length of its lookup's list -> offsets (offsets_kernel)
//   exact_compact_kernel  entries (asset, g * k + rank) in visiting order (given simprint, then key)
//   [stable sort by asset]
//   exact_weights_kernel  per sorted entry: its lookup d and that lookup's frequency, contiguous
//   exact_score_kernel    per asset (thread at the head of its run): coverage x quality (:252-301) -- distinct matched query
//                         simprints in first-seen order (a bit per lookup in the thread's LDS row), min / max frequency, then the
//                         min-max normalised inverse frequencies summed in that order; score < threshold: dropped (:222-223)
// =====================================================================================================================
__global__ __launch_bounds__(BLOCK) void exact_freq_kernel(const isccsearch_record* rec, const uint32_t* cnt, uint32_t* freq_d, uint32_t k) {
    __shared__ uint32_t total;
    const uint32_t d = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) total = 0;
    __syncthreads();
    const uint32_t n = cnt[d] < k ? cnt[d] : k;
    const isccsearch_record* r = rec + (uint64_t)d * k;
    uint32_t mine = 0;
    for (uint32_t i = tid; i < n; i += BLOCK) mine += (i == 0 || r[i].key_hi != r[i - 1].key_hi) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((tid & 63) == 0 && mine) atomicAdd(&total, mine);
    __syncthreads();
    if (tid == 0) freq_d[d] = total;
}

__global__ __launch_bounds__(BLOCK) void exact_count_kernel(const uint32_t* cnt, const uint32_t* d_of_g, uint32_t* hits, uint32_t* zero, uint32_t ng, uint32_t k) {
    const uint32_t g = blockIdx.x * BLOCK + threadIdx.x;
    if (g < ng) {
        const uint32_t c = cnt[d_of_g[g]];
        hits[g] = c < k ? c : k;
        zero[g] = 0;                      // (offsets_kernel's "unknown" input: nothing is unknown here)
    }
}

struct ExactCompactParams {
    const isccsearch_record* rec;
    const uint32_t* d_of_g;
    const uint32_t* hits;
    const uint32_t* offs;
    uint64_t* c_asset;
    uint32_t* c_entry;
    uint32_t k;
};
__global__ __launch_bounds__(BLOCK) void exact_compact_kernel(const ExactCompactParams p) {
    const uint32_t g = blockIdx.x, n = p.hits[g], at = p.offs[g];
    const isccsearch_record* r = p.rec + (uint64_t)p.d_of_g[g] * p.k;
    for (uint32_t i = threadIdx.x; i < n; i += BLOCK) { p.c_asset[at + i] = r[i].key_hi; p.c_entry[at + i] = g * p.k + i; }
}

__global__ __launch_bounds__(BLOCK) void exact_weights_kernel(const uint32_t* entry, const uint32_t* d_of_g, const uint32_t* freq_d, uint32_t* e_d, uint32_t* e_f,
                                                              uint32_t* n_assets, uint32_t entries, uint32_t k) {
    const uint32_t t = blockIdx.x * BLOCK + threadIdx.x;
    if (t == 0) *n_assets = 0;
    if (t < entries) {
        const uint32_t d = d_of_g[entry[t] / k];
        e_d[t] = d;
        e_f[t] = freq_d[d];
    }
}

struct ExactScoreParams {
    const uint64_t* asset;          // sorted
    const uint32_t* e_d;
    const uint32_t* e_f;
    double* score;
    uint32_t* order;
    uint32_t* matches;
    uint32_t* n_assets;
    uint32_t entries, words, queried;
    double threshold;
};
__global__ void exact_score_kernel(const ExactScoreParams p) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* mask = reinterpret_cast<unsigned long long*>(smem);
    const uint32_t T = blockDim.x, tid = threadIdx.x;
    const uint32_t e = blockIdx.x * T + tid;
    bool counted = false;
    if (e < p.entries) {
        const uint64_t a = p.asset[e];
        const bool head = e == 0 || p.asset[e - 1] != a;
        p.order[e] = e;
        double score = -1.0;
        uint32_t len = 0;
        if (head) {
            // query_to_best_freq (:277-283): one entry per distinct matched query simprint, in first-seen order; the frequency of a
            // collision's stored simprint is that of the query simprint itself, so "best" is that one value
            for (uint32_t w = 0; w < p.words; ++w) mask[w * T + tid] = 0;
            uint32_t distinct = 0, fmin = 0xFFFFFFFFu, fmax = 0;
            uint32_t j = e;
            for (; j < p.entries && p.asset[j] == a; ++j) {
                const uint32_t d = p.e_d[j], f = p.e_f[j];
                const unsigned long long bit = 1ull << (d & 63);
                unsigned long long& word = mask[(d >> 6) * T + tid];
                if (!(word & bit)) {
                    word |= bit;
                    distinct += 1;
                    fmin = f < fmin ? f : fmin;
                    fmax = f > fmax ? f : fmax;
                }
            }
            len = j - e;
            double quality = 1.0;
            if (distinct > 1 && fmin != fmax) {                       // (:291-299)
                const double min_inv = 1.0 / (double)fmax, max_inv = 1.0 / (double)fmin;
                const double span = max_inv - min_inv;
                for (uint32_t w = 0; w < p.words; ++w) mask[w * T + tid] = 0;
                double sum = 0.0;
                for (uint32_t i = e; i < j; ++i) {
                    const uint32_t d = p.e_d[i];
                    const unsigned long long bit = 1ull << (d & 63);
                    unsigned long long& word = mask[(d >> 6) * T + tid];
                    if (!(word & bit)) {
                        word |= bit;
                        const double inv = 1.0 / (double)p.e_f[i];
                        const double term = (inv - min_inv) / span;
                        sum = sum + term;
                    }
                }
                quality = sum / (double)distinct;
            }
            const double coverage = (double)distinct / (double)p.queried;
            score = coverage * quality;
            if (score < p.threshold) score = -1.0;                    // (:222-223) dropped: sorts behind every kept asset
            else counted = true;
        }
        p.score[e] = score;
        p.matches[e] = len;
    }
    const unsigned long long kept = __ballot(counted);
    if ((tid & 63) == 0 && kept) atomicAdd(p.n_assets, (uint32_t)__popcll(kept));
}


uint32_t log2_slots(uint32_t k) {   // hash slots of mark_kernel: the power of two >= 2 k (>= 64)
    uint32_t l = 6;
    while ((1u << l) < 2 * k) ++l;
    return l;
}

}  // namespace

size_t sort_temp_bytes(size_t entries) {
    size_t a = 0, b = 0;
    uint64_t* k64 = nullptr;
    double* kd = nullptr;
    uint32_t* v = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, a, k64, k64, v, v, entries, 0, 64, (hipStream_t) nullptr);
    (void)rocprim::radix_sort_pairs_desc(nullptr, b, kd, kd, v, v, entries, 0, 64, (hipStream_t) nullptr);
    return a > b ? a : b;
}

hipError_t queue_batch(const Buffers& b, const BatchArgs& a, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        // k = 4 096: 8 192 slots x 12 bytes
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mark_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const size_t off = (size_t)a.pos * a.k;
    MarkParams mp{b.rec + off, a.cnt, b.best + off, b.nbest + a.pos, b.freq_q + a.pos, b.unknown + a.pos, a.info, a.k, a.h_max, a.dup_limit, log2_slots(a.k)};
    const size_t lds = ((size_t)1 << mp.log2s) * 12;
    hipLaunchKernelGGL(mark_kernel, dim3(a.m), dim3(BLOCK), lds, stream, mp);
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(BLOCK), 0, stream, b.nbest + a.pos, b.unknown + a.pos, b.offs + a.pos, a.m, a.base, a.info);
    CompactParams cp{b.rec + off, b.best + off, b.offs + a.pos, b.c_asset[0], b.c_entry[0], a.k, a.pos};
    hipLaunchKernelGGL(compact_kernel, dim3(a.m), dim3(BLOCK), 0, stream, cp);
    return hipGetLastError();
}

hipError_t queue_exact(Buffers& b, const ExactArgs& a, hipStream_t stream) {
    hipError_t e;
    // (the caller has run exact_prepare: hits / offsets / entries are in place and `entries` is known)
    ExactCompactParams cp{b.rec, a.d_of_g, b.nbest, b.offs, b.c_asset[0], b.c_entry[0], a.k};
    hipLaunchKernelGGL(exact_compact_kernel, dim3(a.ng), dim3(BLOCK), 0, stream, cp);
    size_t bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.c_asset[0], b.c_asset[1], b.c_entry[0], b.c_entry[1], a.entries, 0, 64, stream);
    if (e != hipSuccess) return e;
    // (the outputs of the second sort are free until it runs: the per-entry lookup index and frequency live there meanwhile)
    uint32_t* const e_d = b.order[1];
    uint32_t* const e_f = reinterpret_cast<uint32_t*>(b.score[1]);
    hipLaunchKernelGGL(exact_weights_kernel, dim3((a.entries + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, b.c_entry[1], a.d_of_g, b.freq_q, e_d, e_f, b.n_assets, a.entries, a.k);
    const uint32_t words = (a.nd + 63) / 64;
    const uint32_t T = words <= 32 ? 256 : (words <= 64 ? 128 : 64);
    if (words > MAX_QUERY_SIMPRINTS / 64) return hipErrorInvalidValue;
    ExactScoreParams sp{b.c_asset[1], e_d, e_f, b.score[0], b.order[0], b.matches, b.n_assets, a.entries, words, a.queried, a.threshold};
    hipLaunchKernelGGL(exact_score_kernel, dim3((a.entries + T - 1) / T), dim3(T), (size_t)words * T * 8, stream, sp);
    bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs_desc(b.temp, bytes, b.score[0], b.score[1], b.order[0], b.order[1], a.entries, 0, 64, stream);
    if (e != hipSuccess) return e;
    EmitParams ep{};
    ep.score = b.score[1]; ep.order = b.order[1]; ep.asset = b.c_asset[1]; ep.entry = b.c_entry[1]; ep.matches = b.matches;
    ep.n_assets = b.n_assets; ep.rec = b.rec; ep.rows = nullptr; ep.freq_col = nullptr;
    ep.out_results = a.out_results; ep.out_chunks = a.out_chunks; ep.out_chunk_words = nullptr; ep.out_info = a.out_info;
    ep.limit = a.limit; ep.k = a.k; ep.W = 0; ep.dup_limit = 0;
    ep.d_of_g = a.d_of_g; ep.freq_d = b.freq_q;
    hipLaunchKernelGGL(emit_kernel, dim3(a.limit < a.entries ? a.limit : a.entries), dim3(BLOCK), 0, stream, ep);
    return hipGetLastError();
}

// document frequencies of the lookups, hits per given simprint and their offsets; info[0] = entries (travels to the host)
hipError_t exact_prepare(const Buffers& b, const uint32_t* cnt, const uint32_t* d_of_g, uint32_t nd, uint32_t ng, uint32_t k, uint32_t* info, hipStream_t stream) {
    hipLaunchKernelGGL(exact_freq_kernel, dim3(nd), dim3(BLOCK), 0, stream, b.rec, cnt, b.freq_q, k);
    hipLaunchKernelGGL(exact_count_kernel, dim3((ng + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, cnt, d_of_g, b.nbest, b.unknown, ng, k);
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(BLOCK), 0, stream, b.nbest, b.unknown, b.offs, ng, 0u, info);
    return hipGetLastError();
}

hipError_t queue_score(Buffers& b, const ScoreArgs& a, hipStream_t stream) {
    hipError_t e;
    size_t bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs(b.temp, bytes, b.c_asset[0], b.c_asset[1], b.c_entry[0], b.c_entry[1], a.entries, 0, 64, stream);
    if (e != hipSuccess) return e;
    // (the outputs of the second sort are free until it runs: weights and query indices live there meanwhile)
    WeightParams wp{b.c_entry[1], b.rec, b.rows, a.freq_col, b.freq_q, a.sim_tab, a.idf_tab, b.score[1], b.ws, b.order[1], b.idf_q,
                    b.n_assets, a.entries, a.nq, a.k, a.dup_limit};
    const uint32_t wn = a.entries > a.nq ? a.entries : a.nq;
    hipLaunchKernelGGL(weights_kernel, dim3((wn + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, stream, wp);
    const uint32_t words = (a.nq + 63) / 64;
    const uint32_t T = words <= 32 ? 256 : (words <= 64 ? 128 : 64);         // one LDS row of `words` u64 per thread, <= 64 KB per block
    if (words > MAX_QUERY_SIMPRINTS / 64) return hipErrorInvalidValue;
    ScoreParams sp{b.c_asset[1], b.score[1], b.ws, b.order[1], b.idf_q, b.score[0], b.order[0], b.matches, b.n_assets, a.entries, a.nq, words};
    hipLaunchKernelGGL(score_kernel, dim3((a.entries + T - 1) / T), dim3(T), (size_t)words * T * 8, stream, sp);
    bytes = b.temp_bytes;
    e = rocprim::radix_sort_pairs_desc(b.temp, bytes, b.score[0], b.score[1], b.order[0], b.order[1], a.entries, 0, 64, stream);
    if (e != hipSuccess) return e;
    EmitParams ep{};
    ep.score = b.score[1]; ep.order = b.order[1]; ep.asset = b.c_asset[1]; ep.entry = b.c_entry[1]; ep.matches = b.matches;
    ep.n_assets = b.n_assets; ep.rec = b.rec; ep.rows = b.rows; ep.freq_col = a.freq_col;
    for (uint32_t w = 0; w < 4; ++w) ep.col[w] = a.col[w];
    ep.out_results = a.out_results; ep.out_chunks = a.out_chunks; ep.out_chunk_words = a.out_chunk_words; ep.out_info = a.out_info;
    ep.limit = a.limit; ep.k = a.k; ep.W = a.W; ep.dup_limit = a.dup_limit;
    hipLaunchKernelGGL(emit_kernel, dim3(a.limit < a.entries ? a.limit : a.entries), dim3(BLOCK), 0, stream, ep);
    return hipGetLastError();
}

}  // namespace isksp
