// scan_params.hip.h -- what the scan kernels of kernels.hip.h (XOR + popcount on the VALU) and of mfma_scan.hip (FP4 on the
// matrix cores) share: constants, the launch parameters and the candidate append.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace isk {

constexpr int BLOCK = 256;            // threads per workgroup = 4 waves of 64
constexpr uint32_t HB = 264;          // histogram stride per query (bins 0..256 used)
constexpr uint32_t NBINS = 257;       // hamming distance 0..256
constexpr uint32_t CNT_STRIDE = 32;   // one candidate counter per 128-byte line: device-scope atomics on
                                      // words of one line serialise at the memory side (~90 M/s per line)
constexpr uint32_t BIT31 = 0x80000000u;
constexpr uint32_t BIAS_NEVER = 0x80000000u;  // bias + h always has bit 31 set: never a candidate
constexpr uint32_t COUNT_OVERFLOW = 0xFFFFFFFFu;   // == ISCCSEARCH_COUNT_OVERFLOW: "this list needs the exact fallback"
constexpr int MODE_COLLECT = 0;   // append candidates
constexpr int MODE_HIST = 1;      // count candidates per hamming distance
constexpr int MODE_BOTH = 2;      // both: the threshold levels
constexpr int MODE_STRETCH = 3;   // as MODE_BOTH, for the stretches of the collect pass: a separate instantiation so that
                                  // profilers tell the pass from the levels by kernel name
constexpr int MODE_SELF = 4;      // single pass with SELF-TIGHTENING thresholds (mfma_scan.hip): candidates are appended and counted
                                  // in a cumulative histogram; the lane that makes "k rows within t" true lowers the live
                                  // threshold of its query, and every wave re-reads the live thresholds once per step

struct ScanParams {
    const uint64_t* col[4];   // segment columns (word-major)
    uint64_t row_begin;       // first row scanned (a multiple of the tile size)
    uint64_t n_rows;          // rows [row_begin, n_rows) are scanned
    const uint64_t* queries;  // [nq_pad][4] big-endian packed query words
    const uint32_t* bias;     // [nq_pad] 0x7FFFFFFF - tau  (BIAS_NEVER for padding queries)
    uint32_t* cnt;            // [nq_pad * CNT_STRIDE] candidates appended so far (MODE_COLLECT)
    uint64_t* cand;           // [nq_pad][cap] (hamming << 48) | row          (MODE_COLLECT)
    uint32_t* ghist;          // [nq_pad][HB] histogram of hamming <= tau     (MODE_HIST)
    uint32_t cap;
    uint32_t mask_lo, mask_hi;  // mask of the last compared word (partial-word prefixes)
    uint32_t k;                 // results wanted per query
    uint32_t fold_tau;          // scan_adapt_kernel: groups whose thresholds are all <= this take the folded fast path (0: never)
    uint32_t nq_pad;            // queries (bias entries) behind `queries` / `bias`: the MFMA kernel pads its last chunk itself
    float* thr_live;            // [nq_pad] MODE_SELF: live threshold of every query as the MFMA kernel compares it: tau - popc(query)
                                //          (mfma_pack_kernel: the same number PACKED, see pack_threshold; boot kernels write either form)
    uint32_t refresh_steps;     // MODE_SELF: a wave re-reads its share of the live thresholds every this many steps of a full chunk (power of two)
};

// mfma_pack_kernel (mfma_scan.hip) keeps two dot products per f32 accumulator: bits = 0x4B402000 + d1 + 65536 * d2, i.e. the
// LOW half is PK_LO0 + d1 and the HIGH half PK_HI0 + d2, both positive normal f16 patterns.  A threshold (hit <=> d <= thr)
// travels as the first NON-hit pattern of each half in one word; 0 = no row can hit.  Monotone in thr, so the live
// thresholds of the self-tightening pass are lowered with ONE atomicMin on the packed word.
constexpr uint32_t PK_LO0 = 0x2000u, PK_HI0 = 0x4B40u;                 // bit patterns of the two halves at d = 0
constexpr uint32_t PK_MAGIC = (PK_HI0 << 16) | PK_LO0;                  // 0x4B402000 = 2^23 + 0x402000 as f32
__host__ __device__ __forceinline__ uint32_t pack_threshold(int thr) {
    if (thr < -64) return 0u;
    const uint32_t t1 = (uint32_t)((thr > 64 ? 64 : thr) + 1 + 64);     // 0 .. 129
    return ((PK_HI0 - 64 + t1) << 16) | (PK_LO0 - 64 + t1);
}
__host__ __device__ __forceinline__ int unpack_threshold(uint32_t tpk) { return (int)(tpk & 0xFFFFu) - (int)PK_LO0 - 1; }

// returns the candidate's slot in the query's list (0 in MODE_HIST)
template <int MODE>
__device__ __forceinline__ uint32_t emit(const ScanParams& p, uint32_t qi, uint32_t h, uint64_t row) {
    uint32_t slot = 0;
    if constexpr (MODE == MODE_COLLECT || MODE == MODE_BOTH || MODE == MODE_STRETCH) {
        slot = atomicAdd(&p.cnt[(uint64_t)qi * CNT_STRIDE], 1u);
        if (slot < p.cap) p.cand[(uint64_t)qi * p.cap + slot] = ((uint64_t)h << 48) | row;
    }
    if constexpr (MODE == MODE_HIST || MODE == MODE_BOTH || MODE == MODE_STRETCH) atomicAdd(&p.ghist[(uint64_t)qi * HB + h], 1u);
    return slot;
}

// MODE_SELF.  ghist[q][t] counts the appended rows with hamming <= t, for every t BELOW the threshold the appending wave
// last saw (`tau_seen` >= the live threshold, so every count below the live threshold is complete).  The lane whose row
// makes count[t] reach k has proved "k appended rows lie within t": it lowers the live threshold to t (a looser leftover
// from a lost race is harmless: thresholds only decide what is APPENDED, select_kernel picks the exact top-k afterwards).
__device__ __forceinline__ void lower_threshold(float* addr, float v) {
    int* a = reinterpret_cast<int*>(addr);
    int old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__int_as_float(old) > v) {
        const int prev = atomicCAS(a, old, __float_as_int(v));
        if (prev == old) break;
        old = prev;
    }
}

// the live threshold of query qi := min(itself, thr), in the representation the running kernel keeps (PACKED: mfma_pack_kernel)
template <bool PACKED>
__device__ __forceinline__ void lower_live(const ScanParams& p, uint32_t qi, int thr) {
    if constexpr (PACKED) atomicMin(reinterpret_cast<uint32_t*>(p.thr_live) + qi, pack_threshold(thr));
    else lower_threshold(p.thr_live + qi, (float)thr);
}

}  // namespace isk
