// mfma_scan.hip -- the collect scan for LARGE query batches on the matrix cores of gfx950 (MI355X).
//
// Same contract as scan_kernel (kernels.hip.h): for every row of [row_begin, n_rows) and every query, append
// (hamming, row) to the query's candidate list when hamming over the compared prefix <= tau_q
// (reference call sites: iscc_search/indexes/usearch/index.py:2037, iscc_search/indexes/simprint/usearch_core.py:165;
// metric: docs/explanation/similarity-search.md:24-29).  What differs is the arithmetic:
//
//   hamming(row, q) = popc(q) + dot(row bits as 0/1, query bits as +1/-1)            (exact small integers)
//
// so the (rows x queries x bits) work is a dense contraction.  The XOR + popcount kernel needs 4.5 VALU instructions per
// (row, query, 64-bit word), each issuing at one wave64 per ~4 cycles per SIMD, and is VALU-bound from ~11 queries per
// pass on.  Here the products run on the matrix pipe in its cheapest format: FP4 (e2m1: 0x2 = +1,
// 0xA = -1, 0x0 = 0) through v_mfma_f32_32x32x64_f8f6f4 (the unscaled form: block scales 2^0) -- ONE instruction (32 cycles)
// per 32 rows x 32 queries x 64 bits, f32 accumulation, exact because every partial sum is an integer of magnitude
// <= 256.  (The int8 form, v_mfma_i32_32x32x32_i8, needs two instructions of the same length per word and measured 1.44x
// slower: profiles/r02_proto_mfma_scan.txt.)
//
// A wave owns T = 2 tiles of 32 rows: each lane expands ITS 32 bits of one row per word into 32 nibbles (4 VGPRs, the A
// operand), once per step, and then walks every query group of the block's chunk.  A group is 32 queries pre-expanded to
// +1/-1 nibbles in LDS (one ds_read_b128 per lane and word); W MFMAs per tile give the 32 x 32 dot products; a lane's 16
// results per tile all belong to ONE query (C/D column = lane & 31), so 16 v_min3_f32 fold the two tiles and one compare
// against thr_q = tau_q - popc(q) decides whether the lane enters the rare emit path.  Masked prefixes (NPHD between codes
// of different lengths) cost nothing: the query nibbles beyond the prefix are 0.
//
// What bounds it is vector ISSUE: the fold (16 values per lane per 1 024 pairs, two per v_min3) and the MFMAs share the
// SIMD's issue port (~50 cycles per 1 024 pairs and word against 32 of matrix-pipe time); the XOR + popcount kernel needs
// ~290.  Rows cross the memory system once per chunk of up to 1 024 queries.  Measured numbers: DESIGN.md section 4.
//
// Modes (scan_params.hip.h): MODE_COLLECT (range-limited searches: a given threshold), MODE_BOTH / MODE_STRETCH (the threshold
// levels and the collect pass of the level design: append + histogram, picks between launches) and MODE_SELF -- ONE launch
// over all rows whose thresholds tighten themselves: the live thresholds are re-read from global memory once per step, every
// candidate is counted per distance, and the lane that proves "k rows within t" lowers the threshold (see Pending, emit_self,
// lower_threshold).  The default for k <= 512.
//
// Built with -mllvm -amdgpu-mfma-vgpr-form=1 -ffinite-math-only: hipcc otherwise puts the accumulators in AGPRs and pays one
// v_accvgpr_read per result before the fold (16 extra VALU instructions per tile and group), and canonicalises the inputs of
// every 2-input fminf (two v_max per group: the values are small integers, never NaN).
#include "mfma_scan.h"

#include <hip/hip_runtime.h>

namespace isk {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(3))) v4i* lds_frag_ptr;
typedef const __attribute__((address_space(3))) float* lds_thr_ptr;

// row tiles (32 rows) per wave and step: two share every B fragment, threshold read and compare; experiment switch per W
#ifndef ISK_MFMA_TILES_W4
#define ISK_MFMA_TILES_W4 2
#endif
template <int W> constexpr int mfma_tiles() { return W == 4 ? ISK_MFMA_TILES_W4 : 2; }
#ifndef ISK_ORDERED_STAGE
// experiment switch: 1 = the stage of codes of two to four words as single-instruction asm statements in issue order (fold_op /
// mfma_asm below).  Bit-exact (GPU suite green) but 2-4 % SLOWER than hipcc's own schedule on the same box -- 256-bit 9.0 against
// 8.8 ms per 1 024 queries, 128-bit 4.63 against 4.45, 192-bit 7.77 against 7.57 (profiles/r03_ab_ordered_stage.txt): with 2 W
// MFMAs per 17 fold instructions the matrix pipe, not the issue order, is the bound there.  Off.
#define ISK_ORDERED_STAGE 0
#endif
#ifndef ISK_EXP_NO_CANDIDATE_MEMORY
#define ISK_EXP_NO_CANDIDATE_MEMORY 0     // experiment switch (never in the product build): candidates are found but not appended
#endif
#ifndef ISK_SCALAR_STEPS_FROM_W
#define ISK_SCALAR_STEPS_FROM_W 2     // experiment switch: code widths (in 64-bit words) from which the step number is kept scalar
#endif
constexpr uint32_t PK_RING_ENTRIES = 8, PK_RING_ENTRY_DWORDS = 36;      // per wave: saved result blocks of lanes that hold a hit (144 B each)
constexpr int MBLOCK = 256;           // 4 waves; a chunk's LDS image is <= 40 KB, so LDS admits four blocks per CU
// Registers decide: the rare emit path (both accumulator sets live + 64-bit row numbers) peaks at 130-175 VGPRs, i.e. three
// waves per SIMD for W <= 3 and two for W = 4.  Forcing four (128 VGPRs) spilled to scratch; the prototype measured
// 3.22 ms with three resident blocks against 3.15 ms with four (profiles/r02_proto_mfma_scan.txt) -- not worth a spill.
template <int W> constexpr int mfma_min_waves() { return W <= 3 ? 3 : (mfma_tiles<W>() == 1 ? 4 : 2); }
constexpr int FP4 = 4;                // cbsz / blgp format code of e2m1
// Both scale operands constant 0: hipcc then selects the UNSCALED encoding, v_mfma_f32_32x32x64_f8f6f4 (no
// v_mfma_ld_scale prefix, no scale VGPRs), which multiplies as with block scales 2^0.  Same bits as the scaled form with
// E8M0 scales 0x7F (both checked against a brute-force kernel: tools/proto_mfma_scan.hip, -DPROTO_SCALE=0) and 6 % faster
// (3.10 vs 3.30 ms per 100 M x 1 024 pass): one instruction less to issue per MFMA.
constexpr int SCALE_ONE = 0;

__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
// a live threshold as other CUs last wrote it: device-scope load, past this CU's vector cache
__device__ __forceinline__ float live_threshold(const float* addr) {
    return __int_as_float(__hip_atomic_load(reinterpret_cast<const int*>(addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Dword j (0..3) of a 32-bit half: nibble t holds bit j + 4 t.  Rows (A) and queries (B) use the same map, so the k order
// inside the instruction does not matter; lanes 0..31 carry the low half of a word and lanes 32..63 the high half on both sides.
__device__ __forceinline__ uint32_t nibbles(uint32_t x, int j) { return (x >> j) & 0x11111111u; }

// ---- the stage of mfma_scan_kernel for codes of two to four words, in ISSUE ORDER (round 3) ---------------------------------
// EXPERIMENT (ISK_ORDERED_STAGE, off by default: measured slower, see the switch).  hipcc issues the 2 W MFMAs of a group back
// to back and the 17-instruction fold of the previous group behind them.  Here, as in mfma_pack_kernel below, the MFMAs and the
// fold are single-instruction asm statements in the order they should issue:
// MFMA (tile 0, word 0), MFMA (tile 1, word 0), then after every further MFMA a few fold instructions of the PREVIOUS group.
// hipcc places no hazard nops for asm: the fold starts after the stage's second MFMA and walks accumulator 0 of the old group
// first, so every read of an MFMA result lies >= 12 instructions behind that MFMA (checked by tools/audit_kernels.py).
template <int N>
__device__ __forceinline__ void fold_op(float& mA, float& mB, const v16f& o0, const v16f& o1) {
    if constexpr (N == 0) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(mA) : "v"(o0[0]), "v"(o0[1]), "v"(o0[2]));
    else if constexpr (N <= 6) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mA) : "v"(o0[2 * N + 1]), "v"(o0[2 * N + 2]));
    else if constexpr (N == 7) asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(mB) : "v"(o1[0]), "v"(o1[1]), "v"(o1[2]));
    else if constexpr (N <= 13) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mB) : "v"(o1[2 * (N - 7) + 1]), "v"(o1[2 * (N - 7) + 2]));
    else if constexpr (N == 14) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(mA) : "v"(o0[15]), "v"(o1[15]));
    else asm volatile("v_min_f32 %0, %0, %1" : "+v"(mA) : "v"(mB));
}
template <int FROM, int TO>
__device__ __forceinline__ void fold_ops(float& mA, float& mB, const v16f& o0, const v16f& o1) {
    if constexpr (FROM < TO && FROM < 16) {
        fold_op<FROM>(mA, mB, o0, o1);
        fold_ops<FROM + 1, TO>(mA, mB, o0, o1);
    }
}
template <bool FIRST_WORD>
__device__ __forceinline__ void mfma_asm(v16f& acc, const v4i& a, const v4i& b) {
    if constexpr (FIRST_WORD) asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, 0 cbsz:4 blgp:4" : "=&v"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4" : "+v"(acc) : "v"(a), "v"(b));
}

// LDS image of a chunk: B fragments [groups][W][64] v4i | thr[groups * 32] (float) | popc[groups * 32]
template <int W, int MODE>
__global__ __launch_bounds__(MBLOCK, mfma_min_waves<W>()) void mfma_scan_kernel(const ScanParams p, const uint32_t groups) {
    constexpr int MT = mfma_tiles<W>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i* lb = reinterpret_cast<v4i*>(smem);
    float* lthr = reinterpret_cast<float*>(smem + (size_t)groups * W * 64 * 16);
    int* lpop = reinterpret_cast<int*>(lthr + groups * 32);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t q0 = blockIdx.y * groups * 32;       // first query of this block's chunk

    // prologue: expand the chunk's queries to +1 / -1 nibbles (0 beyond the compared prefix and for padding queries)
    for (uint32_t i = tid; i < groups * 32 * 2 * W; i += MBLOCK) {
        const uint32_t ql = i / (2 * W), rest = i % (2 * W), w = rest >> 1, hh = rest & 1;
        const uint32_t q = q0 + ql;
        const bool live = q < p.nq_pad;
        const uint64_t qw = live ? p.queries[(uint64_t)q * 4 + w] : 0;
        const uint32_t x = hh ? (uint32_t)(qw >> 32) : (uint32_t)qw;
        uint32_t m = live ? 0xFFFFFFFFu : 0u;
        if (w == W - 1) m &= hh ? p.mask_hi : p.mask_lo;
        const uint32_t g = ql >> 5, c = ql & 31;
        v4i frag;
#pragma unroll
        for (int j = 0; j < 4; ++j) frag[j] = (int)((0x22222222u | (nibbles(x, j) << 3)) & (nibbles(m, j) * 0xFu));   // bit ? -1 : +1, masked: 0
        lb[((size_t)g * W + w) * 64 + hh * 32 + c] = frag;
    }
    for (uint32_t ql = tid; ql < groups * 32; ql += MBLOCK) {
        const uint32_t q = q0 + ql;
        int pc = 0, tau = -1;
        if (q < p.nq_pad) {
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint64_t qw = p.queries[(uint64_t)q * 4 + w];
                if (w == W - 1) qw &= ((uint64_t)p.mask_hi << 32) | p.mask_lo;
                pc += __popcll(qw);
            }
            tau = (int)(0x7FFFFFFFu - p.bias[q]);       // BIAS_NEVER -> -1: no row can be a candidate
        }
        lpop[ql] = pc;
        if constexpr (MODE == MODE_SELF) lthr[ql] = q < p.nq_pad ? live_threshold(p.thr_live + q) : -1.0e9f;
        else lthr[ql] = (float)(tau - pc);              // hamming <= tau  <=>  dot <= tau - popc(q)
    }
    __syncthreads();

    const uint64_t first = p.row_begin / (32 * MT);                         // row_begin is a multiple of 64
    const uint64_t nsteps = (p.n_rows + 32 * MT - 1) / (32 * MT);           // the last step may be partial
    const uint64_t stride = (uint64_t)gridDim.x * (MBLOCK / 64);
    // W >= 2: the wave number is read as a SCALAR, so that the step number and the row addresses live on the scalar unit
    // (scalar-base loads) instead of ~14 vector instructions of 64-bit address arithmetic per step.  Same box, A/B: 128-bit
    // 4.586 against 4.605 ms per 1 024 queries and 0.57-0.59 against 0.615 ms per 64; 256-bit 9.20 against 9.36 and no change
    // at 64 queries.  Not for 64-bit codes: nothing at 1 024 queries and 17-64 queries measured 15-20 % slower.
    constexpr bool SCALAR_STEPS = ISK_SCALAR_STEPS_FROM_W <= W;
    const uint32_t wave_u = SCALAR_STEPS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)wave) : wave;
    uint64_t step = first + (uint64_t)blockIdx.x * (MBLOCK / 64) + wave_u;
    if (step >= nsteps) return;
    const uint64_t last_row = p.n_rows - 1;

    struct Acc { v16f t[MT]; };
    const uint32_t* col32[W];
#pragma unroll
    for (int w = 0; w < W; ++w) col32[w] = reinterpret_cast<const uint32_t*>(p.col[w]);

    constexpr bool ORDERED = W >= 2 && MT == 2 && ISK_ORDERED_STAGE;      // the stage in issue order (inline asm), see fold_op above
    static_assert(MT == 2, "the candidate ring holds the two tiles' 32 results of a lane");
    // ---- candidates: as in mfma_pack_kernel (below) -- the lanes that hold a result within their query's threshold copy their
    // 32 results (+ query, threshold) into their wave's LDS ring and the stage loop goes on; at the end of the step the ring is
    // walked with a real loop, TWO saved blocks per trip, lane v on result v & 31 of block v >> 5.  MODE_SELF: the list slot is
    // consumed at the lane's next candidate or at the end of the next step, the distance counts are no-return atomics, and
    // CHECKER lanes notice "k rows within t" (one counter read per look) and lower the live threshold.  (Round 2: 32 unrolled
    // compares and two returned atomics + a dependent chain per candidate inside the stage loop: config 5's table -- 10 M x
    // 128-bit, 512 queries, k = 400 -- scanned at 0.94 ms against 0.22 ms of matrix-pipe time.)
    constexpr uint32_t RING_E = PK_RING_ENTRIES, ENTRY = PK_RING_ENTRY_DWORDS;
    uint32_t* const ring = reinterpret_cast<uint32_t*>(lpop + groups * 32) + wave * (RING_E * ENTRY);
    uint32_t rcount = 0;
    uint32_t pend_slot = 0, pend_lo = 0, pend_hi = 0x80000000u;        // pend_hi bit 31: nothing pending
    auto pend_complete = [&]() {
        if (!(pend_hi & 0x80000000u)) {
            const uint32_t qi = q0 + (pend_hi >> 21);                   // query in chunk : 10 | hamming : 9 | row >> 32 : 12
            if (pend_slot < p.cap) p.cand[(uint64_t)qi * p.cap + pend_slot] = ((uint64_t)((pend_hi >> 12) & 0x1FFu) << 48) | ((uint64_t)(pend_hi & 0xFFFu) << 32) | pend_lo;
            pend_hi = 0x80000000u;
        }
    };
    auto process_ring = [&](uint64_t st) {
        const uint32_t sub = lane >> 5, ri = lane & 31;
        const uint32_t off0 = (ri >> 4) * 32 + (ri & 3) + 8 * ((ri & 15) >> 2);
        for (uint32_t e = 0; e < rcount; e += 2) {
            if (e + sub < rcount) {
                const uint32_t* const blk = ring + (e + sub) * ENTRY;
                const float v = __uint_as_float(blk[ri]), thr = __uint_as_float(blk[33]);
                const uint32_t head = blk[32], ql = head & 0xFFFFu;
                const uint64_t row = st * (32 * MT) + off0 + 4 * (head >> 16);
                if (v <= thr && row <= last_row) {
                    const int pc = lpop[ql];
                    const uint32_t hd = (uint32_t)((int)v + pc);
                    if constexpr (MODE == MODE_SELF) {
                        pend_complete();
                        const uint32_t qi = q0 + ql;
                        const int tau_seen = (int)thr + pc;
                        pend_slot = atomicAdd(&p.cnt[(uint64_t)qi * CNT_STRIDE], 1u);
                        uint32_t* const counts = p.ghist + (uint64_t)qi * HB;
                        for (int t = (int)hd; t < tau_seen; ++t) __hip_atomic_fetch_add(&counts[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        pend_lo = (uint32_t)row;
                        pend_hi = (ql << 21) | (hd << 12) | (uint32_t)(row >> 32);       // rows < 2^44
                    } else {
                        emit<MODE>(p, q0 + ql, hd, row);
                    }
                }
            }
        }
        rcount = 0;
    };
    // `mask`: the lanes whose minimum is within their threshold (query g * 32 + (lane & 31), rows 4 * (lane >> 5) + ... of the tiles)
    auto save_hits = [&](const Acc& acc, uint64_t mask, float thr, uint32_t g, uint64_t st) {
        while (mask) {                              // wave-uniform; more than one trip only when the ring fills up
            const uint32_t room = RING_E - rcount;
            if (room == 0) { process_ring(st); continue; }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const bool mine = ((mask >> lane) & 1) != 0 && rank < room;
            if (mine) {
                uint32_t* const blk = ring + (rcount + rank) * ENTRY;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 16; i += 4)
                        *reinterpret_cast<float4*>(blk + 16 * j + i) = make_float4(acc.t[j][i], acc.t[j][i + 1], acc.t[j][i + 2], acc.t[j][i + 3]);
                *reinterpret_cast<uint2*>(blk + 32) = make_uint2((g * 32 + r) | (h << 16), __float_as_uint(thr));
            }
            const uint64_t taken = __builtin_amdgcn_ballot_w64(mine);
            rcount += (uint32_t)__builtin_popcountll(taken);
            mask &= ~taken;
        }
    };
    // (the stage of 64-bit codes that stay off the packed kernel: fold by builtins, then the same ring)
    auto rare = [&](const Acc& acc, float thr, uint32_t g, uint64_t st, float mall) {
        const uint64_t mask = __builtin_amdgcn_ballot_w64(mall <= thr);
        if (__builtin_expect(mask != 0, 0)) save_hits(acc, mask, thr, g, st);
    };
    auto reduce = [&](const Acc& acc, float thr, uint32_t g, uint64_t st) {
        float m[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) m[t] = min3f(acc.t[t][0], acc.t[t][1], acc.t[t][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2)
#pragma unroll
            for (int t = 0; t < MT; ++t) m[t] = min3f(m[t], acc.t[t][i], acc.t[t][i + 1]);
        float mall;
        if constexpr (MT == 2) mall = fminf(min3f(m[0], acc.t[0][15], acc.t[1][15]), m[1]);
        else mall = fminf(m[0], acc.t[0][15]);
        rare(acc, thr, g, st, mall);
    };

    const v16f zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v8i a[MT][W];     // only the first four dwords carry FP4 data; the instruction ignores the rest
    // one word of one group: one MFMA per tile into the group's accumulators
    auto mm = [&](Acc& acc, int w, const v4i& b) {
        const v8i b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < MT; ++t)
            acc.t[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[t][w], b8, w == 0 ? zero : acc.t[t], FP4, FP4, 0, SCALE_ONE, 0, SCALE_ONE);
    };
    // An empty asm naming BOTH accumulator sets right after the first MFMAs of the next group: the fold of the previous
    // group then depends on it, so hipcc can neither hoist that fold above the MFMAs nor give the two sets the same
    // registers (it did both in the prototype and serialised MFMA -> s_nop 10 -> fold).
    auto pin2 = [&](Acc& x, Acc& y) {
        if constexpr (MT == 2) asm volatile("" : "+v"(x.t[0]), "+v"(x.t[1]), "+v"(y.t[0]), "+v"(y.t[1]));
        else asm volatile("" : "+v"(x.t[0]), "+v"(y.t[0]));
    };
    const v4i* lbl = lb + lane;
    const float* lt = lthr + r;
    auto row_of = [&](uint64_t st, int t) { const uint64_t row = (st * MT + t) * 32 + r; return row <= last_row ? row : last_row; };

    // the rows of step `st`: lane (r, h) of tile t reads dword h of row st * 32 MT + 32 t + r.  SCALAR_STEPS: a uniform base plus a
    // constant per-lane offset; only the table's last step can be partial and clamps per lane as the general form does
    const uint32_t lane_dword = r * 2 + h;
    auto load_rows = [&](uint64_t st, uint32_t (&dst)[MT][W]) {
        if (SCALAR_STEPS && (st + 1) * (32 * MT) <= p.n_rows) {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int w = 0; w < W; ++w) dst[t][w] = (col32[w] + st * (64 * MT) + t * 64)[lane_dword];
        } else {
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int w = 0; w < W; ++w) dst[t][w] = col32[w][row_of(st, t) * 2 + h];
        }
    };
    uint32_t x[MT][W], xn[MT][W];
    load_rows(step, x);
    // MODE_SELF: wave w keeps the block's copy of thresholds [256 w, 256 w + 256) fresh -- requested here, written to LDS
    // after the group loop, picked up by all four waves from their next step on (a stale threshold is only a looser one)
    // (the lane's slice of the thresholds is RECOMPUTED at each use -- mbcnt of a laundered zero -- or hipcc keeps a 64-bit global
    //  address and an LDS address alive through the group loop: registers the 192-bit kernel does not have)
    const uint32_t wave_sc = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    auto fresh_index = [&]() {
        uint32_t z = 0;
        asm volatile("" : "+v"(z));
        return wave_sc * 256 + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)) * 4;
    };
    const bool refresh = MODE == MODE_SELF && wave * 256 + lane * 4 < groups * 32 && q0 + wave * 256 + lane * 4 < p.nq_pad;   // nq_pad is a multiple of 8
    // ... every `refresh_steps` steps when the chunk is full (32 groups), proportionally less often for smaller chunks
    const uint32_t refresh_mask = (groups >= 32 ? 1u : groups >= 16 ? 2u : groups >= 8 ? 4u : groups >= 4 ? 8u : 16u) * p.refresh_steps - 1u;
    uint32_t trip = 0;
    for (; step < nsteps; step += stride, ++trip) {
        const uint64_t ns = step + stride < nsteps ? step + stride : step;
        float fresh[4] = {0.f, 0.f, 0.f, 0.f};
        // (a wave's first steps always look: all waves start under the bootstrap threshold at once, and until the first update
        //  arrives every row within it is appended -- a 4 M-row table would be scanned whole under it at 16 steps per look)
        const bool look = trip < 8 || (trip & refresh_mask) == 0;
        const bool refresh_now = refresh && look;                             // (MODE_SELF only: `refresh` is false otherwise)
        uint32_t chk_count = 0, chk_what = ~0u;       // chk_what: query in chunk | hamming level << 16, ~0: no task
        if constexpr (MODE == MODE_SELF) {
            if (refresh_now) {
                const float* const src = p.thr_live + q0 + fresh_index();
#pragma unroll
                for (int i = 0; i < 4; ++i) fresh[i] = live_threshold(src + i);
            }
            // checkers: task (query, j) reads count[q][tau_q - j], j = 1..4 (see mfma_pack_kernel)
            const uint32_t slices = groups * 32 * 4 / 64, nwaves = gridDim.x * (MBLOCK / 64);
            const uint32_t gw = blockIdx.x * (MBLOCK / 64) + wave_sc;
            const uint32_t slice = nwaves >= slices ? gw : (gw + trip * nwaves) % slices;
            if (look && slice < slices) {
                const uint32_t task = slice * 64 + lane, ql = task >> 2;
                const float thr = lthr[ql];
                const int level = (int)thr + lpop[ql] - 1 - (int)(task & 3);
                if (thr > -1.0e8f && level >= 0 && q0 + ql < p.nq_pad) {
                    chk_what = ql | ((uint32_t)level << 16);
                    chk_count = (uint32_t)__hip_atomic_load(reinterpret_cast<const int*>(p.ghist + (uint64_t)(q0 + ql) * HB + level), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        load_rows(ns, xn);                                                          // next step's rows, in flight during this one
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w)
                a[t][w] = v8i{(int)(nibbles(x[t][w], 0) << 1), (int)(nibbles(x[t][w], 1) << 1), (int)(nibbles(x[t][w], 2) << 1),
                              (int)(nibbles(x[t][w], 3) << 1), 0, 0, 0, 0};                      // bit ? 1.0 (0x2) : 0

        if constexpr (ORDERED) {
            // ---- two to four words: the stage in issue order (fold_op / mfma_asm above) --------------------------------
            // One B buffer PER WORD and group parity (2 W fragments in registers): the fragments of group g + 2 are requested
            // while group g + 1 multiplies.
            v4i a4[MT][W];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int w = 0; w < W; ++w) a4[t][w] = v4i{a[t][w][0], a[t][w][1], a[t][w][2], a[t][w][3]};
            v4i bX[W], bY[W];
            auto fetch = [&](v4i (&dst)[W], uint32_t g) {
#pragma unroll
                for (int w = 0; w < W; ++w) dst[w] = lbl[((size_t)g * W + w) * 64];
            };
            // MFMAs of group `nw` (tile 0 and 1 of word 0 first), the fold of group `od` spread behind every further MFMA;
            // returns the lanes with a result within the threshold as a wave mask and the lane's minimum in `m`
            constexpr int GAPS = 2 * W - 1, PER = (16 + GAPS - 1) / GAPS;
            auto stage = [&](Acc& nw, const Acc& od, const v4i (&b)[W], float thr, float& m) -> uint64_t {
                float mA, mB;
                mfma_asm<true>(nw.t[0], a4[0][0], b[0]);
                mfma_asm<true>(nw.t[1], a4[1][0], b[0]);
                fold_ops<0, PER>(mA, mB, od.t[0], od.t[1]);
                if constexpr (W >= 2) { mfma_asm<false>(nw.t[0], a4[0][1], b[1]); fold_ops<PER, 2 * PER>(mA, mB, od.t[0], od.t[1]);
                                        mfma_asm<false>(nw.t[1], a4[1][1], b[1]); fold_ops<2 * PER, 3 * PER>(mA, mB, od.t[0], od.t[1]); }
                if constexpr (W >= 3) { mfma_asm<false>(nw.t[0], a4[0][2], b[2]); fold_ops<3 * PER, 4 * PER>(mA, mB, od.t[0], od.t[1]);
                                        mfma_asm<false>(nw.t[1], a4[1][2], b[2]); fold_ops<4 * PER, 5 * PER>(mA, mB, od.t[0], od.t[1]); }
                if constexpr (W >= 4) { mfma_asm<false>(nw.t[0], a4[0][3], b[3]); fold_ops<5 * PER, 6 * PER>(mA, mB, od.t[0], od.t[1]);
                                        mfma_asm<false>(nw.t[1], a4[1][3], b[3]); fold_ops<6 * PER, 7 * PER>(mA, mB, od.t[0], od.t[1]); }
                uint64_t mask;
                asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(mA), "v"(thr));
                m = mA;
                return mask;
            };
            auto only_mfmas = [&](Acc& nw, const v4i (&b)[W]) {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    if (w == 0) { mfma_asm<true>(nw.t[0], a4[0][0], b[0]); mfma_asm<true>(nw.t[1], a4[1][0], b[0]); }
                    else { mfma_asm<false>(nw.t[0], a4[0][w], b[w]); mfma_asm<false>(nw.t[1], a4[1][w], b[w]); }
                }
                asm volatile("s_nop 7\ns_nop 3" : "+v"(nw.t[0]), "+v"(nw.t[1]));      // results readable by what follows
            };
            auto only_fold = [&](const Acc& od, float thr, float& m) -> uint64_t {
                float mA, mB;
                asm volatile("s_nop 7\ns_nop 3");
                fold_ops<0, 16>(mA, mB, od.t[0], od.t[1]);
                uint64_t mask;
                asm volatile("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(mA), "v"(thr));
                m = mA;
                return mask;
            };
            Acc accX, accY;
            float thrX, thrY = lt[0], mX, mY;
            fetch(bY, 0);
            fetch(bX, 1);                                   // groups >= 2 (mfma_groups_per_chunk)
            thrX = lt[32];
            only_mfmas(accY, bY);
            uint32_t g = 0;
#pragma unroll 1
            for (; g + 2 < groups; g += 2) {
                fetch(bY, g + 2);
                const float thrYn = lt[(g + 2) * 32];
                if (const uint64_t mk = stage(accX, accY, bX, thrY, mY); __builtin_expect(mk != 0, 0)) save_hits(accY, mk, thrY, g, step);
                thrY = thrYn;
                const uint32_t g3 = g + 3 < groups ? g + 3 : g + 2;
                fetch(bX, g3);
                const float thrXn = lt[g3 * 32];
                if (const uint64_t mk = stage(accY, accX, bY, thrX, mX); __builtin_expect(mk != 0, 0)) save_hits(accX, mk, thrX, g + 1, step);
                thrX = thrXn;
            }
            if (g + 1 < groups) {
                if (const uint64_t mk = stage(accX, accY, bX, thrY, mY); mk != 0) save_hits(accY, mk, thrY, g, step);
                if (const uint64_t mk = only_fold(accX, thrX, mX); mk != 0) save_hits(accX, mk, thrX, g + 1, step);
            } else {
                if (const uint64_t mk = only_fold(accY, thrY, mY); mk != 0) save_hits(accY, mk, thrY, g, step);
            }
        } else {
        // Software pipeline over the (group, word) sequence: two B buffers (one word each) and two accumulator sets.
        // The fragment of the NEXT word is requested right after the MFMAs of the current one are issued (its buffer
        // was consumed one stage earlier), and the results of group g are folded while the MFMAs of group g + 1 run.
        v4i bx = lbl[0], by = lbl[0];
        Acc accX, accY;
        float thrX = 0.f, thrY = lt[0];
        // stage(w): consume one buffer, prefetch fragment `nxt` (counted from the pair's base pointer, so that the offsets
        // are immediates of the ds_read and one pointer increment serves two groups) into the other one
        auto stage = [&](Acc& acc, lds_frag_ptr base, int nxt, int w, bool y_buf, bool more) {
            if (y_buf) {
                mm(acc, w, by);
                if (more) bx = base[nxt * 64];
            } else {
                mm(acc, w, bx);
                if (more) by = base[nxt * 64];
            }
        };
        // group 0
#pragma unroll
        for (int w = 0; w < W; ++w) stage(accY, (lds_frag_ptr)lbl, w + 1, w, (w & 1) == 0, true);
        // LDS addresses of the pair (g, g + 1): 32-bit pointers advanced by hand and laundered, or hipcc rebuilds both
        // from g with a shift-add per group (two more vector instructions per pair in a loop that is issue-bound)
        lds_frag_ptr lg = (lds_frag_ptr)lbl + W * 64;
        lds_thr_ptr ltg = (lds_thr_ptr)lt + 32;
#pragma unroll 1
        for (uint32_t g = 1; g + 1 < groups; g += 2, lg += 2 * W * 64, ltg += 64) {
            asm volatile("" : "+v"(lg), "+v"(ltg));
            // odd group g -> accX; its first word sits in buffer parity (W & 1): Y when W is even
            thrX = ltg[0];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                stage(accX, lg, w + 1, w, ((W + w) & 1) == 0, true);
                if (w == 0) { pin2(accX, accY); reduce(accY, thrY, g - 1, step); }
            }
            // even group g + 1 -> accY; (2 * W + w) & 1 == w & 1
            thrY = ltg[32];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                stage(accY, lg, W + w + 1, w, (w & 1) == 0, true);
                if (w == 0) { pin2(accY, accX); reduce(accX, thrX, g, step); }
            }
        }
        // last (odd) group: nothing further to prefetch after its last word
        {
            const uint32_t g = groups - 1;
            thrX = ltg[0];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                stage(accX, lg, w + 1, w, ((W + w) & 1) == 0, w + 1 < W);
                if (w == 0) { pin2(accX, accY); reduce(accY, thrY, g - 1, step); }
            }
            reduce(accX, thrX, g, step);
        }
        }       // (!ORDERED)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w) x[t][w] = xn[t][w];
        // what the PREVIOUS step's appends returned (issued a whole step ago: no wait), then this step's saved blocks
        if constexpr (MODE == MODE_SELF) pend_complete();
        if (rcount) process_ring(step);
        if constexpr (MODE == MODE_SELF) {
            if (chk_what != ~0u && chk_count >= p.k) {
                const uint32_t ql = chk_what & 0xFFFFu;
                lower_threshold(p.thr_live + q0 + ql, (float)((int)(chk_what >> 16) - lpop[ql]));
            }
            if (refresh_now) *reinterpret_cast<float4*>(lthr + fresh_index()) = make_float4(fresh[0], fresh[1], fresh[2], fresh[3]);
        }
    }
    if constexpr (MODE == MODE_SELF) pend_complete();
}


// =====================================================================================================================
// mfma_pack_kernel -- 64-bit codes (W = 1): TWO row tiles share ONE accumulator, folded as packed f16
// =====================================================================================================================
// The kernel above looks at every (row, query) result once: 15 v_min3_f32 + v_min_f32 + v_cmp per 32 queries x 64 rows,
// 17 vector instructions beside 64 cycles of matrix-pipe time -- vector ISSUE bound it (pipe 0.485 busy, round 2).  Here the
// first MFMA of a tile pair adds its dot products (|d| <= 64) to a constant block C = 2^23 + 0x402000 and the second one is
// the block-SCALED form with scale 2^16 accumulating into the same registers:
//
//     bits(acc) = 0x4B402000 + d1 + 65536 * d2        (an f32 in [2^23, 2^24): ulp = 1, every partial sum an exact integer)
//
// so the LOW half of every register is 0x2000 + d1 and the HIGH half 0x4B40 + d2: positive, normal f16 bit patterns, whose
// order as f16 is their order as integers.  v_pk_minimum3_f16 (new in gfx950) folds FOUR results per instruction, and the
// query's packed threshold T = (first NON-hit pattern of each half) rides in the same fold: "some result <= thr" <=> fold != T.
// A wave owns FOUR tiles (128 rows, two accumulators): 16 fold instructions + 1 compare + 2 scale loads beside four MFMAs
// (128 cycles) -- the matrix pipe is the bound again (prototype: tools/proto_pack_scan.hip, profiles/r03_proto_pack_scan.txt).
//
// The stage (MFMAs of group g + 1 around the fold of group g) is inline assembly in ISSUE ORDER -- MFMA, four fold
// instructions, MFMA, ... -- because hipcc moved the builtin MFMAs across the fold and the hit branch whatever
// sched_barrier said.  hipcc inserts NO hazard nops for assembly, so the distances are kept by construction and counted in
// INSTRUCTIONS (one wait state each, the rule hipcc itself applies; an 8-pass MFMA result may be read by the VALU 11 wait
// states after the MFMA): the first eight fold instructions touch accumulator 0 of the old group only (last written by the
// THIRD MFMA of the previous stage, >= 20 instructions back), accumulator 1 comes after that (>= 20 back as well); a
// v_pk_minimum3_f16 is never followed directly by a consumer of its result (two interleaved chains; s_nop 0 before the join
// and the compare); the first and the last group of a step, whose MFMAs / fold stand alone, are padded with s_nop.
//
// d2 = +64 (query == 0 against a row of all ones) would carry into the exponent and halve the resolution of the low half:
// the host routes a batch holding an all-zero 64-bit query to the kernel above (Batch::begin, isccsearch.hip).
#ifndef ISK_PACK_CARRY
#define ISK_PACK_CARRY 1       // experiment switch: 0 = the general loop never carries its accumulators across steps
#endif
#ifndef ISK_PACK_STRETCH
#define ISK_PACK_STRETCH 1
#endif
constexpr int PK_TILES = 4;                                             // row tiles per wave and step
constexpr uint32_t PK_DEEP_GROUPS = 4;                                  // chunks of up to this many groups (128 queries): one instantiation per count,
                                                                        // four steps of rows in flight, accumulators carried across steps (five and six fit 168 registers no more)
__device__ __forceinline__ uint32_t pkmin3(uint32_t a, uint32_t b, uint32_t c) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 x = __builtin_bit_cast(h2, a), y = __builtin_bit_cast(h2, b), z = __builtin_bit_cast(h2, c);
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_minimum(__builtin_elementwise_minimum(x, y), z));
}
// The A operand of 32 bits of a row (mfma_pack_kernel): nibble t of dword j stands for bit j + 4 t, kept IN PLACE -- the codes
// 0x1, 0x2, 0x4 are e2m1 1/2, 1, 2 (bit 3 would be the sign: that dword moves down one bit) -- and the query fragments carry the
// reciprocal magnitudes (prologue).  5 vector instructions instead of the 7 of "(x >> j) & 0x11111111, << 1".
__device__ __forceinline__ v4i pk_rows(uint32_t x) {
    return v4i{(int)(x & 0x11111111u), (int)(x & 0x22222222u), (int)(x & 0x44444444u), (int)((x & 0x88888888u) >> 1)};
}
__device__ __forceinline__ uint32_t live_packed(const float* addr) {
    return (uint32_t)__hip_atomic_load(reinterpret_cast<const int*>(addr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ISK_EXP_PACK (experiment switch, never in the product build; tools/pack_step_accounting.sh): what the parts of a step cost, by
// leaving them out -- results are WRONG, only the launch time and the counters of the kernel mean anything.
//   bit 0: no fold (the 16 v_pk_minimum3_f16 + v_cmp of a stage: no row is ever a candidate)
//   bit 1: the rows are loaded but expanded only once, before the loop
//   bit 2: no looks (threshold refresh, checkers)
//   bit 3: see ISK_MF2
#ifndef ISK_EXP_PACK
#define ISK_EXP_PACK 0
#endif
#if ISK_EXP_PACK & 1
#define ISK_PKM "; v_pk_minimum3_f16 "
#define ISK_CMP "s_mov_b64 %[mask], 0\n"
#else
#define ISK_PKM "v_pk_minimum3_f16 "
#define ISK_CMP "v_cmp_ne_u32_e64 %[mask], %[t], %[mA]\n"
#endif
#define ISK_MF1(n, av) "v_mfma_f32_32x32x64_f8f6f4 %[" #n "], %[" #av "], %[b], %[mg] cbsz:4 blgp:4\n"
#if ISK_EXP_PACK & 8       // bit 3: the second tile of an accumulator multiplied WITHOUT its block scale (no v_mfma_ld_scale_b32 prefix)
#define ISK_MF2(n, av) "v_mfma_f32_32x32x64_f8f6f4 %[" #n "], %[" #av "], %[b], %[" #n "] cbsz:4 blgp:4\n"
#else
#define ISK_MF2(n, av) "v_mfma_scale_f32_32x32x64_f8f6f4 %[" #n "], %[" #av "], %[b], %[" #n "], %[sh], %[so] op_sel_hi:[0,0,0] cbsz:4 blgp:4\n"
#endif

// DEPTH: steps whose rows a wave keeps in flight.  A chunk of 32 groups works ~5 000 cycles on a step's 1 KB of rows and one
// step ahead hides any latency; a chunk of one or two groups is done in ~400, and with one step (3 waves x 4 SIMDs x 1 KB =
// 12 KB per CU) in flight the scan crawled at 2.5 TB/s, bound by memory latency (32 queries: 0.33 ms per 100 M rows,
// profiles/r03_step_timelines.txt).  Small chunks run the DEPTH = 4 instantiation: the step body four times per loop trip,
// each on its own row registers.
// G: 0 = any number of groups, fragments from LDS, a step's groups pipelined among themselves; 1 / 2 = a chunk of exactly that
// many groups (<= 64 queries), fragments in registers, pipelined ACROSS steps (`few_step` below)
template <int MODE, int DEPTH, int G>
__global__ __launch_bounds__(MBLOCK, 3) void mfma_pack_kernel(const ScanParams p, const uint32_t groups) {
    constexpr int MT = PK_TILES;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i* lb = reinterpret_cast<v4i*>(smem);
    uint32_t* lthr = reinterpret_cast<uint32_t*>(smem + (size_t)groups * 64 * 16);      // packed thresholds
    int* lpop = reinterpret_cast<int*>(lthr + groups * 32);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t q0 = blockIdx.y * groups * 32;

    // prologue: as mfma_scan_kernel<1>, thresholds packed
    for (uint32_t i = tid; i < groups * 32 * 2; i += MBLOCK) {
        const uint32_t ql = i >> 1, hh = i & 1;
        const uint32_t q = q0 + ql;
        // padding queries (bias BIAS_NEVER: beyond the batch's real queries) are all-zero words: as live queries they would score
        // +64 against a row of all ones, the one value the packed high half cannot hold -- their fragment is ZERO (every dot 0)
        const bool live = q < p.nq_pad && p.bias[q] != BIAS_NEVER;
        const uint64_t qw = live ? p.queries[(uint64_t)q * 4] : 0;
        const uint32_t x = hh ? (uint32_t)(qw >> 32) : (uint32_t)qw;
        const uint32_t m = live ? (hh ? p.mask_hi : p.mask_lo) : 0u;
        const uint32_t g = ql >> 5, c = ql & 31;
        v4i frag;
#pragma unroll
        // +-v with v = 2, 1, 1/2, 1/2 for the four dwords (e2m1 0x4, 0x2, 0x1, 0x1; sign = bit 3): the ROW nibbles of dword j are
        // 1/2, 1, 2, 2 (pk_rows), so that every product is +-1 and three of a row's four dwords cost ONE v_and each
        for (int j = 0; j < 4; ++j) frag[j] = (int)(((j == 0 ? 0x44444444u : j == 1 ? 0x22222222u : 0x11111111u) | (nibbles(x, j) << 3)) & (nibbles(m, j) * 0xFu));
        lb[(size_t)g * 64 + hh * 32 + c] = frag;
    }
    for (uint32_t ql = tid; ql < groups * 32; ql += MBLOCK) {
        const uint32_t q = q0 + ql;
        int pc = 0, tau = -1;
        if (q < p.nq_pad) {
            pc = __popcll(p.queries[(uint64_t)q * 4] & (((uint64_t)p.mask_hi << 32) | p.mask_lo));
            tau = (int)(0x7FFFFFFFu - p.bias[q]);
        }
        lpop[ql] = pc;
        if constexpr (MODE == MODE_SELF) lthr[ql] = q < p.nq_pad ? live_packed(p.thr_live + q) : 0u;     // the boot kernel wrote them packed
        else lthr[ql] = pack_threshold(tau - pc);
    }
    __syncthreads();

    const uint64_t first = p.row_begin / (32 * MT);                         // row_begin is a multiple of the XOR kernel's tile (>= 512 rows)
    const uint64_t nsteps = (p.n_rows + 32 * MT - 1) / (32 * MT);
    const uint64_t stride = (uint64_t)gridDim.x * (MBLOCK / 64);
    // the wave number as a SCALAR: step number and row addresses then live on the scalar unit (scalar-base loads)
    // (DEPTH > 1: a wave owns STRETCH consecutive steps at a time -- its loads in flight are DEPTH KB of one stretch of rows)
    constexpr int STRETCH = ISK_PACK_STRETCH ? DEPTH : 1;
    uint64_t step = first + ((uint64_t)blockIdx.x * (MBLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)wave)) * STRETCH;
    if (step >= nsteps) return;
    const uint64_t last_row = p.n_rows - 1;
    const uint32_t* const col32 = reinterpret_cast<const uint32_t*>(p.col[0]);

    struct Acc { v16f t[2]; };
    // ---- candidates ------------------------------------------------------------------------------------------------------------
    // A stage whose fold differs from T in some lane costs the hot loop NO global memory operation and no unrolled search: the
    // lanes that hold a hit copy their 32 accumulator registers (+ query, threshold) into their wave's LDS ring -- slots from
    // the compare's own lane mask, so the count stays wave-uniform -- and the stage loop goes on.  At the end of the step (or
    // when the ring is full) `process_ring` walks the saved blocks with a REAL loop: lane v looks at result v of a block (64
    // results: 32 registers x 2 halves), so the search for the hit is one compare per lane instead of 64 unrolled compares
    // with a branch each, and every hit of a block is appended by its own lane with all atomics in flight together.  Their
    // results are consumed by `Pending` at the lane's next hit or at the end of the NEXT step, when they (and the row prefetch,
    // which shares the in-order vmcnt) have long arrived.  Before: the 64 compares and two returned atomics per candidate sat in
    // the stage loop behind a vmcnt(0) that also waited for the row prefetch -- ~1.9 us of wave time per candidate (k = 100:
    // 3.5 ms per 100 M x 1 024 pass against 2.3 for k = 1, profiles/r03_ab_self.txt).
    constexpr uint32_t RING_E = PK_RING_ENTRIES, ENTRY = PK_RING_ENTRY_DWORDS;       // dwords: 32 registers | query, lane half | T | pad
    uint32_t* const ring = reinterpret_cast<uint32_t*>(lpop + groups * 32) + wave * (RING_E * ENTRY);
    uint32_t rcount = 0;                        // saved blocks in the ring (wave-uniform)
    // MODE_SELF: a candidate's list slot is requested here and its word stored once the slot is known -- at the lane's next
    // candidate or at the end of the next step.  Its distance counts are NO-RETURN atomics (count[q][t] += 1 for every t in
    // [hamming, threshold the compare ran under)): nobody waits for them; "k rows within t" is noticed by the CHECKER lanes
    // below, which read one counter per step each and lower the live threshold with one atomicMin on the packed word.
    // (The lane whose own increment crossed k used to do that: two returned atomics and a dependent chain of further ones per
    //  candidate, ~3.6 us of wave time each -- the level design beat the single pass by 10 % at 100 M rows and by 40 % at 12.5 M.)
    uint32_t pend_slot = 0, pend_lo = 0, pend_hi = 0x80000000u;        // pend_hi bit 31: nothing pending
    auto pend_complete = [&]() __attribute__((always_inline)) {
        if (!(pend_hi & 0x80000000u)) {
            const uint32_t qi = q0 + (pend_hi >> 20);                   // query in chunk : 11 | hamming : 7 | row >> 32 : 12 ... see issue below
            if (pend_slot < p.cap) p.cand[(uint64_t)qi * p.cap + pend_slot] = ((uint64_t)((pend_hi >> 12) & 0x7Fu) << 48) | ((uint64_t)(pend_hi & 0xFFFu) << 32) | pend_lo;
            pend_hi = 0x80000000u;
        }
    };
    auto process_ring = [&](uint64_t st) __attribute__((always_inline)) {
        const uint32_t reg = lane >> 1, hf = lane & 1;
        // result `reg` of half `hf` is tile 2 (reg >> 4) + hf, matrix row (reg & 3) + 8 ((reg & 15) >> 2) + 4 (lane >> 5 of the
        // saving lane); tile t, matrix row m is row 64 (t & 1) + (t >> 1) + 2 m of the step (`expand`)
        const uint32_t off0 = 64 * hf + (reg >> 4) + 2 * ((reg & 3) + 8 * ((reg & 15) >> 2));
        for (uint32_t e = 0; e < rcount; ++e) {
            const uint32_t* const blk = ring + e * ENTRY;
            const uint32_t bits = blk[reg], head = blk[32], tpk = blk[33];
            const uint32_t ql = head & 0xFFFFu, off = off0 + 8 * (head >> 16);
            const bool below = hf ? bits < (tpk & 0xFFFF0000u) : (bits & 0xFFFFu) < (tpk & 0xFFFFu);
            const int d = hf ? (int)(bits >> 16) - (int)PK_HI0 : (int)(bits & 0xFFFFu) - (int)PK_LO0;
            const uint64_t row = st * (32 * MT) + off;
            // (d < -64 is no dot product of 64 bits: never turned into an index)
            if (below && d >= -64 && row <= last_row) {
                const int pc = lpop[ql];
#if ISK_EXP_NO_CANDIDATE_MEMORY
                // EXPERIMENT (results are wrong): what does the candidate path cost WITHOUT its global memory operations?
                asm volatile("" ::"v"(pc), "v"(d), "v"(row));
                continue;
#endif
                if constexpr (MODE == MODE_SELF) {
                    pend_complete();
                    const uint32_t qi = q0 + ql, hd = (uint32_t)(d + pc);
                    const int tau_seen = unpack_threshold(tpk) + pc;
                    pend_slot = atomicAdd(&p.cnt[(uint64_t)qi * CNT_STRIDE], 1u);
                    uint32_t* const counts = p.ghist + (uint64_t)qi * HB;
                    for (int t = (int)hd; t < tau_seen; ++t) __hip_atomic_fetch_add(&counts[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // result unused: no-return form
                    pend_lo = (uint32_t)row;
                    pend_hi = (ql << 20) | (hd << 12) | (uint32_t)(row >> 32);       // rows < 2^44
                } else {
                    emit<MODE>(p, q0 + ql, (uint32_t)(d + pc), row);
                }
            }
        }
        rcount = 0;
    };
    // `mask`: the lanes whose fold differs from T (query g * 32 + (lane & 31), rows 4 * (lane >> 5) + ... of the step's tiles)
    auto save_hits = [&](const Acc& acc, uint64_t mask, uint32_t tpk, uint32_t g, uint64_t st) __attribute__((always_inline)) {
        while (mask) {                              // wave-uniform; more than one trip only when the ring fills up
            const uint32_t room = RING_E - rcount;
            if (room == 0) { process_ring(st); continue; }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const bool mine = ((mask >> lane) & 1) != 0 && rank < room;
            if (mine) {
                uint32_t* const blk = ring + (rcount + rank) * ENTRY;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int i = 0; i < 16; i += 4)
                        *reinterpret_cast<float4*>(blk + 16 * j + i) = make_float4(acc.t[j][i], acc.t[j][i + 1], acc.t[j][i + 2], acc.t[j][i + 3]);
                *reinterpret_cast<uint2*>(blk + 32) = make_uint2((g * 32 + r) | (h << 16), tpk);
            }
            const uint64_t taken = __builtin_amdgcn_ballot_w64(mine);
            rcount += (uint32_t)__builtin_popcountll(taken);
            mask &= ~taken;
        }
    };

    const float mgf = __uint_as_float(PK_MAGIC);
    v16f magic = {mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf, mgf};
    asm volatile("" : "+v"(magic));                        // ONE register block for the whole kernel
    int sc_hi = (int)0x8F8F8F8F, sc_one = 0x7F7F7F7F;      // E8M0 block scales 2^16 and 2^0
    asm volatile("" : "+v"(sc_hi), "+v"(sc_one));
    v4i a[MT];

    // stage: MFMAs of the NEW group into `nw`, fold of the OLD group `od` with its packed threshold; returns the lanes whose
    // fold differs from T as a wave mask and the fold in `m`
    auto stage = [&](Acc& nw, const Acc& od, const v4i& b, uint32_t tpk, uint32_t& m) __attribute__((always_inline)) -> uint64_t {
        uint32_t mA, mB;
        uint64_t mask;
        const v16f& o0 = od.t[0];
        const v16f& o1 = od.t[1];
        asm volatile(ISK_MF1(n0, a0)
                     ISK_PKM "%[mA], %[t], %[u0], %[u1]\n" ISK_PKM "%[mB], %[u8], %[u9], %[u10]\n"
                     ISK_PKM "%[mA], %[mA], %[u2], %[u3]\n" ISK_PKM "%[mB], %[mB], %[u11], %[u12]\n"
                     : [n0] "=&v"(nw.t[0]), [mA] "=&v"(mA), [mB] "=&v"(mB)
                     : [a0] "v"(a[0]), [b] "v"(b), [mg] "v"(magic), [t] "v"(tpk), [u0] "v"(o0[0]), [u1] "v"(o0[1]), [u2] "v"(o0[2]), [u3] "v"(o0[3]),
                       [u8] "v"(o0[8]), [u9] "v"(o0[9]), [u10] "v"(o0[10]), [u11] "v"(o0[11]), [u12] "v"(o0[12]));
        asm volatile(ISK_MF1(n1, a2)
                     ISK_PKM "%[mA], %[mA], %[u4], %[u5]\n" ISK_PKM "%[mB], %[mB], %[u13], %[u14]\n"
                     ISK_PKM "%[mA], %[mA], %[u6], %[u7]\n" ISK_PKM "%[mB], %[mB], %[u15], %[w8]\n"
                     : [n1] "=&v"(nw.t[1]), [mA] "+v"(mA), [mB] "+v"(mB)
                     : [a2] "v"(a[2]), [b] "v"(b), [mg] "v"(magic), [u4] "v"(o0[4]), [u5] "v"(o0[5]), [u6] "v"(o0[6]), [u7] "v"(o0[7]),
                       [u13] "v"(o0[13]), [u14] "v"(o0[14]), [u15] "v"(o0[15]), [w8] "v"(o1[8]));
        asm volatile(ISK_MF2(n0, a1)
                     ISK_PKM "%[mA], %[mA], %[w0], %[w1]\n" ISK_PKM "%[mB], %[mB], %[w9], %[w10]\n"
                     ISK_PKM "%[mA], %[mA], %[w2], %[w3]\n" ISK_PKM "%[mB], %[mB], %[w11], %[w12]\n"
                     : [n0] "+v"(nw.t[0]), [mA] "+v"(mA), [mB] "+v"(mB)
                     : [a1] "v"(a[1]), [b] "v"(b), [sh] "v"(sc_hi), [so] "v"(sc_one), [w0] "v"(o1[0]), [w1] "v"(o1[1]), [w2] "v"(o1[2]), [w3] "v"(o1[3]),
                       [w9] "v"(o1[9]), [w10] "v"(o1[10]), [w11] "v"(o1[11]), [w12] "v"(o1[12]));
        asm volatile(ISK_MF2(n1, a3)
                     ISK_PKM "%[mA], %[mA], %[w4], %[w5]\n" ISK_PKM "%[mB], %[mB], %[w13], %[w14]\n"
                     ISK_PKM "%[mA], %[mA], %[w6], %[w7]\n"
                     "s_nop 0\n"
                     ISK_PKM "%[mA], %[mA], %[mB], %[w15]\n"
                     "s_nop 0\n"
                     ISK_CMP
                     : [n1] "+v"(nw.t[1]), [mA] "+v"(mA), [mB] "+v"(mB), [mask] "=s"(mask)
                     : [a3] "v"(a[3]), [b] "v"(b), [sh] "v"(sc_hi), [so] "v"(sc_one), [t] "v"(tpk), [w4] "v"(o1[4]), [w5] "v"(o1[5]), [w6] "v"(o1[6]), [w7] "v"(o1[7]),
                       [w13] "v"(o1[13]), [w14] "v"(o1[14]), [w15] "v"(o1[15]));
        m = mA;
        return mask;
    };
    // the four MFMAs of a step's FIRST group (nothing to fold beside them), padded so that the first stage may read them
    auto first_group = [&](Acc& nw, const v4i& b) __attribute__((always_inline)) {
        asm volatile(ISK_MF1(n0, a0) ISK_MF1(n1, a2) ISK_MF2(n0, a1) ISK_MF2(n1, a3) "s_nop 7\ns_nop 3\n"
                     : [n0] "=&v"(nw.t[0]), [n1] "=&v"(nw.t[1])
                     : [a0] "v"(a[0]), [a1] "v"(a[1]), [a2] "v"(a[2]), [a3] "v"(a[3]), [b] "v"(b), [mg] "v"(magic), [sh] "v"(sc_hi), [so] "v"(sc_one));
    };
    // the fold of a step's LAST group: same order as in a stage, no MFMA beside it
    auto last_fold = [&](const Acc& od, uint32_t tpk, uint32_t& m) __attribute__((always_inline)) -> uint64_t {
        uint32_t mA, mB;
        uint64_t mask;
        const v16f& o0 = od.t[0];
        const v16f& o1 = od.t[1];
        asm volatile("s_nop 3\n"
                     ISK_PKM "%[mA], %[t], %[u0], %[u1]\n" ISK_PKM "%[mB], %[u8], %[u9], %[u10]\n"
                     ISK_PKM "%[mA], %[mA], %[u2], %[u3]\n" ISK_PKM "%[mB], %[mB], %[u11], %[u12]\n"
                     ISK_PKM "%[mA], %[mA], %[u4], %[u5]\n" ISK_PKM "%[mB], %[mB], %[u13], %[u14]\n"
                     ISK_PKM "%[mA], %[mA], %[u6], %[u7]\n"
                     : [mA] "=&v"(mA), [mB] "=&v"(mB)
                     : [t] "v"(tpk), [u0] "v"(o0[0]), [u1] "v"(o0[1]), [u2] "v"(o0[2]), [u3] "v"(o0[3]), [u4] "v"(o0[4]), [u5] "v"(o0[5]), [u6] "v"(o0[6]), [u7] "v"(o0[7]),
                       [u8] "v"(o0[8]), [u9] "v"(o0[9]), [u10] "v"(o0[10]), [u11] "v"(o0[11]), [u12] "v"(o0[12]), [u13] "v"(o0[13]), [u14] "v"(o0[14]));
        asm volatile(ISK_PKM "%[mB], %[mB], %[u15], %[w8]\n"
                     ISK_PKM "%[mA], %[mA], %[w0], %[w1]\n" ISK_PKM "%[mB], %[mB], %[w9], %[w10]\n"
                     ISK_PKM "%[mA], %[mA], %[w2], %[w3]\n" ISK_PKM "%[mB], %[mB], %[w11], %[w12]\n"
                     ISK_PKM "%[mA], %[mA], %[w4], %[w5]\n" ISK_PKM "%[mB], %[mB], %[w13], %[w14]\n"
                     ISK_PKM "%[mA], %[mA], %[w6], %[w7]\n"
                     "s_nop 0\n"
                     ISK_PKM "%[mA], %[mA], %[mB], %[w15]\n"
                     "s_nop 0\n"
                     ISK_CMP
                     : [mA] "+v"(mA), [mB] "+v"(mB), [mask] "=s"(mask)
                     : [t] "v"(tpk), [u15] "v"(o0[15]), [w0] "v"(o1[0]), [w1] "v"(o1[1]), [w2] "v"(o1[2]), [w3] "v"(o1[3]), [w4] "v"(o1[4]), [w5] "v"(o1[5]), [w6] "v"(o1[6]),
                       [w7] "v"(o1[7]), [w8] "v"(o1[8]), [w9] "v"(o1[9]), [w10] "v"(o1[10]), [w11] "v"(o1[11]), [w12] "v"(o1[12]), [w13] "v"(o1[13]), [w14] "v"(o1[14]), [w15] "v"(o1[15]));
        m = mA;
        return mask;
    };

    const v4i* const lbl = lb + lane;
    const uint32_t* const lt = lthr + r;
    // A step is 128 rows = 1 KB, ONE 16-byte load per lane: lane L holds rows 2 L and 2 L + 1 of the step (x, y | z, w).  The
    // matrix-core operand wants a row's two dwords in lanes m and m + 32: v_permlane32_swap (gfx950) trades the upper half of
    // one register with the lower half of another, so swap(x, y) yields TWO tiles at once -- rows 2 m (lanes < 32 kept their x,
    // lanes >= 32 received y of lane m) and rows 2 m + 64 -- and swap(z, w) the tiles of rows 2 m + 1 and 2 m + 65.  Before:
    // four 4-byte loads per lane and step, four times the address work of the texture path for the same bytes.
    auto load_rows = [&](uint64_t st) __attribute__((always_inline)) -> u32x4 {          // compiler-scheduled: DEPTH == 1, and the table's partial last step
        if ((st + 1) * (32 * MT) <= p.n_rows) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(col32 + st * (64 * MT)) + lane);
        const uint64_t r0 = st * (32 * MT) + 2 * lane, r1 = r0 + 1;
        const uint2 lo = *reinterpret_cast<const uint2*>(col32 + (r0 <= last_row ? r0 : last_row) * 2);
        const uint2 hi = *reinterpret_cast<const uint2*>(col32 + (r1 <= last_row ? r1 : last_row) * 2);
        return u32x4{lo.x, lo.y, hi.x, hi.y};
    };
    auto expand = [&](const u32x4& v) __attribute__((always_inline)) {
        const auto e = __builtin_amdgcn_permlane32_swap(v[0], v[1], false, false);
        const auto o = __builtin_amdgcn_permlane32_swap(v[2], v[3], false, false);
        a[0] = pk_rows(e[0]);
        a[1] = pk_rows(e[1]);
        a[2] = pk_rows(o[0]);
        a[3] = pk_rows(o[1]);
    };
    // DEPTH > 1: the loads are issued from inline asm and retired by COUNTED waits, so that DEPTH steps of rows stay in flight.
    // (Left to hipcc, every step began with s_waitcnt vmcnt(0) and a copy of the whole row-register array: nothing was in flight
    //  while a step computed, and 17..64 queries scanned at 3.4 TB/s.)  Same discipline as load_tile_asm (kernels.hip.h), checked
    // at build time by tools/audit_kernels.py: nothing touches a destination between its load and its wait; `s_nop 4` in front
    // (a VALU-written SGPR base needs 5 wait states before a VMEM instruction reads it); no spills in this kernel.
    const uint32_t lane_bytes = lane * 16;
    auto issue_rows = [&](u32x4& dst, uint64_t st) __attribute__((always_inline)) {
        const uint64_t addr = reinterpret_cast<uint64_t>(col32) + st * (256 * MT);
        // (the step number is wave-uniform by construction; pinned to SGPRs here in case hipcc moved its arithmetic to the VALU)
        const unsigned char* const base = reinterpret_cast<const unsigned char*>(
            ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(addr >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr));
        // ("+v": a slot is ONE register quadruple for the whole kernel, updated in place -- as a fresh output per load, hipcc gave
        //  some instantiations' slots different registers in the loop and rotated them with copies at the back edge, in flight)
        // (`nt`: the rows are read once -- 9..32 queries scanned at 5.2 TB/s without the hint, 5.8 with it: 0.153 -> 0.139 ms per 100 M rows,
        //  profiles/r04_pmc_sq_small_chunks.txt; the XOR + popcount kernel's loads carry it too)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 nt\n\t" : "+v"(dst) : "v"(lane_bytes), "s"(base) : "memory");
    };
    auto await_rows = [&](u32x4& v) __attribute__((always_inline)) {                     // DEPTH - 1 younger row loads are outstanding at every wait
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DEPTH - 1) : "memory");
        asm volatile("" : "+v"(v));
    };
    const uint32_t wave_s = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    // thresholds [256 wave, 256 wave + 256) are this wave's to keep fresh, four per lane.  The lane number is RECOMPUTED at
    // each use (mbcnt of a laundered zero), or hipcc keeps a 64-bit global address and an LDS address alive through the
    // group loop for two instructions per look -- registers the rare path needs (168 with them: one spilled)
    auto fresh_index = [&]() __attribute__((always_inline)) {
        uint32_t z = 0;
        asm volatile("" : "+v"(z));
        return wave_s * 256 + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z)) * 4;
    };
    const bool refresh = MODE == MODE_SELF && wave * 256 + lane * 4 < groups * 32 && q0 + wave * 256 + lane * 4 < p.nq_pad;
    // a step is 128 rows here (64 in mfma_scan_kernel): look half as many steps apart for the same rows per look
    const uint32_t refresh_mask = (groups >= 32 ? 1u : groups >= 16 ? 2u : groups >= 8 ? 4u : groups >= 4 ? 8u : 16u) * ((p.refresh_steps + 1) / 2) - 1u;
    // CHECKERS (MODE_SELF): task (query, j) asks "do k appended rows lie within tau_q - j?", j = 1..4, by reading count[q][tau_q - j]
    // at the start of a look step and lowering the live threshold at its end.  64 tasks per wave and look: the first
    // `slices` waves of the grid take one slice each; a grid with fewer waves rotates through the slices step by step.
    const uint32_t slices = groups * 32 * 4 / 64, nwaves = gridDim.x * (MBLOCK / 64);
    const uint32_t gw = blockIdx.x * (MBLOCK / 64) + wave_s;
    uint32_t trip = 0;
    Acc accX, accY;
    auto all_groups = [&]() __attribute__((always_inline)) {
        // two B buffers and two accumulator sets: the fragment of group g + 2 is requested while group g + 1 multiplies
        // (any number of groups: pairs of stages while at least three groups remain, then one stage + the last fold for an even
        //  rest or the last fold alone for an odd one -- 17..32 queries are ONE group, not one and a padding group)
        v4i by = lbl[0], bx = lbl[groups > 1 ? 64 : 0];
        uint32_t thrY = lt[0], thrX = lt[groups > 1 ? 32 : 0];
        uint32_t mY, mX;
        first_group(accY, by);
        uint32_t g = 0;
#pragma unroll 1
        for (; g + 2 < groups; g += 2) {
            by = lbl[(g + 2) * 64];                       // consumed by the stage before
            const uint32_t thrYn = lt[(g + 2) * 32];
            if (const uint64_t mk = stage(accX, accY, bx, thrY, mY); __builtin_expect(mk != 0, 0)) save_hits(accY, mk, thrY, g, step);
            thrY = thrYn;
            const uint32_t g3 = g + 3 < groups ? g + 3 : g + 2;       // (an odd count has no group g + 3: nothing is multiplied with it)
            bx = lbl[g3 * 64];
            const uint32_t thrXn = lt[g3 * 32];
            if (const uint64_t mk = stage(accY, accX, by, thrX, mX); __builtin_expect(mk != 0, 0)) save_hits(accX, mk, thrX, g + 1, step);
            thrX = thrXn;
        }
        if (g + 1 < groups) {
            if (const uint64_t mk = stage(accX, accY, bx, thrY, mY); mk != 0) save_hits(accY, mk, thrY, g, step);
            if (const uint64_t mk = last_fold(accX, thrX, mX); mk != 0) save_hits(accX, mk, thrX, g + 1, step);
        } else {
            if (const uint64_t mk = last_fold(accY, thrY, mY); mk != 0) save_hits(accY, mk, thrY, g, step);
        }
    };
    // The general loop with the same carry across steps, for EVEN group counts (an odd count would trade the roles of the two
    // accumulator sets from step to step: two copies of the loop, which do not fit 168 registers): the step's first stage
    // multiplies group 0 into accY while it folds what the PREVIOUS step left in accX (its last group) -- `first_group` and
    // `last_fold` stood alone for 128 + 12 and ~80 cycles per step and wave, 5 % of a 32-group step.
    bool carried = false;                                 // accX holds the previous step's last group
    auto even_groups = [&](uint64_t prev) __attribute__((always_inline)) {
        v4i by = lbl[0], bx = lbl[64];
        uint32_t thrY = lt[0], thrX = lt[32];
        const uint32_t thrL = lt[(groups - 1) * 32];
        uint32_t mY, mX;
        if (const uint64_t mk = stage(accY, accX, by, thrL, mX); __builtin_expect(carried && mk != 0, 0)) { save_hits(accX, mk, thrL, groups - 1, prev); process_ring(prev); }
        uint32_t g = 0;
#pragma unroll 1
        for (; g + 2 < groups; g += 2) {
            by = lbl[(g + 2) * 64];                       // consumed by the stage before
            const uint32_t thrYn = lt[(g + 2) * 32];
            if (const uint64_t mk = stage(accX, accY, bx, thrY, mY); __builtin_expect(mk != 0, 0)) save_hits(accY, mk, thrY, g, step);
            thrY = thrYn;
            bx = lbl[(g + 3) * 64];
            const uint32_t thrXn = lt[(g + 3) * 32];
            if (const uint64_t mk = stage(accY, accX, by, thrX, mX); __builtin_expect(mk != 0, 0)) save_hits(accX, mk, thrX, g + 1, step);
            thrX = thrXn;
        }
        if (const uint64_t mk = stage(accX, accY, bx, thrY, mY); __builtin_expect(mk != 0, 0)) save_hits(accY, mk, thrY, g, step);
        carried = true;
    };
    auto even_flush = [&](uint64_t prev) __attribute__((always_inline)) {
        if (carried) {
            uint32_t m;
            const uint32_t t = lt[(groups - 1) * 32];
            if (const uint64_t mk = last_fold(accX, t, m); mk != 0) { save_hits(accX, mk, t, groups - 1, prev); process_ring(prev); }
            carried = false;
        }
    };
    // FEW GROUPS.  With one or two groups a step has nothing of its own to hide behind: the first group's MFMAs and the last
    // group's fold stood alone (128 + 12 idle cycles and ~80 per 1 KB of rows and wave).  Here the accumulators live ACROSS
    // steps: a stage multiplies this step's rows while it folds what the previous stage left -- for one group the previous
    // STEP's products (the two accumulator sets trade roles from step to step: the unrolled DEPTH loop makes that static), for
    // two groups (s, g0) beside the fold of (s - 1, g1) and (s, g1) beside the fold of (s, g0).  The fragments stay in
    // registers.  A stage's candidates are processed at once (the ring then never mixes steps); `have`: something to fold.
    constexpr int FB = G == 1 || G == 2 ? G : 1;          // one or two groups keep their fragments in registers
    v4i fb[FB];
    bool have = false;
    if constexpr (G == 1 || G == 2) {
#pragma unroll
        for (int g = 0; g < G; ++g) fb[g] = lbl[g * 64];
    }
    // P receives the even groups, Q the odd ones; on entry Q holds the previous step's last group.  An even G leaves its last
    // group in Q again; an odd G leaves it in P: the caller swaps the sets from step to step.
    auto few_step = [&](Acc& P, Acc& Q, uint64_t prev) __attribute__((always_inline)) {      // prev: the step before `step`
        uint32_t m;
        v4i bcur;
        if constexpr (G == 1 || G == 2) bcur = fb[0];
        else bcur = lbl[0];
#pragma unroll
        for (int g = 0; g < (G > 0 ? G : 1); ++g) {
            v4i bnext = bcur;
            if (g + 1 < G) {
                if constexpr (G == 2) bnext = fb[FB - 1];
                else bnext = lbl[(g + 1) * 64];                             // requested while group g multiplies
            }
            Acc& nw = (g & 1) ? Q : P;
            Acc& od = (g & 1) ? P : Q;
            const uint32_t t = lt[(g == 0 ? G - 1 : g - 1) * 32];            // the threshold of the group being FOLDED
            const uint64_t mk = stage(nw, od, bcur, t, m);
            if (g == 0) {
                if (__builtin_expect(have && mk != 0, 0)) { save_hits(od, mk, t, G - 1, prev); process_ring(prev); }
            } else {
                if (__builtin_expect(mk != 0, 0)) { save_hits(od, mk, t, g - 1, step); process_ring(step); }
            }
            bcur = bnext;
        }
        have = true;
    };
    auto few_flush = [&](Acc& od, uint64_t prev) __attribute__((always_inline)) {        // the fold of the last products: those of step `prev`
        if (have) {
            uint32_t m;
            const uint32_t t = lt[(G > 0 ? G - 1 : 0) * 32];
            if (const uint64_t mk = last_fold(od, t, m); mk != 0) { save_hits(od, mk, t, G > 0 ? G - 1 : 0, prev); process_ring(prev); }
            have = false;
        }
    };
    // one step over the rows expanded in a[]: `body` multiplies and folds
    auto one_step = [&](auto&& body) __attribute__((always_inline)) {
        uint32_t fresh[4] = {0u, 0u, 0u, 0u};
        const bool look = !(ISK_EXP_PACK & 4) && (trip < 4 || (trip & refresh_mask) == 0);
        const bool refresh_now = refresh && look;
        uint32_t chk_count = 0, chk_what = ~0u;       // chk_what: query in chunk | hamming level << 16, ~0: no task
        if constexpr (MODE == MODE_SELF) {
            if (refresh_now) {
                const float* const src = p.thr_live + q0 + fresh_index();
#pragma unroll
                for (int i = 0; i < 4; ++i) fresh[i] = live_packed(src + i);
            }
            const uint32_t slice = nwaves >= slices ? gw : (gw + trip * nwaves) % slices;
            if (look && slice < slices) {
                uint32_t z = 0;
                asm volatile("" : "+v"(z));
                const uint32_t task = slice * 64 + __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
                const uint32_t ql = task >> 2, tpk = lthr[ql];
                const int level = unpack_threshold(tpk) + lpop[ql] - 1 - (int)(task & 3);
                if (tpk != 0 && level >= 0 && q0 + ql < p.nq_pad) {
                    chk_what = ql | ((uint32_t)level << 16);
                    chk_count = (uint32_t)__hip_atomic_load(reinterpret_cast<const int*>(p.ghist + (uint64_t)(q0 + ql) * HB + level), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
        body();
        // what the PREVIOUS step's appends returned (issued a whole step ago: no wait), then this step's blocks
        if constexpr (MODE == MODE_SELF) pend_complete();
        if (rcount) process_ring(step);
        if constexpr (MODE == MODE_SELF) {
            if (chk_what != ~0u && chk_count >= p.k) {
                const uint32_t ql = chk_what & 0xFFFFu;
                atomicMin(reinterpret_cast<uint32_t*>(p.thr_live) + q0 + ql, pack_threshold((int)(chk_what >> 16) - lpop[ql]));
            }
            if (refresh_now) *reinterpret_cast<uint4*>(lthr + fresh_index()) = make_uint4(fresh[0], fresh[1], fresh[2], fresh[3]);
        }
    };
    if constexpr (DEPTH == 1) {
        u32x4 x = load_rows(step);
        if ((groups & 1) || !ISK_PACK_CARRY) {
            while (step < nsteps) {
                expand(x);
                x = load_rows(step + stride < nsteps ? step + stride : step);    // in flight during this step
                one_step(all_groups);
                step += stride;
                ++trip;
            }
        } else {
            if (ISK_EXP_PACK & 2) expand(x);
            while (step < nsteps) {
                if (ISK_EXP_PACK & 2) asm volatile("" ::"v"(x));
                else expand(x);
                x = load_rows(step + stride < nsteps ? step + stride : step);
                one_step([&]() __attribute__((always_inline)) { even_groups(step - stride); });
                step += stride;
                ++trip;
            }
            even_flush(step - stride);
        }
    } else {
        const uint64_t nfull = p.n_rows / (32 * MT);          // whole steps: [first, nfull); a partial last step is loaded the slow way
        const uint64_t leap = stride * STRETCH;               // from a wave's stretch of steps to its next one
        constexpr int HOP = STRETCH == 1 ? 0 : 1;             // (experiment switch ISK_PACK_STRETCH = 0: substep d is step base + d * stride)
        u32x4 x[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) x[d] = u32x4{0u, 0u, 0u, 0u};
        if (step < nfull) {
            uint64_t base = step, last = step;
            uint32_t last_d = 0;
            bool more = true;
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) issue_rows(x[d], base + (HOP ? d : d * stride) < nfull ? base + (HOP ? d : d * stride) : base);
            while (more) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    step = base + (HOP ? d : d * stride);
                    if (step >= nfull) { more = false; break; }
                    const uint64_t prev = HOP ? (d ? step - 1 : step - leap + (DEPTH - 1)) : step - stride;
                    const uint64_t ahead = HOP ? leap : stride * DEPTH;
                    await_rows(x[d]);
                    expand(x[d]);
                    issue_rows(x[d], step + ahead < nfull ? step + ahead : base);   // (past the end: a re-read nobody uses)
                    if constexpr (G == 0) one_step(all_groups);
                    else if constexpr ((G & 1) != 0) {
                        static_assert((DEPTH & 1) == 0, "the accumulator roles of an odd group count alternate with d");
                        if (d & 1) one_step([&]() __attribute__((always_inline)) { few_step(accY, accX, prev); });
                        else one_step([&]() __attribute__((always_inline)) { few_step(accX, accY, prev); });
                        last_d = d;
                    } else one_step([&]() __attribute__((always_inline)) { few_step(accX, accY, prev); });
                    last = step;
                    ++trip;
                }
                if (more) { base += HOP ? leap : stride * DEPTH; step = base; }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if constexpr ((G & 1) != 0) {                     // the last group waits in P of the last call
                if (last_d & 1) few_flush(accY, last);
                else few_flush(accX, last);
            } else if constexpr (G > 1) few_flush(accY, last);   // ... in Q
        }
        if (step < nsteps) {                                  // == nfull: this wave owns the partial step
            expand(load_rows(step));
            one_step(all_groups);
        }
    }
    if constexpr (MODE == MODE_SELF) pend_complete();
}

template <int DEPTH, int G>
static void launch_pack_depth(int mode, dim3 grid, size_t lds, hipStream_t st, const ScanParams& p, uint32_t groups) {
    if (mode == MODE_COLLECT) hipLaunchKernelGGL((mfma_pack_kernel<MODE_COLLECT, DEPTH, G>), grid, dim3(MBLOCK), lds, st, p, groups);
    else if (mode == MODE_STRETCH) hipLaunchKernelGGL((mfma_pack_kernel<MODE_STRETCH, DEPTH, G>), grid, dim3(MBLOCK), lds, st, p, groups);
    else if (mode == MODE_SELF) hipLaunchKernelGGL((mfma_pack_kernel<MODE_SELF, DEPTH, G>), grid, dim3(MBLOCK), lds, st, p, groups);
    else hipLaunchKernelGGL((mfma_pack_kernel<MODE_BOTH, DEPTH, G>), grid, dim3(MBLOCK), lds, st, p, groups);
}

static int launch_pack(int mode, dim3 grid, size_t lds, hipStream_t st, const ScanParams& p, uint32_t groups) {
    if (lds > (size_t)MFMA_MAX_LDS) return (int)hipErrorInvalidValue;
    static_assert(PK_DEEP_GROUPS == 4, "one case per count");
    switch (groups) {           // up to PK_DEEP_GROUPS groups: one instantiation per count (a step's stages are straight-line code)
        case 1: launch_pack_depth<4, 1>(mode, grid, lds, st, p, groups); break;
        case 2: launch_pack_depth<4, 2>(mode, grid, lds, st, p, groups); break;
        case 3: launch_pack_depth<4, 3>(mode, grid, lds, st, p, groups); break;
        case 4: launch_pack_depth<4, 4>(mode, grid, lds, st, p, groups); break;
        default: launch_pack_depth<1, 0>(mode, grid, lds, st, p, groups);
    }
    return 0;
}

template <int W>
static int launch_w(int mode, dim3 grid, size_t lds, hipStream_t st, const ScanParams& p, uint32_t groups) {
    // a chunk's LDS image is <= 40 KB: inside the default dynamic-LDS limit, no per-device function attribute to set
    if (lds > (size_t)MFMA_MAX_LDS) return (int)hipErrorInvalidValue;
    if (mode == MODE_COLLECT) hipLaunchKernelGGL((mfma_scan_kernel<W, MODE_COLLECT>), grid, dim3(MBLOCK), lds, st, p, groups);
    else if (mode == MODE_STRETCH) hipLaunchKernelGGL((mfma_scan_kernel<W, MODE_STRETCH>), grid, dim3(MBLOCK), lds, st, p, groups);
    else if (mode == MODE_SELF) hipLaunchKernelGGL((mfma_scan_kernel<W, MODE_SELF>), grid, dim3(MBLOCK), lds, st, p, groups);
    else hipLaunchKernelGGL((mfma_scan_kernel<W, MODE_BOTH>), grid, dim3(MBLOCK), lds, st, p, groups);
    return 0;
}

uint32_t mfma_groups_per_chunk(int W, uint32_t nq_pad, bool pack) {
    // LDS per group: 32 queries x (32 * W bytes of +1/-1 nibbles + thr + popc); at most 40 KB per block, four blocks per CU
    static const uint32_t max_groups[5] = {0, 32, 16, 10, 8};
    uint32_t need = (nq_pad + 31) / 32;
    if (pack) return need < 1 ? 1 : (need < max_groups[1] ? need : max_groups[1]);      // mfma_pack_kernel takes any number of groups
    need += need & 1;                         // the pipeline of mfma_scan_kernel walks the groups in pairs
    if (need < 2) need = 2;
    if (need <= max_groups[W]) return need;
    // several chunks: as many as the LDS limit asks for, each as SMALL as that number of chunks allows -- 1 024 queries of 192-bit
    // codes are 32 groups: four chunks of 10 multiplied 40 groups' worth (a fifth of the MFMAs on padding queries: the matrix pipe
    // 0.75 busy for 0.53 of the peak, profiles/r04_pmc_sq_widths.txt), four chunks of 8 multiply 32
    const uint32_t chunks = (need + max_groups[W] - 1) / max_groups[W];
    uint32_t g = (need + chunks - 1) / chunks;
    g += g & 1;
    return g < max_groups[W] ? g : max_groups[W];
}

// B fragments | thresholds | popcounts | the four waves' rings of saved result blocks
size_t mfma_lds_bytes(int W, uint32_t groups) { return (size_t)groups * W * 64 * 16 + (size_t)groups * 32 * 8 + MBLOCK / 64 * PK_RING_ENTRIES * PK_RING_ENTRY_DWORDS * sizeof(uint32_t); }

uint32_t mfma_waves_per_block() { return MBLOCK / 64; }
uint32_t mfma_rows_per_wave_step(int W, bool pack) { return pack ? 32u * PK_TILES : 32u * (uint32_t)(W == 4 ? mfma_tiles<4>() : 2); }

uint32_t mfma_blocks_per_cu(int W, uint32_t groups, bool pack) {
    const uint32_t by_lds = (uint32_t)((160u * 1024u) / mfma_lds_bytes(W, groups));
    const uint32_t by_regs = pack || W <= 3 ? 3u : (mfma_tiles<4>() == 1 ? 4u : 2u);
    return by_lds < by_regs ? (by_lds ? by_lds : 1u) : by_regs;
}

int launch_mfma_scan(int W, int mode, bool pack, uint32_t blocks_x, uint32_t groups, hipStream_t st, const ScanParams& p) {
    const uint32_t chunks = (p.nq_pad + groups * 32 - 1) / (groups * 32);
    const dim3 grid(blocks_x, chunks);
    const size_t lds = mfma_lds_bytes(W, groups);
    if (pack) return W == 1 ? launch_pack(mode, grid, lds, st, p, groups) : (int)hipErrorInvalidValue;
    switch (W) {
        case 1: return launch_w<1>(mode, grid, lds, st, p, groups);
        case 2: return launch_w<2>(mode, grid, lds, st, p, groups);
        case 3: return launch_w<3>(mode, grid, lds, st, p, groups);
        default: return launch_w<4>(mode, grid, lds, st, p, groups);
    }
}

}  // namespace isk
