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
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(3))) v4i* lds_frag_ptr;
typedef const __attribute__((address_space(3))) float* lds_thr_ptr;

// row tiles (32 rows) per wave and step: two share every B fragment, threshold read and compare; experiment switch per W
#ifndef ISK_MFMA_TILES_W4
#define ISK_MFMA_TILES_W4 2
#endif
template <int W> constexpr int mfma_tiles() { return W == 4 ? ISK_MFMA_TILES_W4 : 2; }
#ifndef ISK_SCALAR_STEPS_FROM_W
#define ISK_SCALAR_STEPS_FROM_W 2     // experiment switch: code widths (in 64-bit words) from which the step number is kept scalar
#endif
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

// MODE_SELF: a candidate whose two atomics (list slot, distance counter) are IN FLIGHT.  Waiting for them on the spot costs
// the wave ~1 us per candidate (~180 candidates per query over 100 M rows: 0.1 ms of a 3 ms pass); instead the results are
// consumed at the lane's next candidate or at the end of the step, whichever comes first, when they have long arrived.
struct Pending {
    uint32_t slot, before;    // results of the atomics (valid once `meta` says so)
    uint32_t lo, hi;          // the candidate word (hamming << 48) | row
    uint32_t meta = 0;        // bit 31: pending; bit 30: `before` counts (hamming < tau_seen); bits 16..24: tau_seen; bits 0..9: query in chunk

    __device__ __forceinline__ void issue(const ScanParams& p, uint32_t q0, uint32_t ql, int h, uint64_t row, int tau_seen) {
        const uint32_t qi = q0 + ql;
        const bool counted = h < tau_seen;
        slot = atomicAdd(&p.cnt[(uint64_t)qi * CNT_STRIDE], 1u);
        if (counted) before = atomicAdd(&p.ghist[(uint64_t)qi * HB + (uint32_t)h], 1u);
        lo = (uint32_t)row;
        hi = ((uint32_t)h << 16) | (uint32_t)(row >> 32);
        meta = 0x80000000u | (counted ? 0x40000000u : 0u) | ((uint32_t)tau_seen << 16) | ql;
    }
    __device__ __forceinline__ void complete(const ScanParams& p, uint32_t q0, const int* lpop) {
        if (!(meta & 0x80000000u)) return;
        const uint32_t ql = meta & 0x3FFu, qi = q0 + ql;
        if (slot < p.cap) p.cand[(uint64_t)qi * p.cap + slot] = ((uint64_t)hi << 32) | lo;
        if (meta & 0x40000000u) {
            const int tau_seen = (int)((meta >> 16) & 0x1FFu);
            uint32_t* const counts = p.ghist + (uint64_t)qi * HB;
            uint32_t b = before;
            for (int t = (int)(hi >> 16);;) {       // as emit_self: count under every t < tau_seen until one proves k rows
                if (b + 1 >= p.k) { lower_threshold(p.thr_live + qi, (float)(t - lpop[ql])); break; }
                if (++t >= tau_seen) break;
                b = atomicAdd(&counts[(uint32_t)t], 1u);
            }
        }
        meta = 0;
    }
};

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

    constexpr bool ASYNC = W <= 2;       // five more live registers: W = 3 would spill, W = 4 is at two waves per SIMD already
    Pending pend;
    // a lane's 16 results per tile all belong to query g * 32 + (lane & 31): fold both tiles (two chains), compare once
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
        if (__builtin_expect(mall <= thr, 0)) {
            // rare: result `reg` of tile t is row (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) of that tile.
            // The step number is laundered so that hipcc does not hoist 32 row numbers per lane out of the group loop
            // (that cost 64 VGPRs in the hot loop for a path taken once in ~10^3 group-steps).
            uint32_t st_lo = (uint32_t)st, st_hi = (uint32_t)(st >> 32);
            asm volatile("" : "+v"(st_lo), "+v"(st_hi));
            const uint64_t base = (((uint64_t)st_hi << 32) | st_lo) * (32 * MT) + 4 * h;
            const uint32_t ql = g * 32 + r;
            const int pc = lpop[ql];
            int tau_seen = (int)thr + pc;               // MODE_SELF: the threshold this compare ran under (>= the live one)
            // MODE_SELF: the FIRST hit of the entry (nearly always the only one) is appended asynchronously -- see Pending
            uint32_t hits = 0, first_off = 0;
            float first_dot = 0.f;
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    if (acc.t[t][reg] <= thr) {
                        const uint32_t off = (uint32_t)(t * 32 + (reg & 3) + 8 * (reg >> 2));
                        const uint64_t row = base + off;
                        if (row <= last_row) {
                            if constexpr (MODE == MODE_SELF && !ASYNC) {
                                tau_seen = emit_self(p, q0 + ql, (int)acc.t[t][reg] + pc, row, tau_seen, pc);
                            } else if constexpr (MODE == MODE_SELF) {
                                if (hits == 0) { first_dot = acc.t[t][reg]; first_off = off; }
                                else tau_seen = emit_self(p, q0 + ql, (int)acc.t[t][reg] + pc, row, tau_seen, pc);
                                hits += 1;
                            } else {
                                emit<MODE>(p, q0 + ql, (uint32_t)((int)acc.t[t][reg] + pc), row);
                            }
                        }
                    }
                }
            if constexpr (MODE == MODE_SELF && ASYNC) {
                if (hits) {
                    pend.complete(p, q0, lpop);
                    pend.issue(p, q0, ql, (int)first_dot + pc, base + first_off, tau_seen);
                }
            }
        }
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
    const uint32_t fresh_at = wave * 256 + lane * 4;
    const bool refresh = MODE == MODE_SELF && fresh_at < groups * 32 && q0 + fresh_at < p.nq_pad;   // nq_pad is a multiple of 8
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
        if constexpr (MODE == MODE_SELF) {
            if (refresh_now) {
#pragma unroll
                for (int i = 0; i < 4; ++i) fresh[i] = live_threshold(p.thr_live + q0 + fresh_at + i);
            }
        }
        load_rows(ns, xn);                                                          // next step's rows, in flight during this one
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w)
                a[t][w] = v8i{(int)(nibbles(x[t][w], 0) << 1), (int)(nibbles(x[t][w], 1) << 1), (int)(nibbles(x[t][w], 2) << 1),
                              (int)(nibbles(x[t][w], 3) << 1), 0, 0, 0, 0};                      // bit ? 1.0 (0x2) : 0

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
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w) x[t][w] = xn[t][w];
        if constexpr (MODE == MODE_SELF) {
            if (refresh_now) *reinterpret_cast<float4*>(lthr + fresh_at) = make_float4(fresh[0], fresh[1], fresh[2], fresh[3]);
            if constexpr (ASYNC) {
                if (look) pend.complete(p, q0, lpop);
            }
        }
    }
    if constexpr (MODE == MODE_SELF && ASYNC) pend.complete(p, q0, lpop);
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

uint32_t mfma_groups_per_chunk(int W, uint32_t nq_pad) {
    // LDS per group: 32 queries x (32 * W bytes of +1/-1 nibbles + thr + popc); at most 40 KB per block, four blocks per CU
    static const uint32_t max_groups[5] = {0, 32, 16, 10, 8};
    uint32_t need = (nq_pad + 31) / 32;
    need += need & 1;                         // the pipeline walks the groups in pairs
    if (need < 2) need = 2;
    return need < max_groups[W] ? need : max_groups[W];
}

size_t mfma_lds_bytes(int W, uint32_t groups) { return (size_t)groups * W * 64 * 16 + (size_t)groups * 32 * 8; }

uint32_t mfma_waves_per_block() { return MBLOCK / 64; }
uint32_t mfma_rows_per_wave_step(int W) { return 32u * (uint32_t)(W == 4 ? mfma_tiles<4>() : 2); }

uint32_t mfma_blocks_per_cu(int W, uint32_t groups) {
    const uint32_t by_lds = (uint32_t)((160u * 1024u) / mfma_lds_bytes(W, groups));
    const uint32_t by_regs = W <= 3 ? 3u : (mfma_tiles<4>() == 1 ? 4u : 2u);
    return by_lds < by_regs ? (by_lds ? by_lds : 1u) : by_regs;
}

int launch_mfma_scan(int W, int mode, uint32_t blocks_x, uint32_t groups, hipStream_t st, const ScanParams& p) {
    const uint32_t chunks = (p.nq_pad + groups * 32 - 1) / (groups * 32);
    const dim3 grid(blocks_x, chunks);
    const size_t lds = mfma_lds_bytes(W, groups);
    switch (W) {
        case 1: return launch_w<1>(mode, grid, lds, st, p, groups);
        case 2: return launch_w<2>(mode, grid, lds, st, p, groups);
        case 3: return launch_w<3>(mode, grid, lds, st, p, groups);
        default: return launch_w<4>(mode, grid, lds, st, p, groups);
    }
}

}  // namespace isk
