// kernels.hip.h -- gfx950 (MI355X, CDNA4) device code of the brute-force Hamming / NPHD k-NN path.
//
// What is computed (reference: docs/explanation/similarity-search.md:24-29, call sites
// iscc_search/indexes/usearch/index.py:2037 and iscc_search/indexes/simprint/usearch_core.py:165):
// for every query the k stored codes with the smallest (hamming over the common prefix, key).
//
// Data layout: a table is split into SEGMENTS by code length in bytes; inside a
// segment all codes have the same length, so a (query class, segment) pass is a FIXED-length
// Hamming scan over W = ceil(p/8) 64-bit words, p = min(query bytes, segment bytes).  Codes are
// stored structure-of-arrays by word: col[w][row] (uint64, big-endian packed), keys[row*KW].
//
// Kernels
//   boot_kernel    threshold bootstrap: exact histogram of the first S0 rows -> per-query bias
//   scan_kernel    THE hot kernel: streams col[0..W) once per group of TQ queries; queries and
//                  thresholds live in SGPRs; per (row, query) 2 v_xor + 2 v_bcnt per word and half a
//                  v_min3; a lane leaves the streaming loop only when one of its rows beats a
//                  threshold (MODE_HIST: count it, MODE_COLLECT: append (hamming,row) to the
//                  query's candidate list)
//   pick_kernel    threshold from the sample histogram
//   select_kernel  per query: exact radix select over (hamming, key) of the candidates, bitonic
//                  sort of the k winners in LDS, emit records
//   fullhist_kernel exact histogram of one query over a whole segment (overflow fallback)
//   merge_kernel   k-way merge of sorted record lists (segments of a table, shards of a node)
//   radius_init_kernel  range-limited searches: the given threshold for every query (no bootstrap, no samples)
//   distinct_kernel     document frequency: distinct assets in a key-ordered collision list
//   plus small utilities (synthetic fill, row moves, row / frequency gathers); the sort-based frequency
//   column lives in docfreq.hip
//   Wide query groups (TQ*W >= 24) hold their queries in LDS instead of SGPRs (queries_in_lds).
//
// No MFMA: this is integer bit work bound by HBM reads (roofline in DESIGN.md section 4).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "scan_params.hip.h"

namespace isk {

struct Record {            // == isccsearch_record (24 bytes)
    uint64_t key_hi;
    uint64_t key_lo;
    uint32_t dist_rank;
    uint16_t hamming;
    uint16_t prefix_bits;
};
static_assert(sizeof(Record) == 24, "record layout");

// rows per thread per tile: every thread issues U 16-byte loads per column (2 rows each)
template <int W> struct TileCfg { static constexpr int U = (W == 1) ? 4 : (W == 2 ? 2 : 1); };
template <int W> constexpr int tile_rows() { return BLOCK * 2 * TileCfg<W>::U; }
// Where the TQ queries of a group live while a block scans.
//   SGPRs (scalar operands of v_xor, nothing to load in the loop) as long as they FIT: TQ*W*2 query dwords + TQ
//   biases + the loop's own scalars must stay under ~100 registers, beyond that hipcc spills them into VGPR lanes
//   and every use costs a v_readlane -- a VALU instruction, the very resource the kernel is short of (W=4, TQ=8:
//   52 spilled SGPRs = +19 % VALU work per tile; TQ=16: 211).  (VGPR-resident queries were measured and bring nothing:
//   see vgpr() above.)
//   LDS otherwise: one broadcast ds_read_b128 per four query dwords per tile, on the LDS pipe, into VGPR operands.
template <int W, int TQ> constexpr bool queries_in_lds() { return TQ * W >= 24; }
template <int W> constexpr int query_vecs() { return (2 * W + 3) / 4; }   // u32x4 slots per query in LDS
// tiles in flight ahead of the one being scored (experiment switch; 1 = double buffering)
#ifndef ISK_PREFETCH_DEPTH_W1
#define ISK_PREFETCH_DEPTH_W1 1
#endif
template <int W> constexpr int prefetch_depth() { return W == 1 ? ISK_PREFETCH_DEPTH_W1 : 1; }

__device__ __forceinline__ uint32_t bcnt(uint32_t x, uint32_t acc) {
    return (uint32_t)__builtin_popcount(x) + acc;   // cold paths: let the compiler pick the form
}
// Hot-path forms.  Left alone hipcc reassociates popc(x)+popc(y)+bias into 2 x v_bcnt(.., 0) +
// v_add3 and splits the row-pair minimum into v_min + v_min3 (5.75 VALU ops per (row, query) pair
// instead of 4.5).  An EMPTY asm statement on the running value stops the reassociation while
// instruction selection still folds popc(x)+acc into one v_bcnt_u32_b32 and the two mins into one
// v_min3_u32 (a non-empty asm makes the hazard recogniser pad with s_nop).
__device__ __forceinline__ uint32_t pin(uint32_t v) { asm("" : "+v"(v)); return v; }
__device__ __forceinline__ uint32_t bcnt_s(uint32_t x, uint32_t acc_sgpr) { return pin((uint32_t)__builtin_popcount(x) + acc_sgpr); }
__device__ __forceinline__ uint32_t bcnt_v(uint32_t x, uint32_t acc) { return (uint32_t)__builtin_popcount(x) + acc; }
__device__ __forceinline__ uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) { return pin(min(min(a, b), c)); }
__device__ __forceinline__ uint32_t sgpr(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// A wave-uniform value kept in a VGPR on purpose (experiment switch ISK_QUERIES_IN_VGPRS).  In ISOLATION gfx950 issues the
// plain two-operand ops (v_xor, v_and, v_add, shifts, v_mov, v_fma_f32) of a wave64 in ~2.4 cycles when their sources are
// VGPRs, inline constants or literals, and in ~4.1 when one source is an SGPR; v_bcnt, v_min3 and the other VOP3 integer ops
// take ~4.1-4.4 either way (profiles/r02_micro_valu2.txt).  MIXED with those 4-cycle ops, as in this kernel's inner loop,
// the fast forms gain nothing: 4 v_xor + 4 v_bcnt + 1 v_min3 take 35 cycles with the query words in VGPRs and 35 with them in
// SGPRs (profiles/r02_micro_valu3.txt), and the kernel measured 78.8 k queries/s either way -- so the queries stay in SGPRs,
// which leaves the VGPRs to the tiles in flight (7 instead of 6 waves per SIMD) and streams 2 % faster (0.84 vs 0.82 of HBM).
__device__ __forceinline__ uint32_t vgpr(uint32_t v) { asm volatile("" : "+v"(v)); return v; }

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // one 16-byte global load: .x/.y = row r (lo, hi), .z/.w = row r+1

template <bool NT>
__device__ __forceinline__ u32x4 load16(const uint64_t* p) {
    const u32x4* q = reinterpret_cast<const u32x4*>(p);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

// Streaming loads of the scan kernel, issued from inline asm so that the PREFETCH stays in flight:
// hipcc's own s_waitcnt insertion drained it at the loop head (vmcnt(0) in front of the next
// prefetch's address arithmetic).  hipcc neither counts nor pads what is inside an asm statement
// (cdna_hip_programming.md section 5.7), so:
//   * every use of a destination goes through wait_tile() first (counted s_waitcnt vmcnt + "+v" ties);
//   * the string opens with `s_nop 4`: the scalar bases may come straight from v_readfirstlane /
//     v_readlane (SGPR spill reloads), and a VALU-written SGPR needs 5 wait states before a VMEM
//     instruction reads it -- without the pad the load used a stale base (wrong rows, or a fault);
//   * outputs are early-clobber: a destination must not share a register with a later load's operand;
//   * kernels using these loads must have NO scratch and NO VGPR spills (a compiler copy of a
//     destination between load and wait would read garbage): tools/kernel_resources.py checks it.
// All U*W loads of a tile are ONE statement.  saddr form: 64-bit scalar column base + one 32-bit per-lane
// byte offset + immediate u*1024 (each wave reads U KiB contiguous per column).
#define ISK_LD(dst, off, base, imm, nt) "global_load_dwordx4 " dst ", " off ", " base " offset:" imm nt "\n\t"
template <bool NT, int U, int W>
__device__ __forceinline__ void load_tile_asm(u32x4 (&v)[U][W], const void* const (&tb)[W], uint32_t voff) {
    if constexpr (W == 1) {
        static_assert(U == 4, "tile shape");
        if constexpr (NT)
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%4", "%5", "0", " nt") ISK_LD("%1", "%4", "%5", "1024", " nt")
                         ISK_LD("%2", "%4", "%5", "2048", " nt") ISK_LD("%3", "%4", "%5", "3072", " nt")
                         : "=&v"(v[0][0]), "=&v"(v[1][0]), "=&v"(v[2][0]), "=&v"(v[3][0]) : "v"(voff), "s"(tb[0]) : "memory");
        else
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%4", "%5", "0", "") ISK_LD("%1", "%4", "%5", "1024", "")
                         ISK_LD("%2", "%4", "%5", "2048", "") ISK_LD("%3", "%4", "%5", "3072", "")
                         : "=&v"(v[0][0]), "=&v"(v[1][0]), "=&v"(v[2][0]), "=&v"(v[3][0]) : "v"(voff), "s"(tb[0]) : "memory");
    } else if constexpr (W == 2) {
        static_assert(U == 2, "tile shape");
        if constexpr (NT)
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%4", "%5", "0", " nt") ISK_LD("%1", "%4", "%6", "0", " nt")
                         ISK_LD("%2", "%4", "%5", "1024", " nt") ISK_LD("%3", "%4", "%6", "1024", " nt")
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]) : "v"(voff), "s"(tb[0]), "s"(tb[1]) : "memory");
        else
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%4", "%5", "0", "") ISK_LD("%1", "%4", "%6", "0", "")
                         ISK_LD("%2", "%4", "%5", "1024", "") ISK_LD("%3", "%4", "%6", "1024", "")
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[1][0]), "=&v"(v[1][1]) : "v"(voff), "s"(tb[0]), "s"(tb[1]) : "memory");
    } else if constexpr (W == 3) {
        static_assert(U == 1, "tile shape");
        if constexpr (NT)
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%3", "%4", "0", " nt") ISK_LD("%1", "%3", "%5", "0", " nt") ISK_LD("%2", "%3", "%6", "0", " nt")
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]) : "v"(voff), "s"(tb[0]), "s"(tb[1]), "s"(tb[2]) : "memory");
        else
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%3", "%4", "0", "") ISK_LD("%1", "%3", "%5", "0", "") ISK_LD("%2", "%3", "%6", "0", "")
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]) : "v"(voff), "s"(tb[0]), "s"(tb[1]), "s"(tb[2]) : "memory");
    } else {
        static_assert(W == 4 && U == 1, "tile shape");
        if constexpr (NT)
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%4", "%5", "0", " nt") ISK_LD("%1", "%4", "%6", "0", " nt")
                         ISK_LD("%2", "%4", "%7", "0", " nt") ISK_LD("%3", "%4", "%8", "0", " nt")
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3])
                         : "v"(voff), "s"(tb[0]), "s"(tb[1]), "s"(tb[2]), "s"(tb[3]) : "memory");
        else
            asm volatile("s_nop 4\n\t" ISK_LD("%0", "%4", "%5", "0", "") ISK_LD("%1", "%4", "%6", "0", "")
                         ISK_LD("%2", "%4", "%7", "0", "") ISK_LD("%3", "%4", "%8", "0", "")
                         : "=&v"(v[0][0]), "=&v"(v[0][1]), "=&v"(v[0][2]), "=&v"(v[0][3])
                         : "v"(voff), "s"(tb[0]), "s"(tb[1]), "s"(tb[2]), "s"(tb[3]) : "memory");
    }
}
// wait until at most N vector-memory operations of this wave are outstanding, then tie the tile's
// registers to the wait so that no use can be scheduled above it
template <int N, int U, int W>
__device__ __forceinline__ void wait_tile(u32x4 (&v)[U][W]) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int w = 0; w < W; ++w) asm volatile("" : "+v"(v[u][w]));
}

// ---------------------------------------------------------------------------------------------
// scan_kernel<W, MASK, TQ, MODE, NT>
//   grid = (blocks_x, query_groups); block = 256.  Group g holds queries [g*TQ, (g+1)*TQ).
//   Fast path per tile: U*W coalesced 16-byte loads per lane, then for every query
//       acc = bias_q; acc = bcnt(row_lo ^ q_lo, acc); acc = bcnt(row_hi ^ q_hi, acc)   (per word)
//   so acc < 2^31  <=>  hamming <= tau_q, and one v_min3 folds two rows into the lane's running
//   minimum.  Only lanes whose minimum has bit 31 clear enter the (rare) emit path.
// ---------------------------------------------------------------------------------------------
//   FOLD: the fast path tests popc((lo^q_lo)|(hi^q_hi)) <= tau, a NECESSARY condition, for 3.5 instead of 4.5 ops
//         per pair; the emit path computes the exact distance.  Only pays under a tight threshold: chosen at run
//         time by scan_adapt_kernel.
template <int W, bool MASK, int TQ, int MODE, bool NT, bool FOLD = false>
__device__ __forceinline__ void scan_body(const ScanParams& p) {
    static_assert(!FOLD || (W == 1 && !MASK), "the OR-fold filter is for whole 64-bit codes");
    constexpr int U = TileCfg<W>::U;
    constexpr int TILE = BLOCK * 2 * U;
    constexpr bool QL = queries_in_lds<W, TQ>() && !FOLD;
    constexpr int NV = query_vecs<W>();
    const uint32_t tid = threadIdx.x;
    const uint32_t q0 = blockIdx.y * TQ;

    // biases -> SGPRs; queries -> SGPRs (uniform addresses: scalar loads) or LDS (see queries_in_lds)
    __shared__ u32x4 lq[QL ? TQ * NV : 1];
    uint32_t qlo[QL ? 1 : TQ][W], qhi[QL ? 1 : TQ][W], bias[TQ];
#pragma unroll
    for (int q = 0; q < TQ; ++q) {
        bias[q] = sgpr(p.bias[q0 + q]);
        if constexpr (!QL) {
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint64_t v = p.queries[(uint64_t)(q0 + q) * 4 + w];
#ifdef ISK_QUERIES_IN_VGPRS
                qlo[q][w] = vgpr((uint32_t)v);
                qhi[q][w] = vgpr((uint32_t)(v >> 32));
#else
                qlo[q][w] = sgpr((uint32_t)v);
                qhi[q][w] = sgpr((uint32_t)(v >> 32));
#endif
            }
        }
    }
    if constexpr (QL) {
        // dword d of query q = half (d & 1) of word d / 2; slots past 2*W stay zero
        uint32_t* l = reinterpret_cast<uint32_t*>(lq);
        for (uint32_t i = tid; i < (uint32_t)(TQ * NV * 4); i += BLOCK) {
            const uint32_t q = i / (NV * 4), d = i % (NV * 4);
            uint32_t val = 0;
            if (d < 2 * W) {
                const uint64_t v = p.queries[(uint64_t)(q0 + q) * 4 + d / 2];
                val = (d & 1) ? (uint32_t)(v >> 32) : (uint32_t)v;
            }
            l[i] = val;
        }
        __syncthreads();
    }
    // the words of query q as operands: SGPR copies, or one broadcast LDS read per four dwords
    auto query_words = [&](int q, uint32_t (&ql)[W], uint32_t (&qh)[W]) {
        if constexpr (QL) {
            u32x4 t[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) t[j] = lq[q * NV + j];
#pragma unroll
            for (int w = 0; w < W; ++w) { ql[w] = t[(2 * w) / 4][(2 * w) % 4]; qh[w] = t[(2 * w + 1) / 4][(2 * w + 1) % 4]; }
        } else {
#pragma unroll
            for (int w = 0; w < W; ++w) { ql[w] = qlo[q][w]; qh[w] = qhi[q][w]; }
        }
    };
    const uint32_t mlo = sgpr(p.mask_lo), mhi = sgpr(p.mask_hi);

    // Tile numbers are 32-bit ON PURPOSE: the loop tests below are then scalar compares (s_cmp_lt_u32) and scalar branches.
    // With 64-bit counters hipcc did the unsigned compares on the VALU (there is no s_cmp_lt_u64), parked the operand in a
    // register pair it also uses for load destinations, and structurised the `break`s with EXEC tests -- a control-flow
    // graph tools/audit_kernels.py cannot prove the asm-load invariants on.  The host refuses segments of >= 2^31 tiles.
    const uint32_t n_full = (uint32_t)(p.n_rows / TILE);
    // per-lane byte offset inside a tile (constant over the loop): wave w reads U KiB contiguous per column,
    // load u of a lane sits u*1024 bytes further (immediate offset); the tile base stays scalar.
    //   row(u, lane, r) = tile*TILE + wave*(U*128) + u*128 + lane*2 + r
    const uint32_t wave = tid >> 6, lane = tid & 63;
    const uint32_t voff = wave * (uint32_t)(U * 1024) + lane * 16u;
    const uint32_t row_in_tile = wave * (uint32_t)(U * 128) + lane * 2u;

    auto load_tile = [&](u32x4 (&v)[U][W], uint32_t tile) {
        const void* tb[W];
#pragma unroll
        for (int w = 0; w < W; ++w) {
            // uniform tile base, forced into an SGPR pair for the saddr operand
            const uint64_t ta = reinterpret_cast<uint64_t>(p.col[w]) + (uint64_t)tile * (uint64_t)(TILE * 8);
            tb[w] = reinterpret_cast<const void*>(((uint64_t)sgpr((uint32_t)(ta >> 32)) << 32) | sgpr((uint32_t)ta));
        }
        load_tile_asm<NT, U, W>(v, tb, voff);
    };

    auto process = [&](const u32x4 (&v)[U][W], uint32_t tile) {
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < TQ; ++q) {
            uint32_t ql[W], qh[W];
            query_words(q, ql, qh);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint32_t a0, a1;
                if constexpr (FOLD) {
                    // OR-fold filter for 64-bit codes: y = (lo ^ q_lo) | (hi ^ q_hi) has popc(y) <= hamming, so
                    // popc(y) <= tau is NECESSARY for a candidate.  One v_xor + one v_bitop3 (a | (b ^ c)) + one
                    // v_bcnt per row: 3.5 VALU ops per pair instead of 4.5.  For unrelated codes y is 3/4 ones
                    // (popc ~ 24 +- 2.4), so at tau ~ 12-15 the filter passes ~1e-5 of the pairs; the exact
                    // distance is computed in the emit path below.
                    const uint32_t y0 = __builtin_amdgcn_bitop3_b32(v[u][0].x ^ ql[0], v[u][0].y, qh[0], 0xF6);
                    const uint32_t y1 = __builtin_amdgcn_bitop3_b32(v[u][0].z ^ ql[0], v[u][0].w, qh[0], 0xF6);
                    a0 = (uint32_t)__builtin_popcount(y0) + bias[q];   // one v_bcnt_u32_b32 with the SGPR bias as accumulator
                    a1 = (uint32_t)__builtin_popcount(y1) + bias[q];
                    m = min3u(m, a0, a1);
                    continue;
                }
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    uint32_t x0 = v[u][w].x ^ ql[w], y0 = v[u][w].y ^ qh[w];
                    uint32_t x1 = v[u][w].z ^ ql[w], y1 = v[u][w].w ^ qh[w];
                    if (MASK && w == W - 1) { x0 &= mlo; y0 &= mhi; x1 &= mlo; y1 &= mhi; }
                    if (w == 0) { a0 = bcnt_s(x0, bias[q]); a1 = bcnt_s(x1, bias[q]); }
                    else { a0 = pin(bcnt_v(x0, a0)); a1 = pin(bcnt_v(x1, a1)); }
                    a0 = bcnt_v(y0, a0);
                    a1 = bcnt_v(y1, a1);
                    // multi-word codes: every step of the chain is pinned, or hipcc re-associates the words after the
                    // first into v_bcnt(x, 0) + v_bcnt(y, 0) + v_add3 (W = 4: 48 extra VALU instructions per wave-tile)
                    if (W > 1 && w + 1 < W) { a0 = pin(a0); a1 = pin(a1); }
                }
                m = min3u(m, a0, a1);
            }
        }
        if ((int32_t)m >= 0) {
            // rare: at least one (row, query) pair of this lane is within its threshold.  Rescore per
            // query from the SGPR-resident queries (fully unrolled: no memory loads, no dynamic register
            // indexing), so a tile that takes this path costs about two plain tiles instead of the
            // ~16 a load-per-query loop cost.
            const uint64_t base = (uint64_t)tile * TILE + row_in_tile;
            // launder the row registers: without this the compiler merges the rescoring below with
            // the fast path above (common subexpressions) and keeps all TQ*U*2 accumulators alive
            u32x4 r[U][W];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int w = 0; w < W; ++w) { r[u][w] = v[u][w]; asm volatile("" : "+v"(r[u][w])); }
#pragma unroll
            for (int q = 0; q < TQ; ++q) {
                uint32_t a[U][2];
                uint32_t mq = 0xFFFFFFFFu;
                uint32_t ql[W], qh[W];
                query_words(q, ql, qh);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    uint32_t a0 = bias[q], a1 = bias[q];
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        uint32_t x0 = r[u][w].x ^ ql[w], y0 = r[u][w].y ^ qh[w];
                        uint32_t x1 = r[u][w].z ^ ql[w], y1 = r[u][w].w ^ qh[w];
                        if (MASK && w == W - 1) { x0 &= mlo; y0 &= mhi; x1 &= mlo; y1 &= mhi; }
                        a0 = bcnt(y0, bcnt(x0, a0));
                        a1 = bcnt(y1, bcnt(x1, a1));
                    }
                    a[u][0] = a0; a[u][1] = a1;
                    mq = min(mq, min(a0, a1));
                }
                if ((int32_t)mq >= 0) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const uint64_t row = base + (uint64_t)u * 128;
                        if ((int32_t)a[u][0] >= 0) emit<MODE>(p, q0 + q, a[u][0] - bias[q], row);
                        if ((int32_t)a[u][1] >= 0) emit<MODE>(p, q0 + q, a[u][1] - bias[q], row + 1);
                    }
                }
            }
        }
    };

    // software pipeline: the loads of the next tile are in flight while the current one is scored.
    // The prefetch is UNCONDITIONAL (past the end it re-reads the block's last tile) so that exactly
    // U*W younger loads are outstanding at every wait: s_waitcnt vmcnt(U*W) retires the current tile
    // and leaves the prefetch alone.  (The rare emit path may add compiler-counted stores/atomics in
    // between; more outstanding operations only make the counted wait stricter, never weaker.)
    if constexpr (prefetch_depth<W>() == 1) {
        u32x4 va[U][W], vb[U][W];
        uint32_t tile = (uint32_t)(p.row_begin / TILE) + blockIdx.x;
        if (tile < n_full) {
            const uint32_t last = n_full - 1;
            load_tile(va, tile);
            for (;;) {
                const uint32_t t1 = tile + gridDim.x;
                load_tile(vb, t1 < n_full ? t1 : last);
                wait_tile<U * W>(va);
                process(va, tile);
                if (t1 >= n_full) break;
                const uint32_t t2 = t1 + gridDim.x;
                load_tile(va, t2 < n_full ? t2 : last);
                wait_tile<U * W>(vb);
                process(vb, t1);
                if (t2 >= n_full) break;
                tile = t2;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last (unused) prefetch
        }
    } else {
        // two tiles ahead: three buffers rotate, 2*U*W younger loads are outstanding at every wait
        static_assert(prefetch_depth<W>() == 2, "prefetch depth");
        u32x4 va[U][W], vb[U][W], vc[U][W];
        uint32_t t = (uint32_t)(p.row_begin / TILE) + blockIdx.x;
        const uint32_t G = gridDim.x;
        if (t < n_full) {
            const uint32_t last = n_full - 1;
            auto clamp = [&](uint32_t x) { return x < n_full ? x : last; };
            load_tile(va, t);
            load_tile(vb, clamp(t + G));
            for (;;) {
                load_tile(vc, clamp(t + 2 * G));
                wait_tile<2 * U * W>(va);
                process(va, t);
                if (t + G >= n_full) break;
                load_tile(va, clamp(t + 3 * G));
                wait_tile<2 * U * W>(vb);
                process(vb, t + G);
                if (t + 2 * G >= n_full) break;
                load_tile(vb, clamp(t + 4 * G));
                wait_tile<2 * U * W>(vc);
                process(vc, t + 2 * G);
                if (t + 3 * G >= n_full) break;
                t += 3 * G;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the unused prefetches
        }
    }

    // tail rows [n_full*TILE, n_rows): one row per thread, a slice of 256 rows per block (the host launches at least as many blocks
    // as the tail has slices).  One block used to walk the whole tail, up to 8 rounds of dependent loads: a one-query scan of 10 000
    // rows took 15 us, of 1 M rows 9 -- most of it this loop.
    {
        for (uint64_t row = (uint64_t)n_full * TILE + (uint64_t)blockIdx.x * BLOCK + tid; row < p.n_rows; row += (uint64_t)gridDim.x * BLOCK) {
            uint32_t lo[W], hi[W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                const uint64_t c = p.col[w][row];
                lo[w] = (uint32_t)c; hi[w] = (uint32_t)(c >> 32);
            }
            // (queries and biases as the tiles take them -- SGPRs or LDS, loaded in the prologue: re-read from global memory per query,
            //  a slice cost TQ rounds of dependent scalar loads)
#pragma unroll
            for (int q = 0; q < TQ; ++q) {
                uint32_t ql[W], qh[W];
                query_words(q, ql, qh);
                uint32_t a = bias[q];
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    uint32_t x = lo[w] ^ ql[w], y = hi[w] ^ qh[w];
                    if (MASK && w == W - 1) { x &= mlo; y &= mhi; }
                    a = bcnt(y, bcnt(x, a));
                }
                if ((int32_t)a >= 0) emit<MODE>(p, q0 + q, a - bias[q], row);
            }
        }
    }
}

template <int W, bool MASK, int TQ, int MODE, bool NT>
__global__ __launch_bounds__(BLOCK) void scan_kernel(const ScanParams p) {
    scan_body<W, MASK, TQ, MODE, NT, false>(p);
}

// Whole 64-bit codes: both fast paths in one kernel, chosen per query group at run time.  The folded path saves one
// VALU operation per pair but raises a false alarm (a full rescoring of the tile) for ~3.8e-5 of the pairs at
// tau = 13, 8e-6 at 12, 1.5e-6 at 11 (y = (lo^q_lo)|(hi^q_hi) is Binomial(32, 3/4) for unrelated codes): it pays
// only once the group's thresholds are tight -- which the levels and the picks between stretches bring about as
// the pass advances, and which a collision lookup (max_hamming 0) has from the start.
template <int TQ, int MODE>
__global__ __launch_bounds__(BLOCK) void scan_adapt_kernel(const ScanParams p) {
    const uint32_t q0 = blockIdx.y * TQ;
    bool fold = p.fold_tau != 0;
#pragma unroll
    for (int q = 0; q < TQ; ++q) fold = fold && sgpr(p.bias[q0 + q]) >= 0x7FFFFFFFu - p.fold_tau;   // BIAS_NEVER passes too
    if (fold) scan_body<1, false, TQ, MODE, true, true>(p);
    else scan_body<1, false, TQ, MODE, true, false>(p);
}

// ---------------------------------------------------------------------------------------------
// find the first histogram bin where the running count reaches `need`  (wave 0 does the work)
//   returns the bin in res[0] and the count strictly below it in res[1]; every thread gets both.
//   nbins <= 320.  If the total is below `need` the last bin is returned.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_find_cut(const uint32_t* hist, uint32_t nbins, uint32_t need,
                                               uint32_t* res, uint32_t& bin, uint32_t& less) {
    const uint32_t tid = threadIdx.x;
    if (tid < 64) {
        uint32_t c[5], s = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t b = tid * 5 + j;
            c[j] = b < nbins ? hist[b] : 0u;
            s += c[j];
        }
        uint32_t incl = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t t = __shfl_up(incl, off, 64);
            if (tid >= (uint32_t)off) incl += t;
        }
        const uint32_t excl = incl - s;
        const uint32_t total = __shfl(incl, 63, 64);
        if (tid == 0 && total < need) {          // not enough entries: take everything
            uint32_t last = 0, run = 0, below = 0;
            for (uint32_t b = 0; b < nbins; ++b) { if (hist[b]) { last = b; below = run; } run += hist[b]; }
            res[0] = last; res[1] = below;
        }
        if (total >= need && excl < need && need <= incl) {
            uint32_t run = excl;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                if (run + c[j] >= need) { res[0] = tid * 5 + j; res[1] = run; break; }
                run += c[j];
            }
        }
    }
    __syncthreads();
    bin = res[0];
    less = res[1];
    __syncthreads();
}

// range-limited searches have a given threshold: one launch sets every bias and zeroes the candidate counters
__global__ __launch_bounds__(BLOCK) void radius_init_kernel(uint32_t* bias, uint32_t* cnt, uint32_t nq, uint32_t nq_pad, uint32_t value) {
    const uint32_t q = blockIdx.x * BLOCK + threadIdx.x;
    if (q >= nq_pad) return;
    bias[q] = q < nq ? value : BIAS_NEVER;
    cnt[(uint64_t)q * CNT_STRIDE] = 0;
}

// ... and for a handful of queries the queries themselves, carried in the kernel's arguments ([nq_pad][4] words)
constexpr uint32_t INLINE_QUERIES = 16;
struct InlineQueries { uint64_t w[INLINE_QUERIES * 4]; };
__global__ __launch_bounds__(BLOCK) void radius_init_inline_kernel(uint32_t* bias, uint32_t* cnt, uint32_t nq, uint32_t nq_pad, uint32_t value,
                                                                   uint64_t* queries, const InlineQueries iq) {
    const uint32_t i = threadIdx.x;
    if (i < nq_pad * 4) queries[i] = iq.w[i];
    if (i < nq_pad) {
        bias[i] = i < nq ? value : BIAS_NEVER;
        cnt[(uint64_t)i * CNT_STRIDE] = 0;
    }
}

struct BootParams {
    const uint64_t* col[4];
    const uint64_t* queries;  // [nq_pad][4]
    uint32_t* bias;           // [nq_pad] out
    uint32_t* cnt;            // [nq_pad * CNT_STRIDE] candidate counters: zeroed here (saves the host a memset)
    uint64_t s0;              // rows [0, s0) are sampled (s0 >= 1)
    uint32_t nq;              // real queries; blocks q >= nq write BIAS_NEVER
    uint32_t k;
    uint32_t W;
    uint64_t mask_last;
    float* thr;               // [nq_pad] out, nullable: tau0 - popc(query) as the MFMA scan compares it (MODE_SELF)
    uint32_t thr_packed;      // ... written PACKED (pack_threshold, scan_params.hip.h) for mfma_pack_kernel
    uint32_t* counts;         // [nq_pad][HB], nullable: zeroed here -- the distance counters of the self-tightening pass
    uint32_t hint;            // BOOT_NO_HINT, or the threshold itself (no sample): the k-th distance a previous batch of this size ended at + margin
};
constexpr uint32_t BOOT_NO_HINT = 0xFFFFFFFFu;
// the bootstrap threshold (dot-product form: tau0 - popc(query); never = no row can be a candidate) in the scan's representation
__device__ __forceinline__ void store_boot_threshold(const BootParams& p, uint32_t q, int thr, bool never) {
    if (p.thr_packed) reinterpret_cast<uint32_t*>(p.thr)[q] = never ? 0u : pack_threshold(thr);
    else p.thr[q] = never ? -1.0e9f : (float)thr;
}
constexpr uint64_t BOOT_EXACT_ROWS = 4096;   // rows of the full histogram; the rest of a longer sample only counts under its cut

// the longer part of the bootstrap sample: rows [s1, s0) that lie at or under `cut` go into the histogram.  W is a
// template argument so that the eight rows of a trip are eight INDEPENDENT loads (with a run-time word loop hipcc keeps
// them in program order and the loop waits out one L2 latency per row: 115 us per 65 536 rows instead of ~15)
template <int W>
__device__ __forceinline__ void boot_tail(const BootParams& p, const uint64_t (&qw)[4], uint64_t s1, uint32_t cut, uint32_t* hist) {
    const uint32_t tid = threadIdx.x, nthr = blockDim.x;
    for (uint64_t r0 = s1 + tid; r0 < p.s0; r0 += 8 * nthr) {
        uint64_t x[8][W];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const uint64_t r = r0 + (uint64_t)u * nthr;
            const uint64_t rr = r < p.s0 ? r : s1;          // clamped: the value is discarded below
#pragma unroll
            for (int w = 0; w < W; ++w) x[u][w] = p.col[w][rr];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t h = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                uint64_t y = x[u][w] ^ qw[w];
                if (w == W - 1) y &= p.mask_last;
                h += (uint32_t)__builtin_popcountll(y);
            }
            if (h <= cut && r0 + (uint64_t)u * nthr < p.s0) atomicAdd(&hist[h], 1u);
        }
    }
}

// one block per (padded) query: tau0 = k-th smallest hamming over the first s0 rows.  The first BOOT_EXACT_ROWS rows
// go into a full histogram (LDS atomics on a handful of hot bins: ~6 us); a longer sample then only counts the rows at
// or under THAT cut -- a few per thousand.  Any block size from 64 to 1 024 threads: with a handful of queries the host
// launches wide blocks, or the sample of a query would be one block's latency-bound walk.
__global__ __launch_bounds__(1024) void boot_kernel(const BootParams p) {
    __shared__ uint32_t hist[320];
    __shared__ uint32_t res[2];
    const uint32_t q = blockIdx.x, tid = threadIdx.x, nthr = blockDim.x;
    if (tid == 0) p.cnt[(uint64_t)q * CNT_STRIDE] = 0;
    if (p.counts)
        for (uint32_t i = tid; i < HB; i += nthr) p.counts[(uint64_t)q * HB + i] = 0;      // (saves the host a memset launch)
    if (q >= p.nq) {
        if (tid == 0) {
            p.bias[q] = BIAS_NEVER;
            if (p.thr) store_boot_threshold(p, q, 0, true);       // below every dot product: never a candidate
        }
        return;
    }
    uint64_t qw[4];
    for (uint32_t w = 0; w < 4; ++w) qw[w] = w < p.W ? p.queries[(uint64_t)q * 4 + w] : 0;
    if (p.hint != BOOT_NO_HINT) {      // (uniform: a launch parameter) the threshold is given -- the host verifies that it held k rows
        if (tid == 0) {
            p.bias[q] = 0x7FFFFFFFu - p.hint;
            if (p.thr) {
                uint32_t pc = 0;
                for (uint32_t w = 0; w < p.W; ++w) pc += (uint32_t)__builtin_popcountll(w == p.W - 1 ? qw[w] & p.mask_last : qw[w]);
                store_boot_threshold(p, q, (int)p.hint - (int)pc, false);
            }
        }
        return;
    }
    for (uint32_t i = tid; i < 320; i += nthr) hist[i] = 0;
    __syncthreads();
    // (a large k keeps the whole sample exact: the cut of the first rows must leave >= k rows under it)
    const uint64_t s1 = (p.s0 <= BOOT_EXACT_ROWS || (uint64_t)p.k * 4 > BOOT_EXACT_ROWS) ? p.s0 : BOOT_EXACT_ROWS;
    for (uint64_t r = tid; r < s1; r += nthr) {
        uint32_t h = 0;
        for (uint32_t w = 0; w < p.W; ++w) {
            uint64_t x = p.col[w][r] ^ qw[w];
            if (w == p.W - 1) x &= p.mask_last;
            h += (uint32_t)__builtin_popcountll(x);
        }
        atomicAdd(&hist[h], 1u);
    }
    __syncthreads();
    uint32_t bin, less;
    block_find_cut(hist, NBINS, p.k < s1 ? p.k : (uint32_t)s1, res, bin, less);
    if (p.s0 > s1) {
        // (uniform branch: s0 is a launch parameter)  bins <= `bin` become exact over [0, s0); the k-th smallest lies there
        switch (p.W) {
            case 1: boot_tail<1>(p, qw, s1, bin, hist); break;
            case 2: boot_tail<2>(p, qw, s1, bin, hist); break;
            case 3: boot_tail<3>(p, qw, s1, bin, hist); break;
            default: boot_tail<4>(p, qw, s1, bin, hist); break;
        }
        __syncthreads();
        block_find_cut(hist, NBINS, p.k < p.s0 ? p.k : (uint32_t)p.s0, res, bin, less);
    }
    if (tid == 0) {
        p.bias[q] = 0x7FFFFFFFu - bin;
        if (p.thr) {
            uint32_t pc = 0;
            for (uint32_t w = 0; w < p.W; ++w) pc += (uint32_t)__builtin_popcountll(w == p.W - 1 ? qw[w] & p.mask_last : qw[w]);
            store_boot_threshold(p, q, (int)bin - (int)pc, false);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// boot_multi_kernel<W>: the same bootstrap for LARGE batches, BOOT_QB queries per 1 024-thread block.  With one block per query
// 1 024 blocks each read the whole 512 KB sample from the L2 (~0.5 GB per launch, ~11 TB/s for 45 us): here a row is loaded
// once per BOOT_QB queries and the kernel is bound by its XOR + popcount work instead.  nq_pad is a multiple of 8, hence of 4.
// ---------------------------------------------------------------------------------------------
constexpr int BOOT_QB = 4;      // (8 measured slower: 128 blocks leave the chip short of waves; 1 M rows x 1 024 queries 0.213 against 0.191 ms)
template <int W>
__global__ __launch_bounds__(1024) void boot_multi_kernel(const BootParams p) {
    __shared__ uint32_t hist[BOOT_QB][320];
    __shared__ uint32_t res[2];
    const uint32_t q0 = blockIdx.x * BOOT_QB, tid = threadIdx.x, nthr = blockDim.x;
    if (tid < BOOT_QB) p.cnt[(uint64_t)(q0 + tid) * CNT_STRIDE] = 0;
    if (p.counts)
        for (uint32_t i = tid; i < BOOT_QB * HB; i += nthr) p.counts[(uint64_t)q0 * HB + i] = 0;
    for (uint32_t i = tid; i < BOOT_QB * 320; i += nthr) (&hist[0][0])[i] = 0;
    uint64_t qw[BOOT_QB][W];
#pragma unroll
    for (int i = 0; i < BOOT_QB; ++i)
#pragma unroll
        for (int w = 0; w < W; ++w) qw[i][w] = p.queries[(uint64_t)(q0 + i) * 4 + w];     // uniform addresses: scalar loads
    __syncthreads();
    auto hamming = [&](const uint64_t (&x)[W], int i) {
        uint32_t h = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) {
            uint64_t y = x[w] ^ qw[i][w];
            if (w == W - 1) y &= p.mask_last;
            h += (uint32_t)__builtin_popcountll(y);
        }
        return h;
    };
    // (a large k keeps the whole sample exact: the cut of the first rows must leave >= k rows under it)
    const uint64_t s1 = (p.s0 <= BOOT_EXACT_ROWS || (uint64_t)p.k * 4 > BOOT_EXACT_ROWS) ? p.s0 : BOOT_EXACT_ROWS;
    for (uint64_t r = tid; r < s1; r += nthr) {
        uint64_t x[W];
#pragma unroll
        for (int w = 0; w < W; ++w) x[w] = p.col[w][r];
#pragma unroll
        for (int i = 0; i < BOOT_QB; ++i) atomicAdd(&hist[i][hamming(x, i)], 1u);
    }
    __syncthreads();
    uint32_t cut[BOOT_QB], less;
#pragma unroll
    for (int i = 0; i < BOOT_QB; ++i) block_find_cut(hist[i], NBINS, p.k < s1 ? p.k : (uint32_t)s1, res, cut[i], less);
    if (p.s0 > s1) {
        // bins <= cut[i] become exact over [0, s0); the k-th smallest of query i lies there.  Eight independent rows per
        // thread and trip, each scored against the block's queries.
        for (uint64_t r0 = s1 + tid; r0 < p.s0; r0 += 8 * (uint64_t)nthr) {
            uint64_t x[8][W];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint64_t r = r0 + (uint64_t)u * nthr;
                const uint64_t rr = r < p.s0 ? r : s1;          // clamped: the value is discarded below
#pragma unroll
                for (int w = 0; w < W; ++w) x[u][w] = p.col[w][rr];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool real = r0 + (uint64_t)u * nthr < p.s0;
#pragma unroll
                for (int i = 0; i < BOOT_QB; ++i) {
                    const uint32_t h = hamming(x[u], i);
                    if (h <= cut[i] && real) atomicAdd(&hist[i][h], 1u);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < BOOT_QB; ++i) block_find_cut(hist[i], NBINS, p.k < p.s0 ? p.k : (uint32_t)p.s0, res, cut[i], less);
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < BOOT_QB; ++i) {
            const uint32_t q = q0 + i;
            if (q >= p.nq) {
                p.bias[q] = BIAS_NEVER;
                if (p.thr) store_boot_threshold(p, q, 0, true);       // below every dot product: never a candidate
                continue;
            }
            p.bias[q] = 0x7FFFFFFFu - cut[i];
            if (p.thr) {
                uint32_t pc = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) pc += (uint32_t)__builtin_popcountll(w == W - 1 ? qw[i][w] & p.mask_last : qw[i][w]);
                store_boot_threshold(p, q, (int)cut[i] - (int)pc, false);
            }
        }
    }
}

struct PickParams {
    const uint32_t* ghist;   // [nq_pad][HB]
    uint32_t* bias;          // [nq_pad] in/out
    uint32_t nq;
    uint32_t need;           // min(k, rows seen so far)
    uint32_t* cnt;           // [nq_pad * CNT_STRIDE] candidate counters   (nullptr: no pruning)
    uint64_t* cand;          // [nq_pad][cap] candidate lists, pruned in place to the new threshold
    uint32_t cap;
};

// one block per query: tau = first bin of the running histogram where the count reaches `need`; then the
// candidates collected so far under looser thresholds are pruned to it, so that the list holds ~need entries
// (+ ties) whatever the number of levels.
__global__ __launch_bounds__(BLOCK) void pick_kernel(const PickParams p) {
    __shared__ uint32_t hist[320];
    __shared__ uint32_t res[2];
    __shared__ uint32_t wsum[BLOCK / 64];
    __shared__ uint32_t base;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (q >= p.nq) return;
    for (uint32_t i = tid; i < 320; i += BLOCK) hist[i] = i < NBINS ? p.ghist[(uint64_t)q * HB + i] : 0u;
    __syncthreads();
    uint32_t bin, less;
    block_find_cut(hist, NBINS, p.need, res, bin, less);
    // never loosen the threshold (the histogram only holds bins <= the threshold its rows were scanned under)
    const uint32_t tau0 = 0x7FFFFFFFu - p.bias[q];
    const uint32_t tau = bin < tau0 ? bin : tau0;
    if (tid == 0) { p.bias[q] = 0x7FFFFFFFu - tau; base = 0; }
    if (!p.cand) return;
    const uint32_t total = p.cnt[(uint64_t)q * CNT_STRIDE];
    if (total > p.cap) return;            // overflowed: select_kernel flags it, the host reruns the query exactly
    uint64_t* list = p.cand + (uint64_t)q * p.cap;
    __syncthreads();
    // in-place stable compaction, one 256-entry chunk at a time: a chunk is read into registers before anything
    // of it is written, and the write position never passes the read position
    for (uint32_t start = 0; start < total; start += BLOCK) {
        const uint32_t i = start + tid;
        uint64_t c = 0;
        bool keep = false;
        if (i < total) { c = list[i]; keep = (uint32_t)(c >> 48) <= tau; }
        const uint64_t ball = __ballot(keep);
        const uint32_t lane = tid & 63, wave = tid >> 6;
        const uint32_t before = (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = (uint32_t)__popcll(ball);
        __syncthreads();
        uint32_t off = base;
        for (uint32_t w = 0; w < wave; ++w) off += wsum[w];
        __syncthreads();
        if (keep) list[off + before] = c;
        if (tid == 0) base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    if (tid == 0) p.cnt[(uint64_t)q * CNT_STRIDE] = base;
}

// ---------------------------------------------------------------------------------------------
// select_kernel<KW>: one block per query.  Exact top-k of the candidate list under (hamming, key).
//   1. histogram of hamming over the candidates -> cut h*, `less` = candidates below it
//   2. tie class h == h*: MSB-first radix select on the key (8 bits per pass) until the r smallest
//      keys of the class are pinned down
//   3. compact the keff winners into LDS, bitonic sort by (hamming, key_hi, key_lo), emit records
// dynamic LDS: sh[P] u32 | pad | klo[P] u64 | khi[P] u64 (KW == 2) | row[P] u32 (ROWS)
// ROWS: the segment row of every result travels through the sort and is written beside the records (out_rows) -- the
// simprint scoring kernels (simprint_score.hip) read the stored code and its document frequency by row, not by key.
// ---------------------------------------------------------------------------------------------
struct SelectParams {
    const uint32_t* cnt;      // [nq_pad * CNT_STRIDE]
    const uint64_t* cand;     // [nq_pad][cap]
    uint64_t cap;
    const uint64_t* keys;     // segment key column [rows*KW]
    const uint16_t* rank;     // [257] hamming -> order-preserving distance rank for this prefix
    Record* out;              // [nq][k]
    uint32_t* out_count;      // [nq]
    uint32_t* overflow;       // [nq] set to 1 when the candidate list overflowed
    uint32_t k;
    uint32_t P;               // power of two >= min(k, cap)
    uint32_t prefix_bits;
    uint32_t q_base;          // block b serves query q_base + b
    uint32_t overflow_count;  // what out_count[q] becomes when the candidate list overflowed: 0, or COUNT_OVERFLOW for callers
                              // that cannot look at the flags before the results travel on (search_device_async)
    uint32_t* out_rows;       // [nq][k] segment row of every record (select_kernel<KW, true> only)
    uint32_t* out_kth;        // nullable [nq]: hamming of the query's LAST result (0 when it has none) -- what the host needs of a
                              // result block that stays on the device to seed the next batch's threshold hint
};

template <int KW>
__device__ __forceinline__ void load_key(const uint64_t* keys, uint64_t row, uint64_t& hi, uint64_t& lo) {
    if constexpr (KW == 2) { hi = keys[2 * row]; lo = keys[2 * row + 1]; }
    else { hi = 0; lo = keys[row]; }
}
// digit d (0 = most significant byte) of a KW-word key
template <int KW>
__device__ __forceinline__ uint32_t key_digit(uint64_t hi, uint64_t lo, int d) {
    if constexpr (KW == 2) return d < 8 ? (uint32_t)(hi >> (56 - 8 * d)) & 255u : (uint32_t)(lo >> (56 - 8 * (d - 8))) & 255u;
    else return (uint32_t)(lo >> (56 - 8 * d)) & 255u;
}
// the key with everything below its first d bytes cleared
template <int KW>
__device__ __forceinline__ void key_top(uint64_t hi, uint64_t lo, int d, uint64_t& thi, uint64_t& tlo) {
    if constexpr (KW == 2) {
        if (d >= 16) { thi = hi; tlo = lo; }
        else if (d >= 8) { thi = hi; tlo = d == 8 ? 0 : lo & (~0ULL << (64 - 8 * (d - 8))); }
        else { thi = d == 0 ? 0 : hi & (~0ULL << (64 - 8 * d)); tlo = 0; }
    } else {
        thi = 0;
        tlo = d >= 8 ? lo : (d == 0 ? 0 : lo & (~0ULL << (64 - 8 * d)));
    }
}

// NT: threads per block.  One block per query holds its sort buffer in LDS (P = 4 096 slots with 128-bit keys and rows: 112 KB, one
// block per CU): with 256 threads that CU runs FOUR waves through ~60 bitonic stages and two rounds of dependent gathers -- a k = 400
// select took 47 us for 16 queries, 93 us for 512.  Large buffers (P >= 1 024) are launched with 1 024 threads.
// select_body: the whole select of query q over `total` candidates at `cand` (the block's dynamic LDS at smem); every thread of the block calls it.
template <int KW, bool ROWS, int NT>
__device__ __forceinline__ void select_body(const SelectParams& p, const uint32_t q, const uint64_t* cand, const uint32_t total, unsigned char* smem) {
    __shared__ uint32_t hist[320];
    __shared__ uint32_t res[2];
    __shared__ uint32_t n_out;
    const uint32_t P = p.P;
    uint32_t* sh = reinterpret_cast<uint32_t*>(smem);
    uint64_t* sklo = reinterpret_cast<uint64_t*>(smem + (((size_t)P * 4 + 15) & ~(size_t)15));
    uint64_t* skhi = sklo + P;   // only touched when KW == 2
    uint32_t* srow = reinterpret_cast<uint32_t*>(sklo + (size_t)P * KW);   // only touched when ROWS

    const uint32_t tid = threadIdx.x;
    if (tid == 0) p.overflow[q] = total > p.cap ? 1u : 0u;   // always written: the host never has to clear the flags
    if (total > p.cap) {             // candidate list overflowed: host reruns this query exactly
        if (tid == 0) { p.out_count[q] = p.overflow_count; if (p.out_kth) p.out_kth[q] = 0; }
        return;
    }
    const uint32_t keff = p.k < total ? p.k : total;
    if (keff == 0) {
        if (tid == 0) { p.out_count[q] = 0; if (p.out_kth) p.out_kth[q] = 0; }
        return;
    }

    // 1. cut on the hamming distance
    for (uint32_t i = tid; i < 320; i += NT) hist[i] = 0;
    if (tid == 0) n_out = 0;
    __syncthreads();
    for (uint32_t i = tid; i < total; i += NT) atomicAdd(&hist[(uint32_t)(cand[i] >> 48)], 1u);
    __syncthreads();
    uint32_t hstar, less;
    block_find_cut(hist, NBINS, keff, res, hstar, less);
    uint32_t tie = hist[hstar];
    uint32_t r = keff - less;            // 1 <= r <= tie
    __syncthreads();

    // 2. radix select on the key inside the tie class -- skipped when everything up to and including the tie
    //    class fits the sort buffer (the usual case: a few dozen rows): the sort then orders the ties by key and
    //    the first keff entries are the answer, without up to 8*KW dependent passes over the gathered keys
    uint64_t phi = 0, plo = 0;           // selected key prefix (first d bytes)
    int d = 0;
    const bool fits = less + tie <= P;
    while (!fits && r < tie && d < KW * 8) {
        for (uint32_t i = tid; i < 320; i += NT) hist[i] = 0;
        __syncthreads();
        for (uint32_t i = tid; i < total; i += NT) {
            const uint64_t c = cand[i];
            if ((uint32_t)(c >> 48) != hstar) continue;
            uint64_t khi, klo, thi, tlo;
            load_key<KW>(p.keys, c & 0xFFFFFFFFFFFFULL, khi, klo);
            key_top<KW>(khi, klo, d, thi, tlo);
            if (thi == phi && tlo == plo) atomicAdd(&hist[key_digit<KW>(khi, klo, d)], 1u);
        }
        __syncthreads();
        uint32_t b, below;
        block_find_cut(hist, 256, r, res, b, below);
        tie = hist[b];
        r -= below;
        if (KW == 2 && d < 8) phi |= (uint64_t)b << (56 - 8 * d);
        else plo |= (uint64_t)b << (56 - 8 * (KW == 2 ? d - 8 : d));
        ++d;
        __syncthreads();
    }
    // winners: hamming < h*, or hamming == h* and top-d key bytes <= selected prefix
    // (when the loop stopped with r == tie every key sharing the prefix is taken)

    // 3. compact
    for (uint32_t i = tid; i < total; i += NT) {
        const uint64_t c = cand[i];
        const uint32_t h = (uint32_t)(c >> 48);
        if (h > hstar) continue;
        uint64_t khi, klo;
        load_key<KW>(p.keys, c & 0xFFFFFFFFFFFFULL, khi, klo);
        if (h == hstar) {
            uint64_t thi, tlo;
            key_top<KW>(khi, klo, d, thi, tlo);
            if (thi > phi || (thi == phi && tlo > plo)) continue;
        }
        const uint32_t pos = atomicAdd(&n_out, 1u);
        if (pos < P) { sh[pos] = h; sklo[pos] = klo; if (KW == 2) skhi[pos] = khi; if (ROWS) srow[pos] = (uint32_t)(c & 0xFFFFFFFFFFFFULL); }
    }
    __syncthreads();
    const uint32_t got = n_out < P ? n_out : P;     // == keff when keys are unique
    // sort only as many slots as hold winners (a range-limited search asks for a large k and finds few rows)
    uint32_t Ps = 1;
    while (Ps < got) Ps <<= 1;
    for (uint32_t i = got + tid; i < Ps; i += NT) { sh[i] = 0xFFFFFFFFu; sklo[i] = ~0ULL; if (KW == 2) skhi[i] = ~0ULL; }
    __syncthreads();

    // bitonic sort ascending by (h, khi, klo)
    for (uint32_t size = 2; size <= Ps; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = tid; i < (Ps >> 1); i += NT) {
                const uint32_t lo_i = 2 * i - (i & (stride - 1));
                const uint32_t hi_i = lo_i + stride;
                const bool up = (lo_i & size) == 0;
                const uint32_t ha = sh[lo_i], hb = sh[hi_i];
                const uint64_t la = sklo[lo_i], lb = sklo[hi_i];
                uint64_t ka = 0, kb = 0;
                if (KW == 2) { ka = skhi[lo_i]; kb = skhi[hi_i]; }
                const bool a_gt_b = ha != hb ? ha > hb : (ka != kb ? ka > kb : la > lb);
                if (a_gt_b == up) {
                    sh[lo_i] = hb; sh[hi_i] = ha;
                    sklo[lo_i] = lb; sklo[hi_i] = la;
                    if (KW == 2) { skhi[lo_i] = kb; skhi[hi_i] = ka; }
                    if (ROWS) { const uint32_t ra = srow[lo_i]; srow[lo_i] = srow[hi_i]; srow[hi_i] = ra; }
                }
            }
            __syncthreads();
        }
    }
    const uint32_t nres = got < keff ? got : keff;
    for (uint32_t i = tid; i < nres; i += NT) {
        Record rec;
        rec.key_hi = KW == 2 ? skhi[i] : 0;
        rec.key_lo = sklo[i];
        rec.dist_rank = p.rank[sh[i]];
        rec.hamming = (uint16_t)sh[i];
        rec.prefix_bits = (uint16_t)p.prefix_bits;
        p.out[(uint64_t)q * p.k + i] = rec;
        if (ROWS) p.out_rows[(uint64_t)q * p.k + i] = srow[i];
    }
    if (tid == 0) { p.out_count[q] = nres; if (p.out_kth) p.out_kth[q] = nres ? sh[nres - 1] : 0; }
}

template <int KW, bool ROWS = false, int NT = BLOCK>
__global__ __launch_bounds__(NT) void select_kernel(const SelectParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t q = p.q_base + blockIdx.x;
    select_body<KW, ROWS, NT>(p, q, p.cand + (uint64_t)q * p.cap, p.cnt[(uint64_t)q * CNT_STRIDE], smem);
}

// ---------------------------------------------------------------------------------------------
// tiny_search_kernel: a segment of a few thousand rows (option "tiny_rows", default 16 384) answered by ONE launch, one block per
// query: the block computes the distance of every row (range-limited searches keep the rows within the radius), lists them as
// candidates and runs the select on them -- exact by construction, no threshold to find or verify.  Such a search used to be
// three launches (threshold, scan, select: ~20 us of launches and hand-overs around ~5 us of work): the reference's own call shape
// on the index sizes its deployment guide names (BASELINE config 1: 2 500 rows per unit type).
// ---------------------------------------------------------------------------------------------
struct TinyParams {
    const uint64_t* col[4];
    const uint64_t* queries;  // [nq_pad][4] device copy (used when !use_inline)
    uint64_t* cand;           // [nq_pad][cap]
    uint64_t mask_last;       // of the last compared word
    uint32_t n_rows;          // <= cap
    uint32_t W;               // compared words
    int32_t radius;           // >= 0: rows within it only
    uint32_t use_inline;
};
template <int KW, bool ROWS, int NT>
__global__ __launch_bounds__(NT) void tiny_search_kernel(const TinyParams t, const SelectParams p, const InlineQueries iq) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t n_cand;
    const uint32_t tid = threadIdx.x;
    const uint32_t q = p.q_base + blockIdx.x;
    uint64_t qw[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) qw[w] = t.use_inline ? iq.w[(q % INLINE_QUERIES) * 4 + w] : t.queries[(uint64_t)q * 4 + w];
    if (tid == 0) n_cand = 0;
    __syncthreads();
    uint64_t* const cand = t.cand + (uint64_t)q * p.cap;
#pragma unroll 4
    for (uint32_t row = tid; row < t.n_rows; row += NT) {
        uint32_t hd = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) {
            if (w < t.W) {
                uint64_t x = t.col[w][row] ^ qw[w];
                if (w + 1 == t.W) x &= t.mask_last;
                hd += (uint32_t)__popcll(x);
            }
        }
        if (t.radius < 0 || hd <= (uint32_t)t.radius) cand[atomicAdd(&n_cand, 1u)] = ((uint64_t)hd << 48) | row;
    }
    __syncthreads();             // (a workgroup-scope fence with it: the block reads back its own candidate words)
    select_body<KW, ROWS, NT>(p, q, cand, n_cand, smem);
}

// ---------------------------------------------------------------------------------------------
// fullhist_kernel: exact hamming histogram of ONE query over a whole segment (overflow fallback)
// ---------------------------------------------------------------------------------------------
struct FullHistParams {
    const uint64_t* col[4];
    uint64_t n_rows;
    const uint64_t* query;    // [4]
    uint32_t* ghist;          // [HB] (zeroed by the host)
    uint32_t W;
    uint64_t mask_last;
};
__global__ __launch_bounds__(BLOCK) void fullhist_kernel(const FullHistParams p) {
    __shared__ uint32_t hist[320];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < 320; i += BLOCK) hist[i] = 0;
    __syncthreads();
    uint64_t qw[4];
    for (uint32_t w = 0; w < 4; ++w) qw[w] = w < p.W ? p.query[w] : 0;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + tid; r < p.n_rows; r += (uint64_t)gridDim.x * BLOCK) {
        uint32_t h = 0;
        for (uint32_t w = 0; w < p.W; ++w) {
            uint64_t x = p.col[w][r] ^ qw[w];
            if (w == p.W - 1) x &= p.mask_last;
            h += (uint32_t)__builtin_popcountll(x);
        }
        atomicAdd(&hist[h], 1u);
    }
    __syncthreads();
    for (uint32_t i = tid; i < NBINS; i += BLOCK)
        if (hist[i]) atomicAdd(&p.ghist[i], hist[i]);
}

// ---------------------------------------------------------------------------------------------
// Overflow fallback for tie classes too large to collect: radix select on the KEY over the rows of
// the table itself.  One query at a time; plain (unpipelined) scans -- this path only runs for
// adversarial data (e.g. millions of identical codes).
//   fb_keyhist_kernel  histogram of key byte `d` over rows with hamming == tau whose first d key
//                      bytes equal the selected prefix
//   fb_collect_kernel  append rows with hamming < tau, or hamming == tau and top-d key bytes <= prefix
// ---------------------------------------------------------------------------------------------
struct FbParams {
    const uint64_t* col[4];
    const uint64_t* keys;
    uint64_t n_rows;
    const uint64_t* query;    // [4]
    uint32_t W, KW;
    uint64_t mask_last;
    uint32_t tau;
    int d;                    // key bytes already fixed
    uint64_t phi, plo;        // the fixed prefix (as a key with the lower bytes cleared)
    uint32_t* ghist;          // [256]  (fb_keyhist_kernel)
    uint32_t* cnt;            // [1]    (fb_collect_kernel)
    uint64_t* cand;           // [cap]
    uint32_t cap;
};
__device__ __forceinline__ uint32_t fb_hamming(const FbParams& p, const uint64_t (&qw)[4], uint64_t r) {
    uint32_t h = 0;
    for (uint32_t w = 0; w < p.W; ++w) {
        uint64_t x = p.col[w][r] ^ qw[w];
        if (w == p.W - 1) x &= p.mask_last;
        h += (uint32_t)__builtin_popcountll(x);
    }
    return h;
}
template <int KW>
__global__ __launch_bounds__(BLOCK) void fb_keyhist_kernel(const FbParams p) {
    __shared__ uint32_t hist[256];
    const uint32_t tid = threadIdx.x;
    hist[tid] = 0;
    __syncthreads();
    uint64_t qw[4];
    for (uint32_t w = 0; w < 4; ++w) qw[w] = w < p.W ? p.query[w] : 0;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + tid; r < p.n_rows; r += (uint64_t)gridDim.x * BLOCK) {
        if (fb_hamming(p, qw, r) != p.tau) continue;
        uint64_t khi, klo, thi, tlo;
        load_key<KW>(p.keys, r, khi, klo);
        key_top<KW>(khi, klo, p.d, thi, tlo);
        if (thi == p.phi && tlo == p.plo) atomicAdd(&hist[key_digit<KW>(khi, klo, p.d)], 1u);
    }
    __syncthreads();
    if (hist[tid]) atomicAdd(&p.ghist[tid], hist[tid]);
}
template <int KW>
__global__ __launch_bounds__(BLOCK) void fb_collect_kernel(const FbParams p) {
    const uint32_t tid = threadIdx.x;
    uint64_t qw[4];
    for (uint32_t w = 0; w < 4; ++w) qw[w] = w < p.W ? p.query[w] : 0;
    for (uint64_t r = (uint64_t)blockIdx.x * BLOCK + tid; r < p.n_rows; r += (uint64_t)gridDim.x * BLOCK) {
        const uint32_t h = fb_hamming(p, qw, r);
        if (h > p.tau) continue;
        if (h == p.tau && p.d > 0) {
            uint64_t khi, klo, thi, tlo;
            load_key<KW>(p.keys, r, khi, klo);
            key_top<KW>(khi, klo, p.d, thi, tlo);
            if (thi > p.phi || (thi == p.phi && tlo > p.plo)) continue;
        }
        const uint32_t slot = atomicAdd(p.cnt, 1u);
        if (slot < p.cap) p.cand[slot] = ((uint64_t)h << 48) | r;
    }
}

// ---------------------------------------------------------------------------------------------
// merge_kernel: per query, k-way merge of n_lists record lists, each sorted by (dist_rank, key).
//   rank of an element = its position in its own list + the number of elements of every other
//   list that sort before it (binary search); ranks are distinct because keys are.
// ---------------------------------------------------------------------------------------------
struct MergeParams {
    const unsigned char* lists;   // list l: records at lists + l*list_stride, laid out [nq][k]
    const unsigned char* counts;  // list l: counts  at counts + l*count_stride, laid out [nq]
    uint64_t list_stride;         // bytes between the record blocks of consecutive lists
    uint64_t count_stride;        // bytes between the count blocks of consecutive lists
    Record* out;                  // [nq][k]
    uint32_t* out_count;          // [nq]
    uint32_t n_lists, nq, k;
};
__device__ __forceinline__ bool rec_less(const Record& a, const Record& b) {
    if (a.dist_rank != b.dist_rank) return a.dist_rank < b.dist_rank;
    if (a.key_hi != b.key_hi) return a.key_hi < b.key_hi;
    return a.key_lo < b.key_lo;
}
__global__ __launch_bounds__(BLOCK) void merge_kernel(const MergeParams p) {
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    auto list_of = [&](uint32_t l) { return reinterpret_cast<const Record*>(p.lists + (uint64_t)l * p.list_stride) + (uint64_t)q * p.k; };
    auto raw_count = [&](uint32_t l) { return reinterpret_cast<const uint32_t*>(p.counts + (uint64_t)l * p.count_stride)[q]; };
    auto count_of = [&](uint32_t l) {
        const uint32_t c = raw_count(l);
        return c < p.k ? c : p.k;
    };
    uint32_t total = 0;
    for (uint32_t l = 0; l < p.n_lists; ++l) {
        if (raw_count(l) == COUNT_OVERFLOW) {      // a list that could not be completed without the host: pass the marker on
            if (tid == 0) p.out_count[q] = COUNT_OVERFLOW;
            return;
        }
        total += count_of(l);
    }
    const uint32_t keff = total < p.k ? total : p.k;
    for (uint32_t e = tid; e < p.n_lists * p.k; e += BLOCK) {
        const uint32_t l = e / p.k, i = e % p.k;
        if (i >= count_of(l)) continue;
        const Record me = list_of(l)[i];
        uint32_t rank = i;
        for (uint32_t o = 0; o < p.n_lists && rank < keff; ++o) {
            if (o == l) continue;
            const Record* lst = list_of(o);
            uint32_t lo = 0, hi = count_of(o);
            // elements of list o sorting before `me`; an (impossible) exact tie goes to the lower list id
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                const bool before = o < l ? !rec_less(me, lst[mid]) : rec_less(lst[mid], me);
                if (before) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < keff) p.out[(uint64_t)q * p.k + rank] = me;
    }
    if (tid == 0) p.out_count[q] = keff;
}

// ---------------------------------------------------------------------------------------------
// document frequency: distinct assets in one query's collision list.  The list is ordered by key, so
// the rows of one asset (= first key word of a 2-word key) are adjacent: count the boundaries.
// ---------------------------------------------------------------------------------------------
struct DistinctParams {
    const Record* rec;        // [nq][k] ascending (dist_rank, key)
    const uint32_t* count;    // [nq]
    uint32_t* out;            // [nq]
    uint32_t k, KW;
};
__global__ __launch_bounds__(BLOCK) void distinct_kernel(const DistinctParams p) {
    __shared__ uint32_t total;
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) total = 0;
    __syncthreads();
    const uint32_t n = p.count[q] < p.k ? p.count[q] : p.k;
    const Record* r = p.rec + (uint64_t)q * p.k;
    uint32_t mine = 0;
    for (uint32_t i = tid; i < n; i += BLOCK)
        mine += (i == 0 || p.KW == 1 || r[i].key_hi != r[i - 1].key_hi) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((tid & 63) == 0 && mine) atomicAdd(&total, mine);
    __syncthreads();
    if (tid == 0) p.out[q] = total;
}

// ---------------------------------------------------------------------------------------------
// utilities
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

struct FillParams {
    uint64_t* col[4];
    uint64_t* keys;
    uint64_t dst_row;      // first destination row in the segment
    uint64_t n;
    uint64_t seed, first_row, key_base;
    uint32_t W, KW;
    uint64_t mask_last;    // codes shorter than W words keep zero padding
};
__global__ __launch_bounds__(BLOCK) void fill_kernel(const FillParams p) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < p.n; i += (uint64_t)gridDim.x * BLOCK) {
        const uint64_t src = p.first_row + i;
        for (uint32_t w = 0; w < p.W; ++w) {
            uint64_t v = splitmix64(p.seed + 4 * src + w);
            if (w == p.W - 1) v &= p.mask_last;
            p.col[w][p.dst_row + i] = v;
        }
        if (p.KW == 2) { p.keys[2 * (p.dst_row + i)] = 0; p.keys[2 * (p.dst_row + i) + 1] = p.key_base + src; }
        else p.keys[p.dst_row + i] = p.key_base + src;
    }
}

// sequential row moves (swap-with-last removal): lane c owns column c for every move, in order
struct MoveParams {
    uint64_t* col[4];
    uint64_t* keys;
    const uint64_t* moves;   // [n_moves][2] = (dst, src)
    uint64_t n_moves;
    uint32_t W, KW;
};
__global__ void move_rows_kernel(const MoveParams p) {
    const uint32_t c = threadIdx.x;
    if (c < p.W) {
        uint64_t* col = p.col[c];
        for (uint64_t m = 0; m < p.n_moves; ++m) col[p.moves[2 * m]] = col[p.moves[2 * m + 1]];
    } else if (c < p.W + p.KW) {
        const uint32_t kw = c - p.W;
        for (uint64_t m = 0; m < p.n_moves; ++m) p.keys[p.moves[2 * m] * p.KW + kw] = p.keys[p.moves[2 * m + 1] * p.KW + kw];
    }
}

struct GatherParams {
    const uint64_t* col[4];
    const uint64_t* rows;    // [n]
    uint64_t* out;           // [n][W]
    uint64_t n;
    uint32_t W;
};
__global__ __launch_bounds__(BLOCK) void gather_rows_kernel(const GatherParams p) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < p.n; i += (uint64_t)gridDim.x * BLOCK)
        for (uint32_t w = 0; w < p.W; ++w) p.out[i * p.W + w] = p.col[w][p.rows[i]];
}

// ingest: rows handed over row-major [n][MW] -> the segment's word columns (the last kept word masked to the code length)
struct SplitParams {
    uint64_t* col[4];
    const uint64_t* rows;    // [n][MW]
    uint64_t dst_row, n;
    uint32_t W, MW;
    uint64_t mask_last;
};
__global__ __launch_bounds__(BLOCK) void split_rows_kernel(const SplitParams p) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < p.n; i += (uint64_t)gridDim.x * BLOCK)
        for (uint32_t w = 0; w < p.W; ++w) {
            uint64_t v = p.rows[i * p.MW + w];
            if (w == p.W - 1) v &= p.mask_last;
            p.col[w][p.dst_row + i] = v;
        }
}

// out[i] = src[rows[i]]  (document-frequency column lookups)
__global__ __launch_bounds__(BLOCK) void gather_u32_kernel(const uint32_t* src, const uint64_t* rows, uint32_t* out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) out[i] = src[rows[i]];
}

}  // namespace isk
