// mfma_scan.hip -- the collect scan for LARGE query batches on the int8 matrix cores of gfx950 (MI355X).
//
// Same contract as scan_kernel (kernels.hip.h): for every row of [row_begin, n_rows) and every query, append
// (hamming, row) to the query's candidate list when hamming over the compared prefix <= tau_q
// (reference call sites: iscc_search/indexes/usearch/index.py:2037, iscc_search/indexes/simprint/usearch_core.py:165;
// metric: docs/explanation/similarity-search.md:24-29).  What differs is the arithmetic:
//
//   hamming(row, q) = popc(q) + dot(row bits as 0/1, query bits as +1/-1)            (exact integers)
//
// so the (rows x queries x bits) work is a dense int8 contraction.  The XOR + popcount kernel needs 4.5 VALU
// instructions per (row, query, 64-bit word) -- two of them v_bcnt, which gfx950 issues at one wave64 per 4 cycles --
// and is VALU-bound from ~11 queries per pass on.  Here a wave owns T = 2 tiles of 32 rows:
// each lane expands ITS 32 bits of one row per word into 32 bytes of 0/1 (the A operands of two
// v_mfma_i32_32x32x32_i8), once per step, and then walks every query group of the block's chunk.  A group is 32
// queries pre-expanded to +1/-1 bytes in LDS (two ds_read_b128 per lane and word); 2*W MFMAs per tile give the 32 x 32
// dot products; a lane's 16 results per tile all belong to ONE query (C/D column = lane & 31), so 16 v_min3_i32 fold
// them and one compare against thr_q = tau_q - popc(q) decides whether the lane enters the rare emit path.
// Masked prefixes (NPHD between codes of different lengths) cost nothing: the query bytes beyond the prefix are 0.
//
// The bound is the matrix pipe: 64 * W cycles per 1 024 (row, query) pairs per SIMD, against ~290 * W VALU cycles.
// Rows cross the memory system once per chunk of up to 1 024 / W queries.  Measured numbers: DESIGN.md section 4.
//
// Built with -mllvm -amdgpu-mfma-vgpr-form=1: hipcc otherwise puts the accumulators in AGPRs and pays one
// v_accvgpr_read per result before the fold (16 extra VALU instructions per tile and group).
#include "mfma_scan.h"

#include <hip/hip_runtime.h>

namespace isk {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int MT = 2;   // row tiles (32 rows) per wave and step

__device__ __forceinline__ int min3i(int a, int b, int c) { return min(min(a, b), c); }

// Dword j (0..7) of a 32-bit half: (x >> j) & 0x01010101 holds bits j, j+8, j+16, j+24 as bytes 0/1.  MFMA 2*w + (j >> 2)
// takes it as VGPR j & 3 of its fragment -- for rows (A) and queries (B) alike, so the k order inside the
// instruction does not matter.
__device__ __forceinline__ uint32_t spread(uint32_t x, int j) { return (x >> j) & 0x01010101u; }

// LDS image of a chunk: B fragments [groups][2 * W][64] v4i | thr[groups * 32] | popc[groups * 32]
template <int W, int MODE>
__global__ __launch_bounds__(BLOCK) void mfma_scan_kernel(const ScanParams p, const uint32_t groups) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    v4i* lb = reinterpret_cast<v4i*>(smem);
    int* lthr = reinterpret_cast<int*>(smem + (size_t)groups * (2 * W) * 64 * 16);
    int* lpop = lthr + groups * 32;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t r = lane & 31, h = lane >> 5;
    const uint32_t q0 = blockIdx.y * groups * 32;       // first query of this block's chunk

    // prologue: expand the chunk's queries to +1 / -1 bytes (0 beyond the compared prefix and for padding queries)
    for (uint32_t i = tid; i < groups * 32 * 2 * W; i += BLOCK) {
        const uint32_t ql = i / (2 * W), rest = i % (2 * W), w = rest >> 1, hh = rest & 1;
        const uint32_t q = q0 + ql;
        const bool live = q < p.nq_pad;
        const uint64_t qw = live ? p.queries[(uint64_t)q * 4 + w] : 0;
        uint32_t x = hh ? (uint32_t)(qw >> 32) : (uint32_t)qw;
        uint32_t m = live ? 0xFFFFFFFFu : 0u;
        if (w == W - 1) m &= hh ? p.mask_hi : p.mask_lo;
        const uint32_t g = ql >> 5, c = ql & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t s = (0x01010101u | (spread(x, j) * 0xFEu)) & (spread(m, j) * 0xFFu);
            reinterpret_cast<uint32_t*>(&lb[((size_t)g * (2 * W) + 2 * w + (j >> 2)) * 64 + hh * 32 + c])[j & 3] = s;
        }
    }
    for (uint32_t ql = tid; ql < groups * 32; ql += BLOCK) {
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
        lthr[ql] = tau - pc;                            // hamming <= tau  <=>  dot <= tau - popc(q)
    }
    __syncthreads();

    const uint64_t first = p.row_begin / (32 * MT);                         // row_begin is a multiple of 64
    const uint64_t nsteps = (p.n_rows + 32 * MT - 1) / (32 * MT);           // the last step may be partial
    const uint64_t stride = (uint64_t)gridDim.x * (BLOCK / 64);
    uint64_t step = first + (uint64_t)blockIdx.x * (BLOCK / 64) + wave;
    if (step >= nsteps) return;
    const uint64_t last_row = p.n_rows - 1;

    struct Acc { v16i t[MT]; };
    const uint32_t* col32[W];
#pragma unroll
    for (int w = 0; w < W; ++w) col32[w] = reinterpret_cast<const uint32_t*>(p.col[w]);

    // a lane's 16 results per tile all belong to query g * 32 + (lane & 31): fold them, compare once
    auto reduce = [&](const Acc& acc, int thr, uint32_t g, uint64_t st) {
        int m = min3i(acc.t[0][0], acc.t[0][1], acc.t[0][2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) m = min3i(m, acc.t[0][i], acc.t[0][i + 1]);
        m = min3i(m, acc.t[0][15], acc.t[1][0]);
#pragma unroll
        for (int i = 1; i < 15; i += 2) m = min3i(m, acc.t[1][i], acc.t[1][i + 1]);
        m = min(m, acc.t[1][15]);
        if (__builtin_expect(m <= thr, 0)) {
            // rare: result `reg` of tile t is row (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) of that tile.
            // The step number is laundered so that hipcc does not hoist 32 row numbers per lane out of the group loop
            // (that cost 64 VGPRs in the hot loop for a path taken once in ~10^3 group-steps).
            uint32_t st_lo = (uint32_t)st, st_hi = (uint32_t)(st >> 32);
            asm volatile("" : "+v"(st_lo), "+v"(st_hi));
            const uint64_t base = (((uint64_t)st_hi << 32) | st_lo) * (32 * MT) + 4 * h;
            const uint32_t ql = g * 32 + r;
            const int pc = lpop[ql];
#pragma unroll
            for (int t = 0; t < MT; ++t)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    if (acc.t[t][reg] <= thr) {
                        const uint64_t row = base + (uint32_t)(t * 32 + (reg & 3) + 8 * (reg >> 2));
                        if (row <= last_row) emit<MODE>(p, q0 + ql, (uint32_t)(acc.t[t][reg] + pc), row);
                    }
                }
        }
    };

    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i a[MT][2 * W];
    // one word of one group: 2 MFMAs per tile into the group's accumulators
    auto mm = [&](Acc& acc, int w, const v4i& b0, const v4i& b1) {
#pragma unroll
        for (int t = 0; t < MT; ++t) acc.t[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t][2 * w], b0, w == 0 ? zero : acc.t[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc.t[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[t][2 * w + 1], b1, acc.t[t], 0, 0, 0);
    };
    const v4i* lbl = lb + lane;
    const int* lt = lthr + r;
    auto row_of = [&](uint64_t st, int t) { const uint64_t row = (st * MT + t) * 32 + r; return row <= last_row ? row : last_row; };

    uint32_t x[MT][W], xn[MT][W];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int w = 0; w < W; ++w) x[t][w] = col32[w][row_of(step, t) * 2 + h];
    for (; step < nsteps; step += stride) {
        const uint64_t ns = step + stride < nsteps ? step + stride : step;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w) xn[t][w] = col32[w][row_of(ns, t) * 2 + h];     // next step's rows, in flight during this one
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[t][2 * w + (j >> 2)][j & 3] = (int)spread(x[t][w], j);

        // Software pipeline over the (group, word) sequence: two B buffers (one word each) and two accumulator sets.
        // The fragments of the NEXT word are requested right after the MFMAs of the current one are issued (their buffer
        // was consumed one stage earlier), and the results of group g are folded while the MFMAs of group g + 1 run.
        v4i bx0, bx1, by0, by1;
        Acc accX, accY;
        int thrX = 0, thrY = lt[0];
        by0 = lbl[0];
        by1 = lbl[64];
        // stage(g, w) for group parity P: consume buffer ((P * W + w) & 1), prefetch the next word into the other one
        auto stage = [&](Acc& acc, uint32_t g, int w, bool y_buf, bool more) {
            const uint32_t nxt = (g * W + w + 1) * 128;
            if (y_buf) {
                mm(acc, w, by0, by1);
                if (more) { bx0 = lbl[nxt]; bx1 = lbl[nxt + 64]; }
            } else {
                mm(acc, w, bx0, bx1);
                if (more) { by0 = lbl[nxt]; by1 = lbl[nxt + 64]; }
            }
        };
        // group 0
#pragma unroll
        for (int w = 0; w < W; ++w) stage(accY, 0, w, (w & 1) == 0, true);
#pragma unroll 1
        for (uint32_t g = 1; g + 1 < groups; g += 2) {
            // odd group g -> accX; its first word sits in buffer parity (W & 1): Y when W is even
            thrX = lt[g * 32];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                stage(accX, g, w, ((W + w) & 1) == 0, true);
                if (w == 0) reduce(accY, thrY, g - 1, step);
            }
            // even group g + 1 -> accY; (2 * W + w) & 1 == w & 1
            thrY = lt[(g + 1) * 32];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                stage(accY, g + 1, w, (w & 1) == 0, true);
                if (w == 0) reduce(accX, thrX, g, step);
            }
        }
        // last (odd) group: nothing further to prefetch after its last word
        {
            const uint32_t g = groups - 1;
            thrX = lt[g * 32];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                stage(accX, g, w, ((W + w) & 1) == 0, w + 1 < W);
                if (w == 0) reduce(accY, thrY, g - 1, step);
            }
            reduce(accX, thrX, g, step);
        }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int w = 0; w < W; ++w) x[t][w] = xn[t][w];
    }
}

template <int W>
static int launch_w(int mode, dim3 grid, size_t lds, hipStream_t st, const ScanParams& p, uint32_t groups) {
    static bool attr_set[4] = {false, false, false, false};
    auto set = [&](const void* f) { return hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, MFMA_MAX_LDS); };
    const int mi = mode == MODE_COLLECT ? 0 : (mode == MODE_STRETCH ? 1 : 2);
    if (!attr_set[mi]) {
        hipError_t e = mi == 0 ? set(reinterpret_cast<const void*>(&mfma_scan_kernel<W, MODE_COLLECT>))
                     : mi == 1 ? set(reinterpret_cast<const void*>(&mfma_scan_kernel<W, MODE_STRETCH>))
                               : set(reinterpret_cast<const void*>(&mfma_scan_kernel<W, MODE_BOTH>));
        if (e != hipSuccess) return (int)e;
        attr_set[mi] = true;
    }
    if (mi == 0) hipLaunchKernelGGL((mfma_scan_kernel<W, MODE_COLLECT>), grid, dim3(BLOCK), lds, st, p, groups);
    else if (mi == 1) hipLaunchKernelGGL((mfma_scan_kernel<W, MODE_STRETCH>), grid, dim3(BLOCK), lds, st, p, groups);
    else hipLaunchKernelGGL((mfma_scan_kernel<W, MODE_BOTH>), grid, dim3(BLOCK), lds, st, p, groups);
    return 0;
}

uint32_t mfma_groups_per_chunk(int W, uint32_t nq_pad) {
    // LDS per group: 32 queries x (64 * W bytes of +1/-1 + thr + popc); keep two blocks per CU (<= 76 KB each)
    static const uint32_t max_groups[5] = {0, 32, 16, 10, 8};
    uint32_t need = (nq_pad + 31) / 32;
    need += need & 1;                         // the pipeline walks the groups in pairs
    if (need < 2) need = 2;
    return need < max_groups[W] ? need : max_groups[W];
}

size_t mfma_lds_bytes(int W, uint32_t groups) { return (size_t)groups * (2 * W) * 64 * 16 + (size_t)groups * 32 * 8; }

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
