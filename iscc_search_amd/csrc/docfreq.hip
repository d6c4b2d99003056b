// Document-frequency column: sort the rows of a segment by (code, asset), then count per code.
//
//   1. perm = 0..n-1; least-significant-first passes of a stable 64-bit radix sort (rocPRIM) over
//      the asset word of the key (2-word keys only) and then the code words W-1 .. 0: afterwards rows with
//      equal codes are adjacent and, inside such a run, ordered by asset.
//   2. head_code[i] = row i starts a new code, head_pair[i] = row i starts a new (code, asset) pair.
//      run = inclusive_scan(head_code) - 1.
//   3. count[run] = pair heads among the first dup_limit rows of the run; freq[perm[i]] = count[run(i)].
//
// The first dup_limit rows of a run in (asset, offset, size) order span whole assets in asset order plus at
// most one partial asset, so which pair heads fall below the limit does not depend on the order INSIDE
// an asset: sorting by the asset word alone is enough.
//
// A maintenance operation (runs after ingest, before the first approximate simprint search): HBM-bound
// radix passes, ~(W+1) x 2 reads+writes of 12 bytes per row.
#include <cstring>   // rocPRIM's texture iterator calls memset without including it

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <cerrno>
#include <cstdio>

#include "docfreq.h"

namespace iskdf {
namespace {

constexpr int BLOCK = 256;

struct Cols { const uint64_t* c[4]; };

__global__ __launch_bounds__(BLOCK) void iota_kernel(uint32_t* perm, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) perm[i] = (uint32_t)i;
}

// out[i] = src[perm[i] * stride]
__global__ __launch_bounds__(BLOCK) void gather_kernel(const uint64_t* src, uint32_t stride, const uint32_t* perm, uint64_t* out, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK)
        out[i] = src[(uint64_t)perm[i] * stride];
}

__global__ __launch_bounds__(BLOCK) void heads_kernel(const Cols cols, int W, const uint64_t* keys, int KW, const uint32_t* perm,
                                                      uint32_t* head_code, uint8_t* head_pair, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        uint32_t hc = 1, hp = 1;
        if (i > 0) {
            const uint64_t a = perm[i], b = perm[i - 1];
            bool same = true;
            for (int w = 0; w < W; ++w) same = same && cols.c[w][a] == cols.c[w][b];
            hc = same ? 0u : 1u;
            hp = (!same || KW == 1 || keys[2 * a] != keys[2 * b]) ? 1u : 0u;
        }
        head_code[i] = hc;
        head_pair[i] = (uint8_t)hp;
    }
}

__global__ __launch_bounds__(BLOCK) void run_start_kernel(const uint32_t* head_code, const uint32_t* run_incl, uint32_t* run_start, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK)
        if (head_code[i]) run_start[run_incl[i] - 1] = (uint32_t)i;
}

__global__ __launch_bounds__(BLOCK) void count_kernel(const uint8_t* head_pair, const uint32_t* run_incl, const uint32_t* run_start,
                                                      uint32_t dup_limit, uint32_t* count, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK) {
        if (!head_pair[i]) continue;
        const uint32_t run = run_incl[i] - 1;
        if ((uint32_t)i - run_start[run] < dup_limit) atomicAdd(&count[run], 1u);
    }
}

__global__ __launch_bounds__(BLOCK) void scatter_kernel(const uint32_t* perm, const uint32_t* run_incl, const uint32_t* count, uint32_t* freq, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += (uint64_t)gridDim.x * BLOCK)
        freq[perm[i]] = count[run_incl[i] - 1];
}

struct Scratch {
    void* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    int used = 0;
    ~Scratch() { for (int i = 0; i < used; ++i) if (p[i]) (void)hipFree(p[i]); }
    template <typename T>
    hipError_t alloc(T*& out, size_t count) {
        void* q = nullptr;
        hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) { p[used++] = q; out = static_cast<T*>(q); }
        return e;
    }
};

int failed(std::string* err, int code, const char* what, hipError_t e) {
    (void)hipGetLastError();   // do not leave the code behind for an unrelated later launch check
    if (err) {
        char buf[256];
        snprintf(buf, sizeof buf, "document-frequency column: %s: %s", what, hipGetErrorString(e));
        *err = buf;
    }
    return code;
}

}  // namespace

#define DFOK(expr, what)                                                                   \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return failed(err, e_ == hipErrorOutOfMemory ? -ENOMEM : -EIO, what, e_); \
    } while (0)

int build_freq_column(const uint64_t* const* col, int W, const uint64_t* keys, int KW, uint64_t n,
                      uint32_t dup_limit, uint32_t* freq_out, hipStream_t stream, std::string* err) {
    if (n == 0) return 0;
    if (n > 0xFFFFFFFFull || W < 1 || W > 4 || (KW != 1 && KW != 2) || dup_limit < 1) {
        if (err) *err = "document-frequency column: unsupported shape (rows must be < 2^32)";
        return -EINVAL;
    }
    Scratch mem;
    uint64_t *k0 = nullptr, *k1 = nullptr;
    uint32_t *p0 = nullptr, *p1 = nullptr, *head_code = nullptr, *run_incl = nullptr;
    uint8_t* head_pair = nullptr;
    DFOK(mem.alloc(k0, n), "hipMalloc");
    DFOK(mem.alloc(k1, n), "hipMalloc");
    DFOK(mem.alloc(p0, n), "hipMalloc");
    DFOK(mem.alloc(p1, n), "hipMalloc");
    DFOK(mem.alloc(head_code, n), "hipMalloc");
    DFOK(mem.alloc(run_incl, n), "hipMalloc");
    DFOK(mem.alloc(head_pair, n), "hipMalloc");
    size_t sort_bytes = 0, scan_bytes = 0;
    DFOK(rocprim::radix_sort_pairs(nullptr, sort_bytes, k0, k1, p0, p1, n, 0, 64, stream), "radix_sort_pairs(size)");
    DFOK(rocprim::inclusive_scan(nullptr, scan_bytes, head_code, run_incl, n, rocprim::plus<uint32_t>(), stream), "inclusive_scan(size)");
    unsigned char* temp = nullptr;
    const size_t temp_bytes = sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
    DFOK(mem.alloc(temp, temp_bytes), "hipMalloc");

    const uint32_t grid = (uint32_t)((n + BLOCK - 1) / BLOCK < 65536 ? (n + BLOCK - 1) / BLOCK : 65536);
    hipLaunchKernelGGL(iota_kernel, dim3(grid), dim3(BLOCK), 0, stream, p0, n);

    // least significant sort key first: asset word, then code words W-1 .. 0
    auto pass = [&](const uint64_t* src, uint32_t stride) -> hipError_t {
        hipLaunchKernelGGL(gather_kernel, dim3(grid), dim3(BLOCK), 0, stream, src, stride, p0, k0, n);
        size_t b = sort_bytes;
        hipError_t e = rocprim::radix_sort_pairs(temp, b, k0, k1, p0, p1, n, 0, 64, stream);
        uint32_t* t = p0; p0 = p1; p1 = t;
        return e;
    };
    if (KW == 2) DFOK(pass(keys, 2), "radix_sort_pairs(asset)");
    for (int w = W - 1; w >= 0; --w) DFOK(pass(col[w], 1), "radix_sort_pairs(code word)");

    Cols cols{};
    for (int w = 0; w < W; ++w) cols.c[w] = col[w];
    hipLaunchKernelGGL(heads_kernel, dim3(grid), dim3(BLOCK), 0, stream, cols, W, keys, KW, p0, head_code, head_pair, n);
    {
        size_t b = scan_bytes;
        DFOK(rocprim::inclusive_scan(temp, b, head_code, run_incl, n, rocprim::plus<uint32_t>(), stream), "inclusive_scan");
    }
    // the sort buffers are free again: reuse them for the per-run tables
    uint32_t* run_start = reinterpret_cast<uint32_t*>(k0);
    uint32_t* count = reinterpret_cast<uint32_t*>(k1);
    DFOK(hipMemsetAsync(count, 0, n * sizeof(uint32_t), stream), "hipMemsetAsync");
    hipLaunchKernelGGL(run_start_kernel, dim3(grid), dim3(BLOCK), 0, stream, head_code, run_incl, run_start, n);
    hipLaunchKernelGGL(count_kernel, dim3(grid), dim3(BLOCK), 0, stream, head_pair, run_incl, run_start, dup_limit, count, n);
    hipLaunchKernelGGL(scatter_kernel, dim3(grid), dim3(BLOCK), 0, stream, p0, run_incl, count, freq_out, n);
    DFOK(hipGetLastError(), "kernel launch");
    DFOK(hipStreamSynchronize(stream), "hipStreamSynchronize");
    return 0;
}

}  // namespace iskdf
