// micro_read.hip -- read-only HBM streaming ceiling on gfx950 with the scan kernel's access pattern
// (16 B per lane, 4 loads per lane per tile, grid-stride over 2048-row tiles), almost no ALU work.
// Build: hipcc --offload-arch=gfx950 -O3 -o micro_read micro_read.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, int U>
__global__ __launch_bounds__(256) void rd(const uint64_t* __restrict__ col, uint64_t n_tiles, uint32_t* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const char* tb = reinterpret_cast<const char*>(col) + t * (uint64_t)(256 * 16 * U);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const u32x4* p = reinterpret_cast<const u32x4*>(tb + threadIdx.x * 16u + u * 4096u);
            u32x4 v = NT ? __builtin_nontemporal_load(p) : *p;
            acc ^= v;
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;   // keep the loads alive
}

template <bool NT, int U>
void run(const char* name, const uint64_t* d, uint64_t bytes, int blocks) {
    uint32_t* out; hipMalloc(&out, 4);
    uint64_t n_tiles = bytes / (256 * 16 * U);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rd<NT, U><<<blocks, 256>>>(d, n_tiles, out);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) rd<NT, U><<<blocks, 256>>>(d, n_tiles, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks=%5d  %.1f us/pass  %.0f GB/s\n", name, blocks, ms / reps * 1e3, bytes * (double)reps / (ms * 1e-3) / 1e9);
    hipFree(out);
}

// cache-policy variants of the same stream, loads issued from asm (4 x 16 B per lane per tile, then one wait)
#define RD4(MOD)                                                                                              \
    asm volatile("global_load_dwordx4 %0, %4, off " MOD "\n\tglobal_load_dwordx4 %1, %5, off " MOD "\n\t"       \
                 "global_load_dwordx4 %2, %6, off " MOD "\n\tglobal_load_dwordx4 %3, %7, off " MOD "\n\t"       \
                 "s_waitcnt vmcnt(0)"                                                                         \
                 : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(e) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory")
template <int POLICY>
__global__ __launch_bounds__(256) void rd_policy(const uint64_t* __restrict__ col, uint64_t n_tiles, uint32_t* out) {
    u32x4 acc = {0, 0, 0, 0};
    for (uint64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const char* tb = reinterpret_cast<const char*>(col) + t * (uint64_t)(256 * 16 * 4);
        const char* p0 = tb + threadIdx.x * 16u;
        const char *p1 = p0 + 4096, *p2 = p0 + 8192, *p3 = p0 + 12288;
        u32x4 a, b, c, e;
        if constexpr (POLICY == 0) RD4("nt");
        else if constexpr (POLICY == 1) RD4("sc0 nt");
        else if constexpr (POLICY == 2) RD4("sc1 nt");
        else if constexpr (POLICY == 3) RD4("sc0 sc1 nt");
        else if constexpr (POLICY == 4) RD4("sc1");
        else RD4("sc0 sc1");
        acc ^= a ^ b ^ c ^ e;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
template <int POLICY>
void run_policy(const char* name, const uint64_t* d, uint64_t bytes, int blocks) {
    uint32_t* out; hipMalloc(&out, 4);
    uint64_t n_tiles = bytes / (256 * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rd_policy<POLICY><<<blocks, 256>>>(d, n_tiles, out);
    hipDeviceSynchronize();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) rd_policy<POLICY><<<blocks, 256>>>(d, n_tiles, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s blocks=%5d  %.1f us/pass  %.0f GB/s\n", name, blocks, ms / reps * 1e3, bytes * (double)reps / (ms * 1e-3) / 1e9);
    hipFree(out);
}

int main() {
    const uint64_t bytes = 800000000ull / 16384 * 16384;
    uint64_t* d; hipMalloc(&d, bytes);
    hipMemset(d, 0x5a, bytes);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount;
    for (int b : {4, 8, 16}) {
        run<true, 4>("read nt, 4x16B/lane/tile", d, bytes, cus * b);
        run<false, 4>("read plain, 4x16B/lane/tile", d, bytes, cus * b);
    }
    run<true, 8>("read nt, 8x16B/lane/tile", d, bytes, cus * 8);
    run<true, 2>("read nt, 2x16B/lane/tile", d, bytes, cus * 8);
    run_policy<0>("asm: nt", d, bytes, cus * 8);
    run_policy<1>("asm: sc0 nt", d, bytes, cus * 8);
    run_policy<2>("asm: sc1 nt", d, bytes, cus * 8);
    run_policy<3>("asm: sc0 sc1 nt", d, bytes, cus * 8);
    run_policy<4>("asm: sc1", d, bytes, cus * 8);
    run_policy<5>("asm: sc0 sc1", d, bytes, cus * 8);
    return 0;
}
