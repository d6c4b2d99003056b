// micro_valu3.hip -- do the 2-cycle forms of micro_valu2 (v_xor with VGPR sources, ...) keep their rate when they are
// MIXED with 4-cycle instructions (v_bcnt, v_min3), as in the scan kernel's inner loop?  Sequences per "pair of rows":
//   4 x v_xor, 4 x v_bcnt, 1 x v_min3  (the XOR + popcount kernel's work for two rows against one 64-bit query)
// with the query words in VGPRs or in SGPRs, and a few reorderings.
// Build: hipcc --offload-arch=gfx950 -O3 -o micro_valu3 micro_valu3.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// one "row pair" step on register set j: rows (r0l, r0h, r1l, r1h), running min m, query (ql, qh), bias b
#define STEP(QL, QH, ORDER)                                                                                          \
    if (ORDER == 0) {                                                                                               \
        asm volatile("v_xor_b32 %0, %4, %6\n\tv_xor_b32 %1, %5, %7\n\tv_xor_b32 %2, %4, %8\n\tv_xor_b32 %3, %5, %9\n\t"  \
                     "v_bcnt_u32_b32 %0, %0, %10\n\tv_bcnt_u32_b32 %2, %2, %10\n\t"                                  \
                     "v_bcnt_u32_b32 %0, %1, %0\n\tv_bcnt_u32_b32 %2, %3, %2\n\t"                                    \
                     "v_min3_u32 %11, %11, %0, %2"                                                                  \
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)                                                   \
                     : QL, QH, "v"(r[j][0]), "v"(r[j][1]), "v"(r[j][2]), "v"(r[j][3]), "s"(s1), "v"(m[j]));           \
    } else {                                                                                                        \
        asm volatile("v_xor_b32 %0, %4, %6\n\tv_bcnt_u32_b32 %0, %0, %10\n\tv_xor_b32 %1, %5, %7\n\tv_bcnt_u32_b32 %0, %1, %0\n\t" \
                     "v_xor_b32 %2, %4, %8\n\tv_bcnt_u32_b32 %2, %2, %10\n\tv_xor_b32 %3, %5, %9\n\tv_bcnt_u32_b32 %2, %3, %2\n\t" \
                     "v_min3_u32 %11, %11, %0, %2"                                                                  \
                     : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)                                                   \
                     : QL, QH, "v"(r[j][0]), "v"(r[j][1]), "v"(r[j][2]), "v"(r[j][3]), "s"(s1), "v"(m[j]));           \
    }

template <int QKIND, int ORDER>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters, uint32_t s0, uint32_t s1) {
    uint32_t r[4][4], m[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        m[j] = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[j][i] = threadIdx.x * 2654435761u + (j * 4 + i) * 40503u + blockIdx.x;
    }
    uint32_t qlv = s0 ^ 0x55, qhv = s0 ^ 0xAA;
    asm volatile("" : "+v"(qlv), "+v"(qhv));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                uint32_t t0, t1, t2, t3;
                if (QKIND == 0) { STEP("v"(qlv), "v"(qhv), ORDER) } else { STEP("s"(s0), "s"(s1), ORDER) }
                asm volatile("" : "+v"(m[j]));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = m[0] + m[1] + m[2] + m[3];
}

template <int QKIND, int ORDER>
void run(const char* name, int blocks_per_cu) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * blocks_per_cu;
    uint32_t* out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<QKIND, ORDER><<<blocks, 256>>>(out, 10, 0x12345678u, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<QKIND, ORDER><<<blocks, 256>>>(out, iters, 0x12345678u, 7u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_simd = (double)blocks * 4 * iters * 16.0 / (cus * 4.0);   // 16 row-pair steps per iteration per wave
    printf("%-34s blocks/CU=%d  %.3f ms  %.2f cycles(@2.4GHz) per 9-instruction row-pair step per SIMD (36.0 if all 4, 28.0 if xor at 2)\n", name,
           blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / steps_per_simd);
    (void)hipFree(out);
}

int main() {
    for (int b : {2, 6, 8}) {
        run<0, 0>("query in VGPRs, grouped", b);
        run<0, 1>("query in VGPRs, interleaved", b);
        run<1, 0>("query in SGPRs, grouped", b);
        run<1, 1>("query in SGPRs, interleaved", b);
    }
    return 0;
}
