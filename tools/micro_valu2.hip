// micro_valu2.hip -- which VALU forms issue at 2 cycles per wave64 and which at 4 (gfx950)?
// tools/micro_valu.hip showed `v_lshrrev_b32 v, 1, v` at 2.4 cycles while v_xor/v_bcnt/v_min3/v_fma with an SGPR or a
// second VGPR source take 4.1-4.4.  This sweeps the operand kinds: inline constant / literal / SGPR / VGPR.
// Build: hipcc --offload-arch=gfx950 -O3 -o micro_valu2 micro_valu2.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define KERNEL(NAME, ASM, ...)                                                                         \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters, uint32_t s0, uint32_t s1) { \
        uint32_t a[8];                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 2654435761u + j * 40503u + blockIdx.x; \
        for (int i = 0; i < iters; ++i) {                                                              \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                            \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) { asm volatile(ASM : "+v"(a[j]) : __VA_ARGS__); } \
            }                                                                                          \
        }                                                                                              \
        uint32_t t = 0;                                                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) t += a[j];                                       \
        out[blockIdx.x * 256 + threadIdx.x] = t;                                                       \
    }

#define VN "v"(a[(j + 1) & 7])
#define VM "v"(a[(j + 2) & 7])
KERNEL(k_xor_inl, "v_xor_b32 %0, 1, %0", "s"(s0))
KERNEL(k_xor_lit, "v_xor_b32 %0, 0x12345678, %0", "s"(s0))
KERNEL(k_xor_sgpr, "v_xor_b32 %0, %1, %0", "s"(s0))
KERNEL(k_xor_vgpr, "v_xor_b32 %0, %1, %0", VN)
KERNEL(k_xor_self, "v_xor_b32 %0, %0, %0", "s"(s0))
KERNEL(k_and_inl, "v_and_b32 %0, 15, %0", "s"(s0))
KERNEL(k_and_vgpr, "v_and_b32 %0, %1, %0", VN)
KERNEL(k_add_inl, "v_add_u32 %0, 1, %0", "s"(s0))
KERNEL(k_add_sgpr, "v_add_u32 %0, %1, %0", "s"(s0))
KERNEL(k_add_vgpr, "v_add_u32 %0, %1, %0", VN)
KERNEL(k_shr_inl, "v_lshrrev_b32 %0, 1, %0", "s"(s0))
KERNEL(k_shr_sgpr, "v_lshrrev_b32 %0, %1, %0", "s"(s1))
KERNEL(k_shr_vgpr, "v_lshrrev_b32 %0, %1, %0", VN)
KERNEL(k_mov, "v_mov_b32 %0, %1", VN)
KERNEL(k_not, "v_not_b32 %0, %0", "s"(s0))
KERNEL(k_bcnt_inl, "v_bcnt_u32_b32 %0, %0, 0", "s"(s0))
KERNEL(k_bcnt_sgpr, "v_bcnt_u32_b32 %0, %0, %1", "s"(s1))
KERNEL(k_bcnt_vgpr, "v_bcnt_u32_b32 %0, %0, %1", VN)
KERNEL(k_bcnt_sv, "v_bcnt_u32_b32 %0, %1, %0", "s"(s1))
KERNEL(k_min_vgpr, "v_min_u32 %0, %1, %0", VN)
KERNEL(k_min_inl, "v_min_u32 %0, 7, %0", "s"(s0))
KERNEL(k_min3_vvv, "v_min3_i32 %0, %0, %1, %2", VN, VM)
KERNEL(k_min3_vvi, "v_min3_i32 %0, %0, %1, 5", VN)
KERNEL(k_min3_vii, "v_min3_i32 %0, %0, 3, 5", "s"(s0))
KERNEL(k_and_or, "v_and_or_b32 %0, %0, %1, %2", "s"(s0), VN)
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 1, %1", VN)
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 1, 31", "s"(s0))
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", VN, "s"(s0))
KERNEL(k_fma_vvv, "v_fma_f32 %0, %0, %1, %0", VN)
KERNEL(k_mul_inl, "v_mul_f32 %0, 2.0, %0", "s"(s0))
KERNEL(k_add_f_inl, "v_add_f32 %0, 1.0, %0", "s"(s0))
KERNEL(k_cmp_class, "v_cmp_lt_u32 vcc, %0, %1", VN)
KERNEL(k_xor_dpp, "v_xor_b32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "s"(s0))

typedef void (*kern_t)(uint32_t*, int, uint32_t, uint32_t);

void run(const char* name, kern_t k, int blocks_per_cu) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * blocks_per_cu;
    uint32_t* out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 10, 0x12345678u, 7u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, 0x12345678u, 7u);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * iters * 64.0;
    const double per_simd = wave_instr / (cus * 4.0);
    printf("%-12s blocks/CU=%d  %.3f ms  %.2f cycles(@2.4GHz) per wave-instr per SIMD\n", name, blocks_per_cu, ms, ms * 1e-3 * 2.4e9 / per_simd);
    (void)hipFree(out);
}

#define RUN(K) run(#K, K, b)
int main() {
    for (int b : {2, 8}) {
        RUN(k_xor_inl); RUN(k_xor_lit); RUN(k_xor_sgpr); RUN(k_xor_vgpr); RUN(k_xor_self);
        RUN(k_and_inl); RUN(k_and_vgpr); RUN(k_add_inl); RUN(k_add_sgpr); RUN(k_add_vgpr);
        RUN(k_shr_inl); RUN(k_shr_sgpr); RUN(k_shr_vgpr); RUN(k_mov); RUN(k_not);
        RUN(k_bcnt_inl); RUN(k_bcnt_sgpr); RUN(k_bcnt_vgpr); RUN(k_bcnt_sv);
        RUN(k_min_vgpr); RUN(k_min_inl); RUN(k_min3_vvv); RUN(k_min3_vvi); RUN(k_min3_vii);
        RUN(k_and_or); RUN(k_lshl_or); RUN(k_bfe); RUN(k_perm);
        RUN(k_fma_vvv); RUN(k_mul_inl); RUN(k_add_f_inl); RUN(k_cmp_class); RUN(k_xor_dpp);
    }
    return 0;
}
