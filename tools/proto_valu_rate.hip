// Issue rate of the fold's candidate instructions on gfx950: cycles per wave instruction, from s_memtime around an unrolled
// run of independent instructions (one wave per SIMD: the VALU is that wave's alone).
// build + run (GPU box): hipcc --offload-arch=gfx950 -O3 -o gpurun_out/proto_valu_rate tools/proto_valu_rate.hip && gpurun_out/proto_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define BODY(ins)                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n" : "=s"(t0));                            \
    for (int i = 0; i < 64; ++i) {                                                                                     \
        asm volatile(REP8(ins " %0, %0, %4, %5\n" ins " %1, %1, %4, %5\n" ins " %2, %2, %4, %5\n" ins " %3, %3, %4, %5\n") \
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));                                           \
    }                                                                                                                  \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n" : "=s"(t1));

#define BODY2(ins)                                                                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)\n s_memtime %0\n s_waitcnt lgkmcnt(0)\n" : "=s"(t0));                            \
    for (int i = 0; i < 64; ++i) {                                                                                     \
        asm volatile(REP8(ins " %0, %0, %4\n" ins " %1, %1, %5\n" ins " %2, %2, %4\n" ins " %3, %3, %5\n")               \
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));                                           \
    }                                                                                                                  \
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)\n" : "=s"(t1));

__global__ void rate(int which, unsigned long long* out, unsigned* sink) {
    unsigned a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, x = a * 7 + 0x3c003c00u, y = a * 5 + 0x38003800u;
    unsigned long long t0 = 0, t1 = 0;
    switch (which) {
        case 0: { BODY("v_pk_minimum3_f16") } break;
        case 1: { BODY("v_min3_f32") } break;
        case 2: { BODY("v_or3_b32") } break;
        case 3: { BODY2("v_pk_min_f16") } break;
        case 4: { BODY("v_min3_u32") } break;
        case 5: { BODY2("v_pk_min_u16") } break;
        case 6: { BODY("v_minimum3_f32") } break;
        case 7: { BODY("v_min3_f16") } break;
        case 8: { BODY("v_pk_maximum3_f16") } break;
        case 9: { BODY2("v_pk_max_f16") } break;
        case 10: { BODY("v_and_or_b32") } break;
        case 11: { BODY("v_min3_i16") } break;
    }
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    sink[blockIdx.x * 1024 + threadIdx.x] = a ^ b ^ c ^ d;
}

int main() {
    const char* names[] = {"v_pk_minimum3_f16", "v_min3_f32", "v_or3_b32", "v_pk_min_f16", "v_min3_u32", "v_pk_min_u16", "v_minimum3_f32", "v_min3_f16",
                           "v_pk_maximum3_f16", "v_pk_max_f16", "v_and_or_b32", "v_min3_i16"};
    unsigned long long* out;
    unsigned* sink;
    hipMalloc(&out, 8 * 1024 * 16);
    hipMalloc(&sink, 4 * 64 * 1024 * 16);
    for (int w = 0; w < 12; ++w) {
        for (int waves = 1; waves <= 4; ++waves) {          // waves per SIMD: blocks of 4 * waves waves, one block per CU
            hipLaunchKernelGGL(rate, dim3(256), dim3(256 * waves), 0, 0, w, out, sink);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(16);
            hipMemcpy(h.data(), out, 16 * 8, hipMemcpyDeviceToHost);
            unsigned long long lo = ~0ull, hi = 0;
            for (int i = 0; i < 4 * waves; ++i) { lo = h[i] < lo ? h[i] : lo; hi = h[i] > hi ? h[i] : hi; }
            // s_memtime ticks per instruction a wave issued: the fastest and the slowest wave of block 0 (the oldest wave has priority)
            printf("%-20s %d wave(s)/SIMD: %.3f .. %.3f ticks per instruction and wave\n", names[w], waves, (double)lo / (64.0 * 32), (double)hi / (64.0 * 32));
        }
    }
    return 0;
}
