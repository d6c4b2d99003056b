// micro_cvt_fp6.hip -- v_cvt_scalef32_2xpk16_fp6_f32 packs 32 f32 values of a lane into 32 FP6 values (6 VGPRs) in ONE
// instruction.  Could it replace the 16-instruction v_min3 fold of the matrix-core scan (sign test on the packed result)?
//   1. where does element i's SIGN bit land, and do small negatives keep it?   2. what does the instruction cost?
// Build: hipcc --offload-arch=gfx950 -O3 -o micro_cvt_fp6 micro_cvt_fp6.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float v16f __attribute__((ext_vector_type(16)));
typedef unsigned int v6u __attribute__((ext_vector_type(6)));

__global__ void probe(uint32_t* out, float neg, float pos, float scale) {
    // lane i: element i = neg, every other element = pos   (lane 32: all pos)
    v16f a, b;
    const int l = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = (l == i) ? neg : pos; b[i] = (l == 16 + i) ? neg : pos; }
    v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
#pragma unroll
    for (int i = 0; i < 6; ++i) out[l * 6 + i] = r[i];
}

template <int MODE>
__global__ __launch_bounds__(256) void rate(uint32_t* out, int iters, float s) {
    v16f a[2], b[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) { a[j][i] = threadIdx.x * 0.25f + i + j; b[j][i] = threadIdx.x * 0.5f - i - j; }
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (MODE == 0) {
                    v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a[j], b[j], s);
                    asm volatile("" : "+v"(r));
                    acc |= r[0];
                } else {
                    // the fold it would replace: 15 v_min3 + 1 v_min over the same 32 values
                    float m0 = fminf(fminf(a[j][0], a[j][1]), a[j][2]), m1 = fminf(fminf(b[j][0], b[j][1]), b[j][2]);
#pragma unroll
                    for (int i = 3; i < 15; i += 2) { m0 = fminf(fminf(m0, a[j][i]), a[j][i + 1]); m1 = fminf(fminf(m1, b[j][i]), b[j][i + 1]); }
                    m0 = fminf(fminf(m0, a[j][15]), b[j][15]);
                    float m = fminf(m0, m1);
                    asm volatile("" : "+v"(m));
                    acc |= __float_as_uint(m);
                }
                asm volatile("" : "+v"(a[j]), "+v"(b[j]));
            }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, int bpc) {
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * bpc, iters = 2000;
    uint32_t* out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    rate<MODE><<<blocks, 256>>>(out, 10, 1.0f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    rate<MODE><<<blocks, 256>>>(out, iters, 1.0f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)blocks * 4 * iters * 16.0 / (cus * 4.0);      // 16 folds of 32 values per iteration per wave
    printf("%-28s blocks/CU=%d  %.3f ms  %.1f cycles(@2.4GHz) per 32-value fold per SIMD\n", name, bpc, ms, ms * 1e-3 * 2.4e9 / per_simd);
    (void)hipFree(out);
}

int main() {
    uint32_t* d;
    (void)hipMalloc(&d, 64 * 6 * 4);
    std::vector<uint32_t> h(64 * 6);
    for (float neg : {-1.0f, -0.125f, -64.0f, 0.0f}) {
        probe<<<1, 64>>>(d, neg, 3.0f, 1.0f);
        (void)hipMemcpy(h.data(), d, 64 * 6 * 4, hipMemcpyDeviceToHost);
        printf("neg=%g pos=3 scale=1: baseline %08x %08x %08x %08x %08x %08x\n", neg, h[32 * 6], h[32 * 6 + 1], h[32 * 6 + 2], h[32 * 6 + 3], h[32 * 6 + 4], h[32 * 6 + 5]);
        for (int i = 0; i < 32; ++i) {
            printf("  element %2d:", i);
            for (int w = 0; w < 6; ++w) {
                const uint32_t x = h[i * 6 + w] ^ h[32 * 6 + w];
                for (int b = 0; b < 32; ++b) if (x >> b & 1) printf(" bit %d", w * 32 + b);
            }
            printf("\n");
            if (neg != -1.0f && i >= 2) { printf("  ...\n"); break; }
        }
    }
    for (int b : {1, 2, 4}) { run<0>("v_cvt_scalef32_2xpk16_fp6_f32", b); run<1>("15 v_min3 + 1 v_min", b); }
    return 0;
}
