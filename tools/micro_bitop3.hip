// micro_bitop3.hip -- checks the truth-table convention of v_bitop3_b32 on gfx950:
// __builtin_amdgcn_bitop3_b32(a, b, c, 0xF6) must equal a | (b ^ c).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const uint32_t* in, uint32_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_bitop3_b32(in[3 * i], in[3 * i + 1], in[3 * i + 2], 0xF6);
}
int main() {
    const int n = 4096;
    uint32_t h[3 * n], o[n];
    uint64_t s = 88172645463325252ULL;
    for (int i = 0; i < 3 * n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s >> 16); }
    uint32_t *d, *r;
    hipMalloc(&d, sizeof h); hipMalloc(&r, sizeof o);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(d, r, n);
    hipMemcpy(o, r, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) if (o[i] != (h[3 * i] | (h[3 * i + 1] ^ h[3 * i + 2]))) ++bad;
    printf("bitop3 0xF6 == a | (b ^ c): %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
