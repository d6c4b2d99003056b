// mfma_scan.h -- host interface of mfma_scan.hip (the matrix-core form of the collect scan: FP4 products, f32 sums).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "scan_params.hip.h"

namespace isk {

constexpr int MFMA_MAX_LDS = 48 * 1024;   // a chunk needs <= 40 KB of dynamic LDS: within the default limit, nothing to configure

// query groups (32 queries each; an even number unless `pack`) one block keeps in LDS for W compared words
uint32_t mfma_groups_per_chunk(int W, uint32_t nq_pad, bool pack);
size_t mfma_lds_bytes(int W, uint32_t groups);
uint32_t mfma_waves_per_block();
// pack: 64-bit codes on mfma_pack_kernel (two row tiles per accumulator, packed f16 fold); the batch must not hold an all-zero query
uint32_t mfma_rows_per_wave_step(int W, bool pack);   // rows a wave takes per step (128 packed, else 64, or 32 when a build runs one tile per wave)
uint32_t mfma_blocks_per_cu(int W, uint32_t groups, bool pack);   // resident blocks per CU (LDS and register limits)
// grid = (blocks_x, chunks of groups * 32 queries); returns 0 or a hipError_t when the chunk would not fit (check
// hipGetLastError() for the launch itself, as with every other kernel)
int launch_mfma_scan(int W, int mode, bool pack, uint32_t blocks_x, uint32_t groups, hipStream_t st, const ScanParams& p);

}  // namespace isk
