// ssd_tiles.h -- LDS tile helpers shared by the MFMA scan kernels (ssd_mfma.hip, ssd_v5.hip): 16-byte accessors and the
// XOR swizzles that keep every access pattern of those kernels bank-conflict free on gfx950.
#pragma once
#include "omk_common.h"

namespace omk {

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ s16x8 as_s16x8(u32x4 v) { return __builtin_bit_cast(s16x8, v); }
__device__ __forceinline__ float bf_lo(uint32_t v) { return bf16_to_f32((uint16_t)(v & 0xffffu)); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return bf16_to_f32((uint16_t)(v >> 16)); }

// LDS layouts: unpadded rows, the 16-byte segment index XOR-ed with a function of the row (see ssd_mfma.hip, class A):
//   128-column tiles (256 B rows): seg ^ swzK(row);  64-column tiles (128 B rows): seg ^ swzU(row), swzU = swzK & 7.
__device__ __forceinline__ int swzK(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((((r >> 2) ^ (r >> 3)) & 1) << 1) | ((r >> 2) & 1); }
__device__ __forceinline__ int swzU(int r) { return swzK(r) & 7; }
__device__ __forceinline__ int kx3(int row, int col) { return row * 128 + ((((col >> 3) ^ swzK(row)) << 3) | (col & 7)); }
__device__ __forceinline__ int ux3(int row, int col) { return row * 64 + ((((col >> 3) ^ swzU(row)) << 3) | (col & 7)); }

}  // namespace omk
