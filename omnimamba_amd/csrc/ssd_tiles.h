// ssd_tiles.h -- LDS tile helpers shared by the MFMA scan kernels (ssd_mfma.hip, ssd_v6.hip, ssd_cp.hip): 16-byte accessors and the
// XOR swizzles that keep every access pattern of those kernels bank-conflict free on gfx950.
#pragma once
#include "omk_common.h"

namespace omk {

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ s16x8 as_s16x8(u32x4 v) { return __builtin_bit_cast(s16x8, v); }
__device__ __forceinline__ float bf_lo(uint32_t v) { return bf16_to_f32((uint16_t)(v & 0xffffu)); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return bf16_to_f32((uint16_t)(v >> 16)); }

// ---- buffer resources: a global range addressed as resource (SGPRs) + 32-bit lane offset + 32-bit scalar offset.  The address
// is formed by the memory pipeline (no 64-bit VALU adds per load: the scan kernels are issue bound) and the range check comes
// with it: a load that starts at or behind `nbytes` returns 0, a store there is dropped -- the rows behind the end of a ragged
// last chunk need neither clamped addresses nor selects.
struct BufRes {
#ifdef OMK_EMU
  const char* base; uint32_t nbytes;
#else
  __amdgpu_buffer_rsrc_t r;
#endif
};
__device__ __forceinline__ BufRes make_buf(const void* p, uint32_t nbytes) {
  BufRes b;
#ifdef OMK_EMU
  b.base = (const char*)p; b.nbytes = nbytes;
#else
  b.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)nbytes, 0x00020000);   // raw buffer, 32-bit data format
#endif
  return b;
}
__device__ __forceinline__ u32x4 buf_ld16(const BufRes& b, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;   // (the UNWRAPPED sum: a kernel must not count on 32-bit wrap-around of lane + scalar offset)
  if (o >= b.nbytes) return u32x4{0u, 0u, 0u, 0u};
  return *reinterpret_cast<const u32x4*>(b.base + o);
#else
  return __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)voff, (int)soff, 0);
#endif
}
__device__ __forceinline__ float buf_ld_f32(const BufRes& b, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;   // (the UNWRAPPED sum: a kernel must not count on 32-bit wrap-around of lane + scalar offset)
  if (o >= b.nbytes) return 0.f;
  return *reinterpret_cast<const float*>(b.base + o);
#else
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, (int)voff, (int)soff, 0));
#endif
}
__device__ __forceinline__ void buf_st_f32(const BufRes& b, float v, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;
  if (o < b.nbytes) *reinterpret_cast<float*>(const_cast<char*>(b.base) + o) = v;
#else
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned int, v), b.r, (int)voff, (int)soff, 0);
#endif
}
// 16-bit elements (bf16 / f16 storage): the raw bits
__device__ __forceinline__ uint16_t buf_ld_u16(const BufRes& b, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;   // (the UNWRAPPED sum: a kernel must not count on 32-bit wrap-around of lane + scalar offset)
  if (o >= b.nbytes) return 0;
  return *reinterpret_cast<const uint16_t*>(b.base + o);
#else
  return (uint16_t)__builtin_amdgcn_raw_buffer_load_b16(b.r, (int)voff, (int)soff, 0);
#endif
}
__device__ __forceinline__ void buf_st_u16(const BufRes& b, uint16_t v, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;
  if (o < b.nbytes) *reinterpret_cast<uint16_t*>(const_cast<char*>(b.base) + o) = v;
#else
  __builtin_amdgcn_raw_buffer_store_b16(v, b.r, (int)voff, (int)soff, 0);
#endif
}
// raw bits of one element of storage type T (the conversion stays with the consumer: a load is not waited for where it is issued)
template <class T> __device__ __forceinline__ uint32_t buf_ld_raw(const BufRes& b, uint32_t voff, uint32_t soff) {
  if (sizeof(T) == 4) return __builtin_bit_cast(uint32_t, buf_ld_f32(b, voff, soff));
  return buf_ld_u16(b, voff, soff);
}
template <class T> __device__ __forceinline__ float raw_to_f32(uint32_t r);
template <> __device__ __forceinline__ float raw_to_f32<float>(uint32_t r) { return __builtin_bit_cast(float, r); }
template <> __device__ __forceinline__ float raw_to_f32<bf16_t>(uint32_t r) { return __builtin_bit_cast(float, r << 16); }
template <> __device__ __forceinline__ float raw_to_f32<f16_t>(uint32_t r) { return (float)__builtin_bit_cast(_Float16, (uint16_t)r); }
// one element of storage type T as fp32 / fp32 to one element of T
template <class T> __device__ __forceinline__ float buf_ld_t(const BufRes& b, uint32_t voff, uint32_t soff);
template <> __device__ __forceinline__ float buf_ld_t<float>(const BufRes& b, uint32_t voff, uint32_t soff) { return buf_ld_f32(b, voff, soff); }
template <> __device__ __forceinline__ float buf_ld_t<bf16_t>(const BufRes& b, uint32_t voff, uint32_t soff) { return bf16_to_f32(buf_ld_u16(b, voff, soff)); }
template <> __device__ __forceinline__ float buf_ld_t<f16_t>(const BufRes& b, uint32_t voff, uint32_t soff) {
  const uint16_t h = buf_ld_u16(b, voff, soff);
  return (float)__builtin_bit_cast(_Float16, h);
}
template <class T> __device__ __forceinline__ void buf_st_t(const BufRes& b, float v, uint32_t voff, uint32_t soff);
template <> __device__ __forceinline__ void buf_st_t<float>(const BufRes& b, float v, uint32_t voff, uint32_t soff) { buf_st_f32(b, v, voff, soff); }
template <> __device__ __forceinline__ void buf_st_t<bf16_t>(const BufRes& b, float v, uint32_t voff, uint32_t soff) { buf_st_u16(b, f32_to_bf16(v), voff, soff); }
template <> __device__ __forceinline__ void buf_st_t<f16_t>(const BufRes& b, float v, uint32_t voff, uint32_t soff) {
  const _Float16 h = (_Float16)v;
  buf_st_u16(b, __builtin_bit_cast(uint16_t, h), voff, soff);
}
__device__ __forceinline__ u32x2 buf_ld8(const BufRes& b, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;   // (the UNWRAPPED sum: a kernel must not count on 32-bit wrap-around of lane + scalar offset)
  if (o >= b.nbytes) return u32x2{0u, 0u};
  return *reinterpret_cast<const u32x2*>(b.base + o);
#else
  return __builtin_amdgcn_raw_buffer_load_b64(b.r, (int)voff, (int)soff, 0);
#endif
}
__device__ __forceinline__ void buf_st8(const BufRes& b, u32x2 v, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;
  if (o < b.nbytes) *reinterpret_cast<u32x2*>(const_cast<char*>(b.base) + o) = v;
#else
  __builtin_amdgcn_raw_buffer_store_b64(v, b.r, (int)voff, (int)soff, 0);
#endif
}
__device__ __forceinline__ void buf_st16(const BufRes& b, u32x4 v, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;
  if (o < b.nbytes) *reinterpret_cast<u32x4*>(const_cast<char*>(b.base) + o) = v;
#else
  __builtin_amdgcn_raw_buffer_store_b128(v, b.r, (int)voff, (int)soff, 0);
#endif
}

// ---- LDS-DMA (buffer_load ... lds): the 64 lanes of a wave fetch 16 (4) bytes each from the resource and the memory pipeline writes
// them to LDS at lds_base + 16 (4) * lane -- no staging registers, no ds_write pass.  lds_base must be wave-uniform; the request
// counts on vmcnt like any load, and NOTHING orders a later ds_read behind it except the issuing wave's own s_waitcnt vmcnt (plus a
// barrier for the other waves).  Rows behind the end of the buffer arrive as zeros.
__device__ __forceinline__ void buf_ld16_lds(const BufRes& b, void* lds_base, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;
  u32x4 v = {0u, 0u, 0u, 0u};
  if (o < b.nbytes) v = *reinterpret_cast<const u32x4*>(b.base + o);
  *reinterpret_cast<u32x4*>((char*)lds_base + 16 * lane_id()) = v;
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_base, 16, (int)voff, (int)soff, 0, 0);
#endif
}
__device__ __forceinline__ void buf_ld4_lds(const BufRes& b, void* lds_base, uint32_t voff, uint32_t soff) {
#ifdef OMK_EMU
  const uint64_t o = (uint64_t)voff + soff;
  uint32_t v = 0u;
  if (o < b.nbytes) v = *reinterpret_cast<const uint32_t*>(b.base + o);
  *reinterpret_cast<uint32_t*>((char*)lds_base + 4 * lane_id()) = v;
#else
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b.r, (__attribute__((address_space(3))) void*)lds_base, 4, (int)voff, (int)soff, 0, 0);
#endif
}
// the wave's vector-memory operations except the N youngest have completed (in-order completion, as the compiler's own wait counts assume)
#ifdef OMK_EMU
#define OMK_VMCNT(n) do { } while (0)
#else
#define OMK_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#endif
// workgroup barrier that does NOT drain vector memory (LDS-DMA requests stay in flight across it): the wave's own LDS traffic is
// complete (lgkmcnt(0)), then s_barrier.  __syncthreads() would wait for vmcnt(0) whenever a DMA is pending.
__device__ __forceinline__ void block_sync_lds() {
#ifdef OMK_EMU
  block_sync();
#else
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#endif
}

// LDS layouts: unpadded rows, the 16-byte segment index XOR-ed with a function of the row (see ssd_mfma.hip, class A):
//   128-column tiles (256 B rows): seg ^ swzK(row);  64-column tiles (128 B rows): seg ^ swzU(row), swzU = swzK & 7.
__device__ __forceinline__ int swzK(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((((r >> 2) ^ (r >> 3)) & 1) << 1) | ((r >> 2) & 1); }
__device__ __forceinline__ int swzU(int r) { return swzK(r) & 7; }
__device__ __forceinline__ int kx3(int row, int col) { return row * 128 + ((((col >> 3) ^ swzK(row)) << 3) | (col & 7)); }
// Window-state images in memory (GScan::dump, 16 KB per head and window): the 16-byte segments of the [u][k] kx3 image in the ORDER THE
// COLUMN-SLICE SCAN HOLDS THEM -- segment q = (4 w + i) 64 + lane is k-segment 4 i + (lane >> 4) of row u = 16 w + (lane & 15) -- so
// that every store instruction of the scan writes 1 KB of consecutive bytes (as 16 rows x 64 bytes the images cost the training
// forward 35 us of 237: tools/ubench/store_cost.hip).  img_off: element offset of segment q inside the kx3 image in LDS.
__device__ __forceinline__ int img_off(int q) {
  const int su = 16 * (q >> 8) + (q & 15);
  return su * 128 + (((4 * ((q >> 6) & 3) + ((q >> 4) & 3)) ^ swzK(su)) << 3);
}
__device__ __forceinline__ int ux3(int row, int col) { return row * 64 + ((((col >> 3) ^ swzU(row)) << 3) | (col & 7)); }

// 32x32x16 operand fragment out of a swizzled row-major [contraction][col] tile (K: 128 columns, U: 64 columns):
// rows r0 + 8 h32 + 4 m + {0..3}, column c0 + (lane & 31)
template <bool KT>
__device__ __forceinline__ s16x8 tr_frag3(const uint16_t* tile, int r0, int c0, int lane) {
  const int t16 = lane & 15, g16 = lane >> 4, h32 = lane >> 5;
  const int row = r0 + 8 * h32 + (t16 >> 2), col = c0 + 16 * (g16 & 1) + 4 * (t16 & 3);
  const s16x4 a = lds_read_tr16_b64(tile + (KT ? kx3(row, col) : ux3(row, col)));
  const s16x4 b = lds_read_tr16_b64(tile + (KT ? kx3(row + 4, col) : ux3(row + 4, col)));
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

}  // namespace omk
