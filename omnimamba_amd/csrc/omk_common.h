// omk_common.h -- host-side validation helpers and dtype-generic device load/store shared by all kernels.
#pragma once
#include <cstdarg>
#include <cstdio>

#include "../../include/omk.h"
#include "omk_platform.h"

namespace omk {

// ---- thread-local error text --------------------------------------------------------------------------
char* err_buf();
int fail(int code, const char* fmt, ...);
// which scan kernels the last omk_ssd_scan_fwd / omk_ssd_scan_bwd of this thread launched (omk_ssd_last_kernels: measurement tools tie a
// profile to the kernel that was timed)
void kernels_reset();
void kernels_note(const char* fmt, ...);
#define OMK_REQUIRE(cond, ...) \
  do { if (!(cond)) return ::omk::fail(OMK_EINVAL, __VA_ARGS__); } while (0)

inline bool present(const OmkTensor& t) { return t.data != nullptr; }
inline int64_t numel(const OmkTensor& t) {
  int64_t n = 1;
  for (int i = 0; i < t.ndim; i++) n *= t.shape[i];
  return n;
}
inline bool is_contig_last(const OmkTensor& t) { return t.ndim == 0 || t.shape[t.ndim - 1] == 1 || t.stride[t.ndim - 1] == 1; }
// densely packed, row-major
inline bool is_dense(const OmkTensor& t) {
  int64_t st = 1;
  for (int i = t.ndim - 1; i >= 0; i--) {
    if (t.shape[i] > 1 && t.stride[i] != st) return false;
    st *= t.shape[i];
  }
  return true;
}
inline size_t dtype_size(int dt) { return dt == OMK_F32 ? 4 : dt == OMK_U8 ? 1 : 2; }
inline bool aligned16(const OmkTensor& t) { return ((uintptr_t)t.data & 15) == 0; }
// every stride except the last is a multiple of `elems` (so 16-byte vector rows stay aligned)
inline bool strides_multiple_of(const OmkTensor& t, int64_t elems) {
  for (int i = 0; i + 1 < t.ndim; i++)
    if (t.shape[i] > 1 && (t.stride[i] % elems) != 0) return false;
  return true;
}
int finish_launch(const char* what);

// ---- storage types --------------------------------------------------------------------------------------
struct bf16_t { uint16_t v; };
struct f16_t { _Float16 v; };

template <class T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = OMK_F32; };
template <> struct dtype_of<bf16_t> { static constexpr int value = OMK_BF16; };
template <> struct dtype_of<f16_t> { static constexpr int value = OMK_F16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v.v); }
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v.v; }
template <class T> __device__ __forceinline__ T from_f32(float f);
template <> __device__ __forceinline__ float from_f32<float>(float f) { return f; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float f) { return bf16_t{f32_to_bf16(f)}; }
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float f) { return f16_t{(_Float16)f}; }

// runtime-dtype scalar access (parameters: a handful of loads per thread, never in the streaming loop)
__device__ __forceinline__ float load_rt(const void* p, int64_t i, int dt) {
  if (dt == OMK_F32) return ((const float*)p)[i];
  if (dt == OMK_BF16) return bf16_to_f32(((const uint16_t*)p)[i]);
  return (float)((const _Float16*)p)[i];
}
// The same without control flow -- one aligned dword read, the element picked with selects -- for kernels that are
// nothing but a chain of such loads (the decode-step state update): a load under a branch makes the compiler drain the
// memory pipeline at every use, which serialises the round trips.  Costs address arithmetic and registers, so the
// streaming kernels (conv1d forward: +13 % with this variant) keep the compact one.  (16-bit element: the aligned dword
// that contains it, always inside the allocation.)
__device__ __forceinline__ float load_rt_flat(const void* p, int64_t i, int dt) {
  const uintptr_t addr = (uintptr_t)p + ((uintptr_t)i << (dt == OMK_F32 ? 2 : 1));
  const uint32_t w = *reinterpret_cast<const uint32_t*>(addr & ~(uintptr_t)3);
  const uint32_t h = (addr & 2) ? (w >> 16) : (w & 0xffffu);
  const uint32_t bfb = h << 16;
  const uint16_t hb = (uint16_t)h;
  const float f32v = __builtin_bit_cast(float, w), bfv = __builtin_bit_cast(float, bfb), f16v = (float)__builtin_bit_cast(_Float16, hb);
  return dt == OMK_F32 ? f32v : (dt == OMK_BF16 ? bfv : f16v);
}
// The same in two halves: the request (no branch, no conversion -- several of them in a row stay in ONE basic block and fly together) and
// the conversion.  load_rt_flat's selects on the run-time dtype compile to scalar branches with the conversion inside, i.e. to a wait for
// the value right behind every request: four tied scalars of the decode state update were four dependent round trips (round 5).
struct RawElem { uint32_t w; uint32_t hi; };
__device__ __forceinline__ RawElem raw_rt_flat(const void* p, int64_t i, int dt) {
  const uintptr_t addr = (uintptr_t)p + ((uintptr_t)i << (dt == OMK_F32 ? 2 : 1));
  RawElem r;
  r.w = *reinterpret_cast<const uint32_t*>(addr & ~(uintptr_t)3);
  r.hi = (uint32_t)(addr & 2);
  return r;
}
__device__ __forceinline__ float cvt_rt_flat(RawElem r, int dt) {
  const uint32_t h = r.hi ? (r.w >> 16) : (r.w & 0xffffu);
  const uint32_t bfb = h << 16;
  const uint16_t hb = (uint16_t)h;
  const float f32v = __builtin_bit_cast(float, r.w), bfv = __builtin_bit_cast(float, bfb), f16v = (float)__builtin_bit_cast(_Float16, hb);
  return dt == OMK_F32 ? f32v : (dt == OMK_BF16 ? bfv : f16v);
}
__device__ __forceinline__ void store_rt(void* p, int64_t i, int dt, float v) {
  if (dt == OMK_F32) ((float*)p)[i] = v;
  else if (dt == OMK_BF16) ((uint16_t*)p)[i] = f32_to_bf16(v);
  else ((_Float16*)p)[i] = (_Float16)v;
}

// vector of VEC elements of storage type T <-> float[VEC]; VEC*sizeof(T) must be 4, 8 or 16 bytes
template <class T, int VEC> struct alignas((sizeof(T) * VEC) > 16 ? 16 : (sizeof(T) * VEC)) vec_t { T e[VEC]; };
template <class T, int VEC> __device__ __forceinline__ void load_vec(const T* p, float (&out)[VEC]) {
  vec_t<T, VEC> v = *reinterpret_cast<const vec_t<T, VEC>*>(p);
#pragma unroll
  for (int i = 0; i < VEC; i++) out[i] = to_f32(v.e[i]);
}
template <class T, int VEC> __device__ __forceinline__ void store_vec(T* p, const float (&in)[VEC]) {
  vec_t<T, VEC> v;
#pragma unroll
  for (int i = 0; i < VEC; i++) v.e[i] = from_f32<T>(in[i]);
  *reinterpret_cast<vec_t<T, VEC>*>(p) = v;
}

// dtype dispatch: calls FN<T>(...) for the storage type matching `dt`
#define OMK_DISPATCH_DTYPE(dt, T, ...)                                  \
  do {                                                                  \
    if ((dt) == OMK_F32) { using T = float; __VA_ARGS__; }              \
    else if ((dt) == OMK_BF16) { using T = ::omk::bf16_t; __VA_ARGS__; } \
    else if ((dt) == OMK_F16) { using T = ::omk::f16_t; __VA_ARGS__; }   \
    else return ::omk::fail(OMK_EINVAL, "bad dtype %d", (int)(dt));     \
  } while (0)

}  // namespace omk
