// omk_platform.h -- the one place where the kernels touch the platform.
//   default build : hipcc --offload-arch=gfx950 (CDNA4 only; no other GPU target is supported)
//   -DOMK_EMU     : host build against tests/emu/hip_emu.h, used only by the CPU test-suite
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifdef OMK_EMU
#include "hip_emu.h"
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
using emu::dim3;
#define threadIdx (emu::cur().tIdx)
#define blockIdx (emu::blk()->bIdx)
#define blockDim (emu::blk()->bDim)
#define gridDim (emu::blk()->gDim)
typedef void* hipStream_t;
#define OMK_DYN_SMEM(name) char* name = emu::blk()->dyn_smem
#define OMK_LAUNCH(kern, grid, block, smem, stream, ...) \
  emu::launch(grid, block, smem, [&] { kern(__VA_ARGS__); })
#define OMK_LAUNCH_CHECK() 0
#define OMK_SET_MAX_DYN_SMEM(kern, bytes) 0
#else
#include <hip/hip_runtime.h>
#define OMK_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define OMK_LAUNCH(kern, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kern, grid, block, smem, (hipStream_t)(stream), __VA_ARGS__)
#define OMK_LAUNCH_CHECK() (hipGetLastError() != hipSuccess)
#define OMK_SET_MAX_DYN_SMEM(kern, bytes) \
  (hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)) != hipSuccess)
#endif

namespace omk {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---- bf16 <-> f32 (round-to-nearest-even; identical bits in both builds) ---------------------------------
__device__ __forceinline__ float bf16_to_f32(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
#ifdef OMK_EMU
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
#else
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 on gfx950 (RNE)
  uint16_t r;
  memcpy(&r, &b, 2);
  return r;
#endif
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#ifdef OMK_EMU
  return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
#else
  typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
  const pk_f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pk_bf16x2));   // one v_cvt_pk_bf16_f32 (RNE)
#endif
}

// c + a.lo * b.lo + a.hi * b.hi on packed bf16 pairs (v_dot2c_f32_bf16 on gfx950)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float c) {
#ifdef OMK_EMU
  return c + bf16_to_f32((uint16_t)(a & 0xffffu)) * bf16_to_f32((uint16_t)(b & 0xffffu)) + bf16_to_f32((uint16_t)(a >> 16)) * bf16_to_f32((uint16_t)(b >> 16));
#else
  typedef __bf16 dot_bf16x2 __attribute__((ext_vector_type(2)));
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(dot_bf16x2, a), __builtin_bit_cast(dot_bf16x2, b), c, false);
#endif
}

// ---- wave-level primitives (wave = 64 lanes on CDNA) --------------------------------------------------
#ifdef OMK_EMU
__device__ __forceinline__ int lane_id() { return emu::lane_id(); }
template <class T> __device__ __forceinline__ T shfl(T v, int src) { return emu::shfl_generic(v, src); }
template <class T> __device__ __forceinline__ T shfl_xor(T v, int m) { return emu::shfl_generic(v, emu::lane_id() ^ m); }
template <class T> __device__ __forceinline__ T shfl_up(T v, int d) {
  int l = emu::lane_id();
  return emu::shfl_generic(v, l - d >= 0 ? l - d : l);
}
template <class T> __device__ __forceinline__ T shfl_down(T v, int d) {
  int l = emu::lane_id();
  return emu::shfl_generic(v, l + d < 64 ? l + d : l);
}
__device__ __forceinline__ void block_sync() { emu::block_barrier(); }
__device__ __forceinline__ bool ballot_any(bool pred) { return emu::ballot(pred ? 1 : 0) != 0; }   // true in every lane when any lane's pred is
__device__ __forceinline__ f32x16 mfma32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
  emu::Wave& w = emu::cur_wave();
  int l = emu::lane_id();
  for (int e = 0; e < 8; e++) { w.a16[l][e] = (uint16_t)a[e]; w.b16[l][e] = (uint16_t)b[e]; }
  for (int r = 0; r < 16; r++) w.c32[l][r] = c[r];
  emu::wave_collective(emu::op_mfma32);
  f32x16 d;
  for (int r = 0; r < 16; r++) d[r] = w.d32[l][r];
  return d;
}
__device__ __forceinline__ f32x4 mfma16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) {
  emu::Wave& w = emu::cur_wave();
  int l = emu::lane_id();
  for (int e = 0; e < 8; e++) { w.a16[l][e] = (uint16_t)a[e]; w.b16[l][e] = (uint16_t)b[e]; }
  for (int r = 0; r < 4; r++) w.c32[l][r] = c[r];
  emu::wave_collective(emu::op_mfma16);
  f32x4 d;
  for (int r = 0; r < 4; r++) d[r] = w.d32[l][r];
  return d;
}
__device__ __forceinline__ f32x4 mfma16x16x4_f32(float a, float b, f32x4 c) {
  emu::Wave& w = emu::cur_wave();
  int l = emu::lane_id();
  for (int r = 0; r < 4; r++) w.c32[l][r] = c[r];
  w.c32[l][4] = a; w.c32[l][5] = b;
  emu::wave_collective(emu::op_mfma16_f32);
  f32x4 d;
  for (int r = 0; r < 4; r++) d[r] = w.d32[l][r];
  return d;
}
// ds_read_b64_tr_b16: lane passes the address of ITS 8 bytes; gets column (lane&15) of the 4x16 block that its
// 16-lane group fetched (rows = lanes t/4, columns = 4*(t%4)+e).
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const uint16_t* p) {
  emu::Wave& w = emu::cur_wave();
  int l = emu::lane_id();
  for (int e = 0; e < 4; e++) w.tr_in[l][e] = p[e];
  emu::wave_collective(emu::op_tr16);
  s16x4 r;
  for (int e = 0; e < 4; e++) r[e] = (short)w.tr_out[l][e];
  return r;
}
__device__ __forceinline__ float atomic_add_f32(float* p, float v) { float o = *p; *p = o + v; return o; }
__device__ __forceinline__ void lds_add_f32(float* p, float v) { *p += v; }
#else
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
template <class T> __device__ __forceinline__ T shfl(T v, int src) { return __shfl(v, src, 64); }
template <class T> __device__ __forceinline__ T shfl_xor(T v, int m) { return __shfl_xor(v, m, 64); }
template <class T> __device__ __forceinline__ T shfl_up(T v, int d) { return __shfl_up(v, d, 64); }
template <class T> __device__ __forceinline__ T shfl_down(T v, int d) { return __shfl_down(v, d, 64); }
__device__ __forceinline__ void block_sync() { __syncthreads(); }
__device__ __forceinline__ bool ballot_any(bool pred) { return __builtin_amdgcn_ballot_w64(pred) != 0; }   // wave-uniform
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// v_mfma_f32_16x16x4_f32: fp32 operands (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15]), 32 cycles per SIMD: the
// fp32 matrix rate of this chip equals its fp32 VALU rate -- what the instruction buys is 1024 MACs per issue slot instead of 64
__device__ __forceinline__ f32x4 mfma16x16x4_f32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ s16x4 lds_read_tr16_b64(const uint16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}
__device__ __forceinline__ float atomic_add_f32(float* p, float v) { return unsafeAtomicAdd(p, v); }
// fire-and-forget LDS float add (ds_add_f32): result unused so no return value is requested
__device__ __forceinline__ void lds_add_f32(float* p, float v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif

// Order the LDS traffic of ONE wave: what its lanes wrote before is visible to the reads of any of its lanes behind.  The LDS executes a
// wave's instructions in order, so only the compiler has to be told (a fence at wavefront scope costs no instruction; __syncthreads of a
// one-wave workgroup would also wait for every global load and store in flight -- vmcnt(0) -- and end any prefetch).
#ifdef OMK_EMU
__device__ __forceinline__ void wave_lds_sync() { (void)emu::ballot(1); }   // the lanes are fibers: meet
#else
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#endif

template <class T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += shfl_xor(v, m);
  return v;
}

// ---- math ---------------------------------------------------------------------------------------------
// hardware transcendental forms (v_exp_f32 / v_rcp_f32, ~1 ulp) for the inner loops of the MFMA kernels
#ifdef OMK_EMU
__device__ __forceinline__ float exp2_fast(float x) { return exp2f(x); }
__device__ __forceinline__ float log2_fast(float x) { return log2f(x); }   // log2(0) = -inf on both builds
__device__ __forceinline__ float rcp_fast(float x) { return 1.f / x; }
__device__ __forceinline__ int uniform_i(int v) { return v; }
// inclusive prefix sum over the 64 lanes of a wave / broadcast of one lane
__device__ __forceinline__ float wave_incl_scan_add(float v) {
  for (int off = 1; off < 64; off <<= 1) {
    float o = shfl_up(v, off);
    if (lane_id() >= off) v += o;
  }
  return v;
}
__device__ __forceinline__ float wave_read_lane(float v, int l) { return shfl(v, l); }
// lane N of the lane's own row of 16 / lane 15 of the row of 16 in front of an odd row, lane 15 of its own row for an even row
template <int N> __device__ __forceinline__ float wave_row_bcast(float v) { return shfl(v, (lane_id() & ~15) + N); }
__device__ __forceinline__ float wave_pair_boundary(float v) { const int l = lane_id(); return shfl(v, (l & 16) ? (l & ~15) - 1 : (l & ~15) + 15); }
// v_permlane16_swap: the odd rows of 16 lanes of x change places with the even rows of y (x: lanes 16..31 <-> y: lanes 0..15, x: 48..63 <-> y: 32..47)
__device__ __forceinline__ void wave_swap16(uint32_t& x, uint32_t& y) {
  const int l = lane_id();
  const uint32_t xo = shfl(x, l ^ 16), yo = shfl(y, l ^ 16);
  if (l & 16) x = yo; else y = xo;
}
#else
__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float log2_fast(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32 = log2
__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }   // wave-uniform value -> SGPR
// DPP Hillis-Steele inside rows of 16 lanes, then row_bcast:15 / row_bcast:31 carry the row totals (no LDS traffic)
__device__ __forceinline__ float wave_incl_scan_add(float v) {
#define OMK_DPP_ADD(ctrl, rowmask) \
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rowmask, 0xf, false))
  OMK_DPP_ADD(0x111, 0xf);   // row_shr:1
  OMK_DPP_ADD(0x112, 0xf);   // row_shr:2
  OMK_DPP_ADD(0x114, 0xf);   // row_shr:4
  OMK_DPP_ADD(0x118, 0xf);   // row_shr:8
  OMK_DPP_ADD(0x142, 0xa);   // row_bcast:15 -> rows 1, 3
  OMK_DPP_ADD(0x143, 0xc);   // row_bcast:31 -> rows 2, 3
#undef OMK_DPP_ADD
  return v;
}
__device__ __forceinline__ float wave_read_lane(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
// lane N of the lane's own row of 16 (DPP row_newbcast: no LDS round trip, unlike ds_bpermute behind __shfl)
template <int N> __device__ __forceinline__ float wave_row_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + N, 0xf, 0xf, false));
}
// odd rows of 16: lane 15 of the row in front (row_bcast:15); even rows: lane 15 of their own row (row_newbcast:15)
__device__ __forceinline__ void wave_swap16(uint32_t& x, uint32_t& y) {
  const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  x = r[0]; y = r[1];
}
__device__ __forceinline__ float wave_pair_boundary(float v) {
  const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x15f, 0x5, 0xf, false);
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, false));
}
#endif
// Phase markers for tools/isa_phases.py (developer builds with -DOMK_ISA_MARKS only): a comment line in the ISA
#if defined(OMK_ISA_MARKS) && !defined(OMK_EMU)
#define OMK_ISA_MARK(name) asm volatile("; @@PHASE " name ::: "memory")
#else
#define OMK_ISA_MARK(name) do { } while (0)
#endif
// Instruction-scheduling fence: nothing moves across it
#ifdef OMK_EMU
#define OMK_SCHED_FENCE() do { } while (0)
#else
#define OMK_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
// Register budget hint: at least N waves per SIMD (the compiler caps VGPRs at 512 / N)
#ifdef OMK_EMU
#define OMK_WAVES_PER_EU(n)
#else
#define OMK_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif
// A lambda that must be inlined at every call site (a large body called twice is otherwise left as a function: everything it captures by
// reference -- accumulator tiles, staging registers -- then lives in scratch memory)
#ifdef OMK_EMU
#define OMK_ALWAYS_INLINE_LAMBDA
#else
#define OMK_ALWAYS_INLINE_LAMBDA __attribute__((always_inline))
#endif
// Keep a value alive without using it (ablation builds)
#ifdef OMK_EMU
#define OMK_KEEP(x) do { } while (0)
#else
#define OMK_KEEP(x) asm volatile("" :: "v"(x))
#endif
// Make a lane value opaque to the optimiser at this point: address arithmetic derived from it afterwards stays inside
// the loop instead of being hoisted into (many) loop-invariant registers.
#ifdef OMK_EMU
#define OMK_OPAQUE(x) do { } while (0)
#else
#define OMK_OPAQUE(x) asm volatile("" : "+v"(x))
#endif
// The same for a wave-uniform value (stays in an SGPR): a running scalar offset is advanced by one add per use instead of being
// re-derived as base + k * step with every k * step hoisted into a register of its own.
#ifdef OMK_EMU
#define OMK_OPAQUE_S(x) do { } while (0)
#else
#define OMK_OPAQUE_S(x) asm volatile("" : "+s"(x))
#endif
// Static issue priority of the wave (0 .. 3) from here on.
#ifdef OMK_EMU
#define OMK_SET_PRIO(n) do { } while (0)
#else
#define OMK_SET_PRIO(n) __builtin_amdgcn_s_setprio(n)
#endif
// Every vector-memory operation issued so far has completed, and the COMPILER knows it (a real s_waitcnt, not inline asm).  In
// front of a loop whose back edge carries counted loads behind stores: the wait-count pass merges the scoreboards of the two
// edges of the loop header, and prologue loads still pending there turn the loop's exact vmcnt(N) into vmcnt(0 / 1) -- a
// wait for the previous iteration's STORES.
#ifdef OMK_EMU
#define OMK_VM_DRAIN() do { } while (0)
#else
#define OMK_VM_DRAIN() __builtin_amdgcn_s_waitcnt(0x0F70)   // vmcnt(0), expcnt / lgkmcnt untouched
#endif
// Sum of each of 4 per-lane values over the 16 lanes of a DPP row (lanes 16 k .. 16 k + 15); every lane of the row gets the
// totals.  GPU: four row rotations per value with the DPP modifier on the add (no LDS crossbar round trips).
__device__ __forceinline__ void row16_sum4(float (&v)[4]) {
#ifdef OMK_EMU
#pragma unroll
  for (int i = 0; i < 4; i++) { v[i] += shfl_xor(v[i], 8); v[i] += shfl_xor(v[i], 4); v[i] += shfl_xor(v[i], 2); v[i] += shfl_xor(v[i], 1); }
#else
#define OMK_DPP_ROR(n) \
  _Pragma("unroll") for (int i = 0; i < 4; i++) \
    v[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v[i]), 0x120 + (n), 0xf, 0xf, false))
  OMK_DPP_ROR(8); OMK_DPP_ROR(4); OMK_DPP_ROR(2); OMK_DPP_ROR(1);
#undef OMK_DPP_ROR
#endif
}
// float sums take the DPP path (8 VALU ops instead of six ds_bpermute round trips); every lane gets the total
__device__ __forceinline__ float wave_sum(float v) { return wave_read_lane(wave_incl_scan_add(v), 63); }
#ifdef OMK_EMU
__device__ __forceinline__ uint64_t clock64_() { return 0; }
#else
__device__ __forceinline__ uint64_t clock64_() { return __builtin_readcyclecounter(); }
#endif
// a * b + c with ONE rounding, in every build and at every call site: `acc += a * b` under -ffp-contract=fast is fused where the compiler
// likes it and split where its vectoriser prefers a packed multiply + add (conv1d_fwd_cl8_kernel: tokens 5 .. 7 of every group of eight came
// out unfused, 1e-5 of the outputs one bf16 ulp away from the other forward kernels).  Kernels whose results must agree bit for bit say fma.
__device__ __forceinline__ float fma_f32(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// the same on a pair of fp32 values: one v_pk_fma_f32
__device__ __forceinline__ f32x2 fma_f32x2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float sigmoid_fast(float x) { return rcp_fast(1.f + exp2_fast(-x * LOG2E)); }
// Wave totals of N per-lane values (N a power of two <= 64) with N - 1 + log2(64 / N) shuffles instead of the 6 N
// DPP steps (plus their wait states) of N wave_sum calls: each halving step hands half of the values to the partner
// lane.  On return v[0] of lane L is the total of the value with index L >> (6 - log2 N).
// (compile-time recursion: a run-time halving loop is not unrolled and turns v[] into scratch memory)
template <int N, int S> struct WaveMultiSum {
  template <int M> static __device__ __forceinline__ void run(float (&v)[M], int lane) {
    const bool up = (lane & S) != 0;
#pragma unroll
    for (int i = 0; i < N / 2; i++) {
      const float keep = up ? v[i + N / 2] : v[i], send = up ? v[i] : v[i + N / 2];
      v[i] = keep + shfl_xor(send, S);
    }
    WaveMultiSum<N / 2, S / 2>::run(v, lane);
  }
};
template <int S> struct WaveMultiSum<1, S> {
  template <int M> static __device__ __forceinline__ void run(float (&v)[M], int) {
    v[0] += shfl_xor(v[0], S);
    WaveMultiSum<1, S / 2>::run(v, 0);
  }
};
template <> struct WaveMultiSum<1, 0> {
  template <int M> static __device__ __forceinline__ void run(float (&)[M], int) {}
};
template <int N> __device__ __forceinline__ void wave_multi_sum(float (&v)[N]) {
  static_assert(N >= 1 && N <= 64 && (N & (N - 1)) == 0, "power of two");
  WaveMultiSum<N, 32>::run(v, lane_id());
}

// The same totals for 32 values WITHOUT the LDS crossbar: behind __shfl_xor every halving step was a ds_bpermute with its own
// s_waitcnt (32 round trips per call: ~3 k cycles in selscan_bwd_lanes_kernel).  Steps 32 and 16 are register swaps between lane halves
// (v_permlane32_swap / v_permlane16_swap: the pair of values a lane keeps / sends IS the swap), steps 8 .. 1 DPP row operations folded into
// the add.  On return v[0] of lane L is the total of the value with index L >> 1 (as wave_multi_sum<32>).
__device__ __forceinline__ void wave_sum32(float (&v)[32]) {
#ifdef OMK_EMU
  wave_multi_sum<32>(v);
#else
  const int lane = lane_id();
  // (the two halves of the builtin's result are taken apart through the inline asm below: written as r[0] + r[1] the compiler added the first
  // result to itself -- tools/probe/wave_sum32_probe.hip)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    float x = v[i], y = v[i + 16];
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    v[i] = x + y;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) {
    float x = v[i], y = v[i + 8];
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    v[i] = x + y;
  }
#define OMK_DPP_MOV(x, ctrl, bank) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, bank, false))
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float keep = up ? v[i + 4] : v[i], send = up ? v[i] : v[i + 4];
      v[i] = keep + OMK_DPP_MOV(send, 0x128, 0xf);                       // row_ror:8 = lane ^ 8 inside a row of 16
    }
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float keep = up ? v[i + 2] : v[i], send = up ? v[i] : v[i + 2];
      const int lo = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, send), 0x104, 0xf, 0x5, false);     // row_shl:4 -> banks 0, 2 read lane + 4
      const int x4 = __builtin_amdgcn_update_dpp(lo, __builtin_bit_cast(int, send), 0x114, 0xf, 0xa, false);    // row_shr:4 -> banks 1, 3 read lane - 4
      v[i] = keep + __builtin_bit_cast(float, x4);
    }
  }
  {
    const bool up = (lane & 2) != 0;
    const float keep = up ? v[1] : v[0], send = up ? v[0] : v[1];
    v[0] = keep + OMK_DPP_MOV(send, 0x4e, 0xf);                          // quad_perm [2, 3, 0, 1]
  }
  v[0] += OMK_DPP_MOV(v[0], 0xb1, 0xf);                                  // quad_perm [1, 0, 3, 2]
#undef OMK_DPP_MOV
#endif
}

__device__ __forceinline__ float silu_fast(float x) { return x * sigmoid_fast(x); }
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + expf(-x)); }

}  // namespace omk
