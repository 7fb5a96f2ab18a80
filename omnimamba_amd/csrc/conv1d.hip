// conv1d.hip -- causal depthwise conv1d (width 2..4) + bias + SiLU: forward, backward, single-step update.
//
// HBM-bound (2 * tok * C * s bytes).  The Mamba-2 block stores xBC channel-last
// ((B, L, C) view of the in_proj output, row stride 8512), so the fast path gives each lane 8 adjacent channels
// (one 16-byte load per token, a wave covers 1 KB of one token row) and walks TL consecutive tokens with the
// W-1 previous inputs held in registers: every input byte is fetched once per tile (+ a W-1 token halo that
// hits L2).  Any other layout (channel-first, odd sizes, fp32 rows not 16-byte aligned) takes the strided
// scalar kernel, whose fastest thread index follows the unit-stride dimension.
#include <type_traits>
#include "omk_common.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int CONV_MAXW = 4;

struct ConvArgs {
  const void* x; const void* w; const void* bias; const void* init; const void* dout;
  void* out; void* fin; void* dx; float* dw; float* db; void* dinit;
  float* part;   // backward, optional: [B * tiles][C * (W + 1)] partial dw / db rows instead of atomics (conv1d_bwd_cl4_kernel)
  int FW;   // columns of final_states (>= W - 1)
  int64_t xsb, xsc, xsl, osb, osc, osl, isb, isc, isl, fsb, fsc, fsl, dosb, dosc, dosl, dxsb, dxsc, dxsl, disb, disc, disl;
  int64_t wsc, wsk;
  int B, C, L, W, silu, wdt, bdt, idt, fdt;
};

__device__ __forceinline__ float silu_grad(float pre) {
  float s = sigmoid_fast(pre);
  return s * (1.f + pre * (1.f - s));
}

// ---------------------------------------------------------------------------------------------------------
// channel-last vectorised forward: thread = (b, token tile, 8-channel vector)
// ---------------------------------------------------------------------------------------------------------
template <class T, int VEC, int TL, int W, int TG>
__global__ __launch_bounds__(256) void conv1d_fwd_cl_kernel(ConvArgs a) {
  const int CV = a.C / VEC, NT = (a.L + TL - 1) / TL;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)a.B * NT * CV) return;
  const int cv = (int)(g % CV), tile = (int)((g / CV) % NT), b = (int)(g / ((int64_t)CV * NT));
  const int c0 = cv * VEC, l0 = tile * TL;
  const T* x = (const T*)a.x + (int64_t)b * a.xsb + c0;
  T* out = (T*)a.out + (int64_t)b * a.osb + c0;
  // win[k] holds the input at position l - (W-1) + k  (k = W-1 is the current token)
  float w[W][VEC], bias[VEC], win[W][VEC];
  // The prologue of a thread -- taps, bias, the W - 1 rows in front of its tile -- as ONE batch of requests, conversions behind the last
  // of them (raw_rt_flat / cvt_rt_flat, omk_common.h).  Written as `cond ? load_rt(...) : 0` (rounds 1 - 4) every one of these ~ 20 small
  // loads was a branch with s_waitcnt vmcnt(0) behind it: ~ 20 dependent round trips in front of a main loop of 8 (tools/isa_waits.py).
  {
    RawElem qb[VEC], qw[W][VEC], qi[W][VEC];
    vec_t<T, VEC> qx[W];
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      qb[i] = raw_rt_flat(a.bias ? a.bias : a.w, a.bias ? c0 + i : 0, a.bias ? a.bdt : a.wdt);
#pragma unroll
      for (int k = 0; k < W; k++) qw[k][i] = raw_rt_flat(a.w, (int64_t)(c0 + i) * a.wsc + k * a.wsk, a.wdt);
    }
    // before the first shift slot s (1..W-1) holds position l0 - W + s: the row itself (clamped), or the initial state in front of the sequence
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      qx[s] = *reinterpret_cast<const vec_t<T, VEC>*>(x + (int64_t)(l >= 0 ? l : 0) * a.xsl);
      const bool ini = a.init != nullptr && l < 0 && W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++)
        qi[s][i] = raw_rt_flat(ini ? a.init : a.w, ini ? (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(W - 1 + l) * a.isl : 0, ini ? a.idt : a.wdt);
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      bias[i] = a.bias ? cvt_rt_flat(qb[i], a.bdt) : 0.f;
#pragma unroll
      for (int k = 0; k < W; k++) w[k][i] = cvt_rt_flat(qw[k][i], a.wdt);
    }
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      const bool ini = a.init != nullptr && l < 0 && W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++) win[s][i] = l >= 0 ? to_f32(qx[s].e[i]) : (ini ? cvt_rt_flat(qi[s][i], a.idt) : 0.f);
    }
  }
  const int lend = (l0 + TL < a.L) ? l0 + TL : a.L;
  // groups of TG tokens: all TG row loads are issued before the first one is consumed (a strictly sequential walk
  // exposes one HBM latency per token)
#pragma unroll 1
  for (int lg = l0; lg < lend; lg += TG) {
    vec_t<T, VEC> raw[TG];
#pragma unroll
    for (int j = 0; j < TG; j++) {
      const int l = lg + j < lend ? lg + j : lend - 1;   // clamped, so the loads stay unconditional
      raw[j] = *reinterpret_cast<const vec_t<T, VEC>*>(x + (int64_t)l * a.xsl);
    }
#pragma unroll
    for (int j = 0; j < TG; j++) {
      if (lg + j < lend) {
#pragma unroll
        for (int s = 0; s + 1 < W; s++)
#pragma unroll
          for (int i = 0; i < VEC; i++) win[s][i] = win[s + 1][i];
#pragma unroll
        for (int i = 0; i < VEC; i++) win[W - 1][i] = to_f32(raw[j].e[i]);
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          float acc = bias[i];
#pragma unroll
          for (int k = 0; k < W; k++) acc = fma_f32(w[k][i], win[k][i], acc);
          o[i] = a.silu ? silu_fast(acc) : acc;
        }
        store_vec<T, VEC>(out + (int64_t)(lg + j) * a.osl, o);
      }
    }
  }
  if (a.fin && lend == a.L) {
    // final_states[j] = xpad[L + (W - 1) - FW + j], xpad = [init | x]: the last FW >= W - 1 inputs (FW = W: the conv_state of Mamba2's cache)
    for (int j = 0; j < a.FW; j++) {
      const int l = a.L + j - a.FW;
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        float v = 0.f;
        if (l >= 0) v = to_f32(x[(int64_t)l * a.xsl + i]);
        else if (a.init && W - 1 + l >= 0) v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(W - 1 + l) * a.isl, a.idt);
        store_rt(a.fin, (int64_t)b * a.fsb + (int64_t)(c0 + i) * a.fsc + (int64_t)j * a.fsl, a.fdt, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// strided scalar forward: thread = one output element; fastest thread index along the unit-stride dim
// ---------------------------------------------------------------------------------------------------------
template <class T>
__device__ __forceinline__ float conv_in(const ConvArgs& a, const T* x, int b, int c, int l) {
  if (l >= 0) return to_f32(x[(int64_t)b * a.xsb + (int64_t)c * a.xsc + (int64_t)l * a.xsl]);
  if (a.init && a.W - 1 + l >= 0) return load_rt(a.init, (int64_t)b * a.isb + (int64_t)c * a.isc + (int64_t)(a.W - 1 + l) * a.isl, a.idt);
  return 0.f;
}

// ---------------------------------------------------------------------------------------------------------
// The forward walk with scalar token positions (round 5; see conv1d_bwd_cl4_kernel): four 16-bit channels (8 bytes) per lane, a wave =
// one strip of TL tokens x 64 channel quads, the strip index through readfirstlane -- row offsets are scalar, rows move through buffer
// resources (idle lanes of the last channel block point behind the range), the interior groups of a tile run a test-free token step.
// ---------------------------------------------------------------------------------------------------------
#ifndef TGF
#define TGF 8   // tokens requested per group
#endif
template <class T, int TL, int W, int TG, bool WF>
__global__ __launch_bounds__(256) void conv1d_fwd_cl8_kernel(ConvArgs a) {
  constexpr int VEC = 4;
  static_assert(sizeof(T) == 2, "four 16-bit channels = 8 bytes per lane");
  const int CV = a.C / VEC, NT4 = (a.L + 4 * TL - 1) / (4 * TL), CVB = (CV + 63) / 64;
  const int cvb = blockIdx.x % CVB, t4 = (blockIdx.x / CVB) % NT4, b = blockIdx.x / (CVB * NT4);
  const int cvl = threadIdx.x & 63, strip = uniform_i(threadIdx.x >> 6);
  const int cv = cvb * 64 + cvl;
  const bool cvok = cv < CV;
  const int c0 = (cvok ? cv : 0) * VEC, l0 = (t4 * 4 + strip) * TL;
  if (l0 >= a.L) return;                                           // (scalar: the whole wave)
  const T* x = (const T*)a.x + (int64_t)b * a.xsb + c0;
  float w[W][VEC], bias[VEC], win[W][VEC];
  if constexpr (WF) {
    // SHORT strips need a short prologue (WF: fp32 (C, W) weight rows, contiguous and 16-byte aligned, fp32 bias or none -- checked by the
    // launcher): the lane's VEC x W taps are VEC x W consecutive floats = W 16-byte requests, the bias one more, the W - 1 halo rows three
    // 8-byte requests behind ONE scalar test -- against 16 + 4 + 12 run-time-dtype element requests in the general prologue below, which
    // costs a 16-token strip more than its tokens (profiles/r06_stream_kernels.txt).
    f32x4 wq[W], bq = {0.f, 0.f, 0.f, 0.f};
    const f32x4* wp = reinterpret_cast<const f32x4*>((const float*)a.w + (int64_t)c0 * W);
#pragma unroll
    for (int q = 0; q < W; q++) wq[q] = wp[q];
    if (a.bias) bq = *reinterpret_cast<const f32x4*>((const float*)a.bias + c0);
    const bool interior = l0 >= W - 1;                                // (scalar)
    vec_t<T, VEC> qx[W];
    if (interior) {
#pragma unroll
      for (int s = 1; s < W; s++) qx[s] = *reinterpret_cast<const vec_t<T, VEC>*>(x + (int64_t)(l0 - W + s) * a.xsl);
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      bias[i] = bq[i];
      win[0][i] = 0.f;
#pragma unroll
      for (int k = 0; k < W; k++) w[k][i] = wq[(i * W + k) / 4][(i * W + k) % 4];
    }
    if (interior) {
#pragma unroll
      for (int s = 1; s < W; s++)
#pragma unroll
        for (int i = 0; i < VEC; i++) win[s][i] = to_f32(qx[s].e[i]);
    } else {   // the first strip of a sequence: initial states or zeros, earlier tokens of the strip's own sequence
#pragma unroll
      for (int s = 1; s < W; s++) {
        const int l = l0 - W + s;
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          float v = 0.f;
          if (l >= 0) v = to_f32(x[(int64_t)l * a.xsl + i]);
          else if (a.init && W - 1 + l >= 0) v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(W - 1 + l) * a.isl, a.idt);
          win[s][i] = v;
        }
      }
    }
  } else {
    RawElem qb[VEC], qw[W][VEC], qi[W][VEC];
    vec_t<T, VEC> qx[W];
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      qb[i] = raw_rt_flat(a.bias ? a.bias : a.w, a.bias ? c0 + i : 0, a.bias ? a.bdt : a.wdt);
#pragma unroll
      for (int k = 0; k < W; k++) qw[k][i] = raw_rt_flat(a.w, (int64_t)(c0 + i) * a.wsc + k * a.wsk, a.wdt);
    }
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      qx[s] = *reinterpret_cast<const vec_t<T, VEC>*>(x + (int64_t)(l >= 0 ? l : 0) * a.xsl);
      const bool ini = a.init != nullptr && l < 0 && W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++)
        qi[s][i] = raw_rt_flat(ini ? a.init : a.w, ini ? (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(W - 1 + l) * a.isl : 0, ini ? a.idt : a.wdt);
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      bias[i] = a.bias ? cvt_rt_flat(qb[i], a.bdt) : 0.f;
#pragma unroll
      for (int k = 0; k < W; k++) w[k][i] = cvt_rt_flat(qw[k][i], a.wdt);
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) win[0][i] = 0.f;
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      const bool ini = a.init != nullptr && l < 0 && W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++) win[s][i] = l >= 0 ? to_f32(qx[s].e[i]) : (ini ? cvt_rt_flat(qi[s][i], a.idt) : 0.f);
    }
  }
  const int lend = (l0 + TL < a.L) ? l0 + TL : a.L;
  const uint32_t vch = cvok ? 2u * (uint32_t)c0 : 0x7ffffff0u;
  const BufRes xr = make_buf((const T*)a.x + (int64_t)b * a.xsb, (uint32_t)(((int64_t)(a.L - 1) * a.xsl + a.C) * 2));
  const BufRes orr = make_buf((T*)a.out + (int64_t)b * a.osb, (uint32_t)(((int64_t)(a.L - 1) * a.osl + a.C) * 2));
  const uint32_t xrow = 2u * (uint32_t)a.xsl, orow = 2u * (uint32_t)a.osl;
  const bool silu_on = a.silu != 0;
  auto un4 = [](u32x2 r, float (&o)[VEC]) {
    if constexpr (std::is_same<T, bf16_t>::value) {
      o[0] = __builtin_bit_cast(float, r[0] << 16); o[1] = __builtin_bit_cast(float, r[0] & 0xffff0000u);
      o[2] = __builtin_bit_cast(float, r[1] << 16); o[3] = __builtin_bit_cast(float, r[1] & 0xffff0000u);
    } else {
      o[0] = to_f32(__builtin_bit_cast(T, (uint16_t)r[0])); o[1] = to_f32(__builtin_bit_cast(T, (uint16_t)(r[0] >> 16)));
      o[2] = to_f32(__builtin_bit_cast(T, (uint16_t)r[1])); o[3] = to_f32(__builtin_bit_cast(T, (uint16_t)(r[1] >> 16)));
    }
  };
  auto token = [&](auto fast_c, u32x2 rx_, int l) {   // l: scalar
    constexpr bool FAST = decltype(fast_c)::value;
    if (FAST || l < lend) {
#pragma unroll
      for (int s = 0; s + 1 < W; s++)
#pragma unroll
        for (int i = 0; i < VEC; i++) win[s][i] = win[s + 1][i];
      un4(rx_, win[W - 1]);
      T o4[VEC];
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        float acc = bias[i];
#pragma unroll
        for (int k = 0; k < W; k++) acc = fma_f32(w[k][i], win[k][i], acc);
        o4[i] = from_f32<T>(silu_on ? silu_fast(acc) : acc);
      }
      u32x2 pk;
      pk[0] = (uint32_t)__builtin_bit_cast(uint16_t, o4[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, o4[1]) << 16);
      pk[1] = (uint32_t)__builtin_bit_cast(uint16_t, o4[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, o4[3]) << 16);
      buf_st8(orr, pk, vch, orow * (uint32_t)l);
    }
  };
#pragma unroll 1
  for (int lg = l0; lg < lend; lg += TG) {
    u32x2 raw[TG];
#pragma unroll
    for (int j = 0; j < TG; j++) {
      const int l = lg + j < lend ? lg + j : lend - 1;   // (scalar clamp)
      raw[j] = buf_ld8(xr, vch, xrow * (uint32_t)l);
    }
    if (lg + TG <= lend) {
#pragma unroll
      for (int j = 0; j < TG; j++) token(std::true_type{}, raw[j], lg + j);
    } else {
#pragma unroll
      for (int j = 0; j < TG; j++) token(std::false_type{}, raw[j], lg + j);
    }
  }
  if (a.fin && lend == a.L && cvok) {
    // final_states[j] = xpad[L + (W - 1) - FW + j], xpad = [init | x]: the last FW >= W - 1 inputs (FW = W: the conv_state of Mamba2's cache)
    for (int j = 0; j < a.FW; j++) {
      const int l = a.L + j - a.FW;
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        float v = 0.f;
        if (l >= 0) v = to_f32(x[(int64_t)l * a.xsl + i]);
        else if (a.init && W - 1 + l >= 0) v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(W - 1 + l) * a.isl, a.idt);
        store_rt(a.fin, (int64_t)b * a.fsb + (int64_t)(c0 + i) * a.fsc + (int64_t)j * a.fsl, a.fdt, v);
      }
    }
  }
}

template <class T>
__global__ void conv1d_fwd_generic_kernel(ConvArgs a, int l_fastest) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)a.B * a.C * a.L) return;
  int b, c, l;
  if (l_fastest) { l = (int)(g % a.L); c = (int)((g / a.L) % a.C); b = (int)(g / ((int64_t)a.L * a.C)); }
  else { c = (int)(g % a.C); l = (int)((g / a.C) % a.L); b = (int)(g / ((int64_t)a.L * a.C)); }
  const T* x = (const T*)a.x;
  float acc = a.bias ? load_rt(a.bias, c, a.bdt) : 0.f;
  for (int k = 0; k < a.W; k++) acc = fma_f32(load_rt(a.w, (int64_t)c * a.wsc + k * a.wsk, a.wdt), conv_in<T>(a, x, b, c, l - (a.W - 1) + k), acc);
  if (a.silu) acc = silu_f(acc);
  ((T*)a.out)[(int64_t)b * a.osb + (int64_t)c * a.osc + (int64_t)l * a.osl] = from_f32<T>(acc);
}

template <class T>
__global__ void conv1d_final_states_kernel(ConvArgs a) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Wm = a.FW;
  if (g >= (int64_t)a.B * a.C * Wm) return;
  const int c = (int)(g % a.C), j = (int)((g / a.C) % Wm), b = (int)(g / ((int64_t)a.C * Wm));
  float v = conv_in<T>(a, (const T*)a.x, b, c, a.L + j - Wm);
  store_rt(a.fin, (int64_t)b * a.fsb + (int64_t)c * a.fsc + (int64_t)j * a.fsl, a.fdt, v);
}

// ---------------------------------------------------------------------------------------------------------
// backward.  pre[l] = bias + sum_k w[k] xpad[l+k] ; dpre = dout * silu'(pre)
//   dx[l] = sum_k w[k] dpre[l + (W-1) - k]   (terms with index >= L vanish)
//   dw[k] = sum_{b,l} dpre[l] xpad[l+k] ; db = sum dpre ; dinit[j] = sum_k w[k] dpre[j - k] (j-k >= 0, < L)
// channel-last vectorised: thread = (b, tile of TL tokens, 8 channels); dw/db reduced with one atomic per thread.
// ---------------------------------------------------------------------------------------------------------
// block = 64 channel vectors (one wave row of 1 KB per token) x 4 token strips of TL tokens; the four strips fold
// their dw / db partials through LDS so each (channel, tap) costs one global atomic per block.
template <class T, int VEC, int TL, int W, int TG>
__global__ __launch_bounds__(256) void conv1d_bwd_cl_kernel(ConvArgs a) {
  __shared__ float sred[4][64][VEC * (W + 1)];
  const int CV = a.C / VEC, NT4 = (a.L + 4 * TL - 1) / (4 * TL), CVB = (CV + 63) / 64;
  const int cvb = blockIdx.x % CVB, t4 = (blockIdx.x / CVB) % NT4, b = blockIdx.x / (CVB * NT4);
  const int cvl = threadIdx.x & 63, strip = threadIdx.x >> 6;
  const int cv = cvb * 64 + cvl;
  const bool cvok = cv < CV;
  const int c0 = (cvok ? cv : 0) * VEC, l0 = (t4 * 4 + strip) * TL;
  const T* x = (const T*)a.x;
  const T* dout = (const T*)a.dout + (int64_t)b * a.dosb + c0;
  T* dx = (T*)a.dx + (int64_t)b * a.dxsb + c0;
  float w[W][VEC], bias[VEC], dwacc[W][VEC], dbacc[VEC];
  // (taps, bias and the rows in front of the tile: one batch of requests, conversions behind -- see conv1d_fwd_cl_kernel)
  RawElem qb[VEC], qw[W][VEC];
#pragma unroll
  for (int i = 0; i < VEC; i++) {
    qb[i] = raw_rt_flat(a.bias ? a.bias : a.w, a.bias ? c0 + i : 0, a.bias ? a.bdt : a.wdt);
    dbacc[i] = 0.f;
#pragma unroll
    for (int k = 0; k < W; k++) {
      qw[k][i] = raw_rt_flat(a.w, (int64_t)(c0 + i) * a.wsc + k * a.wsk, a.wdt);
      dwacc[k][i] = 0.f;
    }
  }
  const int lend = !cvok ? l0 : ((l0 + TL < a.L) ? l0 + TL : a.L);   // idle lanes (cv out of range) do zero tokens
  // walking p = l0 .. lend+W-2: xw[k] = xpad at position p-(W-1)+k, dp[k] = dpre[p-(W-1)+k];
  // at step p dpre[p] becomes known and dx[p-(W-1)] = sum_k w[k] dpre[p-k] = sum_k w[k] dp[W-1-k] is complete.
  float xw[W][VEC], dp[W][VEC];
#pragma unroll
  for (int s = 0; s < W; s++)
#pragma unroll
    for (int i = 0; i < VEC; i++) { xw[s][i] = 0.f; dp[s][i] = 0.f; }
  {
    T qx[W][VEC];
    RawElem qi[W][VEC];
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      const bool ini = a.init != nullptr && l < 0 && a.W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        qx[s][i] = x[(int64_t)b * a.xsb + (int64_t)(c0 + i) * a.xsc + (int64_t)(l >= 0 ? l : 0) * a.xsl];
        qi[s][i] = raw_rt_flat(ini ? a.init : a.w, ini ? (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(a.W - 1 + l) * a.isl : 0, ini ? a.idt : a.wdt);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      bias[i] = a.bias ? cvt_rt_flat(qb[i], a.bdt) : 0.f;
#pragma unroll
      for (int k = 0; k < W; k++) w[k][i] = cvt_rt_flat(qw[k][i], a.wdt);
    }
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      const bool ini = a.init != nullptr && l < 0 && a.W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++) xw[s][i] = l >= 0 ? to_f32(qx[s][i]) : (ini ? cvt_rt_flat(qi[s][i], a.idt) : 0.f);
    }
  }
  // dpre of the W-1 positions before l0 is NOT needed: dx[lo] only uses dpre[lo .. lo+W-1], lo >= l0.
  const int pend = lend + W - 1;
  const T* xb = x + (int64_t)b * a.xsb + c0;
#pragma unroll 1
  for (int pg = l0; pg < pend; pg += TG) {
    vec_t<T, VEC> rawx[TG], rawg[TG];
#pragma unroll
    for (int j = 0; j < TG; j++) {
      const int pc = pg + j < a.L ? pg + j : a.L - 1;   // clamped: unconditional loads, masked below
      rawx[j] = *reinterpret_cast<const vec_t<T, VEC>*>(xb + (int64_t)pc * a.xsl);
      rawg[j] = *reinterpret_cast<const vec_t<T, VEC>*>(dout + (int64_t)pc * a.dosl);
    }
#pragma unroll
    for (int j = 0; j < TG; j++) {
      const int p = pg + j;
      if (p < pend) {
#pragma unroll
        for (int s = 0; s + 1 < W; s++)
#pragma unroll
          for (int i = 0; i < VEC; i++) { xw[s][i] = xw[s + 1][i]; dp[s][i] = dp[s + 1][i]; }
        const bool inside = p < a.L;
        float go[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          xw[W - 1][i] = inside ? to_f32(rawx[j].e[i]) : 0.f;
          go[i] = inside ? to_f32(rawg[j].e[i]) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          float d = go[i];
          if (a.silu && inside) {
            float pre = bias[i];
#pragma unroll
            for (int k = 0; k < W; k++) pre += w[k][i] * xw[k][i];
            d *= silu_grad(pre);
          }
          dp[W - 1][i] = inside ? d : 0.f;
          if (inside && p < lend) {   // dw/db: each position is counted by exactly one tile
            dbacc[i] += d;
#pragma unroll
            for (int k = 0; k < W; k++) dwacc[k][i] += d * xw[k][i];
          }
        }
        const int lo = p - (W - 1);
        if (lo >= l0 && lo < lend) {
          float o[VEC];
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < W; k++) acc += w[k][i] * dp[W - 1 - k][i];
            o[i] = acc;
          }
          store_vec<T, VEC>(dx + (int64_t)lo * a.dxsl, o);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; i++) {
    sred[strip][cvl][i * (W + 1) + W] = dbacc[i];
#pragma unroll
    for (int k = 0; k < W; k++) sred[strip][cvl][i * (W + 1) + k] = dwacc[k][i];
  }
  block_sync();
  if (strip == 0 && cvok) {
#pragma unroll
    for (int i = 0; i < VEC; i++) {
#pragma unroll
      for (int k = 0; k <= W; k++) {
        const int j = i * (W + 1) + k;
        const float v = sred[0][cvl][j] + sred[1][cvl][j] + sred[2][cvl][j] + sred[3][cvl][j];
        if (k < W) atomic_add_f32(a.dw + (int64_t)(c0 + i) * W + k, v);
        else if (a.db) atomic_add_f32(a.db + c0 + i, v);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// The same walk for two 16-bit channels per lane (the shipped configuration), round 5.  conv1d_bwd_cl_kernel is VALU bound: ~ 125
// instructions per token and lane, of which ~ 60 are v_cndmask / v_mov for the tile- and sequence-edge masks and ~ 14 are 64-bit address
// arithmetic per request.  Here
//   * the token position of a wave is a SCALAR (strip = readfirstlane): row offsets are s_mul / s_add, every edge test a scalar branch;
//   * rows move through buffer resources: the lane part of an address is one constant VGPR (idle lanes of the last channel block point
//     behind the range: their loads read zeros and their stores are dropped), the row part the instruction's scalar offset;
//   * the token step exists twice: the seven interior groups of a 64-token tile take a copy without any test -- the window shifts are
//     register renames there;
//   * (round 6) the lane's two channels are one packed fp32 pair (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) and contiguous fp32 weights take a
//     prologue of 16-byte requests (WF): 207 -> 185 us at the cfg 2 slice (profiles/r06_stream_kernels.txt).
// ---------------------------------------------------------------------------------------------------------
#ifndef TGX
#define TGX 4   // tokens requested per group (round 6, packed pairs: 4 / 8 / 16 = 76 / 94 / 142 registers, 193.8 / 189.6 / 194.2 us -- no difference)
#endif
template <class T, int TL, int W, int TG, int NS, bool WF>   // NS strips of TL tokens per workgroup (one wave each): NS TL tokens per set of dw / db atomics
__global__ __launch_bounds__(64 * NS) void conv1d_bwd_cl4_kernel(ConvArgs a) {   // (74 - 76 registers with the packed pairs of round 6; 102 before)
  constexpr int VEC = 2;
  static_assert(sizeof(T) == 2, "two 16-bit channels = one dword per lane");
  __shared__ float sred[NS][64][VEC * (W + 1)];
  const int CV = a.C / VEC, NT4 = (a.L + NS * TL - 1) / (NS * TL), CVB = (CV + 63) / 64;
  const int cvb = blockIdx.x % CVB, t4 = (blockIdx.x / CVB) % NT4, b = blockIdx.x / (CVB * NT4);
  const int cvl = threadIdx.x & 63, strip = uniform_i(threadIdx.x >> 6);
  const int cv = cvb * 64 + cvl;
  const bool cvok = cv < CV;
  const int c0 = (cvok ? cv : 0) * VEC, l0 = (t4 * NS + strip) * TL;
  const T* x = (const T*)a.x;
  // the lane's two channels are ONE packed fp32 pair everywhere below: taps, window, sums -- v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do
  // both channels per instruction (the scalar form is VALU bound: ~ 45 vector + 4 transcendental instructions per token, ~ 25 + 4 packed)
  f32x2 w[W], bias, dwacc[W], dbacc;
  const int lend = (l0 + TL < a.L) ? l0 + TL : a.L;   // (scalar; idle lanes walk along on zeros)
  f32x2 xw[W], dp[W];
#pragma unroll
  for (int s = 0; s < W; s++) { xw[s] = f32x2{0.f, 0.f}; dp[s] = f32x2{0.f, 0.f}; dwacc[s] = f32x2{0.f, 0.f}; }
  dbacc = f32x2{0.f, 0.f};
  if constexpr (WF) {
    // the short prologue of the forward kernel (fp32 (C, W) weight rows, contiguous and 16-byte aligned; fp32 bias or none): the lane's 2 W taps are
    // 2 W consecutive floats, the halo rows three dword requests behind one scalar test
    const float* wp = (const float*)a.w + (int64_t)c0 * W;
    float wf[VEC * W];
    if constexpr ((VEC * W) % 4 == 0) {
#pragma unroll
      for (int q = 0; q < VEC * W / 4; q++) {
        const f32x4 v = reinterpret_cast<const f32x4*>(wp)[q];
#pragma unroll
        for (int e = 0; e < 4; e++) wf[4 * q + e] = v[e];
      }
    } else {
#pragma unroll
      for (int q = 0; q < VEC * W / 2; q++) {
        const f32x2 v = reinterpret_cast<const f32x2*>(wp)[q];
        wf[2 * q] = v[0]; wf[2 * q + 1] = v[1];
      }
    }
    bias = a.bias ? *reinterpret_cast<const f32x2*>((const float*)a.bias + c0) : f32x2{0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < W; kk++) w[kk] = f32x2{wf[kk], wf[W + kk]};
    if (l0 >= W - 1 && l0 < a.L) {                                      // (scalar)
      uint32_t qx[W];
#pragma unroll
      for (int s = 1; s < W; s++) qx[s] = *reinterpret_cast<const uint32_t*>(x + (int64_t)b * a.xsb + c0 + (int64_t)(l0 - W + s) * a.xsl);
#pragma unroll
      for (int s = 1; s < W; s++) {
        if constexpr (std::is_same<T, bf16_t>::value) xw[s] = f32x2{__builtin_bit_cast(float, qx[s] << 16), __builtin_bit_cast(float, qx[s] & 0xffff0000u)};
        else xw[s] = f32x2{to_f32(__builtin_bit_cast(T, (uint16_t)qx[s])), to_f32(__builtin_bit_cast(T, (uint16_t)(qx[s] >> 16)))};
      }
    } else {
#pragma unroll
      for (int s = 1; s < W; s++) {
        const int l = l0 - W + s;
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          float v = 0.f;
          if (l >= 0) v = to_f32(x[(int64_t)b * a.xsb + (int64_t)(c0 + i) * a.xsc + (int64_t)(l < a.L ? l : a.L - 1) * a.xsl]);
          else if (a.init && a.W - 1 + l >= 0) v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(a.W - 1 + l) * a.isl, a.idt);
          xw[s][i] = v;
        }
      }
    }
  } else {
  RawElem qb[VEC], qw[W][VEC];
#pragma unroll
  for (int i = 0; i < VEC; i++) {
    qb[i] = raw_rt_flat(a.bias ? a.bias : a.w, a.bias ? c0 + i : 0, a.bias ? a.bdt : a.wdt);
#pragma unroll
    for (int k = 0; k < W; k++) qw[k][i] = raw_rt_flat(a.w, (int64_t)(c0 + i) * a.wsc + k * a.wsk, a.wdt);
  }
  {
    T qx[W][VEC];
    RawElem qi[W][VEC];
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      const bool ini = a.init != nullptr && l < 0 && a.W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        qx[s][i] = x[(int64_t)b * a.xsb + (int64_t)(c0 + i) * a.xsc + (int64_t)(l >= 0 ? (l < a.L ? l : a.L - 1) : 0) * a.xsl];
        qi[s][i] = raw_rt_flat(ini ? a.init : a.w, ini ? (int64_t)b * a.isb + (int64_t)(c0 + i) * a.isc + (int64_t)(a.W - 1 + l) * a.isl : 0, ini ? a.idt : a.wdt);
      }
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      bias[i] = a.bias ? cvt_rt_flat(qb[i], a.bdt) : 0.f;
#pragma unroll
      for (int k = 0; k < W; k++) w[k][i] = cvt_rt_flat(qw[k][i], a.wdt);
    }
#pragma unroll
    for (int s = 1; s < W; s++) {
      const int l = l0 - W + s;
      const bool ini = a.init != nullptr && l < 0 && a.W - 1 + l >= 0;
#pragma unroll
      for (int i = 0; i < VEC; i++) xw[s][i] = l >= 0 ? to_f32(qx[s][i]) : (ini ? cvt_rt_flat(qi[s][i], a.idt) : 0.f);
    }
  }
  }
  const int pend = lend + W - 1;
  // rows of this batch element as buffers; lane part of every address: the channel pair (idle lanes: behind the range)
  const uint32_t vch = cvok ? 2u * (uint32_t)c0 : 0x7ffffff0u;
  const BufRes xr = make_buf(x + (int64_t)b * a.xsb, (uint32_t)(((int64_t)(a.L - 1) * a.xsl + a.C) * 2));
  const BufRes gr = make_buf((const T*)a.dout + (int64_t)b * a.dosb, (uint32_t)(((int64_t)(a.L - 1) * a.dosl + a.C) * 2));
  const BufRes dr = make_buf((T*)a.dx + (int64_t)b * a.dxsb, (uint32_t)(((int64_t)(a.L - 1) * a.dxsl + a.C) * 2));
  const uint32_t xrow = 2u * (uint32_t)a.xsl, grow = 2u * (uint32_t)a.dosl, drow = 2u * (uint32_t)a.dxsl;
  const bool silu_on = a.silu != 0;
  auto un2 = [](uint32_t r) -> f32x2 {
    f32x2 o;
    if constexpr (std::is_same<T, bf16_t>::value) { o[0] = __builtin_bit_cast(float, r << 16); o[1] = __builtin_bit_cast(float, r & 0xffff0000u); }
    else { o[0] = to_f32(__builtin_bit_cast(T, (uint16_t)r)); o[1] = to_f32(__builtin_bit_cast(T, (uint16_t)(r >> 16))); }
    return o;
  };
  auto silu_grad2 = [](f32x2 pre) -> f32x2 {   // silu_grad of both channels: the two exp2 / rcp are the only unpacked steps
    const f32x2 t = pre * (-LOG2E);
    const f32x2 e = {exp2_fast(t[0]), exp2_fast(t[1])};
    const f32x2 q = e + 1.f;
    const f32x2 sg = {rcp_fast(q[0]), rcp_fast(q[1])};
    return sg * (1.f + pre * (1.f - sg));
  };
  auto token = [&](auto fast_c, uint32_t rx_, uint32_t rg_, int p) {   // p: scalar
    constexpr bool FAST = decltype(fast_c)::value;
    if (FAST || p < pend) {
#pragma unroll
      for (int s = 0; s + 1 < W; s++) { xw[s] = xw[s + 1]; dp[s] = dp[s + 1]; }
      const bool inside = FAST ? true : p < a.L;
      f32x2 d = un2(rg_), xn = un2(rx_);
      if (!inside) { d = f32x2{0.f, 0.f}; xn = f32x2{0.f, 0.f}; }
      xw[W - 1] = xn;
      if (silu_on) {
        f32x2 pre = bias;
#pragma unroll
        for (int k = 0; k < W; k++) pre = fma_f32x2(w[k], xw[k], pre);
        d *= silu_grad2(pre);
      }
      dp[W - 1] = d;
      if (FAST || p < lend) {   // dw/db: each position is counted by exactly one tile (behind the sequence d is zero)
        dbacc += d;
#pragma unroll
        for (int k = 0; k < W; k++) dwacc[k] = fma_f32x2(d, xw[k], dwacc[k]);
      }
      const int lo = p - (W - 1);
      if (FAST || (lo >= l0 && lo < lend)) {
        f32x2 acc = w[0] * dp[W - 1];
#pragma unroll
        for (int k = 1; k < W; k++) acc = fma_f32x2(w[k], dp[W - 1 - k], acc);
        T o2[VEC] = {from_f32<T>(acc[0]), from_f32<T>(acc[1])};
        const uint32_t pk = (uint32_t)__builtin_bit_cast(uint16_t, o2[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, o2[1]) << 16);
        buf_st_f32(dr, __builtin_bit_cast(float, pk), vch, drow * (uint32_t)lo);
      }
    }
  };
#pragma unroll 1
  for (int pg = l0; pg < pend; pg += TG) {
    uint32_t rawx[TG], rawg[TG];
#pragma unroll
    for (int j = 0; j < TG; j++) {
      const int pc = pg + j < a.L ? pg + j : a.L - 1;   // (scalar clamp; masked in the edge copy of the step)
      rawx[j] = __builtin_bit_cast(uint32_t, buf_ld_f32(xr, vch, xrow * (uint32_t)pc));
      rawg[j] = __builtin_bit_cast(uint32_t, buf_ld_f32(gr, vch, grow * (uint32_t)pc));
    }
    if (pg >= l0 + W - 1 && pg + TG <= lend) {   // interior group (lend <= L)
#pragma unroll
      for (int j = 0; j < TG; j++) { token(std::true_type{}, rawx[j], rawg[j], pg + j); OMK_SCHED_FENCE(); }   // (fence: eight steps scheduled as one block keep 146 registers live)
    } else {
#pragma unroll
      for (int j = 0; j < TG; j++) token(std::false_type{}, rawx[j], rawg[j], pg + j);
    }
  }
#pragma unroll
  for (int i = 0; i < VEC; i++) {
    sred[strip][cvl][i * (W + 1) + W] = dbacc[i];
#pragma unroll
    for (int k = 0; k < W; k++) sred[strip][cvl][i * (W + 1) + k] = dwacc[k][i];
  }
  block_sync();
  if (a.part) {
    // one partial row per (batch, tile of NS strips): the lane's VEC (W + 1) sums are consecutive floats, spread over the NS waves; a second launch
    // (conv1d_bwd_fold_kernel) adds the rows in a fixed order -- no atomics, the same gradients on every run
    if (cvok) {
      float* prow = a.part + ((int64_t)b * NT4 + t4) * ((int64_t)a.C * (W + 1)) + (int64_t)c0 * (W + 1);
      for (int j = strip; j < VEC * (W + 1); j += NS) {
        float v = sred[0][cvl][j];
#pragma unroll
        for (int q = 1; q < NS; q++) v += sred[q][cvl][j];
        prow[j] = v;
      }
    }
    return;
  }
  if (strip == 0 && cvok) {
#pragma unroll
    for (int i = 0; i < VEC; i++) {
#pragma unroll
      for (int k = 0; k <= W; k++) {
        const int j = i * (W + 1) + k;
        float v = sred[0][cvl][j];
#pragma unroll
        for (int q = 1; q < NS; q++) v += sred[q][cvl][j];
        if (k < W) atomic_add_f32(a.dw + (int64_t)(c0 + i) * W + k, v);
        else if (a.db) atomic_add_f32(a.db + c0 + i, v);
      }
    }
  }
}

// dw[c][k] += sum_p part[p][c (W + 1) + k], db[c] += sum_p part[p][c (W + 1) + W]: 64 columns x 16 row groups per workgroup, four requests in flight per
// lane, rows added in a fixed order (group sums in index order)
__global__ __launch_bounds__(1024) void conv1d_bwd_fold_kernel(const float* part, int P, int C, int W, float* dw, float* db) {
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int64_t ncol = (int64_t)C * (W + 1), col = (int64_t)blockIdx.x * 64 + cl;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (col < ncol) {
    for (int p0 = g; p0 < P; p0 += 64) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int p = p0 + 16 * u; v[u] = p < P ? part[(int64_t)p * ncol + col] : 0.f; }
#pragma unroll
      for (int u = 0; u < 4; u++) s[u] += v[u];
    }
  }
  sh[g][cl] = (s[0] + s[1]) + (s[2] + s[3]);
  block_sync();
  if (g == 0 && col < ncol) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) v += sh[q][cl];
    const int c = (int)(col / (W + 1)), k = (int)(col % (W + 1));
    if (k < W) dw[(int64_t)c * W + k] += v;
    else if (db) db[c] += v;
  }
}

// strided scalar backward: thread = (b, c), sequential over L (fallback for channel-first / odd layouts)
template <class T>
__global__ void conv1d_bwd_generic_kernel(ConvArgs a) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)a.B * a.C) return;
  const int c = (int)(g % a.C), b = (int)(g / a.C), W = a.W;
  const T* x = (const T*)a.x;
  const T* dout = (const T*)a.dout;
  float w[CONV_MAXW], dwacc[CONV_MAXW] = {0.f, 0.f, 0.f, 0.f}, dbacc = 0.f;
  for (int k = 0; k < CONV_MAXW; k++) w[k] = k < W ? load_rt(a.w, (int64_t)c * a.wsc + k * a.wsk, a.wdt) : 0.f;
  const float bias = a.bias ? load_rt(a.bias, c, a.bdt) : 0.f;
  auto dpre = [&](int l) -> float {
    if (l < 0 || l >= a.L) return 0.f;
    float d = to_f32(dout[(int64_t)b * a.dosb + (int64_t)c * a.dosc + (int64_t)l * a.dosl]);
    if (a.silu) {
      float pre = bias;
      for (int k = 0; k < W; k++) pre += w[k] * conv_in<T>(a, x, b, c, l - (W - 1) + k);
      d *= silu_grad(pre);
    }
    return d;
  };
  for (int l = 0; l < a.L; l++) {
    float acc = 0.f;
    for (int k = 0; k < W; k++) acc += w[k] * dpre(l + (W - 1) - k);
    ((T*)a.dx)[(int64_t)b * a.dxsb + (int64_t)c * a.dxsc + (int64_t)l * a.dxsl] = from_f32<T>(acc);
    float d = dpre(l);
    dbacc += d;
    for (int k = 0; k < W; k++) dwacc[k] += d * conv_in<T>(a, x, b, c, l - (W - 1) + k);
  }
  if (a.db) atomic_add_f32(a.db + c, dbacc);
  for (int k = 0; k < W; k++) atomic_add_f32(a.dw + (int64_t)c * W + k, dwacc[k]);
}

// dinitial_states[b, c, j] = sum_k w[k] dpre[j - k] over 0 <= j-k < L  (tiny: B*C*(W-1) threads)
template <class T>
__global__ void conv1d_dinit_kernel(ConvArgs a) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int Wm = a.W - 1, W = a.W;
  if (g >= (int64_t)a.B * a.C * Wm) return;
  const int c = (int)(g % a.C), j = (int)((g / a.C) % Wm), b = (int)(g / ((int64_t)a.C * Wm));
  const T* x = (const T*)a.x;
  const T* dout = (const T*)a.dout;
  const float bias = a.bias ? load_rt(a.bias, c, a.bdt) : 0.f;
  float acc = 0.f;
  for (int k = 0; k < W; k++) {
    int l = j - k;   // xpad index j feeds pre[l] through weight k when l + k == j
    if (l < 0 || l >= a.L) continue;
    float d = to_f32(dout[(int64_t)b * a.dosb + (int64_t)c * a.dosc + (int64_t)l * a.dosl]);
    if (a.silu) {
      float pre = bias;
      for (int kk = 0; kk < W; kk++) pre += load_rt(a.w, (int64_t)c * a.wsc + kk * a.wsk, a.wdt) * conv_in<T>(a, x, b, c, l - (W - 1) + kk);
      d *= silu_grad(pre);
    }
    acc += load_rt(a.w, (int64_t)c * a.wsc + k * a.wsk, a.wdt) * d;
  }
  store_rt(a.dinit, (int64_t)b * a.disb + (int64_t)c * a.disc + (int64_t)j * a.disl, a.idt, acc);
}

// ---------------------------------------------------------------------------------------------------------
// decode update: thread = (b, c); state (B, C, S) shifted left by T, new inputs appended
// ---------------------------------------------------------------------------------------------------------
struct ConvUpdArgs {
  const void* x; void* state; const void* w; const void* bias; void* out;
  int64_t xsb, xsc, xsl, ssb, ssc, ssl, osb, osc, osl, wsc, wsk;
  int B, C, T, S, W, silu, xdt, sdt, wdt, bdt;
};
__global__ void conv1d_update_kernel(ConvUpdArgs a) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)a.B * a.C) return;
  const int c = (int)(g % a.C), b = (int)(g / a.C);
  float w[CONV_MAXW], win[CONV_MAXW];
  for (int k = 0; k < CONV_MAXW; k++) w[k] = k < a.W ? load_rt(a.w, (int64_t)c * a.wsc + k * a.wsk, a.wdt) : 0.f;
  const float bias = a.bias ? load_rt(a.bias, c, a.bdt) : 0.f;
  const int64_t sbase = (int64_t)b * a.ssb + (int64_t)c * a.ssc;
  // window = last W-1 state entries
  for (int k = 0; k < CONV_MAXW; k++) win[k] = 0.f;
  for (int k = 0; k + 1 < a.W; k++) win[k] = load_rt(a.state, sbase + (int64_t)(a.S - (a.W - 1) + k) * a.ssl, a.sdt);
  // shift the stored state left by T (S is tiny: W-1 or W)
  if (a.T < a.S) {
    for (int s = 0; s + a.T < a.S; s++) store_rt(a.state, sbase + (int64_t)s * a.ssl, a.sdt, load_rt(a.state, sbase + (int64_t)(s + a.T) * a.ssl, a.sdt));
  }
  for (int t = 0; t < a.T; t++) {
    float xv = load_rt(a.x, (int64_t)b * a.xsb + (int64_t)c * a.xsc + (int64_t)t * a.xsl, a.xdt);
    win[a.W - 1] = xv;
    float acc = bias;
    for (int k = 0; k < a.W; k++) acc = fma_f32(w[k], win[k], acc);
    if (a.silu) acc = silu_f(acc);
    store_rt(a.out, (int64_t)b * a.osb + (int64_t)c * a.osc + (int64_t)t * a.osl, a.xdt, acc);
    for (int k = 0; k + 1 < a.W; k++) win[k] = win[k + 1];
    int spos = a.S - a.T + t;
    if (spos >= 0) store_rt(a.state, sbase + (int64_t)spos * a.ssl, a.sdt, xv);
  }
}

static bool cl_fast_ok(const OmkTensor& t, int C) {   // (B, C, L) logical, channel contiguous, 16-byte rows (8 x 2-byte)
  return t.stride[1] == 1 && (t.stride[0] % 8) == 0 && (t.stride[2] % 8) == 0 && aligned16(t) && (C % 8) == 0;
}

static int fill_common(ConvArgs& a, const OmkTensor& x, const OmkTensor& w, const OmkTensor& bias, const OmkTensor& init, const char* who) {
  OMK_REQUIRE(x.ndim == 3 && w.ndim == 2, "%s: x must be (B, C, L), weight (C, W)", who);
  a.B = (int)x.shape[0]; a.C = (int)x.shape[1]; a.L = (int)x.shape[2]; a.W = (int)w.shape[1];
  OMK_REQUIRE(w.shape[0] == a.C && a.W >= 2 && a.W <= CONV_MAXW, "%s: weight must be (C, W) with W in 2..4", who);
  a.x = x.data; a.xsb = x.stride[0]; a.xsc = x.stride[1]; a.xsl = x.stride[2];
  a.w = w.data; a.wsc = w.stride[0]; a.wsk = w.stride[1]; a.wdt = w.dtype;
  a.bias = bias.data; a.bdt = bias.dtype;
  if (present(bias)) OMK_REQUIRE(numel(bias) == a.C && is_contig_last(bias), "%s: bias must be (C) contiguous", who);
  a.init = init.data; a.idt = init.dtype;
  if (present(init)) {
    OMK_REQUIRE(init.ndim == 3 && init.shape[0] == a.B && init.shape[1] == a.C && init.shape[2] == a.W - 1, "%s: initial_states must be (B, C, W-1)", who);
    a.isb = init.stride[0]; a.isc = init.stride[1]; a.isl = init.stride[2];
  }
  return OMK_OK;
}

}  // namespace omk

using namespace omk;

extern "C" int omk_causal_conv1d_fwd(const OmkConv1dFwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->x) && present(p->weight) && present(p->out), "causal_conv1d_fwd: x, weight, out required");
  ConvArgs a = {};
  int rc = fill_common(a, p->x, p->weight, p->bias, p->initial_states, "causal_conv1d_fwd");
  if (rc) return rc;
  OMK_REQUIRE(p->out.ndim == 3 && p->out.shape[0] == a.B && p->out.shape[1] == a.C && p->out.shape[2] == a.L && p->out.dtype == p->x.dtype, "causal_conv1d_fwd: out mismatch");
  a.out = p->out.data; a.osb = p->out.stride[0]; a.osc = p->out.stride[1]; a.osl = p->out.stride[2];
  a.silu = p->silu;
  a.fin = p->final_states.data; a.fdt = p->final_states.dtype;
  if (present(p->final_states)) {
    OMK_REQUIRE(p->final_states.ndim == 3 && p->final_states.shape[0] == a.B && p->final_states.shape[1] == a.C && p->final_states.shape[2] >= a.W - 1 &&
                p->final_states.shape[2] <= 2 * a.W, "causal_conv1d_fwd: final_states must be (B, C, state_len) with W - 1 <= state_len <= 2 W");
    a.FW = (int)p->final_states.shape[2];
    a.fsb = p->final_states.stride[0]; a.fsc = p->final_states.stride[1]; a.fsl = p->final_states.stride[2];
  }
  if ((int64_t)a.B * a.C * a.L == 0) return OMK_OK;
  const bool fast = cl_fast_ok(p->x, a.C) && cl_fast_ok(p->out, a.C);   // (fp32 too: the prefill of the reference's fp32 inference ran the scalar kernel at 0.85 TB/s)
  if (fast) {
    // 4 channels (8 bytes) per lane and 8 tokens per load group: 100 VGPRs / 4 waves per SIMD measured fastest on the
    // 1.3B shape (141 us vs 171 us for 8 channels per lane, which needs 158 VGPRs)
#define CONV_FWD_V(T_, VEC_, TL_, TG_) do { int64_t n = (int64_t)a.B * ((a.L + TL_ - 1) / TL_) * (a.C / VEC_); \
      dim3 grid((unsigned)((n + 255) / 256)), block(256); \
      if (a.W == 4) OMK_LAUNCH((conv1d_fwd_cl_kernel<T_, VEC_, TL_, 4, TG_>), grid, block, 0, stream, a); \
      else if (a.W == 3) OMK_LAUNCH((conv1d_fwd_cl_kernel<T_, VEC_, TL_, 3, TG_>), grid, block, 0, stream, a); \
      else OMK_LAUNCH((conv1d_fwd_cl_kernel<T_, VEC_, TL_, 2, TG_>), grid, block, 0, stream, a); } while (0)
    const char* tle = getenv("OMK_CONV_FWD_TL");   // developer A/B of the tokens per thread (bf16)
    const int tl = (tle && *tle) ? atoi(tle) : (a.L >= 1024 ? 64 : 32);   // 64: -4 % on the 1.3B slice (halo rows), 128: worse again
    const bool cl8 = !(getenv("OMK_CONV_FWD_CL8") && getenv("OMK_CONV_FWD_CL8")[0] == '0');
    const int64_t farf = (int64_t)a.L * 2 * (a.xsl > a.osl ? a.xsl : a.osl);
    const char* vce = getenv("OMK_CONV_FWD_VEC");   // developer A/B: "8" = 16 bytes per lane, 4 tokens in flight
    if (p->x.dtype == OMK_BF16 && vce && vce[0] == '8' && a.C % 8 == 0) {
      if (vce[1] == '8') CONV_FWD_V(bf16_t, 8, 64, 8); else if (vce[1] == '2') CONV_FWD_V(bf16_t, 8, 32, 4); else CONV_FWD_V(bf16_t, 8, 64, 4);
    } else
    if (cl8 && !(tle && *tle) && !(vce && *vce) && (p->x.dtype == OMK_BF16 || p->x.dtype == OMK_F16) && a.L >= 256 && a.C % 4 == 0 && farf < ((int64_t)1 << 31) &&
        a.xsc == 1 && a.osc == 1) {
      // scalar token positions (conv1d_fwd_cl8_kernel); OMK_CONV_FWD_CL8=0: the per-thread tiles of rounds 1 - 4
      // strips of 16 tokens: the shorter a wave lives the better these streams run (tools/probe/conv_probe.hip: 128 / 64 / 32 / 16 tokens
      // per strip = 4.73 / 5.06 / 5.17 / 5.45 TB/s, three halo rows per strip included) -- once the prologue is a handful of 16-byte
      // requests (WF); the general prologue keeps 64
      const char* wfe = getenv("OMK_CONV_CL8_WF");   // developer A/B: "0" = the general prologue
      const bool wf = !(wfe && *wfe == '0') && p->weight.dtype == OMK_F32 && a.wsk == 1 && a.wsc == a.W && ((uintptr_t)p->weight.data & 15) == 0 &&
                      (!present(p->bias) || (p->bias.dtype == OMK_F32 && ((uintptr_t)p->bias.data & 15) == 0));
      int tlf = wf ? 16 : 64;
      if (const char* e = getenv("OMK_CONV_CL8_TL")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) tlf = v; }   // developer A/B
      const int CVB = (a.C / 4 + 63) / 64, NT4 = (a.L + 4 * tlf - 1) / (4 * tlf);
      dim3 grid((unsigned)((int64_t)a.B * NT4 * CVB)), block(256);
#define CONV_FWD_8F(T_, TL_, W_) do { if (wf) OMK_LAUNCH((conv1d_fwd_cl8_kernel<T_, TL_, W_, TGF, true>), grid, block, 0, stream, a); \
        else OMK_LAUNCH((conv1d_fwd_cl8_kernel<T_, TL_, W_, TGF, false>), grid, block, 0, stream, a); } while (0)
#define CONV_FWD_8W(T_, TL_) do { if (a.W == 4) CONV_FWD_8F(T_, TL_, 4); else if (a.W == 3) CONV_FWD_8F(T_, TL_, 3); else CONV_FWD_8F(T_, TL_, 2); } while (0)
#define CONV_FWD_8(T_) do { if (tlf == 64) CONV_FWD_8W(T_, 64); else if (tlf == 32) CONV_FWD_8W(T_, 32); else CONV_FWD_8W(T_, 16); } while (0)
      if (p->x.dtype == OMK_BF16) CONV_FWD_8(bf16_t); else CONV_FWD_8(f16_t);
#undef CONV_FWD_8F
#undef CONV_FWD_8W
#undef CONV_FWD_8
    } else
    if (p->x.dtype == OMK_BF16) {
      if (tl == 128) CONV_FWD_V(bf16_t, 4, 128, 8); else if (tl == 64) CONV_FWD_V(bf16_t, 4, 64, 8); else if (tl == 16) CONV_FWD_V(bf16_t, 4, 16, 8);
      else CONV_FWD_V(bf16_t, 4, 32, 8);
    } else if (p->x.dtype == OMK_F32) { if (tl == 64) CONV_FWD_V(float, 4, 64, 8); else CONV_FWD_V(float, 4, 32, 8); }
    else CONV_FWD_V(f16_t, 4, 32, 8);
#undef CONV_FWD_V
  } else {
    int64_t n = (int64_t)a.B * a.C * a.L;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    int lf = (a.xsl == 1 && a.xsc != 1) ? 1 : 0;
    OMK_DISPATCH_DTYPE(p->x.dtype, T, OMK_LAUNCH((conv1d_fwd_generic_kernel<T>), grid, block, 0, stream, a, lf));
    if (a.fin) {
      int64_t m = (int64_t)a.B * a.C * a.FW;
      dim3 g2((unsigned)((m + 255) / 256));
      OMK_DISPATCH_DTYPE(p->x.dtype, T, OMK_LAUNCH((conv1d_final_states_kernel<T>), g2, block, 0, stream, a));
    }
  }
  return finish_launch("causal_conv1d_fwd");
}

// partial dw / db rows of conv1d_bwd_cl4_kernel: one per (batch, tile of 4 x 64 tokens), C (W + 1) floats each
static size_t conv_bwd_part_bytes(int B, int C, int L, int W) { return (size_t)B * ((L + 255) / 256) * (size_t)C * (W + 1) * 4; }

extern "C" size_t omk_causal_conv1d_bwd_workspace_bytes(const OmkConv1dBwd* p) {
  if (!p || !present(p->x) || !present(p->weight) || !present(p->dout) || !present(p->dx) || p->x.ndim != 3 || p->weight.ndim != 2) return 0;
  if (p->x.dtype == OMK_F32) return 0;   // (the 16-bit channel-last kernels only; every other layout keeps its atomics and needs nothing)
  const int C = (int)p->x.shape[1];
  if (!(cl_fast_ok(p->x, C) && cl_fast_ok(p->dout, C) && cl_fast_ok(p->dx, C))) return 0;
  return conv_bwd_part_bytes((int)p->x.shape[0], C, (int)p->x.shape[2], (int)p->weight.shape[1]);
}

extern "C" int omk_causal_conv1d_bwd(const OmkConv1dBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->x) && present(p->weight) && present(p->dout) && present(p->dx) && present(p->dweight), "causal_conv1d_bwd: x, weight, dout, dx, dweight required");
  ConvArgs a = {};
  int rc = fill_common(a, p->x, p->weight, p->bias, p->initial_states, "causal_conv1d_bwd");
  if (rc) return rc;
  OMK_REQUIRE(p->dout.dtype == p->x.dtype && p->dx.dtype == p->x.dtype, "causal_conv1d_bwd: dout/dx dtype must equal x dtype");
  OMK_REQUIRE(p->dweight.dtype == OMK_F32 && p->dweight.stride[1] == 1 && p->dweight.stride[0] == a.W, "causal_conv1d_bwd: dweight must be contiguous f32 (C, W)");
  OMK_REQUIRE(!present(p->dbias) || p->dbias.dtype == OMK_F32, "causal_conv1d_bwd: dbias must be f32");
  a.dout = p->dout.data; a.dosb = p->dout.stride[0]; a.dosc = p->dout.stride[1]; a.dosl = p->dout.stride[2];
  a.dx = p->dx.data; a.dxsb = p->dx.stride[0]; a.dxsc = p->dx.stride[1]; a.dxsl = p->dx.stride[2];
  a.dw = (float*)p->dweight.data; a.db = (float*)p->dbias.data; a.silu = p->silu;
  a.dinit = p->dinitial_states.data;
  if (present(p->dinitial_states)) {
    OMK_REQUIRE(present(p->initial_states) && p->dinitial_states.dtype == p->initial_states.dtype, "causal_conv1d_bwd: dinitial_states needs initial_states of the same dtype");
    a.disb = p->dinitial_states.stride[0]; a.disc = p->dinitial_states.stride[1]; a.disl = p->dinitial_states.stride[2];
  }
  if ((int64_t)a.B * a.C * a.L == 0) return OMK_OK;
  const bool fast = p->x.dtype != OMK_F32 && cl_fast_ok(p->x, a.C) && cl_fast_ok(p->dout, a.C) && cl_fast_ok(p->dx, a.C);
  dim3 block(256);
  if (fast) {
    // strips of 64 tokens, 8 tokens in flight: 402 -> 373 us per call on the 1.3B slice against 32 / 4 (`OMK_CONV_BWD_VAR`: halo
    // re-reads of W - 1 rows per strip and twice the loads in flight); 128-token strips gain nothing more
    constexpr int TL = 64;
#define CONV_BWD_V(T_, VEC_, TG_) do { const int CVB = (a.C / VEC_ + 63) / 64, NT4 = (a.L + 4 * TL - 1) / (4 * TL); \
      dim3 grid((unsigned)((int64_t)a.B * NT4 * CVB)); \
      if (a.W == 4) OMK_LAUNCH((conv1d_bwd_cl_kernel<T_, VEC_, TL, 4, TG_>), grid, block, 0, stream, a); \
      else if (a.W == 3) OMK_LAUNCH((conv1d_bwd_cl_kernel<T_, VEC_, TL, 3, TG_>), grid, block, 0, stream, a); \
      else OMK_LAUNCH((conv1d_bwd_cl_kernel<T_, VEC_, TL, 2, TG_>), grid, block, 0, stream, a); } while (0)
    // 2 channels per lane: 86 VGPRs / 5 waves per SIMD; the silu' recompute makes this kernel VALU- and latency-heavy
    // (4 channels: 154 VGPRs, 383 us; 2 channels: ~290 us on the 1.3B shape)
    const char* var = getenv("OMK_CONV_BWD_VAR");   // developer A/B (bf16, W = 4): "<VEC><TL><TG>" digits, e.g. 2648 = VEC 2, TL 64, TG 8
    if (var && *var && p->x.dtype == OMK_BF16 && a.W == 4) {
#define CONV_BWD_X(VEC_, TL_, TG_) do { const int CVB = (a.C / VEC_ + 63) / 64, NT4 = (a.L + 4 * TL_ - 1) / (4 * TL_); \
        dim3 grid((unsigned)((int64_t)a.B * NT4 * CVB)); \
        OMK_LAUNCH((conv1d_bwd_cl_kernel<bf16_t, VEC_, TL_, 4, TG_>), grid, block, 0, stream, a); } while (0)
      const int v = atoi(var);
      if (v == 2324) CONV_BWD_X(2, 32, 4);
      else if (v == 2328) CONV_BWD_X(2, 32, 8);
      else if (v == 2644) CONV_BWD_X(2, 64, 4);
      else if (v == 2648) CONV_BWD_X(2, 64, 8);
      else if (v == 21288) CONV_BWD_X(2, 128, 8);
      else if (v == 4324) CONV_BWD_X(4, 32, 4);
      else if (v == 4644) CONV_BWD_X(4, 64, 4);
      else if (v == 1648) CONV_BWD_X(1, 64, 8);
      else return fail(OMK_EINVAL, "OMK_CONV_BWD_VAR: unknown variant %d", v);
#undef CONV_BWD_X
    } else {
      // the scalar-position kernel (conv1d_bwd_cl4_kernel) when every row offset fits 31 bits (OMK_CONV_BWD_CL4=0: the round-3 kernel)
      const bool cl4 = !(getenv("OMK_CONV_BWD_CL4") && getenv("OMK_CONV_BWD_CL4")[0] == '0');
      const int64_t far = (int64_t)a.L * 2 * (a.xsl > a.dosl ? (a.xsl > a.dxsl ? a.xsl : a.dxsl) : (a.dosl > a.dxsl ? a.dosl : a.dxsl));
      if (cl4 && far < ((int64_t)1 << 31) && a.xsc == 1 && a.dosc == 1 && a.dxsc == 1) {
        // (measured and not kept, profiles/r06_stream_kernels.txt: 8 / 16 strips per workgroup, 32- and 16-token strips, 8 / 16 tokens requested
        // per group -- 189 ... 194 us all, 16 waves per workgroup 220 ... 262)
        const bool wf = p->weight.dtype == OMK_F32 && a.wsk == 1 && a.wsc == a.W && ((uintptr_t)p->weight.data & 15) == 0 &&
                        (!present(p->bias) || (p->bias.dtype == OMK_F32 && ((uintptr_t)p->bias.data & 7) == 0));
        const int CVB = (a.C / 2 + 63) / 64, NTS = (a.L + 4 * TL - 1) / (4 * TL);
        dim3 grid((unsigned)((int64_t)a.B * NTS * CVB)), blk(256);
        const size_t pbytes = conv_bwd_part_bytes(a.B, a.C, a.L, a.W);
        a.part = (p->workspace && p->workspace_bytes >= pbytes && !getenv("OMK_CONV_BWD_ATOMICS")) ? (float*)p->workspace : nullptr;
#define CONV_BWD_4G(T_, W_) do { if (wf) OMK_LAUNCH((conv1d_bwd_cl4_kernel<T_, TL, W_, TGX, 4, true>), grid, blk, 0, stream, a); \
          else OMK_LAUNCH((conv1d_bwd_cl4_kernel<T_, TL, W_, TGX, 4, false>), grid, blk, 0, stream, a); } while (0)
#define CONV_BWD_4(T_) do { if (a.W == 4) CONV_BWD_4G(T_, 4); else if (a.W == 3) CONV_BWD_4G(T_, 3); else CONV_BWD_4G(T_, 2); } while (0)
        if (p->x.dtype == OMK_BF16) CONV_BWD_4(bf16_t); else CONV_BWD_4(f16_t);
        if (a.part) {
          const int64_t ncol = (int64_t)a.C * (a.W + 1);
          OMK_LAUNCH(conv1d_bwd_fold_kernel, dim3((unsigned)((ncol + 63) / 64)), dim3(1024), 0, stream, a.part, a.B * NTS, a.C, a.W, a.dw, a.db);
        }
#undef CONV_BWD_4G
#undef CONV_BWD_4
      } else if (p->x.dtype == OMK_BF16) CONV_BWD_V(bf16_t, 2, 8); else CONV_BWD_V(f16_t, 2, 8);
    }
#undef CONV_BWD_V
  } else {
    int64_t n = (int64_t)a.B * a.C;
    dim3 grid((unsigned)((n + 255) / 256));
    OMK_DISPATCH_DTYPE(p->x.dtype, T, OMK_LAUNCH((conv1d_bwd_generic_kernel<T>), grid, block, 0, stream, a));
  }
  if (a.dinit) {
    int64_t m = (int64_t)a.B * a.C * (a.W - 1);
    dim3 g2((unsigned)((m + 255) / 256));
    OMK_DISPATCH_DTYPE(p->x.dtype, T, OMK_LAUNCH((conv1d_dinit_kernel<T>), g2, block, 0, stream, a));
  }
  return finish_launch("causal_conv1d_bwd");
}

extern "C" int omk_causal_conv1d_update(const OmkConv1dUpdate* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->x) && present(p->conv_state) && present(p->weight) && present(p->out), "causal_conv1d_update: x, conv_state, weight, out required");
  OMK_REQUIRE(p->x.ndim == 3 && p->conv_state.ndim == 3 && p->out.ndim == 3 && p->weight.ndim == 2, "causal_conv1d_update: x/out (B, C, T), conv_state (B, C, S), weight (C, W)");
  ConvUpdArgs a = {};
  a.B = (int)p->x.shape[0]; a.C = (int)p->x.shape[1]; a.T = (int)p->x.shape[2]; a.S = (int)p->conv_state.shape[2]; a.W = (int)p->weight.shape[1];
  OMK_REQUIRE(a.W >= 2 && a.W <= CONV_MAXW && a.S >= a.W - 1, "causal_conv1d_update: W in 2..4 and state_len >= W-1");
  OMK_REQUIRE(p->conv_state.shape[0] == a.B && p->conv_state.shape[1] == a.C && p->weight.shape[0] == a.C, "causal_conv1d_update: shape mismatch");
  OMK_REQUIRE(p->out.dtype == p->x.dtype, "causal_conv1d_update: out dtype");
  a.x = p->x.data; a.state = p->conv_state.data; a.w = p->weight.data; a.bias = p->bias.data; a.out = p->out.data;
  a.xsb = p->x.stride[0]; a.xsc = p->x.stride[1]; a.xsl = p->x.stride[2];
  a.ssb = p->conv_state.stride[0]; a.ssc = p->conv_state.stride[1]; a.ssl = p->conv_state.stride[2];
  a.osb = p->out.stride[0]; a.osc = p->out.stride[1]; a.osl = p->out.stride[2];
  a.wsc = p->weight.stride[0]; a.wsk = p->weight.stride[1];
  a.silu = p->silu; a.xdt = p->x.dtype; a.sdt = p->conv_state.dtype; a.wdt = p->weight.dtype; a.bdt = p->bias.dtype;
  if ((int64_t)a.B * a.C * a.T == 0) return OMK_OK;
  int64_t n = (int64_t)a.B * a.C;
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  OMK_LAUNCH(conv1d_update_kernel, grid, block, 0, stream, a);
  return finish_launch("causal_conv1d_update");
}
