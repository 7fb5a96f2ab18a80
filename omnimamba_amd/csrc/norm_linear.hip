// norm_linear.hip -- the decode-step projections fused with the normalisation in front of them (SURVEY.md section 8 rows
// a2, a3, a9, a10 at one token per sequence):
//
//   n   = rmsnorm(x [+ residual]) * w                         (pre-norm of the block; residual_out = x + residual)
//      or rmsnorm(x * silu(z)) * w  /  rmsnorm(x) * w * silu(z) (the gated norm in front of out_proj, grouped)
//      or x                                                    (no norm weight)
//   out = n W^T [+ bias] [+ scale * (n A^T) B^T]               (LoRA of the task, dropout is off at decode time)
//
// At batch 1 the eager step spends one launch each on add+norm, the A GEMV, the base GEMV and the B addmm (7 + 13 + 19 +
// 9 us per layer measured, against 12 us of pure weight streaming for the 70 MB of in_proj).  Here a workgroup requests
// its first pair of weight rows, forms u = (x + residual | x silu(z)) * w in LDS while those loads fly, and streams its
// rows of W against the UN-normalised u (RMSNorm is a per-row scale: out = rstd * (W u + scale * B (A u))), so the sum of
// squares and the rank-r LoRA vector need no barrier of their own.  wave = a pair of rows at a time, lane = 4 (fp32) or
// 8 (16-bit) consecutive columns per step, 16 loads of 16 bytes in flight per lane, the next pair requested before the
// current one is reduced.  HBM-bound on W: out * in * sizeof(W) bytes per call; measured 23.6 us (in_proj + norm + LoRA,
// 70 MB) and 12.5 us (gated norm + out_proj, 33.5 MB) per layer-step of the 1.3B model, 12.8 us for the bare in_proj GEMV.
#include <type_traits>
#include "omk_common.h"
#include "ssd_tiles.h"

// the weight stream of a decode step: every byte is read once per step by ONE CU -- non-temporal loads (MI355X_MICROARCH.md, nt-weights:
// issued -> landed - 18 %); OMK_NL_NT=0 at compile time for the A/B
#ifndef OMK_NL_NT
#define OMK_NL_NT 1
#endif
#if OMK_NL_NT && !defined(OMK_EMU)
#define OMK_NL_WLOAD(p) __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p))
#else
#define OMK_NL_WLOAD(p) (*reinterpret_cast<const u32x4*>(p))
#endif

namespace omk {

constexpr int NL_MAXR = 16;       // LoRA rank
constexpr int NL_THREADS = 256;

struct NlArgs {
  const void* x; const void* res; const void* z; const void* nw; const void* W; const void* bias; const void* la; const void* lb;
  void* ro; void* out;
  int64_t xs, rs, zs, ros, os, Ws, las, lbs;     // row strides (elements)
  int B, In, Out, R, G, xdt, rdt, rodt, nwdt, bdt, odt, ldt, nbg;   // G = norm groups, nbg = norm_before_gate
  int nbatch;                                                        // templated variant: row batches per wave
  void* cst; const void* ccw; const void* ccb; int64_t csb, csc, csl, ccws; int cc0, cc1, cS, cW, csilu;   // conv tail (see omk.h)
  float eps, scale;
};

// four consecutive elements of a runtime-typed array as floats (8- or 16-byte load)
__device__ __forceinline__ void ld4_rt(const void* p, int64_t idx, int dt, float (&o)[4]) {
  if (dt == OMK_F32) {
    const f32x4 v = *reinterpret_cast<const f32x4*>((const float*)p + idx);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
  } else if (dt == OMK_BF16) {
    load_vec<bf16_t, 4>((const bf16_t*)p + idx, o);
  } else {
    load_vec<f16_t, 4>((const f16_t*)p + idx, o);
  }
}

// One sequence per call (batch 1).  RMSNorm is a per-row scale: with u = (x + residual | x silu(z)) * w the output is
// rstd * (W u + scale * B (A u)) -- so the rows stream against the UN-normalised u as soon as it is in LDS, and the two
// small reductions (sum of squares, the rank-r LoRA vector) ride on the same single barrier.
template <class TW>
__global__ __launch_bounds__(NL_THREADS) void norm_linear_kernel(NlArgs a) {
  OMK_DYN_SMEM(smem);
  float* sn = (float*)smem;                       // [In] u
  float* part = sn + a.In;                        // [waves][NL_MAXR] LoRA partials
  __shared__ float red[NL_THREADS / 64][8];       // per-wave sum of squares per norm group (G <= 8)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int gs = a.In / a.G;
  // ---- the first pair of weight rows is requested before anything else
  constexpr int VEC = 16 / sizeof(TW);
  constexpr int UN = 8;                            // column steps issued together: 64 * VEC * UN columns per sweep
  const TW* W = (const TW*)a.W;
  const int nwaves = gridDim.x * (NL_THREADS / 64);
  int row0 = blockIdx.x * (NL_THREADS / 64) + wave;
  float w0[UN][VEC], w1[UN][VEC];
#define NL_ISSUE(r0_, cb_) do {                                                                                   \
    const int r0c_ = (r0_) < a.Out ? (r0_) : a.Out - 1, r1c_ = (r0_) + nwaves < a.Out ? (r0_) + nwaves : r0c_;     \
    _Pragma("unroll") for (int u = 0; u < UN; u++) {                                                               \
      const int c0_ = (cb_) + u * 64 * VEC < a.In ? (cb_) + u * 64 * VEC : lane * VEC;   /* clamped: unconditional */ \
      load_vec<TW, VEC>(W + (int64_t)r0c_ * a.Ws + c0_, w0[u]);                                                    \
      load_vec<TW, VEC>(W + (int64_t)r1c_ * a.Ws + c0_, w1[u]);                                                    \
    } } while (0)
  NL_ISSUE(row0, lane * VEC);
  // ---- u, its sum of squares and the LoRA partials: a thread owns the 4-column groups tid, tid + 256, ...; every load
  // below is independent of the others and issued before the first use
  constexpr int MAXQ = 8;                           // In <= 8192
  const int nq = a.In / (4 * NL_THREADS);           // the host guarantees In % 1024 == 0
  float v[MAXQ][4], ssq[MAXQ];
#pragma unroll
  for (int k = 0; k < MAXQ; k++) {
    ssq[k] = 0.f;
    if (k < nq) {
      const int c = 4 * (tid + NL_THREADS * k);
      ld4_rt(a.x, c, a.xdt, v[k]);
      float t4[4];
      if (a.res) {
        ld4_rt(a.res, c, a.rdt, t4);
#pragma unroll
        for (int i = 0; i < 4; i++) v[k][i] += t4[i];
      }
      if (a.ro && blockIdx.x == 0)   // residual_out = x (+ residual): also written when there is no incoming residual
#pragma unroll
        for (int i = 0; i < 4; i++) store_rt(a.ro, c + i, a.rodt, v[k][i]);
      float g4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.z) {
        ld4_rt(a.z, c, a.xdt, g4);
#pragma unroll
        for (int i = 0; i < 4; i++) g4[i] = silu_f(g4[i]);
      }
      if (a.nw) ld4_rt(a.nw, c, a.nwdt, t4);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float q = (a.z && !a.nbg) ? v[k][i] * g4[i] : v[k][i];   // the quantity that is normalised
        ssq[k] += q * q;
        v[k][i] = (a.nw ? q * t4[i] : q) * ((a.z && a.nbg) ? g4[i] : 1.f);
      }
      *reinterpret_cast<f32x4*>(&sn[c]) = f32x4{v[k][0], v[k][1], v[k][2], v[k][3]};
    }
  }
  if (a.nw) {
    for (int g = 0; g < a.G; g++) {   // a 4-column group never straddles a norm group (group size % 4 == 0)
      float ss = 0.f;
#pragma unroll
      for (int k = 0; k < MAXQ; k++)
        if (k < nq && (4 * (tid + NL_THREADS * k)) / gs == g) ss += ssq[k];
      ss = wave_sum(ss);
      if (lane == 0) red[wave][g] = ss;
    }
  }
  if (a.la) {
#pragma unroll
    for (int r = 0; r < NL_MAXR; r++) {
      if (r < a.R) {
        float hr = 0.f;
#pragma unroll
        for (int k = 0; k < MAXQ; k++) {
          if (k < nq) {
            float a4[4];
            ld4_rt(a.la, (int64_t)r * a.las + 4 * (tid + NL_THREADS * k), a.ldt, a4);
#pragma unroll
            for (int i = 0; i < 4; i++) hr += v[k][i] * a4[i];
          }
        }
        hr = wave_sum(hr);
        if (lane == 0) part[wave * NL_MAXR + r] = hr;
      }
    }
  }
  block_sync();   // u, the partial sums of squares and the LoRA partials are visible
  float rstd = 1.f;
  if (a.nw) {
    if (a.G == 1) {
      rstd = rsqrtf((red[0][0] + red[1][0] + red[2][0] + red[3][0]) / (float)gs + a.eps);
    } else {      // grouped norm: fold each group's scale into u (one more barrier; not the 1.3B configuration)
      for (int c = tid; c < a.In; c += NL_THREADS) {
        const int g = c / gs;
        sn[c] *= rsqrtf((red[0][g] + red[1][g] + red[2][g] + red[3][g]) / (float)gs + a.eps);
      }
      block_sync();
      // the LoRA partials were formed with the unscaled u: grouped norm + LoRA is rejected on the host
    }
  }
  float hq = 0.f;   // lane q < R holds h[q] = (A u)[q]
  if (a.la && lane < a.R) hq = part[lane] + part[NL_MAXR + lane] + part[2 * NL_MAXR + lane] + part[3 * NL_MAXR + lane];
  // ---- rows of W: every wave walks the row pairs (w, w + nwaves), (w + 2 nwaves, ...) -- balanced to one row -- sweeping a
  // pair with lane = VEC consecutive columns per step.  The next pair's loads are issued as soon as the registers are
  // free, so the wave reductions and stores of one pair overlap the memory latency of the next.
  const int sweep = 64 * VEC * UN;
  int cb = lane * VEC;
  float acc0 = 0.f, acc1 = 0.f;
  while (row0 < a.Out) {
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int c0 = cb + u * 64 * VEC;
      if (c0 < a.In) {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          const float nv = sn[c0 + i];
          acc0 += w0[u][i] * nv;
          acc1 += w1[u][i] * nv;
        }
      }
    }
    const bool row_done = cb - lane * VEC + sweep >= a.In;
    const int prow = row0;
    if (row_done) { row0 += 2 * nwaves; cb = lane * VEC; } else { cb += sweep; }
    OMK_SCHED_FENCE();   // keep the next pair's loads below the products: hoisted, they need a second register set
    if (row0 < a.Out) NL_ISSUE(row0, cb);
    if (row_done) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int row = prow + j * nwaves;
        float vv = wave_sum(j == 0 ? acc0 : acc1);
        if (a.la) {   // + scale * B[row] . h : lane q multiplies its h[q]
          const float bq = (lane < a.R && row < a.Out) ? load_rt(a.lb, (int64_t)row * a.lbs + lane, a.ldt) : 0.f;
          vv += a.scale * wave_sum(bq * hq);
        }
        if (lane == 0 && row < a.Out) {
          vv *= rstd;
          if (a.bias) vv += load_rt(a.bias, row, a.bdt);
          store_rt(a.out, row, a.odt, vv);
        }
      }
      acc0 = 0.f; acc1 = 0.f;
    }
  }
}
#undef NL_ISSUE

// ---------------------------------------------------------------------------------------------------------
// The 1.3B decode configurations take this variant: every dtype and the row length are template parameters, so the
// preamble is ONE batch of unconditional loads (x, residual, gate, norm weight, the LoRA A rows) and the row loop holds
// nothing but weight loads -- no load or store sits under control flow, where the compiler would have to fall back to
// s_waitcnt vmcnt(0) (the generic kernel above drains the memory pipeline 500+ times per call that way, 21 us for the
// 70 MB in_proj against 12.8 us for the bare GEMV).  Results stay in registers (lane s keeps the s-th row of its wave)
// until the loop is done; the LoRA B term, the bias and the store happen once per wave after it.
//   TW: weight / activation / norm-weight / LoRA / output type, TR: residual type, In = 1024 NQ, R <= RMAX (0 or 8)
// element i of a 16-byte weight vector kept raw in registers (converted at the multiply, not at the load: the converted
// copy of sixteen bf16 loads would be 128 registers)
template <class TW> __device__ __forceinline__ float raw_elem(const u32x4& v, int i);
// (the element goes through a scalar first: __builtin_bit_cast applied to a vector-element lvalue reads element 0)
template <> __device__ __forceinline__ float raw_elem<float>(const u32x4& v, int i) { const uint32_t e = v[i]; return __builtin_bit_cast(float, e); }
template <> __device__ __forceinline__ float raw_elem<bf16_t>(const u32x4& v, int i) {
  const uint32_t e = v[i >> 1], b = (i & 1) ? (e & 0xffff0000u) : (e << 16);
  return __builtin_bit_cast(float, b);
}

template <class TW, class TR, int NQ, int RMAX>
__global__ __launch_bounds__(NL_THREADS) void norm_linear_fast_kernel(NlArgs a) {
  constexpr int In = 1024 * NQ;
  constexpr int VEC = 16 / sizeof(TW);
  constexpr int LOADS = 16;                                   // 16-byte loads in flight per lane
  constexpr int STEPS_ROW = In / (64 * VEC);                  // column steps of one row
  constexpr int UNE = STEPS_ROW;                              // a lane holds a whole row slice: one request per row
  constexpr int RW = LOADS / UNE;                             // rows per batch
  static_assert(UNE <= LOADS && RW * UNE == LOADS, "row length");
  OMK_DYN_SMEM(smem);
  float* sn = (float*)smem;                       // [In] u
  float* part = sn + In;                          // [waves][8] LoRA partials
  __shared__ float red[NL_THREADS / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TW* W = (const TW*)a.W;
  const int nwaves = gridDim.x * (NL_THREADS / 64);
  const int wg = blockIdx.x * (NL_THREADS / 64) + wave;
  int row0 = wg;
  u32x4 wr[RW][UNE];
  // row j of the batch that starts at row r0_ (rows past the end re-read the last row: unconditional, never used)
#define NLF_ISSUE_ROW(r0_, j_) do {                                                                  \
    const int rj_ = (r0_) + (j_) * nwaves, rc_ = rj_ < a.Out ? rj_ : a.Out - 1;                        \
    const TW* wp_ = W + (int64_t)rc_ * a.Ws + lane * VEC;                                              \
    _Pragma("unroll") for (int u = 0; u < UNE; u++) wr[j_][u] = OMK_NL_WLOAD(wp_ + u * 64 * VEC); \
  } while (0)
  // ---- preamble: a thread owns the 4-column groups tid, tid + 256, ...
  // Order of the requests (round 5): the activations and the LoRA A rows FIRST, the weight rows behind them.  The load counter returns in
  // order, so with the weights in front (rounds 1 - 4) the norm could not start before all sixteen weight requests of the lane had landed
  // from HBM, and each batch of A rows was one more round trip behind that.
  const bool hasres = a.res != nullptr, hasz = a.z != nullptr;
  const TW* xp = (const TW*)a.x;
  const TR* rp = hasres ? (const TR*)a.res : (const TR*)a.x;      // absent: any valid 16 bytes, the value is not used
  const TW* zp = hasz ? (const TW*)a.z : xp;
  float v[NQ][4], t4[NQ][4], g4[NQ][4], n4[NQ][4];
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const int c = 4 * (tid + NL_THREADS * k);
    load_vec<TW, 4>(xp + c, v[k]);
    load_vec<TR, 4>(rp + (hasres ? c : 0), t4[k]);
    load_vec<TW, 4>(zp + c, g4[k]);
    load_vec<TW, 4>((const TW*)a.nw + c, n4[k]);
  }
  // the LoRA A rows ride on the same round trip when they fit the registers (each later wait would be another one)
  // (all A rows up front: measured slower twice -- round 2 behind the weights 18.6 vs 17.3 us, round 5 in front of them 17.7 / 13.7 us vs
  // 17.5 / 13.2 us for fp32 / bf16; the activations in front of the weights: out_proj 10.2 -> 10.0 us, 6.6 -> 6.2 us.  OMK_NLF_A_UP=1 for the A/B)
#ifndef OMK_NLF_A_UP
#define OMK_NLF_A_UP 0
#endif
  constexpr bool A_UP = OMK_NLF_A_UP && RMAX > 0 && RMAX * NQ <= 16;
  float aup[A_UP ? RMAX : 1][NQ][4];
  if constexpr (A_UP) {
#pragma unroll
    for (int r = 0; r < RMAX; r++)
#pragma unroll
      for (int k = 0; k < NQ; k++)
        load_vec<TW, 4>((const TW*)a.la + (int64_t)(r < a.R ? r : 0) * a.las + 4 * (tid + NL_THREADS * k), aup[r][k]);
  }
  OMK_SCHED_FENCE();
#pragma unroll
  for (int j = 0; j < RW; j++) NLF_ISSUE_ROW(row0, j);
  OMK_SCHED_FENCE();
  float ssq = 0.f;
#pragma unroll
  for (int k = 0; k < NQ; k++) {
    const int c = 4 * (tid + NL_THREADS * k);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      v[k][i] += hasres ? t4[k][i] : 0.f;
      t4[k][i] = v[k][i];                                                      // residual_out
      const float g = hasz ? silu_fast(g4[k][i]) : 1.f;
      const float q = (hasz && !a.nbg) ? v[k][i] * g : v[k][i];                // the quantity that is normalised
      ssq += q * q;
      v[k][i] = q * n4[k][i] * ((hasz && a.nbg) ? g : 1.f);
    }
    *reinterpret_cast<f32x4*>(&sn[c]) = f32x4{v[k][0], v[k][1], v[k][2], v[k][3]};
  }
  if (a.ro && blockIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NQ; k++) store_vec<TR, 4>((TR*)a.ro + 4 * (tid + NL_THREADS * k), t4[k]);
  }
  ssq = wave_sum(ssq);
  if (lane == 0) red[wave] = ssq;
  if (RMAX > 0) {
    const TW* lap = (const TW*)a.la;
#pragma unroll
    for (int rb = 0; rb < RMAX; rb += 4) {       // (not up front: four rows of A at a time bound the registers)
      float a4[A_UP ? 1 : 4][NQ][4];
      if constexpr (!A_UP) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
          for (int k = 0; k < NQ; k++)
            load_vec<TW, 4>(lap + (int64_t)(rb + r < a.R ? rb + r : 0) * a.las + 4 * (tid + NL_THREADS * k), a4[r][k]);
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float hr = 0.f;
#pragma unroll
        for (int k = 0; k < NQ; k++)
#pragma unroll
          for (int i = 0; i < 4; i++) hr += v[k][i] * (A_UP ? aup[A_UP ? rb + r : 0][k][i] : a4[A_UP ? 0 : r][k][i]);
        hr = wave_sum(hr);
        if (lane == 0) part[wave * 8 + rb + r] = rb + r < a.R ? hr : 0.f;
      }
    }
  }
  block_sync();   // u, the sums of squares and the LoRA partials are visible
  const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)In + a.eps);
  // this lane's slice of u, kept in registers for every row (<= 64 values)
  float ur[UNE][VEC];
#pragma unroll
  for (int u = 0; u < UNE; u++)
#pragma unroll
    for (int i = 0; i < VEC; i += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(&sn[u * 64 * VEC + lane * VEC + i]);
      ur[u][i] = t[0]; ur[u][i + 1] = t[1]; ur[u][i + 2] = t[2]; ur[u][i + 3] = t[3];
    }
  // ---- what lane s needs to finish the s-th row of its wave (LoRA B row, bias, conv taps and conv state), requested HERE, ahead of the
  // row loop: as stored (no conversion next to a load) and from clamped, always valid addresses (no branch between two loads).  In the
  // finish itself these were three dependent round trips behind the last weight -- LoRA B row, then the bias, then taps and state under
  // their branches, each an s_waitcnt vmcnt(0): ~ 4 us of the 17.8 us of the 1.3B in_proj step (round 5, same finding as in
  // norm_linear_mfma_kernel below).
  auto rawld = [](const TW* p_) -> uint32_t {
    if constexpr (sizeof(TW) == 4) return *reinterpret_cast<const uint32_t*>(p_);
    else return *reinterpret_cast<const uint16_t*>(p_);
  };
  auto rawf = [](uint32_t r_) -> float {   // (16-bit: by a shift -- a truncation becomes an AND the compiler hoists up to the load, a wait)
    if constexpr (sizeof(TW) == 4) return __builtin_bit_cast(float, r_);
    else if constexpr (std::is_same<TW, bf16_t>::value) return __builtin_bit_cast(float, r_ << 16);
    else return to_f32(__builtin_bit_cast(TW, (uint16_t)r_));
  };
  const int frow_ = wg + lane * nwaves, frow = frow_ < a.Out ? frow_ : a.Out - 1;
  const bool conv_on = a.cst != nullptr;
  uint32_t q_lb[RMAX > 0 ? RMAX : 1], q_wt[4], q_hist[3], q_cb, q_bias = 0u;
  {
    if constexpr (RMAX > 0) {
      const TW* lbp = (const TW*)a.lb + (int64_t)frow * a.lbs;
#pragma unroll
      for (int r = 0; r < RMAX; r++) q_lb[r] = rawld(lbp + (r < a.R ? r : 0));      // (ranks >= R: their h is zero)
    } else {
      q_lb[0] = 0u;
    }
    if (a.bias) q_bias = rawld((const TW*)a.bias + frow);                           // (uniform; the projections of the model have none)
    const int convC = a.cc1 - a.cc0, c_ = frow - a.cc0, ch = c_ < 0 ? 0 : (c_ < convC ? c_ : convC - 1);
    const TW* wr_ = conv_on ? (const TW*)a.ccw + (int64_t)ch * a.ccws : W;
    const TW* cs = conv_on ? (const TW*)a.cst + (int64_t)ch * a.csc : W;
    const int64_t csl = conv_on ? a.csl : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int col = k - (4 - a.cW); q_wt[k] = rawld(wr_ + (col >= 0 ? col : 0)); }
    q_cb = rawld(a.ccb ? (const TW*)a.ccb + ch : wr_);
#pragma unroll
    for (int k = 0; k < 3; k++) { const int sl = a.cS - 3 + k; q_hist[k] = rawld(cs + (int64_t)(sl >= 0 ? sl : 0) * csl); }
  }
  OMK_SCHED_FENCE();
  // ---- rows: a batch = RW rows; as soon as a row's products are done its registers take the same row of the next
  // batch, so (RW - 1) / RW of the requests stay in flight through the reductions.  The trip count is uniform (the
  // host sizes the grid for nbatch full batches per wave) and the loads are unconditional: the compiler counts them.
  float keep = 0.f;
  for (int b = 0; b < a.nbatch; b++) {
    const int rnext = row0 + RW * nwaves;
#pragma unroll
    for (int j = 0; j < RW; j++) {
      float acc = 0.f;
#pragma unroll
      for (int u = 0; u < UNE; u++)
#pragma unroll
        for (int i = 0; i < VEC; i++) acc += raw_elem<TW>(wr[j][u], i) * ur[u][i];
      OMK_SCHED_FENCE();   // the refill stays below the products (hoisted, it would need a second register set)
      NLF_ISSUE_ROW(rnext, j);
      OMK_SCHED_FENCE();
      const float vv = wave_sum(acc);
      keep = lane == b * RW + j ? vv : keep;
    }
    row0 = rnext;
  }
  const int slot = a.nbatch * RW;
#undef NLF_ISSUE_ROW
  // ---- lane s finishes the s-th row of this wave: LoRA B term, norm scale, bias, conv tail, store
  const int row = wg + lane * nwaves;
  if (lane < slot && row < a.Out) {
    float vv = keep;
    if (RMAX > 0) {
      float d = 0.f;
#pragma unroll
      for (int r = 0; r < RMAX; r++) {
        const float h = part[r] + part[8 + r] + part[16 + r] + part[24 + r];     // zero beyond R
        d += rawf(q_lb[r]) * h;
      }
      vv += a.scale * d;
    }
    vv *= rstd;
    if (a.bias) vv += rawf(q_bias);
    if (conv_on && row >= a.cc0 && row < a.cc1) {
      // this row is a new xBC input: causal_conv1d_update for its channel, right here (no other lane touches it)
      const int ch = row - a.cc0;
      TW* cs = (TW*)a.cst + (int64_t)ch * a.csc;
      float hist[3], wt[4];
#pragma unroll
      for (int k = 0; k < 3; k++) hist[k] = rawf(q_hist[k]);
#pragma unroll
      for (int k = 0; k < 4; k++) wt[k] = k - (4 - a.cW) >= 0 ? rawf(q_wt[k]) : 0.f;
      const TW xr_ = from_f32<TW>(vv);
      const float xin = to_f32(xr_);                          // the value upstream would have stored in zxbcdt
      float cv = (a.ccb ? rawf(q_cb) : 0.f) + wt[0] * hist[0] + wt[1] * hist[1] + wt[2] * hist[2] + wt[3] * xin;
      for (int sl = 0; sl + 1 < a.cS; sl++) {                 // roll: slot sl <- slot sl + 1 = hist[sl + 1 - (S - 3)] (the stored value moves as it is)
        const int k = sl + 1 - (a.cS - 3);
        const uint32_t qv = k == 0 ? q_hist[0] : (k == 1 ? q_hist[1] : q_hist[2]);
        if constexpr (sizeof(TW) == 4) cs[(int64_t)sl * a.csl] = __builtin_bit_cast(TW, qv);
        else cs[(int64_t)sl * a.csl] = __builtin_bit_cast(TW, (uint16_t)qv);
      }
      cs[(int64_t)(a.cS - 1) * a.csl] = xr_;
      vv = a.csilu ? silu_f(cv) : cv;
    }
    ((TW*)a.out)[row] = from_f32<TW>(vv);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Two to eight sequences per step: the same stream of W, every 16-byte weight vector multiplied against the NB
// activation rows while it sits in registers (the separate ops re-stream W per GEMM and per LoRA factor and spend 11
// launches: 4.6 ms per step at batch 8 against 1.8 ms at batch 1).  Differences to the batch-1 kernel: u lives in LDS
// ([NB][In], fp32 for fp32 weights, bf16 -- upstream's own rounding point -- for bf16 weights) and is re-read per batch
// of rows; the preamble walks the sequences in groups that fit the registers; NB accumulators per row.
template <class TW> struct lds_u { using type = float; };
template <> struct lds_u<bf16_t> { using type = bf16_t; };

// GATE: the second input is the gate z (dtype TW) instead of a residual (dtype TR) -- never both in this kernel.
template <class TW, class TR, int NQ, int RMAX, int NB, bool GATE>
__global__ __launch_bounds__(NL_THREADS, 2) void norm_linear_batched_kernel(NlArgs a) {   // two waves per SIMD: <= 256 VGPRs
  using TU = typename lds_u<TW>::type;
  constexpr int In = 1024 * NQ;
  constexpr int VEC = 16 / sizeof(TW);
  constexpr int LOADS = 16;
  constexpr int UNE = In / (64 * VEC);
  constexpr int RW = LOADS / UNE;
  constexpr int CB = (8 / NQ) < NB ? (8 / NQ) : NB;           // sequences per preamble group
  static_assert(UNE <= LOADS && RW * UNE == LOADS && NB % CB == 0, "row length");
  OMK_DYN_SMEM(smem);
  TU* sn = (TU*)smem;                                         // [NB][In] u
  float* part = (float*)(smem + (size_t)NB * In * sizeof(TU));   // [waves][NB][8] LoRA partials
  float* red = part + (NL_THREADS / 64) * NB * 8;             // [waves][NB] sums of squares
  float* res = red + (NL_THREADS / 64) * NB;                  // [waves][64 row slots][NB] row results
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const TW* W = (const TW*)a.W;
  const int nwaves = gridDim.x * (NL_THREADS / 64);
  const int wg = blockIdx.x * (NL_THREADS / 64) + wave;
  int row0 = wg;
  u32x4 wr[RW][UNE];
#define NLB_ISSUE(r0_) do {                                                                          \
    _Pragma("unroll") for (int j = 0; j < RW; j++) {                                                   \
      const int rj_ = (r0_) + j * nwaves, rc_ = rj_ < a.Out ? rj_ : a.Out - 1;                         \
      const TW* wp_ = W + (int64_t)rc_ * a.Ws + lane * VEC;                                            \
      _Pragma("unroll") for (int u = 0; u < UNE; u++) wr[j][u] = OMK_NL_WLOAD(wp_ + u * 64 * VEC); \
    } } while (0)
  NLB_ISSUE(row0);
  // ---- preamble, CB sequences at a time (sequences past B repeat the last one; nothing of theirs is stored).  Every
  // wait on a load costs a round trip (and the first one also waits for the weight rows above, in-order counter), so the
  // LoRA A rows are requested once, up front, and the next group's activations before the current group is reduced.
  using TA = typename std::conditional<GATE, TW, TR>::type;       // second input: gate or residual
  const bool hasaux = GATE ? true : a.res != nullptr;
  float n4[NQ][4];
#pragma unroll
  for (int k = 0; k < NQ; k++) load_vec<TW, 4>((const TW*)a.nw + 4 * (tid + NL_THREADS * k), n4[k]);
  constexpr bool A_UP = RMAX > 0 && RMAX * NQ <= 16;             // 64 registers at most
  float aup[A_UP ? RMAX : 1][NQ][4];
  if constexpr (A_UP) {
#pragma unroll
    for (int r = 0; r < RMAX; r++)
#pragma unroll
      for (int k = 0; k < NQ; k++)
        load_vec<TW, 4>((const TW*)a.la + (int64_t)(r < a.R ? r : 0) * a.las + 4 * (tid + NL_THREADS * k), aup[r][k]);
  }
  constexpr int NG = NB / CB;
  constexpr bool PF = RMAX == 0 && NG > 1;                        // prefetch the next group (registers allow it without LoRA)
  float v[PF ? 2 : 1][CB][NQ][4], t4[PF ? 2 : 1][CB][NQ][4];
  auto issue = [&](int gi, int buf) {
#pragma unroll
    for (int bb = 0; bb < CB; bb++) {
      const int b = gi * CB + bb < a.B ? gi * CB + bb : a.B - 1;
      const TW* xp = (const TW*)a.x + (int64_t)b * a.xs;
      const TA* ap = GATE ? (const TA*)a.z + (int64_t)b * a.zs : (hasaux ? (const TA*)a.res + (int64_t)b * a.rs : (const TA*)xp);
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int c = 4 * (tid + NL_THREADS * k);
        load_vec<TW, 4>(xp + c, v[buf][bb][k]);
        load_vec<TA, 4>(ap + (hasaux ? c : 0), t4[buf][bb][k]);
      }
    }
  };
  issue(0, 0);
#pragma unroll
  for (int gi = 0; gi < NG; gi++) {
    const int cur = PF ? (gi & 1) : 0, b0 = gi * CB;
    if constexpr (PF) { if (gi + 1 < NG) issue(gi + 1, cur ^ 1); }
#pragma unroll
    for (int bb = 0; bb < CB; bb++) {
      float ssq = 0.f;
#pragma unroll
      for (int k = 0; k < NQ; k++) {
        const int c = 4 * (tid + NL_THREADS * k);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          float x_ = v[cur][bb][k][i], g = 1.f;
          if constexpr (GATE) g = silu_fast(t4[cur][bb][k][i]);
          else { x_ += hasaux ? t4[cur][bb][k][i] : 0.f; t4[cur][bb][k][i] = x_; }   // t4 becomes residual_out
          const float q = (GATE && !a.nbg) ? x_ * g : x_;
          ssq += q * q;
          v[cur][bb][k][i] = q * n4[k][i] * ((GATE && a.nbg) ? g : 1.f);
        }
        store_vec<TU, 4>(sn + (size_t)(b0 + bb) * In + c, v[cur][bb][k]);
        if (sizeof(TU) == 2) {   // the LoRA input is the same rounded u the rows multiply
#pragma unroll
          for (int i = 0; i < 4; i++) v[cur][bb][k][i] = to_f32(from_f32<TU>(v[cur][bb][k][i]));
        }
      }
      if constexpr (!GATE) {
        if (a.ro && blockIdx.x == 0 && b0 + bb < a.B) {
#pragma unroll
          for (int k = 0; k < NQ; k++) store_vec<TR, 4>((TR*)a.ro + (int64_t)(b0 + bb) * a.ros + 4 * (tid + NL_THREADS * k), t4[cur][bb][k]);
        }
      }
      ssq = wave_sum(ssq);
      if (lane == 0) red[wave * NB + b0 + bb] = ssq;
    }
    if constexpr (RMAX > 0) {
#pragma unroll
      for (int rb = 0; rb < RMAX; rb += 4) {
        float a4[A_UP ? 1 : 4][NQ][4];
        if constexpr (!A_UP) {
#pragma unroll
          for (int r = 0; r < 4; r++)
#pragma unroll
            for (int k = 0; k < NQ; k++)
              load_vec<TW, 4>((const TW*)a.la + (int64_t)(rb + r < a.R ? rb + r : 0) * a.las + 4 * (tid + NL_THREADS * k), a4[r][k]);
        }
        float hr[CB * 4];                                       // index bb * 4 + r: one butterfly instead of 4 CB wave sums
#pragma unroll
        for (int bb = 0; bb < CB; bb++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            float h = 0.f;
#pragma unroll
            for (int k = 0; k < NQ; k++)
#pragma unroll
              for (int i = 0; i < 4; i++) h += v[cur][bb][k][i] * (A_UP ? aup[A_UP ? rb + r : 0][k][i] : a4[A_UP ? 0 : r][k][i]);
            hr[bb * 4 + r] = rb + r < a.R ? h : 0.f;
          }
        wave_multi_sum<CB * 4>(hr);                             // lane L: value index L >> (6 - log2(4 CB))
        if ((lane & (64 / (CB * 4) - 1)) == 0) {
          const int idx = lane / (64 / (CB * 4));
          part[(wave * NB + b0 + idx / 4) * 8 + rb + idx % 4] = hr[0];
        }
      }
    }
    if constexpr (!PF) { if (gi + 1 < NG) issue(gi + 1, 0); }
  }
  block_sync();
  // ---- rows: every weight vector meets the NB activation rows while it is in registers
  for (int bi = 0; bi < a.nbatch; bi++) {
    float acc[RW][NB];
#pragma unroll
    for (int j = 0; j < RW; j++)
#pragma unroll
      for (int b = 0; b < NB; b++) acc[j][b] = 0.f;
    // all NB reads of a column step are issued before the first multiply: one LDS round trip per step, not one per
    // sequence (a single wave per SIMD cannot hide them: 128 serialised reads were most of the out_proj call)
#pragma unroll
    for (int u = 0; u < UNE; u++) {
      using URaw = typename std::conditional<sizeof(TU) == 4, f32x4, u32x4>::type;   // 16 bytes: VEC elements of TU
      URaw uraw[NB];
#pragma unroll
      for (int b = 0; b < NB; b++)
        uraw[b] = *reinterpret_cast<const URaw*>(sn + (size_t)b * In + u * 64 * VEC + lane * VEC);
      OMK_SCHED_FENCE();
#pragma unroll
      for (int b = 0; b < NB; b++)
#pragma unroll
        for (int j = 0; j < RW; j++)
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            float uvi;
            if constexpr (sizeof(TU) == 4) uvi = uraw[b][i];
            else uvi = raw_elem<bf16_t>(uraw[b], i);
            acc[j][b] += raw_elem<TW>(wr[j][u], i) * uvi;
          }
    }
    row0 += RW * nwaves;
    OMK_SCHED_FENCE();
    NLB_ISSUE(row0);
    OMK_SCHED_FENCE();
    // RW * NB totals with one butterfly (15 + 2 shuffles for 16 values instead of 16 DPP scans and their wait states);
    // lane L ends up with the value of index L >> (6 - log2(RW NB)) = j * NB + b and parks it in LDS for the row's lane
    float tot[RW * NB];
#pragma unroll
    for (int j = 0; j < RW; j++)
#pragma unroll
      for (int b = 0; b < NB; b++) tot[j * NB + b] = acc[j][b];
    wave_multi_sum<RW * NB>(tot);
    if ((lane & (64 / (RW * NB) - 1)) == 0) {
      const int idx = lane / (64 / (RW * NB));
      res[(wave * 64 + bi * RW + idx / NB) * NB + idx % NB] = tot[0];
    }
  }
#undef NLB_ISSUE
  block_sync();   // the row results were parked by other lanes
  const int slot = a.nbatch * RW;
  const int row = wg + lane * nwaves;
  if (lane < slot && row < a.Out) {
    float lbv[RMAX > 0 ? RMAX : 1];
    if (RMAX > 0) {
      const TW* lbp = (const TW*)a.lb + (int64_t)row * a.lbs;
#pragma unroll
      for (int r = 0; r < RMAX; r++) lbv[r] = to_f32(lbp[r < a.R ? r : 0]);
    }
    const float bias = a.bias ? to_f32(((const TW*)a.bias)[row]) : 0.f;
    const bool isconv = a.cst && row >= a.cc0 && row < a.cc1;
    float wt[4] = {0.f, 0.f, 0.f, 0.f}, cbias = 0.f;
    const int ch = row - a.cc0;
    if (isconv) {
      const TW* wr_ = (const TW*)a.ccw + (int64_t)ch * a.ccws;
#pragma unroll
      for (int k = 0; k < 4; k++) { const int col = k - (4 - a.cW); wt[k] = col >= 0 ? to_f32(wr_[col >= 0 ? col : 0]) : 0.f; }
      cbias = a.ccb ? to_f32(((const TW*)a.ccb)[ch]) : 0.f;
    }
    // every load of the tail first (sequences past B clamp to the last one), then the arithmetic, stores last: with the
    // loads inside a per-sequence branch each sequence costs its own round trip
    float hist[NB][3], vout[NB], xin[NB];
    if (isconv) {
#pragma unroll
      for (int b = 0; b < NB; b++) {
        const TW* cs = (const TW*)a.cst + (int64_t)(b < a.B ? b : a.B - 1) * a.csb + (int64_t)ch * a.csc;
#pragma unroll
        for (int k = 0; k < 3; k++) { const int sl = a.cS - 3 + k; hist[b][k] = to_f32(cs[(int64_t)(sl >= 0 ? sl : 0) * a.csl]); }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
      float vv = res[(wave * 64 + lane) * NB + b];
      if (RMAX > 0) {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
          float h = 0.f;
#pragma unroll
          for (int w2 = 0; w2 < NL_THREADS / 64; w2++) h += part[(w2 * NB + b) * 8 + r];
          d += lbv[r] * h;
        }
        vv += a.scale * d;
      }
      float ss = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NL_THREADS / 64; w2++) ss += red[w2 * NB + b];
      vv = vv * rsqrtf(ss / (float)In + a.eps) + bias;
      xin[b] = to_f32(from_f32<TW>(vv));
      if (isconv) {
        const float cv = cbias + wt[0] * hist[b][0] + wt[1] * hist[b][1] + wt[2] * hist[b][2] + wt[3] * xin[b];
        vv = a.csilu ? silu_f(cv) : cv;
      }
      vout[b] = vv;
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
      if (b < a.B) {
        if (isconv) {
          TW* cs = (TW*)a.cst + (int64_t)b * a.csb + (int64_t)ch * a.csc;
          for (int sl = 0; sl + 1 < a.cS; sl++) {
            const int k = sl + 1 - (a.cS - 3);
            cs[(int64_t)sl * a.csl] = from_f32<TW>(k == 0 ? hist[b][0] : (k == 1 ? hist[b][1] : hist[b][2]));
          }
          cs[(int64_t)(a.cS - 1) * a.csl] = from_f32<TW>(xin[b]);
        }
        ((TW*)a.out)[(int64_t)b * a.os + row] = from_f32<TW>(vout[b]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Two to eight sequences, 16-bit weights: the rows on the MATRIX pipe (round 5).
//
// The vector form above spends ~130 issue slots per 16-byte weight vector at eight sequences (8 x 8 multiply-adds + the unpacking of both
// operands): 33 us for the 35 MB of a bf16 in_proj -- the same as for the 70 MB of the fp32 one, not memory bound.  Here the unit of work is
// a TILE of 16 output rows; the waves of a workgroup take a slice of the input features each, and a 16-byte weight vector is an operand as
// it lands: lane (i = lane & 15, g = lane >> 4) loads the 8 features [32 s + 8 g, + 8) of row i of the tile, which IS the A fragment of
// v_mfma_f32_16x16x32_bf16; the B fragment is the same features of sequence lane & 15 from the bf16 u in LDS (one ds_read_b128; u rows 16
// bytes apart in bank space).  One matrix instruction per KB of weights instead of ~130 vector instructions; exact fp32 products and sums
// either way.  D[r] of a lane = row 4 g + r of the tile, sequence i.
//   * ONE workgroup per CU walks its tiles (blockIdx.x, + gridDim.x, ...): the preamble -- u = norm input of every sequence, 64 values per
//     thread at eight sequences -- runs once per CU, not once per tile (a workgroup per tile: 21 us for the in_proj of the 1.3B model, a third
//     of it 532 preambles on 256 CUs);
//   * two tiles of weights are in flight per workgroup at any time (two register sets, all loads of a wave's slice at once), together with
//     the operands thread (row i, sequence b) needs to finish its element: LoRA B row, bias, conv taps and conv state;
//   * x is requested FIRST (the load counter returns in order: a wait for x must not be a wait for the weights);
//   * the LoRA vector h = A u is one more 8-row tile on the same u, once per workgroup (A from L2, requested when the preamble's registers
//     are free);
//   * the waves' partial tiles meet in LDS (two buffers: one barrier per tile); thread (i, b) adds them up: LoRA, rstd, bias, conv tail.
// NT = 512 threads (eight feature slices) for 2048 and 4096 input features, 256 for 1024.
// ROWS = 16 or 8 rows per tile: with 8 the lanes i >= 8 of a fragment repeat the rows of the lanes i - 8 (same addresses: no more HBM
// traffic, twice the matrix instructions per byte, which are free here) -- twice as many units of work, for matrices with few tiles per CU
// (out_proj of the 1.3B model: 128 tiles of 16 rows on 256 CUs; in_proj: 532 tiles = 2 or 3 per workgroup).
template <class TW, class TR, int NQ, int RMAX, int NB, bool GATE, int NT, int ROWS>
__global__ __launch_bounds__(NT, 1) void norm_linear_mfma_kernel(NlArgs a) {
  using TU = typename lds_u<TW>::type;
  constexpr int In = 1024 * NQ, VEC = 16 / sizeof(TW), NWV = NT / 64;
  constexpr int NQT = In / (4 * NT);                                        // 4-element pieces per thread and sequence
  constexpr int US = In + 16 / (int)sizeof(TU);
  constexpr int KL = 4 * VEC, KPW = In / NWV, NLD = KPW / KL;               // features per wave load, per wave; loads per lane and tile
  constexpr bool W32 = sizeof(TW) == 4;                                     // fp32 weights: four v_mfma_f32_16x16x4_f32 per 16-byte vector
  static_assert(NQT >= 1 && NQT * 4 * NT == In && 16 * NB <= NT && NLD <= 16 && RMAX <= 8 && (ROWS == 8 || ROWS == 16), "shape");
  OMK_DYN_SMEM(smem);
  TU* sn = (TU*)smem;                                            // [NB][US] u
  float* part = (float*)(smem + (size_t)NB * US * sizeof(TU));   // [waves][NB][8] LoRA partials
  float* red = part + NWV * NB * 8;                              // [waves][NB] sums of squares
  float* hs = red + NWV * NB;                                    // [NB][8] LoRA vector, [NB] rstd
  float* rstd = hs + NB * 8;
  float* res = rstd + NB;                                        // [2][waves][16 rows][NB] partial tiles
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t16 = lane & 15, g16 = lane >> 4;
  using TA = typename std::conditional<GATE, TW, TR>::type;      // second input: gate or residual
  const bool hasaux = GATE ? true : a.res != nullptr;
  // the inputs of EVERY sequence are requested here, ahead of everything else and as stored (16-bit values stay packed: 48 - 64 registers
  // at eight sequences): a request issued behind the weights returns behind them (in-order counter) -- a second group of sequences
  // fetched inside the preamble made it wait for both tiles of weights to land, + 2.4 us at eight sequences
  constexpr int AW = sizeof(TA) == 4 ? 4 : 2;
  constexpr int XW = W32 ? 4 : 2;
  uint32_t xr[NB][NQT][XW], ar[NB][NQT][AW];
#pragma unroll
  for (int bb = 0; bb < NB; bb++) {
    const int b = bb < a.B ? bb : a.B - 1;
    const TW* xp = (const TW*)a.x + (int64_t)b * a.xs;
    const TA* ap = GATE ? (const TA*)a.z + (int64_t)b * a.zs : (hasaux ? (const TA*)a.res + (int64_t)b * a.rs : (const TA*)xp);
#pragma unroll
    for (int k = 0; k < NQT; k++) {
      const int c = 4 * (tid + NT * k);
      if constexpr (W32) { const u32x4 q = *reinterpret_cast<const u32x4*>(xp + c); xr[bb][k][0] = q[0]; xr[bb][k][1] = q[1]; xr[bb][k][2] = q[2]; xr[bb][k][3] = q[3]; }
      else { const u32x2 q = *reinterpret_cast<const u32x2*>(xp + c); xr[bb][k][0] = q[0]; xr[bb][k][1] = q[1]; }
      if constexpr (AW == 4) { const u32x4 r = *reinterpret_cast<const u32x4*>(ap + (hasaux ? c : 0)); ar[bb][k][0] = r[0]; ar[bb][k][1] = r[1]; ar[bb][k][2] = r[2]; ar[bb][k][3] = r[3]; }
      else { const u32x2 r = *reinterpret_cast<const u32x2*>(ap + (hasaux ? c : 0)); ar[bb][k][0] = r[0]; ar[bb][k][1] = r[1]; }
    }
  }
  float n4[NQT][4];
#pragma unroll
  for (int k = 0; k < NQT; k++) load_vec<TW, 4>((const TW*)a.nw + 4 * (tid + NT * k), n4[k]);
  OMK_SCHED_FENCE();
  // ---- tiles of this workgroup; what travels with a tile: the wave's slice of its 16 rows, the finish operands of thread (i, b)
  const int ntile = (a.Out + ROWS - 1) / ROWS, gstep = (int)gridDim.x;
  const int fi = tid & (ROWS - 1), fb = tid / ROWS;
  constexpr int LBR = RMAX > 0 ? (W32 ? 8 : 4) : 1;
  static_assert(RMAX == 0 || RMAX == 8, "LoRA B rows of 16 bytes");
  // (raw 16-bit values from clamped, always valid addresses: no branch and no conversion between two loads -- with the selects and
  // conversions next to the loads every one of the ~12 small loads of a thread became its own branch + s_waitcnt vmcnt(0), ~4 us per call)
  struct Fin { uint32_t lbq[LBR]; uint32_t hist[3], wt[4], cbias, bias; };   // (one register per value: 16-bit members get packed -- a wait)
  const bool conv_on = a.cst != nullptr;
  const int convC = a.cc1 - a.cc0;
  auto load_w = [&](u32x4 (&w)[NLD], int t) {
    const int rt = ROWS * t + (t16 & (ROWS - 1));
    const TW* wp = (const TW*)a.W + (int64_t)(rt < a.Out ? rt : a.Out - 1) * a.Ws + wave * KPW + VEC * g16;
#pragma unroll
    for (int s = 0; s < NLD; s++) w[s] = OMK_NL_WLOAD(wp + KL * s);
  };
  auto raw16 = [](const TW* p) -> uint32_t {
    if constexpr (sizeof(TW) == 4) return *reinterpret_cast<const uint32_t*>(p);
    else return *reinterpret_cast<const uint16_t*>(p);
  };
  auto load_fin = [&](Fin& f, int t) {   // (no branch in here: without the conv tail the taps and the state read a valid dummy address)
    const int fr = ROWS * t + fi, frow = fr < a.Out ? fr : a.Out - 1, fbc = fb < a.B ? fb : a.B - 1;
    if constexpr (RMAX > 0) {   // the LoRA B row as stored, RMAX values = 16 / 32 bytes (the launcher checks R == RMAX and the alignment)
      const u32x4* lq = reinterpret_cast<const u32x4*>((const TW*)a.lb + (int64_t)frow * a.lbs);
#pragma unroll
      for (int j = 0; j < LBR / 4; j++) { const u32x4 q = lq[j]; f.lbq[4 * j] = q[0]; f.lbq[4 * j + 1] = q[1]; f.lbq[4 * j + 2] = q[2]; f.lbq[4 * j + 3] = q[3]; }
    } else {
      f.lbq[0] = 0u;
    }
    f.bias = a.bias ? raw16((const TW*)a.bias + frow) : 0u;   // (the projections of the model have none: a uniform branch)
    const int c_ = frow - a.cc0, ch = c_ < 0 ? 0 : (c_ < convC ? c_ : convC - 1);
    const TW* wr_ = conv_on ? (const TW*)a.ccw + (int64_t)ch * a.ccws : (const TW*)a.W;
    const TW* cs = conv_on ? (const TW*)a.cst + (int64_t)fbc * a.csb + (int64_t)ch * a.csc : (const TW*)a.W;
    const int64_t csl = conv_on ? a.csl : 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int col = k - (4 - a.cW); f.wt[k] = raw16(wr_ + (col >= 0 ? col : 0)); }
    f.cbias = raw16(a.ccb ? (const TW*)a.ccb + ch : wr_);
#pragma unroll
    for (int k = 0; k < 3; k++) { const int sl = a.cS - 3 + k; f.hist[k] = raw16(cs + (int64_t)(sl >= 0 ? sl : 0) * csl); }
  };
  // (order of the requests = order of their return: everything small in front of the weights, so that neither the preamble nor the LoRA
  // tile nor the finish of the first tile waits for the second tile's weights to land)
  u32x4 wA[NLD], wB[NLD];
  Fin fA, fB;
  const int tfirst = (int)blockIdx.x;
  load_fin(fA, tfirst);
  if (tfirst + gstep < ntile) load_fin(fB, tfirst + gstep);
  // LoRA A rows: eight loads per lane at a time (NLD = 16: two halves, each requested when the registers of the one before are free)
  constexpr int NLH = NLD < 8 ? NLD : 8;
  u32x4 wl[RMAX > 0 ? NLH : 1];
  constexpr bool LORA_EARLY = RMAX > 0 && NLD <= 8 && !W32;   // (next to the preamble's registers only when everything fits 256)
  auto load_lora = [&](int s0) {
    const int r8 = t16 & 7;
    const TW* lap = (const TW*)a.la + (int64_t)(r8 < a.R ? r8 : 0) * a.las + wave * KPW + VEC * g16;
#pragma unroll
    for (int s = 0; s < NLH; s++) wl[s] = *reinterpret_cast<const u32x4*>(lap + KL * (s0 + s));
  };
  if constexpr (LORA_EARLY) load_lora(0);
  OMK_SCHED_FENCE();
  load_w(wA, tfirst);
  if (tfirst + gstep < ntile) load_w(wB, tfirst + gstep);
  OMK_SCHED_FENCE();
  // ---- preamble: u = (x + residual | x silu(z)) * w of every sequence into LDS, sums of squares per wave
  auto un16 = [](uint32_t r, int i) -> float {   // element i (0 / 1) of a packed pair of TW
    if constexpr (std::is_same<TW, bf16_t>::value) return __builtin_bit_cast(float, i ? (r & 0xffff0000u) : (r << 16));
    else if constexpr (sizeof(TW) == 4) return __builtin_bit_cast(float, r);     // (not used: fp32 values are not packed)
    else return to_f32(__builtin_bit_cast(TW, (uint16_t)(i ? r >> 16 : r)));
  };
#pragma unroll
  for (int bb = 0; bb < NB; bb++) {
    float ssq = 0.f;
#pragma unroll
    for (int k = 0; k < NQT; k++) {
      const int c = 4 * (tid + NT * k);
      float vv4[4], t4[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float x_, g = 1.f, aux;
        if constexpr (W32) x_ = __builtin_bit_cast(float, xr[bb][k][i]);
        else x_ = un16(xr[bb][k][i >> 1], i & 1);
        if constexpr (AW == 4) aux = __builtin_bit_cast(float, ar[bb][k][i]);
        else aux = un16(ar[bb][k][i >> 1], i & 1);
        if constexpr (GATE) g = silu_fast(aux);
        else { x_ += hasaux ? aux : 0.f; t4[i] = x_; }                 // t4: residual_out
        const float q = (GATE && !a.nbg) ? x_ * g : x_;
        ssq += q * q;
        vv4[i] = q * n4[k][i] * ((GATE && a.nbg) ? g : 1.f);
      }
      store_vec<TU, 4>(sn + (size_t)bb * US + c, vv4);
      if constexpr (!GATE) {
        if (a.ro && blockIdx.x == 0 && bb < a.B) store_vec<TR, 4>((TR*)a.ro + (int64_t)bb * a.ros + c, t4);
      }
    }
    ssq = wave_sum(ssq);
    if (lane == 0) red[wave * NB + bb] = ssq;
  }
  if constexpr (RMAX > 0 && !LORA_EARLY) load_lora(0);
  block_sync();
  const TU* up = sn + (size_t)(t16 & (NB - 1)) * US + wave * KPW + VEC * g16;
  // one landed 16-byte vector of a row against the same features of the sequences: one bf16 matrix instruction, or -- fp32 -- element e of the
  // vector in instruction e on both operands (feature 4 g + e of the step).  (The elements through a bit cast of the VECTOR:
  // __builtin_bit_cast(float, wv[e]) reads element 0 for every e with this compiler.)
  auto mma = [&](const u32x4& wv, int s_, f32x4 (&ac)[2]) {
    if constexpr (W32) {
      const f32x4 uv = *reinterpret_cast<const f32x4*>(up + KL * s_), wf = __builtin_bit_cast(f32x4, wv);
#pragma unroll
      for (int e = 0; e < 4; e++) ac[e & 1] = mfma16x16x4_f32(wf[e], uv[e], ac[e & 1]);
    } else {
      ac[s_ & 1] = mfma16x16x32_bf16(as_s16x8(wv), as_s16x8(ld16(up + KL * s_)), ac[s_ & 1]);
    }
  };
  if constexpr (RMAX > 0) {   // h[rank][sequence] of this wave's slice: ranks 4 g + r (g < 2), sequence lane & 15
    f32x4 hacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s0 = 0; s0 < NLD; s0 += NLH) {
      if (s0 > 0) load_lora(s0);
#pragma unroll
      for (int s = 0; s < NLH; s++) mma(wl[s], s0 + s, hacc);
    }
    if (t16 < NB && g16 < 2) {
#pragma unroll
      for (int r = 0; r < 4; r++) part[(wave * NB + t16) * 8 + 4 * g16 + r] = 4 * g16 + r < a.R ? hacc[0][r] + hacc[1][r] : 0.f;
    }
    block_sync();
  }
  // the sums over the waves once per workgroup, not once per finishing thread and tile: h[b][r] and rstd[b] (visible behind the barrier of
  // the first tile)
  if (RMAX > 0 && tid < 8 * NB) {
    float h = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NWV; w2++) h += part[w2 * NB * 8 + tid];
    hs[tid] = h;
  }
  if (tid >= NT - NB) {
    const int b = tid - (NT - NB);
    float ss = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NWV; w2++) ss += red[w2 * NB + b];
    rstd[b] = rsqrtf(ss / (float)In + a.eps);
  }
  // one tile: products of the landed slice, the slice and the finish operands of the tile two steps on requested into the freed
  // registers, partial tile to LDS, barrier, finish
  auto tile_step = [&](u32x4 (&w)[NLD], Fin& f, int t, int par) {
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int s = 0; s < NLD; s++) mma(w[s], s, acc);
    const Fin fc = f;
    const int tn = t + 2 * gstep;
    if (tn < ntile) { load_w(w, tn); load_fin(f, tn); }
    float* rs = res + par * (NWV * ROWS * NB);
    if (t16 < NB && 4 * g16 < ROWS) {
#pragma unroll
      for (int r = 0; r < 4; r++) rs[(wave * ROWS + 4 * g16 + r) * NB + t16] = acc[0][r] + acc[1][r];
    }
    block_sync();
    const int frow = ROWS * t + fi;
    if (fb < NB && fb < a.B && frow < a.Out) {
      // (bf16 from the raw register by a shift: a truncation to 16 bits here becomes an AND the compiler hoists up to the load -- a wait)
      auto f16 = [](uint32_t r) -> float {
        if constexpr (sizeof(TW) == 4) return __builtin_bit_cast(float, r);
        else if constexpr (std::is_same<TW, bf16_t>::value) return __builtin_bit_cast(float, r << 16);
        else return to_f32(__builtin_bit_cast(TW, (uint16_t)r));
      };
      float vv = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < NWV; w2++) vv += rs[(w2 * ROWS + fi) * NB + fb];
      if constexpr (RMAX > 0) {
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < RMAX; r++)
          d += (W32 ? f16(fc.lbq[W32 ? r : 0]) : ((r & 1) ? f16(fc.lbq[r >> 1] >> 16) : f16(fc.lbq[r >> 1]))) * hs[fb * 8 + r];   // (ranks >= R: h is zero)
        vv += a.scale * d;
      }
      vv = vv * rstd[fb] + (a.bias ? f16(fc.bias) : 0.f);
      const TW xr = from_f32<TW>(vv);
      const float xin = to_f32(xr);
      if (conv_on && frow >= a.cc0 && frow < a.cc1) {
        float wt[4];
#pragma unroll
        for (int k = 0; k < 4; k++) wt[k] = k - (4 - a.cW) >= 0 ? f16(fc.wt[k]) : 0.f;
        const float cv = (a.ccb ? f16(fc.cbias) : 0.f) + wt[0] * f16(fc.hist[0]) + wt[1] * f16(fc.hist[1]) + wt[2] * f16(fc.hist[2]) + wt[3] * xin;
        vv = a.csilu ? silu_f(cv) : cv;
        TW* cs = (TW*)a.cst + (int64_t)fb * a.csb + (int64_t)(frow - a.cc0) * a.csc;
        for (int sl = 0; sl + 1 < a.cS; sl++) {      // roll: the stored values move as they are
          const int k = sl + 1 - (a.cS - 3);
          const uint32_t hq = k == 0 ? fc.hist[0] : (k == 1 ? fc.hist[1] : fc.hist[2]);
          if constexpr (W32) cs[(int64_t)sl * a.csl] = __builtin_bit_cast(TW, hq);
          else cs[(int64_t)sl * a.csl] = __builtin_bit_cast(TW, (uint16_t)hq);
        }
        cs[(int64_t)(a.cS - 1) * a.csl] = xr;
      }
      ((TW*)a.out)[(int64_t)fb * a.os + frow] = from_f32<TW>(vv);
    }
  };
  for (int t = tfirst; t < ntile; t += 2 * gstep) {
    tile_step(wA, fA, t, 0);
    if (t + gstep < ntile) tile_step(wB, fB, t + gstep, 1);
  }
}

// compute units of the current device (cached; 256 on the MI355X and under the emulator)
static int cu_count() {
#ifdef OMK_EMU
  return 256;
#else
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n;
#endif
}

}  // namespace omk

using namespace omk;

extern "C" int omk_norm_linear(const OmkNormLinear* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->x) && present(p->weight) && present(p->out), "norm_linear: x, weight, out required");
  OMK_REQUIRE(p->x.ndim == 2 && p->out.ndim == 2 && p->weight.ndim == 2, "norm_linear: x (B, in), weight (out, in), out (B, out)");
  NlArgs a = {};
  a.B = (int)p->x.shape[0]; a.In = (int)p->x.shape[1]; a.Out = (int)p->weight.shape[0];
  OMK_REQUIRE(p->weight.shape[1] == a.In && p->out.shape[0] == a.B && p->out.shape[1] == a.Out, "norm_linear: shape mismatch");
  OMK_REQUIRE(p->x.stride[1] == 1 && p->out.stride[1] == 1 && p->weight.stride[1] == 1, "norm_linear: last dims must be contiguous");
  if (a.B == 0 || a.Out == 0) return OMK_OK;
  // up to eight sequences per call (norm_linear_batched_kernel); more than that goes to the separate ops
  if (a.B > 8) return fail(OMK_EUNSUPPORTED, "norm_linear: batch %d > 8 is served by the unfused ops", a.B);
  const int wdt = p->weight.dtype;
  const int vec = wdt == OMK_F32 ? 4 : 8;
  if (a.In % 1024 != 0 || a.In > 8192 || !aligned16(p->weight) || p->weight.stride[0] % vec != 0)
    return fail(OMK_EUNSUPPORTED, "norm_linear: in_features %d must be a multiple of 1024 (<= 8192) with 16-byte aligned weight rows", a.In);
  {
    auto ok4 = [](const OmkTensor& t) { return !present(t) || (aligned16(t) && (t.ndim < 2 || t.stride[0] % 4 == 0)); };
    if (!ok4(p->x) || !ok4(p->residual) || !ok4(p->z) || !ok4(p->norm_weight) || !ok4(p->lora_a) || !ok4(p->residual_out))
      return fail(OMK_EUNSUPPORTED, "norm_linear: activation / LoRA rows must be 16-byte aligned");
  }
  a.G = 1;
  if (present(p->norm_weight)) {
    OMK_REQUIRE(numel(p->norm_weight) == a.In && is_contig_last(p->norm_weight), "norm_linear: norm_weight must be contiguous (in)");
    const int64_t gsz = p->group_size > 0 ? p->group_size : a.In;
    OMK_REQUIRE(a.In % gsz == 0 && gsz % 4 == 0, "norm_linear: group_size must divide in_features and be a multiple of 4");
    a.G = (int)(a.In / gsz);
  }
  if (present(p->residual)) OMK_REQUIRE(p->residual.ndim == 2 && p->residual.shape[0] == a.B && p->residual.shape[1] == a.In && p->residual.stride[1] == 1, "norm_linear: residual mismatch");
  if (present(p->residual_out)) OMK_REQUIRE(p->residual_out.ndim == 2 && p->residual_out.shape[0] == a.B && p->residual_out.shape[1] == a.In && p->residual_out.stride[1] == 1, "norm_linear: residual_out mismatch");
  if (present(p->z)) OMK_REQUIRE(p->z.ndim == 2 && p->z.shape[0] == a.B && p->z.shape[1] == a.In && p->z.stride[1] == 1 && p->z.dtype == p->x.dtype, "norm_linear: z mismatch");
  if (present(p->bias)) OMK_REQUIRE(numel(p->bias) == a.Out && is_contig_last(p->bias), "norm_linear: bias must be contiguous (out)");
  a.R = 0;
  if (present(p->lora_a) || present(p->lora_b)) {
    OMK_REQUIRE(present(p->lora_a) && present(p->lora_b) && p->lora_a.ndim == 2 && p->lora_b.ndim == 2, "norm_linear: lora_a (r, in) and lora_b (out, r) come together");
    a.R = (int)p->lora_a.shape[0];
    OMK_REQUIRE(a.R <= NL_MAXR && p->lora_a.shape[1] == a.In && p->lora_b.shape[0] == a.Out && p->lora_b.shape[1] == a.R, "norm_linear: LoRA shapes");
    OMK_REQUIRE(p->lora_a.stride[1] == 1 && p->lora_b.stride[1] == 1 && p->lora_a.dtype == p->lora_b.dtype, "norm_linear: LoRA factors must be row-major of one dtype");
    a.la = p->lora_a.data; a.lb = p->lora_b.data; a.las = p->lora_a.stride[0]; a.lbs = p->lora_b.stride[0]; a.ldt = p->lora_a.dtype;
  }
  a.x = p->x.data; a.res = p->residual.data; a.z = p->z.data; a.nw = p->norm_weight.data; a.W = p->weight.data; a.bias = p->bias.data;
  a.ro = p->residual_out.data; a.out = p->out.data;
  a.xs = p->x.stride[0]; a.rs = present(p->residual) ? p->residual.stride[0] : 0; a.zs = present(p->z) ? p->z.stride[0] : 0;
  a.ros = present(p->residual_out) ? p->residual_out.stride[0] : 0; a.os = p->out.stride[0]; a.Ws = p->weight.stride[0];
  a.xdt = p->x.dtype; a.rdt = p->residual.dtype; a.rodt = p->residual_out.dtype; a.nwdt = p->norm_weight.dtype; a.bdt = p->bias.dtype; a.odt = p->out.dtype;
  a.nbg = p->norm_before_gate; a.eps = p->eps; a.scale = p->lora_scale;
  const size_t smem = ((size_t)a.In + (size_t)NL_MAXR * (NL_THREADS / 64)) * 4;
  if (a.G > 8 || (a.G > 1 && a.R > 0)) return fail(OMK_EUNSUPPORTED, "norm_linear: more than 8 norm groups, or grouped norm together with LoRA");
  // the templated variant: one dtype for x, z, norm weight, weight, LoRA, bias and out; fp32 or that dtype for the residual
  {
    const int xdt = p->x.dtype;
    auto same = [&](const OmkTensor& t) { return !present(t) || t.dtype == wdt; };
    const int nq = a.In / 1024;
    const bool resok = (!present(p->residual) || p->residual.dtype == OMK_F32 || p->residual.dtype == wdt) &&
                       (!present(p->residual_out) || !present(p->residual) || p->residual_out.dtype == p->residual.dtype);
    const int trdt = present(p->residual) ? p->residual.dtype : (present(p->residual_out) ? p->residual_out.dtype : wdt);
    const bool fast = (wdt == OMK_F32 || wdt == OMK_BF16) && xdt == wdt && same(p->z) && same(p->norm_weight) && same(p->lora_a) &&
                      same(p->bias) && p->out.dtype == wdt && present(p->norm_weight) && a.G == 1 && a.R <= 8 && resok &&
                      (trdt == OMK_F32 || trdt == wdt) && a.In == 1024 * nq && (nq == 1 || nq == 2 || nq == 4) &&
                      (!present(p->lora_b) || p->lora_b.stride[1] == 1) && !getenv("OMK_NORM_LINEAR_GENERIC");
    if (present(p->conv_state)) {
      OMK_REQUIRE(present(p->conv_weight) && p->conv_state.ndim == 3 && p->conv_weight.ndim == 2, "norm_linear: conv_state (B, C, S) needs conv_weight (C, W)");
      const int64_t Cc = p->conv_state.shape[1];
      a.cS = (int)p->conv_state.shape[2]; a.cW = (int)p->conv_weight.shape[1];
      OMK_REQUIRE(p->conv_state.shape[0] == a.B, "norm_linear: conv_state batch");
      OMK_REQUIRE(p->conv_weight.shape[0] == Cc && p->conv_offset >= 0 && p->conv_offset + Cc <= a.Out, "norm_linear: conv channels must be rows [conv_offset, conv_offset + C) of the output");
      const bool cok = fast && a.cW >= 2 && a.cW <= 4 && a.cS >= a.cW - 1 && a.cS <= 4 && p->conv_state.dtype == wdt && p->conv_weight.dtype == wdt &&
                       p->conv_weight.stride[1] == 1 && (!present(p->conv_bias) || (p->conv_bias.dtype == wdt && is_contig_last(p->conv_bias) && numel(p->conv_bias) == Cc));
      if (!cok) return fail(OMK_EUNSUPPORTED, "norm_linear: the conv tail needs the uniform-dtype kernel (W 2..4, state length W-1..4)");
      a.cst = p->conv_state.data; a.ccw = p->conv_weight.data; a.ccb = p->conv_bias.data;
      a.csb = p->conv_state.stride[0]; a.csc = p->conv_state.stride[1]; a.csl = p->conv_state.stride[2]; a.ccws = p->conv_weight.stride[0];
      a.cc0 = (int)p->conv_offset; a.cc1 = (int)(p->conv_offset + Cc); a.csilu = p->conv_silu;
    }
    if (fast && a.B > 1) {
      // two to eight sequences: u for all of them in LDS; one workgroup per CU when that takes more than half of it
      const int nb = a.B <= 2 ? 2 : (a.B <= 4 ? 4 : 8);
      const int vecw = wdt == OMK_F32 ? 4 : 8, steps_row = a.In / (64 * vecw), rw = 16 / steps_row;
      const size_t bsmem = (size_t)nb * a.In * (wdt == OMK_F32 ? 4 : 2) + (size_t)(NL_THREADS / 64) * nb * (9 + 64) * 4;
      // (these two limits belong to the VECTOR form below; the matrix-pipe form walks its tiles persistently and sizes its own LDS -- the
      // checks are applied behind its dispatch: advisor finding, round 5)
      const bool vec_lds_ok = bsmem <= 150 * 1024;
      const int wg_per_cu = bsmem <= 76 * 1024 ? 2 : 1;
      const int maxw = wg_per_cu * cu_count() * (NL_THREADS / 64);
      const int k = (a.Out + rw * maxw - 1) / (rw * maxw);
      const int nw_ = (a.Out + rw * k - 1) / (rw * k);
      a.nbatch = k;
      const bool vec_rows_ok = k * rw <= 64;
      dim3 bgrid((unsigned)((nw_ + NL_THREADS / 64 - 1) / (NL_THREADS / 64))), bblock(NL_THREADS);
      const bool gate = present(p->z);
      if (gate && present(p->residual)) return fail(OMK_EUNSUPPORTED, "norm_linear: residual and gate together are served by the batch-1 kernel only");
      // 16-bit weights: the matrix-pipe form, a workgroup per tile of 16 rows (OMK_NL_MFMA=0: the vector form, for the A/B)
      const bool use_mfma = !(getenv("OMK_NL_MFMA") && atoi(getenv("OMK_NL_MFMA")) == 0);   // (developer switches are read per call, all of them)
      // B rows of the LoRA read as 16-byte loads
      // (both matrices: load_lora reads the A rows as 16-byte vectors too -- advisor finding, round 5)
      const bool lora_rows16 = a.R == 0 || (a.R == 8 && (a.lbs * (wdt == OMK_F32 ? 4 : 2)) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.lb) & 15) == 0 &&
                                            (a.las * (wdt == OMK_F32 ? 4 : 2)) % 16 == 0 && (reinterpret_cast<uintptr_t>(a.la) & 15) == 0);
      // fp32 weights (four v_mfma_f32_16x16x4_f32 per 16-byte vector): only where it was measured ahead of the vector form -- eight sequences
      // with LoRA, rows of up to 2048 features: 29.7 against 31.5 us for the 1.3B in_proj; behind it at two sequences (28.8 / 19.5 us) and
      // without LoRA (23.9 / 19.7 us) -- profiles/r05_decode_projections.txt.  (4096 features: 32 loads per lane and tile, no room for two tiles.)
      const bool mfma_f32 = !(getenv("OMK_NL_MFMA_F32") && atoi(getenv("OMK_NL_MFMA_F32")) == 0);
      const bool f32_ok = wdt == OMK_F32 && nq <= 2 && mfma_f32 && ((nb == 8 && a.R > 0) || (getenv("OMK_NL_MFMA_F32") && atoi(getenv("OMK_NL_MFMA_F32")) == 2));
      if (use_mfma && lora_rows16 && (wdt == OMK_BF16 || f32_ok)) {
        // tiles of 8 rows when there are fewer 16-row tiles than workgroups (out_proj of the 1.3B model: 11.1 -> 10.1 us at eight sequences,
        // 8.4 -> 7.3 us at two; with several tiles per workgroup 8 rows are behind: in_proj 16.2 -> 17.7 us).  OMK_NL_MFMA_ROWS = 8 / 16 for the A/B
        int wgs = cu_count();
        if (const char* e = getenv("OMK_NL_MFMA_WGS")) wgs = atoi(e) > 0 ? atoi(e) : wgs;   // tests: several tiles per workgroup on small matrices
        int rows = (a.Out + 15) / 16 < wgs ? 8 : 16;
        if (const char* e = getenv("OMK_NL_MFMA_ROWS")) rows = atoi(e) == 8 ? 8 : (atoi(e) == 16 ? 16 : rows);
        const int nwv = nq == 1 ? 4 : 8, ntile = (a.Out + rows - 1) / rows, ub = wdt == OMK_F32 ? 4 : 2;
        const size_t msmem = (size_t)nb * (a.In + 16 / ub) * ub + (size_t)nwv * nb * (9 + 2 * 16) * 4 + (size_t)nb * 9 * 4;
        dim3 mgrid((unsigned)(ntile < wgs ? ntile : wgs));
#define NLM_G(TW_, TR_, NQ_, RM_, NB_, G_) do { constexpr int NT_ = NQ_ == 1 ? 256 : 512; \
          if (rows == 8) { \
            if (OMK_SET_MAX_DYN_SMEM((norm_linear_mfma_kernel<TW_, TR_, NQ_, RM_, NB_, G_, NT_, 8>), msmem)) return fail(OMK_ELAUNCH, "norm_linear: cannot raise dynamic LDS to %zu", msmem); \
            OMK_LAUNCH((norm_linear_mfma_kernel<TW_, TR_, NQ_, RM_, NB_, G_, NT_, 8>), mgrid, dim3(NT_), msmem, stream, a); \
          } else { \
            if (OMK_SET_MAX_DYN_SMEM((norm_linear_mfma_kernel<TW_, TR_, NQ_, RM_, NB_, G_, NT_, 16>), msmem)) return fail(OMK_ELAUNCH, "norm_linear: cannot raise dynamic LDS to %zu", msmem); \
            OMK_LAUNCH((norm_linear_mfma_kernel<TW_, TR_, NQ_, RM_, NB_, G_, NT_, 16>), mgrid, dim3(NT_), msmem, stream, a); } } while (0)
#define NLM_GO(TW_, TR_, NQ_, RM_, NB_) do { if (gate) NLM_G(TW_, TR_, NQ_, RM_, NB_, true); else NLM_G(TW_, TR_, NQ_, RM_, NB_, false); } while (0)
#define NLM_B(TW_, TR_, NQ_, RM_) do { if (nb == 2) NLM_GO(TW_, TR_, NQ_, RM_, 2); else if (nb == 4) NLM_GO(TW_, TR_, NQ_, RM_, 4); else NLM_GO(TW_, TR_, NQ_, RM_, 8); } while (0)
#define NLM_R(TW_, TR_, NQ_) do { if (a.R > 0) NLM_B(TW_, TR_, NQ_, 8); else NLM_B(TW_, TR_, NQ_, 0); } while (0)
#define NLM_Q(TW_, TR_) do { if (nq == 1) NLM_R(TW_, TR_, 1); else if (nq == 2) NLM_R(TW_, TR_, 2); else NLM_R(TW_, TR_, 4); } while (0)
#define NLM_Q2(TW_, TR_) do { if (nq == 1) NLM_R(TW_, TR_, 1); else NLM_R(TW_, TR_, 2); } while (0)
        if (wdt == OMK_F32) NLM_Q2(float, float); else if (trdt == OMK_F32) NLM_Q(bf16_t, float); else NLM_Q(bf16_t, bf16_t);
#undef NLM_Q2
#undef NLM_Q
#undef NLM_R
#undef NLM_B
#undef NLM_GO
#undef NLM_G
        return finish_launch("norm_linear");
      }
      if (!vec_lds_ok) return fail(OMK_EUNSUPPORTED, "norm_linear: %d sequences x %d features do not fit the LDS", a.B, a.In);
      if (!vec_rows_ok) return fail(OMK_EUNSUPPORTED, "norm_linear: %d output rows are too many for the batched kernel", a.Out);
#define NLB_G(TW_, TR_, NQ_, RM_, NB_, G_) do { \
        if (OMK_SET_MAX_DYN_SMEM((norm_linear_batched_kernel<TW_, TR_, NQ_, RM_, NB_, G_>), bsmem)) return fail(OMK_ELAUNCH, "norm_linear: cannot raise dynamic LDS to %zu", bsmem); \
        OMK_LAUNCH((norm_linear_batched_kernel<TW_, TR_, NQ_, RM_, NB_, G_>), bgrid, bblock, bsmem, stream, a); } while (0)
#define NLB_GO(TW_, TR_, NQ_, RM_, NB_) do { if (gate) NLB_G(TW_, TR_, NQ_, RM_, NB_, true); else NLB_G(TW_, TR_, NQ_, RM_, NB_, false); } while (0)
#define NLB_B(TW_, TR_, NQ_, RM_) do { if (nb == 2) NLB_GO(TW_, TR_, NQ_, RM_, 2); else if (nb == 4) NLB_GO(TW_, TR_, NQ_, RM_, 4); else NLB_GO(TW_, TR_, NQ_, RM_, 8); } while (0)
#define NLB_R(TW_, TR_, NQ_) do { if (a.R > 0) NLB_B(TW_, TR_, NQ_, 8); else NLB_B(TW_, TR_, NQ_, 0); } while (0)
#define NLB_Q(TW_, TR_) do { if (nq == 1) NLB_R(TW_, TR_, 1); else if (nq == 2) NLB_R(TW_, TR_, 2); else NLB_R(TW_, TR_, 4); } while (0)
      if (wdt == OMK_F32) NLB_Q(float, float);
      else if (trdt == OMK_F32) NLB_Q(bf16_t, float);
      else NLB_Q(bf16_t, bf16_t);
#undef NLB_Q
#undef NLB_R
#undef NLB_B
#undef NLB_GO
#undef NLB_G
      return finish_launch("norm_linear");
    }
    if (a.B > 1) return fail(OMK_EUNSUPPORTED, "norm_linear: batch %d needs the uniform-dtype kernel (one dtype, in_features 1024 / 2048 / 4096)", a.B);
    if (fast) {
      const int vecw = wdt == OMK_F32 ? 4 : 8;
      const int steps_row = a.In / (64 * vecw), rw = 16 / steps_row;   // rows per batch (16 loads of 16 bytes per lane)
      // waves: every wave takes k full batches of rw rows (k as small as two workgroups per CU allow)
      const int wpc = getenv("OMK_NLF_WPC") ? atoi(getenv("OMK_NLF_WPC")) : 2;   // workgroups per CU the grid is sized for
      const int maxw = (wpc > 0 ? wpc : 2) * cu_count() * (NL_THREADS / 64);
      const int k = (a.Out + rw * maxw - 1) / (rw * maxw);
      const int nw_ = (a.Out + rw * k - 1) / (rw * k);
      a.nbatch = k;
      if (k * rw <= 64) {
        dim3 fgrid((unsigned)((nw_ + NL_THREADS / 64 - 1) / (NL_THREADS / 64))), fblock(NL_THREADS);
        const size_t fsmem = ((size_t)a.In + 8 * (NL_THREADS / 64)) * 4;
#define NLF_GO(TW_, TR_, NQ_, RM_) do { \
          if (OMK_SET_MAX_DYN_SMEM((norm_linear_fast_kernel<TW_, TR_, NQ_, RM_>), fsmem)) return fail(OMK_ELAUNCH, "norm_linear: cannot raise dynamic LDS to %zu", fsmem); \
          OMK_LAUNCH((norm_linear_fast_kernel<TW_, TR_, NQ_, RM_>), fgrid, fblock, fsmem, stream, a); } while (0)
#define NLF_R(TW_, TR_, NQ_) do { if (a.R > 0) NLF_GO(TW_, TR_, NQ_, 8); else NLF_GO(TW_, TR_, NQ_, 0); } while (0)
#define NLF_Q(TW_, TR_) do { if (nq == 1) NLF_R(TW_, TR_, 1); else if (nq == 2) NLF_R(TW_, TR_, 2); else NLF_R(TW_, TR_, 4); } while (0)
        if (wdt == OMK_F32) NLF_Q(float, float);
        else if (trdt == OMK_F32) NLF_Q(bf16_t, float);
        else NLF_Q(bf16_t, bf16_t);
#undef NLF_Q
#undef NLF_R
#undef NLF_GO
        return finish_launch("norm_linear");
      }
    }
  }
  if (present(p->conv_state)) return fail(OMK_EUNSUPPORTED, "norm_linear: the conv tail needs the uniform-dtype kernel");
  // two workgroups per CU; small matrices get one wave per row pair
  const int ncu = 2 * cu_count();
  const int want = (a.Out + 7) / 8;   // workgroups if every wave took exactly one row pair
  dim3 grid((unsigned)(want < ncu ? want : ncu)), block(NL_THREADS);
#define NL_GO(TW_, NB_) do { if (OMK_SET_MAX_DYN_SMEM((norm_linear_kernel<TW_>), smem)) return fail(OMK_ELAUNCH, "norm_linear: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((norm_linear_kernel<TW_>), grid, block, smem, stream, a); } while (0)
#define NL_NB(TW_) NL_GO(TW_, 1)
  if (wdt == OMK_F32) NL_NB(float); else if (wdt == OMK_BF16) NL_NB(bf16_t); else NL_NB(f16_t);
#undef NL_NB
#undef NL_GO
  return finish_launch("norm_linear");
}
