// ssd.hip -- Mamba-2 SSD scan: C-ABI entry points, dt preparation, the shape-generic fp32 scan kernel and the
// backward finishing pass.  The fast MFMA chunked kernel lives in ssd_mfma.hip; both consume GScan (ssd_scan.h).
//
// Forward  = 1 scan (GS_Y).  Backward = 3 scans (GS_DC, GS_DX, GS_DB) + a per-(b,h) reverse prefix over tokens
// that turns the two per-token scalars they emit into d(dt) / dA (see tests/test_bwd_derivation.py):
//   e_t = dy_t . y~_t            (from GS_DC:  sum_n C_t[n] O_t[n])
//   w_t = <g_t, x_t (x) B_t>     (from GS_DB:  sum_n B_t[n] O_t[n])
//   dl_t = e_t - dt'_t w_t + dl_{t+1} ;  d(dt')_t = w_t + A_h dl_t ;  dA_h = sum_t dt'_t dl_t
#include <cstdlib>

#include "ssd_scan.h"

namespace omk {

// ---------------------------------------------------------------------------------------------------------
// dt' = clamp(softplus(dt + bias)) -> (B, H, L) f32 ; dsoft = d dt' / d dt
// ---------------------------------------------------------------------------------------------------------
struct DtPrepArgs {
  const void* dt; const void* bias; float* dtp; float* dsoft;
  int64_t sb, sl, sh; int B, L, H, dt_dt, bias_dt, softplus; float lo, hi;
  float* zero[3]; int nzero[3];   // backward: the small accumulators (dA, dD, d dt_bias) cleared by workgroup 0 -- three 4 us launches less
};
// block = 32 tokens x 32 heads of one batch element: loads follow the unit-stride head dimension of (B, L, H), the stores
// the unit-stride token dimension of (B, H, L); the tile turns in LDS (a direct per-element mapping writes 4 bytes per
// lane into H different rows: 26 us instead of 6 for the 1.3B shape)
__global__ __launch_bounds__(256) void ssd_dt_prep_kernel(DtPrepArgs a) {
  __shared__ float sv[32][33], sd[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const int nth = (a.H + 31) / 32, ntl = (a.L + 31) / 32;
  const int hb = blockIdx.x % nth, lb = (blockIdx.x / nth) % ntl, b = blockIdx.x / (nth * ntl);
  const int h = hb * 32 + tx;
  if (blockIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++)
      for (int i = threadIdx.x; i < a.nzero[k]; i += 256) a.zero[k][i] = 0.f;
  }
  const float bias = (a.bias && h < a.H) ? load_rt(a.bias, h, a.bias_dt) : 0.f;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int t = lb * 32 + ty + 8 * j;
    float v = 0.f, d = 1.f;
    if (t < a.L && h < a.H) {
      v = load_rt(a.dt, (int64_t)b * a.sb + (int64_t)t * a.sl + (int64_t)h * a.sh, a.dt_dt) + bias;
      if (a.softplus && v <= 20.f) {
        // softplus and its derivative from ONE hardware exp2: e = exp(v), u = 1 + e; log1p(e) = log(u) e / (u - 1) repairs the
        // rounding of 1 + e (classic compensation), sigmoid(v) = e / u.  The libm forms (expf, log1pf, a second expf) were ~150
        // instructions per element: the 2 M elements of the 1.3B shape kept this launch VALU bound at 17 us.
        const float e = exp2_fast(v * LOG2E), u = 1.f + e;
        d = e * rcp_fast(u);
        v = u == 1.f ? e : log2_fast(u) * 0.6931471805599453f * e * rcp_fast(u - 1.f);
      }
      if (v < a.lo) { v = a.lo; d = 0.f; }
      if (v > a.hi) { v = a.hi; d = 0.f; }
    }
    sv[ty + 8 * j][tx] = v;
    sd[ty + 8 * j][tx] = d;
  }
  block_sync();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int hh = hb * 32 + ty + 8 * j, t = lb * 32 + tx;
    if (hh < a.H && t < a.L) {
      const int64_t o = ((int64_t)b * a.H + hh) * a.L + t;
      a.dtp[o] = sv[tx][ty + 8 * j];
      if (a.dsoft) a.dsoft[o] = sd[tx][ty + 8 * j];
    }
  }
}

// The same for 16-bit dt with unit head stride (the block's zxbcdt view): 64 tokens x 64 heads per block, two adjacent heads per
// 4-byte load, the (B, H, L) rows stored as 16-byte vectors of four tokens.
template <class T, int TOK>   // TOK tokens x 64 heads per block
__global__ __launch_bounds__(256) void ssd_dt_prep_vec_kernel(DtPrepArgs a) {
  __shared__ float sv[64][TOK + 1], sd[64][TOK + 1];
  const int tid = threadIdx.x;
  const int nth = (a.H + 63) / 64, ntl = (a.L + TOK - 1) / TOK;
  const int hb = blockIdx.x % nth, lb = (blockIdx.x / nth) % ntl, b = blockIdx.x / (nth * ntl);
  if (blockIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++)
      for (int i = tid; i < a.nzero[k]; i += 256) a.zero[k][i] = 0.f;
  }
  const int hp = tid & 31, tr = tid >> 5;
  const int h = hb * 64 + 2 * hp;
  const bool hok = h < a.H;   // H is even: both heads of the pair are in or out
  float bias[2] = {0.f, 0.f};
  if (a.bias && hok) { bias[0] = load_rt(a.bias, h, a.bias_dt); bias[1] = load_rt(a.bias, h + 1, a.bias_dt); }
  const T* src = (const T*)a.dt + (int64_t)b * a.sb + (hok ? h : 0);
  // all eight loads of the thread first, without control flow (rows behind the end read the last row and are masked below): with
  // the load under `if (t < L)` the eight round trips to memory ran one after the other (13.5 us per launch; round 4)
  float raw[TOK / 8][2];
#pragma unroll
  for (int j = 0; j < TOK / 8; j++) {
    const int t = lb * TOK + 8 * j + tr;
    load_vec<T, 2>(src + (int64_t)(t < a.L ? t : a.L - 1) * a.sl, raw[j]);
  }
  const bool want_d = a.dsoft != nullptr;
#pragma unroll
  for (int j = 0; j < TOK / 8; j++) {
    const int tl = 8 * j + tr, t = lb * TOK + tl;
    float v2[2] = {0.f, 0.f}, d2[2] = {1.f, 1.f};
    if (t < a.L && hok) {
#pragma unroll
      for (int e = 0; e < 2; e++) {
        float v = raw[j][e] + bias[e], d = 1.f;
        if (a.softplus && v <= 20.f) {
          const float ex = exp2_fast(v * LOG2E), u = 1.f + ex;
          d = ex * rcp_fast(u);
          v = u == 1.f ? ex : log2_fast(u) * 0.6931471805599453f * ex * rcp_fast(u - 1.f);
        }
        if (v < a.lo) { v = a.lo; d = 0.f; }
        if (v > a.hi) { v = a.hi; d = 0.f; }
        v2[e] = v; d2[e] = d;
      }
    }
    sv[2 * hp][tl] = v2[0]; sv[2 * hp + 1][tl] = v2[1];
    if (want_d) { sd[2 * hp][tl] = d2[0]; sd[2 * hp + 1][tl] = d2[1]; }
  }
  block_sync();
  const int hh = tid >> 2, q = tid & 3, hg = hb * 64 + hh;
  if (hg < a.H) {
    const int64_t o = ((int64_t)b * a.H + hg) * a.L + lb * TOK + (TOK / 4) * q;
#pragma unroll
    for (int i = 0; i < TOK / 16; i++) {
      const int tl = (TOK / 4) * q + 4 * i;
      if (lb * TOK + tl < a.L) {   // L % 4 == 0: a vector is in or out as a whole
        *reinterpret_cast<f32x4*>(a.dtp + o + 4 * i) = f32x4{sv[hh][tl], sv[hh][tl + 1], sv[hh][tl + 2], sv[hh][tl + 3]};
        if (a.dsoft) *reinterpret_cast<f32x4*>(a.dsoft + o + 4 * i) = f32x4{sd[hh][tl], sd[hh][tl + 1], sd[hh][tl + 2], sd[hh][tl + 3]};
      }
    }
  }
}

__global__ void zero_f32_kernel(float* p, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}
__global__ void cvt_f32_kernel(const float* src, void* dst, int64_t sb, int64_t sl, int64_t sg, int B, int L, int G, int N, int dt) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n = (int64_t)B * L * G * N;
  for (; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int nn = (int)(i % N), g = (int)((i / N) % G), t = (int)((i / ((int64_t)N * G)) % L), b = (int)(i / ((int64_t)N * G * L));
    store_rt(dst, (int64_t)b * sb + (int64_t)t * sl + (int64_t)g * sg + nn, dt, src[i]);
  }
}

// ---------------------------------------------------------------------------------------------------------
// shape-generic scan: any DU, DK, dtype; fp32 VALU.  256 threads; a state row (one u) is spread over TK lanes
// with NPT contiguous k each; a workgroup covers RW = 256 / TK rows of one (b, h); tokens move in tiles of TT.
// ---------------------------------------------------------------------------------------------------------
constexpr int GEN_NPT = 16;
constexpr int GEN_TT = 16;

// F32: every activation source is fp32 (the reference's inference default): plain loads.  With the run-time dtype switch of load_rt
// every load sits under control flow, the compiler drains the memory pipeline at each one, and the 16 staging loads of a thread
// became 16 serial round trips per tile (65 us for a 72-token prefill layer).
template <bool F32>
__device__ __forceinline__ float src_at(const Src& s, int b, int t, int h, int g, int i) {
  const int64_t o = (int64_t)b * s.sb + (int64_t)t * s.sl + (int64_t)(s.per_group ? g : h) * s.sh + i;
  return F32 ? ((const float*)s.p)[o] : load_rt(s.p, o, s.dt);
}

// NPT: state elements (consecutive k) per thread.  16 everywhere but in the fp32 specialisation of few sequences (one fp32 prompt of
// the MMU path: B H = 64 sequences were 128 workgroups, one wave per SIMD on half the chip), which takes 8: twice the workgroups, half the
// serial work per token and thread.
template <int MODE, bool F32 = false, int NPT = GEN_NPT>
__global__ __launch_bounds__(256) void ssd_generic_kernel(GScan a, int TK) {
  OMK_DYN_SMEM(smem);
  const int RW = 256 / TK;
  float* sU = (float*)smem;              // [TT][RW]
  float* sX = sU + GEN_TT * RW;          // [TT][RW]  X4 rows (DC/DB) -- also Z rows for Y
  float* sO = sX + GEN_TT * RW;          // [TT][RW]
  float* sK = sO + GEN_TT * RW;          // [TT][DK]
  float* sQ = sK + GEN_TT * a.DK;        // [TT][DK]
  float* sdec = sQ + GEN_TT * a.DK;      // [TT] decay exp(a)
  float* sw = sdec + GEN_TT;             // [TT] input scale
  float* sdt = sw + GEN_TT;              // [TT] dt' of the token itself
  const int ublocks = (a.DU + RW - 1) / RW;
  const int ub = blockIdx.x % ublocks, h = (blockIdx.x / ublocks) % a.H, b = blockIdx.x / (ublocks * a.H);
  const int g = h / (a.H / a.G);
  const int tid = threadIdx.x, r = tid / TK, ks = tid % TK;
  const int u = ub * RW + r;
  const bool live = u < a.DU;
  const int k0 = ks * NPT;
  const float Ah = a.A[h];
  const float* dtp = a.dtp + ((int64_t)b * a.H + h) * a.L;
  float s[NPT];
#pragma unroll
  for (int j = 0; j < NPT; j++) {
    const int k = k0 + j;
    s[j] = (a.init && live && k < a.DK) ? load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt) : 0.f;
  }
  float dDacc = 0.f;
  const int nT = (a.L + GEN_TT - 1) / GEN_TT;
  // F32 specialisation (launched only with DK = 128, TK = 8, RW = 32): the next tile's rows are fetched into registers while this
  // tile is computed -- 2 + 2 + 8 + 8 + 2 values per thread -- so the global round trips overlap the token loop
  constexpr int CRW = 256 / (128 / NPT), NU = GEN_TT * CRW / 256;   // rows per workgroup, U / Z values per thread and tile (F32 form: DK = 128)
  float pu[NU], pz[NU], pk[8], pq[8], pd = 0.f, pla = 0.f;
  auto fetch = [&](int ti_) {
    const int tile_ = a.reverse ? nT - 1 - ti_ : ti_, t0_ = tile_ * GEN_TT;
    const int nl_ = (a.L - t0_) < GEN_TT ? (a.L - t0_) : GEN_TT;
#pragma unroll
    for (int j = 0; j < NU; j++) {
      const int i = tid + 256 * j, t = i / CRW, uu = ub * CRW + (i % CRW);
      const bool ok = t < nl_ && uu < a.DU;
      const int tc = ok ? t0_ + t : 0, uc = ok ? uu : 0;                     // clamped: unconditional loads
      const float vu = src_at<true>(a.U, b, tc, h, g, uc);
      pu[j] = ok ? vu : 0.f;
      pz[j] = 0.f;
      if (a.Z.p) { const float vz = src_at<true>(a.Z, b, tc, h, g, uc); pz[j] = ok ? vz : 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = tid + 256 * j, t = i / 128, k = i % 128;
      const int tc = t < nl_ ? t0_ + t : 0;
      const float vk = src_at<true>(a.K, b, tc, h, g, k), vq = src_at<true>(a.Q, b, tc, h, g, k);
      pk[j] = t < nl_ ? vk : 0.f;
      pq[j] = t < nl_ ? vq : 0.f;
    }
    if (tid < GEN_TT) {
      const int t = t0_ + tid;
      const bool ok = tid < nl_;
      const int ta = a.reverse ? t + 1 : t;
      const float d = dtp[ok ? t : 0], da = dtp[(ok && ta < a.L) ? ta : 0];
      pd = ok ? d : 0.f;
      pla = (ok && ta < a.L) ? da * Ah : 0.f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int j = 0; j < NU; j++) { sU[tid + 256 * j] = pu[j]; sX[tid + 256 * j] = pz[j]; }
#pragma unroll
    for (int j = 0; j < 8; j++) { sK[tid + 256 * j] = pk[j]; sQ[tid + 256 * j] = pq[j]; }
    if (tid < GEN_TT) { sdec[tid] = expf(pla); sw[tid] = a.w_is_dt ? pd : 1.f; sdt[tid] = pd; }
  };
  if (F32) { fetch(0); }
  for (int ti = 0; ti < nT; ti++) {
    const int tile = a.reverse ? nT - 1 - ti : ti;
    const int t0 = tile * GEN_TT;
    const int nl = (a.L - t0) < GEN_TT ? (a.L - t0) : GEN_TT;
    if (F32) {
      commit();
      if (tid < GEN_TT && tid >= nl) sw[tid] = 0.f;     // tokens past the end of a ragged tile: no input
      block_sync();
      if (ti + 1 < nT) fetch(ti + 1);
    } else {
    for (int i = tid; i < GEN_TT * RW; i += 256) {
      const int t = i / RW, rr = i % RW, uu = ub * RW + rr;
      const bool ok = t < nl && uu < a.DU;
      sU[i] = ok ? src_at<F32>(a.U, b, t0 + t, h, g, uu) : 0.f;
      if (MODE == GS_DC || MODE == GS_DB) sX[i] = ok ? src_at<F32>(a.X4, b, t0 + t, h, g, uu) : 0.f;
      if (MODE == GS_Y) sX[i] = (ok && a.Z.p) ? src_at<F32>(a.Z, b, t0 + t, h, g, uu) : 0.f;
    }
    for (int i = tid; i < GEN_TT * a.DK; i += 256) {
      const int t = i / a.DK, k = i % a.DK;
      sK[i] = t < nl ? src_at<F32>(a.K, b, t0 + t, h, g, k) : 0.f;
      sQ[i] = t < nl ? src_at<F32>(a.Q, b, t0 + t, h, g, k) : 0.f;
    }
    if (tid < GEN_TT) {
      const int t = t0 + tid;
      float d = 0.f, w = 0.f, la = 0.f;
      if (tid < nl) {
        d = dtp[t];
        const int ta = a.reverse ? t + 1 : t;                 // reverse scans decay with a_{t+1}
        la = ta < a.L ? dtp[ta] * Ah : 0.f;
        w = a.w_is_dt ? d : 1.f;
      }
      sdec[tid] = expf(la); sw[tid] = w; sdt[tid] = d;
    }
    }
    if (!F32)
    block_sync();
    // The recurrence of a thread's 16 state elements needs nothing from other lanes; only the output does (sum over the TK lanes
    // of a row).  So the token loop keeps one partial sum per token in registers and the cross-lane sums of all 16 tokens are
    // formed behind it, as independent shuffle chains -- three DEPENDENT ds_bpermute round trips per token made this loop latency
    // bound (72 us for a 72-token fp32 prefill layer).  Tokens past the end of a ragged tile are no-ops (decay 1, input 0).
    // (four tokens per batch: four independent chains hide the round trips; sixteen cost 63 more registers and half the occupancy
    // of the large-shape launches: 4.9 -> 6.9 ms at B 8, L 4096)
    for (int tb = 0; tb < GEN_TT; tb += 4) {
      float accv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int tt = tb + q;
        const int t = a.reverse ? GEN_TT - 1 - tt : tt;
        const float dec = sdec[t], wu = sw[t] * sU[t * RW + r];
        float acc = 0.f;
        if (k0 + NPT <= a.DK && (a.DK & 3) == 0) {
          // the thread's 16 consecutive k as four 16-byte LDS reads per operand (row offsets are multiples of 16 bytes here; the
          // element-wise form below costs 32 ds_read_b32 per token and thread)
          const f32x4* kq = reinterpret_cast<const f32x4*>(sK + t * a.DK + k0);
          const f32x4* qq = reinterpret_cast<const f32x4*>(sQ + t * a.DK + k0);
          float a4[4] = {0.f, 0.f, 0.f, 0.f};   // four partial sums: one accumulator would be a chain of NPT dependent FMAs per token
#pragma unroll
          for (int v = 0; v < NPT / 4; v++) {
            const f32x4 kv = kq[v], qv = qq[v];
#pragma unroll
            for (int e = 0; e < 4; e++) {
              s[4 * v + e] = s[4 * v + e] * dec + wu * kv[e];
              a4[e] += s[4 * v + e] * qv[e];
            }
          }
          acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
        } else {
#pragma unroll
          for (int j = 0; j < NPT; j++) {
            const int k = k0 + j;
            if (k < a.DK) {
              s[j] = s[j] * dec + wu * sK[t * a.DK + k];
              acc += s[j] * sQ[t * a.DK + k];
            }
          }
        }
        accv[q] = acc;
      }
      for (int m = TK >> 1; m >= 1; m >>= 1) {
#pragma unroll
        for (int q = 0; q < 4; q++) accv[q] += shfl_xor(accv[q], m);
      }
      if (ks == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) sO[(a.reverse ? GEN_TT - 1 - (tb + q) : tb + q) * RW + r] = accv[q];
      }
    }
    block_sync();
    // ---- epilogue over the tile
    if (MODE == GS_Y || MODE == GS_DX) {
      for (int i = tid; i < GEN_TT * RW; i += 256) {
        const int t = i / RW, rr = i % RW, uu = ub * RW + rr;
        if (t < nl && uu < a.DU) {
          float v = sO[i];
          const float Dv = a.D ? load_rt(a.D, (int64_t)h * a.Dsh + (int64_t)uu * a.Dsp, a.D_dt) : 0.f;
          if (MODE == GS_DX) v *= sdt[t];
          v += Dv * sU[i];
          const int64_t o = (int64_t)b * a.osb + (int64_t)(t0 + t) * a.osl + (int64_t)h * a.osh + uu;
          if (MODE == GS_Y) {
            if (a.outx) store_rt(a.outx, o, a.out_dt, v);
            if (a.Z.p) v *= silu_f(sX[i]);
          }
          store_rt(a.out, o, a.out_dt, v);
        }
      }
    } else {
      for (int i = tid; i < GEN_TT * RW; i += 256) {
        const int t = i / RW, rr = i % RW, uu = ub * RW + rr;
        if (t < nl && uu < a.DU) {
          float v = sO[i];
          if (MODE == GS_DB) v *= sdt[t];
          atomic_add_f32(a.acc32 + (((int64_t)b * a.L + (t0 + t)) * a.G + g) * a.DU + uu, v);
        }
      }
      if (tid < nl) {   // token scalar: sum over this workgroup's rows of X4[t][u] * O[t][u]
        float sc = 0.f;
        for (int rr = 0; rr < RW; rr++) sc += sX[tid * RW + rr] * sO[tid * RW + rr];
        atomic_add_f32(a.tokscal + ((int64_t)b * a.H + h) * a.L + t0 + tid, sc);
      }
      if (MODE == GS_DB && a.dD && ub == 0 && tid < a.DK) {
        for (int t = 0; t < nl; t++) dDacc += sK[t * a.DK + tid] * sQ[t * a.DK + tid];
      }
    }
    block_sync();
  }
  if (MODE == GS_DB && a.dD && ub == 0 && tid < a.DK) atomic_add_f32(a.dD + (int64_t)h * a.dDsh + (int64_t)tid * a.dDsp, dDacc);
  if (a.fin && live) {
    const float extra = a.fin_extra_decay ? expf(dtp[0] * Ah) : 1.f;
#pragma unroll
    for (int j = 0; j < NPT; j++) {
      const int k = k0 + j;
      if (k < a.DK) a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)u * a.fsu + (int64_t)k * a.fsk] = s[j] * extra;
    }
  }
}

int ssd_generic_launch(const GScan& g, omk_stream stream) {
  kernels_note("ssd_generic<mode=%d>", g.mode);
  int TK = 1;
  {   // fp32 forward of few sequences (see the kernel): 8 state elements per thread
    bool f32 = g.U.dt == OMK_F32 && g.K.dt == OMK_F32 && g.Q.dt == OMK_F32 && (!g.Z.p || g.Z.dt == OMK_F32);
    if (g.mode == GS_Y && f32 && g.DK == 128 && g.DU % 16 == 0 && (int64_t)g.B * g.H * (g.DU / 32) < 512) {
      const int RW8 = 16;
      dim3 grid8((unsigned)((int64_t)g.B * g.H * (g.DU / RW8))), block8(256);
      const size_t smem8 = (size_t)(GEN_TT * (3 * RW8 + 2 * g.DK) + 3 * GEN_TT) * 4;
      OMK_LAUNCH((ssd_generic_kernel<GS_Y, true, 8>), grid8, block8, smem8, stream, g, 16);
      return OMK_OK;
    }
  }
  while (TK * GEN_NPT < g.DK) TK <<= 1;
  if (TK > 64) return fail(OMK_EUNSUPPORTED, "ssd generic scan: inner dim %d too large", g.DK);
  const int RW = 256 / TK;
  const int ublocks = (g.DU + RW - 1) / RW;
  dim3 grid((unsigned)((int64_t)g.B * g.H * ublocks)), block(256);
  const size_t smem = (size_t)(GEN_TT * (3 * RW + 2 * g.DK) + 3 * GEN_TT) * 4;
  const bool f32 = g.U.dt == OMK_F32 && g.K.dt == OMK_F32 && g.Q.dt == OMK_F32 && (!g.Z.p || g.Z.dt == OMK_F32);
  if (g.mode == GS_Y && f32 && g.DK == 128 && TK == 8) { OMK_LAUNCH((ssd_generic_kernel<GS_Y, true>), grid, block, smem, stream, g, TK); return OMK_OK; }
  switch (g.mode) {
    case GS_Y: OMK_LAUNCH((ssd_generic_kernel<GS_Y>), grid, block, smem, stream, g, TK); break;
    case GS_DC: OMK_LAUNCH((ssd_generic_kernel<GS_DC>), grid, block, smem, stream, g, TK); break;
    case GS_DX: OMK_LAUNCH((ssd_generic_kernel<GS_DX>), grid, block, smem, stream, g, TK); break;
    default: OMK_LAUNCH((ssd_generic_kernel<GS_DB>), grid, block, smem, stream, g, TK); break;
  }
  return OMK_OK;
}

// ---------------------------------------------------------------------------------------------------------
// backward finishing pass: one wave per (b, h), reverse inclusive prefix over tokens with wave shuffles
// ---------------------------------------------------------------------------------------------------------
struct FinishArgs {
  const float* e; const float* wsum; const float* dtp; const float* dsoft; const float* A;
  const float* dfin; int64_t dfsb, dfsh, dfsp, dfsn; const float* sfin;   // sfin: (B, H, N, P) contiguous f32 from the dC scan
  void* ddt; int64_t dsb, dsl, dsh; int ddt_dt;
  float* dA; float* ddtb;
  const float* bnd;   // optional (B, H, nT + 1): < g, h > at the boundary behind tile ti (MFMA path): dl restarts from exp(a) * bnd
  int B, H, L, P, N;
  int ckpt_every;   // bnd is valid at tile boundaries j with j % ckpt_every == 0 and at the end of the sequence
  int bnd_is_q;     // chunk-parallel backward (ssd_cp.hip): bnd[j] already IS dl at token 64 j (the decay of that token included)
};
__global__ __launch_bounds__(64) void ssd_bwd_finish_kernel(FinishArgs a) {
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, lane = threadIdx.x;
  const int64_t base = (int64_t)bh * a.L;
  const float Ah = a.A[h];
  float carry = 0.f;
  if (a.dfin) {   // dl_L = <dfinal_states, S_final>
    float acc = 0.f;
    for (int i = lane; i < a.P * a.N; i += 64) {
      const int n = i / a.P, p = i % a.P;
      acc += a.sfin[(int64_t)bh * a.P * a.N + i] * a.dfin[(int64_t)b * a.dfsb + (int64_t)h * a.dfsh + (int64_t)p * a.dfsp + (int64_t)n * a.dfsn];
    }
    carry = wave_sum(acc);
  }
  float dAacc = 0.f, dbacc = 0.f;
  const int nT = (a.L + 63) / 64;
  for (int ti = nT - 1; ti >= 0; ti--) {
    // bf16-level errors in e / w must not accumulate over the whole sequence: restart from the exact boundary value
    if (a.bnd && (((ti + 1) % (a.ckpt_every > 0 ? a.ckpt_every : 1)) == 0 || ti + 1 == nT)) {
      const int tn = (ti + 1) * 64;   // bnd holds < g, h > at the boundary; the decay of the first token behind it is applied here
      carry = a.bnd[(int64_t)bh * (nT + 1) + ti + 1] * (tn < a.L ? expf(a.dtp[base + tn] * Ah) : 1.f);
    }
    const int t = ti * 64 + lane;
    const bool ok = t < a.L;
    const float d = ok ? a.dtp[base + t] : 0.f, w = ok ? a.wsum[base + t] : 0.f;
    float v = ok ? a.e[base + t] - d * w : 0.f;
    // inclusive suffix sum within the wave: v_l = sum_{j >= l} v_j
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      float o = shfl_down(v, off);
      if (lane + off < 64) v += o;
    }
    const float dl = v + carry;
    carry = shfl(dl, 0);
    if (ok) {
      const float ddtp = w + Ah * dl;
      const float draw = ddtp * a.dsoft[base + t];
      dAacc += d * dl;
      dbacc += draw;
      store_rt(a.ddt, (int64_t)b * a.dsb + (int64_t)t * a.dsl + (int64_t)h * a.dsh, a.ddt_dt, draw);
    }
  }
  dAacc = wave_sum(dAacc); dbacc = wave_sum(dbacc);
  if (lane == 0) {
    atomic_add_f32(a.dA + h, dAacc);
    if (a.ddtb) atomic_add_f32(a.ddtb + h, dbacc);
  }
}

// MFMA path: with the exact restart values bnd every 64-token tile is independent.  One 1024-thread block per (b, h),
// wave w takes tiles w, w + 16, ...; the dA / d(dt_bias) partials meet in LDS so each block issues two atomics.
__global__ __launch_bounds__(1024) void ssd_bwd_finish_par_kernel(FinishArgs a) {
  __shared__ float red[2][16];
  const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t base = (int64_t)bh * a.L;
  const float Ah = a.A[h];
  const int nT = (a.L + 63) / 64;
  const int cke = a.ckpt_every > 0 ? a.ckpt_every : 1;
  const int nR = (nT + cke - 1) / cke;   // runs of cke tiles between two restart boundaries (the sequence end is one)
  float dAacc = 0.f, dbacc = 0.f;
  for (int run = wave; run < nR; run += 16) {
    const int t_hi = ((run + 1) * cke < nT ? (run + 1) * cke : nT) - 1;   // last tile of the run
    const int tn = (t_hi + 1) * 64;   // bnd holds < g, h > at the boundary; apply the decay of the first token behind it
    float carry = a.bnd[(int64_t)bh * (nT + 1) + t_hi + 1] * ((tn < a.L && !a.bnd_is_q) ? expf(a.dtp[base + tn] * Ah) : 1.f);
    for (int ti = t_hi; ti >= run * cke; ti--) {
      const int t = ti * 64 + lane;
      const bool ok = t < a.L;
      const float d = ok ? a.dtp[base + t] : 0.f, w = ok ? a.wsum[base + t] : 0.f;
      const float ds = ok ? a.dsoft[base + t] : 0.f;
      const float v0 = ok ? a.e[base + t] - d * w : 0.f;
      // inclusive suffix sum = total - exclusive prefix
      const float incl = wave_incl_scan_add(v0);
      const float dl = wave_read_lane(incl, 63) - incl + v0 + carry;
      carry = wave_read_lane(dl, 0);
      if (ok) {
        const float draw = (w + Ah * dl) * ds;
        dAacc += d * dl;
        dbacc += draw;
        store_rt(a.ddt, (int64_t)b * a.dsb + (int64_t)t * a.dsl + (int64_t)h * a.dsh, a.ddt_dt, draw);
      }
    }
  }
  dAacc = wave_sum(dAacc); dbacc = wave_sum(dbacc);
  if (lane == 0) { red[0][wave] = dAacc; red[1][wave] = dbacc; }
  block_sync();
  if (threadIdx.x < 2) {
    float s = 0.f;
    for (int i = 0; i < 16; i++) s += red[threadIdx.x][i];
    if (threadIdx.x == 0) atomic_add_f32(a.dA + h, s);
    else if (a.ddtb) atomic_add_f32(a.ddtb + h, s);
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static Src make_src(const OmkTensor& t, bool per_group) {   // (B, L, H|G, dim) with unit stride on dim
  Src s; s.p = t.data; s.sb = t.stride[0]; s.sl = t.stride[1]; s.sh = t.stride[2]; s.dt = t.dtype; s.per_group = per_group ? 1 : 0;
  return s;
}
static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct SsdDims { int B, L, H, P, G, N; };
static int ssd_check_common(const OmkTensor& x, const OmkTensor& dt, const OmkTensor& A, const OmkTensor& Bm, const OmkTensor& Cm,
                            const OmkTensor& D, const OmkTensor& dtb, const OmkTensor& init, SsdDims* d, const char* who) {
  OMK_REQUIRE(present(x) && present(dt) && present(A) && present(Bm) && present(Cm), "%s: x, dt, A, B, C required", who);
  OMK_REQUIRE(x.ndim == 4 && dt.ndim == 3 && Bm.ndim == 4 && Cm.ndim == 4, "%s: x (B,L,H,P), dt (B,L,H), B/C (B,L,G,N)", who);
  d->B = (int)x.shape[0]; d->L = (int)x.shape[1]; d->H = (int)x.shape[2]; d->P = (int)x.shape[3]; d->G = (int)Bm.shape[2]; d->N = (int)Bm.shape[3];
  OMK_REQUIRE(d->G > 0 && d->H % d->G == 0, "%s: nheads must be a multiple of ngroups", who);
  OMK_REQUIRE(dt.shape[0] == d->B && dt.shape[1] == d->L && dt.shape[2] == d->H, "%s: dt shape", who);
  OMK_REQUIRE(Bm.shape[0] == d->B && Bm.shape[1] == d->L && Cm.shape[0] == d->B && Cm.shape[1] == d->L && Cm.shape[2] == d->G && Cm.shape[3] == d->N, "%s: B/C shape", who);
  OMK_REQUIRE(x.stride[3] == 1 && Bm.stride[3] == 1 && Cm.stride[3] == 1, "%s: x, B, C need unit stride on the last dim", who);
  OMK_REQUIRE(A.dtype == OMK_F32 && numel(A) == d->H && is_contig_last(A), "%s: A must be f32 (H)", who);
  if (present(D)) OMK_REQUIRE((D.ndim == 1 && D.shape[0] == d->H) || (D.ndim == 2 && D.shape[0] == d->H && D.shape[1] == d->P), "%s: D must be (H) or (H, P)", who);
  if (present(dtb)) OMK_REQUIRE(numel(dtb) == d->H && is_contig_last(dtb), "%s: dt_bias must be (H)", who);
  if (present(init)) OMK_REQUIRE(init.ndim == 4 && init.shape[0] == d->B && init.shape[1] == d->H && init.shape[2] == d->P && init.shape[3] == d->N, "%s: initial_states must be (B,H,P,N)", who);
  return OMK_OK;
}

static void launch_dt_prep(const OmkTensor& dt, const OmkTensor& dtb, const SsdDims& d, float* dtp, float* dsoft, int softplus, float lo, float hi, omk_stream stream,
                           float* z0 = nullptr, int64_t n0 = 0, float* z1 = nullptr, int64_t n1 = 0, float* z2 = nullptr, int64_t n2 = 0) {
  DtPrepArgs a = {};
  a.zero[0] = z0; a.nzero[0] = z0 ? (int)n0 : 0; a.zero[1] = z1; a.nzero[1] = z1 ? (int)n1 : 0; a.zero[2] = z2; a.nzero[2] = z2 ? (int)n2 : 0;
  a.dt = dt.data; a.bias = dtb.data; a.dtp = dtp; a.dsoft = dsoft; a.sb = dt.stride[0]; a.sl = dt.stride[1]; a.sh = dt.stride[2];
  a.B = d.B; a.L = d.L; a.H = d.H; a.dt_dt = dt.dtype; a.bias_dt = dtb.dtype; a.softplus = softplus; a.lo = lo;
  a.hi = hi > 0.f ? hi : INFINITY;
  dim3 block(256);
  kernels_note("ssd_dt_prep");
  const bool vec = (dt.dtype == OMK_BF16 || dt.dtype == OMK_F16) && a.sh == 1 && (a.sl % 2) == 0 && (a.sb % 2) == 0 && ((uintptr_t)dt.data & 3) == 0 &&
                   (d.H % 2) == 0 && (d.L % 4) == 0 && ((uintptr_t)dtp & 15) == 0 && (!dsoft || ((uintptr_t)dsoft & 15) == 0);
  if (vec) {
    // 32-token tiles: 7.6 us per launch against 9.7 us with 64 (twice the blocks in flight; the (B, H, L) rows still leave as full 128-byte lines)
    const int tok = getenv("OMK_DT_PREP_TOK") ? atoi(getenv("OMK_DT_PREP_TOK")) : 32;
    if (tok == 32) {
      dim3 grid((unsigned)((int64_t)d.B * ((d.L + 31) / 32) * ((d.H + 63) / 64)));
      if (dt.dtype == OMK_BF16) OMK_LAUNCH((ssd_dt_prep_vec_kernel<bf16_t, 32>), grid, block, 0, stream, a);
      else OMK_LAUNCH((ssd_dt_prep_vec_kernel<f16_t, 32>), grid, block, 0, stream, a);
    } else {
      dim3 grid((unsigned)((int64_t)d.B * ((d.L + 63) / 64) * ((d.H + 63) / 64)));
      if (dt.dtype == OMK_BF16) OMK_LAUNCH((ssd_dt_prep_vec_kernel<bf16_t, 64>), grid, block, 0, stream, a);
      else OMK_LAUNCH((ssd_dt_prep_vec_kernel<f16_t, 64>), grid, block, 0, stream, a);
    }
    return;
  }
  dim3 grid((unsigned)((int64_t)d.B * ((d.L + 31) / 32) * ((d.H + 31) / 32)));
  OMK_LAUNCH(ssd_dt_prep_kernel, grid, block, 0, stream, a);
}
static void launch_zero(float* p, int64_t n, omk_stream stream) {
  if (n <= 0) return;
  int64_t blocks = (n + 255) / 256;
  dim3 grid((unsigned)(blocks > 4096 ? 4096 : blocks)), block(256);
  OMK_LAUNCH(zero_f32_kernel, grid, block, 0, stream, p, n);
}
static int run_scan(const GScan& g, int force_generic, omk_stream stream) {
  if (!force_generic) {
    int rc = ssd_mfma_launch(g, stream);
    if (rc != OMK_EUNSUPPORTED) return rc;
    rc = ssd_f32_mfma_launch(g, stream);   // fp32 activations (the reference's inference default): ssd_f32.hip
    if (rc != OMK_EUNSUPPORTED) return rc;
  }
  return ssd_generic_launch(g, stream);
}

}  // namespace omk

using namespace omk;

extern "C" size_t omk_ssd_scan_fwd_workspace_bytes(const OmkSsdFwd* p) {
  if (!p) return 0;
  // dt' + developer profiling slots + the segment states of a split sequence (ssd_scan.h)
  size_t n = align256((size_t)p->x.shape[0] * p->x.shape[1] * p->x.shape[2] * 4) + 1024 + ssd_seg_bytes((int)(p->x.shape[0] * p->x.shape[2]), (int)p->x.shape[1]);
#ifdef OMK_PHASE_PROF
  n += 64 * 1024;   // developer build: first / last wall clock of every workgroup (ssd_a8.hip PT8_END), at the END of the workspace
#endif
  return n;
}

// Can this forward leave its window states behind, and how large are they?  The class-A MFMA kernel of the plain scan only (bf16,
// headdim 64, d_state 128, no gate, no pre-gate copy: the instantiations a training forward of the model uses).
static size_t fwd_window_states_bytes(const OmkSsdFwd* p) {
  if (!p || p->force_generic || !present(p->out) || present(p->z) || present(p->out_x)) return 0;
  if (p->x.ndim != 4 || p->Bm.ndim != 4 || p->x.dtype != OMK_BF16 || p->x.shape[3] != 64 || p->Bm.shape[3] != 128) return 0;
  if (p->flags & (OMK_SSD_KHILO | OMK_SSD_PRECISE)) return 0;
  const int64_t B = p->x.shape[0], L = p->x.shape[1], H = p->x.shape[2];
  // the same dry check the launch makes (strides, alignment, 32-bit row span): a forward the MFMA kernel cannot take must answer 0
  // here, so that the caller allocates nothing and omk_ssd_scan_fwd takes its ordinary fall-back chain (advisor finding, round 3:
  // a bf16 x at a 2-byte storage offset failed with 'window_states asked for on a shape outside the MFMA kernel')
  if (p->x.ndim != 4 || p->Cm.ndim != 4 || p->out.ndim != 4 || p->out.dtype != OMK_BF16 || p->Bm.dtype != OMK_BF16 || p->Cm.dtype != OMK_BF16) return 0;
  GScan g = {};
  g.mode = GS_Y; g.U = make_src(p->x, false); g.K = make_src(p->Bm, true); g.Q = make_src(p->Cm, true);
  g.B = (int)B; g.H = (int)H; g.G = (int)p->Bm.shape[2]; g.L = (int)L; g.DU = 64; g.DK = 128;
  if (g.G < 1 || g.H % g.G != 0) return 0;
  g.out = p->out.data; g.osb = p->out.stride[0]; g.osl = p->out.stride[1]; g.osh = p->out.stride[2]; g.out_dt = p->out.dtype;
  if (present(p->D)) { g.D = p->D.data; g.D_dt = p->D.dtype; g.Dsh = p->D.stride[0]; g.Dsp = p->D.ndim == 2 ? p->D.stride[1] : 0; }
  if (ssd_mfma_launch(g, nullptr, 1) != OMK_OK) return 0;
  return (size_t)B * ((L + 127) / 128) * H * 16384;
}
extern "C" size_t omk_ssd_scan_fwd_window_states_bytes(const OmkSsdFwd* p) { return fwd_window_states_bytes(p); }

extern "C" int omk_ssd_scan_fwd(const OmkSsdFwd* p, omk_stream stream) {
  OMK_REQUIRE(p && (present(p->out) || present(p->final_states)), "ssd_scan_fwd: out (or, for the state-only pass, final_states) required");
  SsdDims d;
  int rc = ssd_check_common(p->x, p->dt, p->A, p->Bm, p->Cm, p->D, p->dt_bias, p->initial_states, &d, "ssd_scan_fwd");
  if (rc) return rc;
  const bool state_only = !present(p->out);   // out absent: only the state behind the sequence (one shard of a context-parallel scan)
  if (!state_only) OMK_REQUIRE(p->out.ndim == 4 && p->out.shape[0] == d.B && p->out.shape[1] == d.L && p->out.shape[2] == d.H && p->out.shape[3] == d.P && p->out.stride[3] == 1, "ssd_scan_fwd: out must be (B,L,H,P) with unit last stride");
  OMK_REQUIRE(p->Bm.dtype == p->x.dtype && p->Cm.dtype == p->x.dtype && (state_only || p->out.dtype == p->x.dtype), "ssd_scan_fwd: B, C, out must have x's dtype");
  if (state_only) OMK_REQUIRE(!present(p->z) && !present(p->out_x), "ssd_scan_fwd: the state-only pass takes no gate and writes no output");
  if (present(p->z)) OMK_REQUIRE(p->z.ndim == 4 && p->z.stride[3] == 1 && p->z.dtype == p->x.dtype, "ssd_scan_fwd: z must be (B,L,H,P) of x's dtype");
  if (present(p->out_x)) OMK_REQUIRE(p->out_x.dtype == p->out.dtype && p->out_x.stride[0] == p->out.stride[0] && p->out_x.stride[1] == p->out.stride[1] && p->out_x.stride[2] == p->out.stride[2], "ssd_scan_fwd: out_x must match out's dtype and strides");
  if (present(p->final_states)) OMK_REQUIRE(p->final_states.dtype == OMK_F32 && p->final_states.ndim == 4, "ssd_scan_fwd: final_states must be f32 (B,H,P,N)");
  OMK_REQUIRE(p->workspace && p->workspace_bytes >= omk_ssd_scan_fwd_workspace_bytes(p), "ssd_scan_fwd: workspace too small");
  if (present(p->conv_weight)) {
    OMK_REQUIRE(p->conv_weight.ndim == 2 && p->conv_weight.shape[0] == (int64_t)d.H * d.P && p->conv_weight.shape[1] >= 1 && p->conv_weight.shape[1] <= 4,
                "ssd_scan_fwd: conv_weight must be (H * P, width <= 4)");
    if (present(p->conv_bias)) OMK_REQUIRE(p->conv_bias.ndim == 1 && p->conv_bias.shape[0] == (int64_t)d.H * d.P && p->conv_bias.stride[0] == 1, "ssd_scan_fwd: conv_bias must be dense (H * P)");
    if (state_only || present(p->z) || present(p->out_x) || present(p->window_states) || (p->flags & OMK_SSD_PRECISE) || p->force_generic || p->x.dtype != OMK_BF16)
      return fail(OMK_EUNSUPPORTED, "ssd_scan_fwd: the fused conv exists for the plain bf16 forward only (no gate / pre-gate copy / window states / PRECISE): run omk_causal_conv1d_fwd and the scan separately");
  }
  if ((int64_t)d.B * d.L * d.H * d.P == 0) return OMK_OK;
  kernels_reset();
  float* dtp = (float*)p->workspace;
  GScan g = {};
  // (tried in round 2: the scan reading the raw (B, L, H) dt itself and applying bias / softplus / clamp in its scalar pass, to
  // save this 16 us launch -- 280 us against 236 + 16 us: the 2-byte loads at stride H are 64 requests per wave and chunk and the
  // softplus lands on wave 0's critical path between the publish and the barrier.  Not kept.)
  launch_dt_prep(p->dt, p->dt_bias, d, dtp, nullptr, p->dt_softplus, p->dt_min, p->dt_max, stream);
  g.mode = GS_Y; g.U = make_src(p->x, false); g.K = make_src(p->Bm, true); g.Q = make_src(p->Cm, true);
  if (present(p->z)) g.Z = make_src(p->z, false);
  g.dtp = dtp; g.A = (const float*)p->A.data; g.B = d.B; g.H = d.H; g.G = d.G; g.L = d.L; g.DU = d.P; g.DK = d.N; g.reverse = 0; g.w_is_dt = 1;
  if (present(p->initial_states)) {
    g.init = p->initial_states.data; g.init_dt = p->initial_states.dtype;
    g.isb = p->initial_states.stride[0]; g.ish = p->initial_states.stride[1]; g.isu = p->initial_states.stride[2]; g.isk = p->initial_states.stride[3];
  }
  if (present(p->final_states)) {
    g.fin = (float*)p->final_states.data;
    g.fsb = p->final_states.stride[0]; g.fsh = p->final_states.stride[1]; g.fsu = p->final_states.stride[2]; g.fsk = p->final_states.stride[3];
  }
  g.out = p->out.data; g.osb = p->out.stride[0]; g.osl = p->out.stride[1]; g.osh = p->out.stride[2]; g.out_dt = p->out.dtype; g.outx = p->out_x.data;
  if (present(p->D)) { g.D = p->D.data; g.D_dt = p->D.dtype; g.Dsh = p->D.stride[0]; g.Dsp = p->D.ndim == 2 ? p->D.stride[1] : 0; }
  if (getenv("OMK_PROF") && p->workspace_bytes >= omk_ssd_scan_fwd_workspace_bytes(p)) g.prof = (unsigned long long*)((char*)p->workspace + align256((size_t)d.B * d.H * d.L * 4));
#ifdef OMK_PHASE_PROF
  if (const char* e = getenv("OMK_ABLATE")) g.ablate = atoi(e);
  if (getenv("OMK_PROF_WG") && p->workspace_bytes >= omk_ssd_scan_fwd_workspace_bytes(p))   // the per-workgroup clocks instead of the phase sums: slots behind everything else
    { g.prof = (unsigned long long*)((char*)p->workspace + omk_ssd_scan_fwd_workspace_bytes(p) - 64 * 1024); g.ablate |= 1 << 20; }
#endif
  g.flags = p->flags & (GSF_PRECISE | GSF_KHILO | GSF_FLUSH | GSF_NO_SPLIT | GSF_COLUMN_SLICE);
  if (ssd_seg_bytes(d.B * d.H, d.L) && !(p->flags & OMK_SSD_NO_SPLIT)) g.seg = (float*)((char*)p->workspace + align256((size_t)d.B * d.H * d.L * 4) + 1024);
  if (present(p->conv_weight)) {
    g.cw = p->conv_weight.data; g.cwsc = p->conv_weight.stride[0]; g.cwsk = p->conv_weight.stride[1]; g.cw_dt = p->conv_weight.dtype; g.cW = (int)p->conv_weight.shape[1];
    g.cb = p->conv_bias.data; g.cb_dt = p->conv_bias.dtype;
    // only ssd_a8.hip knows the fused conv: a shape it does not take (heads that do not pair up, D per (head, column), a sequence it would split) is the caller's to run unfused
    GScan q = g;
    if (q.seg && ssd_segments(d.B * d.H, d.L).nseg > 1) return fail(OMK_EUNSUPPORTED, "ssd_scan_fwd: fused conv on a sequence the scan splits into segments (B * H <= 128): run conv + scan separately");
    g.seg = nullptr;
    rc = (ssd_mfma_launch(g, nullptr, 1) == OMK_OK && ssd_a8_applies(g)) ? ssd_a8_launch(g, stream) : OMK_EUNSUPPORTED;
    if (rc == OMK_EUNSUPPORTED) return fail(OMK_EUNSUPPORTED, "ssd_scan_fwd: fused conv on a shape outside the specialised-wave kernel: run conv + scan separately");
    if (rc) return rc;
    return finish_launch("ssd_scan_fwd");
  }
  if (state_only) {
    rc = (p->force_generic || p->x.dtype != OMK_BF16) ? OMK_EUNSUPPORTED : ssd_mfma_state_only(g, stream);
    if (rc == OMK_EUNSUPPORTED) return fail(OMK_EUNSUPPORTED, "ssd_scan_fwd: the state-only pass exists for the MFMA shape only (bf16, headdim 64, d_state 128); run the scan and drop its output");
    if (rc) return rc;
    return finish_launch("ssd_scan_fwd");
  }
  if (present(p->window_states)) {
    const size_t need = fwd_window_states_bytes(p);
    OMK_REQUIRE(need > 0, "ssd_scan_fwd: this forward cannot save window states (omk_ssd_scan_fwd_window_states_bytes is 0 for these arguments)");
    OMK_REQUIRE(p->window_states.dtype == OMK_BF16 && (size_t)numel(p->window_states) * 2 >= need, "ssd_scan_fwd: window_states must be bf16 with omk_ssd_scan_fwd_window_states_bytes(p) bytes");
    g.dump = (uint16_t*)p->window_states.data; g.dump_nw = (d.L + 127) / 128;
    rc = ssd_mfma_launch(g, stream);
    if (rc == OMK_EUNSUPPORTED) return fail(OMK_EUNSUPPORTED, "ssd_scan_fwd: window_states asked for on a shape outside the MFMA kernel (strides / alignment)");
    if (rc) return rc;
    return finish_launch("ssd_scan_fwd");
  }
  rc = run_scan(g, p->force_generic, stream);
  if (rc) return rc;
  return finish_launch("ssd_scan_fwd");
}

// backward paths: 0 generic fp32 VALU scans, 1 sequential MFMA scans with head-pair partials (round 1 / 2), 2 chunk-parallel dB / dC
// (ssd_cp.hip) behind the dx scan
enum { BWD_GENERIC = 0, BWD_MFMA = 1, BWD_CP = 2 };
struct BwdWs { float *dtp, *dsoft, *e, *wsum, *dB32, *dC32, *sfin, *part, *ckpt, *bnd, *seg, *segf, *pB, *pC; uint16_t *Sf, *Sg; int nhs; size_t total; };
static BwdWs bwd_ws_layout(void* base, int B, int L, int H, int P, int G, int N, bool need_sfin, int path, bool cp_direct = false) {
  BwdWs w = {}; size_t off = 0; char* c = (char*)base;
  auto take = [&](size_t bytes) { float* r = (float*)(c + off); off += align256(bytes); return r; };
  const size_t bhl = (size_t)B * H * L * 4, blgn = (size_t)B * L * G * N * 4;
  bool need_part = path == BWD_MFMA; const bool mfma = path != BWD_GENERIC;
  w.dtp = take(bhl); w.dsoft = take(bhl); w.e = take(bhl); w.wsum = take(bhl);
  if (path != BWD_CP) { w.dB32 = take(blgn); w.dC32 = take(blgn); }
  w.sfin = need_sfin ? take((size_t)B * H * P * N * 4) : nullptr;
  w.part = need_part ? take((size_t)B * (H / 2) * L * 128 * 2) : nullptr;   // bf16 head-pair partial tiles
  const size_t nC = (size_t)(L + 63) / 64;
  w.ckpt = need_part ? take((size_t)B * (H / 2) * nC * (8 * 2 * 8 * 64) * 4) : nullptr;
  w.bnd = mfma ? take((size_t)B * H * (nC + 1) * 4) : nullptr;
  if (path == BWD_CP) {   // window states (16 KB bf16 images) of both directions, fp32 partials of the head subsets
    const size_t nW = (size_t)(L + 127) / 128;
    w.Sf = (uint16_t*)take((size_t)B * H * nW * 16384);
    w.Sg = (uint16_t*)take((size_t)B * H * nW * 16384);
    w.nhs = ssd_cp_heads_split(B, L, H, G);
    // (fp32 partials of the head subsets: not reserved when the kernel stores dB / dC itself -- 2 x 16 MiB at B 8, L 4096; advisor finding, round 5)
    if (!cp_direct) { w.pB = take((size_t)w.nhs * blgn); w.pC = take((size_t)w.nhs * blgn); }
  }
  // split sequences: start states of the segments -- adjoint state (dx and dB scans) and forward state (dC scan)
  w.seg = mfma && ssd_seg_bytes(B * H, L) ? take(ssd_seg_bytes(B * H, L)) : nullptr;
  w.segf = mfma && ssd_seg_bytes(B * H, L) ? take(ssd_seg_bytes(B * H, L)) : nullptr;
  w.total = off;
  return w;
}

// the three backward scans as descriptors (pointers into the workspace filled in by the caller)
static void bwd_scans(const OmkSsdBwd* p, const SsdDims& d, const BwdWs& w, bool mfma, GScan* gdc, GScan* gdx, GScan* gdb) {
  const bool has_dfin = present(p->dfinal_states);
  auto base = [&](int mode) {
    GScan g = {};
    g.mode = mode; g.dtp = w.dtp; g.A = (const float*)p->A.data; g.B = d.B; g.H = d.H; g.G = d.G; g.L = d.L;
    g.flags = p->flags & (GSF_FLUSH | GSF_NO_SPLIT | GSF_COLUMN_SLICE);
#ifdef OMK_PHASE_PROF
    if (const char* e = getenv("OMK_ABLATE_B")) g.ablate = atoi(e);
#endif
    return g;
  };
  {  // dC: state [n][p], forward in time
    GScan g = base(GS_DC);
    g.U = make_src(p->Bm, true); g.K = make_src(p->x, false); g.Q = make_src(p->dout, false); g.X4 = make_src(p->Cm, true);
    g.DU = d.N; g.DK = d.P; g.reverse = 0; g.w_is_dt = 1;
    if (present(p->initial_states)) {
      g.init = p->initial_states.data; g.init_dt = p->initial_states.dtype;
      g.isb = p->initial_states.stride[0]; g.ish = p->initial_states.stride[1]; g.isk = p->initial_states.stride[2]; g.isu = p->initial_states.stride[3];
    }
    if (has_dfin) { g.fin = w.sfin; g.fsb = (int64_t)d.H * d.N * d.P; g.fsh = (int64_t)d.N * d.P; g.fsu = d.P; g.fsk = 1; }
    if (mfma) { g.part = w.part; g.tokscal = w.e; g.ckpt = w.ckpt; g.ckpt_every = ssd_ckpt_every(); } else { g.acc32 = w.dC32; g.tokscal = w.e; }
    *gdc = g;
  }
  {  // dx: state [p][n], reverse in time
    GScan g = base(GS_DX);
    g.U = make_src(p->dout, false); g.K = make_src(p->Cm, true); g.Q = make_src(p->Bm, true);
    g.DU = d.P; g.DK = d.N; g.reverse = 1; g.w_is_dt = 0;
    if (has_dfin) {
      g.init = p->dfinal_states.data; g.init_dt = OMK_F32;
      g.isb = p->dfinal_states.stride[0]; g.ish = p->dfinal_states.stride[1]; g.isu = p->dfinal_states.stride[2]; g.isk = p->dfinal_states.stride[3];
    }
    if (present(p->dinitial_states)) {
      g.fin = (float*)p->dinitial_states.data; g.fin_extra_decay = 1;
      g.fsb = p->dinitial_states.stride[0]; g.fsh = p->dinitial_states.stride[1]; g.fsu = p->dinitial_states.stride[2]; g.fsk = p->dinitial_states.stride[3];
    }
    g.seg = (p->flags & OMK_SSD_NO_SPLIT) ? nullptr : w.seg;
    g.out = p->dx.data; g.osb = p->dx.stride[0]; g.osl = p->dx.stride[1]; g.osh = p->dx.stride[2]; g.out_dt = p->dx.dtype;
    if (present(p->D)) { g.D = p->D.data; g.D_dt = p->D.dtype; g.Dsh = p->D.stride[0]; g.Dsp = p->D.ndim == 2 ? p->D.stride[1] : 0; }
    *gdx = g;
  }
  {  // dB: state [n][p], reverse in time
    GScan g = base(GS_DB);
    g.U = make_src(p->Cm, true); g.K = make_src(p->dout, false); g.Q = make_src(p->x, false); g.X4 = make_src(p->Bm, true);
    g.DU = d.N; g.DK = d.P; g.reverse = 1; g.w_is_dt = 0;
    if (has_dfin) {
      g.init = p->dfinal_states.data; g.init_dt = OMK_F32;
      g.isb = p->dfinal_states.stride[0]; g.ish = p->dfinal_states.stride[1]; g.isk = p->dfinal_states.stride[2]; g.isu = p->dfinal_states.stride[3];
    }
    if (mfma) { g.part = w.part; g.tokscal = w.wsum; g.ckpt = w.ckpt; g.bnd = w.bnd; g.ckpt_every = ssd_ckpt_every(); }
    else { g.acc32 = w.dB32; g.tokscal = w.wsum; }
    if (present(p->dD)) { g.dD = (float*)p->dD.data; g.dDsh = p->dD.stride[0]; g.dDsp = p->dD.ndim == 2 ? p->dD.stride[1] : 0; }
    *gdb = g;
  }
}

static bool bwd_mfma_applies(const OmkSsdBwd* p, const SsdDims& d) {
  if (p->force_generic) return false;
  if (p->dB.dtype != OMK_BF16 && p->dB.dtype != OMK_F32 && p->dB.dtype != OMK_F16) return false;
  BwdWs w = {};
  GScan gdc, gdx, gdb;
  bwd_scans(p, d, w, true, &gdc, &gdx, &gdb);
  return ssd_mfma_launch(gdc, nullptr, 1) == OMK_OK && ssd_mfma_launch(gdx, nullptr, 1) == OMK_OK && ssd_mfma_launch(gdb, nullptr, 1) == OMK_OK;
}

// the chunk-parallel form: bf16 block shape (the MFMA scans apply), one D per head, rows addressable through 32-bit buffer offsets
static bool bwd_cp_applies(const OmkSsdBwd* p, const SsdDims& d) {
  if (p->flags & OMK_SSD_SEQUENTIAL_BWD) return false;
  if (present(p->D) && p->D.ndim != 1) return false;
  if (p->x.dtype != OMK_BF16 || d.P != 64 || d.N != 128) return false;
  const int64_t lim = (int64_t)0xfffff000;
  if ((int64_t)d.L * p->x.stride[1] * 2 >= lim || (int64_t)d.L * p->dout.stride[1] * 2 >= lim || (int64_t)d.L * p->Bm.stride[1] * 2 >= lim ||
      (int64_t)d.L * p->Cm.stride[1] * 2 >= lim) return false;
  return true;
}
static int bwd_path(const OmkSsdBwd* p, const SsdDims& d) {
  if (!bwd_mfma_applies(p, d)) return BWD_GENERIC;
  return bwd_cp_applies(p, d) ? BWD_CP : BWD_MFMA;
}

static bool bwd_cp_direct(const OmkSsdBwd* p, const SsdDims& d) {
  CpArgs c = {};
  c.dB = p->dB.data; c.dbsb = p->dB.stride[0]; c.dbsl = p->dB.stride[1]; c.dbsg = p->dB.stride[2]; c.dB_dt = p->dB.dtype;
  c.dC = p->dC.data; c.dcsb = p->dC.stride[0]; c.dcsl = p->dC.stride[1]; c.dcsg = p->dC.stride[2]; c.dC_dt = p->dC.dtype;
  c.nhs = ssd_cp_heads_split(d.B, d.L, d.H, d.G);
  return ssd_cp_direct(c);
}

extern "C" size_t omk_ssd_scan_bwd_workspace_bytes(const OmkSsdBwd* p) {
  if (!p) return 0;
  SsdDims d = {(int)p->x.shape[0], (int)p->x.shape[1], (int)p->x.shape[2], (int)p->x.shape[3], (int)p->Bm.shape[2], (int)p->Bm.shape[3]};
  const int path = bwd_path(p, d);
  return bwd_ws_layout(nullptr, d.B, d.L, d.H, d.P, d.G, d.N, present(p->dfinal_states), path, path == BWD_CP && bwd_cp_direct(p, d)).total;
}

extern "C" int omk_ssd_scan_bwd(const OmkSsdBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->dout) && present(p->dx) && present(p->ddt) && present(p->dA) && present(p->dB) && present(p->dC), "ssd_scan_bwd: dout, dx, ddt, dA, dB, dC required");
  SsdDims d;
  int rc = ssd_check_common(p->x, p->dt, p->A, p->Bm, p->Cm, p->D, p->dt_bias, p->initial_states, &d, "ssd_scan_bwd");
  if (rc) return rc;
  OMK_REQUIRE(p->dout.ndim == 4 && p->dout.stride[3] == 1 && p->dout.dtype == p->x.dtype, "ssd_scan_bwd: dout must be (B,L,H,P) of x's dtype, unit last stride");
  OMK_REQUIRE(p->dx.ndim == 4 && p->dx.stride[3] == 1 && p->dx.dtype == p->x.dtype, "ssd_scan_bwd: dx must be (B,L,H,P) of x's dtype");
  OMK_REQUIRE(p->Bm.dtype == p->x.dtype && p->Cm.dtype == p->x.dtype, "ssd_scan_bwd: B, C must have x's dtype");
  OMK_REQUIRE(p->dB.ndim == 4 && p->dC.ndim == 4 && p->dB.stride[3] == 1 && p->dC.stride[3] == 1, "ssd_scan_bwd: dB/dC must be (B,L,G,N) with unit last stride");
  OMK_REQUIRE(p->dA.dtype == OMK_F32 && is_contig_last(p->dA), "ssd_scan_bwd: dA must be f32 (H)");
  OMK_REQUIRE(p->ddt.ndim == 3, "ssd_scan_bwd: ddt must be (B,L,H)");
  if (present(p->dD)) OMK_REQUIRE(p->dD.dtype == OMK_F32 && present(p->D) && p->dD.ndim == p->D.ndim, "ssd_scan_bwd: dD must be f32 with D's shape");
  if (present(p->ddt_bias)) OMK_REQUIRE(p->ddt_bias.dtype == OMK_F32, "ssd_scan_bwd: ddt_bias must be f32");
  if (present(p->dinitial_states)) OMK_REQUIRE(p->dinitial_states.dtype == OMK_F32 && p->dinitial_states.ndim == 4, "ssd_scan_bwd: dinitial_states must be f32 (B,H,P,N)");
  if (present(p->dfinal_states)) OMK_REQUIRE(p->dfinal_states.dtype == OMK_F32 && p->dfinal_states.ndim == 4, "ssd_scan_bwd: dfinal_states must be f32 (B,H,P,N)");
  OMK_REQUIRE(p->workspace && p->workspace_bytes >= omk_ssd_scan_bwd_workspace_bytes(p), "ssd_scan_bwd: workspace too small");
  if ((int64_t)d.B * d.L * d.H * d.P == 0) return OMK_OK;
  kernels_reset();
  const bool has_dfin = present(p->dfinal_states);
  const int path = bwd_path(p, d);
  const bool mfma = path != BWD_GENERIC, cp = path == BWD_CP;
  BwdWs w = bwd_ws_layout(p->workspace, d.B, d.L, d.H, d.P, d.G, d.N, has_dfin, path, cp && bwd_cp_direct(p, d));
  const int64_t bhl = (int64_t)d.B * d.H * d.L, blgn = (int64_t)d.B * d.L * d.G * d.N;
  if (!mfma) { launch_zero(w.e, bhl, stream); launch_zero(w.wsum, bhl, stream); launch_zero(w.dB32, blgn, stream); launch_zero(w.dC32, blgn, stream); }
  launch_dt_prep(p->dt, p->dt_bias, d, w.dtp, w.dsoft, p->dt_softplus, p->dt_min, p->dt_max, stream, (float*)p->dA.data, d.H,
                 present(p->dD) ? (float*)p->dD.data : nullptr, present(p->dD) ? numel(p->dD) : 0,
                 present(p->ddt_bias) ? (float*)p->ddt_bias.data : nullptr, d.H);
  const float* A = (const float*)p->A.data;
  GScan gdc, gdx, gdb;
  bwd_scans(p, d, w, mfma, &gdc, &gdx, &gdb);
  if (mfma && gdx.seg && ssd_segments(d.B * d.H, d.L).nseg > 1) {
    // few (batch, head) pairs: every scan cuts the sequence into the same segments.  Their start states come from two
    // state-only passes + folds: the forward state (x, B; shared by the dC scan) and the adjoint state (dy, C; dx and dB)
    GScan gf = {};
    gf.mode = GS_Y; gf.dtp = w.dtp; gf.A = A; gf.B = d.B; gf.H = d.H; gf.G = d.G; gf.L = d.L; gf.DU = d.P; gf.DK = d.N;
    gf.U = make_src(p->x, false); gf.K = make_src(p->Bm, true); gf.Q = make_src(p->Cm, true); gf.reverse = 0; gf.w_is_dt = 1;
    if (present(p->initial_states)) {
      gf.init = p->initial_states.data; gf.init_dt = p->initial_states.dtype;
      gf.isb = p->initial_states.stride[0]; gf.ish = p->initial_states.stride[1]; gf.isu = p->initial_states.stride[2]; gf.isk = p->initial_states.stride[3];
    }
    gf.seg = w.segf; gf.flags = gdx.flags;
    int fmt_f = 0, fmt_a = 0;   // element order each pass left its states in (ssd_scan.h: seg_fmt)
    if (!(cp && present(p->window_states)) && (rc = ssd_mfma_prepare_segments(gf, stream, &fmt_f))) return rc;   // (saved window states: no forward state pass at all)
    if ((rc = ssd_mfma_prepare_segments(gdx, stream, &fmt_a))) return rc;
    gdc.seg = w.segf; gdc.seg_ready = 1; gdc.seg_fmt = fmt_f;
    gdx.seg_ready = 1; gdx.seg_fmt = fmt_a;
    gdb.seg = w.seg; gdb.seg_ready = 1; gdb.seg_fmt = fmt_a;
  }
  if (cp) {
    // chunk-parallel: a state-only forward pass and the dx scan leave the window-boundary states of both directions behind;
    // ssd_cp.hip forms dB, dC, the token scalars, the restart values of the decay gradient and dD window by window
    const int nW = (d.L + 127) / 128;
    GScan gf = {};
    gf.mode = GS_Y; gf.dtp = w.dtp; gf.A = A; gf.B = d.B; gf.H = d.H; gf.G = d.G; gf.L = d.L; gf.DU = d.P; gf.DK = d.N;
    gf.U = make_src(p->x, false); gf.K = make_src(p->Bm, true); gf.Q = make_src(p->Cm, true); gf.reverse = 0; gf.w_is_dt = 1;
    if (present(p->initial_states)) {
      gf.init = p->initial_states.data; gf.init_dt = p->initial_states.dtype;
      gf.isb = p->initial_states.stride[0]; gf.ish = p->initial_states.stride[1]; gf.isu = p->initial_states.stride[2]; gf.isk = p->initial_states.stride[3];
    }
    gf.flags = gdx.flags;
    if (gdc.seg_ready) { gf.seg = w.segf; gf.seg_ready = 1; gf.seg_fmt = gdc.seg_fmt; }
    // (a training forward that saved its window states spares this pass: the same images, written by the same code)
    const uint16_t* Sf = w.Sf;
    if (present(p->window_states)) {
      OMK_REQUIRE(p->window_states.dtype == OMK_BF16 && (size_t)numel(p->window_states) >= (size_t)d.B * nW * d.H * 8192, "ssd_scan_bwd: window_states too small");
      Sf = (const uint16_t*)p->window_states.data;
    } else {
      gf.dump = w.Sf; gf.dump_nw = nW;
      if ((rc = ssd_mfma_state_dump(gf, stream))) return rc;
    }
    gdx.dump = w.Sg; gdx.dump_nw = nW;
    if ((rc = ssd_mfma_launch(gdx, stream))) return rc;
    CpArgs c = {};
    c.X = (const uint16_t*)p->x.data; c.xsb = p->x.stride[0]; c.xsl = p->x.stride[1]; c.xsh = p->x.stride[2];
    c.DY = (const uint16_t*)p->dout.data; c.ysb = p->dout.stride[0]; c.ysl = p->dout.stride[1]; c.ysh = p->dout.stride[2];
    c.Bm = (const uint16_t*)p->Bm.data; c.bsb = p->Bm.stride[0]; c.bsl = p->Bm.stride[1]; c.bsg = p->Bm.stride[2];
    c.Cm = (const uint16_t*)p->Cm.data; c.csb = p->Cm.stride[0]; c.csl = p->Cm.stride[1]; c.csg = p->Cm.stride[2];
    c.dtp = w.dtp; c.A = A; c.Sf = Sf; c.Sg = w.Sg; c.e = w.e; c.wsum = w.wsum; c.bnd = w.bnd;
    if (present(p->dD)) { c.dD = (float*)p->dD.data; c.dDsh = p->dD.stride[0]; }
    c.pB = w.pB; c.pC = w.pC;
    c.dB = p->dB.data; c.dbsb = p->dB.stride[0]; c.dbsl = p->dB.stride[1]; c.dbsg = p->dB.stride[2]; c.dB_dt = p->dB.dtype;
    c.dC = p->dC.data; c.dcsb = p->dC.stride[0]; c.dcsl = p->dC.stride[1]; c.dcsg = p->dC.stride[2]; c.dC_dt = p->dC.dtype;
    c.B = d.B; c.L = d.L; c.H = d.H; c.G = d.G; c.nW = nW; c.nhs = w.nhs;
    if ((rc = ssd_cp_launch(c, stream))) return rc;
  } else if (mfma) {
    if ((rc = ssd_mfma_launch(gdc, stream))) return rc;   // forward in time: e_t, state checkpoints, dC partials
    ssd_reduce_partials(w.part, p->dC.data, p->dC.stride[0], p->dC.stride[1], p->dC.stride[2], p->dC.dtype, d.B, d.L, d.G, d.H, stream);
    if ((rc = ssd_mfma_launch(gdb, stream))) return rc;
    ssd_reduce_partials(w.part, p->dB.data, p->dB.stride[0], p->dB.stride[1], p->dB.stride[2], p->dB.dtype, d.B, d.L, d.G, d.H, stream);
    if ((rc = ssd_mfma_launch(gdx, stream))) return rc;
  } else {
    if ((rc = ssd_generic_launch(gdc, stream))) return rc;
    if ((rc = ssd_generic_launch(gdx, stream))) return rc;
    if ((rc = ssd_generic_launch(gdb, stream))) return rc;
  }
  {
    FinishArgs f = {};
    f.e = w.e; f.wsum = w.wsum; f.dtp = w.dtp; f.dsoft = w.dsoft; f.A = A;
    if (has_dfin && !mfma) {
      f.dfin = (const float*)p->dfinal_states.data; f.sfin = w.sfin;
      f.dfsb = p->dfinal_states.stride[0]; f.dfsh = p->dfinal_states.stride[1]; f.dfsp = p->dfinal_states.stride[2]; f.dfsn = p->dfinal_states.stride[3];
    }
    f.ddt = p->ddt.data; f.dsb = p->ddt.stride[0]; f.dsl = p->ddt.stride[1]; f.dsh = p->ddt.stride[2]; f.ddt_dt = p->ddt.dtype;
    f.bnd = mfma ? w.bnd : nullptr;
    f.ckpt_every = cp ? 1 : (mfma ? ssd_ckpt_every() : 1);
    f.bnd_is_q = cp ? 1 : 0;
    f.dA = (float*)p->dA.data; f.ddtb = (float*)p->ddt_bias.data; f.B = d.B; f.H = d.H; f.L = d.L; f.P = d.P; f.N = d.N;
    if (f.bnd && !f.dfin) {
      dim3 grid((unsigned)(d.B * d.H)), block(1024);
      kernels_note("ssd_bwd_finish_par");
      OMK_LAUNCH(ssd_bwd_finish_par_kernel, grid, block, 0, stream, f);
    } else {
      dim3 grid((unsigned)(d.B * d.H)), block(64);
      kernels_note("ssd_bwd_finish");
      OMK_LAUNCH(ssd_bwd_finish_kernel, grid, block, 0, stream, f);
    }
  }
  if (!mfma) {
    int64_t blocks = (blgn + 255) / 256;
    dim3 grid((unsigned)(blocks > 4096 ? 4096 : blocks)), block(256);
    OMK_LAUNCH(cvt_f32_kernel, grid, block, 0, stream, (const float*)w.dB32, p->dB.data, p->dB.stride[0], p->dB.stride[1], p->dB.stride[2], d.B, d.L, d.G, d.N, (int)p->dB.dtype);
    OMK_LAUNCH(cvt_f32_kernel, grid, block, 0, stream, (const float*)w.dC32, p->dC.data, p->dC.stride[0], p->dC.stride[1], p->dC.stride[2], d.B, d.L, d.G, d.N, (int)p->dC.dtype);
  }
  return finish_launch("ssd_scan_bwd");
}
