// norms.hip -- fused residual-add + RMSNorm/LayerNorm (layer_norm_fn) and gated RMSNorm (Mamba2.norm), fwd + bwd.
//
// Pure HBM streaming ops.  A row (or one (row, group) segment) is owned by WPR
// waves of a 256-thread block and lives in registers as NCHUNK x VEC floats per lane: WPR = 1 for rows up to 2048
// elements (wave-shuffle reductions only, no barrier), WPR = 4 (the whole block, one LDS exchange per reduction) for
// wider rows such as the 4096-lane gated norm of the 1.3B block -- so no array exceeds 32 floats per lane and nothing
// spills.  Every byte is touched once with 16-byte loads/stores; weight-gradient partials go to a [nparts][cols]
// workspace (nparts <= 1024) that a second tiny kernel folds.
// Algorithmic bytes per row of `cols`: add+norm fwd = cols * (sx + sres_in + sx + sres_out); gated fwd = 3 * cols * sx.
#include <cstdlib>
#include "omk_common.h"

namespace omk {

constexpr int NORM_THREADS = 256;
constexpr int NORM_WAVES = NORM_THREADS / 64;
constexpr int NORM_MAX_BLOCKS = 1024;

// Row totals over the WPR waves of a row.  The exchange array is double buffered by the parity of the reduction count, so ONE
// barrier per reduction is enough: a thread reaches the write of reduction k + 2 (same parity as k) only after the barrier of
// k + 1, which every thread passes after its reads of k.  (Round 1: two barriers per reduction -- four per row in the gated
// backward, whose two sums are one exchange now.)
typedef float NormRed[NORM_WAVES][2];
template <int WPR> __device__ __forceinline__ void row_sum2(float& v0, float& v1, NormRed* red, int wave, int& par) {
  v0 = wave_sum(v0);
  v1 = wave_sum(v1);
  if constexpr (WPR != 1) {
    static_assert(WPR == NORM_WAVES, "a row is one wave or the whole block");
    if ((threadIdx.x & 63) == 0) { red[par][wave][0] = v0; red[par][wave][1] = v1; }
    block_sync();
    v0 = red[par][0][0] + red[par][1][0] + red[par][2][0] + red[par][3][0];
    v1 = red[par][0][1] + red[par][1][1] + red[par][2][1] + red[par][3][1];
    par ^= 1;
  }
}
template <int WPR> __device__ __forceinline__ float row_sum(float v, NormRed* red, int wave, int& par) {
  v = wave_sum(v);
  if constexpr (WPR != 1) {
    static_assert(WPR == NORM_WAVES, "a row is one wave or the whole block");
    if ((threadIdx.x & 63) == 0) red[par][wave][0] = v;
    block_sync();
    v = red[par][0][0] + red[par][1][0] + red[par][2][0] + red[par][3][0];
    par ^= 1;
  }
  return v;
}
// element chunk c of this lane covers columns NORM_COL(c) .. + VEC of its row
#define NORM_ROWMAP()                                                         \
  __shared__ NormRed red[2];                                                  \
  int rpar = 0;                                                               \
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;                 \
  constexpr int RPB = NORM_WAVES / WPR; /* rows per block */                  \
  const int wsub = wave % WPR, wrow = wave / WPR;                             \
  (void)red; (void)rpar
#define NORM_COL(c) ((((c) * WPR + wsub) * 64 + lane) * VEC)

template <class T, int VEC>
__device__ __forceinline__ void ld(const T* p, float (&o)[VEC]) { load_vec<T, VEC>(p, o); }
template <class T, int VEC>
__device__ __forceinline__ void st(T* p, const float (&o)[VEC]) { store_vec<T, VEC>(p, o); }

struct NormArgs {
  const void* x; const void* res; const void* w; const void* b; const void* z;
  void* y; void* ro; float* rstd; float* mean;
  int64_t xs, rs, ys, ros, zs;      // row strides (elements)
  int64_t rows; int cols; int ngroups; int wdt, bdt; float eps; int rms; int norm_before_gate;
  int wvec;                         // weight (and bias) rows can be requested in 16-byte pieces and every lane's columns exist (launcher)
};
struct NormBwdArgs {
  const void* dy; const void* dro; const void* xsum; const void* w; const float* rstd; const float* mean;
  const void* x; const void* z;     // gated only
  void* dx; void* dri; void* dz;
  float* dw_part; float* db_part;   // [nparts][cols]
  int64_t dys, dros, xss, dxs, dris, xs, zs, dzs;
  int64_t rows; int cols; int ngroups; int wdt; int rms; float eps; int norm_before_gate;
};

// VEC weights of a lane as fp32: 16-byte requests when the row is fp32 / 16-bit and aligned (WVEC, checked by the launcher) -- a one-shot
// workgroup pays this prologue per row, the per-element run-time-dtype loads (sixteen scalar-branch round trips) cost it its gain
template <int VEC, bool WVEC>
__device__ __forceinline__ void load_w_row(const void* w, int i0, int wdt, float (&o)[VEC]) {
  if constexpr (WVEC) {
    static_assert(VEC == 8, "two 16-byte fp32 groups or one 16-bit group");
    if (wdt == OMK_F32) {
      const f32x4 v0 = reinterpret_cast<const f32x4*>((const float*)w + i0)[0], v1 = reinterpret_cast<const f32x4*>((const float*)w + i0)[1];
#pragma unroll
      for (int i = 0; i < 4; i++) { o[i] = v0[i]; o[4 + i] = v1[i]; }
    } else if (wdt == OMK_BF16) load_vec<bf16_t, VEC>((const bf16_t*)w + i0, o);
    else load_vec<f16_t, VEC>((const f16_t*)w + i0, o);
  } else {
#pragma unroll
    for (int i = 0; i < VEC; i++) o[i] = load_rt(w, i0 + i, wdt);
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward: y = norm(x + residual) * w + b ; residual_out = x + residual
// ---------------------------------------------------------------------------------------------------------
template <class TX, class TR, class TRO, int VEC, int NCHUNK, int WPR>
__global__ __launch_bounds__(NORM_THREADS) void add_norm_fwd_kernel(NormArgs a) {
  NORM_ROWMAP();
  const TX* x = (const TX*)a.x;
  const TR* res = (const TR*)a.res;
  TX* y = (TX*)a.y;
  TRO* ro = (TRO*)a.ro;
  const float inv_n = 1.f / (float)a.cols;
  float wreg[NCHUNK][VEC], breg[NCHUNK][VEC];   // this lane's columns never change: weights live in registers
  if constexpr (VEC == 8) {
    if (a.wvec) {   // (uniform) the short prologue of a short-lived workgroup: 16-byte requests
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        load_w_row<VEC, true>(a.w, NORM_COL(c), a.wdt, wreg[c]);
        if (a.b) load_w_row<VEC, true>(a.b, NORM_COL(c), a.bdt, breg[c]);
        else {
#pragma unroll
          for (int i = 0; i < VEC; i++) breg[c][i] = 0.f;
        }
      }
    }
  }
  if (VEC != 8 || !a.wvec) {
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        const int col = NORM_COL(c) + i;
        wreg[c][i] = col < a.cols ? load_rt(a.w, col, a.wdt) : 0.f;
        breg[c][i] = (a.b && col < a.cols) ? load_rt(a.b, col, a.bdt) : 0.f;
      }
  }
  const int64_t niter = (a.rows + RPB - 1) / RPB;
  for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
    const int64_t rraw = it * RPB + wrow;
    const bool rlive = rraw < a.rows;
    const int64_t row = rlive ? rraw : a.rows - 1;
    float v[NCHUNK][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < a.cols) {
        ld<TX, VEC>(x + row * a.xs + col, v[c]);
        if (res) {
          float r[VEC];
          ld<TR, VEC>(res + row * a.rs + col, r);
#pragma unroll
          for (int i = 0; i < VEC; i++) v[c][i] += r[i];
        }
        if (ro && rlive) st<TRO, VEC>(ro + row * a.ros + col, v[c]);
#pragma unroll
        for (int i = 0; i < VEC; i++) { s1 += v[c][i]; s2 += v[c][i] * v[c][i]; }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) v[c][i] = 0.f;
      }
    }
    float mu = 0.f, var;
    if (a.rms) {
      var = row_sum<WPR>(s2, red, wave, rpar) * inv_n;
    } else {
      mu = row_sum<WPR>(s1, red, wave, rpar) * inv_n;
      float d2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        if (NORM_COL(c) < a.cols) {
#pragma unroll
          for (int i = 0; i < VEC; i++) { float d = v[c][i] - mu; d2 += d * d; }
        }
      }
      var = row_sum<WPR>(d2, red, wave, rpar) * inv_n;
    }
    const float rstd = rsqrtf(var + a.eps);
    if (lane == 0 && wsub == 0 && rlive) {
      if (a.rstd) a.rstd[row] = rstd;
      if (a.mean) a.mean[row] = mu;
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < a.cols && rlive) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          o[i] = (v[c][i] - mu) * rstd * wreg[c][i] + breg[c][i];
        }
        st<TX, VEC>(y + row * a.ys + col, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward.  dx = (wdy - xhat*c1 - c2) * rstd + dresidual_out ; dw += dy * xhat ; db += dy
// ---------------------------------------------------------------------------------------------------------
template <class TX, class TS, class TRI, int VEC, int NCHUNK, int WPR>
__global__ __launch_bounds__(NORM_THREADS) void add_norm_bwd_kernel(NormBwdArgs a) {
  NORM_ROWMAP();
  const TX* dy = (const TX*)a.dy;
  const TS* dro = (const TS*)a.dro;   // grad of residual_out has residual_out's dtype (= xsum's)
  const TS* xsum = (const TS*)a.xsum;
  TX* dx = (TX*)a.dx;
  TRI* dri = (TRI*)a.dri;
  const float inv_n = 1.f / (float)a.cols;
  float wreg[NCHUNK][VEC];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++)
#pragma unroll
    for (int i = 0; i < VEC; i++) { const int col = NORM_COL(c) + i; wreg[c][i] = col < a.cols ? load_rt(a.w, col, a.wdt) : 0.f; }
  float dwacc[NCHUNK][VEC], dbacc[NCHUNK][VEC];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++)
#pragma unroll
    for (int i = 0; i < VEC; i++) { dwacc[c][i] = 0.f; dbacc[c][i] = 0.f; }
  const int64_t niter = (a.rows + RPB - 1) / RPB;
  for (int64_t it = blockIdx.x; it < niter; it += gridDim.x) {
    const int64_t rraw = it * RPB + wrow;
    const bool rlive = rraw < a.rows;
    const int64_t row = rlive ? rraw : a.rows - 1;
    float xh[NCHUNK][VEC], wdy[NCHUNK][VEC];
    const float rstd = a.rstd[row];
    const float mu = a.mean ? a.mean[row] : 0.f;
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < a.cols && rlive) {
        float g[VEC];
        ld<TS, VEC>(xsum + row * a.xss + col, xh[c]);
        ld<TX, VEC>(dy + row * a.dys + col, g);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          xh[c][i] = (xh[c][i] - mu) * rstd;
          wdy[c][i] = g[i] * wreg[c][i];
          dwacc[c][i] += g[i] * xh[c][i];
          dbacc[c][i] += g[i];
          c1 += xh[c][i] * wdy[c][i];
          c2 += wdy[c][i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) { xh[c][i] = 0.f; wdy[c][i] = 0.f; }
      }
    }
    c1 = row_sum<WPR>(c1, red, wave, rpar) * inv_n;
    c2 = a.rms ? 0.f : row_sum<WPR>(c2, red, wave, rpar) * inv_n;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < a.cols && rlive) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) o[i] = (wdy[c][i] - xh[c][i] * c1 - c2) * rstd;
        if (dro) {
          float r[VEC];
          ld<TS, VEC>(dro + row * a.dros + col, r);
#pragma unroll
          for (int i = 0; i < VEC; i++) o[i] += r[i];
        }
        st<TX, VEC>(dx + row * a.dxs + col, o);
        if (dri) st<TRI, VEC>(dri + row * a.dris + col, o);
      }
    }
  }
  // ONE partial row per workgroup.  Waves of the same row (WPR > 1) write disjoint columns; waves of different rows (WPR == 1: four rows per
  // workgroup) add their sums up in LDS first, in wave order -- a quarter of the partial rows to write and to fold (the 1.3B block: 33.5 -> 8.4 MB
  // per gradient, the fold launch 20 -> 7 us)
  if constexpr (RPB > 1) {
    static_assert(WPR == 1, "a wave owns whole rows");
    __shared__ float shw[NCHUNK * 64 * VEC], shb[NCHUNK * 64 * VEC];
    for (int w = 0; w + 1 < RPB; w++) {
      if (wrow == w) {
#pragma unroll
        for (int c = 0; c < NCHUNK; c++)
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            const int j = NORM_COL(c) + i;
            shw[j] = (w ? shw[j] : 0.f) + dwacc[c][i];
            shb[j] = (w ? shb[j] : 0.f) + dbacc[c][i];
          }
      }
      block_sync();
    }
    if (wrow != RPB - 1) return;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
#pragma unroll
      for (int i = 0; i < VEC; i++) { const int j = NORM_COL(c) + i; dwacc[c][i] += shw[j]; dbacc[c][i] += shb[j]; }
  }
  const int64_t part = blockIdx.x;
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) {
    const int col = NORM_COL(c);
    if (col < a.cols) {
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        if (a.dw_part) a.dw_part[part * a.cols + col + i] = dwacc[c][i];
        if (a.db_part) a.db_part[part * a.cols + col + i] = dbacc[c][i];
      }
    }
  }
}

// out[col] = sum_p part[p][col]; block = 32 columns (one 128-byte line per partial row) x 32 part-groups, folded
// through LDS: cols / 32 blocks of 16 waves keep enough loads in flight for the 16 MB of partials of the 1.3B block
constexpr int RP_COLS = 32, RP_GROUPS = 32;
__global__ __launch_bounds__(RP_COLS * RP_GROUPS) void reduce_parts_kernel(const float* part, int nparts, int cols, float* out) {
  __shared__ float sh[RP_GROUPS][RP_COLS + 1];
  const int cl = threadIdx.x % RP_COLS, pg = threadIdx.x / RP_COLS;
  const int col = blockIdx.x * RP_COLS + cl;
  float s0 = 0.f, s1 = 0.f;
  if (col < cols) {
    int p = pg;
    for (; p + RP_GROUPS < nparts; p += 2 * RP_GROUPS) {
      s0 += part[(int64_t)p * cols + col];
      s1 += part[(int64_t)(p + RP_GROUPS) * cols + col];
    }
    if (p < nparts) s0 += part[(int64_t)p * cols + col];
  }
  sh[pg][cl] = s0 + s1;
  block_sync();
  if (pg == 0 && col < cols) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < RP_GROUPS; g++) s += sh[g][cl];
    out[col] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------
// gated RMSNorm.  norm_before_gate = 0 (reference): y = rmsnorm(x * silu(z)) * w ; 1: y = rmsnorm(x) * w * silu(z)
// a segment = (row, group) of group_size = cols / ngroups lanes; blocks are striped so a block stays in one group
// ---------------------------------------------------------------------------------------------------------
template <class TX, int VEC, int NCHUNK, int WPR>
__global__ __launch_bounds__(NORM_THREADS) void norm_gated_fwd_kernel(NormArgs a) {
  NORM_ROWMAP();
  const TX* x = (const TX*)a.x;
  const TX* z = (const TX*)a.z;
  TX* y = (TX*)a.y;
  const int gs = a.cols / a.ngroups;
  const float inv_n = 1.f / (float)gs;
  const int grp = blockIdx.x % a.ngroups;
  const int g0 = grp * gs;
  const int64_t bi = blockIdx.x / a.ngroups, nbg = gridDim.x / a.ngroups;
  float wreg[NCHUNK][VEC], breg[NCHUNK][VEC];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++)
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      const int col = NORM_COL(c) + i;
      wreg[c][i] = col < gs ? load_rt(a.w, g0 + col, a.wdt) : 0.f;
      breg[c][i] = (a.b && col < gs) ? load_rt(a.b, g0 + col, a.bdt) : 0.f;
    }
  const int64_t niter = (a.rows + RPB - 1) / RPB;
  for (int64_t it = bi; it < niter; it += nbg) {
    const int64_t rraw = it * RPB + wrow;
    const bool rlive = rraw < a.rows;
    const int64_t row = rlive ? rraw : a.rows - 1;
    float v[NCHUNK][VEC], sz[NCHUNK][VEC];
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < gs) {
        ld<TX, VEC>(x + row * a.xs + g0 + col, v[c]);
        if (z) {
          ld<TX, VEC>(z + row * a.zs + g0 + col, sz[c]);
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            sz[c][i] = silu_fast(sz[c][i]);
            if (!a.norm_before_gate) v[c][i] *= sz[c][i];
          }
        }
#pragma unroll
        for (int i = 0; i < VEC; i++) s2 += v[c][i] * v[c][i];
      } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) { v[c][i] = 0.f; sz[c][i] = 0.f; }
      }
    }
    const float rstd = rsqrtf(row_sum<WPR>(s2, red, wave, rpar) * inv_n + a.eps);
    if (lane == 0 && wsub == 0 && a.rstd && rlive) a.rstd[row * a.ngroups + grp] = rstd;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < gs && rlive) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          o[i] = v[c][i] * rstd * wreg[c][i] + breg[c][i];
          if (z && a.norm_before_gate) o[i] *= sz[c][i];
        }
        st<TX, VEC>(y + row * a.ys + g0 + col, o);
      }
    }
  }
}

// The reference's mode only (OmniMamba / Mamba-2: gate z present, norm_before_gate = 0, no bias, one segment per block row): the rows of a
// block are software pipelined -- the 16-byte loads of the NEXT row are in flight while this row is reduced and stored (a persistent
// block otherwise has one row of loads outstanding per wave: the general kernel runs at 4.5 TB/s where a plain element-wise kernel with
// the same three streams reaches 6.1) -- and nothing the mode does not need stays in registers (bias, the gate after its use).
template <class TX, int VEC, int NCHUNK, int WPR, bool WVEC>
__global__ __launch_bounds__(NORM_THREADS) void norm_gated_fwd_lean_kernel(NormArgs a) {
  NORM_ROWMAP();
  const TX* x = (const TX*)a.x;
  const TX* z = (const TX*)a.z;
  TX* y = (TX*)a.y;
  const int gs = a.cols / a.ngroups;
  const float inv_n = 1.f / (float)gs;
  const int grp = blockIdx.x % a.ngroups;
  const int g0 = grp * gs;
  const int64_t bi = blockIdx.x / a.ngroups, nbg = gridDim.x / a.ngroups;
  float wreg[NCHUNK][VEC];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) load_w_row<VEC, WVEC>(a.w, g0 + NORM_COL(c), a.wdt, wreg[c]);
  const int64_t niter = (a.rows + RPB - 1) / RPB;
  vec_t<TX, VEC> rx[NCHUNK], rz[NCHUNK];
  auto issue = [&](int64_t it) {
    const int64_t rraw = it * RPB + wrow;
    const int64_t row = rraw < a.rows ? rraw : a.rows - 1;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      if (a.rms & 32) {
        const u32x4 vx = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(x + row * a.xs + g0 + NORM_COL(c)));
        const u32x4 vz = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(z + row * a.zs + g0 + NORM_COL(c)));
        rx[c] = __builtin_bit_cast(vec_t<TX, VEC>, vx); rz[c] = __builtin_bit_cast(vec_t<TX, VEC>, vz);
      } else {
      rx[c] = *reinterpret_cast<const vec_t<TX, VEC>*>(x + row * a.xs + g0 + NORM_COL(c));
      rz[c] = *reinterpret_cast<const vec_t<TX, VEC>*>(z + row * a.zs + g0 + NORM_COL(c));
      }
    }
  };
  if (bi < niter) issue(bi);
  for (int64_t it = bi; it < niter; it += nbg) {
    const int64_t rraw = it * RPB + wrow;
    const bool rlive = rraw < a.rows;
    float v[NCHUNK][VEC];
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        v[c][i] = (a.rms & 4) ? to_f32(rx[c].e[i]) * to_f32(rz[c].e[i]) : to_f32(rx[c].e[i]) * silu_fast(to_f32(rz[c].e[i]));
        s2 += v[c][i] * v[c][i];
      }
    if (it + nbg < niter) issue(it + nbg);   // (the staging registers are free: the next row's loads fly during the reduction)
    const float rstd = (a.rms & 2) ? s2 : rsqrtf(row_sum<WPR>(s2, red, wave, rpar) * inv_n + a.eps);   // (a.rms bits 1, 2: developer ablations, OMK_NORM_ABL)
    if (lane == 0 && wsub == 0 && a.rstd && rlive && !(a.rms & 8)) a.rstd[rraw * a.ngroups + grp] = rstd;
    if (rlive) {
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) o[i] = v[c][i] * rstd * wreg[c][i];
        if (!(a.rms & 16)) {   // (streamed once: non-temporal, - 3 %)
          vec_t<TX, VEC> ov;
#pragma unroll
          for (int i = 0; i < VEC; i++) ov.e[i] = from_f32<TX>(o[i]);
          __builtin_nontemporal_store(__builtin_bit_cast(u32x4, ov), reinterpret_cast<u32x4*>(y + rraw * a.ys + g0 + NORM_COL(c)));
        } else st<TX, VEC>(y + rraw * a.ys + g0 + NORM_COL(c), o);
      }
    }
  }
}

template <class TX, int VEC, int NCHUNK, int WPR>
__global__ __launch_bounds__(NORM_THREADS) void norm_gated_bwd_kernel(NormBwdArgs a) {
  NORM_ROWMAP();
  const TX* x = (const TX*)a.x;
  const TX* z = (const TX*)a.z;
  const TX* dy = (const TX*)a.dy;
  TX* dx = (TX*)a.dx;
  TX* dz = (TX*)a.dz;
  const int gs = a.cols / a.ngroups;
  const float inv_n = 1.f / (float)gs;
  const int grp = blockIdx.x % a.ngroups;
  const int g0 = grp * gs;
  const int64_t bi = blockIdx.x / a.ngroups, nbg = gridDim.x / a.ngroups;
  float dwacc[NCHUNK][VEC], wreg[NCHUNK][VEC];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++)
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      dwacc[c][i] = 0.f;
      const int col = NORM_COL(c) + i;
      wreg[c][i] = col < gs ? load_rt(a.w, g0 + col, a.wdt) : 0.f;
    }
  const int64_t niter = (a.rows + RPB - 1) / RPB;
  for (int64_t it = bi; it < niter; it += nbg) {
    const int64_t rraw = it * RPB + wrow;
    const bool rlive = rraw < a.rows;
    const int64_t row = rlive ? rraw : a.rows - 1;
    // gv: the normalised quantity (x or x*silu(z)); zv: z, later (norm_before_gate) the finished dz; wdy: dy then w*dy'
    float xv[NCHUNK][VEC], zv[NCHUNK][VEC], gv[NCHUNK][VEC], wdy[NCHUNK][VEC];
    // both row sums in ONE exchange: sum g^2 (-> rstd) and sum g w dy' -- the second is c1 / rstd, it does not need rstd itself
    float s2 = 0.f, t2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < gs && rlive) {
        ld<TX, VEC>(x + row * a.xs + g0 + col, xv[c]);
        ld<TX, VEC>(dy + row * a.dys + g0 + col, wdy[c]);
        if (z) ld<TX, VEC>(z + row * a.zs + g0 + col, zv[c]);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          if (!z) zv[c][i] = 0.f;
          gv[c][i] = (z && !a.norm_before_gate) ? xv[c][i] * silu_fast(zv[c][i]) : xv[c][i];
          s2 += gv[c][i] * gv[c][i];
          const float dyn = (z && a.norm_before_gate) ? wdy[c][i] * silu_fast(zv[c][i]) : wdy[c][i];   // what the norm sees
          t2 += gv[c][i] * dyn * wreg[c][i];
        }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) { xv[c][i] = 0.f; zv[c][i] = 0.f; gv[c][i] = 0.f; wdy[c][i] = 0.f; }
      }
    }
    row_sum2<WPR>(s2, t2, red, wave, rpar);
    const float rstd = rsqrtf(s2 * inv_n + a.eps);
    const float c1 = rstd * t2 * inv_n;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < gs && rlive) {
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          const float w = wreg[c][i];
          float dyv = wdy[c][i];
          const float xhat = gv[c][i] * rstd;
          if (z && a.norm_before_gate) {   // y = xhat*w*silu(z): dz from dy*xhat*w, the norm sees dy*silu(z)
            const float sig = sigmoid_fast(zv[c][i]);
            const float sg = zv[c][i] * sig;
            zv[c][i] = dyv * xhat * w * sig * (1.f + zv[c][i] * (1.f - sig));   // finished dz
            dyv *= sg;
          }
          dwacc[c][i] += dyv * xhat;
          wdy[c][i] = dyv * w;
          gv[c][i] = xhat;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      const int col = NORM_COL(c);
      if (col < gs && rlive) {
        float ox[VEC], oz[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          const float dg = (wdy[c][i] - gv[c][i] * c1) * rstd;   // grad wrt the normalised input
          if (z && !a.norm_before_gate) {
            const float sig = sigmoid_fast(zv[c][i]);
            ox[i] = dg * zv[c][i] * sig;
            oz[i] = dg * xv[c][i] * sig * (1.f + zv[c][i] * (1.f - sig));
          } else {
            ox[i] = dg;
            oz[i] = zv[c][i];
          }
        }
        st<TX, VEC>(dx + row * a.dxs + g0 + col, ox);
        if (dz) st<TX, VEC>(dz + row * a.dzs + g0 + col, oz);
      }
    }
  }
  const int64_t part = bi * RPB + wrow;   // partial rows of this group
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) {
    const int col = NORM_COL(c);
    if (col < gs) {
#pragma unroll
      for (int i = 0; i < VEC; i++)
        if (a.dw_part) a.dw_part[part * a.cols + g0 + col + i] = dwacc[c][i];
    }
  }
}

// The reference's mode of the backward (gate present, norm_before_gate = 0): the general kernel above evaluates the sigmoid of the gate
// twice and keeps six fp32 arrays per lane alive across the row reduction (157 registers: three waves per SIMD, and 40 VALU + 4
// transcendental operations per element -- more SIMD time than the five streams take at the memory roof).  Here: one sigmoid, the
// inputs stay in their 16-byte staging registers until their last use, three fp32 arrays cross the reduction.
template <class TX, int VEC, int NCHUNK, int WPR>
__global__ __launch_bounds__(NORM_THREADS) void norm_gated_bwd_lean_kernel(NormBwdArgs a) {
  static_assert(std::is_same<TX, bf16_t>::value && VEC == 8, "packed bf16 pairs: a 16-byte staging register = four element pairs");
  NORM_ROWMAP();
  const TX* x = (const TX*)a.x;
  const TX* z = (const TX*)a.z;
  const TX* dy = (const TX*)a.dy;
  TX* dx = (TX*)a.dx;
  TX* dz = (TX*)a.dz;
  const int gs = a.cols / a.ngroups;
  const float inv_n = 1.f / (float)gs;
  const int grp = blockIdx.x % a.ngroups;
  const int g0 = grp * gs;
  const int64_t bi = blockIdx.x / a.ngroups, nbg = gridDim.x / a.ngroups;
  // (the weight row lives in LDS, read per row: sixteen registers that decide between three and four waves per SIMD)
  __shared__ __attribute__((aligned(16))) float wsh[WPR * NCHUNK * 64 * VEC];
  // Every element pair of a 32-bit input word is ONE packed fp32 pair from its conversion to the v_cvt_pk_bf16_f32 of the outputs (round 6):
  // v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 do both elements per instruction -- only the two exp2 and the two rcp of a pair stay scalar.
  f32x2 dwacc[NCHUNK][4];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) {
#pragma unroll
    for (int e = 0; e < 4; e++) dwacc[c][e] = f32x2{0.f, 0.f};
    if (wrow == 0) {
#pragma unroll
      for (int i = 0; i < VEC; i++) wsh[NORM_COL(c) + i] = load_rt(a.w, g0 + NORM_COL(c) + i, a.wdt);
    }
  }
  block_sync();
  auto unpk = [](uint32_t w) -> f32x2 { return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; };
  const int64_t niter = (a.rows + RPB - 1) / RPB;
  for (int64_t it = bi; it < niter; it += nbg) {
    const int64_t rraw = it * RPB + wrow;
    const bool rlive = rraw < a.rows;
    const int64_t row = rlive ? rraw : a.rows - 1;
    u32x4 rx[NCHUNK], rz[NCHUNK], rd[NCHUNK];
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      rx[c] = *reinterpret_cast<const u32x4*>(x + row * a.xs + g0 + NORM_COL(c));
      rd[c] = *reinterpret_cast<const u32x4*>(dy + row * a.dys + g0 + NORM_COL(c));
      rz[c] = *reinterpret_cast<const u32x4*>(z + row * a.zs + g0 + NORM_COL(c));
    }
    f32x2 gv[NCHUNK][4], wdy[NCHUNK][4], sig[NCHUNK][4];
    f32x2 s2v = {0.f, 0.f}, t2v = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const f32x2 zf = unpk(rz[c][e]);
        const f32x2 t = zf * (-LOG2E);
        const f32x2 q = f32x2{exp2_fast(t[0]), exp2_fast(t[1])} + 1.f;
        sig[c][e] = f32x2{rcp_fast(q[0]), rcp_fast(q[1])};
        const f32x2 wv = *reinterpret_cast<const f32x2*>(&wsh[NORM_COL(c) + 2 * e]);
        gv[c][e] = unpk(rx[c][e]) * (zf * sig[c][e]);           // x silu(z): what the norm sees
        wdy[c][e] = rlive ? unpk(rd[c][e]) * wv : f32x2{0.f, 0.f};
        s2v = fma_f32x2(gv[c][e], gv[c][e], s2v);
        t2v = fma_f32x2(gv[c][e], wdy[c][e], t2v);
      }
    float s2 = s2v[0] + s2v[1], t2 = t2v[0] + t2v[1];
    row_sum2<WPR>(s2, t2, red, wave, rpar);
    const float rstd = rsqrtf(s2 * inv_n + a.eps);
    const float c1 = rstd * t2 * inv_n;
    // (the staging registers made opaque: the second pass converts the packed inputs again instead of keeping 48 fp32 copies alive)
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
#pragma unroll
      for (int e = 0; e < 4; e++) { OMK_OPAQUE(rx[c][e]); OMK_OPAQUE(rz[c][e]); OMK_OPAQUE(rd[c][e]); }
    if (rlive) {
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        u32x4 ox, oz;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const f32x2 xhat = gv[c][e] * rstd, zf = unpk(rz[c][e]), sg = sig[c][e];
          dwacc[c][e] = fma_f32x2(unpk(rd[c][e]), xhat, dwacc[c][e]);
          const f32x2 ds = (wdy[c][e] - xhat * c1) * rstd * sg;     // grad wrt the normalised input x silu(z), times the sigmoid
          const f32x2 vx = ds * zf, vz = ds * unpk(rx[c][e]) * (1.f + zf * (1.f - sg));
          ox[e] = pack_bf16x2(vx[0], vx[1]);
          oz[e] = pack_bf16x2(vz[0], vz[1]);
        }
        *reinterpret_cast<u32x4*>(dx + rraw * a.dxs + g0 + NORM_COL(c)) = ox;
        *reinterpret_cast<u32x4*>(dz + rraw * a.dzs + g0 + NORM_COL(c)) = oz;
      }
    }
  }
  const int64_t part = bi * RPB + wrow;   // partial rows of this group
  if (a.dw_part) {
#pragma unroll
    for (int c = 0; c < NCHUNK; c++)
#pragma unroll
      for (int e = 0; e < 4; e++) *reinterpret_cast<f32x2*>(&a.dw_part[part * a.cols + g0 + NORM_COL(c) + 2 * e]) = dwacc[c][e];
  }
}

// The same backward for rows of exactly 8 x 64 x 8 = 4096 columns (the 1.3B d_inner) with EIGHT waves per row: one 16-byte vector per lane and
// stream -- half the registers, twice the waves per SIMD.  tools/probe/stream3_probe.hip: 288 -> 258 us for the same bytes and arithmetic; nothing
// else moved this kernel (prefetch, launch shape, barrier, instruction count: profiles/r06_stream_kernels.txt).  512 workgroups: 512 partial rows.
constexpr int NORM_W8_COLS = 8 * 64 * 8, NORM_W8_BLOCKS = 512;
__global__ __launch_bounds__(512) void norm_gated_bwd_w8_kernel(NormBwdArgs a) {
  __shared__ float red[2][8][2];
  __shared__ __attribute__((aligned(16))) float wsh[NORM_W8_COLS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bf16_t* x = (const bf16_t*)a.x;
  const bf16_t* z = (const bf16_t*)a.z;
  const bf16_t* dy = (const bf16_t*)a.dy;
  bf16_t* dx = (bf16_t*)a.dx;
  bf16_t* dz = (bf16_t*)a.dz;
  const int col = (wave * 64 + lane) * 8;
#pragma unroll
  for (int i = 0; i < 8; i++) wsh[col + i] = load_rt(a.w, col + i, a.wdt);
  block_sync();
  constexpr float inv_n = 1.f / (float)NORM_W8_COLS;
  f32x2 dwacc[4];
#pragma unroll
  for (int e = 0; e < 4; e++) dwacc[e] = f32x2{0.f, 0.f};
  auto unpk = [](uint32_t w) -> f32x2 { return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; };
  int par = 0;
  for (int64_t row = blockIdx.x; row < a.rows; row += gridDim.x) {
    u32x4 rx = *reinterpret_cast<const u32x4*>(x + row * a.xs + col);
    u32x4 rd = *reinterpret_cast<const u32x4*>(dy + row * a.dys + col);
    u32x4 rz = *reinterpret_cast<const u32x4*>(z + row * a.zs + col);
    f32x2 gv[4], wdy[4], sig[4];
    f32x2 s2v = {0.f, 0.f}, t2v = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const f32x2 zf = unpk(rz[e]);
      const f32x2 t = zf * (-LOG2E);
      const f32x2 q = f32x2{exp2_fast(t[0]), exp2_fast(t[1])} + 1.f;
      sig[e] = f32x2{rcp_fast(q[0]), rcp_fast(q[1])};
      const f32x2 wv = *reinterpret_cast<const f32x2*>(&wsh[col + 2 * e]);
      gv[e] = unpk(rx[e]) * (zf * sig[e]);           // x silu(z): what the norm sees
      wdy[e] = unpk(rd[e]) * wv;
      s2v = fma_f32x2(gv[e], gv[e], s2v);
      t2v = fma_f32x2(gv[e], wdy[e], t2v);
    }
    float s2 = wave_sum(s2v[0] + s2v[1]), t2 = wave_sum(t2v[0] + t2v[1]);
    if (lane == 0) { red[par][wave][0] = s2; red[par][wave][1] = t2; }
    block_sync();                                      // (the exchange array is double buffered by parity: one barrier per row)
    s2 = 0.f; t2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) { s2 += red[par][q][0]; t2 += red[par][q][1]; }
    par ^= 1;
    const float rstd = rsqrtf(s2 * inv_n + a.eps);
    const float c1 = rstd * t2 * inv_n;
    u32x4 ox, oz;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const f32x2 xhat = gv[e] * rstd, zf = unpk(rz[e]), sg = sig[e];
      dwacc[e] = fma_f32x2(unpk(rd[e]), xhat, dwacc[e]);
      const f32x2 ds = (wdy[e] - xhat * c1) * rstd * sg;
      const f32x2 vx = ds * zf, vz = ds * unpk(rx[e]) * (1.f + zf * (1.f - sg));
      ox[e] = pack_bf16x2(vx[0], vx[1]);
      oz[e] = pack_bf16x2(vz[0], vz[1]);
    }
    *reinterpret_cast<u32x4*>(dx + row * a.dxs + col) = ox;
    *reinterpret_cast<u32x4*>(dz + row * a.dzs + col) = oz;
  }
  if (a.dw_part) {
#pragma unroll
    for (int e = 0; e < 4; e++) *reinterpret_cast<f32x2*>(&a.dw_part[(int64_t)blockIdx.x * a.cols + col + 2 * e]) = dwacc[e];
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
struct VecPlan { int vec, nchunk, wpr; };
// VEC=8 needs 16-byte aligned rows; a segment must fit WPR * NCHUNK * 64 * VEC
static bool plan_vec(int64_t seglen, bool can8, VecPlan* out) {
  if (can8 && seglen % 8 == 0) {
    if (seglen <= 1 * 2 * 64 * 8) { *out = {8, 4, 1}; return true; }     // <= 1024: a wave per row
    // 1025 .. 2048 (the 1.3B d_model): four waves per row, ONE 16-byte vector per lane and stream -- the wave-per-row form holds 32 elements of
    // every stream per lane (add_norm_bwd: 214 registers, two waves per SIMD); see norm_gated_bwd_w8_kernel for what registers cost these kernels
    if (seglen <= 4 * 1 * 64 * 8) { *out = {8, 1, 4}; return true; }
    if (seglen <= 4 * 2 * 64 * 8) { *out = {8, 2, 4}; return true; }     // <= 4096 (the 1.3B gated norm): half the registers
    if (seglen <= 4 * 4 * 64 * 8) { *out = {8, 4, 4}; return true; }     // <= 8192
  }
  if (seglen <= 4 * 32 * 64) { *out = {1, 32, 4}; return true; }         // <= 8192, any alignment
  return false;
}
static int norm_blocks(int64_t rows, int ngroups, const VecPlan& pl) {
  const int rpb = NORM_WAVES / pl.wpr;
  int64_t per_group = (rows + rpb - 1) / rpb;
  int64_t cap = NORM_MAX_BLOCKS / ngroups;
  if (const char* e = getenv("OMK_NORM_BLOCKS")) { const int v = atoi(e); if (v >= 64 && v <= (1 << 20)) cap = v / ngroups; }   // developer A/B
  if (cap < 1) cap = 1;
  if (per_group > cap) per_group = cap;
  if (per_group < 1) per_group = 1;
  return (int)(per_group * ngroups);
}
static int norm_parts(int64_t rows, int ngroups, const VecPlan& pl) {   // partial dw rows per group
  return (norm_blocks(rows, ngroups, pl) / ngroups) * (NORM_WAVES / pl.wpr);
}
static bool rows_ok8(const OmkTensor& t) { return !present(t) || (aligned16(t) && t.stride[1] == 1 && t.stride[0] % 8 == 0); }

#define OMK_PLAN_SWITCH(plan, ...)                                                                                   \
  if (plan.vec == 8 && plan.wpr == 1) { constexpr int VEC = 8, NCHUNK = 4, WPR = 1; __VA_ARGS__; }                   \
  else if (plan.vec == 8 && plan.nchunk == 1) { constexpr int VEC = 8, NCHUNK = 1, WPR = 4; __VA_ARGS__; }           \
  else if (plan.vec == 8 && plan.nchunk == 2) { constexpr int VEC = 8, NCHUNK = 2, WPR = 4; __VA_ARGS__; }           \
  else if (plan.vec == 8) { constexpr int VEC = 8, NCHUNK = 4, WPR = 4; __VA_ARGS__; }                               \
  else { constexpr int VEC = 1, NCHUNK = 32, WPR = 4; __VA_ARGS__; }

static void launch_reduce(const float* part, int nparts, int64_t cols, float* out, omk_stream stream) {
  dim3 rg((unsigned)((cols + RP_COLS - 1) / RP_COLS)), rb(RP_COLS * RP_GROUPS);
  OMK_LAUNCH(reduce_parts_kernel, rg, rb, 0, stream, part, nparts, (int)cols, out);
}

}  // namespace omk

using namespace omk;

extern "C" int omk_add_norm_fwd(const OmkAddNormFwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->x) && present(p->y) && present(p->weight), "add_norm_fwd: x, y, weight required");
  OMK_REQUIRE(p->x.ndim == 2 && p->y.ndim == 2, "add_norm_fwd: x, y must be 2-d (rows, cols)");
  const int64_t rows = p->x.shape[0], cols = p->x.shape[1];
  OMK_REQUIRE(p->y.shape[0] == rows && p->y.shape[1] == cols && p->y.dtype == p->x.dtype, "add_norm_fwd: y mismatch");
  OMK_REQUIRE(p->x.stride[1] == 1 && p->y.stride[1] == 1, "add_norm_fwd: last dim must be contiguous");
  OMK_REQUIRE(numel(p->weight) == cols, "add_norm_fwd: weight size");
  if (present(p->residual)) OMK_REQUIRE(p->residual.shape[0] == rows && p->residual.shape[1] == cols && p->residual.stride[1] == 1, "add_norm_fwd: residual mismatch");
  if (present(p->residual_out)) OMK_REQUIRE(p->residual_out.shape[0] == rows && p->residual_out.shape[1] == cols && p->residual_out.stride[1] == 1, "add_norm_fwd: residual_out mismatch");
  const int xdt = p->x.dtype;
  const int rdt = present(p->residual) ? p->residual.dtype : xdt;
  const int rodt = present(p->residual_out) ? p->residual_out.dtype : xdt;
  OMK_REQUIRE(rdt == xdt || rdt == OMK_F32, "add_norm_fwd: residual dtype must be x's or f32");
  OMK_REQUIRE(rodt == xdt || rodt == OMK_F32, "add_norm_fwd: residual_out dtype must be x's or f32");
  if (rows == 0) return OMK_OK;
  VecPlan plan;
  bool can8 = rows_ok8(p->x) && rows_ok8(p->y) && rows_ok8(p->residual) && rows_ok8(p->residual_out);
  if (!plan_vec(cols, can8, &plan)) return fail(OMK_EUNSUPPORTED, "add_norm_fwd: cols=%lld too large", (long long)cols);
  NormArgs a = {};
  a.x = p->x.data; a.res = p->residual.data; a.w = p->weight.data; a.b = p->bias.data; a.y = p->y.data;
  a.ro = p->residual_out.data; a.rstd = (float*)p->rstd.data; a.mean = p->is_rms_norm ? nullptr : (float*)p->mean.data;
  a.xs = p->x.stride[0]; a.rs = present(p->residual) ? p->residual.stride[0] : 0; a.ys = p->y.stride[0];
  a.ros = present(p->residual_out) ? p->residual_out.stride[0] : 0;
  a.rows = rows; a.cols = (int)cols; a.ngroups = 1; a.wdt = p->weight.dtype; a.bdt = p->bias.dtype; a.eps = p->eps; a.rms = p->is_rms_norm;
  // short-lived workgroups (see omk_norm_gated_fwd): two block rows each, behind a prologue of 16-byte requests -- when every lane's columns exist and
  // the weight / bias rows are aligned; otherwise the persistent grid of rounds 1 - 5
  a.wvec = plan.vec == 8 && cols == (int64_t)plan.wpr * plan.nchunk * 64 * 8 && ((uintptr_t)p->weight.data & 15) == 0 &&
           (!present(p->bias) || ((uintptr_t)p->bias.data & 15) == 0) && !getenv("OMK_NORM_PERSISTENT");
  int nblk = norm_blocks(rows, 1, plan);
  if (a.wvec) {
    const int rpb = NORM_WAVES / plan.wpr;
    int64_t per = (rows + rpb - 1) / rpb;
    if (per > 2048) per = (per + 1) / 2;
    if (const char* e = getenv("OMK_NORM_BLOCKS")) { const int v = atoi(e); if (v >= 64 && v <= (1 << 20) && v < per) per = v; }   // developer A/B
    nblk = (int)per;
  }
  dim3 grid(nblk), block(NORM_THREADS);
#define LAUNCH_ADD(TX, TR, TRO) OMK_PLAN_SWITCH(plan, OMK_LAUNCH((add_norm_fwd_kernel<TX, TR, TRO, VEC, NCHUNK, WPR>), grid, block, 0, stream, a))
  OMK_DISPATCH_DTYPE(xdt, TX, {
    if (rdt == xdt && rodt == xdt) { LAUNCH_ADD(TX, TX, TX); }
    else if (rdt == xdt) { LAUNCH_ADD(TX, TX, float); }
    else if (rodt == xdt) { LAUNCH_ADD(TX, float, TX); }
    else { LAUNCH_ADD(TX, float, float); }
  });
#undef LAUNCH_ADD
  return finish_launch("add_norm_fwd");
}

static bool add_bwd_plan(const OmkAddNormBwd* p, VecPlan* plan) {
  bool can8 = rows_ok8(p->dy) && rows_ok8(p->xsum) && rows_ok8(p->dx) && rows_ok8(p->dresidual_out) && rows_ok8(p->dresidual_in);
  return plan_vec(p->dy.shape[1], can8, plan);
}

extern "C" size_t omk_add_norm_bwd_workspace_bytes(const OmkAddNormBwd* p) {
  if (!p) return 0;
  VecPlan plan;
  if (!add_bwd_plan(p, &plan)) return 0;
  return (size_t)norm_blocks(p->dy.shape[0], 1, plan) * p->dy.shape[1] * 4 * 2;   // (one partial row per workgroup, dw and db)
}

extern "C" int omk_add_norm_bwd(const OmkAddNormBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->dy) && present(p->xsum) && present(p->weight) && present(p->rstd) && present(p->dx),
              "add_norm_bwd: dy, xsum, weight, rstd, dx required");
  const int64_t rows = p->dy.shape[0], cols = p->dy.shape[1];
  OMK_REQUIRE(p->dx.dtype == p->dy.dtype, "add_norm_bwd: dx dtype must equal dy dtype");
  OMK_REQUIRE(!present(p->dresidual_out) || p->dresidual_out.dtype == p->xsum.dtype, "add_norm_bwd: dresidual_out dtype must equal xsum dtype");
  OMK_REQUIRE(p->xsum.dtype == p->dy.dtype || p->xsum.dtype == OMK_F32, "add_norm_bwd: xsum dtype");
  OMK_REQUIRE(!present(p->dweight) || p->dweight.dtype == OMK_F32, "add_norm_bwd: dweight must be f32");
  if (rows == 0) return OMK_OK;
  VecPlan plan;
  if (!add_bwd_plan(p, &plan)) return fail(OMK_EUNSUPPORTED, "add_norm_bwd: cols too large");
  OMK_REQUIRE(p->workspace && p->workspace_bytes >= omk_add_norm_bwd_workspace_bytes(p), "add_norm_bwd: workspace too small");
  const int nparts = norm_blocks(rows, 1, plan);
  NormBwdArgs a = {};
  a.dy = p->dy.data; a.dro = p->dresidual_out.data; a.xsum = p->xsum.data; a.w = p->weight.data;
  a.rstd = (const float*)p->rstd.data; a.mean = p->is_rms_norm ? nullptr : (const float*)p->mean.data;
  a.dx = p->dx.data; a.dri = p->dresidual_in.data;
  // a frozen weight (dweight absent): no partial rows, no reduction launch
  a.dw_part = present(p->dweight) ? (float*)p->workspace : nullptr;
  a.db_part = (p->has_bias && present(p->dbias)) ? (float*)p->workspace + (size_t)nparts * cols : nullptr;
  a.dys = p->dy.stride[0]; a.dros = present(p->dresidual_out) ? p->dresidual_out.stride[0] : 0; a.xss = p->xsum.stride[0];
  a.dxs = p->dx.stride[0]; a.dris = present(p->dresidual_in) ? p->dresidual_in.stride[0] : 0;
  a.rows = rows; a.cols = (int)cols; a.ngroups = 1; a.wdt = p->weight.dtype; a.rms = p->is_rms_norm;
  const int xdt = p->dy.dtype, sdt = p->xsum.dtype;
  const int ridt = present(p->dresidual_in) ? p->dresidual_in.dtype : xdt;
  OMK_REQUIRE(ridt == xdt || ridt == OMK_F32, "add_norm_bwd: dresidual_in dtype");
  dim3 grid(norm_blocks(rows, 1, plan)), block(NORM_THREADS);
#define LAUNCH_B(TX, TS, TRI) OMK_PLAN_SWITCH(plan, OMK_LAUNCH((add_norm_bwd_kernel<TX, TS, TRI, VEC, NCHUNK, WPR>), grid, block, 0, stream, a))
  OMK_DISPATCH_DTYPE(xdt, TX, {
    if (sdt == xdt && ridt == xdt) { LAUNCH_B(TX, TX, TX); }
    else if (sdt == xdt) { LAUNCH_B(TX, TX, float); }
    else if (ridt == xdt) { LAUNCH_B(TX, float, TX); }
    else { LAUNCH_B(TX, float, float); }
  });
#undef LAUNCH_B
  if (a.dw_part) launch_reduce(a.dw_part, nparts, cols, (float*)p->dweight.data, stream);
  if (a.db_part) launch_reduce(a.db_part, nparts, cols, (float*)p->dbias.data, stream);
  return finish_launch("add_norm_bwd");
}

extern "C" int omk_norm_gated_fwd(const OmkNormGatedFwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->x) && present(p->y) && present(p->weight), "norm_gated_fwd: x, y, weight required");
  OMK_REQUIRE(p->x.ndim == 2 && p->y.ndim == 2 && p->x.stride[1] == 1 && p->y.stride[1] == 1, "norm_gated_fwd: (rows, cols) with contiguous cols");
  const int64_t rows = p->x.shape[0], cols = p->x.shape[1];
  const int64_t gs = p->group_size > 0 ? p->group_size : cols;
  OMK_REQUIRE(cols % gs == 0, "norm_gated_fwd: group_size must divide cols");
  OMK_REQUIRE(p->y.dtype == p->x.dtype && (!present(p->z) || (p->z.dtype == p->x.dtype && p->z.stride[1] == 1)), "norm_gated_fwd: dtype/stride mismatch");
  if (rows == 0) return OMK_OK;
  VecPlan plan;
  bool can8 = rows_ok8(p->x) && rows_ok8(p->y) && rows_ok8(p->z) && gs % 8 == 0;
  if (!plan_vec(gs, can8, &plan)) return fail(OMK_EUNSUPPORTED, "norm_gated_fwd: group too large");
  NormArgs a = {};
  a.x = p->x.data; a.z = p->z.data; a.w = p->weight.data; a.b = p->bias.data; a.y = p->y.data; a.rstd = (float*)p->rstd.data;
  a.xs = p->x.stride[0]; a.zs = present(p->z) ? p->z.stride[0] : 0; a.ys = p->y.stride[0];
  a.rows = rows; a.cols = (int)cols; a.ngroups = (int)(cols / gs); a.wdt = p->weight.dtype; a.bdt = p->bias.dtype; a.eps = p->eps;
  a.rms = 1; a.norm_before_gate = p->norm_before_gate;
  dim3 grid(norm_blocks(rows, a.ngroups, plan)), block(NORM_THREADS);
  // the reference's mode on full segments (every lane's columns exist): the software-pipelined kernel
  const bool lean = present(p->z) && !present(p->bias) && !p->norm_before_gate && plan.vec == 8 && gs == (int64_t)plan.wpr * plan.nchunk * 64 * 8 &&
                    p->x.dtype == OMK_BF16 && !getenv("OMK_NORM_NO_LEAN");
  if (lean) {
    if (const char* e = getenv("OMK_NORM_ABL")) a.rms |= atoi(e) & 62;
    // SHORT-LIVED workgroups: two block rows each.  tools/probe/stream3_probe.hip (profiles/r06_stream_kernels.txt): the same three streams run
    // at 5.7 TB/s from workgroups that live for one or two rows and at 5.1 - 5.3 from 1024 - 2048 persistent ones, whatever their row map
    // (strided, contiguous ranges, an atomic queue) -- the rate falls steadily with the rows a workgroup walks.  The weight row is requested
    // in 16-byte pieces so that the per-workgroup prologue stays small.
    const bool wvec = ((uintptr_t)p->weight.data & 15) == 0 && (p->weight.ndim < 1 || p->weight.stride[p->weight.ndim - 1] == 1);
    const int rpb = NORM_WAVES / plan.wpr;
    int64_t per_group = (rows + rpb - 1) / rpb;
    if (per_group > 2048) per_group = (per_group + 1) / 2;
    if (const char* e = getenv("OMK_NORM_BLOCKS")) { const int v = atoi(e); if (v >= 64 && v <= (1 << 20) && v / a.ngroups < per_group) per_group = v / a.ngroups; }   // developer A/B
    const dim3 lgrid((unsigned)(per_group * a.ngroups));
#define OMK_LEAN_FWD(NC_, WPR_) do { if (wvec) OMK_LAUNCH((norm_gated_fwd_lean_kernel<bf16_t, 8, NC_, WPR_, true>), lgrid, block, 0, stream, a); \
      else OMK_LAUNCH((norm_gated_fwd_lean_kernel<bf16_t, 8, NC_, WPR_, false>), lgrid, block, 0, stream, a); } while (0)
    if (plan.wpr == 1) OMK_LEAN_FWD(4, 1);
    else if (plan.nchunk == 1) OMK_LEAN_FWD(1, 4);
    else if (plan.nchunk == 2) OMK_LEAN_FWD(2, 4);
    else OMK_LEAN_FWD(4, 4);
#undef OMK_LEAN_FWD
    return finish_launch("norm_gated_fwd");
  }
  OMK_DISPATCH_DTYPE(p->x.dtype, TX, OMK_PLAN_SWITCH(plan, OMK_LAUNCH((norm_gated_fwd_kernel<TX, VEC, NCHUNK, WPR>), grid, block, 0, stream, a)));
  return finish_launch("norm_gated_fwd");
}

static bool gated_bwd_plan(const OmkNormGatedBwd* p, VecPlan* plan, int* ng) {
  const int64_t cols = p->x.shape[1];
  const int64_t gs = p->group_size > 0 ? p->group_size : cols;
  if (gs <= 0 || cols % gs) return false;
  *ng = (int)(cols / gs);
  bool can8 = rows_ok8(p->x) && rows_ok8(p->dy) && rows_ok8(p->z) && rows_ok8(p->dx) && rows_ok8(p->dz) && gs % 8 == 0;
  return plan_vec(gs, can8, plan);
}

extern "C" size_t omk_norm_gated_bwd_workspace_bytes(const OmkNormGatedBwd* p) {
  if (!p) return 0;
  VecPlan plan; int ng;
  if (!gated_bwd_plan(p, &plan, &ng)) return 0;
  const size_t general = (size_t)norm_parts(p->x.shape[0], ng, plan) * p->x.shape[1] * 4;
  const size_t w8 = (size_t)NORM_W8_BLOCKS * p->x.shape[1] * 4;     // (norm_gated_bwd_w8_kernel: never more than the general form needs for >= 512 rows)
  return general > w8 ? general : w8;
}

extern "C" int omk_norm_gated_bwd(const OmkNormGatedBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->dy) && present(p->x) && present(p->weight) && present(p->dx), "norm_gated_bwd: dy, x, weight, dx required");
  const int64_t rows = p->x.shape[0], cols = p->x.shape[1];
  OMK_REQUIRE(!present(p->dweight) || p->dweight.dtype == OMK_F32, "norm_gated_bwd: dweight must be f32");
  OMK_REQUIRE(p->dy.dtype == p->x.dtype && p->dx.dtype == p->x.dtype, "norm_gated_bwd: dtype mismatch");
  if (present(p->z)) OMK_REQUIRE(present(p->dz) && p->z.dtype == p->x.dtype && p->dz.dtype == p->x.dtype, "norm_gated_bwd: z/dz");
  if (rows == 0) return OMK_OK;
  VecPlan plan; int ng;
  if (!gated_bwd_plan(p, &plan, &ng)) return fail(OMK_EUNSUPPORTED, "norm_gated_bwd: unsupported group size");
  OMK_REQUIRE(p->workspace && p->workspace_bytes >= omk_norm_gated_bwd_workspace_bytes(p), "norm_gated_bwd: workspace too small");
  const int nparts = norm_parts(rows, ng, plan);
  NormBwdArgs a = {};
  a.dy = p->dy.data; a.x = p->x.data; a.z = p->z.data; a.w = p->weight.data; a.dx = p->dx.data; a.dz = p->dz.data;
  a.dw_part = present(p->dweight) ? (float*)p->workspace : nullptr;   // frozen weight: no partial rows, no reduction launch
  a.dys = p->dy.stride[0]; a.xs = p->x.stride[0]; a.zs = present(p->z) ? p->z.stride[0] : 0; a.dxs = p->dx.stride[0];
  a.dzs = present(p->dz) ? p->dz.stride[0] : 0;
  a.rows = rows; a.cols = (int)cols; a.ngroups = ng; a.wdt = p->weight.dtype; a.rms = 1; a.eps = p->eps; a.norm_before_gate = p->norm_before_gate;
  dim3 grid(norm_blocks(rows, ng, plan)), block(NORM_THREADS);
  const int64_t gsz = cols / ng;
  const bool lean = present(p->z) && present(p->dz) && !p->norm_before_gate && plan.vec == 8 && gsz == (int64_t)plan.wpr * plan.nchunk * 64 * 8 &&
                    p->x.dtype == OMK_BF16 && !getenv("OMK_NORM_NO_LEAN");
  const bool w8 = lean && ng == 1 && cols == NORM_W8_COLS && !getenv("OMK_NORM_BWD_W8_OFF");
  if (w8) {
    int nb = rows < NORM_W8_BLOCKS ? (int)rows : NORM_W8_BLOCKS;
    OMK_LAUNCH(norm_gated_bwd_w8_kernel, dim3((unsigned)nb), dim3(512), 0, stream, a);
    if (a.dw_part) launch_reduce(a.dw_part, nb, cols, (float*)p->dweight.data, stream);
    return finish_launch("norm_gated_bwd");
  }
  if (lean) {
    // (measured and not kept, profiles/r06_stream_kernels.txt: the next row's requests in a second staging set -- 150 registers, three waves per
    // SIMD: 300 us against 290; the five streams with trivial arithmetic take 244 us on the same box, one-shot or persistent alike)
    if (plan.wpr == 1) OMK_LAUNCH((norm_gated_bwd_lean_kernel<bf16_t, 8, 4, 1>), grid, block, 0, stream, a);
    else if (plan.nchunk == 1) OMK_LAUNCH((norm_gated_bwd_lean_kernel<bf16_t, 8, 1, 4>), grid, block, 0, stream, a);
    else if (plan.nchunk == 2) OMK_LAUNCH((norm_gated_bwd_lean_kernel<bf16_t, 8, 2, 4>), grid, block, 0, stream, a);
    else OMK_LAUNCH((norm_gated_bwd_lean_kernel<bf16_t, 8, 4, 4>), grid, block, 0, stream, a);
  } else
  OMK_DISPATCH_DTYPE(p->x.dtype, TX, OMK_PLAN_SWITCH(plan, OMK_LAUNCH((norm_gated_bwd_kernel<TX, VEC, NCHUNK, WPR>), grid, block, 0, stream, a)));
  if (a.dw_part) launch_reduce(a.dw_part, nparts, cols, (float*)p->dweight.data, stream);
  return finish_launch("norm_gated_bwd");
}
