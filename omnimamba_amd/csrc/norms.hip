// norms.hip -- fused residual-add + RMSNorm/LayerNorm (layer_norm_fn) and gated RMSNorm (Mamba2.norm), fwd + bwd.
//
// Both are pure HBM streaming ops: one 64-lane wave owns one row (or one
// (row, group) segment), keeps it in registers as NCHUNK x VEC floats per lane, reduces with wave shuffles
// (no LDS, no block barrier) and touches every byte exactly once: 16-byte loads/stores per lane.
// Algorithmic bytes per row of `cols`: add+norm fwd = cols * (sx + sres_in + sx + sres_out); gated fwd = 3 * cols * sx.
#include "omk_common.h"

namespace omk {

constexpr int NORM_THREADS = 256;
constexpr int NORM_WAVES = NORM_THREADS / 64;

template <class T, int VEC>
__device__ __forceinline__ void ld(const T* p, float (&o)[VEC]) { load_vec<T, VEC>(p, o); }
template <class T, int VEC>
__device__ __forceinline__ void st(T* p, const float (&o)[VEC]) { store_vec<T, VEC>(p, o); }

struct NormArgs {
  const void* x; const void* res; const void* w; const void* b; const void* z;
  void* y; void* ro; float* rstd; float* mean;
  int64_t xs, rs, ys, ros, zs;      // row strides (elements)
  int64_t rows; int cols; int ngroups; int wdt, bdt; float eps; int rms; int norm_before_gate;
};

// ---------------------------------------------------------------------------------------------------------
// forward: y = norm(x + residual) * w + b ; residual_out = x + residual
// ---------------------------------------------------------------------------------------------------------
template <class TX, class TR, class TRO, int VEC, int NCHUNK>
__global__ __launch_bounds__(NORM_THREADS) void add_norm_fwd_kernel(NormArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const TX* x = (const TX*)a.x;
  const TR* res = (const TR*)a.res;
  TX* y = (TX*)a.y;
  TRO* ro = (TRO*)a.ro;
  const float inv_n = 1.f / (float)a.cols;
  for (int64_t row = (int64_t)blockIdx.x * NORM_WAVES + wave; row < a.rows; row += (int64_t)gridDim.x * NORM_WAVES) {
    float v[NCHUNK][VEC];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      int col = (c * 64 + lane) * VEC;
      if (col < a.cols) {
        ld<TX, VEC>(x + row * a.xs + col, v[c]);
        if (res) {
          float r[VEC];
          ld<TR, VEC>(res + row * a.rs + col, r);
#pragma unroll
          for (int i = 0; i < VEC; i++) v[c][i] += r[i];
        }
        if (ro) st<TRO, VEC>(ro + row * a.ros + col, v[c]);
#pragma unroll
        for (int i = 0; i < VEC; i++) { s1 += v[c][i]; s2 += v[c][i] * v[c][i]; }
      } else {
#pragma unroll
        for (int i = 0; i < VEC; i++) v[c][i] = 0.f;
      }
    }
    float mu = 0.f, var;
    if (a.rms) {
      var = wave_sum(s2) * inv_n;
    } else {
      mu = wave_sum(s1) * inv_n;
      float d2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        int col = (c * 64 + lane) * VEC;
        if (col < a.cols) {
#pragma unroll
          for (int i = 0; i < VEC; i++) { float d = v[c][i] - mu; d2 += d * d; }
        }
      }
      var = wave_sum(d2) * inv_n;
    }
    float rstd = rsqrtf(var + a.eps);
    if (lane == 0) {
      if (a.rstd) a.rstd[row] = rstd;
      if (a.mean) a.mean[row] = mu;
    }
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      int col = (c * 64 + lane) * VEC;
      if (col < a.cols) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          float w = load_rt(a.w, col + i, a.wdt);
          o[i] = (v[c][i] - mu) * rstd * w;
          if (a.b) o[i] += load_rt(a.b, col + i, a.bdt);
        }
        st<TX, VEC>(y + row * a.ys + col, o);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward of the above.  dx = (wdy - xhat*c1 - c2) * rstd + dresidual_out ; dw += dy * xhat ; db += dy
// ---------------------------------------------------------------------------------------------------------
struct NormBwdArgs {
  const void* dy; const void* dro; const void* xsum; const void* w; const float* rstd; const float* mean;
  const void* x; const void* z;     // gated only
  void* dx; void* dri; void* dz;
  float* dw_part; float* db_part;   // [nparts][cols]
  int64_t dys, dros, xss, dxs, dris, xs, zs, dzs;
  int64_t rows; int cols; int ngroups; int wdt; int rms; float eps; int norm_before_gate;
};

template <class TX, class TS, class TRI, int VEC, int NCHUNK>
__global__ __launch_bounds__(NORM_THREADS) void add_norm_bwd_kernel(NormBwdArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const TX* dy = (const TX*)a.dy;
  const TS* dro = (const TS*)a.dro;   // grad of residual_out has residual_out's dtype (= xsum's)
  const TS* xsum = (const TS*)a.xsum;
  TX* dx = (TX*)a.dx;
  TRI* dri = (TRI*)a.dri;
  const float inv_n = 1.f / (float)a.cols;
  float dwacc[NCHUNK][VEC], dbacc[NCHUNK][VEC];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++)
#pragma unroll
    for (int i = 0; i < VEC; i++) { dwacc[c][i] = 0.f; dbacc[c][i] = 0.f; }
  for (int64_t row = (int64_t)blockIdx.x * NORM_WAVES + wave; row < a.rows; row += (int64_t)gridDim.x * NORM_WAVES) {
    float xh[NCHUNK][VEC], wdy[NCHUNK][VEC];
    const float rstd = a.rstd[row];
    const float mu = a.mean ? a.mean[row] : 0.f;
    float c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      int col = (c * 64 + lane) * VEC;
      if (col < a.cols) {
        float g[VEC];
        ld<TS, VEC>(xsum + row * a.xss + col, xh[c]);
        ld<TX, VEC>(dy + row * a.dys + col, g);
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          xh[c][i] = (xh[c][i] - mu) * rstd;
          wdy[c][i] = g[i] * load_rt(a.w, col + i, a.wdt);
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
    c1 = wave_sum(c1) * inv_n;
    c2 = a.rms ? 0.f : wave_sum(c2) * inv_n;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      int col = (c * 64 + lane) * VEC;
      if (col < a.cols) {
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
  const int64_t part = (int64_t)blockIdx.x * NORM_WAVES + wave;
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) {
    int col = (c * 64 + lane) * VEC;
    if (col < a.cols) {
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        a.dw_part[part * a.cols + col + i] = dwacc[c][i];
        if (a.db_part) a.db_part[part * a.cols + col + i] = dbacc[c][i];
      }
    }
  }
}

__global__ void reduce_parts_kernel(const float* part, int nparts, int cols, float* out) {
  int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= cols) return;
  float s = 0.f;
  for (int p = 0; p < nparts; p++) s += part[(int64_t)p * cols + col];
  out[col] = s;
}

// ---------------------------------------------------------------------------------------------------------
// gated RMSNorm.  norm_before_gate = 0 (reference): y = rmsnorm(x * silu(z)) * w ; 1: y = rmsnorm(x) * w * silu(z)
// one wave per (row, group) segment of group_size = cols / ngroups lanes
// ---------------------------------------------------------------------------------------------------------
template <class TX, int VEC, int NCHUNK>
__global__ __launch_bounds__(NORM_THREADS) void norm_gated_fwd_kernel(NormArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const TX* x = (const TX*)a.x;
  const TX* z = (const TX*)a.z;
  TX* y = (TX*)a.y;
  const int gs = a.cols / a.ngroups;
  const float inv_n = 1.f / (float)gs;
  const int64_t nseg = a.rows * a.ngroups;
  for (int64_t seg = (int64_t)blockIdx.x * NORM_WAVES + wave; seg < nseg; seg += (int64_t)gridDim.x * NORM_WAVES) {
    const int64_t row = seg / a.ngroups;
    const int g0 = (int)(seg % a.ngroups) * gs;
    float v[NCHUNK][VEC], sz[NCHUNK][VEC];
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      int col = (c * 64 + lane) * VEC;
      if (col < gs) {
        ld<TX, VEC>(x + row * a.xs + g0 + col, v[c]);
        if (z) {
          ld<TX, VEC>(z + row * a.zs + g0 + col, sz[c]);
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            sz[c][i] = silu_f(sz[c][i]);
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
    float rstd = rsqrtf(wave_sum(s2) * inv_n + a.eps);
    if (lane == 0 && a.rstd) a.rstd[seg] = rstd;
#pragma unroll
    for (int c = 0; c < NCHUNK; c++) {
      int col = (c * 64 + lane) * VEC;
      if (col < gs) {
        float o[VEC];
#pragma unroll
        for (int i = 0; i < VEC; i++) {
          o[i] = v[c][i] * rstd * load_rt(a.w, g0 + col + i, a.wdt);
          if (a.b) o[i] += load_rt(a.b, g0 + col + i, a.bdt);
          if (z && a.norm_before_gate) o[i] *= sz[c][i];
        }
        st<TX, VEC>(y + row * a.ys + g0 + col, o);
      }
    }
  }
}

template <class TX, int VEC, int NCHUNK>
__global__ __launch_bounds__(NORM_THREADS) void norm_gated_bwd_kernel(NormBwdArgs a) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const TX* x = (const TX*)a.x;
  const TX* z = (const TX*)a.z;
  const TX* dy = (const TX*)a.dy;
  TX* dx = (TX*)a.dx;
  TX* dz = (TX*)a.dz;
  const int gs = a.cols / a.ngroups;
  const float inv_n = 1.f / (float)gs;
  const int64_t nseg = a.rows * a.ngroups;
  const int my_g = -1;
  (void)my_g;
  // a wave may visit segments of different groups, so dw partials are indexed by absolute column through a
  // per-wave accumulation buffer in registers only when ngroups == 1; otherwise accumulate per visited segment.
  float dwacc[NCHUNK][VEC];
#pragma unroll
  for (int c = 0; c < NCHUNK; c++)
#pragma unroll
    for (int i = 0; i < VEC; i++) dwacc[c][i] = 0.f;
  const int64_t part = (int64_t)blockIdx.x * NORM_WAVES + wave;
  // iterate so that every wave stays inside ONE group: segment index = row * ngroups + grp, waves are striped over rows
  const int grp = (int)(part % a.ngroups);
  const int64_t wave_in_grp = part / a.ngroups, waves_per_grp = ((int64_t)gridDim.x * NORM_WAVES) / a.ngroups;
  const int g0 = grp * gs;
  if (wave_in_grp < waves_per_grp) {
    for (int64_t row = wave_in_grp; row < a.rows; row += waves_per_grp) {
      float xv[NCHUNK][VEC], zv[NCHUNK][VEC], gv[NCHUNK][VEC], wdy[NCHUNK][VEC];
      float s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        int col = (c * 64 + lane) * VEC;
        if (col < gs) {
          ld<TX, VEC>(x + row * a.xs + g0 + col, xv[c]);
          ld<TX, VEC>(dy + row * a.dys + g0 + col, wdy[c]);
          if (z) ld<TX, VEC>(z + row * a.zs + g0 + col, zv[c]);
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            float sg = z ? silu_f(zv[c][i]) : 1.f;
            gv[c][i] = (z && !a.norm_before_gate) ? xv[c][i] * sg : xv[c][i];   // what gets normalised
            s2 += gv[c][i] * gv[c][i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < VEC; i++) { xv[c][i] = 0.f; zv[c][i] = 0.f; gv[c][i] = 0.f; wdy[c][i] = 0.f; }
        }
      }
      const float rstd = rsqrtf(wave_sum(s2) * inv_n + a.eps);
      float c1 = 0.f;
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        int col = (c * 64 + lane) * VEC;
        if (col < gs) {
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            float w = load_rt(a.w, g0 + col + i, a.wdt);
            float dyv = wdy[c][i];
            float xhat = gv[c][i] * rstd;
            float sg = z ? silu_f(zv[c][i]) : 1.f;
            if (z && a.norm_before_gate) {
              // y = xhat*w*silu(z): dz uses dy*xhat*w, the norm sees dy*silu(z)
              float sig = sigmoid_f(zv[c][i]);
              zv[c][i] = dyv * xhat * w * sig * (1.f + zv[c][i] * (1.f - sig));   // final dz
              dyv *= sg;
            }
            dwacc[c][i] += dyv * xhat;
            wdy[c][i] = dyv * w;
            c1 += xhat * wdy[c][i];
            gv[c][i] = xhat;
          }
        }
      }
      c1 = wave_sum(c1) * inv_n;
#pragma unroll
      for (int c = 0; c < NCHUNK; c++) {
        int col = (c * 64 + lane) * VEC;
        if (col < gs) {
          float ox[VEC], oz[VEC];
#pragma unroll
          for (int i = 0; i < VEC; i++) {
            float dg = (wdy[c][i] - gv[c][i] * c1) * rstd;   // grad wrt the normalised input
            if (z && !a.norm_before_gate) {
              float sig = sigmoid_f(zv[c][i]);
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
  }
#pragma unroll
  for (int c = 0; c < NCHUNK; c++) {
    int col = (c * 64 + lane) * VEC;
    if (col < gs && wave_in_grp < waves_per_grp) {
#pragma unroll
      for (int i = 0; i < VEC; i++) a.dw_part[wave_in_grp * a.cols + g0 + col + i] = dwacc[c][i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
static int norm_grid(int64_t nseg) {
  int64_t g = (nseg + NORM_WAVES - 1) / NORM_WAVES;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

struct VecPlan { int vec, nchunk; };
// VEC=8 needs 16-byte aligned rows; segment length must fit NCHUNK*64*VEC
static bool plan_vec(int64_t seglen, bool can8, VecPlan* out) {
  if (can8 && seglen % 8 == 0) {
    if (seglen <= 4 * 64 * 8) { *out = {8, 4}; return true; }
    if (seglen <= 16 * 64 * 8) { *out = {8, 16}; return true; }
  }
  if (seglen <= 32 * 64) { *out = {1, 32}; return true; }
  return false;
}
static bool rows_ok8(const OmkTensor& t) { return !present(t) || (aligned16(t) && t.stride[1] == 1 && t.stride[0] % 8 == 0); }

#define OMK_PLAN_SWITCH(plan, ...)                                                 \
  if (plan.vec == 8 && plan.nchunk == 4) { constexpr int VEC = 8, NCHUNK = 4; __VA_ARGS__; }        \
  else if (plan.vec == 8 && plan.nchunk == 16) { constexpr int VEC = 8, NCHUNK = 16; __VA_ARGS__; } \
  else { constexpr int VEC = 1, NCHUNK = 32; __VA_ARGS__; }

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
  dim3 grid(norm_grid(rows)), block(NORM_THREADS);
#define LAUNCH_ADD(TX, TR, TRO) OMK_PLAN_SWITCH(plan, OMK_LAUNCH((add_norm_fwd_kernel<TX, TR, TRO, VEC, NCHUNK>), grid, block, 0, stream, a))
  OMK_DISPATCH_DTYPE(xdt, TX, {
    if (rdt == xdt && rodt == xdt) { LAUNCH_ADD(TX, TX, TX); }
    else if (rdt == xdt) { LAUNCH_ADD(TX, TX, float); }
    else if (rodt == xdt) { LAUNCH_ADD(TX, float, TX); }
    else { LAUNCH_ADD(TX, float, float); }
  });
#undef LAUNCH_ADD
  return finish_launch("add_norm_fwd");
}

static int add_norm_bwd_parts(int64_t rows) { return norm_grid(rows) * NORM_WAVES; }

extern "C" size_t omk_add_norm_bwd_workspace_bytes(const OmkAddNormBwd* p) {
  if (!p) return 0;
  return (size_t)add_norm_bwd_parts(p->dy.shape[0]) * p->dy.shape[1] * 4 * 2;
}

extern "C" int omk_add_norm_bwd(const OmkAddNormBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->dy) && present(p->xsum) && present(p->weight) && present(p->rstd) && present(p->dx) && present(p->dweight),
              "add_norm_bwd: dy, xsum, weight, rstd, dx, dweight required");
  const int64_t rows = p->dy.shape[0], cols = p->dy.shape[1];
  OMK_REQUIRE(p->dx.dtype == p->dy.dtype, "add_norm_bwd: dx dtype must equal dy dtype");
  OMK_REQUIRE(!present(p->dresidual_out) || p->dresidual_out.dtype == p->xsum.dtype, "add_norm_bwd: dresidual_out dtype must equal xsum dtype");
  OMK_REQUIRE(p->xsum.dtype == p->dy.dtype || p->xsum.dtype == OMK_F32, "add_norm_bwd: xsum dtype");
  OMK_REQUIRE(p->workspace_bytes >= omk_add_norm_bwd_workspace_bytes(p) && p->workspace, "add_norm_bwd: workspace too small");
  OMK_REQUIRE(p->dweight.dtype == OMK_F32, "add_norm_bwd: dweight must be f32");
  if (rows == 0) return OMK_OK;
  VecPlan plan;
  bool can8 = rows_ok8(p->dy) && rows_ok8(p->xsum) && rows_ok8(p->dx) && rows_ok8(p->dresidual_out) && rows_ok8(p->dresidual_in);
  if (!plan_vec(cols, can8, &plan)) return fail(OMK_EUNSUPPORTED, "add_norm_bwd: cols too large");
  const int nparts = add_norm_bwd_parts(rows);
  NormBwdArgs a = {};
  a.dy = p->dy.data; a.dro = p->dresidual_out.data; a.xsum = p->xsum.data; a.w = p->weight.data;
  a.rstd = (const float*)p->rstd.data; a.mean = p->is_rms_norm ? nullptr : (const float*)p->mean.data;
  a.dx = p->dx.data; a.dri = p->dresidual_in.data;
  a.dw_part = (float*)p->workspace; a.db_part = p->has_bias ? a.dw_part + (size_t)nparts * cols : nullptr;
  a.dys = p->dy.stride[0]; a.dros = present(p->dresidual_out) ? p->dresidual_out.stride[0] : 0; a.xss = p->xsum.stride[0];
  a.dxs = p->dx.stride[0]; a.dris = present(p->dresidual_in) ? p->dresidual_in.stride[0] : 0;
  a.rows = rows; a.cols = (int)cols; a.ngroups = 1; a.wdt = p->weight.dtype; a.rms = p->is_rms_norm;
  const int xdt = p->dy.dtype, sdt = p->xsum.dtype;
  const int ridt = present(p->dresidual_in) ? p->dresidual_in.dtype : xdt;
  OMK_REQUIRE(ridt == xdt || ridt == OMK_F32, "add_norm_bwd: dresidual_in dtype");
  dim3 grid(norm_grid(rows)), block(NORM_THREADS);
#define LAUNCH_B(TX, TS, TRI) OMK_PLAN_SWITCH(plan, OMK_LAUNCH((add_norm_bwd_kernel<TX, TS, TRI, VEC, NCHUNK>), grid, block, 0, stream, a))
  OMK_DISPATCH_DTYPE(xdt, TX, {
    if (sdt == xdt && ridt == xdt) { LAUNCH_B(TX, TX, TX); }
    else if (sdt == xdt) { LAUNCH_B(TX, TX, float); }
    else if (ridt == xdt) { LAUNCH_B(TX, float, TX); }
    else { LAUNCH_B(TX, float, float); }
  });
#undef LAUNCH_B
  dim3 rg((unsigned)((cols + 255) / 256)), rb(256);
  OMK_LAUNCH(reduce_parts_kernel, rg, rb, 0, stream, (const float*)a.dw_part, nparts, (int)cols, (float*)p->dweight.data);
  if (p->has_bias && present(p->dbias))
    OMK_LAUNCH(reduce_parts_kernel, rg, rb, 0, stream, (const float*)a.db_part, nparts, (int)cols, (float*)p->dbias.data);
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
  dim3 grid(norm_grid(rows * a.ngroups)), block(NORM_THREADS);
  OMK_DISPATCH_DTYPE(p->x.dtype, TX, OMK_PLAN_SWITCH(plan, OMK_LAUNCH((norm_gated_fwd_kernel<TX, VEC, NCHUNK>), grid, block, 0, stream, a)));
  return finish_launch("norm_gated_fwd");
}

static int gated_bwd_grid(int64_t rows, int ngroups) {
  int g = norm_grid(rows * ngroups);
  int waves = g * NORM_WAVES;
  waves = ((waves + ngroups - 1) / ngroups) * ngroups;   // multiple of ngroups so every wave stays in one group
  return (waves + NORM_WAVES - 1) / NORM_WAVES;
}

extern "C" size_t omk_norm_gated_bwd_workspace_bytes(const OmkNormGatedBwd* p) {
  if (!p) return 0;
  const int64_t cols = p->x.shape[1];
  const int64_t gs = p->group_size > 0 ? p->group_size : cols;
  int ng = (int)(cols / gs);
  int64_t waves = (int64_t)gated_bwd_grid(p->x.shape[0], ng) * NORM_WAVES;
  return (size_t)(waves / ng + 1) * cols * 4;
}

extern "C" int omk_norm_gated_bwd(const OmkNormGatedBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->dy) && present(p->x) && present(p->weight) && present(p->dx) && present(p->dweight), "norm_gated_bwd: dy, x, weight, dx, dweight required");
  const int64_t rows = p->x.shape[0], cols = p->x.shape[1];
  const int64_t gs = p->group_size > 0 ? p->group_size : cols;
  OMK_REQUIRE(cols % gs == 0, "norm_gated_bwd: group_size must divide cols");
  OMK_REQUIRE(p->workspace && p->workspace_bytes >= omk_norm_gated_bwd_workspace_bytes(p), "norm_gated_bwd: workspace too small");
  OMK_REQUIRE(p->dweight.dtype == OMK_F32, "norm_gated_bwd: dweight must be f32");
  OMK_REQUIRE(p->dy.dtype == p->x.dtype && p->dx.dtype == p->x.dtype, "norm_gated_bwd: dtype mismatch");
  if (present(p->z)) OMK_REQUIRE(present(p->dz) && p->z.dtype == p->x.dtype && p->dz.dtype == p->x.dtype, "norm_gated_bwd: z/dz");
  if (rows == 0) return OMK_OK;
  VecPlan plan;
  bool can8 = rows_ok8(p->x) && rows_ok8(p->dy) && rows_ok8(p->z) && rows_ok8(p->dx) && rows_ok8(p->dz) && gs % 8 == 0;
  if (!plan_vec(gs, can8, &plan)) return fail(OMK_EUNSUPPORTED, "norm_gated_bwd: group too large");
  const int ng = (int)(cols / gs);
  const int gridx = gated_bwd_grid(rows, ng);
  const int nparts = gridx * NORM_WAVES / ng;
  NormBwdArgs a = {};
  a.dy = p->dy.data; a.x = p->x.data; a.z = p->z.data; a.w = p->weight.data; a.dx = p->dx.data; a.dz = p->dz.data;
  a.dw_part = (float*)p->workspace;
  a.dys = p->dy.stride[0]; a.xs = p->x.stride[0]; a.zs = present(p->z) ? p->z.stride[0] : 0; a.dxs = p->dx.stride[0];
  a.dzs = present(p->dz) ? p->dz.stride[0] : 0;
  a.rows = rows; a.cols = (int)cols; a.ngroups = ng; a.wdt = p->weight.dtype; a.rms = 1; a.eps = p->eps; a.norm_before_gate = p->norm_before_gate;
  dim3 grid(gridx), block(NORM_THREADS);
  OMK_DISPATCH_DTYPE(p->x.dtype, TX, OMK_PLAN_SWITCH(plan, OMK_LAUNCH((norm_gated_bwd_kernel<TX, VEC, NCHUNK>), grid, block, 0, stream, a)));
  dim3 rg((unsigned)((cols + 255) / 256)), rb(256);
  OMK_LAUNCH(reduce_parts_kernel, rg, rb, 0, stream, (const float*)a.dw_part, nparts, (int)cols, (float*)p->dweight.data);
  return finish_launch("norm_gated_bwd");
}
