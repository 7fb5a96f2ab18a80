// ce.hip -- cross entropy over a block of logits: one 256-thread workgroup per row, two passes over the row (the second
// one is served by L2: a row of the 50 288-wide vocabulary is 100 KB of bf16), loss out, gradient written over the logits.
// HBM-bound: V * s read + V * s written per row.
#include "omk_common.h"

namespace omk {

struct CeArgs {
  void* logits; const int64_t* labels; float* losses; const float* gscale;
  int64_t ls, ignore; int T, V, write_grad;
};

template <class T>
__global__ __launch_bounds__(256) void cross_entropy_kernel(CeArgs a) {
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float red_m[4], red_s[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  T* x = (T*)a.logits + (int64_t)row * a.ls;
  const int64_t y = a.labels[row];
  const bool counted = y != a.ignore && y >= 0 && y < a.V;
  const int nvec = a.V / VEC;                   // V % VEC elements at the end are handled by the scalar tail
  // ---- pass 1: online max / sum of exponentials per thread, then across the workgroup
  float m = -INFINITY, s = 0.f;
  for (int i = tid; i < nvec; i += 256) {
    float v[VEC];
    load_vec<T, VEC>(x + (int64_t)i * VEC, v);
    float vm = v[0];
#pragma unroll
    for (int e = 1; e < VEC; e++) vm = fmaxf(vm, v[e]);
    if (vm > m) { s *= exp2_fast((m - vm) * LOG2E); m = vm; }
#pragma unroll
    for (int e = 0; e < VEC; e++) s += exp2_fast((v[e] - m) * LOG2E);
  }
  for (int i = nvec * VEC + tid; i < a.V; i += 256) {
    const float v = to_f32(x[i]);
    if (v > m) { s *= exp2_fast((m - v) * LOG2E); m = v; }
    s += exp2_fast((v - m) * LOG2E);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float mo = shfl_xor(m, off), so = shfl_xor(s, off);
    const float mn = fmaxf(m, mo);
    s = (m == -INFINITY ? 0.f : s * exp2_fast((m - mn) * LOG2E)) + (mo == -INFINITY ? 0.f : so * exp2_fast((mo - mn) * LOG2E));
    m = mn;
  }
  if (lane == 0) { red_m[wv] = m; red_s[wv] = s; }
  block_sync();
  float M = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
  float S = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) S += red_m[i] == -INFINITY ? 0.f : red_s[i] * exp2_fast((red_m[i] - M) * LOG2E);
  const float lse = M + logf(S);
  if (tid == 0) a.losses[row] = counted ? lse - to_f32(x[y]) : 0.f;
  if (!a.write_grad) return;
  block_sync();   // the label's logit was read before anyone overwrites it
  // ---- pass 2: gradient over the logits
  const float gs = counted ? (a.gscale ? a.gscale[0] : 1.f) : 0.f;
  const float c = -lse * LOG2E;
  for (int i = tid; i < nvec; i += 256) {
    float v[VEC];
    load_vec<T, VEC>(x + (int64_t)i * VEC, v);
#pragma unroll
    for (int e = 0; e < VEC; e++) {
      const float p = exp2_fast(fmaf(v[e], LOG2E, c));
      v[e] = (p - ((int64_t)i * VEC + e == y ? 1.f : 0.f)) * gs;
    }
    store_vec<T, VEC>(x + (int64_t)i * VEC, v);
  }
  for (int i = nvec * VEC + tid; i < a.V; i += 256) {
    const float p = exp2_fast(fmaf(to_f32(x[i]), LOG2E, c));
    x[i] = from_f32<T>((p - (i == y ? 1.f : 0.f)) * gs);
  }
}

}  // namespace omk

using namespace omk;

extern "C" int omk_cross_entropy(const OmkCrossEntropy* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->logits) && p->labels && present(p->losses), "cross_entropy: logits, labels, losses required");
  OMK_REQUIRE(p->logits.ndim == 2 && p->logits.stride[1] == 1, "cross_entropy: logits must be (T, V) with unit stride on V");
  OMK_REQUIRE(p->losses.dtype == OMK_F32 && numel(p->losses) == p->logits.shape[0] && is_contig_last(p->losses), "cross_entropy: losses must be contiguous f32 (T)");
  const size_t es = dtype_size(p->logits.dtype);
  OMK_REQUIRE(((uintptr_t)p->logits.data & 15) == 0 && (p->logits.stride[0] * es) % 16 == 0, "cross_entropy: logits rows must be 16-byte aligned");
  CeArgs a = {};
  a.logits = p->logits.data; a.labels = p->labels; a.losses = (float*)p->losses.data; a.gscale = p->grad_scale;
  a.ls = p->logits.stride[0]; a.ignore = p->ignore_index; a.T = (int)p->logits.shape[0]; a.V = (int)p->logits.shape[1]; a.write_grad = p->write_grad;
  if (a.T == 0 || a.V == 0) return OMK_OK;
  dim3 grid((unsigned)a.T), block(256);
  OMK_DISPATCH_DTYPE(p->logits.dtype, T, OMK_LAUNCH((cross_entropy_kernel<T>), grid, block, 0, stream, a));
  return finish_launch("cross_entropy");
}
