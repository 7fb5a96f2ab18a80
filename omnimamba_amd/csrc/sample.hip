// sample.hip -- on-device token sampling for the decode loop (reference
// /root/reference/models/stage2/generation.py:87-121 `sample`, the top_k > 0 branch and the top_k == 1 short cut).
//
// One 256-thread workgroup per row of logits.  No host scalar is read after the launch, no allocation, no sync: the kernel
// sits inside the captured 1-token step (generation.SampleLoopGraph) and the per-step random stream comes from a device
// counter the graph itself advances.
//
//   top_k == 0      the whole vocabulary, plain multinomial of softmax(logits / T) (the reference's default arguments of t2i_generate:
//                   top_k = 0, top_p = 1.0); a top-p cut over the whole vocabulary needs a full sort and stays with the host library
//   top_k == 1      argmax (lowest index among equal maxima)
//   1 < top_k <= 64 the k largest logits by a 4-pass 8-bit radix select over order-preserving integer keys (histograms in LDS,
//                   the row re-read from L2: 200 KB of fp32 at vocab 50 288), candidates gathered into LDS and sorted by one
//                   wave (bitonic, ties by index); values / temperature; softmax over the k; top-p exactly as the reference
//                   filters -- a candidate stays iff the probability mass of the candidates LARGER than it is < top_p
//                   (generation.py:64-76: ascending cumulative sum <= 1 - top_p is cut; top_p <= 0 or >= 1 cuts nothing);
//                   inverse-CDF draw with one Philox4x32-10 uniform keyed by (seed, row, step counter).
// The reference draws with torch.multinomial; the streams differ, the distribution is the same (tests: chi-square at fixed
// seeds, bit-exact ids for top_k == 1).
#include "omk_common.h"

namespace omk {

struct SampleArgs {
  const void* logits; int64_t ls; int dt;
  int64_t* out; int B, V, top_k;
  float top_p, inv_temp;
  unsigned long long seed; const int64_t* counter; unsigned long long offset;
};

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
// Philox4x32-10 (Salmon et al., SC'11): counter (c0..c3), key (k0, k1)
__device__ __forceinline__ void philox4x32(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const uint32_t h0 = mulhi32(0xD2511F53u, c[0]), l0 = 0xD2511F53u * c[0];
    const uint32_t h1 = mulhi32(0xCD9E8D57u, c[2]), l1 = 0xCD9E8D57u * c[2];
    const uint32_t n0 = h1 ^ c[1] ^ k0, n2 = h0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = l1; c[2] = n2; c[3] = l0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}
// order-preserving key: larger float <-> larger unsigned (NaN sorts above +inf; -0 below +0)
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t u = __builtin_bit_cast(uint32_t, f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

#ifdef OMK_EMU
__device__ __forceinline__ uint32_t lds_fetch_add_u32(uint32_t* p, uint32_t v) { const uint32_t o = *p; *p = o + v; return o; }
#else
__device__ __forceinline__ uint32_t lds_fetch_add_u32(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif

constexpr int SAMPLE_KMAX = 64;

__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
  __shared__ uint32_t hist[256];
  __shared__ uint32_t sel_prefix, sel_need, n_gt, n_eq;
  __shared__ float cval[SAMPLE_KMAX];
  __shared__ int cidx[SAMPLE_KMAX];
  __shared__ float wmax[4];
  __shared__ int wimax[4];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int V = a.V;
  auto ld = [&](int i) -> float { return load_rt(a.logits, (int64_t)row * a.ls + i, a.dt); };

  if (a.top_k == 1) {
    float m = -INFINITY; int mi = 0x7fffffff;
    for (int i = tid; i < V; i += 256) { const float v = ld(i); if (v > m || (v == m && i < mi)) { m = v; mi = i; } }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float om = shfl_xor(m, off); const int oi = shfl_xor(mi, off);
      if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) { wmax[wv] = m; wimax[wv] = mi; }
    block_sync();
    if (tid == 0) {
      for (int k = 1; k < 4; k++) if (wmax[k] > m || (wmax[k] == m && wimax[k] < mi)) { m = wmax[k]; mi = wimax[k]; }
      a.out[row] = mi == 0x7fffffff ? 0 : mi;
    }
    return;
  }
  if (a.top_k == 0) {
    // full-vocabulary multinomial (the reference's top_k == 0 branch with top_p outside (0, 1): softmax(logits / T), one draw).
    // Thread t owns the contiguous slice [t c, t c + c): block maximum, slice masses, an inclusive scan of the 256 masses (the scan
    // values tile [0, total) exactly: end_t = start_{t + 1}), one Philox number scaled to the total, and the thread whose interval
    // holds it walks its slice in index order -- the inverse CDF in index order.
    __shared__ float endm[256];
    __shared__ float wsum[4];
    __shared__ int pick_s;
    const int c = (V + 255) / 256, lo = tid * c, hi = lo + c < V ? lo + c : V;
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, ld(i));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, shfl_xor(m, off));
    if (lane == 0) wmax[wv] = m;
    if (tid == 0) pick_s = 0;
    block_sync();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float sc = a.inv_temp * LOG2E;
    float mine = 0.f;
    for (int i = lo; i < hi; i++) mine += exp2_fast((ld(i) - m) * sc);
    const float incl = wave_incl_scan_add(mine);
    if (lane == 63) wsum[wv] = incl;
    block_sync();
    float base = 0.f;
    for (int k = 0; k < wv; k++) base += wsum[k];
    endm[tid] = base + incl;
    block_sync();
    const float tot = endm[255], start = tid ? endm[tid - 1] : 0.f, end = endm[tid];
    uint32_t cc[4] = {(uint32_t)row, 0u, 0u, 0u};
    const unsigned long long step = (a.counter ? (unsigned long long)a.counter[0] : 0ull) + a.offset;
    cc[1] = (uint32_t)step; cc[2] = (uint32_t)(step >> 32);
    philox4x32(cc, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    float u = (float)(cc[0] >> 8) * (1.0f / 16777216.0f) * tot;   // uniform in [0, tot)
    if (u >= tot) u = tot * 0.99999994f;                            // (the product may round up to the total)
    if (u >= start && u < end) {   // exactly one thread when 0 < tot < inf
      float run = start; int got = lo;
      for (int i = lo; i < hi; i++) {
        const float e = exp2_fast((ld(i) - m) * sc);
        if (e > 0.f) got = i;                                       // rounding inside the slice: the last token with mass
        run += e;
        if (run > u) break;
      }
      pick_s = got;
    }
    block_sync();
    if (tid == 0) a.out[row] = pick_s;
    return;
  }
  const int K = a.top_k < V ? a.top_k : V;
  // ---- radix select: the key of the K-th largest logit, 8 bits per pass from the top
  if (tid == 0) { sel_prefix = 0u; sel_need = (uint32_t)K; }
  for (int pass = 0; pass < 4; pass++) {
    const int shift = 24 - 8 * pass;
    hist[tid] = 0u;
    block_sync();
    const uint32_t pre = sel_prefix, pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int i = tid; i < V; i += 256) {
      const uint32_t k = fkey(ld(i));
      if ((k & pmask) == pre) lds_fetch_add_u32(&hist[(k >> shift) & 255u], 1u);
    }
    block_sync();
    if (tid == 0) {   // walk the digits from the top until `need` keys are covered
      uint32_t need = sel_need, d = 255u;
      for (;; d--) {
        const uint32_t c = hist[d];
        if (c >= need || d == 0u) break;
        need -= c;
      }
      sel_prefix = pre | (d << shift);
      sel_need = need;
    }
    block_sync();
  }
  const uint32_t kth = sel_prefix;           // keys > kth: all taken; keys == kth: sel_need of them (lowest indices first)
  const uint32_t take_eq = sel_need;
  if (tid == 0) { n_gt = 0u; n_eq = 0u; }
  block_sync();
  const uint32_t base_eq = (uint32_t)K - take_eq;   // slots [0, base_eq) for keys > kth, [base_eq, K) for the ties at the threshold
  for (int i = tid; i < V; i += 256) {
    const float v = ld(i);
    const uint32_t k = fkey(v);
    if (k > kth) { const uint32_t s = lds_fetch_add_u32(&n_gt, 1u); if (s < base_eq) { cval[s] = v; cidx[s] = i; } }
    else if (k == kth) { const uint32_t s = lds_fetch_add_u32(&n_eq, 1u); if (s < take_eq) { cval[base_eq + s] = v; cidx[base_eq + s] = i; } }
  }
  block_sync();
  if (wv != 0) return;
  if (n_eq > take_eq) {
    // more logits equal the threshold than there are slots left (16-bit logits tie often): which of them are candidates must not
    // depend on the order the tickets were handed out -- one wave walks the row in index order and keeps the first take_eq
    uint32_t got = 0u;
    for (int i0 = 0; i0 < V && got < take_eq; i0 += 64) {
      const int i = i0 + lane;
      float v = 0.f;
      bool eq = false;
      if (i < V) { v = ld(i); eq = fkey(v) == kth; }
#ifdef OMK_EMU
      const unsigned long long m = emu::ballot(eq ? 1 : 0);
#else
      const unsigned long long m = __builtin_amdgcn_ballot_w64(eq);
#endif
      const uint32_t rank = got + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
      if (eq && rank < take_eq) { cval[base_eq + rank] = v; cidx[base_eq + rank] = i; }
      got += (uint32_t)__builtin_popcountll(m);
    }
  }
  // ---- one wave: sort the K candidates (value descending, index ascending), softmax, top-p, draw
  float v = lane < K ? cval[lane] : -INFINITY;
  int ix = lane < K ? cidx[lane] : 0x7fffffff;
#pragma unroll
  for (int size = 2; size <= 64; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride >= 1; stride >>= 1) {
      const float ov = shfl_xor(v, stride); const int oi = shfl_xor(ix, stride);
      const bool first = (lane & stride) == 0;                  // lower lane of the pair
      const bool desc = (lane & size) == 0;                     // direction of this bitonic block
      const bool mine_before = v > ov || (v == ov && ix < oi);  // "mine sorts before the other" in descending order
      const bool keep = (first == desc) ? mine_before : !mine_before;
      if (!keep) { v = ov; ix = oi; }
    }
  const float vmax = wave_read_lane(v, 0);
  float p = lane < K ? exp2_fast((v - vmax) * a.inv_temp * LOG2E) : 0.f;
  const float tot = wave_sum(p);
  p /= tot;
  float incl = wave_incl_scan_add(p);
  const float excl = incl - p;
  const bool keep = lane < K && (a.top_p <= 0.f || a.top_p >= 1.f || excl < a.top_p);
  const float pk = keep ? p : 0.f;
  const float ktot = wave_sum(pk);
  const float cum = wave_incl_scan_add(pk);
  uint32_t c[4] = {(uint32_t)row, 0u, 0u, 0u};
  const unsigned long long step = (a.counter ? (unsigned long long)a.counter[0] : 0ull) + a.offset;
  c[1] = (uint32_t)step; c[2] = (uint32_t)(step >> 32);
  philox4x32(c, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
  const float u = (float)(c[0] >> 8) * (1.0f / 16777216.0f) * ktot;   // uniform in [0, ktot)
  // first kept candidate whose cumulative mass exceeds u
  const bool hit = keep && cum > u;
  unsigned long long mask;
#ifdef OMK_EMU
  mask = emu::ballot(hit ? 1 : 0);
#else
  mask = __builtin_amdgcn_ballot_w64(hit);
#endif
  int pick = mask ? __builtin_ctzll(mask) : 0;   // rounding at the top end: fall back to the largest candidate
  const int chosen = shfl(ix, pick);
  if (lane == 0) a.out[row] = chosen;
}

}  // namespace omk

using namespace omk;

extern "C" int omk_sample(const OmkSample* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->logits) && present(p->out_ids), "sample: logits and out_ids required");
  OMK_REQUIRE(p->logits.ndim == 2 && p->logits.stride[1] == 1, "sample: logits must be (batch, vocab) with unit last stride");
  OMK_REQUIRE(p->out_ids.ndim == 1 && p->out_ids.shape[0] == p->logits.shape[0] && p->out_ids.stride[0] == 1, "sample: out_ids must be dense int64 (batch)");
  OMK_REQUIRE(p->logits.dtype == OMK_F32 || p->logits.dtype == OMK_BF16 || p->logits.dtype == OMK_F16, "sample: logits dtype");
  OMK_REQUIRE(p->top_k >= 0 && p->top_k <= SAMPLE_KMAX, "sample: top_k must be in [0, %d]", SAMPLE_KMAX);
  OMK_REQUIRE(p->top_k > 0 || p->top_p <= 0.f || p->top_p >= 1.f, "sample: top_k == 0 (full vocabulary) is the plain multinomial only; a top-p cut over the whole vocabulary stays on the host library");
  OMK_REQUIRE(p->temperature > 0.f, "sample: temperature must be positive");
  OMK_REQUIRE(p->top_p <= 1.f, "sample: top-p should be in (0, 1]");
  if (p->logits.shape[0] == 0) return OMK_OK;
  OMK_REQUIRE(p->logits.shape[1] > 0 && p->logits.shape[1] < (1ll << 31), "sample: vocabulary size");
  SampleArgs a = {};
  a.logits = p->logits.data; a.ls = p->logits.stride[0]; a.dt = p->logits.dtype;
  a.out = (int64_t*)p->out_ids.data; a.B = (int)p->logits.shape[0]; a.V = (int)p->logits.shape[1]; a.top_k = p->top_k;
  a.top_p = p->top_p; a.inv_temp = 1.f / p->temperature;
  a.seed = p->seed; a.counter = (const int64_t*)p->step_counter; a.offset = p->offset;
  dim3 grid((unsigned)a.B), block(256);
  OMK_LAUNCH(sample_kernel, grid, block, 0, stream, a);
  return finish_launch("sample");
}
