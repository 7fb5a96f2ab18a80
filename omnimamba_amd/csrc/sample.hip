// sample.hip -- on-device token sampling for the decode loop (reference
// /root/reference/models/stage2/generation.py:87-121 `sample`, the top_k > 0 branch and the top_k == 1 short cut).
//
// One 1024-thread workgroup per row of logits.  No host scalar is read after the launch, no allocation, no sync: the kernel
// sits inside the captured 1-token step (generation.SampleLoopGraph) and the per-step random stream comes from a device
// counter the graph itself advances.
//
//   top_k == 0      the whole vocabulary, multinomial of softmax(logits / T) (the reference's default arguments of t2i_generate:
//                   top_k = 0, top_p = 1.0).  Round 6: with the reference's two filters of this branch (generation.py:108-119) --
//                   min_p > 0: a token stays iff its raw logit >= min_p x the largest softmax(logits) probability (the reference compares
//                   LOGITS with that probability, :43; kept as it is) -- and 0 < top_p < 1: the ascending cumulative-probability cut of
//                   :57-69 over all V tokens WITHOUT a sort: token masses as 2^-40 fixed-point integers (sums do not depend on the order
//                   they are formed in), a 4-pass 8-bit radix walk over the order-preserving logit keys with MASS histograms finds the
//                   boundary value, ties at it are cut in index order, the draw is an inverse CDF over the kept tokens in index order.
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
  float top_p, inv_temp, min_p;
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
__device__ __forceinline__ void lds_add_u64(unsigned long long* p, unsigned long long v) { *p += v; }
#else
__device__ __forceinline__ uint32_t lds_fetch_add_u32(uint32_t* p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_add_u64(unsigned long long* p, unsigned long long v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
#endif
constexpr int SAMPLE_KMAX = 64;
// threads / waves per row of logits (round 6: 256 -> 1024, typed loads, four requests in flight per lane in every pass: the kernel is a chain
// of passes over one row from L2, bound by latency -- profiles/r06_sampler.txt).  The CPU emulator runs the same code with 256 (it executes the
// lanes one after the other: sixteen waves per row would make the CPU suite four times slower for nothing).
#ifdef OMK_EMU
constexpr int SNT = 256;
#else
constexpr int SNT = 1024;
#endif
constexpr int SNW = SNT / 64;

// inclusive scan over the SNT threads of the workgroup through LDS (buf: 2 x SNT entries); every thread returns its own prefix
template <class T>
__device__ __forceinline__ T block_incl_scan(T v, T* buf, int tid) {
  int cur = 0;
  buf[tid] = v;
  block_sync();
#pragma unroll 1
  for (int off = 1; off < SNT; off <<= 1) {
    T x = buf[cur * SNT + tid];
    if (tid >= off) x += buf[cur * SNT + tid - off];
    buf[(cur ^ 1) * SNT + tid] = x;
    cur ^= 1;
    block_sync();
  }
  const T r = buf[cur * SNT + tid];
  block_sync();
  return r;
}

// inclusive scan over the lanes of a wave (any additive type the shuffles move)
template <class T> __device__ __forceinline__ T wave_incl_scan_t(T v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const T o = shfl(v, lane >= off ? lane - off : lane); if (lane >= off) v += o; }
  return v;
}
// The digit walk of a radix pass by ONE WAVE instead of one thread (a thread reading 60 - 190 counters one after the other was 2 - 8 us per pass):
// cnt(r) for r = 0 .. 255 in WALK order, lane l holds r = 4 l .. 4 l + 3; returns in every lane the first r whose running total reaches `need`
// by the rule `total before + cnt(r) > limit` (limit = need - 1 for "covers need keys"), and the total BEFORE it; r = 255 when none does.
template <class T, class F> __device__ __forceinline__ void wave_find_digit(F cnt, T limit, int lane, uint32_t& r_out, T& before_out) {
  T c[4], s = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) { c[j] = cnt(4 * lane + j); s += c[j]; }
  const T incl = wave_incl_scan_t<T>(s, lane);
  T run = incl - s;
  int jj = -1; T bef = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) { if (jj < 0 && run + c[j] > limit) { jj = j; bef = run; } run += c[j]; }
#ifdef OMK_EMU
  const unsigned long long m = emu::ballot(jj >= 0 ? 1 : 0);
#else
  const unsigned long long m = __builtin_amdgcn_ballot_w64(jj >= 0);
#endif
  if (m) {
    const int l0 = __builtin_ctzll(m);
    r_out = (uint32_t)(4 * l0 + shfl(jj, l0));
    before_out = shfl(bef, l0);
  } else {   // (the walk ends on the last digit: everything in front of it is "before")
    r_out = 255u;
    before_out = shfl(incl, 63) - shfl(c[3], 63);
  }
}

template <class T>
__global__ __launch_bounds__(SNT) void sample_kernel(SampleArgs a) {
  // eight copies of every counter, picked by the lane: the keys of a row share their top byte(s), and 64 lanes adding to ONE LDS address are 64
  // serial atomics (copy c of digit d lives at 8 d + c: the copies of a digit sit in different banks)
  __shared__ uint32_t hist[256 * 8];
  __shared__ uint32_t sel_prefix, sel_need, n_gt, n_eq;
  __shared__ float cval[SAMPLE_KMAX];
  __shared__ int cidx[SAMPLE_KMAX];
  __shared__ float wmax[SNW];
  __shared__ int wimax[SNW];
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int V = a.V;
  const T* rowp = (const T*)a.logits + (int64_t)row * a.ls;
  auto ld = [&](int i) -> float { return to_f32(rowp[i]); };
  // one pass over the row, thread-strided (coalesced), four requests in flight per lane
  auto strided = [&](auto&& fn) {
    for (int i0 = tid; i0 < V; i0 += 4 * SNT) {
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = i0 + u * SNT; v[u] = ld(i < V ? i : V - 1); }
#pragma unroll
      for (int u = 0; u < 4; u++) { const int i = i0 + u * SNT; if (i < V) fn(v[u], i); }
    }
  };
  // block maximum + its lowest index, in every thread
  auto block_argmax = [&](float& m, int& mi) {
    m = -INFINITY; mi = 0x7fffffff;
    strided([&](float v, int i) { if (v > m || (v == m && i < mi)) { m = v; mi = i; } });
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const float om = shfl_xor(m, off); const int oi = shfl_xor(mi, off);
      if (om > m || (om == m && oi < mi)) { m = om; mi = oi; }
    }
    if (lane == 0) { wmax[wv] = m; wimax[wv] = mi; }
    block_sync();
    for (int k = 0; k < SNW; k++) if (wmax[k] > m || (wmax[k] == m && wimax[k] < mi)) { m = wmax[k]; mi = wimax[k]; }
  };
  auto philox_u24 = [&]() -> uint32_t {
    uint32_t cc[4] = {(uint32_t)row, 0u, 0u, 0u};
    const unsigned long long step = (a.counter ? (unsigned long long)a.counter[0] : 0ull) + a.offset;
    cc[1] = (uint32_t)step; cc[2] = (uint32_t)(step >> 32);
    philox4x32(cc, (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
    return cc[0] >> 8;
  };

  if (a.top_k == 1) {
    float m; int mi;
    block_argmax(m, mi);
    if (tid == 0) a.out[row] = mi == 0x7fffffff ? 0 : mi;
    return;
  }
  if (a.top_k == 0 && (a.min_p > 0.f || (a.top_p > 0.f && a.top_p < 1.f))) {
    // ---- the whole vocabulary behind one of the reference's two filters (see the header)
    __shared__ unsigned long long hm[256], hm8[256 * 8], sc64[2 * SNT];   // (hm8 / hc8: eight copies per digit, picked by the lane -- see hist)
    __shared__ uint32_t hc[256], hc8[256 * 8], sc32[2 * SNT];
    __shared__ unsigned long long s_acc, s_ztot;
    __shared__ uint32_t s_prefix, s_r;
    __shared__ float wsumf[SNW];
    __shared__ int pick_f;
    const int c = (V + SNT - 1) / SNT, lo = tid * c < V ? tid * c : V, hi = lo + c < V ? lo + c : V;
    float m; int mi;
    if (tid == 0) pick_f = -1;
    block_argmax(m, mi);
    const float sct = a.inv_temp * LOG2E;
    // mass of a token in the tempered distribution, 2^-40 units of the largest one (an integer: sums are exact and order-free)
    auto mass = [&](float l) -> unsigned long long { return (unsigned long long)(exp2_fast((l - m) * sct) * 1099511627776.0f); };
    const bool use_min_p = a.min_p > 0.f;
    float thr = -INFINITY;           // min_p: raw logits below it are cut
    uint32_t vstar = 0u, rcut = 0u;  // top-p: keys below vstar are cut, and the first rcut tokens (index order) whose key equals it
    if (use_min_p) {
      float z = 0.f;
      strided([&](float v, int) { z += exp2_fast((v - m) * LOG2E); });          // softmax of the UNtempered logits (:109)
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) z += shfl_xor(z, off);
      if (lane == 0) wsumf[wv] = z;
      block_sync();
      z = 0.f;
      for (int k = 0; k < SNW; k++) z += wsumf[k];
      thr = a.min_p / z;                                                        // max probability = 1 / z
    } else {
      unsigned long long zt = 0ull;
      strided([&](float v, int) { zt += mass(v); });
      const unsigned long long incl = block_incl_scan<unsigned long long>(zt, sc64, tid);
      if (tid == SNT - 1) s_ztot = incl;
      if (tid == 0) { s_prefix = 0u; s_acc = 0ull; }
      block_sync();
      const unsigned long long ztot = s_ztot;
      unsigned long long Q = (unsigned long long)((double)(1.f - a.top_p) * (double)ztot);   // masses <= Q (cumulative, ascending) are cut
      if (Q >= ztot) Q = ztot ? ztot - 1 : 0ull;
      for (int pass = 0; pass < 4; pass++) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 256 * 8; i += SNT) { hm8[i] = 0ull; hc8[i] = 0u; }
        block_sync();
        const uint32_t pre = s_prefix, pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        strided([&](float l, int) {
          const uint32_t k = fkey(l);
          if ((k & pmask) == pre) { const uint32_t d = 8u * ((k >> shift) & 255u) + (uint32_t)(lane & 7); lds_add_u64(&hm8[d], mass(l)); lds_fetch_add_u32(&hc8[d], 1u); }
        });
        block_sync();
        if (tid < 256) {
          unsigned long long tm = 0ull; uint32_t tc = 0u;
#pragma unroll
          for (int c8 = 0; c8 < 8; c8++) { tm += hm8[8 * tid + c8]; tc += hc8[8 * tid + c8]; }
          hm[tid] = tm; hc[tid] = tc;
        }
        block_sync();
        if (wv == 0) {   // ascending: digits whose whole mass still fits under Q are cut; the first one that does not holds the boundary
          const unsigned long long acc0 = s_acc;
          uint32_t d; unsigned long long before;
          wave_find_digit<unsigned long long>([&](int rr) -> unsigned long long { return hm[rr]; }, Q - acc0, lane, d, before);
          if (lane == 0) {
            const unsigned long long acc = acc0 + before;
            s_acc = acc; s_prefix = pre | (d << shift);
            if (pass == 3) { const unsigned long long one = hc[d] ? hm[d] / hc[d] : 0ull; s_r = one ? (uint32_t)((Q - acc) / one) : 0u; if (s_r >= hc[d] && hc[d]) s_r = hc[d] - 1u; }
          }
        }
        block_sync();
      }
      vstar = s_prefix; rcut = s_r;
    }
    // ---- kept mass of the thread's tokens.  Ties at the boundary value that are cut (rare) need index-order ranks: contiguous slices then;
    // otherwise thread t owns t, t + SNT, ... (coalesced reads; the inverse CDF may lay the tokens out in any fixed order)
    const bool contig = !use_min_p && rcut > 0u;
    uint32_t tie_before = 0u;
    if (contig) {
      uint32_t tc = 0u;
      for (int i = lo; i < hi; i++) tc += fkey(ld(i)) == vstar ? 1u : 0u;
      tie_before = block_incl_scan<uint32_t>(tc, sc32, tid) - tc;
    }
    auto owned = [&](auto&& fn) {
      if (contig) { for (int i = lo; i < hi; i++) fn(ld(i), i); }
      else strided(fn);
    };
    unsigned long long km = 0ull;
    {
      uint32_t rank = tie_before;
      owned([&](float l, int) {
        bool keep;
        if (use_min_p) keep = l >= thr;
        else { const uint32_t k = fkey(l); keep = k > vstar || (k == vstar && rank++ >= rcut); }
        if (keep) km += mass(l);
      });
    }
    const unsigned long long incl = block_incl_scan<unsigned long long>(km, sc64, tid);
    if (tid == SNT - 1) s_ztot = incl;
    block_sync();
    const unsigned long long ktot = s_ztot, start = incl - km;
    const unsigned long long u24 = philox_u24();
    const unsigned long long target = (ktot >> 24) * u24 + (((ktot & 0xffffffull) * u24) >> 24);   // floor(u ktot), u in [0, 1): < ktot
    if (ktot > 0ull && target >= start && target < incl) {   // exactly one thread
      unsigned long long run = start; uint32_t rank = tie_before; int got = -1; bool done = false;
      owned([&](float l, int i) {
        bool keep;
        if (use_min_p) keep = l >= thr;
        else { const uint32_t k = fkey(l); keep = k > vstar || (k == vstar && rank++ >= rcut); }
        const unsigned long long e = keep ? mass(l) : 0ull;
        if (!done && e > 0ull) got = i;
        run += e;
        if (run > target) done = true;
      });
      pick_f = got;
    }
    block_sync();
    // (nothing kept -- min_p with every logit under the threshold, where the reference's multinomial raises on a row of NaN: the arg max)
    if (tid == 0) a.out[row] = pick_f >= 0 ? pick_f : (mi == 0x7fffffff ? 0 : mi);
    return;
  }
  if (a.top_k == 0) {
    // full-vocabulary multinomial (the reference's top_k == 0 branch with top_p outside (0, 1): softmax(logits / T), one draw).
    // Block maximum, the mass of every thread's tokens, an inclusive scan of the SNT masses (the scan values tile [0, total) exactly:
    // end_t = start_{t + 1}), one Philox number scaled to the total, and the thread whose interval holds it walks its tokens.
    __shared__ float scf[2 * SNT];
    __shared__ float tot_s;
    __shared__ int pick_s;
    // (thread t owns the tokens t, t + SNT, ...: the inverse CDF may lay the tokens out in ANY fixed order, and this one reads coalesced --
    // contiguous slices per thread were 64 scattered lines per request: 17 -> see profiles/r06_sampler.txt)
    float m; int mi;
    if (tid == 0) pick_s = 0;
    block_argmax(m, mi);
    const float sc = a.inv_temp * LOG2E;
    float mine = 0.f;
    strided([&](float v, int) { mine += exp2_fast((v - m) * sc); });
    const float end = block_incl_scan<float>(mine, scf, tid);
    scf[tid] = end;                                                 // (the scan values tile [0, total): a thread starts where its neighbour ends)
    if (tid == SNT - 1) tot_s = end;
    block_sync();
    const float tot = tot_s, start = tid ? scf[tid - 1] : 0.f;
    float u = (float)philox_u24() * (1.0f / 16777216.0f) * tot;   // uniform in [0, tot)
    if (u >= tot) u = tot * 0.99999994f;                            // (the product may round up to the total)
    if (u >= start && u < end) {   // exactly one thread when 0 < tot < inf
      float run = start; int got = tid < V ? tid : 0; bool done = false;
      strided([&](float v, int i) {
        const float e = exp2_fast((v - m) * sc);
        if (!done && e > 0.f) got = i;                              // rounding inside the walk: the last token with mass
        run += e;
        if (run > u) done = true;
      });
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
    for (int i = tid; i < 256 * 8; i += SNT) hist[i] = 0u;
    block_sync();
    const uint32_t pre = sel_prefix, pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    strided([&](float v, int) {   // (aggregating the lanes of a wave per distinct digit with ballots instead: slower, 55.7 -> 66.3 us)
      const uint32_t k = fkey(v);
      if ((k & pmask) == pre) lds_fetch_add_u32(&hist[8u * ((k >> shift) & 255u) + (uint32_t)(lane & 7)], 1u);
    });
    block_sync();
    if (tid < 256) {
      uint32_t t = 0u;
#pragma unroll
      for (int c = 0; c < 8; c++) t += hist[8 * tid + c];
      hist[8 * tid] = t;
    }
    block_sync();
    if (wv == 0) {   // walk the digits from the top until `need` keys are covered
      const uint32_t need = sel_need;
      uint32_t r, before;
      wave_find_digit<uint32_t>([&](int rr) -> uint32_t { return hist[8u * (255u - (uint32_t)rr)]; }, need - 1u, lane, r, before);
      if (lane == 0) { sel_prefix = pre | ((255u - r) << shift); sel_need = need - before; }
    }
    block_sync();
  }
  const uint32_t kth = sel_prefix;           // keys > kth: all taken; keys == kth: sel_need of them (lowest indices first)
  const uint32_t take_eq = sel_need;
  if (tid == 0) { n_gt = 0u; n_eq = 0u; }
  block_sync();
  const uint32_t base_eq = (uint32_t)K - take_eq;   // slots [0, base_eq) for keys > kth, [base_eq, K) for the ties at the threshold
  strided([&](float v, int i) {
    const uint32_t k = fkey(v);
    if (k > kth) { const uint32_t s = lds_fetch_add_u32(&n_gt, 1u); if (s < base_eq) { cval[s] = v; cidx[s] = i; } }
    else if (k == kth) { const uint32_t s = lds_fetch_add_u32(&n_eq, 1u); if (s < take_eq) { cval[base_eq + s] = v; cidx[base_eq + s] = i; } }
  });
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
  const float u = (float)philox_u24() * (1.0f / 16777216.0f) * ktot;   // uniform in [0, ktot)
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
  OMK_REQUIRE(p->min_p >= 0.f && p->min_p < 1.f, "sample: min_p must be in [0, 1)");
  OMK_REQUIRE(p->min_p == 0.f || p->top_k == 0, "sample: min_p belongs to the whole-vocabulary branch (top_k == 0), as in the reference");
  OMK_REQUIRE(p->temperature > 0.f, "sample: temperature must be positive");
  OMK_REQUIRE(p->top_p <= 1.f, "sample: top-p should be in (0, 1]");
  if (p->logits.shape[0] == 0) return OMK_OK;
  OMK_REQUIRE(p->logits.shape[1] > 0 && p->logits.shape[1] < (1ll << 31), "sample: vocabulary size");
  SampleArgs a = {};
  a.logits = p->logits.data; a.ls = p->logits.stride[0]; a.dt = p->logits.dtype;
  a.out = (int64_t*)p->out_ids.data; a.B = (int)p->logits.shape[0]; a.V = (int)p->logits.shape[1]; a.top_k = p->top_k;
  a.top_p = p->top_p; a.inv_temp = 1.f / p->temperature; a.min_p = p->min_p;
  a.seed = p->seed; a.counter = (const int64_t*)p->step_counter; a.offset = p->offset;
  dim3 grid((unsigned)a.B), block(SNT);
  OMK_DISPATCH_DTYPE(a.dt, T, OMK_LAUNCH((sample_kernel<T>), grid, block, 0, stream, a));
  return finish_launch("sample");
}
