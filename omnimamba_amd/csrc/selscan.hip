// selscan.hip -- Mamba-1 selective scan, forward and backward (selective_scan_fn signature, BASELINE.json configs[0]).
//
//   x_t = exp(delta_t A[d,:]) x_{t-1} + delta_t B_t u_t ;  y_t = <C_t, x_t> ;  out = (y + D u) * silu(z)
//
// HBM-bound: B*L*(4*D*s + 2*G*N*s) algorithmic bytes.  One thread owns one channel d and
// keeps its N-vector state in registers; a workgroup covers DT adjacent channels of one group and walks L in
// tiles of TL tokens.  Tiles of u / delta / z (and the output) move between HBM and LDS with the lane index
// following whichever of (d, l) has unit stride, so both (B, L, D) "channel-last" storage (adjacent lanes =
// adjacent channels) and upstream's (B, D, L) storage are read in full coalesced rows; B_t / C_t rows of the tile
// are staged once per workgroup and broadcast from LDS to every channel thread.
#include <cstdlib>

#include "omk_common.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int SS_TL = 32;

struct SsArgs {
  const void* u; const void* delta; const void* A; const void* Bm; const void* Cm; const void* D; const void* z; const void* dbias;
  void* out; float* last;
  float* ckpt; int TLB, nTB;   // optional: state at the start of every TLB-token tile, (B, D, nTB, N) f32 (backward)
  int ckpt_cl;                 // the checkpoints as (B, nTB, N, D) instead: adjacent lanes = adjacent channels (the lanes = channels backward)
  int64_t usb, usd, usl, dsb, dsd, dsl, zsb, zsd, zsl, osb, osd, osl;
  int64_t Asd, Asn, Bsb, Bsg, Bsn, Bsl, Csb, Csg, Csn, Csl;   // for constant B/C: Bsg = stride over d, Bsn over n
  int B, Dm, L, N, G, DT, softplus, Bvar, Cvar, adt, bdt, cdt, ddt, dbdt;
  uint32_t xu, xd, xz, xo, xB, xC;   // lanes = channels form: bytes one batch element (B / C: one (batch, group)) spans -- buffer ranges
};

template <class T, int TL = SS_TL>
__device__ __forceinline__ void ss_load_tile(const T* g, int64_t sd, int64_t sl, int d0, int nd, int l0, int nl, T* s, int DT) {
  // s[t][c] <- g[(d0+c)*sd + (l0+t)*sl]
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (sl == 1 && sd != 1) {
    for (int i = tid; i < DT * TL; i += nthr) {
      int c = i / TL, t = i % TL;
      s[t * DT + c] = (c < nd && t < nl) ? g[(int64_t)(d0 + c) * sd + (int64_t)(l0 + t)] : T{};
    }
  } else {
    for (int i = tid; i < DT * TL; i += nthr) {
      int t = i / DT, c = i % DT;
      s[t * DT + c] = (c < nd && t < nl) ? g[(int64_t)(d0 + c) * sd + (int64_t)(l0 + t) * sl] : T{};
    }
  }
}
template <class T, int TL = SS_TL>
__device__ __forceinline__ void ss_store_tile(T* g, int64_t sd, int64_t sl, int d0, int nd, int l0, int nl, const T* s, int DT) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (sl == 1 && sd != 1) {
    for (int i = tid; i < DT * TL; i += nthr) {
      int c = i / TL, t = i % TL;
      if (c < nd && t < nl) g[(int64_t)(d0 + c) * sd + (int64_t)(l0 + t)] = s[t * DT + c];
    }
  } else {
    for (int i = tid; i < DT * TL; i += nthr) {
      int t = i / DT, c = i % DT;
      if (c < nd && t < nl) g[(int64_t)(d0 + c) * sd + (int64_t)(l0 + t) * sl] = s[t * DT + c];
    }
  }
}

template <class T, int NREG>
__global__ void selscan_fwd_kernel(SsArgs a) {
  OMK_DYN_SMEM(smem);
  const int DT = a.DT;
  T* su = (T*)smem;
  T* sd = su + SS_TL * DT;
  T* sz = sd + SS_TL * DT;
  T* so = sz + SS_TL * DT;
  float* sB = (float*)(so + SS_TL * DT);   // [TL][N]
  float* sC = sB + SS_TL * a.N;
  const int dpg = a.Dm / a.G;                       // channels per group
  const int tiles_per_group = (dpg + DT - 1) / DT;
  const int tg = blockIdx.x % tiles_per_group, g = (blockIdx.x / tiles_per_group) % a.G, b = blockIdx.x / (tiles_per_group * a.G);
  const int d0 = g * dpg + tg * DT;
  const int nd = (dpg - tg * DT) < DT ? (dpg - tg * DT) : DT;
  const int c = threadIdx.x;
  const bool live = c < nd;
  const int d = d0 + (live ? c : 0);
  float A[NREG], x[NREG], Bc[NREG], Cc[NREG];
#pragma unroll
  for (int n = 0; n < NREG; n++) {
    A[n] = n < a.N ? load_rt(a.A, (int64_t)d * a.Asd + (int64_t)n * a.Asn, a.adt) : 0.f;
    x[n] = 0.f;
    Bc[n] = (!a.Bvar && n < a.N) ? load_rt(a.Bm, (int64_t)d * a.Bsg + (int64_t)n * a.Bsn, a.bdt) : 0.f;
    Cc[n] = (!a.Cvar && n < a.N) ? load_rt(a.Cm, (int64_t)d * a.Csg + (int64_t)n * a.Csn, a.cdt) : 0.f;
  }
  const float Dv = a.D ? load_rt(a.D, d, a.ddt) : 0.f;
  const float db = a.dbias ? load_rt(a.dbias, d, a.dbdt) : 0.f;
  const T* ug = (const T*)a.u + (int64_t)b * a.usb;
  const T* dg = (const T*)a.delta + (int64_t)b * a.dsb;
  const T* zg = a.z ? (const T*)a.z + (int64_t)b * a.zsb : nullptr;
  T* og = a.out ? (T*)a.out + (int64_t)b * a.osb : nullptr;
  for (int l0 = 0; l0 < a.L; l0 += SS_TL) {
    const int nl = (a.L - l0) < SS_TL ? (a.L - l0) : SS_TL;
    ss_load_tile<T>(ug, a.usd, a.usl, d0, nd, l0, nl, su, DT);
    ss_load_tile<T>(dg, a.dsd, a.dsl, d0, nd, l0, nl, sd, DT);
    if (zg) ss_load_tile<T>(zg, a.zsd, a.zsl, d0, nd, l0, nl, sz, DT);
    for (int i = threadIdx.x; i < SS_TL * a.N; i += blockDim.x) {
      int n = i / SS_TL, t = i % SS_TL;   // t fastest: B/C are (.., N, L) with L contiguous upstream
      if (a.Bvar) sB[t * a.N + n] = t < nl ? load_rt(a.Bm, (int64_t)b * a.Bsb + (int64_t)g * a.Bsg + (int64_t)n * a.Bsn + (int64_t)(l0 + t) * a.Bsl, a.bdt) : 0.f;
      if (a.Cvar) sC[t * a.N + n] = t < nl ? load_rt(a.Cm, (int64_t)b * a.Csb + (int64_t)g * a.Csg + (int64_t)n * a.Csn + (int64_t)(l0 + t) * a.Csl, a.cdt) : 0.f;
    }
    block_sync();
    for (int t = 0; t < nl; t++) {
      if (a.ckpt && live && ((l0 + t) % a.TLB) == 0) {   // state BEFORE token l0 + t (backward restarts here)
        float* cp = a.ckpt + (((int64_t)b * a.Dm + d) * a.nTB + (l0 + t) / a.TLB) * a.N;
#pragma unroll
        for (int n = 0; n < NREG; n++)
          if (n < a.N) cp[n] = x[n];
      }
      float uu = to_f32(su[t * DT + c]);
      float dl = to_f32(sd[t * DT + c]) + db;
      if (a.softplus) dl = softplus_f(dl);
      const float du = dl * uu;
      float y = 0.f;
#pragma unroll
      for (int n = 0; n < NREG; n++) {
        if (n < a.N) {
          const float Bv = a.Bvar ? sB[t * a.N + n] : Bc[n];
          const float Cv = a.Cvar ? sC[t * a.N + n] : Cc[n];
          x[n] = expf(dl * A[n]) * x[n] + du * Bv;
          y += x[n] * Cv;
        }
      }
      y += Dv * uu;
      if (zg) y *= silu_f(to_f32(sz[t * DT + c]));
      so[t * DT + c] = from_f32<T>(y);
    }
    block_sync();
    if (a.out) ss_store_tile<T>(og, a.osd, a.osl, d0, nd, l0, nl, so, DT);
    // the next iteration's loads overwrite su/sd/sz only (not so); its first block_sync orders them after this store
  }
  if (a.last && live) {
#pragma unroll
    for (int n = 0; n < NREG; n++)
      if (n < a.N) a.last[((int64_t)b * a.Dm + d) * a.N + n] = x[n];
  }
}

// ---------------------------------------------------------------------------------------------------------
// forward for L-contiguous storage (upstream's (B, D, L) / (B, G, N, L)): CHUNKED ASSOCIATIVE SCAN, lanes = time.
//
// One wave owns one (batch, channel) sequence.  A pass covers 64 * SSC_LC tokens: lane j holds the SSC_LC consecutive
// tokens of chunk j (u, delta, z, B, C rows are contiguous in L, so every lane reads its own 16-byte vectors and the wave
// reads the whole span; B / C rows are shared by the channel waves of a workgroup through L1).  Per state index n:
//   reduce     the lane folds its chunk into the affine pair (P, X): x_end = P x_start + X      (a_t = exp2(delta_t A2))
//   scan       inclusive scan of the 64 pairs across the wave, (P2, X2) o (P1, X1) = (P2 P1, P2 X1 + X2): DPP row shifts
//              inside rows of 16 lanes + two row broadcasts -- no LDS, no barrier
//   downsweep  every lane re-walks its chunk from its true start state and accumulates y_t += C_t[n] x_t[n]
// The carry into the next pass of 64 * SSC_LC tokens is lane 63's end state, kept per n in a small LDS array private to
// the wave.  No cross-lane reduction over n (the n loop is inside the lane), no atomics.
// ---------------------------------------------------------------------------------------------------------
constexpr int SSC_LC_MAX = 16;   // tokens per lane and pass: 16 (fewer scans per token) or 8 (half the registers: more waves per SIMD)

__device__ __forceinline__ void wave_scan_affine(float& p, float& x) {
#ifdef OMK_EMU
  for (int off = 1; off < 64; off <<= 1) {
    const float ps = shfl_up(p, off), xs = shfl_up(x, off);
    if (lane_id() >= off) { x = fmaf(p, xs, x); p = p * ps; }
  }
#else
  // One step = v_fmac_f32_dpp (x += p * x_src) + v_mul_f32_dpp (p *= p_src), the DPP modifier on the shifted operand.  A lane
  // without a source (start of a row, rows outside the row mask) is simply not written -- the identity pair for free; through
  // __builtin_amdgcn_update_dpp the compiler built every step from two moves of the identity, two v_mov_dpp, a nop, the fmac and
  // the mul (7 instructions x 6 steps of a VALU-bound loop).  Hand-placed wait states: a DPP read needs two after the VALU write
  // of its source (x: mul + nop; p: nop + fmac); inline assembly gets no hazard recogniser.
#define OMK_AFF_STEP(ctrl) \
  "v_fmac_f32_dpp %0, %0, %1 " ctrl "\n\tv_mul_f32_dpp %1, %1, %1 " ctrl "\n\ts_nop 0\n\t"
  asm volatile("s_nop 1\n\t"
               OMK_AFF_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
               OMK_AFF_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
               OMK_AFF_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
               OMK_AFF_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
               OMK_AFF_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
               OMK_AFF_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
               : "+v"(x), "+v"(p));
#undef OMK_AFF_STEP
#endif
}

// SSC_LC consecutive elements of a unit-stride row starting at t0 (tokens >= L read as 0); 16-byte vectors when possible
template <class T, int SSC_LC>
__device__ __forceinline__ void ssc_load_row(const T* row, int t0, int L, float (&out)[SSC_LC]) {
  constexpr int VEC = 16 / sizeof(T);
  const T* p = row + t0;
  if (t0 + SSC_LC <= L && (((uintptr_t)p) & 15) == 0) {
#pragma unroll
    for (int i = 0; i < SSC_LC / VEC; i++) {
      float v[VEC];
      load_vec<T, VEC>(p + i * VEC, v);
#pragma unroll
      for (int e = 0; e < VEC; e++) out[i * VEC + e] = v[e];
    }
  } else {
#pragma unroll
    for (int i = 0; i < SSC_LC; i++) out[i] = (t0 + i < L) ? to_f32(p[i]) : 0.f;
  }
}
template <class T, int SSC_LC>
__device__ __forceinline__ void ssc_store_row(T* row, int t0, int L, const float (&v)[SSC_LC]) {
  constexpr int VEC = 16 / sizeof(T);
  T* p = row + t0;
  if (t0 + SSC_LC <= L && (((uintptr_t)p) & 15) == 0) {
#pragma unroll
    for (int i = 0; i < SSC_LC / VEC; i++) {
      float w[VEC];
#pragma unroll
      for (int e = 0; e < VEC; e++) w[e] = v[i * VEC + e];
      store_vec<T, VEC>(p + i * VEC, w);
    }
  } else {
#pragma unroll
    for (int i = 0; i < SSC_LC; i++)
      if (t0 + i < L) p[i] = from_f32<T>(v[i]);
  }
}

// general form (any B / C kind and dtype, any channel count): every wave pulls its own rows through L1
template <class T, int SSC_LC>
__global__ __launch_bounds__(256) void selscan_fwd_chunked_kernel(SsArgs a) {
  constexpr bool SHARE = false;
  constexpr int VEC = 16 / sizeof(T);
  __shared__ float scarry[4][64];   // per wave: state at the start of the pass, per n
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t seq = (int64_t)blockIdx.x * 4 + wv;   // the four waves of a workgroup: adjacent channels of one batch element
  if (seq >= (int64_t)a.B * a.Dm) return;
  const int b = (int)(seq / a.Dm), d = (int)(seq % a.Dm);
  const int g = d / (a.Dm / a.G);
  T* sBC = nullptr;
  const T* urow = (const T*)a.u + (int64_t)b * a.usb + (int64_t)d * a.usd;
  const T* drow = (const T*)a.delta + (int64_t)b * a.dsb + (int64_t)d * a.dsd;
  const T* zrow = a.z ? (const T*)a.z + (int64_t)b * a.zsb + (int64_t)d * a.zsd : nullptr;
  T* orow = a.out ? (T*)a.out + (int64_t)b * a.osb + (int64_t)d * a.osd : nullptr;
  const float Dv = a.D ? load_rt(a.D, d, a.ddt) : 0.f;
  const float db = a.dbias ? load_rt(a.dbias, d, a.dbdt) : 0.f;
  float* carry = scarry[wv];
  carry[lane] = 0.f;
  for (int tile0 = 0; tile0 < a.L; tile0 += 64 * SSC_LC) {
    const int t0 = tile0 + lane * SSC_LC;
    // first pass of the chunked backward: the state in front of every pass, (B, D, passes, N) f32
    if (a.ckpt && lane < a.N) a.ckpt[(((int64_t)b * a.Dm + d) * a.nTB + tile0 / (64 * SSC_LC)) * a.N + lane] = carry[lane];
    float u[SSC_LC], dl[SSC_LC], y[SSC_LC];
    ssc_load_row<T, SSC_LC>(urow, t0, a.L, u);
    ssc_load_row<T, SSC_LC>(drow, t0, a.L, dl);
#pragma unroll
    for (int i = 0; i < SSC_LC; i++) {
      float v = dl[i] + db;
      if (a.softplus) v = v > 20.f ? v : 0.6931471805599453f * log2_fast(1.f + exp2_fast(v * LOG2E));
      dl[i] = (t0 + i < a.L) ? v : 0.f;     // tokens past the end: a = 1, b = 0 (the identity pair)
      u[i] *= dl[i];                          // delta_t u_t
      y[i] = 0.f;
    }
    // B_t[n] / C_t[n] rows of the lane's tokens: the rows of n + 1 are requested before n is computed (one L2 round trip per n
    // would otherwise sit in front of every scan)
    auto lds_row = [&](const T* rowp, float (&o)[SSC_LC]) {
#pragma unroll
      for (int v = 0; v < SSC_LC / VEC; v++) {
        float w[VEC];
        load_vec<T, VEC>(rowp + (v * 64 + lane) * VEC, w);
#pragma unroll
        for (int e = 0; e < VEC; e++) o[v * VEC + e] = w[e];
      }
    };
    auto load_bc = [&](int n, float (&Bo)[SSC_LC], float (&Co)[SSC_LC]) {
      if (SHARE && a.Bvar) lds_row(sBC + (size_t)n * (64 * SSC_LC), Bo);
      else if (a.Bvar) {
        const int64_t ro = (int64_t)b * a.Bsb + (int64_t)g * a.Bsg + (int64_t)n * a.Bsn;
        if (a.bdt == dtype_of<T>::value) ssc_load_row<T, SSC_LC>((const T*)a.Bm + ro, t0, a.L, Bo);
        else {
#pragma unroll
          for (int i = 0; i < SSC_LC; i++) Bo[i] = t0 + i < a.L ? load_rt(a.Bm, ro + t0 + i, a.bdt) : 0.f;
        }
      } else {
        const float bc = load_rt(a.Bm, (int64_t)d * a.Bsg + (int64_t)n * a.Bsn, a.bdt);
#pragma unroll
        for (int i = 0; i < SSC_LC; i++) Bo[i] = bc;
      }
      if (SHARE && a.Cvar) lds_row(sBC + ((size_t)(a.Bvar ? a.N : 0) + n) * (64 * SSC_LC), Co);
      else if (a.Cvar) {
        const int64_t ro = (int64_t)b * a.Csb + (int64_t)g * a.Csg + (int64_t)n * a.Csn;
        if (a.cdt == dtype_of<T>::value) ssc_load_row<T, SSC_LC>((const T*)a.Cm + ro, t0, a.L, Co);
        else {
#pragma unroll
          for (int i = 0; i < SSC_LC; i++) Co[i] = t0 + i < a.L ? load_rt(a.Cm, ro + t0 + i, a.cdt) : 0.f;
        }
      } else {
        const float cc = load_rt(a.Cm, (int64_t)d * a.Csg + (int64_t)n * a.Csn, a.cdt);
#pragma unroll
        for (int i = 0; i < SSC_LC; i++) Co[i] = cc;
      }
    };
    float Bn[SSC_LC], Cn[SSC_LC];
    if (!SHARE) load_bc(0, Bn, Cn);
    float sdl = 0.f;   // the chunk's decay is exp2(A2 * sum of delta): one product per n instead of a running one
#pragma unroll
    for (int i = 0; i < SSC_LC; i++) sdl += dl[i];
    for (int n = 0; n < a.N; n++) {
      const float A2 = load_rt(a.A, (int64_t)d * a.Asd + (int64_t)n * a.Asn, a.adt) * LOG2E;
      float Bv[SSC_LC], Cv[SSC_LC];
      if (SHARE) load_bc(n, Bv, Cv);
      else {
#pragma unroll
        for (int i = 0; i < SSC_LC; i++) { Bv[i] = Bn[i]; Cv[i] = Cn[i]; }
        load_bc(n + 1 < a.N ? n + 1 : n, Bn, Cn);
      }
      // ---- reduce: the chunk as an affine map x -> P x + X
      float av[SSC_LC];
      float P = exp2_fast(sdl * A2), X = 0.f;
#pragma unroll
      for (int i = 0; i < SSC_LC; i++) {
        av[i] = exp2_fast(dl[i] * A2);
        Bv[i] *= u[i];                        // b_t = delta_t u_t B_t[n]
        X = fmaf(av[i], X, Bv[i]);
      }
      // ---- scan across the wave, then the start state of this lane's chunk
      wave_scan_affine(P, X);
      const float cin = carry[n];
      const float xend = fmaf(P, cin, X);                 // state at the end of this lane's chunk
      float x = shfl_up(xend, 1);
      if (lane == 0) x = cin;
      const float cout = wave_read_lane(xend, 63);
      // ---- downsweep
#pragma unroll
      for (int i = 0; i < SSC_LC; i++) {
        x = fmaf(av[i], x, Bv[i]);
        y[i] = fmaf(Cv[i], x, y[i]);
      }
      if (lane == 0) carry[n] = cout;
    }
    if (!orow) continue;
    // ---- epilogue: + D u, gate, store   (u[] holds delta u: the raw u is re-read -- it is still in L1)
    float ur[SSC_LC];
    ssc_load_row<T, SSC_LC>(urow, t0, a.L, ur);
    if (zrow) {
      float zv[SSC_LC];
      ssc_load_row<T, SSC_LC>(zrow, t0, a.L, zv);
#pragma unroll
      for (int i = 0; i < SSC_LC; i++) y[i] = fmaf(Dv, ur[i], y[i]) * (zv[i] * rcp_fast(1.f + exp2_fast(-zv[i] * LOG2E)));
    } else {
#pragma unroll
      for (int i = 0; i < SSC_LC; i++) y[i] = fmaf(Dv, ur[i], y[i]);
    }
    ssc_store_row<T, SSC_LC>(orow, t0, a.L, y);
  }
  if (a.last && lane < a.N) a.last[((int64_t)b * a.Dm + d) * a.N + lane] = carry[lane];
}

// ---------------------------------------------------------------------------------------------------------
// the same scan for the common case -- B and C both (B, G, N, L) rows of u's dtype, A fp32, 8 | channels per group: the eight
// waves of a workgroup are eight adjacent channels of one (batch, group) and SHARE the B / C rows of a pass through LDS
// (one coalesced fill per pass; the per-wave form pulls every row through the CU's L1 once per channel), A2 = A log2 e of
// the wave's channel sits in LDS too, and the n loop carries no dtype or layout branches.
// LDS image of a row: the 16-byte piece v of lane j's tokens at slot v * 64 + j, so a ds_read_b128 of a wave is 1 KB contiguous.
// ---------------------------------------------------------------------------------------------------------
// STATE_ONLY (first pass of the chunked backward): no C rows, no y, no output -- the state in front of every pass of 64 * SSC_LC
// tokens goes to a.ckpt as (B, D, passes, N) f32
template <class T, int SSC_LC, int NU, bool STATE_ONLY = false>   // NU = 2: the elementwise half of the n loop on packed pairs of adjacent tokens
__global__ __launch_bounds__(512) void selscan_fwd_shared_kernel(SsArgs a) {
  constexpr int VEC = 16 / sizeof(T), PPR = 64 * SSC_LC / VEC;   // 16-byte pieces per staged row
  __shared__ float scarry[8][64], sA2[8][64];
  OMK_DYN_SMEM(bc_raw);              // [2][N][64 * SSC_LC] of T
  T* sBC = (T*)bc_raw;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tpb = a.Dm / 8;          // workgroup = (batch, 8-channel tile inside one group)
  const int b = blockIdx.x / tpb, d = (blockIdx.x % tpb) * 8 + wv;
  const int g = d / (a.Dm / a.G);
  const T* urow = (const T*)a.u + (int64_t)b * a.usb + (int64_t)d * a.usd;
  const T* drow = (const T*)a.delta + (int64_t)b * a.dsb + (int64_t)d * a.dsd;
  const T* zrow = a.z ? (const T*)a.z + (int64_t)b * a.zsb + (int64_t)d * a.zsd : nullptr;
  T* orow = STATE_ONLY ? nullptr : (T*)a.out + (int64_t)b * a.osb + (int64_t)d * a.osd;
  const T* Bbase = (const T*)a.Bm + (int64_t)b * a.Bsb + (int64_t)g * a.Bsg;
  const T* Cbase = (const T*)a.Cm + (int64_t)b * a.Csb + (int64_t)g * a.Csg;
  const float Dv = a.D ? load_rt(a.D, d, a.ddt) : 0.f;
  const float db = a.dbias ? load_rt(a.dbias, d, a.dbdt) : 0.f;
  float* carry = scarry[wv];
  carry[lane] = 0.f;
  sA2[wv][lane] = lane < a.N ? ((const float*)a.A)[(int64_t)d * a.Asd + (int64_t)lane * a.Asn] * LOG2E : 0.f;
  for (int tile0 = 0; tile0 < a.L; tile0 += 64 * SSC_LC) {
    const int t0 = tile0 + lane * SSC_LC;
    block_sync();   // every wave is done with the rows of the previous pass
    if ((STATE_ONLY || a.ckpt) && lane < a.N) a.ckpt[(((int64_t)b * a.Dm + d) * a.nTB + tile0 / (64 * SSC_LC)) * a.N + lane] = carry[lane];
    for (int i = threadIdx.x; i < (STATE_ONLY ? 1 : 2) * a.N * PPR; i += 512) {
      const int r = i / PPR, q = i % PPR;
      const T* row = r < a.N ? Bbase + (int64_t)r * a.Bsn : Cbase + (int64_t)(r - a.N) * a.Csn;
      const int e0 = q * VEC, tq = tile0 + e0;
      T* dst = sBC + (size_t)r * (64 * SSC_LC) + ((e0 % SSC_LC) / VEC * 64 + e0 / SSC_LC) * VEC;
      const T* src = row + tq;
      if (tq + VEC <= a.L && (((uintptr_t)src) & 15) == 0) {
        *reinterpret_cast<vec_t<T, VEC>*>(dst) = *reinterpret_cast<const vec_t<T, VEC>*>(src);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; e++) dst[e] = tq + e < a.L ? src[e] : T{};
      }
    }
    float u[SSC_LC], dl[SSC_LC], y[SSC_LC];
    ssc_load_row<T, SSC_LC>(urow, t0, a.L, u);
    ssc_load_row<T, SSC_LC>(drow, t0, a.L, dl);
    float sdl = 0.f;   // the chunk's decay is exp2(A2 * sum of delta)
#pragma unroll
    for (int i = 0; i < SSC_LC; i++) {
      float v = dl[i] + db;
      if (a.softplus) v = v > 20.f ? v : 0.6931471805599453f * log2_fast(1.f + exp2_fast(v * LOG2E));
      dl[i] = (t0 + i < a.L) ? v : 0.f;     // tokens past the end: a = 1, b = 0 (the identity pair)
      u[i] *= dl[i];                          // delta_t u_t
      y[i] = 0.f;
      sdl += dl[i];
    }
    block_sync();   // rows staged
    const T* pB = sBC + lane * VEC;
    const T* pC = pB + (size_t)a.N * (64 * SSC_LC);
    if (NU == 2) {
      // the elementwise half of the n loop on PACKED fp32 pairs of adjacent tokens (v_pk_mul_f32 / v_pk_fma_f32: two tokens per
      // issue): delta A2, delta u B and the y accumulation.  The two recurrences (fold, sweep) are serial in the token index and the
      // exponentials are not packable; the loop is VALU-bound, so instructions are time.
      constexpr int HP = SSC_LC / 2;
      f32x2 dl2[HP], u2[HP], y2[HP];
#pragma unroll
      for (int j = 0; j < HP; j++) { dl2[j] = f32x2{dl[2 * j], dl[2 * j + 1]}; u2[j] = f32x2{u[2 * j], u[2 * j + 1]}; y2[j] = f32x2{0.f, 0.f}; }
      for (int n = 0; n < a.N; n++) {
        const float A2 = sA2[wv][n];
        const f32x2 A22 = {A2, A2};
        f32x2 bt2[HP], C2[HP], av2[HP];
#pragma unroll
        for (int v = 0; v < SSC_LC / VEC; v++) {
          float wb[VEC], wc[VEC];
          load_vec<T, VEC>(pB + (size_t)n * (64 * SSC_LC) + v * 64 * VEC, wb);
          if (!STATE_ONLY) load_vec<T, VEC>(pC + (size_t)n * (64 * SSC_LC) + v * 64 * VEC, wc);
#pragma unroll
          for (int e = 0; e < VEC; e += 2) {
            bt2[(v * VEC + e) / 2] = f32x2{wb[e], wb[e + 1]};
            C2[(v * VEC + e) / 2] = STATE_ONLY ? f32x2{0.f, 0.f} : f32x2{wc[e], wc[e + 1]};
          }
        }
        float P = exp2_fast(sdl * A2), X = 0.f;
#pragma unroll
        for (int j = 0; j < HP; j++) {
          const f32x2 e2 = dl2[j] * A22;
          av2[j] = f32x2{exp2_fast(e2[0]), exp2_fast(e2[1])};
          bt2[j] = bt2[j] * u2[j];                 // b_t = delta_t u_t B_t[n]
          X = fmaf(av2[j][0], X, bt2[j][0]);
          X = fmaf(av2[j][1], X, bt2[j][1]);
        }
        wave_scan_affine(P, X);
        const float cin = carry[n];
        const float xend = fmaf(P, cin, X);
        float x = shfl_up(xend, 1);
        if (lane == 0) x = cin;
        const float cout = wave_read_lane(xend, 63);
        if (!STATE_ONLY) {
#pragma unroll
          for (int j = 0; j < HP; j++) {
            const float xa = fmaf(av2[j][0], x, bt2[j][0]);
            x = fmaf(av2[j][1], xa, bt2[j][1]);
            y2[j] = C2[j] * f32x2{xa, x} + y2[j];
          }
        }
        if (lane == 0) carry[n] = cout;
      }
#pragma unroll
      for (int j = 0; j < HP; j++) { y[2 * j] = y2[j][0]; y[2 * j + 1] = y2[j][1]; }
    } else {
    for (int n = 0; n < a.N; n++) {
      const float A2 = sA2[wv][n];
      float Bv[SSC_LC], Cv[SSC_LC];
#pragma unroll
      for (int v = 0; v < SSC_LC / VEC; v++) {
        float wb[VEC], wc[VEC];
        load_vec<T, VEC>(pB + (size_t)n * (64 * SSC_LC) + v * 64 * VEC, wb);
        if (!STATE_ONLY) load_vec<T, VEC>(pC + (size_t)n * (64 * SSC_LC) + v * 64 * VEC, wc);
#pragma unroll
        for (int e = 0; e < VEC; e++) { Bv[v * VEC + e] = wb[e]; Cv[v * VEC + e] = STATE_ONLY ? 0.f : wc[e]; }
      }
      float av[SSC_LC];
      float P = exp2_fast(sdl * A2), X = 0.f;
#pragma unroll
      for (int i = 0; i < SSC_LC; i++) {
        av[i] = exp2_fast(dl[i] * A2);
        Bv[i] *= u[i];                        // b_t = delta_t u_t B_t[n]
        X = fmaf(av[i], X, Bv[i]);
      }
      wave_scan_affine(P, X);
      const float cin = carry[n];
      const float xend = fmaf(P, cin, X);     // state at the end of this lane's chunk
      float x = shfl_up(xend, 1);
      if (lane == 0) x = cin;
      const float cout = wave_read_lane(xend, 63);
      if (!STATE_ONLY) {
#pragma unroll
        for (int i = 0; i < SSC_LC; i++) {
          x = fmaf(av[i], x, Bv[i]);
          y[i] = fmaf(Cv[i], x, y[i]);
        }
      }
      if (lane == 0) carry[n] = cout;
    }
    }
    if (STATE_ONLY) continue;
    float ur[SSC_LC];
    ssc_load_row<T, SSC_LC>(urow, t0, a.L, ur);
    if (zrow) {
      float zv[SSC_LC];
      ssc_load_row<T, SSC_LC>(zrow, t0, a.L, zv);
#pragma unroll
      for (int i = 0; i < SSC_LC; i++) y[i] = fmaf(Dv, ur[i], y[i]) * (zv[i] * rcp_fast(1.f + exp2_fast(-zv[i] * LOG2E)));
    } else {
#pragma unroll
      for (int i = 0; i < SSC_LC; i++) y[i] = fmaf(Dv, ur[i], y[i]);
    }
    ssc_store_row<T, SSC_LC>(orow, t0, a.L, y);
  }
  if (a.last && lane < a.N) a.last[((int64_t)b * a.Dm + d) * a.N + lane] = carry[lane];
}

// ---------------------------------------------------------------------------------------------------------
// forward for MANY sequences: lanes = CHANNELS, one sweep (selscan_fwd_lanes_kernel).
//
// With B * D / 64 >= a few hundred waves the chip is full without cutting time into chunks, and the recurrence can run
// the way it is written: one wave = 64 adjacent channels of one (batch, group), a lane keeps the d_state <= 16 states of ITS channel
// in registers as 8 packed pairs and walks the tokens once -- per token and pair one v_pk_mul (delta A2), two v_exp, one v_pk_mul
// (delta u B), one v_pk_fma (state), one v_pk_fma (y): 3 issue slots per (token, state) element where the chunked form (fold + wave scan +
// second sweep) needs 10.  B_t / C_t are the same for every lane: staged per block of 16 tokens into a wave-private [token][state] LDS
// tile and read back as broadcasts.  Storage:
//   channel-last (B, L, D) -- "(B, L, D) laid out for coalesced HBM loads": a token's 64 channels are one 256-byte row (fp32), the lane
//     loads ITS element straight into the register the sweep consumes, and re-issues the load of the same token slot of the NEXT block
//     as soon as the value is consumed (a block time in flight, no second register set, no LDS); y goes out the same way
//   L-contiguous (B, D, L), upstream's layout: 16 tokens x 64 channels tiles transposed through LDS on the way in and out (row stride 68
//     floats: the 16-token x 4-channel footprint of one load instruction hits 64 different banks)
// All global traffic through buffer resources (ssd_tiles.h): the lane part of an address is a 32-bit constant, the token part a
// scalar, and whatever lies behind the end of the batch element's range reads as zero / is dropped.
// ---------------------------------------------------------------------------------------------------------
constexpr int SCL_TB = 16, SCL_S = 68, SCL_SB = 20;
constexpr uint32_t SCL_OOR = 0x80000000u;   // lane offset behind every range (ranges are < 2^31 bytes): loads return 0, stores are dropped

template <class T, bool LC>
__global__ __launch_bounds__(64) void selscan_fwd_lanes_kernel(SsArgs a) {
  __shared__ __attribute__((aligned(16))) float sB[SCL_TB * SCL_SB];
  __shared__ __attribute__((aligned(16))) float sC[SCL_TB * SCL_SB];
  __shared__ float tu[LC ? SCL_TB * SCL_S : 1], td[LC ? SCL_TB * SCL_S : 1], tz[LC ? SCL_TB * SCL_S : 1], to[LC ? SCL_TB * SCL_S : 1];
  constexpr uint32_t ES = sizeof(T);
  constexpr float LN2 = 0.6931471805599453f;
  const int lane = threadIdx.x;
  const int dpg = a.Dm / a.G, tpg = (dpg + 63) / 64;
  const int tg = blockIdx.x % tpg, g = (blockIdx.x / tpg) % a.G, b = blockIdx.x / (tpg * a.G);
  const int d0 = g * dpg + tg * 64;
  const int nd = (dpg - tg * 64) < 64 ? (dpg - tg * 64) : 64;
  const bool live = lane < nd;
  const int d = d0 + (live ? lane : 0);
  const int r4 = lane >> 4, t16 = lane & 15;
  f32x2 A2[8], x2[8];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    A2[j][0] = 2 * j < a.N ? load_rt(a.A, (int64_t)d * a.Asd + (int64_t)(2 * j) * a.Asn, a.adt) * LOG2E : 0.f;
    A2[j][1] = 2 * j + 1 < a.N ? load_rt(a.A, (int64_t)d * a.Asd + (int64_t)(2 * j + 1) * a.Asn, a.adt) * LOG2E : 0.f;
    x2[j] = f32x2{0.f, 0.f};
  }
  const float Dv = a.D ? load_rt(a.D, d, a.ddt) : 0.f;
  const float db = a.dbias ? load_rt(a.dbias, d, a.dbdt) : 0.f;
  const bool hasz = a.z != nullptr;
  const BufRes Ur = make_buf((const T*)a.u + (int64_t)b * a.usb, a.xu), Dr = make_buf((const T*)a.delta + (int64_t)b * a.dsb, a.xd);
  const BufRes Zr = make_buf(hasz ? (const T*)a.z + (int64_t)b * a.zsb : nullptr, hasz ? a.xz : 0u);
  const BufRes Or = make_buf((T*)a.out + (int64_t)b * a.osb, a.xo);
  const BufRes Br = make_buf((const T*)a.Bm + (int64_t)b * a.Bsb + (int64_t)g * a.Bsg, a.xB);
  const BufRes Cr = make_buf((const T*)a.Cm + (int64_t)b * a.Csb + (int64_t)g * a.Csg, a.xC);
  const uint32_t usl = (uint32_t)a.usl, dsl = (uint32_t)a.dsl, zsl = (uint32_t)a.zsl, osl = (uint32_t)a.osl;
  const uint32_t usd = (uint32_t)a.usd, dsd = (uint32_t)a.dsd, zsd = (uint32_t)a.zsd, osd = (uint32_t)a.osd;
  // ---- B / C rows of a block: 16 tokens x 16 states = four elements per lane and array.  The lane's 16-lane group follows whichever
  // of (state, token) has unit stride; states >= d_state are zeros in LDS (their A2 is 0: x stays 0, y gets nothing)
  const bool bl = a.Bsl == 1, cl = a.Csl == 1;
  const uint32_t bvo = ES * (bl ? (uint32_t)r4 * (uint32_t)a.Bsn + (uint32_t)t16 * (uint32_t)a.Bsl : (uint32_t)t16 * (uint32_t)a.Bsn + (uint32_t)r4 * (uint32_t)a.Bsl);
  const uint32_t cvo = ES * (cl ? (uint32_t)r4 * (uint32_t)a.Csn + (uint32_t)t16 * (uint32_t)a.Csl : (uint32_t)t16 * (uint32_t)a.Csn + (uint32_t)r4 * (uint32_t)a.Csl);
  const uint32_t bks = ES * 4u * (uint32_t)(bl ? a.Bsn : a.Bsl), cks = ES * 4u * (uint32_t)(cl ? a.Csn : a.Csl);
  const int blo = bl ? t16 * SCL_SB + r4 : r4 * SCL_SB + t16, bls = bl ? 4 : 4 * SCL_SB;
  const int clo = cl ? t16 * SCL_SB + r4 : r4 * SCL_SB + t16, cls = cl ? 4 : 4 * SCL_SB;
  uint32_t pb[4], pc[4];
  auto fetch_bc = [&](int l0) {
    const uint32_t sb0 = ES * (uint32_t)l0 * (uint32_t)a.Bsl, sc0 = ES * (uint32_t)l0 * (uint32_t)a.Csl;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      pb[k] = buf_ld_raw<T>(Br, bvo, sb0 + (uint32_t)k * bks);
      pc[k] = buf_ld_raw<T>(Cr, cvo, sc0 + (uint32_t)k * cks);
    }
  };
  auto commit_bc = [&]() {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool okb = (bl ? r4 + 4 * k : t16) < a.N, okc = (cl ? r4 + 4 * k : t16) < a.N;
      sB[blo + k * bls] = okb ? raw_to_f32<T>(pb[k]) : 0.f;
      sC[clo + k * cls] = okc ? raw_to_f32<T>(pc[k]) : 0.f;
    }
  };
  // ---- one token of the lane's channel: row = the token's slot in the B / C tile; a slot behind the end of the sequence (valid = false:
  // its loads came back as zeros) is the identity step delta = 0 -- no branch around the body, whose loads the compiler then counts exactly
  struct Rows { f32x4 b[4], c[4]; };   // B_t / C_t of one token as the broadcast reads deliver them
  auto read_rows = [&](int row) -> Rows {
    Rows r;
    const f32x4* rb = reinterpret_cast<const f32x4*>(&sB[row * SCL_SB]);
    const f32x4* rc = reinterpret_cast<const f32x4*>(&sC[row * SCL_SB]);
#pragma unroll
    for (int q = 0; q < 4; q++) { r.b[q] = rb[q]; r.c[q] = rc[q]; }
    return r;
  };
  auto token = [&](float uu, float draw, float zv, const Rows& rows, bool valid) -> float {
    float dl = draw + db;
    if (a.softplus) dl = dl > 20.f ? dl : LN2 * log2_fast(1.f + exp2_fast(dl * LOG2E));
    dl = valid ? dl : 0.f;
    const float du = dl * uu;
    // real register pairs: as {v, v} the compiler feeds v_pk_* the low half twice out of (v, whatever register follows) -- and when that
    // neighbour is the destination of a load still in flight, every packed op waits for the load
    f32x2 dl2 = {dl, dl}, du2 = {du, du};
    OMK_OPAQUE(dl2); OMK_OPAQUE(du2);
    f32x2 ya = {0.f, 0.f}, yb = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = rows.b[q], c4 = rows.c[q];
      {
        const f32x2 e = dl2 * A2[2 * q];
        const f32x2 av = {exp2_fast(e[0]), exp2_fast(e[1])};
        x2[2 * q] = av * x2[2 * q] + du2 * f32x2{b4[0], b4[1]};
        ya = f32x2{c4[0], c4[1]} * x2[2 * q] + ya;
      }
      {
        const f32x2 e = dl2 * A2[2 * q + 1];
        const f32x2 av = {exp2_fast(e[0]), exp2_fast(e[1])};
        x2[2 * q + 1] = av * x2[2 * q + 1] + du2 * f32x2{b4[2], b4[3]};
        yb = f32x2{c4[2], c4[3]} * x2[2 * q + 1] + yb;
      }
    }
    const f32x2 ys = ya + yb;
    float y = fmaf(Dv, uu, ys[0] + ys[1]);
    if (hasz) y *= zv * rcp_fast(1.f + exp2_fast(-zv * LOG2E));
    return y;
  };
  auto checkpoint = [&](int l0) {   // state in front of token l0 (a multiple of a.TLB): what the chunked backward restarts from
    if (a.ckpt && live && (l0 % a.TLB) == 0) {
      if (a.ckpt_cl) {
        float* cp = a.ckpt + (((int64_t)b * a.nTB + l0 / a.TLB) * a.N) * a.Dm + d;
#pragma unroll
        for (int n = 0; n < 16; n++)
          if (n < a.N) cp[(int64_t)n * a.Dm] = x2[n >> 1][n & 1];
      } else {
        float* cp = a.ckpt + (((int64_t)b * a.Dm + d) * a.nTB + l0 / a.TLB) * a.N;
#pragma unroll
        for (int n = 0; n < 16; n++)
          if (n < a.N) cp[n] = x2[n >> 1][n & 1];
      }
    }
  };

  if (!LC) {
    // ---- channel-last: registers only.  (z absent: the range of Zr is empty, its loads return zeros without touching memory)
    const uint32_t uvo = live ? ES * (uint32_t)(d0 + lane) * usd : SCL_OOR, dvo = live ? ES * (uint32_t)(d0 + lane) * dsd : SCL_OOR;
    const uint32_t zvo = (live && hasz) ? ES * (uint32_t)(d0 + lane) * zsd : SCL_OOR, ovo = live ? ES * (uint32_t)(d0 + lane) * osd : SCL_OOR;
    const uint32_t ustep = ES * usl, dstep = ES * dsl, zstep = hasz ? ES * zsl : 0u, ostep = ES * osl;
    uint32_t pu[SCL_TB], pd[SCL_TB], pz[SCL_TB];
    uint32_t su = 0u, sd = 0u, sz = 0u, so = 0u;   // running scalar offsets: the loads are one block ahead of the stores
#pragma unroll
    for (int k = 0; k < SCL_TB; k++) {
      pu[k] = buf_ld_raw<T>(Ur, uvo, su);
      pd[k] = buf_ld_raw<T>(Dr, dvo, sd);
      pz[k] = buf_ld_raw<T>(Zr, zvo, sz);
      su += ustep; sd += dstep; sz += zstep;
      OMK_OPAQUE_S(su); OMK_OPAQUE_S(sd); OMK_OPAQUE_S(sz);
    }
    fetch_bc(0);
    OMK_VM_DRAIN();   // (nothing of the prologue pending at the loop header: the loop's own wait counts stay exact)
    for (int l0 = 0; l0 < a.L; l0 += SCL_TB) {
      wave_lds_sync();   // the previous block's broadcast reads before the new rows
      commit_bc();
      wave_lds_sync();
      checkpoint(l0);
      // the rows of token k + 1 are requested in front of token k's arithmetic (the scheduling fence behind it would otherwise pin every
      // token's eight LDS reads right in front of their use: counters showed 24 % of the wave's cycles in s_waitcnt, one wave per SIMD)
      Rows rows = read_rows(0);
#pragma unroll
      for (int k = 0; k < SCL_TB; k++) {
        if (k == SCL_TB / 2) fetch_bc(l0 + SCL_TB);   // half a block ahead: 32 younger operations, inside what s_waitcnt can count
        Rows next = rows;
        if (k + 1 < SCL_TB) next = read_rows(k + 1);
        const float y = token(raw_to_f32<T>(pu[k]), raw_to_f32<T>(pd[k]), raw_to_f32<T>(pz[k]), rows, l0 + k < a.L);
        rows = next;
        // the same slot of the next block (behind the end: zeros) into the registers the token just released -- issued BEHIND their
        // last use: with the old and the new value alive together the loop-carried slot becomes a copy at the back edge, and the copy
        // a wait for (nearly) every load of the block
        OMK_SCHED_FENCE();
        pu[k] = buf_ld_raw<T>(Ur, uvo, su);
        pd[k] = buf_ld_raw<T>(Dr, dvo, sd);
        pz[k] = buf_ld_raw<T>(Zr, zvo, sz);
        su += ustep; sd += dstep; sz += zstep;
        OMK_OPAQUE_S(su); OMK_OPAQUE_S(sd); OMK_OPAQUE_S(sz);
        buf_st_t<T>(Or, y, ovo, so);          // (behind the end of the sequence: behind the range, dropped)
        so += ostep;
        OMK_OPAQUE_S(so);
      }
    }
  } else {
    // ---- L-contiguous: load instruction k of a block covers channels r4 + 4 k, tokens t16 (64-byte pieces of four rows)
    const uint32_t uvo = ES * ((uint32_t)(d0 + r4) * usd + (uint32_t)t16), dvo = ES * ((uint32_t)(d0 + r4) * dsd + (uint32_t)t16);
    const uint32_t zvo = hasz ? ES * ((uint32_t)(d0 + r4) * zsd + (uint32_t)t16) : SCL_OOR, ovo = ES * ((uint32_t)(d0 + r4) * osd + (uint32_t)t16);
    const int tlo = t16 * SCL_S + r4;   // tile element [token t16][channel r4 + 4 k]
    uint32_t pu[SCL_TB], pd[SCL_TB], pz[SCL_TB];
    const uint32_t ustep = ES * 4u * usd, dstep = ES * 4u * dsd, zstep = hasz ? ES * 4u * zsd : 0u, ostep = ES * 4u * osd;
    auto fetch_tiles = [&](int l0) {
      uint32_t su = ES * (uint32_t)l0, sd = su, sz = hasz ? su : 0u;
#pragma unroll
      for (int k = 0; k < SCL_TB; k++) {
        pu[k] = buf_ld_raw<T>(Ur, uvo, su);
        pd[k] = buf_ld_raw<T>(Dr, dvo, sd);
        pz[k] = buf_ld_raw<T>(Zr, zvo, sz);
        su += ustep; sd += dstep; sz += zstep;
        OMK_OPAQUE_S(su); OMK_OPAQUE_S(sd); OMK_OPAQUE_S(sz);
      }
    };
    fetch_tiles(0);
    fetch_bc(0);
    for (int l0 = 0; l0 < a.L; l0 += SCL_TB) {
      const int nl = (a.L - l0) < SCL_TB ? (a.L - l0) : SCL_TB;
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < SCL_TB; k++) {
        tu[tlo + 4 * k] = raw_to_f32<T>(pu[k]);
        td[tlo + 4 * k] = raw_to_f32<T>(pd[k]);
        if (hasz) tz[tlo + 4 * k] = raw_to_f32<T>(pz[k]);
      }
      commit_bc();
      fetch_tiles(l0 + SCL_TB);
      fetch_bc(l0 + SCL_TB);
      wave_lds_sync();
      checkpoint(l0);
      if (nl == SCL_TB) {   // a full block as one straight run of 16 tokens (the tile reads of a token overlap its neighbours' arithmetic)
#pragma unroll
        for (int t = 0; t < SCL_TB; t++) {
          const float y = token(tu[t * SCL_S + lane], td[t * SCL_S + lane], hasz ? tz[t * SCL_S + lane] : 0.f, read_rows(t), true);
          to[t * SCL_S + lane] = y;
        }
      } else {
        for (int t = 0; t < nl; t++) {
          const float y = token(tu[t * SCL_S + lane], td[t * SCL_S + lane], hasz ? tz[t * SCL_S + lane] : 0.f, read_rows(t), true);
          to[t * SCL_S + lane] = y;
        }
      }
      wave_lds_sync();
      const bool tok = l0 + t16 < a.L;
      uint32_t so = ES * (uint32_t)l0;
#pragma unroll
      for (int k = 0; k < SCL_TB; k++) {
        const bool ok = tok && r4 + 4 * k < nd;
        buf_st_t<T>(Or, to[tlo + 4 * k], ok ? ovo : SCL_OOR, so);
        so += ostep;
        OMK_OPAQUE_S(so);
      }
    }
  }
  if (a.last && live) {
#pragma unroll
    for (int n = 0; n < 16; n++)
      if (n < a.N) a.last[((int64_t)b * a.Dm + d) * a.N + n] = x2[n >> 1][n & 1];
  }
}

// ---------------------------------------------------------------------------------------------------------
// backward.  With a_t = exp(delta_t A), the adjoint g_t of the state x_t runs backwards in time:
//   g_t = C_t dy_t + a_{t+1} g_{t+1}                       dy_t = dout_t * silu(z_t) (or dout_t)
//   dA   += g_t x_{t-1} delta_t a_t          ddelta_t = sum_n g_t (x_{t-1} A a_t + B_t u_t)   (* softplus' when enabled)
//   du_t  = delta_t sum_n g_t B_t + D dy_t   dB_t = g_t delta_t u_t (summed over the channels of the group)
//   dC_t  = dy_t x_t (same sum)              dz_t = dout_t (y_t + D u_t) silu'(z_t)
// x_t is rebuilt per tile of SSB_TL tokens from the checkpoints the forward kernel leaves in the workspace (a.ckpt) and
// parked in LDS ([t][n][channel]); one wave = 64 channels of one group and batch element, tiles walked last to first.
// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// backward for L-contiguous storage: the SAME chunked associative scan, run in both directions inside the n loop.
//
// Workgroup = NW (16 or 8) adjacent channels of one (batch, group), one wave per channel, lanes = 64 time chunks of 8 tokens,
// passes of 512 tokens walked LAST to FIRST.  Per state index n and pass:
//   forward   fold + wave scan + sweep from the pass-start state (left by the STATE_ONLY forward pass): x_{t-1} stays in 8
//             registers, dy_t x_t goes to the dC accumulator, C_t x_t to the y needed by dz
//   reverse   G_t = a_t (C_t dy_t + G_{t+1}) is affine in G_{t+1} with the SAME chunk decay: fold the chunk backwards, suffix
//             scan of the 64 pairs (DPP row_shl inside rows of 16, the three row totals through v_readlane), sweep backwards:
//             g_t = C_t dy_t + G_{t+1} meets x_{t-1} -> du, ddelta (registers), dA (lane sum, one wave reduction per n),
//             g_t delta_t u_t -> the dB accumulator
// dB / dC are sums over the channels of a group: the waves of a workgroup add into fp32 LDS rows (16-byte read-add-write, the
// waves walk the state indices in rotated order so that no row has two owners -- see the n loop), a workgroup may walk several
// channel tiles per pass (q.nOct) before the rows go to HBM as fp32 atomics -- one atomic per (tile group, n, token) instead of
// one per channel.  State indices are processed in blocks of q.NB rows
// (LDS: NB * 512 * (8 + 2 sizeof(T)) bytes); with more than one block a workgroup keeps one channel tile.
// ---------------------------------------------------------------------------------------------------------
constexpr int SSR_LC = 8, SSR_TP = 64 * SSR_LC;

__device__ __forceinline__ void wave_scan_affine_rev(float& p, float& x) {
  // suffix composition: lane j <- pairs of lanes j .. 63, the pair of the LOWER lane applied last
#ifdef OMK_EMU
  for (int off = 1; off < 64; off <<= 1) {
    const float ps = shfl_down(p, off), xs = shfl_down(x, off);
    if (lane_id() + off < 64) { x = fmaf(p, xs, x); p = p * ps; }
  }
#else
  // (steps as in wave_scan_affine: the DPP modifier on the fmac / mul themselves, lanes without a source are not written)
#define OMK_AFF_STEP(ctrl) \
  "v_fmac_f32_dpp %0, %0, %1 " ctrl "\n\tv_mul_f32_dpp %1, %1, %1 " ctrl "\n\ts_nop 0\n\t"
  asm volatile("s_nop 1\n\t"
               OMK_AFF_STEP("row_shl:1 row_mask:0xf bank_mask:0xf")
               OMK_AFF_STEP("row_shl:2 row_mask:0xf bank_mask:0xf")
               OMK_AFF_STEP("row_shl:4 row_mask:0xf bank_mask:0xf")
               OMK_AFF_STEP("row_shl:8 row_mask:0xf bank_mask:0xf")
               : "+v"(x), "+v"(p));
#undef OMK_AFF_STEP
  // lanes 16 / 32 / 48 hold the totals of rows 1 / 2 / 3; row r still needs rows r + 1 .. 3
  const float p1 = wave_read_lane(p, 16), x1 = wave_read_lane(x, 16), p2 = wave_read_lane(p, 32), x2 = wave_read_lane(x, 32);
  const float p3 = wave_read_lane(p, 48), x3 = wave_read_lane(x, 48);
  const float p23 = p2 * p3, x23 = fmaf(p2, x3, x2), p123 = p1 * p23, x123 = fmaf(p1, x23, x1);
  const int row = lane_id() >> 4;
  const float ps = row == 0 ? p123 : row == 1 ? p23 : row == 2 ? p3 : 1.f;
  const float xs = row == 0 ? x123 : row == 1 ? x23 : row == 2 ? x3 : 0.f;
  x = fmaf(p, xs, x); p = p * ps;
#endif
}

constexpr int SSB_TL = 16, SSB_DT = 64, SSB_N = 16;

struct SsBwdArgs {
  SsArgs f;
  const void* dout; int64_t gsb, gsd, gsl;
  void* du; void* ddelta; void* dz; int64_t dusb, dusd, dusl, ddsb, ddsd, ddsl, dzsb, dzsd, dzsl;
  float* dA; float* dB; float* dC; float* dD; float* ddb;
  int64_t dBsb, dBsg, dBsn, dBsl, dCsb, dCsg, dCsn, dCsl;   // variable: (B, G, N, L); constant: dBsg = stride over d, dBsn over n
  int NB, nOct;   // chunked form: state indices per LDS block, channel tiles per workgroup
  int dbg;        // developer ablation bits (OMK_SELSCAN_BWD_DBG): 2 no flush atomics, 4 no n loop
  uint32_t xg, xdu, xdd, xdz;   // lanes = channels form: bytes one batch element of dout / du / ddelta / dz spans (buffer ranges)
};

template <class T>
__global__ __launch_bounds__(SSB_DT) void selscan_bwd_kernel(SsBwdArgs q) {
  const SsArgs& a = q.f;
  OMK_DYN_SMEM(smem);
  constexpr int DT = SSB_DT, TL = SSB_TL, NR = SSB_N;
  float* sx = (float*)smem;                  // [TL][NR][DT]
  float* sB = sx + TL * NR * DT;             // [TL][N]
  float* sC = sB + TL * NR;
  float* sdB = sC + TL * NR;
  float* sdC = sdB + TL * NR;
  T* su = (T*)(sdC + TL * NR);
  T* sd = su + TL * DT;
  T* sz = sd + TL * DT;
  T* sg = sz + TL * DT;
  T* sdu = sg + TL * DT;
  T* sdd = sdu + TL * DT;
  T* sdz = sdd + TL * DT;
  const int dpg = a.Dm / a.G;
  const int tiles_per_group = (dpg + DT - 1) / DT;
  const int tg = blockIdx.x % tiles_per_group, g = (blockIdx.x / tiles_per_group) % a.G, b = blockIdx.x / (tiles_per_group * a.G);
  const int d0 = g * dpg + tg * DT;
  const int nd = (dpg - tg * DT) < DT ? (dpg - tg * DT) : DT;
  const int c = threadIdx.x;
  const bool live = c < nd;
  const int d = d0 + (live ? c : 0);
  float A[NR], gx[NR], Bc[NR], Cc[NR], dAacc[NR], dBc[NR], dCc[NR];
#pragma unroll
  for (int n = 0; n < NR; n++) {
    A[n] = n < a.N ? load_rt(a.A, (int64_t)d * a.Asd + (int64_t)n * a.Asn, a.adt) : 0.f;
    gx[n] = 0.f; dAacc[n] = 0.f; dBc[n] = 0.f; dCc[n] = 0.f;
    Bc[n] = (!a.Bvar && n < a.N) ? load_rt(a.Bm, (int64_t)d * a.Bsg + (int64_t)n * a.Bsn, a.bdt) : 0.f;
    Cc[n] = (!a.Cvar && n < a.N) ? load_rt(a.Cm, (int64_t)d * a.Csg + (int64_t)n * a.Csn, a.cdt) : 0.f;
  }
  const float Dv = a.D ? load_rt(a.D, d, a.ddt) : 0.f;
  const float db = a.dbias ? load_rt(a.dbias, d, a.dbdt) : 0.f;
  float dDacc = 0.f, ddbacc = 0.f;
  const T* ug = (const T*)a.u + (int64_t)b * a.usb;
  const T* dg = (const T*)a.delta + (int64_t)b * a.dsb;
  const T* zg = a.z ? (const T*)a.z + (int64_t)b * a.zsb : nullptr;
  const T* gg = (const T*)q.dout + (int64_t)b * q.gsb;
  const int nT = (a.L + TL - 1) / TL;
  for (int ti = nT - 1; ti >= 0; ti--) {
    const int l0 = ti * TL;
    const int nl = (a.L - l0) < TL ? (a.L - l0) : TL;
    ss_load_tile<T, TL>(ug, a.usd, a.usl, d0, nd, l0, nl, su, DT);
    ss_load_tile<T, TL>(dg, a.dsd, a.dsl, d0, nd, l0, nl, sd, DT);
    ss_load_tile<T, TL>(gg, q.gsd, q.gsl, d0, nd, l0, nl, sg, DT);
    if (zg) ss_load_tile<T, TL>(zg, a.zsd, a.zsl, d0, nd, l0, nl, sz, DT);
    for (int i = c; i < TL * NR; i += DT) {   // every slot is written: the state loops run over all NR entries
      const int n = i / TL, t = i % TL;
      const bool ok = n < a.N && t < nl;
      sB[t * NR + n] = (a.Bvar && ok) ? load_rt(a.Bm, (int64_t)b * a.Bsb + (int64_t)g * a.Bsg + (int64_t)n * a.Bsn + (int64_t)(l0 + t) * a.Bsl, a.bdt) : 0.f;
      sC[t * NR + n] = (a.Cvar && ok) ? load_rt(a.Cm, (int64_t)b * a.Csb + (int64_t)g * a.Csg + (int64_t)n * a.Csn + (int64_t)(l0 + t) * a.Csl, a.cdt) : 0.f;
      sdB[t * NR + n] = 0.f;
      sdC[t * NR + n] = 0.f;
    }
    block_sync();
    // ---- rebuild x_t of this tile from the checkpoint
    float x0[NR], x[NR];
    {
      const float* cp = a.ckpt + (((int64_t)b * a.Dm + d) * a.nTB + ti) * a.N;
#pragma unroll
      for (int n = 0; n < NR; n++) { x0[n] = (n < a.N && live) ? cp[n] : 0.f; x[n] = x0[n]; }
    }
    for (int t = 0; t < nl; t++) {
      const float uu = to_f32(su[t * DT + c]);
      float dl = to_f32(sd[t * DT + c]) + db;
      if (a.softplus) dl = softplus_f(dl);
#pragma unroll
      for (int n = 0; n < NR; n++) {
        const float Bv = a.Bvar ? sB[t * NR + n] : Bc[n];
        x[n] = expf(dl * A[n]) * x[n] + dl * uu * Bv;
        sx[(t * NR + n) * DT + c] = x[n];
      }
    }
    // ---- adjoint sweep
    for (int t = nl - 1; t >= 0; t--) {
      const float uu = to_f32(su[t * DT + c]);
      const float draw = to_f32(sd[t * DT + c]) + db;
      const float dl = a.softplus ? softplus_f(draw) : draw;
      const float go = live ? to_f32(sg[t * DT + c]) : 0.f;
      float dy = go;
      if (zg) {
        const float zz = to_f32(sz[t * DT + c]);
        float ypre = Dv * uu;
#pragma unroll
        for (int n = 0; n < NR; n++) ypre += sx[(t * NR + n) * DT + c] * (a.Cvar ? sC[t * NR + n] : Cc[n]);
        const float sig = 1.f / (1.f + expf(-zz));
        sdz[t * DT + c] = from_f32<T>(go * ypre * sig * (1.f + zz * (1.f - sig)));
        dy = go * zz * sig;
      }
      dDacc += dy * uu;
      float duv = dy * Dv, ddl = 0.f;
#pragma unroll
      for (int n = 0; n < NR; n++) {
        const float Bv = a.Bvar ? sB[t * NR + n] : Bc[n];
        const float Cv = a.Cvar ? sC[t * NR + n] : Cc[n];
        const float xt = sx[(t * NR + n) * DT + c];
        const float xp = t > 0 ? sx[((t - 1) * NR + n) * DT + c] : x0[n];
        const float at = expf(dl * A[n]);
        const float gn = gx[n] + dy * Cv;            // adjoint of x_t
        const float dCn = dy * xt, dBn = gn * dl * uu;
        dAacc[n] += gn * xp * dl * at;
        ddl += gn * (xp * A[n] * at + Bv * uu);
        duv += gn * dl * Bv;
        gx[n] = at * gn;                             // carried to t - 1
        if (a.Bvar) { const float s = wave_sum(dBn); if (c == 0) sdB[t * NR + n] += s; } else dBc[n] += dBn;
        if (a.Cvar) { const float s = wave_sum(dCn); if (c == 0) sdC[t * NR + n] += s; } else dCc[n] += dCn;
      }
      const float ddraw = a.softplus ? ddl * (1.f / (1.f + expf(-draw))) : ddl;
      ddbacc += live ? ddraw : 0.f;
      sdu[t * DT + c] = from_f32<T>(duv);
      sdd[t * DT + c] = from_f32<T>(ddraw);
    }
    block_sync();
    ss_store_tile<T, TL>((T*)q.du + (int64_t)b * q.dusb, q.dusd, q.dusl, d0, nd, l0, nl, sdu, DT);
    ss_store_tile<T, TL>((T*)q.ddelta + (int64_t)b * q.ddsb, q.ddsd, q.ddsl, d0, nd, l0, nl, sdd, DT);
    if (zg && q.dz) ss_store_tile<T, TL>((T*)q.dz + (int64_t)b * q.dzsb, q.dzsd, q.dzsl, d0, nd, l0, nl, sdz, DT);
    for (int i = c; i < TL * a.N; i += DT) {
      const int n = i / TL, t = i % TL;
      if (t < nl) {
        if (a.Bvar) atomic_add_f32(q.dB + (int64_t)b * q.dBsb + (int64_t)g * q.dBsg + (int64_t)n * q.dBsn + (int64_t)(l0 + t) * q.dBsl, sdB[t * NR + n]);
        if (a.Cvar) atomic_add_f32(q.dC + (int64_t)b * q.dCsb + (int64_t)g * q.dCsg + (int64_t)n * q.dCsn + (int64_t)(l0 + t) * q.dCsl, sdC[t * NR + n]);
      }
    }
    block_sync();
  }
  if (live) {
#pragma unroll
    for (int n = 0; n < NR; n++) {
      if (n < a.N) {
        atomic_add_f32(q.dA + (int64_t)d * a.N + n, dAacc[n]);
        if (!a.Bvar) atomic_add_f32(q.dB + (int64_t)d * q.dBsg + (int64_t)n * q.dBsn, dBc[n]);
        if (!a.Cvar) atomic_add_f32(q.dC + (int64_t)d * q.dCsg + (int64_t)n * q.dCsn, dCc[n]);
      }
    }
    if (q.dD) atomic_add_f32(q.dD + d, dDacc);
    if (q.ddb) atomic_add_f32(q.ddb + d, ddbacc);
  }
}


template <class T>
__global__ __launch_bounds__(1024) void selscan_bwd_chunked_kernel(SsBwdArgs q) {
  const SsArgs& a = q.f;
  constexpr int LC = SSR_LC, TP = SSR_TP, VEC = 16 / sizeof(T), PPR = TP / VEC;
  constexpr float LN2 = 0.6931471805599453f;
  OMK_DYN_SMEM(smem);
  const int NB = q.NB, nOct = q.nOct;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, NW = blockDim.x >> 6;
  const int NS = nOct > 1 ? 16 : 64;                            // stride over n of the per-(tile, wave) arrays (nOct > 1 only with N <= 16)
  float* sAcc = (float*)smem;                                   // [2][NB][TP]: dB rows, dC rows of the pass
  T* sBC = (T*)(sAcc + (size_t)2 * NB * TP);                    // [2][NB][TP]: B rows, C rows (16-byte piece v of lane j at slot v * 64 + j)
  float* sA2 = (float*)(sBC + (size_t)2 * NB * TP);             // per (tile, wave, n): A log2 e
  float* sG = sA2 + nOct * NW * NS;                             //                      adjoint carry between passes
  float* sdA = sG + nOct * NW * NS;                             //                      dA sum
  float* sH = sdA + nOct * NW * NS;                             // per (wave, n): state in front of the pass
  const int tpb = a.Dm / (NW * nOct);
  const int b = blockIdx.x / tpb, dbase = (blockIdx.x % tpb) * NW * nOct;
  const int g = dbase / (a.Dm / a.G);
  const T* Bbase = (const T*)a.Bm + (int64_t)b * a.Bsb + (int64_t)g * a.Bsg;
  const T* Cbase = (const T*)a.Cm + (int64_t)b * a.Csb + (int64_t)g * a.Csg;
  float* dBbase = q.dB + (int64_t)b * q.dBsb + (int64_t)g * q.dBsg;
  float* dCbase = q.dC + (int64_t)b * q.dCsb + (int64_t)g * q.dCsg;
  const int nPass = (a.L + TP - 1) / TP;
  for (int oc = 0; oc < nOct; oc++) {
    const int d = dbase + oc * NW + wv, ix = (oc * NW + wv) * NS + lane;
    if (lane < NS) {
      sA2[ix] = lane < a.N ? ((const float*)a.A)[(int64_t)d * a.Asd + (int64_t)lane * a.Asn] * LOG2E : 0.f;
      sG[ix] = 0.f; sdA[ix] = 0.f;
    }
  }
  for (int i = threadIdx.x; i < 2 * NB * TP; i += blockDim.x) sAcc[i] = 0.f;
  float dDacc[4] = {0.f, 0.f, 0.f, 0.f}, ddbacc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int pass = nPass - 1; pass >= 0; pass--) {
    const int tile0 = pass * TP, t0 = tile0 + lane * LC;
    float dl[LC], dlu[LC], dy[LC], gB[LC], ddl[LC], ypre[LC];   // gB = sum_n g_t B_t[n]: du = delta gB + D dy, ddelta += u gB
    for (int nb0 = 0; nb0 < a.N; nb0 += NB) {
      const int nbn = a.N - nb0 < NB ? a.N - nb0 : NB;
      // ---- B / C rows of this block of state indices (the accumulators are zero: start of the kernel or the flush below)
      for (int i = threadIdx.x; i < 2 * nbn * PPR; i += blockDim.x) {
        const int r = i / PPR, pc = i % PPR, k = r / nbn, nl = r % nbn;
        const T* row = k == 0 ? Bbase + (int64_t)(nb0 + nl) * a.Bsn : Cbase + (int64_t)(nb0 + nl) * a.Csn;
        const int e0 = pc * VEC, tq = tile0 + e0;
        T* dst = sBC + ((size_t)k * NB + nl) * TP + ((e0 % LC) / VEC * 64 + e0 / LC) * VEC;
        const T* src = row + tq;
        if (tq + VEC <= a.L && (((uintptr_t)src) & 15) == 0) {
          *reinterpret_cast<vec_t<T, VEC>*>(dst) = *reinterpret_cast<const vec_t<T, VEC>*>(src);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; e++) dst[e] = tq + e < a.L ? src[e] : T{};
        }
      }
      block_sync();   // rows staged, accumulators zeroed
#pragma unroll 1
      for (int oc = 0; oc < nOct; oc++) {
        const int d = dbase + oc * NW + wv, ixb = (oc * NW + wv) * NS;
        const T* urow = (const T*)a.u + (int64_t)b * a.usb + (int64_t)d * a.usd;
        const T* drow = (const T*)a.delta + (int64_t)b * a.dsb + (int64_t)d * a.dsd;
        const T* zrow = a.z ? (const T*)a.z + (int64_t)b * a.zsb + (int64_t)d * a.zsd : nullptr;
        const T* grow = (const T*)q.dout + (int64_t)b * q.gsb + (int64_t)d * q.gsd;
        const float Dv = a.D ? load_rt(a.D, d, a.ddt) : 0.f;
        const float db = a.dbias ? load_rt(a.dbias, d, a.dbdt) : 0.f;
        if (nb0 == 0) {
          float u[LC];
          ssc_load_row<T, LC>(urow, t0, a.L, u);
          ssc_load_row<T, LC>(drow, t0, a.L, dl);
          ssc_load_row<T, LC>(grow, t0, a.L, dy);
          if (zrow) {
            float zv[LC];
            ssc_load_row<T, LC>(zrow, t0, a.L, zv);
#pragma unroll
            for (int i = 0; i < LC; i++) dy[i] *= zv[i] * rcp_fast(1.f + exp2_fast(-zv[i] * LOG2E));
          }
#pragma unroll
          for (int i = 0; i < LC; i++) {
            float v = dl[i] + db;
            if (a.softplus) v = v > 20.f ? v : LN2 * log2_fast(1.f + exp2_fast(v * LOG2E));
            dl[i] = (t0 + i < a.L) ? v : 0.f;     // tokens past the end: a = 1, b = 0, dy = 0 (identity pairs in both directions)
            dlu[i] = dl[i] * u[i];
            gB[i] = 0.f; ddl[i] = 0.f; ypre[i] = 0.f;
          }
          // the state in front of this pass, one n per lane
          if (lane < a.N) sH[wv * 64 + lane] = a.ckpt[(((int64_t)b * a.Dm + d) * a.nTB + pass) * a.N + lane];
        }
        float sdl = 0.f;
#pragma unroll
        for (int i = 0; i < LC; i++) sdl += dl[i];
        const T* pB = sBC + lane * VEC;
        const T* pC = pB + (size_t)NB * TP;
        float* aB = sAcc + lane * 4;
        float* aC = aB + (size_t)NB * TP;
        // dB / dC rows are sums over the channels (= waves) of the workgroup.  LDS float atomics run at a few cycles per LANE on this
        // chip (measured: 16 ds_add_f32 per wave and state index made the kernel 5x slower), so the waves walk the state indices in
        // ROTATED order -- at step k wave w owns row (k + w) mod R -- and stay in step through one barrier: no two waves ever touch
        // the same accumulator row at the same time and a plain 16-byte read-add-write does the job.
        const int R = nbn > NW ? nbn : NW;
        for (int k = 0; k < ((q.dbg & 4) ? 0 : R); k++) {
          int nl = k + wv;
          nl = nl >= R ? nl - R : nl;
          if (nl < nbn) {
            const int n = nb0 + nl;
            const float A2 = sA2[ixb + n], An = A2 * LN2;
            float Bv[LC], Cv[LC];
#pragma unroll
            for (int v = 0; v < LC / VEC; v++) {
              float wb[VEC], wc[VEC];
              load_vec<T, VEC>(pB + (size_t)nl * TP + v * 64 * VEC, wb);
              load_vec<T, VEC>(pC + (size_t)nl * TP + v * 64 * VEC, wc);
#pragma unroll
              for (int e = 0; e < VEC; e++) { Bv[v * VEC + e] = wb[e]; Cv[v * VEC + e] = wc[e]; }
            }
            // ---- forward: x_{t-1} of every token of the lane's chunk
            float av[LC], hp[LC];
            const float Pc = exp2_fast(sdl * A2);
            float P = Pc, X = 0.f;
#pragma unroll
            for (int i = 0; i < LC; i++) {
              av[i] = exp2_fast(dl[i] * A2);
              X = fmaf(av[i], X, dlu[i] * Bv[i]);
            }
            wave_scan_affine(P, X);
            const float cin = sH[wv * 64 + n];
            const float xend = fmaf(P, cin, X);
            float x = shfl_up(xend, 1);
            if (lane == 0) x = cin;
            {
              float* rc = aC + (size_t)nl * TP;
              f32x4 c0 = *(const f32x4*)rc, c1 = *(const f32x4*)(rc + 256);
#pragma unroll
              for (int i = 0; i < LC; i++) {
                hp[i] = x;
                x = fmaf(av[i], x, dlu[i] * Bv[i]);
                ypre[i] = fmaf(Cv[i], x, ypre[i]);
                if (i < 4) c0[i] = fmaf(dy[i], x, c0[i]); else c1[i - 4] = fmaf(dy[i], x, c1[i - 4]);
              }
              *(f32x4*)rc = c0; *(f32x4*)(rc + 256) = c1;
            }
            // ---- reverse: G_t = a_t (C_t dy_t + G_{t+1})
            float Pr = Pc, Xr = 0.f;
#pragma unroll
            for (int i = LC - 1; i >= 0; i--) Xr = av[i] * fmaf(Cv[i], dy[i], Xr);
            wave_scan_affine_rev(Pr, Xr);
            const float gin = sG[ixb + n];
            const float gout = fmaf(Pr, gin, Xr);          // G at the first token of this lane's chunk
            float Gn = shfl_down(gout, 1);
            if (lane == 63) Gn = gin;
            const float g0 = wave_read_lane(gout, 0);
            float dAl = 0.f;
            {
              float* rb = aB + (size_t)nl * TP;
              f32x4 b0 = *(const f32x4*)rb, b1 = *(const f32x4*)(rb + 256);
#pragma unroll
              for (int i = LC - 1; i >= 0; i--) {
                const float gt = fmaf(Cv[i], dy[i], Gn);      // adjoint of x_t
                const float tmp = gt * hp[i] * av[i];
                dAl = fmaf(tmp, dl[i], dAl);
                ddl[i] = fmaf(tmp, An, ddl[i]);
                gB[i] = fmaf(gt, Bv[i], gB[i]);
                if (i < 4) b0[i] = fmaf(gt, dlu[i], b0[i]); else b1[i - 4] = fmaf(gt, dlu[i], b1[i - 4]);
                Gn = av[i] * gt;
              }
              *(f32x4*)rb = b0; *(f32x4*)(rb + 256) = b1;
            }
            const float tot = wave_sum(dAl);
            if (lane == 0) { sG[ixb + n] = g0; sdA[ixb + n] += tot; }
          }
          block_sync();   // next step: every wave moves to the next row
        }
        if (nb0 + NB >= a.N) {
          // ---- per-token outputs of this channel: every state index is in
          // u and softplus'(raw delta) from what is still in registers instead of second reads of their rows (the counters show those
          // coming from HBM, + 0.4 GB at batch 64): u = (delta u) / delta, sigmoid(raw) = 1 - exp(-softplus(raw)).  Only a token
          // whose delta underflowed to exactly 0 has lost its u: then (rare, wave-uniform branch) the row is read again.
          float u[LC];
          bool lost = false;
#pragma unroll
          for (int i = 0; i < LC; i++) {
            u[i] = dl[i] != 0.f ? dlu[i] * rcp_fast(dl[i]) : 0.f;       // (tokens past the end: delta = 0, u = 0)
            lost = lost || (dl[i] == 0.f && t0 + i < a.L);
          }
          if (__builtin_expect(ballot_any(lost), 0)) ssc_load_row<T, LC>(urow, t0, a.L, u);
          if (zrow) {
            float zv[LC], go[LC];
            ssc_load_row<T, LC>(grow, t0, a.L, go);
            ssc_load_row<T, LC>(zrow, t0, a.L, zv);
#pragma unroll
            for (int i = 0; i < LC; i++) {
              const float sig = rcp_fast(1.f + exp2_fast(-zv[i] * LOG2E));
              go[i] = go[i] * fmaf(Dv, u[i], ypre[i]) * sig * (1.f + zv[i] * (1.f - sig));
            }
            ssc_store_row<T, LC>((T*)q.dz + (int64_t)b * q.dzsb + (int64_t)d * q.dzsd, t0, a.L, go);
          }
          float dDl = 0.f, ddbl = 0.f;
#pragma unroll
          for (int i = 0; i < LC; i++) {
            dDl = fmaf(dy[i], u[i], dDl);
            ddl[i] = fmaf(gB[i], u[i], ddl[i]);
            if (a.softplus) ddl[i] *= dl[i] < 1e-3f ? dl[i] * (1.f - 0.5f * dl[i]) : 1.f - exp2_fast(-dl[i] * LOG2E);   // sigmoid(raw) = 1 - exp(-softplus(raw))
            if (t0 + i >= a.L) ddl[i] = 0.f;
            ddbl += ddl[i];
            gB[i] = fmaf(gB[i], dl[i], Dv * dy[i]);   // du
          }
          ssc_store_row<T, LC>((T*)q.du + (int64_t)b * q.dusb + (int64_t)d * q.dusd, t0, a.L, gB);
          ssc_store_row<T, LC>((T*)q.ddelta + (int64_t)b * q.ddsb + (int64_t)d * q.ddsd, t0, a.L, ddl);
#pragma unroll
          for (int o = 0; o < 4; o++)
            if (o == oc) { dDacc[o] += dDl; ddbacc[o] += ddbl; }
        }
      }
      block_sync();   // every wave has added its rows
      // ---- the block's dB / dC rows -> HBM; the thread that reads a slot zeroes it for the next block
      for (int i = threadIdx.x; i < 2 * nbn * TP; i += blockDim.x) {
        const int r = i / TP, t = i % TP, k = r / nbn, nl = r % nbn;
        float* slot = sAcc + ((size_t)k * NB + nl) * TP + ((t % LC) / 4 * 64 + t / LC) * 4 + t % 4;
        const float v = *slot;
        *slot = 0.f;
        if (tile0 + t < a.L && !(q.dbg & 2)) {
          if (k == 0) atomic_add_f32(dBbase + (int64_t)(nb0 + nl) * q.dBsn + (int64_t)(tile0 + t) * q.dBsl, v);
          else atomic_add_f32(dCbase + (int64_t)(nb0 + nl) * q.dCsn + (int64_t)(tile0 + t) * q.dCsl, v);
        }
      }
      // (the next block's staging overwrites sBC: every read of it sits before the barrier above)
    }
  }
  // ---- per-channel sums
#pragma unroll 1
  for (int oc = 0; oc < nOct; oc++) {
    const int d = dbase + oc * NW + wv, ixb = (oc * NW + wv) * NS;
    if (lane < a.N) atomic_add_f32(q.dA + (int64_t)d * a.N + lane, sdA[ixb + lane]);
    float dDt = 0.f, ddbt = 0.f;
#pragma unroll
    for (int o = 0; o < 4; o++)
      if (o == oc) { dDt = dDacc[o]; ddbt = ddbacc[o]; }
    dDt = wave_sum(dDt); ddbt = wave_sum(ddbt);
    if (lane == 0) {
      if (q.dD) atomic_add_f32(q.dD + d, dDt);
      if (q.ddb) atomic_add_f32(q.ddb + d, ddbt);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------
// backward for MANY sequences on CHANNEL-LAST storage: lanes = channels, one reverse sweep (round 6; the forward's twin,
// selscan_fwd_lanes_kernel).  The module keeps (B, L, D) activations as the in_proj GEMM leaves them
// (/root/reference/models/stage2/mixer_seq_simple.py:16,197-205 builds the mixer; mamba_simple.py here); until now their backward made
// L-contiguous copies for the chunked scan (3.3 ms forward + backward at batch 64) or took the per-channel kernel above (33 ms).
//   one wave = 64 adjacent channels of one (batch, group); a lane owns its channel's d_state <= 16 adjoint states G, its dA row and the
//   running sums of dD / d(delta_bias), and walks the sequence backwards in tiles of 16 tokens (= the checkpoint spacing SSB_TL of pass 1,
//   which is the forward lanes kernel without output, checkpoints stored (B, tile, n, D): a lane reads ITS element of a 256-byte row);
//   a tile is two sub-tiles of 8 tokens: the forward recurrence is re-run from the checkpoint -- 8 tokens to reach the second sub-tile,
//   then each sub-tile once more with p_t = a_t x_{t-1} parked in wave-private LDS (16 float4 rows x 8 tokens x 64 lanes = 32 KB; four
//   workgroups of one wave per CU) -- and the adjoint step of a token reads p_t back: 1.5 forward steps + one adjoint step per token;
//   B_t / C_t are broadcasts out of a wave-private LDS tile like in the forward; dB_t / dC_t (sums over the channels of the group) leave
//   the wave through a 32-value butterfly (wave_sum32: lane-half swaps + DPP, no LDS) and one float atomic per value and token.
// ---------------------------------------------------------------------------------------------------------
constexpr int SBL_T = SSB_TL, SBL_S = 8;
static_assert(SBL_T == 2 * SBL_S && SBL_T == SCL_TB, "a tile = two sub-tiles = one block of the forward lanes kernel");

template <class T>
__global__ __launch_bounds__(64) void selscan_bwd_lanes_kernel(SsBwdArgs q) {
  const SsArgs& a = q.f;
  __shared__ __attribute__((aligned(16))) float sB[SBL_T * SCL_SB];
  __shared__ __attribute__((aligned(16))) float sC[SBL_T * SCL_SB];
  __shared__ __attribute__((aligned(16))) f32x4 sP[SBL_S * 4 * 64];
  __shared__ float sRed[SBL_S * 32];
  __shared__ float sDl[SBL_T * 64];   // delta (softplus applied) of the tile's tokens, [token][lane]
  constexpr uint32_t ES = sizeof(T);
  constexpr float LN2 = 0.6931471805599453f;
  const int lane = threadIdx.x;
  const int dpg = a.Dm / a.G, tpg = (dpg + 63) / 64;
  const int tg = blockIdx.x % tpg, g = (blockIdx.x / tpg) % a.G, b = blockIdx.x / (tpg * a.G);
  const int d0 = g * dpg + tg * 64;
  const int nd = (dpg - tg * 64) < 64 ? (dpg - tg * 64) : 64;
  const bool live = lane < nd;
  const int d = d0 + (live ? lane : 0);
  float A2[16], G[16], dAa[16];
#pragma unroll
  for (int n = 0; n < 16; n++) {
    A2[n] = n < a.N ? load_rt(a.A, (int64_t)d * a.Asd + (int64_t)n * a.Asn, a.adt) * LOG2E : 0.f;
    G[n] = 0.f; dAa[n] = 0.f;
  }
  const float Dv = a.D ? load_rt(a.D, d, a.ddt) : 0.f;
  const float db = a.dbias ? load_rt(a.dbias, d, a.dbdt) : 0.f;
  const bool hasz = a.z != nullptr;
  float dDa = 0.f, dba = 0.f;
  const BufRes Ur = make_buf((const T*)a.u + (int64_t)b * a.usb, a.xu), Dr = make_buf((const T*)a.delta + (int64_t)b * a.dsb, a.xd);
  const BufRes Zr = make_buf(hasz ? (const T*)a.z + (int64_t)b * a.zsb : nullptr, hasz ? a.xz : 0u);
  const BufRes Gr = make_buf((const T*)q.dout + (int64_t)b * q.gsb, q.xg);
  const BufRes DUr = make_buf((T*)q.du + (int64_t)b * q.dusb, q.xdu), DDr = make_buf((T*)q.ddelta + (int64_t)b * q.ddsb, q.xdd);
  const BufRes DZr = make_buf(hasz ? (T*)q.dz + (int64_t)b * q.dzsb : nullptr, hasz ? q.xdz : 0u);
  const BufRes Br = make_buf((const T*)a.Bm + (int64_t)b * a.Bsb + (int64_t)g * a.Bsg, a.xB);
  const BufRes Cr = make_buf((const T*)a.Cm + (int64_t)b * a.Csb + (int64_t)g * a.Csg, a.xC);
  const uint32_t lch = (uint32_t)(d0 + lane);
  const uint32_t uvo = live ? ES * lch * (uint32_t)a.usd : SCL_OOR, dvo = live ? ES * lch * (uint32_t)a.dsd : SCL_OOR;
  const uint32_t zvo = (live && hasz) ? ES * lch * (uint32_t)a.zsd : SCL_OOR, gvo = live ? ES * lch * (uint32_t)q.gsd : SCL_OOR;
  const uint32_t duvo = live ? ES * lch * (uint32_t)q.dusd : SCL_OOR, ddvo = live ? ES * lch * (uint32_t)q.ddsd : SCL_OOR;
  const uint32_t dzvo = (live && hasz) ? ES * lch * (uint32_t)q.dzsd : SCL_OOR;
  // B / C rows of a tile: 16 tokens x 16 states = four elements per lane and array (the forward's scheme)
  const int r4 = lane >> 4, t16 = lane & 15;
  const bool bl = a.Bsl == 1, cl = a.Csl == 1;
  const uint32_t bvo = ES * (bl ? (uint32_t)r4 * (uint32_t)a.Bsn + (uint32_t)t16 * (uint32_t)a.Bsl : (uint32_t)t16 * (uint32_t)a.Bsn + (uint32_t)r4 * (uint32_t)a.Bsl);
  const uint32_t cvo = ES * (cl ? (uint32_t)r4 * (uint32_t)a.Csn + (uint32_t)t16 * (uint32_t)a.Csl : (uint32_t)t16 * (uint32_t)a.Csn + (uint32_t)r4 * (uint32_t)a.Csl);
  const uint32_t bks = ES * 4u * (uint32_t)(bl ? a.Bsn : a.Bsl), cks = ES * 4u * (uint32_t)(cl ? a.Csn : a.Csl);
  const int blo = bl ? t16 * SCL_SB + r4 : r4 * SCL_SB + t16, bls = bl ? 4 : 4 * SCL_SB;
  const int clo = cl ? t16 * SCL_SB + r4 : r4 * SCL_SB + t16, cls = cl ? 4 : 4 * SCL_SB;
  float* dBb = q.dB + (int64_t)b * q.dBsb + (int64_t)g * q.dBsg;
  float* dCb = q.dC + (int64_t)b * q.dCsb + (int64_t)g * q.dCsg;
  const int nT = (a.L + SBL_T - 1) / SBL_T;

  const uint32_t usl_ = ES * (uint32_t)a.usl, dsl_ = ES * (uint32_t)a.dsl, zsl_ = hasz ? ES * (uint32_t)a.zsl : 0u, gsl_ = ES * (uint32_t)q.gsl;
  auto softplus_of = [&](uint32_t raw, int t) -> float {   // delta of token t (zero behind the end of the sequence: the identity step)
    float v = raw_to_f32<T>(raw) + db;
    if (a.softplus) v = v > 20.f ? v : LN2 * log2_fast(1.f + exp2_fast(v * LOG2E));
    return t < a.L ? v : 0.f;
  };
  // The token loops are REAL loops (one token per iteration, the 16 states unrolled inside): unrolled over a sub-tile the compiler kept the
  // rows and inputs of all eight tokens alive at once -- 512 registers + 378 spilled, 6.6 k cycles per token.  A token's inputs are
  // requested one iteration ahead (they hit L2: the tile's rows were touched by the sweep before).
  // (two tokens per iteration on two sets of input registers, each refilled right behind its use: with one set the loop-carried value is a
  // register copy at the back edge, and the copy a wait for the load just issued)
  // ---- forward steps of tokens l0 + k0 .. + 7 from state h; PARK: p_t = a_t x_{t-1} into LDS for the adjoint steps
  auto sweep = [&](float (&h)[16], int l0, int k0, auto park_c) {
    constexpr bool PARK = decltype(park_c)::value;
    auto step = [&](int k, uint32_t cu) OMK_ALWAYS_INLINE_LAMBDA {
      const float dlt = sDl[(k0 + k) * 64 + lane], du_ = dlt * raw_to_f32<T>(cu);
      const f32x4* rb = reinterpret_cast<const f32x4*>(&sB[(k0 + k) * SCL_SB]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 b4 = rb[j];
        f32x4 pv;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int n = 4 * j + e;
          const float pn = exp2_fast(dlt * A2[n]) * h[n];
          pv[e] = pn;
          h[n] = fmaf(du_, b4[e], pn);
        }
        if (PARK) sP[(k * 4 + j) * 64 + lane] = pv;
      }
    };
    const uint32_t t0 = (uint32_t)(l0 + k0);
    uint32_t ua = buf_ld_raw<T>(Ur, uvo, usl_ * t0), ub = buf_ld_raw<T>(Ur, uvo, usl_ * (t0 + 1u));
#pragma unroll 1
    for (int k = 0; k < SBL_S; k += 2) {
      step(k, ua);
      OMK_SCHED_FENCE();
      ua = buf_ld_raw<T>(Ur, uvo, usl_ * (t0 + (uint32_t)k + 2u));       // (behind the end of the batch element's range: zeros)
      step(k + 1, ub);
      OMK_SCHED_FENCE();
      ub = buf_ld_raw<T>(Ur, uvo, usl_ * (t0 + (uint32_t)k + 3u));
    }
  };
  // ---- adjoint steps of tokens l0 + k0 + 7 .. l0 + k0 (p_t of each from LDS); the dB / dC rows of the sub-tile collect in LDS and leave
  // as four atomic instructions behind the loop (one atomic per token inside it made every iteration wait for the atomic's round trip:
  // the load counter returns in order)
  auto adjoint = [&](int l0, int k0) {
    auto step = [&](int kk, uint32_t cu, uint32_t cz, uint32_t cg) OMK_ALWAYS_INLINE_LAMBDA {
      const int k = k0 + kk, t = l0 + k;
      const float go = raw_to_f32<T>(cg), dlt = sDl[k * 64 + lane], u_ = raw_to_f32<T>(cu);
      const float zv = raw_to_f32<T>(cz), sg = sigmoid_fast(zv);
      const float dy = hasz ? go * zv * sg : go;
      dDa = fmaf(dy, u_, dDa);
      float sbg = 0.f, sgap = 0.f, ych = 0.f;
      float red[32];
      const float dlu = dlt * u_;
      const f32x4* rb = reinterpret_cast<const f32x4*>(&sB[k * SCL_SB]);
      const f32x4* rc = reinterpret_cast<const f32x4*>(&sC[k * SCL_SB]);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const f32x4 pv = sP[(kk * 4 + j) * 64 + lane], b4 = rb[j], c4 = rc[j];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int n = 4 * j + e;
          const float gn = fmaf(c4[e], dy, G[n]);              // g_t
          sbg = fmaf(b4[e], gn, sbg);
          const float gp = gn * pv[e];
          sgap = fmaf(gp, A2[n], sgap);
          dAa[n] = fmaf(gp, dlt, dAa[n]);
          const float xn = fmaf(dlu, b4[e], pv[e]);             // x_t[n]
          ych = fmaf(c4[e], xn, ych);
          red[n] = gn * dlu;                                    // dB_t[n] of this channel
          red[16 + n] = dy * xn;                                // dC_t[n]
          G[n] = exp2_fast(dlt * A2[n]) * gn;                   // a_t g_t: what token t - 1 adds its C dy to
        }
      }
      if (hasz) buf_st_t<T>(DZr, go * fmaf(Dv, u_, ych) * sg * (1.f + zv * (1.f - sg)), dzvo, ES * (uint32_t)t * (uint32_t)q.dzsl);
      buf_st_t<T>(DUr, fmaf(dlt, sbg, Dv * dy), duvo, ES * (uint32_t)t * (uint32_t)q.dusl);
      float ddr = fmaf(sgap, LN2, u_ * sbg);
      if (a.softplus) ddr *= 1.f - exp2_fast(-dlt * LOG2E);     // softplus'(x) = sigmoid(x) = 1 - exp(-softplus(x))
      ddr = t < a.L ? ddr : 0.f;
      dba += ddr;
      buf_st_t<T>(DDr, ddr, ddvo, ES * (uint32_t)t * (uint32_t)q.ddsl);
      if (!live) {
#pragma unroll
        for (int i = 0; i < 32; i++) red[i] = 0.f;
      }
      wave_sum32(red);               // lane L: total of value L >> 1
      if ((lane & 1) == 0) sRed[kk * 32 + (lane >> 1)] = red[0];
    };
    const int tl = l0 + k0 + SBL_S - 1;
    auto ld3 = [&](int t, uint32_t& ru, uint32_t& rz, uint32_t& rg) OMK_ALWAYS_INLINE_LAMBDA {
      const uint32_t tt = (uint32_t)(t > 0 ? t : 0);
      ru = buf_ld_raw<T>(Ur, uvo, usl_ * tt); rz = buf_ld_raw<T>(Zr, zvo, zsl_ * tt); rg = buf_ld_raw<T>(Gr, gvo, gsl_ * tt);
    };
    uint32_t ua, za, ga, ub, zb, gb;
    ld3(tl, ua, za, ga);
    ld3(tl - 1, ub, zb, gb);
#pragma unroll 1
    for (int kk = SBL_S - 1; kk >= 0; kk -= 2) {
      step(kk, ua, za, ga);
      OMK_SCHED_FENCE();
      ld3(l0 + k0 + kk - 2, ua, za, ga);
      step(kk - 1, ub, zb, gb);
      OMK_SCHED_FENCE();
      ld3(l0 + k0 + kk - 3, ub, zb, gb);
    }
    wave_lds_sync();
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int idx = lane + 64 * r, kk = idx >> 5, j = idx & 31, n = j & 15, t = l0 + k0 + kk;
      if (t < a.L && n < a.N) {
        const float v = sRed[idx];
        if (j < 16) atomic_add_f32(dBb + (int64_t)n * q.dBsn + (int64_t)t * q.dBsl, v);
        else atomic_add_f32(dCb + (int64_t)n * q.dCsn + (int64_t)t * q.dCsl, v);
      }
    }
  };

  for (int ti = nT - 1; ti >= 0; ti--) {
    const int l0 = ti * SBL_T;
    // ---- B / C rows of the tile into the wave-private LDS tile, the checkpoint in front of the tile
    uint32_t pb[4], pc[4];
    {
      const uint32_t sb0 = ES * (uint32_t)l0 * (uint32_t)a.Bsl, sc0 = ES * (uint32_t)l0 * (uint32_t)a.Csl;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        pb[k] = buf_ld_raw<T>(Br, bvo, sb0 + (uint32_t)k * bks);
        pc[k] = buf_ld_raw<T>(Cr, cvo, sc0 + (uint32_t)k * cks);
      }
    }
    float hs[16], h[16];
    {
      const float* cp = a.ckpt + (((int64_t)b * a.nTB + ti) * a.N) * a.Dm + d;
#pragma unroll
      for (int n = 0; n < 16; n++) hs[n] = (n < a.N && live) ? cp[(int64_t)n * a.Dm] : 0.f;
    }
    // delta of the tile's 16 tokens: sixteen loads in flight, sixteen independent softplus chains, parked in LDS -- inside a token step the
    // chain (load -> exp -> log -> sixteen exps) was 300 - 400 cycles of latency nothing else in a one-wave-per-SIMD kernel covers
    uint32_t rdl[SBL_T];
#pragma unroll
    for (int k = 0; k < SBL_T; k++) rdl[k] = buf_ld_raw<T>(Dr, dvo, dsl_ * (uint32_t)(l0 + k));
    wave_lds_sync();   // the previous tile's broadcast reads before the new rows
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool okb = (bl ? r4 + 4 * k : t16) < a.N, okc = (cl ? r4 + 4 * k : t16) < a.N;
      sB[blo + k * bls] = okb ? raw_to_f32<T>(pb[k]) : 0.f;
      sC[clo + k * cls] = okc ? raw_to_f32<T>(pc[k]) : 0.f;
    }
#pragma unroll
    for (int k = 0; k < SBL_T; k++) sDl[k * 64 + lane] = softplus_of(rdl[k], l0 + k);
    wave_lds_sync();
    // second sub-tile first: the first one's forward steps lead to its start state
#pragma unroll
    for (int n = 0; n < 16; n++) h[n] = hs[n];
    sweep(h, l0, 0, std::false_type{});
    sweep(h, l0, SBL_S, std::true_type{});
    wave_lds_sync();
    adjoint(l0, SBL_S);
    wave_lds_sync();
#pragma unroll
    for (int n = 0; n < 16; n++) h[n] = hs[n];
    sweep(h, l0, 0, std::true_type{});
    wave_lds_sync();
    adjoint(l0, 0);
  }
  if (live) {
#pragma unroll
    for (int n = 0; n < 16; n++)
      if (n < a.N) atomic_add_f32(q.dA + (int64_t)d * a.N + n, dAa[n]);
    if (q.dD) atomic_add_f32(q.dD + d, dDa);
    if (q.ddb) atomic_add_f32(q.ddb + d, dba);
  }
}

}  // namespace omk

using namespace omk;

// shared argument marshalling of forward and backward; `who` names the entry point in error messages
static int ss_fill(SsArgs& a, const OmkTensor& u, const OmkTensor& delta, const OmkTensor& A, const OmkTensor& Bm, const OmkTensor& Cm,
                   const OmkTensor& D, const OmkTensor& z, const OmkTensor& dbias, int softplus, const char* who) {
  OMK_REQUIRE(present(u) && present(delta) && present(A) && present(Bm) && present(Cm), "%s: u, delta, A, B, C required", who);
  OMK_REQUIRE(u.ndim == 3 && delta.ndim == 3 && A.ndim == 2, "%s: u/delta (B, D, L), A (D, N)", who);
  a.B = (int)u.shape[0]; a.Dm = (int)u.shape[1]; a.L = (int)u.shape[2]; a.N = (int)A.shape[1];
  OMK_REQUIRE(A.shape[0] == a.Dm, "%s: A must be (D, N)", who);
  OMK_REQUIRE(delta.dtype == u.dtype && (!present(z) || z.dtype == u.dtype), "%s: delta, z must have u's dtype", who);
  a.Bvar = Bm.ndim == 4; a.Cvar = Cm.ndim == 4;
  OMK_REQUIRE((a.Bvar || Bm.ndim == 2) && (a.Cvar || Cm.ndim == 2), "%s: B/C must be (B, G, N, L) or (D, N)", who);
  a.G = 1;
  if (a.Bvar) a.G = (int)Bm.shape[1];
  if (a.Cvar) { OMK_REQUIRE(!a.Bvar || Cm.shape[1] == a.G, "%s: B and C group counts differ", who); a.G = (int)Cm.shape[1]; }
  OMK_REQUIRE(a.G > 0 && a.Dm % a.G == 0, "%s: D must be a multiple of ngroups", who);
  OMK_REQUIRE(a.N <= 64, "%s: d_state > 64 is not supported by the Mamba-1 kernel", who);
  a.u = u.data; a.delta = delta.data; a.A = A.data; a.Bm = Bm.data; a.Cm = Cm.data; a.D = D.data; a.z = z.data; a.dbias = dbias.data;
  a.usb = u.stride[0]; a.usd = u.stride[1]; a.usl = u.stride[2];
  a.dsb = delta.stride[0]; a.dsd = delta.stride[1]; a.dsl = delta.stride[2];
  if (present(z)) { a.zsb = z.stride[0]; a.zsd = z.stride[1]; a.zsl = z.stride[2]; }
  a.Asd = A.stride[0]; a.Asn = A.stride[1];
  if (a.Bvar) { a.Bsb = Bm.stride[0]; a.Bsg = Bm.stride[1]; a.Bsn = Bm.stride[2]; a.Bsl = Bm.stride[3]; }
  else { a.Bsg = Bm.stride[0]; a.Bsn = Bm.stride[1]; }
  if (a.Cvar) { a.Csb = Cm.stride[0]; a.Csg = Cm.stride[1]; a.Csn = Cm.stride[2]; a.Csl = Cm.stride[3]; }
  else { a.Csg = Cm.stride[0]; a.Csn = Cm.stride[1]; }
  a.softplus = softplus; a.adt = A.dtype; a.bdt = Bm.dtype; a.cdt = Cm.dtype; a.ddt = D.dtype; a.dbdt = dbias.dtype;
  return OMK_OK;
}

// OmkSelScan{Fwd,Bwd}::pass_states comes in two forms: (B, D, ceil(L / 512), N) -- the state in front of every 512-token pass, for the chunked
// backward -- and (B, ceil(L / 16), N, D) -- in front of every 16-token tile with the channels innermost, for the lanes = channels backward
static bool ss_tile_states(const OmkTensor& t, int B, int Dm, int L, int N) {
  if (!present(t) || t.ndim != 4 || t.dtype != OMK_F32 || !is_dense(t)) return false;
  const int nT = (L + SBL_T - 1) / SBL_T, nP = (L + SSR_TP - 1) / SSR_TP;
  if (t.shape[0] == B && t.shape[1] == Dm && t.shape[2] == nP && t.shape[3] == N) return false;   // the chunked form
  return t.shape[0] == B && t.shape[1] == nT && t.shape[2] == N && t.shape[3] == Dm;
}

// The lanes = channels sweep (selscan_fwd_lanes_kernel) takes the call when it applies and the batch fills the chip by itself:
// returns 0 (no), 2 (channel-last storage) or 3 (L-contiguous storage) and fills the buffer ranges.
static int ss_lanes_form(SsArgs& a, int udt) {
  if (!a.out || !a.Bvar || !a.Cvar || a.N > 16 || a.bdt != udt || a.cdt != udt) return 0;
  if (a.ckpt && (a.TLB % SCL_TB) != 0) return 0;
  const bool z = a.z != nullptr;
  const bool cl = a.usd == 1 && a.dsd == 1 && (!z || a.zsd == 1) && a.osd == 1;
  const bool lc = a.usl == 1 && a.dsl == 1 && (!z || a.zsl == 1) && a.osl == 1;
  if (!cl && !lc) return 0;
  const int64_t es = (int64_t)dtype_size(udt), lim = (int64_t)1 << 31;
  auto span = [&](int64_t sd, int64_t sl, int64_t nd) -> int64_t { return (sd < 0 || sl < 0) ? lim : es * ((nd + 64) * sd + ((int64_t)a.L + 64) * sl); };
  const int64_t su = span(a.usd, a.usl, a.Dm), sdl = span(a.dsd, a.dsl, a.Dm), sz = z ? span(a.zsd, a.zsl, a.Dm) : 0, so = span(a.osd, a.osl, a.Dm);
  const int64_t sb = span(a.Bsn, a.Bsl, 16), sc = span(a.Csn, a.Csl, 16);
  if (su >= lim || sdl >= lim || sz >= lim || so >= lim || sb >= lim || sc >= lim) return 0;
  const char* e = getenv("OMK_SELSCAN_LANES");
  if (e && atoi(e) == 0) return 0;
  const int dpg = a.Dm / a.G;
  const int64_t nw = (int64_t)a.B * a.G * ((dpg + 63) / 64);
  // few sequences: time has to be cut (the chunked scan).  Measured crossovers at L 1024, D 768 (tools/bench_selscan.py): channel-last
  // storage from ~200 waves (the alternative pays L-contiguous copies), L-contiguous storage from one wave per SIMD
  if (!(e && atoi(e) == 1) && nw < (cl ? 192 : 1024) && a.L >= 64) return 0;
  auto ext = [&](int64_t sd, int64_t sl, int64_t nd) -> uint32_t { return (uint32_t)(es * ((nd - 1) * sd + ((int64_t)a.L - 1) * sl + 1)); };
  a.xu = ext(a.usd, a.usl, a.Dm); a.xd = ext(a.dsd, a.dsl, a.Dm); a.xz = z ? ext(a.zsd, a.zsl, a.Dm) : 0u; a.xo = ext(a.osd, a.osl, a.Dm);
  a.xB = ext(a.Bsn, a.Bsl, a.N); a.xC = ext(a.Csn, a.Csl, a.N);
  return cl ? 2 : 3;
}

// pass_ckpt: first pass of the chunked backward (a.ckpt = state in front of every 512-token pass, no output)
static int ss_launch_fwd(SsArgs& a, int udt, omk_stream stream, bool pass_ckpt = false) {
  if (const int form = pass_ckpt ? 0 : ss_lanes_form(a, udt)) {
    const int dpg = a.Dm / a.G;
    dim3 grid((unsigned)((int64_t)a.B * a.G * ((dpg + 63) / 64))), block(64);
    if (form == 2) OMK_DISPATCH_DTYPE(udt, T, OMK_LAUNCH((selscan_fwd_lanes_kernel<T, false>), grid, block, 0, stream, a));
    else OMK_DISPATCH_DTYPE(udt, T, OMK_LAUNCH((selscan_fwd_lanes_kernel<T, true>), grid, block, 0, stream, a));
    return OMK_OK;
  }
  // L-contiguous storage (upstream's layout): the chunked associative scan, one wave per (batch, channel)
  const bool lcontig = pass_ckpt || (a.usl == 1 && a.dsl == 1 && (!a.z || a.zsl == 1) && a.out && a.osl == 1 && (!a.Bvar || a.Bsl == 1) &&
                                     (!a.Cvar || a.Csl == 1) && (!a.ckpt || a.TLB == SSR_TP) && a.L >= 64 && !getenv("OMK_SELSCAN_SEQ"));
  if (lcontig) {
    const int64_t nseq = (int64_t)a.B * a.Dm;
    const char* lce = getenv("OMK_SELSCAN_LC");
    const bool lc16 = (pass_ckpt || a.ckpt) ? false : lce ? atoi(lce) == 16 : a.L >= 1024 && nseq < 4096;   // pass states: 512-token passes   // few sequences: fewer, longer passes; many: occupancy
    const int lc = lc16 ? 16 : 8;
    // shared B / C rows: 8 adjacent channels of one group per workgroup, rows of u's dtype, <= 64 KB of LDS
    const size_t es = dtype_size(udt);
    const size_t bc_bytes = (size_t)((a.Bvar ? a.N : 0) + (a.Cvar && !pass_ckpt ? a.N : 0)) * 64 * lc * es;
    const bool share = a.Bvar && a.Cvar && (a.Dm / a.G) % 8 == 0 && a.bdt == udt && a.cdt == udt && a.adt == OMK_F32 && a.N <= 64 &&
                       bc_bytes <= 64 * 1024 && !getenv("OMK_SELSCAN_NOSHARE");
    if (share) {
      dim3 grid((unsigned)(nseq / 8)), block(512);
#define SSC_SH(T, LC_, NU_) do { \
        if (OMK_SET_MAX_DYN_SMEM((selscan_fwd_shared_kernel<T, LC_, NU_>), bc_bytes)) return fail(OMK_ELAUNCH, "selective_scan_fwd: cannot raise dynamic LDS to %zu", bc_bytes); \
        OMK_LAUNCH((selscan_fwd_shared_kernel<T, LC_, NU_>), grid, block, bc_bytes, stream, a); } while (0)
      const char* nue = getenv("OMK_SELSCAN_NU");
      const bool nu2 = nue ? atoi(nue) == 2 : true;   // packed token pairs in the n loop (OMK_SELSCAN_NU=1: the scalar form)
#define SSC_ST(T) do { \
        if (OMK_SET_MAX_DYN_SMEM((selscan_fwd_shared_kernel<T, 8, 1, true>), bc_bytes)) return fail(OMK_ELAUNCH, "selective_scan_bwd: cannot raise dynamic LDS to %zu", bc_bytes); \
        OMK_LAUNCH((selscan_fwd_shared_kernel<T, 8, 1, true>), grid, block, bc_bytes, stream, a); } while (0)
      if (pass_ckpt) OMK_DISPATCH_DTYPE(udt, T, SSC_ST(T));
      else if (lc16) OMK_DISPATCH_DTYPE(udt, T, SSC_SH(T, 16, 1));
      else if (nu2) OMK_DISPATCH_DTYPE(udt, T, SSC_SH(T, 8, 2));
      else OMK_DISPATCH_DTYPE(udt, T, SSC_SH(T, 8, 1));
#undef SSC_SH
#undef SSC_ST
    } else {
      dim3 grid((unsigned)((nseq + 3) / 4)), block(256);
      if (lc16) OMK_DISPATCH_DTYPE(udt, T, OMK_LAUNCH((selscan_fwd_chunked_kernel<T, 16>), grid, block, 0, stream, a));
      else OMK_DISPATCH_DTYPE(udt, T, OMK_LAUNCH((selscan_fwd_chunked_kernel<T, 8>), grid, block, 0, stream, a));
    }
    return OMK_OK;
  }
  const int dpg = a.Dm / a.G;
  a.DT = dpg >= 128 ? 128 : ((dpg + 63) / 64) * 64;
  const int tiles_per_group = (dpg + a.DT - 1) / a.DT;
  dim3 grid((unsigned)((int64_t)a.B * a.G * tiles_per_group)), block(a.DT);
  const size_t smem = (size_t)4 * SS_TL * a.DT * dtype_size(udt) + (size_t)2 * SS_TL * a.N * 4;
#define SS_GO(T, NREG) OMK_LAUNCH((selscan_fwd_kernel<T, NREG>), grid, block, smem, stream, a)
  OMK_DISPATCH_DTYPE(udt, T, { if (a.N <= 16) SS_GO(T, 16); else SS_GO(T, 64); });
#undef SS_GO
  return OMK_OK;
}

extern "C" int omk_selective_scan_fwd(const OmkSelScanFwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->out), "selective_scan_fwd: out required");
  SsArgs a = {};
  int rc = ss_fill(a, p->u, p->delta, p->A, p->Bm, p->Cm, p->D, p->z, p->delta_bias, p->delta_softplus, "selective_scan_fwd");
  if (rc) return rc;
  OMK_REQUIRE(p->out.ndim == 3 && p->out.dtype == p->u.dtype, "selective_scan_fwd: out must be (B, D, L) of u's dtype");
  OMK_REQUIRE(!present(p->last_state) || p->last_state.dtype == OMK_F32, "selective_scan_fwd: last_state must be f32");
  a.out = p->out.data; a.last = (float*)p->last_state.data;
  a.osb = p->out.stride[0]; a.osd = p->out.stride[1]; a.osl = p->out.stride[2];
  if ((int64_t)a.B * a.Dm * a.L == 0) return OMK_OK;
  if (ss_tile_states(p->pass_states, a.B, a.Dm, a.L, a.N)) {
    a.ckpt = (float*)p->pass_states.data; a.TLB = SBL_T; a.nTB = (a.L + SBL_T - 1) / SBL_T; a.ckpt_cl = 1;
    if (ss_lanes_form(a, p->u.dtype) != 2) return fail(OMK_EUNSUPPORTED, "selective_scan_fwd: tile states (B, ceil(L / 16), N, D) belong to the lanes = channels form (channel-last views, d_state <= 16)");
  } else if (present(p->pass_states)) {
    const int nP = (a.L + SSR_TP - 1) / SSR_TP;
    OMK_REQUIRE(p->pass_states.dtype == OMK_F32 && is_dense(p->pass_states) && numel(p->pass_states) == (int64_t)a.B * a.Dm * nP * a.N,
                "selective_scan_fwd: pass_states must be contiguous f32 (B, D, ceil(L / 512), N)");
    a.ckpt = (float*)p->pass_states.data; a.TLB = SSR_TP; a.nTB = nP;
    if (!ss_lanes_form(a, p->u.dtype) &&
        (!(a.usl == 1 && a.dsl == 1 && (!a.z || a.zsl == 1) && a.osl == 1 && (!a.Bvar || a.Bsl == 1) && (!a.Cvar || a.Csl == 1) && a.L >= 64) ||
         getenv("OMK_SELSCAN_SEQ")))
      return fail(OMK_EUNSUPPORTED, "selective_scan_fwd: pass_states need L-contiguous u / delta / z / out / B / C and L >= 64");
  }
  if ((rc = ss_launch_fwd(a, p->u.dtype, stream))) return rc;
  return finish_launch("selective_scan_fwd");
}

extern "C" int omk_selective_scan_fwd_form(const OmkSelScanFwd* p) {
  if (!p || !present(p->out)) return fail(OMK_EINVAL, "selective_scan_fwd_form: out required");
  SsArgs a = {};
  int rc = ss_fill(a, p->u, p->delta, p->A, p->Bm, p->Cm, p->D, p->z, p->delta_bias, p->delta_softplus, "selective_scan_fwd_form");
  if (rc) return rc;
  a.out = p->out.data; a.osb = p->out.stride[0]; a.osd = p->out.stride[1]; a.osl = p->out.stride[2];
  if (present(p->pass_states)) { a.ckpt = (float*)p->pass_states.data; a.TLB = SSR_TP; }
  if (ss_lanes_form(a, p->u.dtype)) return 2;
  const bool lcontig = a.usl == 1 && a.dsl == 1 && (!a.z || a.zsl == 1) && a.osl == 1 && (!a.Bvar || a.Bsl == 1) && (!a.Cvar || a.Csl == 1) &&
                       a.L >= 64 && !getenv("OMK_SELSCAN_SEQ");
  return lcontig ? 1 : 0;
}

// channel-last views of everything, input-dependent B / C of u's dtype, d_state <= 16, and as many waves as the forward's lanes form asks
// for: fills the buffer ranges of both kernels.  q.f and the gradient strides must be filled.
static bool ss_bwd_lanes_applies(SsBwdArgs& q, const OmkSelScanBwd* p, int udt) {
  SsArgs& a = q.f;
  if ((present(p->pass_states) && !ss_tile_states(p->pass_states, a.B, a.Dm, a.L, a.N)) || !a.Bvar || !a.Cvar || a.N > 16 || a.bdt != udt || a.cdt != udt) return false;
  const bool z = a.z != nullptr;
  if (!(a.usd == 1 && a.dsd == 1 && (!z || (a.zsd == 1 && q.dzsd == 1)) && q.gsd == 1 && q.dusd == 1 && q.ddsd == 1)) return false;
  if (a.Dm == 1) return false;   // (a single channel is both layouts: the chunked form takes it)
  SsArgs probe = a;              // the forward's criteria (spans, number of waves, OMK_SELSCAN_LANES) on an output laid out like u
  probe.out = const_cast<void*>(a.u); probe.osb = a.usb; probe.osd = a.usd; probe.osl = a.usl; probe.ckpt = nullptr;
  if (ss_lanes_form(probe, udt) != 2) return false;
  a.xu = probe.xu; a.xd = probe.xd; a.xz = probe.xz; a.xB = probe.xB; a.xC = probe.xC; a.xo = 0u;
  const int64_t es = (int64_t)dtype_size(udt), lim = (int64_t)1 << 31;
  auto ext = [&](int64_t sd, int64_t sl) -> int64_t { return (sd < 0 || sl < 0) ? lim : es * (((int64_t)a.Dm - 1) * sd + ((int64_t)a.L - 1) * sl + 1); };
  const int64_t eg = ext(q.gsd, q.gsl), eu = ext(q.dusd, q.dusl), ed = ext(q.ddsd, q.ddsl), ez = z ? ext(q.dzsd, q.dzsl) : 0;
  if (eg >= lim || eu >= lim || ed >= lim || ez >= lim) return false;
  q.xg = (uint32_t)eg; q.xdu = (uint32_t)eu; q.xdd = (uint32_t)ed; q.xdz = (uint32_t)ez;
  return true;
}

// 2: omk_selective_scan_bwd would run the lanes = channels reverse sweep on these (channel-last) views as they lie; 0 / 1: it wants the
// L-contiguous rows of the chunked form (1) or falls to the per-channel kernel (0) -- the host mirror decides about copies with it
extern "C" int omk_selective_scan_bwd_form(const OmkSelScanBwd* p) {
  if (!p || !present(p->dout) || !present(p->du) || !present(p->ddelta)) return fail(OMK_EINVAL, "selective_scan_bwd_form: dout, du, ddelta required");
  SsBwdArgs q = {};
  int rc = ss_fill(q.f, p->u, p->delta, p->A, p->Bm, p->Cm, p->D, p->z, p->delta_bias, p->delta_softplus, "selective_scan_bwd_form");
  if (rc) return rc;
  if (p->dout.ndim != 3 || p->du.ndim != 3 || p->ddelta.ndim != 3) return 0;
  q.gsd = p->dout.stride[1]; q.gsl = p->dout.stride[2]; q.dusd = p->du.stride[1]; q.dusl = p->du.stride[2];
  q.ddsd = p->ddelta.stride[1]; q.ddsl = p->ddelta.stride[2];
  if (present(p->dz)) { q.dzsd = p->dz.stride[1]; q.dzsl = p->dz.stride[2]; }
  if (ss_bwd_lanes_applies(q, p, p->u.dtype)) return 2;
  const SsArgs& a = q.f;
  const bool lc = a.Bvar && a.Cvar && a.usl == 1 && a.dsl == 1 && (!a.z || a.zsl == 1) && q.gsl == 1 && a.Bsl == 1 && a.Csl == 1 && a.L >= 64;
  return lc ? 1 : 0;
}

extern "C" size_t omk_selective_scan_bwd_workspace_bytes(const OmkSelScanBwd* p) {
  if (!p || p->u.ndim != 3 || p->A.ndim != 2) return 0;
  const int64_t nTB = (p->u.shape[2] + SSB_TL - 1) / SSB_TL;
  return (size_t)(p->u.shape[0] * p->u.shape[1] * nTB * p->A.shape[1]) * 4;   // state checkpoints (B, D, nTB, N) f32
}

extern "C" int omk_selective_scan_bwd(const OmkSelScanBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->dout) && present(p->du) && present(p->ddelta) && present(p->dA) && present(p->dB) && present(p->dC),
              "selective_scan_bwd: dout, du, ddelta, dA, dB, dC required");
  SsBwdArgs q = {};
  SsArgs& a = q.f;
  int rc = ss_fill(a, p->u, p->delta, p->A, p->Bm, p->Cm, p->D, p->z, p->delta_bias, p->delta_softplus, "selective_scan_bwd");
  if (rc) return rc;
  const int udt = p->u.dtype;
  OMK_REQUIRE(p->dout.ndim == 3 && p->du.ndim == 3 && p->ddelta.ndim == 3, "selective_scan_bwd: dout, du, ddelta must be (B, D, L)");
  OMK_REQUIRE(p->dout.dtype == udt && p->du.dtype == udt && p->ddelta.dtype == udt && (!present(p->dz) || p->dz.dtype == udt),
              "selective_scan_bwd: dout, du, ddelta, dz must have u's dtype");
  OMK_REQUIRE(p->dA.dtype == OMK_F32 && is_contig_last(p->dA) && p->dA.ndim == 2 && p->dA.stride[0] == a.N, "selective_scan_bwd: dA must be contiguous f32 (D, N)");
  OMK_REQUIRE(p->dB.dtype == OMK_F32 && p->dC.dtype == OMK_F32, "selective_scan_bwd: dB, dC must be f32 (accumulated into: caller zeroes)");
  OMK_REQUIRE(p->dB.ndim == (a.Bvar ? 4 : 2) && p->dC.ndim == (a.Cvar ? 4 : 2), "selective_scan_bwd: dB / dC must have the rank of B / C");
  OMK_REQUIRE(!present(p->dD) || (p->dD.dtype == OMK_F32 && is_contig_last(p->dD)), "selective_scan_bwd: dD must be contiguous f32 (D)");
  OMK_REQUIRE(!present(p->ddelta_bias) || (p->ddelta_bias.dtype == OMK_F32 && is_contig_last(p->ddelta_bias)), "selective_scan_bwd: ddelta_bias must be contiguous f32 (D)");
  OMK_REQUIRE(!present(p->z) || present(p->dz), "selective_scan_bwd: dz required when z is given");
  OMK_REQUIRE(p->workspace && p->workspace_bytes >= omk_selective_scan_bwd_workspace_bytes(p), "selective_scan_bwd: workspace too small");
  if ((int64_t)a.B * a.Dm * a.L == 0) return OMK_OK;
  q.dout = p->dout.data; q.gsb = p->dout.stride[0]; q.gsd = p->dout.stride[1]; q.gsl = p->dout.stride[2];
  q.du = p->du.data; q.dusb = p->du.stride[0]; q.dusd = p->du.stride[1]; q.dusl = p->du.stride[2];
  q.ddelta = p->ddelta.data; q.ddsb = p->ddelta.stride[0]; q.ddsd = p->ddelta.stride[1]; q.ddsl = p->ddelta.stride[2];
  if (present(p->dz)) { q.dz = p->dz.data; q.dzsb = p->dz.stride[0]; q.dzsd = p->dz.stride[1]; q.dzsl = p->dz.stride[2]; }
  q.dA = (float*)p->dA.data; q.dB = (float*)p->dB.data; q.dC = (float*)p->dC.data; q.dD = (float*)p->dD.data; q.ddb = (float*)p->ddelta_bias.data;
  if (a.Bvar) { q.dBsb = p->dB.stride[0]; q.dBsg = p->dB.stride[1]; q.dBsn = p->dB.stride[2]; q.dBsl = p->dB.stride[3]; }
  else { q.dBsg = p->dB.stride[0]; q.dBsn = p->dB.stride[1]; }
  if (a.Cvar) { q.dCsb = p->dC.stride[0]; q.dCsg = p->dC.stride[1]; q.dCsn = p->dC.stride[2]; q.dCsl = p->dC.stride[3]; }
  else { q.dCsg = p->dC.stride[0]; q.dCsn = p->dC.stride[1]; }
  const int dpg = a.Dm / a.G;
  // ---- channel-last storage with enough sequences to fill the chip: the lanes = channels reverse sweep (selscan_bwd_lanes_kernel)
  if (ss_bwd_lanes_applies(q, p, udt)) {
    a.ckpt = (float*)p->workspace; a.TLB = SBL_T; a.nTB = (a.L + SBL_T - 1) / SBL_T; a.ckpt_cl = 1;
    dim3 grid((unsigned)((int64_t)a.B * a.G * ((dpg + 63) / 64))), block(64);
    if (present(p->pass_states)) a.ckpt = (float*)p->pass_states.data;   // (the training forward left them: no pass 1)
    else {   // pass 1: the forward sweep once more, no output, the state in front of every 16-token tile as (B, tile, n, D)
      SsArgs f = a;
      f.out = nullptr; f.xo = 0u; f.last = nullptr; f.z = nullptr; f.xz = 0u; f.D = nullptr;
      OMK_DISPATCH_DTYPE(udt, T, OMK_LAUNCH((selscan_fwd_lanes_kernel<T, false>), grid, block, 0, stream, f));
    }
    OMK_DISPATCH_DTYPE(udt, T, OMK_LAUNCH((selscan_bwd_lanes_kernel<T>), grid, block, 0, stream, q));
    return finish_launch("selective_scan_bwd");
  }
  // ---- L-contiguous storage, input-dependent B and C of u's dtype: the chunked associative scan in both directions
  const bool chunked = a.Bvar && a.Cvar && a.usl == 1 && a.dsl == 1 && (!a.z || (a.zsl == 1 && q.dzsl == 1)) && q.gsl == 1 && q.dusl == 1 &&
                       q.ddsl == 1 && a.Bsl == 1 && a.Csl == 1 && a.bdt == udt && a.cdt == udt && a.adt == OMK_F32 && dpg % 8 == 0 &&
                       a.L >= 64 && !getenv("OMK_SELSCAN_SEQ");
  OMK_REQUIRE(!present(p->pass_states) || chunked, "selective_scan_bwd: pass_states belong to the chunked form (L-contiguous rows, variable B / C, L >= 64) or, as tile states, to the lanes = channels form");
  if (chunked) {
    a.TLB = SSR_TP; a.nTB = (a.L + SSR_TP - 1) / SSR_TP;
    if (present(p->pass_states)) {
      OMK_REQUIRE(p->pass_states.dtype == OMK_F32 && is_dense(p->pass_states) && numel(p->pass_states) == (int64_t)a.B * a.Dm * a.nTB * a.N,
                  "selective_scan_bwd: pass_states must be contiguous f32 (B, D, ceil(L / 512), N)");
      a.ckpt = (float*)p->pass_states.data;
    } else {
      a.ckpt = (float*)p->workspace;
      SsArgs f = a;
      f.out = nullptr; f.last = nullptr; f.z = nullptr; f.D = nullptr;
      if ((rc = ss_launch_fwd(f, udt, stream, true))) return rc;
    }
    int NW = dpg % 16 == 0 ? 16 : 8;
    if (const char* e = getenv("OMK_SELSCAN_BWD_NW")) { if (atoi(e) == 8) NW = 8; }
    const size_t es = dtype_size(udt);
    q.NB = a.N < 16 ? a.N : 16;
    if (const char* e = getenv("OMK_SELSCAN_BWD_NB")) { const int v = atoi(e); if (v >= 1 && v <= 16) q.NB = v < a.N ? v : a.N; }
    q.nOct = 1;
    if (a.N <= q.NB)
      for (int o = 4; o > 1; o >>= 1)
        if (dpg % (NW * o) == 0 && (int64_t)a.B * a.Dm / (NW * o) >= 512) { q.nOct = o; break; }
    if (const char* e = getenv("OMK_SELSCAN_BWD_OCT")) { const int o = atoi(e); if ((o == 1 || o == 2 || o == 4) && a.N <= q.NB && dpg % (NW * o) == 0) q.nOct = o; }
    if (const char* e = getenv("OMK_SELSCAN_BWD_DBG")) q.dbg = atoi(e);
    const size_t smem = (size_t)q.NB * SSR_TP * (8 + 2 * es) + (size_t)(3 * q.nOct * NW * (q.nOct > 1 ? 16 : 64) + NW * 64) * 4;
    dim3 grid((unsigned)((int64_t)a.B * a.Dm / (NW * q.nOct))), block(NW * 64);
#define SSR_GO(T) do { if (OMK_SET_MAX_DYN_SMEM((selscan_bwd_chunked_kernel<T>), smem)) return fail(OMK_ELAUNCH, "selective_scan_bwd: cannot raise dynamic LDS to %zu", smem); \
      OMK_LAUNCH((selscan_bwd_chunked_kernel<T>), grid, block, smem, stream, q); } while (0)
    OMK_DISPATCH_DTYPE(udt, T, SSR_GO(T));
#undef SSR_GO
    return finish_launch("selective_scan_bwd");
  }
  if (a.N > SSB_N) return fail(OMK_EUNSUPPORTED, "selective_scan_bwd: d_state %d > %d needs L-contiguous u / delta / z / dout / B / C (upstream's layout), L >= 64 "
                                                 "and 8 | channels per group", a.N, SSB_N);
  // pass 1: the forward recurrence once more, leaving the state at every SSB_TL-token boundary in the workspace
  a.ckpt = (float*)p->workspace; a.TLB = SSB_TL; a.nTB = (a.L + SSB_TL - 1) / SSB_TL;
  {
    SsArgs f = a;
    f.out = nullptr; f.last = nullptr; f.z = nullptr;
    if ((rc = ss_launch_fwd(f, udt, stream))) return rc;
  }
  // pass 2: adjoint sweep, tiles last to first
  a.DT = SSB_DT;
  const int tiles_per_group = (dpg + SSB_DT - 1) / SSB_DT;
  dim3 grid((unsigned)((int64_t)a.B * a.G * tiles_per_group)), block(SSB_DT);
  const size_t smem = (size_t)SSB_TL * SSB_N * SSB_DT * 4 + (size_t)4 * SSB_TL * SSB_N * 4 + (size_t)7 * SSB_TL * SSB_DT * dtype_size(udt);
#define SSB_GO(T) do { if (OMK_SET_MAX_DYN_SMEM((selscan_bwd_kernel<T>), smem)) return fail(OMK_ELAUNCH, "selective_scan_bwd: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((selscan_bwd_kernel<T>), grid, block, smem, stream, q); } while (0)
  OMK_DISPATCH_DTYPE(udt, T, SSB_GO(T));
#undef SSB_GO
  return finish_launch("selective_scan_bwd");
}
