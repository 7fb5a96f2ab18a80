// selscan.hip -- Mamba-1 selective scan forward (selective_scan_fn signature, BASELINE.json configs[0]).
//
//   x_t = exp(delta_t A[d,:]) x_{t-1} + delta_t B_t u_t ;  y_t = <C_t, x_t> ;  out = (y + D u) * silu(z)
//
// HBM-bound: B*L*(4*D*s + 2*G*N*s) algorithmic bytes.  One thread owns one channel d and
// keeps its N-vector state in registers; a workgroup covers DT adjacent channels of one group and walks L in
// tiles of TL tokens.  Tiles of u / delta / z (and the output) move between HBM and LDS with the lane index
// following whichever of (d, l) has unit stride, so both (B, L, D) "channel-last" storage (adjacent lanes =
// adjacent channels) and upstream's (B, D, L) storage are read in full coalesced rows; B_t / C_t rows of the tile
// are staged once per workgroup and broadcast from LDS to every channel thread.
#include "omk_common.h"

namespace omk {

constexpr int SS_TL = 32;

struct SsArgs {
  const void* u; const void* delta; const void* A; const void* Bm; const void* Cm; const void* D; const void* z; const void* dbias;
  void* out; float* last;
  int64_t usb, usd, usl, dsb, dsd, dsl, zsb, zsd, zsl, osb, osd, osl;
  int64_t Asd, Asn, Bsb, Bsg, Bsn, Bsl, Csb, Csg, Csn, Csl;   // for constant B/C: Bsg = stride over d, Bsn over n
  int B, Dm, L, N, G, DT, softplus, Bvar, Cvar, adt, bdt, cdt, ddt, dbdt;
};

template <class T>
__device__ __forceinline__ void ss_load_tile(const T* g, int64_t sd, int64_t sl, int d0, int nd, int l0, int nl, T* s, int DT) {
  // s[t][c] <- g[(d0+c)*sd + (l0+t)*sl]
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (sl == 1 && sd != 1) {
    for (int i = tid; i < DT * SS_TL; i += nthr) {
      int c = i / SS_TL, t = i % SS_TL;
      s[t * DT + c] = (c < nd && t < nl) ? g[(int64_t)(d0 + c) * sd + (int64_t)(l0 + t)] : T{};
    }
  } else {
    for (int i = tid; i < DT * SS_TL; i += nthr) {
      int t = i / DT, c = i % DT;
      s[t * DT + c] = (c < nd && t < nl) ? g[(int64_t)(d0 + c) * sd + (int64_t)(l0 + t) * sl] : T{};
    }
  }
}
template <class T>
__device__ __forceinline__ void ss_store_tile(T* g, int64_t sd, int64_t sl, int d0, int nd, int l0, int nl, const T* s, int DT) {
  const int tid = threadIdx.x, nthr = blockDim.x;
  if (sl == 1 && sd != 1) {
    for (int i = tid; i < DT * SS_TL; i += nthr) {
      int c = i / SS_TL, t = i % SS_TL;
      if (c < nd && t < nl) g[(int64_t)(d0 + c) * sd + (int64_t)(l0 + t)] = s[t * DT + c];
    }
  } else {
    for (int i = tid; i < DT * SS_TL; i += nthr) {
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
  T* og = (T*)a.out + (int64_t)b * a.osb;
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
    ss_store_tile<T>(og, a.osd, a.osl, d0, nd, l0, nl, so, DT);
    // the next iteration's loads overwrite su/sd/sz only (not so); its first block_sync orders them after this store
  }
  if (a.last && live) {
#pragma unroll
    for (int n = 0; n < NREG; n++)
      if (n < a.N) a.last[((int64_t)b * a.Dm + d) * a.N + n] = x[n];
  }
}

}  // namespace omk

using namespace omk;

extern "C" int omk_selective_scan_fwd(const OmkSelScanFwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->u) && present(p->delta) && present(p->A) && present(p->Bm) && present(p->Cm) && present(p->out), "selective_scan_fwd: u, delta, A, B, C, out required");
  OMK_REQUIRE(p->u.ndim == 3 && p->delta.ndim == 3 && p->out.ndim == 3 && p->A.ndim == 2, "selective_scan_fwd: u/delta/out (B, D, L), A (D, N)");
  SsArgs a = {};
  a.B = (int)p->u.shape[0]; a.Dm = (int)p->u.shape[1]; a.L = (int)p->u.shape[2]; a.N = (int)p->A.shape[1];
  OMK_REQUIRE(p->A.shape[0] == a.Dm, "selective_scan_fwd: A must be (D, N)");
  OMK_REQUIRE(p->delta.dtype == p->u.dtype && p->out.dtype == p->u.dtype && (!present(p->z) || p->z.dtype == p->u.dtype), "selective_scan_fwd: delta, z, out must have u's dtype");
  a.Bvar = p->Bm.ndim == 4; a.Cvar = p->Cm.ndim == 4;
  OMK_REQUIRE((a.Bvar || p->Bm.ndim == 2) && (a.Cvar || p->Cm.ndim == 2), "selective_scan_fwd: B/C must be (B, G, N, L) or (D, N)");
  a.G = 1;
  if (a.Bvar) a.G = (int)p->Bm.shape[1];
  if (a.Cvar) { OMK_REQUIRE(!a.Bvar || p->Cm.shape[1] == a.G, "selective_scan_fwd: B and C group counts differ"); a.G = (int)p->Cm.shape[1]; }
  OMK_REQUIRE(a.G > 0 && a.Dm % a.G == 0, "selective_scan_fwd: D must be a multiple of ngroups");
  OMK_REQUIRE(a.N <= 64, "selective_scan_fwd: d_state > 64 is not supported by the Mamba-1 kernel");
  a.u = p->u.data; a.delta = p->delta.data; a.A = p->A.data; a.Bm = p->Bm.data; a.Cm = p->Cm.data; a.D = p->D.data; a.z = p->z.data;
  a.dbias = p->delta_bias.data; a.out = p->out.data; a.last = (float*)p->last_state.data;
  OMK_REQUIRE(!present(p->last_state) || p->last_state.dtype == OMK_F32, "selective_scan_fwd: last_state must be f32");
  a.usb = p->u.stride[0]; a.usd = p->u.stride[1]; a.usl = p->u.stride[2];
  a.dsb = p->delta.stride[0]; a.dsd = p->delta.stride[1]; a.dsl = p->delta.stride[2];
  if (present(p->z)) { a.zsb = p->z.stride[0]; a.zsd = p->z.stride[1]; a.zsl = p->z.stride[2]; }
  a.osb = p->out.stride[0]; a.osd = p->out.stride[1]; a.osl = p->out.stride[2];
  a.Asd = p->A.stride[0]; a.Asn = p->A.stride[1];
  if (a.Bvar) { a.Bsb = p->Bm.stride[0]; a.Bsg = p->Bm.stride[1]; a.Bsn = p->Bm.stride[2]; a.Bsl = p->Bm.stride[3]; }
  else { a.Bsg = p->Bm.stride[0]; a.Bsn = p->Bm.stride[1]; }
  if (a.Cvar) { a.Csb = p->Cm.stride[0]; a.Csg = p->Cm.stride[1]; a.Csn = p->Cm.stride[2]; a.Csl = p->Cm.stride[3]; }
  else { a.Csg = p->Cm.stride[0]; a.Csn = p->Cm.stride[1]; }
  a.softplus = p->delta_softplus; a.adt = p->A.dtype; a.bdt = p->Bm.dtype; a.cdt = p->Cm.dtype; a.ddt = p->D.dtype; a.dbdt = p->delta_bias.dtype;
  if ((int64_t)a.B * a.Dm * a.L == 0) return OMK_OK;
  const int dpg = a.Dm / a.G;
  a.DT = dpg >= 128 ? 128 : ((dpg + 63) / 64) * 64;
  const int tiles_per_group = (dpg + a.DT - 1) / a.DT;
  dim3 grid((unsigned)((int64_t)a.B * a.G * tiles_per_group)), block(a.DT);
  const size_t smem = (size_t)4 * SS_TL * a.DT * dtype_size(p->u.dtype) + (size_t)2 * SS_TL * a.N * 4;
#define SS_GO(T, NREG) OMK_LAUNCH((selscan_fwd_kernel<T, NREG>), grid, block, smem, stream, a)
  OMK_DISPATCH_DTYPE(p->u.dtype, T, { if (a.N <= 16) SS_GO(T, 16); else SS_GO(T, 64); });
#undef SS_GO
  return finish_launch("selective_scan_fwd");
}

extern "C" int omk_selective_scan_bwd(const OmkSelScanBwd*, omk_stream) {
  return fail(OMK_EUNSUPPORTED, "selective_scan_bwd: the Mamba-1 backward is not implemented (OmniMamba configs use Mamba2 only, "
                                "models/stage2/config_mamba.py:16)");
}
