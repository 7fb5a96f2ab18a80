// ssd_scan.h -- the generalised linear-recurrence scan all Mamba-2 SSD passes are expressed with.
//
//   S_t = exp(a_t) * S_{t-1} + w_t * U_t (x) K_t        S in R^{DU x DK}
//   O_t[u] = sum_k S_t[u][k] * Q_t[k]
//
// forward  y  : U = x  (head, P)   K = B (group, N)  Q = C (group, N)   a = dt'A   w = dt'   t ascending
// backward dC : U = B  (group, N)  K = x (head, P)   Q = dy (head, P)   a = dt'A   w = dt'   t ascending
//          dx : U = dy (head, P)   K = C (group, N)  Q = B (group, N)   a_t := dt'_{t+1} A, w = 1, t descending
//          dB : U = C  (group, N)  K = dy (head, P)  Q = x (head, P)    a_t := dt'_{t+1} A, w = 1, t descending
// (derivation checked against autograd in tests/test_bwd_derivation.py).  Two implementations consume this
// descriptor: the shape-generic fp32 VALU kernel (ssd.hip) and the MFMA chunked kernel (ssd_mfma.hip).
#pragma once
#include <cstdlib>
#include "omk_common.h"

namespace omk {

enum { GS_Y = 0, GS_DC = 1, GS_DX = 2, GS_DB = 3 };
// GScan::flags (the OMK_SSD_* bits of OmkSsdFwd::flags / OmkSsdBwd::flags, include/omk.h, reach the kernels through these)
enum {
  GSF_PRECISE = 1,    // class A forward: the carried state meets Q as bf16 hi + lo and the state-update operand is hi + lo (bare 1e-3 on every head)
  GSF_KHILO = 2,      // class A forward: the state-update operand as hi + lo even when no final state is kept
  GSF_FLUSH = 4,      // ssd_a8: move the basis of the carried state to the chunk end at every chunk (the arithmetic of ssd_a6.hip, bit for bit)
  GSF_NO_SPLIT = 8,   // never split the sequence into segments
  GSF_COLUMN_SLICE = 16,   // class A scans on ssd_a6.hip (eight identical waves) instead of ssd_a8.hip
};

struct Src {            // element (b, t, hh, i) at p[b*sb + t*sl + hh*sh + i]; hh = group index when per_group
  const void* p;
  int64_t sb, sl, sh;
  int dt;
  int per_group;
};

struct GScan {
  int mode;
  Src U, K, Q, X4, Z;   // X4: group vector of length DU for the token-scalar epilogue (C in dC, B in dB); Z: gate (Y)
  const float* dtp;     // (B, H, L) processed dt' (bias + softplus + clamp applied)
  const float* A;       // (H)
  int B, H, G, L, DU, DK;
  int reverse, w_is_dt;
  const void* init; int64_t isb, ish, isu, isk; int init_dt;     // optional initial state, logical [u][k]
  float* fin; int64_t fsb, fsh, fsu, fsk; int fin_extra_decay;   // optional final state (f32); extra exp(a_0) for dinit
  void* out; int64_t osb, osl, osh; int out_dt;                  // Y: out, DX: dx   (b, t, h, u) with u contiguous
  void* outx;                                                    // Y: optional pre-gate copy of out
  const void* D; int64_t Dsh, Dsp; int D_dt;                     // Y, DX
  float* acc32;                                                  // DC/DB: (B, L, G, DU) f32, atomically accumulated
  float* tokscal;                                                // DC: e, DB: wsum  (B, H, L) f32, atomically accumulated
  float* dD; int64_t dDsh, dDsp;                                 // DB: optional grad of D
  float* part;                                                   // DC/DB (MFMA): (B, H/2, L, 128) per-head-pair partial tiles, stored as bf16
  // DC/DB (MFMA): forward-state checkpoints at every chunk end, written by the dC scan in MFMA fragment order
  // (bf16 pairs) and read back by the dB scan, which emits the exact decay-gradient restart values bnd (B, H, nC + 1)
  void* ckpt; float* bnd;
  // checkpoints (and restart values) only at chunk boundaries j with j % ckpt_every == 0, and at the end of the sequence
  int ckpt_every;
  // class A (MFMA) sequence split for few (batch, head) pairs: nseg segments of cps chunks in scan order.  A state-only
  // pass leaves every segment's end state from a zero start in seg[(bh * nseg + s) * SEG_STATE ..] (MFMA accumulator
  // order) and its total log2 decay in seg[BH * nseg * SEG_STATE + bh * nseg + s]; the scan proper folds them.
  float* seg; int nseg, cps;
  int seg_ready;                                                 // seg already holds the folded START states (slot j - 1 = segment j)
  int seg_fmt;                                                   // element order of seg: 0 = ssd_mfma_a3 accumulator order, 1 = logical [u][k] of the class A state
  // class A (MFMA): bf16 images of the carried state at the 128-token window boundaries, for the chunk-parallel backward
  // (ssd_cp.hip): slot (b * dump_nw + w) * H + h holds the state in front of window w in scan direction -- forward: the state
  // BEFORE token 128 w; reverse: the adjoint state at the first token BEHIND window w (dfinal_states for the last one) -- as
  // the raw 16 KB LDS image [u][k] (kx3 swizzle, ssd_tiles.h) the kernel publishes for its own Q . S product.
  uint16_t* dump; int dump_nw;
  int flags;                                                     // GSF_* bits
  // class A forward on ssd_a8.hip: causal depthwise conv1d (width cW <= 4) + SiLU applied to U while it is staged (U = the PRE-conv x);
  // weight element (channel c = h * DU + u, tap k) at cw[c * cwsc + k * cwsk], optional bias cb[c]
  const void* cw; int64_t cwsc, cwsk; int cw_dt, cW; const void* cb; int cb_dt;
  int state_only;                                                // class A (MFMA): no output, only the state pass from the initial state to `fin` (context-parallel shards)
  unsigned long long* prof;                                      // developer only: per-wave phase cycle sums of workgroup 0 (OMK_PROF env)
  int ablate;                                                    // developer only (OMK_PHASE_PROF builds): phases to skip, wrong results
};

constexpr int SEG_STATE = 64 * 128;
struct SegPlan { int nseg, cps; };
// Few (batch, head) sequences leave CUs idle (B = 1, H = 64: a quarter of the chip).  Cut L so that up to two
// workgroups per CU exist; the extra state pass re-reads x and B and costs about a third of a scan, so the split
// only pays from four segments on (measured: B = 1 L = 8192 264 -> 106 us, B = 2 269 -> 190 us, B = 4 loses),
// and segments stay >= 8 chunks long.
// restart interval of the decay-gradient prefix in chunks.  Every checkpoint is 16 KB per head and boundary (537 MB written
// and read back per backward at B 8, L 4096 with an interval of 1), but thinning them costs accuracy: measured on the
// emulator, L = 1024, d(dt) / dA rel-L2 2.1e-3 / 4.4e-2 at 1, 2.2e-3 / 4.6e-2 at 2, 2.8e-3 / 9.0e-2 at 4, 3.8e-3 / 2.0e-1 at 8,
// and at L = 130 an interval of 2 already breaks the 6e-3 bound on d(dt).  Default 1; OMK_SSD_CKPT is an experiment knob.
inline int ssd_ckpt_every() {
  if (const char* e = getenv("OMK_SSD_CKPT")) { const int v = atoi(e); if (v >= 1 && v <= 64) return v; }
  return 1;
}
inline SegPlan ssd_segments(int BH, int L) {
  const int nC = (L + 63) / 64;
  SegPlan p = {1, nC};
  int minc = 8;   // chunks per segment at least (OMK_SSD_SEG_CHUNKS: test hook, lets short sequences split)
  if (const char* e = getenv("OMK_SSD_SEG_CHUNKS")) minc = atoi(e) > 0 ? atoi(e) : minc;
  if (BH <= 0 || BH > 128) return p;
  int n = 512 / BH;
  if (n > nC / minc) n = nC / minc;
  if (n > 16) n = 16;
  if (n < 2 || (n < 4 && minc >= 8)) return p;
  p.cps = (nC + n - 1) / n;
  p.nseg = (nC + p.cps - 1) / p.cps;
  return p;
}
inline size_t ssd_seg_bytes(int BH, int L) {
  const SegPlan p = ssd_segments(BH, L);
  return p.nseg > 1 ? (((size_t)BH * p.nseg * (SEG_STATE + 1) * 4 + 255) & ~(size_t)255) : 0;
}

int ssd_generic_launch(const GScan& g, omk_stream stream);
int ssd_f32_mfma_launch(const GScan& g, omk_stream stream);   // fp32 activations on the fp32 matrix instruction (forward y only); OMK_EUNSUPPORTED otherwise
// returns OMK_EUNSUPPORTED (without touching the error text) when the shape/dtype/layout is outside the MFMA kernel
int ssd_mfma_launch(const GScan& g, omk_stream stream, int dry = 0);   // dry = 1: only answer whether it applies
// split sequences (GScan::seg set, class A style descriptor): state-only pass + fold; afterwards slot j - 1 of g.seg is the
// state at the START of segment j (initial state included).  Shared by the scans whose state this is (y and dC; dx and dB).
int ssd_mfma_prepare_segments(const GScan& g, omk_stream stream, int* seg_fmt = nullptr);   // *seg_fmt: the order it left the states in
// the column-slice class A kernel (ssd_a6.hip): state slices in registers, 32-token sub-chunks, intra tiles shared through LDS, one
// workgroup per head pair (its 16-token predecessor ssd_a5.hip and their experiments: git history, profiles/r04_a5_a6_experiments.txt)
bool ssd_a6_applies(const GScan& g);
int ssd_a6_launch(const GScan& g, omk_stream stream);
int ssd_a6_state_only(const GScan& g, omk_stream stream);
// the specialised-wave class A kernel (ssd_a8.hip): four compute waves of 32 state columns + four helper waves per head pair
bool ssd_a8_applies(const GScan& g);   // OMK_SSD_A8=0: ssd_a6.hip takes its shapes
int ssd_a8_launch(const GScan& g, omk_stream stream);
int ssd_a6_state_dump(const GScan& g, omk_stream stream);   // OMK_EUNSUPPORTED when the column-slice kernel does not take the shape
// state-only pass over the whole sequence that leaves the window-boundary states in g.dump (class A descriptor, no output)
int ssd_mfma_state_dump(const GScan& g, omk_stream stream);
// state-only pass that leaves the state behind the sequence in g.fin (OMK_EUNSUPPORTED outside the MFMA shape)
int ssd_mfma_state_only(const GScan& g, omk_stream stream);
// chunk-parallel dB / dC / token scalars from the dumped states (ssd_cp.hip)
struct CpArgs {
  const uint16_t *X, *DY; int64_t xsb, xsl, xsh, ysb, ysl, ysh;   // (B, L, H, 64) bf16
  const uint16_t *Bm, *Cm; int64_t bsb, bsl, bsg, csb, csl, csg;  // (B, L, G, 128) bf16
  const float* dtp; const float* A;                               // (B, H, L) dt', (H)
  const uint16_t *Sf, *Sg;                                        // window states (GScan::dump images), forward / adjoint
  float *e, *wsum;                                                // (B, H, L) token scalars
  float* bnd;                                                     // (B, H, nT + 1): decay-gradient restart value q at token 64 j
  float* dD; int64_t dDsh;                                        // optional, one D per head
  float *pB, *pC;                                                 // fp32 partials [nhs][B][L][G][128]
  void *dB, *dC; int64_t dbsb, dbsl, dbsg, dcsb, dcsl, dcsg; int dB_dt, dC_dt;
  int B, L, H, G, nW, nhs;
  int direct;   // set by ssd_cp_launch: one head subset and bf16 gradients -- the kernel writes dB / dC itself
  int ablate;   // developer only (OMK_CP_ABLATE): phases to skip, wrong results
  unsigned long long* prof;   // developer only (OMK_PHASE_PROF builds, OMK_CP_PROF=1): per-wave phase cycle sums of workgroup 0
};
int ssd_cp_heads_split(int B, int L, int H, int G);
bool ssd_cp_direct(const CpArgs& a);   // the kernel writes dB / dC itself (the fp32 partial buffers pB / pC are not needed then)
int ssd_cp_launch(const CpArgs& a, omk_stream stream);
int ssd_reduce_partials(const float* part, void* out, int64_t osb, int64_t osl, int64_t osg, int out_dt, int B, int L, int G, int H, omk_stream stream);

}  // namespace omk
