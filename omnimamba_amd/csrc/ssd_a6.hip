// ssd_a6.hip -- class A scan (forward y and the dx scan of the backward): state in COLUMN SLICES, M tiles shared through LDS.
//
// What the measurements of round 4 say about the scan kernels of this repo (profiles/r04_a5_a6_experiments.txt): at two waves per SIMD the
// pipes of a SIMD hardly hide behind each other -- 400 padding VALU instructions per chunk cost their full 4.4 cycles each, the LDS store
// pass of the staging its full time, a matrix instruction its 16 cycles -- so a chunk costs about the SUM of what it asks of the VALU,
// the matrix pipe and the LDS, and a workgroup moves at the pace of its heaviest wave.  The row-strip kernel (ssd_mfma.hip) is paced by
// its fourth strip (930 instructions per 64-token chunk against 650 on the first).  This kernel gives every wave the same ~530:
//
//   * wave w of a head owns the output COLUMNS u in [16 w, 16 w + 16) and the matching slice S[k = 0..127][u] of the running state (eight
//     16 x 16 accumulator tiles).  The bf16 pack of that slice IS the A operand of S_in^T Q^T: the state never goes through LDS, is never
//     published, and there is no barrier on its account.  Every wave does the same work.
//   * the chunk is walked in SUB-CHUNKS of 32 tokens.  Per sub-chunk: one pack of the state (16 cvt), 8 MFMAs for S_in^T Q^T (two strips of
//     16 rows), 8 MFMAs for the state update with all 32 contraction slots carrying tokens (A = two transposed K fragments, B = the
//     ws-scaled U fragments of the two strips), 3 MFMAs for the intra block.
//   * the intra block M (3 tiles of 16 x 16 per sub-chunk: two diagonal, one full) does not depend on the column slice, and its G = K Q^T
//     not even on the head of the pair: ONE wave of the workgroup builds each of the six tiles of a chunk for both heads (4 MFMAs, then
//     per head: decay as row factor x column factor prepared by the scalar wave, causal mask, D on the diagonal, bf16 hi + lo) and
//     leaves them in LDS as ready B operands.  Tiles are built one chunk AHEAD (K / Q / token scalars are staged two chunks ahead of
//     their use, three LDS buffers), so the one barrier per chunk that the staging needs anyway also publishes them.
//
// One workgroup = 8 waves = the two heads of a head PAIR (the group's K / Q tiles are staged once for both), one workgroup per CU.
// Contraction-slot bookkeeping (an MFMA sums over its 32 slots in any order as long as A and B agree):
//   state tile t (0..7), accumulator register r on lane (n = lane & 15, g = lane >> 4)  <->  u = 16 w + n,
//   k = 32 (t >> 1) + 8 g + 4 (t & 1) + r -- the registers of tiles 2i, 2i + 1 are k = 32 i + 8 g + 0..7, what a 16-byte row read of Q hands
//   lane (l, g) for k-step i.  (The transposed reads of the state update fetch ONE 8-byte half of a segment per lane: 2-way bank
//   conflicts.  Swapping the halves in rows with bit 2 set removes them and costs 8-byte K stores / row reads: measured neutral.)
#include <cstdlib>
#include "ssd_scan.h"
#include "ssd_tiles.h"

#ifndef OMK_A6_ABL
#define OMK_A6_ABL 0
#endif

namespace omk {

constexpr int QA6 = 64;    // tokens staged per barrier
struct SmemA6 {
  uint16_t K[3][QA6 * 128];       // kx3 swizzle
  uint16_t Q[3][QA6 * 128];       // kx3 swizzle
  uint16_t U[2][2][QA6 * 64];     // [buffer][head of the pair], ux3 swizzle
  u32x4 M[2][2][6][64];           // [buffer][head][record][lane]: per sub-chunk jj: 3 jj + 0 = {hi, lo} of tile (strip 0, block 0);
                                  // 3 jj + 1 = {hi of (1, 0), hi of (1, 1)}; 3 jj + 2 = {lo of (1, 0), lo of (1, 1)}
  float rl[3][2][QA6], ws[3][2][QA6];   // [buffer][head][chunk row]: the row's factor of S_in^T Q^T, the weight of its state-update term
  float dec[3][2][2];             // decay over sub-chunk jj
  // the tile builders' scalars, [tile buffer][head][chunk row]: the decay exp2(cs_l - cs_s) w_s of an M entry as a PRODUCT of a row
  // and a column factor around a reference inside the tile (diagonal tiles: the middle of the 16-token block, rfd / cfd; the full
  // tile: the block boundary, fo = the column factor in block 0 and the row factor in block 1 of a sub-chunk) -- three exponentials
  // per token in the scalar wave instead of four per lane and tile in the builders.  wide[][] != 0: a block of the chunk decays by
  // more than 2^-90 and the factors could leave the fp32 range; the arrays then hold the exponents (cs_l, log2 w_s - cs_s) and the
  // builders take the exponential of their sum per entry, as they did before round 4
  float rfd[2][2][QA6], cfd[2][2][QA6], fo[2][2][QA6];
  int wide[2][2];
};
static_assert(sizeof(SmemA6) <= 160 * 1024, "one workgroup per CU");


// STATE: the state-only pass (no output, no S_in^T Q^T, no intra block): window-boundary images (GScan::dump) and / or the state behind
// the sequence, with exactly the arithmetic the scan proper carries its state with
template <int MODE, bool EXTRAS, bool DFOLD, bool DUMP, bool KHILO, bool STATE = false>
__global__ __launch_bounds__(512) void ssd_a6_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemA6& sm = *reinterpret_cast<SmemA6*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_i(tid >> 6);
  const int hh = wave >> 2, w = wave & 3;
  const int g16 = lane >> 4, t16 = lane & 15;
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous (batch, pair)
  const int pairs = a.H >> 1;
  const int hp = vid % pairs, seg = (vid / pairs) % a.nseg, b = vid / (pairs * a.nseg);
  const int h = 2 * hp + hh;
  const int g = (2 * hp) / (a.H / a.G);
  const int nC = (a.L + QA6 - 1) / QA6;
  const int c0 = seg * a.cps, c1 = (c0 + a.cps < nC) ? c0 + a.cps : nC;
  const bool rev = a.reverse != 0;
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QA6; };
  auto clipc = [&](int c) -> int { return c < c1 ? c : c1 - 1; };   // (chunks behind the end re-stage the last one: no branch around loads)
  auto rowtok = [&](int i) -> int { return rev ? QA6 - 1 - i : i; };

  // ---- staging: K, Q two 16-byte segments per thread (rows rowk + 32 r), U of the wave's own head two (rows rowu + 32 r)
  const int rowk = tid >> 4, ck8 = (tid & 15) * 8, rowu = (tid & 255) >> 3, cu8 = (tid & 7) * 8;
  const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
  const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
  const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  const int ksl = (int)a.K.sl, qsl = (int)a.Q.sl, usl = (int)a.U.sl, osl = (int)a.osl;
  const BufRes Kr = make_buf(Kb, (uint32_t)((int64_t)a.L * ksl * 2)), Qr = make_buf(Qb, (uint32_t)((int64_t)a.L * qsl * 2));
  const BufRes Ur = make_buf(Ub, (uint32_t)((int64_t)a.L * usl * 2)), Dr = make_buf(dtrow, (uint32_t)((int64_t)a.L * 4));
  const uint32_t kvo = 2u * (uint32_t)((rev ? 31 - rowk : rowk) * ksl + ck8), qvo = 2u * (uint32_t)((rev ? 31 - rowk : rowk) * qsl + ck8);
  const uint32_t uvo = 2u * (uint32_t)((rev ? 31 - rowu : rowu) * usl + cu8);
  const uint32_t dvo = 4u * (uint32_t)rowtok(lane), dvo_a = 4u * (uint32_t)(rowtok(lane) + (rev ? 1 : 0));
  u32x4 rk[2], rq[2], ru[2];
  float rdt = 0.f, rda = 0.f, rwv = 0.f;
  int stlo = 0;   // first token of the K / Q / dt chunk in the staging registers
  auto prefetch_kq = [&](int tl) {
    stlo = tl;
    const uint32_t sk = 2u * (uint32_t)(tl * ksl), sq = 2u * (uint32_t)(tl * qsl);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int ro = rev ? 32 * (1 - r) : 32 * r;
      rk[r] = buf_ld16(Kr, kvo, sk + 2u * (uint32_t)(ro * ksl));
      if (!STATE) rq[r] = buf_ld16(Qr, qvo, sq + 2u * (uint32_t)(ro * qsl));
    }
    rdt = buf_ld_f32(Dr, dvo, 4u * (uint32_t)tl);
    rda = buf_ld_f32(Dr, dvo_a, 4u * (uint32_t)tl);
  };
  auto prefetch_u = [&](int tl) {
    const uint32_t su = 2u * (uint32_t)(tl * usl);
#pragma unroll
    for (int r = 0; r < 2; r++) ru[r] = buf_ld16(Ur, uvo, su + 2u * (uint32_t)((rev ? 32 * (1 - r) : 32 * r) * usl));
  };
  const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
  // (the commit of an iteration goes in three pieces -- K in front of phase 2 of sub-chunk 0, Q behind it, U in front of the barrier: as one
  // burst of 48 ds_write_b128 per CU it kept the LDS pipe from the fragment reads for ~770 cycles per chunk; in pieces 1 % faster)
  auto commit_k = [&](int kb) {   // rows past the end arrived as zeros
#pragma unroll
    for (int r = 0; r < 2; r++) st16(&sm.K[kb][o_ck + 32 * 128 * r], rk[r]);
  };
  auto commit_q = [&](int kb) {
#pragma unroll
    for (int r = 0; r < 2; r++) if (!STATE) st16(&sm.Q[kb][o_ck + 32 * 128 * r], rq[r]);
  };
  auto commit_kq = [&](int kb) { commit_k(kb); commit_q(kb); };
  auto commit_u = [&](int ub) {
#pragma unroll
    for (int r = 0; r < 2; r++) st16(&sm.U[ub][hh][o_cu + 32 * 64 * r], ru[r]);
  };
  const float Ah = a.A[h];
  const float Ah2 = Ah * LOG2E;
  auto scalars = [&](int kb, int mb) {   // waves with w == 0; lanes = rows of the staged K / Q / dt chunk of head hh
    {
      const int t = stlo + rowtok(lane);
      const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
      rwv = okd ? (a.w_is_dt ? rdt : 1.f) : 0.f;
      rdt = okd ? rdt : 0.f;
      rda = oka ? rda : 0.f;
    }
    const float cs = wave_incl_scan_add(rda * Ah2);
    const float e31 = wave_read_lane(cs, 31), e63 = wave_read_lane(cs, 63);
    const float csb = lane < 32 ? 0.f : e31;    // prefix in front of the lane's sub-chunk
    const float cse = lane < 32 ? e31 : e63;    // prefix at its end
    // dx scan: the output row is scaled by dt'_l -- folded into the row's two factors (of S_in^T Q^T and of the M build), so that
    // D dy can ride on the diagonal of M unscaled like D x does in the forward
    const float rsc = MODE == GS_DX ? rdt : 1.f;
    // LAZY decay: unless the first sub-chunk decays by more than 2^-60, the state is carried through it UNDECAYED (T = S / d0: the
    // update terms of its tokens weighted 2^(-cs_s) instead of 2^(cs_31 - cs_s), its dec == 1.f tells phase 2 to skip the multiply),
    // the rows of the second sub-chunk take d0 into their factor of S_in^T Q^T, and its update applies d0 d1 at once: one pass of
    // multiplies over the state slice per chunk instead of two.  The state is exact again at every chunk boundary (images, final state).
    const bool lazy = e31 > -60.f;
    const float csb_r = lazy ? 0.f : csb, cse_w = lane < 32 ? (lazy ? 0.f : e31) : e63;
    sm.rl[kb][hh][lane] = exp2_fast(cs - csb_r) * rsc;
    sm.ws[kb][hh][lane] = rwv * exp2_fast(cse_w - cs);
    if ((lane & 31) == 31) sm.dec[kb][hh][lane >> 5] = lane < 32 ? (lazy ? 1.f : exp2_fast(e31)) : exp2_fast(lazy ? e63 : e63 - e31);
    if (STATE) return;
    const int b16 = lane & ~15;
    const bool blk1 = (lane & 16) != 0;
    const float cmid = shfl(cs, b16 + 7), cbnd = shfl(cs, blk1 ? b16 - 1 : b16 + 15);
    const bool wide = ballot_any(fabsf(cs - cmid) > 90.f);
    if (!wide) {
      sm.rfd[mb][hh][lane] = exp2_fast(cs - cmid) * rsc;
      sm.cfd[mb][hh][lane] = rwv * exp2_fast(cmid - cs);
      sm.fo[mb][hh][lane] = blk1 ? exp2_fast(cs - cbnd) * rsc : rwv * exp2_fast(cbnd - cs);
    } else {
      const float csr = MODE == GS_DX ? cs + log2_fast(rdt) : cs, lw = log2_fast(rwv) - cs;
      sm.rfd[mb][hh][lane] = csr;
      sm.cfd[mb][hh][lane] = lw;
      sm.fo[mb][hh][lane] = blk1 ? csr : lw;
    }
    if (lane == 0) sm.wide[mb][hh] = wide ? 1 : 0;
  };

  // ---- lane-constant LDS element offsets
  int o_rd[4], o_kt[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    o_rd[i] = kx3(t16, 32 * i + 8 * g16);                               // 16-byte row reads of Q / K: row t16, k = 32 i + 8 g16 ..
    o_kt[i] = kx3(4 * g16 + (t16 >> 2), 32 * i + 8 * (t16 & 3));        // K^T transpose reads: rows 4 g16 + 0..3, k = 32 i + 8 q (+ 4 for odd tiles)
  }
  const int o_uf = ux3(4 * g16 + (t16 >> 2), 16 * w + 4 * (t16 & 3));   // U transpose read: rows 4 g16 + 0..3, columns 16 w + 0..15
  const int o_xu = ux3(t16, 16 * w + 4 * g16);                          // x of the lane's output row, columns 16 w + 4 g16 ..

  // ---- running state: eight 16 x 16 tiles, see the header for the (tile, register) <-> k map
  f32x4 accS[8];
#pragma unroll
  for (int t = 0; t < 8; t++) accS[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t bh = (int64_t)b * a.H + h;
  const int su = 16 * w + t16;   // the lane's state column
  if (seg > 0) {   // folded by ssd_seg_fold_kernel (row-strip accumulator order): slot seg - 1 = state at the start of this segment
    const float* sp = a.seg + (bh * a.nseg + seg - 1) * SEG_STATE;
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r, kk = k & 31;
        accS[t][r] = sp[((2 * (k >> 5) + (su >> 5)) * 16 + (kk & 3) + 4 * (kk >> 3)) * 64 + 32 * ((kk >> 2) & 1) + (su & 31)];
      }
  }
  if (a.init && seg == 0) {
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r;
        accS[t][r] = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)su * a.isu + (int64_t)k * a.isk, a.init_dt);
      }
  }
  f32x4 Du = {0.f, 0.f, 0.f, 0.f};   // D of the lane's four output columns (epilogue form)
  if (!DFOLD && a.D) {
#pragma unroll
    for (int r = 0; r < 4; r++) Du[r] = load_rt(a.D, (int64_t)h * a.Dsh + (int64_t)(16 * w + 4 * g16 + r) * a.Dsp, a.D_dt);
  }

#ifdef OMK_PHASE_PROF   // developer build (tools/phase_prof_a6.py): s_memtime deltas per phase, workgroup 0
  uint64_t pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
#define PT6(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
#define PTW(i, v) do { if (prof) { asm volatile("v_readfirstlane_b32 s0, %0" :: "v"(v) : "s0"); PT6(i); } } while (0)
  uint64_t tprev = prof ? clock64_() : 0;
  const uint64_t t_core0 = tprev, t_ref0 = prof ? __builtin_readsteadycounter() : 0;
#else
#define PT6(i) do { } while (0)
#define PTW(i, v) do { } while (0)
#endif
  // ---- M tiles of one chunk (buffers kb: K / Q, mb: tiles and their scalars).  G = K Q^T is the same for the two heads of the pair:
  // each of the six tiles (sub-chunk jj: (0, 0), (1, 0), (1, 1)) is one wave's -- wave (hh, w) builds tile w - 1 of sub-chunk jj = hh
  // for BOTH heads (4 MFMAs, then per head the decay, the causal mask, D on the diagonal, bf16 hi + lo); the waves w = 0 compute
  // the token scalars.  G^T[s][l]: A = K rows s, B = Q rows l; the lane holds s = 4 g16 + r of its own l = t16.
  auto krow = [&](int kb, int row0, int i) -> u32x4 {   // K[row0 + t16][32 i + 8 g16 .. + 7]
    return ld16(&sm.K[kb][o_rd[i] + 128 * row0]);
  };
  float Dh2[2] = {0.f, 0.f};
  if (DFOLD && a.D) {
    Dh2[0] = load_rt(a.D, (int64_t)(2 * hp) * a.Dsh, a.D_dt);
    Dh2[1] = load_rt(a.D, (int64_t)(2 * hp + 1) * a.Dsh, a.D_dt);
  }
  // The builder's LDS reads (operands of G, the factors of both heads) are requested a phase before its arithmetic
  // (build_loads in front of phase 2 of sub-chunk 0, build_tiles behind it): on their own they were a chain of three exposed LDS
  // round trips, 1300 of the 5600 cycles of a chunk (tools/phase_prof_a6.py).
  struct FragB { u32x4 k[4], q[4]; };
  const int bjj = hh, btt = w - 1;                                                          // the wave's tile: sub-chunk, (0, 0) / (1, 0) / (1, 1)
  const int brb = 32 * bjj + (btt >= 1 ? 16 : 0), bcb = 32 * bjj + (btt == 2 ? 16 : 0);    // its first row (l) / column (s)
  const bool bdiag = btt != 1;
  auto build_loads = [&](FragB& f, int kb, int mb) {
    if (STATE || w == 0) return;   // (the waves that compute the token scalars)
#pragma unroll
    for (int i = 0; i < 4; i++) { f.k[i] = krow(kb, bcb, i); f.q[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * brb]); }
  };
  auto build_tiles = [&](const FragB& f, int mb) {
    if (STATE || w == 0) return;
    float rf2[2];
    f32x4 cf2[2];
    int wide2[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {   // (requested in front of the MFMAs)
      const float* rfa = bdiag ? sm.rfd[mb][h2] : sm.fo[mb][h2];
      const float* cfa = bdiag ? sm.cfd[mb][h2] : sm.fo[mb][h2];
      rf2[h2] = rfa[brb + t16];
      cf2[h2] = *reinterpret_cast<const f32x4*>(&cfa[bcb + 4 * g16]);
      wide2[h2] = sm.wide[mb][h2];
    }
    f32x4 gt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) gt = mfma16x16x32_bf16(as_s16x8(f.k[i]), as_s16x8(f.q[i]), gt);
    PTW(7, gt[0]);
#pragma unroll
    for (int h2 = 0; h2 < 2; h2++) {
      const float rf = rf2[h2];
      const f32x4 cf = cf2[h2];
      float v[4];
      if (uniform_i(wide2[h2]) != 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = gt[r] * exp2_fast(rf + cf[r]);
      } else {
        const f32x4 gc = gt * cf * rf;
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = gc[r];
      }
      if (bdiag) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (DFOLD) v[r] = (4 * g16 + r < t16) ? v[r] : (4 * g16 + r == t16 ? v[r] + Dh2[h2] : 0.f);
          else v[r] = (4 * g16 + r <= t16) ? v[r] : 0.f;
        }
      }
      uint32_t hi[2], lo[2];
#pragma unroll
      for (int p2 = 0; p2 < 2; p2++) {
        hi[p2] = pack_bf16x2(v[2 * p2], v[2 * p2 + 1]);
        lo[p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi[p2]), v[2 * p2 + 1] - bf_hi(hi[p2]));
      }
      PTW(8 + h2, lo[1]);
      if (btt == 0) sm.M[mb][h2][3 * bjj][lane] = u32x4{hi[0], hi[1], lo[0], lo[1]};
      else {
        uint32_t* mh = reinterpret_cast<uint32_t*>(&sm.M[mb][h2][3 * bjj + 1][lane]) + 2 * (btt - 1);
        *reinterpret_cast<u32x2*>(mh) = u32x2{hi[0], hi[1]};
        *reinterpret_cast<u32x2*>(mh + 4 * 64) = u32x2{lo[0], lo[1]};
      }
    }
  };
  FragB fb;

  // ---- prologue: chunks c0 and c0 + 1 staged, tiles of c0 built
  prefetch_kq(chunk_lo(c0));
  prefetch_u(chunk_lo(c0));
  commit_kq(0);
  commit_u(0);
  if (w == 0) scalars(0, 0);
  prefetch_kq(chunk_lo(clipc(c0 + 1)));
  commit_kq(1);
  if (w == 0) scalars(1, 1);
  block_sync();
  build_loads(fb, 0, 0);
  build_tiles(fb, 0);
  prefetch_kq(chunk_lo(clipc(c0 + 2)));
  prefetch_u(chunk_lo(clipc(c0 + 1)));
  block_sync();
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
  const BufRes Or = make_buf(STATE ? nullptr : ob, STATE ? 0u : (uint32_t)((int64_t)a.L * osl * 2));
  uint16_t* oxb = a.outx ? (uint16_t*)a.outx + (int64_t)b * a.osb + (int64_t)h * a.osh : nullptr;
  const uint16_t* zb = (MODE == GS_Y && a.Z.p) ? (const uint16_t*)a.Z.p + (int64_t)b * a.Z.sb + (int64_t)h * a.Z.sh : nullptr;
  const int zsl = (int)a.Z.sl;

  // ---- the sub-chunk pipeline.  A sub-chunk (32 tokens) is two phases: (1) pack of the state slice + S_in^T Q^T on the Q row
  // fragments of its two strips; (2) state update, U^T M^T and the output rows on the transposed K / U fragments, the shared M tiles
  // and the token scalars.  The operands of phase 2 are requested in front of phase 1 of the same sub-chunk, the row fragments of the
  // next sub-chunk in front of phase 2, so every LDS read has a phase of work between request and use.
  struct FragR { u32x4 q0[4], q1[4]; };
  struct FragC { s16x4 u0, u1, kt[8][2]; float rl0, rl1; f32x4 ws0, ws1; float dec; u32x4 m0, mh, ml; u32x2 x0, x1; };
  auto load_rows = [&](FragR& f, int kb, int jj) {
    if (STATE) return;
    if (OMK_A6_ABL & 1) { asm volatile("" : "+v"(f.q0[0]), "+v"(f.q0[1]), "+v"(f.q0[2]), "+v"(f.q0[3]), "+v"(f.q1[0]), "+v"(f.q1[1]), "+v"(f.q1[2]), "+v"(f.q1[3])); return; }
#pragma unroll
    for (int i = 0; i < 4; i++) f.q0[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * (32 * jj)]);
#pragma unroll
    for (int i = 0; i < 4; i++) f.q1[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * (32 * jj + 16)]);
  };
  // The 30 LDS reads of a sub-chunk's phase-2 operands are requested in five GROUPS, the first in front of phase 1 and one behind each of
  // its four k-steps: a wave can have 15 LDS instructions in flight (lgkmcnt), a burst of 30 stops it from issuing anything else for
  // two LDS latencies -- and its partner on the SIMD is doing the same thing at the same time.
  auto load_cols = [&](FragC& f, int kb, int ub, int jj, int grp) {
    if ((OMK_A6_ABL & 1) && grp != 0) return;
    if (OMK_A6_ABL & 1) {
      asm volatile("" : "+v"(f.u0), "+v"(f.u1), "+v"(f.rl0), "+v"(f.rl1), "+v"(f.ws0), "+v"(f.ws1), "+v"(f.dec), "+v"(f.m0), "+v"(f.mh), "+v"(f.ml));
#pragma unroll
      for (int t = 0; t < 8; t++) asm volatile("" : "+v"(f.kt[t][0]), "+v"(f.kt[t][1]));
      return;
    }
    const int r0 = 32 * jj;
    if (grp == 0) {
      f.ws0 = *reinterpret_cast<const f32x4*>(&sm.ws[kb][hh][r0 + 4 * g16]);
      f.ws1 = *reinterpret_cast<const f32x4*>(&sm.ws[kb][hh][r0 + 16 + 4 * g16]);
      f.u0 = lds_read_tr16_b64(&sm.U[ub][hh][o_uf + 64 * r0]);          // U[r0 + 4 g16 + e][16 w + t16]
      f.u1 = lds_read_tr16_b64(&sm.U[ub][hh][o_uf + 64 * (r0 + 16)]);
      f.dec = sm.dec[kb][hh][jj];
    }
    if (STATE ? grp == 0 : true) {
#pragma unroll
      for (int t = 0; t < 8; t++) {
        if (!STATE && (t >> 1) != (grp == 0 ? 0 : grp == 1 ? -1 : grp - 1)) continue;   // k-steps 0 | 1 | 2 | 3 in groups 0 | 2 | 3 | 4 (STATE: all in group 0)
        f.kt[t][0] = lds_read_tr16_b64(&sm.K[kb][o_kt[t >> 1] + 4 * (t & 1) + 128 * r0]);
        f.kt[t][1] = lds_read_tr16_b64(&sm.K[kb][o_kt[t >> 1] + 4 * (t & 1) + 128 * (r0 + 16)]);
      }
    }
    if (STATE) return;
    if (grp == 1) {
      f.m0 = sm.M[ub][hh][3 * jj][lane];
      f.mh = sm.M[ub][hh][3 * jj + 1][lane];
      f.ml = sm.M[ub][hh][3 * jj + 2][lane];
    }
    if (grp == 4) {
      f.rl0 = sm.rl[kb][hh][r0 + t16];
      f.rl1 = sm.rl[kb][hh][r0 + 16 + t16];
      if (!DFOLD) {
        f.x0 = *reinterpret_cast<const u32x2*>(&sm.U[ub][hh][o_xu + 64 * r0]);
        f.x1 = *reinterpret_cast<const u32x2*>(&sm.U[ub][hh][o_xu + 64 * (r0 + 16)]);
      }
    }
  };
  f32x4 accA0, accA1;
  // Window-state images (GScan::dump) leave through a buffer resource in the segment order of ssd_tiles.h (img_off): 1 KB of consecutive
  // bytes per store instruction.  (Measured, training forward B 8 L 4096: images as 16 rows x 64 bytes per instruction 252 us, consecutive
  // 229 - 239 us, consecutive and issued in every chunk with the unwanted ones sent behind the end of the resource 233 - 244 us; without
  // images 202 us.  The cost is the store path of the CU, not HBM: images aimed at one L2-resident spot cost the same.)
  const uint32_t dump_nb = (DUMP && a.dump) ? (uint32_t)((((int64_t)a.dump_nw - 1) * a.H + 1) << 14) : 0u;
  const BufRes Pr = make_buf((DUMP && a.dump) ? a.dump + ((((int64_t)b * a.dump_nw) * a.H + h) << 13) : nullptr, dump_nb);
  const uint32_t pvo = 16u * (uint32_t)(256 * w + lane);   // segment (4 w + i) 64 + lane of the image (img_off in ssd_tiles.h): + 1 KB per k-step i
  auto phase1 = [&](const FragR& f, bool dump_slot, bool dump_here, uint32_t dso, FragC& nf, int nkb, int nub, int njj) {   // n*: the operand groups to request
    if (STATE && !(DUMP && dump_here)) return;
    accA0 = f32x4{0.f, 0.f, 0.f, 0.f}; accA1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u32x4 sp;
      sp[0] = pack_bf16x2(accS[2 * i][0], accS[2 * i][1]);
      sp[1] = pack_bf16x2(accS[2 * i][2], accS[2 * i][3]);
      sp[2] = pack_bf16x2(accS[2 * i + 1][0], accS[2 * i + 1][1]);
      sp[3] = pack_bf16x2(accS[2 * i + 1][2], accS[2 * i + 1][3]);
      if (DUMP && dump_slot && dump_here) buf_st16(Pr, sp, pvo + 1024u * (uint32_t)i, dso);
      if (!STATE) {
        accA0 = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q0[i]), accA0);
        accA1 = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q1[i]), accA1);
        OMK_SCHED_FENCE();
        load_cols(nf, nkb, nub, njj, i + 1);
        OMK_SCHED_FENCE();
      }
    }
  };
  auto out_rows = [&](f32x4 o, int row, int tlo, u32x2 xr) {   // the lane's row, columns 16 w + 4 g16 + r
    const int erow = rowtok(row);
    if (!DFOLD) o = o + Du * f32x4{bf_lo(xr[0]), bf_hi(xr[0]), bf_lo(xr[1]), bf_hi(xr[1])};
    uint32_t eoff = (uint32_t)(erow * osl + 16 * w + 4 * g16);
    if (OMK_A6_ABL & 512) eoff = (uint32_t)(rowtok((row & ~15) + 4 * w + g16) * osl + 4 * t16);   // (ablation: four full 128-byte rows per store, wrong data)
    if (MODE == GS_Y && EXTRAS && tlo + erow < a.L) {
      if (oxb) {
        u32x2 ox = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        *reinterpret_cast<u32x2*>(oxb + (int64_t)tlo * osl + eoff) = ox;
      }
      if (zb) {
        const u32x2 zr = *reinterpret_cast<const u32x2*>(zb + (int64_t)tlo * zsl + erow * zsl + 16 * w + 4 * g16);
        o[0] *= silu_fast(bf_lo(zr[0])); o[1] *= silu_fast(bf_hi(zr[0]));
        o[2] *= silu_fast(bf_lo(zr[1])); o[3] *= silu_fast(bf_hi(zr[1]));
      }
    }
    const u32x2 ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    if (!(OMK_A6_ABL & 16)) buf_st8(Or, ov, 2u * eoff, 2u * (uint32_t)(tlo * osl));
    else asm volatile("" :: "v"(ov));
  };
  auto phase2 = [&](const FragC& f, int jj, int tlo) {
    // ---- state update: S = dec S + K^T (ws U) over the 32 tokens of the sub-chunk (contraction slots: strip 0 | strip 1)
    {
      u32x4 uh, ul;
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const s16x4& uf = s2 ? f.u1 : f.u0;
        const f32x4& ws4 = s2 ? f.ws1 : f.ws0;
        float us[4];
#pragma unroll
        for (int e = 0; e < 4; e++) us[e] = bf16_to_f32((uint16_t)uf[e]) * ws4[e];
#pragma unroll
        for (int p2 = 0; p2 < 2; p2++) {
          const uint32_t hi = pack_bf16x2(us[2 * p2], us[2 * p2 + 1]);
          uh[2 * s2 + p2] = hi;
          if (KHILO) ul[2 * s2 + p2] = pack_bf16x2(us[2 * p2] - bf_lo(hi), us[2 * p2 + 1] - bf_hi(hi));
        }
      }
      if (uniform_i((int)__builtin_bit_cast(uint32_t, f.dec)) != 0x3f800000) {   // (1.f: the scalar wave's mark of a lazily carried sub-chunk)
#pragma unroll
        for (int t = 0; t < 8; t++) accS[t] = accS[t] * f.dec;
      }
#pragma unroll
      for (int t = 0; t < 8; t++) {
        s16x8 kk;
        kk[0] = f.kt[t][0][0]; kk[1] = f.kt[t][0][1]; kk[2] = f.kt[t][0][2]; kk[3] = f.kt[t][0][3];
        kk[4] = f.kt[t][1][0]; kk[5] = f.kt[t][1][1]; kk[6] = f.kt[t][1][2]; kk[7] = f.kt[t][1][3];
        accS[t] = mfma16x16x32_bf16(kk, as_s16x8(uh), accS[t]);
        if (KHILO) accS[t] = mfma16x16x32_bf16(kk, as_s16x8(ul), accS[t]);
      }
    }
    if (STATE) return;
    // ---- U^T M^T: strip 0 against tile (0, 0) (hi | lo in the two halves of the contraction, U twice); strip 1 against (1, 0) | (1, 1)
    s16x8 u00, u01;
    u00[0] = f.u0[0]; u00[1] = f.u0[1]; u00[2] = f.u0[2]; u00[3] = f.u0[3]; u00[4] = f.u0[0]; u00[5] = f.u0[1]; u00[6] = f.u0[2]; u00[7] = f.u0[3];
    u01[0] = f.u0[0]; u01[1] = f.u0[1]; u01[2] = f.u0[2]; u01[3] = f.u0[3]; u01[4] = f.u1[0]; u01[5] = f.u1[1]; u01[6] = f.u1[2]; u01[7] = f.u1[3];
    const f32x4 accB0 = mfma16x16x32_bf16(u00, as_s16x8(f.m0), f32x4{0.f, 0.f, 0.f, 0.f});
    f32x4 accB1 = mfma16x16x32_bf16(u01, as_s16x8(f.mh), f32x4{0.f, 0.f, 0.f, 0.f});
    accB1 = mfma16x16x32_bf16(u01, as_s16x8(f.ml), accB1);
    out_rows(accA0 * f.rl0 + accB0, 32 * jj + t16, tlo, f.x0);
    out_rows(accA1 * f.rl1 + accB1, 32 * jj + 16 + t16, tlo, f.x1);
  };

#ifdef OMK_PHASE_PROF
  tprev = prof ? clock64_() : 0;
#endif
  FragR fr;
  FragC fc;
  load_rows(fr, 0, 0);
  int kb0 = 0, kb1 = 1, kb2 = 2;   // K / Q / scalar buffers of chunks c, c + 1, c + 2
  for (int c = c0; c < c1; c++) {
    const int ub0 = (c - c0) & 1, ub1 = ub0 ^ 1;
    const int tlo = chunk_lo(c);
    bool dump_here = false;
    uint32_t dso = dump_nb;
    if (DUMP && a.dump) {   // window-boundary image of the state in front of this chunk, the [u][k] kx3 image ssd_cp.hip reads
      const int cid = rev ? nC - 1 - c : c;
      dump_here = rev ? (cid == nC - 1 || (cid & 1)) : !(cid & 1);
      if (OMK_A6_ABL & 256) dump_here = false;                                              // (ablation: no image leaves)
      if (dump_here) dso = (uint32_t)(((int64_t)(cid >> 1) * a.H) << 14);
      if (OMK_A6_ABL & 128) dso = dump_here ? 0u : dump_nb;                                 // (ablation: every image of a head lands on its first)
    }
    // ---- sub-chunk 0
    const bool skipc = (OMK_A6_ABL & 32) && hh == 1;   // (ablation: the second head's waves only stage)
    const bool skipb = skipc || (OMK_A6_ABL & 64);     // (ablation: no tile build)
    if (!skipc) {
    load_cols(fc, kb0, ub0, 0, 0);
    OMK_SCHED_FENCE();
    phase1(fr, true, dump_here, dso, fc, kb0, ub0, 0);
    OMK_SCHED_FENCE();
    }
    if (c + 1 < c1 && !skipb) build_loads(fb, kb1, ub1);
    if (!(OMK_A6_ABL & 2)) commit_k(kb2);
    OMK_SCHED_FENCE();
    if (!skipc) {
    phase2(fc, 0, tlo);
    OMK_SCHED_FENCE();
    load_rows(fr, kb0, 1);
    }
    PT6(0);
    // ---- the tiles of the next chunk, the staging of chunk c + 2 (K / Q / scalars) and c + 1 (U)
    if (c + 1 < c1 && !skipb) build_tiles(fb, ub1);
    PT6(1);
    if (!(OMK_A6_ABL & 2)) commit_q(kb2);
    PT6(2);
    if (w == 0) scalars(kb2, ub0);
    PT6(3);
#ifndef OMK_A6_LATEPF
    if (!(OMK_A6_ABL & 4)) prefetch_kq(chunk_lo(clipc(c + 3)));   // a whole iteration ahead of their commit
#endif
    OMK_SCHED_FENCE();
    // ---- sub-chunk 1; the barrier of the chunk behind its last request for the current buffers
    if (!skipc) {
    load_cols(fc, kb0, ub0, 1, 0);
    OMK_SCHED_FENCE();
    phase1(fr, false, false, dump_nb, fc, kb0, ub0, 1);
    OMK_SCHED_FENCE();
    }
    if (!(OMK_A6_ABL & 2)) commit_u(ub1);
    if (!(OMK_A6_ABL & 4)) prefetch_u(chunk_lo(clipc(c + 2)));
    PT6(4);
    if (!(OMK_A6_ABL & 8)) block_sync();
    PT6(5);
    if (!skipc) load_rows(fr, kb1, 0);
#ifdef OMK_A6_LATEPF
    if (!(OMK_A6_ABL & 4)) { prefetch_kq(chunk_lo(clipc(c + 3))); prefetch_u(chunk_lo(clipc(c + 2))); }
#endif
    OMK_SCHED_FENCE();
    if (!skipc) phase2(fc, 1, tlo);
    OMK_SCHED_FENCE();
    PT6(6);
    { const int t_ = kb0; kb0 = kb1; kb1 = kb2; kb2 = t_; }
  }
#ifdef OMK_PHASE_PROF
  if (prof) {
    pt[10] = clock64_() - t_core0;
    pt[11] = (__builtin_readsteadycounter() - t_ref0) | ((uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 40);   // HW_ID above bit 40
  }
  if (prof && lane == 0)
    for (int i = 0; i < 12; i++) a.prof[wave * 12 + i] = pt[i];
#endif
  if (a.fin && seg == a.nseg - 1) {
    const float extra = a.fin_extra_decay ? expf(dtrow[0] * Ah) : 1.f;
#pragma unroll
    for (int t = 0; t < 8; t++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r;
        a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)su * a.fsu + (int64_t)k * a.fsk] = accS[t][r] * extra;
      }
  }
}

// The column-slice kernel takes the class A scans of head PAIRS that share a group (OMK_SSD_A6=0: the row-strip kernel), split
// sequences included (the zero-start state pass + fold of ssd_mfma_prepare_segments provides the segment start states).
bool ssd_a6_applies(const GScan& g) {
  if (const char* e = getenv("OMK_SSD_A6")) if (e[0] == '0') return false;
  if (g.mode != GS_Y && g.mode != GS_DX) return false;
  if (g.H % 2 != 0 || (g.H / g.G) % 2 != 0) return false;
  if (g.state_only) return false;
  return true;
}

// state-only pass that leaves the state behind the sequence in g.fin (context-parallel shards): the scan proper with final states
// carries exactly this state (hi + lo operand)
int ssd_a6_state_only(const GScan& g, omk_stream stream) {
  GScan q = g; q.state_only = 0;
  if (!g.fin || g.mode != GS_Y || !ssd_a6_applies(q)) return OMK_EUNSUPPORTED;
  GScan a = q;
  a.out = nullptr; a.outx = nullptr; a.Z = Src{}; a.D = nullptr; a.dump = nullptr;
  const SegPlan sp = a.seg ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QA6 - 1) / QA6};
  a.nseg = sp.nseg; a.cps = sp.cps;
  if (a.nseg > 1 && !a.seg_ready) {
    int rc = ssd_mfma_prepare_segments(a, stream);
    if (rc) return rc;
  }
  dim3 grid((unsigned)(a.B * (a.H / 2) * a.nseg)), block(512);
  const size_t smem = sizeof(SmemA6);
  kernels_note("ssd_a6<state_only>");
  if (OMK_SET_MAX_DYN_SMEM((ssd_a6_kernel<GS_Y, false, true, false, true, true>), smem)) return fail(OMK_ELAUNCH, "ssd_a6: cannot raise dynamic LDS to %zu", smem);
  OMK_LAUNCH((ssd_a6_kernel<GS_Y, false, true, false, true, true>), grid, block, smem, stream, a);
  return OMK_OK;
}

// state-only pass over the whole sequence that leaves the window-boundary images in g.dump (no output): the recomputing backward
int ssd_a6_state_dump(const GScan& g, omk_stream stream) {
  if (!g.dump || !ssd_a6_applies(g) || g.mode != GS_Y) return OMK_EUNSUPPORTED;
  GScan a = g;
  a.out = nullptr; a.outx = nullptr; a.Z = Src{}; a.D = nullptr; a.fin = nullptr;
  const SegPlan sp = (a.seg && a.seg_ready) ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QA6 - 1) / QA6};   // start states already folded
  a.nseg = sp.nseg; a.cps = sp.cps;
  dim3 grid((unsigned)(a.B * (a.H / 2) * a.nseg)), block(512);
  const size_t smem = sizeof(SmemA6);
  kernels_note("ssd_a6<state_dump>");
  if (OMK_SET_MAX_DYN_SMEM((ssd_a6_kernel<GS_Y, false, true, true, false, true>), smem)) return fail(OMK_ELAUNCH, "ssd_a6: cannot raise dynamic LDS to %zu", smem);
  OMK_LAUNCH((ssd_a6_kernel<GS_Y, false, true, true, false, true>), grid, block, smem, stream, a);
  return OMK_OK;
}

// called by ssd_mfma_launch after its shape / alignment checks (same preconditions as the row-strip kernel)
int ssd_a6_launch(const GScan& g, omk_stream stream) {
  if (ssd_a8_applies(g)) return ssd_a8_launch(g, stream);
  GScan a = g;
  const SegPlan sp = a.seg ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QA6 - 1) / QA6};
  a.nseg = sp.nseg; a.cps = sp.cps;
  if (a.nseg > 1 && !a.seg_ready) {
    int rc = ssd_mfma_prepare_segments(g, stream);
    if (rc) return rc;
  }
  dim3 grid((unsigned)(a.B * (a.H / 2) * a.nseg)), block(512);
  const size_t smem = sizeof(SmemA6);
  // the scaled U operand of the state update as hi + lo whenever the caller keeps the final state (prefill -> decode hand-off,
  // context-parallel shards): the carried state is then exact to fp32 accumulation (8 more MFMAs per sub-chunk)
  const bool khilo = (a.flags & (GSF_KHILO | GSF_PRECISE)) || a.fin != nullptr;
#define OMK_A6K(MODE_, EX_, DF_, DU_, KH_) do { \
    kernels_note("ssd_a6<mode=%d,ex=%d,dfold=%d,dump=%d,khilo=%d>", (int)MODE_, (int)EX_, (int)DF_, (int)DU_, (int)KH_); \
    if (OMK_SET_MAX_DYN_SMEM((ssd_a6_kernel<MODE_, EX_, DF_, DU_, KH_>), smem)) return fail(OMK_ELAUNCH, "ssd_a6: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_a6_kernel<MODE_, EX_, DF_, DU_, KH_>), grid, block, smem, stream, a); } while (0)
#define OMK_A6(MODE_, EX_, DF_, DU_) do { if (khilo && MODE_ == GS_Y) OMK_A6K(MODE_, EX_, DF_, DU_, (MODE_ == GS_Y)); else OMK_A6K(MODE_, EX_, DF_, DU_, false); } while (0)
  const bool dfold = !a.D || a.Dsp == 0;   // one D per head (or none)
  if (a.mode == GS_Y) {
    const bool ex = a.Z.p || a.outx;
    if (a.dump) { if (ex) return OMK_EUNSUPPORTED; if (dfold) OMK_A6(GS_Y, false, true, true); else OMK_A6(GS_Y, false, false, true); }
    else if (ex) { if (dfold) OMK_A6(GS_Y, true, true, false); else OMK_A6(GS_Y, true, false, false); }
    else { if (dfold) OMK_A6(GS_Y, false, true, false); else OMK_A6(GS_Y, false, false, false); }
  } else if (dfold) {
    if (a.dump) OMK_A6(GS_DX, false, true, true); else OMK_A6(GS_DX, false, true, false);
  } else {
    if (a.dump) OMK_A6(GS_DX, false, false, true); else OMK_A6(GS_DX, false, false, false);
  }
#undef OMK_A6
#undef OMK_A6K
  return OMK_OK;
}

}  // namespace omk
