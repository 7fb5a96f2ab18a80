// ssd_cp.hip -- chunk-parallel half of the SSD backward (round 3): dB, dC, the token scalars of the decay gradient and dD from
// window-boundary states, with the sum over the heads of a group formed ON CHIP.
//
// Why.  The sequential backward scans of ssd_mfma.hip give every head its own workgroup, so dB / dC (sums over the 64 heads that
// share B and C) left the chip as per-head-pair partial tiles (2 x 268 MB written, 2 x 268 MB read back at B 8, L 4096) and the
// decay gradient needed a forward-state checkpoint per chunk (537 MB written, 537 MB read): 3.94 GB of HBM traffic for 0.85 GB of
// algorithmic bytes (profiles/r02_pmc_ssd_final.txt).  Here ONE workgroup owns a 128-token window of one (batch, group) and walks
// the heads of the group, so the head sum lives in MFMA accumulators and dB / dC are written once.  What a head needs from outside
// its window are two 64 x 128 states -- the forward state S_in in front of the window and the adjoint state Gn behind it -- which
// the class A scans leave behind as bf16 images (GScan::dump): the dx scan (which runs anyway) dumps Gn, one state-only forward
// pass dumps S_in.
//
// Per head h and window (local tokens m, s; c = log2 of the cumulative decay inside the window; a_t = exp(dt'_t A)):
//   Z[m][s]   = dy_m . x_s                      (s <= m)        16x16x32 MFMAs, contraction over headdim
//   T[m][s]   = Z[m][s] 2^(c_m - c_s)                              fp32, from exact bf16 products
//   W[m][s]  += dt_s T[m][s]                     summed over the heads in registers;  dC += W B,  dB += W^T C  at the very end
//   e_m       = 2^c_m C_m . (dy_m S_in)  +  sum_s dt_s T[m][s] (C_m . B_s)          = dy_m . (C_m s_m)
//   w_s       = dec 2^(c_end - c_s) B_s . (x_s Gn)  +  sum_m T[m][s] (C_m . B_s)    = x_s . (g_s B_s)
//   dC^T[n][m] += 2^c_m (S_in^T dy_m)[n]         32x32x16 MFMAs, bf16 state images as A operand through transpose reads
//   dB^T[n][s] += dt_s dec 2^(c_end - c_s) (Gn^T x_s)[n]
// e and w are the token scalars of ssd_bwd_finish_par_kernel (ssd.hip): dl_t = sum_{j >= t} (e_j - dt_j w_j) restarted at every
// 64-token boundary from the exact value q = < g, a s > the window can form by itself:
//   q(end of window)  = 2^c_end < Gn, S_in > + sum_s dt_s w_inter[s]
//   q(token 64)       = 2^c_end < Gn, S_in > + sum_{s < 64} dt_s w_inter[s] + sum_{m >= 64} e_inter[m] + sum_{m >= 64 > s} dt_s T (C . B)
// Off-diagonal tiles take the decay as a row factor times a column factor around the tile boundary (both <= 1: no overflow, and an
// underflow only where the exact product underflows too); diagonal tiles evaluate exp2 per element under the causal mask.
#include <cstdlib>

#include <type_traits>
#include "ssd_scan.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int TW = 128;   // window (tokens)

struct SmemCp {
  uint16_t X[TW * 64];      // [t][p], ux3 swizzle          (X | DY hold W as a 128 x 128 bf16 tile after the head loop)
  uint16_t DY[TW * 64];
  uint16_t S[64 * 128];     // forward state in front of the window, raw kx3 image [p][n]
  uint16_t Gt[64 * 128];    // adjoint state behind the window (before the decay of the first token behind it)
  uint16_t Bm[TW * 128];    // [t][n], kx3 swizzle
  uint16_t Cm[TW * 128];
  float c2[2][TW], alpha[2][TW], dts[2][TW], ecm[2][TW], wsc0[2][TW], wsc[2][TW];   // per head, double buffered
  // token-scalar pieces, plain stores into per-producer slots (LDS float atomics cost ~3 cycles per LANE: the first version spent
  // 10 k of its 19 k cycles per head in them): column sums per Phase A tile, row sums per (wave, strip), Phase B halves per n half
  float colA[2][36][16], rowA[2][8][2][16], eI[2][2][TW], wI[2][2][TW], mw[2][8][4];
  float g5[4][64 * 4], w5[4][64 * 4];   // G1 and the W accumulator of the fifth tile of the even waves (lane-linear float4): register diet
  float ddgs[3][8][2], cdr[3][2];   // per-wave dD and < G, S > sums; c_end and dec: rings of three -- written one head ahead, read one head behind
};
static_assert(sizeof(SmemCp) <= 160 * 1024, "one workgroup per CU");

// Phase A tiles of a role (see the kernel): row block, column block, which of the role's two strips (0: strip 7 - p, 1: strip p)
constexpr int cp_tile_m(int rl, int t) { const int p = rl >> 1, q = rl & 1; return q == 0 ? 7 - p : (t < 3 - p ? 7 - p : (t < 4 ? p : -1)); }
constexpr int cp_tile_s(int rl, int t) { const int p = rl >> 1, q = rl & 1; return q == 0 ? t : (t < 3 - p ? 5 + t : (t < 4 ? t - (3 - p) : 0)); }
constexpr int cp_tile_l(int rl, int t) { const int p = rl >> 1, q = rl & 1; return (q == 1 && t >= 3 - p && t < 4) ? 1 : 0; }

__global__ __launch_bounds__(512) void ssd_cp_kernel(CpArgs a) {
  OMK_DYN_SMEM(smem_raw);
  SmemCp& sm = *reinterpret_cast<SmemCp*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform_i(tid >> 6);
  const int t16 = lane & 15, g16 = lane >> 4, l31 = lane & 31, h32 = lane >> 5;
  int vid = blockIdx.x;
  const int hs = vid % a.nhs; vid /= a.nhs;
  const int win = vid % a.nW; vid /= a.nW;
  const int g = vid % a.G, b = vid / a.G;
  const int hpg = a.H / a.G, hps = hpg / a.nhs;
  const int hbeg = g * hpg + hs * hps;
  const int t0 = win * TW;
  const int nT = (a.L + 63) / 64;

  // ---- the window's B / C rows (shared by every head of the group)
  {
    const uint16_t* Bb = a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg;
    const uint16_t* Cb = a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg;
    const BufRes Br = make_buf(Bb, (uint32_t)((int64_t)a.L * a.bsl * 2)), Cr = make_buf(Cb, (uint32_t)((int64_t)a.L * a.csl * 2));
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int q = tid + 512 * i, row = q >> 4, seg = q & 15;
      st16(&sm.Bm[kx3(row, seg * 8)], buf_ld16(Br, 2u * (uint32_t)((t0 + row) * (int)a.bsl + seg * 8), 0u));
      st16(&sm.Cm[kx3(row, seg * 8)], buf_ld16(Cr, 2u * (uint32_t)((t0 + row) * (int)a.csl + seg * 8), 0u));
    }
  }
  // ---- per-head staging: x, dy rows (two 16-byte segments per thread each), the two state images (two each)
  u32x4 rx[2], ry[2], rs[2], rg[2];
  float rd0 = 0.f, rd1 = 0.f, rdn = 0.f, rA = 0.f;
  // Who does the per-head bookkeeping.  Measured per head (tools: OMK_PHASE_PROF build, OMK_CP_PROF=1): the first-dispatched half of
  // the workgroup (waves 0 - 3) wins the SIMD arbitration and waits 3.4 - 4.7 k of 15.4 k cycles at the first barrier, waves 4 - 7 set
  // the pace -- so the scalars of the next head (SW), the restart values and the token readout sit on waves 1, 0 and 2.
  constexpr int SW = 1, RW0 = 0, RW1 = 2, MW = 3;   // scalars, token readout (two halves of the window), restart values
  auto issue_xy = [&](int h) {
    const uint16_t* Xb = a.X + (int64_t)b * a.xsb + (int64_t)h * a.xsh;
    const uint16_t* Yb = a.DY + (int64_t)b * a.ysb + (int64_t)h * a.ysh;
    const BufRes Xr = make_buf(Xb, (uint32_t)((int64_t)a.L * a.xsl * 2)), Yr = make_buf(Yb, (uint32_t)((int64_t)a.L * a.ysl * 2));
    int tq = tid;
    OMK_OPAQUE(tq);   // lane offsets are rebuilt per call: hoisted out of the head loop they would sit in registers across it
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int q = tq + 512 * i, row = q >> 3, seg = q & 7;
      rx[i] = buf_ld16(Xr, 2u * (uint32_t)((t0 + row) * (int)a.xsl + seg * 8), 0u);
      ry[i] = buf_ld16(Yr, 2u * (uint32_t)((t0 + row) * (int)a.ysl + seg * 8), 0u);
    }
    if (w == SW) {   // dt' of the window's tokens and of the first token behind it (zeros past the end of the sequence)
      const BufRes Dr = make_buf(a.dtp + ((int64_t)b * a.H + h) * a.L, (uint32_t)((int64_t)a.L * 4));
      rd0 = buf_ld_f32(Dr, 4u * (uint32_t)(t0 + lane), 0u);
      rd1 = buf_ld_f32(Dr, 4u * (uint32_t)(t0 + 64 + lane), 0u);
      rdn = buf_ld_f32(Dr, 4u * (uint32_t)(t0 + TW), 0u);
      // A of the head rides along: loaded inside scalars() it was one more round trip there, and -- the load counter returns in order --
      // a wait for everything this wave has in flight, the state images of the next head included
      rA = a.A[h];
    }
  };
  auto issue_sg = [&](int h) {   // the two state images: issued behind Phase B (the register peak), in flight during Phase A
    const int64_t slot = (a.ablate & 4) ? 0 : ((((int64_t)b * a.nW + win) * a.H + h) << 13);
    const BufRes Fr = make_buf(a.Sf + slot, 16384u), Gr = make_buf(a.Sg + slot, 16384u);
    int tq = tid;
    OMK_OPAQUE(tq);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      rs[i] = buf_ld16(Fr, 16u * (uint32_t)(tq + 512 * i), 0u);
      rg[i] = buf_ld16(Gr, 16u * (uint32_t)(tq + 512 * i), 0u);
    }
  };
  auto commit = [&](int) {   // registers -> LDS (between the two barriers of a head: nothing but the eight stores)
    int tq = tid;
    OMK_OPAQUE(tq);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int q = tq + 512 * i, row = q >> 3, seg = q & 7;
      st16(&sm.X[ux3(row, seg * 8)], rx[i]);
      st16(&sm.DY[ux3(row, seg * 8)], ry[i]);
      st16(&sm.S[img_off(q)], rs[i]);    // (segment order of the images in memory: ssd_tiles.h)
      st16(&sm.Gt[img_off(q)], rg[i]);
    }
  };
  // dD and < Graw, S_in > of the staged head fall out of the staging registers: per-wave partial sums into plain slots, ahead of
  // the first barrier (in the waiting time of the fast waves), summed by the readout
  auto stage_sums = [&](int ring) {
    float dd = 0.f, gs = 0.f;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int e = 0; e < 4; e++) { dd = dot2_bf16(rx[i][e], ry[i][e], dd); gs = dot2_bf16(rs[i][e], rg[i][e], gs); }
    dd = wave_sum(dd); gs = wave_sum(gs);
    if (lane == 0) { sm.ddgs[ring][w][0] = dd; sm.ddgs[ring][w][1] = gs; }
  };
  auto scalars = [&](int h, int sb, int ring) {   // wave 0; lanes = tokens 0 .. 63 and 64 .. 127 of the window
    const float Ah2 = rA * LOG2E;
    const float c0 = wave_incl_scan_add(rd0 * Ah2);
    const float tot0 = wave_read_lane(c0, 63);
    const float c1 = wave_incl_scan_add(rd1 * Ah2) + tot0;
    const float cend = wave_read_lane(c1, 63);
    // row factor of the off-diagonal tiles: decay from the end of the previous 16-token block
    const int src = (lane & ~15) - 1;
    const float p0 = shfl(c0, src < 0 ? 0 : src), p1 = shfl(c1, src < 0 ? 0 : src);
    const float ref0 = lane < 16 ? 0.f : p0, ref1 = lane < 16 ? tot0 : p1;
    const float dec = exp2_fast(rdn * Ah2);
    sm.c2[sb][lane] = c0; sm.c2[sb][64 + lane] = c1;
    sm.alpha[sb][lane] = exp2_fast(c0 - ref0); sm.alpha[sb][64 + lane] = exp2_fast(c1 - ref1);
    sm.dts[sb][lane] = rd0; sm.dts[sb][64 + lane] = rd1;
    sm.ecm[sb][lane] = exp2_fast(c0); sm.ecm[sb][64 + lane] = exp2_fast(c1);
    const float k0 = dec * exp2_fast(cend - c0), k1 = dec * exp2_fast(cend - c1);
    sm.wsc0[sb][lane] = k0; sm.wsc0[sb][64 + lane] = k1;
    sm.wsc[sb][lane] = rd0 * k0; sm.wsc[sb][64 + lane] = rd1 * k1;
    if (lane == 0) { sm.cdr[ring][0] = cend; sm.cdr[ring][1] = dec; }
  };

  // ---- Phase A tiles of this wave: the 36 lower-triangular 16 x 16 tiles (m block, s block).  Row strips i and 7 - i hold
  // 9 tiles together; waves 2 p and 2 p + 1 share the pair (p, 7 - p): the even wave takes five tiles of strip 7 - p, the odd one
  // the rest of it and strip p -- four or five tiles per wave, at most two strips, so the row sums of a strip stay in registers
  // (roles: waves 0 - 3, the first-dispatched half that wins the SIMD arbitration, take the five-tile roles 0, 2, 4, 6)
  const int rl = w < 4 ? 2 * w : 2 * (w - 4) + 1;
  int tmb[5], tsb[5], tsl[5];
#pragma unroll
  for (int t = 0; t < 5; t++) { tmb[t] = cp_tile_m(rl, t); tsb[t] = cp_tile_s(rl, t); tsl[t] = cp_tile_l(rl, t); }
  const int strip0 = 7 - (rl >> 1), strip1 = rl >> 1;
  const bool has0 = !((rl & 1) && (rl >> 1) == 3), has1 = (rl & 1) != 0;
  f32x4 Wt[4], g1[4];   // tiles 0 .. 3; the fifth tile of the even waves keeps both in LDS (sm.w5 / sm.g5)
  f32x16 dCt[2], dBt[2];
#pragma unroll
  for (int t = 0; t < 4; t++) { Wt[t] = f32x4{0.f, 0.f, 0.f, 0.f}; g1[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  if (!(rl & 1)) *reinterpret_cast<f32x4*>(&sm.w5[rl >> 1][4 * lane]) = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) { dCt[j][r] = 0.f; dBt[j][r] = 0.f; }

  block_sync();   // zeros and B / C rows visible
  issue_xy(hbeg);
  issue_sg(hbeg);
  commit(0);
  stage_sums(0);
  if (w == SW) scalars(hbeg, 0, 0);
  // G1[m][s] = C_m . B_s of this wave's tiles (the same for every head)
#pragma unroll
  for (int t = 0; t < 5; t++) {
    if (tmb[t] < 0) continue;
    const int m0 = 16 * tmb[t], s0 = 16 * tsb[t];
    f32x4 gg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 4; ks++)
      gg = mfma16x16x32_bf16(as_s16x8(ld16(&sm.Cm[kx3(m0 + t16, 32 * ks + 8 * g16)])), as_s16x8(ld16(&sm.Bm[kx3(s0 + t16, 32 * ks + 8 * g16)])), gg);
    if (t < 4) g1[t < 4 ? t : 0] = gg;
    else *reinterpret_cast<f32x4*>(&sm.g5[rl >> 1][4 * lane]) = gg;
  }
  block_sync();   // head 0 staged

  const int mbB = w & 3, nh = w >> 2;
  const int mrow = 32 * mbB + l31;
  // token scalars, restart values and dD of a finished head from the slots its waves filled (buffer sb): runs while the NEXT head
  // is being computed (waves 3 and 5: tokens; lane 0 of the scalar wave, ahead of its own next write of c_end / dec: the restart values), so nothing of it sits between the two barriers of a head
  auto readout = [&](int h, int sb, int ring) {
    if (a.ablate & 64) return;
    const int64_t bh = (int64_t)b * a.H + h;
    if (w == RW0 || w == RW1) {
      int m = (w == RW0 ? 0 : 64) + lane;
      OMK_OPAQUE(m);
      const int st = m >> 4, i = m & 15;
      float ev = sm.eI[sb][0][m] + sm.eI[sb][1][m], wv = sm.wI[sb][0][m] + sm.wI[sb][1][m];
      if (st >= 4) ev += sm.rowA[sb][2 * (7 - st)][0][i] + sm.rowA[sb][2 * (7 - st) + 1][0][i];
      else ev += sm.rowA[sb][2 * st + 1][1][i];
      for (int mb = st; mb < 8; mb++) wv += sm.colA[sb][mb * (mb + 1) / 2 + st][i];
      // (buffer stores: rows past the end of the sequence are dropped by the range check)
      const BufRes Er = make_buf(a.e + bh * a.L, (uint32_t)((int64_t)a.L * 4)), Wr = make_buf(a.wsum + bh * a.L, (uint32_t)((int64_t)a.L * 4));
      buf_st_f32(Er, ev, 4u * (uint32_t)(t0 + m), 0u);
      buf_st_f32(Wr, wv, 4u * (uint32_t)(t0 + m), 0u);
    } else if (w == MW) {   // lanes 0 .. 7 fetch the eight waves' partial sums, three butterfly steps add them up
      const int k = lane & 7;
      f32x4 m4 = *reinterpret_cast<const f32x4*>(&sm.mw[sb][k][0]);
      f32x2 d2 = *reinterpret_cast<const f32x2*>(&sm.ddgs[ring][k][0]);
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) {
#pragma unroll
        for (int i = 0; i < 4; i++) m4[i] += shfl_xor(m4[i], m);
        d2[0] += shfl_xor(d2[0], m); d2[1] += shfl_xor(d2[1], m);
      }
      if (lane == 0) {
        const float qb = exp2_fast(sm.cdr[ring][0]) * sm.cdr[ring][1] * d2[1];
        const float q_end = qb + m4[0], q_mid = qb + m4[1] + m4[2] + m4[3];
        if (2 * win + 1 <= nT) a.bnd[bh * (nT + 1) + 2 * win + 1] = q_mid;
        if (2 * win + 2 <= nT) a.bnd[bh * (nT + 1) + 2 * win + 2] = q_end;
        if (a.dD) atomic_add_f32(a.dD + (int64_t)h * a.dDsh, d2[0]);
      }
    }
  };
#ifdef OMK_PHASE_PROF   // developer build: s_memtime deltas per phase, workgroup 0 (printed by ssd_cp_launch under OMK_CP_PROF=1)
  uint64_t pt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
  uint64_t tprev = prof ? clock64_() : 0;
  const uint64_t t_wall0 = __builtin_readsteadycounter();   // (every workgroup: 100 MHz wall clock of its first / last instruction + XCC_ID, slots behind the phase sums)
#define PTC(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
#else
#define PTC(i) do { } while (0)
#endif
  int r_prev = 2, r_cur = 0, r_next = 1;   // hi - 1, hi, hi + 1 modulo 3
  for (int hi = 0; hi < hps; hi++) {
    const int h = hbeg + hi, sb = hi & 1;
    const bool more = hi + 1 < hps;
    if (more && !(a.ablate & 8) && !(a.ablate & 512)) issue_xy(h + 1);   // next head's rows: in flight during both phases
    if (hi > 0) readout(h - 1, sb ^ 1, r_prev);
    PTC(0);
    // (lane bases of the swizzled tiles; tile row blocks, k steps and column blocks enter as uniform adds / XORs on top of them)
    int oA = ux3(t16, 8 * g16), oB = ux3(mrow, 8 * h32);
    int oT0 = kx3(8 * h32 + (t16 >> 2), 16 * (g16 & 1) + 4 * (t16 & 3)), oT1 = kx3(8 * h32 + (t16 >> 2) + 4, 16 * (g16 & 1) + 4 * (t16 & 3));
    int oC = kx3(mrow, 4 * h32);
    OMK_OPAQUE(oA); OMK_OPAQUE(oB); OMK_OPAQUE(oT0); OMK_OPAQUE(oT1); OMK_OPAQUE(oC);
    // ---- Phase B: inter-window terms; this wave owns rows n of two 32-blocks (2 nh, 2 nh + 1) x tokens 32 mbB ..
    float ep = 0.f, wp = 0.f;
    const float ecm_m = sm.ecm[sb][mrow], wsc_s = sm.wsc[sb][mrow], wsc0_s = sm.wsc0[sb][mrow], dts_s = sm.dts[sb][mrow];
    // (the k step of the row operands is an XOR on the swizzled segment index: four lane addresses shared by dy and x, both tiles)
    const int oB0 = oB, oB1 = oB ^ 16, oB2 = oB ^ 32, oB3 = oB ^ 48;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (a.ablate & 2) continue;
      const int nbx = 32 * (2 * nh + j);   // column block of the state images: an XOR on the swizzled segment index
      // oT0, oT1 < 2048 (rows 0 .. 15 of a 128-column tile): the k step (16 rows = 2048 elements) stays an immediate offset
      const int t0 = oT0 ^ nbx, t1 = oT1 ^ nbx, cb = oC ^ nbx;
      f32x16 acc;
#define OMK_CP_STATE_FRAG(tile, ks) ({ const s16x4 f0_ = lds_read_tr16_b64(&(tile)[t0 + 2048 * (ks)]), f1_ = lds_read_tr16_b64(&(tile)[t1 + 2048 * (ks)]); \
                                       s16x8{f0_[0], f0_[1], f0_[2], f0_[3], f1_[0], f1_[1], f1_[2], f1_[3]}; })
      {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.S, 0), as_s16x8(ld16(&sm.DY[oB0])), zero16);
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.S, 1), as_s16x8(ld16(&sm.DY[oB1])), acc);
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.S, 2), as_s16x8(ld16(&sm.DY[oB2])), acc);
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.S, 3), as_s16x8(ld16(&sm.DY[oB3])), acc);
      }
      // acc[r] = (S_in^T dy_m)[n], n = 32 nb + 8 (r >> 2) + 4 h32 + (r & 3), m = mrow
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32x2 cv = *reinterpret_cast<const u32x2*>(&sm.Cm[cb ^ (8 * q)]);
        const float c4[4] = {bf_lo(cv[0]), bf_hi(cv[0]), bf_lo(cv[1]), bf_hi(cv[1])};
#pragma unroll
        for (int i = 0; i < 4; i++) { ep += acc[4 * q + i] * c4[i]; dCt[j][4 * q + i] += ecm_m * acc[4 * q + i]; }
      }
      {
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.Gt, 0), as_s16x8(ld16(&sm.X[oB0])), zero16);
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.Gt, 1), as_s16x8(ld16(&sm.X[oB1])), acc);
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.Gt, 2), as_s16x8(ld16(&sm.X[oB2])), acc);
        acc = mfma32x32x16_bf16(OMK_CP_STATE_FRAG(sm.Gt, 3), as_s16x8(ld16(&sm.X[oB3])), acc);
      }
#undef OMK_CP_STATE_FRAG
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const u32x2 bv = *reinterpret_cast<const u32x2*>(&sm.Bm[cb ^ (8 * q)]);
        const float b4[4] = {bf_lo(bv[0]), bf_hi(bv[0]), bf_lo(bv[1]), bf_hi(bv[1])};
#pragma unroll
        for (int i = 0; i < 4; i++) { wp += acc[4 * q + i] * b4[i]; dBt[j][4 * q + i] += wsc_s * acc[4 * q + i]; }
      }
    }
    // the two state images of the next head behind Phase B: sixteen 1 KB requests per wave in one burst stall on the address path
    if (more && !(a.ablate & 8) && !(a.ablate & 256)) issue_sg(h + 1);
    PTC(1);
    // ---- Phase A: intra-window terms
    float qm = 0.f;
    float ra0[4] = {0.f, 0.f, 0.f, 0.f}, ra1[4] = {0.f, 0.f, 0.f, 0.f};   // e_intra row sums of the wave's two strips
    float colv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};                              // w_intra column sums of the wave's tiles (this lane's 4 rows)
    // The tile positions of a wave are a function of its role, a run-time value -- but only eight of them exist: phase A is compiled
    // once per role (tile rows / columns, diagonal or not, strip membership, the window-half crossing all constants: immediate LDS
    // offsets and no branch in front of a tile), and the wave jumps to its copy.
    auto phaseA = [&](auto role_c) {
      constexpr int RL = decltype(role_c)::value;
#define OMK_CP_TILE(T_) do {                                                                                                    \
      constexpr int mb_ = cp_tile_m(RL, T_), sb_ = cp_tile_s(RL, T_);                                                            \
      if constexpr (mb_ >= 0) {                                                                                                  \
        constexpr int m0 = 16 * mb_, s0 = 16 * sb_;                                                                              \
        f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                                                          \
        _Pragma("unroll") for (int ks = 0; ks < 2; ks++)                                                                         \
          z = mfma16x16x32_bf16(as_s16x8(ld16(&sm.DY[(oA + 64 * m0) ^ (32 * ks)])), as_s16x8(ld16(&sm.X[(oA + 64 * s0) ^ (32 * ks)])), z); \
        /* z[r] = dy_m . x_s, m = m0 + 4 g16 + r, s = s0 + t16 */                                                               \
        const float css = sm.c2[sb][s0 + t16], dss = sm.dts[sb][s0 + t16];                                                       \
        f32x4 t2;                                                                                                                \
        if constexpr (m0 != s0) {                                                                                                \
          const float beta = exp2_fast(sm.c2[sb][m0 > 0 ? m0 - 1 : 0] - css);                                                    \
          const f32x4 al = *reinterpret_cast<const f32x4*>(&sm.alpha[sb][m0 + 4 * g16]);                                         \
          t2 = z * al * beta;                                                                                                    \
        } else {                                                                                                                 \
          const f32x4 cm = *reinterpret_cast<const f32x4*>(&sm.c2[sb][m0 + 4 * g16]);                                            \
          _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                                        \
            const float arg = cm[r] - css;                                                                                       \
            t2[r] = (4 * g16 + r >= t16) ? z[r] * exp2_fast(arg < 0.f ? arg : 0.f) : 0.f;                                        \
          }                                                                                                                      \
        }                                                                                                                        \
        f32x4 gg;                                                                                                                \
        if constexpr (T_ < 4) { Wt[T_ < 4 ? T_ : 0] += t2 * dss; gg = g1[T_ < 4 ? T_ : 0]; }                                      \
        else {                                                                                                                   \
          f32x4* wp5 = reinterpret_cast<f32x4*>(&sm.w5[RL >> 1][4 * lane]);                                                      \
          *wp5 = *wp5 + t2 * dss;                                                                                                \
          gg = *reinterpret_cast<const f32x4*>(&sm.g5[RL >> 1][4 * lane]);                                                       \
        }                                                                                                                        \
        const f32x4 d = t2 * gg;                                                                                                 \
        colv[T_] = (d[0] + d[1]) + (d[2] + d[3]);               /* w_intra: sum over m (this lane's four rows) */                \
        const f32x4 dv = d * dss;                               /* e_intra: sum over s of dt_s T (C . B), kept per strip */      \
        if constexpr (m0 >= 64 && s0 < 64) qm += (dv[0] + dv[1]) + (dv[2] + dv[3]);                                              \
        if constexpr (cp_tile_l(RL, T_) != 0) { ra1[0] += dv[0]; ra1[1] += dv[1]; ra1[2] += dv[2]; ra1[3] += dv[3]; }            \
        else { ra0[0] += dv[0]; ra0[1] += dv[1]; ra0[2] += dv[2]; ra0[3] += dv[3]; }                                             \
      } } while (0)
      OMK_CP_TILE(0); OMK_CP_TILE(1); OMK_SCHED_FENCE(); OMK_CP_TILE(2); OMK_CP_TILE(3); OMK_SCHED_FENCE(); OMK_CP_TILE(4);   // (pairs: five interleaved tiles do not fit the registers)
#undef OMK_CP_TILE
    };
    if (!(a.ablate & 1)) {
      switch (rl) {
        case 0: phaseA(std::integral_constant<int, 0>{}); break;
        case 1: phaseA(std::integral_constant<int, 1>{}); break;
        case 2: phaseA(std::integral_constant<int, 2>{}); break;
        case 3: phaseA(std::integral_constant<int, 3>{}); break;
        case 4: phaseA(std::integral_constant<int, 4>{}); break;
        case 5: phaseA(std::integral_constant<int, 5>{}); break;
        case 6: phaseA(std::integral_constant<int, 6>{}); break;
        default: phaseA(std::integral_constant<int, 7>{}); break;
      }
    }
    if (!(a.ablate & 1)) {
      // column sums: the other twelve rows of a tile sit in the lanes 16, 32, 48 further on
#pragma unroll
      for (int t = 0; t < 5; t++) colv[t] += shfl_xor(colv[t], 16);
#pragma unroll
      for (int t = 0; t < 5; t++) colv[t] += shfl_xor(colv[t], 32);
#pragma unroll
      for (int t = 0; t < 5; t++)
        if (tmb[t] >= 0 && g16 == 0) sm.colA[sb][tmb[t] * (tmb[t] + 1) / 2 + tsb[t]][t16] = colv[t];
      // row sums of the two strips (zeros where the wave has no tile of a strip: every slot is rewritten for every head)
      row16_sum4(ra0);
      row16_sum4(ra1);
      const float v0 = t16 == 0 ? ra0[0] : (t16 == 1 ? ra0[1] : (t16 == 2 ? ra0[2] : ra0[3]));
      const float v1 = t16 == 0 ? ra1[0] : (t16 == 1 ? ra1[1] : (t16 == 2 ? ra1[2] : ra1[3]));
      if (t16 < 4) { sm.rowA[sb][rl][0][4 * g16 + t16] = v0; sm.rowA[sb][rl][1][4 * g16 + t16] = v1; }
    }
    PTC(2);
    ep += shfl_xor(ep, 32);
    wp += shfl_xor(wp, 32);
    if (!(a.ablate & 128)) {
      const float ei = h32 == 0 ? ecm_m * ep : 0.f, wi = h32 == 0 ? wsc0_s * wp : 0.f;   // this wave's n half
      if (h32 == 0) { sm.eI[sb][nh][mrow] = ei; sm.wI[sb][nh][mrow] = wi; }
      const float dtw = dts_s * wi;
      const float s_all = wave_sum(dtw), s_eh = wave_sum(ei), s_qm = wave_sum(qm);
      if (lane == 0) *reinterpret_cast<f32x4*>(&sm.mw[sb][w][0]) = f32x4{s_all, mbB < 2 ? s_all : 0.f, mbB < 2 ? 0.f : s_eh, s_qm};
    }
    if (more && !(a.ablate & 16)) stage_sums(r_next);
    PTC(3);
    if (w == SW && more && !(a.ablate & 32)) scalars(h + 1, sb ^ 1, r_next);
    PTC(4);
    block_sync();   // every read of this head's tiles is done, its token-scalar slots are complete
    PTC(5);
    if (more && !(a.ablate & 16)) commit(sb ^ 1);
    PTC(6);
    block_sync();   // next head staged
    PTC(7);
    { const int t_ = r_prev; r_prev = r_cur; r_cur = r_next; r_next = t_; }
  }
#ifdef OMK_PHASE_PROF
  if (prof && lane == 0)
    for (int i = 0; i < 8; i++) a.prof[w * 8 + i] = pt[i];
#endif
  readout(hbeg + hps - 1, (hps - 1) & 1, r_prev);

  // ---- W (sum over the heads, fp32 registers) -> bf16 tile [m][s] in LDS;  dC^T += B^T W^T,  dB^T += C^T W
  // (W as a bf16 hi + lo pair: one rounding of W would sit on top of the output rounding of dB / dC -- 1.7e-3 -> 2.3e-3 measured)
  const f32x4 w5r = *reinterpret_cast<const f32x4*>(&sm.w5[rl >> 1][4 * lane]);
  uint16_t* Wl = sm.X;        // 32 KB: X | DY
  uint16_t* Wlo = sm.S;       // 32 KB: S | Gt
  {
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; i++) { st16(&Wl[(tid + 512 * i) * 8], zero4); st16(&Wlo[(tid + 512 * i) * 8], zero4); }
  }
  block_sync();
#pragma unroll
  for (int t = 0; t < 5; t++) {
    if (tmb[t] < 0) continue;
    const int m0 = 16 * tmb[t], s0 = 16 * tsb[t];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float wv = t < 4 ? Wt[t < 4 ? t : 0][r] : w5r[r];
      const uint16_t hi = f32_to_bf16(wv);
      Wl[kx3(m0 + 4 * g16 + r, s0 + t16)] = hi;
      Wlo[kx3(m0 + 4 * g16 + r, s0 + t16)] = f32_to_bf16(wv - bf16_to_f32(hi));
    }
  }
  block_sync();
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int nb = 2 * nh + j;
    for (int ks = 0; ks <= 2 * mbB + 1; ks++) {   // s <= m: s blocks up to the end of this m block
      const s16x8 fb = tr_frag3<true>(sm.Bm, 16 * ks, 32 * nb, lane);
      dCt[j] = mfma32x32x16_bf16(fb, as_s16x8(ld16(&Wl[kx3(mrow, 16 * ks + 8 * h32)])), dCt[j]);
      dCt[j] = mfma32x32x16_bf16(fb, as_s16x8(ld16(&Wlo[kx3(mrow, 16 * ks + 8 * h32)])), dCt[j]);
    }
    for (int ks = 2 * mbB; ks < 8; ks++) {        // m >= s: m blocks from this s block on
      const s16x8 fc = tr_frag3<true>(sm.Cm, 16 * ks, 32 * nb, lane);
      dBt[j] = mfma32x32x16_bf16(fc, tr_frag3<true>(Wl, 16 * ks, 32 * mbB, lane), dBt[j]);
      dBt[j] = mfma32x32x16_bf16(fc, tr_frag3<true>(Wlo, 16 * ks, 32 * mbB, lane), dBt[j]);
    }
  }
  // ---- one head subset (every shape with >= 256 windows): the sums ARE the gradients -- straight out in the gradient's dtype, no fp32
  // partial round trip (2 x 16.8 MB written + read at B 8, L 4096) and no fold launch
  if (a.direct) {
    if (t0 + mrow < a.L) {
      uint16_t* dCp = (uint16_t*)a.dC + (int64_t)b * a.dcsb + (int64_t)(t0 + mrow) * a.dcsl + (int64_t)g * a.dcsg;
      uint16_t* dBp = (uint16_t*)a.dB + (int64_t)b * a.dbsb + (int64_t)(t0 + mrow) * a.dbsl + (int64_t)g * a.dbsg;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int n = 32 * (2 * nh + j) + 8 * q + 4 * h32;
          *reinterpret_cast<u32x2*>(dCp + n) = u32x2{pack_bf16x2(dCt[j][4 * q], dCt[j][4 * q + 1]), pack_bf16x2(dCt[j][4 * q + 2], dCt[j][4 * q + 3])};
          *reinterpret_cast<u32x2*>(dBp + n) = u32x2{pack_bf16x2(dBt[j][4 * q], dBt[j][4 * q + 1]), pack_bf16x2(dBt[j][4 * q + 2], dBt[j][4 * q + 3])};
        }
    }
#ifdef OMK_PHASE_PROF
    if (a.prof != nullptr && threadIdx.x == 0) {
      a.prof[64 + 2 * blockIdx.x] = t_wall0;
      a.prof[64 + 2 * blockIdx.x + 1] = __builtin_readsteadycounter() | ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 60);
    }
#endif
    return;
  }
  // ---- fp32 partials of this head subset: [hs][b][t][g][n]
  if (t0 + mrow < a.L) {
    const int64_t row = ((((int64_t)hs * a.B + b) * a.L + t0 + mrow) * a.G + g) * 128;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int n = 32 * (2 * nh + j) + 8 * q + 4 * h32;
        *reinterpret_cast<f32x4*>(a.pC + row + n) = f32x4{dCt[j][4 * q], dCt[j][4 * q + 1], dCt[j][4 * q + 2], dCt[j][4 * q + 3]};
        *reinterpret_cast<f32x4*>(a.pB + row + n) = f32x4{dBt[j][4 * q], dBt[j][4 * q + 1], dBt[j][4 * q + 2], dBt[j][4 * q + 3]};
      }
  }
#ifdef OMK_PHASE_PROF
  if (a.prof != nullptr && threadIdx.x == 0) {
    a.prof[64 + 2 * blockIdx.x] = t_wall0;
    a.prof[64 + 2 * blockIdx.x + 1] = __builtin_readsteadycounter() | ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 60);
  }
#endif
}

// out[b][t][g][n] = sum over the head subsets of part[hs][b][t][g][n], in the gradient's dtype; one thread = 8 consecutive n.
// One launch folds both gradients (blockIdx.y: 0 = dC, 1 = dB): two 6 us launches of a latency-bound kernel were 2 % of the backward.
struct FoldOne { const float* part; void* out; int64_t osb, osl, osg; int out_dt; };
__global__ void ssd_cp_fold_kernel(FoldOne f0, FoldOne f1, int B, int L, int G, int nhs) {
  const FoldOne& f = blockIdx.y == 0 ? f0 : f1;
  const float* part = f.part;
  void* out = f.out;
  const int64_t osb = f.osb, osl = f.osl, osg = f.osg;
  const int out_dt = f.out_dt;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * L * G * 16;
  if (i >= total) return;
  const int n8 = (int)(i % 16) * 8;
  const int64_t blg = i / 16;
  const int g = (int)(blg % G), t = (int)((blg / G) % L), b = (int)(blg / ((int64_t)G * L));
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nhs; s++) {
    const float* p = part + ((int64_t)s * B * L * G + blg) * 128 + n8;
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; e++) { acc[e] += v0[e]; acc[4 + e] += v1[e]; }
  }
  const int64_t o = (int64_t)b * osb + (int64_t)t * osl + (int64_t)g * osg + n8;
  if (out_dt == OMK_BF16 && ((o & 7) == 0) && (((uintptr_t)out & 15) == 0)) {
    u32x4 pv;
#pragma unroll
    for (int e = 0; e < 4; e++) pv[e] = pack_bf16x2(acc[2 * e], acc[2 * e + 1]);
    st16((uint16_t*)out + o, pv);
  } else {
#pragma unroll
    for (int e = 0; e < 8; e++) store_rt(out, o + e, out_dt, acc[e]);
  }
}

// few windows (short batch x sequence): split the heads of a group over several workgroups so that the chip fills
int ssd_cp_heads_split(int B, int L, int H, int G) {
  const int nW = (L + TW - 1) / TW, hpg = H / G;
  const int64_t base = (int64_t)B * G * nW;
  int nhs = 1;
  while (base * nhs < 256 && hpg % (nhs * 2) == 0 && hpg / (nhs * 2) >= 4) nhs *= 2;
  return nhs;
}

// one head subset, bf16 gradients on 8-byte aligned rows: the kernel stores dB / dC itself (no fp32 partials, no fold launch)
bool ssd_cp_direct(const CpArgs& a) {
  const bool al = (((uintptr_t)a.dB | (uintptr_t)a.dC) & 7) == 0 && ((a.dbsb | a.dbsl | a.dbsg | a.dcsb | a.dcsl | a.dcsg) & 3) == 0;
  const char* de = getenv("OMK_CP_DIRECT");
  return a.nhs == 1 && a.dB_dt == OMK_BF16 && a.dC_dt == OMK_BF16 && al && !(de && de[0] == '0');
}

int ssd_cp_launch(const CpArgs& a0, omk_stream stream) {
  CpArgs a = a0;
  if (const char* e = getenv("OMK_CP_ABLATE")) a.ablate = atoi(e);
  // 8-byte stores of four bf16: rows and group blocks of dB / dC 8-byte aligned (OMK_CP_DIRECT=0: the partial buffers + fold launch)
  a.direct = ssd_cp_direct(a) ? 1 : 0;
  kernels_note("ssd_cp<direct=%d,nhs=%d>", a.direct, a.nhs);
  const size_t smem = sizeof(SmemCp);
  if (OMK_SET_MAX_DYN_SMEM(ssd_cp_kernel, smem)) return fail(OMK_ELAUNCH, "ssd_cp: cannot raise dynamic LDS to %zu", smem);
  dim3 grid((unsigned)((int64_t)a.B * a.G * a.nW * a.nhs)), block(512);
#ifdef OMK_PHASE_PROF
  static unsigned long long* dprof = nullptr;
  const bool want = getenv("OMK_CP_PROF") != nullptr;
  if (want && !dprof) (void)hipMalloc((void**)&dprof, (64 + 2 * 8192) * sizeof(unsigned long long));
  a.prof = want ? dprof : nullptr;
#endif
  OMK_LAUNCH(ssd_cp_kernel, grid, block, smem, stream, a);
#ifdef OMK_PHASE_PROF
  if (want) {
    unsigned long long hp[64];
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hp, dprof, sizeof(hp), hipMemcpyDeviceToHost);
    const int hps = a.H / a.G / a.nhs;
    fprintf(stderr, "ssd_cp cycles per head and wave:   top+readout    phaseB    phaseA      tail   scalars     wait1    commit     wait2     total\n");
    for (int w = 0; w < 8; w++) {
      unsigned long long tot = 0;
      fprintf(stderr, "wave %d:                           ", w);
      for (int i = 0; i < 8; i++) { fprintf(stderr, "%10llu", hp[w * 8 + i] / hps); tot += hp[w * 8 + i]; }
      fprintf(stderr, "%10llu\n", tot / hps);
    }
    // when did the workgroups finish?  (the launch lasts as long as its slowest one)
    const int nwg = (int)grid.x < 8192 ? (int)grid.x : 8192;
    unsigned long long* st = (unsigned long long*)malloc((size_t)nwg * 16);
    (void)hipMemcpy(st, dprof + 64, (size_t)nwg * 16, hipMemcpyDeviceToHost);
    const unsigned long long mask = (1ull << 60) - 1;
    unsigned long long t0 = ~0ull, t1 = 0; double busy = 0, fx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mx[8] = {0, 0, 0, 0, 0, 0, 0, 0}; int nx[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nwg; i++) t0 = st[2 * i] < t0 ? st[2 * i] : t0;
    double emin = 1e30;
    for (int i = 0; i < nwg; i++) {
      const unsigned long long e = st[2 * i + 1] & mask; const int xc = (int)(st[2 * i + 1] >> 60) & 7;
      t1 = e > t1 ? e : t1; busy += (double)(e - st[2 * i]);
      const double ef = (double)(e - t0) * 0.01; emin = ef < emin ? ef : emin; fx[xc] += ef; nx[xc]++; mx[xc] = ef > mx[xc] ? ef : mx[xc];
    }
    fprintf(stderr, "ssd_cp workgroups: %d, finish min %.1f max %.1f us, mean busy %.1f us = %.2f of the launch; per XCC mean / max finish:", nwg, emin, (double)(t1 - t0) * 0.01,
            busy / nwg * 0.01, busy / nwg / (double)(t1 - t0));
    for (int xc = 0; xc < 8; xc++) if (nx[xc]) fprintf(stderr, "  %d: %.0f / %.0f", xc, fx[xc] / nx[xc], mx[xc]);
    fprintf(stderr, "\n");
    free(st);
  }
#endif
  if (a.direct) return OMK_OK;   // (written by the kernel itself)
  const int64_t total = (int64_t)a.B * a.L * a.G * 16;
  dim3 fgrid((unsigned)((total + 255) / 256), 2u), fblock(256);
  const FoldOne fC = {(const float*)a.pC, a.dC, a.dcsb, a.dcsl, a.dcsg, a.dC_dt}, fB = {(const float*)a.pB, a.dB, a.dbsb, a.dbsl, a.dbsg, a.dB_dt};
  OMK_LAUNCH(ssd_cp_fold_kernel, fgrid, fblock, 0, stream, fC, fB, a.B, a.L, a.G, a.nhs);
  return OMK_OK;
}

}  // namespace omk
