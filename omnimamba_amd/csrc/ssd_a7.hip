// ssd_a7.hip -- EXPERIMENT (OMK_SSD_A7=1, not the default): the column-slice scan of ssd_a6.hip with ONE wave per SIMD.
//
// Why it was tried.  ssd_a6.hip runs two waves per SIMD with 16 state columns each; every wave reads ALL of Q (row fragments) and ALL of
// K^T (transposed fragments) of a sub-chunk from LDS for its 16 columns, packs and decays its own slice, and the two streams of a SIMD
// hardly overlap.  Here a workgroup is 4 waves (256 threads, up to 512 registers per lane), a wave owns 32 state columns
// (two groups of 16) of one head: the Q / K^T fragments, the M tiles and the token scalars are read once per 32 columns (LDS reads per
// token - 40 %), there is one instruction stream per SIMD to schedule, and the plain-VALU / LDS work of one column group can sit in the
// shadow of the other group's MFMAs (tools/ubench/issue_overlap.hip).  Same LDS layout, staging scheme, tile sharing and arithmetic as
// ssd_a6.hip (same results bit for bit); only the plain variants: one D per head, no gate / pre-gate copy, unsplit sequences.
#include <cstdlib>
#include "ssd_scan.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int QA7 = 64;    // tokens staged per barrier
struct SmemA7 {            // (the layout of SmemA6)
  uint16_t K[3][QA7 * 128];
  uint16_t Q[3][QA7 * 128];
  uint16_t U[2][2][QA7 * 64];
  u32x4 M[2][2][6][64];
  float rl[3][2][QA7], ws[3][2][QA7];
  float dec[3][2][2];
  float rfd[2][2][QA7], cfd[2][2][QA7], fo[2][2][QA7];
  int wide[2][2];
};
static_assert(sizeof(SmemA7) <= 160 * 1024, "one workgroup per CU");

template <int MODE, bool DUMP, bool KHILO>
__global__ __launch_bounds__(256) void ssd_a7_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemA7& sm = *reinterpret_cast<SmemA7*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_i(tid >> 6);
  const int hh = wave >> 1, w = wave & 1;
  const int g16 = lane >> 4, t16 = lane & 15;
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous (batch, pair)
  const int pairs = a.H >> 1;
  const int hp = vid % pairs, b = vid / pairs;
  const int h = 2 * hp + hh;
  const int g = (2 * hp) / (a.H / a.G);
  const int nC = (a.L + QA7 - 1) / QA7;
  const int c0 = 0, c1 = nC;
  const bool rev = a.reverse != 0;
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QA7; };
  auto clipc = [&](int c) -> int { return c < c1 ? c : c1 - 1; };
  auto rowtok = [&](int i) -> int { return rev ? QA7 - 1 - i : i; };

  // ---- staging: K, Q four 16-byte segments per thread (rows rowk + 16 r), U of the wave's own head four (rows rowu + 16 r)
  const int rowk = tid >> 4, ck8 = (tid & 15) * 8, rowu = (tid & 127) >> 3, cu8 = (tid & 7) * 8;
  const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
  const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
  const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  const int ksl = (int)a.K.sl, qsl = (int)a.Q.sl, usl = (int)a.U.sl, osl = (int)a.osl;
  const BufRes Kr = make_buf(Kb, (uint32_t)((int64_t)a.L * ksl * 2)), Qr = make_buf(Qb, (uint32_t)((int64_t)a.L * qsl * 2));
  const BufRes Ur = make_buf(Ub, (uint32_t)((int64_t)a.L * usl * 2)), Dr = make_buf(dtrow, (uint32_t)((int64_t)a.L * 4));
  const uint32_t kvo = 2u * (uint32_t)((rev ? 15 - rowk : rowk) * ksl + ck8), qvo = 2u * (uint32_t)((rev ? 15 - rowk : rowk) * qsl + ck8);
  const uint32_t uvo = 2u * (uint32_t)((rev ? 15 - rowu : rowu) * usl + cu8);
  const uint32_t dvo = 4u * (uint32_t)rowtok(lane), dvo_a = 4u * (uint32_t)(rowtok(lane) + (rev ? 1 : 0));
  u32x4 rk[4], rq[4], ru[4];
  float rdt = 0.f, rda = 0.f, rwv = 0.f;
  int stlo = 0;
  auto prefetch_kq = [&](int tl) {
    stlo = tl;
    const uint32_t sk = 2u * (uint32_t)(tl * ksl), sq = 2u * (uint32_t)(tl * qsl);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int ro = rev ? 16 * (3 - r) : 16 * r;
      rk[r] = buf_ld16(Kr, kvo, sk + 2u * (uint32_t)(ro * ksl));
      rq[r] = buf_ld16(Qr, qvo, sq + 2u * (uint32_t)(ro * qsl));
    }
    rdt = buf_ld_f32(Dr, dvo, 4u * (uint32_t)tl);
    rda = buf_ld_f32(Dr, dvo_a, 4u * (uint32_t)tl);
  };
  auto prefetch_u = [&](int tl) {
    const uint32_t su_ = 2u * (uint32_t)(tl * usl);
#pragma unroll
    for (int r = 0; r < 4; r++) ru[r] = buf_ld16(Ur, uvo, su_ + 2u * (uint32_t)((rev ? 16 * (3 - r) : 16 * r) * usl));
  };
  const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
  auto commit_kq = [&](int kb) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      st16(&sm.K[kb][o_ck + 16 * 128 * r], rk[r]);
      st16(&sm.Q[kb][o_ck + 16 * 128 * r], rq[r]);
    }
  };
  auto commit_u = [&](int ub) {
#pragma unroll
    for (int r = 0; r < 4; r++) st16(&sm.U[ub][hh][o_cu + 16 * 64 * r], ru[r]);
  };
  const float Ah2 = a.A[h] * LOG2E;
  auto scalars = [&](int kb, int mb) {   // waves with w == 0 (see ssd_a6.hip: the same scalars, lazy decay and factored tile decay)
    {
      const int t = stlo + rowtok(lane);
      const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
      rwv = okd ? (a.w_is_dt ? rdt : 1.f) : 0.f;
      rdt = okd ? rdt : 0.f;
      rda = oka ? rda : 0.f;
    }
    const float cs = wave_incl_scan_add(rda * Ah2);
    const float e31 = wave_read_lane(cs, 31), e63 = wave_read_lane(cs, 63);
    const float csb = lane < 32 ? 0.f : e31;
    const float rsc = MODE == GS_DX ? rdt : 1.f;
    const bool lazy = e31 > -60.f;
    const float csb_r = lazy ? 0.f : csb, cse_w = lane < 32 ? (lazy ? 0.f : e31) : e63;
    sm.rl[kb][hh][lane] = exp2_fast(cs - csb_r) * rsc;
    sm.ws[kb][hh][lane] = rwv * exp2_fast(cse_w - cs);
    if ((lane & 31) == 31) sm.dec[kb][hh][lane >> 5] = lane < 32 ? (lazy ? 1.f : exp2_fast(e31)) : exp2_fast(lazy ? e63 : e63 - e31);
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
  int o_rd[4], o_kt[4], o_uf[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    o_rd[i] = kx3(t16, 32 * i + 8 * g16);
    o_kt[i] = kx3(4 * g16 + (t16 >> 2), 32 * i + 8 * (t16 & 3));
  }
#pragma unroll
  for (int c = 0; c < 2; c++) o_uf[c] = ux3(4 * g16 + (t16 >> 2), 32 * w + 16 * c + 4 * (t16 & 3));

  // ---- running state: two column groups of eight 16 x 16 tiles (ssd_a6.hip header for the (tile, register) <-> k map)
  f32x4 accS[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int t = 0; t < 8; t++) accS[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (a.init) {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r, su = 32 * w + 16 * c + t16;
          accS[c][t][r] = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)su * a.isu + (int64_t)k * a.isk, a.init_dt);
        }
  }

  // ---- M tiles of one chunk, G shared by the two heads.  Roles: wave (hh, 0) computes the token scalars of head hh and builds tile (0, 0)
  // of sub-chunk jj = hh; wave (hh, 1) builds the tiles (1, 0) and (1, 1) of sub-chunk hh (one set of Q rows, two of K rows)
  float Dh2[2] = {0.f, 0.f};
  if (a.D) {
    Dh2[0] = load_rt(a.D, (int64_t)(2 * hp) * a.Dsh, a.D_dt);
    Dh2[1] = load_rt(a.D, (int64_t)(2 * hp + 1) * a.Dsh, a.D_dt);
  }
  struct FragB { u32x4 k[2][4], q[4]; };
  const int bjj = hh;
  const int brb = 32 * bjj + (w == 0 ? 0 : 16);   // Q rows (l) of the wave's tiles
  auto build_loads = [&](FragB& f, int kb) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      f.q[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * brb]);
      f.k[0][i] = ld16(&sm.K[kb][o_rd[i] + 128 * (32 * bjj)]);
      if (w == 1) f.k[1][i] = ld16(&sm.K[kb][o_rd[i] + 128 * (32 * bjj + 16)]);
    }
  };
  auto build_tiles = [&](const FragB& f, int mb) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (j == 1 && w == 0) break;
      const int tt = w == 0 ? 0 : 1 + j;
      const bool diag = tt != 1;
      const int cb = 32 * bjj + (tt == 2 ? 16 : 0);
      float rf2[2];
      f32x4 cf2[2];
      int wide2[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        const float* rfa = diag ? sm.rfd[mb][h2] : sm.fo[mb][h2];
        const float* cfa = diag ? sm.cfd[mb][h2] : sm.fo[mb][h2];
        rf2[h2] = rfa[brb + t16];
        cf2[h2] = *reinterpret_cast<const f32x4*>(&cfa[cb + 4 * g16]);
        wide2[h2] = sm.wide[mb][h2];
      }
      f32x4 gt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; i++) gt = mfma16x16x32_bf16(as_s16x8(f.k[j][i]), as_s16x8(f.q[i]), gt);
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        float v[4];
        if (uniform_i(wide2[h2]) != 0) {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = gt[r] * exp2_fast(rf2[h2] + cf2[h2][r]);
        } else {
          const f32x4 gc = gt * cf2[h2] * rf2[h2];
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = gc[r];
        }
        if (diag) {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = (4 * g16 + r < t16) ? v[r] : (4 * g16 + r == t16 ? v[r] + Dh2[h2] : 0.f);
        }
        uint32_t hi[2], lo[2];
#pragma unroll
        for (int p2 = 0; p2 < 2; p2++) {
          hi[p2] = pack_bf16x2(v[2 * p2], v[2 * p2 + 1]);
          lo[p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi[p2]), v[2 * p2 + 1] - bf_hi(hi[p2]));
        }
        if (tt == 0) sm.M[mb][h2][3 * bjj][lane] = u32x4{hi[0], hi[1], lo[0], lo[1]};
        else {
          uint32_t* mh = reinterpret_cast<uint32_t*>(&sm.M[mb][h2][3 * bjj + 1][lane]) + 2 * (tt - 1);
          *reinterpret_cast<u32x2*>(mh) = u32x2{hi[0], hi[1]};
          *reinterpret_cast<u32x2*>(mh + 4 * 64) = u32x2{lo[0], lo[1]};
        }
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
  build_loads(fb, 0);
  build_tiles(fb, 0);
  prefetch_kq(chunk_lo(clipc(c0 + 2)));
  prefetch_u(chunk_lo(clipc(c0 + 1)));
  block_sync();
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
  const BufRes Or = make_buf(ob, (uint32_t)((int64_t)a.L * osl * 2));

  struct FragR { u32x4 q0[4], q1[4]; };
  struct FragC { s16x4 u0[2], u1[2], kt[8][2]; float rl0, rl1; f32x4 ws0, ws1; float dec; u32x4 m0, mh, ml; };
  auto load_rows = [&](FragR& f, int kb, int jj) {
#pragma unroll
    for (int i = 0; i < 4; i++) f.q0[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * (32 * jj)]);
#pragma unroll
    for (int i = 0; i < 4; i++) f.q1[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * (32 * jj + 16)]);
  };
  auto load_cols = [&](FragC& f, int kb, int ub, int jj, int grp) {   // five groups, as in ssd_a6.hip
    const int r0 = 32 * jj;
    if (grp == 0) {
      f.ws0 = *reinterpret_cast<const f32x4*>(&sm.ws[kb][hh][r0 + 4 * g16]);
      f.ws1 = *reinterpret_cast<const f32x4*>(&sm.ws[kb][hh][r0 + 16 + 4 * g16]);
#pragma unroll
      for (int c = 0; c < 2; c++) {
        f.u0[c] = lds_read_tr16_b64(&sm.U[ub][hh][o_uf[c] + 64 * r0]);
        f.u1[c] = lds_read_tr16_b64(&sm.U[ub][hh][o_uf[c] + 64 * (r0 + 16)]);
      }
      f.dec = sm.dec[kb][hh][jj];
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      if ((t >> 1) != (grp == 0 ? 0 : grp == 1 ? -1 : grp - 1)) continue;
      f.kt[t][0] = lds_read_tr16_b64(&sm.K[kb][o_kt[t >> 1] + 4 * (t & 1) + 128 * r0]);
      f.kt[t][1] = lds_read_tr16_b64(&sm.K[kb][o_kt[t >> 1] + 4 * (t & 1) + 128 * (r0 + 16)]);
    }
    if (grp == 1) {
      f.m0 = sm.M[ub][hh][3 * jj][lane];
      f.mh = sm.M[ub][hh][3 * jj + 1][lane];
      f.ml = sm.M[ub][hh][3 * jj + 2][lane];
    }
    if (grp == 4) {
      f.rl0 = sm.rl[kb][hh][r0 + t16];
      f.rl1 = sm.rl[kb][hh][r0 + 16 + t16];
    }
  };
  // window-state images: segment (4 wq + i) 64 + lane of ssd_tiles.h (img_off), wq = 2 w + c the 16-column group
  const uint32_t dump_nb = (DUMP && a.dump) ? (uint32_t)((((int64_t)a.dump_nw - 1) * a.H + 1) << 14) : 0u;
  const BufRes Pr = make_buf((DUMP && a.dump) ? a.dump + ((((int64_t)b * a.dump_nw) * a.H + h) << 13) : nullptr, dump_nb);
  const uint32_t pvo = 16u * (uint32_t)(512 * w + lane);
  f32x4 accA0[2], accA1[2];
  auto phase1 = [&](const FragR& f, bool dump_slot, bool dump_here, uint32_t dso, FragC& nf, int nkb, int nub, int njj) {
#pragma unroll
    for (int c = 0; c < 2; c++) { accA0[c] = f32x4{0.f, 0.f, 0.f, 0.f}; accA1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
      for (int c = 0; c < 2; c++) {
        u32x4 sp;
        sp[0] = pack_bf16x2(accS[c][2 * i][0], accS[c][2 * i][1]);
        sp[1] = pack_bf16x2(accS[c][2 * i][2], accS[c][2 * i][3]);
        sp[2] = pack_bf16x2(accS[c][2 * i + 1][0], accS[c][2 * i + 1][1]);
        sp[3] = pack_bf16x2(accS[c][2 * i + 1][2], accS[c][2 * i + 1][3]);
        if (DUMP && dump_slot && dump_here) buf_st16(Pr, sp, pvo + 4096u * (uint32_t)c + 1024u * (uint32_t)i, dso);
        accA0[c] = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q0[i]), accA0[c]);
        accA1[c] = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q1[i]), accA1[c]);
      }
      OMK_SCHED_FENCE();
      load_cols(nf, nkb, nub, njj, i + 1);
      OMK_SCHED_FENCE();
    }
  };
  auto out_rows = [&](f32x4 o, int row, int tlo, int c) {   // the lane's row, columns 32 w + 16 c + 4 g16 + r
    const uint32_t eoff = (uint32_t)(rowtok(row) * osl + 32 * w + 16 * c + 4 * g16);
    const u32x2 ov = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
    buf_st8(Or, ov, 2u * eoff, 2u * (uint32_t)(tlo * osl));
  };
  auto phase2 = [&](const FragC& f, int jj, int tlo) {
    u32x4 uh[2], ul[2];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const s16x4& uf = s2 ? f.u1[c] : f.u0[c];
        const f32x4& ws4 = s2 ? f.ws1 : f.ws0;
        float us[4];
#pragma unroll
        for (int e = 0; e < 4; e++) us[e] = bf16_to_f32((uint16_t)uf[e]) * ws4[e];
#pragma unroll
        for (int p2 = 0; p2 < 2; p2++) {
          const uint32_t hi = pack_bf16x2(us[2 * p2], us[2 * p2 + 1]);
          uh[c][2 * s2 + p2] = hi;
          if (KHILO) ul[c][2 * s2 + p2] = pack_bf16x2(us[2 * p2] - bf_lo(hi), us[2 * p2 + 1] - bf_hi(hi));
        }
      }
    if (uniform_i((int)__builtin_bit_cast(uint32_t, f.dec)) != 0x3f800000) {
#pragma unroll
      for (int c = 0; c < 2; c++)
#pragma unroll
        for (int t = 0; t < 8; t++) accS[c][t] = accS[c][t] * f.dec;
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
      s16x8 kk;
      kk[0] = f.kt[t][0][0]; kk[1] = f.kt[t][0][1]; kk[2] = f.kt[t][0][2]; kk[3] = f.kt[t][0][3];
      kk[4] = f.kt[t][1][0]; kk[5] = f.kt[t][1][1]; kk[6] = f.kt[t][1][2]; kk[7] = f.kt[t][1][3];
#pragma unroll
      for (int c = 0; c < 2; c++) {
        accS[c][t] = mfma16x16x32_bf16(kk, as_s16x8(uh[c]), accS[c][t]);
        if (KHILO) accS[c][t] = mfma16x16x32_bf16(kk, as_s16x8(ul[c]), accS[c][t]);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      s16x8 u00, u01;
      u00[0] = f.u0[c][0]; u00[1] = f.u0[c][1]; u00[2] = f.u0[c][2]; u00[3] = f.u0[c][3]; u00[4] = f.u0[c][0]; u00[5] = f.u0[c][1]; u00[6] = f.u0[c][2]; u00[7] = f.u0[c][3];
      u01[0] = f.u0[c][0]; u01[1] = f.u0[c][1]; u01[2] = f.u0[c][2]; u01[3] = f.u0[c][3]; u01[4] = f.u1[c][0]; u01[5] = f.u1[c][1]; u01[6] = f.u1[c][2]; u01[7] = f.u1[c][3];
      const f32x4 accB0 = mfma16x16x32_bf16(u00, as_s16x8(f.m0), f32x4{0.f, 0.f, 0.f, 0.f});
      f32x4 accB1 = mfma16x16x32_bf16(u01, as_s16x8(f.mh), f32x4{0.f, 0.f, 0.f, 0.f});
      accB1 = mfma16x16x32_bf16(u01, as_s16x8(f.ml), accB1);
      out_rows(accA0[c] * f.rl0 + accB0, 32 * jj + t16, tlo, c);
      out_rows(accA1[c] * f.rl1 + accB1, 32 * jj + 16 + t16, tlo, c);
    }
  };

#ifdef OMK_PHASE_PROF   // developer build (tools/phase_prof_a6.py with OMK_SSD_A7=1): s_memtime deltas per phase, workgroup 0
  uint64_t pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
#define PT7(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
  uint64_t tprev = prof ? clock64_() : 0;
  const uint64_t t_core0 = tprev, t_ref0 = prof ? __builtin_readsteadycounter() : 0;
#else
#define PT7(i) do { } while (0)
#endif
  FragR fr;
  FragC fc;
  load_rows(fr, 0, 0);
  int kb0 = 0, kb1 = 1, kb2 = 2;
  for (int c = c0; c < c1; c++) {
    const int ub0 = (c - c0) & 1, ub1 = ub0 ^ 1;
    const int tlo = chunk_lo(c);
    bool dump_here = false;
    uint32_t dso = dump_nb;
    if (DUMP && a.dump) {
      const int cid = rev ? nC - 1 - c : c;
      dump_here = rev ? (cid == nC - 1 || (cid & 1)) : !(cid & 1);
      if (dump_here) dso = (uint32_t)(((int64_t)(cid >> 1) * a.H) << 14);
    }
    // ---- sub-chunk 0
    load_cols(fc, kb0, ub0, 0, 0);
    OMK_SCHED_FENCE();
    phase1(fr, true, dump_here, dso, fc, kb0, ub0, 0);
    OMK_SCHED_FENCE();
    if (c + 1 < c1) build_loads(fb, kb1);
    OMK_SCHED_FENCE();
    phase2(fc, 0, tlo);
    OMK_SCHED_FENCE();
    load_rows(fr, kb0, 1);
    PT7(0);
    if (c + 1 < c1) build_tiles(fb, ub1);
    PT7(1);
    commit_kq(kb2);
    commit_u(ub1);
    PT7(2);
    if (w == 0) scalars(kb2, ub0);
    PT7(3);
    prefetch_kq(chunk_lo(clipc(c + 3)));
    prefetch_u(chunk_lo(clipc(c + 2)));
    OMK_SCHED_FENCE();
    // ---- sub-chunk 1
    load_cols(fc, kb0, ub0, 1, 0);
    OMK_SCHED_FENCE();
    phase1(fr, false, false, dump_nb, fc, kb0, ub0, 1);
    OMK_SCHED_FENCE();
    PT7(4);
    block_sync();
    PT7(5);
    load_rows(fr, kb1, 0);
    OMK_SCHED_FENCE();
    phase2(fc, 1, tlo);
    OMK_SCHED_FENCE();
    PT7(6);
    { const int t_ = kb0; kb0 = kb1; kb1 = kb2; kb2 = t_; }
  }
#ifdef OMK_PHASE_PROF
  if (prof) {
    pt[10] = clock64_() - t_core0;
    pt[11] = (__builtin_readsteadycounter() - t_ref0) | ((uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 40);
  }
  if (prof && lane == 0)
    for (int i = 0; i < 12; i++) a.prof[wave * 12 + i] = pt[i];
#endif
  if (a.fin) {
    const float extra = a.fin_extra_decay ? expf(dtrow[0] * a.A[h]) : 1.f;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r, su = 32 * w + 16 * c + t16;
          a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)su * a.fsu + (int64_t)k * a.fsk] = accS[c][t][r] * extra;
        }
  }
}

// OMK_SSD_A7=1: the plain class A scans of unsplit sequences (one D per head or none, no gate, no pre-gate copy)
bool ssd_a7_applies(const GScan& g) {
  const char* e = getenv("OMK_SSD_A7");
  if (!e || e[0] != '1') return false;
  if (g.mode != GS_Y && g.mode != GS_DX) return false;
  if (g.H % 2 != 0 || (g.H / g.G) % 2 != 0 || g.state_only) return false;
  if (g.Z.p || g.outx || (g.D && g.Dsp != 0)) return false;
  if (g.seg && ssd_segments(g.B * g.H, g.L).nseg > 1) return false;
  return true;
}

int ssd_a7_launch(const GScan& g, omk_stream stream) {
  GScan a = g;
  a.nseg = 1; a.cps = (a.L + QA7 - 1) / QA7;
  dim3 grid((unsigned)(a.B * (a.H / 2))), block(256);
  const size_t smem = sizeof(SmemA7);
  const char* khe = getenv("OMK_SSD_KHILO");
  const bool khilo = a.mode == GS_Y && (khe ? khe[0] == '1' : (a.fin != nullptr));
#define OMK_A7K(MODE_, DU_, KH_) do { \
    if (OMK_SET_MAX_DYN_SMEM((ssd_a7_kernel<MODE_, DU_, KH_>), smem)) return fail(OMK_ELAUNCH, "ssd_a7: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_a7_kernel<MODE_, DU_, KH_>), grid, block, smem, stream, a); } while (0)
  if (a.mode == GS_Y) {
    if (a.dump) { if (khilo) OMK_A7K(GS_Y, true, true); else OMK_A7K(GS_Y, true, false); }
    else { if (khilo) OMK_A7K(GS_Y, false, true); else OMK_A7K(GS_Y, false, false); }
  } else {
    if (a.dump) OMK_A7K(GS_DX, true, false); else OMK_A7K(GS_DX, false, false);
  }
#undef OMK_A7K
  return OMK_OK;
}

}  // namespace omk
