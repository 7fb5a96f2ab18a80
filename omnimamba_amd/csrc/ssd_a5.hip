// ssd_a5.hip -- class A scan (forward y and the dx scan of the backward) with the running state in COLUMN SLICES.
//
// The row-strip kernel (ssd_mfma.hip, a3) gives every wave 16 output rows of a 64-token chunk: G and M never leave registers, but the
// running state is shared -- it is rounded to bf16 and published through LDS once per chunk (16 KB written, 64 KB read back by the four
// waves, two barriers), and the causal block makes strip 3 do four times the intra-chunk work of strip 0.  This kernel cuts a head the
// other way: wave w owns the output COLUMNS u in [16 w, 16 w + 16) of its head and the matching slice S[k = 0..127][u] of the state
// (eight 16 x 16 accumulator tiles, 32 registers), and walks the chunk in SUB-CHUNKS of 16 tokens:
//
//   (a)  O^T[u][l]  = rl_l * sum_k S_in^T[u][k] Q^T[k][l]      4 MFMA 16x16x32; the A operand IS the accumulator slice, packed to bf16 in
//                                                             registers (tile pair 2i, 2i+1 = the 8 contraction slots of k-step i), so
//                                                             the state never touches LDS and needs no barrier;
//   (b)  O^T[u][l] += sum_{s <= l} U^T[u][s] M^T[s][l]         only the DIAGONAL 16 x 16 block is left of the intra-chunk product (what is
//        M[l][s] = (Q_l . K_s) w_s exp2(cs_l - cs_s)           further back than the sub-chunk is in S_in): 4 MFMA for G, one M build of
//                                                             4 elements per lane, ONE MFMA whose 32 contraction slots carry M as
//                                                             bf16 hi (16 slots) + lo (16 slots) against U twice;
//   (c)  S[k][u]    = dec S[k][u] + sum_l K^T[k][l] (ws_l U[l][u])   8 MFMA (one per state tile), the scaled U operand as hi + lo in
//                                                             the same way: the carried state is exact to fp32 accumulation.
//
// Every wave does the same work (no strips), the per-sub-chunk recurrence is 16 multiplies + 8 independent MFMAs deep, and (a) / (b)
// hang off it as independent side branches.  The price is that all four waves of a head build the same M (4 tiles per chunk instead of
// 2.5 on average) and read the same K / Q rows.  One workgroup = 8 waves = the two heads of a head PAIR (they share the group's K / Q
// tiles in LDS), one workgroup per CU, one barrier per 64-token chunk.
//
// Contraction-slot bookkeeping (the MFMA sums over its 32 slots in any order as long as A and B agree):
//   state tile t (t = 0..7), accumulator register r on lane (n = lane & 15, g = lane >> 4)  <->  u = 16 w + n,
//   k = 32 (t >> 1) + 8 g + 4 (t & 1) + r -- so the registers of tiles 2i, 2i + 1 are k = 32 i + 8 g + 0..7, exactly what a 16-byte
//   row read of Q hands lane (l, g) for k-step i.
#include <cstdlib>
#include "ssd_scan.h"
#include "ssd_tiles.h"

#ifdef OMK_A5_NOFENCE
#undef OMK_SCHED_FENCE
#define OMK_SCHED_FENCE() do { } while (0)
#endif

#ifndef OMK_A5_ABL
#define OMK_A5_ABL 0
#endif

namespace omk {

constexpr int QA5 = 64;    // tokens staged per barrier
struct SmemA5 {
  uint16_t K[2][QA5 * 128];       // kx3 swizzle
  uint16_t Q[2][QA5 * 128];       // kx3 swizzle
  uint16_t U[2][2][QA5 * 64];     // [buffer][head of the pair], ux3 swizzle
  f32x2 rv[2][2][QA5];            // [buffer][head][chunk row]: {cs, rl} of the row
  float lw[2][2][QA5], ws[2][2][QA5], dtl[2][2][QA5];
  float dec[2][2][4];             // decay over sub-chunk j
  float Dv[2][64];
};
static_assert(sizeof(SmemA5) <= 160 * 1024, "one workgroup per CU");

template <int MODE, bool EXTRAS, bool DFOLD, bool DUMP>
__global__ __launch_bounds__(512) void ssd_a5_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemA5& sm = *reinterpret_cast<SmemA5*>(smem_raw);
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
  const int nC = (a.L + QA5 - 1) / QA5;
  const int c0 = seg * a.cps, c1 = (c0 + a.cps < nC) ? c0 + a.cps : nC;
  const bool rev = a.reverse != 0;
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QA5; };
  auto rowtok = [&](int i) -> int { return rev ? QA5 - 1 - i : i; };

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
  int stlo = 0;
  auto prefetch = [&]() {
    const uint32_t sk = 2u * (uint32_t)(stlo * ksl), sq = 2u * (uint32_t)(stlo * qsl), su = 2u * (uint32_t)(stlo * usl);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int ro = rev ? 32 * (1 - r) : 32 * r;
      rk[r] = buf_ld16(Kr, kvo, sk + 2u * (uint32_t)(ro * ksl));
      rq[r] = buf_ld16(Qr, qvo, sq + 2u * (uint32_t)(ro * qsl));
      ru[r] = buf_ld16(Ur, uvo, su + 2u * (uint32_t)(ro * usl));
    }
    rdt = buf_ld_f32(Dr, dvo, 4u * (uint32_t)stlo);
    rda = buf_ld_f32(Dr, dvo_a, 4u * (uint32_t)stlo);
  };
  const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
  auto commit = [&](int buf) {   // rows past the end arrived as zeros
#pragma unroll
    for (int r = 0; r < 2; r++) {
      st16(&sm.K[buf][o_ck + 32 * 128 * r], rk[r]);
      st16(&sm.Q[buf][o_ck + 32 * 128 * r], rq[r]);
      st16(&sm.U[buf][hh][o_cu + 32 * 64 * r], ru[r]);
    }
  };
  const float Ah = a.A[h];
  const float Ah2 = Ah * LOG2E;
  auto scalars = [&](int buf) {   // waves with w == 0; lanes = rows of the staged chunk of head hh
    {
      const int t = stlo + rowtok(lane);
      const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
      rwv = okd ? (a.w_is_dt ? rdt : 1.f) : 0.f;
      rdt = okd ? rdt : 0.f;
      rda = oka ? rda : 0.f;
    }
    const float cs = wave_incl_scan_add(rda * Ah2);
    const float e15 = wave_read_lane(cs, 15), e31 = wave_read_lane(cs, 31), e47 = wave_read_lane(cs, 47), e63 = wave_read_lane(cs, 63);
    const float csb = g16 == 0 ? 0.f : (g16 == 1 ? e15 : (g16 == 2 ? e31 : e47));   // prefix in front of the lane's sub-chunk
    const float cse = g16 == 0 ? e15 : (g16 == 1 ? e31 : (g16 == 2 ? e47 : e63));   // prefix at its end
    sm.rv[buf][hh][lane] = f32x2{cs, exp2_fast(cs - csb)};
    if (MODE == GS_DX) sm.dtl[buf][hh][lane] = rdt;
    sm.lw[buf][hh][lane] = log2_fast(rwv) - cs;
    sm.ws[buf][hh][lane] = rwv * exp2_fast(cse - cs);
    if (t16 == 15) sm.dec[buf][hh][g16] = exp2_fast(cse - csb);
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
  if (seg > 0) {   // folded by ssd_seg_fold_kernel (a3 accumulator order): slot seg - 1 = state at the start of this segment
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

  stlo = chunk_lo(c0);
  prefetch();
  commit(0);
  if (w == 0) scalars(0);
  if (!DFOLD && tid < 128) sm.Dv[tid >> 6][tid & 63] = a.D ? load_rt(a.D, (int64_t)(2 * hp + (tid >> 6)) * a.Dsh + (int64_t)(tid & 63) * a.Dsp, a.D_dt) : 0.f;
  const float Dh = (DFOLD && a.D) ? load_rt(a.D, (int64_t)h * a.Dsh, a.D_dt) : 0.f;
  block_sync();
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
  const BufRes Or = make_buf(ob, (uint32_t)((int64_t)a.L * osl * 2));
  uint16_t* oxb = a.outx ? (uint16_t*)a.outx + (int64_t)b * a.osb + (int64_t)h * a.osh : nullptr;
  const uint16_t* zb = (MODE == GS_Y && a.Z.p) ? (const uint16_t*)a.Z.p + (int64_t)b * a.Z.sb + (int64_t)h * a.Z.sh : nullptr;
  const int zsl = (int)a.Z.sl;
  f32x4 Du = {0.f, 0.f, 0.f, 0.f};
  if (!DFOLD) Du = *reinterpret_cast<const f32x4*>(&sm.Dv[hh][16 * w + 4 * g16]);

  // ---- the sub-chunk pipeline.  A sub-chunk is two phases: (1) S_in^T Q^T and G (8 MFMAs on the Q / K row fragments), (2) state
  // update, M build, U^T M^T and the output rows (9 MFMAs on the transposed K fragments, the U fragment and the token scalars).  The
  // operands of phase 2 are requested in front of phase 1 of the same sub-chunk, the row fragments of the NEXT sub-chunk in front of
  // phase 2 (into the registers phase 1 has just released), so every LDS read has a phase of matrix work between request and use.
  // The one barrier of a chunk sits inside its last sub-chunk: behind the last request for the current buffers, in front of the
  // first request for the next ones.
  struct FragA { u32x4 qf[4], kf[4]; };
  struct FragB { s16x4 uf, kt[8]; f32x2 rv; f32x4 lw4, ws4; float dec, dts; u32x2 xr; };
  auto load_rows = [&](FragA& f, int buf, int j) {
    if (OMK_A5_ABL & 1) { asm volatile("" : "+v"(f.qf[0]), "+v"(f.qf[1]), "+v"(f.qf[2]), "+v"(f.qf[3]), "+v"(f.kf[0]), "+v"(f.kf[1]), "+v"(f.kf[2]), "+v"(f.kf[3])); return; }
#pragma unroll
    for (int i = 0; i < 4; i++) f.qf[i] = ld16(&sm.Q[buf][o_rd[i] + 16 * 128 * j]);
#ifdef OMK_A5_SKIPM
    if (j == 0)
#endif
#pragma unroll
    for (int i = 0; i < 4; i++) f.kf[i] = ld16(&sm.K[buf][o_rd[i] + 16 * 128 * j]);
  };
  auto load_cols = [&](FragB& f, int buf, int j) {
    if (OMK_A5_ABL & 1) {
      asm volatile("" : "+v"(f.uf), "+v"(f.rv), "+v"(f.lw4), "+v"(f.ws4), "+v"(f.dec));
#pragma unroll
      for (int t = 0; t < 8; t++) asm volatile("" : "+v"(f.kt[t]));
      return;
    }
    f.rv = sm.rv[buf][hh][16 * j + t16];
    f.ws4 = *reinterpret_cast<const f32x4*>(&sm.ws[buf][hh][16 * j + 4 * g16]);
    f.uf = lds_read_tr16_b64(&sm.U[buf][hh][o_uf + 16 * 64 * j]);   // U[16 j + 4 g16 + e][16 w + t16]
    f.dec = sm.dec[buf][hh][j];
#pragma unroll
    for (int t = 0; t < 8; t++) f.kt[t] = lds_read_tr16_b64(&sm.K[buf][o_kt[t >> 1] + 4 * (t & 1) + 16 * 128 * j]);
    f.lw4 = *reinterpret_cast<const f32x4*>(&sm.lw[buf][hh][16 * j + 4 * g16]);
    if (!DFOLD) f.xr = *reinterpret_cast<const u32x2*>(&sm.U[buf][hh][o_xu + 16 * 64 * j]);
    if (MODE == GS_DX) f.dts = sm.dtl[buf][hh][16 * j + t16];
  };
  f32x4 accA, gt;
  auto phase1 = [&](const FragA& f, int j, bool dump_here, uint16_t* dp) {
    // (a) S_in^T Q^T: the bf16 pack of the accumulator slice is the A operand;  (b) G^T[s][l] of the diagonal block
    accA = f32x4{0.f, 0.f, 0.f, 0.f}; gt = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; i++) {
      u32x4 sp;
      sp[0] = pack_bf16x2(accS[2 * i][0], accS[2 * i][1]);
      sp[1] = pack_bf16x2(accS[2 * i][2], accS[2 * i][3]);
      sp[2] = pack_bf16x2(accS[2 * i + 1][0], accS[2 * i + 1][1]);
      sp[3] = pack_bf16x2(accS[2 * i + 1][2], accS[2 * i + 1][3]);
      if (DUMP && j == 0 && dump_here) st16(dp + su * 128 + (((4 * i + g16) ^ swzK(su)) << 3), sp);
#ifdef OMK_A5_SKIPM   // experiment (wrong results): what sharing M between the waves of a head could buy
      if (j == 0)
#endif
      gt = mfma16x16x32_bf16(as_s16x8(f.kf[i]), as_s16x8(f.qf[i]), gt);
      accA = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.qf[i]), accA);
    }
  };
  auto phase2 = [&](const FragB& f, int j, int tlo) {
    // ---- (c) state update: S = dec S + K^T (ws U), the scaled U rows as bf16 hi + lo
    {
      float us[4];
#pragma unroll
      for (int e = 0; e < 4; e++) us[e] = bf16_to_f32((uint16_t)f.uf[e]) * f.ws4[e];
      u32x4 ub;
#pragma unroll
      for (int p2 = 0; p2 < 2; p2++) {
        const uint32_t hi = pack_bf16x2(us[2 * p2], us[2 * p2 + 1]);
        ub[p2] = hi;
#ifdef OMK_A5_NOLO
        ub[2 + p2] = 0u;
#else
        ub[2 + p2] = pack_bf16x2(us[2 * p2] - bf_lo(hi), us[2 * p2 + 1] - bf_hi(hi));
#endif
      }
      s16x4 uh, ul;
      uh[0] = (short)(ub[0] & 0xffffu); uh[1] = (short)(ub[0] >> 16); uh[2] = (short)(ub[1] & 0xffffu); uh[3] = (short)(ub[1] >> 16);
      ul[0] = (short)(ub[2] & 0xffffu); ul[1] = (short)(ub[2] >> 16); ul[2] = (short)(ub[3] & 0xffffu); ul[3] = (short)(ub[3] >> 16);
#pragma unroll
      for (int t = 0; t < 8; t++) {
#ifdef OMK_A5_NODECAY
        accS[t] = mfma16x16x16_bf16(f.kt[t], uh, accS[t]);
#else
        accS[t] = mfma16x16x16_bf16(f.kt[t], uh, accS[t] * f.dec);
#endif
#ifndef OMK_A5_NOLO
        accS[t] = mfma16x16x16_bf16(f.kt[t], ul, accS[t]);
#endif
      }
    }
    // ---- (b) M^T (decay, mask, hi + lo) -> U^T M^T
    float v[4];
#ifdef OMK_A5_SKIPM
    if (j != 0) { v[0] = gt[0]; v[1] = gt[1]; v[2] = gt[2]; v[3] = gt[3]; } else
#endif
#pragma unroll
    for (int r = 0; r < 4; r++) {
      v[r] = gt[r] * exp2_fast(f.rv[0] + f.lw4[r]);
      if (DFOLD) v[r] = (4 * g16 + r < t16) ? v[r] : (4 * g16 + r == t16 ? v[r] + Dh : 0.f);
      else v[r] = (4 * g16 + r <= t16) ? v[r] : 0.f;
    }
    u32x4 mm;
#ifdef OMK_A5_SKIPM
    if (j != 0) mm = __builtin_bit_cast(u32x4, f.lw4); else
#endif
#pragma unroll
    for (int p2 = 0; p2 < 2; p2++) {
      const uint32_t hi = pack_bf16x2(v[2 * p2], v[2 * p2 + 1]);
      mm[p2] = hi;
      mm[2 + p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi), v[2 * p2 + 1] - bf_hi(hi));
    }
    s16x4 mh, ml;
    mh[0] = (short)(mm[0] & 0xffffu); mh[1] = (short)(mm[0] >> 16); mh[2] = (short)(mm[1] & 0xffffu); mh[3] = (short)(mm[1] >> 16);
    ml[0] = (short)(mm[2] & 0xffffu); ml[1] = (short)(mm[2] >> 16); ml[2] = (short)(mm[3] & 0xffffu); ml[3] = (short)(mm[3] >> 16);
    f32x4 accB = mfma16x16x16_bf16(f.uf, mh, f32x4{0.f, 0.f, 0.f, 0.f});
    accB = mfma16x16x16_bf16(f.uf, ml, accB);
    // ---- output rows: the lane's row l = 16 j + t16, columns 16 w + 4 g16 + r
    f32x4 o = accA * f.rv[1] + accB;
    const int erow = rowtok(16 * j + t16);
    if (!DFOLD) {
      const float dts = MODE == GS_DX ? f.dts : 1.f;
      o = o * dts + Du * f32x4{bf_lo(f.xr[0]), bf_hi(f.xr[0]), bf_lo(f.xr[1]), bf_hi(f.xr[1])};
    }
    const uint32_t eoff = (uint32_t)(erow * osl + 16 * w + 4 * g16);
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
    if (!(OMK_A5_ABL & 16)) buf_st8(Or, ov, 2u * eoff, 2u * (uint32_t)(tlo * osl));
    else asm volatile("" :: "v"(ov));
  };

#ifdef OMK_PHASE_PROF   // developer build (tools/phase_prof_a5.py): s_memtime deltas per phase, workgroup 0
  uint64_t pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
#define PT5(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
  uint64_t tprev = prof ? clock64_() : 0;
  const uint64_t t_core0 = tprev, t_ref0 = prof ? __builtin_readsteadycounter() : 0;
#else
#define PT5(i) do { } while (0)
#endif
#ifdef OMK_A5_PRIO
  if (wave >= 4) OMK_SET_PRIO(1);
#endif
  FragA fa;
  FragB fb0, fb1;   // two sets, static names: sub-chunk parity
  load_rows(fa, 0, 0);
  load_cols(fb0, 0, 0);
  stlo = chunk_lo(c0 + 1 < c1 ? c0 + 1 : c0);
  prefetch();
  for (int c = c0; c < c1; c++) {
    const int cur = (c - c0) & 1, nxt = cur ^ 1;
    const int tlo = chunk_lo(c);
    bool dump_here = false;
    uint16_t* dp = nullptr;
    if (DUMP && a.dump) {   // window-boundary image of the state in front of this chunk, the [u][k] kx3 image ssd_cp.hip reads
      const int cid = rev ? nC - 1 - c : c;
      dump_here = rev ? (cid == nC - 1 || (cid & 1)) : !(cid & 1);
      dp = a.dump + ((((int64_t)b * a.dump_nw + (cid >> 1)) * a.H + h) << 13);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      FragB& fb = (j & 1) ? fb1 : fb0;
      FragB& fbn = (j & 1) ? fb0 : fb1;
      PT5(0);
      phase1(fa, j, dump_here, dp);
      OMK_SCHED_FENCE();
      PT5(2);
      if (j < 3) { load_rows(fa, cur, j + 1); load_cols(fbn, cur, j + 1); }
      else {
        // stage the next chunk (requested an iteration ago), its scalars, the one barrier of the chunk
        if (!(OMK_A5_ABL & 2)) commit(nxt);
        if (w == 0) scalars(nxt);
        PT5(4);
        if (!(OMK_A5_ABL & 8)) block_sync();
        PT5(5);
        load_rows(fa, nxt, 0);
        load_cols(fbn, nxt, 0);
        stlo = chunk_lo(c + 2 < c1 ? c + 2 : c1 - 1);   // (the last iterations re-stage the last chunk: no branch around loads)
        if (!(OMK_A5_ABL & 4)) prefetch();
      }
      OMK_SCHED_FENCE();
      PT5(3);
#if defined(OMK_A5_PAD) && OMK_A5_PAD == 1
      asm volatile(".rept 100\n s_nop 0\n .endr" ::: "memory");
#elif defined(OMK_A5_PAD) && OMK_A5_PAD == 2
      { int dmy = lane; asm volatile(".rept 100\n v_mov_b32 %0, %0\n .endr" : "+v"(dmy)); }
#elif defined(OMK_A5_PAD) && OMK_A5_PAD == 3
      { int dmy = lane, dm2 = tid; asm volatile(".rept 50\n v_mov_b32 %0, %0\n v_mov_b32 %1, %1\n .endr" : "+v"(dmy), "+v"(dm2)); }
#endif
      phase2(fb, j, tlo);
      OMK_SCHED_FENCE();
    }
  }
#ifdef OMK_PHASE_PROF
  PT5(0);
  if (prof) { pt[10] = clock64_() - t_core0; pt[11] = __builtin_readsteadycounter() - t_ref0; }
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

bool ssd_a5_applies(const GScan& g) {
  if (const char* e = getenv("OMK_SSD_A5")) { if (e[0] == '0') return false; }
  else return false;   // opt-in until measured
  if (g.mode != GS_Y && g.mode != GS_DX) return false;
  if (g.H % 2 != 0 || (g.H / g.G) % 2 != 0) return false;
  if (g.state_only) return false;
  return true;
}

// called by ssd_mfma_launch after its shape / alignment checks (same preconditions as the row-strip kernel)
int ssd_a5_launch(const GScan& g, omk_stream stream) {
  GScan a = g;
  const SegPlan sp = a.seg ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QA5 - 1) / QA5};
  a.nseg = sp.nseg; a.cps = sp.cps;
  if (a.nseg > 1 && !a.seg_ready) {
    int rc = ssd_mfma_prepare_segments(g, stream);
    if (rc) return rc;
  }
  dim3 grid((unsigned)(a.B * (a.H / 2) * a.nseg)), block(512);
  const size_t smem = sizeof(SmemA5);
#define OMK_A5(MODE_, EX_, DF_, DU_) do { \
    if (OMK_SET_MAX_DYN_SMEM((ssd_a5_kernel<MODE_, EX_, DF_, DU_>), smem)) return fail(OMK_ELAUNCH, "ssd_a5: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_a5_kernel<MODE_, EX_, DF_, DU_>), grid, block, smem, stream, a); } while (0)
  const bool dfold = !a.D || a.Dsp == 0;   // one D per head (or none)
  if (a.mode == GS_Y) {
    const bool ex = a.Z.p || a.outx;
    if (a.dump) { if (ex) return OMK_EUNSUPPORTED; if (dfold) OMK_A5(GS_Y, false, true, true); else OMK_A5(GS_Y, false, false, true); }
    else if (ex) { if (dfold) OMK_A5(GS_Y, true, true, false); else OMK_A5(GS_Y, true, false, false); }
    else { if (dfold) OMK_A5(GS_Y, false, true, false); else OMK_A5(GS_Y, false, false, false); }
  } else {
    if (a.dump) OMK_A5(GS_DX, false, false, true); else OMK_A5(GS_DX, false, false, false);
  }
#undef OMK_A5
  return OMK_OK;
}

}  // namespace omk
