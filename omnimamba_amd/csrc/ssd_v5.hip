// ssd_v5.hip -- the chunked scan of ssd_scan.h with TWO INDEPENDENT WAVES PER HEAD (round 2).
//
// The round-1 kernels (ssd_mfma.hip) spread one head over four waves that meet twice per chunk: every wave needs the whole
// running state for Q . S_in, but owns only a slice of it, so the state goes through LDS as bf16 and two barriers order
// the exchange; the strips are unbalanced (the last one builds four G tiles, the first one).  They run at 27 % of the HBM
// roofline with 43 % of the wave cycles in s_waitcnt.
//
// Here a head is cut along the OUTPUT dimension u instead (class A: headdim 64 -> two halves of 32): a wave owns
//   S^T[k = all DK][u = its half]   the running state, in 32x32 MFMA accumulators from the first token to the last,
//   O^T[u = its half][l = all 64]   the output tile of the chunk,
// and nothing it computes is needed by any other wave: the state slice is the A operand of Q . S_in straight out of its
// accumulator registers (accumulator row = contraction index, so the fp32 -> bf16 pack IS the fragment), the decay matrix
// M^T = G^T o exp2(cs_l - cs_s) w_s comes out of the G^T = K . Q^T MFMA in B-operand layout.  The price: both waves of
// a head build G and M (21 % more MFMA work, the M VALU twice).  What it buys: no state in LDS, no barrier inside a head,
// equal work per wave, 32x32x16 MFMAs (2.38 vs 2.08 PFLOP/s peak).  A workgroup is two heads (four waves, one per SIMD,
// 256 workgroups for the 512 sequences of BASELINE config 2 = one per CU); the only thing the waves share are the
// staged tiles, double buffered: ONE barrier per chunk.
//
// Register-index <-> matrix-index maps (32x32x16: D[i][j] lane = j + 32 * (i-group), see MI355X guide):
//   accumulator tile, lane (c = lane & 31, hi = lane >> 5), register r: row 4 hi + (r & 3) + 8 (r >> 2), column c.
//   A / B operand fragment: lane (i = lane & 31, hi) holds 8 consecutive contraction values 8 hi .. 8 hi + 7 of block j.
//   State rows are stored PERMUTED: register r of the k-tile kt is k = 32 kt + 16 (r >> 3) + 8 hi + (r & 7), which makes
//   pack(r = 8 j .. 8 j + 7) the standard A fragment of contraction block j; the state update reaches that order by
//   pointing the four column chunks of its ds_read_b64_tr_b16 at k-offsets {0, 8, 4, 12}.
//   The s (or l) contraction of M . U and of the state update runs in the order the G^T accumulator dictates:
//   lane-hi holds {4 hi .. 4 hi + 3} u {8 + 4 hi .. 8 + 4 hi + 3} of every 16-block; U^T fragments are read to match.
#include "ssd_scan.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int QC5 = 64;     // chunk length (tokens)

__device__ __forceinline__ s16x8 join8(s16x4 a, s16x4 b) {
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// =========================================================================================================
// class A: U per head (64 wide: y = x, dx = dy), K / Q group rows (128 wide)
// =========================================================================================================
struct SmemA5 {
  uint16_t K[2][QC5 * 128];      // group rows, kx3 swizzle, double buffered
  uint16_t Q[2][QC5 * 128];
  uint16_t U[2][2][QC5 * 64];    // [buffer][head of the pair], ux3 swizzle
  float sc[4][5][QC5];           // per wave: cs, lw = log2 w - cs, ecs = exp2 cs, ws = w exp2(cs_end - cs), dt' of the chunk in flight
  float Dv[2][64];               // D per column (only when D is per (head, column))
};
static_assert(sizeof(SmemA5) <= 160 * 1024, "LDS of one CU");

// EXTRAS / STATE / DFOLD as in ssd_mfma_a3_kernel.  Segment states of a split sequence: logical [u][k] fp32.
template <int MODE, bool EXTRAS, bool STATE, bool DFOLD>
__global__ __launch_bounds__(256, 1) void ssd_v5a_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemA5& sm = *reinterpret_cast<SmemA5*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform_i(tid >> 6);
  const int hh = w >> 1, uh = w & 1;
  const int hi = lane >> 5, l31 = lane & 31, g16 = lane >> 4, t16 = lane & 15;
  // XCD-aware order: every XCD gets a contiguous range of (batch, segment, head pair): the pairs that share B / C rows share an L2
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int pairs = a.H >> 1;
  const int nwseg = STATE ? a.nseg - 1 : a.nseg;
  const int hp = vid % pairs, seg = (vid / pairs) % nwseg, b = vid / (pairs * nwseg);
  const int h0 = 2 * hp, h = h0 + hh;
  const int g = h0 / (a.H / a.G);
  const int nC = (a.L + QC5 - 1) / QC5;
  const int c0 = seg * a.cps, c1 = (c0 + a.cps < nC) ? c0 + a.cps : nC;
  const bool rev = a.reverse != 0;
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QC5; };
  auto rowtok = [&](int i) -> int { return rev ? QC5 - 1 - i : i; };

  // ---- staging lanes: K, Q four 16-byte segments per thread (rows rowk + 16 r); U two per head (rows rowu + 32 r)
  const int rowk = tid >> 4, ck8 = (tid & 15) * 8, rowu = tid >> 3, cu8 = (tid & 7) * 8;
  const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
  const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
  const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h0 * a.U.sh;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  const int ksl = (int)a.K.sl, qsl = (int)a.Q.sl, usl = (int)a.U.sl, ush = (int)a.U.sh, osl = (int)a.osl;
  const int rtk_k = rowtok(rowk), rtk_u = rowtok(rowu), rtk_l = rowtok(lane);
  const uint32_t koff0 = (uint32_t)(rtk_k * ksl + ck8), qoff0 = (uint32_t)(rtk_k * qsl + ck8), uoff0 = (uint32_t)(rtk_u * usl + cu8);
  const int dk16 = rev ? -16 : 16, du32 = rev ? -32 : 32;
  u32x4 rk[4], rq[4], ru[4];
  float rdt = 0.f, rda = 0.f;
  int stlo = 0;   // tlo of the chunk held in the staging registers
  // branch-free loads: rows past the end of a ragged last chunk read the chunk's first row (the commit zeroes them)
  auto prefetch_k = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Kc = Kb + (int64_t)stlo * ksl;
#pragma unroll
    for (int r = 0; r < 4; r++) rk[r] = ld16(Kc + (rtk_k + dk16 * r < lim ? koff0 + (uint32_t)(r * dk16 * ksl) : (uint32_t)ck8));
  };
  auto prefetch_q = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Qc = Qb + (int64_t)stlo * qsl;
#pragma unroll
    for (int r = 0; r < 4; r++) rq[r] = ld16(Qc + (rtk_k + dk16 * r < lim ? qoff0 + (uint32_t)(r * dk16 * qsl) : (uint32_t)ck8));
  };
  auto prefetch_u = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Uc = Ub + (int64_t)stlo * usl;
#pragma unroll
    for (int hd = 0; hd < 2; hd++)
#pragma unroll
      for (int r = 0; r < 2; r++)
        ru[2 * hd + r] = ld16(Uc + hd * ush + (rtk_u + du32 * r < lim ? uoff0 + (uint32_t)(r * du32 * usl) : (uint32_t)cu8));
    // token scalars of this wave's head: lanes = rows
    const int t = stlo + rtk_l, ta = rev ? t + 1 : t;
    rdt = dtrow[t < a.L ? t : 0];
    rda = dtrow[ta < a.L ? ta : 0];
  };
  const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
  auto commit = [&](int buf) {
    const u32x4 zero4 = {0, 0, 0, 0};
    const bool full = stlo + QC5 <= a.L;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const bool ok = full || stlo + rowtok(rowk + 16 * r) < a.L;
      st16(&sm.K[buf][o_ck + 16 * 128 * r], ok ? rk[r] : zero4);
      if (!STATE) st16(&sm.Q[buf][o_ck + 16 * 128 * r], ok ? rq[r] : zero4);
    }
#pragma unroll
    for (int hd = 0; hd < 2; hd++)
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const bool ok = full || stlo + rowtok(rowu + 32 * r) < a.L;
        st16(&sm.U[buf][hd][o_cu + 32 * 64 * r], ok ? ru[2 * hd + r] : zero4);
      }
  };
  const float Ah2 = a.A[h] * LOG2E;
  float segdec = 0.f;   // STATE: log2 of the segment's total decay
  float* scw = &sm.sc[w][0][0];
  auto scalars = [&](bool fresh) {   // every wave for its own head; lanes = rows of the staged chunk
    const int t = stlo + rtk_l;
    const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
    const float wv = okd ? (a.w_is_dt ? rdt : 1.f) : 0.f;
    const float cs = wave_incl_scan_add((oka ? rda : 0.f) * Ah2);
    const float cs_end = wave_read_lane(cs, 63);
    if (STATE && fresh) segdec += cs_end;
    scw[0 * QC5 + lane] = cs;
    scw[1 * QC5 + lane] = log2_fast(wv) - cs;
    scw[2 * QC5 + lane] = exp2_fast(cs);
    scw[3 * QC5 + lane] = wv * exp2_fast(cs_end - cs);
    scw[4 * QC5 + lane] = okd ? rdt : 0.f;
  };

  // ---- lane-constant LDS element offsets
  const int o_kr = kx3(l31, 8 * hi);                               // b128 row fragment of a 128-column tile: row l31, block 0; block kk: ^ (kk << 4)
  int o_ut[2], o_kt[2];
#pragma unroll
  for (int m = 0; m < 2; m++) {
    o_ut[m] = ux3(4 * hi + (t16 >> 2) + 8 * m, 32 * uh + 16 * (g16 & 1) + 4 * (t16 & 3));          // U^T fragment (transpose reads)
    const int coff = ((t16 & 1) << 3) | ((t16 & 2) << 1);                                           // chunk t16 & 3 -> k-offset {0, 8, 4, 12}
    o_kt[m] = kx3(4 * hi + (t16 >> 2) + 8 * m, 16 * (g16 & 1) + coff);                              // K^T fragment, permuted k; tile kt: ^ (kt << 5)
  }
  const int o_xu = ux3(l31, 32 * uh + 4 * hi);                     // x of the lane's output row, columns 32 uh + 4 hi + 8 g4 ..: ^ (g4 << 3)

  // ---- running state S^T[k permuted][u = 32 uh + l31], four k tiles
  f32x16 accS[4];
#pragma unroll
  for (int kt = 0; kt < 4; kt++)
#pragma unroll
    for (int r = 0; r < 16; r++) accS[kt][r] = 0.f;
  const int64_t bh = (int64_t)b * a.H + h;
  const int u_own = 32 * uh + l31;
  auto kof = [&](int kt, int r) -> int { return 32 * kt + 16 * (r >> 3) + 8 * hi + (r & 7); };
  if (!STATE && seg > 0) {   // folded by ssd_seg_fold5_kernel: slot seg - 1 = state at the start of this segment, logical [u][k]
    const float* sp = a.seg + (bh * a.nseg + seg - 1) * SEG_STATE + u_own * 128;
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(sp + 32 * kt + 16 * j + 8 * hi);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(sp + 32 * kt + 16 * j + 8 * hi + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { accS[kt][8 * j + e] = v0[e]; accS[kt][8 * j + 4 + e] = v1[e]; }
      }
  }
  if (a.init && !STATE && seg == 0) {
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        accS[kt][r] = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u_own * a.isu + (int64_t)kof(kt, r) * a.isk, a.init_dt);
  }

  stlo = chunk_lo(c0);
  prefetch_k();
  if (!STATE) prefetch_q();
  prefetch_u();
  commit(0);
  scalars(true);
  if (!STATE && !DFOLD && tid < 128) {
    const int hd = tid >> 6, col = tid & 63;
    sm.Dv[hd][col] = a.D ? load_rt(a.D, (int64_t)(h0 + hd) * a.Dsh + (int64_t)col * a.Dsp, a.D_dt) : 0.f;
  }
  const float Dh = (DFOLD && a.D) ? load_rt(a.D, (int64_t)h * a.Dsh, a.D_dt) : 0.f;
  block_sync();
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
  uint16_t* oxb = a.outx ? (uint16_t*)a.outx + (int64_t)b * a.osb + (int64_t)h * a.osh : nullptr;
  const uint16_t* zb = (MODE == GS_Y && a.Z.p) ? (const uint16_t*)a.Z.p + (int64_t)b * a.Z.sb + (int64_t)h * a.Z.sh : nullptr;
  const int zsl = (int)a.Z.sl;

  for (int c = c0; c < c1; c++) {
    const int cur = (c - c0) & 1, nxt = cur ^ 1;
    const int tlo = chunk_lo(c);
    const int cnext = c + 1 < c1 ? c + 1 : c;   // the last iteration re-stages its own chunk: no branch around loads
    const uint16_t* sK = sm.K[cur];
    const uint16_t* sQ = sm.Q[cur];
    const uint16_t* sU = sm.U[cur][hh];
    stlo = chunk_lo(cnext);
    prefetch_k();
    if (!STATE) {
      // the state slice as A-operand fragments: pack(r = 8 j .. 8 j + 7) = contraction values 16 j + 8 hi .. + 7 of tile kt
      u32x4 sA[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; kt++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) sA[kt][j][e] = pack_bf16x2(accS[kt][8 * j + 2 * e], accS[kt][8 * j + 2 * e + 1]);
#pragma unroll
      for (int lb = 0; lb < 2; lb++) {
        // ---- (1) O^T[u][l] = exp2(cs_l) * sum_k S[u][k] Q[l][k]   (rows l = 32 lb + l31 of the chunk)
        u32x4 qf[8];
#pragma unroll
        for (int kk = 0; kk < 8; kk++) qf[kk] = ld16(&sQ[(o_kr ^ (kk << 4)) + 32 * 128 * lb]);
        f32x16 o;
#pragma unroll
        for (int r = 0; r < 16; r++) o[r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; kt++)
#pragma unroll
          for (int j = 0; j < 2; j++) o = mfma32x32x16_bf16(as_s16x8(sA[kt][j]), as_s16x8(qf[2 * kt + j]), o);
        const float e1 = scw[2 * QC5 + 32 * lb + l31], cs_l = scw[0 * QC5 + 32 * lb + l31];
        o *= e1;
        // ---- (2) intra-chunk: G^T tiles (s block sb <= lb) -> M^T fragments (registers) -> O^T += U^T . M^T
#pragma unroll
        for (int sb = 0; sb <= lb; sb++) {
          f32x16 gt;
#pragma unroll
          for (int r = 0; r < 16; r++) gt[r] = 0.f;
#pragma unroll
          for (int kk = 0; kk < 8; kk++) {
            const u32x4 ka = ld16(&sK[(o_kr ^ (kk << 4)) + 32 * 128 * sb]);
            gt = mfma32x32x16_bf16(as_s16x8(ka), as_s16x8(qf[kk]), gt);
          }
          // gt[r] = G^T[s = 32 sb + 4 hi + (r & 3) + 8 (r >> 2)][l = 32 lb + l31]
          float v[16];
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            const f32x4 lw4 = *reinterpret_cast<const f32x4*>(&scw[1 * QC5 + 32 * sb + 8 * g4 + 4 * hi]);
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const int r = 4 * g4 + e;
              float x = gt[r] * exp2_fast(cs_l + lw4[e]);
              if (sb == lb) {
                const int sl = 4 * hi + e + 8 * g4;
                if (DFOLD) x = sl < l31 ? x : (sl == l31 ? x + Dh : 0.f);
                else x = sl <= l31 ? x : 0.f;
              }
              v[r] = x;
            }
          }
          u32x4 mh[2], ml[2];   // bf16 hi + lo: the rounding of M dominates the error of y otherwise
#pragma unroll
          for (int j2 = 0; j2 < 2; j2++)
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const uint32_t hv = pack_bf16x2(v[8 * j2 + 2 * e], v[8 * j2 + 2 * e + 1]);
              mh[j2][e] = hv;
              ml[j2][e] = pack_bf16x2(v[8 * j2 + 2 * e] - bf_lo(hv), v[8 * j2 + 2 * e + 1] - bf_hi(hv));
            }
#pragma unroll
          for (int j2 = 0; j2 < 2; j2++) {
            const uint16_t* pu = sU + 16 * 64 * (2 * sb + j2);
            const s16x8 uf = join8(lds_read_tr16_b64(pu + o_ut[0]), lds_read_tr16_b64(pu + o_ut[1]));
            o = mfma32x32x16_bf16(uf, as_s16x8(mh[j2]), o);
            o = mfma32x32x16_bf16(uf, as_s16x8(ml[j2]), o);
          }
        }
        if (lb == 0) prefetch_q();
        // ---- epilogue of rows 32 lb + l31: 4 consecutive columns u = 32 uh + 8 g4 + 4 hi + {0..3} per register group
        const int lrow = 32 * lb + l31;
        const int trow = tlo + rowtok(lrow);
        if (trow < a.L) {
          const float dts = MODE == GS_DX ? scw[4 * QC5 + lrow] : 1.f;
          uint16_t* orow = ob + (int64_t)trow * osl + 32 * uh + 4 * hi;
#pragma unroll
          for (int g4 = 0; g4 < 4; g4++) {
            f32x4 y = {o[4 * g4], o[4 * g4 + 1], o[4 * g4 + 2], o[4 * g4 + 3]};
            if (!DFOLD) {
              const u32x2 xr = *reinterpret_cast<const u32x2*>(&sU[(o_xu ^ (g4 << 3)) + 32 * 64 * lb]);
              const f32x4 Du = *reinterpret_cast<const f32x4*>(&sm.Dv[hh][32 * uh + 8 * g4 + 4 * hi]);
              y = y * dts + Du * f32x4{bf_lo(xr[0]), bf_hi(xr[0]), bf_lo(xr[1]), bf_hi(xr[1])};
            }
            if (MODE == GS_Y) {
              if (EXTRAS && oxb) {
                const u32x2 ox = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
                *reinterpret_cast<u32x2*>(oxb + (int64_t)trow * osl + 32 * uh + 4 * hi + 8 * g4) = ox;
              }
              if (EXTRAS && zb) {
                const u32x2 zr = *reinterpret_cast<const u32x2*>(zb + (int64_t)trow * zsl + 32 * uh + 4 * hi + 8 * g4);
                y[0] *= silu_fast(bf_lo(zr[0])); y[1] *= silu_fast(bf_hi(zr[0]));
                y[2] *= silu_fast(bf_lo(zr[1])); y[3] *= silu_fast(bf_hi(zr[1]));
              }
            }
            const u32x2 ov = {pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
            *reinterpret_cast<u32x2*>(orow + 8 * g4) = ov;
          }
        }
      }
    }
    prefetch_u();
    // ---- (3) state update: S^T[k][u] = exp2(cs_end) S^T + sum_l K[l][k] (ws_l U[l][u])
    {
      const float dec = scw[2 * QC5 + QC5 - 1];
#pragma unroll
      for (int kt = 0; kt < 4; kt++) accS[kt] *= dec;
#pragma unroll
      for (int lb4 = 0; lb4 < 4; lb4++) {
        const uint16_t* pu = sU + 16 * 64 * lb4;
        const s16x4 u0 = lds_read_tr16_b64(pu + o_ut[0]), u1 = lds_read_tr16_b64(pu + o_ut[1]);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(&scw[3 * QC5 + 16 * lb4 + 4 * hi]);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(&scw[3 * QC5 + 16 * lb4 + 4 * hi + 8]);
        u32x4 us;
        us[0] = pack_bf16x2(bf16_to_f32((uint16_t)u0[0]) * w0[0], bf16_to_f32((uint16_t)u0[1]) * w0[1]);
        us[1] = pack_bf16x2(bf16_to_f32((uint16_t)u0[2]) * w0[2], bf16_to_f32((uint16_t)u0[3]) * w0[3]);
        us[2] = pack_bf16x2(bf16_to_f32((uint16_t)u1[0]) * w1[0], bf16_to_f32((uint16_t)u1[1]) * w1[1]);
        us[3] = pack_bf16x2(bf16_to_f32((uint16_t)u1[2]) * w1[2], bf16_to_f32((uint16_t)u1[3]) * w1[3]);
#pragma unroll
        for (int kt = 0; kt < 4; kt++) {
          const uint16_t* pk = sK + 16 * 128 * lb4;
          const s16x8 kf = join8(lds_read_tr16_b64(pk + (o_kt[0] ^ (kt << 5))), lds_read_tr16_b64(pk + (o_kt[1] ^ (kt << 5))));
          accS[kt] = mfma32x32x16_bf16(kf, as_s16x8(us), accS[kt]);
        }
      }
    }
    // ---- (4) stage the next chunk, next chunk's scalars; ONE barrier per chunk
    commit(nxt);
    scalars(c + 1 < c1);
    block_sync();
  }
  if (STATE) {
    float* sp = a.seg + (bh * a.nseg + seg) * SEG_STATE + u_own * 128;
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        f32x4 v0, v1;
#pragma unroll
        for (int e = 0; e < 4; e++) { v0[e] = accS[kt][8 * j + e]; v1[e] = accS[kt][8 * j + 4 + e]; }
        *reinterpret_cast<f32x4*>(sp + 32 * kt + 16 * j + 8 * hi) = v0;
        *reinterpret_cast<f32x4*>(sp + 32 * kt + 16 * j + 8 * hi + 4) = v1;
      }
    if (lane == 0 && uh == 0) a.seg[(int64_t)a.B * a.H * a.nseg * SEG_STATE + bh * a.nseg + seg] = segdec;
    return;
  }
  if (a.fin && seg == a.nseg - 1) {
    const float extra = a.fin_extra_decay ? expf(dtrow[0] * a.A[h]) : 1.f;
#pragma unroll
    for (int kt = 0; kt < 4; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++)
        a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)u_own * a.fsu + (int64_t)kof(kt, r) * a.fsk] = accS[kt][r] * extra;
  }
}

// Segment states -> start states, in place (logical [u][k] order): slot j becomes the state at the START of segment j + 1:
// run = exp2(dec_j) run + slot_j, run_0 = the caller's initial state.  One thread = one state element of one (b, h).
__global__ void ssd_seg_fold5_kernel(GScan a) {
  const int64_t bh = blockIdx.x / (SEG_STATE / 256);
  const int e = (blockIdx.x % (SEG_STATE / 256)) * 256 + threadIdx.x;
  const int b = (int)(bh / a.H), h = (int)(bh % a.H);
  float run = 0.f;
  if (a.init) {
    const int u = e >> 7, k = e & 127;
    run = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
  }
  const float* sdec = a.seg + (int64_t)a.B * a.H * a.nseg * SEG_STATE + bh * a.nseg;
  float* sp = a.seg + bh * a.nseg * SEG_STATE + e;
  for (int j = 0; j + 1 < a.nseg; j++) {
    run = exp2_fast(sdec[j]) * run + sp[(int64_t)j * SEG_STATE];
    sp[(int64_t)j * SEG_STATE] = run;
  }
}

static bool src_ok16_5(const Src& s, bool need) {
  if (!s.p) return !need;
  return s.dt == OMK_BF16 && ((uintptr_t)s.p & 15) == 0 && s.sb % 8 == 0 && s.sl % 8 == 0 && s.sh % 8 == 0;
}
static bool stride_ok5(int64_t s) { return s >= 0 && s < ((int64_t)1 << 24); }

// MEASURED SLOWER than the round-1 strips on the MI355X (B 8, L 4096: 400 vs 287 us on the same box, profiles/
// r02_scan_v5_vs_v3.txt): four waves per CU are one wave per SIMD, and a single in-order wave exposes every LDS / MFMA /
// transcendental latency (8 cycles per instruction measured; the loop is 1650 instructions per chunk, 336 of them
// accumulator-file moves because state + output + staging exceed 256 VGPRs).  Kept selectable (OMK_SSD_V5=1) as the
// starting point for a hand-scheduled version; the default stays ssd_mfma_a3_kernel.
bool ssd_v5_enabled() {
  const char* e = getenv("OMK_SSD_V5");
  return e && e[0] == '1';
}

// does the two-waves-per-head class A kernel take this scan?  (same shape / layout conditions as ssd_mfma_a3_kernel, plus
// an even number of heads per group: a workgroup is a head pair that shares its B / C rows)
bool ssd_v5a_applies(const GScan& g) {
  if (!ssd_v5_enabled()) return false;
  if (g.mode != GS_Y && g.mode != GS_DX) return false;
  if (g.DU != 64 || g.DK != 128 || g.G < 1 || g.H % 2 != 0 || (g.H / g.G) % 2 != 0) return false;
  if (!src_ok16_5(g.U, true) || !src_ok16_5(g.K, true) || !src_ok16_5(g.Q, true) || !src_ok16_5(g.Z, false)) return false;
  if (g.out && (g.out_dt != OMK_BF16 || ((uintptr_t)g.out & 15) || g.osb % 8 || g.osl % 8 || g.osh % 8)) return false;
  if (g.outx && ((uintptr_t)g.outx & 15)) return false;
  if (g.mode == GS_DX && g.dD) return false;
  if (!stride_ok5(g.K.sl) || !stride_ok5(g.Q.sl) || !stride_ok5(g.U.sl) || !stride_ok5(g.U.sh) || !stride_ok5(g.osl) || (g.Z.p && !stride_ok5(g.Z.sl))) return false;
  return true;
}

// state-only pass + fold of a split sequence (see ssd_mfma_prepare_segments); leaves logical [u][k] start states
int ssd_v5a_prepare_segments(const GScan& g, omk_stream stream) {
  GScan a = g;
  const SegPlan sp = ssd_segments(a.B * a.H, a.L);
  a.nseg = sp.nseg; a.cps = sp.cps;
  if (!a.seg || a.nseg < 2) return OMK_OK;
  dim3 block(256), sgrid((unsigned)(a.B * (a.H / 2) * (a.nseg - 1)));
  const size_t smem = sizeof(SmemA5);
  if (OMK_SET_MAX_DYN_SMEM((ssd_v5a_kernel<GS_Y, false, true, false>), smem)) return fail(OMK_ELAUNCH, "ssd_v5: cannot raise dynamic LDS to %zu", smem);
  OMK_LAUNCH((ssd_v5a_kernel<GS_Y, false, true, false>), sgrid, block, smem, stream, a);
  dim3 fgrid((unsigned)((int64_t)a.B * a.H * (SEG_STATE / 256)));
  OMK_LAUNCH(ssd_seg_fold5_kernel, fgrid, block, 0, stream, a);
  return OMK_OK;
}

int ssd_v5a_launch(const GScan& g, omk_stream stream) {
  GScan a = g;
  const SegPlan sp = a.seg ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QC5 - 1) / QC5};
  a.nseg = sp.nseg; a.cps = sp.cps;
  dim3 grid((unsigned)(a.B * (a.H / 2) * a.nseg)), block(256);
  const size_t smem = sizeof(SmemA5);
#define OMK_A5(MODE_, EX_, ST_, DF_) do { \
    if (OMK_SET_MAX_DYN_SMEM((ssd_v5a_kernel<MODE_, EX_, ST_, DF_>), smem)) return fail(OMK_ELAUNCH, "ssd_v5: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_v5a_kernel<MODE_, EX_, ST_, DF_>), grid, block, smem, stream, a); } while (0)
  if (a.nseg > 1 && !a.seg_ready) {
    int rc = ssd_v5a_prepare_segments(g, stream);
    if (rc) return rc;
  }
  const bool dfold = !a.D || a.Dsp == 0;   // one D per head (or none)
  if (a.mode == GS_Y && (a.Z.p || a.outx)) { if (dfold) OMK_A5(GS_Y, true, false, true); else OMK_A5(GS_Y, true, false, false); }
  else if (a.mode == GS_Y) { if (dfold) OMK_A5(GS_Y, false, false, true); else OMK_A5(GS_Y, false, false, false); }
  else OMK_A5(GS_DX, false, false, false);
#undef OMK_A5
  return OMK_OK;
}

}  // namespace omk
