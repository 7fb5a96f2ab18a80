// ssd_v6.hip -- the PRECISE forward scan: the row-strip kernel of ssd_mfma.hip (class A, y) with the two bf16 roundings that set
// its error floor -- the copy of the carried state that feeds Q . S_in and the w_l K_l operand of the state update -- entering the
// MFMAs as hi + lo pairs (OMK_SSD_PRECISE=1).
//
// The second 16 KB state tile needs LDS the default kernel does not have at two workgroups per CU, so this variant keeps K SINGLE
// buffered: the next chunk's K tile is committed after the barrier that ends the state update and the Q . S_in product of the NEXT
// chunk runs between that barrier and the second one (still two barriers per chunk):
//     intra(c) -> store O(c) -> state update(c) -> publish S -> barrier A -> commit K(c + 1) -> acc = Q(c + 1) . S -> barrier B
// Same lane layouts and LDS swizzles as ssd_mfma_a3_kernel; no segment / state-only / gate variants (those stay on a3).
//
// Round 2 measured this loop order WITHOUT the hi + lo operands as a way to three workgroups per CU: slower in every variant
// (281 - 397 us against 269 - 273 us, profiles/r02_scan_v6_variants.txt) -- what two co-resident workgroups leave idle is not
// there to be filled by a third.  Those variants were removed in round 3; the numbers stay in profiles/ and DESIGN.md section 4.7.
// Cost of PRECISE: + 27 % scan time (272 -> 345 us, profiles/r02_scan_precise.txt); arithmetic error of y at the production shape
// 1.3e-3 -> 3e-6 on slow-decay heads, final state < 1e-3.  The default path rounds exactly where upstream's kernels round
// (tests/test_ops_ssd.py::test_default_forward_is_no_worse_than_upstream_rounding).
#include <cstdlib>

#include "ssd_scan.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int QC6 = 64;

struct SmemA6 {
  uint16_t K[QC6 * 128];
  uint16_t U[2][QC6 * 64];
  uint16_t S[64 * 128];      // [u][k] bf16 copy of S_in
  float cs[2][QC6], lw[2][QC6], ecs[2][QC6], ws[2][QC6], dtl[2][QC6];
  float Dv[64];
  uint16_t Slo[64 * 128];    // PRECISE only: S_in - bf16(S_in).  LAST member: the other variants are launched without it (51 KB)
};
static_assert(sizeof(SmemA6) <= 80 * 1024, "PRECISE: two workgroups must fit the 160 KB of a CU");

// PRECISE (forward only): the two bf16 roundings that set the error floor of the scan -- the copy of the carried state that feeds
// Q . S_in and the w_l K_l operand of the state update -- both enter as hi + lo pairs (16 + 8 more MFMAs per wave and chunk, a
// second 16 KB state tile: the LDS budget that the single-buffered K of this file frees).  y on slow-decay heads and with random
// initial states then sits below the 1e-3 budget of the north star, the final state below 1e-3 too (tests); the reference itself
// (upstream's Triton kernels) rounds both operands to bf16 like the default path does.
template <int MODE, bool DFOLD, int OCC, bool EARLY, bool PRECISE = false>   // OCC: workgroups per CU the register budget allows; EARLY: staging loads before the intra phase
__global__ __launch_bounds__(256, OCC) void ssd_mfma_a6_kernel(GScan a) {
  constexpr int QC = QC6;
  OMK_DYN_SMEM(smem_raw);
  SmemA6& sm = *reinterpret_cast<SmemA6*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform_i(tid >> 6);
  const int h32 = lane >> 5, l31 = lane & 31, g16 = lane >> 4, t16 = lane & 15;
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-aware order (as a3)
  const int h = vid % a.H, b = vid / a.H;
  const int g = h / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
  const bool rev = a.reverse != 0;
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QC; };
  auto rowtok = [&](int i) -> int { return rev ? QC - 1 - i : i; };

  const int rowk = tid >> 4, ck8 = (tid & 15) * 8, rowu = tid >> 3, cu8 = (tid & 7) * 8;
  const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
  const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
  const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  const int ksl = (int)a.K.sl, qsl = (int)a.Q.sl, usl = (int)a.U.sl, osl = (int)a.osl;
  const uint32_t koff0 = (uint32_t)(rowtok(rowk) * ksl + ck8), uoff0 = (uint32_t)(rowtok(rowu) * usl + cu8);
  const uint32_t qoff0 = (uint32_t)(rowtok(16 * w + t16) * qsl + 8 * g16);
  const int kstep = (rev ? -16 : 16) * ksl, ustep = (rev ? -32 : 32) * usl;
  u32x4 rk[4], ru[2], qf[4];
  float rdt = 0.f, rda = 0.f, rwv = 0.f;
  int stlo = 0;
  const int rtk_k = rowtok(rowk), rtk_u = rowtok(rowu), rtk_q = rowtok(16 * w + t16), rtk_l = rowtok(lane);
  const int dk16 = rev ? -16 : 16, du32 = rev ? -32 : 32;
  // branch-free staging loads (rows past the end of a ragged last chunk read the chunk's first row; the commit zeroes them)
  auto prefetch_k = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Kc = Kb + (int64_t)stlo * ksl;
#pragma unroll
    for (int r = 0; r < 4; r++) rk[r] = ld16(Kc + (rtk_k + dk16 * r < lim ? koff0 + (uint32_t)(r * kstep) : (uint32_t)ck8));
  };
  auto prefetch_q = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Qc = Qb + (int64_t)stlo * qsl;
    const uint32_t qo = rtk_q < lim ? qoff0 : (uint32_t)(8 * g16);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) qf[kk] = ld16(Qc + 32 * kk + qo);
  };
  auto prefetch_u = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Uc = Ub + (int64_t)stlo * usl;
#pragma unroll
    for (int r = 0; r < 2; r++) ru[r] = ld16(Uc + (rtk_u + du32 * r < lim ? uoff0 + (uint32_t)(r * ustep) : (uint32_t)cu8));
    const int t = stlo + rtk_l, ta = rev ? t + 1 : t;
    rdt = dtrow[t < a.L ? t : 0];
    rda = dtrow[ta < a.L ? ta : 0];
  };
  const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
  auto commit_k = [&]() {
    const u32x4 zero4 = {0, 0, 0, 0};
    const bool full = stlo + QC <= a.L;
#pragma unroll
    for (int r = 0; r < 4; r++) st16(&sm.K[o_ck + 16 * 128 * r], (full || stlo + rowtok(rowk + 16 * r) < a.L) ? rk[r] : zero4);
  };
  auto commit_u = [&](int buf) {
    const u32x4 zero4 = {0, 0, 0, 0};
    const bool full = stlo + QC <= a.L;
#pragma unroll
    for (int r = 0; r < 2; r++) st16(&sm.U[buf][o_cu + 32 * 64 * r], (full || stlo + rowtok(rowu + 32 * r) < a.L) ? ru[r] : zero4);
  };
  const float Ah2 = a.A[h] * LOG2E;
  auto scalars = [&](int buf) {   // wave 0 only; lanes = rows of the staged chunk
    {
      const int t = stlo + rowtok(lane);
      const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
      rwv = okd ? (a.w_is_dt ? rdt : 1.f) : 0.f;
      rdt = okd ? rdt : 0.f;
      rda = oka ? rda : 0.f;
    }
    const float cs = wave_incl_scan_add(rda * Ah2);
    const float cs_end = wave_read_lane(cs, 63);
    sm.cs[buf][lane] = cs;
    sm.lw[buf][lane] = log2_fast(rwv) - cs;
    sm.ecs[buf][lane] = exp2_fast(cs);
    sm.ws[buf][lane] = rwv * exp2_fast(cs_end - cs);
    sm.dtl[buf][lane] = rdt;
  };

  int o_rd[4], o_mu[4], o_xu[4], o_ps[4], o_tu[2][2], o_tk[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    o_rd[i] = kx3(t16, 32 * i + 8 * g16);
    o_mu[i] = ux3(4 * g16 + (t16 >> 2), 16 * i + 4 * (t16 & 3));
    o_xu[i] = ux3(t16, 16 * i + 4 * g16);
    o_ps[i] = kx3(l31, 8 * i + 4 * h32);
  }
#pragma unroll
  for (int m = 0; m < 2; m++) {
    o_tk[m] = kx3(8 * h32 + (t16 >> 2) + 4 * m, 16 * (g16 & 1) + 4 * (t16 & 3));
#pragma unroll
    for (int ut = 0; ut < 2; ut++) o_tu[ut][m] = ux3(8 * h32 + (t16 >> 2) + 4 * m, 32 * ut + 16 * (g16 & 1) + 4 * (t16 & 3));
  }
  f32x16 accS[2];
#pragma unroll
  for (int ut = 0; ut < 2; ut++)
#pragma unroll
    for (int r = 0; r < 16; r++) accS[ut][r] = 0.f;
  if (a.init) {
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int k = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 32 * ut + l31;
        accS[ut][r] = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
      }
  }
  auto publish_state = [&]() {
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int rq4 = 0; rq4 < 4; rq4++) {
        u32x2 v;
        v[0] = pack_bf16x2(accS[ut][4 * rq4 + 0], accS[ut][4 * rq4 + 1]);
        v[1] = pack_bf16x2(accS[ut][4 * rq4 + 2], accS[ut][4 * rq4 + 3]);
        *reinterpret_cast<u32x2*>(&sm.S[(o_ps[rq4] ^ (w << 5)) + 32 * 128 * ut]) = v;
        if (PRECISE) {
          u32x2 l;
          l[0] = pack_bf16x2(accS[ut][4 * rq4 + 0] - bf_lo(v[0]), accS[ut][4 * rq4 + 1] - bf_hi(v[0]));
          l[1] = pack_bf16x2(accS[ut][4 * rq4 + 2] - bf_lo(v[1]), accS[ut][4 * rq4 + 3] - bf_hi(v[1]));
          *reinterpret_cast<u32x2*>(&sm.Slo[(o_ps[rq4] ^ (w << 5)) + 32 * 128 * ut]) = l;
        }
      }
  };
  // (1) acc = exp2(cs_l) * (Q . S_in) of the chunk whose scalars sit in buffer `buf` and whose Q fragments are in qf
  f32x4 acc[4];
  auto q_dot_s = [&](int buf) {
#pragma unroll
    for (int ut = 0; ut < 4; ut++) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int ut = 0; ut < 4; ut++) {
        const s16x8 fs = as_s16x8(ld16(&sm.S[o_rd[kk] + 16 * 128 * ut]));
        acc[ut] = mfma16x16x32_bf16(fs, as_s16x8(qf[kk]), acc[ut]);
        if (PRECISE) {
          const s16x8 fl = as_s16x8(ld16(&sm.Slo[o_rd[kk] + 16 * 128 * ut]));
          acc[ut] = mfma16x16x32_bf16(fl, as_s16x8(qf[kk]), acc[ut]);
        }
      }
    const float e1 = sm.ecs[buf][16 * w + t16];
#pragma unroll
    for (int ut = 0; ut < 4; ut++) acc[ut] *= e1;
  };

  // ---- prologue: chunk 0 staged, S_in published, acc = Q(0) . S_in
  stlo = chunk_lo(0);
  prefetch_q();
  prefetch_k();
  prefetch_u();
  commit_k();
  commit_u(0);
  if (w == 0) scalars(0);
  publish_state();
  if (!DFOLD && tid < 64) sm.Dv[tid] = a.D ? load_rt(a.D, (int64_t)h * a.Dsh + (int64_t)tid * a.Dsp, a.D_dt) : 0.f;
  const float Dh = (DFOLD && a.D) ? load_rt(a.D, (int64_t)h * a.Dsh, a.D_dt) : 0.f;
  block_sync();
  q_dot_s(0);
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
  const int erow = rowtok(16 * w + t16);
  const uint32_t eoff = (uint32_t)(erow * osl + 4 * g16);

  for (int c = 0; c < nC; c++) {
    const int cur = c & 1, nxt = cur ^ 1;
    const int tlo = chunk_lo(c);
    const int cnext = c + 1 < nC ? c + 1 : c;   // the last iteration re-stages its own chunk: no branch around loads
    if (EARLY) { stlo = chunk_lo(cnext); prefetch_u(); prefetch_k(); }
    // ---- (2) intra-chunk: G tiles -> M fragments (registers) -> M . U, on top of acc = exp2(cs_l) Q . S_in
    {
      const float cs_l = sm.cs[cur][16 * w + t16];
      constexpr bool HILO = MODE == GS_Y;        // the dx scan runs on the bf16 M alone (ssd_mfma.hip)
      auto block = [&](int kk, bool second, bool diag0, bool diag1) {
        u32x4 mh, ml = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (j == 1 && !second) { mh[2] = mh[3] = ml[2] = ml[3] = 0u; continue; }
          const int ta = 2 * kk + j;
          f32x4 gt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kq = 0; kq < 4; kq++) {
            const s16x8 fa = as_s16x8(ld16(&sm.K[o_rd[kq] + 16 * 128 * ta]));
            gt = mfma16x16x32_bf16(fa, as_s16x8(qf[kq]), gt);
          }
          const f32x4 lw4 = *reinterpret_cast<const f32x4*>(&sm.lw[cur][16 * ta + 4 * g16]);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            v[r] = gt[r] * exp2_fast(cs_l + lw4[r]);
            if (j == 0 ? diag0 : diag1) {
              if (DFOLD) v[r] = (4 * g16 + r < t16) ? v[r] : (4 * g16 + r == t16 ? v[r] + Dh : 0.f);
              else v[r] = (4 * g16 + r <= t16) ? v[r] : 0.f;
            }
          }
#pragma unroll
          for (int p2 = 0; p2 < 2; p2++) {
            const uint32_t hi = pack_bf16x2(v[2 * p2], v[2 * p2 + 1]);
            mh[2 * j + p2] = hi;
            if (HILO) ml[2 * j + p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi), v[2 * p2 + 1] - bf_hi(hi));
          }
        }
#pragma unroll
        for (int ut = 0; ut < 4; ut++) {
          const uint16_t* pu = &sm.U[cur][o_mu[ut] + 32 * 64 * kk];
          const s16x4 u0 = lds_read_tr16_b64(pu);
          const s16x4 u1 = lds_read_tr16_b64(pu + 16 * 64);
          s16x8 fu;
          fu[0] = u0[0]; fu[1] = u0[1]; fu[2] = u0[2]; fu[3] = u0[3]; fu[4] = u1[0]; fu[5] = u1[1]; fu[6] = u1[2]; fu[7] = u1[3];
          acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(mh), acc[ut]);
          if (HILO) acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(ml), acc[ut]);
        }
      };
      if (w == 0) { block(0, false, true, false); }
      else if (w == 1) { block(0, true, false, true); }
      else if (w == 2) { block(0, true, false, false); block(1, false, true, false); }
      else { block(0, true, false, false); block(1, true, false, true); }
    }
    // ---- epilogue of chunk c straight from the MFMA layout (acc is free afterwards)
    {
      const int trow = tlo + erow;
      if (trow < a.L) {
        const float dts = MODE == GS_DX ? sm.dtl[cur][16 * w + t16] : 1.f;
        uint16_t* oc = ob + (int64_t)tlo * osl;
#pragma unroll
        for (int ut = 0; ut < 4; ut++) {
          f32x4 v = acc[ut];
          if (!DFOLD) {
            const u32x2 xr = *reinterpret_cast<const u32x2*>(&sm.U[cur][o_xu[ut] + 16 * 64 * w]);
            const f32x4 Du = *reinterpret_cast<const f32x4*>(&sm.Dv[16 * ut + 4 * g16]);
            v = acc[ut] * dts + Du * f32x4{bf_lo(xr[0]), bf_hi(xr[0]), bf_lo(xr[1]), bf_hi(xr[1])};
          }
          u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(oc + 16 * ut + eoff) = o;
        }
      }
    }
    // ---- loads of the next chunk: after the register peak of the intra phase, in flight during the state update
    stlo = chunk_lo(cnext);
    prefetch_q();
    if (!EARLY) { prefetch_u(); prefetch_k(); }
    // ---- (3) state update: S^T[k][u] = exp2(cs_end) S^T + sum_l (ws_l K^T[k][l]) U[l][u]
    {
      const float dec = sm.ecs[cur][QC - 1];
#pragma unroll
      for (int ut = 0; ut < 2; ut++) accS[ut] *= dec;
#pragma unroll
      for (int ls = 0; ls < 4; ls++) {
        s16x8 fk;
        {
          const s16x4 k0 = lds_read_tr16_b64(&sm.K[(o_tk[0] ^ (w << 5)) + 16 * 128 * ls]);
          const s16x4 k1 = lds_read_tr16_b64(&sm.K[(o_tk[1] ^ (w << 5)) + 16 * 128 * ls]);
          fk[0] = k0[0]; fk[1] = k0[1]; fk[2] = k0[2]; fk[3] = k0[3]; fk[4] = k1[0]; fk[5] = k1[1]; fk[6] = k1[2]; fk[7] = k1[3];
        }
        const int lb = 16 * ls + 8 * h32;
        const f32x4 s0v = *reinterpret_cast<const f32x4*>(&sm.ws[cur][lb]);
        const f32x4 s1v = *reinterpret_cast<const f32x4*>(&sm.ws[cur][lb + 4]);
        u32x4 kp, kl = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          const f32x2 kv = {bf16_to_f32((uint16_t)fk[2 * e2]), bf16_to_f32((uint16_t)fk[2 * e2 + 1])};
          const f32x2 sv = e2 < 2 ? f32x2{s0v[2 * e2], s0v[2 * e2 + 1]} : f32x2{s1v[2 * e2 - 4], s1v[2 * e2 - 3]};
          const f32x2 pr = kv * sv;
          kp[e2] = pack_bf16x2(pr[0], pr[1]);
          if (PRECISE) kl[e2] = pack_bf16x2(pr[0] - bf_lo(kp[e2]), pr[1] - bf_hi(kp[e2]));
        }
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
          s16x8 fu;
          {
            const s16x4 u0 = lds_read_tr16_b64(&sm.U[cur][o_tu[ut][0] + 16 * 64 * ls]);
            const s16x4 u1 = lds_read_tr16_b64(&sm.U[cur][o_tu[ut][1] + 16 * 64 * ls]);
            fu[0] = u0[0]; fu[1] = u0[1]; fu[2] = u0[2]; fu[3] = u0[3]; fu[4] = u1[0]; fu[5] = u1[1]; fu[6] = u1[2]; fu[7] = u1[3];
          }
          accS[ut] = mfma32x32x16_bf16(as_s16x8(kp), fu, accS[ut]);
          if (PRECISE) accS[ut] = mfma32x32x16_bf16(as_s16x8(kl), fu, accS[ut]);
        }
      }
    }
    // ---- publish S_out (every wave left S_in behind before barrier B of the previous iteration), next chunk's U and scalars
    publish_state();
    commit_u(nxt);
    if (w == 0) scalars(nxt);
    block_sync();   // A: K and U[cur] of chunk c are free; S_out, U[nxt], the scalars of chunk c + 1 are visible
    commit_k();
    q_dot_s(nxt);   // acc of chunk c + 1 (unused after the last chunk)
    block_sync();   // B: K of chunk c + 1 is visible; every wave is done with S
  }
  if (a.fin) {
    const float Ah = a.A[h];
    const float extra = a.fin_extra_decay ? expf(dtrow[0] * Ah) : 1.f;
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int k = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 32 * ut + l31;
        a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)u * a.fsu + (int64_t)k * a.fsk] = accS[ut][r] * extra;
      }
  }
}

bool ssd_v6_applies(const GScan& g) {
  if (!((g.flags & GSF_PRECISE) && g.mode == GS_Y) || ssd_a8_applies(g)) return false;   // (the plain scans: PRECISE instantiation of ssd_a8.hip)
  if (g.Z.p || g.outx || g.prof) return false;                   // gate / pre-gate copy stay on a3
  if (g.seg && ssd_segments(g.B * g.H, g.L).nseg > 1) return false;   // so do split sequences
  return true;
}

int ssd_v6_launch(const GScan& g, omk_stream stream) {
  GScan a = g;
  a.nseg = 1; a.cps = (a.L + QC6 - 1) / QC6;
  dim3 grid((unsigned)(a.B * a.H)), block(256);
  const size_t smem = sizeof(SmemA6);
#define OMK_A6P(DF_) do { \
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a6_kernel<GS_Y, DF_, 2, true, true>), smem)) return fail(OMK_ELAUNCH, "ssd_v6: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_mfma_a6_kernel<GS_Y, DF_, 2, true, true>), grid, block, smem, stream, a); } while (0)
  if (!a.D || a.Dsp == 0) OMK_A6P(true); else OMK_A6P(false);
#undef OMK_A6P
  return OMK_OK;
}

}  // namespace omk
