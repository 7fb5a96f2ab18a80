// ssd_mfma.hip -- the MFMA chunked scan for the OmniMamba-1.3B block shape (headdim 64, d_state 128, bf16).
//
// A workgroup walks its sequence in chunks of 64 tokens and carries the running state (64 x 128 fp32 per head) in MFMA
// accumulator registers from the first token to the last: states never touch HBM, so HBM traffic is the algorithmic
// minimum -- every x / dy row read once, every y / dx row written once, B / C (shared by all 64 heads of a group)
// served from the XCD's L2.
//
// Per chunk (local token l, s; a = log-decay, cs = inclusive prefix of a inside the chunk, w = input scale):
//   G^T[s][l] = K_s . Q_l                                    16x16x32 MFMA, stays in registers
//   M[l][s]   = G[l][s] * exp2(cs_l + lw_s)  (s <= l)         lw = log2 w - cs; VALU, hi + lo bf16 -> next MFMA operand
//   O_l       = sum_s M[l][s] U_s  +  exp2(cs_l) * (Q_l . S_in)   U via permuted ds_read_b64_tr_b16
//   S_out     = exp2(cs_63) S_in + sum_l (w_l exp2(cs_63 - cs_l) U_l) (x) K_l   MFMA, both operands via transpose reads
// The intra-chunk prefix cs is a DPP wave scan; lanes = tokens.  S_in reaches the O chain as bf16 through LDS.
// Time-reversed scans (backward) only change the global<->LDS row mapping and the decay index.
//
// Two kernels: ssd_mfma_a3_kernel (class A: y and dx, U per head and 64 wide) and ssd_mfma_b3_kernel (class B: dC and
// dB, U shared by the group and 128 wide).  Earlier designs of this round (two heads per workgroup with G and O through
// LDS; 32x32 output tiles; strips split into intra / state waves) were measured slower and live in the git history --
// DESIGN.md section 4 has the numbers.
#include <cstdlib>
#include "ssd_scan.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int QC = 64;     // chunk length (tokens)
#ifndef OMK_SSD_KHILO_DEFAULT
#define OMK_SSD_KHILO_DEFAULT 0   // decided by measurement (profiles/r03_khilo.txt); OmkSsdFwd::flags & OMK_SSD_KHILO asks for it per call
#endif
#ifndef OMK_SSD_BWD_HILO
#define OMK_SSD_BWD_HILO 0   // 1: the dx scan also splits M into bf16 hi + lo (the forward and the dB / dC scans always do)
#endif

// =========================================================================================================
// class A ("row strips"): one head per 256-thread workgroup, two workgroups per CU.
//
// Wave w owns the output rows l in [16 w, 16 w + 16) of every 64-token chunk (all 64 columns) and the state rows
// k in [32 w, 32 w + 32) (all 64 columns).  What that buys over the first designs of this round (G and O through LDS):
//   * G never leaves registers.  G^T[s][l] = K_s . Q_l comes out of the 16x16x32 MFMA with the lane's own l in every
//     register; the MFMA contraction index may be permuted freely as long as both operands agree, so the two G tiles of
//     a 32-wide s block ARE the A-operand fragment of M (after the decay/mask/bf16 hi+lo split), and U is fetched with
//     the matching permuted transpose reads.  No fp32 G tile in LDS, no barrier between G and M.
//   * M is built exactly once per chunk.
//   * Q is never staged: its MFMA fragments (rows of the wave's strip) are loaded from global memory one chunk ahead
//     and serve both G (as B operand) and Q . S_in (as A operand).
//   * O = exp2(cs_l) (Q . S_in) + M . U accumulates in ONE register tile per wave and goes to HBM from registers.
//   * K / U tiles are double buffered: two barriers per chunk (after the S_in reads, after the S_out publish).
// =========================================================================================================
// LDS layouts: unpadded rows, the 16-byte segment index XOR-ed with a function of the row so that every
// access pattern of the kernel is bank-conflict free under the gfx950 lane groups (MI355X_MICROARCH.md, LDS):
//   K / S tiles (256 B rows): seg ^ swzK(row).  ds_read_b128 "row t16, segment 4 c + g16" (lane groups
//     {0-3,12-15,20-27}, ...) needs swzK bijective on row & 15 with swzK({4..11}) closed under ^1; the transpose reads
//     "4 rows x 4 segments per half wave" need swzK(row) >> 2 distinct over 4 consecutive rows.
//   64-column tiles (128 B rows, two rows per 256 B bank row): seg ^ swzU(row) with swzU = swzK & 7 -- searched the same
//     way over every pattern the class A / class B kernels use on them (8 rows x 2 segments, 4 rows x 4 segments,
//     16 rows x 1 segment, the ds_read_b128 row reads, rows {0-3, 8-11} x 2 segments).
struct SmemA3 {
  uint16_t K[2][QC * 128];
  uint16_t U[2][QC * 64];
  uint16_t S[64 * 128];      // [u][k] bf16 copy of S_in
  float cs[2][QC], lw[2][QC], ecs[2][QC], ws[2][QC], dtl[2][QC];   // lw = log2(w) - cs  (M = G exp2(cs_l + lw_s))
  float Dv[64];
};
static_assert(sizeof(SmemA3) <= 80 * 1024, "two workgroups must fit the 160 KB of a CU");

// EXTRAS: gate z and / or the pre-gate copy of y (forward without the gated norm).
// STATE: the state-only pass of a split sequence (ssd_scan.h, GScan::seg): segments 0 .. nseg - 2, no output, the
// segment's end state from a zero start and its total decay go to the workspace; the scan proper (STATE = false,
// nseg workgroups per head) starts every segment from the fold of the earlier segments' states.
// DFOLD (forward, one D per head): D x_l rides on the diagonal of M (M_ll += D, kept to 16 bits by the hi + lo split)
// instead of an epilogue that re-reads x and D from LDS.
// DUMP (forward of a training step): the scan leaves the carried state in front of every 128-token window behind (GScan::dump), as the
// dx scan and the state-only pass do -- a separate instantiation so that the inference forward carries none of it.
// KHILO (forward, OMK_SSD_KHILO): the w_l K_l operand of the state update enters as a bf16 hi + lo pair (8 more 32x32x16 MFMAs per
// wave and chunk, no LDS): the carried state -- and with it final_states -- is then exact to fp32 accumulation instead of carrying
// one bf16 rounding per chunk (1.7e-3 -> 1e-5 on the final state; y of slow-decay heads 1.4e-3 -> 1.0e-3).
template <int MODE, bool EXTRAS, bool STATE, bool DFOLD = false, bool KHILO = false, bool DUMP = false>
__global__ __launch_bounds__(256, 2) void ssd_mfma_a3_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemA3& sm = *reinterpret_cast<SmemA3*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform_i(tid >> 6);
  const int h32 = lane >> 5, l31 = lane & 31, g16 = lane >> 4, t16 = lane & 15;
  // XCD-aware order: consecutive workgroup ids go round-robin over the 8 XCDs; give every XCD a contiguous range of
  // (batch, head) so the heads that share B / C rows also share an L2
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  // workgroup order (b, segment, h): the heads of one (batch, segment) share B / C rows and an XCD
  // spass: the state-only pass that dumps window states starts like a scan (every segment, folded start states) and leaves no
  // segment state behind
  const bool spass = STATE && (a.dump != nullptr || a.state_only != 0);
  const int nwseg = (STATE && !spass) ? a.nseg - 1 : a.nseg;
  const int h = vid % a.H, seg = (vid / a.H) % nwseg, b = vid / (a.H * nwseg);
  const int g = h / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
  const int c0 = seg * a.cps, c1 = (c0 + a.cps < nC) ? c0 + a.cps : nC;   // chunks of this workgroup, scan order
  const bool rev = a.reverse != 0;
  // chunk row i <-> token tlo + (rev ? 63 - i : i), tlo = 64 * chunk id
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QC; };
  auto rowtok = [&](int i) -> int { return rev ? QC - 1 - i : i; };

  // ---- staging lanes: K four 16-byte segments per thread (rows rowk + 16 r), U two (rows rowu + 32 r), Q fragments
  const int rowk = tid >> 4, ck8 = (tid & 15) * 8, rowu = tid >> 3, cu8 = (tid & 7) * 8;
  const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
  const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
  const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  const int ksl = (int)a.K.sl, qsl = (int)a.Q.sl, usl = (int)a.U.sl, osl = (int)a.osl;
  // Staging through BUFFER loads (ssd_tiles.h): lane offset + scalar offset + resource, the address is formed by the memory
  // pipeline -- no 64-bit VALU adds, and rows behind the end of the sequence (ragged last chunk) come back as zeros by the
  // range check, so neither clamped addresses nor selects.  Chunk row i <-> token tlo + (rev ? 63 - i : i); the lane part of the
  // row goes to the lane offset, the per-load part (always >= 0) to the scalar offset.
  const BufRes Kr = make_buf(Kb, (uint32_t)((int64_t)a.L * ksl * 2)), Qr = make_buf(Qb, (uint32_t)((int64_t)a.L * qsl * 2));
  const BufRes Ur = make_buf(Ub, (uint32_t)((int64_t)a.L * usl * 2)), Dr = make_buf(dtrow, (uint32_t)((int64_t)a.L * 4));
  const uint32_t kvo = 2u * (uint32_t)((rev ? 15 - rowk : rowk) * ksl + ck8), uvo = 2u * (uint32_t)((rev ? 31 - rowu : rowu) * usl + cu8);
  const uint32_t qvo = 2u * (uint32_t)(rowtok(16 * w + t16) * qsl + 8 * g16);
  const uint32_t dvo = 4u * (uint32_t)rowtok(lane), dvo_a = 4u * (uint32_t)(rowtok(lane) + (rev ? 1 : 0));
  u32x4 rk[4], ru[2], qf[4];
  float rdt = 0.f, rda = 0.f, rwv = 0.f;
  int stlo = 0;   // tlo of the chunk held in the staging registers
  // The loads of the next chunk are spread over the phases of the current one (a burst of ten 1 KB loads per wave
  // stalls on the 64 B/clk address path): K after barrier X, Q fragments, U and dt after the intra phase.
  auto prefetch_k = [&]() {
    const uint32_t so = 2u * (uint32_t)(stlo * ksl);
#pragma unroll
    for (int r = 0; r < 4; r++) rk[r] = buf_ld16(Kr, kvo, so + 2u * (uint32_t)((rev ? 16 * (3 - r) : 16 * r) * ksl));
  };
  auto prefetch_q = [&]() {   // straight into the live fragment registers: issued after their last use of the chunk
    const uint32_t so = 2u * (uint32_t)(stlo * qsl);
#pragma unroll
    for (int kk = 0; kk < 4; kk++) qf[kk] = buf_ld16(Qr, qvo, so + 64u * kk);
  };
  auto prefetch_u = [&]() {
    const uint32_t so = 2u * (uint32_t)(stlo * usl);
#pragma unroll
    for (int r = 0; r < 2; r++) ru[r] = buf_ld16(Ur, uvo, so + 2u * (uint32_t)((rev ? 32 * (1 - r) : 32 * r) * usl));
    // token scalars (consumed by wave 0, loaded by every wave to keep the instruction stream uniform): lanes = rows
    rdt = buf_ld_f32(Dr, dvo, 4u * (uint32_t)stlo);
    rda = buf_ld_f32(Dr, dvo_a, 4u * (uint32_t)stlo);
  };
  const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
  auto commit = [&](int buf) {   // rows past the end arrived as zeros
#pragma unroll
    for (int r = 0; r < 4; r++) st16(&sm.K[buf][o_ck + 16 * 128 * r], rk[r]);
#pragma unroll
    for (int r = 0; r < 2; r++) st16(&sm.U[buf][o_cu + 32 * 64 * r], ru[r]);
  };
  const float Ah = a.A[h];
  const float Ah2 = Ah * LOG2E;
  float segdec = 0.f;   // STATE: log2 of the segment's total decay (wave 0)
  auto scalars = [&](int buf, bool fresh) {   // wave 0 only; lanes = rows of the staged chunk (fresh: not a re-stage)
    {
      const int t = stlo + rowtok(lane);
      const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
      rwv = okd ? (a.w_is_dt ? rdt : 1.f) : 0.f;
      rdt = okd ? rdt : 0.f;
      rda = oka ? rda : 0.f;
    }
    const float cs = wave_incl_scan_add(rda * Ah2);
    const float cs_end = wave_read_lane(cs, 63);
    if (STATE && fresh) segdec += cs_end;
    sm.cs[buf][lane] = cs;
    sm.lw[buf][lane] = log2_fast(rwv) - cs;
    sm.ecs[buf][lane] = exp2_fast(cs);
    sm.ws[buf][lane] = rwv * exp2_fast(cs_end - cs);
    sm.dtl[buf][lane] = rdt;
  };

  // ---- lane-constant LDS element offsets (the swizzle only depends on row & 15, so tile / chunk-row-block / buffer
  // selection stays an immediate or scalar offset on top of these)
  int o_rd[4], o_mu[4], o_xu[4], o_ps[4], o_tu[2][2], o_tk[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    o_rd[i] = kx3(t16, 32 * i + 8 * g16);                         // b128 row reads of K (G) and S (Q.S): row t16
    o_mu[i] = ux3(4 * g16 + (t16 >> 2), 16 * i + 4 * (t16 & 3));  // permuted transpose reads of U for M.U
    o_xu[i] = ux3(t16, 16 * i + 4 * g16);                         // x of the lane's output row
    o_ps[i] = kx3(l31, 8 * i + 4 * h32);                          // publish: S[u = l31][k = 8 i + 4 h32 ..]
  }
#pragma unroll
  for (int m = 0; m < 2; m++) {
    o_tk[m] = kx3(8 * h32 + (t16 >> 2) + 4 * m, 16 * (g16 & 1) + 4 * (t16 & 3));
#pragma unroll
    for (int ut = 0; ut < 2; ut++) o_tu[ut][m] = ux3(8 * h32 + (t16 >> 2) + 4 * m, 32 * ut + 16 * (g16 & 1) + 4 * (t16 & 3));
  }
  // ---- running state: S^T[32 w + ..][32 ut + ..], ut = 0, 1
  f32x16 accS[2];
#pragma unroll
  for (int ut = 0; ut < 2; ut++)
#pragma unroll
    for (int r = 0; r < 16; r++) accS[ut][r] = 0.f;
  const int64_t bh = (int64_t)b * a.H + h;
  // accumulator order of the segment states: element (w, ut, r, lane) at ((2 w + ut) * 16 + r) * 64 + lane
  const int segoff = (2 * w * 16) * 64 + lane;
  if ((!STATE || spass) && seg > 0) {   // folded by ssd_seg_fold_kernel: slot seg - 1 = state at the start of this segment
    const float* sp = a.seg + (bh * a.nseg + seg - 1) * SEG_STATE + segoff;
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int r = 0; r < 16; r++) accS[ut][r] = sp[(ut * 16 + r) * 64];
  }
  if (a.init && (!STATE || spass) && seg == 0) {
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
        *reinterpret_cast<u32x2*>(&sm.S[(o_ps[rq4] ^ (w << 5)) + 32 * 128 * ut]) = v;   // column block 32 w = segment bits 2-3
      }
  };

  stlo = chunk_lo(c0);
  if (!STATE) prefetch_q();
  prefetch_k();
  prefetch_u();
  commit(0);
  if (w == 0) scalars(0, true);
  if (spass && a.dump) publish_state();
  if (!STATE) {
    publish_state();
    if (!DFOLD && tid < 64) sm.Dv[tid] = a.D ? load_rt(a.D, (int64_t)h * a.Dsh + (int64_t)tid * a.Dsp, a.D_dt) : 0.f;
  }
  const float Dh = (DFOLD && a.D) ? load_rt(a.D, (int64_t)h * a.Dsh, a.D_dt) : 0.f;
  block_sync();
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
  const BufRes Or = make_buf(STATE ? nullptr : ob, STATE ? 0u : (uint32_t)((int64_t)a.L * osl * 2));
  uint16_t* oxb = a.outx ? (uint16_t*)a.outx + (int64_t)b * a.osb + (int64_t)h * a.osh : nullptr;
  const uint16_t* zb = (MODE == GS_Y && a.Z.p) ? (const uint16_t*)a.Z.p + (int64_t)b * a.Z.sb + (int64_t)h * a.Z.sh : nullptr;
  const int zsl = (int)a.Z.sl;
  // the lane's output row l = 16 w + t16; register r of tile ut is column 16 ut + 4 g16 + r  (8-byte stores)
  const int erow = rowtok(16 * w + t16);
  const uint32_t eoff = (uint32_t)(erow * osl + 4 * g16), zoff = (uint32_t)(erow * zsl + 4 * g16);

#ifdef OMK_PHASE_PROF   // developer build (tools/phase_prof.py): s_memtime deltas per phase, workgroup 0
  uint64_t pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
#define PT3(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
  uint64_t tprev = prof ? clock64_() : 0;
  const uint64_t t_core0 = tprev, t_ref0 = prof ? __builtin_readsteadycounter() : 0;   // core clock vs 100 MHz reference
  const int abl = a.ablate;   // developer experiment: skip phases (bit i), results are wrong
#else
#define PT3(i) do { } while (0)
  constexpr int abl = 0;
#endif
  OMK_VM_DRAIN();
  for (int c = c0; c < c1; c++) {
    const int cur = (c - c0) & 1, nxt = cur ^ 1;
    const int tlo = chunk_lo(c);
    const int cnext = c + 1 < c1 ? c + 1 : c;   // the last iteration re-stages its own chunk: no branch around loads
    PT3(0);
    if ((MODE == GS_DX || STATE || DUMP) && a.dump) {   // (compiled out of the forward scan) window-boundary image of the state in front of this chunk (sm.S, published before the last barrier)
      const int cid = rev ? nC - 1 - c : c;
      const bool here = rev ? (cid == nC - 1 || (cid & 1)) : !(cid & 1);
      if (here) {
        uint16_t* dp = a.dump + ((((int64_t)b * a.dump_nw + (cid >> 1)) * a.H + h) << 13);   // [b][window][h]: a window's heads are contiguous
#pragma unroll
        for (int i = 0; i < 4; i++) st16(dp + (tid + 256 * i) * 8, ld16(&sm.S[img_off(tid + 256 * i)]));   // (segment order of the images: ssd_tiles.h)
        if (STATE) block_sync();   // (the scan proper has barrier X between these reads and the next publish)
      }
    }
    OMK_ISA_MARK("1 Q.S_in (16 MFMA) + scale");
    // ---- (1) O = exp2(cs_l) * (Q . S_in): A = Q fragments (registers), B = S_in[k][u] as [u][k] bf16 rows
    f32x4 acc[4];
#pragma unroll
    for (int ut = 0; ut < 4; ut++) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!STATE) {   // (the state-only pass is: stage, state update, one barrier)
      if (!(abl & 1))
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
#pragma unroll
        for (int ut = 0; ut < 4; ut++) {
          const s16x8 fs = as_s16x8(ld16(&sm.S[o_rd[kk] + 16 * 128 * ut]));
          acc[ut] = mfma16x16x32_bf16(fs, as_s16x8(qf[kk]), acc[ut]);   // O^T[u][l]: 4 consecutive u per lane
        }
      const float e1 = sm.ecs[cur][16 * w + t16];
#pragma unroll
      for (int ut = 0; ut < 4; ut++) acc[ut] *= e1;
      PT3(1);
      block_sync();   // X: every wave is done with S_in
    }
    PT3(2);
    OMK_ISA_MARK("2 prefetch K / U of the next chunk");
    stlo = chunk_lo(cnext);
    prefetch_k();
    prefetch_u();
    OMK_ISA_MARK("3 intra: G tiles, M build (decay, mask, hi + lo), M.U");
    // ---- (2) intra-chunk: G tiles -> M fragments (registers) -> M . U
    if (!STATE) {
      const float cs_l = sm.cs[cur][16 * w + t16];
      // the low half of M where the 1e-3 budget of y asks for it; the dx scan (5e-3, no token scalars) runs on the bf16 M alone
      constexpr bool HILO = MODE == GS_Y || OMK_SSD_BWD_HILO;
      auto block = [&](int kk, bool second, bool diag0, bool diag1) {
        // tiles ta = 2 kk + j (j = 0, 1) of G^T: lane holds s = 16 ta + 4 g16 + r (r = 0..3) for its own l = 16 w + t16
        u32x4 mh, ml = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (j == 1 && !second) { mh[2] = mh[3] = ml[2] = ml[3] = 0u; continue; }
          const int ta = 2 * kk + j;
          f32x4 gt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kq = 0; kq < 4; kq++) {
            const s16x8 fa = as_s16x8(ld16(&sm.K[cur][o_rd[kq] + 16 * 128 * ta]));
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
          for (int p2 = 0; p2 < 2; p2++) {   // bf16 hi + lo: the rounding of M dominates the error of y otherwise
            const uint32_t hi = pack_bf16x2(v[2 * p2], v[2 * p2 + 1]);
            mh[2 * j + p2] = hi;
            if (HILO) ml[2 * j + p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi), v[2 * p2 + 1] - bf_hi(hi));
          }
        }
#pragma unroll
        for (int ut = 0; ut < 4; ut++) {
          // B operand with the SAME permuted contraction order: rows 32 kk + 16 j + 4 g16 + {0..3}, column 16 ut + t16
          const uint16_t* pu = &sm.U[cur][o_mu[ut] + 32 * 64 * kk];
          const s16x4 u0 = lds_read_tr16_b64(pu);
          const s16x4 u1 = lds_read_tr16_b64(pu + 16 * 64);   // + 16 rows: same swizzle
          s16x8 fu;
          fu[0] = u0[0]; fu[1] = u0[1]; fu[2] = u0[2]; fu[3] = u0[3]; fu[4] = u1[0]; fu[5] = u1[1]; fu[6] = u1[2]; fu[7] = u1[3];
          acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(mh), acc[ut]);
          if (HILO) acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(ml), acc[ut]);
        }
      };
      // strip w: s blocks kk = 0 .. w >> 1; the diagonal tile is ta = w, tiles ta > w are empty
      if (abl & 2) { }
      else if (w == 0) { block(0, false, true, false); }
      else if (w == 1) { block(0, true, false, true); }
      else if (w == 2) { block(0, true, false, false); block(1, false, true, false); }
      else { block(0, true, false, false); block(1, true, false, true); }
    }
    PT3(3);
    OMK_ISA_MARK("4 prefetch Q + state update (decay, w K operand, 8 MFMA 32x32)");
    if (!STATE) prefetch_q();
    // ---- (3) state update: S^T[k][u] = exp2(cs_end) S^T + sum_l (ws_l K^T[k][l]) U[l][u]
    if (!(abl & 4)) {
      const float dec = sm.ecs[cur][QC - 1];
#pragma unroll
      for (int ut = 0; ut < 2; ut++) accS[ut] *= dec;
#pragma unroll
      for (int ls = 0; ls < 4; ls++) {
        s16x8 fk;
        {
          const s16x4 k0 = lds_read_tr16_b64(&sm.K[cur][(o_tk[0] ^ (w << 5)) + 16 * 128 * ls]);
          const s16x4 k1 = lds_read_tr16_b64(&sm.K[cur][(o_tk[1] ^ (w << 5)) + 16 * 128 * ls]);
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
          if (KHILO) kl[e2] = pack_bf16x2(pr[0] - bf_lo(kp[e2]), pr[1] - bf_hi(kp[e2]));
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
          if (KHILO) accS[ut] = mfma32x32x16_bf16(as_s16x8(kl), fu, accS[ut]);
        }
      }
    }
    PT3(4);
    OMK_ISA_MARK("5 publish S_out (bf16) + commit next chunk + scalars (wave 0)");
    // ---- (4) publish S_out, stage the next chunk, next chunk's scalars
    // (the state-only dump pass publishes only what the next iteration dumps: forward, the state in front of an even chunk)
    if ((!STATE || (spass && a.dump && (rev ? ((nC - 2 - c) & 1) != 0 : ((c + 1) & 1) == 0))) && !(abl & 8)) publish_state();
    commit(nxt);
    if (w == 0) scalars(nxt, c + 1 < c1);
    PT3(5);
    OMK_ISA_MARK("6 epilogue: convert + store O");
    // ---- epilogue from the MFMA layout: 4 consecutive columns of one row per (lane, ut); buffer stores drop rows >= L
    if (!STATE && !(abl & 16)) {
      const float dts = MODE == GS_DX ? sm.dtl[cur][16 * w + t16] : 1.f;
      const uint32_t oso = 2u * (uint32_t)(tlo * osl);
      const bool live = EXTRAS ? (tlo + erow < a.L) : true;      // the gate / pre-gate copy still go through plain pointers
#pragma unroll
      for (int ut = 0; ut < 4; ut++) {
        f32x4 v = acc[ut];
        if (!DFOLD) {
          const u32x2 xr = *reinterpret_cast<const u32x2*>(&sm.U[cur][o_xu[ut] + 16 * 64 * w]);
          const f32x4 Du = *reinterpret_cast<const f32x4*>(&sm.Dv[16 * ut + 4 * g16]);   // D of columns 16 ut + 4 g16 + r
          v = acc[ut] * dts + Du * f32x4{bf_lo(xr[0]), bf_hi(xr[0]), bf_lo(xr[1]), bf_hi(xr[1])};
        }
        if (MODE == GS_Y && EXTRAS && live) {
          if (oxb) {
            u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(oxb + (int64_t)tlo * osl + 16 * ut + eoff) = o;
          }
          if (zb) {
            const u32x2 zr = *reinterpret_cast<const u32x2*>(zb + (int64_t)tlo * zsl + 16 * ut + zoff);
            v[0] *= silu_fast(bf_lo(zr[0])); v[1] *= silu_fast(bf_hi(zr[0]));
            v[2] *= silu_fast(bf_lo(zr[1])); v[3] *= silu_fast(bf_hi(zr[1]));
          }
        }
        const u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        buf_st8(Or, o, 2u * eoff, oso + 32u * ut);
      }
    }
    PT3(6);
    OMK_ISA_MARK("7 barrier Y + loop");
    block_sync();   // Y: tiles, scalars and the bf16 state of the next chunk are visible
    PT3(7);
  }
#ifdef OMK_PHASE_PROF
  if (prof) { pt[10] = clock64_() - t_core0; pt[11] = __builtin_readsteadycounter() - t_ref0; }
  if (prof && lane == 0)
    for (int i = 0; i < 12; i++) a.prof[w * 12 + i] = pt[i];
#endif
  if (STATE && !spass) {
    float* sp = a.seg + (bh * a.nseg + seg) * SEG_STATE + segoff;
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int r = 0; r < 16; r++) sp[(ut * 16 + r) * 64] = accS[ut][r];
    if (tid == 0) a.seg[(int64_t)a.B * a.H * a.nseg * SEG_STATE + bh * a.nseg + seg] = segdec;
    return;
  }
  if (a.fin && seg == a.nseg - 1) {
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

// Segment states -> start states, in place: slot j (end state of segment j from a zero start, accumulator order) becomes
// the state at the START of segment j + 1: run = exp2(dec_j) run + slot_j, run_0 = the caller's initial state.
// One thread = one state element of one (b, h); nseg - 1 sequential steps.
__global__ void ssd_seg_fold_kernel(GScan a) {
  const int64_t bh = blockIdx.x / (SEG_STATE / 256);
  const int e = (blockIdx.x % (SEG_STATE / 256)) * 256 + threadIdx.x;
  const int b = (int)(bh / a.H), h = (int)(bh % a.H);
  float run = 0.f;
  if (a.init) {   // element e = ((2 w + ut) * 16 + r) * 64 + lane of the class A accumulators
    const int lane = e & 63, r = (e >> 6) & 15, ut = (e >> 10) & 1, w = e >> 11;
    const int k = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), u = 32 * ut + (lane & 31);
    run = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
  }
  const float* sdec = a.seg + (int64_t)a.B * a.H * a.nseg * SEG_STATE + bh * a.nseg;
  float* sp = a.seg + bh * a.nseg * SEG_STATE + e;
  for (int j = 0; j + 1 < a.nseg; j++) {
    run = exp2_fast(sdec[j]) * run + sp[(int64_t)j * SEG_STATE];
    sp[(int64_t)j * SEG_STATE] = run;
  }
}

// =========================================================================================================
// class B: the row-strip design of ssd_mfma_a3_kernel for the dC / dB scans.  One 512-thread workgroup owns
// a head PAIR (the unit of the fp32 partial tiles): waves 0-3 are the strips of head 0, waves 4-7 of head 1; the group
// rows U (B or C) and X4 (C or B) are staged once for both.  Wave (hh, w):
//   output   O^T[n][l]  l in [16 w, 16 w + 16), all 128 n: eight 16x16 register tiles (G, M never leave registers);
//   state    S^T[p][n]  p in [16 w, 16 w + 16), all 128 n: eight 16x16 register tiles, updated with 16x16x32 MFMAs;
//   token scalar  e_l / w_l = X4_l . O_l : 32 products per lane + two cross-lane adds, stored straight to HBM;
//   head 1 hands its (dt-scaled) tiles to head 0 through a lane-linear LDS buffer; head 0 adds and stores the partial.
// Tiles are single buffered: everything that reads U / X4 / K / S / the scalars of a chunk sits before barrier E, the
// commit of the next chunk after it.  Three barriers per chunk (X: S_in reads done, E: exchange, Y: next chunk visible).
// =========================================================================================================
struct SmemB3 {
  uint16_t U[QC * 128];        // group rows, kx3 swizzle
  uint16_t X4[QC * 128];
  uint16_t K[2][QC * 64];      // per head, ux3 swizzle
  uint16_t S[2][128 * 64];     // per head [n][p] bf16 copy of S_in, ux3 swizzle
  float O[4 * 8 * 64 * 4];     // head 1 -> head 0: [strip][ut][lane] float4, lane-linear
  float cs[2][QC], lw[2][QC], ecs[2][QC], ws[2][QC], dtl[2][QC];
  float bred[8];
  float dred[2][4][64];        // dD fold: [head][strip][column]
};

template <int MODE, int DMODE>   // DMODE (dB scan): 0 no dD, 1 dD per head (one running sum), 2 dD per (head, column)
__global__ __launch_bounds__(512) void ssd_mfma_b3_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemB3& sm = *reinterpret_cast<SmemB3*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_i(tid >> 6);
  const int hh = wave >> 2, w = wave & 3;
  const int h32 = lane >> 5, g16 = lane >> 4, t16 = lane & 15;
  (void)h32;
  const int pairs = a.H / 2;
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous (batch, pair)
  // workgroup order (b, segment, pair): the pairs of one (batch, segment) share the group rows and an XCD
  const int hp = vid % pairs, seg = (vid / pairs) % a.nseg, b = vid / (pairs * a.nseg);
  const int h0 = hp * 2, hcur = h0 + hh;
  const int g = h0 / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
  const int c0 = seg * a.cps, c1 = (c0 + a.cps < nC) ? c0 + a.cps : nC;   // chunks of this workgroup, scan order
  const bool rev = a.reverse != 0;
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QC; };
  auto rowtok = [&](int i) -> int { return rev ? QC - 1 - i : i; };

  // ---- staging: U, X4 two 16-byte segments per thread (rows rowg + 32 r); K one segment per thread per head
  const int rowg = tid >> 4, cg8 = (tid & 15) * 8, rowk = tid >> 3, ck8 = (tid & 7) * 8;
  const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)g * a.U.sh;
  const uint16_t* Xb = (const uint16_t*)a.X4.p + (int64_t)b * a.X4.sb + (int64_t)g * a.X4.sh;
  const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)h0 * a.K.sh;
  const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)hcur * a.Q.sh;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + hcur) * a.L;
  const int usl = (int)a.U.sl, xsl = (int)a.X4.sl, ksl = (int)a.K.sl, qsl = (int)a.Q.sl, ksh = (int)a.K.sh;
  const int rtk_g = rowtok(rowg), rtk_k = rowtok(rowk), rtk_q = rowtok(16 * w + t16), rtk_l = rowtok(lane);
  const uint32_t uoff0 = (uint32_t)(rtk_g * usl + cg8), xoff0 = (uint32_t)(rtk_g * xsl + cg8);
  const uint32_t koff0 = (uint32_t)(rtk_k * ksl + ck8), qoff0 = (uint32_t)(rtk_q * qsl + 8 * g16);
  const int d32 = rev ? -32 : 32;
  u32x4 ruu[2], rx4[2], rk[2], qf[2];
  float rdt = 0.f, rda = 0.f;
  int stlo = 0;
  // branch-free loads (see ssd_mfma_a3_kernel): rows past the end of a ragged chunk read the chunk's first row
  auto prefetch_tiles = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Uc = Ub + (int64_t)stlo * usl;
    const uint16_t* Xc = Xb + (int64_t)stlo * xsl;
    const uint16_t* Kc = Kb + (int64_t)stlo * ksl;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const bool ok = rtk_g + d32 * r < lim;
      ruu[r] = ld16(Uc + (ok ? uoff0 + (uint32_t)(r * d32 * usl) : (uint32_t)cg8));
      rx4[r] = ld16(Xc + (ok ? xoff0 + (uint32_t)(r * d32 * xsl) : (uint32_t)cg8));
    }
    const uint32_t ko = rtk_k < lim ? koff0 : (uint32_t)ck8;
#pragma unroll
    for (int r = 0; r < 2; r++) rk[r] = ld16(Kc + r * ksh + ko);
  };
  auto prefetch_q = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Qc = Qb + (int64_t)stlo * qsl;
    const uint32_t qo = rtk_q < lim ? qoff0 : (uint32_t)(8 * g16);
#pragma unroll
    for (int kk = 0; kk < 2; kk++) qf[kk] = ld16(Qc + 32 * kk + qo);
    const int t = stlo + rtk_l, ta = rev ? t + 1 : t;
    rdt = dtrow[t < a.L ? t : 0];
    rda = dtrow[ta < a.L ? ta : 0];
  };
  const int o_cg = kx3(rowg, cg8), o_ck = ux3(rowk, ck8);
  auto commit = [&]() {
    const u32x4 zero4 = {0, 0, 0, 0};
    const int lim = a.L - stlo;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const bool ok = rtk_g + d32 * r < lim;
      st16(&sm.U[o_cg + 32 * 128 * r], ok ? ruu[r] : zero4);
      st16(&sm.X4[o_cg + 32 * 128 * r], ok ? rx4[r] : zero4);
    }
    const bool okk = rtk_k < lim;
#pragma unroll
    for (int r = 0; r < 2; r++) st16(&sm.K[r][o_ck], okk ? rk[r] : zero4);
  };
  const float Ah = a.A[hcur];
  const float Ah2 = Ah * LOG2E;
  auto scalars = [&]() {   // waves with w == 0; lanes = rows of the staged chunk of head hh
    const int t = stlo + rtk_l;
    const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
    const float wv = okd ? (a.w_is_dt ? rdt : 1.f) : 0.f;
    const float cs = wave_incl_scan_add((oka ? rda : 0.f) * Ah2);
    const float cs_end = wave_read_lane(cs, 63);
    sm.cs[hh][lane] = cs;
    sm.lw[hh][lane] = log2_fast(wv) - cs;
    sm.ecs[hh][lane] = exp2_fast(cs);
    sm.ws[hh][lane] = wv * exp2_fast(cs_end - cs);
    sm.dtl[hh][lane] = okd ? rdt : 0.f;
  };

  // ---- lane-constant LDS element offsets
  // (a 16-column block index ut only touches segment bits 1-3, which the swizzle XORs: offset(ut) = offset(0) ^ (ut << 4))
  int o_rdk[2];
#pragma unroll
  for (int i = 0; i < 2; i++) o_rdk[i] = ux3(t16, 32 * i + 8 * g16);        // b128 row reads of K (G) and S (Q.S): row t16
  int o_mu = kx3(4 * g16 + (t16 >> 2), 4 * (t16 & 3));                // permuted transpose reads of U for M.U
  int o_su = kx3(8 * g16 + (t16 >> 2), 4 * (t16 & 3));                // transpose reads of U for the state update
  int o_x4 = kx3(t16, 4 * g16);                                       // X4 of the lane's output row
  int o_ps = ux3(t16, 4 * g16);                                       // publish: rows 16 ut + t16
  int o_tk = ux3(8 * g16 + (t16 >> 2), 4 * (t16 & 3));                // transpose reads of K^T (columns 16 w + ..)

  // ---- running state S^T[p = 16 w + 4 g16 + r][n = 16 ut + t16]
  f32x4 accS[8];
#pragma unroll
  for (int ut = 0; ut < 8; ut++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float v = 0.f;
      const int k = 16 * w + 4 * g16 + r, u = 16 * ut + t16;   // this kernel's state is [n = u][p = k]
      if (seg > 0) {
        // start state of the segment, folded in the class A accumulator order of the transposed state [p][n]:
        // element ((2 (n >> 5) + (p >> 5)) * 16 + 4 ((n >> 3) & 3) + (n & 3)) * 64 + 32 ((n >> 2) & 1) + (p & 31)
        const int n = u, pp = k;
        const int e = a.seg_fmt ? pp * 128 + n
                                : ((2 * (n >> 5) + (pp >> 5)) * 16 + 4 * ((n >> 3) & 3) + (n & 3)) * 64 + 32 * ((n >> 2) & 1) + (pp & 31);
        v = a.seg[(((int64_t)b * a.H + hcur) * a.nseg + seg - 1) * SEG_STATE + e];
      } else if (a.init) {
        v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)hcur * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
      }
      accS[ut][r] = v;
    }
  auto publish_state = [&]() {   // sm.S[hh][n][p]: 4 consecutive p per lane
#pragma unroll
    for (int ut = 0; ut < 8; ut++) {
      u32x2 v = {pack_bf16x2(accS[ut][0], accS[ut][1]), pack_bf16x2(accS[ut][2], accS[ut][3])};
      *reinterpret_cast<u32x2*>(&sm.S[hh][(o_ps ^ (w << 4)) + 16 * 64 * ut]) = v;   // column block 16 w = segment bits 1-2
    }
  };

  stlo = chunk_lo(c0);
  prefetch_tiles();
  prefetch_q();
  commit();
  if (w == 0) scalars();
  publish_state();
  block_sync();
  // head-pair partial tiles, bf16 (round 2: the fp32 form was 537 MB written and 537 MB read back per scan at B 8, L 4096)
  uint16_t* part = (uint16_t*)a.part + ((int64_t)b * pairs + hp) * (int64_t)a.L * 128;
  // state checkpoints in fragment order: [b][pair][chunk][wave 8][q 4][lane 64][4] packed bf16 pairs (u32), entry
  // (q, i) = tile ut = 2 q + (i >> 1), register pair i & 1: four fully coalesced 16-byte accesses per lane (sixteen
  // 4-byte loads per lane cost the dB scan 108 us)
  uint32_t* ck = a.ckpt ? (uint32_t*)a.ckpt + ((int64_t)b * pairs + hp) * (int64_t)nC * 8192 : nullptr;
  float* tokscal = a.tokscal + ((int64_t)b * a.H + hcur) * a.L;
  float dDp[DMODE == 2 ? 2 : 1][DMODE == 2 ? 8 : 1];
#pragma unroll
  for (int i = 0; i < (DMODE == 2 ? 2 : 1); i++)
#pragma unroll
    for (int e = 0; e < (DMODE == 2 ? 8 : 1); e++) dDp[i][e] = 0.f;
  const bool want_bnd = MODE == GS_DB && ck != nullptr && a.bnd != nullptr;
  // checkpoint of the chunk the NEXT iteration closes, fetched while the output registers are dead
  u32x4 ckv[4];
  auto load_ckpt = [&](int c) {
    const uint32_t* cp = ck + (int64_t)(nC - 1 - c) * 8192 + wave * 1024 + lane * 4;
#pragma unroll
    for (int i = 0; i < 4; i++) ckv[i] = ld16(cp + 256 * i);
  };
  // boundary j (between chunks j - 1 and j in token order, j = 1 .. nC) carries a checkpoint / restart value when
  // j % ckpt_every == 0 or j == nC
  const int cke = a.ckpt_every > 0 ? a.ckpt_every : 1;
  auto is_restart = [&](int j) -> bool { return (j % cke) == 0 || j == nC; };
  if (want_bnd && is_restart(nC - c0)) load_ckpt(c0);
#ifdef OMK_PHASE_PROF   // developer build: skip phases (bit i of OMK_ABLATE_B), results are wrong
  const int ablb = a.ablate;
#else
  constexpr int ablb = 0;
#endif

  for (int c = c0; c < c1; c++) {
    const int tlo = chunk_lo(c);
    const int cnext = c + 1 < c1 ? c + 1 : c;
    OMK_OPAQUE(o_mu); OMK_OPAQUE(o_su); OMK_OPAQUE(o_x4); OMK_OPAQUE(o_ps); OMK_OPAQUE(o_tk);
    const bool bnd_here = want_bnd && is_restart(nC - c);   // the boundary behind chunk id = nC - 1 - c is j = nC - c
    if (bnd_here && !(ablb & 8)) {
      // exact restart value of the decay-gradient prefix at the boundary behind chunk id = nC - 1 - c:
      //   dl(first token of chunk id + 1) = exp(a_first(id+1)) * < g_first(id+1) (= accS now), h_last(id) (= dC-scan
      //   checkpoint of chunk id) >;  bnd[id + 1] holds the inner product
      // (g enters as the bf16 pairs the Q . S product of this chunk reads anyway: one v_dot2c_f32_bf16 per pair instead of two
      // unpacks and two FMAs; two running sums break the dependency chain)
      float dot = 0.f, dot1 = 0.f;
#pragma unroll
      for (int ut = 0; ut < 8; ut++) {
        dot = dot2_bf16(pack_bf16x2(accS[ut][0], accS[ut][1]), ckv[ut >> 1][2 * (ut & 1)], dot);
        dot1 = dot2_bf16(pack_bf16x2(accS[ut][2], accS[ut][3]), ckv[ut >> 1][2 * (ut & 1) + 1], dot1);
      }
      dot += dot1;
      dot = wave_sum(dot);
      if (lane == 0) sm.bred[wave] = dot;
    }
    // ---- (1) O^T = exp2(cs_l) * (S_in^T . Q^T)
    f32x4 acc[8];
#pragma unroll
    for (int ut = 0; ut < 8; ut++) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(ablb & 1))
#pragma unroll
    for (int kk = 0; kk < 2; kk++)
#pragma unroll
      for (int ut = 0; ut < 8; ut++) {
        const s16x8 fs = as_s16x8(ld16(&sm.S[hh][o_rdk[kk] + 16 * 64 * ut]));
        acc[ut] = mfma16x16x32_bf16(fs, as_s16x8(qf[kk]), acc[ut]);
      }
    {
      const float e1 = sm.ecs[hh][16 * w + t16];
#pragma unroll
      for (int ut = 0; ut < 8; ut++) acc[ut] *= e1;
    }
    block_sync();   // X: every wave is done with S_in (and bred is complete)
    if (bnd_here && tid < 2)   // raw < g, h >; the finish pass applies exp(a_first(id + 1)) (no dependent global load here)
      a.bnd[((int64_t)b * a.H + h0 + tid) * (nC + 1) + (nC - 1 - c) + 1] = sm.bred[4 * tid] + sm.bred[4 * tid + 1] + sm.bred[4 * tid + 2] + sm.bred[4 * tid + 3];
    stlo = chunk_lo(cnext);
    prefetch_tiles();
    // ---- (2) intra-chunk: G tiles -> M fragments (registers) -> U^T . M^T
    {
      const float cs_l = sm.cs[hh][16 * w + t16];
      // both class-B scans emit the token scalars whose reverse prefix is the decay gradient: with M in bf16 alone d(dt) leaves its
      // 6e-3 bound (8e-3 .. 1e-2 measured), so the split stays here
      constexpr bool HILO = true;
      auto block = [&](int kk, bool second, bool diag0, bool diag1) {
        u32x4 mh, ml = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (j == 1 && !second) { mh[2] = mh[3] = ml[2] = ml[3] = 0u; continue; }
          const int ta = 2 * kk + j;
          f32x4 gt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kq = 0; kq < 2; kq++) {
            const u32x4 fa = ld16(&sm.K[hh][o_rdk[kq] + 16 * 64 * ta]);
            gt = mfma16x16x32_bf16(as_s16x8(fa), as_s16x8(qf[kq]), gt);
            if (MODE == GS_DB && DMODE != 0 && (j == 0 ? diag0 : diag1)) {   // the diagonal tile holds K rows 16 w + t16: dD += dy . x
#pragma unroll
              for (int e2 = 0; e2 < 4; e2++) {
                if (DMODE == 1) dDp[0][0] = dot2_bf16(fa[e2], qf[kq][e2], dDp[0][0]);   // one D per head: both products in one v_dot2c
                else {
                  dDp[kq][2 * e2] += bf_lo(fa[e2]) * bf_lo(qf[kq][e2]);
                  dDp[kq][2 * e2 + 1] += bf_hi(fa[e2]) * bf_hi(qf[kq][e2]);
                }
              }
            }
          }
          const f32x4 lw4 = *reinterpret_cast<const f32x4*>(&sm.lw[hh][16 * ta + 4 * g16]);
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            v[r] = gt[r] * exp2_fast(cs_l + lw4[r]);
            if (j == 0 ? diag0 : diag1) v[r] = (4 * g16 + r <= t16) ? v[r] : 0.f;
          }
#pragma unroll
          for (int p2 = 0; p2 < 2; p2++) {
            const uint32_t hi = pack_bf16x2(v[2 * p2], v[2 * p2 + 1]);
            mh[2 * j + p2] = hi;
            if (HILO) ml[2 * j + p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi), v[2 * p2 + 1] - bf_hi(hi));
          }
        }
#pragma unroll
        for (int ut = 0; ut < 8; ut++) {
          const uint16_t* pu = &sm.U[(o_mu ^ (ut << 4)) + 32 * 128 * kk];
          const s16x4 u0 = lds_read_tr16_b64(pu);
          const s16x4 u1 = lds_read_tr16_b64(pu + 16 * 128);
          s16x8 fu;
          fu[0] = u0[0]; fu[1] = u0[1]; fu[2] = u0[2]; fu[3] = u0[3]; fu[4] = u1[0]; fu[5] = u1[1]; fu[6] = u1[2]; fu[7] = u1[3];
          acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(mh), acc[ut]);
          if (HILO) acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(ml), acc[ut]);
        }
      };
      if (ablb & 2) { }
      else if (w == 0) { block(0, false, true, false); }
      else if (w == 1) { block(0, true, false, true); }
      else if (w == 2) { block(0, true, false, false); block(1, false, true, false); }
      else { block(0, true, false, false); block(1, true, false, true); }
    }
    prefetch_q();
    // ---- (3) state update: S^T[p][n] = exp2(cs_end) S^T + sum_l (ws_l K[l][p]) U[l][n]
    if (!(ablb & 4)) {
      const float dec = sm.ecs[hh][QC - 1];
#pragma unroll
      for (int ut = 0; ut < 8; ut++) accS[ut] *= dec;
#pragma unroll
      for (int lb = 0; lb < 2; lb++) {
        const s16x4 k0 = lds_read_tr16_b64(&sm.K[hh][(o_tk ^ (w << 4)) + 32 * 64 * lb]);
        // rows + 4 flip row bit 2: the swizzle changes by swzK(4) = 3 segments
        const s16x4 k1 = lds_read_tr16_b64(&sm.K[hh][(o_tk ^ (w << 4) ^ (3 << 3)) + 32 * 64 * lb + 4 * 64]);
        const f32x4 s0v = *reinterpret_cast<const f32x4*>(&sm.ws[hh][32 * lb + 8 * g16]);
        const f32x4 s1v = *reinterpret_cast<const f32x4*>(&sm.ws[hh][32 * lb + 8 * g16 + 4]);
        u32x4 kp;
        kp[0] = pack_bf16x2(bf16_to_f32((uint16_t)k0[0]) * s0v[0], bf16_to_f32((uint16_t)k0[1]) * s0v[1]);
        kp[1] = pack_bf16x2(bf16_to_f32((uint16_t)k0[2]) * s0v[2], bf16_to_f32((uint16_t)k0[3]) * s0v[3]);
        kp[2] = pack_bf16x2(bf16_to_f32((uint16_t)k1[0]) * s1v[0], bf16_to_f32((uint16_t)k1[1]) * s1v[1]);
        kp[3] = pack_bf16x2(bf16_to_f32((uint16_t)k1[2]) * s1v[2], bf16_to_f32((uint16_t)k1[3]) * s1v[3]);
#pragma unroll
        for (int ut = 0; ut < 8; ut++) {
          const s16x4 u0 = lds_read_tr16_b64(&sm.U[(o_su ^ (ut << 4)) + 32 * 128 * lb]);
          const s16x4 u1 = lds_read_tr16_b64(&sm.U[(o_su ^ (ut << 4) ^ (3 << 3)) + 32 * 128 * lb + 4 * 128]);
          s16x8 fu;
          fu[0] = u0[0]; fu[1] = u0[1]; fu[2] = u0[2]; fu[3] = u0[3]; fu[4] = u1[0]; fu[5] = u1[1]; fu[6] = u1[2]; fu[7] = u1[3];
          accS[ut] = mfma16x16x32_bf16(as_s16x8(kp), fu, accS[ut]);
        }
      }
    }
    if (MODE == GS_DC && ck && is_restart(c + 1) && !(ablb & 8)) {   // forward state at the END of this chunk, fragment order, bf16 pairs
      uint32_t* cp = ck + (int64_t)c * 8192 + wave * 1024 + lane * 4;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int ut = 2 * q + (i >> 1), j = i & 1;
          v[i] = pack_bf16x2(accS[ut][2 * j], accS[ut][2 * j + 1]);
        }
        st16(cp + 256 * q, v);
      }
    }
    // ---- (4) token scalar of this head: X4_l . O_l (row l = 16 w + t16: 32 products per lane, 4 lanes per row)
    const int trow = tlo + rtk_q;
    if (!(ablb & 16)) {
      float pv = 0.f;
#pragma unroll
      for (int ut = 0; ut < 8; ut++) {
        const u32x2 xr = *reinterpret_cast<const u32x2*>(&sm.X4[(o_x4 ^ (ut << 4)) + 16 * 128 * w]);
        pv += bf_lo(xr[0]) * acc[ut][0] + bf_hi(xr[0]) * acc[ut][1] + bf_lo(xr[1]) * acc[ut][2] + bf_hi(xr[1]) * acc[ut][3];
      }
      pv += shfl_xor(pv, 16);
      pv += shfl_xor(pv, 32);
      if (g16 == 0 && trow < a.L) tokscal[trow] = pv;
    }
    // ---- (5) dB: scale by dt'_l; head 1 hands its tiles over
    if (MODE == GS_DB) {
      const float dts = sm.dtl[hh][16 * w + t16];
#pragma unroll
      for (int ut = 0; ut < 8; ut++) acc[ut] *= dts;
    }
    // the two heads of the pair swap HALF of their tiles (head 0 hands over columns 64 .. 127, head 1 columns 0 .. 63) and each adds
    // and stores the half it kept: the same LDS bytes as handing all eight tiles to head 0, but the adds, conversions and stores
    // are spread over all eight waves instead of four
    {
      float* xo = &sm.O[((hh * 4 + w) * 4 * 64 + lane) * 4];
      if (hh == 0) {   // (wave-uniform branches: no run-time register indexing)
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4*>(xo + q * 256) = acc[4 + q];
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) *reinterpret_cast<f32x4*>(xo + q * 256) = acc[q];
      }
    }
    block_sync();   // E: exchange buffer complete; nobody reads this chunk's tiles / scalars any more
    if (trow < a.L && !(ablb & 32)) {
      const float* xi = &sm.O[(((hh ^ 1) * 4 + w) * 4 * 64 + lane) * 4];
      uint16_t* prow = part + (int64_t)trow * 128 + 4 * g16;
      if (hh == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const f32x4 sv = acc[q] + *reinterpret_cast<const f32x4*>(xi + q * 256);
          const u32x2 pv = {pack_bf16x2(sv[0], sv[1]), pack_bf16x2(sv[2], sv[3])};
          *reinterpret_cast<u32x2*>(prow + 16 * q) = pv;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const f32x4 sv = acc[4 + q] + *reinterpret_cast<const f32x4*>(xi + q * 256);
          const u32x2 pv = {pack_bf16x2(sv[0], sv[1]), pack_bf16x2(sv[2], sv[3])};
          *reinterpret_cast<u32x2*>(prow + 16 * (4 + q)) = pv;
        }
      }
    }
    if (want_bnd && is_restart(nC - cnext) && !(ablb & 8)) load_ckpt(cnext);
    publish_state();
    commit();
    if (w == 0) scalars();
    block_sync();   // Y
  }
  if (MODE == GS_DB && DMODE == 1) {
    float v = wave_sum(dDp[0][0]);
    if (lane == 0) sm.bred[wave] = v;
    block_sync();
    if (tid < 2) atomic_add_f32(a.dD + (int64_t)(h0 + tid) * a.dDsh, sm.bred[4 * tid] + sm.bred[4 * tid + 1] + sm.bred[4 * tid + 2] + sm.bred[4 * tid + 3]);
  }
  if (MODE == GS_DB && DMODE == 2) {
    // dDp[kq][e]: column 32 kq + 8 g16 + e summed over this lane's rows; fold the 16 row lanes, then the 4 strips
#pragma unroll
    for (int kq = 0; kq < 2; kq++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float v = dDp[kq][e];
        v += shfl_xor(v, 1); v += shfl_xor(v, 2); v += shfl_xor(v, 4); v += shfl_xor(v, 8);
        if (t16 == 0) sm.dred[hh][w][32 * kq + 8 * g16 + e] = v;
      }
    block_sync();
    if (tid < 128) {
      const int r = tid >> 6, col = tid & 63;
      float v = sm.dred[r][0][col] + sm.dred[r][1][col] + sm.dred[r][2][col] + sm.dred[r][3][col];
      atomic_add_f32(a.dD + (int64_t)(h0 + r) * a.dDsh + (int64_t)col * a.dDsp, v);
    }
  }
  if (a.fin && seg == a.nseg - 1) {
    const float extra = a.fin_extra_decay ? expf(dtrow[0] * Ah) : 1.f;
#pragma unroll
    for (int ut = 0; ut < 8; ut++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k = 16 * w + 4 * g16 + r, u = 16 * ut + t16;
        a.fin[(int64_t)b * a.fsb + (int64_t)hcur * a.fsh + (int64_t)u * a.fsu + (int64_t)k * a.fsk] = accS[ut][r] * extra;
      }
  }
}

// out[b][t][g][n] = sum over the head pairs of group g of part[b][pair][t][n]   (bf16 partial tiles, fp32 sum)
__global__ void ssd_reduce_partials_kernel(const uint16_t* part, void* out, int64_t osb, int64_t osl, int64_t osg, int out_dt,
                                           int B, int L, int G, int pairs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread = 8 consecutive n
  const int64_t total = (int64_t)B * L * G * 16;
  if (i >= total) return;
  const int n8 = (int)(i % 16) * 8, g = (int)((i / 16) % G), t = (int)((i / (16 * (int64_t)G)) % L), b = (int)(i / (16 * (int64_t)G * L));
  const int ppg = pairs / G;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < ppg; p++) {
    const u32x4 v = ld16(part + (((int64_t)b * pairs + g * ppg + p) * L + t) * 128 + n8);
#pragma unroll
    for (int e = 0; e < 4; e++) { acc[2 * e] += bf_lo(v[e]); acc[2 * e + 1] += bf_hi(v[e]); }
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

int ssd_reduce_partials(const float* part, void* out, int64_t osb, int64_t osl, int64_t osg, int out_dt, int B, int L, int G, int H, omk_stream stream) {
  const int64_t total = (int64_t)B * L * G * 16;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  OMK_LAUNCH(ssd_reduce_partials_kernel, grid, block, 0, stream, (const uint16_t*)part, out, osb, osl, osg, out_dt, B, L, G, H / 2);
  return OMK_OK;
}

static bool src_ok16(const Src& s, bool need) {
  if (!s.p) return !need;
  return s.dt == OMK_BF16 && ((uintptr_t)s.p & 15) == 0 && s.sb % 8 == 0 && s.sl % 8 == 0 && s.sh % 8 == 0;
}

// 32-bit per-lane offsets: row strides below 2^24 elements keep every 64-row tile offset inside 31 bits
static bool stride_ok(int64_t s) { return s >= 0 && s < ((int64_t)1 << 24); }

static int ssd_mfma_launch_b(const GScan& g, omk_stream stream, int dry) {
  if (g.DU != 128 || g.DK != 64 || g.H % 2 != 0 || (g.H / g.G) % 2 != 0 || (!g.part && !dry)) return OMK_EUNSUPPORTED;
  if (!src_ok16(g.U, true) || !src_ok16(g.K, true) || !src_ok16(g.Q, true) || !src_ok16(g.X4, true)) return OMK_EUNSUPPORTED;
  if (!stride_ok(g.U.sl) || !stride_ok(g.K.sl) || !stride_ok(g.Q.sl) || !stride_ok(g.X4.sl) || !stride_ok(g.K.sh)) return OMK_EUNSUPPORTED;
  if (!g.tokscal && !dry) return OMK_EUNSUPPORTED;
  if (dry) return OMK_OK;
  // one head pair (x one segment of a split sequence, start states prepared by ssd_mfma_prepare_segments) per workgroup
  GScan a = g;
  const SegPlan sp = (a.seg && a.seg_ready) ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QC - 1) / QC};
  a.nseg = sp.nseg; a.cps = sp.cps;
  dim3 grid((unsigned)(a.B * (a.H / 2) * a.nseg)), block(512);
  const size_t smem = sizeof(SmemB3);
#define OMK_B3(MODE_, DM_) do { \
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_b3_kernel<MODE_, DM_>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_mfma_b3_kernel<MODE_, DM_>), grid, block, smem, stream, a); } while (0)
  if (a.mode == GS_DC) OMK_B3(GS_DC, 0);
  else if (!a.dD) OMK_B3(GS_DB, 0);
  else if (a.dDsp == 0) OMK_B3(GS_DB, 1);
  else OMK_B3(GS_DB, 2);
#undef OMK_B3
  return OMK_OK;
}

int ssd_mfma_prepare_segments(const GScan& g, omk_stream stream, int* seg_fmt) {
  if (seg_fmt) *seg_fmt = 0;
  GScan a = g;
  a.dump = nullptr; a.state_only = 0;   // the zero-start pass is not the one that dumps window states / leaves the final state
  const SegPlan sp = ssd_segments(a.B * a.H, a.L);
  a.nseg = sp.nseg; a.cps = sp.cps;
  if (!a.seg || a.nseg < 2) return OMK_OK;
  dim3 block(256), sgrid((unsigned)(a.B * a.H * (a.nseg - 1)));
  const size_t smem = sizeof(SmemA3);
  // the state pass does not depend on the mode (no output); it reads U, K, dt' and the scan direction only.  When the caller keeps the
  // final state (prefill -> decode hand-off, context-parallel shards) the segment states carry the hi + lo operand like the scan proper
  // does then: a kept final state of a SPLIT sequence (B = 1 prefill) is exact to fp32 accumulation too, not 1e-3 off.
  const bool khilo = g.mode == GS_Y && ((g.flags & (GSF_KHILO | GSF_PRECISE)) || g.fin != nullptr);
  if (khilo) {
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, false, true, false, true>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
    OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, false, true, false, true>), sgrid, block, smem, stream, a);
  } else {
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, false, true, false>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
    OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, false, true, false>), sgrid, block, smem, stream, a);
  }
  kernels_note("ssd_mfma_a3<segment state pass,khilo=%d>;ssd_seg_fold", (int)khilo);
  dim3 fgrid((unsigned)((int64_t)a.B * a.H * (SEG_STATE / 256)));
  OMK_LAUNCH(ssd_seg_fold_kernel, fgrid, block, 0, stream, a);
  return OMK_OK;
}

int ssd_mfma_state_dump(const GScan& g, omk_stream stream) {
  if (!g.dump || g.DU != 64 || g.DK != 128) return OMK_EUNSUPPORTED;
  {   // the scan proper dumps with the same kernel: identical images
    const int rc = ssd_a6_state_dump(g, stream);
    if (rc != OMK_EUNSUPPORTED) return rc;
  }
  GScan a = g;
  const SegPlan sp = (a.seg && a.seg_ready) ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QC - 1) / QC};
  a.nseg = sp.nseg; a.cps = sp.cps;
  dim3 grid((unsigned)(a.B * a.H * a.nseg)), block(256);
  const size_t smem = sizeof(SmemA3);
  if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, false, true, false>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
  OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, false, true, false>), grid, block, smem, stream, a);
  return OMK_OK;
}

int ssd_mfma_state_only(const GScan& g, omk_stream stream) {
  if (g.mode != GS_Y || g.DU != 64 || g.DK != 128 || !g.fin || (g.H / g.G) < 1) return OMK_EUNSUPPORTED;
  if (!src_ok16(g.U, true) || !src_ok16(g.K, true) || !stride_ok(g.K.sl) || !stride_ok(g.U.sl)) return OMK_EUNSUPPORTED;
  {
    const int64_t ms = g.K.sl > g.U.sl ? g.K.sl : g.U.sl;
    if ((int64_t)g.L * ms * 2 >= (int64_t)0xfffff000) return OMK_EUNSUPPORTED;
  }
  {   // the column-slice kernel's state pass: the state a scan with final states carries
    GScan q = g; q.Q = q.K;
    const int rc = ssd_a6_state_only(q, stream);
    if (rc != OMK_EUNSUPPORTED) return rc;
  }
  GScan a = g;
  a.Q = a.K;   // never read by the state pass; keeps the buffer descriptors well formed
  a.state_only = 1; a.out = nullptr; a.outx = nullptr; a.dump = nullptr;
  const SegPlan sp = a.seg ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QC - 1) / QC};
  if (sp.nseg > 1) {
    int rc = ssd_mfma_prepare_segments(a, stream);
    if (rc) return rc;
    a.seg_ready = 1;
  }
  a.nseg = sp.nseg; a.cps = sp.cps;
  dim3 grid((unsigned)(a.B * a.H * a.nseg)), block(256);
  const size_t smem = sizeof(SmemA3);
  if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, false, true, false>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
  OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, false, true, false>), grid, block, smem, stream, a);
  return OMK_OK;
}

int ssd_mfma_launch(const GScan& g, omk_stream stream, int dry) {
  if (g.mode == GS_DC || g.mode == GS_DB) return ssd_mfma_launch_b(g, stream, dry);
  if (g.mode != GS_Y && g.mode != GS_DX) return OMK_EUNSUPPORTED;
  if (g.DU != 64 || g.DK != 128 || (g.H / g.G) < 1) return OMK_EUNSUPPORTED;
  if (!src_ok16(g.U, true) || !src_ok16(g.K, true) || !src_ok16(g.Q, true) || !src_ok16(g.Z, false)) return OMK_EUNSUPPORTED;
  if (g.out_dt != OMK_BF16 || ((uintptr_t)g.out & 15) || g.osb % 8 || g.osl % 8 || g.osh % 8) return OMK_EUNSUPPORTED;
  if (g.outx && ((uintptr_t)g.outx & 15)) return OMK_EUNSUPPORTED;
  if (g.mode == GS_DX && g.dD) return OMK_EUNSUPPORTED;   // dD comes from the dB scan
  if (!stride_ok(g.K.sl) || !stride_ok(g.Q.sl) || !stride_ok(g.U.sl) || !stride_ok(g.osl) || (g.Z.p && !stride_ok(g.Z.sl))) return OMK_EUNSUPPORTED;
  {   // the class A kernel addresses one (batch, head) slice through 32-bit buffer offsets
    int64_t ms = g.K.sl > g.Q.sl ? g.K.sl : g.Q.sl;
    ms = ms > g.U.sl ? ms : g.U.sl; ms = ms > g.osl ? ms : g.osl;
    if ((int64_t)g.L * ms * 2 >= (int64_t)0xfffff000) return OMK_EUNSUPPORTED;
  }
  if (dry) return OMK_OK;
  // PRECISE exists as an instantiation of the specialised-wave kernel; a PRECISE call on another shape (gate / pre-gate copy in the epilogue, D per
  // (head, column), heads that do not pair up) takes the caller's fall-back chain: the fp32 VALU scan, which rounds nothing
  if ((g.flags & GSF_PRECISE) && g.mode == GS_Y && !ssd_a8_applies(g)) return OMK_EUNSUPPORTED;
  if (ssd_a6_applies(g)) return ssd_a6_launch(g, stream);
  // one head (x one segment of the sequence) per workgroup, two workgroups per CU
  GScan a = g;
  const SegPlan sp = a.seg ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QC - 1) / QC};
  a.nseg = sp.nseg; a.cps = sp.cps;
  dim3 grid((unsigned)(a.B * a.H * a.nseg)), block(256);
  const size_t smem = sizeof(SmemA3);
#define OMK_A3K(MODE_, EX_, ST_, DF_, KH_, GRID_) do { \
    kernels_note("ssd_mfma_a3<mode=%d,ex=%d,state=%d,dfold=%d,khilo=%d>", (int)MODE_, (int)EX_, (int)ST_, (int)DF_, (int)KH_); \
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<MODE_, EX_, ST_, DF_, KH_>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_mfma_a3_kernel<MODE_, EX_, ST_, DF_, KH_>), GRID_, block, smem, stream, a); } while (0)
  const bool khilo = (g.flags & (GSF_KHILO | GSF_PRECISE)) || OMK_SSD_KHILO_DEFAULT != 0;
#define OMK_A3(MODE_, EX_, ST_, DF_, GRID_) do { \
    if (MODE_ == GS_Y && khilo) OMK_A3K(MODE_, EX_, ST_, DF_, (MODE_ == GS_Y), GRID_); else OMK_A3K(MODE_, EX_, ST_, DF_, false, GRID_); } while (0)
  if (a.nseg > 1 && !a.seg_ready) {
    int rc = ssd_mfma_prepare_segments(g, stream);
    if (rc) return rc;
  }
  const bool dfold = !a.D || a.Dsp == 0;   // one D per head (or none)
  if (a.mode == GS_Y && (a.Z.p || a.outx)) { if (dfold) OMK_A3(GS_Y, true, false, true, grid); else OMK_A3(GS_Y, true, false, false, grid); }
  else if (a.mode == GS_Y && a.dump) {   // (ssd.hip only asks for dumps without gate / pre-gate copy / KHILO)
    if (khilo) return OMK_EUNSUPPORTED;
    if (dfold) {
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, false, false, true, false, true>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
      OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, false, false, true, false, true>), grid, block, smem, stream, a);
    } else {
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, false, false, false, false, true>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
      OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, false, false, false, false, true>), grid, block, smem, stream, a);
    }
  }
  else if (a.mode == GS_Y) { if (dfold) OMK_A3(GS_Y, false, false, true, grid); else OMK_A3(GS_Y, false, false, false, grid); }
  else OMK_A3(GS_DX, false, false, false, grid);
#undef OMK_A3
#undef OMK_A3K
  return OMK_OK;
}

}  // namespace omk
