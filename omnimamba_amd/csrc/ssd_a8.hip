// ssd_a8.hip -- class A scan (forward y and the dx scan of the backward) with SPECIALISED waves: four compute waves (one per SIMD, 32 state
// columns each) and four helper waves (one per SIMD beside them) per workgroup of a head pair.
//
// Where it comes from.  ssd_a6.hip runs eight identical waves of 16 state columns: every wave reads all of Q and all of K^T of a sub-chunk
// from LDS for its 16 columns (LDS busy 53 % of a chunk) and the two instruction streams of a SIMD run in lockstep, so the pipes of a
// SIMD add up instead of overlapping.  The round-4 experiment with ONE wave of 32 columns per SIMD (ssd_a7.hip, 363 registers,
// git history) halved the fragment reads but left three serial latency chains -- tile build, commit of the staged chunk, token scalars --
// fully exposed in its single stream (218 us against 197).  Here those chains live in a SECOND wave per SIMD:
//   * compute wave (hh, w), waves 0..3: the state columns [32 w, 32 w + 32) of head hh as two groups of eight 16 x 16 accumulator tiles,
//     and nothing but the two phases of a sub-chunk (ssd_a6.hip header): pack + S_in^T Q^T, then state update + U^T M^T + output rows.
//     Its stream is written in issue order: every MFMA is followed by the few VALU / LDS instructions that fit in its shadow
//     (pack of the next tile pair, the scaled U operand, decay of the next tile, output rows), pinned by scheduling fences;
//   * helper wave, waves 4..7: global loads of chunk c + 3 (K, Q, dt') and c + 2 (U), their commit to LDS, the token scalars of chunk
//     c + 2 and the shared M tiles of chunk c + 1 (G = K Q^T once for both heads).  Its waits (vmcnt,
//     LDS round trips, the DPP scan chain) cost the compute wave nothing; the hardware interleaves the two streams.
// Both kinds take 256 registers (two waves per SIMD).  Same LDS layout, staging distances, barrier (one per chunk) and ARITHMETIC as
// ssd_a6.hip: results are equal bit for bit (tests/test_ops_ssd.py).  Variants here: one D per head (or none), no gate / pre-gate copy;
// split sequences, initial and final states, window-state images, the dx scan.  Everything else stays with ssd_a6.hip.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include "ssd_scan.h"
#include "ssd_tiles.h"

#ifndef OMK_A8_CU
#define OMK_A8_CU 1    // the compute waves stage the U tile of their own head (half each); 0: the helpers do
#endif
#ifndef OMK_A8_VAR
#define OMK_A8_VAR 0   // developer: experiment switches (bit mask), see the uses
#endif

namespace omk {

constexpr int QA8 = 64;    // tokens staged per barrier
struct SmemA8 {            // (the layout of SmemA6)
  uint16_t K[3][QA8 * 128];
  uint16_t Q[3][QA8 * 128];
  uint16_t U[2][2][QA8 * 64];
  u32x4 M[2][2][6][64];
  float rl[3][2][QA8], ws[3][2][QA8];
  float dec[3][2][2];
  float rfd[2][2][QA8], cfd[2][2][QA8], fo[2][2][QA8];
  int wide[2][2];
};
static_assert(sizeof(SmemA8) <= 160 * 1024, "one workgroup per CU");

// PRECISE (OmkSsdFwd::flags & OMK_SSD_PRECISE, forward only): the bf16 copy of the state slice that meets Q^T is a hi + lo pair -- twice the
// MFMAs of phase 1 and six instead of one VALU operation per packed pair -- and KHILO is on: no operand of the scan is rounded to 8 bits
// any more, y is within the bare 1e-3 of the fp32 recurrence on every head (tests/test_configs_gpu.py, profiles/r06_precise.txt)
// CONV (forward only; GScan::cw set -- OmkSsdFwd::conv_weight): U is the PRE-conv x and the staging side applies the causal depthwise conv1d
// (width <= 4) + SiLU of upstream's causal_conv1d_fn to it on the way into LDS -- the x columns (4096 of the 4352 conv channels of the 1.3B
// block) never make the round trip through a conv output buffer (K2 fusion, forward-only path: prefill / inference).  The helper waves
// stage U then: a lane owns FOUR consecutive tokens of eight channels, requests them with their three halo rows (seven 16-byte loads),
// runs the taps in the conv kernel's own order (bit-identical bf16 operand) and writes four 16-byte rows.
template <int MODE, bool DUMP, bool KHILO, bool PRECISE = false, bool CONV = false>
__global__ __launch_bounds__(512) void ssd_a8_kernel(GScan a) {
  constexpr bool CUS = OMK_A8_CU && !CONV;   // the compute waves stage the U tile of their own head
  OMK_DYN_SMEM(smem_raw);
  SmemA8& sm = *reinterpret_cast<SmemA8*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_i(tid >> 6);
  const bool helper = wave >= 4;
  const int hh = (wave >> 1) & 1, w = wave & 1;
  const int g16 = lane >> 4, t16 = lane & 15;
  int vid = blockIdx.x;
  if ((gridDim.x & 7) == 0) vid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);   // XCD-contiguous (batch, pair)
  const int pairs = a.H >> 1;
  const int hp = vid % pairs, seg = (vid / pairs) % a.nseg, b = vid / (pairs * a.nseg);
  const int h = 2 * hp + hh;
  const int g = (2 * hp) / (a.H / a.G);
  const int nC = (a.L + QA8 - 1) / QA8;
  const int c0 = seg * a.cps, c1 = (c0 + a.cps < nC) ? c0 + a.cps : nC;
  const bool rev = a.reverse != 0;
  auto chunk_lo = [&](int c) -> int { return (rev ? nC - 1 - c : c) * QA8; };
  auto clipc = [&](int c) -> int { return c < c1 ? c : c1 - 1; };
  auto rowtok = [&](int i) -> int { return rev ? QA8 - 1 - i : i; };
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  const int osl = (int)a.osl;
#ifdef OMK_PHASE_PROF   // developer build (tools/phase_prof_a8.py): s_memtime deltas per phase, workgroup 0
  uint64_t pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
#define PT8(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
  // (every workgroup also leaves the 100 MHz wall clock of its first and last instruction + its XCC_ID behind the per-wave slots: when do the
  // workgroups of one launch finish -- the launch lasts as long as its slowest one)
#define PT8_END() do { if (a.prof != nullptr && (a.ablate & (1 << 20)) && lane == 0 && (wave == 0 || wave == 4)) { \
      a.prof[128 + 4 * blockIdx.x + (wave >> 2) * 2] = t_wall0; \
      a.prof[128 + 4 * blockIdx.x + (wave >> 2) * 2 + 1] = __builtin_readsteadycounter() | ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 60); } \
    if (prof) { pt[10] = clock64_() - t_core0; \
    pt[11] = (__builtin_readsteadycounter() - t_ref0) | ((uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) << 40); \
    if (lane == 0) for (int i = 0; i < 12; i++) a.prof[wave * 12 + i] = pt[i]; } } while (0)
  uint64_t tprev = 0, t_core0 = 0, t_ref0 = 0;
  const uint64_t t_wall0 = __builtin_readsteadycounter();
#define PT8_START() do { if (prof) { tprev = clock64_(); t_core0 = tprev; t_ref0 = __builtin_readsteadycounter(); } } while (0)
#else
#define PT8(i) do { } while (0)
#define PT8_END() do { } while (0)
#define PT8_START() do { } while (0)
#endif

  if (helper) {
    // =====================================================================================================================
    // helper wave: staging, token scalars, M tiles (the 256 threads of waves 4..7)
    // =====================================================================================================================
    // (the role of a helper -- w: scalars + one tile, or two tiles -- is a compile-time constant of its loop: as a run-time value it cut the loop
    // into ~60 basic blocks with a branch each)
    auto helper_body = [&](auto wtag) {
    constexpr int w = decltype(wtag)::value;
    const int ht = tid & 255;
    const int rowk = ht >> 4, ck8 = (ht & 15) * 8, rowu = (ht & 127) >> 3, cu8 = (ht & 7) * 8;
    const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
    const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
    const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh;
    const int ksl = (int)a.K.sl, qsl = (int)a.Q.sl, usl = (int)a.U.sl;
    const BufRes Kr = make_buf(Kb, (uint32_t)((int64_t)a.L * ksl * 2)), Qr = make_buf(Qb, (uint32_t)((int64_t)a.L * qsl * 2));
    const BufRes Ur = make_buf(Ub, (uint32_t)((int64_t)a.L * usl * 2)), Dr = make_buf(dtrow, (uint32_t)((int64_t)a.L * 4));
    const uint32_t kvo = 2u * (uint32_t)((rev ? 15 - rowk : rowk) * ksl + ck8), qvo = 2u * (uint32_t)((rev ? 15 - rowk : rowk) * qsl + ck8);
    const uint32_t uvo = 2u * (uint32_t)((rev ? 15 - rowu : rowu) * usl + cu8);
    const uint32_t dvo = 4u * (uint32_t)rowtok(lane), dvo_a = 4u * (uint32_t)(rowtok(lane) + (rev ? 1 : 0));
    // OMK_A8_PF2 = 1 (round 6 experiment, measured neutral: profiles/r06_a8_experiments.txt): TWO chunks of K / Q / dt' loads in flight per
    // helper -- register set (c - c0) & 1 is committed in iteration c (chunk c + 2) and refilled at once (chunk c + 4), the other set holds
    // chunk c + 3.  The ablations without commits / without loads are both ~30 us faster, which reads like memory latency x bytes in
    // flight; twice the bytes in flight changed nothing (bit-identical results, 169.7 - 177.6 against 172.2 - 174.9 us), so it is not.
#ifndef OMK_A8_PF2
#define OMK_A8_PF2 0
#endif
    constexpr int NSET = OMK_A8_PF2 ? 2 : 1;
    u32x4 rk[NSET][4], rq[NSET][4], ru[NSET][4];
    float rdt_[NSET], rda_[NSET], rwv = 0.f;
    int stlo[NSET];
#pragma unroll
    for (int i = 0; i < NSET; i++) { rdt_[i] = 0.f; rda_[i] = 0.f; stlo[i] = 0; }
    // (the loads of an iteration are issued in pieces, each right behind the commit that frees its registers: four waves issuing 14 loads
    // in one burst sat ~600 cycles in the issue queue of the CU's memory pipeline)
    auto prefetch_k = [&](auto ps, int tl) {
      constexpr int S = decltype(ps)::value;
      const uint32_t sk = 2u * (uint32_t)(tl * ksl);
#pragma unroll
      for (int r = 0; r < 4; r++) rk[S][r] = buf_ld16(Kr, kvo, sk + 2u * (uint32_t)((rev ? 16 * (3 - r) : 16 * r) * ksl));
    };
    auto prefetch_q = [&](auto ps, int tl) {
      constexpr int S = decltype(ps)::value;
      const uint32_t sq = 2u * (uint32_t)(tl * qsl);
#pragma unroll
      for (int r = 0; r < 4; r++) rq[S][r] = buf_ld16(Qr, qvo, sq + 2u * (uint32_t)((rev ? 16 * (3 - r) : 16 * r) * qsl));
    };
    auto prefetch_dt = [&](auto ps, int tl) {
      constexpr int S = decltype(ps)::value;
      stlo[S] = tl;
      rdt_[S] = buf_ld_f32(Dr, dvo, 4u * (uint32_t)tl);
      rda_[S] = buf_ld_f32(Dr, dvo_a, 4u * (uint32_t)tl);
    };
    auto prefetch_kq = [&](auto ps, int tl) { prefetch_k(ps, tl); prefetch_q(ps, tl); prefetch_dt(ps, tl); };
    // ---- CONV: the lane's four tokens 4 tg .. 4 tg + 3 (+ three halo rows in front) of the channels 8 seg .. 8 seg + 7 of head hh
    const int ctg = (ht & 127) >> 3, cseg = ht & 7;
    u32x4 rc[CONV ? 7 : 1];
    float cw[CONV ? 8 : 1][4], cb[CONV ? 8 : 1];
    if (CONV) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int64_t ch = (int64_t)h * 64 + 8 * cseg + e;
        cb[e] = a.cb ? load_rt(a.cb, ch, a.cb_dt) : 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) cw[e][k] = (k >= 4 - a.cW) ? load_rt(a.cw, ch * a.cwsc + (int64_t)(k - (4 - a.cW)) * a.cwsk, a.cw_dt) : 0.f;   // (taps of a narrower filter: leading zeros)
      }
    }
    auto prefetch_u = [&](auto ps, int tl) {   // (the helpers stage U: OMK_A8_CU = 0, or CONV)
      constexpr int S = decltype(ps)::value;
      if (CONV) {
        // rows tl + 4 tg + j - 3, j = 0 .. 6.  The three rows in front of the sequence are zeros: lane offsets stay non-negative (the range
        // check of a buffer access is made on the unwrapped sum), the first chunk points those rows behind the range instead
        if (tl >= 3) {
          const uint32_t sc = 2u * (uint32_t)((tl - 3) * usl);
#pragma unroll
          for (int j = 0; j < 7; j++) rc[j] = buf_ld16(Ur, 2u * (uint32_t)((4 * ctg + j) * usl + 8 * cseg), sc);
        } else {
#pragma unroll
          for (int j = 0; j < 7; j++) {
            const int t = tl + 4 * ctg + j - 3;
            rc[j] = buf_ld16(Ur, t >= 0 ? 2u * (uint32_t)(t * usl + 8 * cseg) : 0x80000000u, 0u);
          }
        }
        return;
      }
      const uint32_t su_ = 2u * (uint32_t)(tl * usl);
#pragma unroll
      for (int r = 0; r < 4; r++) ru[S][r] = buf_ld16(Ur, uvo, su_ + 2u * (uint32_t)((rev ? 16 * (3 - r) : 16 * r) * usl));
    };
    const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
    auto commit_k = [&](auto ps, int kb) {
      constexpr int S = decltype(ps)::value;
#pragma unroll
      for (int r = 0; r < 4; r++) st16(&sm.K[kb][o_ck + 16 * 128 * r], rk[S][r]);
    };
    auto commit_q = [&](auto ps, int kb) {
      constexpr int S = decltype(ps)::value;
#pragma unroll
      for (int r = 0; r < 4; r++) st16(&sm.Q[kb][o_ck + 16 * 128 * r], rq[S][r]);
    };
    auto commit_u = [&](auto ps, int ub) {
      constexpr int S = decltype(ps)::value;
      if (CONV) {
        // out[t] = silu(bias + sum_k w[k] x[t - 3 + k]) in fp32, taps in the conv kernel's order (conv1d.hip), rounded to bf16 once: the
        // operand the unfused path reads back from the conv output buffer, bit for bit
        u32x4 o[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {      // channel pairs (one 32-bit register of every row)
          float xl[7], xh[7];
#pragma unroll
          for (int j = 0; j < 7; j++) { xl[j] = bf_lo(rc[j][e2]); xh[j] = bf_hi(rc[j][e2]); }
#pragma unroll
          for (int t = 0; t < 4; t++) {
            float al = cb[2 * e2], ah = cb[2 * e2 + 1];
#pragma unroll
            for (int k = 0; k < 4; k++) { al = fma_f32(cw[2 * e2][k], xl[t + k], al); ah = fma_f32(cw[2 * e2 + 1][k], xh[t + k], ah); }
            o[t][e2] = pack_bf16x2(silu_fast(al), silu_fast(ah));
          }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) st16(&sm.U[ub][hh][ux3(4 * ctg + t, 8 * cseg)], o[t]);
        return;
      }
#pragma unroll
      for (int r = 0; r < 4; r++) st16(&sm.U[ub][hh][o_cu + 16 * 64 * r], ru[S][r]);
    };
    // (LDS-DMA staging -- buffer_load ... lds, the swizzle on the source address -- was measured here as in ssd_a6.hip: correct and SLOWER, 207 us
    // against 183: the DMA path of a CU lands ~12 bytes per cycle, a chunk needs 48 KB; profiles/r05_a8_experiments.txt, git history)
    const float Ah2 = a.A[h] * LOG2E;
    // The carried state of a head lives in a BASIS: the compute waves hold S' = S(true, at the basis point) and everything that meets it
    // carries the missing decay -- rl = 2^(c_l - basis) on the output side, ws = w 2^(basis' - c_s) on the update side -- so the 64 accumulator
    // registers of a wave are multiplied by a decay only when the basis MOVES.  ssd_a6.hip moves it to the chunk end at every chunk
    // (and to the end of the first sub-chunk when that sub-chunk alone decays by more than 2^-60); here the basis stays where it is for as
    // long as it is less than 2^60 of decay behind (round 6: the decay multiplies were 62 of the 535 instructions of a chunk, most of
    // them packed fp32, which does not overlap with the matrix pipe), and is pulled up where somebody needs the true state: in front of a
    // window-state image, behind the last chunk (final state / next segment), and at every chunk when the caller asks for the
    // arithmetic of the column-slice kernel (GSF_FLUSH).  `base` = log2 decay coordinate of the basis relative to the start of the chunk
    // whose scalars are computed next (>= 0: in the past).
    float base = 0.f;
    auto dumps_at = [&](int cc) -> bool {
      if (!(DUMP && a.dump)) return false;
      const int cid = rev ? nC - 1 - cc : cc;
      return rev ? (cid == nC - 1 || (cid & 1)) : !(cid & 1);
    };
    auto scalars = [&](auto ps, int kb, int mb, int cc) {   // helper waves with w == 0 (ssd_a6.hip: the same scalars and factored tile decay); cc: chunk in scan order
      constexpr int S = decltype(ps)::value;
      float rdt, rda;
      {
        const int t = stlo[S] + rowtok(lane);
        const bool okd = t < a.L, oka = okd && (rev ? t + 1 : t) < a.L;
        rwv = okd ? (a.w_is_dt ? rdt_[S] : 1.f) : 0.f;
        rdt = okd ? rdt_[S] : 0.f;
        rda = oka ? rda_[S] : 0.f;
      }
      const float cs = wave_incl_scan_add(rda * Ah2);
      const float e31 = wave_read_lane(cs, 31), e63 = wave_read_lane(cs, 63);
      const float rsc = MODE == GS_DX ? rdt : 1.f;
      const bool flush_end = (a.flags & GSF_FLUSH) != 0 || cc >= c1 - 1 || dumps_at(cc + 1);
      const float b0 = base;                                                  // basis under the first sub-chunk's output rows
      const float b0p = (b0 - e31 <= 60.f) ? b0 : e31;                        // ... behind its state update
      const float b1p = (!flush_end && b0p - e63 <= 60.f) ? b0p : e63;        // ... behind the second one
      base = b1p - e63;
      sm.rl[kb][hh][lane] = exp2_fast(cs - (lane < 32 ? b0 : b0p)) * rsc;
      sm.ws[kb][hh][lane] = rwv * exp2_fast((lane < 32 ? b0p : b1p) - cs);
      // (1.f exactly = the basis stays: the compute waves test the bits and skip the multiply)
      if ((lane & 31) == 31) sm.dec[kb][hh][lane >> 5] = lane < 32 ? (b0p == b0 ? 1.f : exp2_fast(b0p - b0)) : (b1p == b0p ? 1.f : exp2_fast(b1p - b0p));
      const bool blk1 = (lane & 16) != 0;
      const float cmid = wave_row_bcast<7>(cs), cbnd = wave_pair_boundary(cs);   // cs of lane b16 + 7 / of lane blk1 ? b16 - 1 : b16 + 15
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
    int o_rd[4];
#pragma unroll
    for (int i = 0; i < 4; i++) o_rd[i] = kx3(t16, 32 * i + 8 * g16);
    // M tiles of one chunk, G shared by the two heads.  Roles: helper (hh, 0) computes the token scalars of head hh and builds tile (0, 0)
    // of sub-chunk jj = hh; helper (hh, 1) builds the tiles (1, 0) and (1, 1) of sub-chunk hh (one set of Q rows, two of K rows)
    float Dh2[2] = {0.f, 0.f};
    if (a.D) {
      Dh2[0] = load_rt(a.D, (int64_t)(2 * hp) * a.Dsh, a.D_dt);
      Dh2[1] = load_rt(a.D, (int64_t)(2 * hp + 1) * a.Dsh, a.D_dt);
    }
    // (the builder's LDS reads -- the operands of G and the decay factors of both heads -- are requested at the top of an iteration, a
    // commit and a prefetch in front of their first use)
    struct FragB { u32x4 k[2][4], q[4]; float rf[2][2]; f32x4 cf[2][2]; int wide[2]; };
    const int bjj = hh;
    const int brb = 32 * bjj + (w == 0 ? 0 : 16);   // Q rows (l) of the wave's tiles
    auto build_loads = [&](FragB& f, int kb, int mb) {
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) f.wide[h2] = sm.wide[mb][h2];   // (first: build_flags() turns them into scalars a commit later)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        f.q[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * brb]);
        f.k[0][i] = ld16(&sm.K[kb][o_rd[i] + 128 * (32 * bjj)]);
        if (w == 1) f.k[1][i] = ld16(&sm.K[kb][o_rd[i] + 128 * (32 * bjj + 16)]);
      }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (j == 1 && w == 0) break;
        const int tt = w == 0 ? 0 : 1 + j;
        const bool diag = tt != 1;
        const int cb = 32 * bjj + (tt == 2 ? 16 : 0);
#pragma unroll
        for (int h2 = 0; h2 < 2; h2++) {
          const float* rfa = diag ? sm.rfd[mb][h2] : sm.fo[mb][h2];
          const float* cfa = diag ? sm.cfd[mb][h2] : sm.fo[mb][h2];
          f.rf[j][h2] = rfa[brb + t16];
          f.cf[j][h2] = *reinterpret_cast<const f32x4*>(&cfa[cb + 4 * g16]);
        }
      }
    };
    // (the flags as wave-uniform scalars.  Left to the compiler, their LDS read sank to its use behind the first MFMA of a tile, where it
    // waited for EVERY LDS operation in flight: one exposed LDS round trip per tile)
    int wide_s[2] = {0, 0};
    auto build_flags = [&](const FragB& f) {
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) wide_s[h2] = uniform_i(f.wide[h2]);
      OMK_SCHED_FENCE();
    };
    auto build_tile = [&](const FragB& f, int mb, int j) {
      const int tt = w == 0 ? 0 : 1 + j;
      const bool diag = tt != 1;
      f32x4 gt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; i++) gt = mfma16x16x32_bf16(as_s16x8(f.k[j][i]), as_s16x8(f.q[i]), gt);
#pragma unroll
      for (int h2 = 0; h2 < 2; h2++) {
        float v[4];
        if (wide_s[h2] != 0) {
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = gt[r] * exp2_fast(f.rf[j][h2] + f.cf[j][h2][r]);
        } else {
          const f32x4 gc = gt * f.cf[j][h2] * f.rf[j][h2];
#pragma unroll
          for (int r = 0; r < 4; r++) v[r] = gc[r];
        }
        if (diag) {   // (a select, not a multiply by 0 / 1: the entries above the diagonal may have left the fp32 range)
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
    };
    FragB fb;
    // ---- prologue: chunks c0 and c0 + 1 staged, tiles of c0 built, chunks c0 + 2 (and c0 + 3) requested
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, NSET - 1>;
    prefetch_kq(S0{}, chunk_lo(c0));
    if (!CUS) prefetch_u(S0{}, chunk_lo(c0));
    commit_k(S0{}, 0); commit_q(S0{}, 0);
    if (!CUS) commit_u(S0{}, 0);
    if (w == 0) scalars(S0{}, 0, 0, c0);
    prefetch_kq(S0{}, chunk_lo(clipc(c0 + 1)));
    commit_k(S0{}, 1); commit_q(S0{}, 1);
    if (w == 0) scalars(S0{}, 1, 1, c0 + 1);
    block_sync();
    build_loads(fb, 0, 0);
    build_flags(fb);
    build_tile(fb, 0, 0);
    if (w == 1) build_tile(fb, 0, 1);
    prefetch_kq(S0{}, chunk_lo(clipc(c0 + 2)));
    if (NSET == 2) prefetch_kq(S1{}, chunk_lo(clipc(c0 + 3)));
    if (!CUS) { prefetch_u(S0{}, chunk_lo(clipc(c0 + 1))); if (NSET == 2 && !CONV) prefetch_u(S1{}, chunk_lo(clipc(c0 + 2))); }
    block_sync();
    int kb1 = 1, kb2 = 2, kb0 = 0;
    PT8_START();
#if !defined(OMK_EMU)
    if (OMK_A8_VAR & 1) __builtin_amdgcn_s_setprio(3);
#endif
    // one iteration: the tiles of chunk c + 1, the staging of chunk c + 2 (K / Q / scalars) and c + 1 (U) out of register set ps, the
    // loads of chunk c + 2 + NSET into the set just freed
    auto iter = [&](auto ps, int c) OMK_ALWAYS_INLINE_LAMBDA {
      const int ub0 = (c - c0) & 1, ub1 = ub0 ^ 1;
      constexpr bool more = true;   // (behind the last chunk the builders redo its tiles from the re-staged buffers: nobody reads them, no branch)
      // (OMK_A8_VAR ablations, wrong results: 16 no tile build, 32 no scalars, 64 no commits, 128 no loads.)  The steps are fenced: left
      // alone the compiler moved the builder's LDS reads down to their first use.
      if (more && !(OMK_A8_VAR & 16)) build_loads(fb, kb1, ub1);
      OMK_SCHED_FENCE();
      if (!(OMK_A8_VAR & 64)) commit_k(ps, kb2);
      if (!(OMK_A8_VAR & 128)) prefetch_k(ps, chunk_lo(clipc(c + 2 + NSET)));
      OMK_SCHED_FENCE();
      if (more && !(OMK_A8_VAR & 16)) build_flags(fb);
      PT8(0);
      if (more && !(OMK_A8_VAR & 16)) build_tile(fb, ub1, 0);
      PT8(1);
      OMK_SCHED_FENCE();
      if (!(OMK_A8_VAR & 64)) commit_q(ps, kb2);
      if (!(OMK_A8_VAR & 128)) prefetch_q(ps, chunk_lo(clipc(c + 2 + NSET)));
      OMK_SCHED_FENCE();
      PT8(2);
      if (w == 0) { if (!(OMK_A8_VAR & 32)) scalars(ps, kb2, ub0, c + 2); } else if (more && !(OMK_A8_VAR & 16)) build_tile(fb, ub1, 1);
      PT8(3);
      OMK_SCHED_FENCE();
      if (!(OMK_A8_VAR & 64) && !CUS) commit_u(ps, ub1);
      if (!(OMK_A8_VAR & 128)) { if (!CUS) prefetch_u(ps, chunk_lo(clipc(c + 1 + (CONV ? 1 : NSET)))); prefetch_dt(ps, chunk_lo(clipc(c + 2 + NSET))); }
      PT8(4);
      block_sync();
      PT8(5);
      { const int t_ = kb0; kb0 = kb1; kb1 = kb2; kb2 = t_; }
    };
    if (NSET == 2) {
      for (int c = c0; c < c1; c += 2) {
        iter(S0{}, c);
        if (c + 1 < c1) iter(S1{}, c + 1);
      }
    } else {
      for (int c = c0; c < c1; c++) iter(S0{}, c);
    }
    PT8_END();
    };
    if (w == 0) helper_body(std::integral_constant<int, 0>{}); else helper_body(std::integral_constant<int, 1>{});
    return;
  }

  // =======================================================================================================================
  // compute wave (hh, w): state columns 32 w + 16 cg + t16, cg = 0, 1
  // =======================================================================================================================
  int o_rd[4], o_kt[4], o_uf[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    o_rd[i] = kx3(t16, 32 * i + 8 * g16);
    o_kt[i] = kx3(4 * g16 + (t16 >> 2), 32 * i + 8 * (t16 & 3));
  }
#pragma unroll
  for (int cg = 0; cg < 2; cg++) o_uf[cg] = ux3(4 * g16 + (t16 >> 2), 32 * w + 16 * cg + 4 * (t16 & 3));

  // ---- running state: two column groups of eight 16 x 16 tiles (ssd_a6.hip header for the (tile, register) <-> k map)
  f32x4 accS[2][8];
#pragma unroll
  for (int cg = 0; cg < 2; cg++)
#pragma unroll
    for (int t = 0; t < 8; t++) accS[cg][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int64_t bh = (int64_t)b * a.H + h;
  if (seg > 0) {   // folded by ssd_seg_fold_kernel (row-strip accumulator order): slot seg - 1 = state at the start of this segment
    const float* sp = a.seg + (bh * a.nseg + seg - 1) * SEG_STATE;
#pragma unroll
    for (int cg = 0; cg < 2; cg++)
#pragma unroll
      for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r, kk = k & 31, su = 32 * w + 16 * cg + t16;
          accS[cg][t][r] = sp[((2 * (k >> 5) + (su >> 5)) * 16 + (kk & 3) + 4 * (kk >> 3)) * 64 + 32 * ((kk >> 2) & 1) + (su & 31)];
        }
  }
  if (a.init && seg == 0) {
#pragma unroll
    for (int cg = 0; cg < 2; cg++)
#pragma unroll
      for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r, su = 32 * w + 16 * cg + t16;
          accS[cg][t][r] = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)su * a.isu + (int64_t)k * a.isk, a.init_dt);
        }
  }
  // ---- the U tile of the wave's own head, rows 16 r + 8 w + (lane >> 3) (r = 0..3), segment lane & 7: loaded a chunk ahead of its commit
  const int usl = (int)a.U.sl;
  const BufRes Ur = make_buf((const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh, (uint32_t)((int64_t)a.L * usl * 2));
  const int urow = 8 * w + (lane >> 3);
  const uint32_t uvo = 2u * (uint32_t)((rev ? 15 - urow : urow) * usl + 8 * (lane & 7));
  const int o_cu = ux3(urow, 8 * (lane & 7));
#ifndef OMK_A8_UPF2
#define OMK_A8_UPF2 0   // 1: two chunks of U loads in flight per compute wave (9 spilled registers: + 3 %); the helpers' loads: OMK_A8_PF2
#endif
  constexpr int USET = OMK_A8_UPF2 ? 2 : 1;
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, USET - 1>;
  u32x4 ruu[USET][4];
  auto prefetch_u = [&](auto ps, int tl) {
    constexpr int S = decltype(ps)::value;
    const uint32_t su_ = 2u * (uint32_t)(tl * usl);
#pragma unroll
    for (int r = 0; r < 4; r++) ruu[S][r] = buf_ld16(Ur, uvo, su_ + 2u * (uint32_t)((rev ? 16 * (3 - r) : 16 * r) * usl));
  };
  auto commit_u = [&](auto ps, int ub) {
    constexpr int S = decltype(ps)::value;
    u32x4 (&ru)[4] = ruu[S];
    if (OMK_A8_VAR & 2048) {
      // (K2 fusion step (iii), priced: a depthwise conv of width 4 + SiLU on the 32 staged x values of this lane -- the arithmetic the staging
      // wave would do if the scan read the PRE-conv x; stand-in taps and halo rows, wrong values, representative instruction count)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float v2[2];
#pragma unroll
          for (int hlf = 0; hlf < 2; hlf++) {
            auto el = [&](int rr) -> float { const uint32_t q = ru[rr & 3][e]; return hlf ? bf_hi(q) : bf_lo(q); };
            const float acc = 0.1f + 0.5f * el(r) + 0.25f * el(r + 1) + 0.125f * el(r + 2) + 0.0625f * el(r + 3);
            v2[hlf] = silu_fast(acc);
          }
          o[e] = pack_bf16x2(v2[0], v2[1]);
        }
        st16(&sm.U[ub][hh][o_cu + 16 * 64 * r], o);
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < 4; r++) st16(&sm.U[ub][hh][o_cu + 16 * 64 * r], ru[r]);
  };
  if (CUS) {
    prefetch_u(U0{}, chunk_lo(c0));
    commit_u(U0{}, 0);
    prefetch_u(U0{}, chunk_lo(clipc(c0 + 1)));
    if (USET == 2) prefetch_u(U1{}, chunk_lo(clipc(c0 + 2)));
  }
  block_sync();   // (the helpers' prologue: two barriers)
  block_sync();
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
  const BufRes Or = make_buf(ob, (uint32_t)((int64_t)a.L * osl * 2));
  // output rows: the lane computes its row (strip s2 of sub-chunk jj: 32 jj + 16 s2 + t16) at the columns 32 w + 16 cg + 4 g16 + r; one
  // v_permlane16_swap per packed pair regroups them so that the lane STORES eight consecutive columns, 32 w + 8 (g16 >> 1) + 16 (g16 & 1) ..:
  // 16 rows x 64 bytes per store instruction instead of two times 16 rows x 32 bytes (- 5 % on the scan: the stores of the compute waves
  // share the CU's memory pipeline with the helpers' loads)
  uint32_t ovo[2][2];
#pragma unroll
  for (int jj = 0; jj < 2; jj++)
#pragma unroll
    for (int s2 = 0; s2 < 2; s2++) ovo[jj][s2] = 2u * (uint32_t)(rowtok(32 * jj + 16 * s2 + t16) * osl + 32 * w + 8 * (g16 >> 1) + 16 * (g16 & 1));

  struct FragR { u32x4 q0[4], q1[4]; };
  // u00 = (U strip 0 | U strip 0), u01 = (U strip 0 | U strip 1): the A operands of U^T M^T as they are read (strip 0 twice: two more LDS
  // reads per sub-chunk instead of eight register moves in a stream that is short of issue slots, not of LDS time)
  struct FragC { s16x8 u00[2], u01[2]; s16x4 kt[8][2]; float rl0, rl1; f32x4 ws0, ws1; float dec; u32x4 m0, mh, ml; };
  auto load_rows = [&](FragR& f, int kb, int jj, int part) {   // part 0 / 1: strip 0 / 1 of the sub-chunk
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (part == 0) f.q0[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * (32 * jj)]);
      else f.q1[i] = ld16(&sm.Q[kb][o_rd[i] + 128 * (32 * jj + 16)]);
    }
  };
  // the phase-2 operands of a sub-chunk, requested in eight pieces between the MFMA pairs of phase 1
  auto load_cols = [&](FragC& f, int kb, int ub, int jj, int piece) {
    const int r0 = 32 * jj;
    if (piece == 0) {
      f.ws0 = *reinterpret_cast<const f32x4*>(&sm.ws[kb][hh][r0 + 4 * g16]);
      f.ws1 = *reinterpret_cast<const f32x4*>(&sm.ws[kb][hh][r0 + 16 + 4 * g16]);
#pragma unroll
      for (int cg = 0; cg < 2; cg++) {
        const s16x4 a0 = lds_read_tr16_b64(&sm.U[ub][hh][o_uf[cg] + 64 * r0]);
        const s16x4 a1 = lds_read_tr16_b64(&sm.U[ub][hh][o_uf[cg] + 64 * (r0 + 16)]);
        f.u01[cg][0] = a0[0]; f.u01[cg][1] = a0[1]; f.u01[cg][2] = a0[2]; f.u01[cg][3] = a0[3];
        f.u01[cg][4] = a1[0]; f.u01[cg][5] = a1[1]; f.u01[cg][6] = a1[2]; f.u01[cg][7] = a1[3];
      }
      f.dec = sm.dec[kb][hh][jj];
    } else if (piece == 1) {
#pragma unroll
      for (int cg = 0; cg < 2; cg++) {
        const s16x4 a0 = lds_read_tr16_b64(&sm.U[ub][hh][o_uf[cg] + 64 * r0]);
        const s16x4 a1 = lds_read_tr16_b64(&sm.U[ub][hh][o_uf[cg] + 64 * r0]);
        f.u00[cg][0] = a0[0]; f.u00[cg][1] = a0[1]; f.u00[cg][2] = a0[2]; f.u00[cg][3] = a0[3];
        f.u00[cg][4] = a1[0]; f.u00[cg][5] = a1[1]; f.u00[cg][6] = a1[2]; f.u00[cg][7] = a1[3];
      }
      f.m0 = sm.M[ub][hh][3 * jj][lane];
      f.mh = sm.M[ub][hh][3 * jj + 1][lane];
      f.ml = sm.M[ub][hh][3 * jj + 2][lane];
    } else if (piece >= 2 && piece <= 5) {
      const int i = piece - 2;
#pragma unroll
      for (int t = 2 * i; t < 2 * i + 2; t++) {
        f.kt[t][0] = lds_read_tr16_b64(&sm.K[kb][o_kt[t >> 1] + 4 * (t & 1) + 128 * r0]);
        f.kt[t][1] = lds_read_tr16_b64(&sm.K[kb][o_kt[t >> 1] + 4 * (t & 1) + 128 * (r0 + 16)]);
      }
    } else if (piece == 6) {
      f.rl0 = sm.rl[kb][hh][r0 + t16];
      f.rl1 = sm.rl[kb][hh][r0 + 16 + t16];
    }
  };
  // window-state images: segment (4 wq + i) 64 + lane of ssd_tiles.h (img_off), wq = 2 w + cg the 16-column group
  const uint32_t dump_nb = (DUMP && a.dump) ? (uint32_t)((((int64_t)a.dump_nw - 1) * a.H + 1) << 14) : 0u;
  const BufRes Pr = make_buf((DUMP && a.dump) ? a.dump + ((((int64_t)b * a.dump_nw) * a.H + h) << 13) : nullptr, dump_nb);
  const uint32_t pvo = 16u * (uint32_t)(512 * w + lane);
  f32x4 accA0[2], accA1[2];
  u32x4 uh[2], ul[2];
  auto pack_pair = [&](int cg, int i) -> u32x4 {
    u32x4 sp;
    sp[0] = pack_bf16x2(accS[cg][2 * i][0], accS[cg][2 * i][1]);
    sp[1] = pack_bf16x2(accS[cg][2 * i][2], accS[cg][2 * i][3]);
    sp[2] = pack_bf16x2(accS[cg][2 * i + 1][0], accS[cg][2 * i + 1][1]);
    sp[3] = pack_bf16x2(accS[cg][2 * i + 1][2], accS[cg][2 * i + 1][3]);
    return sp;
  };
  // the scaled U operand of the state update (bf16 hi, and lo when final states are kept) of column group cg, strip s2
  auto scale_u = [&](const FragC& f, int cg, int s2) {
    const f32x4& ws4 = s2 ? f.ws1 : f.ws0;
    float us[4];
#pragma unroll
    for (int e = 0; e < 4; e++) us[e] = bf16_to_f32((uint16_t)f.u01[cg][4 * s2 + e]) * ws4[e];
#pragma unroll
    for (int p2 = 0; p2 < 2; p2++) {
      const uint32_t hi = pack_bf16x2(us[2 * p2], us[2 * p2 + 1]);
      uh[cg][2 * s2 + p2] = hi;
      if (KHILO) ul[cg][2 * s2 + p2] = pack_bf16x2(us[2 * p2] - bf_lo(hi), us[2 * p2 + 1] - bf_hi(hi));
    }
  };
  auto pack_pair_lo = [&](int cg, int i, const u32x4& hi) -> u32x4 {   // PRECISE: what the bf16 rounding of pack_pair left behind
    u32x4 sp;
    sp[0] = pack_bf16x2(accS[cg][2 * i][0] - bf_lo(hi[0]), accS[cg][2 * i][1] - bf_hi(hi[0]));
    sp[1] = pack_bf16x2(accS[cg][2 * i][2] - bf_lo(hi[1]), accS[cg][2 * i][3] - bf_hi(hi[1]));
    sp[2] = pack_bf16x2(accS[cg][2 * i + 1][0] - bf_lo(hi[2]), accS[cg][2 * i + 1][1] - bf_hi(hi[2]));
    sp[3] = pack_bf16x2(accS[cg][2 * i + 1][2] - bf_lo(hi[3]), accS[cg][2 * i + 1][3] - bf_hi(hi[3]));
    return sp;
  };
  // ---- phase 1 of a sub-chunk: pack of the state slice + S_in^T Q^T on the Q row fragments of its two strips; in the shadows of the MFMA
  // pairs: the pack of the next tile pair, the requests for the phase-2 operands (nf), the scaled U operand
  auto phase1 = [&](const FragR& f, bool dump_slot, bool dump_here, uint32_t dso, FragC& nf, int nkb, int nub, int njj) OMK_ALWAYS_INLINE_LAMBDA {
    u32x4 sp = pack_pair(0, 0);
    load_cols(nf, nkb, nub, njj, 0);
    OMK_SCHED_FENCE();
#pragma unroll
    for (int n = 0; n < 8; n++) {   // n = 2 i + cg
      const int i = n >> 1, cg = n & 1;
      if (DUMP && dump_slot && dump_here) buf_st16(Pr, sp, pvo + 4096u * (uint32_t)cg + 1024u * (uint32_t)i, dso);
      if (i == 0) {
        accA0[cg] = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q0[0]), f32x4{0.f, 0.f, 0.f, 0.f});
        accA1[cg] = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q1[0]), f32x4{0.f, 0.f, 0.f, 0.f});
      } else {
        accA0[cg] = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q0[i]), accA0[cg]);
        accA1[cg] = mfma16x16x32_bf16(as_s16x8(sp), as_s16x8(f.q1[i]), accA1[cg]);
      }
      if (PRECISE) {
        const u32x4 spl = pack_pair_lo(cg, i, sp);
        accA0[cg] = mfma16x16x32_bf16(as_s16x8(spl), as_s16x8(f.q0[i]), accA0[cg]);
        accA1[cg] = mfma16x16x32_bf16(as_s16x8(spl), as_s16x8(f.q1[i]), accA1[cg]);
      }
      if (n < 7) sp = pack_pair((n + 1) & 1, (n + 1) >> 1);
      if (n < 6) load_cols(nf, nkb, nub, njj, n + 1);
      if (n >= 4) scale_u(nf, (n - 4) >> 1, (n - 4) & 1);   // (the U fragments and ws were requested in front of the phase)
      OMK_SCHED_FENCE();
    }
  };
  auto out_rows = [&](f32x4 o0, f32x4 o1, int jj, int s2, uint32_t so) {   // o0 / o1: the lane's row in the column groups 0 / 1
    uint32_t x0 = pack_bf16x2(o0[0], o0[1]), x1 = pack_bf16x2(o0[2], o0[3]), y0 = pack_bf16x2(o1[0], o1[1]), y1 = pack_bf16x2(o1[2], o1[3]);
    wave_swap16(x0, y0);
    wave_swap16(x1, y1);
    const u32x4 ov = {x0, x1, y0, y1};
    if (OMK_A8_VAR & 256) { OMK_KEEP(ov); return; }   // (ablation: no output stores)
    buf_st16(Or, ov, ovo[jj][s2], so);
  };
  // ---- phase 2: U^T M^T (intra block), state update, output rows; in the shadows: decay of the next tile, the output arithmetic, the
  // requests for the Q row fragments of the next sub-chunk (nr)
  // (the decay multiply of the state slice: only when the basis of the carried state moves -- dec != 1.f, the scalar wave's mark -- in one
  // block in front of the MFMAs of the state update)
  auto phase2 = [&](const FragC& f, int jj, uint32_t so, FragR& nr, int nkb, int njj) OMK_ALWAYS_INLINE_LAMBDA {
    f32x4 accB0[2], accB1[2];
#pragma unroll
    for (int cg = 0; cg < 2; cg++) {
      accB0[cg] = mfma16x16x32_bf16(f.u00[cg], as_s16x8(f.m0), f32x4{0.f, 0.f, 0.f, 0.f});
      accB1[cg] = mfma16x16x32_bf16(f.u01[cg], as_s16x8(f.mh), f32x4{0.f, 0.f, 0.f, 0.f});
    }
#pragma unroll
    for (int cg = 0; cg < 2; cg++) accB1[cg] = mfma16x16x32_bf16(f.u01[cg], as_s16x8(f.ml), accB1[cg]);   // (not right behind the MFMA it accumulates on)
    if (uniform_i((int)__builtin_bit_cast(uint32_t, f.dec)) != 0x3f800000) {
#pragma unroll
      for (int cg = 0; cg < 2; cg++)
#pragma unroll
        for (int t = 0; t < 8; t++) accS[cg][t] = accS[cg][t] * f.dec;
    }
    OMK_SCHED_FENCE();
#pragma unroll
    for (int n = 0; n < 16; n++) {   // n = 2 t + cg
      const int t = n >> 1, cg = n & 1;
      s16x8 kk;
      kk[0] = f.kt[t][0][0]; kk[1] = f.kt[t][0][1]; kk[2] = f.kt[t][0][2]; kk[3] = f.kt[t][0][3];
      kk[4] = f.kt[t][1][0]; kk[5] = f.kt[t][1][1]; kk[6] = f.kt[t][1][2]; kk[7] = f.kt[t][1][3];
      accS[cg][t] = mfma16x16x32_bf16(kk, as_s16x8(uh[cg]), accS[cg][t]);
      if (KHILO && n > 0) {   // the lo half of the tile before: one MFMA behind the hi half it accumulates on, not right behind it
        const int t1 = (n - 1) >> 1, c1_ = (n - 1) & 1;
        s16x8 k1;
        k1[0] = f.kt[t1][0][0]; k1[1] = f.kt[t1][0][1]; k1[2] = f.kt[t1][0][2]; k1[3] = f.kt[t1][0][3];
        k1[4] = f.kt[t1][1][0]; k1[5] = f.kt[t1][1][1]; k1[6] = f.kt[t1][1][2]; k1[7] = f.kt[t1][1][3];
        accS[c1_][t1] = mfma16x16x32_bf16(k1, as_s16x8(ul[c1_]), accS[c1_][t1]);
      }
      if (KHILO && n == 15) accS[1][7] = mfma16x16x32_bf16(kk, as_s16x8(ul[1]), accS[1][7]);
      if (n == 4) out_rows(accA0[0] * f.rl0 + accB0[0], accA0[1] * f.rl0 + accB0[1], jj, 0, so);
      if (n == 8) out_rows(accA1[0] * f.rl1 + accB1[0], accA1[1] * f.rl1 + accB1[1], jj, 1, so);
      if (n == 10) load_rows(nr, nkb, njj, 0);
      if (n == 12) load_rows(nr, nkb, njj, 1);
      OMK_SCHED_FENCE();
    }
  };

  FragR fr;
  FragC fc;
  load_rows(fr, 0, 0, 0);
  load_rows(fr, 0, 0, 1);
  int kb0 = 0, kb1 = 1, kb2 = 2;
  PT8_START();
  // one chunk; ps: the register set that holds the U tile of chunk c + 1 (committed here, refilled with chunk c + 1 + USET)
  auto citer = [&](auto ps, int c) OMK_ALWAYS_INLINE_LAMBDA {
    const int ub0 = (c - c0) & 1;
    const uint32_t so = 2u * (uint32_t)(chunk_lo(c) * osl);
    bool dump_here = false;
    uint32_t dso = dump_nb;
    if (DUMP && a.dump) {   // window-boundary image of the state in front of this chunk (the [u][k] kx3 image ssd_cp.hip reads)
      const int cid = rev ? nC - 1 - c : c;
      dump_here = rev ? (cid == nC - 1 || (cid & 1)) : !(cid & 1);
      if (dump_here) dso = (uint32_t)(((int64_t)(cid >> 1) * a.H) << 14);
    }
    if (!(OMK_A8_VAR & 1024)) {   // (ablation 1024: the compute waves only stage and meet the barrier)
    phase1(fr, true, dump_here, dso, fc, kb0, ub0, 0);
    PT8(0);
    phase2(fc, 0, so, fr, kb0, 1);
    PT8(1);
    phase1(fr, false, false, dump_nb, fc, kb0, ub0, 1);
    }
    if ((OMK_A8_VAR & 1024) && (OMK_A8_VAR & 4096)) {   // (ablation 1024 + 4096: the memory skeleton -- every load, commit and output store of a chunk, no arithmetic)
#pragma unroll
      for (int jj = 0; jj < 2; jj++)
#pragma unroll
        for (int s2 = 0; s2 < 2; s2++) {
          if (OMK_A8_VAR & 16384) {  // (... as 256-byte pieces, both heads of the pair: four rows x 256 bytes per instruction, wave 2 hh + w takes rows with bits 3:2 == its index)
            const int row = 32 * jj + 16 * s2 + 4 * (2 * hh + w) + (lane >> 4);
            const BufRes Or2 = make_buf((uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)(2 * hp) * a.osh, (uint32_t)((int64_t)a.L * osl * 2));
            buf_st16(Or2, u32x4{0u, 0u, 0u, 0u}, 2u * (uint32_t)(rowtok(row) * osl + 8 * (lane & 15)), so);
          } else
          if (OMK_A8_VAR & 8192) {   // (the same bytes as FULL 128-byte lines: wave w takes the rows with bit 3 == w, eight rows x 128 bytes per instruction)
            const int row = 32 * jj + 16 * s2 + 8 * w + (lane >> 3);
            buf_st16(Or, u32x4{0u, 0u, 0u, 0u}, 2u * (uint32_t)(rowtok(row) * osl + 8 * (lane & 7)), so);
          } else buf_st16(Or, u32x4{0u, 0u, 0u, 0u}, ovo[jj][s2], so);
        }
    }
    if (CUS && !(OMK_A8_VAR & 64)) commit_u(ps, ub0 ^ 1);                                   // U of chunk c + 1
    if (CUS && !(OMK_A8_VAR & 128)) prefetch_u(ps, chunk_lo(clipc(c + 1 + USET)));
    PT8(2);
    block_sync();   // behind the last request for the buffers of chunk c
    PT8(3);
    if (!(OMK_A8_VAR & 1024)) phase2(fc, 1, so, fr, kb1, 0);
    PT8(4);
    { const int t_ = kb0; kb0 = kb1; kb1 = kb2; kb2 = t_; }
  };
  if (USET == 2) {
    for (int c = c0; c < c1; c += 2) {
      citer(U0{}, c);
      if (c + 1 < c1) citer(U1{}, c + 1);
    }
  } else {
    for (int c = c0; c < c1; c++) citer(U0{}, c);
  }
  PT8_END();
  if (a.fin && seg == a.nseg - 1) {
    const float extra = a.fin_extra_decay ? expf(dtrow[0] * a.A[h]) : 1.f;
#pragma unroll
    for (int cg = 0; cg < 2; cg++)
#pragma unroll
      for (int t = 0; t < 8; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int k = 32 * (t >> 1) + 8 * g16 + 4 * (t & 1) + r, su = 32 * w + 16 * cg + t16;
          a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)su * a.fsu + (int64_t)k * a.fsk] = accS[cg][t][r] * extra;
        }
  }
}

// the plain class A scans (one D per head or none, no gate, no pre-gate copy); GSF_COLUMN_SLICE: ssd_a6.hip takes them
bool ssd_a8_applies(const GScan& g) {
  if (g.flags & GSF_COLUMN_SLICE) return false;
  if (g.mode != GS_Y && g.mode != GS_DX) return false;
  if (g.H % 2 != 0 || (g.H / g.G) % 2 != 0 || g.state_only) return false;
  if (g.Z.p || g.outx || (g.D && g.Dsp != 0)) return false;
  return true;
}

int ssd_a8_launch(const GScan& g, omk_stream stream) {
  if (getenv("OMK_SSD_TRACE")) fprintf(stderr, "[omk] ssd_a8 mode %d B %d L %d H %d dump %d fin %d seg %d flags %d\n", g.mode, g.B, g.L, g.H, g.dump != nullptr, g.fin != nullptr, g.seg != nullptr, g.flags);
  GScan a = g;
  const SegPlan sp = a.seg ? ssd_segments(a.B * a.H, a.L) : SegPlan{1, (a.L + QA8 - 1) / QA8};
  a.nseg = sp.nseg; a.cps = sp.cps;
  const bool precise = a.mode == GS_Y && (a.flags & GSF_PRECISE);
  if (precise && a.dump) return fail(OMK_EINVAL, "ssd_a8: a PRECISE forward does not save window states");
  // the fused conv (forward-only path): plain forward of an unsplit sequence
  if (a.cw && (a.mode != GS_Y || a.dump || precise || a.reverse || a.nseg > 1 || a.cW < 1 || a.cW > 4)) return OMK_EUNSUPPORTED;
  if (a.nseg > 1 && !a.seg_ready) {
    int rc = ssd_mfma_prepare_segments(g, stream);
    if (rc) return rc;
  }
  dim3 grid((unsigned)(a.B * (a.H / 2) * a.nseg)), block(512);
  const size_t smem = sizeof(SmemA8);
  // the scaled U operand of the state update as hi + lo whenever the caller keeps the final state (prefill -> decode hand-off,
  // context-parallel shards) or asks for it (OMK_SSD_KHILO / OMK_SSD_PRECISE)
  const bool khilo = a.mode == GS_Y && ((a.flags & (GSF_KHILO | GSF_PRECISE)) || a.fin != nullptr);
#define OMK_A8K(MODE_, DU_, KH_, PR_) do { \
    kernels_note("ssd_a8<mode=%d,dump=%d,khilo=%d,precise=%d>", (int)MODE_, (int)DU_, (int)KH_, (int)PR_); \
    if (OMK_SET_MAX_DYN_SMEM((ssd_a8_kernel<MODE_, DU_, KH_, PR_>), smem)) return fail(OMK_ELAUNCH, "ssd_a8: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_a8_kernel<MODE_, DU_, KH_, PR_>), grid, block, smem, stream, a); } while (0)
#define OMK_A8C(KH_) do { \
    kernels_note("ssd_a8<mode=0,dump=0,khilo=%d,precise=0,conv=1>", (int)KH_); \
    if (OMK_SET_MAX_DYN_SMEM((ssd_a8_kernel<GS_Y, false, KH_, false, true>), smem)) return fail(OMK_ELAUNCH, "ssd_a8: cannot raise dynamic LDS to %zu", smem); \
    OMK_LAUNCH((ssd_a8_kernel<GS_Y, false, KH_, false, true>), grid, block, smem, stream, a); } while (0)
  if (a.cw) { if (khilo) OMK_A8C(true); else OMK_A8C(false); }
  else if (a.mode == GS_Y) {
    if (precise) OMK_A8K(GS_Y, false, true, true);
    else if (a.dump) { if (khilo) OMK_A8K(GS_Y, true, true, false); else OMK_A8K(GS_Y, true, false, false); }
    else { if (khilo) OMK_A8K(GS_Y, false, true, false); else OMK_A8K(GS_Y, false, false, false); }
  } else {
    if (a.dump) OMK_A8K(GS_DX, true, false, false); else OMK_A8K(GS_DX, false, false, false);
  }
#undef OMK_A8K
#undef OMK_A8C
  return OMK_OK;
}

}  // namespace omk
