// ssd_mfma.hip -- the MFMA chunked scan for the OmniMamba-1.3B block shape (headdim 64, d_state 128, bf16).
//
// One 512-thread workgroup (8 waves, 2 per SIMD) owns TWO heads of one batch element and walks the sequence in
// chunks of 64 tokens, carrying both running states (64 x 128 fp32 each) in MFMA accumulator registers from the
// first token to the last: states never touch HBM, so HBM traffic is the algorithmic minimum -- every x / dy row
// read once, every y / dx row written once, B/C (shared by all 64 heads of a group) served from L2.
//
// Per chunk (local token l, s; a = log-decay, cs = inclusive prefix of a inside the chunk, w = input scale):
//   G^T[s][l] = K_s . Q_l                                    16x16x32 MFMA, shared by both heads, via LDS (fp32)
//   M[l][s]   = G[l][s] * exp(cs_l - cs_s) * w_s  (s <= l)    VALU, rounded to bf16 as the next A operand
//   O_l       = sum_s M[l][s] U_s  +  exp(cs_l) * (Q_l . S_in)   2 x 32x32x16 MFMA chains (U via ds_read_tr16_b64)
//   S_out     = exp(cs_63) S_in + sum_l (w_l exp(cs_63 - cs_l) U_l) (x) K_l   MFMA, both operands via transpose reads
// The intra-chunk prefix cs is a 64-lane wave scan (__shfl_up); lanes = tokens.  S_in reaches the O chain as bf16
// through LDS ([u][k], k contiguous = natural B-operand order).  Time-reversed scans (backward) only change the
// global<->LDS row mapping and the decay index, the core is direction agnostic.
//
// Wave w: head hh = w >> 2, (wi, wj) = ((w >> 1) & 1, w & 1): owns O tile [32 wi .. +32][32 wj .. +32] and the two
// state tiles S^T[64 wi + 32 kt .. +32][32 wj .. +32].
#include "ssd_scan.h"

namespace omk {

constexpr int QC = 64;     // chunk length (tokens)
constexpr int LDK = 136;   // row stride (bf16 elements) of 128-wide tiles: 272 B = 17 x 16 B -> conflict-free b128 rows
constexpr int LDU = 72;    // row stride of 64-wide bf16 tiles
constexpr int LDG = 68;    // row stride (floats) of 64-wide fp32 tiles

struct SmemA {
  uint16_t K[QC * LDK];
  uint16_t Qm[QC * LDK];
  uint16_t U[2][QC * LDU];
  float G[QC * LDG];
  uint16_t S[2][64 * LDK];   // [u][k] bf16 copy of S_in
  float O[2][QC * LDG];
  float cs[2][QC], ecs[2][QC], w[2][QC], ws[2][QC];
  float dtl[2][2][QC];       // [chunk parity][head][row]: dt' of the token itself
  float dta[2][2][QC];       // dt' driving the row's decay (the next token's in reverse scans)
  float Dv[2][64];           // D of the two heads, per column (broadcast when D is (H))
};

__device__ __forceinline__ u32x4 ld16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }
__device__ __forceinline__ s16x8 as_s16x8(u32x4 v) { return __builtin_bit_cast(s16x8, v); }
__device__ __forceinline__ float bf_lo(uint32_t v) { return bf16_to_f32((uint16_t)(v & 0xffffu)); }
__device__ __forceinline__ float bf_hi(uint32_t v) { return bf16_to_f32((uint16_t)(v >> 16)); }

// B/A operand fragment (8 contraction values for this lane's column/row) out of a row-major [contraction][col] LDS
// tile with two transpose reads: rows r0 + 8*h32 + 4*m + {0..3}, column c0 + (lane & 31).
__device__ __forceinline__ s16x8 tr_frag(const uint16_t* tile, int ld, int r0, int c0, int lane) {
  const int t16 = lane & 15, g16 = lane >> 4, h32 = lane >> 5;
  const uint16_t* p = tile + (r0 + 8 * h32 + (t16 >> 2)) * ld + c0 + 16 * (g16 & 1) + 4 * (t16 & 3);
  s16x4 a = lds_read_tr16_b64(p);
  s16x4 b = lds_read_tr16_b64(p + 4 * ld);
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// MODE: GS_Y (forward output y) or GS_DX (time-reversed, dx)
//
// Wave roles inside a head (4 waves): wave (wi, wj) always owns the two state tiles S^T[64 wi + 32 kt ..][32 wj ..];
// for the output it is either a "D" wave -- builds the decay-masked M fragments of l-tile wi ONCE and runs the
// M.U chain for both u-tiles -- or an "S" wave -- runs the Q.S_in chain of l-tile wi for both u-tiles.  The two
// partial outputs meet in LDS (S wave stores exp2(cs) * acc, D wave ds_add's).  Heads use opposite role maps so the two
// waves sharing a SIMD (w, w + 4) are one VALU-heavy D wave and one MFMA-heavy S wave.
template <int MODE>
__global__ __launch_bounds__(512) void ssd_mfma_a_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemA& sm = *reinterpret_cast<SmemA*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const bool roleD = ((wj ^ hh) & 1) == 0;
  const int h32 = lane >> 5, l31 = lane & 31, g16 = lane >> 4, t16 = lane & 15;
  const int pairs = a.H / 2;
  const int b = blockIdx.x / pairs, hp = blockIdx.x % pairs;
  const int h0 = hp * 2;
  const int g = h0 / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
  const bool rev = a.reverse != 0;
  const int tstep = rev ? -QC : QC;

  // ---- staging with pointer-increment addressing: token of (chunk 0, row) once, then +-64 tokens per chunk
  auto tok0 = [&](int row) -> int { return rev ? (nC - 1) * QC + (QC - 1) - row : row; };
  u32x4 rk[2], rq[2], ru[2];
  float rdt = 0.f;
  int tk[2], tu = tok0(tid >> 3), tdt = tok0(tid & 63);
  const uint16_t *pk[2], *pq[2], *pu[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int seg = tid + 512 * r, row = seg >> 4, cs8 = (seg & 15) * 8;
    tk[r] = tok0(row);
    pk[r] = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh + (int64_t)tk[r] * a.K.sl + cs8;
    pq[r] = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh + (int64_t)tk[r] * a.Q.sl + cs8;
    pu[r] = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)(h0 + r) * a.U.sh + (int64_t)tu * a.U.sl + (tid & 7) * 8;
  }
  const int64_t dK = (int64_t)tstep * a.K.sl, dQ = (int64_t)tstep * a.Q.sl, dU = (int64_t)tstep * a.U.sl;
  // Loads are UNCONDITIONAL (a runtime-predicated load makes hipcc branch around it and wait vmcnt(0) at the join,
  // which serialised every chunk behind a full HBM round trip): out-of-range rows read a clamped in-range address and
  // are zeroed when the registers are committed to LDS.
  const uint16_t* safeK = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
  const uint16_t* safeQ = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
  const uint16_t* safeU = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h0 * a.U.sh;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h0 + ((tid >> 6) & 1)) * a.L;
  int cload = 0;   // chunk the staging registers currently point at
  bool okk[2] = {false, false}, oku = false;
  float rda = 0.f;   // dt' that drives the decay of the row (reverse scans: the NEXT token's)
  auto prefetch = [&]() {
    const bool in = cload < nC;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      okk[r] = in && tk[r] < a.L;
      rk[r] = ld16(okk[r] ? pk[r] : safeK);
      rq[r] = ld16(okk[r] ? pq[r] : safeQ);
      oku = in && tu < a.L;
      ru[r] = ld16(oku ? pu[r] : safeU);
      pk[r] += dK; pq[r] += dQ; pu[r] += dU; tk[r] += tstep;
    }
    const bool okd = in && tdt < a.L;
    const int ta = rev ? tdt + 1 : tdt;
    rdt = dtrow[okd ? tdt : 0];
    rda = dtrow[(okd && ta < a.L) ? ta : 0];
    if (!okd) rdt = 0.f;
    if (!(okd && ta < a.L)) rda = 0.f;
    tdt += tstep; tu += tstep;
    cload++;
  };
  auto commit = [&](int par) {   // registers -> LDS tiles (dt' goes to the parity buffer of the chunk it belongs to)
    const u32x4 zero4 = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int seg = tid + 512 * r, row = seg >> 4, cs8 = (seg & 15) * 8;
      st16(&sm.K[row * LDK + cs8], okk[r] ? rk[r] : zero4);
      st16(&sm.Qm[row * LDK + cs8], okk[r] ? rq[r] : zero4);
      st16(&sm.U[r][(tid >> 3) * LDU + (tid & 7) * 8], oku ? ru[r] : zero4);
    }
    if (tid < 128) { sm.dtl[par][tid >> 6][tid & 63] = rdt; sm.dta[par][tid >> 6][tid & 63] = rda; }
  };

  // ---- running state: S^T tiles [k = 64 wi + 32 kt + row][u = 32 wj + l31], head h0 + hh
  f32x16 accS[2];
  const int hcur = h0 + hh;
  const float Ah = a.A[hcur];
  const float Ah2 = Ah * LOG2E;   // decays are carried in log2 units so every exp is a bare v_exp_f32
#pragma unroll
  for (int kt = 0; kt < 2; kt++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float v = 0.f;
      if (a.init) {
        const int k = 64 * wi + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 32 * wj + l31;
        v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)hcur * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
      }
      accS[kt][r] = v;
    }
  auto publish_state = [&]() {   // bf16 copy of this wave's tiles into sm.S[hh][u][k]
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int rq4 = 0; rq4 < 4; rq4++) {
        const int k = 64 * wi + 32 * kt + 8 * rq4 + 4 * h32;
        u32x2 v;
        v[0] = pack_bf16x2(accS[kt][4 * rq4 + 0], accS[kt][4 * rq4 + 1]);
        v[1] = pack_bf16x2(accS[kt][4 * rq4 + 2], accS[kt][4 * rq4 + 3]);
        *reinterpret_cast<u32x2*>(&sm.S[hh][(32 * wj + l31) * LDK + k]) = v;
      }
  };

  prefetch();
  commit(0);
  publish_state();
  if (tid < 128) sm.Dv[tid >> 6][tid & 63] = a.D ? load_rt(a.D, (int64_t)(h0 + (tid >> 6)) * a.Dsh + (int64_t)(tid & 63) * a.Dsp, a.D_dt) : 0.f;
  block_sync();

  // epilogue mapping: thread = (head r, row tid>>3, 8 columns); token of the row advances by +-64 per chunk
  int tep = tok0(tid >> 3);
  const uint16_t* pz[2] = {nullptr, nullptr};
  int64_t po[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    po[r] = (int64_t)b * a.osb + (int64_t)tep * a.osl + (int64_t)(h0 + r) * a.osh + (tid & 7) * 8;
    if (MODE == GS_Y && a.Z.p) pz[r] = (const uint16_t*)a.Z.p + (int64_t)b * a.Z.sb + (int64_t)tep * a.Z.sl + (int64_t)(h0 + r) * a.Z.sh + (tid & 7) * 8;
  }
  const int64_t dO = (int64_t)tstep * a.osl, dZ = (int64_t)tstep * a.Z.sl;

#ifdef OMK_PHASE_PROF   // developer build (tools/phase_prof.py): s_memtime deltas per phase, workgroup 0
  uint64_t pt[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
#define PT(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
  uint64_t tprev = prof ? clock64_() : 0;
#else
#define PT(i) do { } while (0)
#endif
  for (int c = 0; c < nC; c++) {
    prefetch();
    PT(0);
    // ---- scalars: one wave per head, lanes = tokens
    if ((wave & 3) == 0) {
      const int id = rev ? nC - 1 - c : c;
      const int t = rev ? id * QC + (QC - 1) - lane : id * QC + lane;
      const bool ok = t < a.L;
      const float d = sm.dtl[c & 1][hh][lane];
      float cs = sm.dta[c & 1][hh][lane] * Ah2;   // zero for rows past the end
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        float o = shfl_up(cs, off);
        if (lane >= off) cs += o;
      }
      const float cs_end = shfl(cs, 63);
      const float wv = ok ? (a.w_is_dt ? d : 1.f) : 0.f;
      sm.cs[hh][lane] = cs;              // log2 units
      sm.ecs[hh][lane] = exp2_fast(cs);
      sm.w[hh][lane] = wv;
      sm.ws[hh][lane] = wv * exp2_fast(cs_end - cs);
    }
    // ---- G^T = K Q^T, lower triangle of 16x16 tiles (s-tile ta <= l-tile tb), 4 MFMA each
    for (int tile = wave; tile < 10; tile += 8) {
      int ta, tb;
      if (tile < 1) { tb = 0; ta = tile; } else if (tile < 3) { tb = 1; ta = tile - 1; } else if (tile < 6) { tb = 2; ta = tile - 3; } else { tb = 3; ta = tile - 6; }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        s16x8 fa = as_s16x8(ld16(&sm.K[(16 * ta + t16) * LDK + 32 * kk + 8 * g16]));
        s16x8 fb = as_s16x8(ld16(&sm.Qm[(16 * tb + t16) * LDK + 32 * kk + 8 * g16]));
        acc = mfma16x16x32_bf16(fa, fb, acc);
      }
      *reinterpret_cast<f32x4*>(&sm.G[(16 * tb + t16) * LDG + 16 * ta + 4 * g16]) = acc;
    }
    PT(1);
    block_sync();   // B1: G and scalars visible
    PT(2);

    f32x16 accX[2];   // D wave: M.U for u-tiles 0/1 ; S wave: Q.S_in for u-tiles 0/1   (l-tile wi)
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int r = 0; r < 16; r++) accX[ut][r] = 0.f;
    if (roleD) {
      const int l = 32 * wi + l31;
      const float cs_l = sm.cs[hh][l];
      const int nks = 2 * (wi + 1);
      for (int ks = 0; ks < nks; ks++) {
        const int s0 = 16 * ks + 8 * h32;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&sm.G[l * LDG + s0]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&sm.G[l * LDG + s0 + 4]);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(&sm.cs[hh][s0]);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(&sm.cs[hh][s0 + 4]);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(&sm.w[hh][s0]);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(&sm.w[hh][s0 + 4]);
        // M is the one rounding point that dominates the error of y, so it is fed as hi + lo bf16 pairs (2 MFMAs)
        u32x4 mp, ml;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          const f32x2 gv = e2 < 2 ? f32x2{g0[2 * e2], g0[2 * e2 + 1]} : f32x2{g1[2 * e2 - 4], g1[2 * e2 - 3]};
          const f32x2 cv = e2 < 2 ? f32x2{c0[2 * e2], c0[2 * e2 + 1]} : f32x2{c1[2 * e2 - 4], c1[2 * e2 - 3]};
          const f32x2 wv = e2 < 2 ? f32x2{w0[2 * e2], w0[2 * e2 + 1]} : f32x2{w1[2 * e2 - 4], w1[2 * e2 - 3]};
          const f32x2 dd = f32x2{cs_l, cs_l} - cv;
          f32x2 v = gv * wv * f32x2{exp2_fast(dd[0]), exp2_fast(dd[1])};
          const int s = s0 + 2 * e2;
          v[0] = (s <= l) ? v[0] : 0.f;
          v[1] = (s + 1 <= l) ? v[1] : 0.f;
          const uint32_t hi = pack_bf16x2(v[0], v[1]);
          mp[e2] = hi;
          ml[e2] = pack_bf16x2(v[0] - bf_lo(hi), v[1] - bf_hi(hi));
        }
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
          s16x8 fb = tr_frag(sm.U[hh], LDU, 16 * ks, 32 * ut, lane);
          accX[ut] = mfma32x32x16_bf16(as_s16x8(mp), fb, accX[ut]);
          accX[ut] = mfma32x32x16_bf16(as_s16x8(ml), fb, accX[ut]);
        }
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        s16x8 fa = as_s16x8(ld16(&sm.Qm[(32 * wi + l31) * LDK + 16 * ks + 8 * h32]));
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
          s16x8 fb = as_s16x8(ld16(&sm.S[hh][(32 * ut + l31) * LDK + 16 * ks + 8 * h32]));
          accX[ut] = mfma32x32x16_bf16(fa, fb, accX[ut]);
        }
      }
      // S wave lands first: exp2(cs_l) * (Q . S_in) as plain stores (O of the previous chunk was consumed before B4)
#pragma unroll
      for (int ut = 0; ut < 2; ut++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int l = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32;
          sm.O[hh][l * LDG + 32 * ut + l31] = sm.ecs[hh][l] * accX[ut][r];
        }
    }
    PT(3);
    // ---- state update: S^T[k][u] = exp(cs_end) S^T + sum_l K^T[k][l] (ws_l U[l][u])
    {
      const float dec = sm.ecs[hh][QC - 1];
#pragma unroll
      for (int kt = 0; kt < 2; kt++) accS[kt] *= dec;
#pragma unroll
      for (int ls = 0; ls < 4; ls++) {
        s16x8 fu = tr_frag(sm.U[hh], LDU, 16 * ls, 32 * wj, lane);
        const int lb = 16 * ls + 8 * h32;
        const f32x4 s0v = *reinterpret_cast<const f32x4*>(&sm.ws[hh][lb]);
        const f32x4 s1v = *reinterpret_cast<const f32x4*>(&sm.ws[hh][lb + 4]);
        u32x4 up;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          const f32x2 uv = {bf16_to_f32((uint16_t)fu[2 * e2]), bf16_to_f32((uint16_t)fu[2 * e2 + 1])};
          const f32x2 sv = e2 < 2 ? f32x2{s0v[2 * e2], s0v[2 * e2 + 1]} : f32x2{s1v[2 * e2 - 4], s1v[2 * e2 - 3]};
          const f32x2 pr = uv * sv;
          up[e2] = pack_bf16x2(pr[0], pr[1]);
        }
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
          s16x8 fk = tr_frag(sm.K, LDK, 16 * ls, 64 * wi + 32 * kt, lane);
          accS[kt] = mfma32x32x16_bf16(fk, as_s16x8(up), accS[kt]);
        }
      }
    }
    PT(4);
    block_sync();   // B2: every wave is done reading S_in, G, K, Q, U tiles; the S waves' part of O is in LDS
    PT(5);
    // epilogue operands (issued here so they are not live across the MFMA section) straight from HBM into registers
    u32x4 ez[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      ez[r] = u32x4{0, 0, 0, 0};
      if (MODE == GS_Y && a.Z.p) ez[r] = ld16(tep < a.L ? pz[r] : (const uint16_t*)a.Z.p);   // wave-uniform branch, unconditional load
    }
    publish_state();
    if (roleD) {
#pragma unroll
      for (int ut = 0; ut < 2; ut++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int l = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32;
          sm.O[hh][l * LDG + 32 * ut + l31] += accX[ut][r];   // plain read-modify-write: this lane is the only writer after B2
                                                               // (an LDS float atomic here compiled to a CAS loop: 21K of 34K cycles/chunk)
        }
    }
    PT(6);
    block_sync();   // B3: O tile and the new bf16 state are complete
    PT(7);
    // ---- epilogue: thread = (head r, row tid>>3, 8 columns)
    {
      const int row = tid >> 3, c8 = (tid & 7) * 8;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const f32x4 o0 = *reinterpret_cast<const f32x4*>(&sm.O[r][row * LDG + c8]);
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(&sm.O[r][row * LDG + c8 + 4]);
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(&sm.Dv[r][c8]), d1 = *reinterpret_cast<const f32x4*>(&sm.Dv[r][c8 + 4]);
        const u32x4 uv = ld16(&sm.U[r][row * LDU + c8]);
        const float sc = MODE == GS_DX ? sm.dtl[c & 1][r][row] : 1.f;
        float res[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float o = e < 4 ? o0[e & 3] : o1[e & 3];
          const float dv = e < 4 ? d0[e & 3] : d1[e & 3];
          const float uu = (e & 1) ? bf_hi(uv[e >> 1]) : bf_lo(uv[e >> 1]);
          res[e] = sc * o + dv * uu;
        }
        if (MODE == GS_Y) {
          if (a.outx && tep < a.L) {
            u32x4 px;
            px[0] = pack_bf16x2(res[0], res[1]); px[1] = pack_bf16x2(res[2], res[3]); px[2] = pack_bf16x2(res[4], res[5]); px[3] = pack_bf16x2(res[6], res[7]);
            st16((uint16_t*)a.outx + po[r], px);
          }
          if (a.Z.p) {
#pragma unroll
            for (int e = 0; e < 8; e++) res[e] *= silu_fast((e & 1) ? bf_hi(ez[r][e >> 1]) : bf_lo(ez[r][e >> 1]));
          }
        }
        u32x4 pzv;
        pzv[0] = pack_bf16x2(res[0], res[1]); pzv[1] = pack_bf16x2(res[2], res[3]); pzv[2] = pack_bf16x2(res[4], res[5]); pzv[3] = pack_bf16x2(res[6], res[7]);
        if (tep < a.L) st16((uint16_t*)a.out + po[r], pzv);
        po[r] += dO;
        if (pz[r]) pz[r] += dZ;
      }
      tep += tstep;
    }
    commit((c + 1) & 1);   // tiles of chunk c+1 (no reader of chunk c's tiles is left after B2; the epilogue read its own sm.U segment before this)
    PT(8);
    block_sync();   // B4
    PT(9);
  }
#ifdef OMK_PHASE_PROF
  if (prof && lane == 0)
    for (int i = 0; i < 10; i++) a.prof[wave * 10 + i] = pt[i];
#endif
  if (a.fin) {
    const float extra = a.fin_extra_decay ? expf(a.dtp[((int64_t)b * a.H + hcur) * a.L] * Ah) : 1.f;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int k = 64 * wi + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 32 * wj + l31;
        a.fin[(int64_t)b * a.fsb + (int64_t)hcur * a.fsh + (int64_t)u * a.fsu + (int64_t)k * a.fsk] = accS[kt][r] * extra;
      }
  }
}

// =========================================================================================================
// class A, version 2: ONE head per 256-thread workgroup (4 waves), 80 KB of LDS so that TWO workgroups share a CU and
// run out of phase: one workgroup's barrier / LDS-latency bubbles are filled by the other's MFMA and VALU work
// (version 1 keeps 8 waves of one workgroup in lock-step; PMC: 44 % of its wave cycles are waits).  Each wave owns a
// complete output tile O[32 wi ..][32 wj ..] (M.U and Q.S_in chains in the same accumulators' sum) and writes it to
// HBM straight from the MFMA register layout -- no LDS round trip for O -- and the two state tiles
// S^T[64 wi + 32 kt ..][32 wj ..].  G is stored unpadded with a 16-byte XOR swizzle to fit the LDS budget.
// =========================================================================================================
struct SmemA2 {
  uint16_t K[QC * LDK];
  uint16_t Qm[QC * LDK];
  uint16_t U[QC * LDU];
  float G[QC * 64];        // element (l, s) at l*64 + ((((s >> 2) ^ (l & 15)) << 2) | (s & 3))
  uint16_t S[64 * LDK];    // [u][k] bf16 copy of S_in
  float cs[QC], ecs[QC], w[QC], ws[QC];
  float dtl[2][QC], dta[2][QC];
  float Dv[64];
};
static_assert(sizeof(SmemA2) <= 80 * 1024, "two workgroups must fit the 160 KB of a CU");

template <int MODE>
__global__ __launch_bounds__(256, 2) void ssd_mfma_a2_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemA2& sm = *reinterpret_cast<SmemA2*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int h32 = lane >> 5, l31 = lane & 31, g16 = lane >> 4, t16 = lane & 15;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int g = h / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
  const bool rev = a.reverse != 0;
  const int rowdir = rev ? -1 : 1;

  // ---- staging: K, Q four 16-byte segments per thread (rows row0 + 16 r), U two (rows rowu + 32 r)
  const int rowk = tid >> 4, ck8 = (tid & 15) * 8, rowu = tid >> 3, cu8 = (tid & 7) * 8;
  const uint16_t* Kb = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh + ck8;
  const uint16_t* Qb = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh + ck8;
  const uint16_t* Ub = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh + cu8;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  u32x4 rk[4], rq[4], ru[2];
  float rdt = 0.f, rda = 0.f;
  int tbase = 0;   // token of LDS row 0 of the chunk held in the staging registers
  auto chunk_base = [&](int c) -> int { const int id = rev ? nC - 1 - c : c; return rev ? id * QC + (QC - 1) : id * QC; };
  auto prefetch = [&](int c) {   // unconditional loads from clamped addresses; rows past the end are zeroed at commit
    tbase = c < nC ? chunk_base(c) : -(1 << 28);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int t = tbase + rowdir * (rowk + 16 * r);
      const int tc = (t >= 0 && t < a.L) ? t : 0;
      rk[r] = ld16(Kb + (int64_t)tc * a.K.sl);
      rq[r] = ld16(Qb + (int64_t)tc * a.Q.sl);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int t = tbase + rowdir * (rowu + 32 * r);
      const int tc = (t >= 0 && t < a.L) ? t : 0;
      ru[r] = ld16(Ub + (int64_t)tc * a.U.sl);
    }
    {
      const int t = tbase + rowdir * (tid & 63);
      const bool okd = t >= 0 && t < a.L;
      const int ta = rev ? t + 1 : t;
      const bool oka = okd && ta < a.L;
      rdt = dtrow[okd ? t : 0];
      rda = dtrow[oka ? ta : 0];
      if (!okd) rdt = 0.f;
      if (!oka) rda = 0.f;
    }
  };
  auto commit = [&](int par) {
    const u32x4 zero4 = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int t = tbase + rowdir * (rowk + 16 * r);
      const bool ok = t >= 0 && t < a.L;
      st16(&sm.K[(rowk + 16 * r) * LDK + ck8], ok ? rk[r] : zero4);
      st16(&sm.Qm[(rowk + 16 * r) * LDK + ck8], ok ? rq[r] : zero4);
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int t = tbase + rowdir * (rowu + 32 * r);
      const bool ok = t >= 0 && t < a.L;
      st16(&sm.U[(rowu + 32 * r) * LDU + cu8], ok ? ru[r] : zero4);
    }
    if (tid < 64) { sm.dtl[par][tid] = rdt; sm.dta[par][tid] = rda; }
  };

  // ---- running state
  f32x16 accS[2];
  const float Ah = a.A[h];
  const float Ah2 = Ah * LOG2E;
#pragma unroll
  for (int kt = 0; kt < 2; kt++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float v = 0.f;
      if (a.init) {
        const int k = 64 * wi + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 32 * wj + l31;
        v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
      }
      accS[kt][r] = v;
    }
  auto publish_state = [&]() {
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int rq4 = 0; rq4 < 4; rq4++) {
        const int k = 64 * wi + 32 * kt + 8 * rq4 + 4 * h32;
        u32x2 v;
        v[0] = pack_bf16x2(accS[kt][4 * rq4 + 0], accS[kt][4 * rq4 + 1]);
        v[1] = pack_bf16x2(accS[kt][4 * rq4 + 2], accS[kt][4 * rq4 + 3]);
        *reinterpret_cast<u32x2*>(&sm.S[(32 * wj + l31) * LDK + k]) = v;
      }
  };

  prefetch(0);
  commit(0);
  publish_state();
  if (tid < 64) sm.Dv[tid] = a.D ? load_rt(a.D, (int64_t)h * a.Dsh + (int64_t)tid * a.Dsp, a.D_dt) : 0.f;
  block_sync();
  const float Du = sm.Dv[32 * wj + l31];   // this lane's output column never changes
  uint16_t* outp = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh + 32 * wj + l31;
  uint16_t* outxp = a.outx ? (uint16_t*)a.outx + (int64_t)b * a.osb + (int64_t)h * a.osh + 32 * wj + l31 : nullptr;
  const uint16_t* zp = (MODE == GS_Y && a.Z.p) ? (const uint16_t*)a.Z.p + (int64_t)b * a.Z.sb + (int64_t)h * a.Z.sh + 32 * wj + l31 : nullptr;

#ifdef OMK_PHASE_PROF   // developer build (tools/phase_prof.py): s_memtime deltas per phase, workgroup 0
  uint64_t pt[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const bool prof = a.prof != nullptr && blockIdx.x == 0;
#define PT2(i) do { if (prof) { uint64_t n_ = clock64_(); pt[i] += n_ - tprev; tprev = n_; } } while (0)
  uint64_t tprev = prof ? clock64_() : 0;
#else
#define PT2(i) do { } while (0)
#endif
  for (int c = 0; c < nC; c++) {
    const int tb0 = chunk_base(c);
    prefetch(c + 1);
    PT2(0);
    if (wave == 0) {   // scalars: lanes = tokens
      const int t = tb0 + rowdir * lane;
      const bool ok = t >= 0 && t < a.L;
      const float d = sm.dtl[c & 1][lane];
      float cs = sm.dta[c & 1][lane] * Ah2;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        float o = shfl_up(cs, off);
        if (lane >= off) cs += o;
      }
      const float cs_end = shfl(cs, 63);
      const float wv = ok ? (a.w_is_dt ? d : 1.f) : 0.f;
      sm.cs[lane] = cs;
      sm.ecs[lane] = exp2_fast(cs);
      sm.w[lane] = wv;
      sm.ws[lane] = wv * exp2_fast(cs_end - cs);
    }
    for (int tile = wave; tile < 10; tile += 4) {   // G^T = K Q^T, lower triangle of 16x16 tiles
      int ta, tb;
      if (tile < 1) { tb = 0; ta = tile; } else if (tile < 3) { tb = 1; ta = tile - 1; } else if (tile < 6) { tb = 2; ta = tile - 3; } else { tb = 3; ta = tile - 6; }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        s16x8 fa = as_s16x8(ld16(&sm.K[(16 * ta + t16) * LDK + 32 * kk + 8 * g16]));
        s16x8 fb = as_s16x8(ld16(&sm.Qm[(16 * tb + t16) * LDK + 32 * kk + 8 * g16]));
        acc = mfma16x16x32_bf16(fa, fb, acc);
      }
      const int l = 16 * tb + t16;
      *reinterpret_cast<f32x4*>(&sm.G[l * 64 + (((4 * ta + g16) ^ (l & 15)) << 2)]) = acc;
    }
    PT2(1);
    block_sync();   // B1
    PT2(2);

    f32x16 accD, accO;
#pragma unroll
    for (int r = 0; r < 16; r++) { accD[r] = 0.f; accO[r] = 0.f; }
    {
      const int l = 32 * wi + l31;
      const float cs_l = sm.cs[l];
      const int nks = 2 * (wi + 1);
      for (int ks = 0; ks < nks; ks++) {
        const int s0 = 16 * ks + 8 * h32;
        const f32x4 g0 = *reinterpret_cast<const f32x4*>(&sm.G[l * 64 + ((((s0 >> 2)) ^ (l & 15)) << 2)]);
        const f32x4 g1 = *reinterpret_cast<const f32x4*>(&sm.G[l * 64 + ((((s0 >> 2) + 1) ^ (l & 15)) << 2)]);
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(&sm.cs[s0]);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(&sm.cs[s0 + 4]);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(&sm.w[s0]);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(&sm.w[s0 + 4]);
        u32x4 mp, ml;   // M as hi + lo bf16 (the rounding point that dominates the error of y)
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          const f32x2 gv = e2 < 2 ? f32x2{g0[2 * e2], g0[2 * e2 + 1]} : f32x2{g1[2 * e2 - 4], g1[2 * e2 - 3]};
          const f32x2 cv = e2 < 2 ? f32x2{c0[2 * e2], c0[2 * e2 + 1]} : f32x2{c1[2 * e2 - 4], c1[2 * e2 - 3]};
          const f32x2 wv = e2 < 2 ? f32x2{w0[2 * e2], w0[2 * e2 + 1]} : f32x2{w1[2 * e2 - 4], w1[2 * e2 - 3]};
          const f32x2 dd = f32x2{cs_l, cs_l} - cv;
          f32x2 v = gv * wv * f32x2{exp2_fast(dd[0]), exp2_fast(dd[1])};
          const int s = s0 + 2 * e2;
          v[0] = (s <= l) ? v[0] : 0.f;
          v[1] = (s + 1 <= l) ? v[1] : 0.f;
          const uint32_t hi = pack_bf16x2(v[0], v[1]);
          mp[e2] = hi;
          ml[e2] = pack_bf16x2(v[0] - bf_lo(hi), v[1] - bf_hi(hi));
        }
        s16x8 fb = tr_frag(sm.U, LDU, 16 * ks, 32 * wj, lane);
        accD = mfma32x32x16_bf16(as_s16x8(mp), fb, accD);
        accD = mfma32x32x16_bf16(as_s16x8(ml), fb, accD);
      }
    }
    PT2(3);
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      s16x8 fa = as_s16x8(ld16(&sm.Qm[(32 * wi + l31) * LDK + 16 * ks + 8 * h32]));
      s16x8 fb = as_s16x8(ld16(&sm.S[(32 * wj + l31) * LDK + 16 * ks + 8 * h32]));
      accO = mfma32x32x16_bf16(fa, fb, accO);
    }
    PT2(4);
    // O = M.U + exp2(cs_l) Q.S_in + D u : everything this lane needs for its 16 outputs, taken before the tiles die
    float ov[16];
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int l = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32;
      const float uu = bf16_to_f32(sm.U[l * LDU + 32 * wj + l31]);
      float o = accD[r] + sm.ecs[l] * accO[r];
      if (MODE == GS_DX) o *= sm.dtl[c & 1][l];
      ov[r] = o + Du * uu;
    }
    PT2(5);
    {   // state update: S^T[k][u] = exp(cs_end) S^T + sum_l K^T[k][l] (ws_l U[l][u])
      const float dec = sm.ecs[QC - 1];
#pragma unroll
      for (int kt = 0; kt < 2; kt++) accS[kt] *= dec;
#pragma unroll
      for (int ls = 0; ls < 4; ls++) {
        s16x8 fu = tr_frag(sm.U, LDU, 16 * ls, 32 * wj, lane);
        const int lb = 16 * ls + 8 * h32;
        const f32x4 s0v = *reinterpret_cast<const f32x4*>(&sm.ws[lb]);
        const f32x4 s1v = *reinterpret_cast<const f32x4*>(&sm.ws[lb + 4]);
        u32x4 up;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          const f32x2 uv = {bf16_to_f32((uint16_t)fu[2 * e2]), bf16_to_f32((uint16_t)fu[2 * e2 + 1])};
          const f32x2 sv = e2 < 2 ? f32x2{s0v[2 * e2], s0v[2 * e2 + 1]} : f32x2{s1v[2 * e2 - 4], s1v[2 * e2 - 3]};
          const f32x2 pr = uv * sv;
          up[e2] = pack_bf16x2(pr[0], pr[1]);
        }
#pragma unroll
        for (int kt = 0; kt < 2; kt++) {
          s16x8 fk = tr_frag(sm.K, LDK, 16 * ls, 64 * wi + 32 * kt, lane);
          accS[kt] = mfma32x32x16_bf16(fk, as_s16x8(up), accS[kt]);
        }
      }
    }
    PT2(6);
    block_sync();   // B2: nobody reads S_in, G or this chunk's tiles any more
    PT2(7);
    publish_state();
    commit((c + 1) & 1);
    PT2(8);
    // epilogue straight from the MFMA layout: for each register, 32 lanes cover 64 contiguous bytes of one output row
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int l = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32;
      const int t = tb0 + rowdir * l;
      if (t >= 0 && t < a.L) {
        float v = ov[r];
        if (MODE == GS_Y) {
          if (outxp) outxp[(int64_t)t * a.osl] = f32_to_bf16(v);
          if (zp) v *= silu_fast(bf16_to_f32(zp[(int64_t)t * a.Z.sl]));
        }
        outp[(int64_t)t * a.osl] = f32_to_bf16(v);
      }
    }
    PT2(9);
    block_sync();   // B3: next chunk's tiles and the new bf16 state are visible
    PT2(10);
  }
#ifdef OMK_PHASE_PROF
  if (prof && lane == 0)
    for (int i = 0; i < 12; i++) a.prof[wave * 12 + i] = pt[i];
#endif
  if (a.fin) {
    const float extra = a.fin_extra_decay ? expf(dtrow[0] * Ah) : 1.f;
#pragma unroll
    for (int kt = 0; kt < 2; kt++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int k = 64 * wi + 32 * kt + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 32 * wj + l31;
        a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)u * a.fsu + (int64_t)k * a.fsk] = accS[kt][r] * extra;
      }
  }
}

// =========================================================================================================
// class A, version 3 ("row strips"): one head per 256-thread workgroup, two workgroups per CU.
//
// Wave w owns the output rows l in [16 w, 16 w + 16) of every 64-token chunk (all 64 columns) and the state rows
// k in [32 w, 32 w + 32) (all 64 columns).  What that buys over versions 1 / 2:
//   * G never leaves registers.  G^T[s][l] = K_s . Q_l comes out of the 16x16x32 MFMA with the lane's own l in every
//     register; the MFMA contraction index may be permuted freely as long as both operands agree, so the two G tiles of
//     a 32-wide s block ARE the A-operand fragment of M (after the decay/mask/bf16 hi+lo split), and U is fetched with
//     the matching permuted transpose reads.  No fp32 G tile in LDS, no barrier between G and M.
//   * M is built exactly once per chunk (versions 1 / 2 built it in two waves).
//   * Q is never staged: its MFMA fragments (rows of the wave's strip) are loaded from global memory one chunk ahead
//     and serve both G (as B operand) and Q . S_in (as A operand).
//   * O = exp2(cs_l) (Q . S_in) + M . U accumulates in ONE register tile per wave and goes to HBM from registers.
//   * K / U tiles are double buffered: two barriers per chunk (after the S_in reads, after the S_out publish).
// =========================================================================================================
// LDS layouts of version 3: unpadded rows, the 16-byte segment index XOR-ed with a function of the row so that every
// access pattern of the kernel is bank-conflict free under the gfx950 lane groups (MI355X_MICROARCH.md, LDS):
//   K / S tiles (256 B rows): seg ^ swzK(row).  ds_read_b128 "row t16, segment 4 c + g16" (lane groups
//     {0-3,12-15,20-27}, ...) needs swzK bijective on row & 15 with swzK({4..11}) closed under ^1; the transpose reads
//     "4 rows x 4 segments per half wave" need swzK(row) >> 2 distinct over 4 consecutive rows.
//   64-column tiles (128 B rows, two rows per 256 B bank row): seg ^ swzU(row) with swzU = swzK & 7 -- searched the same
//     way over every pattern the class A / class B kernels use on them (8 rows x 2 segments, 4 rows x 4 segments,
//     16 rows x 1 segment, the ds_read_b128 row reads, rows {0-3, 8-11} x 2 segments).
__device__ __forceinline__ int swzK(int r) { return ((r & 1) << 3) | ((r & 2) << 1) | ((((r >> 2) ^ (r >> 3)) & 1) << 1) | ((r >> 2) & 1); }
__device__ __forceinline__ int swzU(int r) { return swzK(r) & 7; }
__device__ __forceinline__ int kx3(int row, int col) { return row * 128 + ((((col >> 3) ^ swzK(row)) << 3) | (col & 7)); }
__device__ __forceinline__ int ux3(int row, int col) { return row * 64 + ((((col >> 3) ^ swzU(row)) << 3) | (col & 7)); }
// 32x32x16 operand fragment out of a swizzled row-major [contraction][col] tile (K: 128 columns, U: 64 columns):
// rows r0 + 8 h32 + 4 m + {0..3}, column c0 + (lane & 31)
template <bool KT>
__device__ __forceinline__ s16x8 tr_frag3(const uint16_t* tile, int r0, int c0, int lane) {
  const int t16 = lane & 15, g16 = lane >> 4, h32 = lane >> 5;
  const int row = r0 + 8 * h32 + (t16 >> 2), col = c0 + 16 * (g16 & 1) + 4 * (t16 & 3);
  const s16x4 a = lds_read_tr16_b64(tile + (KT ? kx3(row, col) : ux3(row, col)));
  const s16x4 b = lds_read_tr16_b64(tile + (KT ? kx3(row + 4, col) : ux3(row + 4, col)));
  s16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}
struct SmemA3 {
  uint16_t K[2][QC * 128];
  uint16_t U[2][QC * 64];
  uint16_t S[64 * 128];      // [u][k] bf16 copy of S_in
  float cs[2][QC], lw[2][QC], ecs[2][QC], ws[2][QC], dtl[2][QC];   // lw = log2(w) - cs  (M = G exp2(cs_l + lw_s))
  float Dv[64];
};
static_assert(sizeof(SmemA3) <= 80 * 1024, "two workgroups must fit the 160 KB of a CU");

template <int MODE, bool EXTRAS>   // EXTRAS: gate z and / or the pre-gate copy of y (forward without the gated norm)
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
  const int b = vid / a.H, h = vid % a.H;
  const int g = h / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
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
  const uint32_t koff0 = (uint32_t)(rowtok(rowk) * ksl + ck8), uoff0 = (uint32_t)(rowtok(rowu) * usl + cu8);
  const uint32_t qoff0 = (uint32_t)(rowtok(16 * w + t16) * qsl + 8 * g16);
  const int kstep = (rev ? -16 : 16) * ksl, ustep = (rev ? -32 : 32) * usl;
  u32x4 rk[4], ru[2], qf[4];
  float rdt = 0.f, rda = 0.f, rwv = 0.f;
  int stlo = 0;   // tlo of the chunk held in the staging registers
  // The loads of the next chunk are spread over the phases of the current one (a burst of ten 1 KB loads per wave
  // stalls on the 64 B/clk address path): K after barrier X, Q fragments, U and dt after the intra phase.  They are
  // branch-free on purpose: rows past the end of a ragged last chunk read the chunk's first row instead (the commit
  // zeroes them).  With vector-memory instructions under control flow the compiler can no longer count what is in
  // flight and falls back to s_waitcnt vmcnt(0) -- which also waits for the output stores of the previous chunk.
  const int rtk_k = rowtok(rowk), rtk_u = rowtok(rowu), rtk_q = rowtok(16 * w + t16), rtk_l = rowtok(lane);
  const int dk16 = rev ? -16 : 16, du32 = rev ? -32 : 32;
  auto prefetch_k = [&]() {
    const int lim = a.L - stlo;   // rows with rowtok < lim are in range
    const uint16_t* Kc = Kb + (int64_t)stlo * ksl;
#pragma unroll
    for (int r = 0; r < 4; r++) rk[r] = ld16(Kc + (rtk_k + dk16 * r < lim ? koff0 + (uint32_t)(r * kstep) : (uint32_t)ck8));
  };
  auto prefetch_q = [&]() {   // straight into the live fragment registers: issued after their last use of the chunk
    const int lim = a.L - stlo;
    const uint16_t* Qc = Qb + (int64_t)stlo * qsl;
    const uint32_t qo = rtk_q < lim ? qoff0 : (uint32_t)(8 * g16);   // rows past the end: any in-range row, never stored
#pragma unroll
    for (int kk = 0; kk < 4; kk++) qf[kk] = ld16(Qc + 32 * kk + qo);
  };
  auto prefetch_u = [&]() {
    const int lim = a.L - stlo;
    const uint16_t* Uc = Ub + (int64_t)stlo * usl;
#pragma unroll
    for (int r = 0; r < 2; r++) ru[r] = ld16(Uc + (rtk_u + du32 * r < lim ? uoff0 + (uint32_t)(r * ustep) : (uint32_t)cu8));
    // token scalars (consumed by wave 0, loaded by every wave to keep the instruction stream uniform): lanes = rows
    const int t = stlo + rtk_l, ta = rev ? t + 1 : t;
    rdt = dtrow[t < a.L ? t : 0];   // raw loads: the selects wait in scalars() so nothing stalls on them here
    rda = dtrow[ta < a.L ? ta : 0];
  };
  const int o_ck = kx3(rowk, ck8), o_cu = ux3(rowu, cu8);
  auto commit = [&](int buf) {
    if (stlo + QC <= a.L) {
#pragma unroll
      for (int r = 0; r < 4; r++) st16(&sm.K[buf][o_ck + 16 * 128 * r], rk[r]);
#pragma unroll
      for (int r = 0; r < 2; r++) st16(&sm.U[buf][o_cu + 32 * 64 * r], ru[r]);
    } else {
      const u32x4 zero4 = {0, 0, 0, 0};
#pragma unroll
      for (int r = 0; r < 4; r++) st16(&sm.K[buf][o_ck + 16 * 128 * r], stlo + rowtok(rowk + 16 * r) < a.L ? rk[r] : zero4);
#pragma unroll
      for (int r = 0; r < 2; r++) st16(&sm.U[buf][o_cu + 32 * 64 * r], stlo + rowtok(rowu + 32 * r) < a.L ? ru[r] : zero4);
    }
  };
  const float Ah = a.A[h];
  const float Ah2 = Ah * LOG2E;
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
    for (int r = 0; r < 16; r++) {
      float v = 0.f;
      if (a.init) {
        const int k = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 32 * ut + l31;
        v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
      }
      accS[ut][r] = v;
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

  stlo = chunk_lo(0);
  prefetch_q();
  prefetch_k();
  prefetch_u();
  commit(0);
  if (w == 0) scalars(0);
  publish_state();
  if (tid < 64) sm.Dv[tid] = a.D ? load_rt(a.D, (int64_t)h * a.Dsh + (int64_t)tid * a.Dsp, a.D_dt) : 0.f;
  block_sync();
  uint16_t* ob = (uint16_t*)a.out + (int64_t)b * a.osb + (int64_t)h * a.osh;
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
#else
#define PT3(i) do { } while (0)
#endif
  for (int c = 0; c < nC; c++) {
    const int cur = c & 1, nxt = cur ^ 1;
    const int tlo = chunk_lo(c);
    const int cnext = c + 1 < nC ? c + 1 : c;   // the last iteration re-stages its own chunk: no branch around loads
    PT3(0);
    // ---- (1) O = exp2(cs_l) * (Q . S_in): A = Q fragments (registers), B = S_in[k][u] as [u][k] bf16 rows
    f32x4 acc[4];
#pragma unroll
    for (int ut = 0; ut < 4; ut++) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int ut = 0; ut < 4; ut++) {
        const s16x8 fs = as_s16x8(ld16(&sm.S[o_rd[kk] + 16 * 128 * ut]));
        acc[ut] = mfma16x16x32_bf16(fs, as_s16x8(qf[kk]), acc[ut]);   // O^T[u][l]: 4 consecutive u per lane
      }
    {
      const float e1 = sm.ecs[cur][16 * w + t16];
#pragma unroll
      for (int ut = 0; ut < 4; ut++) acc[ut] *= e1;
    }
    PT3(1);
    block_sync();   // X: every wave is done with S_in
    PT3(2);
    stlo = chunk_lo(cnext);
    prefetch_k();
    prefetch_u();
    // ---- (2) intra-chunk: G tiles -> M fragments (registers) -> M . U
    {
      const float cs_l = sm.cs[cur][16 * w + t16];
      auto block = [&](int kk, bool second, bool diag0, bool diag1) {
        // tiles ta = 2 kk + j (j = 0, 1) of G^T: lane holds s = 16 ta + 4 g16 + r (r = 0..3) for its own l = 16 w + t16
        u32x4 mh, ml;
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
            if (j == 0 ? diag0 : diag1) v[r] = (4 * g16 + r <= t16) ? v[r] : 0.f;
          }
#pragma unroll
          for (int p2 = 0; p2 < 2; p2++) {   // bf16 hi + lo: the rounding of M dominates the error of y otherwise
            const uint32_t hi = pack_bf16x2(v[2 * p2], v[2 * p2 + 1]);
            mh[2 * j + p2] = hi;
            ml[2 * j + p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi), v[2 * p2 + 1] - bf_hi(hi));
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
          acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(ml), acc[ut]);
        }
      };
      // strip w: s blocks kk = 0 .. w >> 1; the diagonal tile is ta = w, tiles ta > w are empty
      if (w == 0) { block(0, false, true, false); }
      else if (w == 1) { block(0, true, false, true); }
      else if (w == 2) { block(0, true, false, false); block(1, false, true, false); }
      else { block(0, true, false, false); block(1, true, false, true); }
    }
    PT3(3);
    prefetch_q();
    // ---- (3) state update: S^T[k][u] = exp2(cs_end) S^T + sum_l (ws_l K^T[k][l]) U[l][u]
    {
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
        u32x4 kp;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          const f32x2 kv = {bf16_to_f32((uint16_t)fk[2 * e2]), bf16_to_f32((uint16_t)fk[2 * e2 + 1])};
          const f32x2 sv = e2 < 2 ? f32x2{s0v[2 * e2], s0v[2 * e2 + 1]} : f32x2{s1v[2 * e2 - 4], s1v[2 * e2 - 3]};
          const f32x2 pr = kv * sv;
          kp[e2] = pack_bf16x2(pr[0], pr[1]);
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
        }
      }
    }
    PT3(4);
    // ---- (4) publish S_out, stage the next chunk, next chunk's scalars
    publish_state();
    commit(nxt);
    if (w == 0) scalars(nxt);
    PT3(5);
    // ---- epilogue from the MFMA layout: 4 consecutive columns of one row per (lane, ut)
    {
      const int trow = tlo + erow;
      if (trow < a.L) {
        const float dts = MODE == GS_DX ? sm.dtl[cur][16 * w + t16] : 1.f;
        uint16_t* oc = ob + (int64_t)tlo * osl;
#pragma unroll
        for (int ut = 0; ut < 4; ut++) {
          const u32x2 xr = *reinterpret_cast<const u32x2*>(&sm.U[cur][o_xu[ut] + 16 * 64 * w]);
          const f32x4 Du = *reinterpret_cast<const f32x4*>(&sm.Dv[16 * ut + 4 * g16]);   // D of columns 16 ut + 4 g16 + r
          f32x4 v = acc[ut] * dts + Du * f32x4{bf_lo(xr[0]), bf_hi(xr[0]), bf_lo(xr[1]), bf_hi(xr[1])};
          if (MODE == GS_Y) {
            if (EXTRAS && oxb) {
              u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
              *reinterpret_cast<u32x2*>(oxb + (int64_t)tlo * osl + 16 * ut + eoff) = o;
            }
            if (EXTRAS && zb) {
              const u32x2 zr = *reinterpret_cast<const u32x2*>(zb + (int64_t)tlo * zsl + 16 * ut + zoff);
              v[0] *= silu_fast(bf_lo(zr[0])); v[1] *= silu_fast(bf_hi(zr[0]));
              v[2] *= silu_fast(bf_lo(zr[1])); v[3] *= silu_fast(bf_hi(zr[1]));
            }
          }
          u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
          *reinterpret_cast<u32x2*>(oc + 16 * ut + eoff) = o;
        }
      }
    }
    PT3(6);
    block_sync();   // Y: tiles, scalars and the bf16 state of the next chunk are visible
    PT3(7);
  }
#ifdef OMK_PHASE_PROF
  if (prof && lane == 0)
    for (int i = 0; i < 12; i++) a.prof[w * 12 + i] = pt[i];
#endif
  if (a.fin) {
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

// =========================================================================================================
// class B: U shared by the group and 128 wide (B or C), K / Q per head and 64 wide (x, dy): the dC and dB scans.
// State tiles S^T[k = p][u = n]; output O_h[l][n] of both heads is scaled (dB: by dt'_l), summed through LDS and
// written as one fp32 [64][128] tile per chunk to the per-head-pair partial buffer (reduced over head pairs by
// ssd_reduce_partials_kernel -- 64-way fp32 atomics measured 330 G/s on MI355X, 4x slower than partial tiles).
// Wave w: head hh = w >> 2, (wi, wj): O tiles [32 wi ..][64 wj + 32 ut ..], state tiles S^T[32 wi ..][64 wj + 32 ut ..].
// =========================================================================================================
struct SmemB {
  uint16_t U[QC * LDK];
  uint16_t K[2][QC * LDU];
  uint16_t Qm[2][QC * LDU];
  union {
    float G[2][QC * LDG];       // per-head G (dead after the M fragments are built)
    float O[QC * 132];          // summed output tile, written after barrier B2
  };
  uint16_t S[2][128 * LDU];     // [u][k] bf16 copy of S_in
  uint16_t X4[QC * LDK];        // C (dC scan) / B (dB scan) rows for the per-token scalar e_l / w_l = X4_l . O_l
  float cs[2][QC], ecs[2][QC], w[2][QC], ws[2][QC];
  float dtl[2][2][QC];
  float rdot[2][2][QC];         // [wj][head][row] partial row dots
  float bred[8];                // boundary dot partials (one per wave)
};

template <int MODE>
__global__ __launch_bounds__(512) void ssd_mfma_b_kernel(GScan a) {
  OMK_DYN_SMEM(smem_raw);
  SmemB& sm = *reinterpret_cast<SmemB*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hh = wave >> 2, wi = (wave >> 1) & 1, wj = wave & 1;
  const int h32 = lane >> 5, l31 = lane & 31, g16 = lane >> 4, t16 = lane & 15;
  const int pairs = a.H / 2;
  const int b = blockIdx.x / pairs, hp = blockIdx.x % pairs;
  const int h0 = hp * 2;
  const int g = h0 / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
  const bool rev = a.reverse != 0;
  auto tok = [&](int c, int row) -> int {
    const int id = rev ? nC - 1 - c : c;
    return rev ? id * QC + (QC - 1) - row : id * QC + row;
  };
  // staging: U (64 x 128) two segments per thread; K, Q (2 heads x 64 x 64) one segment per thread per head
  u32x4 ruu[2], rk[2], rq[2], rx4[2];
  float dDp[2][8] = {{0.f}};
  float rdt = 0.f;
  const bool has_x4 = a.X4.p != nullptr && a.tokscal != nullptr;
  const uint16_t* X4g = has_x4 ? (const uint16_t*)a.X4.p + (int64_t)b * a.X4.sb + (int64_t)g * a.X4.sh : nullptr;
  const uint16_t* Ug = (const uint16_t*)a.U.p + (int64_t)b * a.U.sb + (int64_t)g * a.U.sh;
  const uint16_t* Kg = (const uint16_t*)a.K.p + (int64_t)b * a.K.sb;
  const uint16_t* Qg = (const uint16_t*)a.Q.p + (int64_t)b * a.Q.sb;
  const float* dtp0 = a.dtp + ((int64_t)b * a.H + h0) * a.L;
  auto prefetch = [&](int c) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int seg = tid + 512 * r, row = seg >> 4, cs8 = (seg & 15) * 8;
      const int t = tok(c, row);
      const bool ok = c < nC && t < a.L;
      ruu[r] = ok ? ld16(Ug + (int64_t)t * a.U.sl + cs8) : u32x4{0, 0, 0, 0};
      rx4[r] = (ok && has_x4) ? ld16(X4g + (int64_t)t * a.X4.sl + cs8) : u32x4{0, 0, 0, 0};
      const int tu = tok(c, tid >> 3), cu8 = (tid & 7) * 8;
      const bool oku = c < nC && tu < a.L;
      rk[r] = oku ? ld16(Kg + (int64_t)tu * a.K.sl + (int64_t)(h0 + r) * a.K.sh + cu8) : u32x4{0, 0, 0, 0};
      rq[r] = oku ? ld16(Qg + (int64_t)tu * a.Q.sl + (int64_t)(h0 + r) * a.Q.sh + cu8) : u32x4{0, 0, 0, 0};
    }
    if (tid < 128) {
      const int t = tok(c, tid & 63);
      rdt = (c < nC && t < a.L) ? dtp0[(int64_t)(tid >> 6) * a.L + t] : 0.f;
    }
  };
  auto commit = [&](int par) {
#pragma unroll
    for (int r = 0; r < 2; r++) {
      const int seg = tid + 512 * r, row = seg >> 4, cs8 = (seg & 15) * 8;
      st16(&sm.U[row * LDK + cs8], ruu[r]);
      if (has_x4) st16(&sm.X4[row * LDK + cs8], rx4[r]);
      if (MODE == GS_DB && a.dD) {   // dD += dy . x over this thread's (head r, 8 columns): K = dy, Q = x
#pragma unroll
        for (int e = 0; e < 8; e++)
          dDp[r][e] += ((e & 1) ? bf_hi(rk[r][e >> 1]) : bf_lo(rk[r][e >> 1])) * ((e & 1) ? bf_hi(rq[r][e >> 1]) : bf_lo(rq[r][e >> 1]));
      }
      st16(&sm.K[r][(tid >> 3) * LDU + (tid & 7) * 8], rk[r]);
      st16(&sm.Qm[r][(tid >> 3) * LDU + (tid & 7) * 8], rq[r]);
    }
    if (tid < 128) sm.dtl[par][tid >> 6][tid & 63] = rdt;
  };
  f32x16 accS[2];
  const int hcur = h0 + hh;
  const float Ah = a.A[hcur];
  const float Ah2 = Ah * LOG2E;   // decays are carried in log2 units so every exp is a bare v_exp_f32
#pragma unroll
  for (int ut = 0; ut < 2; ut++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      float v = 0.f;
      if (a.init) {
        const int k = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 64 * wj + 32 * ut + l31;
        v = load_rt(a.init, (int64_t)b * a.isb + (int64_t)hcur * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt);
      }
      accS[ut][r] = v;
    }
  auto publish_state = [&]() {   // sm.S[hh][u][k]
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int rq4 = 0; rq4 < 4; rq4++) {
        const int k = 32 * wi + 8 * rq4 + 4 * h32;
        u32x2 v;
        v[0] = pack_bf16x2(accS[ut][4 * rq4 + 0], accS[ut][4 * rq4 + 1]);
        v[1] = pack_bf16x2(accS[ut][4 * rq4 + 2], accS[ut][4 * rq4 + 3]);
        *reinterpret_cast<u32x2*>(&sm.S[hh][(64 * wj + 32 * ut + l31) * LDU + k]) = v;
      }
  };
  prefetch(0);
  commit(0);
  publish_state();
  block_sync();
  float* part = a.part + ((int64_t)b * pairs + hp) * (int64_t)a.L * 128;
  // state checkpoints in fragment order: [b][pair][chunk][wave][ut][reg/2][lane] packed bf16 pairs (u32)
  uint32_t* ck = a.ckpt ? (uint32_t*)a.ckpt + ((int64_t)b * pairs + hp) * (int64_t)nC * (8 * 2 * 8 * 64) : nullptr;

  for (int c = 0; c < nC; c++) {
    prefetch(c + 1);
    if (MODE == GS_DB && ck && a.bnd) {
      // exact restart value of the decay-gradient prefix at the boundary behind chunk `id`:
      //   bnd[id + 1] = exp(a_first(id+1)) * < g_first(id+1) (= accS now), h_last(id) (= dC-scan checkpoint of chunk id) >
      const int id = nC - 1 - c;
      const uint32_t* cp = ck + (int64_t)id * (8 * 2 * 8 * 64) + wave * (2 * 8 * 64);
      float dot = 0.f;
#pragma unroll
      for (int ut = 0; ut < 2; ut++)
#pragma unroll
        for (int r2 = 0; r2 < 8; r2++) {
          const uint32_t v = cp[(ut * 8 + r2) * 64 + lane];
          dot += accS[ut][2 * r2] * bf_lo(v) + accS[ut][2 * r2 + 1] * bf_hi(v);
        }
      dot = wave_sum(dot);
      if (lane == 0) sm.bred[wave] = dot;
    }
    if ((wave & 3) == 0) {
      const int t = tok(c, lane);
      const bool ok = t < a.L;
      const float d = sm.dtl[c & 1][hh][lane];
      float la;
      if (rev) la = (ok && t + 1 < a.L) ? a.dtp[((int64_t)b * a.H + hcur) * a.L + t + 1] * Ah2 : 0.f;
      else la = ok ? d * Ah2 : 0.f;
      float cs = la;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        float o = shfl_up(cs, off);
        if (lane >= off) cs += o;
      }
      const float cs_end = shfl(cs, 63);
      const float wv = ok ? (a.w_is_dt ? d : 1.f) : 0.f;
      sm.cs[hh][lane] = cs;              // log2 units (la was scaled by log2 e)
      sm.ecs[hh][lane] = exp2_fast(cs);
      sm.w[hh][lane] = wv;
      sm.ws[hh][lane] = wv * exp2_fast(cs_end - cs);
    }
    // per-head G^T = K Q^T (contraction 64): 10 lower tiles over the head's 4 waves
    for (int tile = (wave & 3); tile < 10; tile += 4) {
      int ta, tb;
      if (tile < 1) { tb = 0; ta = tile; } else if (tile < 3) { tb = 1; ta = tile - 1; } else if (tile < 6) { tb = 2; ta = tile - 3; } else { tb = 3; ta = tile - 6; }
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        s16x8 fa = as_s16x8(ld16(&sm.K[hh][(16 * ta + t16) * LDU + 32 * kk + 8 * g16]));
        s16x8 fb = as_s16x8(ld16(&sm.Qm[hh][(16 * tb + t16) * LDU + 32 * kk + 8 * g16]));
        acc = mfma16x16x32_bf16(fa, fb, acc);
      }
      *reinterpret_cast<f32x4*>(&sm.G[hh][(16 * tb + t16) * LDG + 16 * ta + 4 * g16]) = acc;
    }
    block_sync();   // B1
    if (MODE == GS_DB && ck && a.bnd && tid < 2) {
      const int id = nC - 1 - c, hd = h0 + tid;
      const int tnext = (id + 1) * QC;
      const float ex = tnext < a.L ? expf(a.dtp[((int64_t)b * a.H + hd) * a.L + tnext] * a.A[hd]) : 1.f;
      a.bnd[((int64_t)b * a.H + hd) * (nC + 1) + id + 1] = ex * (sm.bred[4 * tid] + sm.bred[4 * tid + 1] + sm.bred[4 * tid + 2] + sm.bred[4 * tid + 3]);
    }
    f32x16 accD[2], accO[2];
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int r = 0; r < 16; r++) { accD[ut][r] = 0.f; accO[ut][r] = 0.f; }
    {
      const int l = 32 * wi + l31;
      const float cs_l = sm.cs[hh][l];
      const int nks = 2 * (wi + 1);
      for (int ks = 0; ks < nks; ks++) {
        const int s0 = 16 * ks + 8 * h32;
        f32x4 g0 = *reinterpret_cast<const f32x4*>(&sm.G[hh][l * LDG + s0]);
        f32x4 g1 = *reinterpret_cast<const f32x4*>(&sm.G[hh][l * LDG + s0 + 4]);
        u32x4 mp, ml;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          float m2[2];
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const int e = 2 * e2 + q, s = s0 + e;
            const float gv = e < 4 ? g0[e & 3] : g1[e & 3];
            const float v = gv * exp2_fast(cs_l - sm.cs[hh][s]) * sm.w[hh][s];
            m2[q] = (s <= l) ? v : 0.f;
          }
          const uint16_t h0b = f32_to_bf16(m2[0]), h1b = f32_to_bf16(m2[1]);
          mp[e2] = (uint32_t)h0b | ((uint32_t)h1b << 16);
          ml[e2] = pack_bf16x2(m2[0] - bf16_to_f32(h0b), m2[1] - bf16_to_f32(h1b));
        }
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
          s16x8 fb = tr_frag(sm.U, LDK, 16 * ks, 64 * wj + 32 * ut, lane);
          accD[ut] = mfma32x32x16_bf16(as_s16x8(mp), fb, accD[ut]);
          accD[ut] = mfma32x32x16_bf16(as_s16x8(ml), fb, accD[ut]);
        }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      s16x8 fa = as_s16x8(ld16(&sm.Qm[hh][(32 * wi + l31) * LDU + 16 * ks + 8 * h32]));
#pragma unroll
      for (int ut = 0; ut < 2; ut++) {
        s16x8 fb = as_s16x8(ld16(&sm.S[hh][(64 * wj + 32 * ut + l31) * LDU + 16 * ks + 8 * h32]));
        accO[ut] = mfma32x32x16_bf16(fa, fb, accO[ut]);
      }
    }
    {
      const float dec = sm.ecs[hh][QC - 1];
#pragma unroll
      for (int ut = 0; ut < 2; ut++)
#pragma unroll
        for (int r = 0; r < 16; r++) accS[ut][r] *= dec;
#pragma unroll
      for (int ls = 0; ls < 4; ls++) {
        s16x8 fk = tr_frag(sm.K[hh], LDU, 16 * ls, 32 * wi, lane);   // A operand K^T[k][l], scaled along l by ws
        const int lb = 16 * ls + 8 * h32;
        u32x4 kp;
#pragma unroll
        for (int e2 = 0; e2 < 4; e2++) {
          const float lo = bf16_to_f32((uint16_t)fk[2 * e2]) * sm.ws[hh][lb + 2 * e2];
          const float hi = bf16_to_f32((uint16_t)fk[2 * e2 + 1]) * sm.ws[hh][lb + 2 * e2 + 1];
          kp[e2] = pack_bf16x2(lo, hi);
        }
#pragma unroll
        for (int ut = 0; ut < 2; ut++) {
          s16x8 fu = tr_frag(sm.U, LDK, 16 * ls, 64 * wj + 32 * ut, lane);
          accS[ut] = mfma32x32x16_bf16(as_s16x8(kp), fu, accS[ut]);
        }
      }
    }
    if (has_x4) {
      // per-token scalar of THIS head: sum_n X4[l][n] * O_h[l][n] over this wave's 64 columns, folded over the 32 lanes
      // of each half; the two column halves (wj) meet in LDS.  (X4 is read before B2: commit() rewrites it after B4.)
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int l = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32;
        pv[r] = 0.f;
#pragma unroll
        for (int ut = 0; ut < 2; ut++)
          pv[r] += bf16_to_f32(sm.X4[l * LDK + 64 * wj + 32 * ut + l31]) * (accD[ut][r] + sm.ecs[hh][l] * accO[ut][r]);
      }
      // 16 row sums over the 32 lanes of each half as a halving butterfly: 8 + 4 + 2 + 1 + 1 = 16 shuffles instead of 80
      const bool b4 = (l31 & 16) != 0, b3 = (l31 & 8) != 0, b2 = (l31 & 4) != 0, b1 = (l31 & 2) != 0;
      float q8[8], q4[4], q2[2], q1;
#pragma unroll
      for (int i = 0; i < 8; i++) q8[i] = (b4 ? pv[i + 8] : pv[i]) + shfl_xor(b4 ? pv[i] : pv[i + 8], 16);
#pragma unroll
      for (int i = 0; i < 4; i++) q4[i] = (b3 ? q8[i + 4] : q8[i]) + shfl_xor(b3 ? q8[i] : q8[i + 4], 8);
#pragma unroll
      for (int i = 0; i < 2; i++) q2[i] = (b2 ? q4[i + 2] : q4[i]) + shfl_xor(b2 ? q4[i] : q4[i + 2], 4);
      q1 = (b1 ? q2[1] : q2[0]) + shfl_xor(b1 ? q2[0] : q2[1], 2);
      q1 += shfl_xor(q1, 1);
      if ((l31 & 1) == 0) {
        const int rr = (b1 ? 1 : 0) + (b2 ? 2 : 0) + (b3 ? 4 : 0) + (b4 ? 8 : 0);
        sm.rdot[wj][hh][32 * wi + (rr & 3) + 8 * (rr >> 2) + 4 * h32] = q1;
      }
    }
    block_sync();   // B2: G, S_in and the tiles of this chunk are no longer read
    publish_state();
    if (MODE == GS_DC && ck) {   // forward state at the END of this chunk, fragment order, bf16 pairs
      uint32_t* cp = ck + (int64_t)c * (8 * 2 * 8 * 64) + wave * (2 * 8 * 64);
#pragma unroll
      for (int ut = 0; ut < 2; ut++)
#pragma unroll
        for (int r2 = 0; r2 < 8; r2++) cp[(ut * 8 + r2) * 64 + lane] = pack_bf16x2(accS[ut][2 * r2], accS[ut][2 * r2 + 1]);
    }
    if (has_x4 && tid < 128) {
      const int hd = tid >> 6, l = tid & 63, t = tok(c, l);
      if (t < a.L) a.tokscal[((int64_t)b * a.H + h0 + hd) * a.L + t] = sm.rdot[0][hd][l] + sm.rdot[1][hd][l];
    }
    // head 0 writes its scaled tile, head 1 adds
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
      if (hh == pass) {
#pragma unroll
        for (int ut = 0; ut < 2; ut++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            const int l = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32;
            float v = accD[ut][r] + sm.ecs[hh][l] * accO[ut][r];
            if (MODE == GS_DB) v *= sm.dtl[c & 1][hh][l];
            float* o = &sm.O[l * 132 + 64 * wj + 32 * ut + l31];
            *o = pass == 0 ? v : *o + v;
          }
      }
      block_sync();   // B3a / B3b
    }
    {   // [64][128] fp32 tile -> partial buffer, 4 x 16 B per thread
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int seg = tid + 512 * r, row = seg >> 5, c4 = (seg & 31) * 4;
        const int t = tok(c, row);
        if (t < a.L) *reinterpret_cast<f32x4*>(part + (int64_t)t * 128 + c4) = *reinterpret_cast<const f32x4*>(&sm.O[row * 132 + c4]);
      }
    }
    commit((c + 1) & 1);
    block_sync();   // B4
  }
  if (MODE == GS_DB && a.dD) {
    float* redw = sm.G[0];   // [8 waves][2 heads][64 cols]
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float v = dDp[r][e];
        v += shfl_xor(v, 8); v += shfl_xor(v, 16); v += shfl_xor(v, 32);
        if (lane < 8) redw[(wave * 2 + r) * 64 + lane * 8 + e] = v;
      }
    block_sync();
    if (tid < 128) {
      const int r = tid >> 6, col = tid & 63;
      float v = 0.f;
#pragma unroll
      for (int w8 = 0; w8 < 8; w8++) v += redw[(w8 * 2 + r) * 64 + col];
      if (a.dDsp == 0) {
        v = wave_sum(v);
        if (col == 0) atomic_add_f32(a.dD + (int64_t)(h0 + r) * a.dDsh, v);
      } else {
        atomic_add_f32(a.dD + (int64_t)(h0 + r) * a.dDsh + (int64_t)col * a.dDsp, v);
      }
    }
  }
  if (a.fin) {
    const float extra = a.fin_extra_decay ? expf(a.dtp[((int64_t)b * a.H + hcur) * a.L] * Ah) : 1.f;
#pragma unroll
    for (int ut = 0; ut < 2; ut++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int k = 32 * wi + (r & 3) + 8 * (r >> 2) + 4 * h32, u = 64 * wj + 32 * ut + l31;
        a.fin[(int64_t)b * a.fsb + (int64_t)hcur * a.fsh + (int64_t)u * a.fsu + (int64_t)k * a.fsk] = accS[ut][r] * extra;
      }
  }
}

// =========================================================================================================
// class B, version 3: the row-strip design of ssd_mfma_a3_kernel for the dC / dB scans.  One 512-thread workgroup owns
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
  const int b = vid / pairs, hp = vid % pairs;
  const int h0 = hp * 2, hcur = h0 + hh;
  const int g = h0 / (a.H / a.G);
  const int nC = (a.L + QC - 1) / QC;
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
      if (a.init) {
        const int k = 16 * w + 4 * g16 + r, u = 16 * ut + t16;
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

  stlo = chunk_lo(0);
  prefetch_tiles();
  prefetch_q();
  commit();
  if (w == 0) scalars();
  publish_state();
  block_sync();
  float* part = a.part + ((int64_t)b * pairs + hp) * (int64_t)a.L * 128;
  // state checkpoints in fragment order: [b][pair][chunk][wave 8][ut 8][reg pair 2][lane 64] packed bf16 pairs (u32)
  uint32_t* ck = a.ckpt ? (uint32_t*)a.ckpt + ((int64_t)b * pairs + hp) * (int64_t)nC * 8192 : nullptr;
  float* tokscal = a.tokscal + ((int64_t)b * a.H + hcur) * a.L;
  float dDp[DMODE == 2 ? 2 : 1][DMODE == 2 ? 8 : 1];
#pragma unroll
  for (int i = 0; i < (DMODE == 2 ? 2 : 1); i++)
#pragma unroll
    for (int e = 0; e < (DMODE == 2 ? 8 : 1); e++) dDp[i][e] = 0.f;
  const bool want_bnd = MODE == GS_DB && ck != nullptr && a.bnd != nullptr;
  // checkpoint of the chunk the NEXT iteration closes, fetched while the output registers are dead
  uint32_t ckv[16];
  auto load_ckpt = [&](int c) {
    const uint32_t* cp = ck + (int64_t)(nC - 1 - c) * 8192 + wave * 1024;
#pragma unroll
    for (int i = 0; i < 16; i++) ckv[i] = cp[i * 64 + lane];
  };
  if (want_bnd) load_ckpt(0);

  for (int c = 0; c < nC; c++) {
    const int tlo = chunk_lo(c);
    const int cnext = c + 1 < nC ? c + 1 : c;
    OMK_OPAQUE(o_mu); OMK_OPAQUE(o_su); OMK_OPAQUE(o_x4); OMK_OPAQUE(o_ps); OMK_OPAQUE(o_tk);
    if (want_bnd) {
      // exact restart value of the decay-gradient prefix at the boundary behind chunk id = nC - 1 - c:
      //   bnd[id + 1] = exp(a_first(id+1)) * < g_first(id+1) (= accS now), h_last(id) (= dC-scan checkpoint of chunk id) >
      float dot = 0.f;
#pragma unroll
      for (int ut = 0; ut < 8; ut++)
#pragma unroll
        for (int j = 0; j < 2; j++) dot += accS[ut][2 * j] * bf_lo(ckv[2 * ut + j]) + accS[ut][2 * j + 1] * bf_hi(ckv[2 * ut + j]);
      dot = wave_sum(dot);
      if (lane == 0) sm.bred[wave] = dot;
    }
    // ---- (1) O^T = exp2(cs_l) * (S_in^T . Q^T)
    f32x4 acc[8];
#pragma unroll
    for (int ut = 0; ut < 8; ut++) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    if (want_bnd && tid < 2) {
      const int id = nC - 1 - c, hd = h0 + tid;
      const int tnext = (id + 1) * QC;
      const float ex = tnext < a.L ? expf(a.dtp[((int64_t)b * a.H + hd) * a.L + tnext] * a.A[hd]) : 1.f;
      a.bnd[((int64_t)b * a.H + hd) * (nC + 1) + id + 1] = ex * (sm.bred[4 * tid] + sm.bred[4 * tid + 1] + sm.bred[4 * tid + 2] + sm.bred[4 * tid + 3]);
    }
    stlo = chunk_lo(cnext);
    prefetch_tiles();
    // ---- (2) intra-chunk: G tiles -> M fragments (registers) -> U^T . M^T
    {
      const float cs_l = sm.cs[hh][16 * w + t16];
      auto block = [&](int kk, bool second, bool diag0, bool diag1) {
        u32x4 mh, ml;
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
                dDp[DMODE == 2 ? kq : 0][DMODE == 2 ? 2 * e2 : 0] += bf_lo(fa[e2]) * bf_lo(qf[kq][e2]);
                dDp[DMODE == 2 ? kq : 0][DMODE == 2 ? 2 * e2 + 1 : 0] += bf_hi(fa[e2]) * bf_hi(qf[kq][e2]);
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
            ml[2 * j + p2] = pack_bf16x2(v[2 * p2] - bf_lo(hi), v[2 * p2 + 1] - bf_hi(hi));
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
          acc[ut] = mfma16x16x32_bf16(fu, as_s16x8(ml), acc[ut]);
        }
      };
      if (w == 0) { block(0, false, true, false); }
      else if (w == 1) { block(0, true, false, true); }
      else if (w == 2) { block(0, true, false, false); block(1, false, true, false); }
      else { block(0, true, false, false); block(1, true, false, true); }
    }
    prefetch_q();
    // ---- (3) state update: S^T[p][n] = exp2(cs_end) S^T + sum_l (ws_l K[l][p]) U[l][n]
    {
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
    if (MODE == GS_DC && ck) {   // forward state at the END of this chunk, fragment order, bf16 pairs
      uint32_t* cp = ck + (int64_t)c * 8192 + wave * 1024;
#pragma unroll
      for (int ut = 0; ut < 8; ut++)
#pragma unroll
        for (int j = 0; j < 2; j++) cp[(2 * ut + j) * 64 + lane] = pack_bf16x2(accS[ut][2 * j], accS[ut][2 * j + 1]);
    }
    // ---- (4) token scalar of this head: X4_l . O_l (row l = 16 w + t16: 32 products per lane, 4 lanes per row)
    const int trow = tlo + rtk_q;
    {
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
    if (hh == 1) {
#pragma unroll
      for (int ut = 0; ut < 8; ut++) *reinterpret_cast<f32x4*>(&sm.O[((w * 8 + ut) * 64 + lane) * 4]) = acc[ut];
    }
    block_sync();   // E: exchange buffer complete; nobody reads this chunk's tiles / scalars any more
    if (hh == 0 && trow < a.L) {
      float* prow = part + (int64_t)trow * 128 + 4 * g16;
#pragma unroll
      for (int ut = 0; ut < 8; ut++) {
        const f32x4 o1 = *reinterpret_cast<const f32x4*>(&sm.O[((w * 8 + ut) * 64 + lane) * 4]);
        *reinterpret_cast<f32x4*>(prow + 16 * ut) = acc[ut] + o1;
      }
    }
    if (want_bnd) load_ckpt(cnext);
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
  if (a.fin) {
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

// out[b][t][g][n] = sum over the head pairs of group g of part[b][pair][t][n]
__global__ void ssd_reduce_partials_kernel(const float* part, void* out, int64_t osb, int64_t osl, int64_t osg, int out_dt,
                                           int B, int L, int G, int pairs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread = 4 consecutive n
  const int64_t total = (int64_t)B * L * G * 32;
  if (i >= total) return;
  const int n4 = (int)(i % 32) * 4, g = (int)((i / 32) % G), t = (int)((i / (32 * (int64_t)G)) % L), b = (int)(i / (32 * (int64_t)G * L));
  const int ppg = pairs / G;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < ppg; p++) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(part + (((int64_t)b * pairs + g * ppg + p) * L + t) * 128 + n4);
    acc += v;
  }
  const int64_t o = (int64_t)b * osb + (int64_t)t * osl + (int64_t)g * osg + n4;
#pragma unroll
  for (int e = 0; e < 4; e++) store_rt(out, o + e, out_dt, acc[e]);
}

int ssd_reduce_partials(const float* part, void* out, int64_t osb, int64_t osl, int64_t osg, int out_dt, int B, int L, int G, int H, omk_stream stream) {
  const int64_t total = (int64_t)B * L * G * 32;
  dim3 grid((unsigned)((total + 255) / 256)), block(256);
  OMK_LAUNCH(ssd_reduce_partials_kernel, grid, block, 0, stream, part, out, osb, osl, osg, out_dt, B, L, G, H / 2);
  return OMK_OK;
}

static bool src_ok16(const Src& s, bool need) {
  if (!s.p) return !need;
  return s.dt == OMK_BF16 && ((uintptr_t)s.p & 15) == 0 && s.sb % 8 == 0 && s.sl % 8 == 0 && s.sh % 8 == 0;
}

static int ssd_mfma_launch_b(const GScan& g, omk_stream stream, int dry) {
  if (g.DU != 128 || g.DK != 64 || g.H % 2 != 0 || (g.H / g.G) % 2 != 0 || (!g.part && !dry)) return OMK_EUNSUPPORTED;
  if (!src_ok16(g.U, true) || !src_ok16(g.K, true) || !src_ok16(g.Q, true) || !src_ok16(g.X4, false)) return OMK_EUNSUPPORTED;
  if (dry) return OMK_OK;
  dim3 grid((unsigned)(g.B * (g.H / 2))), block(512);
  const int64_t lim = (int64_t)1 << 24;   // the row-strip kernel keeps per-lane offsets in 32 bits
  if (g.X4.p && g.tokscal && g.U.sl < lim && g.K.sl < lim && g.Q.sl < lim && g.X4.sl < lim && g.K.sh < lim && !getenv("OMK_SSD_B_V1")) {
    const size_t smem3 = sizeof(SmemB3);
#define OMK_B3(MODE_, DM_) do { \
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_b3_kernel<MODE_, DM_>), smem3)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem3); \
      OMK_LAUNCH((ssd_mfma_b3_kernel<MODE_, DM_>), grid, block, smem3, stream, g); } while (0)
    if (g.mode == GS_DC) OMK_B3(GS_DC, 0);
    else if (!g.dD) OMK_B3(GS_DB, 0);
    else if (g.dDsp == 0) OMK_B3(GS_DB, 1);
    else OMK_B3(GS_DB, 2);
#undef OMK_B3
    return OMK_OK;
  }
  const size_t smem = sizeof(SmemB);
  if (g.mode == GS_DC) {
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_b_kernel<GS_DC>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
    OMK_LAUNCH((ssd_mfma_b_kernel<GS_DC>), grid, block, smem, stream, g);
  } else {
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_b_kernel<GS_DB>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
    OMK_LAUNCH((ssd_mfma_b_kernel<GS_DB>), grid, block, smem, stream, g);
  }
  return OMK_OK;
}

int ssd_mfma_launch(const GScan& g, omk_stream stream, int dry) {
  if (g.mode == GS_DC || g.mode == GS_DB) return ssd_mfma_launch_b(g, stream, dry);
  if (g.mode != GS_Y && g.mode != GS_DX) return OMK_EUNSUPPORTED;
  if (g.DU != 64 || g.DK != 128 || g.H % 2 != 0 || (g.H / g.G) % 2 != 0) return OMK_EUNSUPPORTED;
  if (!src_ok16(g.U, true) || !src_ok16(g.K, true) || !src_ok16(g.Q, true) || !src_ok16(g.Z, false)) return OMK_EUNSUPPORTED;
  if (g.out_dt != OMK_BF16 || ((uintptr_t)g.out & 15) || g.osb % 8 || g.osl % 8 || g.osh % 8) return OMK_EUNSUPPORTED;
  if (g.outx && ((uintptr_t)g.outx & 15)) return OMK_EUNSUPPORTED;
  if (g.mode == GS_DX && g.dD) return OMK_EUNSUPPORTED;   // dD comes from the dB scan
  if (dry) return OMK_OK;
  const int64_t lim = (int64_t)1 << 24;   // the row-strip kernel keeps per-lane offsets in 32 bits
  const bool small = g.K.sl < lim && g.Q.sl < lim && g.U.sl < lim && g.osl < lim && (!g.Z.p || g.Z.sl < lim);
  if (small && !getenv("OMK_SSD_A_V1") && !getenv("OMK_SSD_A_V2")) {   // default: row-strip kernel, one head per 256-thread workgroup
    dim3 grid3((unsigned)(g.B * g.H)), block3(256);
    const size_t smem3 = sizeof(SmemA3);
    if (g.mode == GS_Y && (g.Z.p || g.outx)) {
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, true>), smem3)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem3);
      OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, true>), grid3, block3, smem3, stream, g);
    } else if (g.mode == GS_Y) {
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_Y, false>), smem3)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem3);
      OMK_LAUNCH((ssd_mfma_a3_kernel<GS_Y, false>), grid3, block3, smem3, stream, g);
    } else {
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a3_kernel<GS_DX, false>), smem3)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem3);
      OMK_LAUNCH((ssd_mfma_a3_kernel<GS_DX, false>), grid3, block3, smem3, stream, g);
    }
    return OMK_OK;
  }
  if (!getenv("OMK_SSD_A_V1")) {   // version 2: 32x32 output tiles, G through LDS
    dim3 grid2((unsigned)(g.B * g.H)), block2(256);
    const size_t smem2 = sizeof(SmemA2);
    if (g.mode == GS_Y) {
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a2_kernel<GS_Y>), smem2)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem2);
      OMK_LAUNCH((ssd_mfma_a2_kernel<GS_Y>), grid2, block2, smem2, stream, g);
    } else {
      if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a2_kernel<GS_DX>), smem2)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem2);
      OMK_LAUNCH((ssd_mfma_a2_kernel<GS_DX>), grid2, block2, smem2, stream, g);
    }
    return OMK_OK;
  }
  dim3 grid((unsigned)(g.B * (g.H / 2))), block(512);
  const size_t smem = sizeof(SmemA);
  if (g.mode == GS_Y) {
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a_kernel<GS_Y>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
    OMK_LAUNCH((ssd_mfma_a_kernel<GS_Y>), grid, block, smem, stream, g);
  } else {
    if (OMK_SET_MAX_DYN_SMEM((ssd_mfma_a_kernel<GS_DX>), smem)) return fail(OMK_ELAUNCH, "ssd_mfma: cannot raise dynamic LDS to %zu", smem);
    OMK_LAUNCH((ssd_mfma_a_kernel<GS_DX>), grid, block, smem, stream, g);
  }
  return OMK_OK;
}

}  // namespace omk
