// ssd_f32.hip -- the SSD forward scan for fp32 activations on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32).
//
// Who needs it: the reference runs its inference in fp32 (scripts/inference_t2i.py:21-26, scripts/inference_mmu.py: the model is
// never cast), so the prefill of a prompt -- 72 tokens for T2I, 4 + 729 image + question tokens for MMU -- is an fp32 scan of ONE
// sequence.  The shape-generic kernel (ssd.hip) walks the tokens one by one: 64 sequences on a quarter of the chip, 5.5 us per 16
// tokens.  This kernel is the chunked form of the same recurrence with every product on the matrix pipe and fp32 operands (no
// rounding of an operand anywhere: the accuracy class of the generic kernel):
//   * one 256-thread workgroup per (batch, head, 16 output columns): the state columns are independent, so a head of 64 columns
//     is four workgroups -- 256 of them for one sequence of the 1.3B model;
//   * chunks of 16 tokens; wave w owns the state rows k in [32 w, 32 w + 32) as two 16 x 16 accumulator tiles and contracts over
//     exactly those k everywhere:  G^T[s][l] = K_s . Q_l  and  O^T[u][l] = S_in^T Q_l  are formed as per-wave partial sums and added
//     through LDS; the accumulator tile of the state IS the A operand of S_in^T Q (register r of lane (g, u) holds k = 4 g + r:
//     the contraction order of step r), the decayed / masked G tile IS the B operand of the intra-chunk product, with x fetched in
//     the matching order -- the tricks of the bf16 kernel (ssd_mfma.hip), here without any conversion;
//   * K and x of a chunk are staged in LDS for the state update (A operand K^T, B operand ws_l x_l); the row operands of G and
//     S_in^T Q come straight from global memory in operand order (16 bytes per lane), one chunk ahead;
//   * two barriers per chunk (partial G visible; partial O and the next chunk's tiles visible).
// Per wave and chunk: 8 (G) + 8 (S^T Q) + 1 (M x) + 8 (state) MFMAs of 32 cycles.
#include "ssd_scan.h"

namespace omk {

constexpr int F_T = 16;      // tokens per chunk
constexpr int F_KS = 144;    // row stride (floats) of the staged K tile: the state update reads K^T -- lanes = 16 consecutive k of rows
                             // 4 st + g -- and 144 = 128 + 16 puts the rows g = 0, 1 (one ds_read_b32 lane group) on disjoint banks

struct SmemF32 {
  float K[2][F_T * F_KS];    // [l][k]
  float X[2][F_T * 16];      // [l][u]
  float Gp[4][256];          // partial G^T of wave w, accumulator order (lane * 4 + r)
  float Op[4][256];          // partial O^T of wave w
  float cs[2][F_T], ecs[2][F_T], wv[2][F_T], ws[2][F_T];
};

template <bool HASZ>
__global__ __launch_bounds__(256) void ssd_f32_mfma_kernel(GScan a) {
  __shared__ SmemF32 sm;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = uniform_i(tid >> 6);
  const int g4 = lane >> 4, i16 = lane & 15;
  const int nuq = a.DU / 16;
  const int uq = blockIdx.x % nuq, h = (blockIdx.x / nuq) % a.H, b = blockIdx.x / (nuq * a.H);
  const int g = h / (a.H / a.G);
  const int nC = (a.L + F_T - 1) / F_T;
  const float* Kb = (const float*)a.K.p + (int64_t)b * a.K.sb + (int64_t)g * a.K.sh;
  const float* Qb = (const float*)a.Q.p + (int64_t)b * a.Q.sb + (int64_t)g * a.Q.sh;
  const float* Xb = (const float*)a.U.p + (int64_t)b * a.U.sb + (int64_t)h * a.U.sh + uq * 16;
  const float* Zb = HASZ ? (const float*)a.Z.p + (int64_t)b * a.Z.sb + (int64_t)h * a.Z.sh + uq * 16 : nullptr;
  const float* dtrow = a.dtp + ((int64_t)b * a.H + h) * a.L;
  const float Ah2 = a.A[h] * LOG2E;
  auto rowc = [&](int t) -> int { return t < a.L ? t : a.L - 1; };   // rows past the end re-read the last row (finite; their weight is 0)

  // ---- running state S[k][u], k in [32 w, 32 w + 32): tiles jj = 0, 1 (rows 16 (2 w + jj) + 4 g4 + r, column i16)
  f32x4 accS[2];
#pragma unroll
  for (int jj = 0; jj < 2; jj++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int k = 32 * w + 16 * jj + 4 * g4 + r, u = uq * 16 + i16;
      accS[jj][r] = a.init ? load_rt(a.init, (int64_t)b * a.isb + (int64_t)h * a.ish + (int64_t)u * a.isu + (int64_t)k * a.isk, a.init_dt) : 0.f;
    }

  // ---- staging registers of the NEXT chunk: operand rows of K / Q (row i16, the 2 x 4 k of lane group g4 in this wave's block),
  // the thread's two 16-byte pieces of the K tile, its x (and token scalar) element
  f32x4 kf[2], qf[2], kt[2];
  float xr = 0.f, dtr = 0.f;
  const int srow = tid >> 5, sc4 = tid & 31;          // K tile: rows srow and srow + 8, columns 4 sc4 ..
  const int xl = tid >> 4, xu = tid & 15;             // x / output element of this thread: token xl, column xu
  auto fetch = [&](int c) {
    const int t0 = c * F_T;
    const int tr = rowc(t0 + i16);
#pragma unroll
    for (int jj = 0; jj < 2; jj++) {
      kf[jj] = *reinterpret_cast<const f32x4*>(Kb + (int64_t)tr * a.K.sl + 32 * w + 16 * jj + 4 * g4);
      qf[jj] = *reinterpret_cast<const f32x4*>(Qb + (int64_t)tr * a.Q.sl + 32 * w + 16 * jj + 4 * g4);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) kt[j] = *reinterpret_cast<const f32x4*>(Kb + (int64_t)rowc(t0 + srow + 8 * j) * a.K.sl + 4 * sc4);
    xr = Xb[(int64_t)rowc(t0 + xl) * a.U.sl + xu];
    if (tid < F_T) dtr = t0 + tid < a.L ? dtrow[t0 + tid] : 0.f;
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 2; j++) *reinterpret_cast<f32x4*>(&sm.K[buf][(srow + 8 * j) * F_KS + 4 * sc4]) = kt[j];
    sm.X[buf][xl * 16 + xu] = xr;
    if (w == 0) {   // token scalars: lanes 0 .. 15 = tokens (the others carry zeros through the scan)
      const float av = lane < F_T ? dtr * Ah2 : 0.f;
      const float cs = wave_incl_scan_add(av);
      const float cend = wave_read_lane(cs, F_T - 1);
      if (lane < F_T) {
        const float wt = a.w_is_dt ? dtr : 1.f;
        sm.cs[buf][lane] = cs; sm.ecs[buf][lane] = exp2_fast(cs); sm.wv[buf][lane] = wt; sm.ws[buf][lane] = wt * exp2_fast(cend - cs);
      }
    }
  };
  fetch(0);
  f32x4 kfc[2] = {kf[0], kf[1]}, qfc[2] = {qf[0], qf[1]};   // operand rows of the CURRENT chunk
  commit(0);
  const float Dv = a.D ? load_rt(a.D, (int64_t)h * a.Dsh + (int64_t)(uq * 16 + xu) * a.Dsp, a.D_dt) : 0.f;
  float xcur = xr;
  block_sync();

  for (int c = 0; c < nC; c++) {
    const int cur = c & 1, nxt = cur ^ 1;
    const int t0 = c * F_T;
    const bool more = c + 1 < nC;
    if (more) fetch(c + 1);
    // ---- partial G^T[s][l] (rows s = 4 g4 + r of column l = i16) and partial O^T[u][l] = S_in^T Q_l over this wave's 32 k
    f32x4 gp = {0.f, 0.f, 0.f, 0.f}, op = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        gp = mfma16x16x4_f32(kfc[jj][r], qfc[jj][r], gp);
        op = mfma16x16x4_f32(accS[jj][r], qfc[jj][r], op);
      }
    *reinterpret_cast<f32x4*>(&sm.Gp[w][4 * lane]) = gp;
    block_sync();   // A: the four partial G tiles are visible
    {
      f32x4 G = *reinterpret_cast<const f32x4*>(&sm.Gp[0][4 * lane]);
#pragma unroll
      for (int k = 1; k < 4; k++) G = G + *reinterpret_cast<const f32x4*>(&sm.Gp[k][4 * lane]);
      // M^T[s][l] = G^T[s][l] w_s 2^(cs_l - cs_s) for s <= l: this wave contracts the four s = 4 g4 + w (step w of the product)
      const float csl = sm.cs[cur][i16];
      const int s = 4 * g4 + w;
      const float Gs = w == 0 ? G[0] : (w == 1 ? G[1] : (w == 2 ? G[2] : G[3]));
      const float arg = csl - sm.cs[cur][s];
      const float m = s <= i16 ? Gs * sm.wv[cur][s] * exp2_fast(arg < 0.f ? arg : 0.f) : 0.f;
      const float e1 = sm.ecs[cur][i16];
      op = op * e1;
      op = mfma16x16x4_f32(sm.X[cur][s * 16 + i16], m, op);   // O^T[u][l] += x[s][u] M^T[s][l]
      *reinterpret_cast<f32x4*>(&sm.Op[w][4 * lane]) = op;
    }
    // ---- state update of this wave's rows: S[k][u] = 2^(cs_end) S + sum_l K[l][k] (ws_l x[l][u])
    {
      const float dec = sm.ecs[cur][F_T - 1];
      accS[0] = accS[0] * dec; accS[1] = accS[1] * dec;
#pragma unroll
      for (int st = 0; st < 4; st++) {
        const int l = 4 * st + g4;
        const float bx = sm.ws[cur][l] * sm.X[cur][l * 16 + i16];
#pragma unroll
        for (int jj = 0; jj < 2; jj++) accS[jj] = mfma16x16x4_f32(sm.K[cur][l * F_KS + 32 * w + 16 * jj + i16], bx, accS[jj]);
      }
    }
    const float xthis = xcur;
    if (more) {
      commit(nxt);
      kfc[0] = kf[0]; kfc[1] = kf[1]; qfc[0] = qf[0]; qfc[1] = qf[1];
      xcur = xr;
    }
    block_sync();   // B: partial O tiles and the next chunk's tiles are visible
    // ---- epilogue: thread (token xl, column xu) adds the four partials (accumulator order: lane (xu / 4) * 16 + xl, register xu % 4)
    {
      const int e = (((xu >> 2) * 16 + xl) << 2) + (xu & 3);
      float y = (sm.Op[0][e] + sm.Op[1][e]) + (sm.Op[2][e] + sm.Op[3][e]);
      y += Dv * xthis;
      const int t = t0 + xl;
      if (t < a.L) {
        const int64_t o = (int64_t)b * a.osb + (int64_t)t * a.osl + (int64_t)h * a.osh + uq * 16 + xu;
        if (HASZ) {
          if (a.outx) ((float*)a.outx)[o] = y;
          y *= silu_f(Zb[(int64_t)t * a.Z.sl + xu]);
        }
        ((float*)a.out)[o] = y;
      }
    }
  }
  if (a.fin) {
#pragma unroll
    for (int jj = 0; jj < 2; jj++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int k = 32 * w + 16 * jj + 4 * g4 + r, u = uq * 16 + i16;
        a.fin[(int64_t)b * a.fsb + (int64_t)h * a.fsh + (int64_t)u * a.fsu + (int64_t)k * a.fsk] = accS[jj][r];
      }
  }
}

// applies: the forward scan (y), every source fp32 with 16-byte aligned rows, d_state 128, headdim a multiple of 16, fp32 output
int ssd_f32_mfma_launch(const GScan& g, omk_stream stream) {
  if (g.mode != GS_Y || g.reverse || g.DK != 128 || g.DU % 16 != 0 || g.out_dt != OMK_F32 || g.fin_extra_decay) return OMK_EUNSUPPORTED;
  if (g.U.dt != OMK_F32 || g.K.dt != OMK_F32 || g.Q.dt != OMK_F32 || (g.Z.p && g.Z.dt != OMK_F32)) return OMK_EUNSUPPORTED;
  if (((uintptr_t)g.K.p & 15) || ((uintptr_t)g.Q.p & 15) || g.K.sb % 4 || g.K.sl % 4 || g.K.sh % 4 || g.Q.sb % 4 || g.Q.sl % 4 || g.Q.sh % 4) return OMK_EUNSUPPORTED;
  if (g.L < 1 || (g.H / g.G) < 1) return OMK_EUNSUPPORTED;
  if (const char* e = getenv("OMK_SSD_F32_MFMA")) if (e[0] == '0') return OMK_EUNSUPPORTED;
  kernels_note("ssd_f32_mfma");
  dim3 grid((unsigned)((int64_t)g.B * g.H * (g.DU / 16))), block(256);
  if (g.Z.p) OMK_LAUNCH((ssd_f32_mfma_kernel<true>), grid, block, 0, stream, g);
  else OMK_LAUNCH((ssd_f32_mfma_kernel<false>), grid, block, 0, stream, g);
  return OMK_OK;
}

}  // namespace omk
