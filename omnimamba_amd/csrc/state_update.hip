// state_update.hip -- single-token SSM recurrence (decode):  s <- s*exp(dt*A) + dt*x (x) B ; y = s.C + D*x ; y *= silu(z)
//
// Pure bandwidth on the state (2 * B*H*P*N * s_state bytes per layer-step, 4.19 MB
// at B=1 fp32 for the 1.3B block).  A row of the state (one (b, h, p), N values) is spread over LPR lanes with
// VEC contiguous elements each (16-byte loads for fp32 VEC=4), a wave holds 64/LPR rows, y is reduced with wave
// shuffles; B_t / C_t rows are shared by every row of a head and stay in L1/L2.  In place, no allocation, no
// host scalars: safe to capture in a hipGraph (models/stage2/generation.py:372-434 replays this step 255 times).
#include "omk_common.h"

namespace omk {

struct SuArgs {
  void* state; const void* x; const void* dt; const void* A; const void* Bm; const void* Cm; const void* D; const void* z;
  const void* dtb; void* out;
  int64_t ssb, ssh, ssp, ssn, xsb, xsh, xsp, dsb, dsh, dsp, ash, asp, asn, bsb, bsg, bsn, csb, csg, csn;
  int64_t Dsh, Dsp, zsb, zsh, zsp, tsh, tsp, osb, osh, osp;
  int B, H, P, N, G, softplus, xdt, dtdt, adt, ddt, tbdt;
};

template <class TS, class TX, int VEC, int LPR>
__global__ __launch_bounds__(256) void state_update_kernel(SuArgs a) {
  constexpr int RPW = 64 / LPR;                       // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane % LPR, rw = lane / LPR;
  const int rows_per_block = RPW * (blockDim.x >> 6);
  const int pblocks = (a.P + rows_per_block - 1) / rows_per_block;
  const int pb = blockIdx.x % pblocks, h = (blockIdx.x / pblocks) % a.H, b = blockIdx.x / (pblocks * a.H);
  const int p = pb * rows_per_block + wave * RPW + rw;
  const bool live = p < a.P;
  const int pp = live ? p : a.P - 1;                 // keep every lane in the shuffles
  const int g = h / (a.H / a.G);
  // every independent load first: the scalars of the row and (vector path, N <= LPR * VEC: one step) its state / B / C
  // (requests first, conversions behind the last request: raw_rt_flat / cvt_rt_flat in omk_common.h; D and the gate of the row ride along
  // from clamped / dummy addresses instead of sitting under branches behind the reduction)
  const RawElem q_dt = raw_rt_flat(a.dt, (int64_t)b * a.dsb + (int64_t)h * a.dsh + (int64_t)pp * a.dsp, a.dtdt);
  const RawElem q_dtb = raw_rt_flat(a.dtb ? a.dtb : a.dt, a.dtb ? (int64_t)h * a.tsh + (int64_t)pp * a.tsp : 0, a.dtb ? a.tbdt : a.dtdt);
  const RawElem q_x = raw_rt_flat(a.x, (int64_t)b * a.xsb + (int64_t)h * a.xsh + (int64_t)pp * a.xsp, a.xdt);
  const bool tied = a.asn == 0;
  const RawElem q_A = raw_rt_flat(a.A, (int64_t)h * a.ash + (int64_t)pp * a.asp, a.adt);
  const RawElem q_D = raw_rt_flat(a.D ? a.D : a.A, a.D ? (int64_t)h * a.Dsh + (int64_t)pp * a.Dsp : 0, a.D ? a.ddt : a.adt);
  const RawElem q_z = raw_rt_flat(a.z ? a.z : a.x, a.z ? (int64_t)b * a.zsb + (int64_t)h * a.zsh + (int64_t)pp * a.zsp : 0, a.xdt);
  TS* s = (TS*)a.state + (int64_t)b * a.ssb + (int64_t)h * a.ssh + (int64_t)pp * a.ssp;
  const TX* Bp = (const TX*)a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg;
  const TX* Cp = (const TX*)a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg;
  float sv0[VEC], bv0[VEC], cv0[VEC];
  const bool one_step = VEC > 1 && a.N <= LPR * VEC;
  if constexpr (VEC > 1) {
    const int n0c = lr * VEC < a.N ? lr * VEC : 0;   // clamped: lanes past N re-read the row start and are not stored
    load_vec<TS, VEC>(s + n0c, sv0); load_vec<TX, VEC>(Bp + n0c, bv0); load_vec<TX, VEC>(Cp + n0c, cv0);
  }
  float dt = cvt_rt_flat(q_dt, a.dtdt);
  const float dtbv = cvt_rt_flat(q_dtb, a.dtb ? a.tbdt : a.dtdt), xv = cvt_rt_flat(q_x, a.xdt), Av = cvt_rt_flat(q_A, a.adt);
  const float Dv = cvt_rt_flat(q_D, a.D ? a.ddt : a.adt), zv = cvt_rt_flat(q_z, a.xdt);
  if (a.dtb) dt += dtbv;
  if (a.softplus) dt = softplus_f(dt);
  const float xdt = xv * dt;
  const float dA_t = tied ? expf(dt * Av) : 0.f;
  float acc = 0.f;
  for (int n0 = lr * VEC; n0 < a.N; n0 += LPR * VEC) {
    float sv[VEC], bv[VEC], cv[VEC];
    if constexpr (VEC > 1) {             // unit stride on n checked on the host
      if (n0 == lr * VEC) {
#pragma unroll
        for (int i = 0; i < VEC; i++) { sv[i] = sv0[i]; bv[i] = bv0[i]; cv[i] = cv0[i]; }
      } else {
        load_vec<TS, VEC>(s + n0, sv); load_vec<TX, VEC>(Bp + n0, bv); load_vec<TX, VEC>(Cp + n0, cv);
      }
    } else {
      bv[0] = to_f32(Bp[(int64_t)n0 * a.bsn]); cv[0] = to_f32(Cp[(int64_t)n0 * a.csn]); sv[0] = to_f32(s[(int64_t)n0 * a.ssn]);
    }
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      float dA = tied ? dA_t : expf(dt * load_rt(a.A, (int64_t)h * a.ash + (int64_t)pp * a.asp + (int64_t)(n0 + i) * a.asn, a.adt));
      sv[i] = sv[i] * dA + xdt * bv[i];
      acc += sv[i] * cv[i];
    }
    if (live) {
      if constexpr (VEC > 1) store_vec<TS, VEC>(s + n0, sv);
      else s[(int64_t)n0 * a.ssn] = from_f32<TS>(sv[0]);
    }
  }
  (void)one_step;
#pragma unroll
  for (int m = LPR / 2; m >= 1; m >>= 1) acc += shfl_xor(acc, m);
  if (live && lr == 0) {
    float y = acc;
    if (a.D) y += xv * Dv;
    if (a.z) y *= silu_f(zv);
    store_rt(a.out, (int64_t)b * a.osb + (int64_t)h * a.osh + (int64_t)p * a.osp, a.xdt, y);
  }
}

// Mamba-2's configuration (A, dt, dt_bias tied over (p, n); one vector step per row: N == 4 LPR; unit strides), RPT rows
// per lane group: the per-head scalars and B / C are loaded once, the RPT state vectors fly together.  With one row per
// lane group (kernel above) a batch-8 step is 16 k waves of ten dependent-ish loads each: 17.4 us for 33.5 MB.
template <class TS, class TX, int LPR, int RPT>
__global__ __launch_bounds__(256) void state_update_tied_kernel(SuArgs a) {
  constexpr int VEC = 4, RPW = 64 / LPR, RPB = RPW * 4;         // rows per workgroup and pass
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane % LPR, rw = lane / LPR;
  const int pblocks = a.P / (RPB * RPT);                       // host: P % (RPB * RPT) == 0
  const int pb = blockIdx.x % pblocks, h = (blockIdx.x / pblocks) % a.H, b = blockIdx.x / (pblocks * a.H);
  const int g = h / (a.H / a.G);
  const int p0 = pb * RPB * RPT + wave * RPW + rw, n0 = lr * VEC;
  TS* s = (TS*)a.state + (int64_t)b * a.ssb + (int64_t)h * a.ssh + n0;
  float sv[RPT][VEC], xv[RPT], bv[VEC], cv[VEC];
#pragma unroll
  for (int k = 0; k < RPT; k++) load_vec<TS, VEC>(s + (int64_t)(p0 + k * RPB) * a.ssp, sv[k]);
  load_vec<TX, VEC>((const TX*)a.Bm + (int64_t)b * a.bsb + (int64_t)g * a.bsg + n0, bv);
  load_vec<TX, VEC>((const TX*)a.Cm + (int64_t)b * a.csb + (int64_t)g * a.csg + n0, cv);
  // every other request of the thread right behind them, none under a branch and none followed by its conversion (raw_rt_flat /
  // cvt_rt_flat): x, gate and D of its four rows, the three tied scalars.  Before round 5 dt, dt_bias, A were three dependent round trips
  // in front of the arithmetic and D, gate two more per row behind the reduction: 19.9 us for 33.5 MB at eight fp32 sequences.
  TX xq[RPT], zq[RPT];
  RawElem Dq[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    const int64_t p = p0 + k * RPB;
    const int64_t xi = (int64_t)b * a.xsb + (int64_t)h * a.xsh + p * a.xsp;
    xq[k] = ((const TX*)a.x)[xi];
    zq[k] = ((const TX*)(a.z ? a.z : a.x))[a.z ? (int64_t)b * a.zsb + (int64_t)h * a.zsh + p * a.zsp : xi];   // (no gate: x once more, not used)
    Dq[k] = raw_rt_flat(a.D ? a.D : a.A, a.D ? (int64_t)h * a.Dsh + p * a.Dsp : 0, a.D ? a.ddt : a.adt);
  }
  const RawElem q_dt = raw_rt_flat(a.dt, (int64_t)b * a.dsb + (int64_t)h * a.dsh, a.dtdt);
  const RawElem q_dtb = raw_rt_flat(a.dtb ? a.dtb : a.dt, a.dtb ? (int64_t)h * a.tsh : 0, a.dtb ? a.tbdt : a.dtdt);
  const RawElem q_A = raw_rt_flat(a.A, (int64_t)h * a.ash, a.adt);
#pragma unroll
  for (int k = 0; k < RPT; k++) xv[k] = to_f32(xq[k]);
  float dt = cvt_rt_flat(q_dt, a.dtdt);
  const float dtbv = cvt_rt_flat(q_dtb, a.dtb ? a.tbdt : a.dtdt), Av = cvt_rt_flat(q_A, a.adt);
  if (a.dtb) dt += dtbv;
  if (a.softplus) dt = softplus_f(dt);
  const float dA = expf(dt * Av);
  float acc[RPT];
#pragma unroll
  for (int k = 0; k < RPT; k++) {
    acc[k] = 0.f;
    const float xdt = xv[k] * dt;
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      sv[k][i] = sv[k][i] * dA + xdt * bv[i];
      acc[k] += sv[k][i] * cv[i];
    }
    store_vec<TS, VEC>(s + (int64_t)(p0 + k * RPB) * a.ssp, sv[k]);
  }
#pragma unroll
  for (int k = 0; k < RPT; k++)
#pragma unroll
    for (int m = LPR / 2; m >= 1; m >>= 1) acc[k] += shfl_xor(acc[k], m);
  if (lr == 0) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int p = p0 + k * RPB;
      float y = acc[k];
      if (a.D) y += xv[k] * cvt_rt_flat(Dq[k], a.ddt);
      if (a.z) y *= silu_f(to_f32(zq[k]));
      ((TX*)a.out)[(int64_t)b * a.osb + (int64_t)h * a.osh + (int64_t)p * a.osp] = from_f32<TX>(y);
    }
  }
}

}  // namespace omk

using namespace omk;

extern "C" int omk_selective_state_update(const OmkStateUpdate* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->state) && present(p->x) && present(p->dt) && present(p->A) && present(p->Bm) && present(p->Cm) && present(p->out),
              "selective_state_update: state, x, dt, A, B, C, out required");
  OMK_REQUIRE(p->state.ndim == 4 && p->x.ndim == 3 && p->dt.ndim == 3 && p->A.ndim == 3 && p->Bm.ndim == 3 && p->Cm.ndim == 3 && p->out.ndim == 3,
              "selective_state_update: state (B,H,P,N), x/dt/out (B,H,P), A (H,P,N), B/C (B,G,N)");
  SuArgs a = {};
  a.B = (int)p->state.shape[0]; a.H = (int)p->state.shape[1]; a.P = (int)p->state.shape[2]; a.N = (int)p->state.shape[3]; a.G = (int)p->Bm.shape[1];
  OMK_REQUIRE(a.G > 0 && a.H % a.G == 0, "selective_state_update: H must be a multiple of ngroups");
  OMK_REQUIRE(p->x.shape[0] == a.B && p->x.shape[1] == a.H && p->x.shape[2] == a.P, "selective_state_update: x shape");
  OMK_REQUIRE(p->Bm.shape[0] == a.B && p->Bm.shape[2] == a.N && p->Cm.shape[1] == a.G && p->Cm.shape[2] == a.N, "selective_state_update: B/C shape");
  OMK_REQUIRE(p->Bm.dtype == p->x.dtype && p->Cm.dtype == p->x.dtype && p->out.dtype == p->x.dtype, "selective_state_update: B, C, out must have x's dtype");
  OMK_REQUIRE(!present(p->z) || p->z.dtype == p->x.dtype, "selective_state_update: z dtype");
  a.state = p->state.data; a.x = p->x.data; a.dt = p->dt.data; a.A = p->A.data; a.Bm = p->Bm.data; a.Cm = p->Cm.data;
  a.D = p->D.data; a.z = p->z.data; a.dtb = p->dt_bias.data; a.out = p->out.data;
  a.ssb = p->state.stride[0]; a.ssh = p->state.stride[1]; a.ssp = p->state.stride[2]; a.ssn = p->state.stride[3];
  a.xsb = p->x.stride[0]; a.xsh = p->x.stride[1]; a.xsp = p->x.stride[2];
  a.dsb = p->dt.stride[0]; a.dsh = p->dt.stride[1]; a.dsp = p->dt.stride[2];
  a.ash = p->A.stride[0]; a.asp = p->A.stride[1]; a.asn = p->A.stride[2];
  a.bsb = p->Bm.stride[0]; a.bsg = p->Bm.stride[1]; a.bsn = p->Bm.stride[2];
  a.csb = p->Cm.stride[0]; a.csg = p->Cm.stride[1]; a.csn = p->Cm.stride[2];
  if (present(p->D)) { a.Dsh = p->D.stride[0]; a.Dsp = p->D.ndim > 1 ? p->D.stride[1] : 0; }
  if (present(p->z)) { a.zsb = p->z.stride[0]; a.zsh = p->z.stride[1]; a.zsp = p->z.stride[2]; }
  if (present(p->dt_bias)) { a.tsh = p->dt_bias.stride[0]; a.tsp = p->dt_bias.ndim > 1 ? p->dt_bias.stride[1] : 0; }
  a.osb = p->out.stride[0]; a.osh = p->out.stride[1]; a.osp = p->out.stride[2];
  a.softplus = p->dt_softplus; a.xdt = p->x.dtype; a.dtdt = p->dt.dtype; a.adt = p->A.dtype; a.ddt = p->D.dtype; a.tbdt = p->dt_bias.dtype;
  if ((int64_t)a.B * a.H * a.P * a.N == 0) return OMK_OK;
  // vector path: N % 4 == 0, unit stride on n everywhere, 16-byte (fp32) / 8-byte (16-bit) aligned rows
  const int sb = (int)dtype_size(p->state.dtype), xb = (int)dtype_size(p->x.dtype);
  bool vec = a.N % 4 == 0 && a.ssn == 1 && a.bsn == 1 && a.csn == 1 &&
             ((uintptr_t)p->state.data % (4 * sb)) == 0 && ((uintptr_t)p->Bm.data % (4 * xb)) == 0 && ((uintptr_t)p->Cm.data % (4 * xb)) == 0 &&
             a.ssb % 4 == 0 && a.ssh % 4 == 0 && a.ssp % 4 == 0 && a.bsb % 4 == 0 && a.bsg % 4 == 0 && a.csb % 4 == 0 && a.csg % 4 == 0;
  dim3 block(256);
  // Mamba-2 decode with several sequences: tied scalars, one vector step per row, four rows per lane group
  {
    const int lpr = a.N / 4;
    const bool tied = vec && a.asn == 0 && a.asp == 0 && a.dsp == 0 && (!present(p->dt_bias) || a.tsp == 0) && (lpr == 32 || lpr == 16) && a.N % 4 == 0 &&
                      (!present(p->z) || p->z.dtype == p->x.dtype) && p->out.dtype == p->x.dtype &&   // (an ABSENT gate has dtype 0 = fp32: the bf16 decode at batch 8 fell to the row kernel)
                      (int64_t)a.B * a.H * a.P * a.N >= ((int64_t)1 << 21);
    const int rpb = tied ? (64 / lpr) * 4 : 1;
    if (tied && a.P % (rpb * 4) == 0 && !getenv("OMK_STATE_UPDATE_GENERIC")) {
      dim3 grid((unsigned)((int64_t)a.B * a.H * (a.P / (rpb * 4))));
#define SU_TIED(TS, TX) do { if (lpr == 32) OMK_LAUNCH((state_update_tied_kernel<TS, TX, 32, 4>), grid, block, 0, stream, a); \
                             else OMK_LAUNCH((state_update_tied_kernel<TS, TX, 16, 4>), grid, block, 0, stream, a); } while (0)
      OMK_DISPATCH_DTYPE(p->state.dtype, TS, OMK_DISPATCH_DTYPE(p->x.dtype, TX, SU_TIED(TS, TX)));
#undef SU_TIED
      return finish_launch("selective_state_update");
    }
  }
#define SU_LAUNCH(TS, TX, VEC, LPR) do { \
    int rpb = (64 / LPR) * 4; int pblocks = (a.P + rpb - 1) / rpb; \
    dim3 grid((unsigned)((int64_t)a.B * a.H * pblocks)); \
    OMK_LAUNCH((state_update_kernel<TS, TX, VEC, LPR>), grid, block, 0, stream, a); } while (0)
#define SU_SHAPE(TS, TX) do { \
    if (vec && a.N >= 128) SU_LAUNCH(TS, TX, 4, 32); \
    else if (vec && a.N >= 64) SU_LAUNCH(TS, TX, 4, 16); \
    else if (vec) SU_LAUNCH(TS, TX, 4, 4); \
    else SU_LAUNCH(TS, TX, 1, 16); } while (0)
  OMK_DISPATCH_DTYPE(p->state.dtype, TS, OMK_DISPATCH_DTYPE(p->x.dtype, TX, SU_SHAPE(TS, TX)));
#undef SU_SHAPE
#undef SU_LAUNCH
  return finish_launch("selective_state_update");
}
