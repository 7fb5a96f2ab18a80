// lora_add.hip -- out += scale * h B^T, in place, for the task LoRA of the in_proj (reference models/stage2/lora.py:263-279:
// result += lora_B(lora_A(dropout(x))) * scaling) at training / prefill token counts.
//
// The rank is 8: as a GEMM this is a K = 8 problem that the library runs as a full pass over the (tokens, out_features)
// result PLUS, in PyTorch's out-of-place addmm, a copy of that result in front of it -- +318 us on top of the 560 us base
// GEMM for 16 k tokens x 8512 features (tools/probe_addmm.py).  It is pure streaming: read the result once, write it once
// (558 MB -> ~110 us), eight FMAs per element with the eight B columns of a lane's features held in registers.
//   thread = 8 consecutive output features (16 bytes) x a strip of tokens; h rows are broadcast loads.
#include "omk_common.h"
#include "ssd_tiles.h"

namespace omk {

constexpr int LA_MAXR = 16;
struct LoraAddArgs {
  void* out; const void* h; const void* B; const uint8_t* mask; int64_t os, hs, bs, ms; int T, N, R, tokens_per_block, bdt; float scale; int bvec;
};

// MASK: one byte per element of `out`; elements whose byte is zero keep their value
template <class TO, int RR, bool MASK = false>   // RR: rank (8 or 16), a template parameter so that B's columns cost RR registers per feature
__global__ __launch_bounds__(256) void lora_add_kernel(LoraAddArgs a) {
  constexpr int VEC = 16 / sizeof(TO) > 8 ? 8 : 16 / sizeof(TO);   // 8 features (16-bit) or 4 (fp32) per thread
  const int nvec = a.N / VEC;
  const int cvb = (nvec + 255) / 256;
  const int cb = blockIdx.x % cvb, tb = blockIdx.x / cvb;
  const int cv = cb * 256 + threadIdx.x;
  if (cv >= nvec) return;
  const int n0 = cv * VEC;
  float bw[VEC][RR];
  if (a.bvec) {   // (uniform) contiguous, 16-byte aligned B rows: the lane's VEC x RR weights are consecutive -- 16-byte requests instead of VEC x RR
                  // run-time-dtype element loads (a workgroup lives for 64 tokens: the element prologue was as many requests as its token walk)
#pragma unroll
    for (int i = 0; i < VEC; i++) {
      if (a.bdt == OMK_F32) {
#pragma unroll
        for (int q = 0; q < RR / 4; q++) {
          const f32x4 v = reinterpret_cast<const f32x4*>((const float*)a.B + (int64_t)(n0 + i) * RR)[q];
#pragma unroll
          for (int e = 0; e < 4; e++) bw[i][4 * q + e] = a.scale * v[e];
        }
      } else {
#pragma unroll
        for (int q = 0; q < RR / 8; q++) {
          float tmp[8];
          if (a.bdt == OMK_BF16) load_vec<bf16_t, 8>((const bf16_t*)a.B + (int64_t)(n0 + i) * RR + 8 * q, tmp);
          else load_vec<f16_t, 8>((const f16_t*)a.B + (int64_t)(n0 + i) * RR + 8 * q, tmp);
#pragma unroll
          for (int e = 0; e < 8; e++) bw[i][8 * q + e] = a.scale * tmp[e];
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < VEC; i++)
#pragma unroll
      for (int r = 0; r < RR; r++) bw[i][r] = a.scale * load_rt(a.B, (int64_t)(n0 + i) * a.bs + r, a.bdt);
  }
  const int t0 = tb * a.tokens_per_block, t1 = t0 + a.tokens_per_block < a.T ? t0 + a.tokens_per_block : a.T;
  TO* out = (TO*)a.out + n0;
  const TO* h = (const TO*)a.h;
#ifndef OMK_LORA_ADD_UN
#define OMK_LORA_ADD_UN 8
#endif
  constexpr int UN = RR == 8 ? OMK_LORA_ADD_UN : 4;         // tokens in flight per thread
  for (int t = t0; t < t1; t += UN) {
    float o[UN][VEC], hv[UN][RR];
    uint8_t mk[UN][VEC];
#pragma unroll
    for (int u = 0; u < UN; u++) {
      const int tc = t + u < t1 ? t + u : t1 - 1;        // clamped: unconditional loads
      load_vec<TO, VEC>(out + (int64_t)tc * a.os, o[u]);
      if (MASK) memcpy(mk[u], a.mask + (int64_t)tc * a.ms + n0, VEC);
#pragma unroll
      for (int r = 0; r < RR; r += 8) {                    // rank 8: one 16-byte (16-bit) broadcast load per token
        float tmp[8];
        load_vec<TO, 8>(h + (int64_t)tc * a.hs + r, tmp);
#pragma unroll
        for (int q = 0; q < 8; q++) hv[u][r + q] = tmp[q];
      }
    }
#pragma unroll
    for (int u = 0; u < UN; u++) {
#pragma unroll
      for (int i = 0; i < VEC; i++) {
        float acc = MASK ? 0.f : o[u][i];
#pragma unroll
        for (int r = 0; r < RR; r++) acc += hv[u][r] * bw[i][r];
        o[u][i] = MASK ? (mk[u][i] ? o[u][i] + acc : o[u][i]) : acc;
      }
      if (t + u < t1) store_vec<TO, VEC>(out + (int64_t)(t + u) * a.os, o[u]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Backward of the up-projection: dh += dy B and dB += dy^T h from ONE read of dy.
// Workgroup = 256 columns of dy x a chunk of tokens, walked in tiles of 64 tokens through LDS (rows padded by 16 bytes).
//   dh   wave w, rows 16 w ..: D[token][r] = sum_n dy[token][n] B[n][r]: A = row reads of the tile, B = lora_b fragments kept in
//        registers for the whole workgroup (its columns are fixed) -> one partial row per column block and token
//   dB   wave w, columns 64 w ..: D[n][r] = sum_token dy[token][n] h[token][r]: A = the tile read TRANSPOSED (ds_read_b64_tr_b16),
//        B = h^T from LDS; four 16 x 16 accumulators live across all tiles -> one partial (N, r) slab per token chunk
// The partials are plain stores (every workgroup owns its slots) and the caller adds them up: deterministic, and cheaper than the
// 4.5 M fp32 atomics of the first version (16 of its 89 us).
// Both MFMAs are 16x16x32 with the rank in the N dimension (8 of 16 columns used): the kernel is a pure stream over dy.
// 16 k tokens x 8512: 84 us (round 6: 60 us, see lora_up_plan) + 9 us for the two reductions of the partials, against 112 - 128 us for the two library GEMMs; the
// load / stage / barrier skeleton alone is 68 us (4.1 TB/s), both MFMA parts together 3 us (OMK_LORA_UP_DBG).
// ---------------------------------------------------------------------------------------------------------
constexpr int LU_TT = 64, LU_TN = 256, LU_LD = LU_TN + 8;
struct LoraUpBwdArgs {
  const uint16_t* dy; int64_t dys; const void* B; int64_t bs; int bdt; const uint16_t* h; int64_t hs; float* dh; float* dB;
  int T, N, R, tchunk, NB, dbg;   // dbg: developer ablation (OMK_LORA_UP_DBG): 1 no dh atomics, 2 no dB MFMAs, 4 no dh MFMAs
};

__global__ __launch_bounds__(256) void lora_up_bwd_kernel(LoraUpBwdArgs a) {
  __shared__ __attribute__((aligned(16))) uint16_t sY[LU_TT * LU_LD];
  __shared__ __attribute__((aligned(16))) uint16_t sH[16 * LU_TT];        // h^T [r][token], rows >= R stay zero
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g16 = lane >> 4, t16 = lane & 15;
  const int nchunk = gridDim.x / a.NB;
  const int tc = blockIdx.x % nchunk, nb = blockIdx.x / nchunk;   // (which of the two runs fastest makes no difference: measured)
  const int n0 = nb * LU_TN, tb = tc * a.tchunk, te = tb + a.tchunk < a.T ? tb + a.tchunk : a.T;
  // lora_b fragments of the eight 32-column steps: B[k = n0 + 32 ks + 8 g16 + j][r = t16]
  // (round 6: the workgroup's 256 x R block of B goes through LDS -- a row per thread, 16-byte requests when the rows are contiguous fp32 -- instead
  // of 64 run-time-dtype element requests per lane: the prologue decided how short a workgroup may live, profiles/r06_stream_kernels.txt)
  s16x8 bfr[LU_TN / 32];
  {
    float* sB = reinterpret_cast<float*>(sY);                 // [256][16] fp32, before the first tile lands in sY
    const int n = n0 + tid;
    float row[16];
#pragma unroll
    for (int r = 0; r < 16; r++) row[r] = 0.f;
    if (n < a.N) {
      if (a.bdt == OMK_F32 && a.bs == 8 && a.R == 8 && (((uintptr_t)a.B) & 15) == 0) {
        const f32x4 v0 = reinterpret_cast<const f32x4*>((const float*)a.B + (int64_t)n * 8)[0], v1 = reinterpret_cast<const f32x4*>((const float*)a.B + (int64_t)n * 8)[1];
#pragma unroll
        for (int e = 0; e < 4; e++) { row[e] = v0[e]; row[4 + e] = v1[e]; }
      } else {
        for (int r = 0; r < a.R; r++) row[r] = load_rt(a.B, (int64_t)n * a.bs + r, a.bdt);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) reinterpret_cast<f32x4*>(sB + tid * 16)[q] = f32x4{row[4 * q], row[4 * q + 1], row[4 * q + 2], row[4 * q + 3]};
    block_sync();
#pragma unroll
    for (int ks = 0; ks < LU_TN / 32; ks++)
#pragma unroll
      for (int j = 0; j < 8; j++) bfr[ks][j] = (short)f32_to_bf16(sB[(32 * ks + 8 * g16 + j) * 16 + t16]);
    block_sync();
  }
  for (int i = tid; i < 16 * LU_TT; i += 256) sH[i] = 0;
  f32x4 accB[4];
#pragma unroll
  for (int ct = 0; ct < 4; ct++) accB[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 stage[8];
  u32x4 hreg = {0u, 0u, 0u, 0u};
  auto prefetch = [&](int t0) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int idx = tid + 256 * q, row = idx >> 5, pc = idx & 31;
      const int t = t0 + row, n = n0 + 8 * pc;
      const bool ok = t < te && n + 8 <= a.N;                  // N is a multiple of 8 (checked by the host)
      const u32x4 v = ld16(a.dy + (int64_t)(ok ? t : tb) * a.dys + (ok ? n : n0));
      stage[q] = ok ? v : u32x4{0u, 0u, 0u, 0u};
    }
    if (tid < LU_TT) {
      const int t = t0 + tid;
      const u32x4 v = ld16(a.h + (int64_t)(t < te ? t : tb) * a.hs);
      hreg = t < te ? v : u32x4{0u, 0u, 0u, 0u};
    }
  };
  prefetch(tb);
  for (int t0 = tb; t0 < te; t0 += LU_TT) {
    block_sync();                                               // the previous tile is consumed
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const int idx = tid + 256 * q, row = idx >> 5, pc = idx & 31;
      st16(&sY[row * LU_LD + 8 * pc], stage[q]);
    }
    if (tid < LU_TT) {
#pragma unroll
      for (int r = 0; r < 8; r++) sH[r * LU_TT + tid] = (uint16_t)((r & 1) ? (hreg[r >> 1] >> 16) : (hreg[r >> 1] & 0xffffu));
    }
    block_sync();
    prefetch(t0 + LU_TT);                                       // past the end: clamped loads, zeros
    // ---- dh: rows 16 w .. 16 w + 15 of the tile
    f32x4 acch = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < ((a.dbg & 4) ? 0 : LU_TN / 32); ks++) {
      const s16x8 fa = as_s16x8(ld16(&sY[(16 * w + t16) * LU_LD + 32 * ks + 8 * g16]));
      acch = mfma16x16x32_bf16(fa, bfr[ks], acch);              // D[token 4 g16 + i][r = t16]
    }
    if (t16 < a.R) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int t = t0 + 16 * w + 4 * g16 + i;
        if (t < te && !(a.dbg & 1)) a.dh[((int64_t)nb * a.T + t) * a.R + t16] = acch[i];
      }
    }
    // ---- dB: columns 64 w .. 64 w + 63, contraction over the 64 tokens of the tile
#pragma unroll
    for (int ks = 0; ks < ((a.dbg & 2) ? 0 : LU_TT / 32); ks++) {
      const s16x8 fh = as_s16x8(ld16(&sH[t16 * LU_TT + 32 * ks + 8 * g16]));      // B[k = token 8 g16 + j][r = t16]
#pragma unroll
      for (int ct = 0; ct < 4; ct++) {
        const uint16_t* pa = &sY[(32 * ks + 8 * g16 + (t16 >> 2)) * LU_LD + 64 * w + 16 * ct + 4 * (t16 & 3)];
        const s16x4 a0 = lds_read_tr16_b64(pa), a1 = lds_read_tr16_b64(pa + 4 * LU_LD);
        s16x8 fa;
        fa[0] = a0[0]; fa[1] = a0[1]; fa[2] = a0[2]; fa[3] = a0[3]; fa[4] = a1[0]; fa[5] = a1[1]; fa[6] = a1[2]; fa[7] = a1[3];
        accB[ct] = mfma16x16x32_bf16(fa, fh, accB[ct]);          // D[column 4 g16 + i][r = t16]
      }
    }
  }
  if (t16 < a.R) {
#pragma unroll
    for (int ct = 0; ct < 4; ct++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int n = n0 + 64 * w + 16 * ct + 4 * g16 + i;
        if (n < a.N) a.dB[((int64_t)tc * a.N + n) * a.R + t16] = accB[ct][i];
      }
  }
}

}  // namespace omk

using namespace omk;

extern "C" int omk_lora_add(const OmkLoraAdd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->out) && present(p->h) && present(p->lora_b), "lora_add: out, h, lora_b required");
  OMK_REQUIRE(p->out.ndim == 2 && p->h.ndim == 2 && p->lora_b.ndim == 2, "lora_add: out (T, N), h (T, r), lora_b (N, r)");
  LoraAddArgs a = {};
  a.T = (int)p->out.shape[0]; a.N = (int)p->out.shape[1]; a.R = (int)p->h.shape[1];
  OMK_REQUIRE(p->h.shape[0] == a.T && p->lora_b.shape[0] == a.N && p->lora_b.shape[1] == a.R, "lora_add: shape mismatch");
  OMK_REQUIRE(p->h.dtype == p->out.dtype, "lora_add: h must have out's dtype");
  if (a.T == 0 || a.N == 0 || a.R == 0) return OMK_OK;
  const int esz = (int)dtype_size(p->out.dtype), vec = esz == 4 ? 4 : 8;
  // rank padded to 8 in h (16-byte rows), features in 16-byte groups: the reference's r = 8 on 8512 features
  if ((a.R != 8 && a.R != 16) || a.N % vec != 0 || p->out.stride[1] != 1 || p->h.stride[1] != 1 || p->lora_b.stride[1] != 1 ||
      !aligned16(p->out) || !aligned16(p->h) || (p->out.stride[0] * esz) % 16 != 0 || (p->h.stride[0] * esz) % 16 != 0)
    return fail(OMK_EUNSUPPORTED, "lora_add: rank must be 8 or 16 and rows 16-byte aligned (use addmm otherwise)");
  a.out = p->out.data; a.h = p->h.data; a.B = p->lora_b.data; a.os = p->out.stride[0]; a.hs = p->h.stride[0]; a.bs = p->lora_b.stride[0];
  a.bdt = p->lora_b.dtype; a.scale = p->scale;
  a.bvec = (a.bs == a.R && ((uintptr_t)p->lora_b.data & 15) == 0 && !getenv("OMK_LORA_ADD_NOVEC")) ? 1 : 0;
  if (present(p->mask)) {
    OMK_REQUIRE(p->mask.dtype == OMK_U8 && p->mask.ndim == 2 && p->mask.shape[0] == a.T && p->mask.shape[1] == a.N && p->mask.stride[1] == 1,
                "lora_add: mask must be u8 (T, N) with contiguous rows");
    a.mask = (const uint8_t*)p->mask.data; a.ms = p->mask.stride[0];
  }
  const int nvec = a.N / vec, cvb = (nvec + 255) / 256;
  // few tokens (a 72-token prefill: 18 workgroups of 64 tokens took 19 us): cut the token blocks until the chip has work
  a.tokens_per_block = 32;   // (round 6, behind the 16-byte weight prologue: 128 / 64 / 32 / 16 / 8 tokens = 120 / 112 / 109.5 / 119 / 144 us at 16384 x 8512)
  if (const char* e = getenv("OMK_LORA_ADD_TPB")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32 || v == 64 || v == 128) a.tokens_per_block = v; }   // developer A/B
  while (a.tokens_per_block > 8 && (int64_t)cvb * ((a.T + a.tokens_per_block - 1) / a.tokens_per_block) < 256) a.tokens_per_block >>= 1;
  const int tbs = (a.T + a.tokens_per_block - 1) / a.tokens_per_block;
  dim3 grid((unsigned)((int64_t)cvb * tbs)), block(256);
  if (a.mask) {
    if (a.R == 8) OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 8, true>), grid, block, 0, stream, a));
    else OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 16, true>), grid, block, 0, stream, a));
  } else if (a.R == 8) OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 8>), grid, block, 0, stream, a));
  else OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 16>), grid, block, 0, stream, a));
  return finish_launch("lora_add");
}


static void lora_up_plan(int64_t T, int64_t N, int* NB, int* tchunk, int* chunks) {
  *NB = (int)((N + LU_TN - 1) / LU_TN);
  // token chunks: ONE resident round -- 188 registers = two workgroups per CU, so as many workgroups as fit 512 slots and no more (round 6: 748
  // workgroups were 1.46 rounds, the second one half empty: 81.5 us; 510: 63.3, with the B block through LDS 59.8; 1020 / 1530 / 2040: 63 / 69 / 75 --
  // every further workgroup pays a prologue and a partial dB slab), at least four tiles each
  int target = 512;
  if (const char* e = getenv("OMK_LORA_UP_WGS")) { const int v = atoi(e); if (v >= 64 && v <= 16384) target = v; }   // developer A/B
  int c = target / *NB;
  const int max_chunks = (int)((T + 4 * LU_TT - 1) / (4 * LU_TT));
  if (c > max_chunks) c = max_chunks;
  if (c < 1) c = 1;
  *tchunk = (int)(((T + c - 1) / c + LU_TT - 1) / LU_TT * LU_TT);
  *chunks = (int)((T + *tchunk - 1) / *tchunk);
}

extern "C" int omk_lora_up_bwd_parts(int64_t T, int64_t N, int32_t* parts) {
  OMK_REQUIRE(parts && T >= 0 && N >= 0, "lora_up_bwd_parts: bad arguments");
  int NB = 0, tchunk = 0, chunks = 0;
  if (T > 0 && N > 0) lora_up_plan(T, N, &NB, &tchunk, &chunks);
  parts[0] = NB; parts[1] = chunks;
  return OMK_OK;
}

extern "C" int omk_lora_up_bwd(const OmkLoraUpBwd* p, omk_stream stream) {
  OMK_REQUIRE(p && present(p->dy) && present(p->lora_b) && present(p->h) && present(p->dh) && present(p->dlora_b), "lora_up_bwd: dy, lora_b, h, dh, dlora_b required");
  OMK_REQUIRE(p->dy.ndim == 2 && p->lora_b.ndim == 2 && p->h.ndim == 2 && p->dh.ndim == 3 && p->dlora_b.ndim == 3, "lora_up_bwd: dy, lora_b, h 2-d; dh, dlora_b 3-d partials");
  LoraUpBwdArgs a = {};
  a.T = (int)p->dy.shape[0]; a.N = (int)p->dy.shape[1]; a.R = (int)p->lora_b.shape[1];
  if (a.T == 0 || a.N == 0) return OMK_OK;
  int chunks = 0;
  lora_up_plan(a.T, a.N, &a.NB, &a.tchunk, &chunks);
  OMK_REQUIRE(p->lora_b.shape[0] == a.N && p->h.shape[0] == a.T && p->h.shape[1] == a.R, "lora_up_bwd: shape mismatch");
  OMK_REQUIRE(p->dh.shape[0] == a.NB && p->dh.shape[1] == a.T && p->dh.shape[2] == a.R && p->dlora_b.shape[0] == chunks && p->dlora_b.shape[1] == a.N &&
              p->dlora_b.shape[2] == a.R, "lora_up_bwd: dh must be (column_blocks, T, r), dlora_b (token_chunks, N, r) -- see omk_lora_up_bwd_parts");
  OMK_REQUIRE(p->dh.dtype == OMK_F32 && p->dlora_b.dtype == OMK_F32 && is_dense(p->dh) && is_dense(p->dlora_b), "lora_up_bwd: dh, dlora_b must be dense f32");
  if (p->dy.dtype != OMK_BF16 || p->h.dtype != OMK_BF16 || a.R != 8 || a.N % 8 != 0 || p->dy.stride[1] != 1 || p->h.stride[1] != 1 ||
      p->lora_b.stride[1] != 1 || !aligned16(p->dy) || !aligned16(p->h) || (p->dy.stride[0] * 2) % 16 != 0 || (p->h.stride[0] * 2) % 16 != 0)
    return fail(OMK_EUNSUPPORTED, "lora_up_bwd: bf16 dy / h with 16-byte aligned rows, rank 8, out_features a multiple of 8 (use two GEMMs otherwise)");
  a.dy = (const uint16_t*)p->dy.data; a.dys = p->dy.stride[0]; a.B = p->lora_b.data; a.bs = p->lora_b.stride[0]; a.bdt = p->lora_b.dtype;
  a.h = (const uint16_t*)p->h.data; a.hs = p->h.stride[0]; a.dh = (float*)p->dh.data; a.dB = (float*)p->dlora_b.data;
  if (const char* e = getenv("OMK_LORA_UP_DBG")) a.dbg = atoi(e);
  dim3 grid((unsigned)((int64_t)a.NB * chunks)), block(256);
  OMK_LAUNCH(lora_up_bwd_kernel, grid, block, 0, stream, a);
  return finish_launch("lora_up_bwd");
}
