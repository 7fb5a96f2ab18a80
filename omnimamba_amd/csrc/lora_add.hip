// lora_add.hip -- out += scale * h B^T, in place, for the task LoRA of the in_proj (reference models/stage2/lora.py:263-279:
// result += lora_B(lora_A(dropout(x))) * scaling) at training / prefill token counts.
//
// The rank is 8: as a GEMM this is a K = 8 problem that the library runs as a full pass over the (tokens, out_features)
// result PLUS, in PyTorch's out-of-place addmm, a copy of that result in front of it -- +318 us on top of the 560 us base
// GEMM for 16 k tokens x 8512 features (tools/probe_addmm.py).  It is pure streaming: read the result once, write it once
// (558 MB -> ~110 us), eight FMAs per element with the eight B columns of a lane's features held in registers.
//   thread = 8 consecutive output features (16 bytes) x a strip of tokens; h rows are broadcast loads.
#include "omk_common.h"

namespace omk {

constexpr int LA_MAXR = 16;
struct LoraAddArgs {
  void* out; const void* h; const void* B; const uint8_t* mask; int64_t os, hs, bs, ms; int T, N, R, tokens_per_block, bdt; float scale;
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
#pragma unroll
  for (int i = 0; i < VEC; i++)
#pragma unroll
    for (int r = 0; r < RR; r++) bw[i][r] = a.scale * load_rt(a.B, (int64_t)(n0 + i) * a.bs + r, a.bdt);
  const int t0 = tb * a.tokens_per_block, t1 = t0 + a.tokens_per_block < a.T ? t0 + a.tokens_per_block : a.T;
  TO* out = (TO*)a.out + n0;
  const TO* h = (const TO*)a.h;
  constexpr int UN = RR == 8 ? 8 : 4;         // tokens in flight per thread
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
  if (present(p->mask)) {
    OMK_REQUIRE(p->mask.dtype == OMK_U8 && p->mask.ndim == 2 && p->mask.shape[0] == a.T && p->mask.shape[1] == a.N && p->mask.stride[1] == 1,
                "lora_add: mask must be u8 (T, N) with contiguous rows");
    a.mask = (const uint8_t*)p->mask.data; a.ms = p->mask.stride[0];
  }
  a.tokens_per_block = 64;
  const int nvec = a.N / vec, cvb = (nvec + 255) / 256, tbs = (a.T + a.tokens_per_block - 1) / a.tokens_per_block;
  dim3 grid((unsigned)((int64_t)cvb * tbs)), block(256);
  if (a.mask) {
    if (a.R == 8) OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 8, true>), grid, block, 0, stream, a));
    else OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 16, true>), grid, block, 0, stream, a));
  } else if (a.R == 8) OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 8>), grid, block, 0, stream, a));
  else OMK_DISPATCH_DTYPE(p->out.dtype, TO, OMK_LAUNCH((lora_add_kernel<TO, 16>), grid, block, 0, stream, a));
  return finish_launch("lora_add");
}
