// developer probe: which launch / access SHAPE streams "two bf16 reads + one bf16 write" (the gated RMSNorm forward: y rows contiguous,
// z a strided slice of the 8512-wide zxbcdt row, out contiguous) at what rate on gfx950.  Same bytes in every variant; the variants differ
// in what the review asks about: one-shot workgroups against persistent ones, a workgroup per row against a wave per row, loads in flight
// per lane, whether the row reduction is there at all.  Prints us and GB/s (algorithmic bytes = 3 * rows * 4096 * 2).
//   hipcc --offload-arch=gfx950 -O3 stream3_probe.hip -o stream3_probe && ./stream3_probe [rows]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int COLS = 4096, ZROW = 8512, VPR = COLS / 8;   // 512 16-byte vectors per row

__device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack(float a, float b) {
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u); ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}
__device__ __forceinline__ float silu(float z) { return z * __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// g = x silu(z) of one 16-byte vector pair; returns the sum of squares
__device__ __forceinline__ float gate8(const u32x4& x, const u32x4& z, float (&g)[8]) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    g[2 * e] = lo(x[e]) * silu(lo(z[e])); g[2 * e + 1] = hi(x[e]) * silu(hi(z[e]));
    s += g[2 * e] * g[2 * e] + g[2 * e + 1] * g[2 * e + 1];
  }
  return s;
}
__device__ __forceinline__ u32x4 scale8(const float (&g)[8], float r, const float* w) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) o[e] = pack(g[2 * e] * r * w[2 * e], g[2 * e + 1] * r * w[2 * e + 1]);
  return o;
}

struct A { const uint16_t* x; const uint16_t* z; uint16_t* y; const float* w; int64_t rows; int nt; unsigned long long* stamp; int perm; };

// (0) element-wise, one shot: UNR 16-byte vectors per lane and stream, a workgroup = 256 * UNR consecutive vectors; no reduction
template <int UNR>
__global__ __launch_bounds__(256) void flat_kernel(A a) {
  const int64_t v0 = ((int64_t)blockIdx.x * UNR) * 256 + threadIdx.x;
  u32x4 rx[UNR], rz[UNR];
#pragma unroll
  for (int k = 0; k < UNR; k++) {
    const int64_t v = v0 + k * 256, row = v / VPR; const int col = (int)(v % VPR) * 8;
    rx[k] = *reinterpret_cast<const u32x4*>(a.x + row * COLS + col);
    rz[k] = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + col);
  }
#pragma unroll
  for (int k = 0; k < UNR; k++) {
    const int64_t v = v0 + k * 256, row = v / VPR; const int col = (int)(v % VPR) * 8;
    float g[8]; gate8(rx[k], rz[k], g);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; e++) o[e] = pack(g[2 * e], g[2 * e + 1]);
    if (a.nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(a.y + row * COLS + col));
    else *reinterpret_cast<u32x4*>(a.y + row * COLS + col) = o;
  }
}

// (1) a workgroup of four waves per row (two vectors per lane and stream), full norm; ONE: grid = rows, else persistent with the next
//     row's loads in flight (what norms.hip does)
template <bool ONE, bool REDUCE>
__global__ __launch_bounds__(256) void wg_row_kernel(A a) {
  __shared__ float red[2][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int col[2]; float w[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    col[c] = ((c * 4 + wave) * 64 + lane) * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) w[c][i] = a.w[col[c] + i];
  }
  u32x4 rx[2], rz[2];
  auto issue = [&](int64_t row) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      rx[c] = *reinterpret_cast<const u32x4*>(a.x + row * COLS + col[c]);
      rz[c] = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + col[c]);
    }
  };
  int par = 0;
  const unsigned long long t_begin = a.stamp ? wall_clock64() : 0;
  // perm: workgroup b takes row index (b / 8) + (b % 8) (G / 8) -- an XCD (= b % 8) then owns a contiguous eighth of every G rows instead of every 8th row
  const int64_t first = a.perm ? (int64_t)(blockIdx.x >> 3) + (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) : blockIdx.x;
  issue(first);
  for (int64_t row = first; row < a.rows; row += gridDim.x) {
    float g[2][8]; float s = 0.f;
#pragma unroll
    for (int c = 0; c < 2; c++) s += gate8(rx[c], rz[c], g[c]);
    if (!ONE && row + gridDim.x < a.rows) issue(row + gridDim.x);
    float r = s;
    if (REDUCE) {
      s = wave_sum(s);
      if (lane == 0) red[par][wave] = s;
      __syncthreads();
      r = rsqrtf((red[par][0] + red[par][1] + red[par][2] + red[par][3]) * (1.f / COLS) + 1e-5f);
      par ^= 1;
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const u32x4 o = scale8(g[c], r, w[c]);
      if (a.nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(a.y + row * COLS + col[c]));
      else *reinterpret_cast<u32x4*>(a.y + row * COLS + col[c]) = o;
    }
    if (ONE) break;
  }
  if (a.stamp && threadIdx.x == 0) { a.stamp[2 * blockIdx.x] = t_begin; a.stamp[2 * blockIdx.x + 1] = wall_clock64(); }
}

// (4) the persistent workgroup-per-row kernel with other row maps: MAP 0 strided without prefetch, 1 a contiguous range of K rows per
//     workgroup (grid = rows / K) with prefetch, 2 rows handed out by an atomic counter (in arrival order, as the dispatcher hands out
//     one-shot workgroups; the next index is fetched while this row is processed)
template <int MAP>
__global__ __launch_bounds__(256) void wg_map_kernel(A a, int K, unsigned* ctr) {
  __shared__ float red[2][4];
  __shared__ unsigned nxt[2];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int col[2]; float w[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    col[c] = ((c * 4 + wave) * 64 + lane) * 8;
#pragma unroll
    for (int i = 0; i < 8; i++) w[c][i] = a.w[col[c] + i];
  }
  u32x4 rx[2], rz[2];
  auto issue = [&](int64_t row) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      rx[c] = *reinterpret_cast<const u32x4*>(a.x + row * COLS + col[c]);
      rz[c] = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + col[c]);
    }
  };
  int par = 0;
  auto body = [&](int64_t row, int64_t next) {
    if (MAP == 0) issue(row);
    float g[2][8]; float s = 0.f;
#pragma unroll
    for (int c = 0; c < 2; c++) s += gate8(rx[c], rz[c], g[c]);
    if (MAP != 0 && next >= 0) issue(next);
    s = wave_sum(s);
    if (lane == 0) red[par][wave] = s;
    __syncthreads();
    const float r = rsqrtf((red[par][0] + red[par][1] + red[par][2] + red[par][3]) * (1.f / COLS) + 1e-5f);
    par ^= 1;
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const u32x4 o = scale8(g[c], r, w[c]);
      if (a.nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(a.y + row * COLS + col[c]));
      else *reinterpret_cast<u32x4*>(a.y + row * COLS + col[c]) = o;
    }
  };
  if (MAP == 0) {
    for (int64_t row = blockIdx.x; row < a.rows; row += gridDim.x) body(row, -1);
  } else if (MAP == 1) {
    const int64_t r0 = (int64_t)blockIdx.x * K;
    issue(r0);
    for (int k = 0; k < K; k++) body(r0 + k, k + 1 < K ? r0 + k + 1 : -1);
  } else {
    // batches of K consecutive rows from an atomic counter; the NEXT batch index is fetched while this batch is processed
    if (threadIdx.x == 0) nxt[0] = atomicAdd(ctr, 1u);
    __syncthreads();
    int64_t b = nxt[0];
    const int64_t nb = a.rows / K;
    int q = 1;
    if (b < nb) issue(b * K);
    while (b < nb) {
      unsigned fetched = 0;
      if (threadIdx.x == 0) fetched = atomicAdd(ctr, 1u);     // consumed K - 2 rows later: the wave never waits for it alone
      int64_t bn = -1;
      for (int k = 0; k < K; k++) {
        int64_t next = b * K + k + 1;
        if (k == K - 2 && threadIdx.x == 0) nxt[q] = fetched;           // (published by this body's barrier)
        if (k + 1 == K) { bn = nxt[q]; next = bn < nb ? bn * K : -1; }
        body(b * K + k, next);
      }
      b = bn; q ^= 1;
    }
  }
}

// (2) a wave per row: eight vectors per lane and stream in flight, reduction inside the wave (no LDS, no barrier), weights from LDS
template <bool ONE>
__global__ __launch_bounds__(256) void wave_row_kernel(A a) {
  __shared__ __attribute__((aligned(16))) float wsh[COLS];
  for (int i = threadIdx.x; i < COLS; i += 256) wsh[i] = a.w[i];
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t nw = (int64_t)gridDim.x * 4;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < a.rows; row += nw) {
    u32x4 rx[8], rz[8];
#pragma unroll
    for (int c = 0; c < 8; c++) {
      rx[c] = *reinterpret_cast<const u32x4*>(a.x + row * COLS + (c * 64 + lane) * 8);
      rz[c] = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + (c * 64 + lane) * 8);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; c++) {   // g is recomputed in the second pass: the staging registers are the only copy
      float g[8]; s += gate8(rx[c], rz[c], g);
    }
    const float r = rsqrtf(wave_sum(s) * (1.f / COLS) + 1e-5f);
#pragma unroll
    for (int c = 0; c < 8; c++) {
      float g[8]; gate8(rx[c], rz[c], g);
      const u32x4 o = scale8(g, r, wsh + (c * 64 + lane) * 8);
      if (a.nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(a.y + row * COLS + (c * 64 + lane) * 8));
      else *reinterpret_cast<u32x4*>(a.y + row * COLS + (c * 64 + lane) * 8) = o;
    }
    if (ONE) break;
  }
}

// (3) half a row per wave pair: four vectors per lane and stream, two waves per row (one LDS exchange, no workgroup barrier needed
//     beyond the pair -- written with a workgroup barrier here)
template <bool ONE>
__global__ __launch_bounds__(256) void pair_row_kernel(A a) {
  __shared__ float red[2][4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wsub = wave & 1, wrow = wave >> 1;
  int par = 0;
  const int64_t nr = (int64_t)gridDim.x * 2;
  for (int64_t row0 = (int64_t)blockIdx.x * 2; row0 < a.rows; row0 += nr) {
    const int64_t row = row0 + wrow;
    u32x4 rx[4], rz[4]; int col[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      col[c] = ((c * 2 + wsub) * 64 + lane) * 8;
      rx[c] = *reinterpret_cast<const u32x4*>(a.x + row * COLS + col[c]);
      rz[c] = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + col[c]);
    }
    float g[4][8]; float s = 0.f;
#pragma unroll
    for (int c = 0; c < 4; c++) s += gate8(rx[c], rz[c], g[c]);
    s = wave_sum(s);
    if (lane == 0) red[par][wave] = s;
    __syncthreads();
    const float r = rsqrtf((red[par][2 * wrow] + red[par][2 * wrow + 1]) * (1.f / COLS) + 1e-5f);
    par ^= 1;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const u32x4 o = scale8(g[c], r, a.w + col[c]);
      if (a.nt) __builtin_nontemporal_store(o, reinterpret_cast<u32x4*>(a.y + row * COLS + col[c]));
      else *reinterpret_cast<u32x4*>(a.y + row * COLS + col[c]) = o;
    }
    if (ONE) break;
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// the BACKWARD's five streams (x, z, dy -> dx, dz; dz into the strided zxbcdt-shaped gradient) with the per-column weight gradient kept in
// registers across the rows of a workgroup and written as ONE partial row per row group at the end (what norms.hip does).
//   NG row groups of four waves per workgroup (256 * NG threads); a group walks K rows: STRIDED (persistent, rows g, g + G, ...) or a
//   contiguous range (one shot, grid = rows / (NG K)); the NG groups' partial sums are added in LDS before the store (NG > 1)
// ---------------------------------------------------------------------------------------------------------------------------------------
struct AB { const uint16_t* x; const uint16_t* z; const uint16_t* dy; uint16_t* dx; uint16_t* dz; const float* w; float* part; int64_t rows; };
template <int NG, bool STRIDED, bool NOSYNC = false>   // NOSYNC: the row sums stay inside the wave (wrong result, same work: what the barrier costs)
__global__ __launch_bounds__(256 * NG) void bwd_kernel(AB a, int K) {
  __shared__ float red[2][NG][4][2];
  __shared__ __attribute__((aligned(16))) float wsh[COLS];
  __shared__ __attribute__((aligned(16))) float acc[NG > 1 ? COLS : 4];
  const int wave = (threadIdx.x >> 6) & 3, grp = threadIdx.x >> 8, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < COLS; i += 256 * NG) { wsh[i] = a.w[i]; if (NG > 1) acc[i] = 0.f; }
  __syncthreads();
  int col[2];
#pragma unroll
  for (int c = 0; c < 2; c++) col[c] = ((c * 4 + wave) * 64 + lane) * 8;
  float dw[2][8];
#pragma unroll
  for (int c = 0; c < 2; c++)
#pragma unroll
    for (int i = 0; i < 8; i++) dw[c][i] = 0.f;
  const int64_t G = (int64_t)gridDim.x * NG, g = (int64_t)blockIdx.x * NG + grp;
  int par = 0;
  for (int k = 0; k < K; k++) {
    const int64_t row = STRIDED ? g + (int64_t)k * G : g * K + k;
    u32x4 rx[2], rz[2], rd[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      rx[c] = *reinterpret_cast<const u32x4*>(a.x + row * COLS + col[c]);
      rd[c] = *reinterpret_cast<const u32x4*>(a.dy + row * COLS + col[c]);
      rz[c] = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + col[c]);
    }
    float gv[2][8], wdy[2][8], sg[2][8];
    float s2 = 0.f, t2 = 0.f;
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const float zf = (i & 1) ? hi(rz[c][i >> 1]) : lo(rz[c][i >> 1]);
        const float xf = (i & 1) ? hi(rx[c][i >> 1]) : lo(rx[c][i >> 1]);
        const float df = (i & 1) ? hi(rd[c][i >> 1]) : lo(rd[c][i >> 1]);
        sg[c][i] = __builtin_amdgcn_rcpf(1.f + __expf(-zf));
        gv[c][i] = xf * (zf * sg[c][i]);
        wdy[c][i] = df * wsh[col[c] + i];
        s2 += gv[c][i] * gv[c][i]; t2 += gv[c][i] * wdy[c][i];
      }
    s2 = wave_sum(s2); t2 = wave_sum(t2);
    if (!NOSYNC) {
      if (lane == 0) { red[par][grp][wave][0] = s2; red[par][grp][wave][1] = t2; }
      __syncthreads();
      s2 = red[par][grp][0][0] + red[par][grp][1][0] + red[par][grp][2][0] + red[par][grp][3][0];
      t2 = red[par][grp][0][1] + red[par][grp][1][1] + red[par][grp][2][1] + red[par][grp][3][1];
      par ^= 1;
    }
    const float rstd = rsqrtf(s2 * (1.f / COLS) + 1e-5f), c1 = rstd * t2 * (1.f / COLS);
#pragma unroll
    for (int c = 0; c < 2; c++) {
      u32x4 ox, oz;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float o[2][2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int i = 2 * e + h;
          const float zf = h ? hi(rz[c][e]) : lo(rz[c][e]);
          const float xf = h ? hi(rx[c][e]) : lo(rx[c][e]);
          const float df = h ? hi(rd[c][e]) : lo(rd[c][e]);
          const float xhat = gv[c][i] * rstd;
          dw[c][i] += df * xhat;
          const float ds = (wdy[c][i] - xhat * c1) * rstd * sg[c][i];
          o[0][h] = ds * zf;
          o[1][h] = ds * xf * (1.f + zf * (1.f - sg[c][i]));
        }
        ox[e] = pack(o[0][0], o[0][1]); oz[e] = pack(o[1][0], o[1][1]);
      }
      *reinterpret_cast<u32x4*>(a.dx + row * COLS + col[c]) = ox;
      *reinterpret_cast<u32x4*>(a.dz + row * ZROW + col[c]) = oz;
    }
  }
  if (NG > 1) {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 8; i++) atomicAdd(&acc[col[c] + i], dw[c][i]);
    __syncthreads();
    for (int i = threadIdx.x; i < COLS; i += 256 * NG) a.part[(int64_t)blockIdx.x * COLS + i] = acc[i];
  } else {
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
      for (int i = 0; i < 8; i++) a.part[g * COLS + col[c] + i] = dw[c][i];
  }
}
// the same backward with EIGHT waves per row (one 16-byte vector per lane and stream: half the registers, twice the waves per SIMD), persistent, strided rows
__global__ __launch_bounds__(512) void bwd8_kernel(AB a, int K) {
  __shared__ float red[2][8][2];
  __shared__ __attribute__((aligned(16))) float wsh[COLS];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < COLS; i += 512) wsh[i] = a.w[i];
  __syncthreads();
  const int col = (wave * 64 + lane) * 8;
  float dw[8];
#pragma unroll
  for (int i = 0; i < 8; i++) dw[i] = 0.f;
  int par = 0;
  for (int k = 0; k < K; k++) {
    const int64_t row = blockIdx.x + (int64_t)k * gridDim.x;
    const u32x4 rx = *reinterpret_cast<const u32x4*>(a.x + row * COLS + col);
    const u32x4 rd = *reinterpret_cast<const u32x4*>(a.dy + row * COLS + col);
    const u32x4 rz = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + col);
    float gv[8], wdy[8], sg[8];
    float s2 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const float zf = (i & 1) ? hi(rz[i >> 1]) : lo(rz[i >> 1]);
      const float xf = (i & 1) ? hi(rx[i >> 1]) : lo(rx[i >> 1]);
      const float df = (i & 1) ? hi(rd[i >> 1]) : lo(rd[i >> 1]);
      sg[i] = __builtin_amdgcn_rcpf(1.f + __expf(-zf));
      gv[i] = xf * (zf * sg[i]);
      wdy[i] = df * wsh[col + i];
      s2 += gv[i] * gv[i]; t2 += gv[i] * wdy[i];
    }
    s2 = wave_sum(s2); t2 = wave_sum(t2);
    if (lane == 0) { red[par][wave][0] = s2; red[par][wave][1] = t2; }
    __syncthreads();
    s2 = 0.f; t2 = 0.f;
#pragma unroll
    for (int q = 0; q < 8; q++) { s2 += red[par][q][0]; t2 += red[par][q][1]; }
    par ^= 1;
    const float rstd = rsqrtf(s2 * (1.f / COLS) + 1e-5f), c1 = rstd * t2 * (1.f / COLS);
    u32x4 ox, oz;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float o[2][2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int i = 2 * e + h;
        const float zf = h ? hi(rz[e]) : lo(rz[e]);
        const float xf = h ? hi(rx[e]) : lo(rx[e]);
        const float df = h ? hi(rd[e]) : lo(rd[e]);
        const float xhat = gv[i] * rstd;
        dw[i] += df * xhat;
        const float ds = (wdy[i] - xhat * c1) * rstd * sg[i];
        o[0][h] = ds * zf;
        o[1][h] = ds * xf * (1.f + zf * (1.f - sg[i]));
      }
      ox[e] = pack(o[0][0], o[0][1]); oz[e] = pack(o[1][0], o[1][1]);
    }
    *reinterpret_cast<u32x4*>(a.dx + row * COLS + col) = ox;
    *reinterpret_cast<u32x4*>(a.dz + row * ZROW + col) = oz;
  }
#pragma unroll
  for (int i = 0; i < 8; i++) a.part[(int64_t)blockIdx.x * COLS + col + i] = dw[i];
}
// the five streams with trivial arithmetic, one shot (the ceiling of the access pattern): UNR vectors per lane and stream
template <int UNR>
__global__ __launch_bounds__(256) void flat5_kernel(AB a) {
  const int64_t v0 = ((int64_t)blockIdx.x * UNR) * 256 + threadIdx.x;
  u32x4 rx[UNR], rz[UNR], rd[UNR];
#pragma unroll
  for (int k = 0; k < UNR; k++) {
    const int64_t v = v0 + k * 256, row = v / VPR; const int col = (int)(v % VPR) * 8;
    rx[k] = *reinterpret_cast<const u32x4*>(a.x + row * COLS + col);
    rd[k] = *reinterpret_cast<const u32x4*>(a.dy + row * COLS + col);
    rz[k] = *reinterpret_cast<const u32x4*>(a.z + row * ZROW + col);
  }
#pragma unroll
  for (int k = 0; k < UNR; k++) {
    const int64_t v = v0 + k * 256, row = v / VPR; const int col = (int)(v % VPR) * 8;
    u32x4 o1, o2;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      o1[e] = pack(lo(rx[k][e]) * lo(rz[k][e]) + lo(rd[k][e]), hi(rx[k][e]) * hi(rz[k][e]) + hi(rd[k][e]));
      o2[e] = pack(lo(rx[k][e]) + lo(rz[k][e]) * lo(rd[k][e]), hi(rx[k][e]) + hi(rz[k][e]) * hi(rd[k][e]));
    }
    *reinterpret_cast<u32x4*>(a.dx + row * COLS + col) = o1;
    *reinterpret_cast<u32x4*>(a.dz + row * ZROW + col) = o2;
  }
}
// the fold of the partial rows (what launch_reduce does): P rows of COLS floats -> one
__global__ __launch_bounds__(256) void fold_kernel(const float* part, int P, float* out) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  __shared__ float s[4][64];
  float v = 0.f;
  for (int p = q; p < P; p += 4) v += part[(int64_t)p * COLS + c];
  s[q][threadIdx.x & 63] = v;
  __syncthreads();
  if (q == 0) out[c] = s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x];
}

template <class F>
static float time_us(F launch, int reps = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int t = 0; t < 3; t++) {
    launch(); launch();
    hipEventRecord(e0);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms * 1000.f / reps < best) best = ms * 1000.f / reps;
  }
  return best;
}

int main(int argc, char** argv) {
  const int64_t rows = argc > 1 ? atoll(argv[1]) : 32768;
  A a{}; a.rows = rows;
  uint16_t *x, *z, *y; float* w;
  hipMalloc(&x, rows * COLS * 2); hipMalloc(&z, rows * ZROW * 2); hipMalloc(&y, rows * COLS * 2); hipMalloc(&w, COLS * 4);
  {
    std::vector<uint16_t> h(rows * ZROW);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 22 & 0x3ff) + ((i & 1) << 15));   // +-[~0.008, ~0.03]
    hipMemcpy(z, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(x, h.data(), rows * COLS * 2, hipMemcpyHostToDevice);
    std::vector<float> hw(COLS, 1.f); hipMemcpy(w, hw.data(), COLS * 4, hipMemcpyHostToDevice);
  }
  a.x = x; a.z = z; a.y = y; a.w = w;
  const double bytes = 3.0 * rows * COLS * 2;
  auto report = [&](const char* name, float us) { printf("%-64s %8.1f us %8.0f GB/s\n", name, us, bytes / us * 1e-3); fflush(stdout); };
  for (int nt = 0; nt < 1; nt++) {
    a.nt = nt;
    printf("-- output stores %s\n", nt ? "non-temporal" : "plain");
    const int64_t nv = rows * VPR;
    report("flat one-shot, 1 vector per lane", time_us([&] { flat_kernel<1><<<dim3(nv / 256), 256>>>(a); }));
    report("flat one-shot, 2 vectors per lane", time_us([&] { flat_kernel<2><<<dim3(nv / 512), 256>>>(a); }));
    report("flat one-shot, 4 vectors per lane", time_us([&] { flat_kernel<4><<<dim3(nv / 1024), 256>>>(a); }));
    report("workgroup per row, one shot, no reduction", time_us([&] { wg_row_kernel<true, false><<<dim3(rows), 256>>>(a); }));
    report("workgroup per row, one shot, norm", time_us([&] { wg_row_kernel<true, true><<<dim3(rows), 256>>>(a); }));
    for (int g : {1024, 2048, 4096}) {
      char nm[96];
      snprintf(nm, sizeof nm, "workgroup per row, persistent x %d, prefetch, no reduction", g);
      report(nm, time_us([&] { wg_row_kernel<false, false><<<dim3(g), 256>>>(a); }));
      snprintf(nm, sizeof nm, "workgroup per row, persistent x %d, prefetch, norm (= norms.hip)", g);
      report(nm, time_us([&] { wg_row_kernel<false, true><<<dim3(g), 256>>>(a); }));
    }
    {
      unsigned* ctr; hipMalloc(&ctr, 4);
      for (int g : {1024, 2048}) {
        char nm[96]; snprintf(nm, sizeof nm, "workgroup per row, persistent x %d, strided, NO prefetch, norm", g);
        report(nm, time_us([&] { wg_map_kernel<0><<<dim3(g), 256>>>(a, 0, ctr); }));
      }
      for (int K : {2, 4, 8, 16, 32}) {
        char nm[96]; snprintf(nm, sizeof nm, "workgroup per %d consecutive rows (grid %d), prefetch, norm", K, (int)(rows / K));
        report(nm, time_us([&] { wg_map_kernel<1><<<dim3(rows / K), 256>>>(a, K, ctr); }));
      }
      for (int g : {1024, 2048})
        for (int K : {2, 4, 8, 16}) {
          char nm[96]; snprintf(nm, sizeof nm, "persistent x %d, batches of %d rows from an atomic queue, norm", g, K);
          report(nm, time_us([&] { hipMemsetAsync(ctr, 0, 4); wg_map_kernel<2><<<dim3(g), 256>>>(a, K, ctr); }));
        }
      hipFree(ctr);
    }
    report("wave per row, one shot (4 rows per workgroup), norm", time_us([&] { wave_row_kernel<true><<<dim3(rows / 4), 256>>>(a); }));
    for (int g : {512, 1024, 2048}) {
      char nm[96]; snprintf(nm, sizeof nm, "wave per row, persistent x %d, norm", g);
      report(nm, time_us([&] { wave_row_kernel<false><<<dim3(g), 256>>>(a); }));
    }
    report("wave pair per row, one shot (2 rows per workgroup), norm", time_us([&] { pair_row_kernel<true><<<dim3(rows / 2), 256>>>(a); }));
    for (int g : {1024, 2048}) {
      char nm[96]; snprintf(nm, sizeof nm, "wave pair per row, persistent x %d, norm", g);
      report(nm, time_us([&] { pair_row_kernel<false><<<dim3(g), 256>>>(a); }));
    }
  }
  {   // when do the workgroups of the persistent kernel start and finish?  (100 MHz wall clock)
    for (int pass = 0; pass < 4; pass++) {
      const int g = (pass & 1) ? 2048 : 1024;
      unsigned long long* st; hipMalloc(&st, (size_t)g * 16);
      A b = a; b.stamp = st; b.nt = 0; b.perm = pass >> 1;
      if (b.perm) printf("(rows permuted: an XCD owns a contiguous eighth of every G rows)\n");
      wg_row_kernel<false, true><<<dim3(g), 256>>>(b); hipDeviceSynchronize();
      wg_row_kernel<false, true><<<dim3(g), 256>>>(b); hipDeviceSynchronize();
      std::vector<unsigned long long> h(2 * g); hipMemcpy(h.data(), st, (size_t)g * 16, hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull; for (int i = 0; i < g; i++) t0 = h[2 * i] < t0 ? h[2 * i] : t0;
      std::vector<double> b0(g), e0(g);
      for (int i = 0; i < g; i++) { b0[i] = (h[2 * i] - t0) * 0.01; e0[i] = (h[2 * i + 1] - t0) * 0.01; }
      std::vector<double> bs = b0, es = e0; std::sort(bs.begin(), bs.end()); std::sort(es.begin(), es.end());
      printf("persistent x %d: workgroup start  min %.1f  median %.1f  max %.1f us | finish  min %.1f  5%% %.1f  median %.1f  95%% %.1f  max %.1f us\n", g,
             bs[0], bs[g / 2], bs[g - 1], es[0], es[g / 20], es[g / 2], es[g - g / 20 - 1], es[g - 1]);
      // by XCD (workgroup id mod 8): median finish
      for (int xcd = 0; xcd < 8; xcd++) { std::vector<double> v; for (int i = xcd; i < g; i += 8) v.push_back(e0[i]); std::sort(v.begin(), v.end()); printf("   xcd %d: finish median %.1f max %.1f\n", xcd, v[v.size() / 2], v.back()); }
      hipFree(st);
    }
  }
  {
    AB b{}; b.rows = rows; b.x = x; b.z = z; b.w = w;
    uint16_t *dy, *dz; float *part, *dwo;
    hipMalloc(&dy, rows * COLS * 2); hipMalloc(&dz, rows * ZROW * 2); hipMalloc(&part, (size_t)rows / 2 * COLS * 4); hipMalloc(&dwo, COLS * 4);
    hipMemcpy(dy, x, rows * COLS * 2, hipMemcpyDeviceToDevice);
    b.dy = dy; b.dx = y; b.dz = dz; b.part = part;
    const double bb = 5.0 * rows * COLS * 2;
    auto rep = [&](const char* name, float us, int P) {
      const float f = time_us([&] { fold_kernel<<<dim3(COLS / 64), 256>>>(part, P, dwo); });
      printf("%-72s %8.1f us %6.0f GB/s   + fold of %5d partial rows %6.1f us = %7.1f\n", name, us, bb / us * 1e-3, P, f, us + f); fflush(stdout);
    };
    printf("-- backward: x, z, dy -> dx, dz + dw partial rows (GB/s on the five streams only)\n");
    char nm[128];
    {
      const int64_t nv = rows * VPR;
      float u = time_us([&] { flat5_kernel<1><<<dim3(nv / 256), 256>>>(b); });
      printf("%-72s %8.1f us %6.0f GB/s\n", "five streams, trivial arithmetic, flat one-shot, 1 vector per lane", u, bb / u * 1e-3);
      u = time_us([&] { flat5_kernel<2><<<dim3(nv / 512), 256>>>(b); });
      printf("%-72s %8.1f us %6.0f GB/s\n", "five streams, trivial arithmetic, flat one-shot, 2 vectors per lane", u, bb / u * 1e-3);
      uint16_t* dzc; hipMalloc(&dzc, rows * COLS * 2);
    }
    for (int G : {1024, 2048}) {
      snprintf(nm, sizeof nm, "persistent x %d, strided rows (= norms.hip)", G);
      rep(nm, time_us([&] { bwd_kernel<1, true><<<dim3(G), 256>>>(b, (int)(rows / G)); }), G);
    }
    for (int G : {1024, 2048}) {
      snprintf(nm, sizeof nm, "persistent x %d, strided rows, NO workgroup barrier (row sums inside the wave)", G);
      rep(nm, time_us([&] { bwd_kernel<1, true, true><<<dim3(G), 256>>>(b, (int)(rows / G)); }), G);
    }
    for (int G : {512, 1024, 2048}) {
      snprintf(nm, sizeof nm, "persistent x %d, EIGHT waves per row (one vector per lane and stream)", G);
      rep(nm, time_us([&] { bwd8_kernel<<<dim3(G), 512>>>(b, (int)(rows / G)); }), G);
    }
    for (int K : {4, 8, 16, 32}) {
      snprintf(nm, sizeof nm, "one shot, %d consecutive rows per workgroup (grid %d)", K, (int)(rows / K));
      rep(nm, time_us([&] { bwd_kernel<1, false><<<dim3(rows / K), 256>>>(b, K); }), (int)(rows / K));
    }
  }
  return 0;
}
