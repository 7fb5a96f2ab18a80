// developer probe: the access SHAPE of the channel-last causal conv1d forward (read the 4352-channel xBC slice of the 8512-wide zxbcdt rows,
// write contiguous 4352-channel rows; a lane walks TL tokens of its channels with a W = 4 register window) -- which of {bytes per lane, tokens
// per strip, tokens in flight, wave -> (strip, column block) map} streams fastest on gfx950.  Arithmetic: the four taps (fma), no activation.
//   hipcc --offload-arch=gfx950 -O3 conv_probe.hip -o conv_probe && ./conv_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

constexpr int C = 4352, XROW = 8512, XOFF = 4096;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack(float a, float b) {
  uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
  ua += 0x7fffu + ((ua >> 16) & 1u); ub += 0x7fffu + ((ub >> 16) & 1u);
  return (ua >> 16) | (ub & 0xffff0000u);
}
struct A { const uint16_t* x; uint16_t* y; const float* w; int B, L; };

// NW: 32-bit words per lane (2 = four channels, 4 = eight); a wave = 64 lanes = 64 NW 2-byte-pairs of one token row; TL tokens per strip,
// TG tokens requested ahead.  MAP 0: workgroup = four strips of one column block (conv1d.hip); 1: consecutive waves = consecutive column
// blocks of one strip (flat); 2: workgroup = one strip, its four waves = four adjacent column blocks (needs CB % 4 == 0 -> padded)
template <int NW, int TL, int TG, int MAP>
__global__ __launch_bounds__(256) void conv_kernel(A a) {
  typedef uint32_t vec __attribute__((ext_vector_type(NW)));
  constexpr int CPL = 2 * NW;                       // channels per lane
  const int CB = (C / CPL + 63) / 64, NT = a.L / TL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int cb, ts, b;
  if (MAP == 0) { const int NT4 = NT / 4; cb = blockIdx.x % CB; ts = (blockIdx.x / CB % NT4) * 4 + wave; b = blockIdx.x / (CB * NT4); }
  else if (MAP == 1) { const int64_t w = (int64_t)blockIdx.x * 4 + wave; cb = (int)(w % CB); ts = (int)(w / CB % NT); b = (int)(w / ((int64_t)CB * NT)); if (b >= a.B) return; }
  else { const int CB4 = (CB + 3) / 4; cb = (blockIdx.x % CB4) * 4 + wave; ts = blockIdx.x / CB4 % NT; b = blockIdx.x / (CB4 * NT); if (cb >= CB) return; }
  const int ch = (cb * 64 + lane) * CPL;
  if (ch >= C) return;
  float w[4][CPL];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int i = 0; i < CPL; i++) w[k][i] = a.w[k * C + ch + i];
  const uint16_t* x = a.x + ((int64_t)b * a.L + (int64_t)ts * TL) * XROW + XOFF + ch;
  uint16_t* y = a.y + ((int64_t)b * a.L + (int64_t)ts * TL) * C + ch;
  float win[4][CPL];
#pragma unroll
  for (int s = 1; s < 4; s++) {
    const vec h = (ts == 0) ? vec{} : *reinterpret_cast<const vec*>(x + (int64_t)(s - 4) * XROW);
#pragma unroll
    for (int i = 0; i < NW; i++) { win[s][2 * i] = lo(h[i]); win[s][2 * i + 1] = hi(h[i]); }
  }
  vec r[TG];
#pragma unroll
  for (int t = 0; t < TG; t++) r[t] = *reinterpret_cast<const vec*>(x + (int64_t)t * XROW);
  for (int t0 = 0; t0 < TL; t0 += TG) {
    vec q[TG];
#pragma unroll
    for (int t = 0; t < TG; t++) q[t] = r[t];
    if (t0 + TG < TL) {
#pragma unroll
      for (int t = 0; t < TG; t++) r[t] = *reinterpret_cast<const vec*>(x + (int64_t)(t0 + TG + t) * XROW);
    }
#pragma unroll
    for (int t = 0; t < TG; t++) {
#pragma unroll
      for (int s = 0; s < 3; s++)
#pragma unroll
        for (int i = 0; i < CPL; i++) win[s][i] = win[s + 1][i];
#pragma unroll
      for (int i = 0; i < NW; i++) { win[3][2 * i] = lo(q[t][i]); win[3][2 * i + 1] = hi(q[t][i]); }
      vec o;
#pragma unroll
      for (int i = 0; i < NW; i++) {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; k++) { a0 = __builtin_fmaf(w[k][2 * i], win[k][2 * i], a0); a1 = __builtin_fmaf(w[k][2 * i + 1], win[k][2 * i + 1], a1); }
        o[i] = pack(a0, a1);
      }
      *reinterpret_cast<vec*>(y + (int64_t)(t0 + t) * C) = o;
    }
  }
}

// the copy a flat element-wise kernel makes of the slice (what torch's copy_ reaches): 16 bytes per lane, one shot
__global__ __launch_bounds__(256) void copy_kernel(A a) {
  const int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x, row = v / (C / 8); const int col = (int)(v % (C / 8)) * 8;
  *reinterpret_cast<u32x4*>(a.y + row * C + col) = *reinterpret_cast<const u32x4*>(a.x + row * XROW + XOFF + col);
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

// the BACKWARD's three streams: x (slice of zxbcdt rows) and dout (contiguous rows) in, dx (slice of the zxbcdt-shaped gradient) out; per
// lane NW words, a W = 4 window over x and dout, 4 + 4 + 4 fma per channel and token, dw / db through LDS + atomics at the end of the
// workgroup (as conv1d_bwd_cl4_kernel: four strips per workgroup)
struct AB { const uint16_t* x; const uint16_t* g; uint16_t* dx; const float* w; float* dw; float* part; int B, L; };
template <int NW, int TL, int TG, int RED>   // RED 0: atomics, 1: one partial row per workgroup (plain stores; folded by a second launch), 2: dropped
__global__ __launch_bounds__(256) void conv_bwd_kernel(AB a) {
  typedef uint32_t vec __attribute__((ext_vector_type(NW)));
  constexpr int CPL = 2 * NW;
  __shared__ float sred[4][64][CPL * 5];
  const int CB = (C / CPL + 63) / 64, NT4 = a.L / TL / 4;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cb = blockIdx.x % CB, ts = (blockIdx.x / CB % NT4) * 4 + wave, b = blockIdx.x / (CB * NT4);
  const int chr = (cb * 64 + lane) * CPL;
  const bool ok = chr < C;
  const int ch = ok ? chr : 0;
  float w[4][CPL], dwa[4][CPL], dba[CPL];
#pragma unroll
  for (int i = 0; i < CPL; i++) {
    dba[i] = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) { w[k][i] = a.w[(ch + i) * 4 + k]; dwa[k][i] = 0.f; }
  }
  const int64_t r0 = (int64_t)b * a.L + (int64_t)ts * TL;
  const uint16_t* x = a.x + r0 * XROW + XOFF + ch;
  const uint16_t* g = a.g + r0 * C + ch;
  uint16_t* dx = a.dx + r0 * XROW + XOFF + ch;
  float xw[4][CPL], gw[4][CPL];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int i = 0; i < CPL; i++) { xw[s][i] = 0.f; gw[s][i] = 0.f; }
  vec rx[TG], rg[TG];
  const int NTOK = TL + 3;                       // the strip's dx rows need three more dout rows (clamped at the end of the data)
  const int64_t last = (int64_t)a.B * a.L - 1 - r0;
  auto req = [&](int t0) {
#pragma unroll
    for (int t = 0; t < TG; t++) {
      int64_t tt = t0 + t; if (tt > last) tt = last;
      rx[t] = *reinterpret_cast<const vec*>(x + tt * XROW);
      rg[t] = *reinterpret_cast<const vec*>(g + tt * C);
    }
  };
  req(0);
  for (int t0 = 0; t0 < NTOK; t0 += TG) {
    vec qx[TG], qg[TG];
#pragma unroll
    for (int t = 0; t < TG; t++) { qx[t] = rx[t]; qg[t] = rg[t]; }
    if (t0 + TG < NTOK) req(t0 + TG);
#pragma unroll
    for (int t = 0; t < TG; t++) {
      if (t0 + t < NTOK) {
#pragma unroll
        for (int s = 0; s < 3; s++)
#pragma unroll
          for (int i = 0; i < CPL; i++) { xw[s][i] = xw[s + 1][i]; gw[s][i] = gw[s + 1][i]; }
#pragma unroll
        for (int i = 0; i < NW; i++) { xw[3][2 * i] = lo(qx[t][i]); xw[3][2 * i + 1] = hi(qx[t][i]); gw[3][2 * i] = lo(qg[t][i]); gw[3][2 * i + 1] = hi(qg[t][i]); }
        vec o;
#pragma unroll
        for (int i = 0; i < NW; i++) {
          float acc[2];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int c = 2 * i + h;
            float pre = 0.f, d = gw[3][c], ax = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) pre = __builtin_fmaf(w[k][c], xw[k][c], pre);
            d *= __builtin_amdgcn_rcpf(1.f + __expf(-pre));                 // (stands for silu_grad: one exp, one rcp)
            gw[3][c] = d; dba[c] += d;
#pragma unroll
            for (int k = 0; k < 4; k++) { dwa[k][c] = __builtin_fmaf(d, xw[k][c], dwa[k][c]); ax = __builtin_fmaf(w[k][c], gw[3 - k][c], ax); }
            acc[h] = ax;
          }
          o[i] = pack(acc[0], acc[1]);
        }
        if (t0 + t >= 3) *reinterpret_cast<vec*>(dx + (int64_t)(t0 + t - 3) * XROW) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < CPL; i++) {
    sred[wave][lane][i * 5 + 4] = dba[i];
#pragma unroll
    for (int k = 0; k < 4; k++) sred[wave][lane][i * 5 + k] = dwa[k][i];
  }
  __syncthreads();
  if (RED == 0) {
    if (wave == 0 && ok) {
#pragma unroll
      for (int j = 0; j < CPL * 5; j++) atomicAdd(a.dw + (int64_t)ch * 5 + j, sred[0][lane][j] + sred[1][lane][j] + sred[2][lane][j] + sred[3][lane][j]);
    }
  } else if (RED == 1) {
    // partial[(b, strip group)][channel * 5 + j]: the lane's CPL * 5 values are consecutive floats -> 16-byte stores, spread over the four waves
    float* row = a.part + ((int64_t)(blockIdx.x / CB)) * (C * 5) + (int64_t)ch * 5;
    if (ok) {
      for (int j = wave; j < CPL * 5; j += 4) row[j] = sred[0][lane][j] + sred[1][lane][j] + sred[2][lane][j] + sred[3][lane][j];
    }
  }
}
__global__ __launch_bounds__(256) void fold_kernel(const float* part, int P, float* out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C * 5) return;
  float v = 0.f;
  for (int p = 0; p < P; p++) v += part[(int64_t)p * (C * 5) + c];
  out[c] = v;
}
template <int NW, int TL, int TG>
static void runb(AB a) {
  constexpr int CPL = 2 * NW;
  const int CB = (C / CPL + 63) / 64;
  const int64_t grid = (int64_t)a.B * (a.L / TL / 4) * CB;
  const int P = (int)(grid / CB);
  const float u0 = time_us([&] { conv_bwd_kernel<NW, TL, TG, 0><<<dim3((unsigned)grid), 256>>>(a); });
  const float u1 = time_us([&] { conv_bwd_kernel<NW, TL, TG, 1><<<dim3((unsigned)grid), 256>>>(a); });
  const float uf = time_us([&] { fold_kernel<<<dim3((C * 5 + 255) / 256), 256>>>(a.part, P, a.dw); });
  const float u2 = time_us([&] { conv_bwd_kernel<NW, TL, TG, 2><<<dim3((unsigned)grid), 256>>>(a); });
  const double bytes = 3.0 * a.B * a.L * C * 2;
  printf("backward  %2d B/lane  strip %3d  %2d in flight  grid %6lld | atomics %7.1f us | partial rows %7.1f + fold of %4d rows %5.1f = %7.1f us | dropped %7.1f us %6.0f GB/s\n",
         4 * NW, TL, TG, (long long)grid, u0, u1, P, uf, u1 + uf, u2, bytes / u2 * 1e-3);
  fflush(stdout);
}

template <int NW, int TL, int TG, int MAP>
static void run(A a, const char* what) {
  constexpr int CPL = 2 * NW;
  const int CB = (C / CPL + 63) / 64, NT = a.L / TL;
  int64_t grid;
  if (MAP == 0) grid = (int64_t)a.B * (NT / 4) * CB;
  else if (MAP == 1) grid = ((int64_t)a.B * NT * CB + 3) / 4;
  else grid = (int64_t)a.B * NT * ((CB + 3) / 4);
  const float us = time_us([&] { conv_kernel<NW, TL, TG, MAP><<<dim3((unsigned)grid), 256>>>(a); });
  const double bytes = 2.0 * a.B * a.L * C * 2;
  printf("%2d B/lane  strip %3d  %2d in flight  %-34s grid %6lld  %7.1f us %6.0f GB/s\n", 4 * NW, TL, TG, what, (long long)grid, us, bytes / us * 1e-3);
  fflush(stdout);
}

int main() {
  A a{}; a.B = 8; a.L = 4096;
  const int64_t rows = (int64_t)a.B * a.L;
  uint16_t *x, *y; float* w;
  hipMalloc(&x, rows * XROW * 2); hipMalloc(&y, rows * C * 2); hipMalloc(&w, 4 * C * 4);
  {
    std::vector<uint16_t> h(rows * XROW);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint16_t)(0x3c00 + (i * 2654435761u >> 22 & 0x3ff));
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> hw(4 * C, 0.25f); hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
  }
  a.x = x; a.y = y; a.w = w;
  const double bytes = 2.0 * rows * C * 2;
  { const float us = time_us([&] { copy_kernel<<<dim3((unsigned)(rows * (C / 8) / 256)), 256>>>(a); }); printf("flat 16-byte copy of the slice %52s %7.1f us %6.0f GB/s\n", "", us, bytes / us * 1e-3); }
#define ROW(NW, TL, TG) run<NW, TL, TG, 0>(a, "workgroup = 4 strips (conv1d.hip)"); run<NW, TL, TG, 1>(a, "consecutive waves = one strip"); run<NW, TL, TG, 2>(a, "workgroup = 4 column blocks")
  {
    AB q{}; q.B = a.B; q.L = a.L; q.x = x; q.w = w;
    uint16_t *g, *dx; float* dw;
    hipMalloc(&g, rows * C * 2); hipMalloc(&dx, rows * XROW * 2); hipMalloc(&dw, C * 5 * 4 + 64); hipMalloc(&q.part, (size_t)2048 * C * 5 * 4);
    hipMemcpy(g, x, rows * C * 2, hipMemcpyDeviceToDevice); hipMemset(dw, 0, C * 5 * 4);
    q.g = g; q.dx = dx; q.dw = dw;
    if (0) { runb<1, 128, 4>(q); runb<1, 64, 4>(q); runb<1, 64, 8>(q); runb<1, 32, 4>(q); runb<1, 32, 8>(q); runb<1, 16, 4>(q); runb<1, 16, 8>(q);
    runb<2, 128, 4>(q); runb<2, 64, 4>(q); runb<2, 32, 4>(q); runb<2, 16, 4>(q); }
  }
  ROW(2, 16, 8); ROW(2, 8, 8); ROW(2, 8, 4); ROW(4, 8, 8); ROW(4, 8, 4); ROW(2, 16, 16);
  return 0;
  ROW(2, 64, 8); ROW(2, 64, 4); ROW(2, 64, 16); ROW(2, 32, 8); ROW(2, 128, 8); ROW(2, 16, 8); ROW(2, 16, 16);
  ROW(4, 64, 8); ROW(4, 64, 4); ROW(4, 32, 8); ROW(4, 32, 4); ROW(4, 16, 8); ROW(4, 16, 4); ROW(4, 128, 4);

  return 0;
}
