// Developer microbenchmark (MI355X): what does a global store instruction cost a CU, by bytes per lane and by address pattern?
// One 512-thread workgroup per CU (the scan kernels' shape); every wave issues `n` stores per iteration into its own region and
// pads the iteration with MFMAs so that stores are issued at about the density of the scan (one per few hundred cycles).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_cost.hip -o tools/ubench/store_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// PAT 0: no store.  1: 16 B / lane, lanes contiguous (1 KB).  2: 16 B / lane as 16 rows x 64 B, row stride 256 B (the state images).
// 3: 8 B / lane as 16 rows x 32 B, row stride 8 KB (the output rows of one wave).  4: 8 B / lane contiguous (512 B).
// 5: pattern 2 but all iterations onto the same 4 KB (L2 resident).  6: pattern 3, rows contiguous in 128 B lines across the 4 waves
template <int PAT>
__global__ __launch_bounds__(512) void k(char* out, int iters, int nst, size_t region) {
  f32x4 acc[8];
  s16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); acc[i] = f32x4{0, 0, 0, 0}; }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, t16 = lane & 15, g16 = lane >> 4;
  char* base = out + (size_t)(blockIdx.x * 8 + w) * region;
  u32x4 v = {(uint32_t)lane, 1u, 2u, 3u};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    for (int s = 0; s < nst; s++) {
      const size_t o = ((size_t)it * nst + s) * 4096 % (region - 65536);
      if (PAT == 1) *reinterpret_cast<u32x4*>(base + o + 16 * lane) = v;
      if (PAT == 2) *reinterpret_cast<u32x4*>(base + o + 256 * t16 + 16 * g16 + 64 * (s & 3)) = v;
      if (PAT == 3) *reinterpret_cast<u32x2*>(base + (o * 16) % (region - 262144) + 8192 * t16 + 8 * g16) = u32x2{v[0], v[1]};
      if (PAT == 4) *reinterpret_cast<u32x2*>(base + o + 8 * lane) = u32x2{v[0], v[1]};
      if (PAT == 5) *reinterpret_cast<u32x4*>(base + 256 * t16 + 16 * g16 + 64 * (s & 3)) = v;
      if (PAT == 6) *reinterpret_cast<u32x2*>(out + (size_t)blockIdx.x * 8 * region + (o * 16) % (region - 262144) + 8192 * t16 + 32 * w + 8 * g16) = u32x2{v[0], v[1]};
      v[0] += 1;
    }
  }
  float sum = 0.f;
  for (int i = 0; i < 8; i++) sum += acc[i][0];
  if (sum == 123.456f) out[0] = 1;
}

template <int PAT>
void run(const char* name, char* out, size_t region) {
  const int iters = 4000;
  for (int nst : {1, 2, 4}) {
    k<PAT><<<256, 512>>>(out, 10, nst, region);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<PAT><<<256, 512>>>(out, iters, nst, region);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-64s %d stores / wave / iteration: %7.1f ns / iteration\n", name, nst, ms * 1e6 / iters);
    if (PAT == 0) break;
  }
}

int main() {
  const size_t region = 4u << 20;   // per wave
  char* out;
  if (hipMalloc(&out, region * 8 * 256) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
  run<0>("no stores (8 MFMA per wave = 16 per SIMD)", out, region);
  run<1>("16 B / lane, contiguous 1 KB", out, region);
  run<2>("16 B / lane, 16 rows x 64 B (state image)", out, region);
  run<5>("the same onto one 4 KB spot per wave", out, region);
  run<4>("8 B / lane, contiguous 512 B", out, region);
  run<3>("8 B / lane, 16 rows x 32 B, one wave per 32 B piece", out, region);
  run<6>("8 B / lane, 16 rows x 32 B, four waves fill 128 B lines", out, region);
  return 0;
}
