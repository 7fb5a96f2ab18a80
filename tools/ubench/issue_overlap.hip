// Developer microbenchmark (MI355X): do the matrix pipe and the VALU of one SIMD overlap -- inside ONE wave (independent instructions
// interleaved) and across the TWO waves of a SIMD?  Decides between "more waves" and "one fat wave per SIMD" for the scan kernels.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_overlap.hip -o gpurun_out/issue_overlap && gpurun_out/issue_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define MFMA(i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b))
#define VAL(j) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(c))
#define VMUL(j) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(y[j]) : "v"(c[0]))
#define VCVT(j) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(z[j]) : "v"(y[j]), "v"(c[1]))
#define VADD(j) asm volatile("v_add_u32 %0, %0, %1" : "+v"(z[j]) : "v"(z[(j + 1) & 15]))
#define LDS(j) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j]) : "v"(laddr) : "memory")

// MODE 0: 16 MFMA; 1: 48 VALU; 2: 16 x (MFMA + 3 VALU) interleaved; 3: 16 MFMA then 48 VALU (blocks); 4: 32 VALU;
// 5: 16 x (MFMA + 2 VALU); 6: 16 x (MFMA + 1 ds_read_b128); 7: 16 ds_read_b128; 8: 16 x (MFMA + 2 VALU + 1 ds_read_b128)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, uint64_t* cyc, int iters) {
  __shared__ f32x4 lds[1024];
  f32x4 acc[16];
  f32x2 x[16];
  f32x4 q[16];
  float y[16];
  uint32_t z[16];
  s16x8 a, b;
  f32x2 c = {1.0001f, 0.9999f};
  for (int i = 0; i < 8; i++) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
  for (int i = 0; i < 16; i++) { acc[i] = f32x4{0, 0, 0, 0}; x[i] = f32x2{1.f + i, 2.f + i}; q[i] = f32x4{0, 0, 0, 0}; y[i] = 1.f + i; z[i] = i + threadIdx.x; }
  lds[threadIdx.x] = f32x4{1, 2, 3, 4};
  __syncthreads();
  const uint32_t laddr = (uint32_t)(uintptr_t)&lds[threadIdx.x & 63];
  uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) MFMA(i);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 16; j++) VAL(j);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; i++) { MFMA(i); VAL(i); VAL((i + 5) & 15); VAL((i + 10) & 15); }
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; i++) MFMA(i);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 16; j++) VAL(j);
    } else if (MODE == 4) {
#pragma unroll
      for (int r = 0; r < 2; r++)
#pragma unroll
        for (int j = 0; j < 16; j++) VAL(j);
    } else if (MODE == 5) {
#pragma unroll
      for (int i = 0; i < 16; i++) { MFMA(i); VAL(i); VAL((i + 8) & 15); }
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; i++) { MFMA(i); LDS(i); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 16; i++) LDS(i);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if (MODE == 9) {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 16; j++) VMUL(j);
    } else if (MODE == 10) {
#pragma unroll
      for (int i = 0; i < 16; i++) MFMA(i);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 16; j++) VMUL(j);
    } else if (MODE == 11) {
#pragma unroll
      for (int i = 0; i < 16; i++) { MFMA(i); VMUL(i); VMUL((i + 5) & 15); VMUL((i + 10) & 15); }
    } else if (MODE == 12) {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int j = 0; j < 16; j++) VADD(j);
    } else if (MODE == 13) {
#pragma unroll
      for (int i = 0; i < 16; i++) { MFMA(i); VADD(i); VADD((i + 5) & 15); VADD((i + 10) & 15); }
    } else if (MODE == 14) {
#pragma unroll
      for (int i = 0; i < 16; i++) { MFMA(i); VCVT(i); VCVT((i + 5) & 15); VCVT((i + 10) & 15); }
    } else if (MODE == 15) {
#pragma unroll
      for (int i = 0; i < 16; i += 4) { MFMA(i); MFMA(i + 1); MFMA(i + 2); MFMA(i + 3);
#pragma unroll
        for (int j = 0; j < 12; j++) VMUL((3 * i + j) & 15); }
    } else if (MODE == 8) {
#pragma unroll
      for (int i = 0; i < 16; i++) { MFMA(i); VAL(i); LDS(i); VAL((i + 8) & 15); }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  uint64_t t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][3] + x[i][0] + x[i][1] + q[i][0] + y[i] + (float)z[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, uint64_t* cyc) {
  const int iters = 2000;
  for (int threads : {256, 512}) {
    k<MODE><<<256, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    uint64_t h = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE><<<256, threads>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s %d waves/SIMD: %8.1f shader cycles per iteration per wave-slot  (%.3f ms -> %.1f ns / iteration)\n", name, threads / 256,
           (double)h / iters, ms, ms * 1e6 / iters);
  }
}

int main() {
  float* out; uint64_t* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 8);
  run<0>("16 MFMA 16x16x32", out, cyc);
  run<1>("48 v_pk_mul_f32", out, cyc);
  run<4>("32 v_pk_mul_f32", out, cyc);
  run<2>("16 x (MFMA + 3 VALU) interleaved", out, cyc);
  run<5>("16 x (MFMA + 2 VALU) interleaved", out, cyc);
  run<3>("16 MFMA, then 48 VALU", out, cyc);
  run<7>("16 ds_read_b128", out, cyc);
  run<6>("16 x (MFMA + ds_read_b128)", out, cyc);
  run<8>("16 x (MFMA + 2 VALU + ds_read_b128)", out, cyc);
  run<9>("48 v_mul_f32", out, cyc);
  run<10>("16 MFMA, then 48 v_mul_f32", out, cyc);
  run<11>("16 x (MFMA + 3 v_mul_f32)", out, cyc);
  run<15>("4 x (4 MFMA + 12 v_mul_f32)", out, cyc);
  run<12>("48 v_add_u32", out, cyc);
  run<13>("16 x (MFMA + 3 v_add_u32)", out, cyc);
  run<14>("16 x (MFMA + 3 v_cvt_pk_bf16_f32)", out, cyc);
  return 0;
}
