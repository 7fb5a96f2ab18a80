// developer probe: wave_sum32 (omk_platform.h) against a plain sum over the 64 lanes, per value
#include <cstdio>
#include <hip/hip_runtime.h>
#include "../../omnimamba_amd/csrc/omk_platform.h"
__global__ void k(float* out) {
  float v[32];
  const int lane = threadIdx.x;
  for (int i = 0; i < 32; i++) v[i] = (float)(lane * 100 + i);   // total of value i over lanes = 100 * 2016 + 64 i
  omk::wave_sum32(v);
  out[lane] = v[0];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4);
  k<<<1, 64>>>(d);
  float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    const float want = 100.f * 2016.f + 64.f * (l >> 1);
    if (h[l] != want) { if (bad < 12) printf("lane %2d: got %.0f want %.0f (diff %.0f)\n", l, h[l], want, h[l] - want); bad++; }
  }
  printf("%d lanes wrong\n", bad);
  return 0;
}
