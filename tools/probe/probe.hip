// Hardware probe for MI355X (gfx950): pins the facts the kernels in omnimamba_amd/csrc rely on.
//   1. MFMA 32x32x16 / 16x16x32 bf16 operand + accumulator lane layouts (hypothesis check, asymmetric data)
//   2. ds_read_b64_tr_b16 semantics (dump)
//   3. fp32 global atomicAdd throughput in the cross-head dB/dC reduction pattern
//   4. HBM streaming copy bandwidth
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics probe.hip -o probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef short v4s __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// A: [32][16] row-major, B: [16][32] row-major, D: [32][32]
__global__ void mfma32(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  v8bf a, b;
  for (int e = 0; e < 8; e++) {
    a[e] = (__bf16)A[(l & 31) * 16 + 8 * (l >> 5) + e];
    b[e] = (__bf16)B[(8 * (l >> 5) + e) * 32 + (l & 31)];
  }
  v16f c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    D[row * 32 + col] = c[r];
  }
}
// A: [16][32], B: [32][16], D: [16][16]
__global__ void mfma16(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  v8bf a, b;
  for (int e = 0; e < 8; e++) {
    a[e] = (__bf16)A[(l & 15) * 32 + 8 * (l >> 4) + e];
    b[e] = (__bf16)B[(8 * (l >> 4) + e) * 16 + (l & 15)];
  }
  v4f c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) {
    int row = (l >> 4) * 4 + r, col = l & 15;
    D[row * 16 + col] = c[r];
  }
}
// permuted-k check: accumulator tile used directly as the B operand of the next MFMA (the "P.V" trick)
// S = X(32x16) * Y(16x32) (32x32 acc), then Z = W(32x32, k permuted) * S  with S rows as contraction index.
__global__ void mfma32_chain(const float* X, const float* Y, const float* W, float* Z) {
  int l = threadIdx.x;
  v8bf a, b;
  for (int e = 0; e < 8; e++) {
    a[e] = (__bf16)X[(l & 31) * 16 + 8 * (l >> 5) + e];
    b[e] = (__bf16)Y[(8 * (l >> 5) + e) * 32 + (l & 31)];
  }
  v16f s = {0};
  s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, s, 0, 0, 0);   // s[r] = S[row(r,l)][l&31]
  v16f z = {0};
  int h = l >> 5;
  for (int ks = 0; ks < 2; ks++) {     // two K-steps of 16 rows of S: rows 16*ks + {4h+r, 8+4h+r}
    v8bf sb, wa;
    for (int e = 0; e < 8; e++) {
      int reg = 8 * ks + e;            // acc regs 8ks..8ks+7 -> rows (e&3) + 8*(reg>>2) + 4h
      sb[e] = (__bf16)s[reg];
      int krow = (reg & 3) + 8 * (reg >> 2) + 4 * h;
      wa[e] = (__bf16)W[(l & 31) * 32 + krow];
    }
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, sb, z, 0, 0, 0);
  }
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
    Z[row * 32 + col] = z[r];
  }
}

__global__ void trread(short* out, int pattern) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x, t = l & 15, g = l >> 4;
  int off;
  if (pattern == 0) off = l * 4;                                            // contiguous
  else if (pattern == 1) off = g * 1024 + (t >> 2) * 64 + (t & 3) * 4;      // 4 rows x 16 cols, row stride 64 elems
  else off = (g >> 1) * 4 * 72 + (t >> 2) * 72 + (g & 1) * 16 + (t & 3) * 4; // B-operand use: row stride 72, groups = (half, col-half)
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + off));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = r[j];
}

// atomics: WG w -> batch b = w / pairs, all `pairs` WGs of a batch add into the same [T][128] region.
__global__ void atom_shared(float* buf, int T, int pairs) {
  int b = blockIdx.x / pairs;
  float* base = buf + (size_t)b * T * 128;
  int lane = threadIdx.x & 127, sub = threadIdx.x >> 7;
  for (int t = sub; t < T; t += 2) unsafeAtomicAdd(base + (size_t)t * 128 + lane, 1.0f);
}
__global__ void atom_private(float* buf, int T) {
  float* base = buf + (size_t)blockIdx.x * T * 128;
  int lane = threadIdx.x & 127, sub = threadIdx.x >> 7;
  for (int t = sub; t < T; t += 2) unsafeAtomicAdd(base + (size_t)t * 128 + lane, 1.0f);
}
__global__ void store_private(float* buf, int T) {
  float* base = buf + (size_t)blockIdx.x * T * 128;
  int lane = threadIdx.x & 127, sub = threadIdx.x >> 7;
  for (int t = sub; t < T; t += 2) base[(size_t)t * 128 + lane] = 1.0f;
}
__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = a[i];
}

template <class F> float timeit(F f, int iters = 5) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; i++) f();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch=%s CUs=%d clock=%d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  srand(1);
  auto rnd = [](int n) { std::vector<float> v(n); for (auto& x : v) x = (float)((rand() % 15) - 7); return v; };
  {  // 32x32x16
    auto A = rnd(32 * 16), B = rnd(16 * 32);
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    mfma32<<<1, 64>>>(dA, dB, dD); std::vector<float> D(1024); CK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float r = 0; for (int k = 0; k < 16; k++) r += A[i * 16 + k] * B[k * 32 + j]; if (r != D[i * 32 + j]) bad++; }
    printf("MFMA32x32x16 layout hypothesis: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad);
  }
  {  // 16x16x32
    auto A = rnd(16 * 32), B = rnd(32 * 16);
    float *dA, *dB, *dD; CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, 256 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
    mfma16<<<1, 64>>>(dA, dB, dD); std::vector<float> D(256); CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float r = 0; for (int k = 0; k < 32; k++) r += A[i * 32 + k] * B[k * 16 + j]; if (r != D[i * 16 + j]) bad++; }
    printf("MFMA16x16x32 layout hypothesis: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad);
  }
  {  // chain
    auto X = rnd(32 * 16), Y = rnd(16 * 32), W = rnd(32 * 32);
    for (auto& x : X) x = (float)((int)x % 2); for (auto& y : Y) y = (float)((int)y % 3);  // keep S exactly representable in bf16
    float *dX, *dY, *dW, *dZ; CK(hipMalloc(&dX, 2048)); CK(hipMalloc(&dY, 2048)); CK(hipMalloc(&dW, 4096)); CK(hipMalloc(&dZ, 4096));
    CK(hipMemcpy(dX, X.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dY, Y.data(), 2048, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, W.data(), 4096, hipMemcpyHostToDevice));
    mfma32_chain<<<1, 64>>>(dX, dY, dW, dZ); std::vector<float> Z(1024); CK(hipMemcpy(Z.data(), dZ, 4096, hipMemcpyDeviceToHost));
    std::vector<float> S(1024, 0.f);
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) for (int k = 0; k < 16; k++) S[i * 32 + j] += X[i * 16 + k] * Y[k * 32 + j];
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float r = 0; for (int k = 0; k < 32; k++) r += W[i * 32 + k] * S[k * 32 + j]; if (r != Z[i * 32 + j]) bad++; }
    printf("MFMA32 accumulator-as-B-operand (permuted k) chain: %s (bad=%d)\n", bad ? "FAIL" : "PASS", bad);
  }
  for (int pat = 0; pat < 3; pat++) {
    short* d; CK(hipMalloc(&d, 512)); trread<<<1, 64>>>(d, pat); short h[256]; CK(hipMemcpy(h, d, 512, hipMemcpyDeviceToHost));
    printf("TRREAD pattern %d (lane: 4 values):\n", pat);
    for (int l = 0; l < 64; l++) printf("  l%02d: %5d %5d %5d %5d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l % 4 == 3) ? "\n" : "");
    // hypothesis H1: lane t of 16-lane group, elem j = M_{4j + t/4}[t%4], M_t = 4 shorts at lane t's address
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
      int t = l & 15, g = l >> 4; int src = g * 16 + 4 * j + (t >> 2); int st = src & 15, sg = src >> 4;
      int off = pat == 0 ? src * 4 : pat == 1 ? sg * 1024 + (st >> 2) * 64 + (st & 3) * 4 : (sg >> 1) * 4 * 72 + (st >> 2) * 72 + (sg & 1) * 16 + (st & 3) * 4;
      if (h[l * 4 + j] != (short)(off + (t & 3))) bad++;
    }
    printf("TRREAD pattern %d hypothesis H1: %s (bad=%d)\n", pat, bad ? "FAIL" : "PASS", bad);
  }
  {  // atomics
    int T = 2048, pairs = 32, nb = 8;
    size_t nshared = (size_t)nb * T * 128, npriv = (size_t)nb * pairs * T * 128;
    float *bs, *bp; CK(hipMalloc(&bs, nshared * 4)); CK(hipMalloc(&bp, npriv * 4)); CK(hipMemset(bs, 0, nshared * 4)); CK(hipMemset(bp, 0, npriv * 4));
    double nat = (double)nb * pairs * T * 128;
    float ms = timeit([&] { atom_shared<<<nb * pairs, 256>>>(bs, T, pairs); });
    printf("ATOMIC shared (32 WGs -> same [T][128] fp32 region, 256 WGs): %.3f ms, %.1f G atomics/s\n", ms, nat / ms / 1e6);
    ms = timeit([&] { atom_private<<<nb * pairs, 256>>>(bp, T); });
    printf("ATOMIC private (no contention): %.3f ms, %.1f G atomics/s\n", ms, nat / ms / 1e6);
    ms = timeit([&] { store_private<<<nb * pairs, 256>>>(bp, T); });
    printf("STORE private (plain stores, same pattern): %.3f ms, %.1f GB/s\n", ms, nat * 4 / ms / 1e6);
    std::vector<float> h(16); CK(hipMemcpy(h.data(), bs, 64, hipMemcpyDeviceToHost));
    printf("  shared[0]=%g (expect %d)\n", h[0], pairs * 6);
  }
  {  // HBM copy
    size_t n = (size_t)1 << 26;  // 64M float4 = 1 GiB each
    float4 *a, *b; CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16)); CK(hipMemset(a, 1, n * 16));
    float ms = timeit([&] { copy4<<<256 * 8, 256>>>(a, b, n); });
    printf("HBM copy float4 1GiB->1GiB: %.3f ms, %.1f GB/s (read+write)\n", ms, 2.0 * n * 16 / ms / 1e6);
  }
  return 0;
}
