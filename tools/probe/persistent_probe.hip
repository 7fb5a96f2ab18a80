// Probe: one PERSISTENT kernel with grid barriers against a hipGraph of launches, for the decode step of the 1.3B model.
// Per "layer": y1 (8512) = W1 (8512 x 2048) x ; z (4096) = f(y1) (stands for conv update + state update) ; x' (2048) = W2 (2048 x 4096) z.
// 48 layers, fp32 weights (70 MB + 33.5 MB per layer, each read once).  Data really flows through global memory between the phases,
// so the barrier carries an agent-scope release / acquire (L2 write-back + invalidate across the 8 XCDs).
//   graph      : 3 launches per layer, replayed from a hipGraph (what generation.StepGraph does today: ~4.5 us per launch)
//   persistent : gridDim = CUs x k workgroups, all resident, sense-reversing barrier on one global counter
//   persistent + prefetch : the first row block of the NEXT phase's weights is requested before the barrier is entered
// Build: hipcc --offload-arch=gfx950 -O3 persistent_probe.hip -o persistent_probe ; run: ./persistent_probe [workgroups per CU]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int D = 2048, N1 = 8512, DI = 4096, NL = 48;

__device__ __forceinline__ float row_dot(const float* __restrict__ w, const float* __restrict__ x, int cols, int lane) {
  const f4* wr = reinterpret_cast<const f4*>(w);
  const f4* xr = reinterpret_cast<const f4*>(x);
  float acc = 0.f;
  for (int c = lane; c < cols / 4; c += 64) {
    const f4 a = __builtin_nontemporal_load(wr + c), b = xr[c];
    acc += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
  }
  for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
  return acc;
}

__global__ __launch_bounds__(256) void gemv(const float* __restrict__ W, const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
  for (int r = wave; r < rows; r += nw) {
    const float v = row_dot(W + (size_t)r * cols, x, cols, lane);
    if (lane == 0) y[r] = v * 1e-2f;
  }
}
__global__ void mid(const float* __restrict__ y1, float* __restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < DI) z[i] = tanhf(y1[i] + y1[i + DI]);
}

// sense-reversing grid barrier: count + generation in global memory, agent scope
__device__ __forceinline__ void grid_barrier(unsigned* count, unsigned* gen, unsigned nblocks) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned g = __hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();   // release: this workgroup's writes are visible device-wide
    if (__hip_atomic_fetch_add(count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1) {
      __hip_atomic_store(count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(gen, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // (bounded: a probe must not hang the box if some workgroup is not resident)
      for (unsigned spin = 0; spin < (1u << 24) && __hip_atomic_load(gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == g; spin++) __builtin_amdgcn_s_sleep(1);
    }
    __threadfence();   // acquire
  }
  __syncthreads();
}

template <int MODE>   // 0: barriers only (no work), 1: the layer work, 2: work + touch of the next phase's first rows before the barrier
__global__ __launch_bounds__(256) void persistent(const float* __restrict__ W1, const float* __restrict__ W2, float* x, float* y1, float* z,
                                                  unsigned* count, unsigned* gen, int layers, float* sink) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
  const unsigned nb = gridDim.x;
  float keep = 0.f;
  for (int l = 0; l < layers; l++) {
    const float* w1 = W1 + (size_t)l * N1 * D;
    const float* w2 = W2 + (size_t)l * D * DI;
    if (MODE) {
      for (int r = wave; r < N1; r += nw) {
        const float v = row_dot(w1 + (size_t)r * D, x, D, lane);
        if (lane == 0) y1[r] = v * 1e-2f;
      }
      if (MODE == 2 && wave < D) keep += __builtin_nontemporal_load(w2 + (size_t)wave * DI + lane * 4);   // first 1 KB of this wave's first W2 row
    }
    grid_barrier(count, gen, nb);
    if (MODE) {
      const int i = blockIdx.x * 256 + threadIdx.x;
      if (i < DI) z[i] = tanhf(y1[i] + y1[i + DI]);
    }
    grid_barrier(count, gen, nb);
    if (MODE) {
      for (int r = wave; r < D; r += nw) {
        const float v = row_dot(w2 + (size_t)r * DI, z, DI, lane);
        if (lane == 0) x[r] = v * 1e-2f;
      }
      if (MODE == 2 && l + 1 < layers && wave < N1) keep += __builtin_nontemporal_load(w1 + (size_t)N1 * D + (size_t)wave * D + lane * 4);
    }
    grid_barrier(count, gen, nb);
  }
  if (keep == 123.456f) *sink = keep;
}

int main(int argc, char** argv) {
  const int per_cu = argc > 1 ? atoi(argv[1]) : 1;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount, nb = cus * per_cu;
  float *W1, *W2, *x, *y1, *z, *sink;
  unsigned* bar;
  CK(hipMalloc(&W1, (size_t)NL * N1 * D * 4)); CK(hipMalloc(&W2, (size_t)NL * D * DI * 4));
  CK(hipMalloc(&x, D * 4)); CK(hipMalloc(&y1, N1 * 4)); CK(hipMalloc(&z, DI * 4)); CK(hipMalloc(&sink, 4)); CK(hipMalloc(&bar, 8));
  CK(hipMemset(W1, 0, (size_t)NL * N1 * D * 4)); CK(hipMemset(W2, 0, (size_t)NL * D * DI * 4));
  CK(hipMemset(x, 0, D * 4)); CK(hipMemset(bar, 0, 8));
  hipStream_t s;
  CK(hipStreamCreate(&s));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time_ms = [&](auto fn, int reps) {
    fn(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  // ---- graph of launches
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int l = 0; l < NL; l++) {
    gemv<<<cus * 4, 256, 0, s>>>(W1 + (size_t)l * N1 * D, x, y1, N1, D);
    mid<<<DI / 256, 256, 0, s>>>(y1, z);
    gemv<<<cus * 2, 256, 0, s>>>(W2 + (size_t)l * D * DI, z, x, D, DI);
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  const float tg = time_ms([&] { CK(hipGraphLaunch(ge, s)); }, 10);
  const double gb = (double)NL * ((double)N1 * D + (double)D * DI) * 4 / 1e9;
  printf("device: %d CUs; weights per step %.2f GB\n", cus, gb);
  printf("graph of 3 x 48 launches          : %8.3f ms per step  (%.2f TB/s)\n", tg, gb / tg);
  // ---- persistent
  const float t0 = time_ms([&] { persistent<0><<<nb, 256, 0, s>>>(W1, W2, x, y1, z, bar, bar + 1, NL, sink); }, 10);
  printf("persistent, %4d workgroups, barriers only (144): %8.3f ms  = %.2f us per barrier\n", nb, t0, t0 * 1e3 / (3 * NL));
  const float t1 = time_ms([&] { persistent<1><<<nb, 256, 0, s>>>(W1, W2, x, y1, z, bar, bar + 1, NL, sink); }, 10);
  printf("persistent, %4d workgroups, the layer work      : %8.3f ms per step  (%.2f TB/s)\n", nb, t1, gb / t1);
  const float t2 = time_ms([&] { persistent<2><<<nb, 256, 0, s>>>(W1, W2, x, y1, z, bar, bar + 1, NL, sink); }, 10);
  printf("persistent, %4d workgroups, + next-phase touch  : %8.3f ms per step  (%.2f TB/s)\n", nb, t2, gb / t2);
  return 0;
}
