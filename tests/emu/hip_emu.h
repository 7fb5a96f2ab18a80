// hip_emu.h -- a minimal SIMT emulator used ONLY by the CPU test-suite (tests/emu).
//
// Purpose: this container has no GPU and a gpurun round trip costs minutes, so the HIP sources in
// omnimamba_amd/csrc are written against the thin portability layer omk_platform.h and can also be
// compiled for the host (clang++ -DOMK_EMU).  Every GPU thread becomes a ucontext fiber; the cross-lane
// primitives the kernels use (block barrier, wave shuffles, ballot, MFMA, LDS transpose-read) are
// rendezvous points where the last-arriving lane computes the result for the whole wave from the
// documented gfx950 lane layouts (cdna_hip_programming.md section 3; checked on hardware by
// tools/probe/probe.hip).  It is test infrastructure: the product library never links it.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }

namespace emu {

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

constexpr int WAVE = 64;
constexpr size_t STACK_BYTES = 256 * 1024;

struct Wave {
  int alive = 0, arrived = 0, gen = 0;
  void (*op)(Wave&) = nullptr;
  // staging
  uint64_t in64[WAVE];
  uint64_t out64[WAVE];
  int src[WAVE];
  uint16_t a16[WAVE][8], b16[WAVE][8];
  float c32[WAVE][16];
  float d32[WAVE][16];
  uint16_t tr_in[WAVE][4], tr_out[WAVE][4];
  int iparam = 0;
};

struct Fiber {
  ucontext_t uc;
  char* stack = nullptr;
  int tid = 0;
  bool done = false;
  dim3 tIdx;
};

struct Block {
  std::vector<Fiber> fib;
  std::vector<Wave> waves;
  int alive = 0, arrived = 0, gen = 0;
  ucontext_t sched;
  Fiber* cur = nullptr;
  char* dyn_smem = nullptr;
  dim3 bIdx, bDim, gDim;
  std::function<void()> body;
};

inline Block*& blk() {
  static Block* b = nullptr;
  return b;
}
inline Fiber& cur() { return *blk()->cur; }
inline Wave& cur_wave() { return blk()->waves[cur().tid / WAVE]; }
inline int lane_id() { return cur().tid % WAVE; }

inline void yield() { swapcontext(&blk()->cur->uc, &blk()->sched); }

inline void block_barrier() {
  Block& b = *blk();
  int g = b.gen;
  if (++b.arrived == b.alive) {
    b.arrived = 0;
    b.gen++;
  } else {
    while (b.gen == g) yield();
  }
}

inline void wave_collective(void (*op)(Wave&)) {
  Wave& w = cur_wave();
  int g = w.gen;
  w.op = op;
  if (++w.arrived == w.alive) {
    if (op) op(w);
    w.arrived = 0;
    w.gen++;
  } else {
    while (w.gen == g) yield();
  }
}

inline void fiber_entry() {
  Block& b = *blk();
  b.body();
  Fiber& f = *b.cur;
  f.done = true;
  Wave& w = b.waves[f.tid / WAVE];
  b.alive--;
  w.alive--;
  if (b.alive > 0 && b.arrived == b.alive) { b.arrived = 0; b.gen++; }
  if (w.alive > 0 && w.arrived == w.alive) { if (w.op) w.op(w); w.arrived = 0; w.gen++; }
  swapcontext(&f.uc, &b.sched);
}

inline void run_block(Block& b) {
  blk() = &b;
  int n = (int)b.fib.size();
  for (int i = 0; i < n; i++) {
    Fiber& f = b.fib[i];
    getcontext(&f.uc);
    f.uc.uc_stack.ss_sp = f.stack;
    f.uc.uc_stack.ss_size = STACK_BYTES;
    f.uc.uc_link = &b.sched;
    f.done = false;
    makecontext(&f.uc, (void (*)())fiber_entry, 0);
  }
  int remaining = n;
  long spins = 0;
  while (remaining > 0) {
    int progressed = 0;
    for (int i = 0; i < n; i++) {
      Fiber& f = b.fib[i];
      if (f.done) continue;
      b.cur = &f;
      swapcontext(&b.sched, &f.uc);
      if (f.done) { remaining--; progressed++; }
    }
    if (++spins > 50000000L) { fprintf(stderr, "hip_emu: deadlock (divergent collective?)\n"); abort(); }
  }
}

// launch: grid/block are dim3; body invokes the kernel function with its arguments.
inline void launch(dim3 grid, dim3 block, size_t smem_bytes, std::function<void()> body) {
  int nthreads = block.x * block.y * block.z;
  static std::vector<char*> stacks;
  while ((int)stacks.size() < nthreads) stacks.push_back((char*)malloc(STACK_BYTES));
  std::vector<char> smem(smem_bytes + 64);
  Block b;
  b.body = body;
  b.bDim = block;
  b.gDim = grid;
  b.dyn_smem = (char*)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
  b.fib.resize(nthreads);
  int nw = (nthreads + WAVE - 1) / WAVE;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        b.bIdx = dim3(bx, by, bz);
        // LDS holds garbage on the GPU: poison the dynamic segment (NaN as f32 / bf16) so that a kernel reading LDS it
        // never wrote fails here too instead of silently seeing zeros
        memset(smem.data(), 0xFF, smem.size());
        b.alive = nthreads;
        b.arrived = 0;
        b.gen = 0;
        b.waves.assign(nw, Wave());
        for (int t = 0; t < nthreads; t++) {
          Fiber& f = b.fib[t];
          f.stack = stacks[t];
          f.tid = t;
          f.tIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
          b.waves[t / WAVE].alive++;
        }
        run_block(b);
      }
  blk() = nullptr;
}

// ---------------------------------------------------------------- bf16 helpers
inline float bf2f(uint16_t h) {
  uint32_t u = (uint32_t)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// ---------------------------------------------------------------- collectives
inline void op_shfl(Wave& w) {
  for (int l = 0; l < WAVE; l++) {
    int s = w.src[l];
    w.out64[l] = (s >= 0 && s < WAVE) ? w.in64[s] : w.in64[l];
  }
}
template <class T> inline T shfl_generic(T v, int srclane) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  Wave& w = cur_wave();
  int l = lane_id();
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  w.in64[l] = bits;
  w.src[l] = srclane;
  wave_collective(op_shfl);
  T r;
  memcpy(&r, &w.out64[l], sizeof(T));
  return r;
}
inline void op_ballot(Wave& w) {
  uint64_t m = 0;
  for (int l = 0; l < WAVE; l++)
    if (w.in64[l]) m |= (1ull << l);
  for (int l = 0; l < WAVE; l++) w.out64[l] = m;
}
inline uint64_t ballot(int pred) {
  Wave& w = cur_wave();
  int l = lane_id();
  w.in64[l] = pred ? 1 : 0;
  wave_collective(op_ballot);
  return w.out64[l];
}

// v_mfma_f32_32x32x16_bf16: A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31],
// C/D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
inline void op_mfma32(Wave& w) {
  static float A[32][16], B[16][32], C[32][32];
  for (int l = 0; l < WAVE; l++)
    for (int e = 0; e < 8; e++) {
      A[l & 31][8 * (l >> 5) + e] = bf2f(w.a16[l][e]);
      B[8 * (l >> 5) + e][l & 31] = bf2f(w.b16[l][e]);
    }
  for (int l = 0; l < WAVE; l++)
    for (int r = 0; r < 16; r++) C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31] = w.c32[l][r];
  for (int i = 0; i < 32; i++)
    for (int j = 0; j < 32; j++) {
      float acc = C[i][j];
      for (int k = 0; k < 16; k++) acc = fmaf(A[i][k], B[k][j], acc);
      C[i][j] = acc;
    }
  for (int l = 0; l < WAVE; l++)
    for (int r = 0; r < 16; r++) w.d32[l][r] = C[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][l & 31];
}
// v_mfma_f32_16x16x32_bf16: A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15], C/D[row=(l>>4)*4+r][col=l&15]
inline void op_mfma16(Wave& w) {
  static float A[16][32], B[32][16], C[16][16];
  for (int l = 0; l < WAVE; l++)
    for (int e = 0; e < 8; e++) {
      A[l & 15][8 * (l >> 4) + e] = bf2f(w.a16[l][e]);
      B[8 * (l >> 4) + e][l & 15] = bf2f(w.b16[l][e]);
    }
  for (int l = 0; l < WAVE; l++)
    for (int r = 0; r < 4; r++) C[(l >> 4) * 4 + r][l & 15] = w.c32[l][r];
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      float acc = C[i][j];
      for (int k = 0; k < 32; k++) acc = fmaf(A[i][k], B[k][j], acc);
      C[i][j] = acc;
    }
  for (int l = 0; l < WAVE; l++)
    for (int r = 0; r < 4; r++) w.d32[l][r] = C[(l >> 4) * 4 + r][l & 15];
}
// v_mfma_f32_16x16x4_f32 (fp32 operands, one block): A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C/D[row=(l>>4)*4+r][col=l&15];
// the operands arrive in c32[l][4] (A) and c32[l][5] (B)
inline void op_mfma16_f32(Wave& w) {
  static float A[16][4], B[4][16], C[16][16];
  for (int l = 0; l < WAVE; l++) { A[l & 15][l >> 4] = w.c32[l][4]; B[l >> 4][l & 15] = w.c32[l][5]; }
  for (int l = 0; l < WAVE; l++)
    for (int r = 0; r < 4; r++) C[(l >> 4) * 4 + r][l & 15] = w.c32[l][r];
  for (int i = 0; i < 16; i++)
    for (int j = 0; j < 16; j++) {
      float acc = C[i][j];
      for (int k = 0; k < 4; k++) acc = fmaf(A[i][k], B[k][j], acc);
      C[i][j] = acc;
    }
  for (int l = 0; l < WAVE; l++)
    for (int r = 0; r < 4; r++) w.d32[l][r] = C[(l >> 4) * 4 + r][l & 15];
}
// ds_read_b64_tr_b16: within each 16-lane group, lane t fetches 4 x 16 bit at its own address (M_t);
// result lane t, element j = M_{4j + t/4}[t % 4]  (column t of the 4x16 block the group fetched).
inline void op_tr16(Wave& w) {
  for (int l = 0; l < WAVE; l++) {
    int t = l & 15, g = l >> 4;
    for (int j = 0; j < 4; j++) w.tr_out[l][j] = w.tr_in[g * 16 + 4 * j + (t >> 2)][t & 3];
  }
}

}  // namespace emu
