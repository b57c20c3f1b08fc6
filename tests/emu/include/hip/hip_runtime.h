// TEST INFRASTRUCTURE ONLY -- never shipped, never loaded by riders_amd.
//
// A tiny host-side emulator of the HIP execution model, used by tests/ to run
// the *same* kernel sources that hipcc compiles for gfx950 on the CPU of the
// build container (which has no GPU).  It exists so index math, LDS layouts,
// barrier placement and the MFMA fragment mapping assumed by the kernels can
// be checked against the oracle before GPU minutes are spent, and so the CPU
// build can be run under sanitizers (GPU ASan is not available on the pool).
//
// Model: one OS thread; every HIP thread of a block is a ucontext fiber;
// fibers switch only at __syncthreads() and at wave-collective operations
// (shuffles, MFMA).  Blocks run one after another.  wave = 64 lanes.
//
// The product library (riders_amd/csrc -> libriders_hip.so) is built by hipcc
// from the same sources with the real <hip/hip_runtime.h>; this directory is
// only put on the include path by tests/emu/build_emu.py.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>

#define RD_EMU 1

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef int hipError_t;
typedef void* hipStream_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline hipError_t hipPeekAtLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
// round-to-nearest single operations that the compiler must not contract into an FMA (the emulator build uses -ffp-contract=off)
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return 0; }
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2 };
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

using std::min;
using std::max;

namespace emu {

struct Wave {
  int alive = 0, arrived = 0, gen = 0;
  alignas(16) unsigned char buf[2][64][16];  // two 16-byte operands per lane
};
struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  int wave = 0, lane = 0;
};
struct State {
  ucontext_t sched;
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int cur = -1;
  int alive = 0, arrived = 0, gen = 0;
  const std::function<void()>* body = nullptr;
  dim3 block, grid, bid;
};
inline State& st() { static State s; return s; }
constexpr size_t kStack = 256 * 1024;

}  // namespace emu

// HIP builtin index variables (plain globals: single OS thread, set on every fiber switch-in).
inline dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace emu {

inline void yield() {
  State& s = st();
  Fiber& f = s.fibers[s.cur];
  swapcontext(&f.ctx, &s.sched);
}
inline int lane_id() { State& s = st(); return s.fibers[s.cur].lane; }
inline Wave& my_wave() { State& s = st(); return s.waves[s.fibers[s.cur].wave]; }

inline void block_sync() {
  State& s = st();
  int g = s.gen;
  if (++s.arrived >= s.alive) { s.arrived = 0; s.gen++; }
  else while (s.gen == g) yield();
}
inline void wave_sync() {
  Wave& w = my_wave();
  int g = w.gen;
  if (++w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
  else while (w.gen == g) yield();
}
inline void trampoline() {
  State& s = st();
  (*s.body)();
  Fiber& f = s.fibers[s.cur];
  f.done = true;
  // a finished thread no longer takes part in barriers
  s.alive--;
  if (s.alive > 0 && s.arrived >= s.alive) { s.arrived = 0; s.gen++; }
  Wave& w = s.waves[f.wave];
  w.alive--;
  if (w.alive > 0 && w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
  swapcontext(&f.ctx, &s.sched);
}

inline void run_block(const std::function<void()>& body) {
  State& s = st();
  unsigned n = s.block.x * s.block.y * s.block.z;
  if (s.fibers.size() < n) {
    size_t old = s.fibers.size();
    s.fibers.resize(n);
    for (size_t i = old; i < n; i++) s.fibers[i].stack = (char*)malloc(kStack);
  }
  unsigned nw = (n + 63) / 64;
  s.waves.assign(nw, Wave());
  s.body = &body;
  s.alive = n; s.arrived = 0; s.gen = 0;
  for (unsigned i = 0; i < n; i++) {
    Fiber& f = s.fibers[i];
    f.done = false;
    f.tid = dim3(i % s.block.x, (i / s.block.x) % s.block.y, i / (s.block.x * s.block.y));
    f.wave = i / 64; f.lane = i % 64;
    s.waves[f.wave].alive++;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = nullptr;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  unsigned remaining = n;
  long spins = 0;
  while (remaining) {
    unsigned progressed = 0;
    for (unsigned i = 0; i < n; i++) {
      Fiber& f = s.fibers[i];
      if (f.done) continue;
      s.cur = (int)i;
      threadIdx = f.tid;
      swapcontext(&s.sched, &f.ctx);
      if (f.done) { remaining--; progressed++; }
    }
    if (!progressed && ++spins > 100000000L) { fprintf(stderr, "emu: deadlock (divergent barrier?)\n"); abort(); }
  }
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  State& s = st();
  s.grid = grid; s.block = block;
  gridDim = grid; blockDim = block;
  for (unsigned z = 0; z < grid.z; z++)
    for (unsigned y = 0; y < grid.y; y++)
      for (unsigned x = 0; x < grid.x; x++) {
        blockIdx = dim3(x, y, z);
        run_block(body);
      }
}

template <typename T>
inline T shfl_generic(T v, int src_lane) {
  static_assert(sizeof(T) <= 16, "shfl payload");
  Wave& w = my_wave();
  int l = lane_id();
  memcpy(w.buf[0][l], &v, sizeof(T));
  wave_sync();
  T r;
  memcpy(&r, w.buf[0][src_lane & 63], sizeof(T));
  wave_sync();
  return r;
}

}  // namespace emu

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { emu::block_sync(); }

template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  int l = emu::lane_id();
  int base = l & ~(width - 1);
  return emu::shfl_generic(v, base + (src & (width - 1)));
}
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = emu::lane_id();
  return emu::shfl_generic(v, l ^ mask);
}
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = emu::lane_id();
  int src = l + (int)d;
  if ((src & ~(width - 1)) != (l & ~(width - 1))) src = l;
  return emu::shfl_generic(v, src);
}
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = emu::lane_id();
  int src = l - (int)d;
  if (src < 0 || (src & ~(width - 1)) != (l & ~(width - 1))) src = l;
  return emu::shfl_generic(v, src);
}

// wave vote: bit l of the result = predicate of lane l (wave-collective)
static inline unsigned long long __ballot(int pred) {
  emu::Wave& w = emu::my_wave();
  int l = emu::lane_id();
  w.buf[0][l][0] = pred ? 1 : 0;
  emu::wave_sync();
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) if (i < w.alive && w.buf[0][i][0]) m |= 1ull << i;
  emu::wave_sync();
  return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }

static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicMax(int* p, int v) { int o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicMin(unsigned* p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { int o = *p; if (v < o) *p = v; return o; }

#define __expf expf
#define __logf logf
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }

// ---- MFMA (wave-collective), lane/register maps per cdna_hip_programming.md section 3 ----
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef short emu_s16x8 __attribute__((ext_vector_type(8)));

static inline float emu_bf16_to_f32(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// D[i][j] = C[i][j] + sum_k A[i][k] B[k][j]; lane l supplies A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// lane l holds D[row=(l>>4)*4+r][col=l&15] in register r.
static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c) {
  emu::Wave& w = emu::my_wave();
  int l = emu::lane_id();
  memcpy(w.buf[0][l], &a, 4);
  memcpy(w.buf[1][l], &b, 4);
  emu::wave_sync();
  int j = l & 15;
  for (int r = 0; r < 4; r++) {
    int i = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; k++) {
      float av, bv;
      memcpy(&av, w.buf[0][i + 16 * k], 4);
      memcpy(&bv, w.buf[1][j + 16 * k], 4);
      acc = fmaf(av, bv, acc);
    }
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}
// gfx950 ds_read_b64_tr_b16, modelled on the mapping MEASURED on an MI355X (tools/probes/tr_b16_probe.hip, output committed as
// profiles/r01_tr_b16_probe.txt): within each 16-lane group, lane i receives out[j] = mem16[A_{4j + (i>>2)} + (i&3)], j = 0..3,
// where A_m is the (8-byte aligned) address lane m of the group supplied.
struct emu_u16x4 { unsigned short v[4]; };
static inline emu_u16x4 emu_ds_read_tr16_b64(const void* addr) {
  emu::Wave& w = emu::my_wave();
  int l = emu::lane_id();
  memcpy(w.buf[0][l], &addr, sizeof(addr));
  emu::wave_sync();
  emu_u16x4 o;
  int base = l & ~15, i = l & 15;
  for (int j = 0; j < 4; j++) {
    const unsigned short* a;
    memcpy(&a, w.buf[0][base + 4 * j + (i >> 2)], sizeof(a));
    o.v[j] = a[i & 3];
  }
  emu::wave_sync();
  return o;
}
// bf16: lane l supplies 8 consecutive k (k = 8*(l>>4)+e) of A row i=l&15 / B column j=l&15.
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
// v_mfma_f32_32x32x16_bf16: lane l supplies A[l & 31][8 (l >> 5) + e], B[8 (l >> 5) + e][l & 31]; D[(v & 3) + 8 (v >> 2) + 4 (l >> 5)][l & 31] in v
static inline emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c) {
  emu::Wave& w = emu::my_wave();
  int l = emu::lane_id();
  memcpy(w.buf[0][l], &a, 16);
  memcpy(w.buf[1][l], &b, 16);
  emu::wave_sync();
  int j = l & 31;
  for (int v = 0; v < 16; v++) {
    int i = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
    float acc = c[v];
    for (int g = 0; g < 2; g++) {
      unsigned short av[8], bv[8];
      memcpy(av, w.buf[0][i + 32 * g], 16);
      memcpy(bv, w.buf[1][j + 32 * g], 16);
      for (int e = 0; e < 8; e++) acc += emu_bf16_to_f32(av[e]) * emu_bf16_to_f32(bv[e]);
    }
    c[v] = acc;
  }
  emu::wave_sync();
  return c;
}
static inline float emu_f16_to_f32(unsigned short h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
static inline emu_f32x4 emu_mfma_f32_16x16x32_f16(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c) {
  emu::Wave& w = emu::my_wave();
  int l = emu::lane_id();
  memcpy(w.buf[0][l], &a, 16);
  memcpy(w.buf[1][l], &b, 16);
  emu::wave_sync();
  int j = l & 15;
  for (int r = 0; r < 4; r++) {
    int i = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int g = 0; g < 4; g++) {
      unsigned short av[8], bv[8];
      memcpy(av, w.buf[0][i + 16 * g], 16);
      memcpy(bv, w.buf[1][j + 16 * g], 16);
      for (int e = 0; e < 8; e++) acc += emu_f16_to_f32(av[e]) * emu_f16_to_f32(bv[e]);
    }
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}
static inline emu_f32x16 emu_mfma_f32_32x32x16_f16(emu_s16x8 a, emu_s16x8 b, emu_f32x16 c) {
  emu::Wave& w = emu::my_wave();
  int l = emu::lane_id();
  memcpy(w.buf[0][l], &a, 16);
  memcpy(w.buf[1][l], &b, 16);
  emu::wave_sync();
  int j = l & 31;
  for (int v = 0; v < 16; v++) {
    int i = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5);
    float acc = c[v];
    for (int g = 0; g < 2; g++) {
      unsigned short av[8], bv[8];
      memcpy(av, w.buf[0][i + 32 * g], 16);
      memcpy(bv, w.buf[1][j + 32 * g], 16);
      for (int e = 0; e < 8; e++) acc += emu_f16_to_f32(av[e]) * emu_f16_to_f32(bv[e]);
    }
    c[v] = acc;
  }
  emu::wave_sync();
  return c;
}
// global_load_lds_dwordx4: lane l's 16 bytes land at (wave-uniform) base + 16 l; executed immediately (the model has no asynchrony)
static inline void emu_global_load_lds16(const void* gsrc, void* lds_wave_base) {
  memcpy((char*)lds_wave_base + 16 * emu::lane_id(), gsrc, 16);
}

static inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_s16x8 a, emu_s16x8 b, emu_f32x4 c) {
  emu::Wave& w = emu::my_wave();
  int l = emu::lane_id();
  memcpy(w.buf[0][l], &a, 16);
  memcpy(w.buf[1][l], &b, 16);
  emu::wave_sync();
  int j = l & 15;
  for (int r = 0; r < 4; r++) {
    int i = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int g = 0; g < 4; g++) {
      unsigned short av[8], bv[8];
      memcpy(av, w.buf[0][i + 16 * g], 16);
      memcpy(bv, w.buf[1][j + 16 * g], 16);
      for (int e = 0; e < 8; e++) acc += emu_bf16_to_f32(av[e]) * emu_bf16_to_f32(bv[e]);
    }
    c[r] = acc;
  }
  emu::wave_sync();
  return c;
}
