// Shared device-side helpers for the riders_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Device lambdas that capture register arrays by reference MUST be inlined: an out-of-line closure call forces every captured
// array into scratch (or LDS via promote-alloca) -- seen as private_segment 48..80 bytes / +16 KB LDS on the conv kernels.
#define RD_INLINE_LAMBDA __attribute__((always_inline))

// Precision variant.  Every translation unit is compiled twice: as is (16-bit activations = bf16) and with -DRD_HALF_F16 (16-bit activations =
// IEEE fp16: conversions and MFMA opcodes change, nothing else).  The second build lives in namespace rd_f16 (the token `rd` is renamed), so
// both variants link into one library and rd_api.cpp picks the namespace from the dtype code (RD_BF16 / RD_F16).  The element type keeps
// its name bf16_t in both builds: it is "the 16-bit activation type of this build".
#ifdef RD_HALF_F16
#define rd rd_f16
#endif

// Launch-time sized LDS (hipLaunchKernelGGL's shmem argument).  The host emulator runs blocks one after the other: a static arena of the
// CU's full 160 KiB stands in.
#ifdef RD_EMU
#define RD_DYN_SMEM(name) static __attribute__((aligned(16))) unsigned char name[160 * 1024]
#define RD_WAVE_UNIFORM(x) (x)
#else
#define RD_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define RD_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

// the 16-bit element type as rocprofv3's kernel trace prints it in this build's instantiation names
#ifdef RD_HALF_F16
#define RD_T16_NAME "rd_f16::bf16_t"
#else
#define RD_T16_NAME "rd::bf16_t"
#endif

namespace rd {

// ---- element types -------------------------------------------------------------------------
// Activations are stored either as fp32 or as bf16 (raw uint16 bit patterns); every kernel
// computes in fp32.  dtype codes match RD_F32 / RD_BF16 in include/riders_hip.h.
struct bf16_t { unsigned short v; };

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#ifdef RD_HALF_F16
// ---- fp16 build: v_cvt_f32_f16 / v_cvt_f16_f32 (round to nearest even); the host emulator uses the compiler's _Float16
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  _Float16 x; __builtin_memcpy(&x, &h, 2);
  return (float)x;
}
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
  const _Float16 x = (_Float16)f;
  unsigned short h; __builtin_memcpy(&h, &x, 2);
  return h;
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}
// low / high 16-bit element of a 32-bit word -> float
__device__ __forceinline__ float half_lo_f32(unsigned w) { return bf16_to_f32((unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float half_hi_f32(unsigned w) { return bf16_to_f32((unsigned short)(w >> 16)); }
#else
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float(((unsigned)h) << 16);
}
// round-to-nearest-even, NaN preserved.  gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32, two values per instruction); the
// integer sequence (7 VALU instructions per value) is what the host emulator runs and what the epilogues used to spend their time on.
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
#ifdef RD_EMU
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
#else
  const __bf16 b = (__bf16)f;
  unsigned short h; __builtin_memcpy(&h, &b, 2);
  return h;
#endif
}
// two values -> one 32-bit word (lo in bits 0..15)
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
#ifdef RD_EMU
  return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
#else
  typedef __bf16 rd_bf2 __attribute__((ext_vector_type(2)));
  typedef float rd_f2 __attribute__((ext_vector_type(2)));
  const rd_f2 v = {lo, hi};
  const rd_bf2 b = __builtin_convertvector(v, rd_bf2);
  unsigned u; __builtin_memcpy(&u, &b, 4);
  return u;
#endif
}
// low / high 16-bit element of a 32-bit word -> float (bf16: a shift / a mask)
__device__ __forceinline__ float half_lo_f32(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float half_hi_f32(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
#endif

template <typename T> struct Elem;
template <> struct Elem<float> {
  static constexpr int VE = 4;  // elements per 16-byte vector
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static constexpr int VE = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->v); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// 4 consecutive elements <-> 4 floats (16 B for f32, 8 B for bf16)
__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {
  float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&o)[4]) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  o[0] = half_lo_f32(v.x); o[1] = half_hi_f32(v.x);
  o[2] = half_lo_f32(v.y); o[3] = half_hi_f32(v.y);
}
// XCD-aware block order: the dispatcher places block b on XCD b % 8 (speed only, never correctness).  Giving every XCD a CONTIGUOUS range of
// the launch's work items keeps neighbours -- which share halo rows -- behind the same 4 MiB L2 instead of fetching them once per XCD.
// Bijective on [0, nb).
__device__ __forceinline__ int xcd_contiguous(int bx, int nb) {
  const int q = nb >> 3, r = nb & 7, xcd = bx & 7, idx = bx >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
// occupancy hint (waves per SIMD the register allocation should leave room for); the host emulator build has no such attribute
#ifdef RD_EMU
#define RD_WAVES_PER_EU(n)
#else
#define RD_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))
#endif
// a raw 16-byte vector (kept unconverted while its load is in flight) -> VE floats; the pointer argument only selects the element type
__device__ __forceinline__ void raw16_to_f32(const float*, const uint4& r, float (&o)[4]) {
  o[0] = __uint_as_float(r.x); o[1] = __uint_as_float(r.y); o[2] = __uint_as_float(r.z); o[3] = __uint_as_float(r.w);
}
__device__ __forceinline__ void raw16_to_f32(const bf16_t*, const uint4& r, float (&o)[8]) {
  o[0] = half_lo_f32(r.x); o[1] = half_hi_f32(r.x); o[2] = half_lo_f32(r.y); o[3] = half_hi_f32(r.y);
  o[4] = half_lo_f32(r.z); o[5] = half_hi_f32(r.z); o[6] = half_lo_f32(r.w); o[7] = half_hi_f32(r.w);
}
// guarded forms: the address is always dereferenced (callers clamp it into the tensor) and the result is zero unless `ok` -- no branch,
// so a group of such loads issues back to back instead of one branch + one wait each
__device__ __forceinline__ void ld4z(const float* p, bool ok, float (&o)[4]) {
  float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = ok ? v.x : 0.f; o[1] = ok ? v.y : 0.f; o[2] = ok ? v.z : 0.f; o[3] = ok ? v.w : 0.f;
}
__device__ __forceinline__ void ld4z(const bf16_t* p, bool ok, float (&o)[4]) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  v.x = ok ? v.x : 0u; v.y = ok ? v.y : 0u;
  o[0] = half_lo_f32(v.x); o[1] = half_hi_f32(v.x);
  o[2] = half_lo_f32(v.y); o[3] = half_hi_f32(v.y);
}
__device__ __forceinline__ void st4(float* p, const float (&o)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
__device__ __forceinline__ void st4(bf16_t* p, const float (&o)[4]) {
  uint2 v;
  v.x = pack_bf16x2(o[0], o[1]);
  v.y = pack_bf16x2(o[2], o[3]);
  *reinterpret_cast<uint2*>(p) = v;
}

// VE consecutive elements (one 16-byte access) <-> VE floats
__device__ __forceinline__ void ldv(const float* p, float (&o)[4]) { ld4(p, o); }
__device__ __forceinline__ void stv(float* p, const float (&o)[4]) { st4(p, o); }
__device__ __forceinline__ void ldv(const bf16_t* p, float (&o)[8]) {
  uint4 v = *reinterpret_cast<const uint4*>(p);
  o[0] = half_lo_f32(v.x); o[1] = half_hi_f32(v.x);
  o[2] = half_lo_f32(v.y); o[3] = half_hi_f32(v.y);
  o[4] = half_lo_f32(v.z); o[5] = half_hi_f32(v.z);
  o[6] = half_lo_f32(v.w); o[7] = half_hi_f32(v.w);
}
__device__ __forceinline__ void stv(bf16_t* p, const float (&o)[8]) {
  uint4 v;
  v.x = pack_bf16x2(o[0], o[1]);
  v.y = pack_bf16x2(o[2], o[3]);
  v.z = pack_bf16x2(o[4], o[5]);
  v.w = pack_bf16x2(o[6], o[7]);
  *reinterpret_cast<uint4*>(p) = v;
}

// ---- MFMA ------------------------------------------------------------------------------------
// D = A*B + C on one 64-lane wave; lane l supplies A[i=l&15][k-group l>>4] and B[k-group l>>4][j=l&15],
// and holds D[row=(l>>4)*4+r][col=l&15] in register r (cdna_hip_programming.md section 3).
__device__ __forceinline__ f32x4 mfma_16x16x4_f32(float a, float b, f32x4 c) {
#ifdef RD_EMU
  return emu_mfma_f32_16x16x4f32(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 mfma_16x16x32_bf16(s16x8 a, s16x8 b, f32x4 c) {
#ifdef RD_HALF_F16
  typedef _Float16 rd_h8 __attribute__((ext_vector_type(8)));
#ifdef RD_EMU
  return emu_mfma_f32_16x16x32_f16(a, b, c);
#else
  rd_h8 ha, hb; __builtin_memcpy(&ha, &a, 16); __builtin_memcpy(&hb, &b, 16);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, c, 0, 0, 0);     // same lane maps as the bf16 form (C/D layout is dtype independent)
#endif
#else
#ifdef RD_EMU
  return emu_mfma_f32_16x16x32_bf16(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
#endif
}

// 32x32x16 bf16 MFMA (measured on MI355X, tools/probes/dma_mfma32_probe.hip -> profiles/r02_dma_mfma32_probe.txt): lane l supplies
// A[i = l & 31][k = 8 (l >> 5) + e] and B[k = 8 (l >> 5) + e][j = l & 31], e = 0..7, and holds D[(v & 3) + 8 (v >> 2) + 4 (l >> 5)][l & 31]
// in register v = 0..15.  Twice the FLOPs of 16x16x32 per instruction at the same operand bytes per lane.
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ f32x16 mfma_32x32x16_bf16(s16x8 a, s16x8 b, f32x16 c) {
#ifdef RD_HALF_F16
  typedef _Float16 rd_h8 __attribute__((ext_vector_type(8)));
#ifdef RD_EMU
  return emu_mfma_f32_32x32x16_f16(a, b, c);
#else
  rd_h8 ha, hb; __builtin_memcpy(&ha, &a, 16); __builtin_memcpy(&hb, &b, 16);
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c, 0, 0, 0);
#endif
#else
#ifdef RD_EMU
  return emu_mfma_f32_32x32x16_bf16(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
#endif
}

// LDS DMA (global_load_lds_dwordx4): 16 bytes per lane from global memory straight into LDS, no VGPR staging and no ds_write pass.
// The destination is NOT per-lane: lane l's bytes land at `lds_wave_base + 16 l`, lds_wave_base wave-uniform (probe: as above; lanes
// that are masked off leave their 16 bytes untouched).  Swizzled LDS images are therefore produced by permuting the per-lane SOURCE
// addresses.  Completion is tracked by vmcnt: `dma_wait_all()` (or the vmcnt(0) hipcc puts in front of __syncthreads() while a DMA is
// in flight) before any wave reads the bytes.
// Issued through inline asm ON PURPOSE: hipcc treats the builtin (__builtin_amdgcn_global_load_lds) as an LDS write that may alias any
// later ds_read / ds_write and puts `s_waitcnt vmcnt(0)` in front of the next LDS access -- a prefetch issued before a compute phase is then
// waited for before that phase starts (seen in the ISA of the first version of rd_conv3x3_dma.hip: no overlap at all).  An asm statement
// is invisible to that bookkeeping: the kernel counts its DMAs itself (dma_wait_le<N>, they complete in issue order) and orders them with
// barriers.  M0 carries the destination base and is restored (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void dma16_to_lds(const void* gsrc, void* lds_wave_base) {
#ifdef RD_EMU
  emu_global_load_lds16(gsrc, lds_wave_base);
#else
  // the caller derives lds_wave_base from wave-uniform values only (RD_WAVE_UNIFORM(wave index)): it is an SGPR already
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory", "m0");
#endif
}
// Same DMA with the source given as wave-uniform base (SGPR pair) + per-lane 32-bit byte offset: the per-lane part is a constant of the
// thread in the convolution kernels, so a piece costs no vector address arithmetic.  M0 is written and not restored (hipcc keeps nothing
// live in M0 across an asm statement; it reloads M0 for its own LDS-DMA / interp uses).
__device__ __forceinline__ void dma16_to_lds_base(const void* gbase_uniform, unsigned lane_byte_offset, void* lds_wave_base) {
#ifdef RD_EMU
  emu_global_load_lds16((const char*)gbase_uniform + lane_byte_offset, lds_wave_base);
#else
  const unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_byte_offset), "s"(gbase_uniform), "s"(dst) : "memory", "m0");
#endif
}
__device__ __forceinline__ void dma_wait_all() {
#ifndef RD_EMU
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// LDS transpose read (ds_read_b64_tr_b16): every lane passes an 8-byte aligned LDS address of 4 consecutive 16-bit elements; within a
// 16-lane group lane i receives out[j] = lds16[A_{4j + (i>>2)} + (i&3)], A_m = lane m's address.  The ISA text the programming guide
// cites for this instruction is not in the build image; the mapping was measured on an MI355X (tools/probes/tr_b16_probe.hip ->
// profiles/r01_tr_b16_probe.txt) and the host emulator models exactly that table.  Use: an NHWC (pixel-major) LDS image feeds MFMA
// operands whose k axis is PIXELS (weight gradient) without a transposing store pass.
__device__ __forceinline__ uint2 lds_read_tr16_b64(const void* p) {
#ifdef RD_EMU
  emu_u16x4 r = emu_ds_read_tr16_b64(p);
  uint2 o; o.x = (unsigned)r.v[0] | ((unsigned)r.v[1] << 16); o.y = (unsigned)r.v[2] | ((unsigned)r.v[3] << 16);
  return o;
#else
  typedef __bf16 rd_bf16x4 __attribute__((ext_vector_type(4)));
  rd_bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((rd_bf16x4 __attribute__((address_space(3)))*)p);
  uint2 o; __builtin_memcpy(&o, &v, 8);
  return o;
#endif
}

// 16-bit interleave of two words (one v_perm_b32 each): lo16(l) | lo16(h) << 16 and hi16(l) | hi16(h) << 16
__device__ __forceinline__ unsigned pack_lo16(unsigned l, unsigned h) {
#ifdef RD_EMU
  return (l & 0xffffu) | (h << 16);
#else
  return __builtin_amdgcn_perm(h, l, 0x05040100u);
#endif
}
__device__ __forceinline__ unsigned pack_hi16(unsigned l, unsigned h) {
#ifdef RD_EMU
  return (l >> 16) | (h & 0xffff0000u);
#else
  return __builtin_amdgcn_perm(h, l, 0x07060302u);
#endif
}

// Scheduling fence: no instruction moves across it (used to keep a block of prefetch loads where it is written).
__device__ __forceinline__ void sched_fence() {
#ifndef RD_EMU
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// Orders one wave's LDS traffic (lanes exchanging data through a wave-private LDS region): LDS operations of a wave execute in
// order, so a wavefront-scope fence is enough and the other waves of the workgroup are not held up.  The host emulator runs lanes
// as fibers, where only a block barrier orders them -- every wave of the block must therefore reach the same wave_sync() calls.
__device__ __forceinline__ void wave_sync() {
#ifdef RD_EMU
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// ---- wave / block reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// Sums in butterfly order 1, 2, 4, 8 (row16_sum: the 16 lanes of a DPP row) then 16, 32 (wave_sum_up).  __shfl_xor is a ds_bpermute: address
// arithmetic + an LDS-queue round trip + a wait per step; the fused LoFTR layer is vector-issue bound and spent 128 of them per backward
// launch.  On the GPU the in-row steps are DPP operands of the adds: after the two quad steps every lane of a quad holds the quad's sum, so
// the half-mirror / mirror lanes hold exactly what lane ^ 4 / lane ^ 8 hold; row_bcast15 / row_bcast31 then form (R0 + R1), (R2 + R3) and
// (R2 + R3) + (R0 + R1) in row 3 -- the same two additions the xor steps 16 and 32 perform -- and lane 63 is broadcast through an SGPR.
// The emulator build keeps the xor butterfly: bit-identical sums.  RD_DPP_SUM=0: xor butterfly on the GPU too (A/B), 1: DPP in-row only.
#ifndef RD_DPP_SUM
#define RD_DPP_SUM 2
#endif
#if RD_DPP_SUM && !defined(RD_EMU)
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_f32(float v) {     // lanes outside the row mask read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROWS, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f32<0xB1, 0xF>(v);     // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
  v += dpp_f32<0x141, 0xF>(v);    // row_half_mirror
  v += dpp_f32<0x140, 0xF>(v);    // row_mirror
  return v;
}
#else
__device__ __forceinline__ float row16_sum(float v) {
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o);
  return v;
}
#endif
// sum over the 32 lanes of a wave half (lanes 0-31 / 32-63): valid in the UPPER row of each half on the GPU (lanes 16-31 and 48-63: the in-row
// sums, then row_bcast15 adds the lower row's total), in every lane of the emulator build (xor butterfly through 16); same additions
__device__ __forceinline__ float row32_sum(float v) {
  v = row16_sum(v);
#if RD_DPP_SUM && !defined(RD_EMU)
  v += dpp_f32<0x142, 0xA>(v);    // row_bcast15 into rows 1 and 3
  return v;
#else
  return v + __shfl_xor(v, 16);
#endif
}
__device__ __forceinline__ float wave_sum_up(float v) {
  v = row16_sum(v);
#if RD_DPP_SUM >= 2 && !defined(RD_EMU)
  v += dpp_f32<0x142, 0xA>(v);    // row_bcast15 into rows 1 and 3
  v += dpp_f32<0x143, 0xC>(v);    // row_bcast31 into rows 2 and 3
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#else
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
  return v;
}

// activation codes (RD_ACT_* in include/riders_hip.h)
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_RELU6 = 3 };

__device__ __forceinline__ float act_fwd(float x, int act, float slope) {
  if (act == ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == ACT_LRELU) return x > 0.f ? x : x * slope;
  if (act == ACT_RELU6) return fminf(fmaxf(x, 0.f), 6.f);
  return x;
}
// derivative expressed through the activation OUTPUT z (valid for slope > 0 / relu / relu6)
__device__ __forceinline__ float act_grad_from_out(float z, int act, float slope) {
  if (act == ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == ACT_LRELU) return z > 0.f ? 1.f : slope;
  if (act == ACT_RELU6) return (z > 0.f && z < 6.f) ? 1.f : 0.f;
  return 1.f;
}

__host__ __device__ __forceinline__ int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// one channel's (a, b) partials summed over the rows by a block's 256 threads: four independent 8-byte loads in flight per thread (a plain
// `for (r = t; r < rows; r += 256)` loop is a chain of dependent round trips: 17 us for the 13 K rows of a one-row-per-tile producer)
__device__ __forceinline__ void bn_rows_sum(const float2* __restrict__ p2, int rows, int C, int c, int t, double& a, double& b) {
  int r = t;
  for (; r + 768 < rows; r += 1024) {
    const float2 v0 = p2[(int64_t)r * C + c], v1 = p2[(int64_t)(r + 256) * C + c], v2 = p2[(int64_t)(r + 512) * C + c], v3 = p2[(int64_t)(r + 768) * C + c];
    a += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
    b += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
  }
  for (; r < rows; r += 256) { const float2 v = p2[(int64_t)r * C + c]; a += (double)v.x; b += (double)v.y; }
}

}  // namespace rd
