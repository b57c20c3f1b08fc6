// Decoder head: BatchNorm + activation of the last decoder convolution fused with the 3x3 output convolution (one output channel).
//
// Reference: RCNet/networks.py:773-779 (MultiScaleDecoder: deconv0 -> output0), utils/net_utils.py:84-91 (conv -> BatchNorm2d -> act) and
// :50-82 (output0 = Conv2d(n_filters[-1] = 16 -> 1, 3x3, bias=False, no BatchNorm, linear)).
//
// At RoI resolution (R = 240 crops of 240 x 100: 5.76 M pixels) the 16-channel tensors are 184 MB each (bf16) and every pass over one is
// 35-100 us of HBM time.  The unfused chain moved nine of them around this one layer pair (forward: apply reads y writes a, head reads a;
// backward: the head's data gradient writes da, its weight gradient reads a, the BatchNorm reduce reads da + y, the apply reads da + y and
// writes dy).  The output convolution has ONE output channel: its arithmetic is 144 multiply-adds per pixel and its gradient is a
// function of nine neighbouring dlogits (2 bytes per pixel), so everything around it can be recomputed from y instead of stored:
//   forward   logits = conv3x3(round(act(scale y + shift)))                  reads y (+ halo rows), writes logits      (a never exists)
//   backward  pass 1: sums of g, g xhat (BatchNorm) and of a (x) dlogits (head weight gradient), g = round(da) act'     reads y
//             pass 2: dy = scale (g - c1 - xhat c2)                                                                     reads y, writes dy
// with da = conv3x3^T(dlogits) recomputed in both passes (da never exists).  Every value is rounded where the unfused kernels round it
// (a and da to the activation type, weights to the activation type), sums are fp32 in a fixed order.
//
// Work item = (pixel, group of CPI = 4 or 8 channels), the groups of a pixel in neighbouring lanes; the group's 9 CPI weights in registers.
#include "rd_conv_common.h"
#include <type_traits>
#include <stdio.h>

namespace rd {

static constexpr int HC = 16;      // channels of the producer (RC-Net: n_filters[-1], rcnet_model.py:77-94)

struct HeadArgs {
  const void* y; const void* dl; const float* w;      // y (N,H,W,16); dlogits / logits (N,H,W,1); w fp32 [1][16][3][3]
  const float *scale, *shift, *mean, *rstd, *c1, *c2;
  void* out;                                           // forward: logits; pass 2: dy
  float* partial;                                      // pass 1: rows x HEAD_PC x 2
  int N, H, W, th, tw, tilesH, tilesW, act;
  float slope;
};
static constexpr int HEAD_PC = HC + HC * 9 / 2;        // pairs per partial row: 16 x (sum g, sum g xhat) + the 144 weight-gradient sums

// CPI consecutive channels: raw registers (kept unconverted while the load is in flight) <-> floats
template <typename T, int CPI> struct ChanIO;
template <> struct ChanIO<bf16_t, 8> {
  typedef uint4 Raw;
  static __device__ __forceinline__ Raw ldraw(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ void cvt(const Raw& r, float (&o)[8]) { raw16_to_f32((const bf16_t*)nullptr, r, o); }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&o)[8]) { stv(p, o); }
};
template <> struct ChanIO<bf16_t, 4> {
  typedef uint2 Raw;
  static __device__ __forceinline__ Raw ldraw(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ void cvt(const Raw& r, float (&o)[4]) { o[0] = half_lo_f32(r.x); o[1] = half_hi_f32(r.x); o[2] = half_lo_f32(r.y); o[3] = half_hi_f32(r.y); }
  static __device__ __forceinline__ void st(bf16_t* p, const float (&o)[4]) { st4(p, o); }
};
template <> struct ChanIO<float, 4> {
  typedef float4 Raw;
  static __device__ __forceinline__ Raw ldraw(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ void cvt(const Raw& r, float (&o)[4]) { o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w; }
  static __device__ __forceinline__ void st(float* p, const float (&o)[4]) { st4(p, o); }
};
template <> struct ChanIO<float, 8> {
  struct Raw { float4 a, b; };
  static __device__ __forceinline__ Raw ldraw(const float* p) { Raw r; r.a = *reinterpret_cast<const float4*>(p); r.b = *reinterpret_cast<const float4*>(p + 4); return r; }
  static __device__ __forceinline__ void cvt(const Raw& r, float (&o)[8]) { o[0] = r.a.x; o[1] = r.a.y; o[2] = r.a.z; o[3] = r.a.w; o[4] = r.b.x; o[5] = r.b.y; o[6] = r.b.z; o[7] = r.b.w; }
  static __device__ __forceinline__ void st(float* p, const float (&o)[8]) {
    const float a[4] = {o[0], o[1], o[2], o[3]}, b[4] = {o[4], o[5], o[6], o[7]};
    st4(p, a); st4(p + 4, b);
  }
};
// explicit packed fp32 pairs (v_pk_fma_f32): the compiler pairs the straight multiply-add chains by itself, not the broadcast forms
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// two values rounded to the activation type and back (one v_cvt_pk + two unpacks for the 16-bit types)
template <typename T> __device__ __forceinline__ void rnd2(float& a, float& b);
template <> __device__ __forceinline__ void rnd2<float>(float&, float&) {}
template <> __device__ __forceinline__ void rnd2<bf16_t>(float& a, float& b) { const unsigned w = pack_bf16x2(a, b); a = half_lo_f32(w); b = half_hi_f32(w); }

// activation of the producer, branch-free in the LeakyReLU instantiation: max(u, slope u) = (u > 0 ? u : slope u) bit for bit when
// 0 <= slope <= 1 (the launcher instantiates ACT = LRELU only then)
template <int ACT> __device__ __forceinline__ float head_act(float u, int act, float slope) {
  if (ACT == ACT_LRELU) return fmaxf(u, u * slope);
  return act_fwd(u, act, slope);
}
template <int ACT> __device__ __forceinline__ f32x2 head_act(f32x2 u, int act, float slope) {
  if (ACT == ACT_LRELU) { const f32x2 v = u * f32x2{slope, slope}; return f32x2{fmaxf(u.x, v.x), fmaxf(u.y, v.y)}; }
  return f32x2{act_fwd(u.x, act, slope), act_fwd(u.y, act, slope)};
}
// g = da * act'(u)
template <int ACT> __device__ __forceinline__ float head_act_bwd(float da, float u, int act, float slope) {
  if (ACT == ACT_LRELU) return u > 0.f ? da : da * slope;
  return act ? da * act_grad_from_out(u, act, slope) : da;
}

// sum over the NI = 16 / CPI neighbouring lanes that hold the channel groups of one pixel
template <int NI>
__device__ __forceinline__ float parts_sum(float v) {
#if RD_DPP_SUM && !defined(RD_EMU)
  v += dpp_f32<0xB1, 0xF>(v);                  // quad_perm [1,0,3,2]
  if (NI == 4) v += dpp_f32<0x4E, 0xF>(v);     // quad_perm [2,3,0,1]
#else
  v += __shfl_xor(v, 1);
  if (NI == 4) v += __shfl_xor(v, 2);
#endif
  return v;
}

struct HeadTile { int n, r0, c0; };
// (row, column) of a thread's pixel inside a tile of width w, advanced by a constant number of pixels per iteration without a division
struct PixWalk {
  int r, c, dr, dc, w;
  __device__ __forceinline__ PixWalk() {}
  __device__ __forceinline__ PixWalk(int p0, int step, int width) : r(p0 / width), c(p0 - (p0 / width) * width), dr(step / width), dc(step - (step / width) * width), w(width) {}
  __device__ __forceinline__ void advance() { r += dr; c += dc; if (c >= w) { c -= w; r++; } }
};
__device__ __forceinline__ HeadTile head_tile(const HeadArgs& a, int tile) {
  HeadTile t;
  const int tc = tile % a.tilesW, q = tile / a.tilesW;
  t.c0 = tc * a.tw; t.r0 = (q % a.tilesH) * a.th; t.n = q / a.tilesH;
  return t;
}
// the channel group's weights, rounded to the activation type (what the packed operand of the unfused convolution holds): wq[c][kh * 3 + kw]
template <typename T, int CPI>
__device__ __forceinline__ void head_weights(const float* __restrict__ w, int part, float (&wq)[CPI][9]) {
  const float4* w4 = reinterpret_cast<const float4*>(w + part * CPI * 9);      // 36 / 72 floats (fp32 parameter storage is 16-byte aligned)
  float f[CPI * 9];
#pragma unroll
  for (int i = 0; i < CPI * 9 / 4; i++) { const float4 v = w4[i]; f[i * 4] = v.x; f[i * 4 + 1] = v.y; f[i * 4 + 2] = v.z; f[i * 4 + 3] = v.w; }
#pragma unroll
  for (int c = 0; c < CPI; c++)
#pragma unroll
    for (int k = 0; k < 9; k++) wq[c][k] = Elem<T>::rnd(f[c * 9 + k]);
}
// dlogits of the tile + one pixel of halo -> LDS as floats (zero outside the image): sdl[(r - r0 + 1) * (tw + 2) + (c - c0 + 1)]
template <typename T>
__device__ __forceinline__ void stage_dl(const HeadArgs& a, const HeadTile& tl, float* sdl) {
  const int twh = a.tw + 2, np = (a.th + 2) * twh;
  const T* dl = (const T*)a.dl + (int64_t)tl.n * a.H * a.W;
  for (int i = threadIdx.x; i < np; i += 256) {
    const int lr = i / twh, lc = i - lr * twh;
    const int r = tl.r0 - 1 + lr, c = tl.c0 - 1 + lc;
    const bool ok = (unsigned)r < (unsigned)a.H && (unsigned)c < (unsigned)a.W;
    const float v = Elem<T>::ld(dl + (int64_t)min(max(r, 0), a.H - 1) * a.W + min(max(c, 0), a.W - 1));
    sdl[i] = ok ? v : 0.f;
  }
}
// da[c] = sum over the data gradient's taps tp = (kh', kw') ascending of w[c][2 - kh'][2 - kw'] * dlogits[q + (kh' - 1, kw' - 1)] (the order of
// conv3x3_c1_kernel, rd_conv3x3.hip, the kernel this replaces; fused multiply-adds here: the library is built with -ffp-contract=off and the 288
// multiply-adds per pixel of the two backward passes are what they issue most); e[tp] = that neighbour's dlogit
__device__ __forceinline__ void head_neighbours(const float* sdl, int twh, int pr, int pc, float (&e)[9]) {
#pragma unroll
  for (int kh = 0; kh < 3; kh++)
#pragma unroll
    for (int kw = 0; kw < 3; kw++) e[kh * 3 + kw] = sdl[(pr + kh) * twh + pc + kw];
}
template <typename T, int CPI>
__device__ __forceinline__ void head_dgrad(const float (&wq)[CPI][9], const float (&e)[9], float (&da)[CPI]) {
#pragma unroll
  for (int c = 0; c < CPI; c += 2) {      // two channels per packed multiply-add, taps in ascending order
    f32x2 s = {0.f, 0.f};
#pragma unroll
    for (int tp = 0; tp < 9; tp++) s = fma2(f32x2{wq[c][8 - tp], wq[c + 1][8 - tp]}, f32x2{e[tp], e[tp]}, s);
    da[c] = s.x; da[c + 1] = s.y;
    rnd2<T>(da[c], da[c + 1]);      // the unfused path stores da in the activation type
  }
}

// A thread's items, U at a time, the raw loads of the next U in flight while the current U are computed.  load(q) must be safe for any q
// (it clamps its address); item(q, raw) is called for the valid ones.
template <int U, typename Raw, typename LoadF, typename ItemF>
__device__ __forceinline__ void head_stream(int p0, int step, int width, int rows, LoadF load, ItemF item) {
  PixWalk q[U]; Raw r[U];
#pragma unroll
  for (int j = 0; j < U; j++) { q[j] = PixWalk(p0 + j * step, U * step, width); r[j] = load(q[j]); }
  while (q[0].r < rows) {
    PixWalk qc[U]; Raw rc[U];
#pragma unroll
    for (int j = 0; j < U; j++) { qc[j] = q[j]; rc[j] = r[j]; q[j].advance(); r[j] = load(q[j]); }
    sched_fence();
#pragma unroll
    for (int j = 0; j < U; j++)
      if (qc[j].r < rows) item(qc[j], rc[j]);
  }
}

// ---- forward -----------------------------------------------------------------------------------------------------------------------
// phase 1: every pixel of the tile + halo: a = round(act(scale y + shift)) (zero outside the image: the convolution pads a, not y), its nine
// per-tap dot products over the 16 channels -> LDS; phase 2: an output pixel sums the nine taps of its neighbours.
template <typename T, int ACT, int CPI>
__global__ __launch_bounds__(256) RD_WAVES_PER_EU(CPI == 8 ? 3 : 4) void bn_head_fwd_kernel(HeadArgs a) {
  constexpr int NI = HC / CPI, U = 2;
  typedef typename ChanIO<T, CPI>::Raw Raw;
  RD_DYN_SMEM(smem);
  float* tp = reinterpret_cast<float*>(smem);      // [np][9]
  const int t = threadIdx.x, part = t % NI;
  const HeadTile tl = head_tile(a, xcd_contiguous(blockIdx.x, gridDim.x));
  const int twh = a.tw + 2, thh = a.th + 2, np = thh * twh;
  float wq[CPI][9], sc[CPI], sh[CPI];
  head_weights<T, CPI>(a.w, part, wq);
#pragma unroll
  for (int c = 0; c < CPI; c++) { sc[c] = a.scale[part * CPI + c]; sh[c] = a.shift[part * CPI + c]; }
  const T* yb = (const T*)a.y + (int64_t)tl.n * a.H * a.W * HC + part * CPI;
  auto load = [&](const PixWalk& q) RD_INLINE_LAMBDA {
    const int r = min(max(tl.r0 - 1 + q.r, 0), a.H - 1), c = min(max(tl.c0 - 1 + q.c, 0), a.W - 1);
    return ChanIO<T, CPI>::ldraw(yb + (r * a.W + c) * HC);
  };
  auto taps = [&](const PixWalk& q, const Raw& raw) RD_INLINE_LAMBDA {
    float av[CPI];
    ChanIO<T, CPI>::cvt(raw, av);
#pragma unroll
    for (int c = 0; c < CPI; c += 2) {
      const f32x2 u = f32x2{av[c], av[c + 1]} * f32x2{sc[c], sc[c + 1]} + f32x2{sh[c], sh[c + 1]};
      const f32x2 z = head_act<ACT>(u, a.act, a.slope);
      av[c] = z.x; av[c + 1] = z.y;
      rnd2<T>(av[c], av[c + 1]);
    }
    f32x2 d2[5] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};      // taps (0,1) (2,3) (4,5) (6,7) (8,-)
#pragma unroll
    for (int c = 0; c < CPI; c++) {
      const f32x2 a2 = {av[c], av[c]};
#pragma unroll
      for (int j = 0; j < 4; j++) d2[j] = fma2(f32x2{wq[c][2 * j], wq[c][2 * j + 1]}, a2, d2[j]);
      d2[4].x = fmaf(wq[c][8], av[c], d2[4].x);
    }
    float d[9];
#pragma unroll
    for (int k = 0; k < 9; k++) d[k] = parts_sum<NI>(d2[k >> 1][k & 1]);
    if (q.r < thh) {      // the lanes of a pixel share the nine stores; a pixel outside the image is padding of the convolution's input: zero
      const bool ok = (unsigned)(tl.r0 - 1 + q.r) < (unsigned)a.H && (unsigned)(tl.c0 - 1 + q.c) < (unsigned)a.W;
      const int p = q.r * twh + q.c;
#pragma unroll
      for (int j = 0; j * NI < 9; j++) {      // lane `part` stores taps part, part + NI, ...: its value picked by selects (no store under a branch per tap)
        float v = d[j * NI];
#pragma unroll
        for (int pp = 1; pp < NI; pp++)
          if (j * NI + pp < 9) v = part == pp ? d[j * NI + pp] : v;
        if ((j + 1) * NI <= 9 || j * NI + part < 9) tp[p * 9 + j * NI + part] = ok ? v : 0.f;
      }
    }
  };
  // whole waves run every iteration (parts_sum needs all lanes of a pixel): the trip count is the block's, not the thread's
  {
    const int step = 256 / NI, nit = (np * NI + 255) >> 8;
    PixWalk q[U]; Raw r[U];
#pragma unroll
    for (int j = 0; j < U; j++) { q[j] = PixWalk(t / NI + j * step, U * step, twh); r[j] = load(q[j]); }
    for (int it = 0; it < nit; it += U) {
      PixWalk qc[U]; Raw rc[U];
#pragma unroll
      for (int j = 0; j < U; j++) { qc[j] = q[j]; rc[j] = r[j]; q[j].advance(); r[j] = load(q[j]); }
      sched_fence();
#pragma unroll
      for (int j = 0; j < U; j++)
        if (it + j < nit) taps(qc[j], rc[j]);
    }
  }
  __syncthreads();
  T* out = (T*)a.out + (int64_t)tl.n * a.H * a.W;
  const int npx = a.th * a.tw;
  for (int o = t; o < npx; o += 256) {
    const int pr = o / a.tw, pc = o - pr * a.tw;
    const int r = tl.r0 + pr, c = tl.c0 + pc;
    if (r < a.H && c < a.W) {
      float s = 0.f;
#pragma unroll
      for (int kh = 0; kh < 3; kh++)
#pragma unroll
        for (int kw = 0; kw < 3; kw++) s += tp[((pr + kh) * twh + pc + kw) * 9 + kh * 3 + kw];
      Elem<T>::st(out + (int64_t)r * a.W + c, s);
    }
  }
}

// ---- backward, pass 1: BatchNorm sums + head weight gradient ------------------------------------------------------------------------------
// persistent blocks (block b owns tiles b, b + grid, ...), sums of all its tiles in registers, one partial row per block
template <typename T, int ACT, int CPI>
__global__ __launch_bounds__(256) void bn_head_bwd_reduce_kernel(HeadArgs a, int ntiles) {
  constexpr int NI = HC / CPI, PQ = CPI * 11;      // per channel group: CPI x (sum g, sum g xhat) + CPI x 9 weight-gradient sums
  typedef typename ChanIO<T, CPI>::Raw Raw;
  RD_DYN_SMEM(smem);
  float* sdl = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, part = t % NI, lane = t & 63, wv = t >> 6;
  const int twh = a.tw + 2;
  float wq[CPI][9], sc[CPI], sh[CPI], mu[CPI], rs[CPI];
  head_weights<T, CPI>(a.w, part, wq);
#pragma unroll
  for (int c = 0; c < CPI; c++) {
    const int cc = part * CPI + c;
    sc[c] = a.scale[cc]; sh[c] = a.shift[cc]; mu[c] = a.mean[cc]; rs[c] = a.rstd[cc];
  }
  float sa[CPI], sb[CPI], nmr[CPI];
  f32x2 wacc[CPI][5];      // the nine weight-gradient sums of a channel as pairs (k, k + 1); the tenth slot stays zero
#pragma unroll
  for (int c = 0; c < CPI; c++) {
    sa[c] = 0.f; sb[c] = 0.f; nmr[c] = -(mu[c] * rs[c]);
#pragma unroll
    for (int k = 0; k < 5; k++) wacc[c][k] = f32x2{0.f, 0.f};
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const HeadTile tl = head_tile(a, tile);
    __syncthreads();      // the previous tile's readers
    stage_dl<T>(a, tl, sdl);
    __syncthreads();
    const T* yb = (const T*)a.y + (int64_t)tl.n * a.H * a.W * HC + part * CPI;
    const bool ragged = tl.r0 + a.th > a.H || tl.c0 + a.tw > a.W;      // (block uniform) a last tile with pixels outside the image
    auto load = [&](const PixWalk& q) RD_INLINE_LAMBDA {
      const int r = min(tl.r0 + q.r, a.H - 1), c = min(tl.c0 + q.c, a.W - 1);
      return ChanIO<T, CPI>::ldraw(yb + (r * a.W + c) * HC);
    };
    auto item = [&](const PixWalk& q, const Raw& raw) RD_INLINE_LAMBDA {
      float yy[CPI], e[9], da[CPI], av[CPI];
      ChanIO<T, CPI>::cvt(raw, yy);
      head_neighbours(sdl, twh, q.r, q.c, e);
      if (ragged) {      // a pixel outside the image contributes nothing
        const bool ok = tl.r0 + q.r < a.H && tl.c0 + q.c < a.W;
#pragma unroll
        for (int k = 0; k < 9; k++) e[k] = ok ? e[k] : 0.f;
      }
      head_dgrad<T, CPI>(wq, e, da);
#pragma unroll
      for (int c = 0; c < CPI; c += 2) {
        const f32x2 y2 = {yy[c], yy[c + 1]};
        const f32x2 u = y2 * f32x2{sc[c], sc[c + 1]} + f32x2{sh[c], sh[c + 1]};
        const f32x2 z = head_act<ACT>(u, a.act, a.slope);
        av[c] = z.x; av[c + 1] = z.y;
        rnd2<T>(av[c], av[c + 1]);
        const f32x2 g = {head_act_bwd<ACT>(da[c], u.x, a.act, a.slope), head_act_bwd<ACT>(da[c + 1], u.y, a.act, a.slope)};
        const f32x2 xh = fma2(y2, f32x2{rs[c], rs[c + 1]}, f32x2{nmr[c], nmr[c + 1]});      // xhat = (y - mean) rstd
        const f32x2 s1 = f32x2{sa[c], sa[c + 1]} + g, s2 = fma2(g, xh, f32x2{sb[c], sb[c + 1]});
        sa[c] = s1.x; sa[c + 1] = s1.y; sb[c] = s2.x; sb[c + 1] = s2.y;
      }
      // dw[c][kh][kw] = sum_q a[q][c] * dlogits[q - (kh - 1, kw - 1)] = a * e[8 - (kh * 3 + kw)]
      const f32x2 er[5] = {f32x2{e[8], e[7]}, f32x2{e[6], e[5]}, f32x2{e[4], e[3]}, f32x2{e[2], e[1]}, f32x2{e[0], 0.f}};
#pragma unroll
      for (int c = 0; c < CPI; c++) {
        const f32x2 a2 = {av[c], av[c]};
#pragma unroll
        for (int k = 0; k < 5; k++) wacc[c][k] = fma2(a2, er[k], wacc[c][k]);
      }
    };
    head_stream<2, Raw>(t / NI, 256 / NI, a.tw, a.th, load, item);
  }
  // block reduction: lanes of the same channel group (xor NI .. 32, fixed tree), then the four waves through LDS in wave order
  __syncthreads();
  float* red = sdl;      // [4 waves][NI groups][PQ]
  auto lanes = [&](float v) RD_INLINE_LAMBDA {
#pragma unroll
    for (int o = NI; o < 64; o <<= 1) v += __shfl_xor(v, o);
    return v;
  };
#pragma unroll
  for (int c = 0; c < CPI; c++) {
    const float x0 = lanes(sa[c]), x1 = lanes(sb[c]);
    if (lane < NI) { red[(wv * NI + part) * PQ + c * 2] = x0; red[(wv * NI + part) * PQ + c * 2 + 1] = x1; }
#pragma unroll
    for (int k = 0; k < 9; k++) {
      const float x = lanes(wacc[c][k >> 1][k & 1]);
      if (lane < NI) red[(wv * NI + part) * PQ + 2 * CPI + c * 9 + k] = x;
    }
  }
  __syncthreads();
  // partial row: pairs [0, 16) = (sum g, sum g xhat) of channel c; pairs [16, 88) = the weight-gradient sums dw[c * 9 + k], two per pair
  float* row = a.partial + (int64_t)blockIdx.x * HEAD_PC * 2;
  for (int j = t; j < NI * PQ; j += 256) {
    const int h = j / PQ, q = j - h * PQ;
    const float v = (red[(0 * NI + h) * PQ + q] + red[(1 * NI + h) * PQ + q]) + (red[(2 * NI + h) * PQ + q] + red[(3 * NI + h) * PQ + q]);
    if (q < 2 * CPI) row[(h * CPI + (q >> 1)) * 2 + (q & 1)] = v;
    else row[HC * 2 + h * CPI * 9 + (q - 2 * CPI)] = v;
  }
}

// partial rows -> BatchNorm coefficients + parameter gradients (blocks [0, 16): exactly bn_bwd_finalize_kernel) and the head's weight gradient
// (blocks [16, 88): two of its 144 sums each); fixed order, double precision
__global__ __launch_bounds__(256) void bn_head_finalize_kernel(const float* __restrict__ partial, int rows, double count, float* dgamma, float* dbeta,
                                                               int bn_acc, float* c1, float* c2, float* dw, int w_acc) {
  __shared__ double s1[4], s2[4];
  const int c = blockIdx.x, t = threadIdx.x;
  double x = 0.0, y = 0.0;
  bn_rows_sum(reinterpret_cast<const float2*>(partial), rows, HEAD_PC, c, t, x, y);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { x += __shfl_xor(x, o); y += __shfl_xor(y, o); }
  if ((t & 63) == 0) { s1[t >> 6] = x; s2[t >> 6] = y; }
  __syncthreads();
  if (t == 0) {
    x = (s1[0] + s1[1]) + (s1[2] + s1[3]); y = (s2[0] + s2[1]) + (s2[2] + s2[3]);
    if (c < HC) {
      if (dbeta) dbeta[c] = bn_acc ? dbeta[c] + (float)x : (float)x;
      if (dgamma) dgamma[c] = bn_acc ? dgamma[c] + (float)y : (float)y;
      c1[c] = (float)(x / count); c2[c] = (float)(y / count);
    } else if (dw) {
      const int j = (c - HC) * 2;
      dw[j] = w_acc ? dw[j] + (float)x : (float)x;
      dw[j + 1] = w_acc ? dw[j + 1] + (float)y : (float)y;
    }
  }
}

// ---- backward, pass 2: dy = scale (g - c1 - xhat c2), g = round(conv3x3^T(dlogits)) act'(scale y + shift) ------------------------------------
template <typename T, int ACT, int CPI>
__global__ __launch_bounds__(256) void bn_head_bwd_apply_kernel(HeadArgs a) {
  constexpr int NI = HC / CPI;
  typedef typename ChanIO<T, CPI>::Raw Raw;
  RD_DYN_SMEM(smem);
  float* sdl = reinterpret_cast<float*>(smem);
  const int t = threadIdx.x, part = t % NI;
  const int twh = a.tw + 2;
  const HeadTile tl = head_tile(a, xcd_contiguous(blockIdx.x, gridDim.x));
  float wq[CPI][9], sc[CPI], sh[CPI], cA[CPI], cB[CPI];
  head_weights<T, CPI>(a.w, part, wq);
  // dy = scale (g - c1 - (y - mean) rstd c2) = fma(B, y, fma(scale, g, A)): two fused multiply-adds per element
#pragma unroll
  for (int c = 0; c < CPI; c++) {
    const int cc = part * CPI + c;
    sc[c] = a.scale[cc]; sh[c] = a.shift[cc];
    const float rk = a.rstd[cc] * a.c2[cc];
    cB[c] = -(sc[c] * rk); cA[c] = sc[c] * (rk * a.mean[cc] - a.c1[cc]);
  }
  stage_dl<T>(a, tl, sdl);
  __syncthreads();
  const int64_t ib = (int64_t)tl.n * a.H * a.W * HC + part * CPI;
  const T* yb = (const T*)a.y + ib;
  T* ob = (T*)a.out + ib;
  auto load = [&](const PixWalk& q) RD_INLINE_LAMBDA {
    const int r = min(tl.r0 + q.r, a.H - 1), c = min(tl.c0 + q.c, a.W - 1);
    return ChanIO<T, CPI>::ldraw(yb + (r * a.W + c) * HC);
  };
  auto item = [&](const PixWalk& q, const Raw& raw) RD_INLINE_LAMBDA {
    float yy[CPI], e[9], da[CPI], dy[CPI];
    ChanIO<T, CPI>::cvt(raw, yy);
    head_neighbours(sdl, twh, q.r, q.c, e);
    head_dgrad<T, CPI>(wq, e, da);
#pragma unroll
    for (int c = 0; c < CPI; c += 2) {
      const f32x2 y2 = {yy[c], yy[c + 1]}, s2 = {sc[c], sc[c + 1]};
      const f32x2 u = y2 * s2 + f32x2{sh[c], sh[c + 1]};
      const f32x2 g = {head_act_bwd<ACT>(da[c], u.x, a.act, a.slope), head_act_bwd<ACT>(da[c + 1], u.y, a.act, a.slope)};
      const f32x2 o = fma2(f32x2{cB[c], cB[c + 1]}, y2, fma2(s2, g, f32x2{cA[c], cA[c + 1]}));
      dy[c] = o.x; dy[c + 1] = o.y;
    }
    const int r = tl.r0 + q.r, c = tl.c0 + q.c;
    if (r < a.H && c < a.W) ChanIO<T, CPI>::st(ob + (r * a.W + c) * HC, dy);
  };
  head_stream<4, Raw>(t / NI, 256 / NI, a.tw, a.th, load, item);
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
struct HeadPlan { int th, tw, tilesH, tilesW; };
static HeadPlan head_plan(int H, int W, int max_np) {
  HeadPlan p;
  p.tilesW = W <= 160 ? 1 : (int)cdiv(W, 128);
  p.tw = (int)cdiv(W, p.tilesW);
  int th = std::max(1, std::min(H, max_np / (p.tw + 2) - 2));
  p.tilesH = (int)cdiv(H, th);
  p.th = (int)cdiv(H, p.tilesH);
  return p;
}
bool bn_head_ok(int N, int H, int W, int C, int dtype) {
  (void)dtype;
  return C == HC && N > 0 && H > 0 && W > 0 && (int64_t)N * H * W < ((int64_t)1 << 31) / HC;
}
static int head_fwd_np() { return std::min(rd_opt(OPT_HEAD_NP, 1360), 1800); }     // pixels (tile + halo) of a forward tile: 36 bytes of LDS each (<= 64 KiB per block)
static int head_bwd_np() { return 2048; }
static int head_cpi(int which) { return ((rd_opt(OPT_HEAD_CPI, 0x448) >> (4 * which)) & 15) == 8 ? 8 : 4; }      // channels per work item of kernel `which`
static HeadArgs head_args(const HeadPlan& p, int N, int H, int W, int act, float slope) {
  HeadArgs a = {};
  a.N = N; a.H = H; a.W = W; a.th = p.th; a.tw = p.tw; a.tilesH = p.tilesH; a.tilesW = p.tilesW; a.act = act; a.slope = slope;
  return a;
}
int bn_head_rows(int N, int H, int W) {
  const HeadPlan p = head_plan(H, W, head_bwd_np());
  return (int)std::min<int64_t>((int64_t)N * p.tilesH * p.tilesW, 1024);
}
// (activation, channels per item) -> compile-time
template <typename F> static void head_dispatch(int act, float slope, int cpi, F f) {
  if (act == ACT_LRELU && slope >= 0.f && slope <= 1.f) { if (cpi == 8) f(std::integral_constant<int, ACT_LRELU>(), std::integral_constant<int, 8>()); else f(std::integral_constant<int, ACT_LRELU>(), std::integral_constant<int, 4>()); }
  else { if (cpi == 8) f(std::integral_constant<int, -1>(), std::integral_constant<int, 8>()); else f(std::integral_constant<int, -1>(), std::integral_constant<int, 4>()); }
}
void launch_bn_head_fwd(const void* y, const float* scale, const float* shift, int act, float slope, const float* w, void* logits, int N, int H, int W,
                        int dtype, hipStream_t st) {
  const HeadPlan p = head_plan(H, W, head_fwd_np());
  HeadArgs a = head_args(p, N, H, W, act, slope);
  a.y = y; a.scale = scale; a.shift = shift; a.w = w; a.out = logits;
  const unsigned grid = (unsigned)(N * p.tilesH * p.tilesW);
  const size_t lds = (size_t)(p.th + 2) * (p.tw + 2) * 9 * sizeof(float);
  head_dispatch(act, slope, head_cpi(0), [&](auto ac, auto cp) {
    constexpr int A = decltype(ac)::value, CPI = decltype(cp)::value;
    if (dtype == 0) hipLaunchKernelGGL((bn_head_fwd_kernel<float, A, CPI>), dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((bn_head_fwd_kernel<bf16_t, A, CPI>), dim3(grid), dim3(256), lds, st, a);
  });
}
void launch_bn_head_bwd_reduce(const void* dl, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, int act,
                               float slope, const float* w, float* partial, int N, int H, int W, int dtype, hipStream_t st) {
  const HeadPlan p = head_plan(H, W, head_bwd_np());
  HeadArgs a = head_args(p, N, H, W, act, slope);
  a.y = y; a.dl = dl; a.scale = scale; a.shift = shift; a.mean = mean; a.rstd = rstd; a.w = w; a.partial = partial;
  const int ntiles = N * p.tilesH * p.tilesW;
  const unsigned grid = (unsigned)bn_head_rows(N, H, W);
  const size_t lds = std::max((size_t)(p.th + 2) * (p.tw + 2), (size_t)4 * HC * 11) * sizeof(float);
  head_dispatch(act, slope, head_cpi(1), [&](auto ac, auto cp) {
    constexpr int A = decltype(ac)::value, CPI = decltype(cp)::value;
    if (dtype == 0) hipLaunchKernelGGL((bn_head_bwd_reduce_kernel<float, A, CPI>), dim3(grid), dim3(256), lds, st, a, ntiles);
    else hipLaunchKernelGGL((bn_head_bwd_reduce_kernel<bf16_t, A, CPI>), dim3(grid), dim3(256), lds, st, a, ntiles);
  });
}
void launch_bn_head_bwd_apply(const void* dl, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, int act,
                              float slope, const float* w, const float* partial, int rows, float* coef, float* dgamma, float* dbeta, int bn_acc,
                              float* dw, int w_acc, void* dy, int N, int H, int W, int dtype, hipStream_t st) {
  hipLaunchKernelGGL(bn_head_finalize_kernel, dim3(HEAD_PC), dim3(256), 0, st, partial, rows, (double)N * H * W, dgamma, dbeta, bn_acc, coef, coef + HC, dw,
                     w_acc);
  const HeadPlan p = head_plan(H, W, head_bwd_np());
  HeadArgs a = head_args(p, N, H, W, act, slope);
  a.y = y; a.dl = dl; a.scale = scale; a.shift = shift; a.mean = mean; a.rstd = rstd; a.w = w; a.c1 = coef; a.c2 = coef + HC; a.out = dy;
  const unsigned grid = (unsigned)(N * p.tilesH * p.tilesW);
  const size_t lds = (size_t)(p.th + 2) * (p.tw + 2) * sizeof(float);
  head_dispatch(act, slope, head_cpi(2), [&](auto ac, auto cp) {
    constexpr int A = decltype(ac)::value, CPI = decltype(cp)::value;
    if (dtype == 0) hipLaunchKernelGGL((bn_head_bwd_apply_kernel<float, A, CPI>), dim3(grid), dim3(256), lds, st, a);
    else hipLaunchKernelGGL((bn_head_bwd_apply_kernel<bf16_t, A, CPI>), dim3(grid), dim3(256), lds, st, a);
  });
}
const char* bn_head_kernel_name(int which, int dtype, int act) {
  static thread_local char buf[96];
  static const char* const names[3] = {"bn_head_fwd_kernel", "bn_head_bwd_reduce_kernel", "bn_head_bwd_apply_kernel"};
  snprintf(buf, sizeof(buf), "%s<%s, %d, %d>", names[which], dtype == 0 ? "float" : RD_T16_NAME, act == ACT_LRELU ? ACT_LRELU : -1, head_cpi(which));
  return buf;
}

}  // namespace rd
