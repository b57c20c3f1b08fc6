// Fused LoFTR encoder layer (forward and backward), one workgroup per ROI.
//
// Reference: RCNet/linear_attention.py:84-135 (LoFTREncoderLayer.forward):
//   q = Wq x, k = Wk src, v = Wv src;  message = LinearAttention(q, k, v);  message = norm1(Wm message);
//   message = norm2(W2 relu(W0 [x | message]));  return x + message
// RC-Net applies it 16 times per step to R = 240 ROIs x L = 21 tokens x C = 128 channels.  As separate launches that is ~600 small
// latency-bound kernels (3.5 ms of a 17 ms step, 26 GFLOP).  Every ROI is independent through the whole layer, so here a workgroup
// keeps one ROI's token tile (21 rows padded to 32) in LDS and walks the layer: six token GEMMs on the MFMA (weights streamed from
// L2 in fragment layout, tokens are the B operand so padded rows never contaminate real ones), the per-head linear attention of
// rd_attention_head.h, LayerNorm on the VALU.  Intermediates the backward / the weight gradients need are written once to HBM
// (q, k, v, attention output, pre-norm activations, hidden) -- 21 x 1.3 K elements per ROI.  The backward kernel mirrors the walk with the
// transposed (mode 1) packed weights and leaves (input, output-gradient) pairs for the grouped weight gradient (rd_linear_wgrad.hip);
// LayerNorm parameter gradients leave per-ROI partials that bn_bwd_finalize sums in a fixed order.
#include "rd_attention_head.h"

namespace rd {

#ifdef RD_LOFTR_PROF   // phase timing probe (tools/loftr_prof.py builds a separate library with this define; never in the product build)
__device__ unsigned long long g_loftr_prof[32];
#define LPROF_INIT unsigned long long pt_ = wall_clock64();
#define LPROF(i) { const unsigned long long n_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&g_loftr_prof[i], n_ - pt_); pt_ = n_; }
#define LPROFW(i) if ((threadIdx.x & 63) == 0) atomicAdd(&g_loftr_prof[i], wall_clock64() - pt_);   // per wave, time since the phase began (sum over 8 waves)
#else
#define LPROF_INIT
#define LPROF(i)
#define LPROFW(i)
#endif
static constexpr int LC = 128, LC2 = 256, LTOK = 32, LF = 132;   // channels, hidden, padded tokens, float row pitch
static constexpr int LNW = 8, LNT = LNW * 64;                     // waves / threads per workgroup: one attention head per wave, two waves
                                                                  // per SIMD (the layer is a chain of short latency-bound phases)

template <typename T> struct LoftrGeom {
  static constexpr int PADE = 16 / (int)sizeof(T);        // one 16-byte slot of padding per row: 16 token rows land on 16 bank groups
  static constexpr int LDA = LC + PADE, LDH = LC2 + PADE;
};

template <typename T>
struct LoftrSmemFwd {
  T bX[LTOK * LoftrGeom<T>::LDA], bS[LTOK * LoftrGeom<T>::LDA], bM[LTOK * LoftrGeom<T>::LDA];
  union U {
    AttnSmem at[LNW];
    struct FH { float f[LTOK * LF]; T h[LTOK * LoftrGeom<T>::LDH]; } fh;
  } u;
};
template <typename T>
struct LoftrSmemBwd {
  T bD[LTOK * LoftrGeom<T>::LDA];
  float acc[LTOK * LF];
  float red[LNW][2][LC];
  union U {
    AttnSmem at[LNW];
    struct FH { float f[LTOK * LF]; T h[LTOK * LoftrGeom<T>::LDH]; } fh;
  } u;
};

// rows x 128 channels of a token matrix -> LDS tile (rows >= `rows` zero filled).  RowsReg splits the copy into fetch (global loads
// issued) and commit (LDS stores) so that other loads can be issued in between without being waited for (vmcnt counts in order).
template <typename T> struct RowsReg { static constexpr int NIT = LTOK * (LC / Elem<T>::VE) / LNT; uint4 v[NIT]; };
template <typename T>
__device__ __forceinline__ void rows_fetch(const T* __restrict__ g, int rows, RowsReg<T>& rr) {
  constexpr int VE = Elem<T>::VE, SPR = LC / VE;
#pragma unroll
  for (int i = 0; i < RowsReg<T>::NIT; i++) {
    const int idx = threadIdx.x + i * LNT, r = idx / SPR, sl = idx - r * SPR;
    const uint4 v = *reinterpret_cast<const uint4*>(g + (int64_t)min(r, rows - 1) * LC + sl * VE);   // clamped row: no branch around the load
    rr.v[i] = v;
  }
}
template <typename T>
__device__ __forceinline__ void rows_commit(const RowsReg<T>& rr, int rows, T* lds, int ld) {
  constexpr int VE = Elem<T>::VE, SPR = LC / VE;
#pragma unroll
  for (int i = 0; i < RowsReg<T>::NIT; i++) {
    const int idx = threadIdx.x + i * LNT, r = idx / SPR, sl = idx - r * SPR;
    *reinterpret_cast<uint4*>(lds + r * ld + sl * VE) = (r < rows) ? rr.v[i] : make_uint4(0, 0, 0, 0);
  }
}
template <typename T>
__device__ __forceinline__ void load_rows(const T* __restrict__ g, int rows, T* lds, int ld) {
  RowsReg<T> rr;
  rows_fetch<T>(g, rows, rr);
  rows_commit<T>(rr, rows, lds, ld);
}

// tile[32 tokens][NO] = A[32][K] . W^T with A in LDS (columns [0,K0) from a0, [K0,K) from a1, row pitch lda) and W packed [NO][K].
// epi(ct, acc): this lane holds acc[tt][r] = (token tt*16 + (lane&15), channel ct*16 + (lane>>4)*4 + r).
// The weight fragments of one GEMM live in a TokW: tok_load() issues every load of the GEMM at once (when they fit in 64 VGPRs) and
// is called a phase EARLY -- before the barrier / attention / LayerNorm that precedes the GEMM -- so the L2 round trip of the weights
// overlaps that phase instead of starting the GEMM (the layer is a chain of latency-bound phases; profiles/r01_loftr_phases.txt).
template <typename T, int K, int NO> struct TokW {
  static constexpr int VE = Elem<T>::VE, SE = 4 * VE, NS = K / SE, KSL = K / VE, NCT = NO / 16 / LNW;
  static constexpr bool HOIST = NCT * NS <= 16;
  uint4 f[HOIST ? NCT : 1][NS];
  const uint4* wp;
};
template <typename T, int K, int NO>
__device__ __forceinline__ void tok_load(const void* wpacked, TokW<T, K, NO>& w) {
  using W = TokW<T, K, NO>;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  // the fragment-ordered second copy of the packed operand (pack_has_tokfrag, rd_conv.hip): behind the NO x K row-major one
  w.wp = reinterpret_cast<const uint4*>(wpacked) + NO * W::KSL;
  if (W::HOIST) {
#pragma unroll
    for (int ci = 0; ci < W::NCT; ci++)
#pragma unroll
      for (int ks = 0; ks < W::NS; ks++) {
#if defined(RD_LOFTR_PROBE) && RD_LOFTR_PROBE == 5      // timing probe: every fragment from one 1-KiB region (L1 hits): what the weight streaming costs
        const uint4 v = w.wp[fr * 4 + fg + ((ks + ci) & 1) * 64];
#elif defined(RD_LOFTR_PROBE) && RD_LOFTR_PROBE == 7    // timing probe: the row-major copy (16 rows x 64 bytes per load instruction), as before round 4
        const uint4 v = (w.wp - NO * W::KSL)[(int64_t)((wv + ci * LNW) * 16 + fr) * W::KSL + ks * 4 + fg];
#else
        const uint4 v = w.wp[((wv + ci * LNW) * W::NS + ks) * 64 + lane];      // one contiguous 1-KiB fragment per instruction
#endif
        w.f[ci][ks] = v;
      }
    sched_fence();   // keep the loads here: the scheduler otherwise sinks them next to their MFMA, two at a time
  }
}
template <typename T, int K, int NO, typename Epi>
__device__ __forceinline__ void tok_mma(const T* a0, const T* a1, int K0, int lda, TokW<T, K, NO>& w, Epi epi) {
  using W = TokW<T, K, NO>;
  constexpr int VE = W::VE, SE = W::SE, NS = W::NS;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int ci = 0; ci < W::NCT; ci++) {
    const int ct = wv + ci * LNW;
    if (!W::HOIST) {
#pragma unroll
      for (int ks = 0; ks < NS; ks++) { const uint4 v = w.wp[(ct * NS + ks) * 64 + lane]; w.f[0][ks] = v; }
    }
    const uint4 (&wf)[NS] = w.f[W::HOIST ? ci : 0];
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
    for (int ks = 0; ks < NS; ks++) {
      const T* ap = (ks * SE < K0) ? a0 + ks * SE : a1 + (ks * SE - K0);
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const uint4 pf = *reinterpret_cast<const uint4*>(ap + (tt * 16 + fr) * lda + fg * VE);
        if (sizeof(T) == 4) {
          acc[tt] = mfma_16x16x4_f32(__uint_as_float(wf[ks].x), __uint_as_float(pf.x), acc[tt]);
          acc[tt] = mfma_16x16x4_f32(__uint_as_float(wf[ks].y), __uint_as_float(pf.y), acc[tt]);
          acc[tt] = mfma_16x16x4_f32(__uint_as_float(wf[ks].z), __uint_as_float(pf.z), acc[tt]);
          acc[tt] = mfma_16x16x4_f32(__uint_as_float(wf[ks].w), __uint_as_float(pf.w), acc[tt]);
        } else {
          s16x8 wa, pb;
          __builtin_memcpy(&wa, &wf[ks], 16);
          __builtin_memcpy(&pb, &pf, 16);
          acc[tt] = mfma_16x16x32_bf16(wa, pb, acc[tt]);
        }
      }
    }
    epi(ct, acc);
  }
}
// ---- forward ----------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LNT) void loftr_layer_fwd_kernel(const T* __restrict__ x, const T* __restrict__ src, LoftrW w,
                                                              T* __restrict__ out, LoftrSaved sv, int L, int S, float eps_attn,
                                                              float eps_ln) {
  constexpr int LDA = LoftrGeom<T>::LDA, LDH = LoftrGeom<T>::LDH;
  __shared__ LoftrSmemFwd<T> sm;
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  const bool self = (src == x);
  const int64_t xo = (int64_t)n * L * LC, so = (int64_t)n * S * LC;

  LPROF_INIT
  TokW<T, LC, LC> Wq, Wk, Wv, Wm;
  TokW<T, LC2, LC2> W0;
  TokW<T, LC2, LC> W2;
  constexpr bool EARLY = sizeof(T) == 2;   // fp32 fragments are twice the registers: its weights are fetched right before each GEMM
  // LayerNorm parameters first: loads return in order, so requested behind a weight prefetch they would wait for all of it
  const float lg1[2] = {w.g1[lane], w.g1[64 + lane]}, lb1[2] = {w.b1[lane], w.b1[64 + lane]};
  const float lg2[2] = {w.g2[lane], w.g2[64 + lane]}, lb2[2] = {w.b2[lane], w.b2[64 + lane]};
  {
    RowsReg<T> rx, rs;
    rows_fetch<T>(x + xo, L, rx);
    if (!self) rows_fetch<T>(src + so, S, rs);
    sched_fence();
    tok_load<T, LC, LC>(w.wq, Wq);
    if (EARLY) { tok_load<T, LC, LC>(w.wk, Wk); tok_load<T, LC, LC>(w.wv, Wv); }
    rows_commit<T>(rx, L, sm.bX, LDA);
    if (!self) rows_commit<T>(rs, S, sm.bS, LDA);
  }
  __syncthreads();
  LPROF(0)
  const T* sp = self ? sm.bX : sm.bS;

  auto to_global = [&](T* base, int64_t off, int rows, int ldg) RD_INLINE_LAMBDA {
    return [=](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const int tok = tt * 16 + fr;
        if (tok < rows) {
          float v[4] = {acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]};
          st4(base + off + (int64_t)tok * ldg + ct * 16 + fg * 4, v);
        }
      }
    };
  };
#ifndef RD_LOFTR_ATT_GLOBAL
  // The wave that owns head h (one head per wave) is the wave whose projection tile is channels 16 h .. 16 h + 15: its q / k / v accumulators
  // go to global memory for the backward AND, rounded the same way and feature-mapped, straight into its own attention scratch -- no store ->
  // barrier -> L2 load round trip in front of the attention phase (RD_LOFTR_ATT_GLOBAL: the round trip, for A/B).
  AttnSmem& as = sm.u.at[wv];
  wave_sync();      // (a previous phase's reads of this scratch)
  auto to_both = [&](T* base, int64_t off, int rows, float* lds, int mode, float scale) RD_INLINE_LAMBDA {
    return [=](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const int tok = tt * 16 + fr;
        float v[4] = {acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]};
        if (tok < rows) st4(base + off + (int64_t)tok * LC + ct * 16 + fg * 4, v);
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float x = Elem<T>::rnd(v[r]);      // what the stored tensor holds (and what the backward recomputes from)
          lds[tok * LD + fg * 4 + r] = tok < rows ? (mode ? (x > 0.f ? x + 1.f : __expf(x)) : x * scale) : 0.f;
        }
      }
    };
  };
  tok_mma<T, LC, LC>(sm.bX, sm.bX, LC, LDA, Wq, to_both((T*)sv.q, xo, L, as.q, 1, 1.f));
  if (!EARLY) tok_load<T, LC, LC>(w.wk, Wk);
  tok_mma<T, LC, LC>(sp, sp, LC, LDA, Wk, to_both((T*)sv.k, so, S, as.k, 1, 1.f));
  if (!EARLY) tok_load<T, LC, LC>(w.wv, Wv);
  tok_mma<T, LC, LC>(sp, sp, LC, LDA, Wv, to_both((T*)sv.v, so, S, as.v, 0, 1.f / (float)S));
  wave_sync();
  LPROF(1)
#else
  tok_mma<T, LC, LC>(sm.bX, sm.bX, LC, LDA, Wq, to_global((T*)sv.q, xo, L, LC));
  if (!EARLY) tok_load<T, LC, LC>(w.wk, Wk);
  tok_mma<T, LC, LC>(sp, sp, LC, LDA, Wk, to_global((T*)sv.k, so, S, LC));
  if (!EARLY) tok_load<T, LC, LC>(w.wv, Wv);
  tok_mma<T, LC, LC>(sp, sp, LC, LDA, Wv, to_global((T*)sv.v, so, S, LC));
  __syncthreads();
  LPROF(1)
#endif

  // linear attention: wave wv owns head wv (q, k, v come back from L2; the head routine stages them per head)
#if defined(RD_LOFTR_PROBE) && RD_LOFTR_PROBE == 4
  tok_load<T, LC, LC>(w.wm, Wm);
#else
  {
    auto hook = [&]() RD_INLINE_LAMBDA { tok_load<T, LC, LC>(w.wm, Wm); };      // merge weights: in flight across the phase
#ifndef RD_LOFTR_ATT_GLOBAL
    attn_head<T, false, decltype(hook), true>(sm.u.at[wv], (const T*)sv.q, (const T*)sv.k, (const T*)sv.v, (const T*)nullptr, (T*)sv.att, (T*)nullptr,
                                                (T*)nullptr, (T*)nullptr, n, wv, true, L, S, LC, LC, LC, LC, eps_attn, hook, sm.bM, LDA);
#else
    attn_head<T, false, decltype(hook), false>(sm.u.at[wv], (const T*)sv.q, (const T*)sv.k, (const T*)sv.v, (const T*)nullptr, (T*)sv.att, (T*)nullptr,
                                                 (T*)nullptr, (T*)nullptr, n, wv, true, L, S, LC, LC, LC, LC, eps_attn, hook);
#endif
  }
#endif
  __syncthreads();
  LPROF(2)
#ifndef RD_LOFTR_ATT_GLOBAL
  const T* const attT = sm.bM;      // every head wrote its 16 columns of the attention output into the (still unused) message tile
#else
  load_rows<T>((const T*)sv.att + xo, L, sm.bS, LDA);
  __syncthreads();
  const T* const attT = sm.bS;
#endif
  LPROF(3)

  // merge projection -> fp32 scratch (values as the unfused path stores them) + saved pre-norm activation
  auto to_f = [&](T* gbase, int rows) RD_INLINE_LAMBDA {
    return [=](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const int tok = tt * 16 + fr;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) { v[r] = Elem<T>::rnd(acc[tt][r]); sm.u.fh.f[tok * LF + ct * 16 + fg * 4 + r] = v[r]; }
        if (tok < rows) st4(gbase + xo + (int64_t)tok * LC + ct * 16 + fg * 4, v);
      }
    };
  };
  tok_load<T, LC2, LC2>(w.w0, W0);     // in flight across the merge GEMM and norm1
  tok_mma<T, LC, LC>(attT, attT, LC, LDA, Wm, to_f((T*)sv.mpre, L));
  __syncthreads();
  LPROF(4)

  // LayerNorm (one wave per row, two channels per lane); which = 0: norm1 -> bM + saved message, 1: norm2 + residual -> out
  auto ln_rows = [&](const float (&gamma)[2], const float (&beta)[2], int which) RD_INLINE_LAMBDA {
    for (int r = wv; r < LTOK; r += LNW) {
      if (r >= L) {
        if (which == 0) { Elem<T>::st(&sm.bM[r * LDA + lane], 0.f); Elem<T>::st(&sm.bM[r * LDA + 64 + lane], 0.f); }
        continue;
      }
      const float v0 = sm.u.fh.f[r * LF + lane], v1 = sm.u.fh.f[r * LF + 64 + lane];
      const float mu = wave_sum_up(v0 + v1) / (float)LC;
      const float d0 = v0 - mu, d1 = v1 - mu;
      const float var = wave_sum_up(d0 * d0 + d1 * d1) / (float)LC;
      const float rs = 1.0f / sqrtf(var + eps_ln);
      float o0 = d0 * rs * gamma[0] + beta[0], o1 = d1 * rs * gamma[1] + beta[1];
      const int64_t go = xo + (int64_t)r * LC;
      if (which == 0) {
        Elem<T>::st(&sm.bM[r * LDA + lane], o0); Elem<T>::st(&sm.bM[r * LDA + 64 + lane], o1);
        Elem<T>::st((T*)sv.msg + go + lane, o0); Elem<T>::st((T*)sv.msg + go + 64 + lane, o1);
      } else {
        o0 += Elem<T>::ld(&sm.bX[r * LDA + lane]); o1 += Elem<T>::ld(&sm.bX[r * LDA + 64 + lane]);
        Elem<T>::st(out + go + lane, o0); Elem<T>::st(out + go + 64 + lane, o1);
      }
      if (lane == 0) { sv.stats[((int64_t)n * L + r) * 4 + which * 2] = mu; sv.stats[((int64_t)n * L + r) * 4 + which * 2 + 1] = rs; }
    }
  };
  ln_rows(lg1, lb1, 0);
  __syncthreads();
  LPROF(5)

  // hidden = relu(W0 [x | message])
  tok_load<T, LC2, LC>(w.w2, W2);
  tok_mma<T, LC2, LC2>(sm.bX, sm.bM, LC, LDA, W0, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      const int tok = tt * 16 + fr;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; r++) v[r] = fmaxf(acc[tt][r], 0.f);
      st4(&sm.u.fh.h[tok * LDH + ct * 16 + fg * 4], v);
      if (tok < L) st4((T*)sv.hid + ((int64_t)n * L + tok) * LC2 + ct * 16 + fg * 4, v);
    }
  });
  __syncthreads();
  LPROF(6)
  tok_mma<T, LC2, LC>(sm.u.fh.h, sm.u.fh.h, LC2, LDH, W2, to_f((T*)sv.m2pre, L));
  __syncthreads();
  LPROF(7)
  ln_rows(lg2, lb2, 1);
  LPROF(8)
}

// ---- backward ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(LNT) void loftr_layer_bwd_kernel(const T* __restrict__ x, const T* __restrict__ src, LoftrW w,
                                                              LoftrSaved sv, LoftrGrads gr, int L, int S, float eps_attn) {
  constexpr int LDA = LoftrGeom<T>::LDA, LDH = LoftrGeom<T>::LDH;
  __shared__ LoftrSmemBwd<T> sm;
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, fr = lane & 15, fg = lane >> 4;
  const bool self = (src == x);
  const int64_t xo = (int64_t)n * L * LC, so = (int64_t)n * S * LC;

  // LayerNorm backward over the rows of this ROI.  d(row, c) supplies the upstream gradient, pre = saved pre-norm activation,
  // which selects the statistics; result -> bD (T) + saved copy for the weight gradient; per-ROI (sum d, sum d*xhat) -> lnp[n][c][2].
  auto ln_bwd = [&](auto dfn, const T* pre, const float (&gm)[2], int which, T* gout, float* lnp, bool seed_acc, auto after_loads) RD_INLINE_LAMBDA {
    float ag[2] = {0.f, 0.f}, ab[2] = {0.f, 0.f};
    constexpr int RPW = LTOK / LNW;   // rows per wave
    // every global value of this wave's rows is requested before the first reduction (clamped row: no branch around the loads);
    // row by row the loop was one round trip per row
    float dv_[RPW][2], pv_[RPW][2], mu_[RPW], rs_[RPW];
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int rc = min(wv + j * LNW, L - 1);
      const int64_t go = xo + (int64_t)rc * LC;
      mu_[j] = sv.stats[((int64_t)n * L + rc) * 4 + which * 2]; rs_[j] = sv.stats[((int64_t)n * L + rc) * 4 + which * 2 + 1];
#pragma unroll
      for (int e = 0; e < 2; e++) { dv_[j][e] = dfn(rc, e * 64 + lane); pv_[j][e] = Elem<T>::ld(pre + go + e * 64 + lane); }
    }
    sched_fence();
    after_loads();   // weight prefetch of the next GEMMs goes BEHIND this phase's own loads (loads return in order)
#pragma unroll
    for (int j = 0; j < RPW; j++) {
      const int r = wv + j * LNW;
      if (r >= L) {
        Elem<T>::st(&sm.bD[r * LDA + lane], 0.f); Elem<T>::st(&sm.bD[r * LDA + 64 + lane], 0.f);
        if (seed_acc) { sm.acc[r * LF + lane] = 0.f; sm.acc[r * LF + 64 + lane] = 0.f; }
        continue;
      }
      const int64_t go = xo + (int64_t)r * LC;
      const float mu = mu_[j], rs = rs_[j];
      float d[2], xh[2], g[2];
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int c = e * 64 + lane;
        d[e] = dv_[j][e];
        xh[e] = (pv_[j][e] - mu) * rs;
        g[e] = d[e] * gm[e];
        ag[e] += d[e] * xh[e]; ab[e] += d[e];
        if (seed_acc) sm.acc[r * LF + c] = d[e];   // residual path: dx starts as the upstream gradient
      }
      const float s1 = wave_sum_up(g[0] + g[1]) / (float)LC, s2 = wave_sum_up(g[0] * xh[0] + g[1] * xh[1]) / (float)LC;
#pragma unroll
      for (int e = 0; e < 2; e++) {
        const int c = e * 64 + lane;
        const float o = rs * (g[e] - s1 - xh[e] * s2);
        Elem<T>::st(&sm.bD[r * LDA + c], o);
        Elem<T>::st(gout + go + c, o);
      }
    }
#pragma unroll
    for (int e = 0; e < 2; e++) { sm.red[wv][0][e * 64 + lane] = ab[e]; sm.red[wv][1][e * 64 + lane] = ag[e]; }
    __syncthreads();
    if (threadIdx.x < LC) {
      const int c = threadIdx.x;
      float a = 0.f, b = 0.f;
      for (int q = 0; q < LNW; q++) { b += sm.red[q][0][c]; a += sm.red[q][1][c]; }
      lnp[((int64_t)n * LC + c) * 2] = b;       // dbeta terms
      lnp[((int64_t)n * LC + c) * 2 + 1] = a;   // dgamma terms
    }
  };

  LPROF_INIT
  TokW<T, LC, LC2> W2;
  TokW<T, LC2, LC2> W0;
  TokW<T, LC, LC> Wm, Wq, Wk, Wv;
  constexpr bool EARLY = sizeof(T) == 2;   // fp32 fragments are twice the registers: fewer GEMMs' weights in flight at once
  const float lg1[2] = {w.g1[lane], w.g1[64 + lane]}, lg2[2] = {w.g2[lane], w.g2[64 + lane]};
  // 1. out = x + norm2(m2pre)
  const T* dout = (const T*)gr.dout;
  ln_bwd([&](int r, int c) RD_INLINE_LAMBDA { return Elem<T>::ld(dout + xo + (int64_t)r * LC + c); }, (const T*)sv.m2pre, lg2, 1, (T*)gr.dm2pre, gr.lnp2, true,
         [&]() RD_INLINE_LAMBDA { tok_load<T, LC, LC2>(w.w2, W2); if (EARLY) tok_load<T, LC2, LC2>(w.w0, W0); });   // in flight across the norm2 backward
  __syncthreads();
  LPROF(10)

  // 2. dhid = (dm2pre W2) * relu'(hid)
  tok_mma<T, LC, LC2>(sm.bD, sm.bD, LC, LDA, W2, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      const int tok = tt * 16 + fr;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (tok < L) {
        float h[4];
        ld4((const T*)sv.hid + ((int64_t)n * L + tok) * LC2 + ct * 16 + fg * 4, h);
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = h[r] > 0.f ? acc[tt][r] : 0.f;
        st4((T*)gr.dhid + ((int64_t)n * L + tok) * LC2 + ct * 16 + fg * 4, v);
      }
      st4(&sm.u.fh.h[tok * LDH + ct * 16 + fg * 4], v);
    }
  });
  __syncthreads();
  LPROF(11)

  // 3. dcat = dhid W0: channels [0,128) add into dx, [128,256) are the gradient of the normalised message
  if (!EARLY) tok_load<T, LC2, LC2>(w.w0, W0);
  tok_mma<T, LC2, LC2>(sm.u.fh.h, sm.u.fh.h, LC2, LDH, W0, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      const int tok = tt * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int c = ct * 16 + fg * 4 + r;
        const float v = Elem<T>::rnd(acc[tt][r]);
        if (c < LC) sm.acc[tok * LF + c] += v; else sm.u.fh.f[tok * LF + c - LC] = v;
      }
    }
  });
  __syncthreads();
  LPROF(12)

  // 4. message = norm1(mpre)
  ln_bwd([&](int r, int c) RD_INLINE_LAMBDA { return sm.u.fh.f[r * LF + c]; }, (const T*)sv.mpre, lg1, 0, (T*)gr.dmpre, gr.lnp1, false,
         [&]() RD_INLINE_LAMBDA { tok_load<T, LC, LC>(w.wm, Wm); });
  __syncthreads();
  LPROF(13)

  // 5. datt = dmpre Wm
  auto to_global = [&](T* base, int64_t off, int rows) RD_INLINE_LAMBDA {
    return [=](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
      for (int tt = 0; tt < 2; tt++) {
        const int tok = tt * 16 + fr;
        if (tok < rows) {
          float v[4] = {acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]};
          st4(base + off + (int64_t)tok * LC + ct * 16 + fg * 4, v);
        }
      }
    };
  };
#if defined(RD_LOFTR_PROBE) && RD_LOFTR_PROBE >= 2
  tok_mma<T, LC, LC>(sm.bD, sm.bD, LC, LDA, Wm, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA { if (acc[0][0] == 12345.f) sm.acc[0] = 1.f; });
  __syncthreads();
  LPROF(14)
#elif !defined(RD_LOFTR_ATT_GLOBAL)
  // as in the forward: head h's datt tile is computed by the wave that owns head h -- stored for the weight gradient and written straight into
  // its attention scratch (rounded as stored); no barrier, the attention phase only adds its own q / k / v loads
  wave_sync();
  tok_mma<T, LC, LC>(sm.bD, sm.bD, LC, LDA, Wm, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
    float* const ld_ = sm.u.at[wv].d;
#pragma unroll
    for (int tt = 0; tt < 2; tt++) {
      const int tok = tt * 16 + fr;
      float v[4] = {acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]};
      if (tok < L) st4((T*)gr.datt + xo + (int64_t)tok * LC + ct * 16 + fg * 4, v);
#pragma unroll
      for (int r = 0; r < 4; r++) ld_[tok * LD + fg * 4 + r] = tok < L ? Elem<T>::rnd(v[r]) : 0.f;
    }
  });
  LPROF(14)
#else
  tok_mma<T, LC, LC>(sm.bD, sm.bD, LC, LDA, Wm, to_global((T*)gr.datt, xo, L));
  __syncthreads();
  LPROF(14)
#endif

  // 6. attention backward (recomputes KV / P from the saved q, k, v)
#if defined(RD_LOFTR_PROBE) && RD_LOFTR_PROBE == 4
  tok_load<T, LC, LC>(w.wq, Wq); if (EARLY) { tok_load<T, LC, LC>(w.wk, Wk); tok_load<T, LC, LC>(w.wv, Wv); }
  if (false)
#endif
  {
    auto hookb = [&]() RD_INLINE_LAMBDA { tok_load<T, LC, LC>(w.wq, Wq); if (EARLY) { tok_load<T, LC, LC>(w.wk, Wk); tok_load<T, LC, LC>(w.wv, Wv); } };
#if defined(RD_LOFTR_PROBE)
    attn_head<T, true, decltype(hookb), false, false>(sm.u.at[wv], (const T*)sv.q, (const T*)sv.k, (const T*)sv.v, (const T*)gr.dout, (T*)nullptr, (T*)gr.dq,
                                                        (T*)gr.dk, (T*)gr.dv, n, wv, true, L, S, LC, LC, LC, LC, eps_attn, hookb);
#elif !defined(RD_LOFTR_ATT_GLOBAL)
    attn_head<T, true, decltype(hookb), false, true>(sm.u.at[wv], (const T*)sv.q, (const T*)sv.k, (const T*)sv.v, (const T*)gr.datt, (T*)nullptr, (T*)gr.dq,
                                                       (T*)gr.dk, (T*)gr.dv, n, wv, true, L, S, LC, LC, LC, LC, eps_attn, hookb);
#else
    attn_head<T, true, decltype(hookb), false, false>(sm.u.at[wv], (const T*)sv.q, (const T*)sv.k, (const T*)sv.v, (const T*)gr.datt, (T*)nullptr, (T*)gr.dq,
                                                        (T*)gr.dk, (T*)gr.dv, n, wv, true, L, S, LC, LC, LC, LC, eps_attn, hookb);
#endif
  }
  __syncthreads();
  LPROF(15)

  // 7. dx += dq Wq;  dsrc = dk Wk + dv Wv  (self-attention: dsrc adds into dx)
  RowsReg<T> rdq, rdk, rdv;     // all three token matrices requested now; dk / dv arrive while the dq GEMM runs
  rows_fetch<T>((const T*)gr.dq + xo, L, rdq);
  rows_fetch<T>((const T*)gr.dk + so, S, rdk);
  rows_fetch<T>((const T*)gr.dv + so, S, rdv);
  sched_fence();
  rows_commit<T>(rdq, L, sm.bD, LDA);
  __syncthreads();
  tok_mma<T, LC, LC>(sm.bD, sm.bD, LC, LDA, Wq, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int r = 0; r < 4; r++) sm.acc[(tt * 16 + fr) * LF + ct * 16 + fg * 4 + r] += Elem<T>::rnd(acc[tt][r]);
  });
  __syncthreads();
  LPROF(16)
  float* sacc = self ? sm.acc : sm.u.fh.f;
  rows_commit<T>(rdk, S, sm.bD, LDA);
  __syncthreads();
  if (!EARLY) tok_load<T, LC, LC>(w.wk, Wk);
  tok_mma<T, LC, LC>(sm.bD, sm.bD, LC, LDA, Wk, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float* p = &sacc[(tt * 16 + fr) * LF + ct * 16 + fg * 4 + r];
        const float v = Elem<T>::rnd(acc[tt][r]);
        *p = self ? *p + v : v;
      }
  });
  __syncthreads();
  LPROF(17)
  rows_commit<T>(rdv, S, sm.bD, LDA);
  __syncthreads();
  if (!EARLY) tok_load<T, LC, LC>(w.wv, Wv);
  tok_mma<T, LC, LC>(sm.bD, sm.bD, LC, LDA, Wv, [&](int ct, f32x4 (&acc)[2]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int tt = 0; tt < 2; tt++)
#pragma unroll
      for (int r = 0; r < 4; r++) sacc[(tt * 16 + fr) * LF + ct * 16 + fg * 4 + r] += Elem<T>::rnd(acc[tt][r]);
  });
  __syncthreads();

  LPROF(18)
  // 8. results
  for (int idx = threadIdx.x; idx < L * (LC / 4); idx += LNT) {
    const int r = idx / (LC / 4), c4 = (idx - r * (LC / 4)) * 4;
    float v[4] = {sm.acc[r * LF + c4], sm.acc[r * LF + c4 + 1], sm.acc[r * LF + c4 + 2], sm.acc[r * LF + c4 + 3]};
    st4((T*)gr.dx + xo + (int64_t)r * LC + c4, v);
  }
  if (!self)
    for (int idx = threadIdx.x; idx < S * (LC / 4); idx += LNT) {
      const int r = idx / (LC / 4), c4 = (idx - r * (LC / 4)) * 4;
      float v[4] = {sm.u.fh.f[r * LF + c4], sm.u.fh.f[r * LF + c4 + 1], sm.u.fh.f[r * LF + c4 + 2], sm.u.fh.f[r * LF + c4 + 3]};
      if (gr.dsrc_accumulate) {      // the source's earlier gradient contribution rides along: sum rounded once, no separate add pass
        float o[4];
        ld4((const T*)gr.dsrc + so + (int64_t)r * LC + c4, o);
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] += o[e];
      }
      st4((T*)gr.dsrc + so + (int64_t)r * LC + c4, v);
    }
  LPROF(19)
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------------
void launch_loftr_layer_fwd(const void* x, const void* src, const LoftrW& w, void* out, const LoftrSaved& sv, int N, int L, int S,
                            float eps_attn, float eps_ln, int dtype, hipStream_t st) {
  if (N <= 0) return;
  if (dtype == 0) hipLaunchKernelGGL((loftr_layer_fwd_kernel<float>), dim3((unsigned)N), dim3(LNT), 0, st, (const float*)x, (const float*)src, w, (float*)out, sv, L, S, eps_attn, eps_ln);
  else hipLaunchKernelGGL((loftr_layer_fwd_kernel<bf16_t>), dim3((unsigned)N), dim3(LNT), 0, st, (const bf16_t*)x, (const bf16_t*)src, w, (bf16_t*)out, sv, L, S, eps_attn, eps_ln);
}
void launch_loftr_layer_bwd(const void* x, const void* src, const LoftrW& w, const LoftrSaved& sv, const LoftrGrads& gr, int N, int L,
                            int S, float eps_attn, int dtype, hipStream_t st) {
  if (N <= 0) return;
  if (dtype == 0) hipLaunchKernelGGL((loftr_layer_bwd_kernel<float>), dim3((unsigned)N), dim3(LNT), 0, st, (const float*)x, (const float*)src, w, sv, gr, L, S, eps_attn);
  else hipLaunchKernelGGL((loftr_layer_bwd_kernel<bf16_t>), dim3((unsigned)N), dim3(LNT), 0, st, (const bf16_t*)x, (const bf16_t*)src, w, sv, gr, L, S, eps_attn);
  // LayerNorm parameter gradients: ordered sum of the per-ROI partials
  if (gr.defer_ln) return;      // the caller finishes them later (rd_ln_grad_batch: one launch for every layer application of a stage)
  launch_bn_bwd_finalize(gr.lnp1, N, LC, 1.0, gr.dg1, gr.db1, gr.accumulate, nullptr, nullptr, st);
  launch_bn_bwd_finalize(gr.lnp2, N, LC, 1.0, gr.dg2, gr.db2, gr.accumulate, nullptr, nullptr, st);
}

}  // namespace rd

#ifdef RD_LOFTR_PROF
extern "C" int rd_debug_loftr_prof(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(rd::g_loftr_prof), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
  if (reset) { unsigned long long z[32] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rd::g_loftr_prof), z, sizeof(z)) != hipSuccess) return 1; }
  return 0;
}
#endif
