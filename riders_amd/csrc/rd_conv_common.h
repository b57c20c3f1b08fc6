// Device helpers shared by the convolution kernels (rd_conv.hip, rd_conv3x3.hip).
#pragma once
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

static constexpr int BM = 128;          // pixels per block tile
static constexpr int STAGE_BYTES = 128; // K bytes per row per stage

// LDS byte offset of 16-byte slot `slot` (0..7) of tile row `row`; XOR swizzle keeps the
// ds_read_b128 fragment reads (16 rows x 4 k-groups per wave) bank-conflict free.
__device__ __forceinline__ int lds_slot(int row, int slot) { return row * 8 + (slot ^ ((row >> 1) & 7)); }

// x / d for 0 <= x < 2^22 through the rounded reciprocal (the float product is off by less than one): the linear-strip tilings of
// rd_conv3x3_frag.hip / rd_wgrad3x3.hip decode pixel indices with it instead of ~40-instruction integer divisions
__device__ __forceinline__ int fdiv_small(int x, int d, float rd_, int& rem) {
  int q = (int)((float)x * rd_);
  int r = x - q * d;
  if (r < 0) { q--; r += d; }
  if (r >= d) { q++; r -= d; }
  rem = r;
  return q;
}

template <typename T>
__device__ __forceinline__ bool conv_src_ptr(const ConvArgs& a, int n, int ih, int iw, int ci, const T*& p) {
  if (a.dil > 1) {
    if (ih < 0 || iw < 0) return false;
    if ((ih % a.dil) | (iw % a.dil)) return false;
    ih /= a.dil; iw /= a.dil;
  }
  if ((unsigned)ih >= (unsigned)a.Hin || (unsigned)iw >= (unsigned)a.Win) return false;
  int hs = ih, ws = iw, Hp = a.Hin, Wp = a.Win;
  if (a.ups) {  // F.interpolate(mode='nearest') source index, ATen float formula
    hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
    ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
    Hp = a.H1; Wp = a.W1;
  }
  if (ci < a.C1) p = (const T*)a.src1 + ((((int64_t)n * Hp + hs) * Wp + ws) * a.C1 + ci);
  else p = (const T*)a.src2 + ((((int64_t)n * Hp + hs) * Wp + ws) * a.C2 + (ci - a.C1));
  return true;
}


// Branch-free form of conv_src_ptr for loaders that must issue their loads unconditionally: ALWAYS yields a dereferenceable pointer (the
// coordinates are clamped into the tensor) and returns whether the element exists.  No early returns: a load guarded by control flow is
// sunk into the branch by the compiler, which then cannot count the requests in flight any more.
template <typename T>
__device__ __forceinline__ bool conv_src_ptr_nb(const ConvArgs& a, int n, int ih, int iw, int ci, const T*& p) {
  bool ok = true;
  if (a.dil > 1) {      // uniform
    ok = ih >= 0 && iw >= 0 && ((ih % a.dil) | (iw % a.dil)) == 0;
    ih = (ih < 0 ? 0 : ih) / a.dil; iw = (iw < 0 ? 0 : iw) / a.dil;
  }
  ok = ok & ((unsigned)ih < (unsigned)a.Hin) & ((unsigned)iw < (unsigned)a.Win);
  ih = min(max(ih, 0), a.Hin - 1); iw = min(max(iw, 0), a.Win - 1);
  int hs = ih, ws = iw, Hp = a.Hin, Wp = a.Win;
  if (a.ups) {      // uniform
    hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
    ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
    Hp = a.H1; Wp = a.W1;
  }
  const bool first = ci < a.C1;
  const T* base = first ? (const T*)a.src1 : (const T*)a.src2;
  const int cc = first ? ci : ci - a.C1, cw = first ? a.C1 : a.C2;
  p = base + ((((int64_t)n * Hp + hs) * Wp + ws) * cw + cc);
  return ok;
}

// ---- consumer-side BatchNorm apply (ConvArgs::in_scale): one 16-byte vector of the producer's raw output y -> act(scale * y + shift),
// rounded to the activation type.  The expression is rd_affine_act's (rd_norm.hip, bn_apply1): the value a consumer stages is bit for bit
// the z that pass would have written.  sc / sh: the VE per-channel coefficients of this vector (registers or LDS).
__device__ __forceinline__ float bn_apply1(float y, float s, float b, int act, float slope) { return act_fwd(y * s + b, act, slope); }
__device__ __forceinline__ uint4 affine16(const float*, const uint4& r, const float* sc, const float* sh, int act, float slope) {
  uint4 o;
  o.x = __float_as_uint(bn_apply1(__uint_as_float(r.x), sc[0], sh[0], act, slope));
  o.y = __float_as_uint(bn_apply1(__uint_as_float(r.y), sc[1], sh[1], act, slope));
  o.z = __float_as_uint(bn_apply1(__uint_as_float(r.z), sc[2], sh[2], act, slope));
  o.w = __float_as_uint(bn_apply1(__uint_as_float(r.w), sc[3], sh[3], act, slope));
  return o;
}
__device__ __forceinline__ uint4 affine16(const bf16_t*, const uint4& r, const float* sc, const float* sh, int act, float slope) {
  float v[8];
  raw16_to_f32((const bf16_t*)nullptr, r, v);
#pragma unroll
  for (int e = 0; e < 8; e++) v[e] = bn_apply1(v[e], sc[e], sh[e], act, slope);
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}
// the block's copy of the coefficients: aff[c] = scale, aff[Cin + c] = shift for the Cin channels [c0, c0 + Cin) of the gather's K axis;
// channels of the second source (c >= C1) get the identity (they are never transformed, the values only keep the reads in bounds)
__device__ __forceinline__ void affine_fill(float* aff, const float* scale, const float* shift, int c0, int Cin, int C1, int t, int nt) {
  for (int i = t; i < Cin; i += nt) {
    const int c = c0 + i;
    const bool in1 = c < C1;
    const float s = scale[in1 ? c : 0], b = shift[in1 ? c : 0];
    aff[i] = in1 ? s : 1.f; aff[Cin + i] = in1 ? b : 0.f;
  }
}

// ---- shared epilogue: bias, activation, NHWC store (dual destination), BatchNorm (sum, sum^2) partials ------------------------------
// acc[c][pt][r] = output channel n0 + (wn*CT + c)*16 + fg*4 + r of pixel m[pt] (valid iff mv[pt]); WMV waves share the pixel axis.
// Part 1 stores the tile and ADDS its values into the caller's per-lane statistics registers; part 2 reduces those registers over
// the block and writes one statistics row.  One-tile-per-block kernels call both per tile; persistent kernels keep the registers across
// their tiles and call part 2 once (one row per block instead of one per tile: 45 000 -> 1024 rows on RC-Net's ROI-resolution layers).
// 4 consecutive channels: round to T once, hand back the rounded values (BatchNorm statistics are taken over what is stored)
__device__ __forceinline__ void round_store4(float* d, const float (&x)[4], float (&xr)[4]) {
  *reinterpret_cast<float4*>(d) = make_float4(x[0], x[1], x[2], x[3]);
#pragma unroll
  for (int r = 0; r < 4; r++) xr[r] = x[r];
}
__device__ __forceinline__ void round_store4(bf16_t* d, const float (&x)[4], float (&xr)[4]) {
  uint2 u;
  u.x = pack_bf16x2(x[0], x[1]); u.y = pack_bf16x2(x[2], x[3]);
  xr[0] = half_lo_f32(u.x); xr[1] = half_hi_f32(u.x);
  xr[2] = half_lo_f32(u.y); xr[3] = half_hi_f32(u.y);
  *reinterpret_cast<uint2*>(d) = u;
}

// The narrow RC-Net layers are VALU-bound in this routine (rocprofv3: ~380 VALU instructions per wave and 128-pixel tile, 4 cycles
// each), so everything uniform is decided once: no bias / no activation (every BatchNorm-ed convolution) skips both per element,
// invalid pixels skip the whole channel loop, the destination row pointers are formed once per pixel, and a value is rounded once.
// General form: accumulator c of the lane holds the 4 consecutive output channels cbase + c * CSTR .. + 3 (16x16x32 tiles: cbase = n0 + (wn CT) 16
// + 4 fg, CSTR = 16; 32x32x16 tiles: cbase = n0 + 32 (channel tile) + 4 (lane >> 5), CSTR = 8).
// BNB: compile the BatchNorm-backward sums (ConvArgs::bn_y) in -- only the kernels conv_bn_bwd_ok() names pass true: in the narrow-layer
// kernels, which sit at their register caps, the mere presence of that code cost 8-12 registers and 4-22 spilled ones (round 5, measured:
// 0.29 ms per RC-Net step).
template <typename T, int CT, bool ADD = true, int CSTR = 16, bool BNB = false>
__device__ __forceinline__ void conv_epilogue_store_at(const ConvArgs& a, f32x4 (&acc)[CT][2], const int64_t (&m)[2], const bool (&mv)[2], int cbase,
                                                       float (&ssum)[CT][4], float (&ssq)[CT][4]) {
  const int D2 = a.Cout - a.D1;
  const bool vec_ok = ((a.D1 & 3) == 0) && ((D2 & 3) == 0);
  const bool has_bias = a.bias != nullptr;
  const bool plain = !has_bias && a.act == ACT_NONE;
  // the lane's bias values, fetched ONCE and unconditionally (clamped index): inside the store loop every `if (co + r < Cout) x += bias[..]`
  // was a guarded scalar load followed by its own s_waitcnt -- up to 64 dependent memory round trips per tile on the FullyConnected layers
  float bv[CT][4];
  if (has_bias) {
#pragma unroll
    for (int c = 0; c < CT; c++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int cb = cbase + c * CSTR + r;
        const float b = a.bias[cb < a.Cout ? cb : a.Cout - 1];
        bv[c][r] = cb < a.Cout ? b : 0.f;
      }
  }
#pragma unroll
  for (int pt = 0; pt < 2; pt++) {
    if (!mv[pt]) continue;
    T* const p1 = (T*)a.dst1 + m[pt] * a.D1;
    T* const p2 = (T*)a.dst2 + m[pt] * D2 - a.D1;     // indexed by the global channel (only dereferenced for co >= D1)
#pragma unroll
    for (int c = 0; c < CT; c++) {
      const int co = cbase + c * CSTR;
      float x[4], xr[4];
#pragma unroll
      for (int r = 0; r < 4; r++) x[r] = acc[c][pt][r];
      if (!plain) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (has_bias) x[r] += bv[c][r];
          x[r] = act_fwd(x[r], a.act, a.slope);
        }
      }
      if (ADD && a.add1) {      // the tensor's earlier gradient contribution rides along: cur + this, rounded once (no separate add pass)
        const T* ap = (const T*)a.add1 + m[pt] * a.Cout + co;
        if (vec_ok && co + 3 < a.Cout) { float av[4]; ld4(ap, av);
#pragma unroll
          for (int r = 0; r < 4; r++) x[r] += av[r];
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) if (co + r < a.Cout) x[r] += Elem<T>::ld(ap + r);
        }
      }
      if (vec_ok && co + 3 < a.Cout) {
        round_store4((co < a.D1 ? p1 : p2) + co, x, xr);
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) {
          xr[r] = Elem<T>::rnd(x[r]);
          const int cc = co + r;
          if (cc < a.Cout) Elem<T>::st((cc < a.D1 ? p1 : p2) + cc, xr[r]);
        }
      }
      if (BNB && a.bn_y) {
        // data gradient whose first destination is dz of a BatchNorm-ed producer (ConvArgs::bn_y = its raw output, same layout as dst1): the
        // statistics registers collect the BatchNorm backward's sums over the STORED (rounded) dz instead -- g = dz * act'(scale y + shift),
        // (sum g, sum g * xhat) -- with col_reduce_vec_kernel's expressions (rd_norm.hip), so the producer's backward skips its reduce pass
        if (co + 3 < a.D1) {
          float yy[4];
          ld4((const T*)a.bn_y + m[pt] * a.D1 + co, yy);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int cc = co + r;
            float g = xr[r];
            if (a.bn_act) g *= act_grad_from_out(yy[r] * a.bn_scale[cc] + a.bn_shift[cc], a.bn_act, a.bn_slope);
            ssum[c][r] += g; ssq[c][r] += g * ((yy[r] - a.bn_mean[cc]) * a.bn_rstd[cc]);
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; r++) { ssum[c][r] += xr[r]; ssq[c][r] += xr[r] * xr[r]; }
      }
    }
  }
}
template <typename T, int CT, bool ADD = true, bool BNB = false>
__device__ __forceinline__ void conv_epilogue_store(const ConvArgs& a, f32x4 (&acc)[CT][2], const int64_t (&m)[2], const bool (&mv)[2], int n0,
                                                    int wn, int fr, int fg, float (&ssum)[CT][4], float (&ssq)[CT][4]) {
  conv_epilogue_store_at<T, CT, ADD, 16, BNB>(a, acc, m, mv, n0 + wn * CT * 16 + fg * 4, ssum, ssq);
}
template <int CT, int BN, int WMV, int NTH = 256>
__device__ __forceinline__ void conv_epilogue_stats(const ConvArgs& a, const float (&ssum)[CT][4], const float (&ssq)[CT][4], int n0, int wn,
                                                    int wm, int fr, int fg, int t, int64_t stats_row, float* red /* >= WMV*BN*2 floats */) {
  if (!a.stats) return;  // per-block partials, combined later in a fixed order (deterministic)
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      float s1 = ssum[c][r], s2 = ssq[c][r];
      s1 = row16_sum(s1); s2 = row16_sum(s2);   // over fr = lane & 15: DPP adds, the xor butterfly's additions (rd_common.h)
      if (fr == 0) {
        int col = (wn * CT + c) * 16 + fg * 4 + r;
        red[(wm * BN + col) * 2 + 0] = s1;
        red[(wm * BN + col) * 2 + 1] = s2;
      }
    }
  __syncthreads();
  for (int col = t; col < BN; col += NTH) {
    int co = n0 + col;
    if (co < a.Cout) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int w = 0; w < WMV; w++) { s1 += red[(w * BN + col) * 2]; s2 += red[(w * BN + col) * 2 + 1]; }
      a.stats[(stats_row * a.Cout + co) * 2 + 0] = s1;
      a.stats[(stats_row * a.Cout + co) * 2 + 1] = s2;
    }
  }
}
template <typename T, int CT, int BN, int WMV>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, f32x4 (&acc)[CT][2], const int64_t (&m)[2], const bool (&mv)[2], int n0,
                                              int wn, int wm, int fr, int fg, int t, int64_t stats_row, float* red /* >= WMV*BN*2 floats */) {
  float ssum[CT][4], ssq[CT][4];
#pragma unroll
  for (int c = 0; c < CT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) { ssum[c][r] = 0.f; ssq[c][r] = 0.f; }
  conv_epilogue_store<T, CT, true, true>(a, acc, m, mv, n0, wn, fr, fg, ssum, ssq);      // (conv_epilogue: the implicit-GEMM kernel)
  conv_epilogue_stats<CT, BN, WMV>(a, ssum, ssq, n0, wn, wm, fr, fg, t, stats_row, red);
}

}  // namespace rd
