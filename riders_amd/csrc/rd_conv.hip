// Implicit-GEMM convolution family for gfx950 (MFMA 16x16 tiles, LDS-staged, NHWC activations).
//
// Replaces the cuDNN/ATen convolutions dispatched by the reference at
//   utils/net_utils.py:84-91   (Conv2d.forward: conv -> BN -> act)
//   utils/net_utils.py:195-198 (UpConv2d: nearest interpolate -> conv; the upsample is folded
//                               into this kernel's gather, never materialised)
//   utils/net_utils.py:564-569 (DecoderBlock: cat([deconv, skip]) -> conv; the concat is folded
//                               into the gather as a second source tensor)
//   RCNet/linear_attention.py:121-131 (nn.Linear projections = 1x1 convolutions over tokens)
//   modules/midas/blocks.py (all dense convs of the Scale Map Learner)
//
// One kernel serves forward and data-gradient:
//   forward : out[m, co] = sum_{kh,kw,ci} X[n, oh*s-p+kh, ow*s-p+kw, ci] * W[co, kh, kw, ci]
//   dgrad   : the same gather over dY with stride 1, pad' = K-1-p, the input viewed as zero-dilated
//             by s (in_dilate) and the weights packed flipped/transposed by rd_conv_pack_weights.
// The weight-gradient kernel reduces over pixels into per-split slabs that a second kernel sums in
// a fixed order (deterministic), writing the reference's OIHW fp32 layout directly.
//
// GEMM view: rows = output pixels (block tile 128), cols = output channels (block tile BN),
// K = KH*KW*Cin walked in 128-byte stages (32 fp32 / 64 bf16 per row).  MFMA operand roles are
// swapped (A = weights, B = pixels) so each lane ends up holding 4 consecutive output channels of
// one pixel -> one 16-byte (fp32) / 8-byte (bf16) NHWC store per lane per tile.
#include "rd_conv_common.h"
#include <type_traits>
#include <stdio.h>

namespace rd {

// WM = waves along the pixel axis (block tile = 32*WM pixels x BN channels; the other 4/WM wave factor splits the channels).
// WM = 2 halves the tile for small-M launches (LoFTR projections, deep encoder stages) so they spread over more CUs.
// DEEP = two register sets, global loads issued two stages ahead: small-M launches with a long K axis (SML's 1x1 convolutions on
// 1.7-7 K pixels with up to 1392 channels) are a chain of stages each waiting one L2 round trip (~1.3 us per stage measured).
// PAR (round 4) = the data gradient of a STRIDE-2 layer (ConvArgs::dil == 2: dY viewed as zero-dilated): an output pixel only meets the
// taps whose parity matches its own -- (1, 2, 2, 4) of the 9 taps of a 3x3 layer for the four (row, column) parity classes, (1, 0, 0, 0)
// for a 1x1 projection -- but a tile of consecutive pixels mixes the classes and walked all nine (4x the staging and the MFMAs: 81 us for
// the 64 -> 128 stage against 20 us of its forward).  Here the pixel axis is ordered class by class (a block's tile lies in ONE class, the
// grid is the sum of the classes' tiles) and the K walk visits the class's taps only; channel counts are whole stages (Cin % BKE == 0).
__host__ __device__ __forceinline__ void par_class(int c, int OH, int OW, int N, int& ph, int& pw, int& oh2, int& ow2, int& mc) {
  ph = c >> 1; pw = c & 1;
  oh2 = (OH - ph + 1) >> 1; ow2 = (OW - pw + 1) >> 1;
  mc = N * oh2 * ow2;
}
template <typename T, int BN, bool VEC, int WM, bool DEEP = false, bool PAR = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs a) {
  constexpr int VE = Elem<T>::VE;
  constexpr int BKE = STAGE_BYTES / (int)sizeof(T);  // K elements per stage
  constexpr int WN = 4 / WM;
  constexpr int BMV = 32 * WM;                       // pixels per block tile
  constexpr int CT = BN / 16 / WN;                   // cout tiles per wave
  constexpr int BITER = (BN * 8 + 255) / 256;        // weight-tile vec loads per thread
  __shared__ uint4 sA[2][BMV * 8];
  __shared__ uint4 sB[2][BN * 8];

  const int t = threadIdx.x;
  const int lane = t & 63, wv = t >> 6;
  // XCD-aware tile order: the dispatcher places block b on XCD b % 8 (speed only, never correctness), so give every XCD a CONTIGUOUS
  // range of pixel tiles -- neighbouring tiles share their 3x3 halo rows and then hit the same 4 MiB L2 (bijective remap; +3 % measured).
  int bx = blockIdx.x;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bx & 7, idx = bx >> 3;
    bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  int m0 = bx * BMV;
  const int wm = wv % WM, wn = wv / WM;
  const int n0 = blockIdx.y * BN;
  const int Cin = a.C1 + a.C2;
  // PAR: this block's parity class, its pixel count and the class's taps kh0 + 2 i, kw0 + 2 j
  int ph = 0, pw = 0, oh2 = 1, ow2 = 1, mcl = a.M, kh0 = 0, kw0 = 0, nkw = 1, spt = 1, npar = 0;
  if (PAR) {
    int c = 0, tb = bx;
    for (; c < 4; c++) {
      par_class(c, a.OH, a.OW, a.N, ph, pw, oh2, ow2, mcl);
      const int tc = (mcl + BMV - 1) / BMV;
      if (tb < tc || c == 3) break;
      tb -= tc;
    }
    m0 = tb * BMV;
    kh0 = (ph + a.pad) & 1; kw0 = (pw + a.pad) & 1;
    const int nkh = kh0 < a.KH ? (a.KH - kh0 + 1) >> 1 : 0;
    nkw = kw0 < a.KW ? (a.KW - kw0 + 1) >> 1 : 0;
    spt = Cin / BKE;
    npar = nkh * nkw * spt;
  }
  // pixel of local row ml (PAR: row of the class) -> image, row, column; false past the end
  auto row_pixel = [&](int ml, int& n, int& oh, int& ow) RD_INLINE_LAMBDA {
    if (PAR) {
      const int c2 = ml % ow2, q = ml / ow2, r2 = q % oh2;
      n = q / oh2; oh = 2 * r2 + ph; ow = 2 * c2 + pw;
      return ml < mcl;
    }
    ow = ml % a.OW; const int q = ml / a.OW; oh = q % a.OH; n = q / a.OH;
    return ml < a.M;
  };

  // ---- per-thread staging state: slot s of rows r0+32*i --------------------------------------
  const int s = t & 7, r0 = t >> 3;
  int rn[WM], rih[WM], riw[WM];
#pragma unroll
  for (int i = 0; i < WM; i++) {
    int n, oh, ow;
    if (row_pixel(m0 + r0 + 32 * i, n, oh, ow)) {
      rn[i] = n; rih[i] = oh * a.stride - a.pad; riw[i] = ow * a.stride - a.pad;
    } else { rn[i] = -1; rih[i] = 0; riw[i] = 0; }
  }
  // k-state of this thread's vector (VEC path): k = kt*BKE + s*VE
  int kci = 0, kkw = 0, kkh = 0;
  if (VEC) {
    int k = s * VE; int tap = k / Cin; kci = k - tap * Cin; kkh = tap / a.KW; kkw = tap - kkh * a.KW;
  }

  uint4 ra[WM]; uint4 rb[BITER];
  uint4 ra2[DEEP ? WM : 1]; uint4 rb2[DEEP ? BITER : 1];
  uint4 ra3[DEEP ? WM : 1]; uint4 rb3[DEEP ? BITER : 1];
  uint4 ra4[DEEP ? WM : 1]; uint4 rb4[DEEP ? BITER : 1];

  const int nk_ = PAR ? npar : a.Kpad / BKE;      // stages; a request past the last one (issued unconditionally by the deep path) re-reads the last weights
  auto load_tile = [&](int kt, uint4 (&ra)[WM], uint4 (&rb)[BITER]) RD_INLINE_LAMBDA {
    if (PAR) {      // logical stage -> (tap of the class, channel stage) -> physical stage of the packed operand
      const int j = min(kt, nk_ - 1), tp = j / spt, js = j - tp * spt, th = tp / nkw;
      kkh = kh0 + 2 * th; kkw = kw0 + 2 * (tp - th * nkw); kci = js * BKE + s * VE;
      kt = (kkh * a.KW + kkw) * spt + js;
    }
#pragma unroll
    for (int i = 0; i < WM; i++) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (VEC) {
        // unconditional load (an element that does not exist reads the tensor's first vector and is zeroed afterwards): behind a branch
        // the compiler cannot count the requests in flight and waits for ALL of them (vmcnt(0)) before every LDS store, which turned the
        // multi-stage prefetch of the small-M path into one memory round trip per stage
        const T* p;
        const bool inb = conv_src_ptr_nb<T>(a, rn[i] >= 0 ? rn[i] : 0, rih[i] + kkh, riw[i] + kkw, kkh < a.KH ? kci : 0, p);
        const bool ok = inb & (rn[i] >= 0) & (kkh < a.KH);
        const uint4 lv = *reinterpret_cast<const uint4*>(p);
        v.x = ok ? lv.x : 0u; v.y = ok ? lv.y : 0u; v.z = ok ? lv.z : 0u; v.w = ok ? lv.w : 0u;
      } else if (rn[i] >= 0) {
        {
          T tmp[VE];
#pragma unroll
          for (int e = 0; e < VE; e++) {
            int k = kt * BKE + s * VE + e;
            float f = 0.f;
            if (k < a.K) {
              int tap = k / Cin; int ci = k - tap * Cin; int kh = tap / a.KW; int kw = tap - kh * a.KW;
              const T* p;
              if (conv_src_ptr<T>(a, rn[i], rih[i] + kh, riw[i] + kw, ci, p)) f = Elem<T>::ld(p);
            }
            Elem<T>::st(&tmp[e], f);
          }
          v = *reinterpret_cast<uint4*>(tmp);
        }
      }
      ra[i] = v;
    }
    const uint4* wp = reinterpret_cast<const uint4*>(a.w);
    const int kslots = a.Kpad / VE;  // 16-byte slots per packed weight row
#pragma unroll
    for (int i = 0; i < BITER; i++) {
      int idx = t + 256 * i;
      if (BN * 8 % 256 == 0 || idx < BN * 8) {  // compile-time when the tile divides evenly: a runtime guard parks rb[] in scratch
        int row = idx >> 3, sl = idx & 7;
        const uint4 v = wp[(int64_t)(n0 + row) * kslots + (PAR ? kt : min(kt, nk_ - 1)) * 8 + sl];  // via a value: a direct global->array struct copy stays a memcpy
        rb[i] = v;                                                        // through a private alloca (scratch / LDS-promoted)
      }
    }
    if (VEC && !PAR) {  // advance this thread's k-state by one stage
      kci += BKE;
      while (kci >= Cin) { kci -= Cin; if (++kkw == a.KW) { kkw = 0; ++kkh; } }
    }
  };
  auto store_tile = [&](int buf, const uint4 (&ra)[WM], const uint4 (&rb)[BITER]) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < WM; i++) sA[buf][lds_slot(r0 + 32 * i, s)] = ra[i];
#pragma unroll
    for (int i = 0; i < BITER; i++) {
      int idx = t + 256 * i;
      if (BN * 8 % 256 == 0 || idx < BN * 8) sB[buf][lds_slot(idx >> 3, idx & 7)] = rb[i];
    }
  };

  f32x4 acc[CT][2];
#pragma unroll
  for (int c = 0; c < CT; c++) { acc[c][0] = f32x4{0, 0, 0, 0}; acc[c][1] = f32x4{0, 0, 0, 0}; }

  const int nk = nk_;
  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int buf) RD_INLINE_LAMBDA {
#pragma unroll
    for (int ch = 0; ch < 2; ch++) {  // two 64-byte chunks per stage
      uint4 pf[2];
#pragma unroll
      for (int pt = 0; pt < 2; pt++) pf[pt] = sA[buf][lds_slot(wm * 32 + pt * 16 + fr, ch * 4 + fg)];
#pragma unroll
      for (int c = 0; c < CT; c++) {
        uint4 wf = sB[buf][lds_slot((wn * CT + c) * 16 + fr, ch * 4 + fg)];
#pragma unroll
        for (int pt = 0; pt < 2; pt++) {
          if (sizeof(T) == 4) {
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.x), __uint_as_float(pf[pt].x), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.y), __uint_as_float(pf[pt].y), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.z), __uint_as_float(pf[pt].z), acc[c][pt]);
            acc[c][pt] = mfma_16x16x4_f32(__uint_as_float(wf.w), __uint_as_float(pf[pt].w), acc[c][pt]);
          } else {
            s16x8 wa, pb;
            __builtin_memcpy(&wa, &wf, 16);
            __builtin_memcpy(&pb, &pf[pt], 16);
            acc[c][pt] = mfma_16x16x32_bf16(wa, pb, acc[c][pt]);
          }
        }
      }
    }
  };
  if (PAR && nk == 0) {
    // a class without taps (the odd rows / columns of a 1x1 stride-2 projection): zeros (+ the addend) are stored below
  } else if (!DEEP) {
    load_tile(0, ra, rb);
    store_tile(0, ra, rb);
    __syncthreads();
    for (int kt = 0; kt < nk; kt++) {
      const int buf = kt & 1;
      if (kt + 1 < nk) load_tile(kt + 1, ra, rb);
      compute(buf);
      if (kt + 1 < nk) store_tile(buf ^ 1, ra, rb);
      __syncthreads();
    }
  } else {
    // FOUR register sets: stages kt+1 .. kt+4 are in flight while stage kt is multiplied (two sets left every stage waiting ~half an L2
    // round trip: the small-M / long-K launches this path serves are pure latency chains).  Stage kt sits in LDS buffer kt & 1.
    auto& a1 = ra; auto& b1 = rb;
    auto& a2 = reinterpret_cast<uint4 (&)[WM]>(ra2); auto& b2 = reinterpret_cast<uint4 (&)[BITER]>(rb2);
    auto& a3 = reinterpret_cast<uint4 (&)[WM]>(ra3); auto& b3 = reinterpret_cast<uint4 (&)[BITER]>(rb3);
    auto& a4 = reinterpret_cast<uint4 (&)[WM]>(ra4); auto& b4 = reinterpret_cast<uint4 (&)[BITER]>(rb4);
    // every load_tile below is issued UNCONDITIONALLY (stages past the end fetch clamped, unused data): only then does the compiler know
    // how many requests are younger than the set it is about to store and waits for exactly that set
    load_tile(0, a1, b1);
    store_tile(0, a1, b1);
    load_tile(1, a1, b1);
    load_tile(2, a2, b2);
    load_tile(3, a3, b3);
    load_tile(4, a4, b4);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 4) {
      compute(0);
      store_tile(1, a1, b1);          // stage kt + 1 (garbage past the end, never multiplied)
      __syncthreads();
      load_tile(kt + 5, a1, b1);
      if (kt + 1 >= nk) break;
      compute(1);
      store_tile(0, a2, b2);
      __syncthreads();
      load_tile(kt + 6, a2, b2);
      if (kt + 2 >= nk) break;
      compute(0);
      store_tile(1, a3, b3);
      __syncthreads();
      load_tile(kt + 7, a3, b3);
      if (kt + 3 >= nk) break;
      compute(1);
      store_tile(0, a4, b4);
      __syncthreads();
      load_tile(kt + 8, a4, b4);
    }
  }

  // ---- epilogue (shared): bias, activation, NHWC store (dual destination), BN statistics ----------------
  int64_t mm[2]; bool mvv[2];
#pragma unroll
  for (int pt = 0; pt < 2; pt++) {
    mm[pt] = m0 + wm * 32 + pt * 16 + fr; mvv[pt] = mm[pt] < a.M;
    if (PAR) {
      int n, oh, ow;
      mvv[pt] = row_pixel((int)mm[pt], n, oh, ow);
      mm[pt] = ((int64_t)n * a.OH + oh) * a.OW + ow;
    }
  }
  conv_epilogue<T, CT, BN, WM>(a, acc, mm, mvv, n0, wn, wm, fr, fg, t, bx, reinterpret_cast<float*>(&sA[0][0]));
}

// ---- weight packing -------------------------------------------------------------------------------
// OIHW fp32 (the reference's parameter layout) -> [rows_pad][Kpad] of T with k = (kh*KW+kw)*C + c.
// mode 0 (forward): rows = Cout, C = Cin.   mode 1 (dgrad): rows = Cin, C = Cout, taps flipped.
// 3x3 operands whose channel axis is a whole number of 128-byte chunks carry a SECOND copy behind the row-major one, in MFMA fragment
// order for conv3x3_frag_kernel (rd_conv3x3_frag.hip): [chunk][tap][16-row tile][k half][lane = 16 * k-group + row] x 16 bytes, so that a
// wave's fragment of one (chunk, tap, row tile, k half) is one contiguous 1-KiB load.  rd_conv_packed_elems sizes the buffer for both.
__host__ __device__ __forceinline__ bool pack_has_frag(int KH, int KW, int C, int dtype) { return KH == 3 && KW == 3 && (C % (dtype == 0 ? 32 : 64)) == 0; }
template <typename T>
__device__ __forceinline__ int64_t pack_frag_index(int row, int k, int C, int rows_pad, int tap_ = -1, int c_ = 0) {
  constexpr int VE = Elem<T>::VE, CKE = 8 * VE;
  const int tap = tap_ >= 0 ? tap_ : k / C, c = tap_ >= 0 ? c_ : k - tap * C;
  const int chunk = c / CKE, kk = c - chunk * CKE, kh = kk / (4 * VE), kg = (kk % (4 * VE)) / VE, e = kk % VE;
  return ((((int64_t)(chunk * 9 + tap) * (rows_pad >> 4) + (row >> 4)) * 2 + kh) * 64 + kg * 16 + (row & 15)) * VE + e;
}
// Linear / 1x1 operands of the fused LoFTR layer's sizes (128 or 256 channels on either side) also carry a second copy, in the fragment
// order of its token GEMMs (rd_loftr.hip tok_load): [16-row tile][k step of 4 vectors][lane = 16 * k-group + row] x 16 bytes.  Row-major, a
// wave's fragment load touched 16 rows x 64 bytes -- half a cache line per row, the other half wanted one k step later by which time the
// eight waves' 256 lines had passed through a 128-line L1: probe builds (profiles/r04_microbench/loftr_weight_stream.txt) put 6 of the
// forward's 22 us on the weight stream, and 3 of them on this access shape alone.
__host__ __device__ __forceinline__ bool pack_has_tokfrag(int rows, int K) { return (rows == 128 || rows == 256) && (K == 128 || K == 256); }
template <typename T>
__device__ __forceinline__ int64_t pack_tokfrag_index(int row, int k, int K) {
  constexpr int VE = Elem<T>::VE, SE = 4 * VE;
  const int ks = k / SE, kk = k - ks * SE, kg = kk / VE, e = kk - kg * VE;
  return ((((int64_t)(row >> 4) * (K / SE) + ks) * 64) + kg * 16 + (row & 15)) * VE + e;
}
// mode 2 (forward operand of an exact-2x nearest up-sampling 3x3 layer computed ON THE SOURCE, ConvArgs::d2s): row = (class (a, b), co) of 4 Cout rows,
// k = (source tap (kh', kw'), ci).  Output pixel (2 i + a, 2 j + b) reads source pixel (i + floor((a + kh - 1) / 2), ...) through tap kh, so source tap
// kh' collects a = 0: kh' = 0 <- {0}, kh' = 1 <- {1, 2};  a = 1: kh' = 1 <- {0, 1}, kh' = 2 <- {2}  (columns likewise with b): the class's 2x2
// effective kernel, zero elsewhere.  Summed in fp32 in ascending (kh, kw) order, rounded once when stored.
__device__ __forceinline__ float pack_up2_value(const float* __restrict__ w, int Cout, int Cin, int row, int c, int khs, int kws) {
  const int cls = row / Cout, co = row - cls * Cout, a = cls >> 1, b = cls & 1;
  const int h0 = a == 0 ? (khs == 0 ? 0 : (khs == 1 ? 1 : 3)) : (khs == 0 ? 3 : (khs == 1 ? 0 : 2));      // first original tap row (3: none)
  const int h1 = a == 0 ? (khs == 0 ? 0 : (khs == 1 ? 2 : -1)) : (khs == 0 ? -1 : (khs == 1 ? 1 : 2));    // last
  const int w0 = b == 0 ? (kws == 0 ? 0 : (kws == 1 ? 1 : 3)) : (kws == 0 ? 3 : (kws == 1 ? 0 : 2));
  const int w1 = b == 0 ? (kws == 0 ? 0 : (kws == 1 ? 2 : -1)) : (kws == 0 ? -1 : (kws == 1 ? 1 : 2));
  float v = 0.f;
  for (int kh = h0; kh <= h1; kh++)
    for (int kw = w0; kw <= w1; kw++) v += w[(((int64_t)co * Cin + c) * 3 + kh) * 3 + kw];
  return v;
}
template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int KH,
                                    int KW, int mode, int rows_pad, int Kpad, int CinSrc) {
  int64_t total = (int64_t)rows_pad * Kpad;
  int rows = mode == 2 ? 4 * Cout : (mode ? Cin : Cout), C = mode == 3 ? 4 * Cout : (mode == 1 ? Cout : Cin);
  int K = KH * KW * C;
  const bool frag = pack_has_frag(KH, KW, C, sizeof(T) == 4 ? 0 : 1);
  const bool tokfrag = KH == 1 && KW == 1 && pack_has_tokfrag(rows, K);      // rows_pad == rows, Kpad == K at these sizes
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int row = (int)(i / Kpad), k = (int)(i - (int64_t)row * Kpad);
    float v = 0.f;
    if (row < rows && k < K) {
      int tap = k / C, c = k - tap * C, kh = tap / KW, kw = tap - kh * KW;
      // CinSrc < Cin: the packed layout carries zero-padded input channels (3-channel stems run on the 16-byte-vector paths)
      if (mode == 0) v = c < CinSrc ? w[(((int64_t)row * CinSrc + c) * KH + kh) * KW + kw] : 0.f;
      else if (mode == 2) v = pack_up2_value(w, Cout, Cin, row, c, kh, kw);
      else if (mode == 3) v = pack_up2_value(w, Cout, Cin, c, row, 2 - kh, 2 - kw);      // (class, channel) column c of the flipped tap, input channel = row
      else v = w[(((int64_t)c * Cin + row) * KH + (KH - 1 - kh)) * KW + (KW - 1 - kw)];
    }
    Elem<T>::st(&out[i], v);
    if (frag) Elem<T>::st(&out[total + pack_frag_index<T>(row, k, C, rows_pad)], v);    // Kpad == K here
    if (tokfrag) Elem<T>::st(&out[total + pack_tokfrag_index<T>(row, k, K)], v);
  }
}

// batch form: blockIdx.y = item (device descriptor table), blockIdx.x strides over the item's packed elements
struct PackItem { const float* w; void* out; int Cout, Cin, KH, KW, mode, dtype, CinSrc, pad_; };
__device__ __forceinline__ bool rd_pack_vec_enabled(const PackItem& it) { return it.pad_ == 0; }      // (rd_pack_item.reserved = 1: the element-wise form, for A/B)
template <typename T>
__device__ __forceinline__ void pack_one(const PackItem& it, int bx, int nbx) {
  const int rows = it.mode == 2 ? 4 * it.Cout : (it.mode ? it.Cin : it.Cout), C = it.mode == 3 ? 4 * it.Cout : (it.mode == 1 ? it.Cout : it.Cin);
  const int K = it.KH * it.KW * C;
  const int bn = rows <= 16 ? 16 : (rows <= 32 ? 32 : (rows <= 64 ? 64 : 128));
  const int rows_pad = (rows + bn - 1) / bn * bn;
  const int bke = sizeof(T) == 4 ? 32 : 64;
  const int Kpad = (K + bke - 1) / bke * bke;
  const int64_t total = (int64_t)rows_pad * Kpad;
  const bool frag = pack_has_frag(it.KH, it.KW, C, sizeof(T) == 4 ? 0 : 1);
  const bool tokfrag = it.KH == 1 && it.KW == 1 && pack_has_tokfrag(rows, K);
  T* out = (T*)it.out;
  // (row, k) advance with the element index instead of being divided out of it, and k -> (tap, channel) goes through a rounded float
  // reciprocal (k < 2^22): 93 -> 70 us per RC-Net step.  (The SML step's 210 us are the strided OIHW reads -- one cache line per four
  // useful bytes; a source-tiled LDS transpose was measured at 183 us there and 105-128 us on RC-Net's many small operands: not kept.)
  const int64_t stride = (int64_t)nbx * 256;
  const int srow = (int)(stride / Kpad), sk = (int)(stride - (int64_t)srow * Kpad);
  const float rC = 1.0f / (float)C, rKW = 1.0f / (float)it.KW;
  const int cs = it.CinSrc > 0 ? it.CinSrc : it.Cin;
  int64_t i = (int64_t)bx * 256 + threadIdx.x;
  int row = (int)(i / Kpad), k = (int)(i - (int64_t)row * Kpad);
  for (; i < total; i += stride) {
    float v = 0.f;
    int c, kw;
    const int tap = fdiv_small(k, C, rC, c), kh = fdiv_small(tap, it.KW, rKW, kw);
    if (row < rows && k < K) {
      if (it.mode == 0) v = c < cs ? it.w[(((int64_t)row * cs + c) * it.KH + kh) * it.KW + kw] : 0.f;
      else if (it.mode == 2) v = pack_up2_value(it.w, it.Cout, it.Cin, row, c, kh, kw);
      else if (it.mode == 3) v = pack_up2_value(it.w, it.Cout, it.Cin, c, row, 2 - kh, 2 - kw);
      else v = it.w[(((int64_t)c * it.Cin + row) * it.KH + (it.KH - 1 - kh)) * it.KW + (it.KW - 1 - kw)];
    }
    Elem<T>::st(&out[i], v);
    if (frag) Elem<T>::st(&out[total + pack_frag_index<T>(row, k, C, rows_pad, tap, c)], v);
    if (tokfrag) Elem<T>::st(&out[total + pack_tokfrag_index<T>(row, k, K)], v);
    row += srow; k += sk;
    if (k >= Kpad) { k -= Kpad; row++; }
  }
}
// Round 6: the same packing in 16-byte units with coalesced reads (the batch kernel was 84 us per RC-Net step and ~300 us per SML step -- 9 x
// the operands' bytes -- because every 2-byte output element was its own strided 4-byte OIHW read and its own 2-byte store, and the fragment-
// ordered copies were scattered 2-byte stores).  A work item is one (packed row, group of VE consecutive channels of the packed K axis) over
// ALL taps of a 1x1 or 3x3 kernel: VE x KK source values, KK 16-byte stores per layout.
//   mode 0 / 2 (rows follow Cout): the VE x KK values of an item are ONE contiguous run of the OIHW tensor (VE input channels x KK taps of one
//     output channel): read as 16-byte vectors, consecutive items (channel groups of a row) = consecutive runs.
//   mode 1 / 3 (rows follow Cin, the transposed operand of the data gradient): an item needs KK-float pieces of VE different output channels;
//     consecutive threads take consecutive ROWS (input channels), whose pieces are adjacent in memory.
// Bit-identical to pack_one (same fp32 sums in the same order for the pre-summed up-convolution taps, one rounding) -- pack_batch_case.
template <typename T>
__device__ __forceinline__ bool pack_vec_ok(const PackItem& it) {
  constexpr int VE = Elem<T>::VE;
  const int C = it.mode == 3 ? 4 * it.Cout : (it.mode == 1 ? it.Cout : it.Cin);
  const int KK = it.KH * it.KW;
  if (!(KK == 1 || (it.KH == 3 && it.KW == 3))) return false;
  if (it.CinSrc > 0 && it.CinSrc != it.Cin) return false;      // zero-padded stems keep the element-wise form
  if (C % VE || (it.Cin % 4) || ((uintptr_t)it.w & 15) || ((uintptr_t)it.out & 15)) return false;      // 16-byte source runs and destination units
  if ((it.mode == 2 || it.mode == 3) && (KK != 9 || (it.Cout % VE))) return false;
  return true;
}
template <typename T>
__device__ __forceinline__ void pack_one_vec(const PackItem& it, int bx, int nbx) {
  constexpr int VE = Elem<T>::VE;
  const int rows = it.mode == 2 ? 4 * it.Cout : (it.mode ? it.Cin : it.Cout), C = it.mode == 3 ? 4 * it.Cout : (it.mode == 1 ? it.Cout : it.Cin);
  const int KK = it.KH * it.KW, K = KK * C;
  const int bn = rows <= 16 ? 16 : (rows <= 32 ? 32 : (rows <= 64 ? 64 : 128));
  const int rows_pad = (rows + bn - 1) / bn * bn;
  const int bke = sizeof(T) == 4 ? 32 : 64;
  const int Kpad = (K + bke - 1) / bke * bke;
  const int64_t total = (int64_t)rows_pad * Kpad;
  const bool frag = pack_has_frag(it.KH, it.KW, C, sizeof(T) == 4 ? 0 : 1);
  const bool tokfrag = KK == 1 && pack_has_tokfrag(rows, K);
  T* out = (T*)it.out;
  const float* __restrict__ w = it.w;
  const int CG = C / VE;                                   // channel groups per row
  const bool by_row = it.mode == 1 || it.mode == 3;        // source rows follow Cin (the transposed operand)
  // A wave owns a tile of 8 rows x 8 channel groups, lane = 8 * group + row: the row-major stores of one tap are 8 rows x 128 contiguous bytes,
  // the fragment-ordered stores 8 groups x 128 contiguous bytes (16 consecutive rows of a k-group are adjacent there), the source reads are
  // 8 adjacent runs (modes 0 / 2: 8 x VE x KK floats of one output channel; modes 1 / 3: KK floats of 8 adjacent input channels).  Whole
  // 128-byte lines in every store is what matters: with scattered 16-byte stores the launch was SLOWER than the element-wise form on
  // RC-Net's many small operands (104 vs 84 us; profiles/r06_microbench/pack_vec.txt).
  const int tiles_c = (CG + 7) >> 3, tiles_r = (rows + 7) >> 3;
  const int64_t items = (int64_t)tiles_r * tiles_c * 64;
  for (int64_t u = (int64_t)bx * 256 + threadIdx.x; u < items; u += (int64_t)nbx * 256) {
    const int64_t tile = u >> 6;
    const int ln = (int)(u & 63), tr = (int)(tile / tiles_c), tc = (int)(tile - (int64_t)tr * tiles_c);
    const int row = tr * 8 + (ln & 7), cg = tc * 8 + (ln >> 3);
    if (row >= rows || cg >= CG) continue;
    const int c0 = cg * VE;
    if (KK == 1) {
      float v[VE];
      if (it.mode == 0) {
#pragma unroll
        for (int q = 0; q < VE / 4; q++) { const float4 f = *reinterpret_cast<const float4*>(w + (int64_t)row * it.Cin + c0 + 4 * q); v[4 * q] = f.x; v[4 * q + 1] = f.y; v[4 * q + 2] = f.z; v[4 * q + 3] = f.w; }
      } else {
#pragma unroll
        for (int e = 0; e < VE; e++) v[e] = w[(int64_t)(c0 + e) * it.Cin + row];
      }
      stv(out + (int64_t)row * Kpad + c0, v);
      if (tokfrag) stv(out + total + pack_tokfrag_index<T>(row, c0, K), v);
      continue;
    }
    // 3x3: src[e][t] = w[co][ci][t] of the VE (co, ci) pairs this item touches, t = kh * 3 + kw of the ORIGINAL kernel
    float src[VE][9];
    if (!by_row) {      // modes 0 / 2: one contiguous run of VE x 9 floats
      const int co = it.mode == 2 ? row % it.Cout : row;
      const float4* run = reinterpret_cast<const float4*>(w + ((int64_t)co * it.Cin + c0) * 9);
      float flat[VE * 9];
#pragma unroll
      for (int q = 0; q < VE * 9 / 4; q++) { const float4 f = run[q]; flat[4 * q] = f.x; flat[4 * q + 1] = f.y; flat[4 * q + 2] = f.z; flat[4 * q + 3] = f.w; }
#pragma unroll
      for (int e = 0; e < VE; e++)
#pragma unroll
        for (int t = 0; t < 9; t++) src[e][t] = flat[e * 9 + t];
    } else {            // modes 1 / 3: nine floats of each of VE output channels, input channel = row
      const int co0 = it.mode == 3 ? c0 % it.Cout : c0;
#pragma unroll
      for (int e = 0; e < VE; e++) {
        const float* pw = w + ((int64_t)(co0 + e) * it.Cin + row) * 9;
#pragma unroll
        for (int t = 0; t < 9; t++) src[e][t] = pw[t];
      }
    }
    const int cls = it.mode == 2 ? row / it.Cout : (it.mode == 3 ? c0 / it.Cout : 0), ca = cls >> 1, cb = cls & 1;
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {      // packed tap (kh, kw) of this layout
      const int kh = tap / 3, kw = tap - kh * 3;
      float v[VE];
      if (it.mode == 0) {
#pragma unroll
        for (int e = 0; e < VE; e++) v[e] = src[e][tap];
      } else if (it.mode == 1) {
#pragma unroll
        for (int e = 0; e < VE; e++) v[e] = src[e][(2 - kh) * 3 + (2 - kw)];
      } else {
        // pre-summed taps of parity class (ca, cb) (pack_up2_value): source tap (khs, kws) collects original rows h0..h1, columns w0..w1
        const int khs = it.mode == 2 ? kh : 2 - kh, kws = it.mode == 2 ? kw : 2 - kw;
        const int h0 = ca == 0 ? (khs == 0 ? 0 : (khs == 1 ? 1 : 3)) : (khs == 0 ? 3 : (khs == 1 ? 0 : 2));
        const int h1 = ca == 0 ? (khs == 0 ? 0 : (khs == 1 ? 2 : -1)) : (khs == 0 ? -1 : (khs == 1 ? 1 : 2));
        const int w0 = cb == 0 ? (kws == 0 ? 0 : (kws == 1 ? 1 : 3)) : (kws == 0 ? 3 : (kws == 1 ? 0 : 2));
        const int w1 = cb == 0 ? (kws == 0 ? 0 : (kws == 1 ? 2 : -1)) : (kws == 0 ? -1 : (kws == 1 ? 1 : 2));
#pragma unroll
        for (int e = 0; e < VE; e++) {
          float a = 0.f;
#pragma unroll
          for (int hh = 0; hh < 3; hh++)
#pragma unroll
            for (int ww = 0; ww < 3; ww++)
              if (hh >= h0 && hh <= h1 && ww >= w0 && ww <= w1) a += src[e][hh * 3 + ww];
          v[e] = a;
        }
      }
      const int k0 = tap * C + c0;
      stv(out + (int64_t)row * Kpad + k0, v);
      if (frag) stv(out + total + pack_frag_index<T>(row, k0, C, rows_pad, tap, c0), v);
    }
  }
  // padding (rows beyond `rows`, k beyond K) is zero: 16-byte units of the row-major copy, and of the fragment-ordered copy for padded rows
  const float z[VE] = {};
  const int KV = Kpad / VE, KVr = K / VE;
  const int64_t pad_units = (int64_t)(rows_pad - rows) * KV + (int64_t)rows * (KV - KVr);
  for (int64_t u = (int64_t)bx * 256 + threadIdx.x; u < pad_units; u += (int64_t)nbx * 256) {
    int row, kv;
    if (u < (int64_t)rows * (KV - KVr)) { row = (int)(u / (KV - KVr)); kv = KVr + (int)(u - (int64_t)row * (KV - KVr)); }
    else { const int64_t r = u - (int64_t)rows * (KV - KVr); row = rows + (int)(r / KV); kv = (int)(r - (int64_t)(row - rows) * KV); }
    stv(out + (int64_t)row * Kpad + kv * VE, z);
    if (frag && kv < KVr) { const int k0 = kv * VE, tap = k0 / C; stv(out + total + pack_frag_index<T>(row, k0, C, rows_pad, tap, k0 - tap * C), z); }
    if (tokfrag && kv < KVr) stv(out + total + pack_tokfrag_index<T>(row, kv * VE, K), z);
  }
}
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackItem* __restrict__ items) {
  const PackItem it = items[blockIdx.y];
  const bool vec = rd_pack_vec_enabled(it);
  if (it.dtype == 0) { if (vec && pack_vec_ok<float>(it)) pack_one_vec<float>(it, blockIdx.x, gridDim.x); else pack_one<float>(it, blockIdx.x, gridDim.x); }
  else { if (vec && pack_vec_ok<bf16_t>(it)) pack_one_vec<bf16_t>(it, blockIdx.x, gridDim.x); else pack_one<bf16_t>(it, blockIdx.x, gridDim.x); }
}

// ---- weight gradient ----------------------------------------------------------------------------------
// dW[co, k] = sum_m dY[m, co] * Xg[m, k]   (Xg = the same gathered/virtual input as the forward).
// Block = (k-tile of 128, cout-tile of 128, pixel split).  Reduction runs over pixels in stages of 32;
// both operands are converted to fp32 in LDS and fed to the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32),
// for which each lane supplies one scalar -> the pixel-major tiles need no transpose.
__device__ __forceinline__ int wg_idx(int p, int col) { return p * 128 + (col ^ ((p & 1) << 4)); }
// dY tile [32 pixels][COT]: the XOR keeps the two pixel rows a ds_read_b32 half-wave touches on different banks
// (COT = 16 rows are 16 words apart, which already does that)
template <int COT>
__device__ __forceinline__ int wgy_idx(int p, int col) { return p * COT + (COT >= 32 ? (col ^ ((p & 1) << 4)) : col); }

// XCD-aware block order of the split-K weight-gradient kernels.  Consecutive workgroup ids go round-robin to the 8 XCDs, each with
// its own L2; with (k-tile, cout-tile, split) = blockIdx.(x, y, z) every XCD touched every pixel of x and dY (PMC: 225 MB fetched for
// 28 MB of operands on the 384->256 layer).  Here the grid is 1-D and XCD x owns the pixel splits s = x, x+8, ...: all (k-tile,
// cout-tile) blocks of a split -- nine taps over the same pixels, every cout tile over the same x -- hit one L2.
struct WgBlock { int kt, ct, split; };
__device__ __forceinline__ bool wg_block(int K, int Cout, int cot, int nsplit, WgBlock& b) {
  const int nkt = (K + 127) / 128, nct = (Cout + cot - 1) / cot, per = nkt * nct;
  const int xcd = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
  b.split = xcd + 8 * (j / per);
  const int r = j % per;
  b.kt = r % nkt; b.ct = r / nkt;
  return b.split < nsplit;
}
static unsigned wg_grid(int K, int Cout, int cot, int nsplit) { return (unsigned)(8 * cdiv(K, 128) * cdiv(Cout, cot) * cdiv(nsplit, 8)); }

template <typename T, bool VEC, int COT>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int PK = 32;  // pixels per stage
  constexpr int WROWS = COT == 128 ? 2 : 1, WCOLS = 4 / WROWS;
  constexpr int TI = COT / (16 * WROWS), TJ = 128 / (16 * WCOLS);
  constexpr int YV = PK * COT / 4;              // float4 vectors in the dY tile
  constexpr int YIT = (YV + 255) / 256;
  __shared__ float sX[2][PK * 128];
  __shared__ float sY[2][PK * COT];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  WgBlock wb;
  if (!wg_block(a.K, a.Cout, COT, a.nsplit, wb)) return;
  const int kt0 = wb.kt * 128, c0 = wb.ct * COT;
  const int Cin = a.C1 + a.C2;
  const int mbeg = wb.split * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);

  // X staging role: column group cg (4 consecutive k columns, fixed for the whole pixel loop), pixel rows pr + 8*i
  const int cg = t & 31, pr = t >> 5;
  int xkh[4], xkw[4], xci[4]; bool xv[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int k = kt0 + cg * 4 + e;
    xv[e] = k < a.K;
    int kk = xv[e] ? k : 0;
    int tap = kk / Cin; xci[e] = kk - tap * Cin; xkh[e] = tap / a.KW; xkw[e] = tap - xkh[e] * a.KW;
  }
  ConvArgs g;  // reuse the forward gather
  g.src1 = a.src1; g.src2 = a.src2; g.Hin = a.Hin; g.Win = a.Win; g.C1 = a.C1; g.C2 = a.C2; g.H1 = a.H1; g.W1 = a.W1;
  g.dil = 1; g.ups = a.ups; g.scale_h = a.scale_h; g.scale_w = a.scale_w;

  float rx[4][4], ry[YIT][4];
  auto load_stage = [&](int mb) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int m = mb + pr + 8 * i;
      // every load is unconditional (an element that does not exist reads the tensor's first element and is zeroed afterwards): guarded
      // loads with the conversion inside the guard cost one memory round trip each, four to eight per stage
      const bool mv = m < mend;
      const int mm_ = mv ? m : mbeg;
      int ow = mm_ % a.OW; int q = mm_ / a.OW; int oh = q % a.OH; int n = q / a.OH;
      int ihb = oh * a.stride - a.pad, iwb = ow * a.stride - a.pad;
      if (VEC) {
        const T* p;
        const bool ok = conv_src_ptr<T>(g, n, ihb + xkh[0], iwb + xkw[0], xci[0], p) && mv && xv[0];
        ld4z(ok ? p : (const T*)a.src1, ok, rx[i]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const T* p;
          const bool ok = conv_src_ptr<T>(g, n, ihb + xkh[e], iwb + xkw[e], xci[e], p) && mv && xv[e];
          const float v = Elem<T>::ld(ok ? p : (const T*)a.src1);
          rx[i][e] = ok ? v : 0.f;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      int idx = t + 256 * i;
      int py = idx / (COT / 4), cy = (idx % (COT / 4)) * 4;
      int m = mb + py, co = c0 + cy;
      const bool yv = idx < YV && m < mend;
      const T* yp = (const T*)a.dy + (yv ? (int64_t)m * a.Cout + co : 0);
      if ((a.Cout & 3) == 0) {
        const bool ok = yv && co + 3 < a.Cout;
        ld4z(ok ? yp : (const T*)a.dy, ok, ry[i]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool ok = yv && co + e < a.Cout;
          const float v = Elem<T>::ld(ok ? yp + e : (const T*)a.dy);
          ry[i][e] = ok ? v : 0.f;
        }
      }
    }
  };
  auto store_stage = [&](int buf) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      int p = pr + 8 * i;
      *reinterpret_cast<float4*>(&sX[buf][wg_idx(p, cg * 4)]) = make_float4(rx[i][0], rx[i][1], rx[i][2], rx[i][3]);
    }
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      int idx = t + 256 * i;
      if (idx < YV) {
        int py = idx / (COT / 4), cy = (idx % (COT / 4)) * 4;
        *reinterpret_cast<float4*>(&sY[buf][wgy_idx<COT>(py, cy)]) = make_float4(ry[i][0], ry[i][1], ry[i][2], ry[i][3]);
      }
    }
  };

  // wave tile: cout rows wr .. wr+16*TI, k cols wc .. wc+16*TJ
  const int wr = (wv / WCOLS) * TI * 16, wc = (wv % WCOLS) * TJ * 16;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0, 0, 0, 0};
  const int fr = lane & 15, fg = lane >> 4;
  bool jv[TJ];
#pragma unroll
  for (int j = 0; j < TJ; j++) jv[j] = kt0 + wc + j * 16 < a.K;  // wave-uniform: skip all-padding column tiles

  // wave-uniform count of column tiles that are not all padding (a prefix): selects a fully unrolled MFMA body without per-MFMA branches
  int nj = 0;
#pragma unroll
  for (int j = 0; j < TJ; j++) nj += jv[j] ? 1 : 0;
  auto mma_stage = [&](int buf, auto njc) RD_INLINE_LAMBDA {
    constexpr int NJ = decltype(njc)::value;
#pragma unroll
    for (int p4 = 0; p4 < PK / 4; p4++) {
      int p = p4 * 4 + fg;
      float ya[TI], xb[NJ];
#pragma unroll
      for (int i = 0; i < TI; i++) ya[i] = sY[buf][wgy_idx<COT>(p, wr + i * 16 + fr)];
#pragma unroll
      for (int j = 0; j < NJ; j++) xb[j] = sX[buf][wg_idx(p, wc + j * 16 + fr)];
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int i = 0; i < TI; i++) acc[i][j] = mfma_16x16x4_f32(ya[i], xb[j], acc[i][j]);
    }
  };

  int nst = (mend > mbeg) ? (mend - mbeg + PK - 1) / PK : 0;
  if (nst > 0) {
    load_stage(mbeg);
    store_stage(0);
  }
  __syncthreads();
  for (int st = 0; st < nst; st++) {
    int buf = st & 1;
    if (st + 1 < nst) load_stage(mbeg + (st + 1) * PK);
    if (nj == TJ) mma_stage(buf, std::integral_constant<int, TJ>{});
    else if (TJ > 1 && nj == TJ - 1) mma_stage(buf, std::integral_constant<int, (TJ > 1 ? TJ - 1 : 1)>{});
    else if (TJ > 2 && nj == TJ - 2) mma_stage(buf, std::integral_constant<int, (TJ > 2 ? TJ - 2 : 1)>{});
    else if (TJ > 3 && nj == TJ - 3) mma_stage(buf, std::integral_constant<int, (TJ > 3 ? TJ - 3 : 1)>{});
    if (st + 1 < nst) store_stage(buf ^ 1);
    __syncthreads();
  }
  // slab[split][co][k]: lane holds rows (cout) fg*4+r, col (k) fr
  float* slab = a.slab + (int64_t)wb.split * a.Cout * a.K;
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) {
      int k = kt0 + wc + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int co = c0 + wr + i * 16 + fg * 4 + r;
        if (co < a.Cout && k < a.K) slab[(int64_t)co * a.K + k] = acc[i][j][r];
      }
    }
}


// ---- bf16 weight gradient on the bf16 MFMA (16x16x32): both operands need 8 CONSECUTIVE PIXELS per lane, i.e. the pixel-major
// NHWC tiles transposed.  The transpose happens on the way into LDS: a thread loads the same 8 channels of two neighbouring pixels
// (two 16-byte loads), packs the pairs and writes eight 32-bit words into channel-major rows [col][64 pixels]; fragments are then
// single ds_read_b128 (16-byte slots XOR-swizzled as in the forward kernel).  64 pixels per stage = two MFMA k-steps. ------------
template <int COT>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(WgradArgs a) {
  typedef bf16_t T;
  constexpr int PK = 64;
  constexpr int WROWS = COT == 128 ? 2 : 1, WCOLS = 4 / WROWS;
  constexpr int TI = COT / (16 * WROWS), TJ = 128 / (16 * WCOLS);
  constexpr int YG = COT / 8;                          // 8-channel groups of the dY tile
  constexpr int YIT = (YG * 32 + 255) / 256;           // (group, pixel pair) items per thread
  __shared__ uint4 sX[2][128 * 8];                     // [k col][64 px] bf16 = 8 slots of 16 B per row
  __shared__ uint4 sY[2][COT * 8];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  WgBlock wb;
  if (!wg_block(a.K, a.Cout, COT, a.nsplit, wb)) return;
  const int kt0 = wb.kt * 128, c0 = wb.ct * COT;
  const int Cin = a.C1 + a.C2;
  const int mbeg = wb.split * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);

  // X role: pixel pair pp = t % 16 (+16 for the second item), 8-column group cg = t / 16 (fixed k columns for the whole loop)
  const int xpp = t & 15, xcg = t >> 4;
  const int kx = kt0 + xcg * 8;
  const bool xv = kx < a.K;
  int xkh, xkw, xci;
  { int kk = xv ? kx : 0; int tap = kk / Cin; xci = kk - tap * Cin; xkh = tap / a.KW; xkw = tap - xkh * a.KW; }
  // This thread's filter tap and channels are fixed, so the source tensor is picked once; its four pixels advance by one stage (PK
  // pixels) per call, and (image, row, column) follow by carries.  Decoding every pixel with two runtime divisions and the general
  // gather cost ~110 VALU instructions per 16-byte load -- 5x the MFMA cycles of a stage (the kernel was VALU-bound at 218 TFLOP/s).
  const T* xbase; int xC;
  if (xci < a.C1) { xbase = (const T*)a.src1 + xci; xC = a.C1; } else { xbase = (const T*)a.src2 + (xci - a.C1); xC = a.C2; }
  const int Hp = a.ups ? a.H1 : a.Hin, Wp = a.ups ? a.W1 : a.Win;
  struct PX { int n, oh, ow; };
  const PX pstep = {(PK / a.OW) / a.OH, (PK / a.OW) % a.OH, PK % a.OW};
  PX pxs[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int m = mbeg + (xpp + 16 * i) * 2 + h;
      pxs[i][h].ow = m % a.OW; const int q = m / a.OW; pxs[i][h].oh = q % a.OH; pxs[i][h].n = q / a.OH;
    }

  uint4 rx[2][2], ry[YIT][2];
  auto load_px = [&](int m, PX& c, uint4& v) RD_INLINE_LAMBDA {   // called once per stage and pixel stream, in stage order
    v = make_uint4(0, 0, 0, 0);
    const int ih = c.oh * a.stride - a.pad + xkh, iw = c.ow * a.stride - a.pad + xkw;
    if (xv && m < mend && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win) {
      int hs = ih, ws = iw;
      if (a.ups) {  // F.interpolate(mode='nearest') source index, ATen float formula (as conv_src_ptr)
        hs = min((int)floorf((float)ih * a.scale_h), a.H1 - 1);
        ws = min((int)floorf((float)iw * a.scale_w), a.W1 - 1);
      }
      v = *reinterpret_cast<const uint4*>(xbase + (((int64_t)c.n * Hp + hs) * Wp + ws) * xC);
    }
    c.ow += pstep.ow; if (c.ow >= a.OW) { c.ow -= a.OW; c.oh++; }
    c.oh += pstep.oh; if (c.oh >= a.OH) { c.oh -= a.OH; c.n++; }
    if (c.oh >= a.OH) { c.oh -= a.OH; c.n++; }
    c.n += pstep.n;
  };
  auto load_stage = [&](int mb) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      int m = mb + (xpp + 16 * i) * 2;
      load_px(m, pxs[i][0], rx[i][0]);
      load_px(m + 1, pxs[i][1], rx[i][1]);
    }
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      int idx = t + 256 * i;
      int pp = idx & 31, yg = idx >> 5;
      int co = c0 + yg * 8;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        int m = mb + pp * 2 + h;
        ry[i][h] = make_uint4(0, 0, 0, 0);
        if (yg < YG && m < mend && co < a.Cout) ry[i][h] = *reinterpret_cast<const uint4*>((const T*)a.dy + (int64_t)m * a.Cout + co);
      }
    }
  };
  // word w (two bf16: pixel 2pp in the low half, 2pp+1 in the high half) of channel e of a pixel-pair.  The eight LDS word offsets of
  // a (row group, pixel pair) are the same in every stage: computed once; the halves are interleaved by one v_perm_b32 per word (the
  // shift / mask / or form plus the per-word swizzle arithmetic were ~250 of the ~400 VALU instructions per wave and stage).
  int xw[2][8], yw[YIT][8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
#pragma unroll
    for (int i = 0; i < 2; i++) { const int pp = xpp + 16 * i; xw[i][e] = lds_slot(xcg * 8 + e, pp >> 2) * 4 + (pp & 3); }
#pragma unroll
    for (int i = 0; i < YIT; i++) { const int idx = t + 256 * i, pp = idx & 31, yg = idx >> 5; yw[i][e] = lds_slot(yg * 8 + e, pp >> 2) * 4 + (pp & 3); }
  }
  auto put = [&](uint4* tile, const int (&wo)[8], const uint4& lo, const uint4& hi) RD_INLINE_LAMBDA {
    const unsigned l[4] = {lo.x, lo.y, lo.z, lo.w}, h[4] = {hi.x, hi.y, hi.z, hi.w};
    unsigned* words = reinterpret_cast<unsigned*>(tile);
#pragma unroll
    for (int e = 0; e < 8; e++) words[wo[e]] = (e & 1) ? pack_hi16(l[e >> 1], h[e >> 1]) : pack_lo16(l[e >> 1], h[e >> 1]);
  };
  auto store_stage = [&](int buf) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 2; i++) put(sX[buf], xw[i], rx[i][0], rx[i][1]);
#pragma unroll
    for (int i = 0; i < YIT; i++) {
      int idx = t + 256 * i;
      if ((idx >> 5) < YG) put(sY[buf], yw[i], ry[i][0], ry[i][1]);
    }
  };

  const int wr = (wv / WCOLS) * TI * 16, wc = (wv % WCOLS) * TJ * 16;
  f32x4 acc[TI][TJ];
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) acc[i][j] = f32x4{0, 0, 0, 0};
  const int fr = lane & 15, fg = lane >> 4;

  int nst = (mend > mbeg) ? (mend - mbeg + PK - 1) / PK : 0;
  if (nst > 0) { load_stage(mbeg); store_stage(0); }
  __syncthreads();
  for (int st = 0; st < nst; st++) {
    int buf = st & 1;
    if (st + 1 < nst) load_stage(mbeg + (st + 1) * PK);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {  // 32 pixels per MFMA k-step: lane group fg supplies pixels ks*32 + fg*8 .. +7 (one 16-byte slot)
      s16x8 ya[TI], xb[TJ];
#pragma unroll
      for (int i = 0; i < TI; i++) { uint4 v = sY[buf][lds_slot(wr + i * 16 + fr, ks * 4 + fg)]; __builtin_memcpy(&ya[i], &v, 16); }
#pragma unroll
      for (int j = 0; j < TJ; j++) { uint4 v = sX[buf][lds_slot(wc + j * 16 + fr, ks * 4 + fg)]; __builtin_memcpy(&xb[j], &v, 16); }
#pragma unroll
      for (int i = 0; i < TI; i++)
#pragma unroll
        for (int j = 0; j < TJ; j++) acc[i][j] = mfma_16x16x32_bf16(ya[i], xb[j], acc[i][j]);
    }
    if (st + 1 < nst) store_stage(buf ^ 1);
    __syncthreads();
  }
  float* slab = a.slab + (int64_t)wb.split * a.Cout * a.K;
#pragma unroll
  for (int i = 0; i < TI; i++)
#pragma unroll
    for (int j = 0; j < TJ; j++) {
      int k = kt0 + wc + j * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int co = c0 + wr + i * 16 + fg * 4 + r;
        if (co < a.Cout && k < a.K) slab[(int64_t)co * a.K + k] = acc[i][j][r];
      }
    }
}

// ---- halo-tile weight gradient for 3x3 / stride-1 layers with few channels (decoder tail: Cin <= 64, Cout <= 32) -----------
// The generic wgrad re-gathers every input pixel once per tap (9x through L1/L2), which bounds these layers far below both the
// MFMA and the HBM roofline.  Here a block stages an (TH+2) x (TW+2) x Cin input patch (upsample / concat folded in) and the
// TH x TW x Cout output-gradient tile in LDS ONCE, converts to fp32, and runs all 9 taps from LDS on v_mfma_f32_16x16x4_f32:
// D[cout][(tap, cin)] += dY[pixel][cout] * X[pixel + tap][cin], k = 4 consecutive pixels of a row.  Blocks are persistent over
// tiles and write one slab each, summed in order by wgrad_reduce_kernel (deterministic).
__device__ __forceinline__ int hswz(int C, int c, int parity) { return C >= 32 ? (c ^ (parity << 4)) : c; }

// tile geometry (compile time): the (TH, TW) with TW in {16, 32} that fits 60 KiB and has the best useful / staged pixel ratio
constexpr int halo_th(int cti, int rt) { return (cti == 4 && rt == 2) ? 4 : 8; }
constexpr int halo_tw(int cti, int rt) { return (cti == 4 || (cti == 2 && rt == 2)) ? 16 : 32; }

template <typename T, int CTI, int RT>  // CTI = Cin/16, RT = ceil(Cout/16)
__global__ __launch_bounds__(256) void conv_wgrad_halo_kernel(WgradArgs a, int tilesH, int tilesW) {
  constexpr int CIN = CTI * 16, COP = RT * 16;
  constexpr int TH = halo_th(CTI, RT), TW = halo_tw(CTI, RT), HT = TH + 2, WT = TW + 2;
  constexpr int NCW = (9 * CTI + 3) / 4;  // (tap, cin-tile) column tiles per wave
  constexpr int VX = CIN / 4, PLX = 256 / VX, NX = (HT * WT + PLX - 1) / PLX;  // staged input vectors per thread
  constexpr int VY = COP / 4, PLY = 256 / VY, NY = (TH * TW + PLY - 1) / PLY;
  static_assert(HT * WT * CIN + TH * TW * COP <= 15360, "halo tile exceeds 60 KiB of LDS");
  __shared__ float smem[HT * WT * CIN + TH * TW * COP];
  float* sX = smem;                       // [HT][WT][CIN]   (channel index XOR-swizzled by column parity)
  float* sY = smem + HT * WT * CIN;       // [TH*TW][COP]
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  ConvArgs g;
  g.src1 = a.src1; g.src2 = a.src2; g.Hin = a.Hin; g.Win = a.Win; g.C1 = a.C1; g.C2 = a.C2; g.H1 = a.H1; g.W1 = a.W1;
  g.dil = 1; g.ups = a.ups; g.scale_h = a.scale_h; g.scale_w = a.scale_w;

  f32x4 acc[RT][NCW];
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < NCW; j++) acc[i][j] = f32x4{0, 0, 0, 0};
  int jtap[NCW], jct[NCW]; bool jv[NCW];
#pragma unroll
  for (int j = 0; j < NCW; j++) {
    int idx = wv + 4 * j;
    jv[j] = idx < 9 * CTI;
    int id2 = jv[j] ? idx : 0;
    jtap[j] = id2 / CTI; jct[j] = id2 - jtap[j] * CTI;
  }
  // per-lane LDS offsets: px0 and 4*qd are even, so the swizzle parity of a lane's pixel is (fg + kw) & 1 and every LDS address of the
  // MFMA loop is  row base + per-lane constant + compile-time offset (the first version spent 41 VALU instructions per MFMA here)
  int offx[NCW], offy[RT];
#pragma unroll
  for (int j = 0; j < NCW; j++) {
    int kh = jtap[j] / 3, kw = jtap[j] - kh * 3;
    offx[j] = (kh * WT + fg + kw) * CIN + hswz(CIN, jct[j] * 16 + fr, (fg + kw) & 1);
  }
#pragma unroll
  for (int i = 0; i < RT; i++) offy[i] = fg * COP + hswz(COP, i * 16 + fr, fg & 1);

  // staging roles: thread = (fixed 4-channel vector, pixel lane); the next tile is fetched into registers while the current one is
  // being multiplied, so global-memory latency is off the critical path (2 blocks per CU would otherwise idle through it)
  const int xc = (t % VX) * 4, xp0 = t / VX;
  const int yc = (t % VY) * 4, yp0 = t / VY;
  float fx[NX][4], fy[NY][4];
  const int ntiles = a.N * tilesH * tilesW;
  auto fetch = [&](int tile) RD_INLINE_LAMBDA {
    int tw_ = tile % tilesW; int q = tile / tilesW; int th_ = q % tilesH; int n = q / tilesH;
    const int oh0 = th_ * TH, ow0 = tw_ * TW;
#pragma unroll
    for (int it = 0; it < NX; it++) {
      int pq = xp0 + it * PLX;
      int py = pq / WT, px = pq - py * WT;
      const T* p;
      const bool ok = conv_src_ptr<T>(g, n, oh0 - 1 + py, ow0 - 1 + px, xc, p) && pq < HT * WT;
      ld4z(ok ? p : (const T*)a.src1, ok, fx[it]);      // unconditional, masked: the tile's requests issue together
    }
#pragma unroll
    for (int it = 0; it < NY; it++) {
      int pq = yp0 + it * PLY;
      int py = pq / TW, px = pq - py * TW;
      int oh = oh0 + py, ow = ow0 + px;
      const bool yv = pq < TH * TW && oh < a.OH && ow < a.OW;
      const T* yp = (const T*)a.dy + (yv ? (((int64_t)n * a.OH + oh) * a.OW + ow) * a.Cout + yc : 0);
      if ((a.Cout & 3) == 0) {
        const bool ok = yv && yc + 3 < a.Cout;
        ld4z(ok ? yp : (const T*)a.dy, ok, fy[it]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool ok = yv && yc + e < a.Cout;
          const float v = Elem<T>::ld(ok ? yp + e : (const T*)a.dy);
          fy[it][e] = ok ? v : 0.f;
        }
      }
    }
  };
  auto stash = [&]() RD_INLINE_LAMBDA {
#pragma unroll
    for (int it = 0; it < NX; it++) {
      int pq = xp0 + it * PLX;
      int px = pq % WT;
      if (pq < HT * WT)
        *reinterpret_cast<float4*>(&sX[pq * CIN + hswz(CIN, xc, px & 1)]) = make_float4(fx[it][0], fx[it][1], fx[it][2], fx[it][3]);
    }
#pragma unroll
    for (int it = 0; it < NY; it++) {
      int pq = yp0 + it * PLY;
      int px = pq % TW;
      if (pq < TH * TW)
        *reinterpret_cast<float4*>(&sY[pq * COP + hswz(COP, yc, px & 1)]) = make_float4(fy[it][0], fy[it][1], fy[it][2], fy[it][3]);
    }
  };

  auto compute = [&](auto njc) RD_INLINE_LAMBDA {
    constexpr int NJ = decltype(njc)::value;
    constexpr int QD = CTI == 4 ? 2 : 4;  // pixel quads whose LDS reads are batched ahead of their MFMAs (register budget)
#pragma unroll
    for (int py = 0; py < TH; py++)
#pragma unroll
      for (int px0 = 0; px0 < TW; px0 += 4 * QD) {
        const float* xrow = sX + (py * WT + px0) * CIN;
        const float* yrow = sY + (py * TW + px0) * COP;
        float ya[QD][RT], xb[QD][NJ > 0 ? NJ : 1];
#pragma unroll
        for (int qd = 0; qd < QD; qd++) {
#pragma unroll
          for (int i = 0; i < RT; i++) ya[qd][i] = yrow[offy[i] + qd * 4 * COP];
#pragma unroll
          for (int j = 0; j < NJ; j++) xb[qd][j] = xrow[offx[j] + qd * 4 * CIN];
        }
#pragma unroll
        for (int qd = 0; qd < QD; qd++)
#pragma unroll
          for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int i = 0; i < RT; i++) acc[i][j] = mfma_16x16x4_f32(ya[qd][i], xb[qd][j], acc[i][j]);
      }
  };

  int tile = blockIdx.x;
  if (tile < ntiles) fetch(tile);
  for (; tile < ntiles; tile += gridDim.x) {
    __syncthreads();  // previous tile fully consumed
    stash();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) fetch(tile + gridDim.x);
    // the number of column tiles a wave owns (NCW or NCW-1) selects one of two fully unrolled bodies: a per-MFMA `if` made the
    // compiler shuttle the accumulators between register files around every MFMA (33 VALU instructions per MFMA, measured)
    if (jv[NCW - 1]) compute(std::integral_constant<int, NCW>{});
    else compute(std::integral_constant<int, NCW - 1>{});
  }
  float* slab = a.slab + (int64_t)blockIdx.x * a.Cout * a.K;
#pragma unroll
  for (int i = 0; i < RT; i++)
#pragma unroll
    for (int j = 0; j < NCW; j++)
      if (jv[j]) {
        int k = jtap[j] * CIN + jct[j] * 16 + fr;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          int co = i * 16 + fg * 4 + r;
          if (co < a.Cout) slab[(int64_t)co * a.K + k] = acc[i][j][r];
        }
      }
}

static bool wgrad_halo_ok(const WgradArgs& a, int dtype) {
  const int Cin = a.C1 + a.C2;
  if (dtype == 1 && Cin == 64 && a.Cout % 8 == 0) return false;  // measured: the bf16-MFMA generic kernel is ~1.8x faster there
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && (Cin == 16 || Cin == 32 || Cin == 64) && a.Cout <= 32 &&
         (a.C1 % 8 == 0) && a.OH == a.Hin && a.OW == a.Win;
}
static void halo_geom(int Cin, int Cout, int& TH, int& TW) {
  const int cti = Cin / 16, rt = Cout <= 16 ? 1 : 2;
  TH = halo_th(cti, rt); TW = halo_tw(cti, rt);
}
static const int HALO_BLOCKS = 1024;

// sum slabs and write / accumulate the OIHW fp32 gradient.  thread = (output o, split lane sl): lane sl adds splits sl, sl+SL, ...
// in order, the SL lane sums are then added in lane order -> a fixed summation tree (deterministic) with SL-fold memory parallelism
// (the halo-tile path produces up to 1024 slabs for a few thousand outputs).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin, int KH,
                                                           int KW, int nsplit, int accumulate, int SL) {
  __shared__ float red[256];
  const int K = KH * KW * Cin;
  const int64_t total = (int64_t)Cout * K;
  const int OB = 256 / SL;
  const int t = threadIdx.x, ol = t % OB, sl = t / OB;
  for (int64_t base = (int64_t)blockIdx.x * OB; base < total; base += (int64_t)gridDim.x * OB) {
    const int64_t i = base + ol;
    float s = 0.f;
    if (i < total)
      for (int sp = sl; sp < nsplit; sp += SL) s += slab[(int64_t)sp * total + i];
    red[t] = s;
    __syncthreads();
    if (sl == 0 && i < total) {
      float acc = 0.f;
      for (int q = 0; q < SL; q++) acc += red[q * OB + ol];
      int co = (int)(i / K), k = (int)(i - (int64_t)co * K);
      int tap = k / Cin, ci = k - tap * Cin, kh = tap / KW, kw = tap - kh * KW;
      int64_t o = (((int64_t)co * Cin + ci) * KH + kh) * KW + kw;
      dw[o] = accumulate ? dw[o] + acc : acc;
    }
    __syncthreads();
  }
}

// float4 version (Cout*K a multiple of 4): thread = (float4 column ol, slab lane sl); each slab lane sums every SL-th slab with
// 16-byte loads (a quarter of the load instructions of the scalar kernel, whose per-thread loop was the whole latency of this launch),
// then a fixed-order LDS tree over the slab lanes.  Deterministic.
__device__ __forceinline__ void wgrad_reduce_vec_body(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin, int KH, int KW,
                                                      int nsplit, int accumulate, int SL, int bx, int nb, float4* red) {
  const int K = KH * KW * Cin;
  const int64_t total = (int64_t)Cout * K, total4 = total >> 2;
  const int OB = 256 / SL;
  const int t = threadIdx.x, ol = t % OB, sl = t / OB;
  const float4* s4 = reinterpret_cast<const float4*>(slab);
  for (int64_t base = (int64_t)bx * OB; base < total4; base += (int64_t)nb * OB) {
    const int64_t i4 = base + ol;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < total4) {
      int sp = sl;
      for (; sp + 3 * SL < nsplit; sp += 4 * SL) {   // four independent loads in flight
        const float4 v0 = s4[(int64_t)sp * total4 + i4], v1 = s4[(int64_t)(sp + SL) * total4 + i4];
        const float4 v2 = s4[(int64_t)(sp + 2 * SL) * total4 + i4], v3 = s4[(int64_t)(sp + 3 * SL) * total4 + i4];
        s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
        s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      for (; sp < nsplit; sp += SL) { const float4 v = s4[(int64_t)sp * total4 + i4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    red[t] = s;
    __syncthreads();
    for (int h = SL >> 1; h > 0; h >>= 1) {
      if (sl < h) { const float4 o = red[t + h * OB]; float4 m = red[t]; m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w; red[t] = m; }
      __syncthreads();
    }
    if (sl == 0 && i4 < total4) {
      const float4 m = red[t];
      const float acc[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int64_t i = i4 * 4 + e;
        int co = (int)(i / K), k = (int)(i - (int64_t)co * K);
        int tap = k / Cin, ci = k - tap * Cin, kh = tap / KW, kw = tap - kh * KW;
        int64_t o = (((int64_t)co * Cin + ci) * KH + kh) * KW + kw;
        dw[o] = accumulate ? dw[o] + acc[e] : acc[e];
      }
    }
    __syncthreads();
  }
}
__global__ __launch_bounds__(256) void wgrad_reduce_vec_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin,
                                                                int KH, int KW, int nsplit, int accumulate, int SL) {
  __shared__ float4 red[256];
  wgrad_reduce_vec_body(slab, dw, Cout, Cin, KH, KW, nsplit, accumulate, SL, blockIdx.x, gridDim.x, red);
}
// The slab reductions of MANY weight gradients in one launch (items by value in the kernel arguments; a block finds its item from the
// running block counts): 30+ launches of ~9 us, each too small to fill the chip, become one per backward stage.
__global__ __launch_bounds__(256) void wgrad_reduce_batch_kernel(WgradReduceBatch b) {
  __shared__ float4 red[256];
  int it = 0;
  while (it + 1 < b.n && (int)blockIdx.x >= b.first[it + 1]) it++;
  const WgradReduceItem& q = b.item[it];
  wgrad_reduce_vec_body(q.slab, q.dw, q.Cout, q.Cin, q.KH, q.KW, q.nsplit, q.accumulate, b.SL[it], (int)blockIdx.x - b.first[it],
                        b.first[it + 1] - b.first[it], red);
}
static int wgrad_reduce_sl(int nsplit) { int SL = 1; while (SL * 8 < nsplit && SL < 64) SL <<= 1; return SL; }   // about 8 slabs per slab lane

static void launch_wgrad_reduce(const float* slab, float* dw, int Cout, int Cin, int KH, int KW, int nsplit, int accumulate, hipStream_t st) {
  int64_t total = (int64_t)Cout * KH * KW * Cin;
  if (total % 4 == 0) {
    int SL = wgrad_reduce_sl(nsplit);
    unsigned rg = (unsigned)std::min<int64_t>(cdiv(total / 4, 256 / SL), 4096);
    hipLaunchKernelGGL(wgrad_reduce_vec_kernel, dim3(rg), dim3(256), 0, st, slab, dw, Cout, Cin, KH, KW, nsplit, accumulate, SL);
    return;
  }
  int SL = 1; while (SL < nsplit && SL < 16) SL <<= 1;
  unsigned rg = (unsigned)std::min<int64_t>(cdiv(total, 256 / SL), 4096);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rg), dim3(256), 0, st, slab, dw, Cout, Cin, KH, KW, nsplit, accumulate, SL);
}

// ---- host-side launchers ------------------------------------------------------------------------------------
static int pick_bn(int cout) {
  if (cout <= 16) return 16;
  if (cout <= 32) return 32;
  if (cout <= 64) return 64;
  return 128;
}
int conv_rows_pad(int rows) { int bn = pick_bn(rows); return (int)cdiv(rows, bn) * bn; }
int conv_kpad(int K, int dtype) { int bke = dtype == 0 ? 32 : 64; return (int)cdiv(K, bke) * bke; }
// elements of a packed operand: K axes that can be 9 taps x whole 128-byte chunks leave room for the fragment-ordered copy
int64_t conv_packed_elems(int rows, int K, int dtype) {
  const int64_t n = (int64_t)conv_rows_pad(rows) * conv_kpad(K, dtype);
  return (K % (9 * (dtype == 0 ? 32 : 64)) == 0 || pack_has_tokfrag(rows, K)) ? 2 * n : n;
}

// tile choice: small-M launches (fewer than ~2 blocks per CU with 128-pixel tiles) halve the pixel tile, then the channel tile
static void conv_tiles(int M, int Cout, int& bn, int& wm) {
  bn = pick_bn(Cout);
  wm = 4;
  // (round 6: outputs of >= 768 channels halve the pixel tile up to 768 blocks -- the SML's 136 -> 816 expansions and the data gradients of its
  // 816 -> 136 projections on 10 368 pixels are 567 blocks of 128 x 128 and ran 22.3 / 25.9 us (plain / with statistics) against 17.5 / 19.6 on
  // 64-pixel tiles; the narrower layers of that stage lose 0.7 us with them: tools/bench_pw.py, profiles/r06_microbench/pw_tiles.txt)
  // (... and the general limits 384 -> 512 blocks, both of them: swept in the captured steps, three alternating rounds on one box -- SML 11.77 -> 11.68 ms,
  // RC-Net unchanged at 7.08; 768 for the second one costs RC-Net 0.01 ms)
  if (bn >= 32 && cdiv(M, BM) * cdiv(Cout, bn) < (Cout >= 768 ? 768 : 512)) {
    wm = 2;
    if (bn == 128 && cdiv(M, 64) * cdiv(Cout, bn) < 512) bn = 64;
    // still fewer blocks than CUs (a few hundred rows: RC-Net's FullyConnected layers, the deep encoder stages on 8 x 16 maps): narrower
    // channel tiles.  Such a launch is a chain of stages whose length is the MFMA work of ONE block per stage (measured: 1.1 us per fp32
    // stage at 64 channels whatever the prefetch depth), so spreading the channels over more CUs shortens every stage.
    if (bn == 64 && cdiv(M, 64) * cdiv(Cout, bn) < 128) bn = 32;
  }
}
int conv_block_pixels(int M, int Cout) { int bn, wm; conv_tiles(M, Cout, bn, wm); return 32 * wm; }

// parity-class walk of a stride-2 layer's data gradient (conv_gemm_kernel PAR): whole 128-byte channel stages, one source, no statistics
static bool conv_gemm_par(const ConvArgs& a, int dtype) {
  const int bke = STAGE_BYTES / (dtype == 0 ? 4 : 2), Cin = a.C1 + a.C2;
  const bool off = rd_opt(OPT_CONV_PAR, 1) == 0;      // A/B switch (rd_set_option)
  return !off && a.dil == 2 && a.stride == 1 && !a.ups && a.C2 == 0 && Cin % bke == 0 && !a.stats && !a.pool2 && a.K == a.KH * a.KW * Cin &&
         a.Kpad == a.K;
}
static bool conv_gemm_deep(const ConvArgs& a, int dtype) {
  const int ve = dtype == 0 ? 4 : 8, Cin = a.C1 + a.C2, es = dtype == 0 ? 4 : 2;
  const bool vec = (Cin % ve == 0) && (a.C1 % ve == 0);
  int bn, wm;
  conv_tiles(a.M, a.Cout, bn, wm);
  int nk = a.Kpad / (STAGE_BYTES / es);
  if (conv_gemm_par(a, dtype)) nk = ((a.KH + 1) / 2) * ((a.KW + 1) / 2) * (Cin / (STAGE_BYTES / es));      // the longest class
  return vec && wm == 2 && cdiv(a.M, 32 * wm) * cdiv(a.Cout, bn) <= 512 && nk >= 6;
}
template <typename T>
static void launch_conv_t(const ConvArgs& a, hipStream_t st) {
  constexpr int VE = Elem<T>::VE;
  const int Cin = a.C1 + a.C2;
  const bool vec = (Cin % VE == 0) && (a.C1 % VE == 0);
  int bn, wm;
  conv_tiles(a.M, a.Cout, bn, wm);
  dim3 grid((unsigned)cdiv(a.M, 32 * wm), (unsigned)cdiv(a.Cout, bn));
  const bool par = conv_gemm_par(a, sizeof(T) == 4 ? 0 : 1);
  if (par) {      // the tiles of the four parity classes, one after the other
    unsigned gx = 0;
    for (int c = 0; c < 4; c++) { int ph, pw, oh2, ow2, mc; par_class(c, a.OH, a.OW, a.N, ph, pw, oh2, ow2, mc); gx += (unsigned)cdiv(mc, 32 * wm); }
    grid.x = gx;
  }
  // few blocks and a long K axis: latency-bound stage chain -> two-stage-ahead loads
  const bool deep = conv_gemm_deep(a, sizeof(T) == 4 ? 0 : 1);
#define RD_CONV_CASE(BNV, WMV)                                                                                  \
  if (bn == BNV && wm == WMV) {                                                                                  \
    if (par) hipLaunchKernelGGL((conv_gemm_kernel<T, BNV, true, WMV, false, true>), grid, dim3(256), 0, st, a);  \
    else if (vec) hipLaunchKernelGGL((conv_gemm_kernel<T, BNV, true, WMV>), grid, dim3(256), 0, st, a);         \
    else hipLaunchKernelGGL((conv_gemm_kernel<T, BNV, false, WMV>), grid, dim3(256), 0, st, a);                 \
  }
#define RD_CONV_DEEP(BNV) if (bn == BNV) { if (par) hipLaunchKernelGGL((conv_gemm_kernel<T, BNV, true, 2, true, true>), grid, dim3(256), 0, st, a); \
                                            else hipLaunchKernelGGL((conv_gemm_kernel<T, BNV, true, 2, true>), grid, dim3(256), 0, st, a); return; }
  if (deep) { RD_CONV_DEEP(32) RD_CONV_DEEP(64) RD_CONV_DEEP(128) }
#undef RD_CONV_DEEP
  RD_CONV_CASE(16, 4) RD_CONV_CASE(32, 4) RD_CONV_CASE(64, 4) RD_CONV_CASE(128, 4)
  RD_CONV_CASE(32, 2) RD_CONV_CASE(64, 2) RD_CONV_CASE(128, 2)
#undef RD_CONV_CASE
}

// 3x3/stride-1 layers with enough tiles to fill the chip go to the patch-staged kernel (rd_conv3x3.hip)
static int conv3x3_min_blocks() {
  return rd_opt(OPT_CONV3X3_MIN_BLOCKS, 256);  // test hook (rd_set_option): 0 forces the patch kernel on small cases
}
static bool use_conv3x3(const ConvArgs& a, int dtype) {
  if (!conv3x3_ok(a, dtype)) return false;
  return (int64_t)conv3x3_tiles(a) * cdiv(a.Cout, pick_bn(a.Cout)) >= conv3x3_min_blocks();
}
static bool use_conv3x3_small(const ConvArgs& a, int dtype) {
  return conv3x3_small_ok(a, dtype) && conv3x3_tiles(a) >= conv3x3_min_blocks();
}
static bool use_conv3x3_frag(const ConvArgs& a, int dtype) {   // wide layers: weights in fragment order from L2, only the patch in LDS
  // also far below one block per CU: a block's chain (patch load, 9 taps per chunk, store) is short next to the implicit-GEMM kernel's
  // stage chain on the same shape -- deep encoder stages, 9 672 / 2 560 pixels x 128 channels: 21 -> 13 us and 15 -> 12 us per launch
  return conv3x3_frag_ok(a, dtype) && conv3x3_frag_blocks(a, dtype) >= std::min(conv3x3_min_blocks(), 8);
}
static bool conv_stem_ok(const ConvArgs& a, int dtype);
static int conv_stem_blocks(const ConvArgs& a);
// rd_conv_pw.hip: pointwise layers on a few thousand pixels
bool conv_pw_shape(const ConvArgs& a, int dtype);
bool conv_pw_ok(const ConvArgs& a, int dtype);
int conv_pw_rows(const ConvArgs& a);
const char* conv_pw_name(const ConvArgs& a, int dtype);
void launch_conv_pw(const ConvArgs& a, int dtype, hipStream_t st);
static bool conv_skinny_ok(const ConvArgs& a, int dtype);
static bool conv_pw_route(const ConvArgs& a, int dtype) {      // as launch_conv orders its routes
  return !conv_skinny_ok(a, dtype) && !conv_stem_ok(a, dtype) && !conv_few_ok(a) && !conv1x1_direct_ok(a, dtype) && conv_pw_ok(a, dtype);
}
static bool conv_d2s_small(const ConvArgs& a, int dtype);
int conv_stats_rows(const ConvArgs& a, int dtype) {
  if (a.d2s) return conv_d2s_small(a, dtype) ? conv3x3_small_blocks(a, dtype) : conv3x3_frag_tiles(a, dtype);
  if (conv_stem_ok(a, dtype)) return conv_stem_blocks(a);
  if (conv_few_ok(a)) return conv_few_blocks(a);
  if (conv1x1_direct_ok(a, dtype)) return conv1x1_direct_rows(a);
  if (conv_pw_route(a, dtype)) return conv_pw_rows(a);
  if (use_conv3x3_small(a, dtype)) return conv3x3_small_blocks(a, dtype);   // persistent blocks: one statistics row each
  if (use_conv3x3_frag(a, dtype)) return conv3x3_frag_tiles(a, dtype);
  if (use_conv3x3(a, dtype)) return conv3x3_tiles(a);
  return (int)cdiv(a.M, conv_block_pixels(a.M, a.Cout));
}

// ---- skinny linear layers: few output tiles, long K axis ----------------------------------------------------------------------------
// The point MLP's last layer (RCNet/networks.py:299-329: 128 -> 2688 over R = 240 radar points) has a data gradient of 240 x 128 outputs
// with K = 2688: eight 64 x 64 tiles of the implicit-GEMM kernel walk 84 dependent stages each (63 us for 165 MFLOP).  Here a block owns
// ONE 16 x 16 output tile and its eight waves split the K axis (steps of one 16-byte vector per lane, interleaved); both operands are
// k-contiguous (activation rows, packed weight rows) and come straight from L2 into the MFMA registers; the eight partial tiles are summed
// through LDS in wave order (fixed summation order: reproducible), bias / activation in the epilogue.
template <typename T>
__global__ __launch_bounds__(512) void linear_skinny_kernel(ConvArgs a) {
  constexpr int VE = Elem<T>::VE, STEP = 4 * VE, NWV = 8;
  __shared__ float red[NWV][64][4];
  const int t = threadIdx.x, lane = t & 63, wv = RD_WAVE_UNIFORM(t >> 6), fr = lane & 15, fg = lane >> 4;
  const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
  const int K = a.C1, nsteps = K / STEP;
  const T* ap = (const T*)a.src1 + (int64_t)min(m0 + fr, a.M - 1) * K + fg * VE;
  const T* wp = (const T*)a.w + (int64_t)(n0 + fr) * a.Kpad + fg * VE;
  f32x4 acc = {0, 0, 0, 0};
  for (int s = wv; s < nsteps; s += NWV) {
    const uint4 av = *reinterpret_cast<const uint4*>(ap + s * STEP);
    const uint4 wf = *reinterpret_cast<const uint4*>(wp + s * STEP);
    if (sizeof(T) == 4) {
      acc = mfma_16x16x4_f32(__uint_as_float(wf.x), __uint_as_float(av.x), acc);
      acc = mfma_16x16x4_f32(__uint_as_float(wf.y), __uint_as_float(av.y), acc);
      acc = mfma_16x16x4_f32(__uint_as_float(wf.z), __uint_as_float(av.z), acc);
      acc = mfma_16x16x4_f32(__uint_as_float(wf.w), __uint_as_float(av.w), acc);
    } else {
      s16x8 wa, pb;
      __builtin_memcpy(&wa, &wf, 16);
      __builtin_memcpy(&pb, &av, 16);
      acc = mfma_16x16x32_bf16(wa, pb, acc);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) red[wv][lane][r] = acc[r];
  __syncthreads();
  if (wv != 0) return;
  // this lane: output row m0 + fr, channels n0 + 4 fg .. + 3
  const int m = m0 + fr, co = n0 + fg * 4;
  float x[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float v = red[0][lane][r];
#pragma unroll
    for (int w = 1; w < NWV; w++) v += red[w][lane][r];
    if (a.bias && co + r < a.Cout) v += a.bias[co + r];
    x[r] = Elem<T>::rnd(act_fwd(v, a.act, a.slope));
  }
  if (m >= a.M) return;
  T* const op = (T*)a.dst1 + (int64_t)m * a.Cout + co;
  if (co + 3 < a.Cout && a.Cout % 4 == 0) st4(op, x);
  else {
#pragma unroll
    for (int r = 0; r < 4; r++) if (co + r < a.Cout) Elem<T>::st(op + r, x[r]);
  }
}
static bool conv_skinny_ok(const ConvArgs& a, int dtype) {
  const int step = dtype == 0 ? 16 : 32;
  return a.KH == 1 && a.KW == 1 && a.stride == 1 && a.dil == 1 && a.pad == 0 && a.C2 == 0 && !a.ups && !a.pool2 && a.D1 == a.Cout && a.K == a.C1 &&
         a.C1 % step == 0 && a.C1 >= 1024 && (int64_t)a.M * a.Cout <= 64 * 1024 && !a.stats && !a.add1 && !a.in_scale && !a.bn_y;
}
static void launch_linear_skinny(const ConvArgs& a, int dtype, hipStream_t st) {
  const dim3 grid((unsigned)cdiv(a.M, 16), (unsigned)cdiv(a.Cout, 16));
  if (dtype == 0) hipLaunchKernelGGL((linear_skinny_kernel<float>), grid, dim3(512), 0, st, a);
  else hipLaunchKernelGGL((linear_skinny_kernel<bf16_t>), grid, dim3(512), 0, st, a);
}

// ---- image stems: ONE 16-byte channel vector per input pixel (3 channels zero-padded to 8 bf16 / 4 fp32 by the caller), <= 32 outputs ----------
// RC-Net's 7x7 / stride-2 stem (utils/net_utils.py:29-91 via RCNet/networks.py ResNetEncoder) and the SML backbone's 3x3 / stride-2 stem on
// 0.6 M output pixels.  On the implicit-GEMM kernel a block tile of 128 pixels x 32 channels walks K = 448 in seven LDS stages of 8 MFMAs
// per wave behind a barrier each: 77 us for 78 MB.  Here a tap of a pixel IS one MFMA operand vector (8 channels = 16 bytes), so a lane
// fetches its fragments straight from global memory -- lane (pixel fr, k-group fg) of k-step ks reads tap 4 ks + fg of its pixel -- with every
// tap's request of a 16-pixel tile in flight together; the weight operand (Cout x ceil(taps / 4) k-steps) sits in LDS in fragment order for the
// block's persistent loop.  No pixel staging, no barrier until the statistics epilogue.  Taps past KH * KW read a clamped address and meet the
// packed operand's zero padding.
template <typename T, int NCT, int NS>
__global__ __launch_bounds__(256) void conv_stem_kernel(ConvArgs a, int ntile) {
  constexpr int VE = Elem<T>::VE, SE = 4 * VE;
  __shared__ uint4 sw[NCT * NS * 64];      // the weight operand in fragment order [channel tile][k step][lane]: one conflict-free 1-KiB read per use
  __shared__ float red[4 * 16 * NCT * 2];
  const int t = threadIdx.x, lane = t & 63, wv = RD_WAVE_UNIFORM(t >> 6), fr = lane & 15, fg = lane >> 4;
  const int ntap = a.KH * a.KW;
  for (int i = t; i < NCT * NS * 64; i += 256) {
    const int l = i & 63, cs = i >> 6, c = cs / NS, ks = cs - c * NS;
    sw[i] = *reinterpret_cast<const uint4*>((const T*)a.w + (int64_t)(c * 16 + (l & 15)) * a.Kpad + ks * SE + (l >> 4) * VE);
  }
  // this lane's taps: (kh, kw) of tap 4 ks + fg, -1 past the end
  int tkh[NS], tkw[NS];
#pragma unroll
  for (int ks = 0; ks < NS; ks++) {
    const int tap = ks * 4 + fg;
    tkh[ks] = tap < ntap ? tap / a.KW : -1; tkw[ks] = tap < ntap ? tap - (tap / a.KW) * a.KW : 0;
  }
  float ssum[NCT][4], ssq[NCT][4];
#pragma unroll
  for (int c = 0; c < NCT; c++)
#pragma unroll
    for (int r = 0; r < 4; r++) { ssum[c][r] = 0.f; ssq[c][r] = 0.f; }
  const T* const x = (const T*)a.src1;
  __syncthreads();
  // a wave takes 16-pixel tiles (one at a time: the 13 fragment requests of a 7x7 tile + accumulators stay under 128 registers, four waves
  // per SIMD hide the gather's round trip; two tiles and the weights in registers were 312 registers, one wave per SIMD, and SLOWER than
  // the implicit-GEMM kernel)
  for (int p = (int)blockIdx.x * 4 + wv; p < ntile; p += (int)gridDim.x * 4) {
    int64_t mm[2]; bool mvv[2];
    mm[0] = (int64_t)p * 16 + fr; mvv[0] = mm[0] < a.M; mm[1] = 0; mvv[1] = false;
    const int m = mvv[0] ? (int)mm[0] : 0;
    const int ow = m % a.OW, q = m / a.OW, pn = q / a.OH;
    const int ph = (q - pn * a.OH) * a.stride - a.pad, pw_ = ow * a.stride - a.pad;
    uint4 xv[NS];
#pragma unroll
    for (int ks = 0; ks < NS; ks++) {
      const int ih = ph + tkh[ks], iw = pw_ + tkw[ks];
      const bool ok = mvv[0] && tkh[ks] >= 0 && (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
      const uint4 v = *reinterpret_cast<const uint4*>(x + (((int64_t)pn * a.Hin + min(max(ih, 0), a.Hin - 1)) * a.Win + min(max(iw, 0), a.Win - 1)) * VE);
      xv[ks] = ok ? v : make_uint4(0, 0, 0, 0);      // unconditional load from a clamped address, zero selected afterwards
    }
    f32x4 acc[NCT][2];
#pragma unroll
    for (int c = 0; c < NCT; c++) { acc[c][0] = f32x4{0, 0, 0, 0}; acc[c][1] = f32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int ks = 0; ks < NS; ks++)
#pragma unroll
      for (int c = 0; c < NCT; c++) {
        const uint4 wfr = sw[(c * NS + ks) * 64 + lane];
        if (sizeof(T) == 4) {
          acc[c][0] = mfma_16x16x4_f32(__uint_as_float(wfr.x), __uint_as_float(xv[ks].x), acc[c][0]);
          acc[c][0] = mfma_16x16x4_f32(__uint_as_float(wfr.y), __uint_as_float(xv[ks].y), acc[c][0]);
          acc[c][0] = mfma_16x16x4_f32(__uint_as_float(wfr.z), __uint_as_float(xv[ks].z), acc[c][0]);
          acc[c][0] = mfma_16x16x4_f32(__uint_as_float(wfr.w), __uint_as_float(xv[ks].w), acc[c][0]);
        } else {
          s16x8 wa, pb;
          __builtin_memcpy(&wa, &wfr, 16);
          __builtin_memcpy(&pb, &xv[ks], 16);
          acc[c][0] = mfma_16x16x32_bf16(wa, pb, acc[c][0]);
        }
      }
    conv_epilogue_store<T, NCT>(a, acc, mm, mvv, 0, 0, fr, fg, ssum, ssq);
  }
  conv_epilogue_stats<NCT, 16 * NCT, 4>(a, ssum, ssq, 0, 0, wv, fr, fg, t, blockIdx.x, red);
}
static int conv_stem_min_m() { return rd_opt(OPT_CONV_STEM_MIN_M, 65536); }      // test hook: 0 routes small cases here
static bool conv_stem_ok(const ConvArgs& a, int dtype) {
  const int ve = dtype == 0 ? 4 : 8, taps = a.KH * a.KW;
  return a.C1 == ve && a.C2 == 0 && !a.ups && a.dil == 1 && !a.pool2 && !a.in_scale && !a.bn_y && a.KH == a.KW && (taps == 9 || taps == 49) &&
         (a.Cout == 16 || a.Cout == 32) && a.K == taps * ve && a.Kpad >= ((taps + 3) / 4) * 4 * ve && a.M >= conv_stem_min_m() &&
         (int64_t)a.N * a.Hin * a.Win < ((int64_t)1 << 27);
}
static int conv_stem_blocks(const ConvArgs& a) { return (int)std::min<int64_t>(cdiv(a.M, 16 * 4), 2048); }      // persistent blocks = statistics rows
static void launch_conv_stem(const ConvArgs& a, int dtype, hipStream_t st) {
  const int ntile = (int)cdiv(a.M, 16), taps = a.KH * a.KW;
  const dim3 grid((unsigned)conv_stem_blocks(a));
#define RD_STEM(TT, NCTV, NSV) hipLaunchKernelGGL((conv_stem_kernel<TT, NCTV, NSV>), grid, dim3(256), 0, st, a, ntile)
#define RD_STEM_T(TT) { if (a.Cout == 32) { if (taps == 49) RD_STEM(TT, 2, 13); else RD_STEM(TT, 2, 3); } else { if (taps == 49) RD_STEM(TT, 1, 13); else RD_STEM(TT, 1, 3); } }
  if (dtype == 0) RD_STEM_T(float) else RD_STEM_T(bf16_t)
#undef RD_STEM_T
#undef RD_STEM
}

void launch_conv(const ConvArgs& a, int dtype, hipStream_t st) {
  if (a.s2d) { launch_conv3x3_frag(a, dtype, st); return; }      // (rd_api.cpp checked conv_s2d_ok)
  if (a.d2s) { if (conv_d2s_small(a, dtype)) launch_conv3x3_small(a, dtype, st); else launch_conv3x3_frag(a, dtype, st); return; }      // (rd_api.cpp checked conv_d2s_ok)
  if (conv_skinny_ok(a, dtype)) { launch_linear_skinny(a, dtype, st); return; }
  if (conv_stem_ok(a, dtype)) { launch_conv_stem(a, dtype, st); return; }
  if (conv_few_ok(a)) { launch_conv_few(a, dtype, st); return; }
  if (conv1x1_direct_ok(a, dtype)) { launch_conv1x1_direct(a, dtype, st); return; }
  if (conv_pw_ok(a, dtype)) { launch_conv_pw(a, dtype, st); return; }
  if (conv3x3_c1_ok(a)) { launch_conv3x3_c1(a, dtype, st); return; }
  if (use_conv3x3_small(a, dtype)) { launch_conv3x3_small(a, dtype, st); return; }
  if (use_conv3x3_frag(a, dtype)) { launch_conv3x3_frag(a, dtype, st); return; }
  if (use_conv3x3(a, dtype)) { launch_conv3x3(a, dtype, st); return; }
  if (dtype == 0) launch_conv_t<float>(a, st);
  else launch_conv_t<bf16_t>(a, st);
}

static bool wgrad_tiny_shape(const WgradArgs& a);
static bool wgrad_tiny_shape_fwd(const WgradArgs& a) { return wgrad_tiny_shape(a); }
// out_reduce2 (ConvArgs::pool2): only the narrow-layer 3x3 kernel pairs rows / columns of its output tile in registers
bool conv_pool2_ok(const ConvArgs& a, int dtype) {
  return !conv_few_ok(a) && !conv1x1_direct_ok(a, dtype) && !conv3x3_c1_ok(a) && use_conv3x3_small(a, dtype) && !(a.OH & 1) && !(a.OW & 1) &&
         a.D1 == a.Cout && !a.bias && a.act == ACT_NONE;
}
// out_d2s (ConvArgs::d2s): the narrow-layer 3x3 kernel's D2S instantiation -- 16-bit activations, 64-byte source pixels (32 channels), 4 x 16 output
// channels -- or the register-fed kernel's (whole 128-byte channel chunks, D1 a multiple of 32)
static bool conv_d2s_small(const ConvArgs& a, int dtype) {
  return dtype != 0 && a.C2 == 0 && a.C1 == 32 && a.Cout == 64 && a.D1 == 16 && conv3x3_small_ok(a, dtype);
}
bool conv_d2s_ok(const ConvArgs& a, int dtype) {
  if (a.ups || a.pool2 || a.bias || a.act != ACT_NONE || a.in_scale || a.add1 || a.Cout != 4 * a.D1) return false;
  return conv_d2s_small(a, dtype) || conv3x3_frag_d2s_ok(a, dtype);      // (launch_conv sends a d2s descriptor straight to that kernel, whatever its tile count)
}
// in_s2d (ConvArgs::s2d): the register-fed kernel's S2D instantiations
bool conv_s2d_ok(const ConvArgs& a, int dtype) {
  if (a.ups || a.pool2 || a.d2s || a.bias || a.act != ACT_NONE || a.in_scale || a.add1 || a.C2 || a.D1 != a.Cout || a.stats) return false;
  return conv3x3_frag_s2d_ok(a, dtype);
}
// ConvArgs::add1: the kernels that store through conv_epilogue_store except the narrow-layer one (not the few-channel / single-channel
// streaming kernels, not the experimental LDS-DMA kernel), one destination, no 2x2 reduction
bool conv_add_ok(const ConvArgs& a, int dtype) {
  if (a.D1 != a.Cout || a.pool2 || conv_few_ok(a)) return false;
  if (conv1x1_direct_ok(a, dtype)) return true;
  if (conv3x3_c1_ok(a) || use_conv3x3_small(a, dtype)) return false;      // (the narrow-layer variants sit at their register caps)
  return true;
}
// ConvArgs::in_scale (consumer-side BatchNorm apply while staging): the two 3x3 / stride-1 kernels that stage whole 16-byte channel
// vectors of a pixel patch through registers (every 3x3 layer of RC-Net that reads a BatchNorm-ed convolution's output)
bool conv_in_affine_ok(const ConvArgs& a, int dtype) {
  if (conv_few_ok(a) || conv1x1_direct_ok(a, dtype) || conv3x3_c1_ok(a)) return false;
  return use_conv3x3_small(a, dtype) || use_conv3x3_frag(a, dtype);
}
// ConvArgs::bn_y (BatchNorm-backward sums in the data-gradient epilogue): kernels that store through conv_epilogue_store and write
// statistics rows; dz = the first destination
bool conv_bn_bwd_ok(const ConvArgs& a, int dtype) {
  // the register-fed 3x3 kernel (16x16x32 form) and the implicit-GEMM kernel: both write one statistics row per tile through
  // conv_epilogue_stats; dz must be whole 4-channel groups of the first destination, no 2x2 reduction, no bias / activation of its own
  if (a.pool2 || a.bias || a.act != ACT_NONE || (a.D1 & 3) || ((a.Cout - a.D1) & 3) || a.in_scale) return false;
  if (conv_skinny_ok(a, dtype) || conv_stem_ok(a, dtype) || conv_few_ok(a) || conv1x1_direct_ok(a, dtype) || conv3x3_c1_ok(a) || use_conv3x3_small(a, dtype)) return false;
  if (conv_pw_shape(a, dtype)) return false;      // (its epilogue takes the forward statistics only)
  if (use_conv3x3_frag(a, dtype)) return !conv3x3_frag_is32(a, dtype);
  if (use_conv3x3(a, dtype)) return false;      // (the patch-staged kernel is only a fallback since round 3)
  return true;                                  // implicit GEMM
}
bool wgrad_in_affine_ok(const WgradArgs& a, int dtype) { return !wgrad_tiny_shape_fwd(a) && wgrad3x3_tr_affine_ok(a, dtype); }
// name of the kernel launch_conv picks for this shape (bench.py groups its per-launch timings by the names rocprofv3 reports)
const char* conv_kernel_name(const ConvArgs& a, int dtype) {
  if (conv_skinny_ok(a, dtype)) return dtype == 0 ? "linear_skinny_kernel<float>" : "linear_skinny_kernel<" RD_T16_NAME ">";
  if (conv_stem_ok(a, dtype)) return "conv_stem_kernel";
  if (conv_few_ok(a)) return "conv_few_kernel";
  if (conv1x1_direct_ok(a, dtype)) return "conv1x1_direct_kernel";
  if (conv_pw_ok(a, dtype)) return conv_pw_name(a, dtype);
  if (conv3x3_c1_ok(a)) return "conv3x3_c1_kernel";
  if (a.s2d) return conv3x3_frag_name(a, dtype);
  if (a.d2s) return conv_d2s_small(a, dtype) ? conv3x3_small_name(a, dtype) : conv3x3_frag_name(a, dtype);
  if (use_conv3x3_small(a, dtype)) return conv3x3_small_name(a, dtype);
  if (use_conv3x3_frag(a, dtype)) return conv3x3_frag_name(a, dtype);
  if (use_conv3x3(a, dtype)) return conv3x3_patch_name(a, dtype);
  {     // the implicit-GEMM kernel's instantiation, as launch_conv_t picks it
    static thread_local char buf[96];
    const int ve = dtype == 0 ? 4 : 8, Cin = a.C1 + a.C2, es = dtype == 0 ? 4 : 2;
    const bool vec = (Cin % ve == 0) && (a.C1 % ve == 0);
    int bn, wm;
    conv_tiles(a.M, a.Cout, bn, wm);
    const bool deep = conv_gemm_deep(a, dtype), par = conv_gemm_par(a, dtype);
    (void)es;
    snprintf(buf, sizeof(buf), "conv_gemm_kernel<%s, %d, %s, %d, %s, %s>", dtype == 0 ? "float" : RD_T16_NAME, bn, vec ? "true" : "false", wm, deep ? "true" : "false", par ? "true" : "false");
    return buf;
  }
}
void launch_pack_weights(const float* w, void* out, int Cout, int Cin, int KH, int KW, int mode, int dtype,
                         hipStream_t st, int CinSrc) {
  if (CinSrc <= 0) CinSrc = Cin;
  int rows = mode == 2 ? 4 * Cout : (mode ? Cin : Cout), C = mode == 3 ? 4 * Cout : (mode == 1 ? Cout : Cin);
  int rows_pad = conv_rows_pad(rows), Kpad = conv_kpad(KH * KW * C, dtype);
  int64_t total = (int64_t)rows_pad * Kpad;
  unsigned grid = (unsigned)std::min<int64_t>(cdiv(total, 256), 4096);
  if (dtype == 0)
    hipLaunchKernelGGL((pack_weights_kernel<float>), dim3(grid), dim3(256), 0, st, w, (float*)out, Cout, Cin, KH, KW, mode, rows_pad, Kpad, CinSrc);
  else
    hipLaunchKernelGGL((pack_weights_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, w, (bf16_t*)out, Cout, Cin, KH, KW, mode, rows_pad, Kpad, CinSrc);
}

// the same with a block map (round 6): block b works on item map[b].x as block map[b].y of map[b].z -- the grid is the SUM of what the items
// need instead of 256 x n blocks of which 97 % exit at once (RC-Net: ~320 operands, 82 K blocks, a launch bound by block dispatch)
__global__ __launch_bounds__(256) void pack_weights_batch_map_kernel(const PackItem* __restrict__ items, const int4* __restrict__ map) {
  const int4 m = map[blockIdx.x];
  const PackItem it = items[m.x];
  const bool vec = rd_pack_vec_enabled(it);
  if (it.dtype == 0) { if (vec && pack_vec_ok<float>(it)) pack_one_vec<float>(it, m.y, m.z); else pack_one<float>(it, m.y, m.z); }
  else { if (vec && pack_vec_ok<bf16_t>(it)) pack_one_vec<bf16_t>(it, m.y, m.z); else pack_one<bf16_t>(it, m.y, m.z); }
}
void launch_pack_weights_batch_map(const void* items, int n, const void* map, int blocks, hipStream_t st) {
  (void)n;
  hipLaunchKernelGGL(pack_weights_batch_map_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const PackItem*)items, (const int4*)map);
}

void launch_pack_weights_batch(const void* items, int n, hipStream_t st) {
  static_assert(sizeof(PackItem) == 48, "PackItem mirrors rd_pack_item");
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(256, (unsigned)n), dim3(256), 0, st, (const PackItem*)items);  // small items: most blocks exit at once
}

// split the pixel reduction so the launch has a few blocks per CU, stage-aligned
int wgrad_nsplit(int M, int K, int Cout);
int wgrad_slabs(int M, int K, int Cout) {
  int ns = wgrad_nsplit(M, K, Cout);
  return (Cout <= 32 && K <= 9 * 64) ? std::max(ns, HALO_BLOCKS) : ns;  // the halo-tile path writes one slab per persistent block
}
int wgrad_nsplit(int M, int K, int Cout) {
  int cot = pick_bn(Cout);
  int64_t tiles = cdiv(K, 128) * cdiv(Cout, cot);
  const int target = rd_opt(OPT_WGRAD_BLOCKS, 1024);   // experiment hook (rd_set_option)
  int64_t want = cdiv(target, tiles);
  int64_t maxs = cdiv(M, 64);  // at least two 32-pixel stages per split (small-M GEMMs such as the LoFTR projections need the blocks)
  int64_t s = std::max<int64_t>(1, std::min(want, maxs));
  if (s >= 8) s -= s % 8;   // XCD x owns the splits x, x+8, ...: a multiple of 8 keeps the XCDs balanced
  return (int)s;
}

template <typename T>
static void launch_wgrad_t(const WgradArgs& a, bool vec, hipStream_t st) {
  int cot = pick_bn(a.Cout);
  dim3 grid(wg_grid(a.K, a.Cout, cot, a.nsplit));
#define RD_WG_CASE(C)                                                                              \
  if (cot == C) {                                                                                  \
    if (vec) hipLaunchKernelGGL((conv_wgrad_kernel<T, true, C>), grid, dim3(256), 0, st, a);        \
    else hipLaunchKernelGGL((conv_wgrad_kernel<T, false, C>), grid, dim3(256), 0, st, a);           \
  }
  RD_WG_CASE(16) RD_WG_CASE(32) RD_WG_CASE(64) RD_WG_CASE(128)
#undef RD_WG_CASE
}

template <typename T>
static void launch_wgrad_halo_t(const WgradArgs& a, int nblk, hipStream_t st) {
  const int Cin = a.C1 + a.C2, rt = a.Cout <= 16 ? 1 : 2;
  int TH, TW;
  halo_geom(Cin, a.Cout, TH, TW);
  int tilesH = (int)cdiv(a.OH, TH), tilesW = (int)cdiv(a.OW, TW);
#define RD_HALO(CTI, RT) hipLaunchKernelGGL((conv_wgrad_halo_kernel<T, CTI, RT>), dim3(nblk), dim3(256), 0, st, a, tilesH, tilesW)
  if (Cin == 16) { if (rt == 1) RD_HALO(1, 1); else RD_HALO(1, 2); }
  else if (Cin == 32) { if (rt == 1) RD_HALO(2, 1); else RD_HALO(2, 2); }
  else { if (rt == 1) RD_HALO(4, 1); else RD_HALO(4, 2); }
#undef RD_HALO
}

// ---- weight gradients with a handful of channels and millions of pixels (SML: the 3->3 `first` convolution and the 32->1 output head).  On the MFMA kernels they use 2-4 % of a tile (0.15-0.32 ms each for 16-32 MB of operands).  Here a thread owns a
// pixel stride and keeps all K x G products (K = KH*KW*Cin taps, G output channels of its group) in registers: HBM-bound streaming with
// unconditional, clamped tap loads; the block's sums go out as one slab, summed by the common reduction. ----
template <typename T, int KH, int CIN, int G, int S>
__global__ __launch_bounds__(256) void conv_wgrad_tiny_kernel(WgradArgs a) {
  constexpr int K = KH * KH * CIN, VE = Elem<T>::VE;
  __shared__ float red[4][K * G];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int co0 = blockIdx.y * G;
  const int64_t per = cdiv((int64_t)a.M, (int64_t)gridDim.x);
  const int64_t mbeg = (int64_t)blockIdx.x * per, mend = mbeg + per < a.M ? mbeg + per : a.M;
  const T* x = (const T*)a.src1;
  const T* dy = (const T*)a.dy;
  float acc[K][G];
#pragma unroll
  for (int k = 0; k < K; k++)
#pragma unroll
    for (int e = 0; e < G; e++) acc[k][e] = 0.f;
  for (int64_t m = mbeg + t; m < mend; m += 256) {
    const int ow = (int)(m % a.OW); const int64_t q = m / a.OW; const int oh = (int)(q % a.OH); const int n = (int)(q / a.OH);
    float g[G];
#pragma unroll
    for (int e = 0; e < G; e++) {
      const bool ok = co0 + e < a.Cout;
      const float v = Elem<T>::ld(dy + m * a.Cout + (ok ? co0 + e : 0));
      g[e] = ok ? v : 0.f;
    }
#pragma unroll
    for (int kh = 0; kh < KH; kh++)
#pragma unroll
      for (int kw = 0; kw < KH; kw++) {
        const int ih = oh * S - a.pad + kh, iw = ow * S - a.pad + kw;
        const bool ok = (unsigned)ih < (unsigned)a.Hin && (unsigned)iw < (unsigned)a.Win;
        const T* px = x + (((int64_t)n * a.Hin + min(max(ih, 0), a.Hin - 1)) * a.Win + min(max(iw, 0), a.Win - 1)) * CIN;
        float xv[CIN];
        if constexpr (CIN % VE == 0) {
#pragma unroll
          for (int c = 0; c < CIN; c += VE) {
            float v[VE];
            ldv(px + c, v);
#pragma unroll
            for (int e = 0; e < VE; e++) xv[c + e] = ok ? v[e] : 0.f;
          }
        } else {
#pragma unroll
          for (int c = 0; c < CIN; c++) { const float v = Elem<T>::ld(px + c); xv[c] = ok ? v : 0.f; }
        }
#pragma unroll
        for (int c = 0; c < CIN; c++)
#pragma unroll
          for (int e = 0; e < G; e++) acc[(kh * KH + kw) * CIN + c][e] += xv[c] * g[e];
      }
  }
#pragma unroll
  for (int k = 0; k < K; k++)
#pragma unroll
    for (int e = 0; e < G; e++) {
      const float v = wave_sum(acc[k][e]);
      if (lane == 0) red[wv][k * G + e] = v;
    }
  __syncthreads();
  for (int i = t; i < K * G; i += 256) {
    const int k = i / G, e = i - k * G;
    if (co0 + e < a.Cout)
      a.slab[((int64_t)blockIdx.x * a.Cout + co0 + e) * K + k] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}
// -> number of slabs written (0: shape not handled here)
static bool wgrad_tiny_shape(const WgradArgs& a) {
  const int min_m = rd_opt(OPT_WGRAD_TINY_MIN_M, 1 << 16);      // test hook (rd_set_option): 0 sends small cases through this kernel
  if (a.ups || a.C2 || a.KH != a.KW || a.M < min_m || wgrad_slabs(a.M, a.K, a.Cout) < HALO_BLOCKS) return false;
  // (the 3->32 stride-2 stem was tried here too, 8 groups of 4 output channels: 0.33 ms against 0.15 ms on the generic kernel -- every
  // group re-reads the 27-tap patch -- so it stays there)
  return (a.KH == 3 && a.C1 == 3 && a.stride == 1 && a.Cout <= 4) || (a.KH == 1 && a.C1 == 32 && a.stride == 1 && a.Cout == 1);
}
bool wgrad_streams(const WgradArgs& a) { return wgrad_tiny_shape(a); }
template <typename T>
static int launch_wgrad_tiny(const WgradArgs& a, hipStream_t st) {
  if (!wgrad_tiny_shape(a)) return 0;
  const int nblk = (int)std::min<int64_t>(HALO_BLOCKS, cdiv(a.M, 256 * 4));      // the workspace holds >= HALO_BLOCKS slabs for these shapes
#define RD_TINY(KHV, CINV, GV, SV) { dim3 grid((unsigned)nblk, (unsigned)cdiv(a.Cout, GV));                                       \
    hipLaunchKernelGGL((conv_wgrad_tiny_kernel<T, KHV, CINV, GV, SV>), grid, dim3(256), 0, st, a); return nblk; }
  if (a.KH == 3 && a.C1 == 3 && a.stride == 1 && a.Cout <= 4) { if (a.Cout <= 3) RD_TINY(3, 3, 3, 1) else RD_TINY(3, 3, 4, 1) }
  if (a.KH == 1 && a.C1 == 32 && a.stride == 1 && a.Cout == 1) RD_TINY(1, 32, 1, 1)
#undef RD_TINY
  return 0;
}

void launch_wgrad_reduce_batch(const WgradReduceItem* items, int n, hipStream_t st) {
  int i = 0;
  while (i < n) {
    WgradReduceBatch b; b.n = 0; b.first[0] = 0;
    for (; i < n && b.n < WGRAD_BATCH_MAX; i++) {
      const WgradReduceItem& q = items[i];
      const int64_t total = (int64_t)q.Cout * q.KH * q.KW * q.Cin;
      if (total % 4) { launch_wgrad_reduce(q.slab, q.dw, q.Cout, q.Cin, q.KH, q.KW, q.nsplit, q.accumulate, st); continue; }
      const int SL = wgrad_reduce_sl(q.nsplit);
      const int nb = (int)std::min<int64_t>(cdiv(total / 4, 256 / SL), 1024);
      b.item[b.n] = q; b.SL[b.n] = SL; b.first[b.n + 1] = b.first[b.n] + nb; b.n++;
    }
    if (b.n) hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3((unsigned)b.first[b.n]), dim3(256), 0, st, b);
  }
}

// defer != nullptr: the partial slabs are produced, the reduction into dw is left to a later launch_wgrad_reduce_batch over *defer
void launch_wgrad(WgradArgs a, int dtype, float* dw, int accumulate, hipStream_t st, WgradReduceItem* defer) {
  const int Cin = a.C1 + a.C2;
  auto reduce = [&](int nsplit) {
    if (defer) { defer->slab = a.slab; defer->dw = dw; defer->Cout = a.Cout; defer->Cin = Cin; defer->KH = a.KH; defer->KW = a.KW;
                 defer->nsplit = nsplit; defer->accumulate = accumulate; }
    else launch_wgrad_reduce(a.slab, dw, a.Cout, Cin, a.KH, a.KW, nsplit, accumulate, st);
  };
  if (wgrad_tiny_shape(a)) {      // few channels, millions of pixels: register-accumulating streaming kernel
    const int ns = dtype == 0 ? launch_wgrad_tiny<float>(a, st) : launch_wgrad_tiny<bf16_t>(a, st);
    if (ns) { reduce(ns); return; }
  }
  if (wgrad3x3_tr_ok(a, dtype)) {  // bf16 narrow layers: transpose-read kernel, one slab per persistent block (<= HALO_BLOCKS)
    launch_wgrad3x3_tr(a, st);
    reduce(wgrad3x3_tr_blocks(a));
    return;
  }
  if (wgrad_halo_ok(a, dtype)) {
    int TH, TW;
    halo_geom(Cin, a.Cout, TH, TW);
    int64_t ntiles = (int64_t)a.N * cdiv(a.OH, TH) * cdiv(a.OW, TW);
    int nblk = (int)std::min<int64_t>(ntiles, HALO_BLOCKS);
    if (dtype == 0) launch_wgrad_halo_t<float>(a, nblk, st);
    else launch_wgrad_halo_t<bf16_t>(a, nblk, st);
    reduce(nblk);
    return;
  }
  a.nsplit = wgrad_nsplit(a.M, a.K, a.Cout);
  a.rows_per_split = (int)(cdiv(cdiv(a.M, a.nsplit), 32) * 32);
  a.nsplit = (int)cdiv(a.M, a.rows_per_split);
  bool vec = (Cin % 4 == 0) && (a.C1 % 4 == 0);
  if (dtype == 1 && (Cin % 8 == 0) && (a.C1 % 8 == 0) && (a.Cout % 8 == 0)) {
    // bf16 MFMA path: 64-pixel stages
    a.rows_per_split = (int)(cdiv(cdiv(a.M, wgrad_nsplit(a.M, a.K, a.Cout)), 64) * 64);
    a.nsplit = (int)cdiv(a.M, a.rows_per_split);
    int cot = pick_bn(a.Cout);
    dim3 grid(wg_grid(a.K, a.Cout, cot, a.nsplit));
    if (cot == 16) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<16>), grid, dim3(256), 0, st, a);
    else if (cot == 32) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<32>), grid, dim3(256), 0, st, a);
    else if (cot == 64) hipLaunchKernelGGL((conv_wgrad_bf16_kernel<64>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgrad_bf16_kernel<128>), grid, dim3(256), 0, st, a);
    reduce(a.nsplit);
    return;
  }
  if (dtype == 0) launch_wgrad_t<float>(a, vec, st);
  else launch_wgrad_t<bf16_t>(a, vec, st);
  reduce(a.nsplit);
}

const char* wgrad_kernel_name(const WgradArgs& a, int dtype) {
  const int Cin = a.C1 + a.C2;
  if (wgrad_tiny_shape(a)) return "conv_wgrad_tiny_kernel";
  if (wgrad3x3_tr_ok(a, dtype)) return wgrad3x3_tr_name(a);
  if (wgrad_halo_ok(a, dtype)) return "conv_wgrad_halo_kernel";
  if (dtype == 1 && (Cin % 8 == 0) && (a.C1 % 8 == 0) && (a.Cout % 8 == 0)) {
    static thread_local char buf[48];
    snprintf(buf, sizeof(buf), "conv_wgrad_bf16_kernel<%d>", pick_bn(a.Cout));
    return buf;
  }
  return "conv_wgrad_kernel";
}

}  // namespace rd
