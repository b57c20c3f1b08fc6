// Depthwise convolution, bilinear x2 resampling, the SML prediction head and small elementwise helpers.
// All HBM-bound (depthwise: 2*k*k FLOP per 4-byte element), NHWC, coalesced over channels.
//
// Reference call sites:
//   modules/midas/blocks.py:44-64  tf_efficientnet_lite3 (torch.hub, third-party: geffnet DepthwiseSeparableConv /
//                                  InvertedResidual: conv_dw k3/k5 stride 1/2, TF-"SAME" padding, BN eps 1e-3, ReLU6)
//   modules/midas/blocks.py:168-170 F.interpolate(scale_factor=2, bilinear, align_corners=True)
//   modules/midas/blocks.py:187     nn.Upsample(scale_factor=2, mode="bilinear")  (align_corners=False)
//   modules/midas/midas_net_custom.py:121-133  scales = relu(1 + out); pred = d * scales; in-place clamps
//   train_zju.py:355-356            d = 1/d; sml_pred = 1/sml_pred
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

static unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 4096)); }

// weights: OIHW with I = 1 -> w[c*k*k + kh*k + kw]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ y,
                                                         int N, int H, int W, int C, int OH, int OW, int k, int s, int p) {
  const int64_t total = (int64_t)N * OH * OW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int ow = (int)(q % OW); q /= OW; int oh = (int)(q % OH); int n = (int)(q / OH);
    float acc = 0.f;
    for (int kh = 0; kh < k; kh++) {
      int ih = oh * s - p + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < k; kw++) {
        int iw = ow * s - p + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        acc += Elem<T>::ld(x + (((int64_t)n * H + ih) * W + iw) * C + c) * w[(c * k + kh) * k + kw];
      }
    }
    Elem<T>::st(y + i, acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                           int N, int H, int W, int C, int OH, int OW, int k, int s, int p) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int iw = (int)(q % W); q /= W; int ih = (int)(q % H); int n = (int)(q / H);
    float acc = 0.f;
    for (int kh = 0; kh < k; kh++) {
      int t = ih + p - kh;
      if (t < 0 || (t % s)) continue;
      int oh = t / s;
      if (oh >= OH) continue;
      for (int kw = 0; kw < k; kw++) {
        int u = iw + p - kw;
        if (u < 0 || (u % s)) continue;
        int ow = u / s;
        if (ow >= OW) continue;
        acc += Elem<T>::ld(dy + (((int64_t)n * OH + oh) * OW + ow) * C + c) * w[(c * k + kh) * k + kw];
      }
    }
    Elem<T>::st(dx + i, acc);
  }
}

// partial[row][c][k*k]: thread (channel cl, pixel lane pl) accumulates all taps of its channel over its pixels
template <typename T, int KK>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                           int N, int H, int W, int C, int OH, int OW, int k, int s, int p, int CB,
                                                           int PL) {
  __shared__ float red[256];
  const int t = threadIdx.x, cl = t % CB, pl = t / CB;
  const int c = blockIdx.y * CB + cl;
  const int64_t pixels = (int64_t)N * OH * OW;
  const int64_t per = cdiv(pixels, gridDim.x);
  const int64_t pbeg = (int64_t)blockIdx.x * per, pend = pbeg + per < pixels ? pbeg + per : pixels;
  float acc[KK];
#pragma unroll
  for (int j = 0; j < KK; j++) acc[j] = 0.f;
  if (c < C) {
    for (int64_t m = pbeg + pl; m < pend; m += PL) {
      int ow = (int)(m % OW); int64_t q = m / OW; int oh = (int)(q % OH); int n = (int)(q / OH);
      float g = Elem<T>::ld(dy + m * C + c);
#pragma unroll
      for (int j = 0; j < KK; j++) {
        int kh = j / k, kw = j - kh * k;
        int ih = oh * s - p + kh, iw = ow * s - p + kw;
        if (j < k * k && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W)
          acc[j] += g * Elem<T>::ld(x + (((int64_t)n * H + ih) * W + iw) * C + c);
      }
    }
  }
  for (int j = 0; j < k * k; j++) {
    float v = 0.f;
#pragma unroll
    for (int jj = 0; jj < KK; jj++) if (jj == j) v = acc[jj];
    red[t] = v;
    __syncthreads();
    if (pl == 0 && c < C) {
      float sacc = 0.f;
      for (int q = 0; q < PL; q++) sacc += red[q * CB + cl];
      partial[((int64_t)blockIdx.x * C + c) * (k * k) + j] = sacc;
    }
    __syncthreads();
  }
}


// ---- channel-vectorised depthwise kernels (C % 4 == 0): thread = (4 channels, pixel lane); the k*k x 4 weights of the
// thread's channels live in registers, every tap is one 16-byte (fp32) / 8-byte (bf16) load ------------------------------------
template <typename T, int K, int MODE>  // MODE 0: forward, 1: data gradient
__global__ __launch_bounds__(256) void dwconv_vec_kernel(const T* __restrict__ src, const float* __restrict__ w, T* __restrict__ dst, int N,
                                                         int H, int W, int C, int OH, int OW, int s, int p, int C4B, int PL, int PPB) {
  const int t = threadIdx.x, cl = t % C4B, pl = t / C4B;
  const int c = (blockIdx.y * C4B + cl) * 4;
  if (c >= C) return;
  float wr[K * K][4];
#pragma unroll
  for (int j = 0; j < K * K; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) wr[j][e] = w[(c + e) * K * K + j];
  const int DH = MODE ? H : OH, DW = MODE ? W : OW;           // destination spatial size
  const int64_t M = (int64_t)N * DH * DW;
  const int64_t mbeg = (int64_t)blockIdx.x * PPB, mend = mbeg + PPB < M ? mbeg + PPB : M;
  for (int64_t m = mbeg + pl; m < mend; m += PL) {
    int dw_ = (int)(m % DW); int64_t q = m / DW; int dh = (int)(q % DH); int n = (int)(q / DH);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < K; kh++) {
      int sh;
      if (MODE == 0) { sh = dh * s - p + kh; if ((unsigned)sh >= (unsigned)H) continue; }
      else { int u = dh + p - kh; if (u < 0 || (u % s)) continue; sh = u / s; if (sh >= OH) continue; }
#pragma unroll
      for (int kw = 0; kw < K; kw++) {
        int sw;
        if (MODE == 0) { sw = dw_ * s - p + kw; if ((unsigned)sw >= (unsigned)W) continue; }
        else { int u = dw_ + p - kw; if (u < 0 || (u % s)) continue; sw = u / s; if (sw >= OW) continue; }
        const int SH = MODE ? OH : H, SW = MODE ? OW : W;
        float v[4];
        ld4(src + (((int64_t)n * SH + sh) * SW + sw) * C + c, v);
#pragma unroll
        for (int e = 0; e < 4; e++) acc[e] += v[e] * wr[kh * K + kw][e];
      }
    }
    st4(dst + m * C + c, acc);
  }
}

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_vec_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                               int N, int H, int W, int C, int OH, int OW, int s, int p, int C4B, int PL) {
  __shared__ float red[256][4];
  const int t = threadIdx.x, cl = t % C4B, pl = t / C4B;
  const int c = (blockIdx.y * C4B + cl) * 4;
  const bool cv = c < C;
  const int64_t pixels = (int64_t)N * OH * OW;
  const int64_t per = cdiv(pixels, gridDim.x);
  const int64_t pbeg = (int64_t)blockIdx.x * per, pend = pbeg + per < pixels ? pbeg + per : pixels;
  float acc[K * K][4];
#pragma unroll
  for (int j = 0; j < K * K; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[j][e] = 0.f;
  if (cv) {
    for (int64_t m = pbeg + pl; m < pend; m += PL) {
      int ow = (int)(m % OW); int64_t q = m / OW; int oh = (int)(q % OH); int n = (int)(q / OH);
      float g[4];
      ld4(dy + m * C + c, g);
#pragma unroll
      for (int kh = 0; kh < K; kh++) {
        int ih = oh * s - p + kh;
        if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
        for (int kw = 0; kw < K; kw++) {
          int iw = ow * s - p + kw;
          if ((unsigned)iw >= (unsigned)W) continue;
          float v[4];
          ld4(x + (((int64_t)n * H + ih) * W + iw) * C + c, v);
#pragma unroll
          for (int e = 0; e < 4; e++) acc[kh * K + kw][e] += g[e] * v[e];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K * K; j++) {
#pragma unroll
    for (int e = 0; e < 4; e++) red[t][e] = acc[j][e];
    __syncthreads();
    if (pl == 0 && cv) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float sacc = 0.f;
        for (int q = 0; q < PL; q++) sacc += red[q * C4B + cl][e];
        partial[((int64_t)blockIdx.x * C + c + e) * (K * K) + j] = sacc;
      }
    }
    __syncthreads();
  }
}


// ---- stride-1 forms with a sliding register window: a thread produces DWR = 4 consecutive outputs of a row, so a filter row costs
// DWR + K - 1 loads instead of DWR * K (5x5: 10 loads per output instead of 25; the per-tap form above is bound by L1 load issue). ----
static constexpr int DWR = 4;
template <typename T, int K, int MODE>  // MODE 0: forward, 1: data gradient (stride 1)
__global__ __launch_bounds__(256) void dwconv_run_kernel(const T* __restrict__ src, const float* __restrict__ w, T* __restrict__ dst, int N,
                                                         int H, int W, int C, int OH, int OW, int p, int C4B, int PL, int PPB) {
  const int t = threadIdx.x, cl = t % C4B, pl = t / C4B;
  const int c = (blockIdx.y * C4B + cl) * 4;
  if (c >= C) return;
  float wr[K * K][4];
#pragma unroll
  for (int j = 0; j < K * K; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) wr[j][e] = w[(c + e) * K * K + j];
  const int DH = MODE ? H : OH, DW = MODE ? W : OW;           // destination size
  const int SH = MODE ? OH : H, SW = MODE ? OW : W;           // source size
  const int runs = (DW + DWR - 1) / DWR;
  const int64_t M = (int64_t)N * DH * runs;
  const int64_t mbeg = (int64_t)blockIdx.x * PPB, mend = mbeg + PPB < M ? mbeg + PPB : M;
  for (int64_t m = mbeg + pl; m < mend; m += PL) {
    const int rw = (int)(m % runs); int64_t q = m / runs; const int dh = (int)(q % DH); const int n = (int)(q / DH);
    const int dw0 = rw * DWR;
    const int cbase = MODE ? dw0 + p - (K - 1) : dw0 - p;     // source column of window slot 0
    float acc[DWR][4];
#pragma unroll
    for (int r = 0; r < DWR; r++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[r][e] = 0.f;
#pragma unroll
    for (int kh = 0; kh < K; kh++) {
      const int sh = MODE ? dh + p - kh : dh - p + kh;
      if ((unsigned)sh >= (unsigned)SH) continue;
      const T* row = src + (((int64_t)n * SH + sh) * SW) * C + c;
      float v[DWR + K - 1][4];
#pragma unroll
      for (int j = 0; j < DWR + K - 1; j++) {
        const int sw = cbase + j;
        if ((unsigned)sw < (unsigned)SW) ld4(row + (int64_t)sw * C, v[j]);
        else { v[j][0] = 0.f; v[j][1] = 0.f; v[j][2] = 0.f; v[j][3] = 0.f; }
      }
#pragma unroll
      for (int kw = 0; kw < K; kw++)
#pragma unroll
        for (int r = 0; r < DWR; r++)
#pragma unroll
          for (int e = 0; e < 4; e++) acc[r][e] += v[MODE ? r + K - 1 - kw : r + kw][e] * wr[kh * K + kw][e];
    }
#pragma unroll
    for (int r = 0; r < DWR; r++)
      if (dw0 + r < DW) st4(dst + (((int64_t)n * DH + dh) * DW + dw0 + r) * C + c, acc[r]);
  }
}

template <typename T, int K>
__global__ __launch_bounds__(256) void dwconv_wgrad_run_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                               int N, int H, int W, int C, int OH, int OW, int p, int C4B, int PL) {
  __shared__ float red[256][4];
  const int t = threadIdx.x, cl = t % C4B, pl = t / C4B;
  const int c = (blockIdx.y * C4B + cl) * 4;
  const bool cv = c < C;
  const int runs = (OW + DWR - 1) / DWR;
  const int64_t units = (int64_t)N * OH * runs;
  const int64_t per = cdiv(units, gridDim.x);
  const int64_t ubeg = (int64_t)blockIdx.x * per, uend = ubeg + per < units ? ubeg + per : units;
  float acc[K * K][4];
#pragma unroll
  for (int j = 0; j < K * K; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[j][e] = 0.f;
  if (cv) {
    for (int64_t m = ubeg + pl; m < uend; m += PL) {
      const int rw = (int)(m % runs); int64_t q = m / runs; const int oh = (int)(q % OH); const int n = (int)(q / OH);
      const int ow0 = rw * DWR;
      float g[DWR][4];
#pragma unroll
      for (int r = 0; r < DWR; r++) {
        if (ow0 + r < OW) ld4(dy + (((int64_t)n * OH + oh) * OW + ow0 + r) * C + c, g[r]);
        else { g[r][0] = 0.f; g[r][1] = 0.f; g[r][2] = 0.f; g[r][3] = 0.f; }
      }
#pragma unroll
      for (int kh = 0; kh < K; kh++) {
        const int ih = oh - p + kh;
        if ((unsigned)ih >= (unsigned)H) continue;
        const T* row = x + (((int64_t)n * H + ih) * W) * C + c;
        float v[DWR + K - 1][4];
#pragma unroll
        for (int j = 0; j < DWR + K - 1; j++) {
          const int iw = ow0 - p + j;
          if ((unsigned)iw < (unsigned)W) ld4(row + (int64_t)iw * C, v[j]);
          else { v[j][0] = 0.f; v[j][1] = 0.f; v[j][2] = 0.f; v[j][3] = 0.f; }
        }
#pragma unroll
        for (int kw = 0; kw < K; kw++)
#pragma unroll
          for (int r = 0; r < DWR; r++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[kh * K + kw][e] += g[r][e] * v[r + kw][e];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < K * K; j++) {
#pragma unroll
    for (int e = 0; e < 4; e++) red[t][e] = acc[j][e];
    __syncthreads();
    if (pl == 0 && cv) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float sacc = 0.f;
        for (int q = 0; q < PL; q++) sacc += red[q * C4B + cl][e];
        partial[((int64_t)blockIdx.x * C + c + e) * (K * K) + j] = sacc;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void dwconv_wgrad_finalize_kernel(const float* __restrict__ partial, int rows, int CK, float* dw,
                                                                   int accumulate) {
  const int i = blockIdx.x, lane = threadIdx.x;
  double a = 0.0;
  for (int r = lane; r < rows; r += 64) a += partial[(int64_t)r * CK + i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
  if (lane == 0) dw[i] = accumulate ? dw[i] + (float)a : (float)a;
}

// per-channel (sum, sum^2) partials of an NHWC tensor: BatchNorm statistics for layers whose producer has no fused epilogue
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ y, float* __restrict__ partial, int64_t pixels, int C,
                                                       int CB, int PL) {
  __shared__ float red[2][256];
  const int t = threadIdx.x, cl = t % CB, pl = t / CB;
  const int c = blockIdx.y * CB + cl;
  const int64_t per = cdiv(pixels, gridDim.x);
  const int64_t pbeg = (int64_t)blockIdx.x * per, pend = pbeg + per < pixels ? pbeg + per : pixels;
  float a = 0.f, b = 0.f;
  if (c < C)
    for (int64_t m = pbeg + pl; m < pend; m += PL) { float v = Elem<T>::ld(y + m * C + c); a += v; b += v * v; }
  red[0][t] = a; red[1][t] = b;
  __syncthreads();
  if (pl == 0 && c < C) {
    float sa = 0.f, sb = 0.f;
    for (int q = 0; q < PL; q++) { sa += red[0][q * CB + cl]; sb += red[1][q * CB + cl]; }
    partial[((int64_t)blockIdx.x * C + c) * 2] = sa;
    partial[((int64_t)blockIdx.x * C + c) * 2 + 1] = sb;
  }
}

// ---- bilinear x2 (ATen upsample_bilinear2d index arithmetic) ------------------------------------------------------
__device__ __forceinline__ void bil_src(int d, int in, int out, int align, int& i0, int& i1, float& l1) {
  float src;
  if (align) src = (out > 1) ? (float)d * ((float)(in - 1) / (float)(out - 1)) : 0.f;
  else { src = ((float)d + 0.5f) * ((float)in / (float)out) - 0.5f; if (src < 0.f) src = 0.f; }
  i0 = (int)src; if (i0 > in - 1) i0 = in - 1;
  i1 = i0 < in - 1 ? i0 + 1 : i0;
  l1 = src - (float)i0;
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int OH,
                                                           int OW, int align) {
  const int64_t total = (int64_t)N * OH * OW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int ow = (int)(q % OW); q /= OW; int oh = (int)(q % OH); int n = (int)(q / OH);
    int h0, h1, w0, w1; float lh, lw;
    bil_src(oh, H, OH, align, h0, h1, lh);
    bil_src(ow, W, OW, align, w0, w1, lw);
    const T* b = x + (int64_t)n * H * W * C + c;
    float v00 = Elem<T>::ld(b + ((int64_t)h0 * W + w0) * C), v01 = Elem<T>::ld(b + ((int64_t)h0 * W + w1) * C);
    float v10 = Elem<T>::ld(b + ((int64_t)h1 * W + w0) * C), v11 = Elem<T>::ld(b + ((int64_t)h1 * W + w1) * C);
    Elem<T>::st(y + i, (1.f - lh) * ((1.f - lw) * v00 + lw * v01) + lh * ((1.f - lw) * v10 + lw * v11));
  }
}
// deterministic gather: every source pixel collects the weights with which destination pixels read it
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                           int OW, int align) {
  const int64_t total = (int64_t)N * H * W * C;
  const int rh = (OH + H - 1) / H + 2, rw = (OW + W - 1) / W + 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int w = (int)(q % W); q /= W; int h = (int)(q % H); int n = (int)(q / H);
    int ohc = (int)(((int64_t)h * OH) / H), owc = (int)(((int64_t)w * OW) / W);
    float g = 0.f;
    for (int oh = max(ohc - rh, 0); oh <= min(ohc + rh, OH - 1); oh++) {
      int h0, h1; float lh; int dum0, dum1; (void)dum0; (void)dum1;
      bil_src(oh, H, OH, align, h0, h1, lh);
      float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
      if (wh == 0.f) continue;
      for (int ow = max(owc - rw, 0); ow <= min(owc + rw, OW - 1); ow++) {
        int w0, w1; float lw;
        bil_src(ow, W, OW, align, w0, w1, lw);
        float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
        if (ww == 0.f) continue;
        g += wh * ww * Elem<T>::ld(dy + (((int64_t)n * OH + oh) * OW + ow) * C + c);
      }
    }
    Elem<T>::st(dx + i, g);
  }
}

// ---- SML head ---------------------------------------------------------------------------------------------------------
// pred = d * relu(1 + out); pred > hi -> hi; pred < lo -> lo   (hi = 1/min_pred, lo = 1/max_pred; <0 disables)
template <typename T>
__global__ __launch_bounds__(256) void sml_head_fwd_kernel(const T* __restrict__ out, const float* __restrict__ d, float* __restrict__ pred,
                                                           int64_t n, float hi, float lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float sc = 1.f + Elem<T>::ld(out + i); sc = sc > 0.f ? sc : 0.f;
    float p = d[i] * sc;
    if (hi > 0.f && p > hi) p = hi;
    if (lo > 0.f && p < lo) p = lo;
    pred[i] = p;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void sml_head_bwd_kernel(const T* __restrict__ out, const float* __restrict__ d, const float* __restrict__ dpred,
                                                           T* __restrict__ dout, int64_t n, float hi, float lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float sc = 1.f + Elem<T>::ld(out + i);
    float p = d[i] * (sc > 0.f ? sc : 0.f);
    bool clamped = (hi > 0.f && p > hi);
    if (!clamped && lo > 0.f && p < lo) clamped = true;
    Elem<T>::st(dout + i, (sc > 0.f && !clamped) ? dpred[i] * d[i] : 0.f);
  }
}
// y = 1/x ; dx = -dy / x^2   (fp32)
__global__ __launch_bounds__(256) void reciprocal_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                                                         int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float v = x[i];
    out[i] = dy ? -dy[i] / (v * v) : 1.f / v;
  }
}

// ---- launchers -----------------------------------------------------------------------------------------------------------
struct RG { int CB, PL, nchunk; };
static RG rgeom(int C) { RG g; int cb = 1; while (cb < C && cb < 256) cb <<= 1; g.CB = cb; g.PL = 256 / cb; g.nchunk = (int)cdiv(C, cb); return g; }
int dw_rows(int64_t pixels, int C) {
  RG g = rgeom(C);
  return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(pixels, (int64_t)g.PL * 64), 512));
}

struct VG { int C4B, PL, PPB, nchunk; };
static VG vgeom(int C) {
  VG g; int c4 = C / 4; int cb = 1; while (cb < c4 && cb < 64) cb <<= 1;
  g.C4B = cb; g.PL = 256 / cb; g.PPB = g.PL * 16; g.nchunk = (int)cdiv(c4, cb);
  return g;
}
template <typename T, int MODE>
static void launch_dw_vec(const void* src, const float* w, void* dst, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, hipStream_t st) {
  VG g = vgeom(C);
  if (s == 1) {  // sliding-window form: units of DWR outputs
    const int DH = MODE ? H : OH, DW = MODE ? W : OW;
    int64_t units = (int64_t)N * DH * cdiv(DW, DWR);
    const int ppb = g.PL * 4;
    dim3 rgrid((unsigned)cdiv(units, ppb), g.nchunk);
    if (k == 3) hipLaunchKernelGGL((dwconv_run_kernel<T, 3, MODE>), rgrid, dim3(256), 0, st, (const T*)src, w, (T*)dst, N, H, W, C, OH, OW, p, g.C4B, g.PL, ppb);
    else hipLaunchKernelGGL((dwconv_run_kernel<T, 5, MODE>), rgrid, dim3(256), 0, st, (const T*)src, w, (T*)dst, N, H, W, C, OH, OW, p, g.C4B, g.PL, ppb);
    return;
  }
  int64_t M = (int64_t)N * (MODE ? (int64_t)H * W : (int64_t)OH * OW);
  dim3 grid((unsigned)cdiv(M, g.PPB), g.nchunk);
  if (k == 3) hipLaunchKernelGGL((dwconv_vec_kernel<T, 3, MODE>), grid, dim3(256), 0, st, (const T*)src, w, (T*)dst, N, H, W, C, OH, OW, s, p, g.C4B, g.PL, g.PPB);
  else hipLaunchKernelGGL((dwconv_vec_kernel<T, 5, MODE>), grid, dim3(256), 0, st, (const T*)src, w, (T*)dst, N, H, W, C, OH, OW, s, p, g.C4B, g.PL, g.PPB);
}

void launch_dwconv_fwd(const void* x, const float* w, void* y, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, int dtype,
                       hipStream_t st) {
  if (C % 4 == 0 && (k == 3 || k == 5)) {
    if (dtype == 0) launch_dw_vec<float, 0>(x, w, y, N, H, W, C, OH, OW, k, s, p, st);
    else launch_dw_vec<bf16_t, 0>(x, w, y, N, H, W, C, OH, OW, k, s, p, st);
    return;
  }
  int64_t n = (int64_t)N * OH * OW * C;
  if (dtype == 0) hipLaunchKernelGGL((dwconv_fwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)x, w, (float*)y, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((dwconv_fwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)x, w, (bf16_t*)y, N, H, W, C, OH, OW, k, s, p);
}
void launch_dwconv_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int OH, int OW, int k, int s, int p,
                         int dtype, hipStream_t st) {
  if (C % 4 == 0 && (k == 3 || k == 5)) {
    if (dtype == 0) launch_dw_vec<float, 1>(dy, w, dx, N, H, W, C, OH, OW, k, s, p, st);
    else launch_dw_vec<bf16_t, 1>(dy, w, dx, N, H, W, C, OH, OW, k, s, p, st);
    return;
  }
  int64_t n = (int64_t)N * H * W * C;
  if (dtype == 0) hipLaunchKernelGGL((dwconv_dgrad_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)dy, w, (float*)dx, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((dwconv_dgrad_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)dy, w, (bf16_t*)dx, N, H, W, C, OH, OW, k, s, p);
}
void launch_dwconv_wgrad(const void* x, const void* dy, float* partial, float* dw, int accumulate, int N, int H, int W, int C, int OH,
                         int OW, int k, int s, int p, int dtype, hipStream_t st) {
  int rows = dw_rows((int64_t)N * OH * OW, C);
  if (C % 4 == 0 && (k == 3 || k == 5)) {
    VG v = vgeom(C);
    dim3 vgrid(rows, v.nchunk);
    if (s == 1) {
#define RD_DWR(T, K) hipLaunchKernelGGL((dwconv_wgrad_run_kernel<T, K>), vgrid, dim3(256), 0, st, (const T*)x, (const T*)dy, partial, N, H, W, C, OH, OW, p, v.C4B, v.PL)
      if (dtype == 0) { if (k == 3) RD_DWR(float, 3); else RD_DWR(float, 5); }
      else { if (k == 3) RD_DWR(bf16_t, 3); else RD_DWR(bf16_t, 5); }
#undef RD_DWR
      hipLaunchKernelGGL(dwconv_wgrad_finalize_kernel, dim3(C * k * k), dim3(64), 0, st, partial, rows, C * k * k, dw, accumulate);
      return;
    }
#define RD_DWV(T, K) hipLaunchKernelGGL((dwconv_wgrad_vec_kernel<T, K>), vgrid, dim3(256), 0, st, (const T*)x, (const T*)dy, partial, N, H, W, C, OH, OW, s, p, v.C4B, v.PL)
    if (dtype == 0) { if (k == 3) RD_DWV(float, 3); else RD_DWV(float, 5); }
    else { if (k == 3) RD_DWV(bf16_t, 3); else RD_DWV(bf16_t, 5); }
#undef RD_DWV
    hipLaunchKernelGGL(dwconv_wgrad_finalize_kernel, dim3(C * k * k), dim3(64), 0, st, partial, rows, C * k * k, dw, accumulate);
    return;
  }
  RG g = rgeom(C);
  dim3 grid(rows, g.nchunk);
#define RD_DW(T, KK) hipLaunchKernelGGL((dwconv_wgrad_kernel<T, KK>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, partial, N, H, W, C, OH, OW, k, s, p, g.CB, g.PL)
  if (dtype == 0) { if (k <= 3) RD_DW(float, 9); else RD_DW(float, 25); }
  else { if (k <= 3) RD_DW(bf16_t, 9); else RD_DW(bf16_t, 25); }
#undef RD_DW
  hipLaunchKernelGGL(dwconv_wgrad_finalize_kernel, dim3(C * k * k), dim3(64), 0, st, partial, rows, C * k * k, dw, accumulate);
}
void launch_bn_stats(const void* y, float* partial, int64_t pixels, int C, int dtype, hipStream_t st) {
  RG g = rgeom(C);
  dim3 grid(dw_rows(pixels, C), g.nchunk);
  if (dtype == 0) hipLaunchKernelGGL((bn_stats_kernel<float>), grid, dim3(256), 0, st, (const float*)y, partial, pixels, C, g.CB, g.PL);
  else hipLaunchKernelGGL((bn_stats_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)y, partial, pixels, C, g.CB, g.PL);
}
// 16-byte-vector forms (C % VE == 0): one thread interpolates / gathers VE channels
template <typename T>
__global__ __launch_bounds__(256) void bilinear_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int OH,
                                                               int OW, int align) {
  constexpr int VE = Elem<T>::VE;
  const int CV = C / VE;
  const int64_t total = (int64_t)N * OH * OW * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int cv = (int)(i % CV); int64_t q = i / CV;
    int ow = (int)(q % OW); q /= OW; int oh = (int)(q % OH); int n = (int)(q / OH);
    int h0, h1, w0, w1; float lh, lw;
    bil_src(oh, H, OH, align, h0, h1, lh);
    bil_src(ow, W, OW, align, w0, w1, lw);
    const T* b = x + (int64_t)n * H * W * C + cv * VE;
    float v00[VE], v01[VE], v10[VE], v11[VE], o[VE];
    ldv(b + ((int64_t)h0 * W + w0) * C, v00); ldv(b + ((int64_t)h0 * W + w1) * C, v01);
    ldv(b + ((int64_t)h1 * W + w0) * C, v10); ldv(b + ((int64_t)h1 * W + w1) * C, v11);
#pragma unroll
    for (int e = 0; e < VE; e++) o[e] = (1.f - lh) * ((1.f - lw) * v00[e] + lw * v01[e]) + lh * ((1.f - lw) * v10[e] + lw * v11[e]);
    stv(y + i * VE, o);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void bilinear_bwd_vec_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int H, int W, int C, int OH,
                                                               int OW, int align) {
  constexpr int VE = Elem<T>::VE;
  const int CV = C / VE;
  const int64_t total = (int64_t)N * H * W * CV;
  const int rh = (OH + H - 1) / H + 2, rw = (OW + W - 1) / W + 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int cv = (int)(i % CV); int64_t q = i / CV;
    int w = (int)(q % W); q /= W; int h = (int)(q % H); int n = (int)(q / H);
    int ohc = (int)(((int64_t)h * OH) / H), owc = (int)(((int64_t)w * OW) / W);
    float g[VE];
#pragma unroll
    for (int e = 0; e < VE; e++) g[e] = 0.f;
    for (int oh = max(ohc - rh, 0); oh <= min(ohc + rh, OH - 1); oh++) {
      int h0, h1; float lh;
      bil_src(oh, H, OH, align, h0, h1, lh);
      float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
      if (wh == 0.f) continue;
      for (int ow = max(owc - rw, 0); ow <= min(owc + rw, OW - 1); ow++) {
        int w0, w1; float lw;
        bil_src(ow, W, OW, align, w0, w1, lw);
        float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
        if (ww == 0.f) continue;
        float v[VE];
        ldv(dy + (((int64_t)n * OH + oh) * OW + ow) * C + cv * VE, v);
#pragma unroll
        for (int e = 0; e < VE; e++) g[e] += wh * ww * v[e];
      }
    }
    stv(dx + i * VE, g);
  }
}
void launch_bilinear(const void* x, void* y, int N, int H, int W, int C, int OH, int OW, int align, int backward, int dtype, hipStream_t st) {
  const int ve = dtype == 0 ? 4 : 8;
  if (C % ve == 0) {
    if (!backward) {
      unsigned g = ew_grid((int64_t)N * OH * OW * (C / ve));
      if (dtype == 0) hipLaunchKernelGGL((bilinear_fwd_vec_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, OH, OW, align);
      else hipLaunchKernelGGL((bilinear_fwd_vec_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW, align);
    } else {
      unsigned g = ew_grid((int64_t)N * H * W * (C / ve));
      if (dtype == 0) hipLaunchKernelGGL((bilinear_bwd_vec_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, OH, OW, align);
      else hipLaunchKernelGGL((bilinear_bwd_vec_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW, align);
    }
    return;
  }
  if (!backward) {
    unsigned g = ew_grid((int64_t)N * OH * OW * C);
    if (dtype == 0) hipLaunchKernelGGL((bilinear_fwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, OH, OW, align);
    else hipLaunchKernelGGL((bilinear_fwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW, align);
  } else {  // x = dy (N,OH,OW,C), y = dx (N,H,W,C)
    unsigned g = ew_grid((int64_t)N * H * W * C);
    if (dtype == 0) hipLaunchKernelGGL((bilinear_bwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, OH, OW, align);
    else hipLaunchKernelGGL((bilinear_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, H, W, C, OH, OW, align);
  }
}
void launch_sml_head_fwd(const void* out, const float* d, float* pred, int64_t n, float hi, float lo, int dtype, hipStream_t st) {
  if (dtype == 0) hipLaunchKernelGGL((sml_head_fwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)out, d, pred, n, hi, lo);
  else hipLaunchKernelGGL((sml_head_fwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)out, d, pred, n, hi, lo);
}
void launch_sml_head_bwd(const void* out, const float* d, const float* dpred, void* dout, int64_t n, float hi, float lo, int dtype,
                         hipStream_t st) {
  if (dtype == 0) hipLaunchKernelGGL((sml_head_bwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)out, d, dpred, (float*)dout, n, hi, lo);
  else hipLaunchKernelGGL((sml_head_bwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)out, d, dpred, (bf16_t*)dout, n, hi, lo);
}
void launch_reciprocal(const float* x, const float* dy, float* out, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(reciprocal_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, dy, out, n);
}

}  // namespace rd
