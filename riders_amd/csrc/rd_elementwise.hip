// Layout, cast and resampling helpers (all HBM-bound, grid-stride, coalesced over the channel axis).
//
// Reference call sites:
//   RCNet/rcnet_transforms.py:258-261   images / 255.0          -> rd_cast(scale = 1/255) fused with NCHW->NHWC
//   utils/net_utils.py:196              F.interpolate(x, size)  -> nearest (forward folded into the conv gather;
//                                       the standalone forward/backward live here)
//   RCNet/networks.py:441-450           view/permute of tokens, cat([image_tf, depth_tf], dim=1)
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

static unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 4096)); }

template <typename S, typename D>
__global__ __launch_bounds__(256) void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, int64_t n, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Elem<D>::st(dst + i, Elem<S>::ld(src + i) * scale);
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    Elem<T>::st(out + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
}
// 16-byte vectors for the aligned body, scalars for the tail
template <typename T>
__global__ __launch_bounds__(256) void add_vec_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, int64_t n) {
  constexpr int VE = Elem<T>::VE;
  const int64_t nv = n / VE, stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = i0; i < nv; i += stride) {
    float x[VE], y[VE];
    ldv(a + i * VE, x); ldv(b + i * VE, y);
#pragma unroll
    for (int e = 0; e < VE; e++) x[e] += y[e];
    stv(out + i * VE, x);
  }
  for (int64_t i = nv * VE + i0; i < n; i += stride) Elem<T>::st(out + i, Elem<T>::ld(a + i) + Elem<T>::ld(b + i));
}

// [N][C][H][W] -> [N][H][W][C] (and back); small C on the hot path (3-channel thermal image), so a plain
// gather with coalesced writes is enough.
template <typename S, typename D>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const S* __restrict__ src, D* __restrict__ dst, int N, int C, int H,
                                                           int W, float scale) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int w = (int)(q % W); q /= W; int h = (int)(q % H); int n = (int)(q / H);
    Elem<D>::st(dst + i, Elem<S>::ld(src + (((int64_t)n * C + c) * H + h) * W + w) * scale);
  }
}
template <typename S, typename D>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const S* __restrict__ src, D* __restrict__ dst, int N, int C, int H,
                                                           int W) {
  const int64_t total = (int64_t)N * C * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int w = (int)(i % W); int64_t q = i / W;
    int h = (int)(q % H); q /= H; int c = (int)(q % C); int n = (int)(q / C);
    Elem<D>::st(dst + i, Elem<S>::ld(src + (((int64_t)n * H + h) * W + w) * C + c));
  }
}

// [B][R][Cc] -> [B][Cc][R]
template <typename T>
__global__ __launch_bounds__(256) void transpose_last2_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t B, int R, int Cc) {
  const int64_t total = B * R * Cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int r = (int)(i % R); int64_t q = i / R; int c = (int)(q % Cc); int64_t b = q / Cc;
    dst[i] = src[(b * R + r) * Cc + c];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void concat2_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                                                      int64_t rows, int Ca, int Cb) {
  const int Cc = Ca + Cb; const int64_t total = rows * Cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % Cc); int64_t r = i / Cc;
    out[i] = c < Ca ? a[r * Ca + c] : b[r * Cb + (c - Ca)];
  }
}
template <typename T>
__global__ __launch_bounds__(256) void split2_kernel(const T* __restrict__ in, T* __restrict__ a, T* __restrict__ b, int64_t rows,
                                                     int Ca, int Cb) {
  const int Cc = Ca + Cb; const int64_t total = rows * Cc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % Cc); int64_t r = i / Cc;
    if (c < Ca) a[r * Ca + c] = in[i]; else b[r * Cb + (c - Ca)] = in[i];
  }
}

__device__ __forceinline__ int nearest_src(int d, float scale, int in) { return min((int)floorf((float)d * scale), in - 1); }

template <typename T>
__global__ __launch_bounds__(256) void upsample_nearest_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int Hs, int Ws,
                                                                   int Hv, int Wv, int C, float sh, float sw) {
  const int64_t total = (int64_t)N * Hv * Wv * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int w = (int)(q % Wv); q /= Wv; int h = (int)(q % Hv); int n = (int)(q / Hv);
    y[i] = x[(((int64_t)n * Hs + nearest_src(h, sh, Hs)) * Ws + nearest_src(w, sw, Ws)) * C + c];
  }
}
// dx[source pixel] = sum of dy over the replicated destination pixels (deterministic gather)
template <typename T>
__global__ __launch_bounds__(256) void upsample_nearest_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int Hs, int Ws,
                                                                   int Hv, int Wv, int C, float sh, float sw) {
  const int64_t total = (int64_t)N * Hs * Ws * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int w = (int)(q % Ws); q /= Ws; int h = (int)(q % Hs); int n = (int)(q / Hs);
    int h0 = (int)(((int64_t)h * Hv) / Hs) - 1, h1 = (int)(((int64_t)(h + 1) * Hv + Hs - 1) / Hs) + 1;
    int w0 = (int)(((int64_t)w * Wv) / Ws) - 1, w1 = (int)(((int64_t)(w + 1) * Wv + Ws - 1) / Ws) + 1;
    h0 = max(h0, 0); w0 = max(w0, 0); h1 = min(h1, Hv - 1); w1 = min(w1, Wv - 1);
    float g = 0.f;
    for (int hv = h0; hv <= h1; hv++) {
      if (nearest_src(hv, sh, Hs) != h) continue;
      for (int wv = w0; wv <= w1; wv++) {
        if (nearest_src(wv, sw, Ws) != w) continue;
        g += Elem<T>::ld(dy + (((int64_t)n * Hv + hv) * Wv + wv) * C + c);
      }
    }
    Elem<T>::st(dx + i, g);
  }
}

// 16-byte-vector form (C % VE == 0): one thread sums VE channels of a source pixel
template <typename T>
__global__ __launch_bounds__(256) void upsample_nearest_bwd_vec_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int Hs, int Ws,
                                                                       int Hv, int Wv, int C, float sh, float sw) {
  constexpr int VE = Elem<T>::VE;
  const int CV = C / VE;
  const int64_t total = (int64_t)N * Hs * Ws * CV;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int cv = (int)(i % CV); int64_t q = i / CV;
    int w = (int)(q % Ws); q /= Ws; int h = (int)(q % Hs); int n = (int)(q / Hs);
    int h0 = (int)(((int64_t)h * Hv) / Hs) - 1, h1 = (int)(((int64_t)(h + 1) * Hv + Hs - 1) / Hs) + 1;
    int w0 = (int)(((int64_t)w * Wv) / Ws) - 1, w1 = (int)(((int64_t)(w + 1) * Wv + Ws - 1) / Ws) + 1;
    h0 = max(h0, 0); w0 = max(w0, 0); h1 = min(h1, Hv - 1); w1 = min(w1, Wv - 1);
    float g[VE];
#pragma unroll
    for (int e = 0; e < VE; e++) g[e] = 0.f;
    // the destination rows / columns that read this source pixel are contiguous (nearest_src is monotonic): first one and count, ALU only
    int hf = 0, nh = 0, wf = 0, nw = 0;
    for (int hv = h0; hv <= h1; hv++)
      if (nearest_src(hv, sh, Hs) == h) { if (nh == 0) hf = hv; nh++; }
    for (int wv = w0; wv <= w1; wv++)
      if (nearest_src(wv, sw, Ws) == w) { if (nw == 0) wf = wv; nw++; }
    if (nh <= 3 && nw <= 3) {
      // up-sampling by ~2 (RC-Net: 7x3 -> 15x6 ... 60x25 -> 120x50): at most 3 x 3 readers.  All nine are requested together,
      // unconditionally (clamped address, masked sum) -- a load inside the candidate loop is one memory round trip per reader
      uint4 raw[9];
#pragma unroll
      for (int q = 0; q < 9; q++) {
        const int a = q / 3, bq = q - a * 3;
        const int hv = min(hf + (a < nh ? a : 0), Hv - 1), wv = min(wf + (bq < nw ? bq : 0), Wv - 1);
        raw[q] = *reinterpret_cast<const uint4*>(dy + ((((int64_t)n * Hv + hv) * Wv + wv) * CV + cv) * VE);
      }
#pragma unroll
      for (int q = 0; q < 9; q++) {
        const int a = q / 3, bq = q - a * 3;
        float v[VE];
        raw16_to_f32(reinterpret_cast<const T*>(0), raw[q], v);
        const bool ok = a < nh && bq < nw;
#pragma unroll
        for (int e = 0; e < VE; e++) g[e] += ok ? v[e] : 0.f;
      }
    } else {
      for (int hv = hf; hv < hf + nh; hv++)
        for (int wv = wf; wv < wf + nw; wv++) {
          float v[VE];
          ldv(dy + ((((int64_t)n * Hv + hv) * Wv + wv) * CV + cv) * VE, v);
#pragma unroll
          for (int e = 0; e < VE; e++) g[e] += v[e];
        }
    }
    stv(dx + i * VE, g);
  }
}

// [rows][C] -> [rows][Cpad] with zero fill (3-channel images onto the 16-byte-vector convolution paths)
template <typename T>
__global__ __launch_bounds__(256) void pad_channels_kernel(const T* __restrict__ src, T* __restrict__ dst, int64_t rows, int C, int Cpad) {
  const int64_t total = rows * Cpad;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cpad); const int64_t r = i / Cpad;
    if (c < C) dst[i] = src[r * C + c]; else Elem<T>::st(dst + i, 0.f);
  }
}
// dw[co][ci][tap] (+)= dwp[co][ci][tap] for ci < Cin (dwp has CinPad input channels)
__global__ __launch_bounds__(256) void unpad_weight_grad_kernel(const float* __restrict__ dwp, float* __restrict__ dw, int Cout, int Cin,
                                                                int CinPad, int taps, int accumulate) {
  const int total = Cout * Cin * taps;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int tp = i % taps, ci = (i / taps) % Cin, co = i / (taps * Cin);
    const float v = dwp[((int64_t)co * CinPad + ci) * taps + tp];
    dw[i] = accumulate ? dw[i] + v : v;
  }
}
void launch_pad_channels(const void* src, void* dst, int64_t rows, int C, int Cpad, int dtype, hipStream_t st) {
  unsigned g = ew_grid(rows * Cpad);
  if (dtype == 0) hipLaunchKernelGGL((pad_channels_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, rows, C, Cpad);
  else hipLaunchKernelGGL((pad_channels_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, rows, C, Cpad);
}
void launch_unpad_weight_grad(const float* dwp, float* dw, int Cout, int Cin, int CinPad, int taps, int accumulate, hipStream_t st) {
  hipLaunchKernelGGL(unpad_weight_grad_kernel, dim3((unsigned)cdiv(Cout * Cin * taps, 256)), dim3(256), 0, st, dwp, dw, Cout, Cin, CinPad, taps, accumulate);
}

void launch_cast(const void* src, void* dst, int64_t n, int sd, int dd, float scale, hipStream_t st) {
  unsigned g = ew_grid(n);
  if (sd == 0 && dd == 0) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, n, scale);
  else if (sd == 0 && dd == 1) hipLaunchKernelGGL((cast_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, n, scale);
  else if (sd == 1 && dd == 0) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, n, scale);
  else hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, n, scale);
}
void launch_add(const void* a, const void* b, void* out, int64_t n, int dtype, hipStream_t st) {
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0) {
    unsigned g = ew_grid(n / (dtype == 0 ? 4 : 8) + 1);
    if (dtype == 0) hipLaunchKernelGGL((add_vec_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n);
    else hipLaunchKernelGGL((add_vec_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
    return;
  }
  if (dtype == 0) hipLaunchKernelGGL((add_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n);
  else hipLaunchKernelGGL((add_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, n);
}
void launch_nchw_to_nhwc(const void* src, void* dst, int N, int C, int H, int W, int sd, int dd, float scale, hipStream_t st) {
  unsigned g = ew_grid((int64_t)N * C * H * W);
  if (sd == 0 && dd == 0) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, N, C, H, W, scale);
  else if (sd == 0 && dd == 1) hipLaunchKernelGGL((nchw_to_nhwc_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, N, C, H, W, scale);
  else if (sd == 1 && dd == 0) hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, N, C, H, W, scale);
  else hipLaunchKernelGGL((nchw_to_nhwc_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, N, C, H, W, scale);
}
void launch_nhwc_to_nchw(const void* src, void* dst, int N, int C, int H, int W, int sd, int dd, hipStream_t st) {
  unsigned g = ew_grid((int64_t)N * C * H * W);
  if (sd == 0 && dd == 0) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, N, C, H, W);
  else if (sd == 0 && dd == 1) hipLaunchKernelGGL((nhwc_to_nchw_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)src, (bf16_t*)dst, N, C, H, W);
  else if (sd == 1 && dd == 0) hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (float*)dst, N, C, H, W);
  else hipLaunchKernelGGL((nhwc_to_nchw_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, N, C, H, W);
}
void launch_transpose_last2(const void* src, void* dst, int64_t B, int R, int Cc, int dtype, hipStream_t st) {
  unsigned g = ew_grid(B * R * Cc);
  if (dtype == 0) hipLaunchKernelGGL((transpose_last2_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)src, (float*)dst, B, R, Cc);
  else hipLaunchKernelGGL((transpose_last2_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, R, Cc);
}
void launch_concat2(const void* a, const void* b, void* out, int64_t rows, int Ca, int Cb, int dtype, hipStream_t st) {
  unsigned g = ew_grid(rows * (Ca + Cb));
  if (dtype == 0) hipLaunchKernelGGL((concat2_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, rows, Ca, Cb);
  else hipLaunchKernelGGL((concat2_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, rows, Ca, Cb);
}
void launch_split2(const void* in, void* a, void* b, int64_t rows, int Ca, int Cb, int dtype, hipStream_t st) {
  unsigned g = ew_grid(rows * (Ca + Cb));
  if (dtype == 0) hipLaunchKernelGGL((split2_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)in, (float*)a, (float*)b, rows, Ca, Cb);
  else hipLaunchKernelGGL((split2_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)a, (bf16_t*)b, rows, Ca, Cb);
}
void launch_upsample_nearest_fwd(const void* x, void* y, int N, int Hs, int Ws, int Hv, int Wv, int C, int dtype, hipStream_t st) {
  float sh = (float)Hs / (float)Hv, sw = (float)Ws / (float)Wv;
  unsigned g = ew_grid((int64_t)N * Hv * Wv * C);
  if (dtype == 0) hipLaunchKernelGGL((upsample_nearest_fwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)x, (float*)y, N, Hs, Ws, Hv, Wv, C, sh, sw);
  else hipLaunchKernelGGL((upsample_nearest_fwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, N, Hs, Ws, Hv, Wv, C, sh, sw);
}
void launch_upsample_nearest_bwd(const void* dy, void* dx, int N, int Hs, int Ws, int Hv, int Wv, int C, int dtype, hipStream_t st) {
  float sh = (float)Hs / (float)Hv, sw = (float)Ws / (float)Wv;
  const int ve = dtype == 0 ? 4 : 8;
  if (C % ve == 0) {
    unsigned gv = ew_grid((int64_t)N * Hs * Ws * (C / ve));
    if (dtype == 0) hipLaunchKernelGGL((upsample_nearest_bwd_vec_kernel<float>), dim3(gv), dim3(256), 0, st, (const float*)dy, (float*)dx, N, Hs, Ws, Hv, Wv, C, sh, sw);
    else hipLaunchKernelGGL((upsample_nearest_bwd_vec_kernel<bf16_t>), dim3(gv), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, Hs, Ws, Hv, Wv, C, sh, sw);
    return;
  }
  unsigned g = ew_grid((int64_t)N * Hs * Ws * C);
  if (dtype == 0) hipLaunchKernelGGL((upsample_nearest_bwd_kernel<float>), dim3(g), dim3(256), 0, st, (const float*)dy, (float*)dx, N, Hs, Ws, Hv, Wv, C, sh, sw);
  else hipLaunchKernelGGL((upsample_nearest_bwd_kernel<bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)dy, (bf16_t*)dx, N, Hs, Ws, Hv, Wv, C, sh, sw);
}

}  // namespace rd
