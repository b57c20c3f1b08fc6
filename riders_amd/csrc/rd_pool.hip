// Max-pool and ROI max-pool (sparse-radar gather) kernels.  HBM-bound, NHWC, coalesced over channels.
//
// Reference call sites:
//   RCNet/networks.py:73-76,245   torch.nn.MaxPool2d(3, stride 2, padding 1)
//   RCNet/networks.py:418-433     torchvision.ops.roi_pool(latent / skips, b_boxes, spatial_scale, output_size)
// torchvision (0.14.0, environment.yaml:9) is not part of the reference tree; the ROI arithmetic below
// restates its published CPU kernel (roi_pool_kernel.cpp): C round() of the scaled box corners,
// +1 widths, floor/ceil bin edges clamped to the map, empty bin -> 0 / argmax -1, strict '>' scan
// in row-major order (first maximum wins).  Index results are bit-exact by construction.
#include "rd_common.h"
#include "rd_kernels.h"
#include <float.h>

namespace rd {

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                          unsigned char* __restrict__ arg, int N, int H, int W, int C,
                                                          int OH, int OW, int k, int s, int p) {
  const int64_t total = (int64_t)N * OH * OW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int ow = (int)(q % OW); q /= OW; int oh = (int)(q % OH); int n = (int)(q / OH);
    float best = -INFINITY; int bi = 0; bool first = true;
    for (int kh = 0; kh < k; kh++) {
      int ih = oh * s - p + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < k; kw++) {
        int iw = ow * s - p + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        float v = Elem<T>::ld(x + (((int64_t)n * H + ih) * W + iw) * C + c);
        if (first || v > best || v != v) { best = v; bi = kh * k + kw; first = false; }
      }
    }
    Elem<T>::st(out + i, best);
    arg[i] = (unsigned char)bi;
  }
}

// deterministic gather: every input pixel sums the outputs whose arg-max points at it
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int OH, int OW,
                                                          int k, int s, int p) {
  const int64_t total = (int64_t)N * H * W * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int w = (int)(q % W); q /= W; int h = (int)(q % H); int n = (int)(q / H);
    float g = 0.f;
    int oh_lo = (h + p - k + 1 + s - 1); oh_lo = oh_lo < 0 ? 0 : oh_lo / s;
    int oh_hi = (h + p) / s; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_lo = (w + p - k + 1 + s - 1); ow_lo = ow_lo < 0 ? 0 : ow_lo / s;
    int ow_hi = (w + p) / s; if (ow_hi > OW - 1) ow_hi = OW - 1;
    for (int oh = oh_lo; oh <= oh_hi; oh++)
      for (int ow = ow_lo; ow <= ow_hi; ow++) {
        int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + c;
        int a = arg[o];
        int kh = a / k, kw = a - kh * k;
        if (oh * s - p + kh == h && ow * s - p + kw == w) g += Elem<T>::ld(dout + o);
      }
    Elem<T>::st(dx + i, g);
  }
}

// rois: [R][5] = (batch index, x1, y1, x2, y2) in input-image pixels
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ rois,
                                                           T* __restrict__ out, int* __restrict__ argmax, int R, int N,
                                                           int H, int W, int C, int PH, int PW, float scale) {
  const int64_t total = (int64_t)R * PH * PW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int c = (int)(i % C); int64_t q = i / C;
    int pw = (int)(q % PW); q /= PW; int ph = (int)(q % PH); int r = (int)(q / PH);
    const float* roi = rois + (int64_t)r * 5;
    int b = (int)roi[0];
    int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
    int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
    int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
    float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    int hs = (int)floorf((float)ph * bh), he = (int)ceilf((float)(ph + 1) * bh);
    int ws = (int)floorf((float)pw * bw), we = (int)ceilf((float)(pw + 1) * bw);
    hs = min(max(hs + sh, 0), H); he = min(max(he + sh, 0), H);
    ws = min(max(ws + sw, 0), W); we = min(max(we + sw, 0), W);
    bool empty = (he <= hs) || (we <= ws);
    float best = empty ? 0.f : -FLT_MAX; int bi = -1;
    if (b >= 0 && b < N) {
      const T* xb = x + (int64_t)b * H * W * C + c;
      for (int h = hs; h < he; h++)
        for (int w = ws; w < we; w++) {
          float v = Elem<T>::ld(xb + ((int64_t)h * W + w) * C);
          if (v > best) { best = v; bi = h * W + w; }
        }
    }
    Elem<T>::st(out + i, best);
    argmax[i] = bi;
  }
}

// scatter-add through the saved arg-max into an fp32 accumulator (zeroed by the launcher)
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(const T* __restrict__ dout, const float* __restrict__ rois,
                                                           const int* __restrict__ argmax, float* __restrict__ dx, int R,
                                                           int N, int H, int W, int C, int PH, int PW) {
  const int64_t total = (int64_t)R * PH * PW * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int a = argmax[i];
    if (a < 0) continue;
    int c = (int)(i % C); int r = (int)(i / ((int64_t)C * PH * PW));
    int b = (int)rois[(int64_t)r * 5];
    if (b < 0 || b >= N) continue;
    atomicAdd(dx + ((int64_t)b * H * W + a) * C + c, Elem<T>::ld(dout + i));
  }
}

// explicit zero-fill kernel (kept as a kernel node so hipGraph replays order it like every other launch)
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

static unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 4096)); }

void launch_maxpool_fwd(const void* x, void* out, unsigned char* arg, int N, int H, int W, int C, int OH, int OW, int k, int s,
                        int p, int dtype, hipStream_t st) {
  int64_t n = (int64_t)N * OH * OW * C;
  if (dtype == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)x, (float*)out, arg, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, arg, N, H, W, C, OH, OW, k, s, p);
}
void launch_maxpool_bwd(const void* dout, const unsigned char* arg, void* dx, int N, int H, int W, int C, int OH, int OW, int k,
                        int s, int p, int dtype, hipStream_t st) {
  int64_t n = (int64_t)N * H * W * C;
  if (dtype == 0) hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)dout, arg, (float*)dx, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)dout, arg, (bf16_t*)dx, N, H, W, C, OH, OW, k, s, p);
}
void launch_roi_pool_fwd(const void* x, const float* rois, void* out, int* argmax, int R, int N, int H, int W, int C, int PH,
                         int PW, float scale, int dtype, hipStream_t st) {
  int64_t n = (int64_t)R * PH * PW * C;
  if (n == 0) return;
  if (dtype == 0) hipLaunchKernelGGL((roi_pool_fwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)x, rois, (float*)out, argmax, R, N, H, W, C, PH, PW, scale);
  else hipLaunchKernelGGL((roi_pool_fwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)x, rois, (bf16_t*)out, argmax, R, N, H, W, C, PH, PW, scale);
}
void launch_roi_pool_bwd(const void* dout, const float* rois, const int* argmax, float* dx_f32, int R, int N, int H, int W, int C,
                         int PH, int PW, int dtype, hipStream_t st) {
  int64_t nz = (int64_t)N * H * W * C;
  hipLaunchKernelGGL(zero_f32_kernel, dim3(ew_grid(nz / 4 + 1)), dim3(256), 0, st, dx_f32, nz);
  int64_t n = (int64_t)R * PH * PW * C;
  if (n == 0) return;
  if (dtype == 0) hipLaunchKernelGGL((roi_pool_bwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)dout, rois, argmax, dx_f32, R, N, H, W, C, PH, PW);
  else hipLaunchKernelGGL((roi_pool_bwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)dout, rois, argmax, dx_f32, R, N, H, W, C, PH, PW);
}

}  // namespace rd
