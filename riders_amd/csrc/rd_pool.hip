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
#include <cstdlib>

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

// 16-byte-vector form of the forward (C % VE == 0): a thread owns VE channels of one output pixel; per element the same rule as above (the
// first tap, then strictly greater or NaN).  The element-per-thread form ran RC-Net's 248 x 306 x 32 pool at 46 us for a 49 MB pass.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_vec_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                              unsigned char* __restrict__ arg, int N, int H, int W, int C,
                                                              int OH, int OW, int k, int s, int p) {
  constexpr int VE = Elem<T>::VE;
  const int G = C / VE;
  const int64_t total = (int64_t)N * OH * OW * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % G); int64_t q = i / G;
    const int ow = (int)(q % OW); q /= OW; const int oh = (int)(q % OH); const int n = (int)(q / OH);
    float best[VE]; int bi[VE]; bool first = true;
#pragma unroll
    for (int e = 0; e < VE; e++) { best[e] = -INFINITY; bi[e] = 0; }
    for (int kh = 0; kh < k; kh++) {
      const int ih = oh * s - p + kh;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int kw = 0; kw < k; kw++) {
        const int iw = ow * s - p + kw;
        if ((unsigned)iw >= (unsigned)W) continue;
        float v[VE];
        ldv(x + (((int64_t)n * H + ih) * W + iw) * C + g * VE, v);
#pragma unroll
        for (int e = 0; e < VE; e++)
          if (first || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * k + kw; }
        first = false;
      }
    }
    stv(out + i * VE, best);
    unsigned char* ap = arg + i * VE;
#pragma unroll
    for (int e = 0; e < VE; e += 4)
      *reinterpret_cast<unsigned*>(ap + e) = (unsigned)bi[e] | ((unsigned)bi[e + 1] << 8) | ((unsigned)bi[e + 2] << 16) | ((unsigned)bi[e + 3] << 24);
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

// 16-byte-vector form (C % VE == 0)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_vec_kernel(const T* __restrict__ dout, const unsigned char* __restrict__ arg,
                                                              T* __restrict__ dx, int N, int H, int W, int C, int OH, int OW,
                                                              int k, int s, int p) {
  constexpr int VE = Elem<T>::VE;
  const int G = C / VE;
  const int64_t total = (int64_t)N * H * W * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int g = (int)(i % G); int64_t q = i / G;
    int w = (int)(q % W); q /= W; int h = (int)(q % H); int n = (int)(q / H);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; e++) acc[e] = 0.f;
    int oh_lo = (h + p - k + 1 + s - 1); oh_lo = oh_lo < 0 ? 0 : oh_lo / s;
    int oh_hi = (h + p) / s; if (oh_hi > OH - 1) oh_hi = OH - 1;
    int ow_lo = (w + p - k + 1 + s - 1); ow_lo = ow_lo < 0 ? 0 : ow_lo / s;
    int ow_hi = (w + p) / s; if (ow_hi > OW - 1) ow_hi = OW - 1;
    for (int oh = oh_lo; oh <= oh_hi; oh++)
      for (int ow = ow_lo; ow <= ow_hi; ow++) {
        const int64_t o = (((int64_t)n * OH + oh) * OW + ow) * C + g * VE;
        // the window tap that maps this output back to (h, w); an output contributes where its saved arg-max equals it
        const int want = (h - (oh * s - p)) * k + (w - (ow * s - p));
        float dv[VE];
        ldv(dout + o, dv);
        const unsigned char* ap = arg + o;
#pragma unroll
        for (int e = 0; e < VE; e++) if ((int)ap[e] == want) acc[e] += dv[e];
      }
    stv(dx + i * VE, acc);
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


// Compact arg-max (round 4): one BYTE per pooled element instead of the int32 pixel index -- the arg-max as an offset inside its bin window,
// (h - hs) * 16 + (w - ws), 0xFF for an empty bin.  RC-Net's bins are ~1.02 pixels (windows of 2 x 2), the int32 was 4 of the 6 bytes the
// forward writes per element and 4 of the 6 the backward reads (83.6 M elements per step).  Windows wider or taller than 15 pixels cannot
// be encoded: the forward then raises *flag (sticky) and the backward kernels turn the whole gradient into NaN -- loud, not silent.
static constexpr int ROI_U8_MAXWIN = 15;
__device__ __forceinline__ void roi_store_arg(int* p, const int (&bi)[4]) { *reinterpret_cast<int4*>(p) = make_int4(bi[0], bi[1], bi[2], bi[3]); }
__device__ __forceinline__ void roi_store_arg(int* p, const int (&bi)[8]) {
  *reinterpret_cast<int4*>(p) = make_int4(bi[0], bi[1], bi[2], bi[3]);
  *reinterpret_cast<int4*>(p + 4) = make_int4(bi[4], bi[5], bi[6], bi[7]);
}
__device__ __forceinline__ void roi_store_arg(unsigned char* p, const int (&bi)[4]) {
  *reinterpret_cast<unsigned*>(p) = (unsigned)(bi[0] & 255) | ((unsigned)(bi[1] & 255) << 8) | ((unsigned)(bi[2] & 255) << 16) | ((unsigned)(bi[3] & 255) << 24);
}
__device__ __forceinline__ void roi_store_arg(unsigned char* p, const int (&bi)[8]) {
  uint2 v;
  v.x = (unsigned)(bi[0] & 255) | ((unsigned)(bi[1] & 255) << 8) | ((unsigned)(bi[2] & 255) << 16) | ((unsigned)(bi[3] & 255) << 24);
  v.y = (unsigned)(bi[4] & 255) | ((unsigned)(bi[5] & 255) << 8) | ((unsigned)(bi[6] & 255) << 16) | ((unsigned)(bi[7] & 255) << 24);
  *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void roi_load_arg(const int* p, int (&am)[4]) { const int4 a = *reinterpret_cast<const int4*>(p); am[0] = a.x; am[1] = a.y; am[2] = a.z; am[3] = a.w; }
__device__ __forceinline__ void roi_load_arg(const int* p, int (&am)[8]) {
  const int4 a = *reinterpret_cast<const int4*>(p), b = *reinterpret_cast<const int4*>(p + 4);
  am[0] = a.x; am[1] = a.y; am[2] = a.z; am[3] = a.w; am[4] = b.x; am[5] = b.y; am[6] = b.z; am[7] = b.w;
}
__device__ __forceinline__ void roi_load_arg(const unsigned char* p, int (&am)[4]) {
  const unsigned a = *reinterpret_cast<const unsigned*>(p);
  am[0] = a & 255; am[1] = (a >> 8) & 255; am[2] = (a >> 16) & 255; am[3] = a >> 24;
}
__device__ __forceinline__ void roi_load_arg(const unsigned char* p, int (&am)[8]) {
  const uint2 a = *reinterpret_cast<const uint2*>(p);
  am[0] = a.x & 255; am[1] = (a.x >> 8) & 255; am[2] = (a.x >> 16) & 255; am[3] = a.x >> 24;
  am[4] = a.y & 255; am[5] = (a.y >> 8) & 255; am[6] = (a.y >> 16) & 255; am[7] = a.y >> 24;
}

// 16-byte-vector form (C % VE == 0): one thread pools VE channels of a bin.  A = int (pixel index h * W + w, -1 = empty bin) or unsigned
// char (compact code, above)
template <typename T, typename A>
__global__ __launch_bounds__(256) void roi_pool_fwd_vec_kernel(const T* __restrict__ x, const float* __restrict__ rois,
                                                               T* __restrict__ out, A* __restrict__ argmax, int R, int N,
                                                               int H, int W, int C, int PH, int PW, float scale, int* __restrict__ flag) {
  constexpr bool U8 = sizeof(A) == 1;
  constexpr int VE = Elem<T>::VE;
  const int G = C / VE;
  const int64_t total = (int64_t)R * PH * PW * G;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int g = (int)(i % G); int64_t q = i / G;
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
    float best[VE]; int bi[VE];
#pragma unroll
    for (int e = 0; e < VE; e++) { best[e] = empty ? 0.f : -FLT_MAX; bi[e] = -1; }
    if (U8 && (he - hs > ROI_U8_MAXWIN || we - ws > ROI_U8_MAXWIN)) *flag = 1;      // cannot be encoded: the backward poisons the gradient
    const bool bok = b >= 0 && b < N;
    const T* xb = x + (int64_t)(bok ? b : 0) * H * W * C + g * VE;
    if (he - hs <= 2 && we - ws <= 2) {
      // windows of at most 2 x 2 pixels (every RC-Net pooling: bins of ~1.02 pixels): the four candidates are requested together,
      // unconditionally (clamped address, masked afterwards) -- the window loop below is one memory round trip per pixel -- and
      // compared in the loop's order (row-major, strict >), so ties resolve identically
      uint4 raw[4]; bool ok[4]; int idx[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int h = hs + (q >> 1), w = ws + (q & 1);
        ok[q] = bok && h < he && w < we;
        const int hc = min(max(h, 0), H - 1), wc = min(max(w, 0), W - 1);
        idx[q] = U8 ? (q >> 1) * 16 + (q & 1) : h * W + w;
        raw[q] = *reinterpret_cast<const uint4*>(xb + ((int64_t)hc * W + wc) * C);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float v[VE];
        raw16_to_f32(reinterpret_cast<const T*>(0), raw[q], v);
#pragma unroll
        for (int e = 0; e < VE; e++) if (ok[q] && v[e] > best[e]) { best[e] = v[e]; bi[e] = idx[q]; }
      }
    } else if (bok) {
      for (int h = hs; h < he; h++)
        for (int w = ws; w < we; w++) {
          float v[VE];
          ldv(xb + ((int64_t)h * W + w) * C, v);
#pragma unroll
          for (int e = 0; e < VE; e++) if (v[e] > best[e]) { best[e] = v[e]; bi[e] = U8 ? (((h - hs) & 15) << 4) | ((w - ws) & 15) : h * W + w; }
        }
    }
    stv(out + i * VE, best);
    roi_store_arg(argmax + i * VE, bi);      // (compact form: -1 & 255 = 0xFF = empty)
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


// the same for the compact arg-max: the window start of the element's bin is recomputed with the forward's expressions
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_bwd_u8_kernel(const T* __restrict__ dout, const float* __restrict__ rois,
                                                              const unsigned char* __restrict__ argmax, const int* __restrict__ flag,
                                                              float* __restrict__ dx, int R, int N, int H, int W, int C, int PH, int PW, float scale) {
  const int64_t total = (int64_t)R * PH * PW * C;
  const bool poison = *flag != 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int a = argmax[i];
    if (a == 255 && !poison) continue;
    int c = (int)(i % C); int64_t q = i / C;
    int pw = (int)(q % PW); q /= PW; int ph = (int)(q % PH); int r = (int)(q / PH);
    const float* roi = rois + (int64_t)r * 5;
    const int b = (int)roi[0];
    if (b < 0 || b >= N) continue;
    const int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
    const int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
    const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
    const float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    const int hs = min(max((int)floorf((float)ph * bh) + sh, 0), H), ws = min(max((int)floorf((float)pw * bw) + sw, 0), W);
    if (poison) { atomicAdd(dx + ((int64_t)b * H * W + min(hs, H - 1) * W + min(ws, W - 1)) * C + c, __builtin_nanf("")); continue; }
    atomicAdd(dx + ((int64_t)b * H * W + (hs + (a >> 4)) * W + ws + (a & 15)) * C + c, Elem<T>::ld(dout + i));
  }
}

// ---- deterministic (gather) forms of the RoI-pool backward: a block owns a 16 x 16 pixel tile of one image, lists the RoIs of that image whose
// (scaled) box meets the tile in ascending order (ordered wave-ballot compaction into LDS), and every pixel adds dout of the bins whose window
// contains it and whose saved arg-max names it -- no atomics, fixed order.  (The first form of this, one window loop per pixel and RoI, was
// replaced by the pixel-owner kernel below in round 2 and removed in round 4.)
static constexpr int RPB_T = 16, RPB_MAXL = 64;
struct RoiGeo { int r, sh, sw, eh, ew; float bh, bw; };   // eh / ew: one past the last row / column any bin window can reach

// ---- pixel-owner form of the RoI-pool backward ------------------------------------------------------------------------------------
// As the gather kernel above: a block owns a 16 x 16 pixel tile (x 32 channels) of one image, lists the RoIs of that image whose box
// meets the tile, and every (pixel, VE-channel group) item adds dout of the bins whose window contains the pixel and whose saved
// arg-max names it -- registers only, no LDS accumulators, no atomics, fixed order (RoI, bin row, bin column).  What differs is where
// the window arithmetic happens: for each listed RoI, 32 threads work out ONCE, with the forward's float expressions, which bins
// cover each of the tile's 16 rows and 16 columns (first bin and count, windows of consecutive bins being contiguous) and leave that
// in LDS; an item then only looks up its row and its column (one bin each in RC-Net, where the RoI has the size of its output), issues the
// arg-max / dout requests of all its pixels together and compares.  ~45 VALU instructions per (pixel group, RoI) instead of ~400 in
// the tile form (per-channel arg-max decode + LDS read-add-write) and ~150 in the first gather form (window loops per pixel).
static constexpr int RPT_CC = 32;      // channels per block of the pixel-owner and tile forms
struct RoiCand { int r; int rowc[RPB_T]; int colc[RPB_T]; };    // (first bin << 12) | count per tile row / column; 0 = none
// compact arg-max: offset of the tile row / column inside the window of its first and of its second covering bin (d0 | d1 << 8)
struct RoiCandOff { int rowd[RPB_T]; int cold[RPB_T]; };
template <typename T, typename A>
__global__ __launch_bounds__(256) void roi_pool_bwd_pix_kernel(const T* __restrict__ dout, const float* __restrict__ rois,
                                                               const A* __restrict__ argmax, T* __restrict__ dx, int R, int H,
                                                               int W, int C, int PH, int PW, float scale, int tilesW, const int* __restrict__ flag) {
  constexpr bool U8 = sizeof(A) == 1;
  __shared__ RoiCandOff coff[U8 ? RPB_MAXL : 1];
  constexpr int VE = Elem<T>::VE, GC = RPT_CC / VE;        // channel groups per block = items per thread (256 pixels x GC groups)
  constexpr int PR = 256 / GC / RPB_T;                     // tile rows covered by one pass of the block's threads
  __shared__ RoiGeo geo[RPB_MAXL];
  __shared__ RoiCand cand[RPB_MAXL];
  __shared__ int wcount[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int b = blockIdx.y, c0 = blockIdx.z * RPT_CC;
  const int h0 = ((int)blockIdx.x / tilesW) * RPB_T, w0 = ((int)blockIdx.x % tilesW) * RPB_T;
  const int g = t % GC, lw = (t / GC) % RPB_T, lh0 = t / (GC * RPB_T);      // item k of the thread: pixel (lh0 + k * PR, lw), group g
  const bool cvalid = c0 + g * VE < C && w0 + lw < W;
  float acc[GC][VE];
  int tgt[GC];
#pragma unroll
  for (int k = 0; k < GC; k++) {
    const int h = h0 + lh0 + k * PR;
    tgt[k] = (cvalid && h < H) ? h * W + w0 + lw : -2;     // -2 never equals a saved arg-max (>= -1)
#pragma unroll
    for (int e = 0; e < VE; e++) acc[k][e] = 0.f;
  }
  auto fetch = [&](int64_t o, int (&am)[VE], uint4& dv) RD_INLINE_LAMBDA {
    roi_load_arg(argmax + o, am);
    dv = *reinterpret_cast<const uint4*>(dout + o);
  };
  auto take = [&](const int (&am)[VE], const uint4& raw, int target, float (&a)[VE]) RD_INLINE_LAMBDA {
    float dv[VE];
    raw16_to_f32(reinterpret_cast<const T*>(0), raw, dv);
#pragma unroll
    for (int e = 0; e < VE; e++) a[e] += am[e] == target ? dv[e] : 0.f;
  };

  for (int rbase = 0; rbase < R; rbase += 256) {
    const int r = rbase + t;
    bool hit = false; RoiGeo gme;
    if (r < R) {
      const float* roi = rois + (int64_t)r * 5;
      const int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
      const int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
      const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
      gme.r = r; gme.sh = sh; gme.sw = sw; gme.eh = sh + rh + 1; gme.ew = sw + rw + 1;
      gme.bh = (float)rh / (float)PH; gme.bw = (float)rw / (float)PW;
      // conservative box: windows end at ceil((p+1)*bin) <= extent + 1
      hit = ((int)roi[0] == b) && (sh <= h0 + RPB_T - 1) && (sh + rh + 1 >= h0) && (sw <= w0 + RPB_T - 1) && (sw + rw + 1 >= w0);
    }
    const unsigned long long m = __ballot(hit);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wv] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wv; q++) woff += wcount[q];
    const int total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    const int mypos = woff + before;
    for (int cbase = 0; cbase < total; cbase += RPB_MAXL) {
      __syncthreads();
      if (hit && mypos >= cbase && mypos < cbase + RPB_MAXL) geo[mypos - cbase] = gme;
      __syncthreads();
      const int nl = min(total - cbase, RPB_MAXL);
      // ---- which bins cover each tile row / column of each listed RoI (the forward's expressions, roi_pool_fwd_vec_kernel) ---------
      for (int idx = t; idx < nl * 2 * RPB_T; idx += 256) {
        const int li = idx / (2 * RPB_T), j = idx - li * 2 * RPB_T;
        const bool rows = j < RPB_T;
        const int pos = rows ? j : j - RPB_T;
        const RoiGeo q = geo[li];
        const int x = (rows ? h0 : w0) + pos, start = rows ? q.sh : q.sw, lim = rows ? H : W, P = rows ? PH : PW;
        const float bin = rows ? q.bh : q.bw;
        int first = 0, cnt = 0;
        if (x < lim && x >= start - 1 && x < (rows ? q.eh : q.ew)) {
          // p with floor(p*bin) <= d < ceil((p+1)*bin), d = x - start: p in ((d-1)/bin - 1, (d+1)/bin)
          const float d = (float)(x - start);
          const int p_lo = max((int)floorf((d - 1.f) / bin) - 1, 0), p_hi = min((int)ceilf((d + 1.f) / bin), P - 1);
          for (int pp = p_lo; pp <= p_hi; pp++) {
            int s0 = (int)floorf((float)pp * bin), e0 = (int)ceilf((float)(pp + 1) * bin);
            s0 = min(max(s0 + start, 0), lim); e0 = min(max(e0 + start, 0), lim);
            if (x >= s0 && x < e0) { if (cnt == 0) first = pp; cnt++; }
          }
        }
        const int code = cnt ? ((first << 12) | min(cnt, 4095)) : 0;
        if (rows) cand[li].rowc[pos] = code; else cand[li].colc[pos] = code;
        if (j == 0) cand[li].r = q.r;
        if (U8) {      // x - window start of the first and the second covering bin (what the forward encoded for this pixel)
          int d[2];
#pragma unroll
          for (int k2 = 0; k2 < 2; k2++) {
            const int s0 = min(max((int)floorf((float)(first + k2) * bin) + start, 0), lim);
            d[k2] = (x - s0) & 255;
          }
          const int dcode = d[0] | (d[1] << 8);
          if (rows) coff[li].rowd[pos] = dcode; else coff[li].cold[pos] = dcode;
        }
      }
      __syncthreads();
      // ---- every item walks the list.  RC-Net's bins are ~1.02 pixels, so nearly every window is 2 x 2 pixels and nearly every pixel
      // lies in 2 x 2 bins: the 2 x 2 candidate block of KB pixels is requested in one go -- UNCONDITIONALLY (a candidate that does
      // not exist reads bin (0, 0) and is compared against a target that never matches): exec-masked or looped requests made every
      // candidate its own memory round trip, which was most of the kernel.  Candidates beyond 2 x 2 (bins smaller than a pixel) follow
      // in a plain loop. ----
      constexpr int KB = 2;
      static_assert(GC % KB == 0, "items per batch");
      const int goff = cvalid ? c0 + g * VE : 0;
      for (int li = 0; li < nl; li++) {
        const int cc = cand[li].colc[lw];
        const int rr = cand[li].r;
        const int nc = cc & 4095, cf = cc >> 12;
        const int cdw = U8 ? coff[li].cold[lw] : 0;
#pragma unroll
        for (int kb = 0; kb < GC; kb += KB) {
          int am[KB][4][VE]; uint4 dv[KB][4]; int nr[KB], rf[KB];
          int rdh[KB];
          bool any = false;
#pragma unroll
          for (int k = 0; k < KB; k++) {
            const int rc = nc ? cand[li].rowc[lh0 + (kb + k) * PR] : 0;
            nr[k] = tgt[kb + k] >= 0 ? (rc & 4095) : 0; rf[k] = rc >> 12;
            rdh[k] = U8 ? coff[li].rowd[lh0 + (kb + k) * PR] : 0;
            any = any || nr[k] > 0;
          }
          // a listed RoI meets the TILE, not necessarily this wave's rows (one tile row per item, all 16 columns): when no lane has a
          // covering bin the batch's 2 x 2 x KB requests -- one memory round trip of the serial walk -- are skipped for the whole wave
          if (__ballot(any) == 0ull) continue;
#pragma unroll
          for (int k = 0; k < KB; k++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int ia = q >> 1, ib = q & 1;
              const bool ok = ia < nr[k] && ib < nc;
              fetch((((int64_t)rr * PH + (ok ? rf[k] + ia : 0)) * PW + (ok ? cf + ib : 0)) * C + goff, am[k][q], dv[k][q]);
            }
          }
#pragma unroll
          for (int k = 0; k < KB; k++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int ia = q >> 1, ib = q & 1;
              // compact form: the pixel's code inside candidate bin (rf + ia, cf + ib) = (row offset << 4) | column offset
              const int want = U8 ? ((((rdh[k] >> (8 * ia)) & 255) << 4) | ((cdw >> (8 * ib)) & 255)) : tgt[kb + k];
              take(am[k][q], dv[k][q], (ia < nr[k] && ib < nc) ? want : -2, acc[kb + k]);
            }
            if (nr[k] > 2 || (nc > 2 && nr[k] > 0)) {      // bins smaller than a pixel: the candidates outside the 2 x 2 block
              const RoiGeo gq = geo[li];
              const int hh = h0 + lh0 + (kb + k) * PR, ww = w0 + lw;
              for (int ia = 0; ia < nr[k]; ia++)
                for (int ib = (ia < 2 ? 2 : 0); ib < nc; ib++) {
                  int am1[VE]; uint4 dv1;
                  fetch((((int64_t)rr * PH + rf[k] + ia) * PW + cf + ib) * C + goff, am1, dv1);
                  int want = tgt[kb + k];
                  if (U8) {
                    const int hs1 = min(max((int)floorf((float)(rf[k] + ia) * gq.bh) + gq.sh, 0), H);
                    const int ws1 = min(max((int)floorf((float)(cf + ib) * gq.bw) + gq.sw, 0), W);
                    want = (((hh - hs1) & 15) << 4) | ((ww - ws1) & 15);
                  }
                  take(am1, dv1, want, acc[kb + k]);
                }
            }
          }
        }
      }
    }
    __syncthreads();
  }
  const bool poison = U8 && *flag != 0;      // a window the compact arg-max could not encode: the gradient is made visibly wrong
#pragma unroll
  for (int k = 0; k < GC; k++) {
    const int h = h0 + lh0 + k * PR;
    if (poison) {
#pragma unroll
      for (int e = 0; e < VE; e++) acc[k][e] = __builtin_nanf("");
    }
    if (tgt[k] >= 0) stv(dx + (((int64_t)b * H + h) * W + w0 + lw) * C + c0 + g * VE, acc[k]);
  }
}

// ---- tile-accumulate form of the RoI-pool backward ---------------------------------------------------------------------------------
// A block owns a 16 x 16 pixel tile x 32 channels of one image as fp32 accumulators in LDS.  It lists the RoIs of the image that can
// reach the tile (as the gather kernel), and for each of them walks the sub-rectangle of bins whose windows can touch the tile:
// (bin, VE-channel) items read arg-max and dout with 16-byte vectors and add into the LDS tile where the arg-max falls inside it
// (ds_add_f32); bins on tile borders are scanned by both neighbours, each adds only its own pixels.  The tile is written once, in the
// activation dtype: no global atomics (83.6 M per RC-Net step before), no zero fill, no fp32 -> bf16 cast pass.
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_bwd_tile_kernel(const T* __restrict__ dout, const float* __restrict__ rois,
                                                                const int* __restrict__ argmax, T* __restrict__ dx, int R, int H,
                                                                int W, int C, int PH, int PW, float scale, int tilesW) {
  constexpr int VE = Elem<T>::VE, GC = RPT_CC / VE;
  __shared__ float tacc[RPB_T * RPB_T * RPT_CC];
  __shared__ RoiGeo list[RPB_MAXL];
  __shared__ int wcount[4];
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int b = blockIdx.y, c0 = blockIdx.z * RPT_CC;
  const int h0 = ((int)blockIdx.x / tilesW) * RPB_T, w0 = ((int)blockIdx.x % tilesW) * RPB_T;
  const float invW = 1.0f / (float)W;
  for (int i = t; i < RPB_T * RPB_T * RPT_CC; i += 256) tacc[i] = 0.f;

  for (int rbase = 0; rbase < R; rbase += 256) {
    const int r = rbase + t;
    bool hit = false; RoiGeo gme;
    if (r < R) {
      const float* roi = rois + (int64_t)r * 5;
      const int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
      const int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
      const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
      gme.r = r; gme.sh = sh; gme.sw = sw; gme.eh = sh + rh + 1; gme.ew = sw + rw + 1;
      gme.bh = (float)rh / (float)PH; gme.bw = (float)rw / (float)PW;
      hit = ((int)roi[0] == b) && (sh <= h0 + RPB_T - 1) && (sh + rh + 1 >= h0) && (sw <= w0 + RPB_T - 1) && (sw + rw + 1 >= w0);
    }
    const unsigned long long m = __ballot(hit);
    const int before = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wcount[wv] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < wv; q++) woff += wcount[q];
    const int total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
    const int mypos = woff + before;
    for (int cbase = 0; cbase < total; cbase += RPB_MAXL) {
      __syncthreads();
      if (hit && mypos >= cbase && mypos < cbase + RPB_MAXL) list[mypos - cbase] = gme;
      __syncthreads();
      const int nl = min(total - cbase, RPB_MAXL);
      for (int li = 0; li < nl; li++) {
        const RoiGeo q = list[li];
        // bins whose window [floor(p*bin), ceil((p+1)*bin)) + start can touch rows h0..h0+15 / columns w0..w0+15 (conservative)
        const int ph_lo = max((int)floorf((float)(h0 - q.sh - 1) / q.bh) - 1, 0), ph_hi = min((int)ceilf((float)(h0 + RPB_T - q.sh) / q.bh), PH - 1);
        const int pw_lo = max((int)floorf((float)(w0 - q.sw - 1) / q.bw) - 1, 0), pw_hi = min((int)ceilf((float)(w0 + RPB_T - q.sw) / q.bw), PW - 1);
        const int nph = ph_hi - ph_lo + 1, npw = pw_hi - pw_lo + 1;
        if (nph <= 0 || npw <= 0) continue;
        constexpr int NB = 256 / GC;
        static_assert(256 % GC == 0, "bin lanes");
        const int g = t % GC, tb = t / GC;
        // one item = (bin, 8 channels): arg-max + gradient (two / one 16-byte loads), then `apply(tile slot, value)` per channel
        auto fetch = [&](int ph, int pw, int (&am)[VE], float (&dv)[VE]) RD_INLINE_LAMBDA {
          const int64_t o = (((int64_t)q.r * PH + ph) * PW + pw) * C + c0 + g * VE;
#pragma unroll
          for (int e4 = 0; e4 < VE / 4; e4++) {
            const int4 a4 = *reinterpret_cast<const int4*>(argmax + o + e4 * 4);
            am[e4 * 4] = a4.x; am[e4 * 4 + 1] = a4.y; am[e4 * 4 + 2] = a4.z; am[e4 * 4 + 3] = a4.w;
          }
          ldv(dout + o, dv);
        };
        auto scatter = [&](const int (&am)[VE], const float (&dv)[VE], auto apply) RD_INLINE_LAMBDA {
          // bins are about one pixel: the VE channels usually share their arg-max, decode a pixel index only when it changes
          int pa = -2, pslot = -1;
#pragma unroll
          for (int e = 0; e < VE; e++) {
            const int a = am[e];
            if (a != pa) {
              pa = a; pslot = -1;
              if (a >= 0) {
                // float reciprocal estimate of a / W (error ~ ah * 2^-22: off by one row at most for a < 2^24) + one exact integer correction
                int ah = (int)(((float)a + 0.5f) * invW);
                int rem = a - ah * W;
                if (rem < 0) { ah--; rem += W; } else if (rem >= W) { ah++; rem -= W; }
                const int lh = ah - h0, lw = rem - w0;
                if ((unsigned)lh < (unsigned)RPB_T && (unsigned)lw < (unsigned)RPB_T) pslot = (lh * RPB_T + lw) * RPT_CC + g * VE;
              }
            }
            if (pslot >= 0) apply(pslot + e, dv[e]);
          }
        };
        if (q.bh >= 1.001f && q.bw >= 1.001f) {
          // Bins at least one pixel large (every RC-Net pooling: the RoI has the size of its output): the window of bin p ends before the
          // window of bin p + 2 begins (ceil((p+1) b) <= floor((p+2) b) for b >= 1; the 0.001 margin covers the float products), so bins
          // of one parity class (ph & 1, pw & 1) pick DIFFERENT pixels.  The four classes run one after the other with plain LDS
          // read-add-write: no atomics -- fp32 LDS atomics retire about one lane per clock (PMC: 100 LDS-busy cycles per wave instruction,
          // 60 % of this kernel) -- and the summation order is fixed (RoI order, then class order).  The first item of the NEXT class
          // is requested before the barrier that ends the current one, so the barriers do not expose the load latency.
          auto plain = [&](int slot, float v) RD_INLINE_LAMBDA { tacc[slot] += v; };
          int cph0[4], cpw0[4], cnw[4], cnb[4];
#pragma unroll
          for (int cls = 0; cls < 4; cls++) {
            cph0[cls] = ph_lo + (((cls >> 1) ^ ph_lo) & 1); cpw0[cls] = pw_lo + (((cls & 1) ^ pw_lo) & 1);
            const int nh = cph0[cls] <= ph_hi ? (ph_hi - cph0[cls]) / 2 + 1 : 0;
            cnw[cls] = cpw0[cls] <= pw_hi ? (pw_hi - cpw0[cls]) / 2 + 1 : 0;
            cnb[cls] = nh * cnw[cls];
          }
          int amA[VE]; float dvA[VE];
          auto fetch_cls = [&](int cls, int bq, int (&am)[VE], float (&dv)[VE]) RD_INLINE_LAMBDA {
            const int i = bq / cnw[cls], j = bq - i * cnw[cls];
            fetch(cph0[cls] + 2 * i, cpw0[cls] + 2 * j, am, dv);
          };
          if (tb < cnb[0]) fetch_cls(0, tb, amA, dvA);
#pragma unroll
          for (int cls = 0; cls < 4; cls++) {
            __syncthreads();   // the previous class (or a previous RoI's atomics) is complete in every wave
            if (tb < cnb[cls]) scatter(amA, dvA, plain);
            for (int bq = tb + NB; bq < cnb[cls]; bq += NB) {
              int am[VE]; float dv[VE];
              fetch_cls(cls, bq, am, dv);
              scatter(am, dv, plain);
            }
            if (cls < 3 && tb < cnb[cls + 1]) fetch_cls(cls + 1, tb, amA, dvA);
          }
          __syncthreads();     // before the next RoI touches the tile (it may use atomics)
        } else {
          // general bins (smaller than a pixel: many bins per pixel): LDS atomics; the bin index advances by NB per iteration and
          // (ph, pw) follow by carries
          const int nbins = nph * npw;
          const int dph = NB / npw, dpw = NB - dph * npw;
          int phi = tb / npw, pwi = tb - phi * npw;
          for (int bq = tb; bq < nbins; bq += NB) {
            const int pw = pw_lo + pwi, ph = ph_lo + phi;
            pwi += dpw; phi += dph;
            if (pwi >= npw) { pwi -= npw; phi++; }
            int am[VE]; float dv[VE];
            fetch(ph, pw, am, dv);
            scatter(am, dv, [&](int slot, float v) RD_INLINE_LAMBDA { atomicAdd(&tacc[slot], v); });
          }
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  for (int i = t; i < RPB_T * RPB_T * (RPT_CC / 4); i += 256) {
    const int c4 = (i % (RPT_CC / 4)) * 4, pix = i / (RPT_CC / 4);
    const int h = h0 + pix / RPB_T, w = w0 + pix % RPB_T;
    if (h < H && w < W) {
      float v[4] = {tacc[pix * RPT_CC + c4], tacc[pix * RPT_CC + c4 + 1], tacc[pix * RPT_CC + c4 + 2], tacc[pix * RPT_CC + c4 + 3]};
      st4(dx + (((int64_t)b * H + h) * W + w) * C + c0 + c4, v);
    }
  }
}

// explicit zero-fill kernel (kept as a kernel node so hipGraph replays order it like every other launch)
__global__ __launch_bounds__(256) void zero_f32_kernel(float* __restrict__ p, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
    reinterpret_cast<float4*>(p)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

static unsigned ew_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 8192)); }      // (4096 until round 6: 7.04 -> 7.03 ms per RC-Net step, two alternating rounds)

void launch_maxpool_fwd(const void* x, void* out, unsigned char* arg, int N, int H, int W, int C, int OH, int OW, int k, int s,
                        int p, int dtype, hipStream_t st) {
  int64_t n = (int64_t)N * OH * OW * C;
  const int ve = dtype == 0 ? 4 : 8;
  if (C % ve == 0 && (reinterpret_cast<uintptr_t>(arg) & 3) == 0) {
    const unsigned gv = ew_grid(n / ve);
    if (dtype == 0) hipLaunchKernelGGL((maxpool_fwd_vec_kernel<float>), dim3(gv), dim3(256), 0, st, (const float*)x, (float*)out, arg, N, H, W, C, OH, OW, k, s, p);
    else hipLaunchKernelGGL((maxpool_fwd_vec_kernel<bf16_t>), dim3(gv), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, arg, N, H, W, C, OH, OW, k, s, p);
    return;
  }
  if (dtype == 0) hipLaunchKernelGGL((maxpool_fwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)x, (float*)out, arg, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((maxpool_fwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, arg, N, H, W, C, OH, OW, k, s, p);
}
void launch_maxpool_bwd(const void* dout, const unsigned char* arg, void* dx, int N, int H, int W, int C, int OH, int OW, int k,
                        int s, int p, int dtype, hipStream_t st) {
  int64_t n = (int64_t)N * H * W * C;
  const int ve = dtype == 0 ? 4 : 8;
  if (C % ve == 0) {
    unsigned gv = ew_grid(n / ve);
    if (dtype == 0) hipLaunchKernelGGL((maxpool_bwd_vec_kernel<float>), dim3(gv), dim3(256), 0, st, (const float*)dout, arg, (float*)dx, N, H, W, C, OH, OW, k, s, p);
    else hipLaunchKernelGGL((maxpool_bwd_vec_kernel<bf16_t>), dim3(gv), dim3(256), 0, st, (const bf16_t*)dout, arg, (bf16_t*)dx, N, H, W, C, OH, OW, k, s, p);
    return;
  }
  if (dtype == 0) hipLaunchKernelGGL((maxpool_bwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)dout, arg, (float*)dx, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((maxpool_bwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)dout, arg, (bf16_t*)dx, N, H, W, C, OH, OW, k, s, p);
}
void launch_roi_pool_fwd(const void* x, const float* rois, void* out, int* argmax, int R, int N, int H, int W, int C, int PH,
                         int PW, float scale, int dtype, hipStream_t st) {
  int64_t n = (int64_t)R * PH * PW * C;
  if (n == 0) return;
  const int ve = dtype == 0 ? 4 : 8;
  if (C % ve == 0) {
    unsigned gv = ew_grid(n / ve);
    if (dtype == 0) hipLaunchKernelGGL((roi_pool_fwd_vec_kernel<float, int>), dim3(gv), dim3(256), 0, st, (const float*)x, rois, (float*)out, argmax, R, N, H, W, C, PH, PW, scale, (int*)nullptr);
    else hipLaunchKernelGGL((roi_pool_fwd_vec_kernel<bf16_t, int>), dim3(gv), dim3(256), 0, st, (const bf16_t*)x, rois, (bf16_t*)out, argmax, R, N, H, W, C, PH, PW, scale, (int*)nullptr);
    return;
  }
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

void launch_roi_pool_bwd_tile(const void* dout, const float* rois, const int* argmax, void* dx, int R, int N, int H, int W, int C,
                              int PH, int PW, float scale, int dtype, hipStream_t st) {
  const int tilesH = (int)cdiv(H, RPB_T), tilesW = (int)cdiv(W, RPB_T);
  dim3 grid((unsigned)(tilesH * tilesW), (unsigned)N, (unsigned)(C / RPT_CC));
  if (dtype == 0) hipLaunchKernelGGL((roi_pool_bwd_tile_kernel<float>), grid, dim3(256), 0, st, (const float*)dout, rois, argmax, (float*)dx, R, H, W, C, PH, PW, scale, tilesW);
  else hipLaunchKernelGGL((roi_pool_bwd_tile_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)dout, rois, argmax, (bf16_t*)dx, R, H, W, C, PH, PW, scale, tilesW);
}
void launch_roi_pool_bwd_gather(const void* dout, const float* rois, const int* argmax, void* dx, int R, int N, int H, int W, int C,
                                int PH, int PW, float scale, int dtype, hipStream_t st) {
  const int tilesH = (int)cdiv(H, RPB_T), tilesW = (int)cdiv(W, RPB_T);
  dim3 grid((unsigned)(tilesH * tilesW), (unsigned)N, (unsigned)cdiv(C, RPT_CC));
  if (dtype == 0) hipLaunchKernelGGL((roi_pool_bwd_pix_kernel<float, int>), grid, dim3(256), 0, st, (const float*)dout, rois, argmax, (float*)dx, R, H, W, C, PH, PW, scale, tilesW, (const int*)nullptr);
  else hipLaunchKernelGGL((roi_pool_bwd_pix_kernel<bf16_t, int>), grid, dim3(256), 0, st, (const bf16_t*)dout, rois, argmax, (bf16_t*)dx, R, H, W, C, PH, PW, scale, tilesW, (const int*)nullptr);
}

// ---- compact (one byte per element) arg-max forms: C must be a multiple of the 16-byte vector -----------------------------------------
void launch_roi_pool_fwd_u8(const void* x, const float* rois, void* out, unsigned char* argmax, int* flag, int R, int N, int H, int W, int C,
                            int PH, int PW, float scale, int dtype, hipStream_t st) {
  const int64_t n = (int64_t)R * PH * PW * C;
  if (n == 0) return;
  const unsigned gv = ew_grid(n / (dtype == 0 ? 4 : 8));
  if (dtype == 0) hipLaunchKernelGGL((roi_pool_fwd_vec_kernel<float, unsigned char>), dim3(gv), dim3(256), 0, st, (const float*)x, rois, (float*)out, argmax, R, N, H, W, C, PH, PW, scale, flag);
  else hipLaunchKernelGGL((roi_pool_fwd_vec_kernel<bf16_t, unsigned char>), dim3(gv), dim3(256), 0, st, (const bf16_t*)x, rois, (bf16_t*)out, argmax, R, N, H, W, C, PH, PW, scale, flag);
}
void launch_roi_pool_bwd_u8(const void* dout, const float* rois, const unsigned char* argmax, const int* flag, float* dx_f32, int R, int N, int H,
                            int W, int C, int PH, int PW, float scale, int dtype, hipStream_t st) {
  const int64_t nz = (int64_t)N * H * W * C;
  hipLaunchKernelGGL(zero_f32_kernel, dim3(ew_grid(nz / 4 + 1)), dim3(256), 0, st, dx_f32, nz);
  const int64_t n = (int64_t)R * PH * PW * C;
  if (n == 0) return;
  if (dtype == 0) hipLaunchKernelGGL((roi_pool_bwd_u8_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)dout, rois, argmax, flag, dx_f32, R, N, H, W, C, PH, PW, scale);
  else hipLaunchKernelGGL((roi_pool_bwd_u8_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)dout, rois, argmax, flag, dx_f32, R, N, H, W, C, PH, PW, scale);
}
void launch_roi_pool_bwd_gather_u8(const void* dout, const float* rois, const unsigned char* argmax, const int* flag, void* dx, int R, int N, int H,
                                   int W, int C, int PH, int PW, float scale, int dtype, hipStream_t st) {
  const int tilesH = (int)cdiv(H, RPB_T), tilesW = (int)cdiv(W, RPB_T);
  dim3 grid((unsigned)(tilesH * tilesW), (unsigned)N, (unsigned)cdiv(C, RPT_CC));
  if (dtype == 0) hipLaunchKernelGGL((roi_pool_bwd_pix_kernel<float, unsigned char>), grid, dim3(256), 0, st, (const float*)dout, rois, argmax, (float*)dx, R, H, W, C, PH, PW, scale, tilesW, flag);
  else hipLaunchKernelGGL((roi_pool_bwd_pix_kernel<bf16_t, unsigned char>), grid, dim3(256), 0, st, (const bf16_t*)dout, rois, argmax, (bf16_t*)dx, R, H, W, C, PH, PW, scale, tilesW, flag);
}

}  // namespace rd
