// RC-Net batch augmentation on the device (reference: RCNet/rcnet_transforms.py:58-240, Transforms.transform as configured by
// train_rcnet_zju.py:52-59: brightness / contrast / saturation in [0.8, 1.2], horizontal flip, probability 1) and the ground-truth
// crop extraction of the dataset (data/datasets.py:254-272).
//
// The photometric functions are torchvision.transforms.functional.adjust_{brightness,contrast,saturation} (torchvision 0.14, NOT in the
// reference tree) applied to INT32 images -- the reference converts a 0..255 float image with images.int() (:104-107) -- i.e. the tensor
// path  _blend(img1, img2, ratio) = (ratio * img1 + (1 - ratio) * img2).clamp(0, 255).to(int32)  (float32 arithmetic, truncation) with
//   brightness: img2 = 0;  contrast: img2 = mean over the image of gray(img), gray = (0.2989 r + 0.587 g + 0.114 b).to(int32);
//   saturation: img2 = gray(img) per pixel.
// Order: brightness -> contrast -> saturation -> float -> normalise -> [point noise, host-drawn, rd_add] -> horizontal flip of image,
// ground-truth crops and boxes (x1' = W - x2, x2' = W - x1) -> vertical flip of image and crops (:199-217); the radar points are NOT
// flipped (:174-197), and the vertical flip's box update indexes bounding_boxes[b, 1] and [b, 3] -- BOXES 1 and 3 of the sample, all four
// coordinates -- rather than y1 / y2 of every box (:213-217): both quirks are kept as they are.
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

// per-image parameters (floats): 0 do_brightness, 1 factor, 2 do_contrast, 3 factor, 4 do_saturation, 5 factor, 6 do_hflip, 7 do_vflip
static constexpr int AUGP = 8;
static constexpr int AUG_CHUNKS = 32;

__device__ __forceinline__ int aug_blend(int v, float w, float ratio) {   // torchvision _blend on an int image, one element
  const float om = (float)(1.0 - (double)ratio);
  float t = __fadd_rn(__fmul_rn(ratio, (float)v), __fmul_rn(om, w));
  t = fminf(fmaxf(t, 0.f), 255.f);
  return (int)t;
}
__device__ __forceinline__ int aug_gray(int r, int g, int b) {
  return (int)__fadd_rn(__fadd_rn(__fmul_rn(0.2989f, (float)r), __fmul_rn(0.587f, (float)g)), __fmul_rn(0.114f, (float)b));
}
__device__ __forceinline__ void aug_brightness(const float* __restrict__ p, int (&v)[3]) {
  if (p[0] != 0.f) {
#pragma unroll
    for (int c = 0; c < 3; c++) v[c] = aug_blend(v[c], 0.f, p[1]);
  }
}

// sum over the image of gray(brightness-adjusted image): integer, so the partial sums are exact and order independent
__global__ __launch_bounds__(256) void augment_gray_partials_kernel(const float* __restrict__ image, int HW, const float* __restrict__ params,
                                                                    long long* __restrict__ partial) {
  __shared__ long long sh[4];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const float* p = params + b * AUGP;
  const float* img = image + (int64_t)b * 3 * HW;
  long long s = 0;
  if (p[2] != 0.f) {
    for (int i = chunk * 256 + threadIdx.x; i < HW; i += AUG_CHUNKS * 256) {
      int v[3] = {(int)img[i], (int)img[HW + i], (int)img[2 * HW + i]};
      aug_brightness(p, v);
      s += aug_gray(v[0], v[1], v[2]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[b * AUG_CHUNKS + chunk] = sh[0] + sh[1] + sh[2] + sh[3];
}

// image (B,3,H,W) float 0..255 -> out (B,H,W,3) in the activation dtype, normalised (v * scale + shift), horizontally flipped where asked
template <typename T>
__global__ __launch_bounds__(256) void augment_image_kernel(const float* __restrict__ image, int B, int H, int W, const float* __restrict__ params,
                                                            const long long* __restrict__ partial, T* __restrict__ out, float scale, float shift) {
  const int HW = H * W;
  const int64_t total = (int64_t)B * HW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / HW), pix = (int)(i - (int64_t)b * HW);
    const int y = pix / W, x = pix - y * W;
    const float* p = params + b * AUGP;
    const float* img = image + (int64_t)b * 3 * HW;
    int v[3] = {(int)img[pix], (int)img[HW + pix], (int)img[2 * HW + pix]};
    aug_brightness(p, v);
    if (p[2] != 0.f) {
      long long s = 0;
#pragma unroll 8
      for (int c = 0; c < AUG_CHUNKS; c++) s += partial[b * AUG_CHUNKS + c];
      const float mean = (float)((double)s / (double)HW);
#pragma unroll
      for (int c = 0; c < 3; c++) v[c] = aug_blend(v[c], mean, p[3]);
    }
    if (p[4] != 0.f) {
      const float g = (float)aug_gray(v[0], v[1], v[2]);
#pragma unroll
      for (int c = 0; c < 3; c++) v[c] = aug_blend(v[c], g, p[5]);
    }
    const int xo = p[6] != 0.f ? W - 1 - x : x;
    const int yo = p[7] != 0.f ? H - 1 - y : y;
    T* o = out + ((int64_t)b * HW + (int64_t)yo * W + xo) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) Elem<T>::st(o + c, __fadd_rn(__fmul_rn((float)v[c], scale), shift));
  }
}

// ground-truth crops (B, K, 1, ph, pw): flipped copy for the flagged images; boxes (B, K, 4) in place: x1' = n_width - x2, x2' = n_width - x1
__global__ __launch_bounds__(256) void augment_flip_labels_kernel(const float* __restrict__ lin, float* __restrict__ lout, int B, int K, int ph, int pw,
                                                                  float* __restrict__ boxes, const float* __restrict__ params, float n_width) {
  const int64_t per = (int64_t)K * ph * pw, total = (int64_t)B * per;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per);
    const int x = (int)(i % pw), y = (int)((i / pw) % ph);
    int64_t src = i;
    if (params[b * AUGP + 6] != 0.f) src += (pw - 1 - x) - x;
    if (params[b * AUGP + 7] != 0.f) src += (int64_t)((ph - 1 - y) - y) * pw;
    lout[i] = lin[src];
  }
  if (boxes)
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < B * K; j += gridDim.x * blockDim.x) {
      if (params[(j / K) * AUGP + 6] != 0.f) {
        const float x1 = boxes[j * 4 + 0], x2 = boxes[j * 4 + 2];
        boxes[j * 4 + 0] = n_width - x2;
        boxes[j * 4 + 2] = n_width - x1;
      }
    }
}

// :213-217 as written: for a vertically flipped sample, boxes[b][1][:] <- n_height - boxes[b][3][:] and boxes[b][3][:] <- n_height - (old) boxes[b][1][:]
// (runs after the horizontal update, as in the reference: a separate launch, the rows were written by other threads)
__global__ __launch_bounds__(64) void augment_vflip_boxes_kernel(float* __restrict__ boxes, int B, int K, const float* __restrict__ params, float n_height) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;      // (sample, coordinate)
  if (j >= B * 4) return;
  const int b = j >> 2, c = j & 3;
  if (params[b * AUGP + 7] == 0.f) return;
  float* r1 = boxes + ((int64_t)b * K + 1) * 4 + c;
  float* r3 = boxes + ((int64_t)b * K + 3) * 4 + c;
  const float t = *r1;
  *r1 = n_height - *r3;
  *r3 = n_height - t;
}

// data/datasets.py:254-272: crops[b][k] = padded_gt[b][0][int(y) - pad_y : int(y) + pad_y][int(x) - pad_x : int(x) + pad_x], (x, y) = radar point
// in padded coordinates; the zero padding of the reference's np.pad is the caller's (gt is the padded map)
__global__ __launch_bounds__(256) void crop_patches_kernel(const float* __restrict__ gt, const float* __restrict__ points, float* __restrict__ crops,
                                                           int B, int K, int Hp, int Wp, int ph, int pw) {
  const int64_t per = (int64_t)ph * pw, total = (int64_t)B * K * per;
  const int pad_y = ph / 2, pad_x = pw / 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / per), o = (int)(i - (int64_t)r * per);
    const int b = r / K, cy = o / pw, cx = o - cy * pw;
    const int y = (int)(points[r * 3 + 1] - (float)pad_y) + cy, x = (int)(points[r * 3 + 0] - (float)pad_x) + cx;
    crops[i] = ((unsigned)y < (unsigned)Hp && (unsigned)x < (unsigned)Wp) ? gt[((int64_t)b * Hp + y) * Wp + x] : 0.f;
  }
}

static unsigned aug_grid(int64_t n) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 4096)); }

void launch_augment_gray_partials(const float* image, int B, int H, int W, const float* params, long long* partial, hipStream_t st) {
  hipLaunchKernelGGL(augment_gray_partials_kernel, dim3(AUG_CHUNKS, B), dim3(256), 0, st, image, H * W, params, partial);
}
void launch_augment_image(const float* image, int B, int H, int W, const float* params, const long long* partial, void* out, int dtype, float scale,
                          float shift, hipStream_t st) {
  const unsigned g = aug_grid((int64_t)B * H * W);
  if (dtype == 0) hipLaunchKernelGGL((augment_image_kernel<float>), dim3(g), dim3(256), 0, st, image, B, H, W, params, partial, (float*)out, scale, shift);
  else hipLaunchKernelGGL((augment_image_kernel<bf16_t>), dim3(g), dim3(256), 0, st, image, B, H, W, params, partial, (bf16_t*)out, scale, shift);
}
void launch_augment_flip_labels(const float* lin, float* lout, int B, int K, int ph, int pw, float* boxes, const float* params, float n_width,
                                hipStream_t st) {
  hipLaunchKernelGGL(augment_flip_labels_kernel, dim3(aug_grid((int64_t)B * K * ph * pw)), dim3(256), 0, st, lin, lout, B, K, ph, pw, boxes, params, n_width);
}
void launch_augment_vflip_boxes(float* boxes, int B, int K, const float* params, float n_height, hipStream_t st) {
  hipLaunchKernelGGL(augment_vflip_boxes_kernel, dim3(cdiv(B * 4, 64)), dim3(64), 0, st, boxes, B, K, params, n_height);
}
void launch_crop_patches(const float* gt, const float* points, float* crops, int B, int K, int Hp, int Wp, int ph, int pw, hipStream_t st) {
  hipLaunchKernelGGL(crop_patches_kernel, dim3(aug_grid((int64_t)B * K * ph * pw)), dim3(256), 0, st, gt, points, crops, B, K, Hp, Wp, ph, pw);
}

// ---- offline point-cloud projection / scatter (data/preprocess/project_transform.py:67-97, pointcloud_project_zju.py:57-76,81-103) ------------
// p_cam = T p (homogeneous, float64 as numpy computes it: the calibration matrices are float64), depth = p_cam.z, (u, v) = rint(P p_cam / w)
// (np.round: half to even), kept if 0 < u < W, 0 < v < H, depth > 0 (canvas_crop, strict at 0) and min < depth < max (min_max_filter);
// depth_map[v][u] = max(depth, 1) with NEARER points overwriting farther ones (the reference sorts by depth, descending, and writes in
// that order).  "The nearest wins" is order independent: an atomic minimum on the float32 bit pattern (positive floats order like
// unsigned integers), so the scatter is bit-exact and deterministic whatever the point order.  The map must be pre-filled with 0xFFFFFFFF;
// pass 2 turns untouched pixels into 0.  Optionally emits the kept points (u, v, depth) compacted in input order + their count.
__global__ __launch_bounds__(256) void project_scatter_kernel(const float* __restrict__ pts, int n, int stride, const double* __restrict__ T,
                                                              const double* __restrict__ P, int H, int W, double dmin, double dmax,
                                                              unsigned* __restrict__ map, float* __restrict__ kept, int* __restrict__ nkept) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const double x = (double)pts[(int64_t)i * stride], y = (double)pts[(int64_t)i * stride + 1], z = (double)pts[(int64_t)i * stride + 2];
    double c[4];
#pragma unroll
    for (int r = 0; r < 4; r++) c[r] = ((T[r * 4 + 0] * x + T[r * 4 + 1] * y) + T[r * 4 + 2] * z) + T[r * 4 + 3] * 1.0;
    double q[3];
#pragma unroll
    for (int r = 0; r < 3; r++) q[r] = ((P[r * 4 + 0] * c[0] + P[r * 4 + 1] * c[1]) + P[r * 4 + 2] * c[2]) + P[r * 4 + 3] * c[3];
    const double depth = c[2];
    const double uf = rint(q[0] / q[2]), vf = rint(q[1] / q[2]);
    if (!(uf > 0.0 && uf < (double)W && vf > 0.0 && vf < (double)H && depth > 0.0 && depth < dmax && depth > dmin)) continue;
    const int u = (int)uf, v = (int)vf;
    const float val = fmaxf((float)depth, 1.0f);
    atomicMin(&map[v * W + u], __float_as_uint(val));
    if (kept) {
      const int k = atomicAdd(nkept, 1);
      kept[k * 3 + 0] = (float)u; kept[k * 3 + 1] = (float)v; kept[k * 3 + 2] = (float)depth;
    }
  }
}
__global__ __launch_bounds__(256) void scatter_finish_kernel(unsigned* __restrict__ map, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (map[i] == 0xFFFFFFFFu) map[i] = 0u;
}
__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned* __restrict__ p, int n, unsigned v) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
// ---- lidar interpolation: barycentric raster of a Delaunay triangulation (data/data_utils.py:231-275, :333-367) ----------------------
// The reference builds scipy's LinearNDInterpolator over the valid pixels and evaluates it at every pixel.  The triangulation (Qhull) stays
// on the host, as there; what is evaluated H*W times -- point location + barycentric weights -- runs here.  Data points AND queries are
// integer pixel coordinates, so the edge functions are exact in int64: a pixel lies in a triangle iff its three edge functions do not
// have opposite signs (edges and vertices included).  Pass 1: one wave per triangle walks its bounding box and claims the pixels it
// contains with atomicMin(triangle index) -- a pixel on a shared edge belongs to the lowest-numbered triangle, so the result does not
// depend on scheduling (the interpolant is continuous across the edge anyway).  Pass 2: every pixel evaluates its owner in double.
__device__ __forceinline__ int64_t tri_edge(int r0, int c0, int r1, int c1, int r, int c) {
  return (int64_t)(r1 - r0) * (c - c0) - (int64_t)(c1 - c0) * (r - r0);
}
__global__ __launch_bounds__(256) void tri_owner_fill_kernel(int* __restrict__ owner, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) owner[i] = 0x7fffffff;
}
__global__ __launch_bounds__(256) void tri_owner_kernel(const int* __restrict__ tri, const int* __restrict__ prow, const int* __restrict__ pcol,
                                                        int M, int H, int W, int* __restrict__ owner) {
  const int lane = threadIdx.x & 63;
  for (int t = blockIdx.x * 4 + (threadIdx.x >> 6); t < M; t += gridDim.x * 4) {
    const int a = tri[t * 3], b = tri[t * 3 + 1], c = tri[t * 3 + 2];
    const int r0 = prow[a], c0 = pcol[a], r1 = prow[b], c1 = pcol[b], r2 = prow[c], c2 = pcol[c];
    if (tri_edge(r0, c0, r1, c1, r2, c2) == 0) continue;                    // degenerate simplex
    const int rlo = max(min(r0, min(r1, r2)), 0), rhi = min(max(r0, max(r1, r2)), H - 1);
    const int clo = max(min(c0, min(c1, c2)), 0), chi = min(max(c0, max(c1, c2)), W - 1);
    const int bw = chi - clo + 1, n = (rhi - rlo + 1) * bw;
    for (int i = lane; i < n; i += 64) {
      const int r = rlo + i / bw, cc = clo + i % bw;
      const int64_t e0 = tri_edge(r1, c1, r2, c2, r, cc), e1 = tri_edge(r2, c2, r0, c0, r, cc), e2 = tri_edge(r0, c0, r1, c1, r, cc);
      const bool neg = e0 < 0 || e1 < 0 || e2 < 0, pos = e0 > 0 || e1 > 0 || e2 > 0;
      if (!(neg && pos)) atomicMin(&owner[(int64_t)r * W + cc], t);
    }
  }
}
__global__ __launch_bounds__(256) void tri_interp_kernel(const int* __restrict__ tri, const int* __restrict__ prow, const int* __restrict__ pcol,
                                                         const double* __restrict__ values, const int* __restrict__ owner, int H, int W,
                                                         double fill, double* __restrict__ out) {
  const int64_t n = (int64_t)H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = owner[i];
    if (t == 0x7fffffff) { out[i] = fill; continue; }
    const int r = (int)(i / W), cc = (int)(i % W);
    const int a = tri[t * 3], b = tri[t * 3 + 1], c = tri[t * 3 + 2];
    const int r0 = prow[a], c0 = pcol[a], r1 = prow[b], c1 = pcol[b], r2 = prow[c], c2 = pcol[c];
    const double area = (double)tri_edge(r0, c0, r1, c1, r2, c2);
    const double w0 = (double)tri_edge(r1, c1, r2, c2, r, cc) / area, w1 = (double)tri_edge(r2, c2, r0, c0, r, cc) / area;
    const double w2 = 1.0 - w0 - w1;
    out[i] = w0 * values[a] + w1 * values[b] + w2 * values[c];
  }
}
void launch_tri_raster(const int* tri, const int* prow, const int* pcol, const double* values, int M, int H, int W, double fill, int* owner,
                       double* out, hipStream_t st) {
  const int64_t n = (int64_t)H * W;
  hipLaunchKernelGGL(tri_owner_fill_kernel, dim3(aug_grid(n)), dim3(256), 0, st, owner, n);
  if (M > 0) hipLaunchKernelGGL(tri_owner_kernel, dim3((unsigned)std::min<int64_t>(cdiv(M, 4), 4096)), dim3(256), 0, st, tri, prow, pcol, M, H, W, owner);
  hipLaunchKernelGGL(tri_interp_kernel, dim3(aug_grid(n)), dim3(256), 0, st, tri, prow, pcol, values, owner, H, W, fill, out);
}

// ---- nearest-knot map (modules/interpolator.py:7-18 with interpolate = 'nearest': scipy griddata -> NearestNDInterpolator) ------------
// Every pixel takes the value of the closest knot.  Knots and queries are integer pixel coordinates, so the squared distances are exact
// integers; among equidistant knots the LOWEST index wins (scipy's cKDTree leaves ties implementation-defined; the fixture has none).
// Brute force with the knots staged through LDS in blocks of 256: K is the number of radar / lidar returns of one frame.
__global__ __launch_bounds__(256) void nearest_knot_kernel(const int* __restrict__ prow, const int* __restrict__ pcol, const double* __restrict__ values,
                                                           int K, int H, int W, double fill, double* __restrict__ out) {
  __shared__ int sr[256], sc[256];
  const int64_t n = (int64_t)H * W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int r = (int)(i / W), c = (int)(i % W);
  int64_t best = INT64_MAX; int arg = -1;
  for (int k0 = 0; k0 < K; k0 += 256) {
    const int kk = k0 + (int)threadIdx.x;
    sr[threadIdx.x] = kk < K ? prow[kk] : 0; sc[threadIdx.x] = kk < K ? pcol[kk] : 0;
    __syncthreads();
    const int m = min(256, K - k0);
    for (int j = 0; j < m; j++) {
      const int64_t dr = r - sr[j], dc = c - sc[j], d2 = dr * dr + dc * dc;
      if (d2 < best) { best = d2; arg = k0 + j; }
    }
    __syncthreads();
  }
  if (i < n) out[i] = arg >= 0 ? values[arg] : fill;
}
void launch_nearest_knot(const int* prow, const int* pcol, const double* values, int K, int H, int W, double fill, double* out, hipStream_t st) {
  hipLaunchKernelGGL(nearest_knot_kernel, dim3((unsigned)cdiv((int64_t)H * W, 256)), dim3(256), 0, st, prow, pcol, values, K, H, W, fill, out);
}

void launch_project_scatter(const float* pts, int n, int stride, const double* T, const double* P, int H, int W, double dmin, double dmax, float* depth_map,
                            float* kept, int* nkept, hipStream_t st) {
  unsigned* map = reinterpret_cast<unsigned*>(depth_map);
  hipLaunchKernelGGL(fill_u32_kernel, dim3(aug_grid((int64_t)H * W)), dim3(256), 0, st, map, H * W, 0xFFFFFFFFu);
  if (nkept) hipLaunchKernelGGL(fill_u32_kernel, dim3(1), dim3(64), 0, st, reinterpret_cast<unsigned*>(nkept), 1, 0u);
  if (n > 0) hipLaunchKernelGGL(project_scatter_kernel, dim3(aug_grid(n)), dim3(256), 0, st, pts, n, stride, T, P, H, W, dmin, dmax, map, kept, nkept);
  hipLaunchKernelGGL(scatter_finish_kernel, dim3(aug_grid((int64_t)H * W)), dim3(256), 0, st, map, H * W);
}

}  // namespace rd
