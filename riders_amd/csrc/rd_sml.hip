// Scale-Map-Learner pre-step, loss, outlier removal and validation kernels (all HBM-bound; reductions by wave shuffles +
// per-block partials combined in a fixed order).
//
// Reference:
//   train_zju.py:246-343        per-sample CPU pre-step (valid masks, 1/depth, Optimizer.optimize_scale, int_scales,
//                               min-max normalise, cv2 nearest resize, fixed mean/std normalise, gray image)
//   modules/estimator.py:129-176 objective sum(mask*|s*p - t|), bounded minimisation, clamps
//   utils/net_utils.py:591-638  OutlierRemoval.remove_outliers
//   utils/loss.py:5-135,187-274 compute_loss ('l1'), sobel_smoothness_loss_func, sobel_filter
//   val_zju.py:198-231, utils/eval_utils.py  bicubic resize (A=-0.75, align_corners=False) and the metric set
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

static unsigned ew_grid(int64_t n, int cap = 2048) { return (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), cap)); }

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wv] = v;
  __syncthreads();
  double r = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); w++) r += sh[w];
  return r;
}

// The per-sample reductions below run ONE block per sample (B <= a few dozen): the block is 1024 threads wide and every thread keeps two
// 16-byte requests per tensor in flight -- a 256-thread block issuing one scalar load per iteration spent 512 dependent memory round trips
// per sample (~0.3 ms per launch at 256 x 512).
static constexpr int SML_SCAN_T = 1024;
__device__ __forceinline__ bool sml_vec_ok(const void* a, const void* b, const void* c, int HW) {
  return (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// ---- S1: global scale alignment --------------------------------------------------------------------------------------
// s* = argmin_{s in [lo,hi]} sum_valid |s*p - t|, t = 1/depth on valid radar pixels.  The objective is convex piecewise
// linear, so s* is where its slope g(s) = sum p*sign(s*p - t) changes sign: 50 bisection steps, one block per sample.
// (scipy's bounded Brent stops within xatol = 1e-5 of the same point; with no valid pixel it returns 0.299996..., kept.)
__global__ __launch_bounds__(SML_SCAN_T) void sml_scale_align_kernel(const float* __restrict__ mono, const float* __restrict__ sparse, int HW,
                                                              float dmin, float dmax, float lo, float hi, float* __restrict__ scale,
                                                              int* __restrict__ nvalid) {
  constexpr int CAP = 4096;  // valid radar pixels kept in LDS (sparse radar: a few hundred per frame); more -> rescan global memory
  __shared__ double sh[SML_SCAN_T / 64];
  __shared__ float sp[CAP], st[CAP];
  __shared__ int scount;
  const int b = blockIdx.x, T = blockDim.x;
  const float* p = mono + (int64_t)b * HW;
  const float* z = sparse + (int64_t)b * HW;
  if (threadIdx.x == 0) scount = 0;
  __syncthreads();
  auto keep = [&](float zi, float pi) {
    if (zi < dmax && zi > dmin) {
      int k = atomicAdd(&scount, 1);
      if (k < CAP) { sp[k] = pi; st[k] = 1.0f / zi; }
    }
  };
  int done = 0;
  if (sml_vec_ok(p, z, p, HW)) {
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const int n4 = HW >> 2;
    for (int i = threadIdx.x; i < n4; i += 2 * T) {
      const int i1 = i + T < n4 ? i + T : i;
      const float4 za = z4[i], pa = p4[i], zb = z4[i1], pb = p4[i1];
      keep(za.x, pa.x); keep(za.y, pa.y); keep(za.z, pa.z); keep(za.w, pa.w);
      if (i + T < n4) { keep(zb.x, pb.x); keep(zb.y, pb.y); keep(zb.z, pb.z); keep(zb.w, pb.w); }
    }
    done = HW;
  }
  for (int i = done + threadIdx.x; i < HW; i += T) keep(z[i], p[i]);
  __syncthreads();
  const int cnt = scount;
  const bool in_lds = cnt <= CAP;
  auto slope = [&](float s) {  // the sum is order-independent up to double rounding: compaction order does not matter
    double g = 0.0;
    if (in_lds) {
      for (int i = threadIdx.x; i < cnt; i += T) {
        float r = s * sp[i] - st[i];
        g += r > 0.f ? (double)sp[i] : (r < 0.f ? -(double)sp[i] : 0.0);
      }
    } else {
      for (int i = threadIdx.x; i < HW; i += T) {
        float zi = z[i];
        if (zi < dmax && zi > dmin) {
          float r = s * p[i] - 1.0f / zi;
          g += r > 0.f ? (double)p[i] : (r < 0.f ? -(double)p[i] : 0.0);
        }
      }
    }
    return block_sum_d(g, sh);
  };
  float a = lo, c = hi, res;
  if (cnt == 0) res = 0.29999601510536417f * (hi / 0.3f);
  else if (slope(a) >= 0.0) res = a;
  else if (slope(c) <= 0.0) res = c;
  else {
    for (int it = 0; it < 50; it++) {
      float m = 0.5f * (a + c);
      if (m == a || m == c) break;      // a and c are neighbouring floats: the remaining steps would change nothing (same result, ~25 block sums less)
      if (slope(m) < 0.0) a = m; else c = m;
    }
    res = c;
  }
  if (threadIdx.x == 0) { scale[b] = res; nvalid[b] = cnt; }
}

// 'st' global alignment (modules/estimator.py:5-29 compute_scale_and_shift_ls, LeastSquaresEstimator :90-118): closed-form least squares
// of scale * mono + shift against the inverse radar depth over the valid radar pixels; the 2x2 normal equations are accumulated in
// double (the reference sums float32 arrays with numpy's pairwise fp32 sum: same quantities, ~1e-6 apart); singular -> (0, 0).
__global__ __launch_bounds__(SML_SCAN_T) void sml_scale_shift_ls_kernel(const float* __restrict__ mono, const float* __restrict__ sparse, int HW,
                                                                 float dmin, float dmax, float* __restrict__ scale,
                                                                 float* __restrict__ shift, int* __restrict__ nvalid) {
  __shared__ double sh[SML_SCAN_T / 64];
  const int b = blockIdx.x, T = blockDim.x;
  const float* p = mono + (int64_t)b * HW;
  const float* z = sparse + (int64_t)b * HW;
  double a00 = 0.0, a01 = 0.0, a11 = 0.0, b0 = 0.0, b1 = 0.0;
  auto add = [&](float zi, float pf) {
    if (zi < dmax && zi > dmin) {
      const double pi = (double)pf, ti = (double)(1.0f / zi);
      a00 += pi * pi; a01 += pi; a11 += 1.0; b0 += pi * ti; b1 += ti;
    }
  };
  int done = 0;
  if (sml_vec_ok(p, z, p, HW)) {
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const float4* p4 = reinterpret_cast<const float4*>(p);
    const int n4 = HW >> 2;
    for (int i = threadIdx.x; i < n4; i += 2 * T) {
      const int i1 = i + T < n4 ? i + T : i;
      const float4 za = z4[i], pa = p4[i], zb = z4[i1], pb = p4[i1];
      add(za.x, pa.x); add(za.y, pa.y); add(za.z, pa.z); add(za.w, pa.w);
      if (i + T < n4) { add(zb.x, pb.x); add(zb.y, pb.y); add(zb.z, pb.z); add(zb.w, pb.w); }
    }
    done = HW;
  }
  for (int i = done + threadIdx.x; i < HW; i += T) add(z[i], p[i]);
  a00 = block_sum_d(a00, sh); a01 = block_sum_d(a01, sh); a11 = block_sum_d(a11, sh);
  b0 = block_sum_d(b0, sh); b1 = block_sum_d(b1, sh);
  if (threadIdx.x == 0) {
    const double det = a00 * a11 - a01 * a01;
    double x0 = 0.0, x1 = 0.0;
    if (det > 0.0) { x0 = (a11 * b0 - a01 * b1) / det; x1 = (-a01 * b0 + a00 * b1) / det; }
    scale[b] = (float)x0; shift[b] = (float)x1; nvalid[b] = (int)a11;
  }
}

// int_depth = clamp(s*mono + shift), int_scales (1 / rcnet / radar override) -> per-sample min & max of int_scales
__device__ __forceinline__ float int_depth_of(float s, float mono, float hi, float lo, float sft = 0.f) {
  float v = s * mono + sft;
  if (hi > 0.f && v > hi) v = hi;
  if (lo > 0.f && v < lo) v = lo;
  return v;
}
__device__ __forceinline__ float int_scale_of(float idp, float radar, float rc, float dmin, float dmax, int use_rcnet) {
  float sc = 1.f;
  if (use_rcnet && rc < dmax && rc > dmin) sc = (1.f / rc) / idp;
  if (radar < dmax && radar > dmin) sc = (1.f / radar) / idp;
  return sc;
}
__global__ __launch_bounds__(SML_SCAN_T) void sml_scales_minmax_kernel(const float* __restrict__ mono, const float* __restrict__ sparse,
                                                                const float* __restrict__ rcnet, const float* __restrict__ scale,
                                                                const float* __restrict__ shift, int HW,
                                                                float dmin, float dmax, float hi, float lo, int use_rcnet,
                                                                float* __restrict__ mm /* [B][3] = min, max, nvalid(radar+rcnet) */) {
  constexpr int NW = SML_SCAN_T / 64;
  __shared__ float smin[NW], smax[NW], scnt[NW];
  const int b = blockIdx.x, T = blockDim.x;
  const float s = scale[b], sft = shift ? shift[b] : 0.f;
  const float* pm = mono + (int64_t)b * HW;
  const float* ps = sparse + (int64_t)b * HW;
  const float* pr = use_rcnet ? rcnet + (int64_t)b * HW : pm;     // not read without use_rcnet; a valid pointer for the alignment test
  float mn = INFINITY, mx = -INFINITY, cn = 0.f;      // min / max / an integer count in fp32: independent of the visiting order
  auto add = [&](float m, float sp_, float rc) {
    const float v = int_scale_of(int_depth_of(s, m, hi, lo, sft), sp_, rc, dmin, dmax, use_rcnet);
    mn = fminf(mn, v); mx = fmaxf(mx, v);
    cn += (sp_ < dmax && sp_ > dmin) ? 1.f : 0.f;
    cn += (use_rcnet && rc < dmax && rc > dmin) ? 1.f : 0.f;
  };
  int done = 0;
  if (sml_vec_ok(pm, ps, pr, HW)) {
    const float4* m4 = reinterpret_cast<const float4*>(pm);
    const float4* s4 = reinterpret_cast<const float4*>(ps);
    const float4* r4 = reinterpret_cast<const float4*>(pr);
    const int n4 = HW >> 2;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = threadIdx.x; i < n4; i += 2 * T) {
      const int i1 = i + T < n4 ? i + T : i;
      const float4 ma = m4[i], sa = s4[i], mb = m4[i1], sb = s4[i1];
      const float4 ra = use_rcnet ? r4[i] : zero, rb = use_rcnet ? r4[i1] : zero;
      add(ma.x, sa.x, ra.x); add(ma.y, sa.y, ra.y); add(ma.z, sa.z, ra.z); add(ma.w, sa.w, ra.w);
      if (i + T < n4) { add(mb.x, sb.x, rb.x); add(mb.y, sb.y, rb.y); add(mb.z, sb.z, rb.z); add(mb.w, sb.w, rb.w); }
    }
    done = HW;
  }
  for (int i = done + threadIdx.x; i < HW; i += T) add(pm[i], ps[i], use_rcnet ? pr[i] : 0.f);
  mn = wave_min(mn); mx = wave_max(mx); cn = wave_sum(cn);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { smin[wv] = mn; smax[wv] = mx; scnt[wv] = cn; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = smin[0], c = smax[0], n = scnt[0];
    for (int w = 1; w < (T >> 6); w++) { a = fminf(a, smin[w]); c = fmaxf(c, smax[w]); n += scnt[w]; }
    mm[b * 3 + 0] = a; mm[b * 3 + 1] = c; mm[b * 3 + 2] = n;
  }
}
// network input x (B,h,w,3) NHWC = [(int_depth-m0)/s0, (int_scales_n-m1)/s1, gray]; d (B,h,w) = int_depth;
// cv2.INTER_NEAREST source index = min(floor(dst * src/dst_size), src-1)
__global__ __launch_bounds__(256) void sml_build_inputs_kernel(const float* __restrict__ image /* B,3,H,W */, const float* __restrict__ mono,
                                                               const float* __restrict__ sparse, const float* __restrict__ rcnet,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               const float* __restrict__ mm, int B, int H, int W, int h, int w, float dmin, float dmax, float hi, float lo,
                                                               int use_rcnet, float m0, float s0, float m1, float s1,
                                                               float* __restrict__ x, float* __restrict__ d) {
  const int64_t total = (int64_t)B * h * w;
  const double fy = (double)H / (double)h, fx = (double)W / (double)w;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int xx = (int)(i % w); int64_t q = i / w; int yy = (int)(q % h); int b = (int)(q / h);
    int sy = min((int)floor((double)yy * fy), H - 1), sx = min((int)floor((double)xx * fx), W - 1);
    int64_t j = ((int64_t)b * H + sy) * W + sx;
    float idp = int_depth_of(scale[b], mono[j], hi, lo, shift ? shift[b] : 0.f);
    float rc = use_rcnet ? rcnet[j] : 0.f;
    float sc = int_scale_of(idp, sparse[j], rc, dmin, dmax, use_rcnet);
    float mn = mm[b * 3], mx = mm[b * 3 + 1];
    if (mm[b * 3 + 2] > 1.f && (mx - mn) > 2.220446049250313e-16f) sc = (sc - mn) / (mx - mn);
    const float* im = image + (int64_t)b * 3 * H * W + (int64_t)sy * W + sx;
    float gray = im[0] * 0.299f + im[(int64_t)H * W] * 0.587f + im[(int64_t)2 * H * W] * 0.114f;
    x[i * 3 + 0] = (idp - m0) / s0;
    x[i * 3 + 1] = (sc - m1) / s1;
    x[i * 3 + 2] = gray;
    d[i] = idp;
  }
}

// ---- S9: outlier removal -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void max_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float sm[4];
  float m = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = fmaxf(m, x[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}
__global__ __launch_bounds__(256) void outlier_removal_kernel(const float* __restrict__ depth, const float* __restrict__ partial, int nparts,
                                                              float* __restrict__ out, int N, int H, int W, int k, float thr) {
  // the maximum over the partials, by the block (a max is order independent): every thread walking all <= 256 partials on its own was
  // most of this kernel's 55 us
  __shared__ float smx[4];
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < nparts; i += 256) mx = fmaxf(mx, partial[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((threadIdx.x & 63) == 0) smx[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  const float fill = 10.f * mx;
  const int r = k / 2;
  const int64_t total = (int64_t)N * H * W;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int w = (int)(i % W); int64_t q = i / W; int h = (int)(q % H); int n = (int)(q / H);
    float mn = INFINITY;
    for (int dy = -r; dy <= r; dy++)
      for (int dx = -r; dx <= r; dx++) {
        int hh = h + dy, ww = w + dx;
        float v = fill;
        if ((unsigned)hh < (unsigned)H && (unsigned)ww < (unsigned)W) {
          float dd = depth[((int64_t)n * H + hh) * W + ww];
          v = dd > 0.f ? dd : fill;  // validity_map <= 0 -> filled
        }
        mn = fminf(mn, v);
      }
    float dv = depth[i];
    out[i] = (mn < dv - thr) ? 0.f : dv;
  }
}

// ---- S10/S11: loss ----------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sobel_gx(int u, int v, int fs) {
  int c = fs / 2;
  if (v == c) return 0.f;
  float m = (u == c && (v == c - 1 || v == c + 1)) ? 2.f : 1.f;
  return v < c ? m : -m;
}
__device__ __forceinline__ float sobel_gy(int u, int v, int fs) {
  int c = fs / 2;
  if (u == c) return 0.f;
  float m = (v == c && (u == c - 1 || u == c + 1)) ? 2.f : 1.f;
  return u < c ? m : -m;
}
__device__ __forceinline__ float sgn(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// per pixel: masked L1 terms and Sobel terms; block partials[blk][8]; gfx/gfy = weights * w_{x,y} * sign(pred_{dx,dy})
// FS: compile-time Sobel size (3 / 5 / 7; 0 = the run-time `fs_rt`): with it the tap loops unroll and their 2 * FS^2 + 9 loads -- always
// in bounds, the indices are clamped -- issue as one batch; as run-time loops every tap was its own memory round trip.
// supervised term of one pixel and its derivative: 'l1' |d|, 'l2' d^2 (F.mse_loss), 'smoothl1' 0.5 d^2 below 1, |d| - 0.5 above (F.smooth_l1_loss, beta 1)
__device__ __forceinline__ float sml_loss_term(float d, int kind) {
  const float a = fabsf(d);
  return kind == 0 ? a : (kind == 1 ? d * d : (a < 1.f ? 0.5f * d * d : a - 0.5f));
}
__device__ __forceinline__ float sml_loss_dterm(float d, int kind) {
  return kind == 0 ? sgn(d) : (kind == 1 ? 2.f * d : (fabsf(d) < 1.f ? d : sgn(d)));
}
template <int FS>
__global__ __launch_bounds__(256) void sml_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ image,
                                                           const float* __restrict__ gt_interp, const float* __restrict__ gt_sparse,
                                                           const float* __restrict__ weights, int N, int H, int W, int fs_rt, int flags,
                                                           float edge_ratio, float* __restrict__ gfx, float* __restrict__ gfy, double* __restrict__ partial) {
  // flags: bit 0 = mask the interpolated ground truth where sparse lidar exists (loss.py:26-33); bits 1-2 = loss_func 0 'l1', 1 'l2', 2 'smoothl1'
  // (utils/loss.py:55-100).  edge_ratio = w_edge / w_smoothness: the gradient fields then carry the edge-matching term too (loss.py:241-249).
  const int mask_interp = flags & 1, kind = (flags >> 1) & 3;
  __shared__ double sh[4];
  const int64_t total = (int64_t)N * H * W;
  const int fs = FS ? FS : fs_rt;
  const int r = fs / 2;
  double acc[8];
  for (int j = 0; j < 8; j++) acc[j] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int w = (int)(i % W); int64_t q = i / W; int h = (int)(q % H); int n = (int)(q / H);
    const float* P = pred + (int64_t)n * H * W; const float* I = image + (int64_t)n * H * W;
    float o = P[(int64_t)h * W + w];
    float gs = gt_sparse[i], gi = gt_interp[i];
    if (mask_interp && gs > 0.f) gi = 0.f;
    if (gi > 0.f) { acc[0] += sml_loss_term(o - gi, kind); acc[1] += 1.0; }
    if (gs > 0.f) { acc[2] += sml_loss_term(o - gs, kind); acc[3] += 1.0; }
    float pdx = 0.f, pdy = 0.f, idx = 0.f, idy = 0.f;
#pragma unroll
    for (int u = 0; u < fs; u++) {
      int hh = min(max(h + u - r, 0), H - 1);
#pragma unroll
      for (int v = 0; v < fs; v++) {
        int ww = min(max(w + v - r, 0), W - 1);
        float gx = sobel_gx(u, v, fs), gy = sobel_gy(u, v, fs);
        float pv = P[(int64_t)hh * W + ww], iv = I[(int64_t)hh * W + ww];
        pdx += pv * gx; pdy += pv * gy; idx += iv * gx; idy += iv * gy;
      }
    }
    float sdx = 0.f, sdy = 0.f;
#pragma unroll
    for (int u = 0; u < 3; u++) {
      int hh = min(max(h + u - 1, 0), H - 1);
#pragma unroll
      for (int v = 0; v < 3; v++) {
        int ww = min(max(w + v - 1, 0), W - 1);
        float iv = I[(int64_t)hh * W + ww];
        sdx += iv * sobel_gx(u, v, 3); sdy += iv * sobel_gy(u, v, 3);
      }
    }
    float wt = weights ? weights[i] : 1.f;
    float wx = wt * __expf(-fabsf(sdy)), wy = wt * __expf(-fabsf(sdx));  // x-term weighted by the image's y-gradient (loss.py:235-239)
    acc[4] += wx * fabsf(pdx); acc[5] += wy * fabsf(pdy);
    acc[6] += wt * fabsf(fabsf(pdx) - fabsf(idx)); acc[7] += wt * fabsf(fabsf(pdy) - fabsf(idy));
    float fx = wx * sgn(pdx), fy = wy * sgn(pdy);
    if (edge_ratio != 0.f) {      // d | |p'| - |i'| | / d p' = sgn(|p'| - |i'|) sgn(p'), weighted like the loss term
      fx += edge_ratio * wt * sgn(fabsf(pdx) - fabsf(idx)) * sgn(pdx);
      fy += edge_ratio * wt * sgn(fabsf(pdy) - fabsf(idy)) * sgn(pdy);
    }
    gfx[i] = fx; gfy[i] = fy;
  }
  for (int j = 0; j < 8; j++) {
    double s = block_sum_d(acc[j], sh);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * 8 + j] = s;
  }
}
// info = [loss, supervised, lidar, smoothness, edge, n_interp, n_lidar]
__global__ void sml_loss_finalize_kernel(const double* __restrict__ partial, int rows, double npix, int fs, float w_lidar, float w_smooth,
                                         float w_edge, float* __restrict__ info) {
  // one wave: lanes stride over the rows, double-precision xor tree (fixed order)
  const int lane = threadIdx.x;
  double a[8];
  for (int j = 0; j < 8; j++) {
    double v = 0.0;
    for (int r = lane; r < rows; r += 64) v += partial[(int64_t)r * 8 + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    a[j] = v;
  }
  if (lane) return;
  double sup = a[0] / a[1], lid = w_lidar > 0.f ? a[2] / a[3] : 0.0;
  double sm = (a[4] / npix + a[5] / npix) / (double)(fs * fs), ed = (a[6] / npix + a[7] / npix) / (double)(fs * fs);
  if (!(w_smooth > 0.f || w_edge > 0.f)) { sm = 0.0; ed = 0.0; }
  info[0] = (float)(sup + w_lidar * lid + w_smooth * sm + w_edge * ed);
  info[1] = (float)sup; info[2] = (float)lid; info[3] = (float)sm; info[4] = (float)ed; info[5] = (float)a[1]; info[6] = (float)a[3];
}
// d loss / d pred: L1 signs + transposed Sobel through the replicate padding (gather over padded positions clamping to p)
template <int FS>
__global__ __launch_bounds__(256) void sml_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ gt_interp,
                                                           const float* __restrict__ gt_sparse, const float* __restrict__ gfx,
                                                           const float* __restrict__ gfy, const float* __restrict__ info,
                                                           const float* __restrict__ dloss, int N, int H, int W, int fs_rt, int flags,
                                                           float w_lidar, float w_smooth, float* __restrict__ dpred) {
  const int mask_interp = flags & 1, kind = (flags >> 1) & 3;
  const int64_t total = (int64_t)N * H * W;
  const int fs = FS ? FS : fs_rt;
  const int r = fs / 2;
  const float gl = dloss[0];
  const float c_sup = 1.f / info[5], c_lid = w_lidar > 0.f ? w_lidar / info[6] : 0.f;
  const float c_sm = w_smooth / ((float)total * (float)(fs * fs));
  const int64_t gstride = (int64_t)gridDim.x * blockDim.x, gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // Interior pixels (no padded position clamps onto them, every tap lies inside the map) take a path without clamps, masks or 64-bit
  // index arithmetic per tap -- the same products in the same order; the 2r-wide frame goes through the general gather below, compacted
  // into its own waves (as one loop over all pixels the general form cost 193 us per SML step: ~20 vector instructions per tap).
  const int Hi = H - 2 * r, Wi = W - 2 * r;
  const bool split = FS > 0 && Hi > 0 && Wi > 0 && w_smooth > 0.f;
  if (split) {
    const int64_t nint = (int64_t)N * Hi * Wi;
    for (int64_t j = gtid; j < nint; j += gstride) {
      const int wi = (int)(j % Wi); const int64_t q = j / Wi; const int hi = (int)(q % Hi); const int n = (int)(q / Hi);
      const int64_t i = ((int64_t)n * H + hi + r) * W + wi + r;
      float o = pred[i], gs = gt_sparse[i], gi = gt_interp[i];
      if (mask_interp && gs > 0.f) gi = 0.f;
      float g = 0.f;
      if (gi > 0.f) g += c_sup * sml_loss_dterm(o - gi, kind);
      if (gs > 0.f) g += c_lid * sml_loss_dterm(o - gs, kind);
      const float* FX = gfx + i + (int64_t)r * W + r; const float* FY = gfy + i + (int64_t)r * W + r;      // tap (u, v) reads (h - u + r, w - v + r)
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < fs; u++) {
        const float* rx = FX - u * W; const float* ry = FY - u * W;      // one pointer per filter row, the column is an immediate offset
#pragma unroll
        for (int v = 0; v < fs; v++) {
          const float fx = rx[-v], fy = ry[-v];
          s += fx * sobel_gx(u, v, fs) + fy * sobel_gy(u, v, fs);
        }
      }
      g += c_sm * s;
      dpred[i] = gl * g;
    }
  }
  const int64_t E = split ? (int64_t)2 * r * W + (int64_t)Hi * 2 * r : (int64_t)H * W;      // frame pixels per image (all of them when not split)
  for (int64_t e = gtid; e < (int64_t)N * E; e += gstride) {
    int n = (int)(e / E), h, w;
    {
      const int64_t ei = e - (int64_t)n * E;
      if (!split) { h = (int)(ei / W); w = (int)(ei - (int64_t)h * W); }
      else if (ei < (int64_t)2 * r * W) { const int row = (int)(ei / W); h = row < r ? row : H - 2 * r + row; w = (int)(ei - (int64_t)row * W); }
      else { const int64_t e2 = ei - (int64_t)2 * r * W; const int cc = (int)(e2 % (2 * r)); h = (int)(e2 / (2 * r)) + r; w = cc < r ? cc : W - 2 * r + cc; }
    }
    const int64_t i = ((int64_t)n * H + h) * W + w;
    float o = pred[i], gs = gt_sparse[i], gi = gt_interp[i];
    if (mask_interp && gs > 0.f) gi = 0.f;
    float g = 0.f;
    if (gi > 0.f) g += c_sup * sml_loss_dterm(o - gi, kind);
    if (gs > 0.f) g += c_lid * sml_loss_dterm(o - gs, kind);
    if (w_smooth > 0.f) {
      // padded rows that clamp onto h: h itself, plus -r..-1 when h == 0, plus H..H+r-1 when h == H-1 (same for columns)
      int ph0 = h == 0 ? -r : h, ph1 = h == H - 1 ? H - 1 + r : h;
      int pw0 = w == 0 ? -r : w, pw1 = w == W - 1 ? W - 1 + r : w;
      const float* FX = gfx + (int64_t)n * H * W; const float* FY = gfy + (int64_t)n * H * W;
      float s = 0.f;
      for (int ph = ph0; ph <= ph1; ph++)
        for (int pw = pw0; pw <= pw1; pw++)
#pragma unroll
          for (int u = 0; u < fs; u++) {
            const int a = ph - u + r;  // output row whose tap u reads padded row ph
            const bool aok = (unsigned)a < (unsigned)H;
            const int ac = min(max(a, 0), H - 1);
#pragma unroll
            for (int v = 0; v < fs; v++) {
              const int b = pw - v + r;
              const bool ok = aok && (unsigned)b < (unsigned)W;
              const int bc = min(max(b, 0), W - 1);
              // unconditional loads (clamped), masked sum: the taps of a pixel are requested together
              const float fx = FX[(int64_t)ac * W + bc], fy = FY[(int64_t)ac * W + bc];
              s += ok ? fx * sobel_gx(u, v, fs) + fy * sobel_gy(u, v, fs) : 0.f;
            }
          }
      g += c_sm * s;
    }
    dpred[i] = gl * g;
  }
}

// ---- S12: validation ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void cubic_w(float t, float (&w)[4]) {
  const float A = -0.75f;
  float x = t + 1.f; w[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t; w[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t; w[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t; w[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__global__ __launch_bounds__(256) void bicubic_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int OH,
                                                      int OW) {
  const int64_t total = (int64_t)N * OH * OW;
  const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int ow = (int)(i % OW); int64_t q = i / OW; int oh = (int)(q % OH); int n = (int)(q / OH);
    float fy = sh * ((float)oh + 0.5f) - 0.5f, fx = sw * ((float)ow + 0.5f) - 0.5f;
    int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - (float)iy, wy); cubic_w(fx - (float)ix, wx);
    const float* b = x + (int64_t)n * H * W;
    float acc = 0.f;
    for (int u = 0; u < 4; u++) {
      int yy = min(max(iy - 1 + u, 0), H - 1);
      float row = 0.f;
      for (int v = 0; v < 4; v++) row += wx[v] * b[(int64_t)yy * W + min(max(ix - 1 + v, 0), W - 1)];
      acc += wy[u] * row;
    }
    y[i] = acc;
  }
}
// per image n: sums over the mask (gt > 0 & dmin < gt < dmax) of the eval_utils terms; out[n][8] =
// [count, sum|1000(o-g)|, sum(1000(o-g))^2, sum|1/(.001g) - 1/(.001o)|, sum(.)^2, sum|o-g|/g, sum 1000(o-g)^2/g, count(max(o/g,g/o)<1.25)]
__global__ __launch_bounds__(256) void depth_metrics_kernel(const float* __restrict__ out, const float* __restrict__ gt, int HW, float dmin,
                                                            float dmax, double* __restrict__ res) {
  __shared__ double sh[4];
  const int n = blockIdx.x;
  double a[8];
  for (int j = 0; j < 8; j++) a[j] = 0.0;
  for (int i = threadIdx.x; i < HW; i += 256) {
    float g = gt[(int64_t)n * HW + i], o = out[(int64_t)n * HW + i];
    if (g > 0.f && g > dmin && g < dmax) {
      double e = 1000.0 * (double)o - 1000.0 * (double)g;
      double ie = 1.0 / (0.001 * (double)g) - 1.0 / (0.001 * (double)o);
      a[0] += 1.0; a[1] += fabs(e); a[2] += e * e; a[3] += fabs(ie); a[4] += ie * ie;
      a[5] += fabs(e) / (1000.0 * (double)g); a[6] += e * e / (1000.0 * (double)g);
      double rr = (double)o / (double)g, r2 = (double)g / (double)o;
      a[7] += (rr > r2 ? rr : r2) < 1.25 ? 1.0 : 0.0;
    }
  }
  for (int j = 0; j < 8; j++) {
    double s = block_sum_d(a[j], sh);
    if (threadIdx.x == 0) res[n * 8 + j] = s;
  }
}

// ---- launchers ---------------------------------------------------------------------------------------------------------------
void launch_sml_scale_align(const float* mono, const float* sparse, int B, int HW, float dmin, float dmax, float lo, float hi, float* scale,
                            int* nvalid, hipStream_t st) {
  hipLaunchKernelGGL(sml_scale_align_kernel, dim3(B), dim3(SML_SCAN_T), 0, st, mono, sparse, HW, dmin, dmax, lo, hi, scale, nvalid);
}
void launch_sml_scale_shift_ls(const float* mono, const float* sparse, int B, int HW, float dmin, float dmax, float* scale, float* shift,
                               int* nvalid, hipStream_t st) {
  hipLaunchKernelGGL(sml_scale_shift_ls_kernel, dim3(B), dim3(SML_SCAN_T), 0, st, mono, sparse, HW, dmin, dmax, scale, shift, nvalid);
}
void launch_sml_build_inputs(const float* image, const float* mono, const float* sparse, const float* rcnet, const float* scale,
                             const float* shift, float* mm, int B, int H, int W, int h, int w, float dmin, float dmax, float hi, float lo,
                             int use_rcnet, float m0, float s0, float m1, float s1, float* x, float* d, hipStream_t st) {
  hipLaunchKernelGGL(sml_scales_minmax_kernel, dim3(B), dim3(SML_SCAN_T), 0, st, mono, sparse, rcnet, scale, shift, H * W, dmin, dmax, hi, lo, use_rcnet, mm);
  hipLaunchKernelGGL(sml_build_inputs_kernel, dim3(ew_grid((int64_t)B * h * w)), dim3(256), 0, st, image, mono, sparse, rcnet, scale, shift, mm, B,
                     H, W, h, w, dmin, dmax, hi, lo, use_rcnet, m0, s0, m1, s1, x, d);
}
int outlier_parts(int64_t n) { return (int)ew_grid(n, 256); }
void launch_outlier_removal(const float* depth, float* partial, float* out, int N, int H, int W, int k, float thr, hipStream_t st) {
  int64_t n = (int64_t)N * H * W;
  int parts = outlier_parts(n);
  hipLaunchKernelGGL(max_partial_kernel, dim3(parts), dim3(256), 0, st, depth, n, partial);
  hipLaunchKernelGGL(outlier_removal_kernel, dim3(ew_grid(n)), dim3(256), 0, st, depth, partial, parts, out, N, H, W, k, thr);
}
int sml_loss_rows(int64_t n) { return (int)ew_grid(n, 512); }
void launch_sml_loss_fwd(const float* pred, const float* image, const float* gi, const float* gs, const float* weights, int N, int H, int W,
                         int fs, int mask_interp, float w_lidar, float w_smooth, float w_edge, float* gfx, float* gfy, double* partial,
                         float* info, hipStream_t st) {
  int64_t n = (int64_t)N * H * W;
  int rows = sml_loss_rows(n);
  const float edge_ratio = (w_edge > 0.f && w_smooth > 0.f) ? w_edge / w_smooth : 0.f;      // (mask_interp carries the loss kind in bits 1-2: rd_sml_loss_fwd_kind)
#define RD_LF(F) hipLaunchKernelGGL((sml_loss_fwd_kernel<F>), dim3(rows), dim3(256), 0, st, pred, image, gi, gs, weights, N, H, W, fs, mask_interp, edge_ratio, gfx, gfy, partial)
  if (fs == 7) RD_LF(7); else if (fs == 5) RD_LF(5); else if (fs == 3) RD_LF(3); else RD_LF(0);
#undef RD_LF
  hipLaunchKernelGGL(sml_loss_finalize_kernel, dim3(1), dim3(64), 0, st, partial, rows, (double)n, fs, w_lidar, w_smooth, w_edge, info);
}
void launch_sml_loss_bwd(const float* pred, const float* gi, const float* gs, const float* gfx, const float* gfy, const float* info,
                         const float* dloss, int N, int H, int W, int fs, int mask_interp, float w_lidar, float w_smooth, float* dpred,
                         hipStream_t st) {
#define RD_LB(F) hipLaunchKernelGGL((sml_loss_bwd_kernel<F>), dim3(ew_grid((int64_t)N * H * W)), dim3(256), 0, st, pred, gi, gs, gfx, gfy, info, dloss, N, H, W, fs, \
                                    mask_interp, w_lidar, w_smooth, dpred)
  if (fs == 7) RD_LB(7); else if (fs == 5) RD_LB(5); else if (fs == 3) RD_LB(3); else RD_LB(0);
#undef RD_LB
}
void launch_bicubic(const float* x, float* y, int N, int H, int W, int OH, int OW, hipStream_t st) {
  hipLaunchKernelGGL(bicubic_kernel, dim3(ew_grid((int64_t)N * OH * OW)), dim3(256), 0, st, x, y, N, H, W, OH, OW);
}
void launch_depth_metrics(const float* out, const float* gt, int N, int HW, float dmin, float dmax, double* res, hipStream_t st) {
  hipLaunchKernelGGL(depth_metrics_kernel, dim3(N), dim3(256), 0, st, out, gt, HW, dmin, dmax, res);
}

}  // namespace rd
