// Fused flat-arena Adam (one launch for every parameter of a model).
//
// Reference: torch.optim.Adam as constructed at RCNet/rcnet_main.py:233-238 and train_zju.py:205-211
// (betas (0.9, 0.999), eps 1e-8, weight_decay 0 -> L2-coupled form, no amsgrad).  The arithmetic follows
// torch's single-tensor Adam: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).  28 algorithmic bytes per parameter.
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

// fp16 mode (static loss scale): any non-finite scaled gradient raises flag[0]; flag[1] counts the steps that were skipped because of it.
// One pass over the gradient arena (4 bytes per parameter) on top of Adam's 28.
__global__ __launch_bounds__(256) void grad_finite_kernel(const float* __restrict__ g, int64_t n, int* __restrict__ flag) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float gg[4];
    ld4(g + (i << 2), gg);
#pragma unroll
    for (int e = 0; e < 4; e++) bad |= !(fabsf(gg[e]) <= 3.4028234e38f);      // false for inf and NaN
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) bad |= !(fabsf(g[i]) <= 3.4028234e38f);
  if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicAdd(flag, 1);      // any non-zero value raises the flag
}
__global__ void adam_skip_count_kernel(int* __restrict__ flag) {   // after the guarded Adam launches of a step: count the skip, clear the flag
  if (flag[0]) { flag[1] += 1; flag[0] = 0; }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                   float wd, float bc1, float bc2_sqrt, float gscale, const int* __restrict__ skip) {
  if (skip && *skip) return;      // a non-finite gradient somewhere in the arena: parameters and moments stay untouched (uniform branch)
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float step = lr / bc1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float pp[4], gg[4], mm[4], vv[4];
    ld4(p + (i << 2), pp); ld4(g + (i << 2), gg); ld4(m + (i << 2), mm); ld4(v + (i << 2), vv);
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float gr = gg[e] * gscale + wd * pp[e];
      mm[e] = b1 * mm[e] + (1.f - b1) * gr;
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
      pp[e] -= step * mm[e] / (sqrtf(vv[e]) / bc2_sqrt + eps);
    }
    st4(p + (i << 2), pp); st4(m + (i << 2), mm); st4(v + (i << 2), vv);
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float gr = g[i] * gscale + wd * p[i];
    float mi = b1 * m[i] + (1.f - b1) * gr, vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

void launch_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                 float bc1, float bc2_sqrt, float gscale, hipStream_t st, const int* skip) {
  unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 1024), 2048));
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, bc1, bc2_sqrt, gscale, skip);
}
void launch_grad_finite(const float* g, int64_t n, int* flag, hipStream_t st) {
  unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 1024), 2048));
  hipLaunchKernelGGL(grad_finite_kernel, dim3(grid), dim3(256), 0, st, g, n, flag);
}
void launch_adam_skip_count(int* flag, hipStream_t st) { hipLaunchKernelGGL(adam_skip_count_kernel, dim3(1), dim3(1), 0, st, flag); }

}  // namespace rd
