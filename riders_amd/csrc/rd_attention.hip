// Linear attention  phi(Q) (phi(K)^T V)  on MFMA with LDS-staged per-head tiles (fp32 accumulate).
//
// Reference: RCNet/linear_attention.py:18-45 (LinearAttention.forward, masks are always None on this path)
//   Q = elu(q)+1, K = elu(k)+1, V = v / S
//   KV[n,h]   = sum_s K[n,s,h,:]^T V[n,s,h,:]            (D x D,  D = 16)
//   Z[n,l,h]  = 1 / (Q[n,l,h,:] . sum_s K[n,s,h,:] + eps)
//   out       = (Q KV) * Z * S
// One wave owns one (n, h) pair: the three 21x16 head slices are staged in LDS (rows padded to 32 with zeros),
// the two contractions K^T V (16x16x32) and Q KV (32x16x16) run on v_mfma_f32_16x16x4_f32 (exact fp32),
// the normaliser on the VALU.  The backward recomputes KV / P from q,k,v (nothing but q,k,v is saved).
#include "rd_attention_head.h"

namespace rd {

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void linear_attention_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ v, const T* __restrict__ dout,
                                                               T* __restrict__ out, T* __restrict__ dq, T* __restrict__ dk,
                                                               T* __restrict__ dv, int N, int L, int S, int H, int ldq,
                                                               int ldk, int ldv, int ldo, float eps) {
  __shared__ AttnSmem sm[4];
  const int wv = threadIdx.x >> 6;
  const int pair = blockIdx.x * 4 + wv;
  const bool active = pair < N * H;
  const int n = active ? pair / H : 0, h = active ? pair % H : 0;
  attn_head<T, BWD>(sm[wv], q, k, v, dout, out, dq, dk, dv, n, h, active, L, S, ldq, ldk, ldv, ldo, eps, []() {});
}

void launch_linear_attention_fwd(const void* q, const void* k, const void* v, void* out, int N, int L, int S, int H, int ldq,
                                 int ldk, int ldv, int ldo, float eps, int dtype, hipStream_t st) {
  unsigned grid = (unsigned)cdiv((int64_t)N * H, 4);
  if (grid == 0) return;
  if (dtype == 0)
    hipLaunchKernelGGL((linear_attention_kernel<float, false>), dim3(grid), dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)nullptr, (float*)out, (float*)nullptr, (float*)nullptr, (float*)nullptr, N, L, S, H, ldq, ldk, ldv, ldo, eps);
  else
    hipLaunchKernelGGL((linear_attention_kernel<bf16_t, false>), dim3(grid), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)nullptr, (bf16_t*)out, (bf16_t*)nullptr, (bf16_t*)nullptr, (bf16_t*)nullptr, N, L, S, H, ldq, ldk, ldv, ldo, eps);
}
void launch_linear_attention_bwd(const void* q, const void* k, const void* v, const void* dout, void* dq, void* dk, void* dv,
                                 int N, int L, int S, int H, int ldq, int ldk, int ldv, int ldo, float eps, int dtype,
                                 hipStream_t st) {
  unsigned grid = (unsigned)cdiv((int64_t)N * H, 4);
  if (grid == 0) return;
  if (dtype == 0)
    hipLaunchKernelGGL((linear_attention_kernel<float, true>), dim3(grid), dim3(256), 0, st, (const float*)q, (const float*)k, (const float*)v, (const float*)dout, (float*)nullptr, (float*)dq, (float*)dk, (float*)dv, N, L, S, H, ldq, ldk, ldv, ldo, eps);
  else
    hipLaunchKernelGGL((linear_attention_kernel<bf16_t, true>), dim3(grid), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)dout, (bf16_t*)nullptr, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, N, L, S, H, ldq, ldk, ldv, ldo, eps);
}

}  // namespace rd
