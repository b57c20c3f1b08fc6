// BatchNorm over WIDE layers on SMALL maps as ONE launch per direction: a workgroup owns one 16-byte channel vector (8 channels of the 16-bit
// types, 4 of fp32) of EVERY pixel, so the batch statistics it needs are its own -- no partial rows, no finalize launch, no second pass.
//
// Reference: torch.nn.BatchNorm2d in training mode + ReLU6 inside the inverted-residual blocks of the Scale Map Learner's EfficientNet-Lite3
// backbone (modules/midas/midas_net_custom.py:84-111 via geffnet), i.e. utils/net_utils.py:84-91's conv -> BatchNorm -> act pattern on the
// expanded tensors of the /16 and /32 stages: 576 / 816 channels on 18 x 36 maps, 1392 channels on 9 x 18 maps at batch 16.
//
// Why (profiles/r06_sml_b16_bf16_per_shape.txt): those layers are 2.6 K ... 10 K pixels -- 7 ... 17 MB -- and every launch over them is a
// latency floor: forward finalize 9 us + apply 10-13 us, backward reduce 11-13 us + finalize 9 us + apply 9-13 us, 31 such layers per step.
// The pixel-row decomposition of the general kernels needs the three-step protocol because a channel's sum lives in every workgroup.  In the
// channel-vector decomposition it lives in ONE: thread t of the workgroup takes pixels t, t + 256, ... (at most NP of them), keeps their raw
// vectors in registers (NP = 11 or 22: 88 / 176 registers), reduces over the workgroup, and writes
// the result from the registers -- every byte is read once.  A lane's 16 bytes are 1/8 of a 128-byte line, so the eight workgroups that
// share a line are placed on the same XCD (block -> vector mapping below): its L2 fetches the line once.
#include "rd_common.h"
#include "rd_kernels.h"
#include <type_traits>

namespace rd {

static constexpr int SLAB_G = 8;      // 16-byte vectors per 128-byte line

// workgroup -> channel vector: the dispatcher places block b on XCD b % 8; vectors of one line (v / 8 equal) get blocks of one XCD
__device__ __forceinline__ int slab_vector(int b) {
  const int xcd = b & 7, i = b >> 3;
  return ((i / SLAB_G) * 8 + xcd) * SLAB_G + (i % SLAB_G);
}
static unsigned slab_grid(int nvec) { return (unsigned)(8 * cdiv(cdiv(nvec, SLAB_G), 8) * SLAB_G); }

// the compiler may not carry anything derived from v across this point (the backward keeps the RAW vectors over its reduction and converts
// them twice: hoisted conversions tripled its registers -- 325 for 11 pixels per thread, spills at 41)
__device__ __forceinline__ void slab_opaque(uint4& v) {
#ifndef RD_EMU
  asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
#endif
}

// sum of 2 VE per-thread values over the workgroup, result to every thread: wave shuffles, then the four waves in order (fixed: reproducible)
template <int NV>
__device__ __forceinline__ void slab_block_sum(float (&v)[NV], float (*sh)[NV], int t) {
#pragma unroll
  for (int e = 0; e < NV; e++) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) v[e] += __shfl_xor(v[e], o);
  }
  if ((t & 63) == 0) {
#pragma unroll
    for (int e = 0; e < NV; e++) sh[t >> 6][e] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < NV; e++) v[e] = (sh[0][e] + sh[1][e]) + (sh[2][e] + sh[3][e]);
}

// ---- backward: dy = scale (g - mean(g) - xhat mean(g xhat)), g = dz act'(scale y + shift); d gamma (+)= sum g xhat, d beta (+)= sum g --------------
template <typename T, int ACT, int NP>
__global__ __launch_bounds__(256) void bn_bwd_slab_kernel(const T* __restrict__ dz, const T* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ scale, const float* __restrict__ shift,
                                                          float* dgamma, float* dbeta, int accumulate, T* __restrict__ dy, int pixels, int C, int nvec,
                                                          int act, float slope) {
  constexpr int VE = Elem<T>::VE;
  const int actv = ACT >= 0 ? ACT : act;
  __shared__ float sh[4][2 * VE];
  const int t = threadIdx.x;
  const int vec = slab_vector(blockIdx.x);
  if (vec >= nvec) return;
  const int c0 = vec * VE;
  float mu[VE], rs[VE], sc[VE], hf[VE];
#pragma unroll
  for (int e = 0; e < VE; e++) { mu[e] = mean[c0 + e]; rs[e] = rstd[c0 + e]; sc[e] = scale[c0 + e]; hf[e] = shift[c0 + e]; }
  // every request of the workgroup issued before the first use (a pixel past the end re-reads the last one and is not counted)
  uint4 rg[NP], ry[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = t + 256 * j, pc = p < pixels ? p : pixels - 1;
    rg[j] = *reinterpret_cast<const uint4*>(dz + (int64_t)pc * C + c0);
    ry[j] = *reinterpret_cast<const uint4*>(y + (int64_t)pc * C + c0);
  }
  float ab[2 * VE];
#pragma unroll
  for (int e = 0; e < 2 * VE; e++) ab[e] = 0.f;
#pragma unroll
  for (int j = 0; j < NP; j++) {
    if (t + 256 * j < pixels) {
      float g[VE], yy[VE];
      raw16_to_f32((const T*)nullptr, rg[j], g);
      raw16_to_f32((const T*)nullptr, ry[j], yy);
#pragma unroll
      for (int e = 0; e < VE; e++) {
        float gv = g[e];
        if (actv) gv *= act_grad_from_out(yy[e] * sc[e] + hf[e], actv, slope);
        ab[e] += gv; ab[VE + e] += gv * ((yy[e] - mu[e]) * rs[e]);
      }
    }
  }
  slab_block_sum<2 * VE>(ab, sh, t);
#pragma unroll
  for (int j = 0; j < NP; j++) { slab_opaque(rg[j]); slab_opaque(ry[j]); }
  const float inv = 1.0f / (float)pixels;
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = t + 256 * j;
    if (p < pixels) {
      float g[VE], yy[VE], ov[VE];
      raw16_to_f32((const T*)nullptr, rg[j], g);
      raw16_to_f32((const T*)nullptr, ry[j], yy);
#pragma unroll
      for (int e = 0; e < VE; e++) {
        float gv = g[e];
        if (actv) gv *= act_grad_from_out(yy[e] * sc[e] + hf[e], actv, slope);
        const float xh = (yy[e] - mu[e]) * rs[e];
        ov[e] = sc[e] * (gv - ab[e] * inv - xh * (ab[VE + e] * inv));
      }
      stv(dy + (int64_t)p * C + c0, ov);
    }
  }
  if (t < VE) {
    if (dbeta) dbeta[c0 + t] = accumulate ? dbeta[c0 + t] + ab[t] : ab[t];
    if (dgamma) dgamma[c0 + t] = accumulate ? dgamma[c0 + t] + ab[VE + t] : ab[VE + t];
  }
}

// ---- forward: the statistics rows of the producer -> mean / rstd / scale / shift (+ running statistics), then z = act(scale y + shift) ------
// stats[rows][C][2] (sum, sum^2 per row) as the convolution epilogues write them; the finalize arithmetic is bn_finalize_kernel's (rd_norm.hip)
template <typename T, int ACT, int NP>
__global__ __launch_bounds__(256) void bn_fwd_slab_kernel(const float* __restrict__ stats, int rows, const T* __restrict__ y, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, float momentum, float* running_mean,
                                                          float* running_var, float* mean_out, float* rstd_out, float* scale_out, float* shift_out,
                                                          T* __restrict__ out, int pixels, int C, int nvec, int act, float slope) {
  constexpr int VE = Elem<T>::VE;
  constexpr int QV = VE / 2;            // float4 pieces (two channels' pairs) of a row's 2 VE floats
  constexpr int RL = 256 / QV;          // row lanes
  const int actv = ACT >= 0 ? ACT : act;
  __shared__ double red[RL][QV][4];
  __shared__ float coef[2][VE];
  const int t = threadIdx.x;
  const int vec = slab_vector(blockIdx.x);
  if (vec >= nvec) return;
  const int c0 = vec * VE;
  uint4 ry[NP];
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = t + 256 * j, pc = p < pixels ? p : pixels - 1;
    ry[j] = *reinterpret_cast<const uint4*>(y + (int64_t)pc * C + c0);
  }
  {      // the rows of this vector's channels: thread (row lane, piece) sums its rows in double precision, the row lanes are combined in order
    const int q = t % QV, rl = t / QV;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = rl; r < rows; r += RL) {
      const float4 v = *reinterpret_cast<const float4*>(stats + ((int64_t)r * C + c0) * 2 + q * 4);
      s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
    }
#pragma unroll
    for (int i = 0; i < 4; i++) red[rl][q][i] = s[i];
  }
  __syncthreads();
  if (t < VE) {
    const int q = t >> 1, i0 = (t & 1) * 2;
    double sa = 0.0, sb = 0.0;
    for (int rl = 0; rl < RL; rl++) { sa += red[rl][q][i0]; sb += red[rl][q][i0 + 1]; }
    const double count = (double)pixels;
    const double m = sa / count;
    double v = sb / count - m * m;
    if (v < 0.0) v = 0.0;
    const float mean = (float)m, var = (float)v;
    const int c = c0 + t;
    if (running_mean) {
      const double unb = count > 1.0 ? v * count / (count - 1.0) : v;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
    const float rstd = 1.0f / sqrtf(var + eps);
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    if (mean_out) mean_out[c] = mean;
    if (rstd_out) rstd_out[c] = rstd;
    const float scl = g * rstd, sft = b - mean * g * rstd;
    scale_out[c] = scl; shift_out[c] = sft;
    coef[0][t] = scl; coef[1][t] = sft;
  }
  __syncthreads();
  float sc[VE], hf[VE];
#pragma unroll
  for (int e = 0; e < VE; e++) { sc[e] = coef[0][e]; hf[e] = coef[1][e]; }
#pragma unroll
  for (int j = 0; j < NP; j++) {
    const int p = t + 256 * j;
    if (p < pixels) {
      float yy[VE], ov[VE];
      raw16_to_f32((const T*)nullptr, ry[j], yy);
#pragma unroll
      for (int e = 0; e < VE; e++) ov[e] = act_fwd(yy[e] * sc[e] + hf[e], actv, slope);      // rd_affine_act's expression (bn_apply1)
      stv(out + (int64_t)p * C + c0, ov);
    }
  }
}

// ---- routing / launchers --------------------------------------------------------------------------------------------------------------------
// Where (tools/bench_bn.py on MI355X, profiles/r06_microbench/bn_slab.txt): a lane's 16 bytes are one line request, so the form is bound by
// the L1's line rate, not by bytes: at 2 592 pixels (11 vectors per thread) finalize + apply 15.3 -> 9.5 us, backward 21.6 -> 14.1 us (C = 1392;
// 15.6 -> 11.8 at 816), at 10 368 pixels (41 per thread) 16.7 -> 21.2 and 20.6 -> 51 us.  Default: >= 64 workgroups and <= 11 pixels per
// thread; "bn_slab" (rd_set_option): 0 off, 2 = any channel count and up to 22 pixels per thread (tests).
bool bn_slab_ok(int64_t pixels, int C, int dtype) {
  const int ve = dtype == 0 ? 4 : 8, opt = rd_opt(OPT_BN_SLAB, 1);
  if (opt == 0 || pixels <= 0 || (C % ve) != 0) return false;
  if (opt == 2) return pixels <= 256 * 22;
  return pixels <= 256 * 11 && C / ve >= 64;
}
template <typename F>
static void slab_act_dispatch(int act, F&& f) {
  switch (act) {
    case 0: f(std::integral_constant<int, 0>{}); break;
    case 3: f(std::integral_constant<int, 3>{}); break;      // ReLU6: the backbone's activation
    default: f(std::integral_constant<int, -1>{}); break;
  }
}
static int slab_np(int64_t pixels) { return cdiv(pixels, 256) <= 11 ? 11 : 22; }

template <typename T>
static void launch_bn_bwd_slab_t(const void* dz, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, float* dgamma,
                                 float* dbeta, int accumulate, void* dy, int64_t pixels, int C, int act, float slope, hipStream_t st) {
  const int nvec = C / Elem<T>::VE, np = slab_np(pixels);
  const dim3 grid(slab_grid(nvec));
  slab_act_dispatch(act, [&](auto ac) {
    constexpr int A = decltype(ac)::value;
#define RD_SLAB_B(NPV) hipLaunchKernelGGL((bn_bwd_slab_kernel<T, A, NPV>), grid, dim3(256), 0, st, (const T*)dz, (const T*)y, mean, rstd, scale, shift, dgamma, dbeta, \
                                          accumulate, (T*)dy, (int)pixels, C, nvec, act, slope)
    if (np == 11) RD_SLAB_B(11); else RD_SLAB_B(22);
#undef RD_SLAB_B
  });
}
void launch_bn_bwd_slab(const void* dz, const void* y, const float* mean, const float* rstd, const float* scale, const float* shift, float* dgamma,
                        float* dbeta, int accumulate, void* dy, int64_t pixels, int C, int act, float slope, int dtype, hipStream_t st) {
  if (dtype == 0) launch_bn_bwd_slab_t<float>(dz, y, mean, rstd, scale, shift, dgamma, dbeta, accumulate, dy, pixels, C, act, slope, st);
  else launch_bn_bwd_slab_t<bf16_t>(dz, y, mean, rstd, scale, shift, dgamma, dbeta, accumulate, dy, pixels, C, act, slope, st);
}
template <typename T>
static void launch_bn_fwd_slab_t(const float* stats, int rows, const void* y, const float* gamma, const float* beta, float eps, float momentum,
                                 float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift, void* out, int64_t pixels,
                                 int C, int act, float slope, hipStream_t st) {
  const int nvec = C / Elem<T>::VE, np = slab_np(pixels);
  const dim3 grid(slab_grid(nvec));
  slab_act_dispatch(act, [&](auto ac) {
    constexpr int A = decltype(ac)::value;
#define RD_SLAB_F(NPV) hipLaunchKernelGGL((bn_fwd_slab_kernel<T, A, NPV>), grid, dim3(256), 0, st, stats, rows, (const T*)y, gamma, beta, eps, momentum, running_mean, \
                                          running_var, mean, rstd, scale, shift, (T*)out, (int)pixels, C, nvec, act, slope)
    if (np == 11) RD_SLAB_F(11); else RD_SLAB_F(22);
#undef RD_SLAB_F
  });
}
void launch_bn_fwd_slab(const float* stats, int rows, const void* y, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                        float* running_var, float* mean, float* rstd, float* scale, float* shift, void* out, int64_t pixels, int C, int act, float slope,
                        int dtype, hipStream_t st) {
  if (dtype == 0) launch_bn_fwd_slab_t<float>(stats, rows, y, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, out, pixels, C, act, slope, st);
  else launch_bn_fwd_slab_t<bf16_t>(stats, rows, y, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, out, pixels, C, act, slope, st);
}
const char* bn_slab_kernel_name(int which, int64_t pixels, int dtype, int act) {      // which 0 = forward, 1 = backward (as rocprofv3 prints them)
  static thread_local char buf[96];
  const int A = (act == 0 || act == 3) ? act : -1;
  snprintf(buf, sizeof(buf), "%s<%s, %d, %d>", which ? "bn_bwd_slab_kernel" : "bn_fwd_slab_kernel", dtype == 0 ? "float" : RD_T16_NAME, A, slab_np(pixels));
  return buf;
}

}  // namespace rd
