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


// ---- channel-vectorised depthwise kernels (C % 4 == 0, k in {3, 5}) ---------------------------------------------------------------------
// thread = (quad of 4 channels, pixel lane); the k*k x 4 weights of the quad live in registers; a thread produces DWR = 4 consecutive
// outputs of a row from a sliding register window, so a filter row costs (DWR-1)*S + K loads for DWR outputs.
// Every load is UNCONDITIONAL (clamped address, zero selected afterwards): a bounds `if` around each load made the compiler emit one
// branch + one s_waitcnt per load, i.e. 40 serialised memory latencies per unit; without them a row's (or a unit's) loads issue together.
// Block geometry: QB quads x PL pixel lanes <= 256 threads with QB = ceil(C/4 / nchunk) chosen for the fewest idle threads (C/4 is
// 36..348 in EfficientNet-Lite3: a power-of-two QB idled up to 44 % of the lanes).
#ifndef RD_DW_ST_MODE      // forward statistics epilogue: 1 = every pixel lane sums one of the quad's eight statistics (0.58 -> 0.53 ms per SML step of
#define RD_DW_ST_MODE 1    // depthwise forwards; tools/bench_dw.py with BD_STATS=1), 0 = lane 0 sums all eight serially, 2 = probe without the reduction (0.46)
#endif
static constexpr int DWR = 4;
// The tap products are explicit fmaf(): the library is built with -ffp-contract=off (emulator and GPU round alike), which turned the 400
// multiply-adds of a 5x5 unit into 208 v_pk_mul_f32 + 254 v_pk_add_f32; fused they are ~200 v_pk_fma_f32 (the host emulator's fmaf is the
// same correctly-rounded operation).  These kernels are vector-issue bound (~750 instructions per 16 outputs at 5x5), not HBM bound.

// The block's weights are a contiguous run of w (QB quads x 4 channels x k*k floats): staged into LDS with coalesced 16-byte loads, then
// each thread picks up its quad (k*k x 4 registers).  Reading them straight from global memory is a 400-byte-strided gather -- 64 cache
// lines per instruction -- and cost more than the convolution itself on the small late-stage maps.  STR (floats per quad in LDS) keeps the
// 16-byte LDS reads of 16 consecutive lanes on distinct banks.
static constexpr int DW_QMAX = 64;
template <int K> struct DwLds { static constexpr int STR = K == 5 ? 108 : 4 * K * K; };
template <int K>
__device__ __forceinline__ void dw_stage_weights(const float* __restrict__ w, int c0, int nq, float* lds, int cl, bool active,
                                                 float (&wr)[K * K][4]) {
  constexpr int QF = 4 * K * K, STR = DwLds<K>::STR;
  const float* src = w + (int64_t)c0 * K * K;
  if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    for (int i = threadIdx.x; i < nq * (QF / 4); i += blockDim.x) {
      const float4 v = reinterpret_cast<const float4*>(src)[i];
      const int qd = (i * 4) / QF, off = i * 4 - qd * QF;
      *reinterpret_cast<float4*>(lds + qd * STR + off) = v;
    }
  } else {
    for (int i = threadIdx.x; i < nq * QF; i += blockDim.x) { const int qd = i / QF; lds[qd * STR + i - qd * QF] = src[i]; }
  }
  __syncthreads();
  if (active) {
#pragma unroll
    for (int j = 0; j < K * K; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) wr[j][e] = lds[cl * STR + e * K * K + j];
  }
}

// ST: the forward also emits BatchNorm statistics of what it stores -- stats[blockIdx.x][C][2] = (sum, sum of squares) of the block's rounded
// outputs -- so the layer's BatchNorm needs no separate pass over y.
template <typename T, int K, int S, int MODE, bool ST>  // MODE 0: forward (stride S); MODE 1: data gradient of a stride-1 layer (S == 1)
// (the 5x5 stride-1 forward with statistics lands a few registers above 256 without the occupancy hint, i.e. at one wave per SIMD)
__global__ __launch_bounds__(256) RD_WAVES_PER_EU((ST && S == 1) ? 2 : 1) void dw_run_kernel(const T* __restrict__ src, const float* __restrict__ w, T* __restrict__ dst, int N,
                                                     int H, int W, int C, int OH, int OW, int p, int QB, int PL, int PPB,
                                                     float* __restrict__ stats) {
  __shared__ __attribute__((aligned(16))) float wl[DW_QMAX * DwLds<K>::STR];
  const int t = threadIdx.x, pl = t / QB, cl = t - pl * QB;
  const int c = (blockIdx.y * QB + cl) * 4;
  const bool active = pl < PL && c < C;
  float wr[K * K][4];
  dw_stage_weights<K>(w, blockIdx.y * QB * 4, min(QB, C / 4 - (int)blockIdx.y * QB), wl, cl, active, wr);
  if (!ST && !active) return;
  // statistics accumulate in the thread's own LDS slot (the weights' LDS, free once every thread holds its registers): 8 more live
  // registers across the unit loop would push the 5x5 kernels past 256 VGPRs, i.e. down to one wave per SIMD
  float* red = wl;      // [256][8] floats <= DW_QMAX * STR
  if (ST) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) red[t * 8 + e] = 0.f;
  }
  constexpr int WIN = (DWR - 1) * S + K;
  const int DH = MODE ? H : OH, DW = MODE ? W : OW;           // destination size
  const int SH = MODE ? OH : H, SW = MODE ? OW : W;           // source size
  const int runs = (DW + DWR - 1) / DWR;
  const int64_t M = (int64_t)N * DH * runs;
  const int64_t mbeg = (int64_t)xcd_contiguous(blockIdx.x, gridDim.x) * PPB, mend = !active ? 0 : (mbeg + PPB < M ? mbeg + PPB : M);
  for (int64_t m = mbeg + pl; m < mend; m += PL) {
    const int rw = (int)(m % runs); int64_t q = m / runs; const int dh = (int)(q % DH); const int n = (int)(q / DH);
    const int dw0 = rw * DWR;
    const int cbase = MODE ? dw0 + p - (K - 1) : dw0 * S - p;     // source column of window slot 0
    int coff[WIN]; bool cok[WIN];
#pragma unroll
    for (int j = 0; j < WIN; j++) {
      const int sw = cbase + j;
      cok[j] = (unsigned)sw < (unsigned)SW;
      coff[j] = (sw < 0 ? 0 : (sw >= SW ? SW - 1 : sw)) * C;
    }
    float acc[DWR][4];
#pragma unroll
    for (int r = 0; r < DWR; r++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[r][e] = 0.f;
#pragma unroll
    for (int kh = 0; kh < K; kh++) {
      const int sh = MODE ? dh + p - kh : dh * S - p + kh;
      const bool rok = (unsigned)sh < (unsigned)SH;
      const T* row = src + (((int64_t)n * SH + (sh < 0 ? 0 : (sh >= SH ? SH - 1 : sh))) * SW) * C + c;
      float v[WIN][4];
#pragma unroll
      for (int j = 0; j < WIN; j++) ld4z(row + coff[j], rok && cok[j], v[j]);
#pragma unroll
      for (int kw = 0; kw < K; kw++)
#pragma unroll
        for (int r = 0; r < DWR; r++)
#pragma unroll
          for (int e = 0; e < 4; e++) acc[r][e] = fmaf(v[MODE ? r + K - 1 - kw : r * S + kw][e], wr[kh * K + kw][e], acc[r][e]);
    }
#pragma unroll
    for (int r = 0; r < DWR; r++)
      if (dw0 + r < DW) {
        st4(dst + (((int64_t)n * DH + dh) * DW + dw0 + r) * C + c, acc[r]);
      }
    if (ST) {
      float u1[4] = {0.f, 0.f, 0.f, 0.f}, u2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < DWR; r++)
        if (dw0 + r < DW) {
#pragma unroll
          for (int e = 0; e < 4; e++) { const float v = Elem<T>::rnd(acc[r][e]); u1[e] += v; u2[e] += v * v; }
        }
#pragma unroll
      for (int e = 0; e < 4; e++) { red[t * 8 + e] += u1[e]; red[t * 8 + 4 + e] += u2[e]; }
    }
  }
  if (ST) {      // pixel lanes of the block summed in a fixed order
    __syncthreads();
#if RD_DW_ST_MODE == 0
    if (active && pl == 0) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float a = 0.f, b = 0.f;
        for (int q = 0; q < PL; q++) { a += red[(q * QB + cl) * 8 + e]; b += red[(q * QB + cl) * 8 + 4 + e]; }
        stats[((int64_t)blockIdx.x * C + c + e) * 2] = a;
        stats[((int64_t)blockIdx.x * C + c + e) * 2 + 1] = b;
      }
    }
#elif RD_DW_ST_MODE == 1
    // pixel lane pl sums statistic e8 = pl (, pl + PL, ...) of its quad over all lanes, four LDS reads in flight, same order of additions
    if (active) {
      for (int e8 = pl; e8 < 8; e8 += PL) {
        float a = 0.f;
        for (int q0 = 0; q0 < PL; q0 += 4) {
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = red[(min(q0 + u, PL - 1) * QB + cl) * 8 + e8];
#pragma unroll
          for (int u = 0; u < 4; u++) if (q0 + u < PL) a += v[u];
        }
        stats[((int64_t)blockIdx.x * C + c + (e8 & 3)) * 2 + (e8 >> 2)] = a;
      }
    }
#else
    if (active && pl == 0 && red[t * 8] == 12345.f) stats[0] = 1.f;      // probe: no reduction
#endif
  }
}

// data gradient of a stride-2 layer: dx[ih, iw] = sum over (kh, kw) with (ih + p - kh), (iw + p - kw) even of dy[.. / 2] * w[kh, kw].
// A run of 4 dx columns starts at a multiple of 4, so which kw pairs with which dy column is a compile-time pattern given the parity PP
// of the pad; the row parity (ih + p) & 1 picks one of two fully unrolled bodies.
template <typename T, int K, int PP, int PH>
__device__ __forceinline__ void dw_dgrad2_rows(const T* __restrict__ dy, const float (&wr)[K * K][4], int n, int dh, int p, int OH, int OW, int C,
                                               int c, const int* coff, const bool* cok, float (&acc)[DWR][4]) {
  constexpr int NJ = (K + 2 - PP) / 2 + 1;
#pragma unroll
  for (int kh = PH; kh < K; kh += 2) {
    const int oh = (dh + p - kh) >> 1;            // (dh + p - kh) is even here; negative -> negative
    const bool rok = (unsigned)oh < (unsigned)OH;
    const T* row = dy + (((int64_t)n * OH + (oh < 0 ? 0 : (oh >= OH ? OH - 1 : oh))) * OW) * C + c;
    float v[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; j++) ld4z(row + coff[j], rok && cok[j], v[j]);
#pragma unroll
    for (int r = 0; r < DWR; r++)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const int kw = r + (K - 1 - PP) - 2 * j;
        if (kw >= 0 && kw < K) {
#pragma unroll
          for (int e = 0; e < 4; e++) acc[r][e] = fmaf(v[j][e], wr[kh * K + kw][e], acc[r][e]);
        }
      }
  }
}
template <typename T, int K, int PP>
__global__ __launch_bounds__(256) void dw_dgrad2_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int N,
                                                        int H, int W, int C, int OH, int OW, int p, int QB, int PL, int PPB) {
  __shared__ __attribute__((aligned(16))) float wl[DW_QMAX * DwLds<K>::STR];
  const int t = threadIdx.x, pl = t / QB, cl = t - pl * QB;
  const int c = (blockIdx.y * QB + cl) * 4;
  const bool active = pl < PL && c < C;
  float wr[K * K][4];
  dw_stage_weights<K>(w, blockIdx.y * QB * 4, min(QB, C / 4 - (int)blockIdx.y * QB), wl, cl, active, wr);
  if (!active) return;
  constexpr int NJ = (K + 2 - PP) / 2 + 1;
  const int runs = (W + DWR - 1) / DWR;
  const int64_t M = (int64_t)N * H * runs;
  const int64_t mbeg = (int64_t)xcd_contiguous(blockIdx.x, gridDim.x) * PPB, mend = mbeg + PPB < M ? mbeg + PPB : M;
  for (int64_t m = mbeg + pl; m < mend; m += PL) {
    const int rw = (int)(m % runs); int64_t q = m / runs; const int dh = (int)(q % H); const int n = (int)(q / H);
    const int iw0 = rw * DWR;
    const int obase = (iw0 + p - K + 1 + PP) >> 1;     // first dy column that reaches this run (the numerator is even)
    int coff[NJ]; bool cok[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const int ow = obase + j;
      cok[j] = (unsigned)ow < (unsigned)OW;
      coff[j] = (ow < 0 ? 0 : (ow >= OW ? OW - 1 : ow)) * C;
    }
    float acc[DWR][4];
#pragma unroll
    for (int r = 0; r < DWR; r++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[r][e] = 0.f;
    if ((dh + p) & 1) dw_dgrad2_rows<T, K, PP, 1>(dy, wr, n, dh, p, OH, OW, C, c, coff, cok, acc);
    else dw_dgrad2_rows<T, K, PP, 0>(dy, wr, n, dh, p, OH, OW, C, c, coff, cok, acc);
#pragma unroll
    for (int r = 0; r < DWR; r++)
      if (iw0 + r < W) st4(dx + (((int64_t)n * H + dh) * W + iw0 + r) * C + c, acc[r]);
  }
}

// weight gradient: thread accumulates the k*k x 4 taps of its quad over its units; the PL pixel lanes of a block are summed through LDS
// one filter row at a time and written as partial[block][tap][C] (16-byte stores, coalesced over channels).
template <typename T, int K, int S>
__global__ __launch_bounds__(256) void dw_wgrad_run_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ partial,
                                                           int N, int H, int W, int C, int OH, int OW, int p, int QB, int PL) {
  __shared__ float red[256][K][4];
  const int t = threadIdx.x, pl = t / QB, cl = t - pl * QB;
  const int c = (blockIdx.y * QB + cl) * 4;
  const bool cv = pl < PL && c < C;
  constexpr int WIN = (DWR - 1) * S + K;
  const int runs = (OW + DWR - 1) / DWR;
  const int64_t units = (int64_t)N * OH * runs;
  const int64_t per = cdiv(units, gridDim.x);
  const int64_t ubeg = (int64_t)xcd_contiguous(blockIdx.x, gridDim.x) * per, uend = ubeg + per < units ? ubeg + per : units;
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
      const T* grow = dy + (((int64_t)n * OH + oh) * OW) * C + c;
#pragma unroll
      for (int r = 0; r < DWR; r++) ld4z(grow + (int64_t)(ow0 + r < OW ? ow0 + r : OW - 1) * C, ow0 + r < OW, g[r]);
      const int cbase = ow0 * S - p;
      int coff[WIN]; bool cok[WIN];
#pragma unroll
      for (int j = 0; j < WIN; j++) {
        const int iw = cbase + j;
        cok[j] = (unsigned)iw < (unsigned)W;
        coff[j] = (iw < 0 ? 0 : (iw >= W ? W - 1 : iw)) * C;
      }
#pragma unroll
      for (int kh = 0; kh < K; kh++) {
        const int ih = oh * S - p + kh;
        const bool rok = (unsigned)ih < (unsigned)H;
        const T* row = x + (((int64_t)n * H + (ih < 0 ? 0 : (ih >= H ? H - 1 : ih))) * W) * C + c;
        float v[WIN][4];
#pragma unroll
        for (int j = 0; j < WIN; j++) ld4z(row + coff[j], rok && cok[j], v[j]);
#pragma unroll
        for (int kw = 0; kw < K; kw++)
#pragma unroll
          for (int r = 0; r < DWR; r++)
#pragma unroll
            for (int e = 0; e < 4; e++) acc[kh * K + kw][e] = fmaf(g[r][e], v[r * S + kw][e], acc[kh * K + kw][e]);
      }
    }
  }
#pragma unroll
  for (int kh = 0; kh < K; kh++) {
#pragma unroll
    for (int kw = 0; kw < K; kw++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[t][kw][e] = acc[kh * K + kw][e];
    __syncthreads();
    // pixel lane pl sums tap kw = pl (, pl + PL, ...) of this filter row over all lanes, four LDS reads in flight (same order of
    // additions as one lane walking q = 0 .. PL-1).  With lane 0 alone doing all K taps one dependent read at a time this epilogue
    // took longer than the unit loop: 25-30 of the 50 us of EVERY 5x5 layer whatever its size (probe builds: unit loop only 22-27 us,
    // epilogue only 30-35).  Also measured and not kept: all 44 loads of a unit issued up front (0.88 -> 0.87 ms per SML step, mixed per
    // layer), one or two units per thread instead of four (0.91 / 0.99: more blocks = more epilogues and partial rows).
    if (cv) {
      for (int kw = pl; kw < K; kw += PL) {
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < PL; q0 += 4) {
          float4 v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(&red[min(q0 + u, PL - 1) * QB + cl][kw][0]);
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (q0 + u < PL) { s4[0] += v[u].x; s4[1] += v[u].y; s4[2] += v[u].z; s4[3] += v[u].w; }
        }
        st4(partial + ((int64_t)blockIdx.x * (K * K) + kh * K + kw) * C + c, s4);
      }
    }
    __syncthreads();
  }
}

// partial[rows][taps][C] -> dw[c][tap]: 64 elements x 4 row groups per block, rows read coalesced over (tap, c)
__global__ __launch_bounds__(256) void dw_wgrad_finalize_tc_kernel(const float* __restrict__ partial, int rows, int C, int KK, float* dw,
                                                                   int accumulate) {
  __shared__ double red[4][64];
  const int e = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + e, CK = C * KK;
  double a = 0.0;
  if (i < CK)
    for (int r = rg; r < rows; r += 4) a += partial[(int64_t)r * CK + i];
  red[rg][e] = a;
  __syncthreads();
  if (rg == 0 && i < CK) {
    a = red[0][e] + red[1][e] + red[2][e] + red[3][e];
    const int j = i / C, c = i - j * C;
    const int o = c * KK + j;
    dw[o] = accumulate ? dw[o] + (float)a : (float)a;
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

// block geometry of the quad kernels: QB quads x PL pixel lanes, nchunk blocks along the channels, fewest idle threads
struct QG { int QB, PL, nchunk; };
static QG qgeom(int C, int qmax) {
  const int c4 = C / 4;
  QG best = {1, 256, c4}; double bu = -1.0;
  for (int nc = 1; nc <= 256; nc++) {
    const int qb = (int)cdiv(c4, nc);
    if (qb > qmax) continue;
    const int pl = 256 / qb;
    const double u = (double)c4 * pl / (256.0 * nc);
    if (u > bu + 1e-9) { bu = u; best = {qb, pl, nc}; }
  }
  return best;
}
// units a thread works through.  ONE (round 6): rounds 2-5 gave a thread up to 8 (>= ~1024 blocks per launch) so that its k*k x 4 weight registers
// were loaded once per several units; swept in the captured SML step on one box (two alternating rounds): 1024 blocks / <= 8 units 11.90 ms,
// 2048 / 8 11.85, 1024 / 4 11.85, 4096 / 4 11.82, one unit per thread 11.80 -- the units of a thread are a serial chain of load round trips,
// and more blocks hide them better than fewer weight loads pay
static int dw_upt(int64_t units, const QG& g) {
  (void)units; (void)g;
  return 1;
}
// units per block of the forward with fused statistics: as dw_upt, but never more than 1024 blocks (= statistics rows) along the units
static int dw_stats_ppb(int64_t units, const QG& g) {
  const int64_t upt = std::max<int64_t>(dw_upt(units, g), cdiv(units, (int64_t)g.PL * 1024));
  return (int)(g.PL * upt);
}
template <typename T, int MODE>
static void launch_dw_quad(const void* src, const float* w, void* dst, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, hipStream_t st,
                           float* stats = nullptr) {
  const QG g = qgeom(C, DW_QMAX);      // the block's weights are staged in LDS: at most DW_QMAX quads
  const int DH = MODE ? H : OH, DW = MODE ? W : OW;
  const int64_t units = (int64_t)N * DH * cdiv(DW, DWR);
  const int ppb = stats ? dw_stats_ppb(units, g) : g.PL * dw_upt(units, g);
  const dim3 grid((unsigned)cdiv(units, ppb), g.nchunk);
#define RD_DWQ(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, st, (const T*)src, w, (T*)dst, N, H, W, C, OH, OW, p, g.QB, g.PL, ppb, stats)
#define RD_DWQ2(KERNEL) hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, st, (const T*)src, w, (T*)dst, N, H, W, C, OH, OW, p, g.QB, g.PL, ppb)
  if (MODE == 0 && stats) {
    if (s == 1) { if (k == 3) RD_DWQ((dw_run_kernel<T, 3, 1, 0, true>)); else RD_DWQ((dw_run_kernel<T, 5, 1, 0, true>)); }
    else { if (k == 3) RD_DWQ((dw_run_kernel<T, 3, 2, 0, true>)); else RD_DWQ((dw_run_kernel<T, 5, 2, 0, true>)); }
  } else if (MODE == 0) {
    if (s == 1) { if (k == 3) RD_DWQ((dw_run_kernel<T, 3, 1, 0, false>)); else RD_DWQ((dw_run_kernel<T, 5, 1, 0, false>)); }
    else { if (k == 3) RD_DWQ((dw_run_kernel<T, 3, 2, 0, false>)); else RD_DWQ((dw_run_kernel<T, 5, 2, 0, false>)); }
  } else if (s == 1) {
    if (k == 3) RD_DWQ((dw_run_kernel<T, 3, 1, 1, false>)); else RD_DWQ((dw_run_kernel<T, 5, 1, 1, false>));
  } else if (p & 1) {
    if (k == 3) RD_DWQ2((dw_dgrad2_kernel<T, 3, 1>)); else RD_DWQ2((dw_dgrad2_kernel<T, 5, 1>));
  } else {
    if (k == 3) RD_DWQ2((dw_dgrad2_kernel<T, 3, 0>)); else RD_DWQ2((dw_dgrad2_kernel<T, 5, 0>));
  }
#undef RD_DWQ
#undef RD_DWQ2
}
static bool dw_quad_ok(int C, int k, int s) { return C % 4 == 0 && (k == 3 || k == 5) && (s == 1 || s == 2); }

int dwconv_stats_rows(int N, int OH, int OW, int C, int k, int s) {
  if (!dw_quad_ok(C, k, s)) return 0;
  const QG g = qgeom(C, DW_QMAX);
  const int64_t units = (int64_t)N * OH * cdiv(OW, DWR);
  return (int)cdiv(units, dw_stats_ppb(units, g));
}
void launch_dwconv_fwd(const void* x, const float* w, void* y, int N, int H, int W, int C, int OH, int OW, int k, int s, int p, int dtype,
                       hipStream_t st, float* stats) {
  if (dw_quad_ok(C, k, s)) {
    if (dtype == 0) launch_dw_quad<float, 0>(x, w, y, N, H, W, C, OH, OW, k, s, p, st, stats);
    else launch_dw_quad<bf16_t, 0>(x, w, y, N, H, W, C, OH, OW, k, s, p, st, stats);
    return;
  }
  int64_t n = (int64_t)N * OH * OW * C;
  if (dtype == 0) hipLaunchKernelGGL((dwconv_fwd_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)x, w, (float*)y, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((dwconv_fwd_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)x, w, (bf16_t*)y, N, H, W, C, OH, OW, k, s, p);
}
void launch_dwconv_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int OH, int OW, int k, int s, int p,
                         int dtype, hipStream_t st) {
  if (dw_quad_ok(C, k, s)) {
    if (dtype == 0) launch_dw_quad<float, 1>(dy, w, dx, N, H, W, C, OH, OW, k, s, p, st);
    else launch_dw_quad<bf16_t, 1>(dy, w, dx, N, H, W, C, OH, OW, k, s, p, st);
    return;
  }
  int64_t n = (int64_t)N * H * W * C;
  if (dtype == 0) hipLaunchKernelGGL((dwconv_dgrad_kernel<float>), dim3(ew_grid(n)), dim3(256), 0, st, (const float*)dy, w, (float*)dx, N, H, W, C, OH, OW, k, s, p);
  else hipLaunchKernelGGL((dwconv_dgrad_kernel<bf16_t>), dim3(ew_grid(n)), dim3(256), 0, st, (const bf16_t*)dy, w, (bf16_t*)dx, N, H, W, C, OH, OW, k, s, p);
}
// many depthwise weight gradients finished by ONE launch (item i exactly as dw_wgrad_finalize_tc_kernel: 64 elements x 4 row groups per block)
struct DwWgradBatch { DwWgradItem it[DW_WGRAD_BATCH_MAX]; };
__global__ __launch_bounds__(256) void dw_wgrad_finalize_batch_kernel(DwWgradBatch b) {
  __shared__ double red[4][64];
  const DwWgradItem it = b.it[blockIdx.y];
  const int e = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int CK = it.C * it.KK;
  for (int base = blockIdx.x * 64; base < CK; base += gridDim.x * 64) {      // (uniform per block)
    const int i = base + e;
    double a = 0.0;
    if (i < CK) {
      // eight rows in flight per iteration, added in row order (see colsum_finalize_batch_kernel): 58 us for 17 items before
      for (int r = rg; r < it.rows; r += 32) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { const int rq = r + 4 * q; const float x = it.partial[(int64_t)(rq < it.rows ? rq : 0) * CK + i]; v[q] = rq < it.rows ? x : 0.f; }
#pragma unroll
        for (int q = 0; q < 8; q++) a += v[q];
      }
    }
    red[rg][e] = a;
    __syncthreads();
    if (rg == 0 && i < CK) {
      a = red[0][e] + red[1][e] + red[2][e] + red[3][e];
      const int j = i / it.C, c = i - j * it.C;
      const int o = c * it.KK + j;
      it.dw[o] = it.accumulate ? it.dw[o] + (float)a : (float)a;
    }
    __syncthreads();
  }
}
void launch_dw_wgrad_finalize_batch(const DwWgradItem* items, int n, hipStream_t st) {
  for (int base = 0; base < n; base += DW_WGRAD_BATCH_MAX) {
    const int m = std::min(DW_WGRAD_BATCH_MAX, n - base);
    DwWgradBatch b;
    int maxg = 1;
    for (int i = 0; i < m; i++) { b.it[i] = items[base + i]; maxg = std::max(maxg, (int)cdiv(b.it[i].C * b.it[i].KK, 64)); }
    for (int i = m; i < DW_WGRAD_BATCH_MAX; i++) b.it[i] = b.it[0];
    hipLaunchKernelGGL(dw_wgrad_finalize_batch_kernel, dim3((unsigned)std::min(maxg, 1024), (unsigned)m), dim3(256), 0, st, b);
  }
}
// item != nullptr: the vector path leaves its partial rows and fills *item for a later launch_dw_wgrad_finalize_batch (item->rows = 0 when
// the gradient was finished here: shapes on the scalar path)
void launch_dwconv_wgrad(const void* x, const void* dy, float* partial, float* dw, int accumulate, int N, int H, int W, int C, int OH,
                         int OW, int k, int s, int p, int dtype, hipStream_t st, DwWgradItem* item) {
  if (item) { item->partial = partial; item->dw = dw; item->rows = 0; item->C = C; item->KK = k * k; item->accumulate = accumulate; }
  const int rows = dw_rows((int64_t)N * OH * OW, C);      // what the caller sized `partial` for
  if (dw_quad_ok(C, k, s)) {
    const QG g = qgeom(C, 256);
    const int64_t units = (int64_t)N * OH * cdiv(OW, DWR);
    // blocks along the units: ~512 blocks in all, >= 3 units per thread, and no more partial rows than the tensors justify (round 6, swept in
    // the captured SML step on one box, two alternating rounds: 1024 / 4 -- the values since round 3 -- 12.02 ms, 768 / 3 11.95, 512 / 3 11.89,
    // 448 / 3 12.00, 256 / 4 12.10: every block ends in the K-row LDS epilogue and writes a partial row the finalize launch re-reads)
    int64_t gx = std::min<int64_t>(512 / g.nchunk, cdiv(units, (int64_t)g.PL * 3));
    gx = std::min<int64_t>(gx, std::max<int64_t>(64, (int64_t)N * OH * OW / 32));
    gx = std::max<int64_t>(1, std::min<int64_t>(gx, rows));
    const dim3 grid((unsigned)gx, g.nchunk);
#define RD_DWG(T, K, S) hipLaunchKernelGGL((dw_wgrad_run_kernel<T, K, S>), grid, dim3(256), 0, st, (const T*)x, (const T*)dy, partial, N, H, W, C, OH, OW, p, g.QB, g.PL)
    if (dtype == 0) {
      if (s == 1) { if (k == 3) RD_DWG(float, 3, 1); else RD_DWG(float, 5, 1); }
      else { if (k == 3) RD_DWG(float, 3, 2); else RD_DWG(float, 5, 2); }
    } else {
      if (s == 1) { if (k == 3) RD_DWG(bf16_t, 3, 1); else RD_DWG(bf16_t, 5, 1); }
      else { if (k == 3) RD_DWG(bf16_t, 3, 2); else RD_DWG(bf16_t, 5, 2); }
    }
#undef RD_DWG
    if (item) { item->rows = (int)gx; return; }
    hipLaunchKernelGGL(dw_wgrad_finalize_tc_kernel, dim3((unsigned)cdiv(C * k * k, 64)), dim3(256), 0, st, partial, (int)gx, C, k * k, dw, accumulate);
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
    // the output rows / columns that interpolate from this pixel, with their weights: ALU only, at most four each for the x2 resampling
    int ohp[4] = {0, 0, 0, 0}, owp[4] = {0, 0, 0, 0}, nr = 0, nc = 0;
    float whv[4] = {0.f, 0.f, 0.f, 0.f}, wwv[4] = {0.f, 0.f, 0.f, 0.f};
    for (int oh = max(ohc - rh, 0); oh <= min(ohc + rh, OH - 1); oh++) {
      int h0, h1; float lh;
      bil_src(oh, H, OH, align, h0, h1, lh);
      const float wh = (h0 == h ? 1.f - lh : 0.f) + (h1 == h ? lh : 0.f);
      if (wh != 0.f) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (nr == j) { ohp[j] = oh; whv[j] = wh; }
        nr++;
      }
    }
    for (int ow = max(owc - rw, 0); ow <= min(owc + rw, OW - 1); ow++) {
      int w0, w1; float lw;
      bil_src(ow, W, OW, align, w0, w1, lw);
      const float ww = (w0 == w ? 1.f - lw : 0.f) + (w1 == w ? lw : 0.f);
      if (ww != 0.f) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (nc == j) { owp[j] = ow; wwv[j] = ww; }
        nc++;
      }
    }
    if (nr <= 4 && nc <= 4) {
      // all sixteen readers requested together, unconditionally (slots beyond nr / nc point at output (0, 0) of the image with weight 0):
      // a load inside the candidate loops is one memory round trip per reader.  Same summation order as the loops (row, then column).
      uint4 raw[16];
#pragma unroll
      for (int q = 0; q < 16; q++)
        raw[q] = *reinterpret_cast<const uint4*>(dy + (((int64_t)n * OH + ohp[q >> 2]) * OW + owp[q & 3]) * C + cv * VE);
#pragma unroll
      for (int q = 0; q < 16; q++) {
        float v[VE];
        raw16_to_f32(reinterpret_cast<const T*>(0), raw[q], v);
        const bool ok = (q >> 2) < nr && (q & 3) < nc;
        const float wgt = whv[q >> 2] * wwv[q & 3];
#pragma unroll
        for (int e = 0; e < VE; e++) g[e] += ok ? wgt * v[e] : 0.f;
      }
    } else {
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
