// Grouped weight gradient of 1x1 / linear layers:  dW_p[Cout][Cin] = dY_p^T [Cout x M] . [X1_p | X2_p] [M x Cin]  for MANY layers p at once.
//
// RC-Net's LoFTR transformer (reference RCNet/linear_attention.py:84-135: q/k/v/merge projections and the two MLP linears of 16 layer
// applications) has 96 such products per step with M = R*L = 5040 tokens and 128..256 channels: 26 GFLOP in total, but as 96 separate
// weight-gradient + slab-reduce launches they cost 2.9 ms (15.7 + 8 us each, latency bound).  Here the engine defers them to the end
// of the backward sweep and issues ONE launch over (layer, 64x64 output tile, token split) blocks plus ONE ordered reduction.
// Descriptors travel by value in the kernel arguments (no device table, nothing to keep alive under hipGraph capture).
// bf16: token-major tiles are copied to LDS as they are and the pixel(k)-major MFMA fragments come from ds_read_b64_tr_b16 (see
// rd_common.h); fp32: v_mfma_f32_16x16x4_f32 takes one k per lane, so plain ds_read_b32 of the same image suffices.
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

struct LwgBatch { LwgGemm it[LWG_MAX_ITEMS]; };
struct LwgRedBatch { LwgReduce it[LWG_MAX_REDS]; };

// Block tile = (NI x 64 input channels) x (NO x 64 output channels), NI + NO <= 4 (round 4; 64 x 64 before).  The EfficientNet-Lite3 1x1 layers
// of the SML are 24..232 channels on one side and 6x that on the other: with 64 x 64 tiles the narrow operand's token rows were staged once
// per 64-wide tile of the wide one (x of a 24 -> 144 expansion three times, dY of a 576 -> 96 projection nine times; PMC round 3: 1.47 GB
// fetched per launch against ~0.5 GB of operands).  A block now stages up to three tiles of the wide operand next to one of the narrow one
// (or 2 + 2) per token stage, and every wave keeps its 32 x 32 corner of each of the NI x NO sub-tiles.
__host__ __device__ __forceinline__ void lwg_shape(int Cin, int Cout, int& ni, int& no) {
  const int tci = (Cin + 63) >> 6, tco = (Cout + 63) >> 6;
  if (tci == 1) { ni = 1; no = tco < 3 ? tco : 3; }
  else if (tco == 1) { no = 1; ni = tci < 3 ? tci : 3; }
  else { ni = 2; no = 2; }
}
__host__ __device__ __forceinline__ int lwg_blocks_per_split(int Cin, int Cout) {
  int ni, no;
  lwg_shape(Cin, Cout, ni, no);
  return ((((Cin + 63) >> 6) + ni - 1) / ni) * ((((Cout + 63) >> 6) + no - 1) / no);
}

template <typename T, int NI, int NO>
__device__ __forceinline__ void lwg_body(const LwgGemm& g, uint4 (&sT)[2][4][(sizeof(T) == 2 ? 64 * 8 : 32 * 17)]) {
  constexpr bool BF = sizeof(T) == 2;
  constexpr int ST = BF ? 64 : 32;             // tokens per stage (a 64-channel row is 128 / 256 bytes)
  constexpr int RS = BF ? 8 : 17;              // 16-byte slots per LDS row (fp32 rows padded by one slot: 4 k-groups on 4 bank sets)
  constexpr int SPR = BF ? 8 : 16;             // data slots per row
  constexpr int VE = Elem<T>::VE;
  const int Cin = g.C1 + g.C2;
  const int tci = (Cin + 63) >> 6, tco = (g.Cout + 63) >> 6;   // partial edge tiles: channel counts need only be multiples of VE
  const int gci = (tci + NI - 1) / NI, gco = (tco + NO - 1) / NO;
  const int per = gci * gco;
  if ((int)blockIdx.x >= per * g.nsplit) return;
  const int sp = blockIdx.x / per, rem = blockIdx.x - sp * per;
  const int co0 = (rem / gci) * 64 * NO, ci0 = (rem % gci) * 64 * NI;
  const int mbeg = sp * g.rows_per_split, mend = min(g.M, mbeg + g.rows_per_split);
  const T* xs[NI]; int xld[NI], xc0[NI];   // a 64-channel tile lies in ONE source (C1 % 64 == 0 whenever C2 > 0)
#pragma unroll
  for (int ti = 0; ti < NI; ti++) {
    const int c = ci0 + ti * 64;
    if (c < g.C1 || g.C2 == 0) { xs[ti] = (const T*)g.x1; xld[ti] = g.C1; xc0[ti] = c; } else { xs[ti] = (const T*)g.x2; xld[ti] = g.C2; xc0[ti] = c - g.C1; }
  }
  const T* ys = (const T*)g.dy;

  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int wr = wv >> 1, wc = wv & 1;        // wave tile inside every 64 x 64 sub-tile: couts wr*32..+31, cins wc*32..+31

  uint4 rx[NI][2], ry[NO][2];
  auto fetch = [&](int m0) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = t + 256 * i;             // ST * SPR = 512 slots per tile
      const int row = idx / SPR, sl = idx - row * SPR;
      const int m = m0 + row;
      const bool mv = m < mend;
      const int64_t mc = mv ? m : mbeg;        // unconditional loads from a clamped row, zero selected afterwards
#pragma unroll
      for (int ti = 0; ti < NI; ti++) {
        const bool ok = mv && xc0[ti] + sl * VE < xld[ti];
        const uint4 v = *reinterpret_cast<const uint4*>(xs[ti] + mc * xld[ti] + (ok ? xc0[ti] + sl * VE : 0));
        rx[ti][i] = ok ? v : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int to = 0; to < NO; to++) {
        const bool ok = mv && co0 + to * 64 + sl * VE < g.Cout;
        const uint4 v = *reinterpret_cast<const uint4*>(ys + mc * g.Cout + (ok ? co0 + to * 64 + sl * VE : 0));
        ry[to][i] = ok ? v : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto stash = [&](int buf) RD_INLINE_LAMBDA {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = t + 256 * i;
      const int row = idx / SPR, sl = idx - row * SPR;
#pragma unroll
      for (int ti = 0; ti < NI; ti++) sT[buf][ti][row * RS + sl] = rx[ti][i];
#pragma unroll
      for (int to = 0; to < NO; to++) sT[buf][NI + to][row * RS + sl] = ry[to][i];
    }
  };

  f32x4 acc[NO][NI][2][2];
#pragma unroll
  for (int to = 0; to < NO; to++)
#pragma unroll
    for (int ti = 0; ti < NI; ti++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[to][ti][i][j] = f32x4{0, 0, 0, 0};

  const int nst = mend > mbeg ? (mend - mbeg + ST - 1) / ST : 0;
  if (nst > 0) fetch(mbeg);
  for (int s = 0; s < nst; s++) {
    const int buf = s & 1;
    stash(buf);
    __syncthreads();
    if (s + 1 < nst) fetch(mbeg + (s + 1) * ST);
    if (BF) {
      // transpose-read roles: this lane supplies token (fg*8 + (fr>>2)) (+4), channels 4*(fr&3)..+3 of the 16-channel tile being read
      const int lo_ = (fg * 8 + (fr >> 2)) * 64 + (fr & 3) * 4;
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        s16x8 ya[NO][2], xb[NI][2];
#pragma unroll
        for (int to = 0; to < NO; to++)
#pragma unroll
          for (int i = 0; i < 2; i++) {
            const unsigned short* by = reinterpret_cast<const unsigned short*>(&sT[buf][NI + to][0]) + lo_ + ks * 32 * 64 + (wr * 2 + i) * 16;
            uint2 lo = lds_read_tr16_b64(by), hi = lds_read_tr16_b64(by + 4 * 64);
            uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
            __builtin_memcpy(&ya[to][i], &v, 16);
          }
#pragma unroll
        for (int ti = 0; ti < NI; ti++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const unsigned short* bx = reinterpret_cast<const unsigned short*>(&sT[buf][ti][0]) + lo_ + ks * 32 * 64 + (wc * 2 + j) * 16;
            uint2 lo = lds_read_tr16_b64(bx), hi = lds_read_tr16_b64(bx + 4 * 64);
            uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
            __builtin_memcpy(&xb[ti][j], &v, 16);
          }
#pragma unroll
        for (int to = 0; to < NO; to++)
#pragma unroll
          for (int ti = 0; ti < NI; ti++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
              for (int j = 0; j < 2; j++) acc[to][ti][i][j] = mfma_16x16x32_bf16(ya[to][i], xb[ti][j], acc[to][ti][i][j]);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < ST / 4; ks++) {
        const int row = (ks * 4 + fg) * RS * 4;
        float ya[NO][2], xb[NI][2];
#pragma unroll
        for (int to = 0; to < NO; to++)
#pragma unroll
          for (int i = 0; i < 2; i++) ya[to][i] = reinterpret_cast<const float*>(&sT[buf][NI + to][0])[row + (wr * 2 + i) * 16 + fr];
#pragma unroll
        for (int ti = 0; ti < NI; ti++)
#pragma unroll
          for (int j = 0; j < 2; j++) xb[ti][j] = reinterpret_cast<const float*>(&sT[buf][ti][0])[row + (wc * 2 + j) * 16 + fr];
#pragma unroll
        for (int to = 0; to < NO; to++)
#pragma unroll
          for (int ti = 0; ti < NI; ti++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
              for (int j = 0; j < 2; j++) acc[to][ti][i][j] = mfma_16x16x4_f32(ya[to][i], xb[ti][j], acc[to][ti][i][j]);
      }
    }
  }
  float* slab = g.slab + (int64_t)sp * g.Cout * Cin;
#pragma unroll
  for (int to = 0; to < NO; to++)
#pragma unroll
    for (int ti = 0; ti < NI; ti++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int ci = ci0 + ti * 64 + (wc * 2 + j) * 16 + fr;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int co = co0 + to * 64 + (wr * 2 + i) * 16 + fg * 4 + r;
            if (co < g.Cout && ci < Cin) slab[(int64_t)co * Cin + ci] = acc[to][ti][i][j][r];
          }
        }
}

template <typename T>
__global__ __launch_bounds__(256) void linear_wgrad_kernel(LwgBatch b) {
  __shared__ uint4 sT[2][4][(sizeof(T) == 2 ? 64 * 8 : 32 * 17)];      // [buffer][tile: NI input tiles, then NO output tiles][token row][slot]
  const LwgGemm& g = b.it[blockIdx.y];
  int ni, no;
  lwg_shape(g.C1 + g.C2, g.Cout, ni, no);
  if (ni == 1 && no == 1) lwg_body<T, 1, 1>(g, sT);
  else if (ni == 1 && no == 2) lwg_body<T, 1, 2>(g, sT);
  else if (ni == 1) lwg_body<T, 1, 3>(g, sT);
  else if (no == 1 && ni == 2) lwg_body<T, 2, 1>(g, sT);
  else if (no == 1) lwg_body<T, 3, 1>(g, sT);
  else lwg_body<T, 2, 2>(g, sT);
}

// dw[e] (+)= sum of the item's slabs, fixed order (deterministic).  thread = (float4 column ol, slab lane sl): lane sl adds slabs sl, sl+SL, ...
// four at a time (independent 16-byte loads in flight), the SL lane sums are then combined by an LDS tree.  The first form -- one thread per
// element walking all its slabs in a scalar loop, at most 64 blocks per item -- took 110 us for a 4-slab 2688 x 128 gradient and for
// many-slab small ones alike: a chain of dependent memory round trips.
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(LwgRedBatch b) {
  __shared__ float4 red[256];
  const LwgReduce& r = b.it[blockIdx.y];
  const int t = threadIdx.x;
  if ((r.elems & 3) || ((reinterpret_cast<uintptr_t>(r.slab) | reinterpret_cast<uintptr_t>(r.dw)) & 15)) {      // not vectorisable: scalar form
    for (int64_t e = (int64_t)blockIdx.x * 256 + t; e < r.elems; e += (int64_t)gridDim.x * 256) {
      float s = 0.f;
      for (int sp = 0; sp < r.nsplit; sp++) s += r.slab[(int64_t)sp * r.elems + e];
      r.dw[e] = r.accumulate ? r.dw[e] + s : s;
    }
    return;
  }
  int SL = 1; while (SL * 8 < r.nsplit && SL < 64) SL <<= 1;      // about 8 slabs per slab lane
  const int OB = 256 / SL, ol = t % OB, sl = t / OB;
  const int64_t n4 = r.elems >> 2;
  const float4* s4 = reinterpret_cast<const float4*>(r.slab);
  float4* d4 = reinterpret_cast<float4*>(r.dw);
  for (int64_t base = (int64_t)blockIdx.x * OB; base < n4; base += (int64_t)gridDim.x * OB) {
    const int64_t i4 = base + ol;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i4 < n4) {
      int sp = sl;
      for (; sp + 3 * SL < r.nsplit; sp += 4 * SL) {
        const float4 v0 = s4[(int64_t)sp * n4 + i4], v1 = s4[(int64_t)(sp + SL) * n4 + i4];
        const float4 v2 = s4[(int64_t)(sp + 2 * SL) * n4 + i4], v3 = s4[(int64_t)(sp + 3 * SL) * n4 + i4];
        s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
        s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
      }
      for (; sp < r.nsplit; sp += SL) { const float4 v = s4[(int64_t)sp * n4 + i4]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    }
    red[t] = s;
    __syncthreads();
    for (int h = SL >> 1; h > 0; h >>= 1) {
      if (sl < h) { const float4 o = red[t + h * OB]; float4 m = red[t]; m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w; red[t] = m; }
      __syncthreads();
    }
    if (sl == 0 && i4 < n4) {
      float4 m = red[t];
      if (r.accumulate) { const float4 o = d4[i4]; m.x += o.x; m.y += o.y; m.z += o.z; m.w += o.w; }
      d4[i4] = m;
    }
    __syncthreads();
  }
}

void launch_linear_wgrad_batch(const LwgGemm* gemms, int n_gemm, const LwgReduce* reds, int n_red, int dtype, hipStream_t st) {
  for (int base = 0; base < n_gemm; base += LWG_MAX_ITEMS) {
    const int n = std::min(LWG_MAX_ITEMS, n_gemm - base);
    LwgBatch b;
    int maxb = 1;
    for (int i = 0; i < n; i++) {
      b.it[i] = gemms[base + i];
      const LwgGemm& g = b.it[i];
      maxb = std::max(maxb, lwg_blocks_per_split(g.C1 + g.C2, g.Cout) * g.nsplit);
    }
    for (int i = n; i < LWG_MAX_ITEMS; i++) b.it[i] = b.it[0];
    if (dtype == 0) hipLaunchKernelGGL((linear_wgrad_kernel<float>), dim3((unsigned)maxb, (unsigned)n), dim3(256), 0, st, b);
    else hipLaunchKernelGGL((linear_wgrad_kernel<bf16_t>), dim3((unsigned)maxb, (unsigned)n), dim3(256), 0, st, b);
  }
  for (int base = 0; base < n_red; base += LWG_MAX_REDS) {
    const int n = std::min(LWG_MAX_REDS, n_red - base);
    LwgRedBatch b;
    int64_t maxe = 1;
    for (int i = 0; i < n; i++) { b.it[i] = reds[base + i]; maxe = std::max(maxe, b.it[i].elems); }
    for (int i = n; i < LWG_MAX_REDS; i++) b.it[i] = b.it[0];
    unsigned gx = (unsigned)std::min<int64_t>(cdiv(maxe, 4 * 32), 256);      // >= 32 float4 columns per block (SL <= 8 lanes for typical splits)
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3(gx, (unsigned)n), dim3(256), 0, st, b);
  }
}

}  // namespace rd
