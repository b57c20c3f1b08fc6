// Linear-attention head routine shared by rd_attention.hip (stand-alone op) and rd_loftr.hip (fused LoFTR encoder layer).
#pragma once
#include "rd_common.h"
#include "rd_kernels.h"

namespace rd {

static constexpr int LD = 17;    // padded row pitch (floats) of the [rows][16] LDS tiles
static constexpr int MAXR = 32;  // max tokens per sequence handled by this kernel

// acc(16x16) += A(16xK) * B(Kx16); A(i,k) = a[i*ai + k*ak], B(k,j) = b[k*bk + j*bj]; K % 4 == 0.
// lane (r = lane&15, g = lane>>4) ends with D[row = g*4 + reg][col = r].
__device__ __forceinline__ f32x4 wave_mm(const float* a, int ai, int ak, const float* b, int bk, int bj, int K, f32x4 acc) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  for (int k0 = 0; k0 < K; k0 += 4) acc = mfma_16x16x4_f32(a[r * ai + (k0 + g) * ak], b[(k0 + g) * bk + r * bj], acc);
  return acc;
}

// the same with the contraction length fixed at compile time: fully unrolled, so that all LDS operand reads of the chain are issued before the
// first MFMA waits for one (the rolled loop was read -> wait -> MFMA per step of four: ~160 cycles per step).  Rows beyond the sequence are
// zero in every scratch tile, so contracting over all MAXR rows adds exact zeros: bit-identical to the rolled loop over the padded length.
template <int K>
__device__ __forceinline__ f32x4 wave_mm_k(const float* a, int ai, int ak, const float* b, int bk, int bj, f32x4 acc) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  float av[K / 4], bv[K / 4];
#pragma unroll
  for (int i = 0; i < K / 4; i++) { av[i] = a[r * ai + (4 * i + g) * ak]; bv[i] = b[(4 * i + g) * bk + r * bj]; }
#pragma unroll
  for (int i = 0; i < K / 4; i++) acc = mfma_16x16x4_f32(av[i], bv[i], acc);
  return acc;
}

struct AttnSmem {
  float q[MAXR * LD], k[MAXR * LD], v[MAXR * LD], d[MAXR * LD];
  float kv[16 * LD], dkv[16 * LD];
  float ksum[16], dksum[16], dden[MAXR], zinv[MAXR];
};

template <typename T>
__device__ __forceinline__ void stage_head(const T* __restrict__ src, int64_t row0, int ld, int col0, int rows, float* dst,
                                            int mode, float scale, bool active) {
  // mode 0: raw*scale, mode 1: elu(x)+1
  const int lane = threadIdx.x & 63;
  for (int idx = lane; idx < MAXR * 16; idx += 64) {
    int r = idx >> 4, c = idx & 15;
    float x = 0.f;
    if (active && r < rows) {
      x = Elem<T>::ld(src + (row0 + r) * ld + col0 + c);
      x = mode ? (x > 0.f ? x + 1.f : __expf(x)) : x * scale;
    }
    dst[r * LD + c] = x;
  }
}

// One wave, one (n, h) head.  `s` is this wave's PRIVATE LDS scratch; q/k/v/dout/out/dq/dk/dv are token matrices with leading
// dimensions ldq/ldk/ldv/ldo (global memory).  Phases are ordered by wave_sync() (no block barrier on the GPU; a block barrier under
// the host emulator): every wave of the block must call it the same number of times, and the caller owns any block-level ordering.
// after_loads() runs right after the staging loads are issued (a caller's prefetch for its next phase goes behind them).
// PRE (forward only): the caller has already written the feature-mapped q, k and the scaled v of this head into s.q / s.k / s.v (rows beyond
// L / S zero) and ordered them with a wave_sync(): no staging loads (rd_loftr.hip: the wave that owns head h is the wave whose projection
// GEMM tile IS head h, so its accumulators go straight into its scratch).
// PRED (backward only): s.d (the gradient of this head's attention output, rows beyond L zero) is already in place likewise.
template <typename T, bool BWD, typename Hook, bool PRE = false, bool PRED = false>
__device__ __forceinline__ void attn_head(AttnSmem& s, const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                          const T* __restrict__ dout, T* __restrict__ out, T* __restrict__ dq, T* __restrict__ dk,
                                          T* __restrict__ dv, int n, int h, bool active, int L, int S, int ldq, int ldk, int ldv,
                                          int ldo, float eps, Hook after_loads, T* lds_out = nullptr, int ldl = 0) {
  const int lane = threadIdx.x & 63;
  const int r = lane & 15, g = lane >> 4;
  const int col0 = h * 16;
  const float fS = (float)S;
  const int Lp = (L + 3) & ~3, Sp = (S + 3) & ~3;
  const int lt = (L + 15) >> 4, stl = (S + 15) >> 4;  // 16-row tiles

  if (PRE) { after_loads(); }
  else {
  if (!PRED) wave_sync();  // a previous call's reads of this wave's scratch are complete before it is restaged (back-to-back calls; PRED: the caller's)
  constexpr int VE = Elem<T>::VE;
  const bool vec = L > 0 && S > 0 && (ldq % VE == 0) && (ldk % VE == 0) && (ldv % VE == 0) && (!BWD || ldo % VE == 0) &&
                   (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (BWD ? (uintptr_t)dout : (uintptr_t)0)) & 15) == 0;
  if (vec) {
    // 16-byte loads, all tensors' requests in flight together (the element-wise path below is one 2-byte round trip after another:
    // it was the whole cost of the attention phase inside the fused LoFTR layer)
    constexpr int SPR = 16 / VE, NIT = MAXR * SPR / 64;   // slots per row, slots per lane (1 for bf16, 2 for fp32)
    float xq[NIT][VE], xk[NIT][VE], xv[NIT][VE], xd[BWD ? NIT : 1][VE];
    const int64_t nn = active ? n : 0;     // inactive waves read sequence 0 / head 0 and discard: no branch around the loads
    const int cc0 = active ? col0 : 0;     // (a load under a condition is waited for at the join, one round trip per tensor)
#pragma unroll
    for (int i = 0; i < NIT; i++) {
      const int idx = lane + 64 * i, rr = idx / SPR, c = (idx - rr * SPR) * VE;
      const int rl = min(rr, L - 1), rs = min(rr, S - 1);
#if defined(RD_LOFTR_PROBE) && RD_LOFTR_PROBE == 3   // timing probe (tools/ab_loftr.sh): no staging loads, wrong results
#pragma unroll
      for (int e = 0; e < VE; e++) { xq[i][e] = 0.01f * (float)(lane + e); xk[i][e] = 0.02f * (float)(rl + e); xv[i][e] = 0.03f * (float)(rs + c); if (BWD) xd[i][e] = 0.01f * (float)(cc0 + e); }
#else
      rd::ldv(q + (nn * L + rl) * ldq + cc0 + c, xq[i]);
      rd::ldv(k + (nn * S + rs) * ldk + cc0 + c, xk[i]);
      rd::ldv(v + (nn * S + rs) * ldv + cc0 + c, xv[i]);
      if (BWD && !PRED) rd::ldv(dout + (nn * L + rl) * ldo + cc0 + c, xd[i]);
#endif
    }
    sched_fence();
    after_loads();
#pragma unroll
    for (int i = 0; i < NIT; i++) {
      const int idx = lane + 64 * i, rr = idx / SPR, c = (idx - rr * SPR) * VE;
      const bool okq = active && rr < L, okk = active && rr < S;
#pragma unroll
      for (int e = 0; e < VE; e++) {
        const float a = xq[i][e], b = xk[i][e];
        s.q[rr * LD + c + e] = okq ? (a > 0.f ? a + 1.f : __expf(a)) : 0.f;
        s.k[rr * LD + c + e] = okk ? (b > 0.f ? b + 1.f : __expf(b)) : 0.f;
        s.v[rr * LD + c + e] = okk ? xv[i][e] * (1.f / fS) : 0.f;
        if (BWD && !PRED) s.d[rr * LD + c + e] = okq ? xd[i][e] * 1.f : 0.f;
      }
    }
  } else {
    stage_head<T>(q, (int64_t)n * L, ldq, col0, L, s.q, 1, 1.f, active);
    stage_head<T>(k, (int64_t)n * S, ldk, col0, S, s.k, 1, 1.f, active);
    stage_head<T>(v, (int64_t)n * S, ldv, col0, S, s.v, 0, 1.f / fS, active);
    if (BWD && !PRED) stage_head<T>(dout, (int64_t)n * L, ldo, col0, L, s.d, 0, 1.f, active);
    after_loads();
  }
  wave_sync();
  }

  // KV = K^T V  (16 x 16), Ksum
  {
    f32x4 acc = f32x4{0, 0, 0, 0};
    acc = wave_mm_k<MAXR>(s.k, 1, LD, s.v, LD, 1, acc);
#pragma unroll
    for (int e = 0; e < 4; e++) s.kv[(g * 4 + e) * LD + r] = acc[e];
    if (lane < 16) {      // (rows >= S are zero: the sum over all MAXR rows in ascending order is the sum over S; unrolled, the reads are in flight together)
      float kx[MAXR];
#pragma unroll
      for (int ss = 0; ss < MAXR; ss++) kx[ss] = s.k[ss * LD + lane];
      float t = 0.f;
#pragma unroll
      for (int ss = 0; ss < MAXR; ss++) t += kx[ss];
      s.ksum[lane] = t;
    }
  }
  wave_sync();

  // normaliser, one token per lane (dd ascending from eps), then P = Q KV
  if (lane < MAXR) {
    float den = eps;
    for (int dd = 0; dd < 16; dd++) den += s.q[lane * LD + dd] * s.ksum[dd];
    s.zinv[lane] = 1.f / den;
  }
  wave_sync();
  f32x4 P[2]; float Z[2][4];
#pragma unroll
  for (int tI = 0; tI < 2; tI++) {
    P[tI] = f32x4{0, 0, 0, 0};
    if (tI < lt) P[tI] = wave_mm_k<16>(s.q + tI * 16 * LD, LD, 1, s.kv, LD, 1, P[tI]);
#pragma unroll
    for (int e = 0; e < 4; e++) Z[tI][e] = s.zinv[tI * 16 + g * 4 + e];
  }

  if (!BWD) {
#pragma unroll
    for (int tI = 0; tI < 2; tI++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        int l = tI * 16 + g * 4 + e;
        const float o = P[tI][e] * Z[tI][e] * fS;
        if (active && l < L) Elem<T>::st(out + ((int64_t)n * L + l) * ldo + col0 + r, o);
        // lds_out: the caller's [MAXR][ldl] token tile of the attention output (rows beyond L zero), so that it need not read `out` back
        if (lds_out) Elem<T>::st(lds_out + l * ldl + col0 + r, (active && l < L) ? o : 0.f);
      }
    return;
  }

  // ---- backward ------------------------------------------------------------------------------------------
  // dP = S*Z*dO (in place over s.d), dden = -Z^2 * S * sum_e P*dO
#pragma unroll
  for (int tI = 0; tI < 2; tI++)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      int l = tI * 16 + g * 4 + e;
      float dO = s.d[l * LD + r];
      float dz = P[tI][e] * dO;
      dz = row16_sum(dz);
      s.d[l * LD + r] = fS * Z[tI][e] * dO;
      if (r == 0) s.dden[l] = -Z[tI][e] * Z[tI][e] * fS * dz;
    }
  wave_sync();

  // dQ = dP KV^T + dden * Ksum ; dq = dQ * phi'(q)
#pragma unroll
  for (int tI = 0; tI < 2; tI++) {
    if (tI < lt) {
      f32x4 a = f32x4{0, 0, 0, 0};
      a = wave_mm_k<16>(s.d + tI * 16 * LD, LD, 1, s.kv, 1, LD, a);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        int l = tI * 16 + g * 4 + e;
        float Q = s.q[l * LD + r];
        float gq = (a[e] + s.dden[l] * s.ksum[r]) * (Q > 1.f ? 1.f : Q);
        if (active && l < L) Elem<T>::st(dq + ((int64_t)n * L + l) * ldq + col0 + r, gq);
      }
    }
  }
  // dKV = Q^T dP (16 x 16), dKsum = sum_l dden[l] Q[l,:]
  {
    f32x4 a = f32x4{0, 0, 0, 0};
    a = wave_mm_k<MAXR>(s.q, 1, LD, s.d, LD, 1, a);
#pragma unroll
    for (int e = 0; e < 4; e++) s.dkv[(g * 4 + e) * LD + r] = a[e];
    if (lane < 16) {      // (q rows >= L are zero: their products are exact zeros)
      float dx[MAXR], qx[MAXR];
#pragma unroll
      for (int l = 0; l < MAXR; l++) { dx[l] = s.dden[l]; qx[l] = s.q[l * LD + lane]; }
      float t = 0.f;
#pragma unroll
      for (int l = 0; l < MAXR; l++) t += (l < L ? dx[l] : 0.f) * qx[l];
      s.dksum[lane] = t;
    }
  }
  wave_sync();
  // dK = V' dKV^T + dKsum ; dV' = K dKV
#pragma unroll
  for (int tI = 0; tI < 2; tI++) {
    if (tI < stl) {
      f32x4 a = f32x4{0, 0, 0, 0}, b = f32x4{0, 0, 0, 0};
      a = wave_mm_k<16>(s.v + tI * 16 * LD, LD, 1, s.dkv, 1, LD, a);
      b = wave_mm_k<16>(s.k + tI * 16 * LD, LD, 1, s.dkv, LD, 1, b);
#pragma unroll
      for (int e = 0; e < 4; e++) {
        int ss = tI * 16 + g * 4 + e;
        float K = s.k[ss * LD + r];
        if (active && ss < S) {
          Elem<T>::st(dk + ((int64_t)n * S + ss) * ldk + col0 + r, (a[e] + s.dksum[r]) * (K > 1.f ? 1.f : K));
          Elem<T>::st(dv + ((int64_t)n * S + ss) * ldv + col0 + r, b[e] / fS);
        }
      }
    }
  }
}

}  // namespace rd
